"""BASELINE.json configs[2..4] at their stated tree shapes, CPU side: our host layer (sequential and speculative-frontier
forms) on the tiny CPU oracle pipe must reproduce the UNCHANGED reference's runs frozen in tests/golden/configs.json -
identical plan / fractions / injection indices / census, similarities and frames within CPU summation-order noise."""
import numpy as np
import pytest

from oracle import pipe as OP
from oracle import sdxl_ref as R

from _baseline_cfgs import (check_structure, check_structure_cfg4_plain, check_values, gold_configs, lpips_base, setup_cfg3, setup_cfg4,
                            setup_cfg5, spread_metric)

CPU_TOL = dict(sim_rtol=2e-3, norm_rtol=2e-3, mean_tol=0.25, head_tol=2, ds_tol=0.5)


@pytest.fixture()
def cpu_backend():
    from latentblending_amd.backend import set_backend
    set_backend(R.TorchCpuBackend())
    yield
    set_backend(None)


def tiny_pipe(turbo):
    return OP.StableDiffusionXLPipeline(turbo=turbo, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())


def test_cfg3_stated_tree_matches_reference(cpu_backend):
    from latentblending_amd import BlendingEngine
    c = gold_configs()["cfg3"]
    p = tiny_pipe(False)
    np.random.seed(0)
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False)
    setup_cfg3(be)
    p.noise.reset()
    p.unet.calls = p.vae.calls = 0
    imgs = be.run_transition(fixed_seeds=[420, 421])
    assert c["list_idx_injection"] == [15, 18, 21, 24, 27] and c["list_nmb_stems"] == [4, 3, 3, 2, 1]     # 
    assert (p.unet.calls, p.vae.calls) == (c["unet_calls"], c["vae_calls"]) == (198, 15)
    check_structure(be, imgs, c)
    assert [[x is None for x in l] for l in be.tree_latents] == c["none_pattern"]
    check_values(be, imgs, c, **CPU_TOL)


@pytest.mark.parametrize("frontier", [1, 16])
def test_cfg4_stated_tree_matches_reference(frontier, cpu_backend):
    from latentblending_amd import BlendingEngine
    c = gold_configs()["cfg4"]
    p = tiny_pipe(True)
    np.random.seed(0)
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=frontier)
    setup_cfg4(be)
    p.noise.reset()
    p.unet.calls = p.vae.calls = 0
    imgs = be.run_transition(fixed_seeds=[420, 421])
    assert c["frames"] == 66 and c["list_nmb_stems"] == [64]
    if frontier == 1:
        assert (p.unet.calls, p.vae.calls, p.noise.draws) == (c["unet_calls"], c["vae_calls"], c["noise_draws"]) == (136, 66, c["noise_draws"])
    if frontier == 1:
        check_structure(be, imgs, c)
        check_values(be, imgs, c, **CPU_TOL)
    else:                   # (a speculative frontier draws ancestral noise in evaluation order: other samples, same structure)
        check_structure_cfg4_plain(be, imgs, c)


@pytest.mark.parametrize("frontier", [1, 16])
def test_cfg4_spread_metric_tree_is_the_references(frontier, cpu_backend):
    """cfg 4 under the metric with spread (fixture cfg4_skew: every greedy choice >= 5 % clear of the runner-up): the
    sequential engine reproduces the reference's tree, commit order, noise draws and frames; the speculative frontier commits
    the SAME tree (its ancestral noise is drawn in evaluation order, so its samples are others: structure only)."""
    from latentblending_amd import BlendingEngine
    c = gold_configs()["cfg4_skew"]
    assert c["min_separation"] >= 1.05
    p = tiny_pipe(True)
    np.random.seed(0)
    lp = R.OracleLPIPS(7)
    be = BlendingEngine(p, metric=lp, verbose=False, frontier_width=frontier)
    be.pair_metric = spread_metric(c, lpips_base(lp))
    setup_cfg4(be)
    p.noise.reset()
    imgs = be.run_transition(fixed_seeds=[420, 421])
    check_structure(be, imgs, c)
    if frontier == 1:
        assert p.noise.draws == c["noise_draws"]
        check_values(be, imgs, c, **CPU_TOL)


def test_cfg5_chain_first_segments_match_reference(cpu_backend):
    """example_multi_trans.py:39-58 through replay.run_multi_transition: the first two of the five chained transitions
    (the GPU test runs all five; a CPU run of all 870 UNet calls would take minutes)."""
    from latentblending_amd import BlendingEngine
    from latentblending_amd.replay import run_multi_transition
    g = gold_configs()["cfg5"]
    p = tiny_pipe(False)
    np.random.seed(0)
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False)
    setup_cfg5(be, g["negative_prompt"])
    p.noise.reset()
    seen = []

    def on_segment(i, frames):
        c = g["segments"][i]
        check_structure(be, frames, c)
        check_values(be, frames, c, **CPU_TOL)
        seen.append(i)

    p.unet.calls = 0
    run_multi_transition(be, g["prompts"][:3], g["seeds"][:3], fp_movie=None, on_segment=on_segment)
    assert seen == [0, 1]
    assert p.unet.calls == g["segments"][0]["unet_calls"] + g["segments"][1]["unet_calls"] == 198 + 168
