"""BASELINE.json configs[2..4] at their STATED tree shapes on a real MI355X (tiny model width): the native engine - HIP
kernels behind the C-ABI, hipGraph replay, speculative frontier / fused wavefront - against the UNCHANGED reference's
runs of the same calls on the CPU fp32 oracle pipe, frozen in tests/golden/configs.json (oracle/make_golden.py).

Identical: plan, fractions, injection indices, frame count.  Within the fp16 tolerance of SURVEY.md §8(d): similarities,
final latents, frames (mean |du8| of a frame's first pixels <= 4, frame means within 1 grey level)."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pipe as OP  # noqa: E402  (checker only: the noise tape and the tiny configs)
from oracle import sdxl_ref as R  # noqa: E402

from _baseline_cfgs import (box16, check_structure, check_structure_cfg4_plain, check_values, gold_configs, lpips_base, setup_cfg3,  # noqa: E402
                            setup_cfg4, setup_cfg5, spread_metric)

GPU_TOL = dict(sim_rtol=5e-2, norm_rtol=3e-2, mean_tol=1.0, head_tol=4, ds_tol=2.0)


def native_pipe(turbo):
    import latentblending_amd.native as n
    ucfg, vcfg = R.tiny_unet_cfg(), R.tiny_vae_cfg()
    p = n.NativeSDXLPipe(turbo=turbo, unet_cfg=n.UNetConfig(**dataclasses.asdict(ucfg)),
                         vae_cfg=n.VAEConfig(**dataclasses.asdict(vcfg)), seed=0)
    tape = OP.NoiseTape(12345)
    p.scheduler.noise_source = tape
    return p, tape


@pytest.mark.parametrize("frontier", [1, 16])
def test_cfg3_stated_tree_native(frontier, results_log):
    """SDXL base, 30 steps, guidance 4.0 (mid-damped), depth 0.5, 15 branches: five injection levels
    [15,18,21,24,27] x [4,3,3,2,1] -> multi-level parents (blending_engine.py:550-561 of the reference)."""
    from latentblending_amd import BlendingEngine
    c = gold_configs()["cfg3"]
    p, tape = native_pipe(False)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=frontier)
    setup_cfg3(be)
    tape.reset()
    imgs = be.run_transition(fixed_seeds=[420, 421])
    assert [int(i) for i in be.list_idx_injection] == [15, 18, 21, 24, 27] and [int(s) for s in be.list_nmb_stems] == [4, 3, 3, 2, 1]
    check_structure(be, imgs, c)
    check_values(be, imgs, c, **GPU_TOL)
    results_log[f"cfg3_stated_tree_frontier{frontier}"] = {"frames": len(imgs), "same_tree": True,
                                                           "sims": [float(s) for s in be.tree_similarities]}


@pytest.mark.parametrize("frontier", [1, 8])
def test_guidance_chain_native(frontier, results_log):
    """tests/golden/guidance_chain.json natively: two chained SDXL-base transitions (one level 3 x 6 stems, guidance 4.0); the scale
    left behind is the last COMMITTED branch's (3.25, reference blending_engine.py:155-164, 358-362) at every frontier width,
    and the second transition - its new anchor denoised under that leftover scale - is the reference's."""
    from latentblending_amd import BlendingEngine
    from _baseline_cfgs import check_guidance_chain, run_guidance_chain
    p, tape = native_pipe(False)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=frontier)
    tape.reset()
    runs = run_guidance_chain(be)
    assert [r[1] for r in runs] == [3.25, 3.25]
    check_guidance_chain(be, runs, **GPU_TOL)
    results_log[f"guidance_chain_frontier{frontier}"] = {"left_behind": [r[1] for r in runs]}


def native_spread_metric(pipe, c):
    return spread_metric(c, lambda a, b: pipe.native_frame_distances([(a, b)])[0])


def test_cfg4_stated_tree_native_sequential(results_log):
    """SDXL-Turbo, 4 steps, 64 branches on one level (66 frames), sequential engine, under the metric with spread (fixture
    cfg4_skew: the reference's tree logic, every greedy choice >= 5 % clear): IDENTICAL tree, commit order and noise draws,
    similarities / latents / every frame (16 x 16 downsample of all 66) within the fp16 tolerance."""
    from latentblending_amd import BlendingEngine
    c = gold_configs()["cfg4_skew"]
    p, tape = native_pipe(True)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=1)
    be.pair_metric = native_spread_metric(p, c)
    setup_cfg4(be)
    tape.reset()
    imgs = be.run_transition(fixed_seeds=[420, 421])
    check_structure(be, imgs, c)
    check_values(be, imgs, c, **GPU_TOL)
    results_log["cfg4_spread_metric_sequential"] = {"frames": len(imgs), "same_tree": True, "min_separation_of_fixture": c["min_separation"]}


def test_cfg4_stated_tree_native_frontier64_matches_oracle_engine(results_log):
    """The same config as ONE speculative frontier of 64 (fused wavefront + virtual gaps: the form an 8-GPU farm runs) against
    the ORACLE engine - this repo's host layer on the CPU fp32 oracle pipe - run at the SAME frontier width with the same
    noise tape (a batched frontier consumes the ancestral tape in evaluation order, so the sequential reference run is not the
    comparison): identical tree = the reference's (fixture), then every frame, final latent and similarity numerically."""
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    c = gold_configs()["cfg4_skew"]
    p, tape = native_pipe(True)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=64)
    be.pair_metric = native_spread_metric(p, c)
    setup_cfg4(be)
    tape.reset()
    imgs = be.run_transition(fixed_seeds=[420, 421])
    check_structure(be, imgs, c)                                  # the reference's tree
    op = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    lp = R.OracleLPIPS(7)
    set_backend(R.TorchCpuBackend())
    import os
    import torch
    threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 8))          # (tiny-width CPU oracle: hundreds of host threads only slow it down)
    try:
        np.random.seed(0)
        ob = BlendingEngine(op, metric=lp, verbose=False, frontier_width=64)
        ob.pair_metric = spread_metric(c, lpips_base(lp))
        setup_cfg4(ob)
        op.noise.reset()
        want = ob.run_transition(fixed_seeds=[420, 421])
    finally:
        set_backend(None)
        torch.set_num_threads(threads)
    assert [float(f) for f in ob.tree_fracts] == [float(f) for f in be.tree_fracts] == c["tree_fracts"]
    assert be.stats.get("frontier_rounds", 0) == ob.stats.get("frontier_rounds", 0)
    sims, osims = np.array([float(x) for x in be.tree_similarities]), np.array([float(x) for x in ob.tree_similarities])
    assert np.allclose(sims, osims, rtol=GPU_TOL["sim_rtol"]), (sims.tolist(), osims.tolist())
    worst_ds = worst_mean = 0.0
    for k, (a, b, la, lb) in enumerate(zip(imgs, want, be.tree_latents, ob.tree_latents)):
        worst_ds = max(worst_ds, float(np.abs(box16(a) - box16(b)).max()))
        worst_mean = max(worst_mean, abs(float(np.asarray(a).mean()) - float(np.asarray(b).mean())))
        n = float(lb[-1].float().norm())
        assert abs(float(la[-1].float().norm()) - n) <= GPU_TOL["norm_rtol"] * n, k
    assert worst_ds <= GPU_TOL["ds_tol"] and worst_mean <= GPU_TOL["mean_tol"], (worst_ds, worst_mean)
    results_log["cfg4_spread_metric_frontier64"] = {"frames": len(imgs), "same_tree_as_reference": True, "rounds": be.stats.get("frontier_rounds", 0),
                                                    "worst_ds16_cell": worst_ds, "worst_frame_mean": worst_mean}


@pytest.mark.parametrize("frontier", [1, 64])
def test_cfg4_plain_lpips_structure_native(frontier, results_log):
    """cfg 4 under plain LPIPS (fixture cfg4): the structural facts - grid filled, census, injection indices."""
    from latentblending_amd import BlendingEngine
    c = gold_configs()["cfg4"]
    p, tape = native_pipe(True)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=frontier)
    setup_cfg4(be)
    tape.reset()
    imgs = be.run_transition(fixed_seeds=[420, 421])
    check_structure_cfg4_plain(be, imgs, c)
    results_log[f"cfg4_plain_lpips_frontier{frontier}"] = {"frames": len(imgs), "rounds": be.stats.get("frontier_rounds", 0)}


@pytest.mark.parametrize("frontier", [1, 16])
def test_cfg5_six_prompt_chain_native(frontier, results_log):
    """example_multi_trans.py:39-58 with 6 prompts on the base model through replay.run_multi_transition: five chained
    transitions, swap_forward + recycle_img1, each with the five-level tree; every segment against the reference's run."""
    from latentblending_amd import BlendingEngine
    from latentblending_amd.replay import run_multi_transition
    g = gold_configs()["cfg5"]
    p, tape = native_pipe(False)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=frontier)
    setup_cfg5(be, g["negative_prompt"])
    tape.reset()
    seen = []

    def on_segment(i, frames):
        check_structure(be, frames, g["segments"][i])
        check_values(be, frames, g["segments"][i], **GPU_TOL)
        seen.append(len(frames))

    segs = run_multi_transition(be, g["prompts"], g["seeds"], fp_movie=None, on_segment=on_segment)
    assert seen == [15] * 5 and len(segs) == 5
    for a, b in zip(segs[:-1], segs[1:]):               # the recycled key frame IS the previous transition's last frame
        assert np.array_equal(np.asarray(a[-1]), np.asarray(b[0]))
    results_log[f"cfg5_chain_frontier{frontier}"] = {"segments": len(segs), "frames": seen, "same_trees": True}


def test_state_round_trip_and_sessions_native(tmp_path, results_log):
    """SURVEY.md §8f rank 4 on the native path: get_state_dict -> yml_save -> yml_load -> load_state_dict on a second engine
    over the same pipe reproduces run_transition bit for bit; two sessions sharing that engine do not see each other's state."""
    from latentblending_amd import BlendingEngine, SessionRouter, yml_load, yml_save
    p, tape = native_pipe(True)
    np.random.seed(0)
    be = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=8)
    be.set_dimensions((128, 128))
    be.set_parental_crossfeed(0.8, 0.5, 0.5)
    be.set_branching(nmb_max_branches=6)
    be.set_prompt1("photo of a reef")
    be.set_prompt2("rendering of an alien planet")
    be.seed1, be.seed2 = 420, 421
    fp = str(tmp_path / "state.yml")
    yml_save(fp, be.get_state_dict())
    tape.reset()
    want = [np.asarray(i).copy() for i in be.run_transition()]
    be2 = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=8)
    be2.load_state_dict(yml_load(fp))
    tape.reset()
    got = [np.asarray(i).copy() for i in be2.run_transition()]
    assert be2.tree_fracts == be.tree_fracts and all(np.array_equal(a, b) for a, b in zip(got, want))
    router = SessionRouter({"turbo": be})
    ua, ub = router.register_new_user("turbo", 128, 128), router.register_new_user("turbo", 128, 128)
    with router.session(ub).bound() as e:
        e.set_branching(nmb_max_branches=3)
        e.set_prompt1("fog"); e.set_prompt2("a harbour")
        tape.reset()
        other = e.run_transition(fixed_seeds=[1, 2])
    with router.session(ua).bound() as e:       # user A still has the engine's original settings and gets the original frames
        assert e.prompt1 == "photo of a reef" and int(e.list_nmb_stems[0]) == 6
        tape.reset()
        again = [np.asarray(i).copy() for i in e.run_transition()]
    assert len(other) == 5 and all(np.array_equal(a, b) for a, b in zip(again, want))
    results_log["state_round_trip_native"] = {"frames": len(got), "bit_identical": True}
