"""Shared set-up of BASELINE.json configs[2..4] at their STATED tree shapes on the tiny model width (tests/golden/configs.json
holds the unchanged reference's runs of exactly these calls: oracle/make_golden.py::configs_fixture)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gold_configs():
    with open(os.path.join(ROOT, "tests", "golden", "configs.json")) as fh:
        return json.load(fh)


def setup_cfg3(be):
    """SDXL base, 30 steps, guidance 4.0, depth 0.5, 15 branches -> levels [15,18,21,24,27] x [4,3,3,2,1]
    (/root/reference/latentblending/blending_engine.py:467-529)."""
    be.set_dimensions((128, 128))
    be.set_num_inference_steps(30)
    be.set_guidance_scale(4.0)
    be.set_branching(depth_strength=0.5, nmb_max_branches=15)
    be.set_prompt1("photo of a reef")
    be.set_prompt2("rendering of an alien planet")


def setup_cfg4(be):
    """SDXL-Turbo, 4 steps, 64 branches on one level."""
    be.set_dimensions((128, 128))
    be.set_branching(nmb_max_branches=64)
    be.set_prompt1("photo of a reef")
    be.set_prompt2("rendering of an alien planet")


def setup_cfg5(be, negative_prompt):
    """example_multi_trans.py:17-24 before its loop (base model, 30 steps, depth 0.5, 15 branches per transition)."""
    be.set_negative_prompt(negative_prompt)
    be.set_dimensions((128, 128))
    be.set_num_inference_steps(30)
    be.set_branching(depth_strength=0.5, nmb_max_branches=15)


def check_structure(be, imgs, c):
    """What must be IDENTICAL to the reference run: plan, census of frames, fractions, injection indices, None pattern."""
    assert len(imgs) == c["frames"]
    assert [int(i) for i in be.list_idx_injection] == c["list_idx_injection"]
    assert [int(s) for s in be.list_nmb_stems] == c["list_nmb_stems"]
    assert [float(f) for f in be.tree_fracts] == c["tree_fracts"], ([float(f) for f in be.tree_fracts], c["tree_fracts"])
    assert [int(i) for i in be.tree_idx_injection] == c["tree_idx_injection"]


def check_values(be, imgs, c, *, sim_rtol, norm_rtol, mean_tol, head_tol):
    """Numeric agreement with the reference run (CPU fp32 oracle arithmetic): similarities, latent norms, frames."""
    sims = np.array([float(s) for s in be.tree_similarities])
    assert np.allclose(sims, c["tree_similarities"], rtol=sim_rtol), (sims.tolist(), c["tree_similarities"])
    for lat, norm in zip(be.tree_latents, c["final_latent_norm"]):
        assert abs(float(lat[-1].float().norm()) - norm) <= norm_rtol * norm
    for img, mean, head in zip(imgs, c["frame_mean"], c["frame_head"]):
        a = np.asarray(img)
        assert abs(float(a.mean()) - mean) <= mean_tol, (float(a.mean()), mean)
        assert np.abs(a.flatten()[:24].astype(int) - np.array(head)).max() <= head_tol


def check_structure_cfg4_batched(be, imgs, c):
    """cfg 4 evaluated as ONE speculative batch: the ancestral sampler's noise tape is then consumed in evaluation order
    instead of the sequential engine's commit order, so every mid branch is a different (equally valid) sample and the
    near-tied choice of the LAST gap may differ from the sequential run.  What is structural stays: 63 of the 64 mid
    branches fill the 1/64 grid completely (any metric: a gap's child is its midpoint and the greedy order exhausts a level
    of the binary splitting before it can reach the next finer one only if all gaps of that level were taken - which the
    frame census of the reference run confirms), the 64th halves one of those gaps."""
    assert len(imgs) == c["frames"] == 66
    assert [int(i) for i in be.list_idx_injection] == c["list_idx_injection"] and [int(s) for s in be.list_nmb_stems] == c["list_nmb_stems"]
    fr = [float(f) for f in be.tree_fracts]
    grid = [k / 64 for k in range(65)]
    assert all(g in fr for g in grid), fr
    extra = [f for f in fr if f not in grid]
    assert len(extra) == 1 and (extra[0] * 128) % 2 == 1, extra
    ref_extra = [f for f in c["tree_fracts"] if f not in grid]
    assert len(ref_extra) == 1               # (the reference's sequential run has the same shape)
    assert [int(i) for i in be.tree_idx_injection] == c["tree_idx_injection"]


def check_cfg4_sequential(be, imgs, c, tol, rel_tie=0.01):
    """cfg 4, sequential engine, fp16 device arithmetic against the fp32 reference run.  The first 63 mid branches fill the 1/64 grid; the
    64th halves the gap with the LARGEST distance among 64 gaps whose reference distances lie within ~0.5 % of each other (tiny
    width, synthetic weights: 1.483e-5 .. 1.492e-5) - a choice below the noise floor of any fp16 pipeline.  Identical tree: full
    structural + numeric comparison.  Otherwise the device must have halved a gap that is a NEAR TIE of the reference's choice
    (its reference distance within `rel_tie` of the largest unsplit one), and every frame both trees hold must still agree
    numerically (same noise draws in the same order up to that last split).  Returns True when the tree was identical."""
    fr = [float(f) for f in be.tree_fracts]
    if fr == c["tree_fracts"]:
        check_structure(be, imgs, c)
        check_values(be, imgs, c, **tol)
        return True
    check_structure_cfg4_batched(be, imgs, c)
    grid = [k / 64 for k in range(65)]
    extra = [f for f in fr if f not in grid][0]
    lo = extra - 1 / 128
    ref_fr, ref_sims = c["tree_fracts"], c["tree_similarities"]
    unsplit = {ref_fr[i]: ref_sims[i] for i in range(len(ref_sims)) if abs(ref_fr[i + 1] - ref_fr[i] - 1 / 64) < 1e-12}
    assert lo in unsplit, (extra, "the reference split this very gap: the trees would be identical")
    assert unsplit[lo] >= (1.0 - rel_tie) * max(unsplit.values()), (extra, unsplit[lo], max(unsplit.values()))
    # frames on the shared grid (everything but the two 1/128 frames): same samples as the reference's
    for f, img, lat in zip(fr, imgs, be.tree_latents):
        if f not in grid:
            continue
        j = ref_fr.index(f)
        a = np.asarray(img)
        assert abs(float(a.mean()) - c["frame_mean"][j]) <= tol["mean_tol"], (f, float(a.mean()), c["frame_mean"][j])
        assert np.abs(a.flatten()[:24].astype(int) - np.array(c["frame_head"][j])).max() <= tol["head_tol"]
        assert abs(float(lat[-1].float().norm()) - c["final_latent_norm"][j]) <= tol["norm_rtol"] * c["final_latent_norm"][j]
    return False
