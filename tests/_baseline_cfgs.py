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


def box16(img):
    """16 x 16 box-downsample of a frame's channel mean (the twin of oracle/make_golden.py::box16)."""
    a = np.asarray(img).astype(np.float64).mean(axis=2)
    h, w = a.shape
    return a.reshape(16, h // 16, 16, w // 16).mean(axis=(1, 3)).flatten()


def check_values(be, imgs, c, *, sim_rtol, norm_rtol, mean_tol, head_tol, ds_tol):
    """Numeric agreement with the reference run (CPU fp32 oracle arithmetic): similarities, latent norms, and EVERY frame
    as a whole - its mean, its first pixels, and all 256 cells of its 16 x 16 box-downsample (a frame that is wrong anywhere
    moves the cell it is wrong in)."""
    sims = np.array([float(s) for s in be.tree_similarities])
    assert np.allclose(sims, c["tree_similarities"], rtol=sim_rtol), (sims.tolist(), c["tree_similarities"])
    for lat, norm in zip(be.tree_latents, c["final_latent_norm"]):
        assert abs(float(lat[-1].float().norm()) - norm) <= norm_rtol * norm
    assert len(c["frame_ds16"]) == len(imgs)
    for k, (img, mean, head, ds) in enumerate(zip(imgs, c["frame_mean"], c["frame_head"], c["frame_ds16"])):
        a = np.asarray(img)
        assert abs(float(a.mean()) - mean) <= mean_tol, (k, float(a.mean()), mean)
        assert np.abs(a.flatten()[:24].astype(int) - np.array(head)).max() <= head_tol
        worst = float(np.abs(box16(img) - np.array(ds)).max())
        assert worst <= ds_tol, (k, worst)


def spread_metric(c, base):
    """The pair_metric of the cfg4_skew fixture: base(frame_a, frame_b) x |fb - fa|^width_power x exp(skew x mean position)
    (oracle/make_golden.py::position_skewed_engine wraps the reference's distance with the same factor)."""
    from oracle.make_golden import spread_weight
    return lambda a, b, fa, fb: base(a, b) * spread_weight(fa, fb, c["skew"], c["width_power"])


def lpips_base(metric):
    """frame pair -> the distance `metric` (an LPIPS callable on [-1, 1] NCHW tensors) gives, as the reference computes it
    (/root/reference/latentblending/blending_engine.py:745-758)."""
    import torch

    def to_tensor(img):
        t = torch.from_numpy(np.array(img)).float()
        return (2 * t / 255.0 - 1).permute([2, 0, 1]).unsqueeze(0)
    return lambda a, b: float(metric(to_tensor(a), to_tensor(b))[0][0][0][0])


def check_structure_cfg4_plain(be, imgs, c):
    """cfg 4 under PLAIN LPIPS (fixture "cfg4"): what is structural whatever the arithmetic.  63 of the 64 mid branches fill
    the 1/64 grid completely, the 64th halves one of those gaps; census and injection indices are the reference's.  (WHICH gap
    the 64th halves is decided among 64 distances within 0.5 % of each other on the tiny synthetic model; the identical-tree
    assertion lives on the "cfg4_skew" fixture, whose every greedy choice is >= 5 % clear.)"""
    assert len(imgs) == c["frames"] == 66
    assert [int(i) for i in be.list_idx_injection] == c["list_idx_injection"] and [int(s) for s in be.list_nmb_stems] == c["list_nmb_stems"]
    fr = [float(f) for f in be.tree_fracts]
    grid = [k / 64 for k in range(65)]
    assert all(g in fr for g in grid), fr
    extra = [f for f in fr if f not in grid]
    assert len(extra) == 1 and (extra[0] * 128) % 2 == 1, extra
    assert [int(i) for i in be.tree_idx_injection] == c["tree_idx_injection"]


def gold_guidance_chain():
    with open(os.path.join(ROOT, "tests", "golden", "guidance_chain.json")) as fh:
        return json.load(fh)


def run_guidance_chain(be, reset_counters=None):
    """The calls of oracle/make_golden.py::guidance_chain_fixture on engine `be`: SDXL base, 6 steps, guidance 4.0, one level
    idx 3 x 6 stems, two chained transitions (swap_forward + recycle_img1).  Returns [(frames, guidance left behind, holder's
    guidance, state-dict guidance)] per transition.  The reference's loop leaves the dampened scale of the LAST COMMITTED
    branch (/root/reference/latentblending/blending_engine.py:155-164, 358-362) and denoises the next new anchor under it
    (:370-423)."""
    g = gold_guidance_chain()
    be.set_dimensions((128, 128))
    be.set_num_inference_steps(g["steps"])
    be.set_guidance_scale(g["guidance"])
    be.list_idx_injection, be.list_nmb_stems = list(g["list_idx_injection"]), list(g["list_nmb_stems"])
    out = []
    for i in range(2):
        if i == 0:
            be.set_prompt1(g["prompts"][0])
            be.set_prompt2(g["prompts"][1])
        else:
            be.swap_forward()
            be.set_prompt2(g["prompts"][i + 1])
        if reset_counters is not None:
            reset_counters()
        imgs = be.run_transition(recycle_img1=i > 0, fixed_seeds=g["seeds"][i:i + 2])
        out.append((list(imgs), float(be.guidance_scale), float(be.dh.guidance_scale), float(be.get_state_dict()["guidance_scale"]),
                    [float(f) for f in be.tree_fracts], [float(s) for s in be.tree_similarities],
                    [float(l[-1].float().norm()) for l in be.tree_latents]))
    return out


def check_guidance_chain(be, runs, *, sim_rtol, norm_rtol, mean_tol, head_tol, ds_tol):
    g = gold_guidance_chain()
    for seg, (imgs, gs, gs_holder, gs_state, fracts, sims, norms) in zip(g["segments"], runs):
        assert gs == gs_holder == gs_state == seg["guidance_scale_left_behind"], (gs, gs_holder, gs_state, seg["guidance_scale_left_behind"])
        assert fracts == seg["tree_fracts"], (fracts, seg["tree_fracts"])
        assert np.allclose(sims, seg["tree_similarities"], rtol=sim_rtol), (sims, seg["tree_similarities"])
        for n, ref in zip(norms, seg["final_latent_norm"]):
            assert abs(n - ref) <= norm_rtol * ref, (n, ref)
        assert len(imgs) == seg["frames"]
        for k, (img, mean, head, ds) in enumerate(zip(imgs, seg["frame_mean"], seg["frame_head"], seg["frame_ds16"])):
            a = np.asarray(img)
            assert abs(float(a.mean()) - mean) <= mean_tol, (k, float(a.mean()), mean)
            assert np.abs(a.flatten()[:24].astype(int) - np.array(head)).max() <= head_tol
            assert float(np.abs(box16(img) - np.array(ds)).max()) <= ds_tol, k
