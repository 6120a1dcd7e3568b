"""ORACLE — test infrastructure.  Generates tests/golden/*.json by EXECUTING the unchanged reference
(/root/reference, imported through oracle/ref_harness.py) in this container.  The GPU box has no
/root/reference, so everything the tests need from it is frozen here.

    python -m oracle.make_golden

Fixtures
  planner.json    get_time_based_branching / set_branching / crossfeed coefficient vectors /
                  guidance mid-dampening / get_closest_idx produced by the reference's own methods
  slerp.json      interpolate_spherical / interpolate_linear of the reference on seeded small inputs
                  (fp16 results stored as int16 bit patterns -> bit-exact comparisons)
  scheduler.json  closed-form SDXL sigma / timestep known answers
  tree.json       a full run_transition of the reference BlendingEngine on the tiny CPU oracle pipe:
                  census, tree_fracts, tree_idx_injection, similarities, latent / frame checksums
  configs.json    BASELINE.json configs[2..4] at their STATED tree shape on the tiny CPU oracle pipe, run by the unchanged
                  reference: cfg 3 (base, 30 steps, depth 0.5, 15 branches, guidance 4.0 -> five injection levels), cfg 4
                  (turbo, 64 branches, one level), cfg 5 (base, 30 steps, 6 prompts chained with swap_forward +
                  recycle_img1 as in example_multi_trans.py:39-58); every frame also as a 16 x 16 box-downsample.
                  cfg4_skew (round 5): cfg 4 again under a metric WITH SPREAD - the reference's tree logic untouched, its
                  perceptual distance multiplied by |fb - fa|^5 x exp(4.4 x mean position of the two frames) (PositionSkewed
                  below) - because under plain LPIPS the tiny synthetic model gives 64 gap distances within 0.5 % of each
                  other and the 64th split is below the noise floor of any fp16 pipeline; here every greedy choice is clear of
                  the runner-up by the margin stored in "min_separation" (asserted >= 5 %)
  frames.json     add_frames_linear_interp of the reference (utils.py:105-178) on seeded uint8 key frames with a seeded
                  numpy RNG: per-gap insert counts and a checksum of every output frame (numpy 2.x arithmetic: the
                  float32 frames are blended in float64)

  guidance_chain.json  (round 6) two chained SDXL-base transitions, one level 3 x 6 stems: the mid-dampened guidance scale each
                  transition leaves behind (the LAST COMMITTED branch's) and the second transition computed under it

    python -m oracle.make_golden frames      # regenerate one fixture
"""
from __future__ import annotations

import hashlib
import json
import os
import re

import numpy as np
import torch

from . import pipe as OP
from . import ref_harness as H
from . import sdxl_ref as R

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def bits16(t: torch.Tensor):
    return t.detach().contiguous().view(torch.int16).flatten().tolist()


def sha(t) -> str:
    a = np.ascontiguousarray(np.asarray(t) if not isinstance(t, torch.Tensor) else t.detach().cpu().numpy())
    return hashlib.sha256(a.tobytes()).hexdigest()[:16]


def seeded(n, seed, dtype=torch.float16, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(n, generator=g) * scale).to(dtype)


def tiny_pipe(turbo=True):
    return OP.StableDiffusionXLPipeline(turbo=turbo, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())


def planner_fixture(ref):
    out = {"time_based": [], "turbo": [], "parental_coeffs": [], "anchor_coeffs": [], "guidance": [], "closest_idx": []}
    with H.cuda_is_identity():
        be = ref.BlendingEngine(tiny_pipe(turbo=False))
        for steps, depth, nmb in [(30, 0.5, 15), (30, 0.5, 64), (30, 0.5, 5), (30, 0.5, 3), (30, 0.2, 15), (50, 0.5, 15),
                                  (10, 0.2, 7), (6, 0.5, 6), (20, 0.35, 9)]:
            be.num_inference_steps = steps
            idx, stems = be.get_time_based_branching(depth, None, nmb)
            out["time_based"].append({"steps": steps, "depth": depth, "nmb": nmb, "idx": [int(i) for i in idx],
                                      "stems": [int(s) for s in stems]})
        for steps, depth, tmax, dtu, dtv in [(30, 0.5, 20, 0.05, 0.1), (30, 0.3, 5, 0.02, 0.2), (50, 0.5, 60, 0.1, 0.5)]:
            be.num_inference_steps, be.dt_unet_step, be.dt_vae = steps, dtu, dtv
            idx, stems = be.get_time_based_branching(depth, tmax, None)
            out["time_based"].append({"steps": steps, "depth": depth, "tmax": tmax, "dt_unet": dtu, "dt_vae": dtv,
                                      "idx": [int(i) for i in idx], "stems": [int(s) for s in stems]})
        # guidance dampening (blending_engine.py:155-164)
        for base, damper in [(4.0, 0.5), (7.5, 0.3), (0.0, 0.5)]:
            be.guidance_scale_base, be.guidance_scale_mid_damper = base, damper
            row = []
            for f in [0.0, 0.125, 0.25, 0.5, 0.75, 1.0]:
                be.set_guidance_mid_dampening(f)
                row.append(float(be.guidance_scale))
            out["guidance"].append({"base": base, "damper": damper, "fracts": [0.0, 0.125, 0.25, 0.5, 0.75, 1.0], "values": row})
        # get_closest_idx (docstring example + more)
        for fracts, q in [([0, 0.3, 0.6, 1.0], 0.4), ([0.0, 1.0], 0.5), ([0.0, 0.25, 0.5, 1.0], 0.125), ([0.0, 0.5, 1.0], 0.75)]:
            be.tree_fracts = list(fracts)
            a, b = be.get_closest_idx(q)
            out["closest_idx"].append({"fracts": fracts, "q": q, "result": [int(a), int(b)]})

        # crossfeed coefficient vectors as the reference builds them inside compute_latents_mix / compute_latents2:
        # capture the mixing_coeffs argument that reaches run_diffusion.
        def capture(engine, call):
            seen = {}
            original = engine.run_diffusion

            def spy(list_conditionings, latents_start=None, idx_start=0, list_latents_mixing=None, mixing_coeffs=0.0,
                    return_image=False):
                seen["coeffs"] = [float(c) for c in mixing_coeffs] if isinstance(mixing_coeffs, list) else mixing_coeffs
                raise StopIteration
            engine.run_diffusion = spy
            try:
                call()
            except StopIteration:
                pass
            engine.run_diffusion = original
            return seen["coeffs"]

        for turbo, steps, idx_inj, pw, rg, dc in [(True, 4, 2, 1.0, 1.0, 1.0), (True, 4, 1, 0.8, 0.5, 0.5), (True, 8, 3, 0.6, 0.75, 0.3),
                                                  (False, 30, 15, None, None, None), (False, 30, 24, None, None, None)]:
            e = ref.BlendingEngine(tiny_pipe(turbo=turbo))
            e.set_num_inference_steps(steps)
            if turbo:
                e.set_parental_crossfeed(pw, rg, dc)
            e.set_prompt1("a"); e.set_prompt2("b")
            z = e.get_noise(1)
            e.tree_latents = [[z] * steps, [z] * steps]
            e.tree_fracts = [0.0, 1.0]
            coeffs = capture(e, lambda: e.compute_latents_mix(0.5, 0, 1, idx_inj))
            out["parental_coeffs"].append({"turbo": turbo, "steps": steps, "idx_injection": idx_inj,
                                           "power": float(e.parental_crossfeed_power), "range": float(e.parental_crossfeed_range),
                                           "decay": float(e.parental_crossfeed_decay), "coeffs": coeffs})
        for steps, pw, rg, dc in [(4, 0.5, 0.5, 0.5), (30, 0.3, 0.6, 0.9), (10, 1.0, 1.0, 0.1)]:
            e = ref.BlendingEngine(tiny_pipe(turbo=True))
            e.set_num_inference_steps(steps)
            e.set_branch1_crossfeed(pw, rg, dc)
            e.set_prompt1("a"); e.set_prompt2("b")
            z = e.get_noise(1)
            e.tree_latents = [[z] * steps, None]
            coeffs = capture(e, e.compute_latents2)
            out["anchor_coeffs"].append({"steps": steps, "power": pw, "range": rg, "decay": dc, "coeffs": coeffs})
        # turbo set_branching (blending_engine.py:273-283)
        e = ref.BlendingEngine(tiny_pipe(turbo=True))
        for steps, depth, nmb in [(4, None, None), (4, None, 15), (4, 0.5, 3), (2, 0.5, 3), (8, 0.3, 64), (5, 0.5, 7)]:
            e.set_num_inference_steps(steps)
            e.set_branching(depth_strength=depth, nmb_max_branches=nmb)
            out["turbo"].append({"steps": steps, "depth": depth, "nmb": nmb, "idx": [int(i) for i in e.list_idx_injection],
                                 "stems": [int(s) for s in e.list_nmb_stems]})
    return out


def slerp_fixture(ref):
    cases = []
    U = ref.utils

    def add(name, p0, p1, f):
        r = U.interpolate_spherical(p0, p1, f)
        entry = {"name": name, "fract": f, "in_dtype": str(p0.dtype), "out_dtype": str(r.dtype), "n": p0.numel(),
                 "seed0": None, "nan": bool(torch.isnan(r).all())}
        if r.dtype == torch.float16:
            entry["out_bits"] = bits16(r)
        else:
            entry["out_f32"] = [float(x) for x in r.flatten().tolist()]
        return entry

    for n in (64, 257):
        p0, p1 = seeded(n, 100 + n, scale=3.0), seeded(n, 200 + n, scale=3.0)
        for f in (0.0, 0.25, 0.37, 0.5, 1.0):
            e = add(f"f16_n{n}", p0, p1, f)
            e.update(seed0=100 + n, seed1=200 + n, scale=3.0)
            cases.append(e)
    a = seeded(64, 7)
    for name, x, y, f in [("identical", a, a.clone(), 0.3), ("antipodal", a, -a, 0.5), ("zero_norm", torch.zeros(64, dtype=torch.float16), a, 0.5)]:
        e = add(name, x, y, f)
        e.update(seed0=7, seed1=None, scale=1.0)
        cases.append(e)
    for dt in (torch.float32, torch.float64):
        x, y = seeded(33, 11, dt), seeded(33, 12, dt)
        e = add(f"dtype_{str(dt).split('.')[-1]}", x, y, 0.41)
        e.update(seed0=11, seed1=12, scale=1.0)
        cases.append(e)
    lerps = []
    for f in (0.0, 0.125, 0.5, 0.7321, 1.0):
        x, y = seeded(96, 21), seeded(96, 22)
        lerps.append({"fract": f, "seed0": 21, "seed1": 22, "n": 96, "out_bits": bits16(U.interpolate_linear(x, y, f))})
    u8a = (np.arange(48, dtype=np.uint8).reshape(4, 4, 3) * 5) % 255
    u8b = (np.arange(48, dtype=np.uint8)[::-1].reshape(4, 4, 3) * 3) % 255
    lerps.append({"fract": 0.3, "uint8": True, "a": u8a.flatten().tolist(), "b": u8b.flatten().tolist(),
                  "out": U.interpolate_linear(u8a, u8b, 0.3).flatten().tolist()})
    return {"slerp": cases, "lerp": lerps}


def scheduler_fixture():
    return {
        "sigma_999": 14.614641, "trailing4_timesteps": [999, 749, 499, 249],
        "trailing4_sigmas": [14.61464, 4.08173, 1.61289, 0.69320, 0.0],
        "trailing4_ancestral": [[3.91930, 1.13999], [1.48163, 0.63733], [0.62591, 0.29793], [0.0, 0.0]],
        "leading30_first": 958, "leading30_last": 1, "leading30_sigma0": 11.47685, "leading30_init_noise_sigma": 11.52033,
        "source": "closed form from SDXL scaled_linear betas [0.00085, 0.012], 1000 steps (SURVEY.md §4)",
        # DDIM (round 5): abar_t = prod_{s<=t} (1 - beta_s) in float64, leading spacing with steps_offset 1
        "ddim_alphas_cumprod": {"0": 0.99915, "1": 0.9982960278384514, "34": 0.9678813931124362, "925": 0.0108601253829753,
                                "958": 0.007534772714533716, "999": 0.004660098513077238},
        "ddim_leading30_timesteps_head": [958, 925, 892, 859], "ddim_leading30_timesteps_tail": [67, 34, 1],
        "ddim_final_alpha_cumprod": 0.99915,
    }


def tree_fixture(ref):
    runs = []
    for turbo, cfgd in [(True, dict(nmb=5)), (True, dict(nmb=3, steps=2, depth=0.5)), (False, dict(nmb=6, steps=6, depth=0.5, gs=3.0))]:
        p = tiny_pipe(turbo=turbo)
        np.random.seed(0)
        with H.cuda_is_identity():
            be = ref.BlendingEngine(p)
            be.set_dimensions((128, 128))
            if "steps" in cfgd:
                be.set_num_inference_steps(cfgd["steps"])
            if "gs" in cfgd:
                be.set_guidance_scale(cfgd["gs"])
            be.set_branching(depth_strength=cfgd.get("depth"), nmb_max_branches=cfgd["nmb"])
            be.set_prompt1("photo of a reef")
            be.set_prompt2("rendering of an alien planet")
            p.noise.reset()
            p.unet.calls = p.vae.calls = 0
            imgs = be.run_transition(fixed_seeds=[420, 421])
        runs.append({
            "turbo": turbo, "config": cfgd, "frames": len(imgs), "unet_calls": p.unet.calls, "vae_calls": p.vae.calls,
            "noise_draws": p.noise.draws, "list_idx_injection": [int(i) for i in be.list_idx_injection],
            "list_nmb_stems": [int(s) for s in be.list_nmb_stems],
            "tree_fracts": [float(f) for f in be.tree_fracts], "tree_idx_injection": [int(i) for i in be.tree_idx_injection],
            "tree_similarities": [float(s) for s in be.tree_similarities],
            "final_latent_sha": [sha(l[-1]) for l in be.tree_latents], "frame_sha": [sha(i) for i in imgs],
            # numeric summaries (compared with a tolerance: CPU conv/GEMM summation order depends on
            # the thread count and ISA of the machine that runs the test)
            "final_latent_head": [[float(v) for v in l[-1].flatten()[:24].float()] for l in be.tree_latents],
            "final_latent_norm": [float(l[-1].float().norm()) for l in be.tree_latents],
            "frame_mean": [float(np.asarray(i).mean()) for i in imgs],
            "frame_head": [[int(v) for v in np.asarray(i).flatten()[:24]] for i in imgs],
            "none_pattern": [[x is None for x in l] for l in be.tree_latents],
        })
    return runs


def box16(img):
    """16 x 16 box-downsample of a frame's channel mean (0..255), rounded to 0.01 grey levels: 256 numbers per frame."""
    a = np.asarray(img).astype(np.float64).mean(axis=2)
    h, w = a.shape
    return [round(float(v), 2) for v in a.reshape(16, h // 16, 16, w // 16).mean(axis=(1, 3)).flatten()]


def spread_weight(fa, fb, skew, width_power):
    """The factor the spread metric multiplies a perceptual distance by: |fb - fa|^width_power x exp(skew x mean position).
    None (an anchor compared before the tree exists) counts as its end of the axis."""
    import math
    fa, fb = (0.0 if fa is None else float(fa)), (1.0 if fb is None else float(fb))
    return abs(fb - fa) ** width_power * math.exp(skew * 0.5 * (fa + fb))


def position_skewed_engine(ref, skew, width_power):
    """The reference's BlendingEngine with ONE quantity changed: the perceptual distance of two frames is multiplied by
    spread_weight(their positions on the transition axis).  get_mixing_parameters / compute_latents_mix / insert_into_tree /
    run_transition are the reference's own code; the subclass only remembers the fraction of the frame being inserted
    (insert_into_tree's argument) so that the wrapped distance can look positions up, and records how far the greedy argmax
    was clear of the runner-up at every choice.

    Why this shape.  On the tiny synthetic model plain LPIPS gives gaps of one width distances within 0.5 % of each other and
    a factor ~2.9 between widths.  exp(skew x position) alone separates neighbours of one width by exp(skew x width) but lets
    a narrow gap far right tie with a wide gap far left (measured: some choice within 0.3 % for every skew in 1.5..6); the
    width power makes a halving worth more than the whole position range (2.9 x 2^5 = 92 > e^4.4 = 81), so the greedy
    order is level by level, right to left, and every choice is >= 5 % clear."""

    class PositionSkewed(ref.BlendingEngine):
        _fract_new = None
        separations = None

        def _position(self, img):
            for f, im in zip(self.tree_fracts, self.tree_final_imgs):
                if im is img:
                    return float(f)
            return self._fract_new

        def get_lpips_similarity(self, imgA, imgB):
            d = super().get_lpips_similarity(imgA, imgB)
            return d * spread_weight(self._position(imgA), self._position(imgB), skew, width_power)

        def insert_into_tree(self, fract_mixing, idx_injection, list_latents):
            self._fract_new = fract_mixing
            try:
                return super().insert_into_tree(fract_mixing, idx_injection, list_latents)
            finally:
                self._fract_new = None

        def get_mixing_parameters(self, idx_injection):
            if self.separations is None:
                self.separations = []
            sims = [s for s in self.tree_similarities if isinstance(s, float)]
            if len(sims) >= 2:
                top = sorted(sims)[-2:]
                self.separations.append(top[1] / top[0])
            return super().get_mixing_parameters(idx_injection)

    return PositionSkewed


def snapshot(be, imgs, p):
    """What the tests compare of one finished run_transition."""
    return {
        "frame_ds16": [box16(i) for i in imgs],
        "frames": len(imgs), "unet_calls": p.unet.calls, "vae_calls": p.vae.calls, "noise_draws": p.noise.draws,
        "list_idx_injection": [int(i) for i in be.list_idx_injection], "list_nmb_stems": [int(s) for s in be.list_nmb_stems],
        "tree_fracts": [float(f) for f in be.tree_fracts], "tree_idx_injection": [int(i) for i in be.tree_idx_injection],
        "tree_similarities": [float(s) for s in be.tree_similarities],
        "final_latent_norm": [float(l[-1].float().norm()) for l in be.tree_latents],
        "final_latent_head": [[float(v) for v in l[-1].flatten()[:24].float()] for l in be.tree_latents],
        "frame_mean": [float(np.asarray(i).mean()) for i in imgs],
        "frame_head": [[int(v) for v in np.asarray(i).flatten()[:24]] for i in imgs],
        "none_pattern": [[x is None for x in l] for l in be.tree_latents],
    }


CFG5_PROMPTS = ["lake and forest", "alien desolate landscapes", "psychedelic skyscraper city", "a reef at dawn",
                "fog over a harbour", "desert under two moons"]
CFG5_SEEDS = [420, 421, 977, 12, 90001, 5]
CFG5_NEGATIVE = "blurry, pale, low-res, lofi"
CFG4_SKEW, CFG4_WIDTH_POWER = 4.4, 5.0


def configs_fixture(ref):
    """BASELINE.json configs[2..4] at their stated tree shapes (tiny width), run by the unchanged reference."""
    out = {}
    # cfg 3: SDXL base, 30 steps, guidance 4.0, depth 0.5, 15 branches (blending_engine.py:467-529 plans the levels)
    p = tiny_pipe(turbo=False)
    np.random.seed(0)
    with H.cuda_is_identity():
        be = ref.BlendingEngine(p)
        be.set_dimensions((128, 128))
        be.set_num_inference_steps(30)
        be.set_guidance_scale(4.0)
        be.set_branching(depth_strength=0.5, nmb_max_branches=15)
        be.set_prompt1("photo of a reef")
        be.set_prompt2("rendering of an alien planet")
        p.noise.reset()
        p.unet.calls = p.vae.calls = 0
        imgs = be.run_transition(fixed_seeds=[420, 421])
    out["cfg3"] = snapshot(be, imgs, p)
    # cfg 4: SDXL-Turbo, 4 steps, 64 branches on one level
    p = tiny_pipe(turbo=True)
    np.random.seed(0)
    with H.cuda_is_identity():
        be = ref.BlendingEngine(p)
        be.set_dimensions((128, 128))
        be.set_branching(nmb_max_branches=64)
        be.set_prompt1("photo of a reef")
        be.set_prompt2("rendering of an alien planet")
        p.noise.reset()
        p.unet.calls = p.vae.calls = 0
        imgs = be.run_transition(fixed_seeds=[420, 421])
    out["cfg4"] = snapshot(be, imgs, p)
    # cfg 4 under a metric with spread (see position_skewed_engine)
    p = tiny_pipe(turbo=True)
    np.random.seed(0)
    with H.cuda_is_identity():
        be = position_skewed_engine(ref, CFG4_SKEW, CFG4_WIDTH_POWER)(p)
        be.set_dimensions((128, 128))
        be.set_branching(nmb_max_branches=64)
        be.set_prompt1("photo of a reef")
        be.set_prompt2("rendering of an alien planet")
        p.noise.reset()
        p.unet.calls = p.vae.calls = 0
        imgs = be.run_transition(fixed_seeds=[420, 421])
    out["cfg4_skew"] = snapshot(be, imgs, p)
    out["cfg4_skew"].update(metric="reference LPIPS x |fb - fa|^width_power x exp(skew x mean position of the two frames)",
                            skew=CFG4_SKEW, width_power=CFG4_WIDTH_POWER,
                            min_separation=float(min(be.separations)), last_separation=float(be.separations[-1]))
    assert min(be.separations) >= 1.05, ("cfg4_skew: a greedy choice within 5 % of the runner-up", min(be.separations))
    # cfg 5: example_multi_trans.py:39-58 with 6 prompts on the base model
    p = tiny_pipe(turbo=False)
    np.random.seed(0)
    segs = []
    with H.cuda_is_identity():
        be = ref.BlendingEngine(p)
        be.set_negative_prompt(CFG5_NEGATIVE)
        be.set_dimensions((128, 128))
        be.set_num_inference_steps(30)
        be.set_branching(depth_strength=0.5, nmb_max_branches=15)
        p.noise.reset()
        for i in range(len(CFG5_PROMPTS) - 1):
            if i == 0:
                be.set_prompt1(CFG5_PROMPTS[i])
                be.set_prompt2(CFG5_PROMPTS[i + 1])
                recycle = False
            else:
                be.swap_forward()
                be.set_prompt2(CFG5_PROMPTS[i + 1])
                recycle = True
            p.unet.calls = p.vae.calls = 0
            imgs = be.run_transition(recycle_img1=recycle, fixed_seeds=CFG5_SEEDS[i:i + 2])
            segs.append(snapshot(be, imgs, p))
    out["cfg5"] = {"prompts": CFG5_PROMPTS, "seeds": CFG5_SEEDS, "negative_prompt": CFG5_NEGATIVE, "segments": segs}
    return out


GCHAIN_PROMPTS = ["photo of a reef", "rendering of an alien planet", "fog over a harbour"]
GCHAIN_SEEDS = [420, 421, 977]


def guidance_chain_fixture(ref):
    """The guidance scale a transition LEAVES BEHIND (blending_engine.py:358-362 of the reference: set_guidance_mid_dampening
    of the last branch committed, :155-164) and what it does to the next transition's new anchor (compute_latents2 runs
    under it, :370-423): SDXL base tiny, 6 steps, guidance 4.0, one level idx 3 x 6 stems, two chained transitions
    (swap_forward + recycle_img1).  With 6 stems the last branch the greedy order commits (f = 0.125 / 0.375 / ... ) is NOT the
    last one a best-first batched round evaluates, which is what round 5's frontier got wrong."""
    p = tiny_pipe(turbo=False)
    np.random.seed(0)
    segs = []
    with H.cuda_is_identity():
        be = ref.BlendingEngine(p)
        be.set_dimensions((128, 128))
        be.set_num_inference_steps(6)
        be.set_guidance_scale(4.0)
        be.list_idx_injection, be.list_nmb_stems = [3], [6]
        p.noise.reset()
        for i in range(2):
            if i == 0:
                be.set_prompt1(GCHAIN_PROMPTS[0])
                be.set_prompt2(GCHAIN_PROMPTS[1])
            else:
                be.swap_forward()
                be.set_prompt2(GCHAIN_PROMPTS[i + 1])
            p.unet.calls = p.vae.calls = 0
            imgs = be.run_transition(recycle_img1=i > 0, fixed_seeds=GCHAIN_SEEDS[i:i + 2])
            seg = snapshot(be, imgs, p)
            seg["guidance_scale_left_behind"] = float(be.guidance_scale)
            seg["holder_guidance_scale"] = float(be.dh.guidance_scale)
            segs.append(seg)
    return {"prompts": GCHAIN_PROMPTS, "seeds": GCHAIN_SEEDS, "steps": 6, "guidance": 4.0, "list_idx_injection": [3],
            "list_nmb_stems": [6], "segments": segs}


def key_frames(seed, n, h, w):
    rng = np.random.RandomState(seed)
    return [rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in range(n)]


def frames_fixture(ref):
    """reference utils.add_frames_linear_interp, executed unchanged."""
    cases = []
    for seed, n, h, w, target, rng_seed in [(11, 4, 16, 16, 23, 5), (12, 3, 8, 32, 9, 6), (13, 5, 16, 16, 64, 7), (14, 2, 16, 16, 2, 8)]:
        imgs = key_frames(seed, n, h, w)
        np.random.seed(rng_seed)
        out = ref.utils.add_frames_linear_interp([i.copy() for i in imgs], nmb_frames_target=target)
        cases.append({"seed": seed, "n": n, "h": h, "w": w, "target": target, "rng_seed": rng_seed,
                      "numpy": np.__version__, "count": len(out), "sha": [sha(np.asarray(o)) for o in out],
                      "head": [[int(v) for v in np.asarray(o).flatten()[:8]] for o in out]})
    return cases


def main():
    import sys
    os.makedirs(OUT, exist_ok=True)
    ref = H.load_reference()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    only = set(sys.argv[1:])
    makers = [("planner", lambda: planner_fixture(ref)), ("slerp", lambda: slerp_fixture(ref)), ("scheduler", scheduler_fixture),
              ("tree", lambda: tree_fixture(ref)), ("configs", lambda: configs_fixture(ref)), ("frames", lambda: frames_fixture(ref)),
              ("guidance_chain", lambda: guidance_chain_fixture(ref))]
    for name, make in makers:
        if only and name not in only:
            continue
        text = json.dumps(make(), indent=1)
        # (rows of plain numbers on one line each: the 16 x 16 downsamples would otherwise be 40 k lines)
        text = re.sub(r"\[\s*((?:-?[0-9.e+-]+,\s*)+-?[0-9.e+-]+)\s*\]", lambda m: "[" + re.sub(r"\s+", " ", m.group(1)) + "]", text)
        with open(os.path.join(OUT, name + ".json"), "w") as fh:
            fh.write(text)
        print("wrote", name)


if __name__ == "__main__":
    main()
