"""``BlendingEngine`` — latent-blending transitions between two prompts.

Drop-in for the reference class (``latentblending/blending_engine.py:19-789`` in
/root/reference): same constructor, setters, ``run_transition`` / ``compute_latents*`` /
``swap_forward`` / writers, same attribute names for the tree state
(``tree_latents``, ``tree_fracts``, ``tree_final_imgs``, ``tree_idx_injection``,
``tree_similarities``) and the same model-dependent defaults.

What is different underneath:

* tree policy lives in :mod:`latentblending_amd.tree`, schedules in
  :mod:`latentblending_amd.planner` (both pure host code, pinned by golden vectors generated with
  the reference's own functions);
* every tensor operation on the path (parental-mix slerps, conditioning lerps, the denoising
  loop, VAE decode, perceptual distance) is a gfx950 HIP kernel behind the C-ABI when the pipe is
  a ``NativeSDXLPipe``;
* with a native pipe and ``frontier_width > 1`` the greedy insertion loop evaluates the children
  of the ``frontier_width`` widest gaps in ONE batched launch sequence and then commits them in
  the reference's order — the committed tree is identical to the sequential one (a gap's child
  does not depend on insertions elsewhere), unconsumed speculation is dropped at level end.
"""
from __future__ import annotations

import os
import platform
import time
from typing import List, Optional, Sequence

import numpy as np
import torch
from PIL import Image

from . import planner
from .diffusers_holder import DiffusersHolder, _is_native
from .tree import UNSCORED, TransitionTree
from .utils import interpolate_linear, interpolate_spherical

try:  # progress bars are optional
    from tqdm.auto import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **_):
        return x


class BlendingEngine:
    def __init__(self, pipe: None, do_compile: bool = False,
                 guidance_scale_mid_damper: float = 0.5, mid_compression_scaler: float = 1.2,
                 metric=None, frontier_width: int = 1, verbose: bool = True, farm=None):
        """
        Args:
            pipe: SDXL pipeline (``NativeSDXLPipe`` for the gfx950 path, or any diffusers-like pipe).
            do_compile: native pipe -> replay the UNet / VAE launch sequences from hipGraphs.
                (The reference's stable-fast / triton compile does not exist here.)
            guidance_scale_mid_damper: (0, 1]; lowers guidance towards the middle of the transition.
            mid_compression_scaler: kept for signature compatibility (unused upstream as well).
            metric: optional perceptual-distance callable ``metric(a, b)`` on ``[1,3,H,W]`` tensors
                in [-1, 1]; default: the pipe's native LPIPS, else the ``lpips`` package.
            frontier_width: number of gap children evaluated per speculative round (batched on a
                native pipe); 1 = the reference's strictly sequential loop.
            farm: ``latentblending_amd.dist.BranchFarm`` — SPMD over ``torch.distributed``: every rank
                runs this same engine, each round's branches are split over the ranks and
                all-gathered (RCCL over xGMI on GPUs); decisions are taken identically everywhere.
        """
        assert guidance_scale_mid_damper > 0 and guidance_scale_mid_damper <= 1.0, \
            f"guidance_scale_mid_damper neees to be in interval (0,1], you provided {guidance_scale_mid_damper}"

        self.verbose = verbose
        self.dh = DiffusersHolder(pipe)
        self.device = self.dh.device
        self.set_dimensions()

        self.guidance_scale_mid_damper = guidance_scale_mid_damper
        self.mid_compression_scaler = mid_compression_scaler
        self.frontier_width = int(frontier_width)
        self.farm = farm
        self.pair_metric = None             # optional callable(frame_a, frame_b, fract_a, fract_b) -> distance replacing the
        #                                     LPIPS metric on every path (policy stress tests: bench.py --metric-skew, tests)
        self.speculate_virtual = True       # frontier mode: also evaluate children of not-yet-existing gaps
        self.speculation_oversubscribe = 1.0   # frontier rounds after the first: evaluate this many candidates per branch still missing
        self.learn_child_ratio = True          # frontier: predict a virtual gap's distances from the ratios child / parent measured so far
        self.two_stage_speculation = False  # fused first round (single-level trees): False = ALL stems at once in level order - one
        #                                     round when the metric is balanced, several small latency-bound rounds and dropped
        #                                     branches when it is not; True = only the complete top levels of the binary splitting
        #                                     that fit HALF the stems (7 of 15) share the anchors' batches, the second round picks the
        #                                     rest best-first from the 8 gap distances then KNOWN: two rounds whatever the metric
        #                                     (tree identical either way: a gap's child does not depend on the order of evaluation)
        self.speculate_from_previous_tree = False   # opt-in: the blind first round of a level (no distance known yet) takes its candidates
        #                                     from the COMMIT ORDER of the same level of the previous transition on this engine (same
        #                                     injection index and stem count) instead of the level order of the binary splitting.  A
        #                                     prior, nothing more: every branch is still denoised, decoded and scored now, the tree is
        #                                     the reference's greedy tree either way, a wrong prior costs what a blind guess costs today
        #                                     (dropped speculation + another round).  Pays where consecutive transitions bend the tree
        #                                     the same way (re-renders of one transition, a metric with a positional bias): the skewed
        #                                     bench line needs ONE round instead of two
        self._level_priors = {}             # (idx_injection, stems) -> [(f_left, f_right, f_mid)] in the order the level was committed
        self.fuse_anchor_round = True       # single-level trees: first round shares the anchors' UNet batches
        self.fuse_recycled_anchor = True    # ... also when an anchor is recycled (swap_forward chains, precomputed key frames): the
        #                                     fused wavefront takes the stored trajectory as given and denoises only the other one
        self.host_frames = False            # True: run_transition hands back HOST PIL images (the reference's return type in full): one
        #                                     device->host copy of all frames through a pinned buffer + their PIL cores, ~1.6 ms per
        #                                     17-frame transition; False (default): lazy DeviceImage frames, copied when first touched.
        #                                     (Two overlapped forms - decode as 3/4 + 1/4 batches, or copy on a side stream and build
        #                                     behind the LPIPS kernels - both measured SLOWER than the plain copy at the end:
        #                                     profiles/r03_materialise_ab.txt)
        self.elide_dead_steps = False       # opt-in (native fused wavefront): skip mid steps the next step's crossfeed (coefficient
        #                                     exactly 1.0, the Turbo defaults) overwrites completely - bit-identical frames, fewer
        #                                     UNet forwards than the reference performs; tree_latents entries of the
        #                                     skipped steps are None
        self.keyframe_chunk = 16            # precompute_keyframes / batched decodes: at most this many samples per UNet / VAE batch
        self.seed1 = 0
        self.seed2 = 0
        self.prompt1 = ""
        self.prompt2 = ""

        self._tree = TransitionTree()
        self.idx_injection = []
        self.tree_status = None
        self.text_embedding1 = None
        self.text_embedding2 = None
        self.image1_lowres = None
        self.image2_lowres = None
        self.negative_prompt = None
        self.stats = {}
        self._preset_anchor_frames = None      # (frame of anchor 1, frame of anchor 2) for the next transition: see preset_anchors
        self._anchor_frames_given = (None, None)    # ... the ones of them that belong to anchors really recycled in the running call

        self.set_guidance_scale()
        self.multi_transition_img_first = None
        self.multi_transition_img_last = None
        self.dt_unet_step = 0
        self.lpips = self._resolve_metric(metric)

        self.set_prompt1("")
        self.set_prompt2("")
        self.set_branch1_crossfeed()
        self.set_parental_crossfeed()
        self.set_num_inference_steps()
        self.benchmark_speed()
        self.set_branching()

        if do_compile:
            if _is_native(self.dh.pipe):
                self.dh.pipe.enable_graphs(True)
            else:
                self._say("do_compile: no compiler for non-native pipes on this platform; ignored")

    # ------------------------------------------------------------------ tree state (API) ---
    tree_latents = property(lambda s: s._tree.latents, lambda s, v: setattr(s._tree, "latents", v))
    tree_fracts = property(lambda s: s._tree.fracts, lambda s, v: setattr(s._tree, "fracts", v))
    tree_final_imgs = property(lambda s: s._tree.frames, lambda s, v: setattr(s._tree, "frames", v))
    tree_idx_injection = property(lambda s: s._tree.idx_injection,
                                  lambda s, v: setattr(s._tree, "idx_injection", v))
    tree_similarities = property(lambda s: s._tree.similarities,
                                 lambda s, v: setattr(s._tree, "similarities", v))

    def _say(self, msg):
        if self.verbose:
            print(msg)

    def _resolve_metric(self, metric):
        if metric is not None:
            return metric
        native = getattr(self.dh.pipe, "lpips_metric", None)
        if native is not None:
            return native
        import lpips  # the reference's dependency; absent -> ImportError, as upstream
        net = lpips.LPIPS(net='alex')
        return net if platform.system() == "Darwin" else net.cuda(self.device)

    # ------------------------------------------------------------------ setters -----------
    def benchmark_speed(self):
        """Time one UNet step and one VAE decode; both feed the time-budget planner."""
        self._say("starting speed benchmark...")
        emb = self.dh.get_text_embedding("test")
        start = self.dh.get_noise(np.random.randint(111111))
        last = self.num_inference_steps - 1
        self.dh.run_diffusion_sd_xl(text_embeddings=emb, latents_start=start,
                                    return_image=False, idx_start=last)  # warm-up
        self._sync()
        t0 = time.time()
        traj = self.dh.run_diffusion_sd_xl(text_embeddings=emb, latents_start=start,
                                           return_image=False, idx_start=last)
        self._sync()
        self.dt_unet_step = time.time() - t0
        t0 = time.time()
        self.dh.latent2image(traj[-1])
        self._sync()
        self.dt_vae = time.time() - t0
        self._say(f"time per unet iteration: {self.dt_unet_step} time for vae: {self.dt_vae}")

    def _sync(self):
        if _is_native(self.dh.pipe):
            self.dh.pipe.synchronize()

    def set_dimensions(self, size_output=None):
        """(width, height) of the output; default 512² for turbo, 1024² otherwise."""
        if size_output is None:
            size_output = (512, 512) if self.dh.is_sdxl_turbo else (1024, 1024)
        self.dh.set_dimensions(size_output)

    def set_guidance_scale(self, guidance_scale=None):
        if guidance_scale is None:
            guidance_scale = 0.0 if self.dh.is_sdxl_turbo else 4.0
        self.guidance_scale_base = guidance_scale
        self.guidance_scale = guidance_scale
        self.dh.guidance_scale = guidance_scale

    def set_negative_prompt(self, negative_prompt):
        """One negative prompt.  As upstream, prompts embedded before this call are not re-embedded."""
        self.negative_prompt = negative_prompt
        self.dh.set_negative_prompt(negative_prompt)

    def set_guidance_mid_dampening(self, fract_mixing):
        g = planner.damped_guidance(self.guidance_scale_base, self.guidance_scale_mid_damper,
                                    fract_mixing)
        self.guidance_scale = g
        self.dh.guidance_scale = g

    def set_branch1_crossfeed(self, crossfeed_power=0, crossfeed_range=0, crossfeed_decay=0):
        """Crossfeed from the first anchor's trajectory into the second anchor (all in [0,1])."""
        self.branch1_crossfeed_power = np.clip(crossfeed_power, 0, 1)
        self.branch1_crossfeed_range = np.clip(crossfeed_range, 0, 1)
        self.branch1_crossfeed_decay = np.clip(crossfeed_decay, 0, 1)

    def set_parental_crossfeed(self, crossfeed_power=None, crossfeed_range=None, crossfeed_decay=None):
        """Crossfeed from the parents into every mid branch.  Turbo defaults 1/1/1; for other pipes
        the arguments are overridden with 0.3/0.6/0.9 exactly like upstream (blending_engine.py:200-203)."""
        if self.dh.is_sdxl_turbo:
            crossfeed_power = 1.0 if crossfeed_power is None else crossfeed_power
            crossfeed_range = 1.0 if crossfeed_range is None else crossfeed_range
            crossfeed_decay = 1.0 if crossfeed_decay is None else crossfeed_decay
        else:
            crossfeed_power, crossfeed_range, crossfeed_decay = 0.3, 0.6, 0.9
        self.parental_crossfeed_power = np.clip(crossfeed_power, 0, 1)
        self.parental_crossfeed_range = np.clip(crossfeed_range, 0, 1)
        self.parental_crossfeed_decay = np.clip(crossfeed_decay, 0, 1)

    def set_prompt1(self, prompt: str):
        self.prompt1 = prompt.replace("_", " ")
        self.text_embedding1 = self.get_text_embeddings(self.prompt1)

    def set_prompt2(self, prompt: str):
        self.prompt2 = prompt.replace("_", " ")
        self.text_embedding2 = self.get_text_embeddings(self.prompt2)

    def set_image1(self, image: Image):
        self.image1_lowres = image

    def set_image2(self, image: Image):
        self.image2_lowres = image

    def set_num_inference_steps(self, num_inference_steps=None):
        if num_inference_steps is None:
            num_inference_steps = 4 if self.dh.is_sdxl_turbo else 30
        self.num_inference_steps = num_inference_steps
        self.dh.set_num_inference_steps(num_inference_steps)

    def set_branching(self, depth_strength=None, t_compute_max_allowed=None, nmb_max_branches=None):
        """Plan ``list_idx_injection`` / ``list_nmb_stems``.  Turbo: one level (default index 2,
        10 mid branches).  Otherwise: time budget (default 20 s) or a frame budget."""
        if self.dh.is_sdxl_turbo:
            assert t_compute_max_allowed is None, "time-based branching not supported for SDXL Turbo"
            self.list_idx_injection, self.list_nmb_stems = planner.turbo_branching(
                self.num_inference_steps, depth_strength, nmb_max_branches)
            return
        if depth_strength is None:
            depth_strength = 0.5
        if t_compute_max_allowed is None and nmb_max_branches is None:
            t_compute_max_allowed = 20
        elif t_compute_max_allowed is not None and nmb_max_branches is not None:
            raise ValueError("Either specify t_compute_max_allowed or nmb_max_branches")
        if self._farm_on() and t_compute_max_allowed is not None:
            # a time budget is planned from measured step times: every rank must plan from the SAME numbers
            self.dt_unet_step, self.dt_vae = self.farm.broadcast_floats([self.dt_unet_step, self.dt_vae], src=0)
        self.list_idx_injection, self.list_nmb_stems = self.get_time_based_branching(
            depth_strength, t_compute_max_allowed, nmb_max_branches)

    def get_time_based_branching(self, depth_strength, t_compute_max_allowed=None, nmb_max_branches=None):
        return planner.time_based_branching(self.num_inference_steps, depth_strength,
                                            self.dt_unet_step, self.dt_vae,
                                            t_compute_max_allowed, nmb_max_branches)

    # ------------------------------------------------------------------ transition --------
    def run_transition(self, recycle_img1: Optional[bool] = False,
                       recycle_img2: Optional[bool] = False,
                       fixed_seeds: Optional[List[int]] = None):
        """Compute the transition; returns the list of frames (PIL images), prompt1 -> prompt2.

        recycle_img1 / recycle_img2: reuse the stored anchor trajectory (after ``swap_forward``).
        fixed_seeds: two seeds for the anchors, or 'randomize'; ``None`` keeps the current seeds.
        """
        assert self.text_embedding1 is not None, 'Set the first text embedding with .set_prompt1(...) before'
        assert self.text_embedding2 is not None, 'Set the second text embedding with .set_prompt2(...) before'

        if fixed_seeds is not None:
            if isinstance(fixed_seeds, str) and fixed_seeds == 'randomize':
                fixed_seeds = list(np.random.randint(0, 1000000, 2).astype(np.int32))
            else:
                assert len(fixed_seeds) == 2, "Supply a list with len = 2"
            self.seed1, self.seed2 = fixed_seeds[0], fixed_seeds[1]

        steps = self.num_inference_steps
        use_frontier = self.frontier_width > 1 or self.farm is not None
        keep1 = recycle_img1 and len(self.tree_latents[0]) == steps
        keep2 = recycle_img2 and len(self.tree_latents[-1]) == steps
        # frames handed over by preset_anchors belong to the trajectories it installed: an anchor that is denoised again
        # in this call must be decoded again too (and a preset must never outlive the call it was made for)
        preset, self._preset_anchor_frames = self._preset_anchor_frames, None
        self._anchor_frames_given = (preset[0] if preset is not None and keep1 else None,
                                     preset[1] if preset is not None and keep2 else None)
        prefilled = None
        restore_noise = self._farm_begin(keep1, keep2) if self._farm_on() else None
        try:
            recycled_ok = self.fuse_recycled_anchor and not self._farm_on()
            fuse = (use_frontier and self.fuse_anchor_round and _is_native(self.dh.pipe)
                    and ((not keep1 and not keep2) or recycled_ok) and self.branch1_crossfeed_power == 0.0 and self.speculate_virtual
                    and len(self.list_idx_injection) == 1 and int(self.list_idx_injection[0]) >= 1
                    and int(self.list_nmb_stems[0]) >= 1 and self._uniform_cfg())
            if fuse:
                first, last, prefilled = self._anchors_with_first_round(keep1, keep2)
            elif self._farm_on():
                first, last = self._anchors_distributed(keep1, keep2)
            elif use_frontier and _is_native(self.dh.pipe) and not keep1 and not keep2 \
                    and self.branch1_crossfeed_power == 0.0:
                first, last = self._compute_anchors_batched()
            else:
                first = self.tree_latents[0] if keep1 else self.compute_latents1()
                last = self.tree_latents[-1] if keep2 else self.compute_latents2()
            frames = self._grow_tree(first, last, prefilled, use_frontier)
            if self.host_frames and _is_native(self.dh.pipe):
                from .native.frames import materialise_frames
                materialise_frames(frames)              # ONE device->host copy of the whole transition + the PIL pixel cores
            return frames
        finally:
            if restore_noise is not None:
                restore_noise()

    def _grow_tree(self, first, last, prefilled, use_frontier):

        if prefilled is None:
            frames = list(self._anchor_frames_given)
            todo = [k for k in (0, 1) if frames[k] is None]
            if todo:
                for k, f in zip(todo, self._decode_many([(first, last)[k][-1] for k in todo])):
                    frames[k] = f
            self._tree.reset(first, last, frames[0], frames[1])

        for level in tqdm(range(len(self.list_idx_injection)), disable=not self.verbose):
            stems = int(self.list_nmb_stems[level])
            idx_injection = int(self.list_idx_injection[level])
            if use_frontier:
                self._grow_level_frontier(idx_injection, stems, ready=prefilled)
                prefilled = None
            else:
                for _ in range(stems):
                    fract, p1, p2 = self.get_mixing_parameters(idx_injection)
                    self.set_guidance_mid_dampening(fract)
                    branch = self.compute_latents_mix(fract, p1, p2, idx_injection)
                    self.insert_into_tree(fract, idx_injection, branch)
        return self.tree_final_imgs

    # ------------------------------------------------------------------ farm (SPMD) ---------
    def _farm_on(self) -> bool:
        return self.farm is not None and self.farm.world > 1

    def _farm_begin(self, keep1, keep2):
        """Start of a farmed transition: (1) every rank must hold the same plan and settings — checked with one tiny
        all-gather, so a divergence stops here with a message instead of hanging in a later collective whose shapes
        depend on the plan; (2) ancestral noise must be the same stream on every rank (ranks compute anchors
        redundantly and branches independently): unless the user installed a noise source, rank 0 draws a seed,
        broadcasts it, and every rank samples from a device generator seeded with it for this transition.
        Returns a callable that restores the pipe's noise source, or None."""
        plan = [float(v) for v in self.list_idx_injection] + [-1.0] + [float(v) for v in self.list_nmb_stems] + [
            float(self.num_inference_steps), float(self.frontier_width), float(self.guidance_scale_base),
            float(self.branch1_crossfeed_power), float(self.parental_crossfeed_power), float(self.parental_crossfeed_range),
            float(self.parental_crossfeed_decay), float(self.seed1), float(self.seed2), float(bool(keep1)), float(bool(keep2))]
        self.farm.check_consistent(plan, "branching plan / settings")
        return self._farm_shared_noise()

    def _farm_shared_noise(self):
        """Ancestral noise as ONE stream on every rank (see ``_farm_begin``); returns the restore callable or None."""
        sched = getattr(self.dh.pipe, "scheduler", None)
        if _is_native(self.dh.pipe) and getattr(sched, "ancestral", False) and sched.noise_source is None:
            from .native.scheduler import SeededDeviceNoise
            seed = int(self.farm.broadcast_floats([float(np.random.randint(0, 2 ** 31 - 1))], src=0)[0])
            sched.noise_source = SeededDeviceNoise(seed, self.dh.pipe.device)

            def restore():
                sched.noise_source = None
            return restore
        return None

    def _uniform_cfg(self) -> bool:
        """True when the anchors and every possible mid branch agree on using classifier-free guidance (the mid
        dampening can push a sub-unity guidance above 1 towards the middle of the transition)."""
        pipe = self.dh.pipe
        g_mid = planner.damped_guidance(self.guidance_scale_base, self.guidance_scale_mid_damper, 0.5)
        return pipe.uses_cfg(self.guidance_scale_base) == pipe.uses_cfg(g_mid)

    def compute_latents1(self, return_image=False):
        """Full trajectory of the first anchor (pure prompt1)."""
        self._say("starting compute_latents1")
        cond = self.get_mixed_conditioning(0)
        t0 = time.time()
        start = self.get_noise(self.seed1)
        traj = self.run_diffusion(cond, latents_start=start, idx_start=0)
        self._sync()                                     # (native launches are asynchronous)
        self.dt_unet_step = (time.time() - t0) / self.num_inference_steps
        self.tree_latents[0] = traj
        return self.dh.latent2image(traj[-1]) if return_image else traj

    def compute_latents2(self, return_image=False):
        """Full trajectory of the second anchor, optionally crossfed from the first."""
        self._say("starting compute_latents2")
        cond = self.get_mixed_conditioning(1)
        start = self.get_noise(self.seed2)
        if self.branch1_crossfeed_power > 0.0:
            coeffs = planner.anchor_crossfeed_coeffs(
                self.num_inference_steps, self.branch1_crossfeed_power,
                self.branch1_crossfeed_range, self.branch1_crossfeed_decay)
            traj = self.run_diffusion(cond, latents_start=start, idx_start=0,
                                      list_latents_mixing=self.tree_latents[0],
                                      mixing_coeffs=coeffs)
        else:
            traj = self.run_diffusion(cond, start)
        self.tree_latents[-1] = traj
        return self.dh.latent2image(traj[-1]) if return_image else traj

    def _compute_anchors_batched(self):
        """Both anchors as ONE batch-2 denoising run (native pipes; they are independent when the
        first anchor is not crossfed into the second).  Halves the number of weight-streaming
        UNet passes of the anchor phase; results equal the two sequential runs."""
        self.dh.set_num_inference_steps(self.num_inference_steps)
        conds = [self.get_mixed_conditioning(0)[0], self.get_mixed_conditioning(1)[0]]
        starts = [self.get_noise(self.seed1), self.get_noise(self.seed2)]
        zeros = [0.0] * self.num_inference_steps
        first, last = self.dh.pipe.native_run_diffusion_batch(
            conds, starts, 0, [None, None], [zeros, zeros], num_inference_steps=self.num_inference_steps,
            guidance_scales=[self.guidance_scale, self.guidance_scale])
        # (dt_unet_step keeps its benchmark_speed() value: a batched, asynchronous run is not a per-step timing)
        self.tree_latents[0], self.tree_latents[-1] = first, last
        return first, last

    def precompute_keyframes(self, embeddings: Sequence, seeds: Sequence[int]):
        """Cross-transition pipelining of a multi-transition chain (the loop of the reference's
        example_multi_trans.py:39-58 diffuses key frame k+1 only when transition k starts, one latency-bound batch-1
        trajectory per transition): ALL key frames are denoised ahead of the transitions - one lock-step batch on a
        native pipe, key frame k on rank ``k % world`` under a farm (then broadcast, C1 of §8e) - and decoded in one
        VAE batch.  ``embeddings[k]`` is the embedding tuple of prompt k, ``seeds[k]`` its seed.  Returns
        ``(trajectories, frames)``; ``replay.run_multi_transition(pipeline_keyframes=True)`` hands them to
        ``run_transition(recycle_img1=True, recycle_img2=True)`` through ``tree_latents`` / ``preset_anchor_frames``.

        Differences from the sequential chain, by construction: every key frame is denoised with the guidance scale
        current at this call (the sequential loop leaves the mid-dampened scale of the previous transition's last
        branch for the next key frame, blending_engine.py:266-270 + :531 of the reference), and an ancestral noise
        stream is consumed key frame by key frame before any mid branch draws from it.  Not available with
        ``branch1_crossfeed_power > 0`` (key frame k+1 then depends on key frame k)."""
        assert self.branch1_crossfeed_power == 0.0, "precompute_keyframes: key frames are only independent without branch1 crossfeed"
        assert len(embeddings) == len(seeds) and len(embeddings) >= 1
        steps, n = self.num_inference_steps, len(embeddings)
        self.dh.set_num_inference_steps(steps)
        pipe = self.dh.pipe
        farm = self.farm if self._farm_on() else None
        ancestral = bool(getattr(getattr(pipe, "scheduler", None), "ancestral", False))
        restore_noise = self._farm_shared_noise() if farm else None
        try:
            mine = [k for k in range(n) if farm is None or farm.owner_of(k) == farm.rank]
            trajs = [None] * n
            if _is_native(pipe):
                if mine:
                    zeros = [0.0] * steps
                    # bounded program shapes: a long chain is denoised in chunks of ``keyframe_chunk`` key frames (one UNet
                    # program per distinct batch size stays cached with its arena).  Ancestral noise is drawn sample-major
                    # by the batch call, so consecutive chunks consume a stream in the same order as one big batch; under
                    # a farm a rank's share (n / world key frames) goes in one call (its noise slots span the whole round)
                    width = len(mine) if farm else max(1, int(self.keyframe_chunk))
                    for c0 in range(0, len(mine), width):
                        part = mine[c0:c0 + width]
                        got = pipe.native_run_diffusion_batch(
                            [embeddings[k] for k in part], [self.get_noise(int(seeds[k])) for k in part], 0, [None] * len(part),
                            [zeros] * len(part), num_inference_steps=steps, guidance_scales=[self.guidance_scale] * len(part),
                            noise_slots=(n, part) if farm else None)
                        for k, t in zip(part, got):
                            trajs[k] = t
                elif ancestral:          # fewer key frames than ranks: a rank without one still advances a shared noise stream
                    self._skip_noise_draws(n * steps)
            else:
                for k in range(n):                       # (generic pipes: one trajectory at a time, in chain order)
                    if k in mine:
                        trajs[k] = self.run_diffusion([embeddings[k]], latents_start=self.get_noise(int(seeds[k])), idx_start=0)
                    elif ancestral:
                        self._skip_noise_draws(steps)
            if farm:
                shape = (1,) + self._latent_chw()
                for k in range(n):
                    owner = farm.owner_of(k)
                    trajs[k] = self._on_pipe_device(farm.share_trajectory(trajs[k] if owner == farm.rank else None, owner, steps, shape))
            frames = self._decode_many([t[-1] for t in trajs])      # every rank decodes the same batch: identical frames
        finally:
            if restore_noise is not None:
                restore_noise()
        self._sync()
        self.stats["keyframes_precomputed"] = self.stats.get("keyframes_precomputed", 0) + n
        return trajs, frames

    def preset_anchors(self, first, last, frame_first=None, frame_last=None):
        """Install two finished anchor trajectories (and, optionally, their decoded frames) for the next
        ``run_transition(recycle_img1=True, recycle_img2=True)``."""
        assert len(first) == len(last) == self.num_inference_steps, \
            "preset_anchors: both trajectories must have num_inference_steps entries"
        self.tree_latents[0], self.tree_latents[-1] = first, last
        self._preset_anchor_frames = (frame_first, frame_last) if frame_first is not None and frame_last is not None else None

    def _parental_mix(self, b_parent1, b_parent2, fract_parental):
        """Slerp the two parents' trajectories step by step (``None`` where either has no latent)."""
        lat1, lat2 = self.tree_latents[b_parent1], self.tree_latents[b_parent2]
        have = [i for i in range(self.num_inference_steps)
                if lat1[i] is not None and lat2[i] is not None]
        mixed = [None] * self.num_inference_steps
        if _is_native(self.dh.pipe) and have:
            from .backend import get_backend
            outs = get_backend().slerp_pairs([lat1[i] for i in have], [lat2[i] for i in have],
                                             [fract_parental] * len(have))
            for i, o in zip(have, outs):
                mixed[i] = o
        else:
            for i in have:
                mixed[i] = interpolate_spherical(lat1[i], lat2[i], fract_parental)
        return mixed

    def compute_latents_mix(self, fract_mixing, b_parent1, b_parent2, idx_injection):
        """Trajectory of a mid branch at ``fract_mixing`` injected at step ``idx_injection`` from the
        slerp of its parents' trajectories, with parental crossfeed."""
        cond = self.get_mixed_conditioning(fract_mixing)
        f1, f2 = self.tree_fracts[b_parent1], self.tree_fracts[b_parent2]
        fract_parental = (fract_mixing - f1) / (f2 - f1)
        mixed = self._parental_mix(b_parent1, b_parent2, fract_parental)
        coeffs = planner.parental_crossfeed_coeffs(
            self.num_inference_steps, idx_injection, self.parental_crossfeed_power,
            self.parental_crossfeed_range, self.parental_crossfeed_decay)
        return self.run_diffusion(cond, latents_start=mixed[idx_injection - 1],
                                  idx_start=idx_injection, list_latents_mixing=mixed,
                                  mixing_coeffs=coeffs)

    def get_mixing_parameters(self, idx_injection):
        """(fract, parent1, parent2) of the next branch: midpoint of the least-similar gap."""
        return self._tree.next_split(idx_injection)

    def insert_into_tree(self, fract_mixing, idx_injection, list_latents):
        """Decode the branch, measure it against both neighbours and commit it."""
        frame = self.dh.latent2image(list_latents[-1])
        lo, hi = self.get_closest_idx(fract_mixing)
        if self.pair_metric is not None:
            left = float(self.pair_metric(frame, self.tree_final_imgs[lo], fract_mixing, self.tree_fracts[lo]))
            right = float(self.pair_metric(frame, self.tree_final_imgs[hi], fract_mixing, self.tree_fracts[hi]))
        else:
            left = self.get_lpips_similarity(frame, self.tree_final_imgs[lo])
            right = self.get_lpips_similarity(frame, self.tree_final_imgs[hi])
        self._tree.commit(fract_mixing, idx_injection, list_latents, frame, left, right)

    def _decode_many(self, latents: list):
        if _is_native(self.dh.pipe) and len(latents) > 1:
            width = max(1, int(self.keyframe_chunk))        # bounded VAE program shapes / arenas (see precompute_keyframes)
            out = []
            for c0 in range(0, len(latents), width):
                part = latents[c0:c0 + width]
                out += self.dh.pipe.native_latent2image_batch(part, "pil") if len(part) > 1 else [self.dh.latent2image(part[0])]
            return out
        return [self.dh.latent2image(z) for z in latents]

    # speculative frontier (native pipes) ---------------------------------------------------
    @staticmethod
    def _bfs_midpoints(count: int):
        """(left, right, mid) of the first ``count`` gaps of [0,1] in level order — the order in which the
        best-first speculation visits gaps while no distance is known yet."""
        out, level = [], [(0.0, 1.0)]
        while len(out) < count:
            nxt = []
            for fl, fr in level:
                if len(out) < count:
                    out.append((fl, fr, (fl + fr) / 2))
                nxt += [(fl, (fl + fr) / 2), ((fl + fr) / 2, fr)]
            level = nxt
        return out

    def _anchors_with_first_round(self, keep1=False, keep2=False):
        """Single-level trees on a native pipe: every mid branch mixes the two ANCHORS, and at step i
        it only needs their latents of step i-1.  So the first speculative round (level-order
        midpoints, exactly what the best-first frontier would pick before any distance is known) is
        denoised in the same UNet batches as the anchors' own steps >= idx_injection, all frames
        are decoded in one batch, and the greedy loop then starts from a pre-filled pool.
        ``keep1`` / ``keep2``: that anchor is recycled (``tree_latents[0]`` / ``[-1]`` hold its trajectory): only the
        other one is denoised; frames handed over by ``preset_anchors`` are used instead of decoding the anchors again."""
        pipe, steps = self.dh.pipe, self.num_inference_steps
        idx_injection, stems = int(self.list_idx_injection[0]), int(self.list_nmb_stems[0])
        self.dh.set_num_inference_steps(steps)
        width = min(self.frontier_width, stems)
        if self.two_stage_speculation:
            w = 1
            while 2 * w + 1 <= (stems + 1) // 2:
                w = 2 * w + 1               # complete levels of the binary splitting: 1, 3, 7, 15, ...
            width = min(width, w)
        gaps = self._bfs_midpoints(width)
        prior = self._level_priors.get((idx_injection, stems)) if self.speculate_from_previous_tree and not self.two_stage_speculation else None
        if prior is not None and len(prior) >= width:
            gaps = list(prior[:width])          # (commit order of the previous transition: every gap's ends are 0, 1 or earlier midpoints)
        coeffs = planner.parental_crossfeed_coeffs(steps, idx_injection, self.parental_crossfeed_power,
                                                   self.parental_crossfeed_range, self.parental_crossfeed_decay)
        guid = [planner.damped_guidance(self.guidance_scale_base, self.guidance_scale_mid_damper, m) for _, _, m in gaps]
        # farm: every rank runs BOTH anchors (no exchange, no idle ranks) plus its round-robin share of the mids
        farm = self.farm if self._farm_on() else None
        mine = farm.my_indices(len(gaps)) if farm else list(range(len(gaps)))
        first, last, mids = pipe.native_run_wavefront(
            [self.get_mixed_conditioning(0)[0], self.get_mixed_conditioning(1)[0]],
            [self.get_noise(self.seed1), self.get_noise(self.seed2)],
            (lambda: [self.get_mixed_conditioning(gaps[k][2])[0] for k in mine]), [gaps[k][2] for k in mine],     # (built behind the first launch)
            [coeffs] * len(mine), idx_injection, steps, self.guidance_scale, [guid[k] for k in mine],
            noise_slots=(len(gaps), mine) if farm else None, elide_dead_steps=self.elide_dead_steps and not farm,
            known_anchors=(self.tree_latents[0] if keep1 else None, self.tree_latents[-1] if keep2 else None))
        self.tree_latents[0], self.tree_latents[-1] = first, last       # (what compute_latents1 / 2 leave behind)
        given = self._anchor_frames_given
        if given[0] is not None and given[1] is not None and not farm:
            frames = list(given) + (pipe.native_latent2image_batch([t[-1] for t in mids], "pil") if mids else [])
        elif farm and farm.rank != 0:
            # the anchors' FRAMES come from rank 0 in the broadcast below: only their owner decodes them (at 8 ranks the
            # decode batch of a non-owner halves: 2 mid frames instead of 2 + 2)
            frames = [None, None] + (pipe.native_latent2image_batch([t[-1] for t in mids], "pil") if mids else [])
        else:
            frames = pipe.native_latent2image_batch([first[-1], last[-1]] + [t[-1] for t in mids], "pil")
        if farm:    # C1: rank 0's anchors (stacks + frames) become everybody's - bit-identical parents / end frames on all ranks
            (first, last), anchor_frames = farm.share_anchor_pair([first, last], frames[:2], 0, steps, self._frame_from_u8,
                                                                  self._latent_chw(), (self.dh.height_img, self.dh.width_img))
            first, last = self._on_pipe_device(first), self._on_pipe_device(last)
            frames = list(anchor_frames) + list(frames[2:])
        self._tree.reset(first, last, frames[0], frames[1])
        mid_frames = frames[2:]
        if farm:
            got = farm.exchange_branches(list(zip(mids, mid_frames)), len(gaps), steps - idx_injection, steps,
                                         self._frame_from_u8, self._latent_chw(), (self.dh.height_img, self.dh.width_img))
            mids, mid_frames = [self._on_pipe_device(t) for t, _ in got], [f for _, f in got]
            for k in mine:                      # keep this rank's own frame objects (their LPIPS features may be cached)
                mid_frames[k] = frames[2 + mine.index(k)]
        frame_at = {0.0: frames[0], 1.0: frames[1]}
        frame_at.update({m: f for (_, _, m), f in zip(gaps, mid_frames)})
        sims = self._gap_child_distances([(frame_at[m], frame_at[fl], frame_at[fr]) for (fl, fr, m) in gaps],
                                         [(m, fl, fr) for (fl, fr, m) in gaps])
        ready = {(fl, fr): dict(fract=m, traj=traj, frame=frame_at[m], sl=sims[k][0], sr=sims[k][1])
                 for k, ((fl, fr, m), traj) in enumerate(zip(gaps, mids))}
        # (the guidance scale this transition leaves behind is set where the branches are COMMITTED: _grow_level_frontier)
        self.stats["frontier_rounds"] = self.stats.get("frontier_rounds", 0) + 1
        self.stats["speculation_evaluated"] = self.stats.get("speculation_evaluated", 0) + len(gaps)
        return first, last, ready

    def _grow_level_frontier(self, idx_injection: int, stems: int, ready=None):
        """Commit ``stems`` branches at this level.  Each round evaluates up to ``frontier_width`` gap
        children in one batch, chosen best-first over the REAL gaps of the tree and the VIRTUAL gaps
        that will exist once an already evaluated (or just picked) child is committed — so the early
        rounds of the binary splitting run full batches instead of 1, 2, 4 branches.

        Exactness: a gap (left fraction, right fraction) has exactly one possible child — its midpoint,
        mixed from the parents of the enclosing real gap (same-level nodes are never parents) — and its
        distances to the gap's two end frames do not depend on anything else.  Evaluated children wait
        in ``ready`` until the reference's greedy order asks for their gap; whatever is never asked
        for is dropped at the end of the level."""
        import heapq
        import math
        tree = self._tree
        ready = dict(ready) if ready else {}     # (f_left, f_right) -> dict(fract, traj, frame, sl, sr)
        remaining = stems
        committed_gaps = []                      # (f_left, f_right, f_mid) in commit order: next transition's prior (speculate_from_previous_tree)
        prior = self._level_priors.get((idx_injection, stems)) if self.speculate_from_previous_tree else None
        prior_rank = {(fl, fr): k for k, (fl, fr, _) in enumerate(prior)} if prior else {}
        last_committed = None                    # fraction of the branch the greedy order committed last at this level
        ratio_l, ratio_r = [], []                # measured (child-to-left-end, child-to-right-end) distance / parent gap distance

        def learn(key, r, parent):               # one sample per evaluated child whose parent gap distance is known
            if parent is not None and parent is not UNSCORED and float(parent) > 0 and math.isfinite(float(parent)):
                ratio_l.append(float(r["sl"]) / float(parent))
                ratio_r.append(float(r["sr"]) / float(parent))

        def virtual_parent(fl, fr):              # distance of the not-yet-real gap (fl, fr): a half of an evaluated child's gap
            width = fr - fl
            for key in ((fl, fr + width), (fl - width, fr)):
                r = ready.get(key)
                if r is not None and r["fract"] in (fl, fr):
                    return r["sl"] if r["fract"] == fr else r["sr"]
            return None
        while remaining > 0:
            # 1) commit everything the greedy order can already consume
            progressed = True
            while remaining > 0 and progressed:
                progressed = False
                gap = tree.widest_gap()
                key = (tree.fracts[gap], tree.fracts[gap + 1])
                if key in ready:
                    r = ready.pop(key)
                    learn(key, r, tree.similarities[gap])
                    tree.commit(r["fract"], idx_injection, r["traj"], r["frame"], r["sl"], r["sr"])
                    committed_gaps.append((key[0], key[1], r["fract"]))
                    last_committed = r["fract"]
                    remaining -= 1
                    progressed = True
            if remaining == 0:
                break
            # 2) pick what to evaluate next: best-first over real + virtual gaps.  A virtual gap's distance is PREDICTED from its
            #    parent's: parent x the median ratio measured on this level so far (left and right halves separately: a metric
            #    that grows along the transition bends the tree, and the halves of a gap are not equally wide in its eyes);
            #    parent / 2 until something was measured.
            r_l = r_r = 0.5
            if self.learn_child_ratio:
                samples_l, samples_r = list(ratio_l), list(ratio_r)
                for (fl, fr), r in ready.items():          # evaluated, not yet committed: their parents may be virtual themselves
                    par = virtual_parent(fl, fr)
                    if par is not None and float(par) > 0:
                        samples_l.append(float(r["sl"]) / float(par))
                        samples_r.append(float(r["sr"]) / float(par))
                if samples_l:
                    r_l, r_r = sorted(samples_l)[len(samples_l) // 2], sorted(samples_r)[len(samples_r) // 2]
            heap, tick = [], 0
            unscored = any(s is UNSCORED for s in tree.similarities)
            blind_prior = unscored and bool(prior_rank)     # no distance known yet: rank gaps by the previous transition's commit order

            def blind_estimate(a, b):                # earlier in the prior = larger estimate; gaps the prior never split come last
                k = prior_rank.get((a, b))
                return float(len(prior_rank) - k) if k is not None else -1.0
            for g in range(len(tree.fracts) - 1):
                est = float("inf") if unscored else float(tree.similarities[g])
                if blind_prior:
                    est = blind_estimate(tree.fracts[g], tree.fracts[g + 1])
                heap.append((-est, tick, tree.fracts[g], tree.fracts[g + 1], g))
                tick += 1
            heapq.heapify(heap)
            # Walk the greedy order FORWARD on exact + predicted distances: every pop is one commit the reference's loop would make
            # next - a child already evaluated (`ready`: consumed virtually, its halves carry exact distances) or a candidate to
            # evaluate now.  `remaining` pops = exactly the candidates this level still needs if the predictions hold (evaluated
            # children the greedy order never asks for do not count: round 3 charged them against the budget and crawled one
            # candidate per round under a skewed metric); `speculation_oversubscribe` > 1 walks further, so that a near miss
            # is already covered (small batches cost little more than one sample).
            over = 1.0 if unscored else float(self.speculation_oversubscribe)      # (a blind first round gains nothing from guessing further)
            target = max(remaining, int(math.ceil(remaining * over)))
            specs, pops = [], 0
            while heap and pops < target and len(specs) < self.frontier_width:
                pops += 1
                neg_est, _, fl, fr, g = heapq.heappop(heap)
                mid = (fl + fr) / 2
                if (fl, fr) in ready:                      # child known: its halves have exact distances
                    est_l, est_r = ready[(fl, fr)]["sl"], ready[(fl, fr)]["sr"]
                else:
                    _, p1, p2 = tree.gap_child(g, idx_injection)
                    f1, f2 = tree.fracts[p1], tree.fracts[p2]
                    specs.append(dict(
                        gap=g, left=fl, right=fr, fract=mid,
                        guidance=planner.damped_guidance(self.guidance_scale_base, self.guidance_scale_mid_damper, mid),
                        cond=self.get_mixed_conditioning(mid)[0],
                        mixed=self._parental_mix(p1, p2, (mid - f1) / (f2 - f1)),
                        coeffs=planner.parental_crossfeed_coeffs(
                            self.num_inference_steps, idx_injection, self.parental_crossfeed_power,
                            self.parental_crossfeed_range, self.parental_crossfeed_decay)))
                    est_l, est_r = -neg_est * r_l, -neg_est * r_r     # prediction until the child exists
                if blind_prior:
                    est_l, est_r = blind_estimate(fl, mid), blind_estimate(mid, fr)
                if self.speculate_virtual:
                    for a, b, e in ((fl, mid, est_l), (mid, fr, est_r)):
                        heapq.heappush(heap, (-float(e), tick, a, b, g))
                        tick += 1
            # 3) evaluate (batched on a native pipe, split over the ranks of a farm), then score
            if self._farm_on():
                own = self.farm.my_indices(len(specs))
                mine = self._evaluate_specs(specs, idx_injection, only=own)
                got = self.farm.exchange_branches(mine, len(specs), self.num_inference_steps - idx_injection,
                                                  self.num_inference_steps, self._frame_from_u8, self._latent_chw(),
                                                  (self.dh.height_img, self.dh.width_img))
                results = [(self._on_pipe_device(t), f) for t, f in got]
                for k, r in zip(own, mine):     # keep this rank's own frame objects
                    results[k] = (results[k][0], r[1])
            else:
                results = self._evaluate_specs(specs, idx_injection)
            frame_at = {f: tree.frames[i] for i, f in enumerate(tree.fracts)}
            frame_at.update({r["fract"]: r["frame"] for r in ready.values()})
            frame_at.update({s["fract"]: fr_ for s, (_, fr_) in zip(specs, results)})
            sims = self._gap_child_distances([(frame, frame_at[s["left"]], frame_at[s["right"]])
                                              for s, (_, frame) in zip(specs, results)],
                                             [(s["fract"], s["left"], s["right"]) for s in specs])
            for k, (s, (traj, frame)) in enumerate(zip(specs, results)):
                ready[(s["left"], s["right"])] = dict(fract=s["fract"], traj=traj, frame=frame,
                                                      sl=sims[k][0], sr=sims[k][1])
            self.stats["frontier_rounds"] = self.stats.get("frontier_rounds", 0) + 1
            self.stats["speculation_evaluated"] = self.stats.get("speculation_evaluated", 0) + len(specs)
        self.stats["speculation_dropped"] = self.stats.get("speculation_dropped", 0) + len(ready)
        # What the reference's loop leaves behind (blending_engine.py:358-362 of the reference: set_guidance_mid_dampening runs
        # right before every branch it commits, :155-164): the dampened scale of the LAST COMMITTED branch - not of the last
        # spec evaluated (the evaluation order of a batched round is best-first over real + virtual gaps, the commit order is
        # the greedy one).  The next transition's anchors are denoised under this value (compute_latents1 / 2, :370-423).
        if last_committed is not None:
            self.set_guidance_mid_dampening(last_committed)
        if len(committed_gaps) == stems and stems > 0:
            self._level_priors[(idx_injection, stems)] = committed_gaps

    def _gap_child_distances(self, triples, fracts):
        """[(child frame, left neighbour, right neighbour)], [(f_child, f_left, f_right)] -> [(d_left, d_right)].
        Under a farm the perceptual metric is SHARDED: every rank measures only the children it owns (features of
        the frames involved are computed on demand and cached) and the scalars are all-gathered, so all ranks
        decide on bit-identical numbers."""
        own = self.farm.my_indices(len(triples)) if self._farm_on() else list(range(len(triples)))
        flat, flat_f = [], []
        for k in own:
            child, left, right = triples[k]
            fc, fl, fr = fracts[k]
            flat += [(child, left), (child, right)]
            flat_f += [(fc, fl), (fc, fr)]
        sims = self._frame_distances(flat, flat_f)
        if not self._farm_on():
            return [(sims[2 * k], sims[2 * k + 1]) for k in range(len(triples))]
        got = self.farm.exchange_scalars([(sims[2 * j], sims[2 * j + 1]) for j in range(len(own))], len(triples))
        return [(g[0], g[1]) for g in got]

    def _latent_chw(self):
        return (int(self.dh.pipe.unet.config.in_channels), int(self.dh.height_latent), int(self.dh.width_latent))

    def _frame_distances(self, pairs, fracts=None):
        if self.pair_metric is not None:
            fracts = fracts or [(None, None)] * len(pairs)
            return [float(self.pair_metric(a, b, fa, fb)) for (a, b), (fa, fb) in zip(pairs, fracts)]
        pipe = self.dh.pipe
        if _is_native(pipe) and self.lpips is getattr(pipe, "lpips_metric", None):
            return pipe.native_frame_distances(pairs) if pairs else []
        return [self.get_lpips_similarity(a, b) for a, b in pairs]

    def _evaluate_specs(self, specs, idx_injection, only=None):
        """(trajectory, decoded frame) for every speculated gap child.  Native pipe: ONE batched
        denoising run + one batched decode.  Generic pipe: spec by spec through the diffusers-style API.
        ``only``: indices of the specs this rank evaluates (farm); the others are skipped, but a shared noise
        stream is advanced past them so that it stays aligned across ranks."""
        if not specs:
            return []
        pipe = self.dh.pipe
        chosen = list(range(len(specs))) if only is None else list(only)
        if _is_native(pipe):
            if not chosen:
                if getattr(pipe.scheduler, "ancestral", False):
                    self._skip_noise_draws(len(specs) * (self.num_inference_steps - idx_injection))
                return []
            sel = [specs[k] for k in chosen]
            trajs = pipe.native_run_diffusion_batch(
                [s["cond"] for s in sel], [s["mixed"][idx_injection - 1] for s in sel],
                idx_injection, [s["mixed"] for s in sel], [s["coeffs"] for s in sel],
                num_inference_steps=self.num_inference_steps,
                guidance_scales=[s["guidance"] for s in sel],
                noise_slots=None if only is None else (len(specs), chosen))
            frames = pipe.native_latent2image_batch([t[-1] for t in trajs], "pil")
            return list(zip(trajs, frames))
        out = []
        for k, s in enumerate(specs):
            if k not in chosen:
                self._skip_noise_draws(self.num_inference_steps - idx_injection)
                continue
            self.guidance_scale = self.dh.guidance_scale = s["guidance"]
            traj = self.run_diffusion([s["cond"]], latents_start=s["mixed"][idx_injection - 1], idx_start=idx_injection,
                                      list_latents_mixing=s["mixed"], mixing_coeffs=s["coeffs"])
            out.append((traj, self.dh.latent2image(traj[-1])))
        return out

    def _on_pipe_device(self, traj):
        """Exchanged latents live wherever the farm's collectives ran (host for gloo); a native pipe
        wants them in HBM."""
        if not _is_native(self.dh.pipe):
            return traj
        dev = self.dh.pipe.device
        return [None if t is None else t.to(dev) for t in traj]

    def _frame_from_u8(self, u8: torch.Tensor):
        """uint8 [H,W,3] tensor (as exchanged between ranks) -> the pipe's frame type."""
        if _is_native(self.dh.pipe):
            from .native.frames import DeviceImage
            return DeviceImage(u8)
        return Image.fromarray(u8.cpu().numpy())

    def _anchors_distributed(self, keep1=False, keep2=False):
        """Farm mode without the fused wavefront (generic pipes, crossfed or recycled anchors): every anchor that is
        not recycled is denoised on ONE owner rank and broadcast (C1 of SURVEY.md §8e), so all ranks continue from
        bit-identical stacks whatever their local RNG state is.  Anchor 1 lives on rank 0; anchor 2 on rank 1 when it
        can run concurrently (anchor 1 is being computed and is not crossfed into it), else on rank 0.  A recycled
        anchor is already identical everywhere (it was exchanged, or computed from exchanged data, last transition).
        Ranks that do not compute an anchor advance a shared noise stream past its draws."""
        farm, steps = self.farm, self.num_inference_steps
        independent = self.branch1_crossfeed_power == 0.0
        shape = (1,) + self._latent_chw()
        ancestral = bool(getattr(getattr(self.dh.pipe, "scheduler", None), "ancestral", False))
        first = self.tree_latents[0] if keep1 else None
        if not keep1:
            if farm.rank == 0:
                first = self.compute_latents1()
            elif ancestral:
                self._skip_noise_draws(steps)
            first = self._on_pipe_device(farm.share_trajectory(first if farm.rank == 0 else None, 0, steps, shape))
            self.tree_latents[0] = first
        last = self.tree_latents[-1] if keep2 else None
        if not keep2:
            owner2 = 1 % farm.world if (independent and not keep1) else 0
            if farm.rank == owner2:
                last = self.compute_latents2()
            elif ancestral:
                self._skip_noise_draws(steps)
            last = self._on_pipe_device(farm.share_trajectory(last if farm.rank == owner2 else None, owner2, steps, shape))
        self.tree_latents[0], self.tree_latents[-1] = first, last
        return first, last

    def _skip_noise_draws(self, n):
        sched = getattr(self.dh.pipe, "scheduler", None)
        src = getattr(sched, "noise_source", None)
        if src is not None and getattr(sched, "ancestral", False) and n > 0:
            shape = (1, self.dh.pipe.unet.config.in_channels, self.dh.height_latent, self.dh.width_latent)
            if hasattr(src, "many"):        # the ranks that DO draw make one generator call for these n latents (native pipe,
                src.many(n, shape)          # scheduler.draw_noise_many): the same call keeps a shared device stream aligned
                return
            for _ in range(n):
                src(shape)

    # ------------------------------------------------------------------ plumbing ----------
    def get_noise(self, seed):
        return self.dh.get_noise(seed)

    @torch.no_grad()
    def run_diffusion(self, list_conditionings, latents_start: torch.Tensor = None,
                      idx_start: int = 0, list_latents_mixing=None, mixing_coeffs=0.0,
                      return_image: Optional[bool] = False):
        """Pass-through to the holder; ``list_conditionings[0]`` is the embedding 4-tuple."""
        self.dh.set_num_inference_steps(self.num_inference_steps)
        assert type(list_conditionings) is list, "list_conditionings need to be a list"
        return self.dh.run_diffusion_sd_xl(
            text_embeddings=list_conditionings[0], latents_start=latents_start,
            idx_start=idx_start, list_latents_mixing=list_latents_mixing,
            mixing_coeffs=mixing_coeffs, return_image=return_image)

    @torch.no_grad()
    def get_mixed_conditioning(self, fract_mixing):
        """Lerp every non-None member of the two embedding tuples; returned wrapped in a list."""
        mixed = [None if a is None else interpolate_linear(a, b, fract_mixing)
                 for a, b in zip(self.text_embedding1, self.text_embedding2)]
        return [mixed]

    @torch.no_grad()
    def get_text_embeddings(self, prompt: str):
        return self.dh.get_text_embedding(prompt)

    # ------------------------------------------------------------------ output ------------
    def write_imgs_transition(self, dp_img):
        """Write the transition frames as ``lowres_img_XXXX.jpg`` into ``dp_img``."""
        os.makedirs(dp_img, exist_ok=True)
        for i, img in enumerate(self.tree_final_imgs):
            leaf = img if isinstance(img, Image.Image) else Image.fromarray(np.asarray(img))
            leaf.save(os.path.join(dp_img, f"lowres_img_{str(i).zfill(4)}.jpg"))

    def write_movie_transition(self, fp_movie, duration_transition, fps=30):
        """Linearly in-between the frames to ``duration_transition*fps`` frames and write a movie."""
        from .movie import MovieSaver, fill_up_frames_linear_interpolation
        frames = fill_up_frames_linear_interpolation(self.tree_final_imgs, duration_transition, fps)
        if os.path.isfile(fp_movie):
            os.remove(fp_movie)
        saver = MovieSaver(fp_movie, fps=fps, shape_hw=[self.dh.height_img, self.dh.width_img])
        for frame in tqdm(frames, disable=not self.verbose):
            saver.write_frame(frame)
        saver.finalize()

    def get_state_dict(self):
        """Scalar settings of the engine (the upstream list has a missing comma and names
        attributes that do not exist, blending_engine.py:711-715; fixed here)."""
        names = ['prompt1', 'prompt2', 'seed1', 'seed2', 'num_inference_steps', 'guidance_scale',
                 'guidance_scale_mid_damper', 'mid_compression_scaler', 'negative_prompt',
                 'branch1_crossfeed_power', 'branch1_crossfeed_range', 'branch1_crossfeed_decay',
                 'parental_crossfeed_power', 'parental_crossfeed_range', 'parental_crossfeed_decay']
        state = {}
        for name in names:
            if not hasattr(self, name):
                continue
            value = getattr(self, name)
            if name in ('seed1', 'seed2'):
                value = int(value)
            elif name == 'guidance_scale' or isinstance(value, (np.floating, np.integer)):
                value = float(value)
            state[name] = value
        state['width'] = self.dh.width_img
        state['height'] = self.dh.height_img
        # beyond the upstream list: what is needed to restore the engine (upstream never stores its branching plan; its
        # `depth_strength` key names an attribute that does not exist)
        state['guidance_scale_base'] = float(self.guidance_scale_base)
        state['list_idx_injection'] = [int(i) for i in self.list_idx_injection]
        state['list_nmb_stems'] = [int(s) for s in self.list_nmb_stems]
        return state

    def load_state_dict(self, state):
        """Inverse of ``get_state_dict`` (the reference has none: its UI re-enters every value by hand,
        gradio_ui.py:139-149): size, step count, guidance, negative prompt, prompts (re-embedded under that negative
        prompt and guidance), seeds, crossfeed settings and the branching plan.  A dict read back with ``yml_load``
        gives the same ``run_transition`` as the engine it was taken from."""
        if 'width' in state and 'height' in state:
            self.set_dimensions((int(state['width']), int(state['height'])))
        if 'num_inference_steps' in state:
            self.set_num_inference_steps(int(state['num_inference_steps']))
        for name in ('guidance_scale_mid_damper', 'mid_compression_scaler'):
            if name in state:
                setattr(self, name, float(state[name]))
        if 'guidance_scale_base' in state or 'guidance_scale' in state:
            self.set_guidance_scale(float(state.get('guidance_scale_base', state.get('guidance_scale'))))
        if 'negative_prompt' in state:          # (always applied when stored - also None: an engine / session that already has a
            neg = state['negative_prompt']      # negative prompt must not keep it and re-embed the prompts under it)
            if neg is None:                     # never set when the state was taken: the holder's default "" (diffusers_holder.py:23)
                self.negative_prompt = None
                self.dh.negative_prompt = ""
            else:
                self.set_negative_prompt(neg if isinstance(neg, str) else list(neg))
        if 'prompt1' in state:
            self.set_prompt1(state['prompt1'])
        if 'prompt2' in state:
            self.set_prompt2(state['prompt2'])
        if 'guidance_scale' in state:           # (the mid-dampened value the last branch left behind, as upstream stores it)
            self.guidance_scale = self.dh.guidance_scale = float(state['guidance_scale'])
        self.seed1, self.seed2 = int(state.get('seed1', self.seed1)), int(state.get('seed2', self.seed2))
        for name in ('branch1_crossfeed_power', 'branch1_crossfeed_range', 'branch1_crossfeed_decay',
                     'parental_crossfeed_power', 'parental_crossfeed_range', 'parental_crossfeed_decay'):
            if name in state:
                setattr(self, name, float(state[name]))
        if 'list_idx_injection' in state and 'list_nmb_stems' in state:
            self.list_idx_injection = [int(i) for i in state['list_idx_injection']]
            self.list_nmb_stems = [int(s) for s in state['list_nmb_stems']]

    def swap_forward(self):
        """Keyframe two becomes keyframe one (multi-transition chains)."""
        self.tree_latents[0] = self.tree_latents[-1]
        self.prompt1 = self.prompt2
        self.text_embedding1 = self.text_embedding2
        self.tree_final_imgs = []

    # ------------------------------------------------------------------ metric ------------
    def get_lpips_similarity(self, imgA, imgB):
        """Perceptual distance of two frames (high = dissimilar)."""
        if self.pair_metric is not None:
            return float(self.pair_metric(imgA, imgB, None, None))
        pipe = self.dh.pipe
        if _is_native(pipe) and self.lpips is getattr(pipe, "lpips_metric", None):
            return pipe.native_frame_distances([(imgA, imgB)])[0]

        def to_tensor(img):
            t = torch.from_numpy(np.array(img)).float()           # (copy: PIL buffers are read-only)
            t = t.cuda(self.device) if str(self.device).startswith("cuda") else t
            t = 2 * t / 255.0 - 1
            return t.permute([2, 0, 1]).unsqueeze(0)
        return float(self.lpips(to_tensor(imgA), to_tensor(imgB))[0][0][0][0])

    def get_tree_similarities(self):
        imgs = self.tree_final_imgs
        return [self.get_lpips_similarity(imgs[i], imgs[i + 1]) for i in range(len(imgs) - 1)]

    def get_closest_idx(self, fract_mixing: float):
        """Indices of the two committed branches enclosing ``fract_mixing``
        (e.g. 0.4 in [0, 0.3, 0.6, 1.0] -> (1, 2))."""
        return self._tree.neighbours(fract_mixing)
