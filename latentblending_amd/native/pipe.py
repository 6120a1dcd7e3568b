"""``NativeSDXLPipe`` — the SDXL pipeline object of the MI355X path.

It is two things at once:

1. a **duck-typed diffusers pipeline** (SURVEY.md Appendix A): every attribute the reference's
   ``DiffusersHolder`` touches exists with the same meaning (``unet(...)``, ``scheduler``,
   ``vae.decode``, ``encode_prompt``, ``prepare_latents``, ``image_processor`` ...), so the
   unchanged reference host code can drive it step by step.  The class is named
   ``StableDiffusionXLPipeline`` because the reference dispatches on that class name
   (/root/reference/latentblending/diffusers_holder.py:41);
2. the **native fast path** (``is_lb_native``) used by ``latentblending_amd.DiffusersHolder``: the
   whole restartable denoising loop for one branch or a BATCH of branches — crossfeed slerps,
   input scaling, UNet program, CFG + Euler(-ancestral) update — with device-resident state and
   one recorded launch program per UNet forward; VAE decode to uint8 frames on device; LPIPS on
   device frames with cached features.

Text encoders: CLIP weights/vocabularies are not available offline, so ``encode_prompt`` returns
seeded synthetic embeddings of the right shapes unless ``text_encoder_fn`` is supplied
(the step before the hot path).
"""
from __future__ import annotations

import os
import warnings
import zlib
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

from ..hip import ops
from ..hip.lib import api
from .frames import DeviceImage
from .lpips import NativeLPIPS
from .scheduler import NativeDDIMScheduler, NativeEulerScheduler
from .unet import NativeUNet, UNetConfig, UNetProgram
from .vae import NativeVAEDecoder, VAEConfig, VAEProgram
from .weights import SyntheticProvider

F16, F32 = torch.float16, torch.float32


def _synthetic_embedding(text: str, shape, salt: int) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(text.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g, dtype=torch.float32)


class _UNetFacade:
    def __init__(self, pipe):
        self._pipe = pipe
        c = pipe.unet_cfg
        self.config = SimpleNamespace(sample_size=c.sample_size, in_channels=c.in_channels,
                                      time_cond_proj_dim=c.time_cond_proj_dim)

    def __call__(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None,
                 cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False):
        pipe = self._pipe
        B, _, L, _ = sample.shape
        prog = pipe.unet_program(B, L)
        prog.set_conditioning(encoder_hidden_states, added_cond_kwargs["text_embeds"],
                              added_cond_kwargs["time_ids"])
        t = torch.as_tensor(timestep, dtype=F32).reshape(-1).expand(B)
        return (prog.forward(sample.to(F16), t.to(sample.device)).clone(),)


class _VAEFacade:
    def __init__(self, pipe):
        self._pipe = pipe
        c = pipe.vae_cfg
        self.dtype = F16
        self.config = SimpleNamespace(force_upcast=False, scaling_factor=c.scaling_factor)
        self.post_quant_conv = SimpleNamespace(parameters=lambda: iter([torch.zeros(1, dtype=F16)]))

    def decode(self, z, return_dict=False):
        """z = latents / scaling_factor -> float image [B,3,H,W] (diffusers convention)."""
        prog = self._pipe.vae_program(z.shape[0], z.shape[-1])
        prog.decode((z.float() * self._pipe.vae_cfg.scaling_factor).to(F16))
        return (prog.image_f32[..., :3].permute(0, 3, 1, 2).clone(),)

    def to(self, *a, **k):
        return self


class _ImageProcessor:
    @staticmethod
    def postprocess(image, output_type="pil"):
        x = image.permute(0, 2, 3, 1).contiguous()
        if x.shape[-1] < 4:
            x = torch.cat([x, torch.zeros_like(x[..., :1])], dim=-1).contiguous()
        if output_type == "np":     # diffusers' VaeImageProcessor: denormalise + clamp, float32 HWC, NOT quantised (diffusers_holder.py:141)
            return [f.cpu().numpy() for f in (x[..., :3].float() / 2 + 0.5).clamp(0, 1)]
        u8 = ops.postprocess_u8(x.float())
        return [DeviceImage(f) for f in u8]


class StableDiffusionXLPipeline:
    is_lb_native = True

    def __init__(self, turbo: bool = True, unet_cfg: Optional[UNetConfig] = None,
                 vae_cfg: Optional[VAEConfig] = None, unet_provider=None, vae_provider=None,
                 lpips_provider=None, device="cuda", seed: int = 0, name_or_path: Optional[str] = None,
                 text_encoder_fn=None, unet_native: Optional[NativeUNet] = None,
                 vae_native: Optional[NativeVAEDecoder] = None, allow_synthetic: Optional[bool] = None,
                 scheduler: Optional[str] = None):
        """``scheduler``: None or "euler" = the reference pipes' choice (Euler-ancestral for Turbo, Euler for base); "ddim" = diffusers'
        DDIMScheduler (eta 0) behind the same native loops - also with ``turbo=True`` (the Turbo guidance / branching defaults
        stay, only the sampler changes: deterministic, "leading" spacing).  Case-insensitive; anything else raises ``ValueError``
        (a misspelt "DDIM " or a scheduler object used to fall through to Euler silently).
        ``allow_synthetic``: seeded synthetic stand-ins (UNet / VAE / LPIPS weights when no provider is given,
        prompt embeddings when no ``text_encoder_fn`` is given) are used silently when True (tests, bench: also
        ``LB_ALLOW_SYNTHETIC=1``); otherwise each stand-in announces itself ONCE with a ``UserWarning`` - frames
        rendered from them are noise-like and prompts have no semantic effect."""
        if scheduler is not None and not isinstance(scheduler, str):
            raise ValueError(f"NativeSDXLPipe: scheduler must be None, 'euler' or 'ddim' (got an object of type {type(scheduler).__name__})")
        scheduler = None if scheduler is None else scheduler.strip().lower()
        if scheduler not in (None, "euler", "ddim"):
            raise ValueError(f"NativeSDXLPipe: unknown scheduler {scheduler!r} (None, 'euler' or 'ddim')")
        if not torch.cuda.is_available():
            raise RuntimeError("NativeSDXLPipe needs an MI355X (HIP device); there is no CPU fallback")
        self.device = torch.device(device)
        self._execution_device = self.device
        # (already packed modules may be shared between pipes: 5.5 GB of UNet weights need not be packed twice)
        self.unet_cfg = unet_native.cfg if unet_native is not None else (unet_cfg or UNetConfig(sample_size=64 if turbo else 128))
        self.vae_cfg = vae_native.cfg if vae_native is not None else (vae_cfg or VAEConfig())
        self._name_or_path = name_or_path or ("stabilityai/sdxl-turbo" if turbo else
                                              "stabilityai/stable-diffusion-xl-base-1.0")
        self.dtype = F16
        self.allow_synthetic = bool(allow_synthetic) if allow_synthetic is not None else os.environ.get("LB_ALLOW_SYNTHETIC") == "1"
        self._warned = set()
        if unet_native is None and unet_provider is None:
            self._synthetic_notice("UNet weights", "pass unet_provider=native.from_safetensors(<dir>/unet) or set LB_WEIGHTS_DIR")
        if vae_native is None and vae_provider is None:
            self._synthetic_notice("VAE weights", "pass vae_provider=native.from_safetensors(<dir>/vae) or set LB_WEIGHTS_DIR")
        if lpips_provider is None:
            self._synthetic_notice("LPIPS-Alex weights (the branch-insertion metric is then NOT LPIPS)",
                                   "pass lpips_provider=native.lpips_provider(alexnet_state_dict, lpips_lin_state_dict)")
        self.unet_native = unet_native or NativeUNet(self.unet_cfg, unet_provider or SyntheticProvider(seed), self.device)
        self.vae_native = vae_native or NativeVAEDecoder(self.vae_cfg, vae_provider or SyntheticProvider(seed + 1), self.device)
        self.lpips_metric = NativeLPIPS(lpips_provider or SyntheticProvider(7), self.device)
        # (the reference's SDXL pipes carry Euler / Euler-ancestral schedulers; scheduler="ddim" = diffusers' DDIMScheduler, eta 0)
        self.scheduler = NativeDDIMScheduler(device=self.device) if scheduler == "ddim" else NativeEulerScheduler(ancestral=turbo, device=self.device)
        self.unet = _UNetFacade(self)
        self.vae = _VAEFacade(self)
        self.image_processor = _ImageProcessor()
        self.vae_scale_factor = self.vae_cfg.scale_factor
        self.default_sample_size = self.unet_cfg.sample_size
        self.text_encoder_2 = None
        self.text_encoder_fn = text_encoder_fn
        self._guidance_scale = 0.0 if turbo else 5.0
        self._guidance_rescale = 0.0
        self._clip_skip = None
        self._cross_attention_kwargs = None
        self._denoising_end = None
        self._interrupt = False
        self._num_timesteps = 0
        # recorded launch programs, keyed by (batch, latent side); each owns its activation arena (+ hipGraph), so the
        # caches are LRU-bounded: speculative rounds of varying width must not pile up SDXL-sized arenas
        self._unet_programs: "OrderedDict[Tuple[int, int], UNetProgram]" = OrderedDict()
        self._vae_programs: "OrderedDict[Tuple[int, int], VAEProgram]" = OrderedDict()
        self.max_cached_programs = 8
        self._embed_cache: OrderedDict = OrderedDict()           # text -> (prompt_embeds, pooled); cleared when the encoder changes
        self._use_graphs = False
        self._feat_scratch: Dict[int, list] = {}
        self.stats = {"unet_forwards": 0, "unet_samples": 0, "vae_decodes": 0, "slerps": 0, "lpips_pairs": 0}
        self._side_stream = None            # lazily created: conditioning programs of big batches run here beside the small anchor steps

    def _synthetic_notice(self, what: str, how: str):
        if self.allow_synthetic or what in self._warned:
            return
        self._warned.add(what)
        warnings.warn(f"NativeSDXLPipe: using seeded SYNTHETIC {what} ({how}; allow_synthetic=True silences this)",
                      UserWarning, stacklevel=3)

    # ---- diffusers duck type ------------------------------------------------------------
    guidance_scale = property(lambda s: s._guidance_scale)
    guidance_rescale = property(lambda s: s._guidance_rescale)
    cross_attention_kwargs = property(lambda s: s._cross_attention_kwargs)

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet.config.time_cond_proj_dim is None

    def uses_cfg(self, guidance_scale: float) -> bool:
        return float(guidance_scale) > 1 and self.unet.config.time_cond_proj_dim is None

    def to(self, *_a, **_k):
        return self

    def upcast_vae(self):
        pass

    def prepare_extra_step_kwargs(self, generator, eta):
        return {"generator": generator}

    @property
    def text_encoder_fn(self):
        return self._text_encoder_fn

    @text_encoder_fn.setter
    def text_encoder_fn(self, fn):
        self._text_encoder_fn = fn
        cache = getattr(self, "_embed_cache", None)
        if cache is not None:
            cache.clear()                                 # embeddings of another encoder are not this one's

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt=1,
                      do_classifier_free_guidance=True, negative_prompt=None, negative_prompt_2=None, **_):
        c = self.unet_cfg

        def embed(text):
            # embedding cache keyed by the text: chained transitions re-encode the prompt they share
            hit = self._embed_cache.get(text)
            if hit is not None:
                self._embed_cache.move_to_end(text)
                return hit[0].clone(), hit[1].clone()
            if self.text_encoder_fn is not None:
                pe, pooled = self.text_encoder_fn(text)
            else:
                self._synthetic_notice("prompt embeddings (prompts have no semantic effect)",
                                       "pass text_encoder_fn=native.clip.NativeTextEncoders(...).encode")
                pe = _synthetic_embedding(text, (1, 77, c.cross_dim), 1)
                pooled = _synthetic_embedding(text, (1, c.pooled_dim), 2)
            pe, pooled = pe.to(self.device, F16), pooled.to(self.device, F16)
            self._embed_cache[text] = (pe.clone(), pooled.clone())
            while len(self._embed_cache) > 64:
                self._embed_cache.popitem(last=False)
            return pe, pooled

        text = prompt if isinstance(prompt, str) else prompt[0]
        pe, pooled = embed(text)
        if not do_classifier_free_guidance:
            return pe, None, pooled, None
        # diffusers' StableDiffusionXLPipeline.encode_prompt: zeros ONLY for ``negative_prompt is None`` (with the SDXL
        # checkpoints' force_zeros_for_empty_prompt); any given negative prompt - including the reference holder's default
        # "" (/root/reference/latentblending/diffusers_holder.py:23,87) - is tokenised and ENCODED
        if negative_prompt is None:
            return pe, torch.zeros_like(pe), pooled, torch.zeros_like(pooled)
        neg = negative_prompt[0] if isinstance(negative_prompt, (list, tuple)) else negative_prompt
        npe, npooled = embed(neg if neg is not None else "")
        return pe, npe, pooled, npooled

    def prepare_latents(self, batch, channels, height, width, dtype, device, generator, latents=None):
        """Seeded noise drawn on the HOST (so a seed means the same latent on every device and in
        the CPU oracle), scaled by the scheduler's initial sigma.  Drawn IN ``dtype`` like diffusers'
        ``randn_tensor(shape, generator, device, dtype)`` (reached from
        /root/reference/latentblending/diffusers_holder.py:98-111 with dtype = float16): on the CPU generator an fp16
        draw is a different stream from an fp32 draw that is cast afterwards."""
        shape = (batch, channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        host_gen = torch.Generator().manual_seed(generator.initial_seed())
        z = torch.randn(shape, generator=host_gen, dtype=dtype)
        return (z * self.scheduler.init_noise_sigma).to(self.device)

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype,
                          text_encoder_projection_dim=None):
        ids = list(original_size) + list(crops_coords_top_left) + list(target_size)
        return torch.tensor([ids], dtype=dtype, device=self.device)

    # ---- programs -----------------------------------------------------------------------
    def _cached(self, cache: OrderedDict, key, build):
        if key in cache:
            cache.move_to_end(key)
            return cache[key]
        while len(cache) >= max(1, self.max_cached_programs):
            torch.cuda.synchronize(self.device)          # (its launches may still be in flight on the stream)
            cache.popitem(last=False)                    # least recently used: arena, workspaces and graph go with it
        cache[key] = build()
        return cache[key]

    def unet_program(self, B: int, L: int) -> UNetProgram:
        def build():
            prog = self.unet_native.build(B, L)
            if self._use_graphs:
                prog.enable_graphs()
            return prog
        return self._cached(self._unet_programs, (B, L), build)

    def vae_program(self, B: int, L: int) -> VAEProgram:
        def build():
            prog = self.vae_native.build(B, L)
            if self._use_graphs:
                prog.prog.instantiate()
            return prog
        return self._cached(self._vae_programs, (B, L), build)

    def enable_graphs(self, flag: bool = True):
        self._use_graphs = bool(flag)
        if flag:
            for p in self._unet_programs.values():
                p.enable_graphs()
            for p in self._vae_programs.values():
                p.prog.instantiate()

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # ---- native fast path -----------------------------------------------------------------
    def _time_ids_row(self) -> List[float]:
        side = float(self.default_sample_size * self.vae_scale_factor)
        return [side, side, 0.0, 0.0, side, side]      # native size, not render size (reference quirk)

    def native_run_diffusion(self, text_embeddings, latents_start, idx_start, list_latents_mixing, coeffs,
                             num_inference_steps, guidance_scale):
        return self.native_run_diffusion_batch([text_embeddings], [latents_start], idx_start,
                                               [list_latents_mixing], [coeffs], num_inference_steps,
                                               [guidance_scale])[0]

    @torch.no_grad()
    def native_run_diffusion_batch(self, conds: Sequence[tuple], starts: Sequence[torch.Tensor], idx_start: int,
                                   mixings: Sequence[Optional[list]], coeffs_list: Sequence[Sequence[float]],
                                   num_inference_steps: int, guidance_scales: Sequence[float],
                                   noise_slots: Optional[Tuple[int, Sequence[int]]] = None):
        """Denoise G branches in lock-step from ``idx_start``.  Returns per branch a list with one
        entry per step (``None`` below ``idx_start``, else the [1,4,L,L] latent after that step).
        ``noise_slots`` = (n_total, my_indices): these G branches are numbers ``my_indices`` of a round of
        ``n_total`` branches evaluated by several ranks; ancestral noise is then drawn for all ``n_total`` in
        order and only this rank's share is used, so a shared noise stream stays aligned across ranks."""
        G = len(conds)
        sched = self.scheduler
        if sched.num_inference_steps != num_inference_steps:
            sched.set_timesteps(num_inference_steps)
        # classifier-free guidance is a per-branch predicate in the reference (guidance_scale > 1,
        # diffusers_holder.py:80,282): a batch mixing both kinds is run as two homogeneous sub-batches
        wants = [self.uses_cfg(g) for g in guidance_scales]
        if any(wants) and not all(wants):
            out: List[Optional[list]] = [None] * G
            assert noise_slots is None, "mixed CFG / non-CFG rounds are not supported under a farm"
            for flag in (False, True):
                idx = [g for g in range(G) if wants[g] == flag]
                part = self.native_run_diffusion_batch([conds[g] for g in idx], [starts[g] for g in idx], idx_start,
                                                       [mixings[g] for g in idx], [coeffs_list[g] for g in idx],
                                                       num_inference_steps, [guidance_scales[g] for g in idx])
                for g, traj in zip(idx, part):
                    out[g] = traj
            return out
        self._guidance_scale = float(guidance_scales[-1])
        cfg = wants[0]
        L = starts[0].shape[-1]
        per_sample = starts[0][0].numel()
        prog = self.unet_program(G * (2 if cfg else 1), L)

        pos_ctx = torch.cat([c[0] for c in conds]).to(self.device, F16)
        pos_pool = torch.cat([c[2] for c in conds]).to(self.device, F16)
        if cfg:
            neg_ctx = torch.cat([c[1] for c in conds]).to(self.device, F16)
            neg_pool = torch.cat([c[3] for c in conds]).to(self.device, F16)
            ctx, pooled = torch.cat([neg_ctx, pos_ctx]), torch.cat([neg_pool, pos_pool])
        else:
            ctx, pooled = pos_ctx, pos_pool
        time_ids = torch.tensor([self._time_ids_row()] * ctx.shape[0], dtype=F32, device=self.device)
        prog.set_conditioning(ctx, pooled, time_ids)

        latents = torch.cat([s.to(self.device, F16).reshape(1, -1, L, L) for s in starts]).contiguous()
        trajs: List[List[Optional[torch.Tensor]]] = [[None] * idx_start for _ in range(G)]
        stream = torch.cuda.current_stream().cuda_stream
        # every step's scalar coefficients in ONE upload: [steps][G][8]
        rows = [sched.step_row(i, float(guidance_scales[g])) for i in range(idx_start, num_inference_steps)
                for g in range(G)]
        params_all = ops.step_params(rows, self.device).view(-1, G, 8) if rows else None
        # ancestral noise is drawn sample-major (all steps of branch 0, then branch 1, ...): the same
        # order a sequential engine consumes a noise tape in, so batching does not change results
        noise_all = None
        if sched.ancestral and num_inference_steps > idx_start:
            shape1 = (1,) + tuple(latents.shape[1:])
            n_draw, keep = (G, list(range(G))) if noise_slots is None else (int(noise_slots[0]), list(noise_slots[1]))
            nm = num_inference_steps - idx_start
            drawn = sched.draw_noise_many(n_draw * nm, shape1, self.device).view(n_draw, nm, *shape1[1:])
            noise_all = drawn[keep].transpose(0, 1).contiguous()                             # [steps, G, 4, L, L]
        for i in range(idx_start, num_inference_steps):
            if i > 0:
                mix = [g for g in range(G) if coeffs_list[g][i] > 0]
                if mix:
                    outs = ops.slerp_pairs([latents[g:g + 1] for g in mix],
                                           [mixings[g][i - 1].to(self.device, F16) for g in mix],
                                           [float(coeffs_list[g][i]) for g in mix])
                    self.stats["slerps"] += len(mix)
                    if len(mix) == G:
                        latents = torch.cat(outs)
                    else:
                        latents = latents.clone()
                        for g, o in zip(mix, outs):
                            latents[g:g + 1] = o
            params = params_all[i - idx_start]
            api.lb_scale_model_input_f16(latents.data_ptr(), prog.x_in.data_ptr(), params.data_ptr(), per_sample, G,
                                         int(cfg), stream)
            prog.tvals.fill_(float(sched.timesteps_np[i]))
            prog.prog_step.launch(stream)
            self.stats["unet_forwards"] += 1
            self.stats["unet_samples"] += prog.B
            noise = noise_all[i - idx_start] if noise_all is not None else None
            latents = sched.device_step(latents, prog.eps, params, noise=noise, cfg=cfg)
            for g in range(G):
                trajs[g].append(latents[g:g + 1])
        return trajs

    @torch.no_grad()
    def native_run_wavefront(self, anchor_conds: Sequence[tuple], anchor_starts: Sequence[torch.Tensor],
                             mid_conds: Sequence[tuple], mid_fracts: Sequence[float],
                             mid_coeffs: Sequence[Sequence[float]], idx_injection: int, num_inference_steps: int,
                             guidance_anchor: float, guidance_mids: Sequence[float],
                             noise_slots: Optional[Tuple[int, Sequence[int]]] = None, elide_dead_steps: bool = False,
                             known_anchors: Sequence[Optional[Sequence[torch.Tensor]]] = (None, None)):
        """Both anchors AND a set of mid branches whose parents are the anchors, in one wavefront.

        A mid branch at step i only needs the anchors' latents of step i-1 (its start latent and its
        crossfeed targets are slerps of the anchors' previous-step latents, blending_engine.py:443-464
        of the reference), so from ``idx_injection`` on the anchors' step i and every mid branch's step
        i share ONE UNet batch of 2+G samples: 2 small + (steps-idx) large forwards instead of
        ``steps`` small + (steps-idx) large ones.  Arithmetic per sample is the same as in the
        separate runs.  Returns (trajectory anchor 1, trajectory anchor 2, [mid trajectories]).
        ``noise_slots`` = (n_total, my_indices) as in ``native_run_diffusion_batch`` (farm: every rank runs both
        anchors plus its own share of the round's mid branches).
        ``elide_dead_steps`` (opt-in): a mid step whose result the NEXT step's crossfeed overwrites
        completely (crossfeed coefficient exactly 1.0 - the SDXL-Turbo defaults, reference blending_engine.py:193-199,
        452-457 and diffusers_holder.py:322-324: slerp(x, target, 1.0) == target) is not computed: that step runs the
        anchors only, the mids' trajectory entry is ``None``.  Frames and final latents are bit-identical (noise draws
        are still consumed in the same order); the reference itself performs the dead forward, so the default is off.
        ``known_anchors[k]`` = a finished trajectory of anchor k (a recycled key frame, blending_engine.py:333-342 of the
        reference: ``recycle_img1`` after ``swap_forward``): that anchor is not denoised again - the small batches carry only
        the other anchor, the mids mix from the stored latents - and is returned as given."""
        known = [None if t is None else list(t) for t in known_anchors]
        live = [k for k in (0, 1) if known[k] is None]          # anchors denoised here
        # ``mid_conds`` may be a CALLABLE returning the list (round 6): the mids' conditionings (two lerp launches each) are then built -
        # and the big batch's conditioning program launched - AFTER the first anchors-only step is on the stream, i.e. the host work runs
        # beside 11 ms of GPU work instead of in front of it (a cfg-2 transition spent ~2 ms of host time before its first launch)
        lazy_mids = callable(mid_conds)
        A, G = len(live), len(mid_fracts)
        assert lazy_mids or len(mid_conds) == G
        assert A + G > 0, "native_run_wavefront: nothing to denoise"
        assert all(t is None or len(t) == num_inference_steps for t in known), "known anchors must be full trajectories"
        anchor_conds = [anchor_conds[k] for k in live]
        sched, steps = self.scheduler, num_inference_steps
        if sched.num_inference_steps != steps:
            sched.set_timesteps(steps)
        all_g = [float(guidance_anchor)] * A + [float(g) for g in guidance_mids]
        self._guidance_scale = all_g[-1]
        cfg = self.uses_cfg(all_g[0])
        ref_start = anchor_starts[0]
        anchor_starts = [anchor_starts[k] for k in live]
        assert all(self.uses_cfg(g) == cfg for g in all_g), "wavefront batches must be uniformly CFG or non-CFG"
        mul = 2 if cfg else 1
        L = ref_start.shape[-1]
        per_sample = ref_start[0].numel()

        def conditioning(conds):
            pos_ctx = torch.cat([c[0] for c in conds]).to(self.device, F16)
            pos_pool = torch.cat([c[2] for c in conds]).to(self.device, F16)
            if not cfg:
                return pos_ctx, pos_pool
            neg_ctx = torch.cat([c[1] for c in conds]).to(self.device, F16)
            neg_pool = torch.cat([c[3] for c in conds]).to(self.device, F16)
            return torch.cat([neg_ctx, pos_ctx]), torch.cat([neg_pool, pos_pool])

        def prepared(conds, side=None, after=None):
            # (the program - arena, workspaces, first-time graph instantiation - is always obtained on the MAIN stream: its
            # allocations then belong to the main stream's allocator pool; only the conditioning launches move to `side`)
            # `conds` may be a callable (evaluated on the stream the conditioning launches run on); `after` = an event of the main
            # stream the side stream waits for INSTEAD of everything queued on the main stream so far (the deferred form: the first
            # small UNet step is already queued - waiting for it would also park the host in the side stream's host-to-device copies)
            n_conds = A + G if callable(conds) else len(conds)
            prog = self.unet_program(n_conds * mul, L)
            if side is None:
                conds = conds() if callable(conds) else conds
                ctx, pooled = conditioning(conds)
                ids = torch.tensor([self._time_ids_row()] * ctx.shape[0], dtype=F32, device=self.device)
                prog.set_conditioning(ctx, pooled, ids)
                return prog, None
            if after is not None:
                side.wait_event(after)
            else:
                side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                conds = conds() if callable(conds) else conds
                ctx, pooled = conditioning(conds)
                ids = torch.tensor([self._time_ids_row()] * ctx.shape[0], dtype=F32, device=self.device)
                prog.set_conditioning(ctx, pooled, ids)     # (copies into program-owned buffers, then the conditioning program: all
                return prog, side.record_event()            # on `side` - the temporaries above never meet another stream)

        # G == 0: a farm rank that owns no mid branch of the round (fewer gaps than ranks) still runs both anchors
        dead = [bool(elide_dead_steps) and G > 0 and i >= idx_injection and i + 1 < steps and
                all(float(mid_coeffs[g][i + 1]) == 1.0 for g in range(G)) for i in range(steps)]
        prog_a = prepared(list(anchor_conds))[0] if A and (idx_injection > 0 or G == 0 or any(dead)) else None
        # The big batch's conditioning program (every branch's context K | V projection: ~1.1 ms at 17 samples) does not depend
        # on any latent: when small anchor-only steps come first it runs on a SIDE stream beside them (they leave most of the
        # chip idle) and the main stream waits for it only before the first big step.  Different programs own different
        # arenas / workspaces, so the two streams share nothing but read-only weights.
        cond_ready = None
        deferred = False
        if G and prog_a is not None and idx_injection > 0:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            if lazy_mids and A:
                deferred, prog_all = True, None         # (prepared right behind the first small step's launch, below)
                before_first = torch.cuda.Event()
                before_first.record()                   # everything the mids' conditionings read (the prompt embeddings) is older than this
            else:
                prog_all, cond_ready = prepared(list(anchor_conds) + list(mid_conds() if lazy_mids else mid_conds), side=self._side_stream)
        else:
            prog_all = prepared(list(anchor_conds) + list(mid_conds() if lazy_mids else mid_conds))[0] if G else prog_a
        stream = torch.cuda.current_stream().cuda_stream
        rows_a = [sched.step_row(i, all_g[0]) for i in range(steps)]
        par_a = ops.step_params([r for r in rows_a for _ in range(A)], self.device).view(steps, A, 8) if A else None
        par_all = ops.step_params([sched.step_row(i, all_g[s]) for i in range(idx_injection, steps)
                                   for s in range(A + G)], self.device).view(-1, A + G, 8) if G else None
        shape1 = (1,) + tuple(ref_start.shape[1:])
        noise_a = noise_m = None
        if sched.ancestral:       # sample-major draws: anchor 1, anchor 2 (those denoised here), then every mid branch
            n_draw, keep = (G, list(range(G))) if noise_slots is None else (int(noise_slots[0]), list(noise_slots[1]))
            nm = steps - idx_injection
            flat = sched.draw_noise_many(A * steps + n_draw * nm, shape1, self.device)      # ONE launch on the device RNG
            # (contiguous per step: the step kernels take noise[i] by pointer)
            noise_a = flat[:A * steps].view(A, steps, *shape1[1:]).transpose(0, 1).contiguous() if A else None          # [steps, A, 4, L, L]
            noise_m = flat[A * steps:].view(n_draw, nm, *shape1[1:])[keep].transpose(0, 1).contiguous() if G else None   # [nm, G, 4, L, L]
        lat_shape = (int(ref_start.shape[-3]), L, L)
        lat_a = torch.cat([s.to(self.device, F16).reshape(1, -1, L, L) for s in anchor_starts]).contiguous() if A else \
            torch.empty((0,) + lat_shape, dtype=F16, device=self.device)
        lat_m = None
        traj_a = [[] if known[k] is None else [t.to(self.device, F16).reshape(1, -1, L, L) for t in known[k]] for k in (0, 1)]
        traj_m: List[List[Optional[torch.Tensor]]] = [[None] * idx_injection for _ in range(G)]
        # mixing fractions / crossfeed coefficients live on the device: every step's parental mix (ONE pair of anchor
        # latents at G fractions) and crossfeed (G pairs) is one strided-slerp launch, no host pointer tables
        n_lat = per_sample
        fr_dev = torch.tensor([float(f) for f in mid_fracts], dtype=torch.float64, device=self.device) if G else None
        coef_dev = torch.tensor([[float(mid_coeffs[g][i]) for g in range(G)] for i in range(steps)],
                                dtype=torch.float64, device=self.device) if G else None
        try:
            for i in range(steps):
                if i < idx_injection or G == 0 or dead[i]:
                    if dead[i] and i == idx_injection:
                        # the mids' (never denoised) start value: the parental mix of step i-1, exactly what the live path starts
                        # from - the next step's crossfeed slerp at coefficient 1.0 replaces it bit for bit, but its FIRST operand
                        # must be a proper latent (a zero tensor has no direction: 0 / 0 in the slerp's cosine)
                        lat_m = ops.slerp_strided(traj_a[0][i - 1].contiguous(), traj_a[1][i - 1].contiguous(), fr_dev, n_lat,
                                                  broadcast0=True, broadcast1=True).view(G, *lat_shape)
                        self.stats["slerps"] += G
                    if A == 0:                  # both anchors known: nothing runs before the injection step (or in a dead one)
                        if i >= idx_injection and G and dead[i]:
                            for g in range(G):
                                traj_m[g].append(None)
                        continue
                    prog, lat, params, n = prog_a, lat_a, par_a[i], A
                    noise = noise_a[i] if noise_a is not None else None
                else:
                    prev1, prev2 = traj_a[0][i - 1].contiguous(), traj_a[1][i - 1].contiguous()
                    mix_prev = ops.slerp_strided(prev1, prev2, fr_dev, n_lat, broadcast0=True, broadcast1=True)   # parental mix of step i-1
                    self.stats["slerps"] += G
                    if i == idx_injection:
                        lat_m = mix_prev.view(G, *lat_shape)
                    elif dead[i - 1]:
                        assert all(float(mid_coeffs[g][i]) == 1.0 for g in range(G))
                    nfeed = sum(1 for g in range(G) if mid_coeffs[g][i] > 0)
                    if nfeed:       # (a coefficient of 0 returns the first operand bit-exactly, like the reference's skipped slerp)
                        lat_m = ops.slerp_strided(lat_m.contiguous().view(G, n_lat), mix_prev, coef_dev[i], n_lat).view(G, *lat_shape)
                        self.stats["slerps"] += nfeed
                    prog, lat, params, n = prog_all, (torch.cat([lat_a, lat_m]) if A else lat_m.contiguous()), par_all[i - idx_injection], A + G
                    if cond_ready is not None:
                        torch.cuda.current_stream().wait_event(cond_ready)
                        cond_ready = None
                    noise = None
                    if noise_m is not None:
                        noise = torch.cat([noise_a[i], noise_m[i - idx_injection]]) if A else noise_m[i - idx_injection]
                api.lb_scale_model_input_f16(lat.data_ptr(), prog.x_in.data_ptr(), params.data_ptr(), per_sample, n,
                                             int(cfg), stream)
                prog.tvals.fill_(float(sched.timesteps_np[i]))
                prog.prog_step.launch(stream)
                self.stats["unet_forwards"] += 1
                self.stats["unet_samples"] += prog.B
                out = sched.device_step(lat, prog.eps, params, noise=noise, cfg=cfg)
                if deferred:            # the first small step is launched: now the host work the big batch needs (runs beside it)
                    deferred = False
                    prog_all, cond_ready = prepared(lambda: list(anchor_conds) + list(mid_conds()), side=self._side_stream, after=before_first)
                lat_a = out[:A]
                for j, k in enumerate(live):
                    traj_a[k].append(out[j:j + 1])
                if i >= idx_injection and G and dead[i]:
                    for g in range(G):
                        traj_m[g].append(None)
                elif i >= idx_injection and G:
                    lat_m = out[A:]
                    for g in range(G):
                        traj_m[g].append(out[A + g:A + g + 1])
        finally:
            if cond_ready is not None:      # (an exception before the first big step: never leave the side stream's conditioning
                torch.cuda.current_stream().wait_event(cond_ready)    # launches un-joined - a later call reuses the cached program)
        return traj_a[0], traj_a[1], traj_m

    @torch.no_grad()
    def native_latent2image_batch(self, latents: Sequence[torch.Tensor], output_type="pil"):
        z = torch.cat([t.to(self.device, F16).reshape(1, -1, t.shape[-2], t.shape[-1]) for t in latents])
        prog = self.vae_program(z.shape[0], z.shape[-1])
        frames = prog.decode(z).clone()
        self.stats["vae_decodes"] += z.shape[0]
        if output_type == "np":     # the reference's postprocess(..., "np"): (x / 2 + 0.5).clamp(0, 1) as float32 HWC, unquantised
            img = (prog.image_f32[..., :3].float() / 2 + 0.5).clamp(0, 1)
            return [f.cpu().numpy() for f in img]
        return [DeviceImage(f) for f in frames]

    def native_latent2image(self, latents, output_type="pil"):
        return self.native_latent2image_batch([latents], output_type)[0]

    @torch.no_grad()
    def native_frame_distances(self, pairs) -> List[float]:
        """LPIPS distance for (frameA, frameB) pairs of ``DeviceImage`` / PIL / numpy frames."""
        frames = []
        for a, b in pairs:
            frames += [a, b]
        missing, seen = [], set()
        for f in frames:
            if getattr(f, "_lb_feats", None) is None and id(f) not in seen:
                seen.add(id(f))
                missing.append(f)
        if missing:
            u8 = torch.stack([self._frame_u8(f) for f in missing])
            taps = self.lpips_metric.features(u8)
            for n, f in enumerate(missing):
                feats = [t[n] for t in taps]
                try:
                    f._lb_feats = feats
                except AttributeError:
                    pass
                self._feat_scratch[id(f)] = feats
        feat_of = lambda f: getattr(f, "_lb_feats", None) or self._feat_scratch[id(f)]
        d = self.lpips_metric.distances([(feat_of(a), feat_of(b)) for a, b in pairs])
        self.stats["lpips_pairs"] += len(pairs)
        self._feat_scratch.clear()
        return [float(v) for v in d.tolist()]

    def _frame_u8(self, f) -> torch.Tensor:
        if isinstance(f, DeviceImage):
            return f._lb_u8.to(self.device)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(f, dtype=np.uint8))).to(self.device)


NativeSDXLPipe = StableDiffusionXLPipeline
