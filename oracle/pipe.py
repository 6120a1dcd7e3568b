"""ORACLE — test infrastructure, NOT product code.

A CPU pipeline object with exactly the duck type the reference's ``DiffusersHolder`` drives
(every attribute touched in
/root/reference/latentblending/diffusers_holder.py:29-366), backed by the fp32 restatement in
``oracle/sdxl_ref.py`` with seeded synthetic weights.  It serves three purposes:

* the pipe that the UNCHANGED reference host classes drive in differential tests
  (``oracle/ref_harness.py``),
* the CPU checker for the gfx950 path (same weights, same synthetic conditioning, same noise),
* the ``cpu_baseline`` leg of ``bench.py``.

The class is deliberately named ``StableDiffusionXLPipeline``: the reference dispatches on
``pipe.__class__.__name__`` (diffusers_holder.py:41).
"""
from __future__ import annotations

import zlib
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch
from PIL import Image

from . import sdxl_ref as R


def synthetic_embedding(text: str, shape, dtype, salt: int = 0) -> torch.Tensor:
    """Deterministic stand-in for CLIP: N(0,1) seeded by the prompt string."""
    g = torch.Generator().manual_seed((zlib.crc32(text.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)


class NoiseTape:
    """Ancestral noise in draw order from a seeded CPU generator (the reference draws from the
    global device RNG, diffusers_holder.py:192,255 — not reproducible; both sides of a parity
    test consume an identically seeded tape instead)."""

    def __init__(self, seed: int = 12345, dtype=torch.float16):
        self.seed, self.dtype = seed, dtype
        self.reset()

    def reset(self):
        self.gen = torch.Generator().manual_seed(self.seed)
        self.draws = 0

    def __call__(self, shape):
        self.draws += 1
        return torch.randn(shape, generator=self.gen, dtype=torch.float32).to(self.dtype)


class _UNet:
    def __init__(self, cfg: R.UNetCfg, weights: Dict[str, torch.Tensor]):
        self.cfg, self.w = cfg, weights
        self.config = SimpleNamespace(sample_size=cfg.sample_size, in_channels=cfg.in_channels,
                                      time_cond_proj_dim=cfg.time_cond_proj_dim)
        self.calls = 0

    def __call__(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None,
                 cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False):
        self.calls += 1
        out = R.unet_forward(self.cfg, self.w, sample, timestep, encoder_hidden_states,
                             added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"])
        return (out.to(sample.dtype),)


class _VAE:
    def __init__(self, cfg: R.VAECfg, weights, dtype):
        self.cfg, self.w = cfg, weights
        self.dtype = dtype
        self.config = SimpleNamespace(force_upcast=cfg.force_upcast, scaling_factor=cfg.scaling_factor)
        self.post_quant_conv = SimpleNamespace(
            parameters=lambda: iter([torch.zeros(1, dtype=self.dtype)]))
        self.calls = 0

    def decode(self, z, return_dict=False):
        self.calls += 1
        return (R.vae_decode(self.cfg, self.w, z),)

    def to(self, dtype=None, **_):
        if dtype is not None:
            self.dtype = dtype
        return self


class _ImageProcessor:
    @staticmethod
    def postprocess(image, output_type="pil"):
        if output_type == "np":     # diffusers VaeImageProcessor.postprocess(..., "np"): denormalised, clamped, float32 HWC - not quantised
            return [a for a in (image.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous().numpy()]
        arr = R.postprocess_u8(image)
        return [Image.fromarray(a) for a in arr]


class StableDiffusionXLPipeline:
    def __init__(self, turbo: bool = True, unet_cfg: Optional[R.UNetCfg] = None,
                 vae_cfg: Optional[R.VAECfg] = None, seed: int = 0, dtype=torch.float16,
                 noise_seed: int = 12345, weights=None, vae_weights=None):
        self.unet_cfg = unet_cfg or R.UNetCfg(sample_size=64 if turbo else 128)
        self.vae_cfg = vae_cfg or R.VAECfg()
        self._name_or_path = "stabilityai/sdxl-turbo" if turbo else "stabilityai/stable-diffusion-xl-base-1.0"
        self._execution_device = torch.device("cpu")
        self.dtype = dtype
        self.noise = NoiseTape(noise_seed)
        self.unet = _UNet(self.unet_cfg, weights if weights is not None
                          else R.make_weights(R.unet_spec(self.unet_cfg), seed))
        self.vae = _VAE(self.vae_cfg, vae_weights if vae_weights is not None
                        else R.make_weights(R.vae_decoder_spec(self.vae_cfg), seed + 1), dtype)
        self.scheduler = R.EulerScheduler(ancestral=turbo, noise_source=self.noise)
        self.vae_scale_factor = 2 ** (len(self.vae_cfg.block_channels) - 1)
        self.default_sample_size = self.unet_cfg.sample_size
        self.image_processor = _ImageProcessor()
        self.text_encoder_2 = None
        self._guidance_scale = 0.0 if turbo else 5.0
        self._guidance_rescale = 0.0
        self._clip_skip = None
        self._cross_attention_kwargs = None
        self._denoising_end = None
        self._interrupt = False
        self._num_timesteps = 0
        self.encode_calls = 0

    # properties the loop reads ------------------------------------------------------------
    guidance_scale = property(lambda s: s._guidance_scale)
    guidance_rescale = property(lambda s: s._guidance_rescale)
    cross_attention_kwargs = property(lambda s: s._cross_attention_kwargs)

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet.config.time_cond_proj_dim is None

    def to(self, *_a, **_k):
        return self

    def upcast_vae(self):
        self.vae.to(dtype=torch.float32)

    def prepare_extra_step_kwargs(self, generator, eta):
        return {"generator": generator}

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt=1,
                      do_classifier_free_guidance=True, negative_prompt=None,
                      negative_prompt_2=None, **_):
        self.encode_calls += 1
        c = self.unet_cfg
        text = prompt if isinstance(prompt, str) else prompt[0]
        pe = synthetic_embedding(text, (1, 77, c.cross_dim), self.dtype, 1)
        pooled = synthetic_embedding(text, (1, c.pooled_dim), self.dtype, 2)
        if not do_classifier_free_guidance:
            return pe, None, pooled, None
        # diffusers semantics: zero embeddings only when negative_prompt is None (force_zeros_for_empty_prompt); a given
        # negative prompt, "" included (the reference holder's default), is encoded like any other text
        if negative_prompt is None:
            return pe, torch.zeros_like(pe), pooled, torch.zeros_like(pooled)
        neg = negative_prompt[0] if isinstance(negative_prompt, (list, tuple)) else negative_prompt
        neg = "" if neg is None else neg
        npe = synthetic_embedding(neg, (1, 77, c.cross_dim), self.dtype, 1)
        npooled = synthetic_embedding(neg, (1, c.pooled_dim), self.dtype, 2)
        return pe, npe, pooled, npooled

    def prepare_latents(self, batch, channels, height, width, dtype, device, generator, latents=None):
        shape = (batch, channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        cpu_gen = torch.Generator().manual_seed(generator.initial_seed())
        # diffusers' randn_tensor draws in the requested dtype (fp16 from the reference holder, diffusers_holder.py:98-111):
        # NOT an fp32 draw cast afterwards - that is a different stream on the CPU generator
        z = torch.randn(shape, generator=cpu_gen, dtype=dtype)
        return z * self.scheduler.init_noise_sigma

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype,
                          text_encoder_projection_dim=None):
        ids = list(original_size) + list(crops_coords_top_left) + list(target_size)
        return torch.tensor([ids], dtype=dtype)
