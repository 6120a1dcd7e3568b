"""``DiffusersHolder`` — the diffusion runner behind the blending engine.

Same public surface as the reference class (``latentblending/diffusers_holder.py:20-366`` in
/root/reference): constructor, ``set_num_inference_steps``, ``set_dimensions``,
``set_negative_prompt``, ``get_text_embedding``, ``get_noise``, ``latent2image``,
``prepare_mixing``, ``run_diffusion`` and ``run_diffusion_sd_xl`` with identical argument
order, defaults and return types.

Two execution paths behind that surface:

* **native** — the pipe is a ``NativeSDXLPipe`` (``pipe.is_lb_native``): the restartable
  denoising loop (skip ``i < idx_start``, crossfeed slerp, scale, UNet, CFG, Euler step) runs as
  hand-written gfx950 kernels on device-resident state; this module only forwards arguments.
* **generic** — any object with the SDXL pipeline duck type, e.g. a real
  diffusers ``StableDiffusionXLPipeline`` or the CPU oracle pipe in ``oracle/``.  The loop below
  drives ``pipe.unet`` / ``pipe.scheduler`` step by step exactly like the reference and takes the
  crossfeed slerp from the active mixing backend.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .backend import get_backend


def _is_native(pipe) -> bool:
    return bool(getattr(pipe, "is_lb_native", False))


class DiffusersHolder:
    def __init__(self, pipe):
        self.negative_prompt = ""
        self.guidance_scale = 5.0
        self.num_inference_steps = 30

        self.pipe = pipe
        self.device = str(pipe._execution_device)
        self.init_types()

        latent_side = self.pipe.unet.config.sample_size
        self.width_latent = latent_side
        self.height_latent = latent_side
        self.width_img = latent_side * self.pipe.vae_scale_factor
        self.height_img = latent_side * self.pipe.vae_scale_factor

    # ---------------------------------------------------------------- configuration -------
    def init_types(self):
        cls = getattr(self.pipe, "__class__", None)
        assert cls is not None and hasattr(cls, "__name__"), "No valid diffusers pipeline found."
        if cls.__name__ == 'StableDiffusionXLPipeline':
            self.pipe.scheduler.set_timesteps(self.num_inference_steps, device=self.device)
            probe = self.pipe.encode_prompt("test")[0]
        else:
            probe = self.pipe._encode_prompt("test", self.device, 1, True)
        self.dtype = probe.dtype
        self.is_sdxl_turbo = 'turbo' in self.pipe._name_or_path

    def set_num_inference_steps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        self.pipe.scheduler.set_timesteps(self.num_inference_steps, device=self.device)

    def set_dimensions(self, size_output):
        """``size_output`` = (width, height); snapped to a multiple of the VAE scale factor."""
        scale = self.pipe.vae_scale_factor
        if size_output is None:
            width = height = self.pipe.unet.config.sample_size
        else:
            width, height = size_output
        self.width_img = int(round(width / scale) * scale)
        self.width_latent = int(self.width_img / scale)
        self.height_img = int(round(height / scale) * scale)
        self.height_latent = int(self.height_img / scale)
        print(f"set_dimensions to width={width} and height={height}")

    def set_negative_prompt(self, negative_prompt):
        """Only one negative prompt is supported: a string or the first entry of a list."""
        as_list = [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
        self.negative_prompt = as_list[:1] if len(as_list) > 1 else as_list

    # ---------------------------------------------------------------- pipeline helpers ----
    def get_text_embedding(self, prompt):
        """4-tuple (prompt_embeds, negative_prompt_embeds, pooled, negative_pooled)."""
        wants_cfg = self.guidance_scale > 1 and self.pipe.unet.config.time_cond_proj_dim is None
        return self.pipe.encode_prompt(
            prompt=prompt, prompt_2=prompt, device=self.pipe._execution_device,
            num_images_per_prompt=1, do_classifier_free_guidance=wants_cfg,
            negative_prompt=self.negative_prompt, negative_prompt_2=self.negative_prompt,
            prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
            negative_pooled_prompt_embeds=None, lora_scale=None, clip_skip=None)

    def get_noise(self, seed=420):
        """Initial latent for ``seed``: always fp16, scaled by the scheduler's init sigma
        (the reference hard-codes float16 here, ``diffusers_holder.py:105``)."""
        generator = torch.Generator(device=self.device).manual_seed(int(seed))
        return self.pipe.prepare_latents(
            1, self.pipe.unet.config.in_channels, self.height_img, self.width_img,
            torch.float16, self.pipe._execution_device, generator, None)

    @torch.no_grad()
    def latent2image(self, latents: torch.Tensor, output_type="pil"):
        """Decode a final latent to an image ("pil" or "np")."""
        assert output_type in ["pil", "np"]
        if _is_native(self.pipe):
            return self.pipe.native_latent2image(latents, output_type)

        vae = self.pipe.vae
        upcast = vae.dtype == torch.float16 and vae.config.force_upcast
        if upcast:
            self.pipe.upcast_vae()
            latents = latents.to(next(iter(vae.post_quant_conv.parameters())).dtype)
        image = vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]
        if upcast:
            vae.to(dtype=torch.float16)
        return self.pipe.image_processor.postprocess(image, output_type=output_type)[0]

    def prepare_mixing(self, mixing_coeffs, list_latents_mixing):
        steps = self.num_inference_steps
        if type(mixing_coeffs) == float:
            coeffs = (1 + steps) * [mixing_coeffs]
        elif type(mixing_coeffs) == list:
            assert len(mixing_coeffs) == steps, \
                f"len(mixing_coeffs) {len(mixing_coeffs)} != self.num_inference_steps {steps}"
            coeffs = mixing_coeffs
        else:
            raise ValueError("mixing_coeffs should be float or list with len=num_inference_steps")
        if np.sum(coeffs) > 0:
            assert len(list_latents_mixing) == steps, \
                f"len(list_latents_mixing) {len(list_latents_mixing)} != self.num_inference_steps {steps}"
        return coeffs

    # ---------------------------------------------------------------- diffusion -----------
    @torch.no_grad()
    def run_diffusion(self, text_embeddings, latents_start, idx_start: int = 0,
                      list_latents_mixing=None, mixing_coeffs=0.0,
                      return_image: Optional[bool] = False):
        return self.run_diffusion_sd_xl(text_embeddings, latents_start, idx_start,
                                        list_latents_mixing, mixing_coeffs, return_image)

    @torch.no_grad()
    def run_diffusion_sd_xl(self, text_embeddings: tuple, latents_start: torch.Tensor,
                            idx_start: int = 0, list_latents_mixing=None, mixing_coeffs=0.0,
                            return_image: Optional[bool] = False):
        """Denoise from step ``idx_start``; returns a list with one entry per step (``None`` for
        skipped steps, otherwise the latent after that step) or the decoded image."""
        coeffs = self.prepare_mixing(mixing_coeffs, list_latents_mixing)
        if _is_native(self.pipe):
            trajectory = self.pipe.native_run_diffusion(
                text_embeddings, latents_start, idx_start, list_latents_mixing, coeffs,
                num_inference_steps=self.num_inference_steps, guidance_scale=self.guidance_scale)
        else:
            trajectory = self._denoise_generic(text_embeddings, latents_start, idx_start,
                                               list_latents_mixing, coeffs)
        if return_image:
            return self.latent2image(trajectory[-1])
        return trajectory

    # generic duck-typed pipe ---------------------------------------------------------------
    def _conditioning_generic(self, text_embeddings):
        pipe = self.pipe
        prompt_embeds, neg_prompt_embeds, pooled, neg_pooled = text_embeddings
        # The micro-conditioning always describes the UNet's native size, not the render size
        # (reference quirk, diffusers_holder.py:216-220).
        side = pipe.default_sample_size * pipe.vae_scale_factor
        if pipe.text_encoder_2 is None:
            proj_dim = int(pooled.shape[-1])
        else:
            proj_dim = pipe.text_encoder_2.config.projection_dim
        time_ids = pipe._get_add_time_ids((side, side), (0, 0), (side, side),
                                          dtype=prompt_embeds.dtype,
                                          text_encoder_projection_dim=proj_dim)
        text_embeds = pooled
        if pipe.do_classifier_free_guidance:
            prompt_embeds = torch.cat([neg_prompt_embeds, prompt_embeds], dim=0)
            text_embeds = torch.cat([neg_pooled, text_embeds], dim=0)
            time_ids = torch.cat([time_ids, time_ids], dim=0)
        device = pipe._execution_device
        return prompt_embeds.to(device), text_embeds.to(device), time_ids.to(device).repeat(1, 1)

    def _denoise_generic(self, text_embeddings, latents_start, idx_start, list_latents_mixing,
                         coeffs):
        pipe = self.pipe
        slerp = get_backend().slerp

        pipe._guidance_scale = self.guidance_scale
        pipe._guidance_rescale = 0.0
        pipe._clip_skip = None
        pipe._cross_attention_kwargs = None
        pipe._denoising_end = None
        pipe._interrupt = False

        device = pipe._execution_device
        pipe.scheduler.set_timesteps(self.num_inference_steps, device=device)
        timesteps = pipe.scheduler.timesteps
        step_kwargs = pipe.prepare_extra_step_kwargs(None, 0.0)
        prompt_embeds, text_embeds, time_ids = self._conditioning_generic(text_embeddings)

        latents = latents_start.clone()
        timestep_cond = None
        if pipe.unet.config.time_cond_proj_dim is not None:
            w = torch.tensor(pipe.guidance_scale - 1).repeat(1)
            timestep_cond = pipe.get_guidance_scale_embedding(
                w, embedding_dim=pipe.unet.config.time_cond_proj_dim
            ).to(device=device, dtype=latents.dtype)
        pipe._num_timesteps = len(timesteps)

        trajectory = []
        for i, t in enumerate(timesteps):
            if i < idx_start:
                trajectory.append(None)
                continue
            if i == idx_start:
                latents = latents_start.clone()
            if i > 0 and coeffs[i] > 0:
                latents = slerp(latents, list_latents_mixing[i - 1].clone(), coeffs[i])

            cfg = pipe.do_classifier_free_guidance
            model_in = torch.cat([latents] * 2) if cfg else latents
            model_in = pipe.scheduler.scale_model_input(model_in, t)
            eps = pipe.unet(
                model_in, t, encoder_hidden_states=prompt_embeds, timestep_cond=timestep_cond,
                cross_attention_kwargs=pipe.cross_attention_kwargs,
                added_cond_kwargs={"text_embeds": text_embeds, "time_ids": time_ids},
                return_dict=False)[0]
            if cfg:
                eps_uncond, eps_text = eps.chunk(2)
                eps = eps_uncond + pipe.guidance_scale * (eps_text - eps_uncond)
            latents = pipe.scheduler.step(eps, t, latents, **step_kwargs, return_dict=False)[0]
            trajectory.append(latents.clone())
        return trajectory
