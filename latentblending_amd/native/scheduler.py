"""Host side of the Euler / Euler-ancestral schedulers (diffusers 0.25 semantics for SDXL base
and SDXL-Turbo).  Only the sigma / timestep tables and per-step scalar
coefficients live here (float64 numpy); the tensor work — ``x / sqrt(sigma^2+1)``, CFG combine
and the update — runs in ``lb_scale_model_input_f16`` / ``lb_euler_step_f16``.

Reference call sites: /root/reference/latentblending/diffusers_holder.py:42,53,247 (set_timesteps),
:301 (order), :330 (scale_model_input), :356 (step).  Known answers: tests/golden/scheduler.json.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch

from ..hip import ops

NUM_TRAIN_TIMESTEPS = 1000


def _sigma_table() -> np.ndarray:
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, NUM_TRAIN_TIMESTEPS, dtype=np.float64) ** 2
    alphas_cumprod = np.cumprod(1.0 - betas)
    return ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5


class SeededDeviceNoise:
    """Noise stream ``callable(shape) -> fp16 tensor`` from a device generator with an explicit seed: the same
    seed gives the same stream on every rank of a branch farm (same GPU model, same torch build), which is what
    lets ranks compute the anchor trajectories redundantly instead of exchanging them."""

    def __init__(self, seed: int, device):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)

    def __call__(self, shape):
        return torch.randn(tuple(shape), generator=self.gen, device=self.device, dtype=torch.float16)

    def many(self, n: int, shape):
        """``n`` draws of ``shape`` ([1, ...]) stacked along dim 0 as ONE generator call (one launch instead of n; every rank of a
        farm makes the same call, so the streams stay aligned)."""
        return torch.randn((int(n),) + tuple(shape[1:]), generator=self.gen, device=self.device, dtype=torch.float16)


class NativeEulerScheduler:
    order = 1
    kind = "euler"

    def __init__(self, ancestral: bool, timestep_spacing: Optional[str] = None, device="cuda"):
        self.ancestral = ancestral
        self.timestep_spacing = timestep_spacing or ("trailing" if ancestral else "leading")
        self.device = device
        self._table = _sigma_table()
        self.noise_source = None          # callable(shape) -> tensor; None = device RNG
        self._step_index = None
        self.set_timesteps(30)

    # ---- tables -----------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device=None):
        n = int(num_inference_steps)
        if self.timestep_spacing == "trailing":
            ts = np.round(np.arange(NUM_TRAIN_TIMESTEPS, 0, -NUM_TRAIN_TIMESTEPS / n)) - 1
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n) * (NUM_TRAIN_TIMESTEPS // n)).round()[::-1].copy() + 1   # steps_offset = 1
        else:
            raise ValueError(f"unsupported timestep_spacing {self.timestep_spacing}")
        ts = ts.astype(np.float32)
        sig = np.interp(ts, np.arange(0, NUM_TRAIN_TIMESTEPS), self._table)
        self.sigmas_np = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps_np = ts
        self.timesteps = torch.from_numpy(ts)                  # host tensor: iterating it never syncs
        self.sigmas = torch.from_numpy(self.sigmas_np)
        self.num_inference_steps = n
        self._step_index = None

    @property
    def init_noise_sigma(self) -> float:
        m = float(self.sigmas_np.max())
        return m if self.timestep_spacing in ("linspace", "trailing") else (m * m + 1) ** 0.5

    def index_of(self, t) -> int:
        hits = np.nonzero(self.timesteps_np == float(t))[0]
        return int(hits[0])

    def step_row(self, i: int, guidance: float = 0.0) -> Tuple[float, float, float, float, float]:
        """(sigma_from, sigma_next, sigma_up, guidance, dt) for ``lb_euler_step_f16``; computed in
        Python floats exactly like the scheduler's own scalar arithmetic."""
        s_from, s_to = float(self.sigmas_np[i]), float(self.sigmas_np[i + 1])
        if self.ancestral:
            s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
            s_down = (s_to ** 2 - s_up ** 2) ** 0.5
            return (s_from, s_down, s_up, guidance, s_down - s_from)
        return (s_from, s_to, 0.0, guidance, s_to - s_from)

    # ---- diffusers-style tensor API (generic holder loop / foreign callers) ------------------
    def _locate(self, t) -> int:
        if self._step_index is None:
            self._step_index = self.index_of(float(t))
        return self._step_index

    def scale_model_input(self, sample: torch.Tensor, timestep) -> torch.Tensor:
        i = self._locate(timestep)
        params = ops.step_params([self.step_row(i)] * sample.shape[0], sample.device)
        return ops.scale_model_input(sample.contiguous(), params)

    def draw_noise(self, shape, device) -> torch.Tensor:
        if self.noise_source is not None:
            return self.noise_source(tuple(shape)).to(device=device, dtype=torch.float16)
        return torch.randn(shape, device=device, dtype=torch.float16)

    def draw_noise_many(self, n: int, shape, device) -> torch.Tensor:
        """``n`` draws of ``shape`` ([1, C, L, L]) stacked along dim 0, in draw order.  The device RNG (and a noise source with a
        ``many`` method) serves them with ONE launch - a cfg-2 transition draws 38 latents' worth of ancestral noise, 38 launches of
        ~14 us each before round 6; a source without ``many`` (a recorded tape) is called once per draw, in the same order."""
        n = int(n)
        if n == 0:
            return torch.empty((0,) + tuple(shape[1:]), device=device, dtype=torch.float16)
        if self.noise_source is None:
            return torch.randn((n,) + tuple(shape[1:]), device=device, dtype=torch.float16)
        many = getattr(self.noise_source, "many", None)
        if many is not None:
            return many(n, tuple(shape)).to(device=device, dtype=torch.float16)
        return torch.cat([self.draw_noise(shape, device) for _ in range(n)])

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, **_):
        i = self._locate(timestep)
        B = sample.shape[0]
        params = ops.step_params([self.step_row(i)] * B, sample.device)
        noise = self.draw_noise(sample.shape, sample.device) if self.ancestral else None
        out = ops.euler_step(sample.contiguous(), model_output.contiguous(), params, noise=noise,
                             ancestral=self.ancestral)
        self._step_index += 1
        return (out,)

    def device_step(self, latents, eps, params, noise=None, cfg=False):
        """The batched step of the native loops (native/pipe.py): one launch for all samples, per-sample rows in ``params``."""
        return ops.euler_step(latents, eps, params, noise=noise, cfg=cfg, ancestral=self.ancestral)


class NativeDDIMScheduler:
    """DDIM (eta = 0, epsilon prediction) as diffusers configures it for SD / SDXL: scaled-linear betas, leading spacing,
    steps_offset = 1, set_alpha_to_one = False, clip_sample = False.  Host side = the fp32 abar table built exactly as
    diffusers builds it (``torch.cumprod`` of fp32 alphas) and the per-step coefficients; the tensor work is
    ``lb_ddim_step_f16``.  Same interface as :class:`NativeEulerScheduler` (``step_row`` / ``device_step`` for the native loops,
    ``scale_model_input`` / ``step`` for the generic diffusers-style loop); ``scale_model_input`` is the identity and
    ``init_noise_sigma`` is 1.  Reached from /root/reference/latentblending/diffusers_holder.py:330,356 when a pipe carries
    this scheduler (``NativeSDXLPipe(scheduler="ddim")``); the reference's own SDXL pipes carry Euler schedulers (:42)."""
    order = 1
    kind = "ddim"
    ancestral = False
    init_noise_sigma = 1.0

    def __init__(self, device="cuda"):
        self.device = device
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, NUM_TRAIN_TIMESTEPS, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)            # fp32, as diffusers holds it
        self.final_alpha_cumprod = self.alphas_cumprod[0]                   # set_alpha_to_one = False
        self.noise_source = None
        self._step_index = None
        self.set_timesteps(30)

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = int(num_inference_steps)
        ratio = NUM_TRAIN_TIMESTEPS // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + 1          # leading, steps_offset = 1
        self.timesteps_np = ts.astype(np.float32)
        self.timesteps = torch.from_numpy(self.timesteps_np)
        self.num_inference_steps = n
        self._ratio = ratio
        self._step_index = None

    def index_of(self, t) -> int:
        return int(np.nonzero(self.timesteps_np == float(t))[0][0])

    def alpha_pair(self, i: int):
        """(abar_t, abar_prev) of step ``i`` as 0-dim fp32 tensors (prev_timestep = t - 1000 // n; below 0: final_alpha_cumprod)."""
        t = int(self.timesteps_np[i])
        prev = t - self._ratio
        return self.alphas_cumprod[t], (self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod)

    def step_row(self, i: int, guidance: float = 0.0):
        """(0, sqrt(abar_t), sqrt(abar_prev), guidance, sqrt(1 - abar_t), sqrt(1 - abar_prev), 1 / sqrt(abar_t)) in fp32 tensor
        arithmetic, the scalars diffusers forms inside DDIMScheduler.step (beta_prod_t ** 0.5, alpha_prod_t ** 0.5, ...).  The last
        one is the fp32 reciprocal the device library forms when a tensor is divided by a 0-dim host tensor (its true-division
        kernel multiplies by ``1 / b`` for a CPU scalar ``b``): ``lb_ddim_step_f16`` multiplies by it instead of dividing."""
        a_t, a_p = self.alpha_pair(i)
        sa_t = a_t ** 0.5
        inv = torch.ones((), dtype=torch.float32) / sa_t.to(torch.float32)
        return (0.0, float(sa_t), float(a_p ** 0.5), guidance, float((1 - a_t) ** 0.5), float((1 - a_p) ** 0.5), float(inv))

    def _locate(self, t) -> int:
        if self._step_index is None:
            self._step_index = self.index_of(float(t))
        return self._step_index

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def draw_noise(self, shape, device) -> torch.Tensor:        # (never used: eta = 0)
        return torch.zeros(shape, device=device, dtype=torch.float16)

    def device_step(self, latents, eps, params, noise=None, cfg=False):
        return ops.ddim_step(latents, eps, params, cfg=cfg)

    def step(self, model_output, timestep, sample, generator=None, return_dict=False, **_):
        i = self._locate(timestep)
        params = ops.step_params([self.step_row(i)] * sample.shape[0], sample.device)
        out = ops.ddim_step(sample.contiguous(), model_output.contiguous(), params)
        self._step_index += 1
        return (out,)
