"""ORACLE — test infrastructure, NOT product code.

CPU float32 restatement (plain PyTorch) of the third-party arithmetic the reference's hot path
executes through ``diffusers==0.25.0`` / ``lpips==0.1.4`` (both absent from /root/reference and
from this image; pinned in /root/reference/requirements.txt:1,3).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

PARITY UNPINNED for the UNet, the schedulers' step arithmetic and LPIPS in this file: the reference
ships no tests, golden vectors or weights for them, and the third-party sources
are not available here; the restatement follows the published architecture (SURVEY.md Appendix B,
validated by the exact parameter counts 2,567,463,684 / 49,490,199 — see ``count_params``).
PINNED (round 6, tests/test_oracle_pins_cpu.py): ``vae_decode`` equals — same weights, fp32
round-off — the latent-diffusion decoder that ``transformers`` ships as ``JanusVQVAEDecoder``
(the module diffusers' ``AutoencoderKL.decoder`` was ported from; its extra lowest-level
attention blocks emptied, as the SD autoencoders have none), at the tiny and at the full SDXL
width; ``attention`` equals torch's ``scaled_dot_product_attention``.  Call sites restated:
  latentblending/diffusers_holder.py:330      scheduler.scale_model_input
  latentblending/diffusers_holder.py:336-344  pipe.unet(...)
  latentblending/diffusers_holder.py:356      scheduler.step
  latentblending/diffusers_holder.py:135,141  vae.decode + image_processor.postprocess
  latentblending/blending_engine.py:744-758   lpips.LPIPS(net='alex')
Scheduler tables ARE pinned by closed-form known answers (tests/golden/scheduler.json).
State-dict key names follow the HF diffusers layout so real checkpoints could be dropped in.
"""
from __future__ import annotations

import math
import os
import zlib
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# =============================================================================================
# configs
# =============================================================================================
@dataclass
class UNetCfg:
    in_channels: int = 4
    out_channels: int = 4
    block_channels: Tuple[int, ...] = (320, 640, 1280)
    layers_per_block: int = 2
    transformer_depth: Tuple[int, ...] = (0, 2, 10)   # per down block; mid uses the last entry
    head_dim: int = 64
    cross_dim: int = 2048
    pooled_dim: int = 1280
    add_time_dim: int = 256
    sample_size: int = 128                            # 128 base, 64 turbo
    norm_groups: int = 32
    time_cond_proj_dim: Optional[int] = None

    @property
    def time_embed_dim(self) -> int:
        return self.block_channels[0] * 4

    @property
    def add_in_dim(self) -> int:
        return self.pooled_dim + 6 * self.add_time_dim


@dataclass
class VAECfg:
    latent_channels: int = 4
    out_channels: int = 3
    block_channels: Tuple[int, ...] = (128, 256, 512, 512)   # encoder order; decoder walks reversed
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.13025
    force_upcast: bool = True


def tiny_unet_cfg() -> UNetCfg:
    return UNetCfg(block_channels=(64, 128, 256), transformer_depth=(0, 1, 2), cross_dim=256,
                   pooled_dim=128, add_time_dim=32, sample_size=16)


def tiny_vae_cfg() -> VAECfg:
    return VAECfg(block_channels=(32, 64, 128, 128))


# =============================================================================================
# synthetic weights (HF key layout)
# =============================================================================================
_BIG = 3 << 19        # tensors above this many elements (1.57 M: the 1280 x 1280 projections and up; the tiny test models
#                       stop at 1.18 M) take the chunked numpy path
_CHUNK = 1 << 18
_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1)))
    return _POOL


def _gen(name: str, shape, std: float, seed: int, mean: float = 0.0) -> Tensor:
    """N(mean, std^2) values from a seed that depends only on (name, seed).  Tensors up to 1.5 * 2^20 elements: one torch CPU
    generator (the scheme every golden fixture was made with).  Larger ones (only the full-size SDXL / CLIP models have them;
    2.6 G values took 76 s of one core per process): 2^18-element chunks, each from its own numpy PCG64 stream, filled by a
    small thread pool (numpy releases the GIL) - same values whatever the thread count."""
    base = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF
    shape = tuple(int(v) for v in shape)
    n = 1
    for v in shape:
        n *= v
    if n <= _BIG:
        g = torch.Generator().manual_seed(base)
        return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean
    buf = np.empty(n, dtype=np.float32)

    def fill(c):
        lo, hi = c * _CHUNK, min(n, (c + 1) * _CHUNK)
        np.random.Generator(np.random.PCG64([base, c])).standard_normal(hi - lo, dtype=np.float32, out=buf[lo:hi])
    list(_pool().map(fill, range((n + _CHUNK - 1) // _CHUNK)))
    return torch.from_numpy(buf).view(shape).mul_(std).add_(mean)


class _Spec:
    """Collects (name -> shape, kind) so that weights and parameter counts share one enumeration."""

    def __init__(self):
        self.items: List[Tuple[str, Tuple[int, ...], str, float]] = []

    def linear(self, name, cin, cout, bias=True, gain=1.0):
        self.items.append((name + ".weight", (cout, cin), "w", gain / math.sqrt(cin)))
        if bias:
            self.items.append((name + ".bias", (cout,), "b", 0.02))

    def conv(self, name, cin, cout, k, gain=1.0):
        self.items.append((name + ".weight", (cout, cin, k, k), "w", gain / math.sqrt(cin * k * k)))
        self.items.append((name + ".bias", (cout,), "b", 0.02))

    def norm(self, name, c):
        self.items.append((name + ".weight", (c,), "g", 0.05))
        self.items.append((name + ".bias", (c,), "b", 0.02))


def _resnet_spec(s: _Spec, p: str, cin: int, cout: int, temb: Optional[int]):
    s.norm(p + ".norm1", cin)
    s.conv(p + ".conv1", cin, cout, 3)
    if temb:
        s.linear(p + ".time_emb_proj", temb, cout)
    s.norm(p + ".norm2", cout)
    s.conv(p + ".conv2", cout, cout, 3, gain=0.5)
    if cin != cout:
        s.conv(p + ".conv_shortcut", cin, cout, 1)


def _transformer_spec(s: _Spec, p: str, c: int, depth: int, cross: int):
    s.norm(p + ".norm", c)
    s.linear(p + ".proj_in", c, c)
    for d in range(depth):
        b = f"{p}.transformer_blocks.{d}"
        s.norm(b + ".norm1", c)
        for nm in ("to_q", "to_k", "to_v"):
            s.linear(f"{b}.attn1.{nm}", c, c, bias=False)
        s.linear(b + ".attn1.to_out.0", c, c, gain=0.5)
        s.norm(b + ".norm2", c)
        s.linear(b + ".attn2.to_q", c, c, bias=False)
        s.linear(b + ".attn2.to_k", cross, c, bias=False)
        s.linear(b + ".attn2.to_v", cross, c, bias=False)
        s.linear(b + ".attn2.to_out.0", c, c, gain=0.5)
        s.norm(b + ".norm3", c)
        s.linear(b + ".ff.net.0.proj", c, 8 * c)
        s.linear(b + ".ff.net.2", 4 * c, c, gain=0.5)
    s.linear(p + ".proj_out", c, c, gain=0.5)


def unet_spec(cfg: UNetCfg) -> _Spec:
    s = _Spec()
    ch = cfg.block_channels
    T = cfg.time_embed_dim
    s.conv("conv_in", cfg.in_channels, ch[0], 3)
    s.linear("time_embedding.linear_1", ch[0], T)
    s.linear("time_embedding.linear_2", T, T)
    s.linear("add_embedding.linear_1", cfg.add_in_dim, T)
    s.linear("add_embedding.linear_2", T, T)
    prev = ch[0]
    for bi, c in enumerate(ch):
        for li in range(cfg.layers_per_block):
            _resnet_spec(s, f"down_blocks.{bi}.resnets.{li}", prev, c, T)
            if cfg.transformer_depth[bi]:
                _transformer_spec(s, f"down_blocks.{bi}.attentions.{li}", c,
                                  cfg.transformer_depth[bi], cfg.cross_dim)
            prev = c
        if bi < len(ch) - 1:
            s.conv(f"down_blocks.{bi}.downsamplers.0.conv", c, c, 3)
    c = ch[-1]
    _resnet_spec(s, "mid_block.resnets.0", c, c, T)
    _transformer_spec(s, "mid_block.attentions.0", c, cfg.transformer_depth[-1], cfg.cross_dim)
    _resnet_spec(s, "mid_block.resnets.1", c, c, T)
    for ui, (c, cin_list) in enumerate(unet_up_plan(cfg)):
        depth = list(reversed(cfg.transformer_depth))[ui]
        for li, cin in enumerate(cin_list):
            _resnet_spec(s, f"up_blocks.{ui}.resnets.{li}", cin, c, T)
            if depth:
                _transformer_spec(s, f"up_blocks.{ui}.attentions.{li}", c, depth, cfg.cross_dim)
        if ui < len(ch) - 1:
            s.conv(f"up_blocks.{ui}.upsamplers.0.conv", c, c, 3)
    s.norm("conv_norm_out", ch[0])
    s.conv("conv_out", ch[0], cfg.out_channels, 3, gain=0.5)
    return s


def unet_skip_channels(cfg: UNetCfg) -> List[int]:
    ch = cfg.block_channels
    skips = [ch[0]]
    for bi, c in enumerate(ch):
        skips += [c] * cfg.layers_per_block
        if bi < len(ch) - 1:
            skips.append(c)
    return skips


def unet_up_plan(cfg: UNetCfg) -> List[Tuple[int, List[int]]]:
    """Per up block: (out channels, [resnet input channels = hidden + popped skip])."""
    skips = unet_skip_channels(cfg)
    plan = []
    hidden = cfg.block_channels[-1]
    for c in reversed(cfg.block_channels):
        cins = []
        for _ in range(cfg.layers_per_block + 1):
            cins.append(hidden + skips.pop())
            hidden = c
        plan.append((c, cins))
    return plan


def vae_decoder_spec(cfg: VAECfg) -> _Spec:
    s = _Spec()
    rev = list(reversed(cfg.block_channels))
    top = rev[0]
    s.conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    s.conv("decoder.conv_in", cfg.latent_channels, top, 3)
    _resnet_spec(s, "decoder.mid_block.resnets.0", top, top, None)
    a = "decoder.mid_block.attentions.0"
    s.norm(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v"):
        s.linear(f"{a}.{nm}", top, top)
    s.linear(a + ".to_out.0", top, top, gain=0.5)
    _resnet_spec(s, "decoder.mid_block.resnets.1", top, top, None)
    prev = top
    for ui, c in enumerate(rev):
        for li in range(cfg.layers_per_block + 1):
            _resnet_spec(s, f"decoder.up_blocks.{ui}.resnets.{li}", prev, c, None)
            prev = c
        if ui < len(rev) - 1:
            s.conv(f"decoder.up_blocks.{ui}.upsamplers.0.conv", c, c, 3)
    s.norm("decoder.conv_norm_out", rev[-1])
    s.conv("decoder.conv_out", rev[-1], cfg.out_channels, 3, gain=0.5)
    return s


LPIPS_CONVS = [(3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1),
               (256, 256, 3, 1, 1)]   # (cin, cout, k, stride, pad); maxpool(3,2) before conv 2 and 3


def lpips_spec() -> _Spec:
    s = _Spec()
    for i, (cin, cout, k, _, _) in enumerate(LPIPS_CONVS):
        s.conv(f"net.conv{i + 1}", cin, cout, k, gain=1.4)
        s.items.append((f"lin{i}.weight", (cout,), "lin", 1.0))
    return s


def count_params(spec: _Spec) -> int:
    return sum(int(np.prod(shape)) for _, shape, _, _ in spec.items)


def make_weights(spec: _Spec, seed: int = 0, round_fp16: bool = True) -> Dict[str, Tensor]:
    """Seeded synthetic weights.  ``round_fp16``: values are fp16-representable (stored as fp32) so
    that the oracle and the fp16 device path use bit-identical parameters."""
    out = {}
    for name, shape, kind, scale in spec.items:
        if kind == "w":
            t = _gen(name, shape, scale, seed)
        elif kind == "b":
            t = _gen(name, shape, scale, seed)
        elif kind == "g":
            t = _gen(name, shape, scale, seed, mean=1.0)
        elif kind == "lin":
            t = _gen(name, shape, 1.0, seed).abs() / shape[0]
        else:
            raise KeyError(kind)
        out[name] = t.half().float() if round_fp16 else t
    return out


# =============================================================================================
# mixing primitives  (reference: latentblending/utils.py:29-71 and :74-102)
# =============================================================================================
def slerp(p0: Tensor, p1: Tensor, fract: float) -> Tensor:
    """Whole-tensor spherical interpolation in float64; fp16 in -> fp16 out, else fp32 out."""
    to_half = p0.dtype == torch.float16
    a, b = p0.double(), p1.double()
    cos = torch.sum(a * b) / (torch.linalg.norm(a) * torch.linalg.norm(b))
    cos = cos.clamp(-1 + 1e-7, 1 - 1e-7)
    theta = torch.arccos(cos)
    sin_theta = torch.sin(theta)
    w0 = torch.sin(theta - theta * fract) / sin_theta
    w1 = torch.sin(theta * fract) / sin_theta
    mix = a * w0 + b * w1
    return mix.half() if to_half else mix.float()


def lerp(p0, p1, fract: float):
    return (1 - fract) * p0 + fract * p1


class TorchCpuBackend:
    """Checker backend for host-logic tests on GPU-less machines (see latentblending_amd.backend)."""
    name = "oracle-cpu"

    def slerp(self, p0, p1, fract):
        return slerp(p0, p1, fract)

    def lerp(self, p0, p1, fract):
        return lerp(p0, p1, fract)

    def slerp_pairs(self, p0, p1, fracts):
        return [slerp(a, b, f) for a, b, f in zip(p0, p1, fracts)]


# =============================================================================================
# schedulers
# =============================================================================================
def sdxl_sigma_table() -> np.ndarray:
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    acp = np.cumprod(1.0 - betas)
    return ((1 - acp) / acp) ** 0.5


def make_timesteps(n: int, spacing: str) -> np.ndarray:
    if spacing == "trailing":
        return (np.round(np.arange(1000, 0, -1000 / n)) - 1).astype(np.float32)
    if spacing == "leading":
        ratio = 1000 // n
        return ((np.arange(0, n) * ratio).round()[::-1].copy() + 1).astype(np.float32)
    raise ValueError(spacing)


class DDIMScheduler:
    """diffusers DDIMScheduler as SD / SDXL configure it (scaled-linear betas, leading spacing, steps_offset 1,
    set_alpha_to_one False, clip_sample False), eta = 0, epsilon prediction - restated from its published algorithm
    (diffusers 0.25.0 schedulers/scheduling_ddim.py: set_timesteps, step formulas (12) / (16) of the DDIM paper).  Like
    diffusers it does NOT upcast: the tensor arithmetic runs in the sample's dtype with 0-dim fp32 coefficients."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.set_timesteps(30)

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy((np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype(np.int64) + 1)

    def scale_model_input(self, sample, t=None):
        return sample

    def step(self, model_output, t, sample, eta=0.0, generator=None, return_dict=False, **_):
        t = int(t)
        prev_t = t - 1000 // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        if sample.dtype != torch.float16:
            pred_original_sample = (sample - beta_prod_t ** (0.5) * model_output) / alpha_prod_t ** (0.5)
            pred_sample_direction = (1 - alpha_prod_t_prev) ** (0.5) * model_output
            return (alpha_prod_t_prev ** (0.5) * pred_original_sample + pred_sample_direction,)
        # fp16 tensors: the same expressions as torch evaluates them ON THE DEVICE - the 0-dim fp32 coefficients live on the
        # host, so every kernel takes them as fp32 scalars (opmath), computes in fp32 and rounds its result to fp16.  (torch's
        # CPU kernels would first round the coefficient itself to fp16: not what a GPU pipeline computes, so not restated.)
        rt = lambda v: v.to(torch.float16).float()                                         # noqa: E731
        sb_t, sa_t = float(beta_prod_t ** 0.5), float(alpha_prod_t ** 0.5)
        sb_p, sa_p = float((1 - alpha_prod_t_prev) ** 0.5), float(alpha_prod_t_prev ** 0.5)
        x, e = sample.float(), model_output.float()
        # (division by a 0-dim HOST tensor: the device library multiplies by the fp32 reciprocal 1 / b, formed once)
        inv = float(torch.ones((), dtype=torch.float32) / torch.tensor(sa_t, dtype=torch.float32))
        x0 = rt(rt(x - rt(sb_t * e)) * inv)
        direction = rt(sb_p * e)
        return ((rt(sa_p * x0) + direction).to(torch.float16),)


class EulerScheduler:
    """Euler (``ancestral=False``, SDXL base: leading spacing, offset 1) and Euler-ancestral
    (``ancestral=True``, SDXL-Turbo: trailing spacing), epsilon prediction."""
    order = 1

    def __init__(self, ancestral: bool, spacing: Optional[str] = None,
                 noise_source: Optional[Callable] = None):
        self.ancestral = ancestral
        self.spacing = spacing or ("trailing" if ancestral else "leading")
        self.noise_source = noise_source
        self._table = sdxl_sigma_table()
        self.timesteps = None
        self.sigmas = None
        self._step_index = None
        self.set_timesteps(30)

    def set_timesteps(self, n, device=None):
        ts = make_timesteps(n, self.spacing)
        sig = np.interp(ts, np.arange(0, 1000), self._table)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self._step_index = None

    @property
    def init_noise_sigma(self):
        m = float(self.sigmas.max())
        return m if self.spacing in ("linspace", "trailing") else (m * m + 1) ** 0.5

    def _locate(self, t):
        if self._step_index is None:
            hits = (self.timesteps == float(t)).nonzero()
            self._step_index = int(hits[0])  # mid-schedule starts locate by value
        return self._step_index

    def scale_model_input(self, sample, t):
        s = float(self.sigmas[self._locate(t)])
        return (sample / ((s * s + 1) ** 0.5)).to(sample.dtype)

    def step(self, model_output, t, sample, generator=None, return_dict=False, **_):
        i = self._locate(t)
        s_from, s_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        x = sample.float()
        eps = model_output.float()
        # x0 = x - s*eps ; d = (x - x0)/s == eps up to rounding; keep the two-step form
        x0 = x - s_from * eps
        d = (x - x0) / s_from
        if self.ancestral:
            s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
            s_down = (s_to ** 2 - s_up ** 2) ** 0.5
            nxt = x + d * (s_down - s_from)
            if self.noise_source is not None:
                noise = self.noise_source(tuple(model_output.shape))
            else:
                noise = torch.randn(model_output.shape, dtype=model_output.dtype, generator=generator)
            nxt = nxt + noise.float() * s_up
        else:
            nxt = x + d * (s_to - s_from)
        self._step_index += 1
        return (nxt.to(model_output.dtype),)


def ancestral_sigmas(s_from: float, s_to: float) -> Tuple[float, float]:
    s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
    return s_up, (s_to ** 2 - s_up ** 2) ** 0.5


# =============================================================================================
# UNet
# =============================================================================================
def sinusoid(values: Tensor, dim: int) -> Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=values.device) / half)
    ang = values.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _gn(x, w, p, groups, eps):
    return F.group_norm(x, groups, w[p + ".weight"], w[p + ".bias"], eps)


def _lin(x, w, p):
    return F.linear(x, w[p + ".weight"], w.get(p + ".bias"))


def _conv(x, w, p, stride=1, pad=1):
    return F.conv2d(x, w[p + ".weight"], w[p + ".bias"], stride=stride, padding=pad)


def resnet_block(x, emb, w, p, groups, eps):
    h = _conv(F.silu(_gn(x, w, p + ".norm1", groups, eps)), w, p + ".conv1")
    if emb is not None:
        h = h + _lin(F.silu(emb), w, p + ".time_emb_proj")[:, :, None, None]
    h = _conv(F.silu(_gn(h, w, p + ".norm2", groups, eps)), w, p + ".conv2")
    if (p + ".conv_shortcut.weight") in w:
        x = _conv(x, w, p + ".conv_shortcut", pad=0)
    return x + h


def attention(q, k, v, heads):
    """q [B,Sq,C], k/v [B,Sk,C] -> [B,Sq,C]; softmax(QK^T/sqrt(d))V per head."""
    B, Sq, C = q.shape
    d = C // heads
    qh = q.view(B, Sq, heads, d).transpose(1, 2)
    kh = k.view(B, -1, heads, d).transpose(1, 2)
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, Sq, C)


def transformer_block(x, ctx, w, b, heads):
    n = F.layer_norm(x, x.shape[-1:], w[b + ".norm1.weight"], w[b + ".norm1.bias"], 1e-5)
    a = attention(_lin(n, w, b + ".attn1.to_q"), _lin(n, w, b + ".attn1.to_k"),
                  _lin(n, w, b + ".attn1.to_v"), heads)
    x = x + _lin(a, w, b + ".attn1.to_out.0")
    n = F.layer_norm(x, x.shape[-1:], w[b + ".norm2.weight"], w[b + ".norm2.bias"], 1e-5)
    a = attention(_lin(n, w, b + ".attn2.to_q"), _lin(ctx, w, b + ".attn2.to_k"),
                  _lin(ctx, w, b + ".attn2.to_v"), heads)
    x = x + _lin(a, w, b + ".attn2.to_out.0")
    n = F.layer_norm(x, x.shape[-1:], w[b + ".norm3.weight"], w[b + ".norm3.bias"], 1e-5)
    h, gate = _lin(n, w, b + ".ff.net.0.proj").chunk(2, dim=-1)
    return x + _lin(h * F.gelu(gate), w, b + ".ff.net.2")


def transformer_2d(x, ctx, w, p, depth, heads, groups):
    B, C, H, W = x.shape
    h = _gn(x, w, p + ".norm", groups, 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = _lin(h, w, p + ".proj_in")
    for d in range(depth):
        h = transformer_block(h, ctx, w, f"{p}.transformer_blocks.{d}", heads)
    h = _lin(h, w, p + ".proj_out").reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + x


def unet_embedding(cfg: UNetCfg, w, t: Tensor, text_embeds: Tensor, time_ids: Tensor) -> Tensor:
    B = text_embeds.shape[0]
    t = t.reshape(-1).float().expand(B)
    temb = _lin(F.silu(_lin(sinusoid(t, cfg.block_channels[0]), w, "time_embedding.linear_1")),
                w, "time_embedding.linear_2")
    ids = sinusoid(time_ids.reshape(-1), cfg.add_time_dim).reshape(B, -1)
    add_in = torch.cat([text_embeds.float(), ids], dim=-1)
    aug = _lin(F.silu(_lin(add_in, w, "add_embedding.linear_1")), w, "add_embedding.linear_2")
    return temb + aug


def unet_forward(cfg: UNetCfg, w: Dict[str, Tensor], sample: Tensor, t, ctx: Tensor,
                 text_embeds: Tensor, time_ids: Tensor, taps: Optional[dict] = None) -> Tensor:
    """sample [B,4,L,L] -> eps [B,4,L,L]; all arithmetic fp32.  ``taps`` collects intermediates."""
    g = cfg.norm_groups
    x = sample.float()
    ctx = ctx.float()
    t = torch.as_tensor(t)
    emb = unet_embedding(cfg, w, t, text_embeds, time_ids)
    ch = cfg.block_channels
    h = _conv(x, w, "conv_in")
    skips = [h]
    if taps is not None:
        taps["emb"] = emb
        taps["conv_in"] = h
    for bi, c in enumerate(ch):
        for li in range(cfg.layers_per_block):
            h = resnet_block(h, emb, w, f"down_blocks.{bi}.resnets.{li}", g, 1e-5)
            if cfg.transformer_depth[bi]:
                h = transformer_2d(h, ctx, w, f"down_blocks.{bi}.attentions.{li}",
                                   cfg.transformer_depth[bi], c // cfg.head_dim, g)
            skips.append(h)
        if bi < len(ch) - 1:
            h = _conv(h, w, f"down_blocks.{bi}.downsamplers.0.conv", stride=2, pad=1)
            skips.append(h)
        if taps is not None:
            taps[f"down{bi}"] = h
    c = ch[-1]
    h = resnet_block(h, emb, w, "mid_block.resnets.0", g, 1e-5)
    h = transformer_2d(h, ctx, w, "mid_block.attentions.0", cfg.transformer_depth[-1],
                       c // cfg.head_dim, g)
    h = resnet_block(h, emb, w, "mid_block.resnets.1", g, 1e-5)
    if taps is not None:
        taps["mid"] = h
    depths = list(reversed(cfg.transformer_depth))
    for ui, (c, cins) in enumerate(unet_up_plan(cfg)):
        for li in range(len(cins)):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(h, emb, w, f"up_blocks.{ui}.resnets.{li}", g, 1e-5)
            if depths[ui]:
                h = transformer_2d(h, ctx, w, f"up_blocks.{ui}.attentions.{li}", depths[ui],
                                   c // cfg.head_dim, g)
        if ui < len(ch) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, w, f"up_blocks.{ui}.upsamplers.0.conv")
        if taps is not None:
            taps[f"up{ui}"] = h
    h = F.silu(_gn(h, w, "conv_norm_out", g, 1e-5))
    return _conv(h, w, "conv_out")


# =============================================================================================
# VAE decoder + postprocess
# =============================================================================================
def vae_decode(cfg: VAECfg, w: Dict[str, Tensor], z: Tensor, taps: Optional[dict] = None) -> Tensor:
    """z (already divided by the scaling factor) [B,4,L,L] -> image [B,3,8L,8L] in ~[-1,1]."""
    g, eps = cfg.norm_groups, 1e-6
    h = _conv(z.float(), w, "post_quant_conv", pad=0)
    h = _conv(h, w, "decoder.conv_in")
    h = resnet_block(h, None, w, "decoder.mid_block.resnets.0", g, eps)
    a = "decoder.mid_block.attentions.0"
    B, C, H, W = h.shape
    n = _gn(h, w, a + ".group_norm", g, eps).view(B, C, H * W).transpose(1, 2)
    att = attention(_lin(n, w, a + ".to_q"), _lin(n, w, a + ".to_k"), _lin(n, w, a + ".to_v"), 1)
    h = h + _lin(att, w, a + ".to_out.0").transpose(1, 2).reshape(B, C, H, W)
    h = resnet_block(h, None, w, "decoder.mid_block.resnets.1", g, eps)
    if taps is not None:
        taps["mid"] = h
    rev = list(reversed(cfg.block_channels))
    for ui in range(len(rev)):
        for li in range(cfg.layers_per_block + 1):
            h = resnet_block(h, None, w, f"decoder.up_blocks.{ui}.resnets.{li}", g, eps)
        if ui < len(rev) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, w, f"decoder.up_blocks.{ui}.upsamplers.0.conv")
        if taps is not None:
            taps[f"up{ui}"] = h
    h = F.silu(_gn(h, w, "decoder.conv_norm_out", g, eps))
    return _conv(h, w, "decoder.conv_out")


def postprocess_u8(image: Tensor) -> np.ndarray:
    """``VaeImageProcessor.postprocess``: (x/2+0.5).clamp(0,1) -> NHWC -> round(x*255) uint8."""
    x = (image.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    return (x * 255).round().astype("uint8")


# =============================================================================================
# LPIPS-Alex
# =============================================================================================
_LP_SHIFT = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
_LP_SCALE = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)


def lpips_features(w: Dict[str, Tensor], img: Tensor) -> List[Tensor]:
    h = (img.float() - _LP_SHIFT) / _LP_SCALE
    feats = []
    for i, (_, _, _, stride, pad) in enumerate(LPIPS_CONVS):
        if i in (1, 2):
            h = F.max_pool2d(h, 3, 2)
        h = F.relu(_conv(h, w, f"net.conv{i + 1}", stride=stride, pad=pad))
        feats.append(h)
    return feats


def lpips_distance(w: Dict[str, Tensor], a: Tensor, b: Tensor) -> Tensor:
    total = 0.0
    for i, (fa, fb) in enumerate(zip(lpips_features(w, a), lpips_features(w, b))):
        na = fa / (fa.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = fb / (fb.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        diff = (na - nb).pow(2)
        total = total + (diff * w[f"lin{i}.weight"].view(1, -1, 1, 1)).sum(1, keepdim=True).mean((2, 3), keepdim=True)
    return total


class OracleLPIPS:
    def __init__(self, seed: int = 7):
        self.w = make_weights(lpips_spec(), seed, round_fp16=True)

    def cuda(self, *_a, **_k):
        return self

    def __call__(self, a, b):
        return lpips_distance(self.w, a.cpu(), b.cpu())
