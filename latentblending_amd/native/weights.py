"""Parameter providers for the native SDXL modules.

The model builders walk their architecture and ask a provider for every parameter by its HF
diffusers state-dict name (``down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight``,
``decoder.up_blocks.2.resnets.0.conv1.weight`` ...), so the same walk serves

* ``DictProvider``      — a state dict: a real checkpoint's tensors (``from_safetensors``) or the
  seeded weights a test shares with the CPU oracle;
* ``SyntheticProvider`` — seeded random weights of the right shapes and scales, generated on the
  fly (no checkpoint on disk and no network here; ``bench.py`` says ``"data": "synthetic"``).
  Init: W ~ N(0, gain^2 / fan_in), b ~ N(0, 0.02^2), norm gamma ~ N(1, 0.05^2); residual-branch
  output projections use gain 0.5 so that 70 stacked transformer blocks stay tame in fp16.
  Values are rounded to fp16 so every consumer (fp16 device path, fp32 CPU oracle) sees
  bit-identical parameters.  Seeds depend only on (name, seed): order independent.
"""
from __future__ import annotations

import math
import os
import zlib
from typing import Dict, Optional, Sequence

import numpy as np
import torch


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


def _seeded(name: str, shape: Sequence[int], std: float, seed: int, mean: float = 0.0) -> torch.Tensor:
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


class SyntheticProvider:
    def __init__(self, seed: int = 0, keep: bool = False, cache_file: Optional[str] = None):
        """``keep``: remember every generated tensor in ``self.state`` (HF key -> fp32 tensor), e.g.
        so that a benchmark can time a CPU checker on exactly the weights the device path uses.
        ``cache_file``: a scratch file holding the generated tensors (fp16, exact: every value is rounded to fp16
        anyway).  Drawing 2.6 G normal deviates from one CPU generator takes about a minute per process; profiling
        sessions that start the same benchmark several times load the file instead (``save_cache`` writes it)."""
        self.seed = seed
        self.state: Optional[Dict[str, torch.Tensor]] = {} if keep else None
        self.cache_file = cache_file
        self._cache: Optional[Dict[str, torch.Tensor]] = None
        self._fresh: Dict[str, torch.Tensor] = {}
        if cache_file and os.path.isfile(cache_file):
            self._cache = torch.load(cache_file, map_location="cpu", mmap=True, weights_only=True)

    def save_cache(self) -> None:
        if self.cache_file and self._cache is None and self._fresh:
            tmp = self.cache_file + f".tmp{os.getpid()}"
            torch.save(self._fresh, tmp)
            os.replace(tmp, self.cache_file)
        self._fresh = {}

    def _out(self, name: str, shape: Sequence[int], make) -> torch.Tensor:
        shape = tuple(shape)
        if self._cache is not None and name in self._cache and tuple(self._cache[name].shape) == shape:
            t = self._cache[name].float()
        else:
            h = make().half()
            if self.cache_file and self._cache is None:
                self._fresh[name] = h
            t = h.float()
        if self.state is not None:
            self.state[name] = t
        return t

    def weight(self, name: str, shape: Sequence[int], fan_in: int, gain: float = 1.0) -> torch.Tensor:
        return self._out(name, shape, lambda: _seeded(name, shape, gain / math.sqrt(fan_in), self.seed))

    def bias(self, name: str, n: int) -> torch.Tensor:
        return self._out(name, (n,), lambda: _seeded(name, (n,), 0.02, self.seed))

    def norm_weight(self, name: str, n: int) -> torch.Tensor:
        return self._out(name, (n,), lambda: _seeded(name, (n,), 0.05, self.seed, mean=1.0))

    def positive(self, name: str, n: int) -> torch.Tensor:        # LPIPS lin layers (non-negative)
        return self._out(name, (n,), lambda: _seeded(name, (n,), 1.0, self.seed).abs() / n)


class DictProvider:
    """State dict lookup; shapes are checked, scales/gains ignored."""

    def __init__(self, state: Dict[str, torch.Tensor], prefix: str = ""):
        self.state, self.prefix = state, prefix

    def _get(self, name: str, shape) -> torch.Tensor:
        key = self.prefix + name
        if key not in self.state:
            raise KeyError(f"parameter '{key}' missing from the state dict")
        t = self.state[key]
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"parameter '{key}': expected shape {tuple(shape)}, found {tuple(t.shape)}")
        return t.detach().to(torch.float32)

    def weight(self, name, shape, fan_in=0, gain=1.0):
        return self._get(name, shape)

    def bias(self, name, n):
        return self._get(name, (n,))

    def norm_weight(self, name, n):
        return self._get(name, (n,))

    def positive(self, name, n):
        t = self.state[self.prefix + name]
        return t.detach().to(torch.float32).reshape(n)


def from_safetensors(path: str, prefix: str = "") -> DictProvider:
    """HF-layout ``*.safetensors`` file or directory (e.g. ``unet/diffusion_pytorch_model.fp16.safetensors``)."""
    from safetensors.torch import load_file
    files = [path] if os.path.isfile(path) else sorted(
        os.path.join(path, f) for f in os.listdir(path) if f.endswith(".safetensors"))
    state: Dict[str, torch.Tensor] = {}
    for f in files:
        state.update(load_file(f))
    return DictProvider(state, prefix)


# AlexNet trunk conv indices inside torchvision's ``alexnet().features`` / the slices of ``lpips.pretrained_networks.alexnet``
_ALEX_FEATURE_IDX = (0, 3, 6, 8, 10)


def lpips_provider(alexnet_state: Dict[str, torch.Tensor], lin_state: Dict[str, torch.Tensor]) -> DictProvider:
    """Provider for ``NativeLPIPS`` from REAL checkpoints: the torchvision AlexNet trunk (keys ``features.N.weight`` /
    ``.bias``, or the ``net.sliceK.N.*`` names the same tensors carry inside a saved ``lpips.LPIPS`` module) and the
    learned linear layers of ``lpips`` v0.1 (``linK.model.1.weight``, shape [1, C, 1, 1]; the file
    ``lpips/weights/v0.1/alex.pth``).  The reference builds exactly this net with ``lpips.LPIPS(net='alex')``
    (/root/reference/latentblending/blending_engine.py:73-76)."""
    out: Dict[str, torch.Tensor] = {}
    for i, idx in enumerate(_ALEX_FEATURE_IDX):
        for kind in ("weight", "bias"):
            for key in (f"features.{idx}.{kind}", f"net.slice{i + 1}.{idx}.{kind}", f"net.features.{idx}.{kind}"):
                if key in alexnet_state:
                    out[f"net.conv{i + 1}.{kind}"] = alexnet_state[key]
                    break
            else:
                raise KeyError(f"AlexNet conv {i + 1} ({kind}): none of features.{idx}.{kind} / net.slice{i + 1}.{idx}.{kind} found")
        for key in (f"lin{i}.model.1.weight", f"lin{i}.weight", f"lins.{i}.model.1.weight"):
            if key in lin_state:
                out[f"lin{i}.weight"] = lin_state[key].reshape(-1)
                break
        else:
            raise KeyError(f"LPIPS linear layer {i}: lin{i}.model.1.weight not found")
    return DictProvider(out)
