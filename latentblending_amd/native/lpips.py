"""Native LPIPS-Alex perceptual distance on device-resident uint8 frames (gfx950).

Replaces ``lpips.LPIPS(net='alex')`` and the host round trips around it in the reference
(/root/reference/latentblending/blending_engine.py:73-76, 744-758: PIL -> numpy -> float ->
``.cuda()`` twice per call, ``float(...)`` sync).  Here frames never leave the GPU:

* ``features(frames_u8)`` runs the 5-conv AlexNet trunk ONCE per frame (implicit-GEMM MFMA convs
  with fused bias+ReLU, NHWC max-pool) and returns the five taps; the engine caches them per
  committed frame, so a transition needs one trunk pass per frame instead of four.
* ``distances(pairs)`` evaluates channel-normalised squared differences, the learned 1x1 "lin"
  weights and the spatial mean for up to 16 pairs per launch, deterministically.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import torch

from ..hip import lib
from ..hip.lib import api
from .runtime import Arena, Emitter, F16, F32, _stream
from .unet import _pad

ALEX_CONVS = [(3, 64, 11, 4, 2), (64, 192, 5, 1, 2), (192, 384, 3, 1, 1), (384, 256, 3, 1, 1), (256, 256, 3, 1, 1)]


class NativeLPIPS:
    def __init__(self, provider, device="cuda"):
        self.device = torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        for i, (cin, cout, k, _, _) in enumerate(ALEX_CONVS):
            name = f"net.conv{i + 1}"
            w = provider.weight(name + ".weight", (cout, cin, k, k), cin * k * k, 1.4)
            cin_p = _pad(cin, 8)
            packed = torch.zeros(cout, k, k, cin_p, dtype=torch.float32)
            packed[..., :cin] = w.permute(0, 2, 3, 1)
            self.w[name + ".weight"] = packed.reshape(cout, k * k * cin_p).to(self.device, F16).contiguous()
            self.w[name + ".bias"] = provider.bias(name + ".bias", cout).to(self.device, F32)
            self.w[f"lin{i}.weight"] = provider.positive(f"lin{i}.weight", cout).to(self.device, F32)
        self.arena = Arena(self.device)
        self.em = Emitter(self.arena)
        self._acc = torch.zeros(16, dtype=F32, device=self.device)
        self._ws = torch.zeros(16 * 128, dtype=F32, device=self.device)

    def features(self, frames_u8: torch.Tensor) -> List[torch.Tensor]:
        """frames_u8: [N, H, W, 3] uint8 on device -> five taps, each [N, Hi*Wi, Ci] fp16."""
        N, H, W, _ = frames_u8.shape
        x = torch.empty(N, H, W, 8, dtype=F16, device=self.device)
        api.lb_lpips_prep_u8(frames_u8.contiguous().data_ptr(), x.data_ptr(), N * H * W, _stream())
        taps = []
        h, hh, ww, cin = x, H, W, 3
        for i, (_, cout, k, stride, pad) in enumerate(ALEX_CONVS):
            if i in (1, 2):
                ho, wo = (hh - 3) // 2 + 1, (ww - 3) // 2 + 1
                pooled = torch.empty(N, ho, wo, cin, dtype=F16, device=self.device)
                api.lb_maxpool3s2_nhwc_f16(h.data_ptr(), pooled.data_ptr(), N, hh, ww, cin, _stream())
                h, hh, ww = pooled, ho, wo
            ho, wo = (hh + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1
            out = torch.empty(N, ho, wo, cout, dtype=F16, device=self.device)
            cin_p = _pad(cin, 8)
            self.em.gemm(h, self.w[f"net.conv{i + 1}.weight"], out, M=N * ho * wo, bias=self.w[f"net.conv{i + 1}.bias"],
                         flags=lib.GEMM_RELU,
                         conv=dict(Hin=hh, Win=ww, Cin=cin_p, Hout=ho, Wout=wo, KH=k, KW=k, stride=stride, pad=pad,
                                   ups=0, ldx=cin_p))
            taps.append(out.view(N, ho * wo, cout))
            h, hh, ww, cin = out, ho, wo, cout
        return taps

    def distances(self, pairs: Sequence[Tuple[List[torch.Tensor], List[torch.Tensor]]]) -> torch.Tensor:
        """pairs of per-frame tap lists (each tap [HWi, Ci]) -> float32 [npairs] on device."""
        out = []
        for base in range(0, len(pairs), 16):
            chunk = pairs[base:base + 16]
            n = len(chunk)
            api.lb_fill_f32(self._acc.data_ptr(), 16, 0.0, _stream())
            for t in range(5):
                hw, c = chunk[0][0][t].shape[-2], chunk[0][0][t].shape[-1]
                pa = (C.c_void_p * n)(*[p[0][t].data_ptr() for p in chunk])
                pb = (C.c_void_p * n)(*[p[1][t].data_ptr() for p in chunk])
                api.lb_lpips_tap(C.cast(pa, lib.c_void_pp), C.cast(pb, lib.c_void_pp), self.w[f"lin{t}.weight"].data_ptr(),
                                 self._acc.data_ptr(), self._ws.data_ptr(), n, hw, c, _stream())
            out.append(self._acc[:n].clone())
        return torch.cat(out)

    def __call__(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """lpips-package calling convention: [1,3,H,W] float tensors in [-1,1] -> [1,1,1,1]."""
        def to_u8(t):
            return ((t.to(self.device).float() + 1) * 127.5).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
        fa = [t[0] for t in self.features(to_u8(a))]
        fb = [t[0] for t in self.features(to_u8(b))]
        return self.distances([(fa, fb)]).view(1, 1, 1, 1)
