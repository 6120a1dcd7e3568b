"""Native SDXL VAE decoder (diffusers ``AutoencoderKL.decode`` + ``VaeImageProcessor.postprocess``,
reached from /root/reference/latentblending/diffusers_holder.py:115-143) for gfx950.

Precision plan (replaces the reference's "upcast the whole VAE to fp32 on every decode",
diffusers_holder.py:129-139): the SDXL VAE overflows fp16 in its residual stream, not in its
normalised activations.  GroupNorm(+SiLU) outputs are bounded, so they are the fp16 MFMA operands
(fp32 accumulate); the residual stream itself (and conv outputs that only feed the next GroupNorm)
is kept in one of two formats:
  * ``stream_fp16_scaled`` (default): fp16 holding value * 2^-4 — range +-1e6, ~5e-4 relative
    precision, half the HBM traffic of fp32 for the bandwidth-bound GroupNorm passes and conv
    epilogues.  The scale rides along for free: conv epilogues use alpha = 2^-4 and biases
    pre-multiplied by 2^-4 (exact), GroupNorm is scale invariant once eps is scaled by 2^-8, and the
    1x1 shortcut / upsampler convs consume the scaled stream directly (linear ops);
  * fp32 stream (``stream_fp16_scaled=False``): ``OUT_F32`` epilogues, GroupNorm reads fp32, the
    two raw-stream contractions cast with the same 2^-4 scale and multiply back in the epilogue.
Either way the MFMA work runs at the fp16 rate (16x the fp32 MFMA rate on gfx950).
The mid-block attention (1 head, d = 512) is GEMM -> row softmax -> GEMM; its V bias is folded
into the output projection bias at load time (softmax rows sum to 1).
Output is quantised on device to uint8 NHWC frames.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch

from ..hip import lib
from ..hip.lib import api
from .runtime import Arena, Emitter, Program, F16, F32, _stream
from .unet import _pad

STREAM_SCALE = 1.0 / 16.0


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.13025
    force_upcast: bool = True
    stream_fp16_scaled: bool = True      # residual stream as fp16 * 2^-4 (False: fp32 stream)
    fuse_gn_stats: bool = True           # GroupNorm statistics from the producing conv's epilogue (LB_GEMM_CH_STATS): one pass over x
    fused_mid_attention: bool = True     # 512-channel mid-block attention as one lb_attn_fwd_d512 launch (no S x S score buffer)

    @property
    def scale_factor(self) -> int:
        return 2 ** (len(self.block_channels) - 1)


class NativeVAEDecoder:
    def __init__(self, cfg: VAEConfig, provider, device="cuda"):
        self.cfg, self.device = cfg, torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        self._load(provider)

    def _dev(self, t, dtype):
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _conv(self, pv, name, cin, cout, k, gain=1.0):
        w = pv.weight(name + ".weight", (cout, cin, k, k), cin * k * k, gain)
        cin_p, cout_p = _pad(cin, 8), _pad(cout, 4)
        packed = torch.zeros(cout_p, k, k, cin_p, dtype=torch.float32)
        packed[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
        self.w[name + ".weight"] = self._dev(packed.reshape(cout_p, k * k * cin_p), F16)
        b = torch.zeros(cout_p, dtype=torch.float32)
        b[:cout] = pv.bias(name + ".bias", cout)
        self.w[name + ".bias"] = self._dev(b, F32)

    def _upconv(self, pv, name, c):
        from ..hip.ops import subpixel_upsample_weights
        w = pv.weight(name + ".weight", (c, c, 3, 3), c * 9, 1.0)
        subs = subpixel_upsample_weights(w, _pad(c, 8))
        for (py, px), k in subs.items():
            self.w[f"{name}.weight.sub{py}{px}"] = k.to(self.device)
        # the same four kernels stacked [4][N][4 Cin] (parity py*2+px) for the one-launch halo form
        self.w[f"{name}.weight.sub4"] = torch.stack([subs[(0, 0)], subs[(0, 1)], subs[(1, 0)], subs[(1, 1)]]).to(self.device).contiguous()
        self.w[name + ".bias"] = self._dev(pv.bias(name + ".bias", c), F32)

    def _norm(self, pv, name, c):
        self.w[name + ".weight"] = self._dev(pv.norm_weight(name + ".weight", c), F32)
        self.w[name + ".bias"] = self._dev(pv.bias(name + ".bias", c), F32)

    def _resnet(self, pv, p, cin, cout):
        self._norm(pv, p + ".norm1", cin)
        self._conv(pv, p + ".conv1", cin, cout, 3)
        self._norm(pv, p + ".norm2", cout)
        self._conv(pv, p + ".conv2", cout, cout, 3, gain=0.5)
        if cin != cout:
            w = pv.weight(p + ".conv_shortcut.weight", (cout, cin, 1, 1), cin, 1.0)
            self.w[p + ".conv_shortcut.weight"] = self._dev(w.reshape(cout, cin), F16)
            self.w[p + ".conv_shortcut.bias"] = self._dev(pv.bias(p + ".conv_shortcut.bias", cout), F32)

    def _load(self, pv):
        cfg = self.cfg
        rev = list(reversed(cfg.block_channels))
        top = rev[0]
        self._conv(pv, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
        self._conv(pv, "decoder.conv_in", cfg.latent_channels, top, 3)
        self._resnet(pv, "decoder.mid_block.resnets.0", top, top)
        a = "decoder.mid_block.attentions.0"
        self._norm(pv, a + ".group_norm", top)
        for nm in ("to_q", "to_k"):
            self.w[f"{a}.{nm}.weight"] = self._dev(pv.weight(f"{a}.{nm}.weight", (top, top), top), F16)
            self.w[f"{a}.{nm}.bias"] = self._dev(pv.bias(f"{a}.{nm}.bias", top), F32)
        wv = pv.weight(a + ".to_v.weight", (top, top), top)
        bv = pv.bias(a + ".to_v.bias", top)
        wo = pv.weight(a + ".to_out.0.weight", (top, top), top, 0.5)
        bo = pv.bias(a + ".to_out.0.bias", top)
        self.w[a + ".to_v.weight"] = self._dev(wv, F16)
        self.w[a + ".to_out.0.weight"] = self._dev(wo, F16)
        self.w[a + ".to_out.0.bias"] = self._dev(bo + wo.half().float() @ bv, F32)   # V bias folded
        if top == 512:                      # fused form: ONE q | k | v projection feeding lb_attn_fwd_d512 (V bias folded as above)
            self.w[a + ".qkv.weight"] = torch.cat([self.w[a + ".to_q.weight"], self.w[a + ".to_k.weight"], self.w[a + ".to_v.weight"]], 0).contiguous()
            self.w[a + ".qkv.bias"] = torch.cat([self.w[a + ".to_q.bias"], self.w[a + ".to_k.bias"],
                                                 torch.zeros_like(self.w[a + ".to_q.bias"])], 0).contiguous()
        self._resnet(pv, "decoder.mid_block.resnets.1", top, top)
        prev = top
        for ui, c in enumerate(rev):
            for li in range(cfg.layers_per_block + 1):
                self._resnet(pv, f"decoder.up_blocks.{ui}.resnets.{li}", prev, c)
                prev = c
            if ui < len(rev) - 1:
                self._upconv(pv, f"decoder.up_blocks.{ui}.upsamplers.0.conv", c)
        self._norm(pv, "decoder.conv_norm_out", rev[-1])
        self._conv(pv, "decoder.conv_out", rev[-1], cfg.out_channels, 3, gain=0.5)
        for key in [k for k in self.w if k.endswith(".bias")]:          # biases in stream units (x 2^-4, exact)
            self.w[key + "_s"] = self.w[key] * STREAM_SCALE

    def build(self, B: int, L: int) -> "VAEProgram":
        return VAEProgram(self, B, L)


class VAEProgram:
    def __init__(self, net: NativeVAEDecoder, B: int, L: int):
        cfg = net.cfg
        self.net, self.B, self.L = net, B, L
        self.scaled = bool(cfg.stream_fp16_scaled)
        self.fuse_gn_stats = bool(getattr(cfg, "fuse_gn_stats", True))
        self.fused_mid_attention = bool(getattr(cfg, "fused_mid_attention", True))
        dev = net.device
        self.arena = Arena(dev)
        self.em = Emitter(self.arena)
        S = L * cfg.scale_factor
        self.z_in = torch.zeros(B, cfg.latent_channels, L, L, dtype=F16, device=dev)
        self.frames = torch.zeros(B, S, S, 3, dtype=torch.uint8, device=dev)
        self.image_f32 = torch.zeros(B, S, S, _pad(cfg.out_channels, 4), dtype=F32, device=dev)
        self._pq = torch.zeros(B, L, L, _pad(cfg.latent_channels, 8), dtype=F16, device=dev)  # pad channels stay 0
        self.prog = Program("vae-decode")
        with self.prog.record():
            self._emit()

    # ---- residual-stream format --------------------------------------------------------------
    # scaled (default): stream tensors hold value * 2^-4 in fp16;  f32: plain fp32 (see module docstring)
    def _stats_for(self, out, B, H, W, cin_p, cout_p, ks=3):
        """GroupNorm statistics from the producing conv's epilogue (LB_GEMM_CH_STATS): when the conv runs on the halo-tile
        kernel, attach a statistics buffer to its output tensor; the GroupNorm that consumes the tensor then skips its
        statistics pass (one read of the activation instead of two).  Returns the buffer or None."""
        rows = self.em.halo_stat_rows(B, H, W, cin_p, cout_p, ks) if self.fuse_gn_stats else 0
        if not rows:
            return None
        buf = self.arena.alloc((cout_p, B * rows, 2), F32)        # channel-major
        out._lb_chstats = (buf, rows)
        return buf

    def _stream_conv(self, x, name, B, H, W, cin, cout, *, residual=None, out=None, stats=True):
        """3x3 conv whose output goes to the residual stream (or feeds only the next GroupNorm)."""
        w, sc = self.net.w, self.scaled
        cin_p, cout_p = _pad(cin, 8), _pad(cout, 4)
        if out is None:
            out = self.arena.alloc((B, H, W, cout_p), F16 if sc else F32)
        flags = 0 if sc else (lib.GEMM_OUT_F32 | (lib.GEMM_RES_F32 if residual is not None else 0))
        st = self._stats_for(out, B, H, W, cin_p, cout_p) if stats else None
        self.em.gemm(x, w[name + ".weight"], out, M=B * H * W, bias=w[name + (".bias_s" if sc else ".bias")],
                     residual=residual, flags=flags, alpha=STREAM_SCALE if sc else 1.0, ch_stats=st,
                     conv=dict(Hin=H, Win=W, Cin=cin_p, Hout=H, Wout=W, KH=3, KW=3, stride=1, pad=1, ups=0, ldx=cin_p))
        return out

    def _gn(self, x, out, name, B, HW, c, silu):
        w = self.net.w
        eps = 1e-6 * (STREAM_SCALE * STREAM_SCALE if self.scaled else 1.0)     # GroupNorm of a scaled tensor
        st = getattr(x, "_lb_chstats", None)                                   # left by the conv that produced x?
        self.em.groupnorm(x, out, w[name + ".weight"], w[name + ".bias"], B=B, HW=HW, C_=c, eps=eps, silu=silu,
                          groups=self.net.cfg.norm_groups, ch_stats=st)
        if st is not None:
            self.arena.release(st[0])
            x._lb_chstats = None

    def _stream_as_operand(self, x, B, H, W, c):
        """The raw stream as an fp16 MFMA operand carrying the 2^-4 scale: free in scaled mode, a
        saturating down-scaling cast in fp32 mode.  Returns (tensor, owned)."""
        if self.scaled:
            return x, False
        x16 = self.arena.alloc((B, H, W, c))
        api.lb_cast_f32_to_f16(x.data_ptr(), x16.data_ptr(), x.numel(), STREAM_SCALE, _stream())
        return x16, True

    def _resnet(self, x, p, B, H, W, cin, cout):
        em, w, ar, sc = self.em, self.net.w, self.arena, self.scaled
        n1 = ar.alloc((B, H, W, cin))
        self._gn(x, n1, p + ".norm1", B, H * W, cin, True)
        h = self._stream_conv(n1, p + ".conv1", B, H, W, cin, cout)
        ar.release(n1)
        n2 = ar.alloc((B, H, W, cout))
        self._gn(h, n2, p + ".norm2", B, H * W, cout, True)
        ar.release(h)
        if cin != cout:
            x16, owned = self._stream_as_operand(x, B, H, W, cin)
            xs = ar.alloc((B, H, W, cout), F16 if sc else F32)
            # operand carries the 2^-4 scale: scaled mode keeps it (bias pre-scaled), fp32 mode multiplies back
            em.gemm(x16, w[p + ".conv_shortcut.weight"], xs, M=B * H * W,
                    bias=w[p + (".conv_shortcut.bias_s" if sc else ".conv_shortcut.bias")],
                    flags=0 if sc else lib.GEMM_OUT_F32, alpha=1.0 if sc else 1.0 / STREAM_SCALE)
            if owned:
                ar.release(x16)
        else:
            xs = x
        out = self._stream_conv(n2, p + ".conv2", B, H, W, cout, cout, residual=xs)
        ar.release(n2)
        if xs is not x:
            ar.release(xs)
        return out

    def _mid_attention(self, h, B, H, W, c):
        em, w, ar, sc = self.em, self.net.w, self.arena, self.scaled
        a = "decoder.mid_block.attentions.0"
        S = H * W
        n = ar.alloc((B * S, c))
        self._gn(h, n, a + ".group_norm", B, S, c, False)
        if self.fused_mid_attention and (a + ".qkv.weight") in w:
            qkv = ar.alloc((B * S, 3 * c))
            em.gemm(n, w[a + ".qkv.weight"], qkv, M=B * S, bias=w[a + ".qkv.bias"])
            ar.release(n)
            o = ar.alloc((B * S, c))
            em.attention(qkv.data_ptr(), qkv.data_ptr() + 2 * c, qkv.data_ptr() + 4 * c, o, B=B, H=1, Sq=S, Skv=S, valid=S,
                         ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, head_dim=512)
            ar.release(qkv)
            return self._attn_out(o, h, B, S)
        q, k = ar.alloc((B * S, c)), ar.alloc((B * S, c))
        em.gemm(n, w[a + ".to_q.weight"], q, M=B * S, bias=w[a + ".to_q.bias"])
        em.gemm(n, w[a + ".to_k.weight"], k, M=B * S, bias=w[a + ".to_k.bias"])
        vt = ar.alloc((c, B * S))
        em.gemm(w[a + ".to_v.weight"], n, vt, M=c)                      # V^T (bias folded into to_out)
        ar.release(n)
        o = ar.alloc((B * S, c))
        scores = ar.alloc((S, S))
        for b in range(B):
            qb, kb, ob = q[b * S:(b + 1) * S], k[b * S:(b + 1) * S], o[b * S:(b + 1) * S]
            em.gemm(qb, kb, scores, M=S, alpha=float(c) ** -0.5)
            api.lb_softmax_rows_f16(scores.data_ptr(), S, S, S, 1.0, _stream())
            em.gemm(scores, vt[:, b * S:(b + 1) * S], ob, M=S, lda=S)    # W = V^T slice [c, S], ldw = B*S
        ar.release(scores); ar.release(q); ar.release(k); ar.release(vt)
        return self._attn_out(o, h, B, S)

    def _attn_out(self, o, h, B, S):
        em, w, ar, sc = self.em, self.net.w, self.arena, self.scaled
        a = "decoder.mid_block.attentions.0"
        em.gemm(o, w[a + ".to_out.0.weight"], h, M=B * S, bias=w[a + (".to_out.0.bias_s" if sc else ".to_out.0.bias")],
                residual=h, flags=0 if sc else (lib.GEMM_OUT_F32 | lib.GEMM_RES_F32), alpha=STREAM_SCALE if sc else 1.0)
        ar.release(o)
        return h

    def _emit(self):
        net, cfg, em, w, ar, B, L = self.net, self.net.cfg, self.em, self.net.w, self.arena, self.B, self.L
        sc = self.scaled
        rev = list(reversed(cfg.block_channels))
        top, lc = rev[0], cfg.latent_channels
        lc_p = _pad(lc, 8)
        z8 = ar.alloc((B, L, L, lc_p))
        api.lb_nchw_to_nhwc_f16(self.z_in.data_ptr(), z8.data_ptr(), B, lc, L * L, lc_p, 1.0 / cfg.scaling_factor,
                                _stream())
        # post_quant_conv (1x1) writes the first `lc` channels of a zero-padded buffer
        em.gemm(z8, w["post_quant_conv.weight"], self._pq, M=B * L * L, bias=w["post_quant_conv.bias"], ldc=lc_p)
        ar.release(z8)
        h = self._stream_conv(self._pq, "decoder.conv_in", B, L, L, lc, top)
        nxt = self._resnet(h, "decoder.mid_block.resnets.0", B, L, L, top, top); ar.release(h); h = nxt
        h = self._mid_attention(h, B, L, L, top)
        nxt = self._resnet(h, "decoder.mid_block.resnets.1", B, L, L, top, top); ar.release(h); h = nxt
        side, prev = L, top
        for ui, c in enumerate(rev):
            for li in range(cfg.layers_per_block + 1):
                nxt = self._resnet(h, f"decoder.up_blocks.{ui}.resnets.{li}", B, side, side, prev, c)
                ar.release(h); h = nxt
                prev = c
            if ui < len(rev) - 1:
                h16, owned = self._stream_as_operand(h, B, side, side, c)
                name = f"decoder.up_blocks.{ui}.upsamplers.0.conv"
                up = ar.alloc((B, 2 * side, 2 * side, c), F16 if sc else F32)
                one = em.upconv_one_launch(B, side, side, c, c)
                if one:                     # all four sub-pixel convs in ONE launch of the halo kernel
                    em.gemm(h16, w[f"{name}.weight.sub4"][0], up, M=B * side * side,
                            bias=w[name + (".bias_s" if sc else ".bias")], ldc=c,
                            flags=0 if sc else lib.GEMM_OUT_F32, alpha=1.0 if sc else 1.0 / STREAM_SCALE,
                            ch_stats=self._stats_for(up, B, side, side, c, c, ks=2),
                            conv=dict(Hin=side, Win=side, Cin=c, Hout=side, Wout=side, KH=2, KW=2, stride=1, pad=0,
                                      ups=0, ldx=c, parity="all"))
                for py in (() if one else (0, 1)):           # else: four 2x2 convs on the low-res grid
                    for px in (0, 1):
                        em.gemm(h16, w[f"{name}.weight.sub{py}{px}"], up, M=B * side * side,
                                bias=w[name + (".bias_s" if sc else ".bias")], ldc=c,
                                flags=0 if sc else lib.GEMM_OUT_F32, alpha=1.0 if sc else 1.0 / STREAM_SCALE,
                                conv=dict(Hin=side, Win=side, Cin=c, Hout=side, Wout=side, KH=2, KW=2, stride=1, pad=0,
                                          ups=0, ldx=c, parity=(py, px)))
                if owned:
                    ar.release(h16)
                ar.release(h)
                h = up
                side *= 2
        n = ar.alloc((B, side, side, rev[-1]))
        self._gn(h, n, "decoder.conv_norm_out", B, side * side, rev[-1], True)
        ar.release(h)
        cin_p = _pad(rev[-1], 8)
        em.gemm(n, w["decoder.conv_out.weight"], self.image_f32, M=B * side * side, bias=w["decoder.conv_out.bias"],
                flags=lib.GEMM_OUT_F32,
                conv=dict(Hin=side, Win=side, Cin=cin_p, Hout=side, Wout=side, KH=3, KW=3, stride=1, pad=1, ups=0, ldx=cin_p))
        ar.release(n)
        api.lb_postprocess_u8(self.image_f32.data_ptr(), self.frames.data_ptr(), B * side * side,
                              _pad(cfg.out_channels, 4), 1, _stream())

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B,4,L,L] fp16 final latents (NOT yet divided by the scaling factor) -> uint8 [B,8L,8L,3]."""
        self.z_in.copy_(z)
        self.prog.launch()
        return self.frames
