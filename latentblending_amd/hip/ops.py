"""Tensor-level wrappers over the C-ABI launchers (torch is used for device memory and the
current stream only).  These are what the parity tests and the thin host layer call; the model
programs in ``latentblending_amd.native`` talk to ``lib.api`` directly with raw pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import lib
from .lib import api, LbGemmParams, LbAttnParams

F16, F32, F64 = torch.float16, torch.float32, torch.float64


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return C.cast(arr, lib.c_void_pp), arr


# ------------------------------------------------------------------------------ mixing -----
def slerp_pairs(p0: List[torch.Tensor], p1: List[torch.Tensor], fracts: List[float]) -> List[torch.Tensor]:
    assert len(p0) == len(p1) == len(fracts) and len(p0) > 0
    dt = p0[0].dtype
    if dt not in (F16, F32, F64):            # bf16 etc.: the reference would return fp32 as well
        p0 = [t.float() for t in p0]
        p1 = [t.float() for t in p1]
        dt = F32
    a = [t.contiguous() for t in p0]
    b = [t.to(dt).contiguous() for t in p1]
    n = a[0].numel()
    assert all(t.numel() == n and t.is_cuda for t in a + b), "slerp_pairs: equal-sized device tensors"
    outs = [torch.empty(t.shape, dtype=F16 if dt == F16 else F32, device=t.device) for t in a]
    pa, keep_a = _ptr_array(a)
    pb, keep_b = _ptr_array(b)
    po, keep_o = _ptr_array(outs)
    fr = (C.c_double * len(fracts))(*[float(f) for f in fracts])
    fn = {F16: api.lb_slerp_pairs_f16, F32: api.lb_slerp_pairs_f32, F64: api.lb_slerp_pairs_f64}[dt]
    fn(pa, pb, po, fr, len(a), n, stream_ptr())
    return outs


def slerp(p0: torch.Tensor, p1: torch.Tensor, fract: float) -> torch.Tensor:
    return slerp_pairs([p0], [p1], [fract])[0]


def slerp_batched(p0: torch.Tensor, p1: torch.Tensor, fracts_dev: torch.Tensor) -> torch.Tensor:
    """p0, p1: [npairs, n] fp16 contiguous; fracts_dev: float64 [npairs] on device."""
    assert p0.dtype == F16 and p0.is_contiguous() and p1.is_contiguous() and fracts_dev.dtype == F64
    out = torch.empty_like(p0)
    api.lb_slerp_batched_f16(p0.data_ptr(), p1.data_ptr(), out.data_ptr(), fracts_dev.data_ptr(),
                             p0.shape[0], p0.shape[1], stream_ptr())
    return out


def slerp_strided(p0: torch.Tensor, p1: torch.Tensor, fracts_dev: torch.Tensor, n: int,
                  broadcast0: bool = False, broadcast1: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[g] = slerp(p0[g or 0], p1[g or 0], fracts_dev[g]) for g < len(fracts_dev); inputs fp16, contiguous
    [G, n] (or one tensor of n elements when broadcast); fracts_dev float64 on device.  One launch, no host pointers."""
    G = fracts_dev.numel()
    assert p0.dtype == F16 and p1.dtype == F16 and fracts_dev.dtype == F64 and p0.is_contiguous() and p1.is_contiguous()
    if out is None:
        out = torch.empty(G, n, dtype=F16, device=p0.device)
    api.lb_slerp_strided_f16(p0.data_ptr(), 0 if broadcast0 else n, p1.data_ptr(), 0 if broadcast1 else n, out.data_ptr(),
                             fracts_dev.data_ptr(), G, n, stream_ptr())
    return out


def lerp(p0: torch.Tensor, p1: torch.Tensor, fract: float) -> torch.Tensor:
    a, b = p0.contiguous(), p1.contiguous()
    if a.dtype == F16 and b.dtype == F16:
        out = torch.empty_like(a)
        api.lb_lerp_f16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), float(fract), stream_ptr())
        return out
    a, b = a.float(), b.float()
    out = torch.empty_like(a)
    api.lb_lerp_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), float(fract), stream_ptr())
    return out


# ------------------------------------------------------------------------------ scheduler --
def step_params(rows: Sequence[Sequence[float]], device) -> torch.Tensor:
    """rows of (sigma_from, sigma_next, sigma_up, guidance, dt) -> float32 [B, 8] on device."""
    # (ONE host tensor from nested lists - a cfg-2 wavefront uploads 34 + 8 rows: the per-row tensor constructions of rounds 1-5 cost
    #  ~0.1 ms of host time in front of a transition's first launch)
    t = torch.tensor([[float(v) for v in r] + [0.0] * (8 - len(r)) for r in rows], dtype=F32).reshape(len(rows), 8)
    return t.to(device)


def scale_model_input(x: torch.Tensor, params: torch.Tensor, dup_for_cfg: bool = False) -> torch.Tensor:
    B = x.shape[0]
    out = torch.empty((2 * B if dup_for_cfg else B,) + tuple(x.shape[1:]), dtype=F16, device=x.device)
    api.lb_scale_model_input_f16(x.data_ptr(), out.data_ptr(), params.data_ptr(), x[0].numel(), B,
                                 int(dup_for_cfg), stream_ptr())
    return out


def euler_step(x, eps, params, noise=None, cfg=False, ancestral=False) -> torch.Tensor:
    assert x.is_contiguous() and eps.is_contiguous(), "euler_step: the kernel takes its operands by pointer"
    if noise is not None and not noise.is_contiguous():
        noise = noise.contiguous()
    out = torch.empty_like(x)
    api.lb_euler_step_f16(x.data_ptr(), eps.data_ptr(), _ptr(noise), out.data_ptr(), params.data_ptr(),
                          x[0].numel(), x.shape[0], int(cfg), int(ancestral), stream_ptr())
    return out


def ddim_step(x, eps, params, cfg=False) -> torch.Tensor:
    """DDIM step (eta = 0); ``params`` rows as NativeDDIMScheduler.step_row builds them."""
    out = torch.empty_like(x)
    api.lb_ddim_step_f16(x.data_ptr(), eps.data_ptr(), out.data_ptr(), params.data_ptr(), x[0].numel(), x.shape[0], int(cfg),
                         stream_ptr())
    return out


# ------------------------------------------------------------------------------ GEMM / conv
def pack_linear_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> fp16 [N, K] contiguous (K padded to a multiple of 8 with zeros)."""
    n, k = w.shape
    kp = (k + 7) // 8 * 8
    out = torch.zeros(n, kp, dtype=F16, device=w.device)
    out[:, :k] = w.to(F16)
    return out


def pack_conv_weight(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    """[Cout, Cin, KH, KW] -> fp16 [Cout, KH*KW*Cin_pad] with (ky, kx, cin) K order."""
    cout, cin, kh, kw = w.shape
    cp = cin_pad or (cin + 7) // 8 * 8
    out = torch.zeros(cout, kh, kw, cp, dtype=F16, device=w.device)
    out[..., :cin] = w.permute(0, 2, 3, 1).to(F16)
    return out.reshape(cout, kh * kw * cp)


def subpixel_upsample_weights(w: torch.Tensor, cin_pad: Optional[int] = None):
    """3x3 conv weight [Cout, Cin, 3, 3] applied after a nearest-2x upsample  ->  four packed 2x2 kernels
    {(py, px): fp16 [Cout, 2*2*Cin_pad]} acting on the low-res grid: output pixel (2y+py, 2x+px) sees only
    a 2x2 low-res neighbourhood, the taps that fall on the same source pixel are summed (in fp32)."""
    cout, cin, _, _ = w.shape
    cp = cin_pad or (cin + 7) // 8 * 8
    w = w.float()
    rows = {0: [w[:, :, 0], w[:, :, 1] + w[:, :, 2]], 1: [w[:, :, 0] + w[:, :, 1], w[:, :, 2]]}   # [Cout,Cin,3(kx)]
    out = {}
    for py in (0, 1):
        for px in (0, 1):
            k = torch.zeros(cout, 2, 2, cp, dtype=torch.float32, device=w.device)
            for a in (0, 1):
                r = rows[py][a]                                            # [Cout, Cin, 3]
                cols = [r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if px == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]]
                for b in (0, 1):
                    k[:, a, b, :cin] = cols[b]
            out[(py, px)] = k.reshape(cout, 4 * cp).to(F16).contiguous()
    return out


def gemm(A: torch.Tensor, W: torch.Tensor, bias=None, residual=None, rowvec=None, rows_per_batch=0,
         flags: int = 0, alpha: float = 1.0, out: Optional[torch.Tensor] = None, splitk_ws: bool = True,
         conv: Optional[dict] = None, M: Optional[int] = None, ln=None, ch_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(A . W^T).  A: [M, K] fp16 (or NHWC [B,H,W,C] with ``conv``), W: [N, K] fp16.
    ``conv``: dict(KH, KW, stride, pad, ups) for an implicit-GEMM convolution.
    ``ln`` = (colsum [N] fp32, eps): LayerNorm over A's columns folded into the GEMM (see ``fold_layernorm``)."""
    p = LbGemmParams()
    N, K = W.shape
    dev = A.device
    if conv is not None:
        B, H, Wd, Cin = A.shape
        ups = int(conv.get("ups", 0))
        kh, kw, st, pad = conv["KH"], conv["KW"], conv.get("stride", 1), conv.get("pad", 0)
        hout = ((H << ups) + 2 * pad - kh) // st + 1
        wout = ((Wd << ups) + 2 * pad - kw) // st + 1
        Mv = B * hout * wout
        p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout = 1, H, Wd, Cin, hout, wout
        p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = kh, kw, st, pad, ups, A.stride(2)
        out_shape = (B, hout, wout)
        if "parity" in conv:                          # sub-pixel upsampling conv: caller supplies `out`
            if conv["parity"] == "all":
                p.scatter = 2                         # W = parity 0 of a stacked [4][N][K] tensor
            else:
                p.scatter, p.sc_py, p.sc_px = 1, int(conv["parity"][0]), int(conv["parity"][1])
            hout, wout = H, Wd
            Mv = B * H * Wd
            p.Hout, p.Wout = H, Wd
    else:
        Mv = M if M is not None else A.shape[0]
        p.lda = A.stride(0)
        out_shape = (Mv,)
    n_out = N // 2 if flags & lib.GEMM_GEGLU else N
    if out is None:
        odt = F32 if flags & lib.GEMM_OUT_F32 else F16
        if flags & lib.GEMM_TRANS_OUT:
            out = torch.empty(n_out, Mv, dtype=odt, device=dev)
        else:
            out = torch.empty(out_shape + (n_out,), dtype=odt, device=dev)
    p.A, p.W, p.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
    p.bias, p.residual, p.rowvec = _ptr(bias), _ptr(residual), _ptr(rowvec)
    p.M, p.N, p.K, p.ldw = Mv, N, K, W.stride(0)
    p.ldc = out.stride(0) if (flags & lib.GEMM_TRANS_OUT) else out.stride(-2)
    if residual is not None:
        p.ldr = residual.stride(-2)
    if rowvec is not None:
        p.ld_rowvec, p.rows_per_batch = rowvec.stride(0), rows_per_batch
    p.alpha, p.flags = alpha, flags
    zp = zero_page(dev)
    p.zero_page = zp.data_ptr()
    if ln is not None:
        p.flags |= lib.GEMM_LN_A
        p.ln_colsum, p.ln_eps = ln[0].data_ptr(), float(ln[1])
    if ch_stats is not None:                          # halo-tile convs only: GroupNorm statistics of the stored output
        p.flags |= lib.GEMM_CH_STATS
        p.ch_stats, p.ch_stats_rows = ch_stats.data_ptr(), ch_stats.shape[1]      # [N][B * rows][2] fp32
    ws = None
    if splitk_ws and not (flags & lib.GEMM_GEGLU) and ln is None:
        ws = torch.empty(api.lb_gemm_workspace_bytes(Mv, N) // 4, dtype=F32, device=dev)
        p.partial = ws.data_ptr()
    if conv is not None and conv.get("halo"):         # experimental halo-tile 3x3 kernel (csrc/conv3_halo.hip)
        api.lb_conv3x3_halo_f16(C.byref(p), stream_ptr())
    else:
        api.lb_gemm_f16(C.byref(p), stream_ptr())
    return out


def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """(W', colsum, b') for ``gemm(..., ln=(colsum, eps))``: LN(x) W^T + b = rstd (x W'^T - mean colsum) + b'."""
    wf = (w.double() * gamma.double()[None, :]).to(F16)
    b2 = w.double() @ beta.double()
    if bias is not None:
        b2 = b2 + bias.double()
    return wf, wf.double().sum(dim=1).to(F32), b2.to(F32)


# ------------------------------------------------------------------------------ norms ------
def groupnorm_nhwc(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                   silu: bool) -> torch.Tensor:
    """x: [B, H, W, C] (or [B, HW, C]) fp16 / fp32 -> fp16."""
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty(x.shape, dtype=F16, device=x.device)
    ws = torch.empty(api.lb_groupnorm_workspace_bytes(B, groups) // 8, dtype=F64, device=x.device)
    api.lb_groupnorm_nhwc(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(),
                          B, HW, C, C, C, groups, eps, int(silu), int(x.dtype == F32), stream_ptr())
    return y


def conv_halo_plan(B: int, H: int, W: int, cin: int, cout: int, ks: int = 3):
    """(kind, tile width, work items, grid) of the halo-tile kernel for this conv geometry (kind 0: not eligible)."""
    p = LbGemmParams()
    p.conv, p.M, p.N, p.K = 1, B * H * W, cout, ks * ks * cin
    p.Hin, p.Win, p.Hout, p.Wout, p.Cin, p.KH, p.KW, p.stride, p.ldx = H, W, H, W, cin, ks, ks, 1, cin
    p.pad, p.scatter = (1, 0) if ks == 3 else (0, 2)
    p.zero_page = 64
    kind, tw, items, grid = C.c_int(), C.c_int(), C.c_long(), C.c_long()
    api.lb_conv_halo_plan(C.byref(p), C.byref(kind), C.byref(tw), C.byref(items), C.byref(grid))
    return kind.value, tw.value, items.value, grid.value


def conv_ch_stat_rows(B: int, H: int, W: int, cin: int, cout: int, ks: int = 3) -> int:
    """Row blocks per sample of the LB_GEMM_CH_STATS buffer a 3x3 conv (ks = 3) / one-launch sub-pixel upsampler conv (ks = 2) of
    this geometry writes when ``gemm`` launches it (lb_gemm_ch_stat_rows: the library's own routing and tile constants); 0 = it
    does not run on a halo-tile kernel."""
    p = LbGemmParams()
    p.conv, p.M, p.N, p.K = 1, B * H * W, cout, ks * ks * cin
    p.Hin, p.Win, p.Hout, p.Wout, p.Cin, p.KH, p.KW, p.stride, p.ldx = H, W, H, W, cin, ks, ks, 1, cin
    p.pad, p.scatter = (1, 0) if ks == 3 else (0, 2)
    p.zero_page = 64
    return int(api.lb_gemm_ch_stat_rows(C.byref(p)))


def groupnorm_from_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, silu: bool,
                         ch_stats: torch.Tensor, rows_per_sample: int) -> torch.Tensor:
    """GroupNorm of x [B, H, W, C] whose (sum, sum of squares) per (64-pixel row block, channel) the producing conv left in
    ``ch_stats`` ([C, B * rows_per_sample, 2] fp32, channel-major, LB_GEMM_CH_STATS)."""
    B, C_ = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C_)
    y = torch.empty(x.shape, dtype=F16, device=x.device)
    ws = torch.empty(api.lb_groupnorm_workspace_bytes(B, groups) // 8, dtype=F64, device=x.device)
    api.lb_groupnorm_from_stats(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ch_stats.data_ptr(), ws.data_ptr(),
                                B, HW, C_, C_, C_, groups, eps, int(silu), int(x.dtype == F32), rows_per_sample, stream_ptr())
    return y


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    M, Cc = x.shape
    y = torch.empty_like(x)
    api.lb_layernorm_f16(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, Cc,
                         x.stride(0), y.stride(0), eps, stream_ptr())
    return y


# ------------------------------------------------------------------------------ attention --
_ZERO_PAGES = {}


def zero_page(device) -> torch.Tensor:
    """64 zero bytes per device: the source of masked direct-to-LDS loads (GEMM tails, attention rows >= Skv)."""
    key = str(device)
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros(64, dtype=torch.uint8, device=device)
    return _ZERO_PAGES[key]


def attention_d64(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Sq: int, Skv: int,
                  skv_valid: Optional[int] = None, ldq=None, ldk=None, ldv=None,
                  out: Optional[torch.Tensor] = None, causal: bool = False) -> torch.Tensor:
    """q: [B*Sq, >=H*64], k, v: [B*Skv, >=H*64] (row strides may exceed widths: slices of a fused QKV buffer)."""
    p = LbAttnParams()
    if out is None:
        out = torch.empty(B * Sq, H * 64, dtype=F16, device=q.device)
    p.Q, p.K, p.V, p.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.B, p.H, p.Sq, p.Skv, p.Skv_valid = B, H, Sq, Skv, skv_valid or Skv
    p.ldq, p.ldk, p.ldv, p.ldo = ldq or q.stride(0), ldk or k.stride(0), ldv or v.stride(0), out.stride(0)
    p.scale = 0.125
    p.causal = int(causal)
    p.zero_page = zero_page(q.device).data_ptr()
    api.lb_attn_fwd_d64(C.byref(p), stream_ptr())
    return out


def attention_d512(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, H: int, Sq: int, Skv: int,
                   skv_valid: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Head dim 512 (the VAE mid-block attention): q [B*Sq, >=H*512], k, v [B*Skv, >=H*512]; one launch, no S x S buffer."""
    p = LbAttnParams()
    if out is None:
        out = torch.empty(B * Sq, H * 512, dtype=F16, device=q.device)
    p.Q, p.K, p.V, p.O = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    p.B, p.H, p.Sq, p.Skv, p.Skv_valid = B, H, Sq, Skv, skv_valid or Skv
    p.ldq, p.ldk, p.ldv, p.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    p.scale = 512.0 ** -0.5
    p.causal = 0
    p.zero_page = zero_page(q.device).data_ptr()
    api.lb_attn_fwd_d512(C.byref(p), stream_ptr())
    return out


def softmax_rows_(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    M, N = x.shape
    api.lb_softmax_rows_f16(x.data_ptr(), M, N, x.stride(0), scale, stream_ptr())
    return x


# ------------------------------------------------------------------------------ small ------
def sinusoid(vals: torch.Tensor, dim: int, out: Optional[torch.Tensor] = None, col_off: int = 0) -> torch.Tensor:
    """vals: float32 [rows, per_row] on device -> fp16 [rows, per_row*dim] ([cos|sin] per value)."""
    rows, per_row = vals.shape
    if out is None:
        out = torch.empty(rows, per_row * dim, dtype=F16, device=vals.device)
    api.lb_sinusoid_f16(vals.data_ptr(), rows, per_row, vals.stride(0), dim, out.data_ptr(), out.stride(0),
                        col_off, stream_ptr())
    return out


def copy_cols(src: torch.Tensor, dst: torch.Tensor, dst_off: int):
    rows = src.numel() // src.shape[-1]
    api.lb_copy_cols_f16(src.data_ptr(), dst.data_ptr(), rows, src.shape[-1], src.stride(-2), dst.stride(-2),
                         dst_off, stream_ptr())


def nchw_to_nhwc(x: torch.Tensor, ld: int, mul: float = 1.0) -> torch.Tensor:
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, ld, dtype=F16, device=x.device)
    api.lb_nchw_to_nhwc_f16(x.contiguous().data_ptr(), y.data_ptr(), B, Cc, H * W, ld, mul, stream_ptr())
    return y


def nhwc_to_nchw(x: torch.Tensor, channels: int) -> torch.Tensor:
    B, H, W, ld = x.shape
    y = torch.empty(B, channels, H, W, dtype=F16, device=x.device)
    api.lb_nhwc_to_nchw_f16(x.data_ptr(), y.data_ptr(), B, channels, H * W, ld, stream_ptr())
    return y


def postprocess_u8(x: torch.Tensor) -> torch.Tensor:
    """x: [B, H, W, ld>=3] fp16/fp32 -> uint8 [B, H, W, 3]."""
    B, H, W, ld = x.shape
    out = torch.empty(B, H, W, 3, dtype=torch.uint8, device=x.device)
    api.lb_postprocess_u8(x.data_ptr(), out.data_ptr(), B * H * W, ld, int(x.dtype == F32), stream_ptr())
    return out


def maxpool3s2(x: torch.Tensor) -> torch.Tensor:
    N, H, W, Cc = x.shape
    y = torch.empty(N, (H - 3) // 2 + 1, (W - 3) // 2 + 1, Cc, dtype=F16, device=x.device)
    api.lb_maxpool3s2_nhwc_f16(x.data_ptr(), y.data_ptr(), N, H, W, Cc, stream_ptr())
    return y


def frames_lerp_u8(frames: torch.Tensor, left, weights) -> torch.Tensor:
    """frames: [n_key, H, W, 3] uint8 on the device; out[k] = uint8((1 - w[k]) * frames[left[k]] + w[k] * frames[left[k] + 1])
    in float64 with a truncating cast (reference utils.py:97 as numpy >= 2 evaluates it on the movie's key frames)."""
    assert frames.dtype == torch.uint8 and frames.is_contiguous()
    n_out = len(left)
    fb = frames[0].numel()
    left_d = torch.tensor(list(left), dtype=torch.int32, device=frames.device)
    w_d = torch.tensor(list(weights), dtype=F64, device=frames.device)
    out = torch.empty((n_out,) + tuple(frames.shape[1:]), dtype=torch.uint8, device=frames.device)
    for k0 in range(0, n_out, 65535):
        k1 = min(n_out, k0 + 65535)
        api.lb_frames_lerp_u8(frames.data_ptr(), left_d[k0:].data_ptr(), w_d[k0:].data_ptr(), out[k0:].data_ptr(), k1 - k0, fb,
                              stream_ptr())
    return out
