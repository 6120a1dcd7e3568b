"""Device-side plumbing for the native model programs.

* ``Arena``   — size-class pool over torch device tensors.  A recorded program replays in stream
  order, so a buffer can be handed to the next op as soon as its last reader has been emitted;
  reuse keeps the working set of a UNet forward inside L2 / Infinity Cache instead of walking
  through fresh HBM for every op.
* ``Program`` — owner of a C++ launch program (``lb_program_*``): ``with prog.record(): ...`` runs
  the emitting Python ONCE; afterwards ``prog.launch()`` replays ~1000 kernel launches from C++
  (optionally as one hipGraph) with a single ctypes call.
* ``Emitter`` — thin typed helpers that fill the C structs and call the launchers (in record mode
  these calls are captured, in eager mode they launch immediately — same code path).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from ..hip import lib
from ..hip.lib import api, LbGemmParams, LbAttnParams

F16, F32 = torch.float16, torch.float32


class Arena:
    def __init__(self, device):
        self.device = device
        self.free: Dict[int, List[torch.Tensor]] = {}
        self.all: List[torch.Tensor] = []
        self.bytes_allocated = 0

    @staticmethod
    def _cls(nbytes: int) -> int:
        return max(512, (nbytes + 511) // 512 * 512)

    def alloc(self, shape, dtype=F16) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        esz = torch.empty((), dtype=dtype).element_size()
        cls = self._cls(n * esz)
        pool = self.free.get(cls)
        if pool:
            raw = pool.pop()
        else:
            raw = torch.empty(cls, dtype=torch.uint8, device=self.device)
            self.all.append(raw)
            self.bytes_allocated += cls
        t = raw[: n * esz].view(dtype).view(*shape)
        t._lb_raw = raw
        return t

    def release(self, t: Optional[torch.Tensor]) -> None:
        if t is None:
            return
        st = getattr(t, "_lb_chstats", None)     # statistics buffer of a conv output nobody normalised: goes back with it
        if st is not None:
            t._lb_chstats = None
            self.release(st[0])
        raw = getattr(t, "_lb_raw", None)
        if raw is None:
            return
        t._lb_raw = None
        self.free.setdefault(raw.numel(), []).append(raw)


class Program:
    def __init__(self, name: str = "program"):
        self.name = name
        self.handle = api.lb_program_create()
        self.keep: List[object] = []       # tensors / arenas the recorded pointers refer to
        self.graph_ready = False

    def __del__(self):
        try:
            if self.handle:
                api.lb_program_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @contextlib.contextmanager
    def record(self):
        api.lb_program_begin_record(self.handle)
        try:
            yield self
        finally:
            api.lb_program_end_record(self.handle)

    @property
    def num_ops(self) -> int:
        return api.lb_program_num_ops(self.handle)

    def op_names(self) -> List[str]:
        return [api.lb_program_op_name(self.handle, i).decode() for i in range(self.num_ops)]

    def run(self, stream: Optional[int] = None) -> None:
        api.lb_program_run(self.handle, stream if stream is not None else _stream())

    def run_range(self, begin: int, end: int, stream: Optional[int] = None) -> None:
        api.lb_program_run_range(self.handle, begin, end, stream if stream is not None else _stream())

    def instantiate(self) -> None:
        api.lb_program_instantiate(self.handle)
        self.graph_ready = True

    def launch(self, stream: Optional[int] = None) -> None:
        api.lb_program_launch(self.handle, stream if stream is not None else _stream())

    def time_ops(self, stream: Optional[int] = None) -> List[float]:
        """Eager replay with hipEvents between ops -> per-op milliseconds (synchronises)."""
        n = self.num_ops
        buf = (C.c_float * n)()
        api.lb_program_time_ops(self.handle, stream if stream is not None else _stream(), buf)
        return list(buf)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Emitter:
    """Helpers shared by the UNet / VAE / LPIPS program builders.  All tensors are views into an
    ``Arena`` (activations) or packed weights; nothing here synchronises or allocates through
    torch while a program is being recorded except via the arena."""

    def __init__(self, arena: Arena):
        self.arena = arena
        self.device = arena.device
        # one split-K slab region and one GroupNorm workspace shared by all ops of a program
        self.ws_gemm: Optional[torch.Tensor] = None
        self.ws_gn: Optional[torch.Tensor] = None
        # algorithmic work of every emitted contraction, in emission order (bench.py roofline)
        self.gemm_log: List[dict] = []
        self.attn_log: List[dict] = []
        self.norm_log: List[dict] = []      # GroupNorm / LayerNorm launches: algorithmic HBM bytes
        self._retired: List[torch.Tensor] = []
        self.zero_page = torch.zeros(64, dtype=torch.uint8, device=self.device)

    # -- workspaces -------------------------------------------------------------------------
    def _gemm_ws(self, M: int, N: int) -> Optional[torch.Tensor]:
        """Split-K slab region.  The launcher only splits grids of at most 640 64x64 tiles (<= 256 blocks of
        the tile it then picks), so larger problems get no workspace (and cannot split).  A grown workspace never replaces the
        old one in already-recorded ops: superseded buffers are kept alive."""
        if ((M + 63) // 64) * ((N + 63) // 64) > 640:
            return None
        need = api.lb_gemm_workspace_bytes(M, N) // 4
        if self.ws_gemm is None or self.ws_gemm.numel() < need:
            if self.ws_gemm is not None:
                self._retired.append(self.ws_gemm)
            self.ws_gemm = torch.empty(max(need, 1 << 22), dtype=F32, device=self.device)
        return self.ws_gemm

    def _gn_ws(self, B: int, groups: int) -> torch.Tensor:
        need = api.lb_groupnorm_workspace_bytes(B, groups) // 8
        if self.ws_gn is None or self.ws_gn.numel() < need:
            if self.ws_gn is not None:
                self._retired.append(self.ws_gn)
            self.ws_gn = torch.empty(need, dtype=torch.float64, device=self.device)
        return self.ws_gn

    # -- dense ------------------------------------------------------------------------------
    def gemm(self, A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, *, M: int, bias=None,
             residual=None, rowvec=None, rows_per_batch: int = 0, flags: int = 0, alpha: float = 1.0,
             lda: Optional[int] = None, ldc: Optional[int] = None, ldr: Optional[int] = None,
             ld_rowvec: Optional[int] = None, conv: Optional[dict] = None, splitk: bool = True,
             ln: Optional[Tuple[torch.Tensor, float]] = None, ch_stats: Optional[torch.Tensor] = None):
        """``ln`` = (colsum [N] fp32, eps): A is consumed through a LayerNorm folded into this GEMM (LB_GEMM_LN_A; W and
        bias must already carry gamma / beta, see ``NativeUNet._ln_linear``); row statistics from the A fragments in the K loop."""
        p = LbGemmParams()
        N, K = W.shape
        p.A, p.W, p.C = A.data_ptr(), W.data_ptr(), out.data_ptr()
        p.bias, p.residual, p.rowvec = _p(bias), _p(residual), _p(rowvec)
        p.M, p.N, p.K, p.ldw = M, N, K, W.stride(0)
        n_out = N // 2 if flags & lib.GEMM_GEGLU else N
        if conv is not None:
            p.conv = 1
            for k in ("Hin", "Win", "Cin", "Hout", "Wout", "KH", "KW", "stride", "pad", "ups", "ldx"):
                setattr(p, k, int(conv[k]))
            if conv.get("parity") == "all":            # all four sub-pixel convs in ONE halo-kernel launch (W stacked [4][N][K])
                p.scatter = 2
            elif "parity" in conv:                     # one sub-pixel upsampling conv
                p.scatter, p.sc_py, p.sc_px = 1, int(conv["parity"][0]), int(conv["parity"][1])
        else:
            p.lda = lda if lda is not None else K
        p.ldc = ldc if ldc is not None else (M if flags & lib.GEMM_TRANS_OUT else n_out)
        p.ldr = ldr if ldr is not None else n_out
        if rowvec is not None:
            p.ld_rowvec, p.rows_per_batch = ld_rowvec if ld_rowvec is not None else N, rows_per_batch
        p.alpha, p.flags = alpha, flags
        p.zero_page = self.zero_page.data_ptr()
        if ln is not None:
            p.flags |= lib.GEMM_LN_A
            p.ln_colsum, p.ln_eps = ln[0].data_ptr(), float(ln[1])
        if ch_stats is not None:       # halo-tile convs only (see halo_stat_rows): GroupNorm statistics of the stored output
            assert conv is not None and ch_stats.dtype == F32
            p.flags |= lib.GEMM_CH_STATS
            p.ch_stats, p.ch_stats_rows = ch_stats.data_ptr(), ch_stats.shape[1]    # [N][B * rows][2]: checked by the launcher
        if splitk and not (flags & lib.GEMM_GEGLU) and ln is None:
            p.partial = _p(self._gemm_ws(M, N))
        api.lb_gemm_f16(C.byref(p), _stream())
        mult = 4.0 if p.scatter == 2 else 1.0          # (four parities: four times the rows, weights and outputs)
        self.gemm_log.append({"M": M, "N": N, "K": K, "flops": mult * 2.0 * M * N * K, "conv": conv is not None,
                              "bytes": mult * 2.0 * (N * K + M * n_out) + (2.0 * M * K if conv is None else 0.0)})
        return out

    def plan(self, M: int, N: int, K: int, flags: int = 0) -> Tuple[int, int]:
        """(tile, split-K) the library would choose for a plain GEMM emitted by ``gemm`` (lb_gemm_plan)."""
        p = LbGemmParams()
        p.M, p.N, p.K, p.flags = M, N, K, flags
        p.zero_page = self.zero_page.data_ptr()
        if not (flags & lib.GEMM_GEGLU):
            p.partial = _p(self._gemm_ws(M, N))
        t, sk, nb = C.c_int(), C.c_int(), C.c_long()
        api.lb_gemm_plan(C.byref(p), C.byref(t), C.byref(sk), C.byref(nb))
        return t.value, sk.value

    def halo_stat_rows(self, B: int, H: int, W: int, cin: int, cout: int, ks: int = 3) -> int:
        """Rows per sample of the LB_GEMM_CH_STATS buffer ([cout][B * rows] float2, channel-major) if a 3x3 conv (ks = 3) or a one-launch
        sub-pixel upsampler conv (ks = 2) of this geometry runs on the halo-tile kernel - whose epilogue can leave the
        GroupNorm statistics of what it stores - else 0 (the consumer then runs the two-pass GroupNorm)."""
        p = LbGemmParams()
        p.conv, p.M, p.N, p.K = 1, B * H * W, cout, ks * ks * cin
        p.Hin, p.Win, p.Hout, p.Wout, p.Cin, p.KH, p.KW, p.stride, p.ldx = H, W, H, W, cin, ks, ks, 1, cin
        p.pad, p.scatter = (1, 0) if ks == 3 else (0, 2)
        p.zero_page = self.zero_page.data_ptr()
        if ks == 2 and not self.upconv_one_launch(B, H, W, cin, cout):
            return 0
        return int(api.lb_gemm_ch_stat_rows(C.byref(p)))      # the library's own routing + tile constants (0: not a halo launch)

    @staticmethod
    def upconv_one_launch(B: int, H: int, W: int, cin: int, cout: int) -> bool:
        """The halo kernel's sub-pixel form takes this upsampler conv in one launch (else: four implicit-GEMM launches)."""
        shape_ok = cin % 64 == 0 and ((W % 32 == 0 and H % 8 == 0) or (W % 16 == 0 and H % 16 == 0))
        blocks = B * (H * W // 256) * 4 * ((cout + 127) // 128)
        return shape_ok and blocks >= 96

    def groupnorm(self, x: torch.Tensor, out: torch.Tensor, gamma, beta, *, B: int, HW: int, C_: int,
                  eps: float, silu: bool, groups: int = 32, ldx: Optional[int] = None,
                  ch_stats: Optional[Tuple[torch.Tensor, int]] = None):
        """``ch_stats`` = (buffer, rows per sample) left by the conv that produced ``x`` (LB_GEMM_CH_STATS): one pass over x."""
        if ch_stats is not None:
            api.lb_groupnorm_from_stats(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ch_stats[0].data_ptr(),
                                        self._gn_ws(B, groups).data_ptr(), B, HW, C_, ldx or C_, C_, groups, eps, int(silu),
                                        int(x.dtype == F32), int(ch_stats[1]), _stream())
            self.norm_log.append({"op": "lb_groupnorm_from_stats", "bytes": float(B * HW * C_) * (x.element_size() + 2)})
            return out
        api.lb_groupnorm_nhwc(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                              self._gn_ws(B, groups).data_ptr(), B, HW, C_, ldx or C_, C_, groups, eps,
                              int(silu), int(x.dtype == F32), _stream())
        # algorithmic traffic: the statistics pass reads x, the apply pass reads x and writes fp16 y; the one-launch form (round 6: slab in
        # registers) reads x once
        one = bool(api.lb_groupnorm_plan(HW, C_, groups, int(x.dtype == F32)))
        self.norm_log.append({"op": "lb_groupnorm_nhwc", "bytes": float(B * HW * C_) * ((1 if one else 2) * x.element_size() + 2), "one_launch": one})
        return out

    def layernorm(self, x, out, gamma, beta, *, M: int, C_: int, eps: float = 1e-5):
        api.lb_layernorm_f16(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, C_, C_, C_,
                             eps, _stream())
        self.norm_log.append({"op": "lb_layernorm_f16", "bytes": float(M * C_) * 4})
        return out

    def attention(self, q_ptr: int, k_ptr: int, v_ptr: int, out: torch.Tensor, *, B, H, Sq, Skv, valid,
                  ldq, ldk, ldv, ldo, causal: bool = False, head_dim: int = 64):
        """``head_dim`` 64: lb_attn_fwd_d64 (UNet, CLIP towers); 512: lb_attn_fwd_d512 (VAE mid block, no causal form)."""
        assert head_dim in (64, 512) and not (causal and head_dim == 512)
        p = LbAttnParams()
        p.Q, p.K, p.V, p.O = q_ptr, k_ptr, v_ptr, out.data_ptr()
        p.B, p.H, p.Sq, p.Skv, p.Skv_valid = B, H, Sq, Skv, valid
        p.ldq, p.ldk, p.ldv, p.ldo = ldq, ldk, ldv, ldo
        p.scale = float(head_dim) ** -0.5
        p.causal = int(causal)
        p.zero_page = self.zero_page.data_ptr()
        (api.lb_attn_fwd_d512 if head_dim == 512 else api.lb_attn_fwd_d64)(C.byref(p), _stream())
        self.attn_log.append({"flops": 4.0 * B * H * Sq * valid * head_dim, "Sq": Sq, "Skv": Skv, "head_dim": head_dim})
        return out

    def copy_cols(self, src, dst, *, rows, cols, ld_src, ld_dst, dst_off):
        api.lb_copy_cols_f16(src.data_ptr(), dst.data_ptr(), rows, cols, ld_src, ld_dst, dst_off, _stream())

    def copy(self, dst: torch.Tensor, src: torch.Tensor):
        assert dst.numel() * dst.element_size() == src.numel() * src.element_size()
        api.lb_copy_d2d(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), _stream())
