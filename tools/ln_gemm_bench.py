"""LayerNorm + GEMM as two launches vs the LayerNorm folded into the GEMM (LB_GEMM_LN_A) with the row statistics
accumulated inside the K loop ("fused") or read from a buffer the producing GEMM wrote ("stats"), on the transformer
shapes of the UNet at B = 17 and B = 2, per tile choice; hipGraph-timed (tools/bench_round2.graph_time).  Also the cost
of LB_GEMM_ROW_STATS on the producer side (N = K = C output projection with residual)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib, ops as o
from tools.bench_round2 import graph_time

DEV = "cuda"


def params(A, W, out, bias, flags, ln=None, ws=None, ln_stats=None, row_stats=None, residual=None):
    p = lib.LbGemmParams()
    p.A, p.W, p.C, p.bias = A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
    p.M, p.N, p.K, p.lda, p.ldw, p.ldc = A.shape[0], W.shape[0], W.shape[1], A.shape[1], W.shape[1], out.shape[1]
    p.flags = flags
    p.zero_page = o.zero_page(DEV).data_ptr()
    if residual is not None:
        p.residual, p.ldr = residual.data_ptr(), residual.shape[1]
    if ln is not None:
        p.flags |= lib.GEMM_LN_A
        p.ln_colsum, p.ln_eps = ln.data_ptr(), 1e-5
        if ln_stats is not None:
            p.row_stats, p.ln_nslots = ln_stats.data_ptr(), W.shape[1] // 32
    elif row_stats is not None:
        p.flags |= lib.GEMM_ROW_STATS
        p.row_stats = row_stats.data_ptr()
    elif ws is not None:
        p.partial = ws.data_ptr()
    return p


def main():
    rows = []
    for B in (17, 2):
        for (S, Cc) in ((256, 1280), (1024, 640)):
            M = B * S
            for (N, geglu, tag) in ((3 * Cc, False, "qkv"), (Cc, False, "to_q"), (8 * Cc, True, "geglu")):
                x = torch.randn(M, Cc, device=DEV).half()
                y = torch.empty_like(x)
                W = (torch.randn(N, Cc, device=DEV) * Cc ** -0.5).half()
                bias = torch.zeros(N, device=DEV)
                cs = W.float().sum(1)
                g = torch.ones(Cc, device=DEV); bt = torch.zeros(Cc, device=DEV)
                out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
                fl = lib.GEMM_GEGLU if geglu else 0
                small = ((M + 63) // 64) * ((N + 63) // 64) <= 640 and not geglu
                ws = torch.empty(lib.api.lb_gemm_workspace_bytes(M, N) // 4, dtype=torch.float32, device=DEV) if small else None
                t_ln = graph_time(lambda: lib.api.lb_layernorm_f16(x.data_ptr(), y.data_ptr(), g.data_ptr(), bt.data_ptr(), M, Cc, Cc, Cc, 1e-5, 0))
                res = {"shape": f"{tag} M{M} N{N} K{Cc}", "ln_us": t_ln}
                stats = torch.zeros(Cc // 32, M, 2, device=DEV)
                stats[..., 1] = 32.0                                  # (sum 0, sum of squares 32 per slot: unit variance)
                for tile in (0, 4, 5, 7):
                    lib.api.lb_gemm_set_tuning(tile, 0)
                    try:
                        ps = params(x, W, out, bias, fl, ln=cs, ln_stats=stats)
                        res[f"lnstats_t{tile}"] = graph_time(lambda: lib.api.lb_gemm_f16(C.byref(ps), 0))
                    finally:
                        lib.api.lb_gemm_set_tuning(0, 0)
                if tag == "to_q":                                     # producer side: the C x C output projection (+ residual)
                    h = torch.randn(M, Cc, device=DEV).half()
                    pa = params(y, W, h, bias, 0, ws=ws, residual=h)
                    pb = params(y, W, h, bias, 0, row_stats=stats, residual=h)
                    res["producer_plain"] = graph_time(lambda: lib.api.lb_gemm_f16(C.byref(pa), 0))
                    res["producer_stats"] = graph_time(lambda: lib.api.lb_gemm_f16(C.byref(pb), 0))
                for tile in (0, 1, 2, 3, 4, 5):
                    lib.api.lb_gemm_set_tuning(tile, 0)
                    try:
                        pp = params(y, W, out, bias, fl, ws=ws)
                        res[f"plain_t{tile}"] = graph_time(lambda: lib.api.lb_gemm_f16(C.byref(pp), 0))
                        pl = params(x, W, out, bias, fl, ln=cs)
                        res[f"lnfused_t{tile}"] = graph_time(lambda: lib.api.lb_gemm_f16(C.byref(pl), 0))
                    finally:
                        lib.api.lb_gemm_set_tuning(0, 0)
                rows.append(res)
                best_p = min(res[f"plain_t{t}"] for t in range(6)); best_f = min(res[f"lnfused_t{t}"] for t in range(6))
                print(f"{res['shape']:28s} LN {t_ln:6.1f} | plain auto {res['plain_t0']:7.1f} best {best_p:7.1f} | fused auto {res['lnfused_t0']:7.1f} best {best_f:7.1f} | "
                      f"stats auto {res['lnstats_t0']:7.1f} t4 {res['lnstats_t4']:.0f} t5 {res['lnstats_t5']:.0f} t7 {res['lnstats_t7']:.0f} | "
                      + " ".join(f"t{t}:{res[f'plain_t{t}']:.0f}/{res[f'lnfused_t{t}']:.0f}" for t in range(1, 6))
                      + (f" | producer {res['producer_plain']:.1f} -> {res['producer_stats']:.1f} with row stats" if "producer_plain" in res else ""), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/ln_gemm_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
