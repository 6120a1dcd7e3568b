"""A/B of the L2-prefetch wave (lb_gemm_set_prefetch) on the B = 17 GEMM shapes of the UNet step program, COLD weights
(every launch of the timed graph reads a different weight matrix), bias + residual epilogue as in situ.
Usage (GPU box): python tools/prefetch_ab.py > gpurun_out/prefetch_ab.txt"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from latentblending_amd.native.runtime import Program

DEV = "cuda"
NW = int(os.environ.get("LB_SWEEP_NW", "24"))


def timed_graph(ps, tile, prefetch):
    lib.api.lb_gemm_set_tuning(tile, 0)
    lib.api.lb_gemm_set_prefetch(prefetch)
    prog = Program("ab")
    try:
        with prog.record():
            for p in ps:
                lib.api.lb_gemm_f16(C.byref(p), 0)
    finally:
        lib.api.lb_gemm_set_tuning(0, 0)
        lib.api.lb_gemm_set_prefetch(0)
    prog.instantiate()
    st = torch.cuda.current_stream().cuda_stream
    prog.launch(st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(2):
            prog.launch(st)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / (2 * len(ps)) * 1e3)
    return best


def main():
    shapes = [("lin", 4352, 1280, 1280), ("lin", 4352, 1280, 5120), ("lin", 4352, 3840, 1280), ("geglu", 4352, 10240, 1280),
              ("lin", 17408, 640, 640), ("lin", 17408, 640, 2560), ("lin", 17408, 1920, 640), ("geglu", 17408, 5120, 640),
              ("lin", 1360, 166400, 2048), ("lin", 8192, 8192, 8192)]
    zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
    for kind, M, N, K in shapes:
        geglu = kind == "geglu"
        nout = N // 2 if geglu else N
        nw = max(2, min(NW, int(1.5e9 // (N * K * 2))))
        A = torch.randn(M, K, device=DEV).half()
        Ws = [(torch.randn(N, K, device=DEV) * K ** -0.5).half() for _ in range(nw)]
        bias = torch.randn(N, device=DEV)
        res = torch.randn(M, nout, device=DEV).half()
        out = torch.empty(M, nout, device=DEV, dtype=torch.float16)
        ps = []
        for w in Ws:
            p = lib.LbGemmParams()
            p.A, p.W, p.C, p.lda = A.data_ptr(), w.data_ptr(), out.data_ptr(), K
            p.M, p.N, p.K, p.ldw, p.ldc, p.ldr = M, N, K, K, nout, nout
            p.bias = bias.data_ptr()
            if not geglu and N <= 8192:
                p.residual = res.data_ptr()
            p.flags = lib.GEMM_GEGLU if geglu else 0
            p.zero_page = zp.data_ptr()
            ps.append(p)
        flops = 2.0 * M * N * K
        t = lib.C.c_int() if hasattr(lib, "C") else C.c_int()
        sk, nb = C.c_int(), C.c_long()
        lib.api.lb_gemm_plan(C.byref(ps[0]), C.byref(t), C.byref(sk), C.byref(nb))
        row = []
        for tile in (0, 4, 5, 7):
            if tile == 7 and geglu:
                continue
            for pf in (0, 1):
                us = timed_graph(ps, tile, pf)
                row.append(f"t{tile}{'+pf' if pf else ''}:{us:7.1f}us({flops / us / 1e6:5.0f}TF)")
        print(f"{kind} M{M} N{N} K{K} (auto tile {t.value}, {nw} cold W): " + "  ".join(row), flush=True)
        # same results with and without the prefetch wave
        lib.api.lb_gemm_set_prefetch(0); lib.api.lb_gemm_f16(C.byref(ps[0]), torch.cuda.current_stream().cuda_stream); ref = out.clone()
        lib.api.lb_gemm_set_prefetch(1); lib.api.lb_gemm_f16(C.byref(ps[0]), torch.cuda.current_stream().cuda_stream)
        lib.api.lb_gemm_set_prefetch(0)
        torch.cuda.synchronize()
        assert torch.equal(ref, out), "prefetch wave changed the result"
        del Ws


if __name__ == "__main__":
    main()
