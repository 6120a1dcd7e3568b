"""Small-M (B = 2 anchor phase) GEMM study: (tile, ring stages, split-K) over the shapes of the B = 2 UNet step program,
with COLD weights - every launch of the timed graph reads a different weight matrix (as the model does: 5.1 GB of
weights per forward never stay in the 256 MB Infinity Cache), bias + residual epilogue as in situ.
Usage (GPU box): python tools/small_m_sweep.py > gpurun_out/small_m_sweep.txt   (LB_SWEEP_SET=b17: the B = 17 shapes, LB_SWEEP_NW=24)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from latentblending_amd.native.runtime import Program

DEV = "cuda"
NW = int(os.environ.get("LB_SWEEP_NW", "48"))        # distinct weight matrices per timed graph


def timed_graph(ps, tile, splitk, stages):
    lib.api.lb_gemm_set_halo(0)
    lib.api.lb_gemm_set_tuning(tile, splitk)
    lib.api.lb_gemm_set_variant(-1 if (tile == 0 and stages == 0) else 1, stages)
    prog = Program("sweep")
    try:
        with prog.record():
            for p in ps:
                lib.api.lb_gemm_f16(C.byref(p), 0)
    finally:
        lib.api.lb_gemm_set_tuning(0, 0)
        lib.api.lb_gemm_set_variant(-1, 0)
        lib.api.lb_gemm_set_halo(1)
    prog.instantiate()
    st = torch.cuda.current_stream().cuda_stream
    prog.launch(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        prog.launch(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * len(ps)) * 1e3, prog


def main():
    shapes = [("lin", 512, 1280, 1280), ("lin", 512, 1280, 5120), ("lin", 512, 3840, 1280), ("geglu", 512, 10240, 1280),
              ("lin", 2048, 640, 640), ("lin", 2048, 640, 2560), ("lin", 2048, 1920, 640), ("geglu", 2048, 5120, 640),
              ("lin", 8192, 320, 640), ("lin", 512, 1280, 2560), ("conv", 512, 1280, 11520), ("conv", 2048, 640, 5760)]
    if os.environ.get("LB_SWEEP_SET") == "b17":     # the B = 17 programs' shapes that run on 128-wide tiles or the 192x128 one: does a deeper
        # ring pay once every CU is busy?  ("open hypothesis"; tiles 4 / 5 / 7 have fixed depths and are the `auto` column)
        shapes = [("lin", 4352, 1280, 1280), ("lin", 17408, 640, 640), ("lin", 17408, 1920, 640), ("lin", 17408, 640, 2560),
                  ("lin", 4352, 1280, 5120), ("lin", 4352, 3840, 1280), ("lin", 69632, 320, 640)]
    if len(sys.argv) > 1:
        shapes = [s for s in shapes if f"{s[1]}x{s[2]}x{s[3]}" in sys.argv[1:]]
    zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
    for kind, M, N, K in shapes:
        geglu = kind == "geglu"
        nout = N // 2 if geglu else N
        conv = kind == "conv"
        if conv:
            cin = K // 9
            side = 16 if M == 512 else 32
            A = torch.randn(2, side, side, cin, device=DEV).half()
        else:
            A = torch.randn(M, K, device=DEV).half()
        Ws = [(torch.randn(N, K, device=DEV) * K ** -0.5).half() for _ in range(NW)]
        bias = torch.randn(N, device=DEV)
        res = torch.randn(M, nout, device=DEV).half()
        out = torch.empty(M, nout, device=DEV, dtype=torch.float16)
        ws = torch.empty(lib.api.lb_gemm_workspace_bytes(M, N) // 4, dtype=torch.float32, device=DEV)
        ps = []
        for w in Ws:
            p = lib.LbGemmParams()
            p.A, p.W, p.C, p.lda = A.data_ptr(), w.data_ptr(), out.data_ptr(), K
            p.M, p.N, p.K, p.ldw, p.ldc, p.ldr = M, N, K, K, nout, nout
            p.bias = bias.data_ptr()
            if not geglu:
                p.residual = res.data_ptr()
                p.partial = ws.data_ptr()
            p.flags = lib.GEMM_GEGLU if geglu else 0
            p.zero_page = zp.data_ptr()
            if conv:
                p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = 1, side, side, cin, side, side, 3, 3, 1, 1, 0, cin
            ps.append(p)
        flops = 2.0 * M * N * K
        auto, _ = timed_graph(ps, 0, 0, 0)
        row = {}
        for tile in (3, 2, 1):
            for stages in (2, 3, 4):
                for sk in ((1,) if geglu else (1, 2, 4)):
                    if K // 64 // sk < 2:
                        continue
                    us, _ = timed_graph(ps, tile, sk, stages)
                    row[f"t{tile}s{stages}k{sk}"] = us
        best = sorted(row.items(), key=lambda kv: kv[1])
        print(f"{kind} M{M} N{N} K{K}: auto {auto:6.1f} us ({flops / auto / 1e6:5.0f} TF) | " +
              " ".join(f"{k}:{v:.1f}" for k, v in best[:8]) + f" | worst {best[-1][0]}:{best[-1][1]:.1f}", flush=True)
        del Ws


if __name__ == "__main__":
    main()
