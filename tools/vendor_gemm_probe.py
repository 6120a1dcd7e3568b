"""Where do the hand-written GEMMs stand against the vendor library on the very shapes of the SDXL programs?  Times
torch.nn.functional.linear (hipBLASLt / rocBLAS behind PyTorch-ROCm) and lb_gemm_f16 on the same operands, weights rotated
through more copies than the Infinity Cache holds (as in the model: 5.1 GB of weights per forward), fp16 in / fp32 accumulate /
fp16 out, bias + residual like the out-projections.  A DIAGNOSTIC (tools/ only): the product calls its own kernels; the numbers
say how much head-room a better main loop has on each shape, and `rocprofv3 --kernel-trace --stats -- python
tools/vendor_gemm_probe.py` names the library's tile configurations.  Prepared at the end of round 3, not yet run.
Usage: python tools/vendor_gemm_probe.py > gpurun_out/vendor_gemm_probe.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from latentblending_amd.hip import ops

DEV = "cuda"
SHAPES = [(4352, 1280, 1280), (4352, 1280, 5120), (4352, 3840, 1280), (4352, 10240, 1280), (17408, 640, 640),
          (17408, 640, 2560), (17408, 1920, 640), (17408, 5120, 640), (512, 1280, 1280), (512, 1280, 5120),
          (512, 3840, 1280), (2048, 640, 640), (8192, 8192, 8192)]


def timed(fn, n):
    """fn(i) for i in range(n), captured ONCE into a graph (no Python / dispatcher time between the launches), replayed 3x."""
    for i in range(min(n, 3)):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3          # us


def main():
    for M, N, K in SHAPES:
        nw = max(2, min(48, int(400e6 // (N * K * 2))))          # > 256 MB of distinct weights where they are small
        x = torch.randn(M, K, device=DEV).half()
        ws = [(torch.randn(N, K, device=DEV) * K ** -0.5).half() for _ in range(nw)]
        bias16, bias32 = torch.randn(N, device=DEV).half(), None
        bias32 = bias16.float()
        res = torch.randn(M, N, device=DEV).half()
        out = torch.empty(M, N, device=DEV, dtype=torch.float16)
        n = nw
        t_lib = timed(lambda i: torch.add(F.linear(x, ws[i % nw], bias16), res, out=out), n)        # library GEMM + bias, then the residual add
        t_lib_plain = timed(lambda i: F.linear(x, ws[i % nw], bias16), n)
        t_own = timed(lambda i: ops.gemm(x, ws[i % nw], out=out, bias=bias32, residual=res), n)
        fl = 2.0 * M * N * K
        tf = lambda us: fl / us / 1e6
        err = float((ops.gemm(x, ws[0], bias=bias32, residual=res).float() - (F.linear(x, ws[0], bias16).float() + res.float())).abs().max())
        print(f"M={M:6d} N={N:6d} K={K:5d}: library {t_lib_plain:8.1f} us ({tf(t_lib_plain):7.1f} TF/s), + residual add {t_lib:8.1f} us; "
              f"lb_gemm_f16 (bias + residual fused) {t_own:8.1f} us ({tf(t_own):7.1f} TF/s)   max|diff| {err:.3f}", flush=True)
        del ws


if __name__ == "__main__":
    main()
