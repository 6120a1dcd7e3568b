"""Why are GEMMs slower inside the UNet program than in tools/sweep_gemm.py?  Three suspects, one probe:
(a) hot loop, 20 launches (what the sweep measures); (b) the same loop sustained for ~1.5 s (clock /
power management); (c) every launch reads a DIFFERENT weight matrix out of a > 256 MiB pool (weights
cold in L2 and Infinity Cache, as in a real forward); (d) cold weights AND a fresh activation per launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from latentblending_amd.native.runtime import Program

DEV = "cuda"


def graph_of(ps):
    prog = Program("probe")
    with prog.record():
        for p in ps:
            lib.api.lb_gemm_f16(C.byref(p), 0)
    prog.instantiate()
    return prog


def timed(prog, reps, n_launch):
    st = torch.cuda.current_stream().cuda_stream
    prog.launch(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        prog.launch(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * n_launch) * 1e3


def main():
    zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
    for (M, N, K) in [(4352, 1280, 1280), (4352, 1280, 5120), (17408, 640, 640), (4352, 10240, 1280)]:
        geglu = N == 10240
        n_w = max(8, int(400e6 / (N * K * 2)))
        n_a = max(4, int(400e6 / (M * K * 2)))
        Ws = [(torch.randn(N, K, device=DEV) * K ** -0.5).half() for _ in range(n_w)]
        As = [torch.randn(M, K, device=DEV).half() for _ in range(n_a)]
        out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)

        def params(A, W):
            p = lib.LbGemmParams()
            p.A, p.W, p.C, p.lda, p.ldw, p.ldc = A.data_ptr(), W.data_ptr(), out.data_ptr(), K, K, out.shape[1]
            p.M, p.N, p.K, p.zero_page = M, N, K, zp.data_ptr()
            p.flags = lib.GEMM_GEGLU if geglu else 0
            return p
        hot = graph_of([params(As[0], Ws[0]) for _ in range(20)])
        cold_w = graph_of([params(As[0], Ws[i % n_w]) for i in range(n_w)])
        cold_aw = graph_of([params(As[i % n_a], Ws[i % n_w]) for i in range(max(n_w, n_a))])
        t_hot = timed(hot, 3, 20)
        t_sus = timed(hot, int(1.5e6 / (t_hot * 20)), 20)
        t_cw = timed(cold_w, 3, n_w)
        t_caw = timed(cold_aw, 3, max(n_w, n_a))
        t_hot2 = timed(hot, 3, 20)
        fl = 2.0 * M * N * K / 1e6
        print(f"M{M} N{N} K{K}: hot {t_hot:7.1f} us ({fl / t_hot:5.0f} TF) | sustained 1.5 s {t_sus:7.1f} us | "
              f"cold W ({n_w} mats) {t_cw:7.1f} us | cold A+W {t_caw:7.1f} us | hot again {t_hot2:7.1f} us", flush=True)
        del Ws, As
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
