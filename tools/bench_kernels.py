"""Micro-benchmarks of the hot kernels on one MI355X (run through gpurun).  Random data."""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import ops as o, lib

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    out = {}
    print("device", torch.cuda.get_device_name(0))
    # GEMM shapes of the SDXL UNet at 512^2 (B = 1, 4, 8)
    shapes = []
    for B in (1, 4, 8):
        shapes += [(256 * B, 3840, 1280, "qkv1280"), (256 * B, 1280, 1280, "out1280"),
                   (256 * B, 10240, 1280, "geglu1280"), (256 * B, 1280, 5120, "ffout1280"),
                   (1024 * B, 1920, 640, "qkv640"), (1024 * B, 5120, 640, "geglu640")]
    shapes += [(4096, 4096, 4096, "square4k"), (8192, 8192, 8192, "square8k")]
    for (M, N, K, tag) in shapes:
        A = torch.randn(M, K, device=DEV).half()
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
        geglu = tag.startswith("geglu")
        for tile in (0, 1, 2, 3):
            lib.api.lb_gemm_set_tuning(tile, 0)
            try:
                ws = torch.empty(lib.api.lb_gemm_workspace_bytes(M, N) // 4, dtype=torch.float32, device=DEV)
                outbuf = torch.empty(M, N // 2 if geglu else N, dtype=torch.float16, device=DEV)
                import ctypes as C
                p = lib.LbGemmParams()
                p.A, p.W, p.C = A.data_ptr(), W.data_ptr(), outbuf.data_ptr()
                p.M, p.N, p.K, p.lda, p.ldw, p.ldc = M, N, K, K, K, outbuf.stride(0)
                p.flags = lib.GEMM_GEGLU if geglu else 0
                p.partial = None if geglu else ws.data_ptr()
                st = torch.cuda.current_stream().cuda_stream
                dt = timeit(lambda: lib.api.lb_gemm_f16(C.byref(p), st))
            finally:
                lib.api.lb_gemm_set_tuning(0, 0)
            tf = 2.0 * M * N * K / dt / 1e12
            gbs = (M * K + N * K + M * N) * 2 / dt / 1e9
            out[f"gemm_{tag}_M{M}_tile{tile}"] = {"us": dt * 1e6, "TF": tf, "GBs": gbs}
            print(f"gemm {tag:10s} M={M:5d} N={N:5d} K={K:5d} tile={tile}: {dt*1e6:9.1f} us {tf:8.1f} TF/s {gbs:8.0f} GB/s")
    # conv 3x3 shapes
    for (B, H, C1, C2, tag) in [(1, 64, 320, 320, "res320"), (1, 32, 640, 640, "res640"), (1, 16, 1280, 1280, "res1280"),
                                (8, 16, 1280, 1280, "res1280b8"), (1, 512, 128, 128, "vae128"), (1, 256, 256, 256, "vae256"),
                                (1, 128, 512, 512, "vae512")]:
        x = torch.randn(B, H, H, C1, device=DEV).half()
        w = (torch.randn(C2, 9 * C1, device=DEV) * (9 * C1) ** -0.5).half()
        fn = lambda: o.gemm(x, w, conv=dict(KH=3, KW=3, stride=1, pad=1), splitk_ws=False)
        dt = timeit(fn, iters=10)
        tf = 2.0 * B * H * H * C2 * 9 * C1 / dt / 1e12
        out[f"conv_{tag}"] = {"us": dt * 1e6, "TF": tf}
        print(f"conv3x3 {tag:10s}: {dt*1e6:9.1f} us {tf:8.1f} TF/s")
    # attention
    for (B, H, S, tag) in [(1, 10, 1024, "self640"), (1, 20, 256, "self1280"), (8, 20, 256, "self1280b8"), (8, 10, 1024, "self640b8")]:
        C = H * 64
        q = torch.randn(B * S, C, device=DEV).half(); k = torch.randn(B * S, C, device=DEV).half()
        vt = torch.randn(C, B * S, device=DEV).half()
        dt = timeit(lambda: o.attention_d64(q, k, vt, B, H, S, S))
        tf = 4.0 * B * H * S * S * 64 / dt / 1e12
        out[f"attn_{tag}"] = {"us": dt * 1e6, "TF": tf}
        print(f"attn {tag:12s}: {dt*1e6:9.1f} us {tf:8.1f} TF/s")
    # slerp batched: HBM roofline on a >1 GiB problem (6 B/element algorithmic)
    for n in (16384, 65536):
        pairs = (1 << 30) // (n * 2 * 3) * 2
        p0 = torch.randn(pairs, n, device=DEV).half(); p1 = torch.randn(pairs, n, device=DEV).half()
        fr = torch.rand(pairs, device=DEV, dtype=torch.float64)
        dt = timeit(lambda: o.slerp_batched(p0, p1, fr), iters=5)
        gbs = pairs * n * 6 / dt / 1e9
        out[f"slerp_batched_n{n}"] = {"us": dt * 1e6, "GBs": gbs, "pairs": pairs}
        print(f"slerp batched n={n} pairs={pairs}: {dt*1e6:9.1f} us {gbs:8.0f} GB/s")
    a = torch.randn(1, 4, 64, 64, device=DEV).half(); b = torch.randn(1, 4, 64, 64, device=DEV).half()
    dt = timeit(lambda: o.slerp(a, b, 0.3), iters=50)
    out["slerp_single_L64_us"] = dt * 1e6
    print(f"slerp single L=64: {dt*1e6:.1f} us per call (launch-latency bound)")
    # groupnorm / layernorm
    for (B, HW, C) in [(1, 4096, 320), (1, 1024, 640), (1, 256, 1280), (8, 256, 1280), (1, 262144, 128)]:
        x = torch.randn(B, HW, C, device=DEV).half()
        g = torch.ones(C, device=DEV); bt = torch.zeros(C, device=DEV)
        dt = timeit(lambda: o.groupnorm_nhwc(x, g, bt, 32, 1e-5, True))
        out[f"gn_{B}_{HW}_{C}"] = {"us": dt * 1e6, "GBs": B * HW * C * 4 / dt / 1e9}
        print(f"groupnorm B={B} HW={HW} C={C}: {dt*1e6:9.1f} us")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_kernels.json", "w"), indent=1)


if __name__ == "__main__":
    main()
