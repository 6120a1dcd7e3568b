"""A/B of the halo-tile conv kernel's step loop: lock-step (every wave reads, then every wave multiplies) vs ping-pong (two wave
groups one barrier apart), on the conv shapes of the B = 17 programs, bit-identity included.  Each variant timed as a hipGraph of
20 launches, variants interleaved, median of 5.   Usage (GPU box): python tools/halo_pp_ab.py > gpurun_out/r04_halo_pp_ab.txt"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from tools.sweep_gemm import time_variant

DEV = "cuda"


def main():
    shapes = [(17, 512, 128, 128, 3), (17, 256, 256, 256, 3), (17, 128, 512, 512, 3), (17, 64, 512, 512, 3), (17, 512, 256, 128, 3),
              (17, 64, 320, 320, 3), (17, 32, 640, 640, 3), (17, 16, 1280, 1280, 3), (17, 16, 2560, 1280, 3), (17, 32, 1920, 640, 3),
              (17, 64, 960, 320, 3), (2, 64, 512, 512, 3), (2, 64, 320, 320, 3)]
    for (B, H, C1, C2, ks) in shapes:
        p = lib.LbGemmParams()
        x = torch.randn(B, H, H, C1, device=DEV).half()
        w = (torch.randn(C2, 9 * C1, device=DEV) * (9 * C1) ** -0.5).half()
        bias = torch.randn(C2, device=DEV)
        outs = [torch.empty(B, H, H, C2, device=DEV, dtype=torch.float16) for _ in range(2)]
        M, N, K = B * H * H, C2, 9 * C1
        p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = 1, H, H, C1, H, H, 3, 3, 1, 1, 0, C1
        p.A, p.W, p.bias = x.data_ptr(), w.data_ptr(), bias.data_ptr()
        p.M, p.N, p.K, p.ldw, p.ldc = M, N, K, K, C2
        zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
        p.zero_page = zp.data_ptr()
        flops = 2.0 * M * N * K
        t = {0: [], 1: []}
        for r in range(5):
            for pp in (0, 1):
                lib.api.lb_conv_halo_set_pingpong(pp)
                p.C = outs[pp].data_ptr()
                t[pp].append(time_variant(p, 0, 0, 0, halo=True))
        lib.api.lb_conv_halo_set_pingpong(0)
        torch.cuda.synchronize()
        same = torch.equal(outs[0], outs[1])
        a, b = statistics.median(t[0]), statistics.median(t[1])
        print(f"B{B} {H}x{H} {C1}->{C2}: lock-step {a:9.1f} us {flops / a / 1e6:7.0f} TF/s | ping-pong {b:9.1f} us {flops / b / 1e6:7.0f} TF/s | x{a / b:.3f} | "
              f"{'bit-identical' if same else 'DIFFERENT'}", flush=True)
        del x, w, outs


if __name__ == "__main__":
    main()
