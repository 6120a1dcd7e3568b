"""Where does a halo-conv tile's time go?  The VAE / UNet B=17 conv shapes timed with parts of the kernel switched off
(lb_conv_halo_set_study): no epilogue (no global stores), halo requests served from the zero page (no activation reads),
weight requests from the zero page, and combinations; persistent and one-item-per-block grids.
Usage: python tools/halo_study.py > profiles/r02_halo_study.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # NEEDS a study build: python -m latentblending_amd.csrc.build --study; LB_HIP_LIBRARY=latentblending_amd/hip/liblbhip_study.so (LB_STUDY_BUILD)
from latentblending_amd.hip import lib
from tools.sweep_gemm import time_variant

DEV = "cuda"
MODES = [(0, "real"), (1, "no epilogue"), (2, "no act reads"), (3, "no epilogue, no act reads"), (7, "MFMA + LDS only")]


def main():
    shapes = [(17, 512, 128, 128), (17, 512, 256, 128), (17, 256, 256, 256), (17, 128, 512, 512), (17, 64, 320, 320), (17, 16, 1280, 1280)]
    print(f"{'shape':24s} {'grid':>10s} " + " ".join(f"{n:>26s}" for _, n in MODES))
    for (B, H, C1, C2) in shapes:
        p = lib.LbGemmParams()
        x = torch.randn(B, H, H, C1, device=DEV).half()
        w = (torch.randn(C2, 9 * C1, device=DEV) * (9 * C1) ** -0.5).half()
        out = torch.empty(B, H, H, C2, device=DEV, dtype=torch.float16)
        M, N, K = B * H * H, C2, 9 * C1
        p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = 1, H, H, C1, H, H, 3, 3, 1, 1, 0, C1
        p.A, p.W, p.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
        p.M, p.N, p.K, p.ldw, p.ldc = M, N, K, K, C2
        zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
        p.zero_page = zp.data_ptr()
        tiles = M // 256 * ((N + 127) // 128)
        steps = 9 * (C1 // 64)
        for persistent in (1, 0):
            lib.api.lb_conv_halo_set_persistent(persistent)
            cells = []
            for bits, _ in MODES:
                lib.api.lb_conv_halo_set_study(bits)
                us = time_variant(p, 0, 0, 0, halo=True)
                per_tile = us / max(1.0, tiles / 256.0)
                cells.append(f"{us:8.1f} us {per_tile:6.2f}/tile {per_tile / steps * 1e3:4.0f}ns/st")
            lib.api.lb_conv_halo_set_study(0)
            print(f"{f'B{B} {H}x{H} {C1}->{C2}':24s} {'persistent' if persistent else 'one item':>10s} " + " ".join(f"{c:>26s}" for c in cells), flush=True)
        lib.api.lb_conv_halo_set_persistent(1)
        del x, w, out


if __name__ == "__main__":
    main()
