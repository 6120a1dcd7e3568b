"""VAE conv_out (128 -> 3(4) channels at 512^2, B = 17) and the other narrow-N convs: halo-tile kernel (BN = 128: 32x the
useful columns) vs the implicit GEMM's 64x64 tile.  Usage: python tools/small_n_conv_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from tools.sweep_gemm import time_variant

DEV = "cuda"
for (B, H, C1, C2) in [(17, 512, 128, 4), (2, 512, 128, 4), (17, 512, 128, 32), (17, 512, 128, 64), (17, 256, 256, 64)]:
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
    auto = time_variant(p, 0, 0, 0)
    t3 = time_variant(p, 3, 0, 0, glds_stages=3)
    t2 = time_variant(p, 2, 0, 0, glds_stages=3)
    halo = time_variant(p, 0, 0, 0, halo=True)
    gb = (x.numel() + out.numel()) * 2 / 1e3
    print(f"B{B} {H}x{H} {C1}->{C2}: implicit auto {auto:8.1f} us | 64x64 {t3:8.1f} | 128x64 {t2:8.1f} | halo {halo:8.1f} us  (read+write floor at 6.3 TB/s: {gb / 6.3e3:6.1f} us)", flush=True)
