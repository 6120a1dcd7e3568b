"""The 3x3 convolutions of the benchmark's programs: the halo-tile kernel (lb_gemm_f16's route) against the VENDOR library on the same
operands - MIOpen through torch.nn.functional.conv2d (fp16, channels-last activations and weights, `cudnn.benchmark = True` = MIOpen's
find mode: it times its own solvers - implicit-GEMM / Winograd / direct / CK - and keeps the fastest).  The conv counterpart of
tools/gemm_bench.cpp's rocBLAS rows: a diagnostic, never linked into the product.  No bias / residual on either side (the library's
call computes the bare convolution).  hipEvents on the launch stream, median of 5 rounds of 10 launches.
Usage: python tools/conv_vs_miopen.py > gpurun_out/conv_vs_miopen.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from latentblending_amd.hip import ops as o

DEV = "cuda:0"
# (B, H = W, Cin, Cout, where)
SHAPES = [(17, 512, 128, 128, "VAE up 512^2 (5 / decode)"), (17, 256, 256, 256, "VAE up 256^2 (5)"), (17, 128, 512, 512, "VAE up 128^2 (6)"),
          (17, 64, 512, 512, "VAE mid / up 64^2 (10)"), (17, 512, 256, 128, "VAE 256 -> 128 @512^2"), (17, 256, 512, 256, "VAE 512 -> 256 @256^2"),
          (17, 16, 1280, 1280, "UNet 16^2 1280 (10 / step)"), (17, 64, 320, 320, "UNet 64^2 320 (7)"), (17, 32, 640, 640, "UNet 32^2 640 (6)"),
          (17, 16, 2560, 1280, "UNet up 16^2 2560 -> 1280 (2)"), (2, 512, 128, 128, "VAE up 512^2, B = 2"), (2, 64, 320, 320, "UNet 64^2 320, B = 2")]


def timed(fn, n=10, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / n * 1e3)
    return sorted(out)[len(out) // 2]


def main():
    torch.backends.cudnn.benchmark = True
    print("# 3x3 / stride 1 / pad 1 convolutions, fp16, NHWC: halo-tile kernel vs MIOpen (torch conv2d, channels_last, find mode); us per launch")
    for (B, H, Cin, Cout, what) in SHAPES:
        g = torch.Generator().manual_seed(H + Cin)
        x = torch.randn(B, Cin, H, H, generator=g).half().to(DEV).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().to(DEV)
        wcl = w.contiguous(memory_format=torch.channels_last)
        xn = x.permute(0, 2, 3, 1)                      # the same storage, viewed NHWC
        assert xn.is_contiguous()
        wp = o.pack_conv_weight(w.cpu(), Cin).to(DEV)
        conv = dict(KH=3, KW=3, stride=1, pad=1)
        flops = 2.0 * B * H * H * Cout * 9 * Cin
        ref = F.conv2d(x, wcl, padding=1)
        got = o.gemm(xn, wp, conv=conv)
        err = float((got.float() - ref.permute(0, 2, 3, 1).float()).norm() / ref.float().norm())
        t_ours = timed(lambda: o.gemm(xn, wp, conv=conv, out=got))
        t_lib = timed(lambda: F.conv2d(x, wcl, padding=1))
        print(f"B={B:2d} {H:3d}^2 {Cin:4d} -> {Cout:4d}  [{what:32s}]  ours {t_ours:8.1f} us ({flops / t_ours / 1e6:6.0f} TF/s)   "
              f"MIOpen {t_lib:8.1f} us ({flops / t_lib / 1e6:6.0f} TF/s)   ours / MIOpen {t_ours / t_lib:5.2f}   rel-L2 between them {err:.1e}", flush=True)
        del x, w, wcl, xn, wp, ref, got
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
