"""GroupNorm (+ SiLU) and LayerNorm of the programs' shapes: the hand-written HBM-bound kernels against torch's own on the same GPU
(ATen's GroupNorm / LayerNorm kernels on ROCm; GroupNorm on a channels_last tensor = the NHWC layout the programs use, SiLU as the
separate elementwise op torch eager runs).  A diagnostic like tools/conv_vs_miopen.py, never linked into the product.  GB/s = the
6 bytes per element our two-pass form moves (read, read, write) resp. 4 for LayerNorm, over the measured time, for BOTH sides.
Usage: python tools/norms_vs_torch.py > gpurun_out/norms_vs_torch.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from latentblending_amd.hip import ops as o

DEV = "cuda:0"


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
    print("# GroupNorm(32 groups) + SiLU, fp16 NHWC; LayerNorm rows, fp16; us per call")
    for (B, H, C, eps, what) in [(17, 512, 128, 1e-6, "VAE 512^2 x 128"), (17, 256, 256, 1e-6, "VAE 256^2 x 256"), (17, 128, 512, 1e-6, "VAE 128^2 x 512"),
                                 (17, 32, 640, 1e-5, "UNet 32^2 x 640"), (17, 16, 1280, 1e-5, "UNet 16^2 x 1280"), (17, 16, 2560, 1e-5, "UNet 16^2 x 2560 (concat)"),
                                 (2, 16, 1280, 1e-5, "UNet 16^2 x 1280, B = 2")]:
        g = torch.Generator().manual_seed(C)
        x = torch.randn(B, C, H, H, generator=g).half().to(DEV).contiguous(memory_format=torch.channels_last)
        gm, bt = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        xn = x.permute(0, 2, 3, 1)
        ours = o.groupnorm_nhwc(xn, gm, bt, 32, eps, True)
        ref = F.silu(F.group_norm(x, 32, gm.half(), bt.half(), eps))
        err = float((ours.float() - ref.permute(0, 2, 3, 1).float()).norm() / ref.float().norm())
        t_o = timed(lambda: o.groupnorm_nhwc(xn, gm, bt, 32, eps, True))
        t_t = timed(lambda: F.silu(F.group_norm(x, 32, gm.half(), bt.half(), eps)))
        nbytes = 6.0 * x.numel()
        print(f"GroupNorm+SiLU B={B:2d} {H:3d}^2 x {C:4d}  [{what:26s}]  ours {t_o:8.1f} us ({nbytes / t_o / 1e3:6.0f} GB/s)   torch {t_t:8.1f} us   "
              f"ours / torch {t_o / t_t:5.2f}   rel-L2 {err:.1e}", flush=True)
        del x, xn, ours, ref
        torch.cuda.empty_cache()
    for (M, C, what) in [(4352, 1280, "UNet B = 17, 16^2 tokens x 1280 (180 / step)"), (17408, 640, "UNet B = 17, 32^2 tokens x 640 (30)"), (512, 1280, "UNet B = 2")]:
        g = torch.Generator().manual_seed(M)
        x = torch.randn(M, C, generator=g).half().to(DEV)
        gm, bt = torch.randn(C, generator=g).to(DEV), torch.randn(C, generator=g).to(DEV)
        ours = o.layernorm(x, gm, bt, 1e-5)
        ref = F.layer_norm(x, (C,), gm.half(), bt.half(), 1e-5)
        err = float((ours.float() - ref.float()).norm() / ref.float().norm())
        t_o = timed(lambda: o.layernorm(x, gm, bt, 1e-5), n=20)
        t_t = timed(lambda: F.layer_norm(x, (C,), gm.half(), bt.half(), 1e-5), n=20)
        print(f"LayerNorm {M:6d} x {C:4d}  [{what:44s}]  ours {t_o:8.1f} us ({4.0 * x.numel() / t_o / 1e3:6.0f} GB/s)   torch {t_t:8.1f} us   "
              f"ours / torch {t_o / t_t:5.2f}   rel-L2 {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
