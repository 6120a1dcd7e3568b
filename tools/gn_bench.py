"""GroupNorm(+SiLU) at the VAE-decode shapes (B = 17): time per launch and algorithmic GB/s (read x twice, write y once)
with the statistics+apply pairs walked in Infinity-Cache-sized sample groups (lb_groupnorm_set_l3_chunk) vs one pair
over the whole batch.  Usage: python tools/gn_bench.py > profiles/r02_groupnorm_l3.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from latentblending_amd.hip import lib, ops  # noqa: E402


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = "cuda:0"
    B = 17
    print(f"{'shape':>22} {'chunk MiB':>10} {'us':>9} {'GB/s (3 passes)':>16} {'GB/s (2 passes)':>16}")
    for side, c in ((512, 128), (512, 256), (256, 256), (256, 512), (128, 512), (64, 512)):
        x = torch.randn(B, side, side, c, device=dev, dtype=torch.float16)
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        ref = None
        for mib in (0, 32, 64, 96, 128, 192):
            lib.api.lb_groupnorm_set_l3_chunk(mib << 20)
            y = ops.groupnorm_nhwc(x, g, b, 32, 1e-6, True)
            if ref is None:
                ref = y
            else:
                assert (y.float() - ref.float()).abs().max().item() < 4e-3
            ms = timed(lambda: ops.groupnorm_nhwc(x, g, b, 32, 1e-6, True))
            nbytes = x.numel() * 2
            print(f"{f'{B}x{side}x{side}x{c}':>22} {mib:>10} {ms * 1e3:9.1f} {3 * nbytes / ms / 1e6:16.0f} {2 * nbytes / ms / 1e6:16.0f}")
        del x, ref, y
    lib.api.lb_groupnorm_set_l3_chunk(0)


if __name__ == "__main__":
    main()
