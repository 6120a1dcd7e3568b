"""A/B of the GroupNorm handling inside the full SDXL VAE decode program (B = 17, 512^2): two-pass GroupNorm everywhere vs
statistics from the producing halo conv's epilogue (LB_GEMM_CH_STATS + lb_groupnorm_from_stats, VAEConfig.fuse_gn_stats).
hipGraph replays, same synthetic weights, outputs compared.  Usage: python tools/vae_gn_ab.py > gpurun_out/vae_gn_ab.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N

DEV = "cuda:0"


def main():
    B, L = int(os.environ.get("LB_AB_BATCH", "17")), 64
    z = torch.randn(B, 4, L, L, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    outs, times = {}, {}
    for fuse in (False, True, False, True):
        net = N.NativeVAEDecoder(N.VAEConfig(fuse_gn_stats=fuse), N.SyntheticProvider(1), DEV)
        prog = net.build(B, L)
        prog.decode(z)
        prog.prog.instantiate()
        for _ in range(2):
            prog.prog.launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            prog.prog.launch()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        names = prog.prog.op_names()
        times.setdefault(fuse, []).append(ms)
        outs[fuse] = prog.decode(z).clone()
        print(f"fuse_gn_stats={fuse!s:5}: {ms:7.3f} ms per decode batch (B={B}, hipGraph), {len(names)} launches, "
              f"{sum(1 for n in names if n == 'lb_groupnorm_from_stats')} GroupNorms from conv statistics, "
              f"{sum(1 for n in names if n == 'lb_groupnorm_nhwc')} two-pass", flush=True)
        del prog, net
        torch.cuda.empty_cache()
    d = (outs[True].int() - outs[False].int()).abs()
    print(f"frames: mean |du8| between the two forms {d.float().mean():.4f}, max {int(d.max())}")
    print(f"best: two-pass {min(times[False]):.3f} ms, fused statistics {min(times[True]):.3f} ms")


if __name__ == "__main__":
    main()
