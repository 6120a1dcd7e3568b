"""Timing + checksum leg for A/Bs of COMPILE-TIME kernel switches: the full SDXL VAE decode program (B = 17, 512^2) and the full UNet
step programs (B = 17 / B = 2), hipGraph replays on the same seeded synthetic weights and inputs, under whichever build of the
library LB_HIP_LIBRARY names (default: the shipped liblbhip.so).  Prints the best of three recordings per program and a checksum of
every output, so that two runs in ONE gpurun call (same box) compare both speed and bits:
  LB_HIP_LIBRARY=$PWD/latentblending_amd/hip/liblbhip_ab0.so python tools/programs_lib_ab.py      (e.g. built with -DLB_HALO_LEAN_ADDR=0)
  python tools/programs_lib_ab.py
Usage: LB_SYNTH_CACHE=/tmp python tools/programs_lib_ab.py >> gpurun_out/programs_lib_ab.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N

DEV = "cuda:0"


def timed(launch, iters):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    tag = os.path.basename(os.environ.get("LB_HIP_LIBRARY", "liblbhip.so"))
    cdir = os.environ.get("LB_SYNTH_CACHE")
    cfile = (lambda s: os.path.join(cdir, f"lb_synth_seed{s}.pt")) if cdir else (lambda s: None)
    vprov = N.SyntheticProvider(1, cache_file=cfile(1))
    vae = N.NativeVAEDecoder(N.VAEConfig(), vprov, DEV)
    vprov.save_cache()
    B, L = 17, 64
    z = torch.randn(B, 4, L, L, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    best = 1e9
    for rep in range(3):
        prog = vae.build(B, L)
        out = prog.decode(z).clone()
        prog.prog.instantiate()
        best = min(best, timed(prog.prog.launch, 5))
        del prog
    print(f"[{tag}] VAE decode B={B}: best {best:.3f} ms, checksum {int(out.to(torch.int64).sum())} / {int((out.to(torch.int64) * 31 % 1009).sum())}", flush=True)
    del vae
    torch.cuda.empty_cache()
    uprov = N.SyntheticProvider(0, cache_file=cfile(0))
    net = N.NativeUNet(N.UNetConfig(), uprov, DEV)
    uprov.save_cache()
    for B in (17, 2):
        g = torch.Generator().manual_seed(B)
        ctx, te = torch.randn(B, 77, 2048, generator=g).half().to(DEV), torch.randn(B, 1280, generator=g).half().to(DEV)
        ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B).to(DEV)
        x = torch.randn(B, 4, 64, 64, generator=g).half().to(DEV)
        best = 1e9
        for rep in range(3):
            prog = net.build(B, 64)
            prog.set_conditioning(ctx, te, ids)
            out = prog.forward(x, torch.full((B,), 499.0)).clone()
            prog.enable_graphs()
            best = min(best, timed(prog.prog_step.launch, 10 if B == 2 else 5))
            del prog
        bits = out.view(torch.int16).to(torch.int64)
        print(f"[{tag}] UNet step B={B}: best {best:.3f} ms, checksum {int(bits.sum())} / {int((bits * 31 % 1009).sum())}", flush=True)


if __name__ == "__main__":
    main()
