"""A/B of the one-round-trip tile epilogue (lb_gemm_set_lean_epilogue, round 5) inside the full programs: VAE decode B=17 and UNet
step B=17 / B=2 at 512^2, hipGraph replays in ONE process, outputs must be bit-identical.
Usage: LB_SYNTH_CACHE=/tmp python tools/epilogue_ab.py [--unet]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N
from latentblending_amd.hip import lib

DEV = "cuda:0"
MODES = (0, 1)          # lb_gemm_set_lean_epilogue values: 0 = per-row epilogue everywhere (rounds 2-4), 1 = one round trip per tile


def timed(launch, iters):
    for _ in range(2):
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
    z = torch.randn(17, 4, 64, 64, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    vae = N.NativeVAEDecoder(N.VAEConfig(), N.SyntheticProvider(1), DEV)
    outs = {}
    for rep in range(2):
        for wide in MODES:
            lib.api.lb_gemm_set_lean_epilogue(wide)
            prog = vae.build(17, 64)                      # (the flag is read when the program is RECORDED)
            prog.decode(z)
            prog.prog.instantiate()
            ms = timed(prog.prog.launch, 5)
            outs[wide] = prog.decode(z).clone()
            print(f"VAE decode B=17: lean_epilogue={wide}: {ms:7.3f} ms", flush=True)
            del prog
    print("VAE frames identical:", bool(torch.equal(outs[MODES[0]], outs[MODES[1]])))
    if "--unet" in sys.argv:
        cdir = os.environ.get("LB_SYNTH_CACHE")
        prov = N.SyntheticProvider(0, cache_file=os.path.join(cdir, "lb_synth_seed0.pt") if cdir else None)
        net = N.NativeUNet(N.UNetConfig(), prov, DEV)
        prov.save_cache()
        for B in (17, 2):
            g = torch.Generator().manual_seed(B)
            ctx, te = torch.randn(B, 77, 2048, generator=g).half().to(DEV), torch.randn(B, 1280, generator=g).half().to(DEV)
            ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B).to(DEV)
            x = torch.randn(B, 4, 64, 64, generator=g).half().to(DEV)
            res = {}
            for rep in range(2):
                for wide in MODES:
                    lib.api.lb_gemm_set_lean_epilogue(wide)
                    prog = net.build(B, 64)
                    prog.set_conditioning(ctx, te, ids)
                    prog.forward(x, torch.full((B,), 499.0))
                    prog.enable_graphs()
                    ms = timed(prog.prog_step.launch, 10 if B == 2 else 5)
                    res[wide] = prog.forward(x, torch.full((B,), 499.0)).clone()
                    print(f"UNet step B={B}: lean_epilogue={wide}: {ms:7.3f} ms", flush=True)
                    del prog
            print(f"UNet B={B} outputs identical:", bool(torch.equal(res[MODES[0]], res[MODES[1]])))
    lib.api.lb_gemm_set_lean_epilogue(1)


if __name__ == "__main__":
    main()
