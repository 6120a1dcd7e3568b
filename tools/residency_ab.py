"""Why does the same UNet B=17 step program take ~35.3 ms in a UNet-only process (tools/unet_knob_ab.py, round 4) and ~38 ms inside
the bench process (phases_per_transition) and after a VAE was built (tools/epilogue_ab.py, round 5)?  One process: time the step
graph (a) alone, (b) after the VAE decoder + its B=17 program exist and ran, (c) after also the B=2 program exists, (d) after the
VAE objects were dropped and the allocator cache emptied, (e) with the step replayed back to back with the VAE decode (the
bench's order); (b') / (b'') separate a clock effect from a memory-placement effect.  Usage: LB_SYNTH_CACHE=/tmp python tools/residency_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N

DEV = "cuda:0"


def timed(launch, iters=5):
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


def mem():
    return f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB"


def main():
    cdir = os.environ.get("LB_SYNTH_CACHE")
    prov = N.SyntheticProvider(0, cache_file=os.path.join(cdir, "lb_synth_seed0.pt") if cdir else None)
    net = N.NativeUNet(N.UNetConfig(), prov, DEV)
    prov.save_cache()

    def unet_prog(B):
        g = torch.Generator().manual_seed(B)
        ctx, te = torch.randn(B, 77, 2048, generator=g).half().to(DEV), torch.randn(B, 1280, generator=g).half().to(DEV)
        ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B).to(DEV)
        x = torch.randn(B, 4, 64, 64, generator=g).half().to(DEV)
        prog = net.build(B, 64)
        prog.set_conditioning(ctx, te, ids)
        prog.forward(x, torch.full((B,), 499.0))
        prog.enable_graphs()
        return prog

    p17 = unet_prog(17)
    for rep in range(2):
        print(f"(a) UNet B=17 step alone                         : {timed(p17.prog_step.launch):7.3f} ms   [{mem()}]", flush=True)
    vae = N.NativeVAEDecoder(N.VAEConfig(), N.SyntheticProvider(1), DEV)
    z = torch.randn(17, 4, 64, 64, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    vp = vae.build(17, 64)
    vp.decode(z)
    vp.prog.instantiate()
    print(f"    VAE decode B=17                              : {timed(vp.prog.launch):7.3f} ms", flush=True)
    for rep in range(2):
        print(f"(b) UNet B=17 step, VAE resident                 : {timed(p17.prog_step.launch):7.3f} ms   [{mem()}]", flush=True)
    import time
    time.sleep(5.0)         # (clocks / temperature: the VAE's halo convs run the chip at its power limit just before (b))
    print(f"(b') the same after 5 s of idle                  : {timed(p17.prog_step.launch):7.3f} ms", flush=True)
    print(f"(b'') and again immediately (20 iterations)      : {timed(p17.prog_step.launch, 20):7.3f} ms", flush=True)
    p2 = unet_prog(2)
    print(f"    UNet B=2 step                                : {timed(p2.prog_step.launch, 10):7.3f} ms", flush=True)
    print(f"(c) UNet B=17 step, VAE + B=2 program resident   : {timed(p17.prog_step.launch):7.3f} ms   [{mem()}]", flush=True)

    def bench_order():
        p2.prog_step.launch(); p2.prog_step.launch(); p17.prog_step.launch(); p17.prog_step.launch(); vp.prog.launch()
    t = timed(bench_order, 4)
    print(f"(e) 2 x B=2 + 2 x B=17 + VAE back to back        : {t:7.3f} ms per round", flush=True)
    del vp, vae, p2
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    print(f"(d) UNet B=17 step, VAE dropped + cache emptied  : {timed(p17.prog_step.launch):7.3f} ms   [{mem()}]", flush=True)


if __name__ == "__main__":
    main()
