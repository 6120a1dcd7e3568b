"""A/B of the LayerNorm handling inside the full SDXL UNet programs: stand-alone LayerNorm launches vs the LayerNorm folded
into the consuming GEMMs (fuse_layernorm=True: row statistics inside the K loop; "auto" picks per program from this A/B).
hipGraph replay of the B=2 and B=17 step programs at 512^2, same synthetic weights.
Usage: LB_SYNTH_CACHE=/tmp python tools/ln_stats_ab.py > profiles/r02_ln_stats_ab.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N

DEV = "cuda:0"


def timed(prog, B, L, iters=10):
    x = torch.randn(B, 4, L, L, device=DEV).half()
    t = torch.full((B,), 499.0)
    prog.forward(x, t)
    prog.enable_graphs()
    for _ in range(2):
        prog.forward(x, t)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        prog.prog_step.launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters, prog.forward(x, t).float().cpu(), x


def main():
    cdir = os.environ.get("LB_SYNTH_CACHE")
    cfile = os.path.join(cdir, "lb_synth_seed0.pt") if cdir else None
    outs = {}
    modes = [False, True]         # stand-alone LayerNorms vs folded into the consumers (in-loop statistics)
    for mode in modes:
        t0 = time.time()
        prov = N.SyntheticProvider(0, cache_file=cfile)
        net = N.NativeUNet(N.UNetConfig(), prov, DEV, fuse_layernorm=mode)
        prov.save_cache()
        print(f"# mode={mode!r}: weights ready in {time.time() - t0:.0f} s", flush=True)
        for B in (2, 17):
            prog = net.build(B, 64)
            g = torch.Generator().manual_seed(B)
            cfg = net.cfg
            prog.set_conditioning(torch.randn(B, 77, cfg.cross_dim, generator=g).half().to(DEV),
                                  torch.randn(B, cfg.pooled_dim, generator=g).half().to(DEV),
                                  torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B).to(DEV))
            torch.manual_seed(B)
            ms, out, _ = timed(prog, B, 64)
            names = prog.prog_step.op_names()
            outs[(mode, B)] = out
            print(f"mode={mode!r:8} B={B:2d}: {ms:7.3f} ms per forward (hipGraph), {len(names)} launches, "
                  f"{sum(1 for n in names if n == 'lb_layernorm_f16')} LayerNorm launches", flush=True)
            del prog
        del net
        torch.cuda.empty_cache()
    for B in (2, 17):
        for mode in modes[1:]:
            a, b = outs[(False, B)], outs[(mode, B)]
            print(f"B={B}: rel-L2 between default and mode={mode!r} {float((a - b).norm() / a.norm()):.2e}")


if __name__ == "__main__":
    main()
