"""Experiment (round 6): the B = 17 UNet step as ONE program vs as TWO programs (B = 9 and B = 8) replayed concurrently on two
streams - does asynchrony between the halves hide the launch boundaries / kernel tails of the dependent chain?
Usage: LB_SYNTH_CACHE=/tmp python tools/dual_stream_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N

DEV = "cuda:0"


def main():
    cdir = os.environ.get("LB_SYNTH_CACHE")
    prov = N.SyntheticProvider(0, cache_file=os.path.join(cdir, "lb_synth_seed0.pt") if cdir else None)
    net = N.NativeUNet(N.UNetConfig(), prov, DEV)
    prov.save_cache()

    def build(B):
        g = torch.Generator().manual_seed(B)
        ctx, te = torch.randn(B, 77, 2048, generator=g).half().to(DEV), torch.randn(B, 1280, generator=g).half().to(DEV)
        ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B).to(DEV)
        x = torch.randn(B, 4, 64, 64, generator=g).half().to(DEV)
        prog = net.build(B, 64)
        prog.set_conditioning(ctx, te, ids)
        prog.forward(x, torch.full((B,), 499.0))
        prog.enable_graphs()
        return prog

    whole, a, b = build(17), build(9), build(8)
    s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    main_s = torch.cuda.current_stream()

    def one():
        whole.prog_step.launch(main_s.cuda_stream)

    def two():
        s1.wait_stream(main_s)
        s2.wait_stream(main_s)
        a.prog_step.launch(s1.cuda_stream)
        b.prog_step.launch(s2.cuda_stream)
        main_s.wait_stream(s1)
        main_s.wait_stream(s2)

    def seq():
        a.prog_step.launch(main_s.cuda_stream)
        b.prog_step.launch(main_s.cuda_stream)

    def timed(fn, iters=5):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    for rep in range(3):
        print(f"rep {rep}: B=17 one program {timed(one):8.3f} ms | B=9 + B=8 on two streams {timed(two):8.3f} ms | B=9 then B=8 on one stream {timed(seq):8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
