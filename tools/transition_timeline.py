"""Where the ~3 ms of a cfg-2 transition outside the UNet / VAE / LPIPS programs go (round 6): hipEvents at the start of
run_transition, around every program launch and at the end, printed as a timeline (one process, graphs on, frontier 16).
Usage: LB_SYNTH_CACHE=/tmp python tools/transition_timeline.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N
from latentblending_amd import BlendingEngine


def main():
    cdir = os.environ.get("LB_SYNTH_CACHE")
    cfile = (lambda s: os.path.join(cdir, f"lb_synth_seed{s}.pt")) if cdir else (lambda s: None)
    pipe = N.NativeSDXLPipe(turbo=True, unet_provider=N.SyntheticProvider(0, cache_file=cfile(0)),
                            vae_provider=N.SyntheticProvider(1, cache_file=cfile(1)), allow_synthetic=True)
    be = BlendingEngine(pipe, do_compile=True, frontier_width=16, verbose=False)
    be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
    be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
    be.set_branching(nmb_max_branches=15)
    for _ in range(3):
        be.run_transition(fixed_seeds=[420, 421])
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e, time.perf_counter()))

    def wrap(obj, attr, label):
        orig = getattr(obj, attr)

        def timed(*a, **k):
            mark(label + " begin")
            r = orig(*a, **k)
            mark(label + " end")
            return r
        setattr(obj, attr, timed)
    for key, prog in pipe._unet_programs.items():
        wrap(prog.prog_step, "launch", f"unet B={key[0]}")
        wrap(prog.prog_cond, "launch", f"unet-cond B={key[0]}")
    for key, prog in pipe._vae_programs.items():
        wrap(prog.prog, "launch", f"vae B={key[0]}")
    wrap(pipe, "native_frame_distances", "lpips")
    for rep in range(2):
        marks.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mark("run_transition begin")
        imgs = be.run_transition(fixed_seeds=[420, 421])
        mark("run_transition returned")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f"== transition {rep}: {1e3 * (t1 - t0):.3f} ms wall, {len(imgs)} frames", flush=True)
        base_e, base_t = marks[0][1], marks[0][2]
        prev = 0.0
        for name, e, th in marks:
            tg = base_e.elapsed_time(e)
            print(f"   gpu {tg:9.3f} ms (+{tg - prev:7.3f})   host {1e3 * (th - base_t):9.3f} ms   {name}", flush=True)
            prev = tg


if __name__ == "__main__":
    main()
