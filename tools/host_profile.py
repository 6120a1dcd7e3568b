"""cProfile of the HOST side of cfg-2 transitions (SDXL-Turbo 512^2, 4 steps, 15 mid branches, hipGraphs, frontier 16, host frames): where
does the Python time of a transition go (the device work is asynchronous; the waits show up under the synchronising calls)?
Usage: LB_SYNTH_CACHE=/tmp python tools/host_profile.py > gpurun_out/host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys

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
    be.host_frames = True
    be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
    be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
    be.set_branching(nmb_max_branches=15)
    for _ in range(3):
        be.run_transition(fixed_seeds=[420, 421])
    torch.cuda.synchronize()
    n = 10
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        be.run_transition(fixed_seeds=[420, 421])
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        out = io.StringIO()
        pstats.Stats(pr, stream=out).strip_dirs().sort_stats(key).print_stats(45)
        print(f"==== {n} transitions, sorted by {key} ====")
        print(out.getvalue())


if __name__ == "__main__":
    main()
