"""A chain of transitions (the loop of the reference's example_multi_trans.py:39-58) at the metric's settings (SDXL-Turbo 512^2,
4 steps, 15 mid branches, hipGraphs, frontier 16, frames copied to host PIL images): the sequential loop (swap_forward +
recycle_img1: one batch-1 key-frame trajectory per transition) vs replay.run_multi_transition(pipeline_keyframes=True) (all key
frames denoised and decoded ahead of the transitions).  Usage: python tools/chain_ab.py > gpurun_out/chain_ab.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N
from latentblending_amd import BlendingEngine, replay

PROMPTS = ["photo of underwater landscape, fish, und the sea, incredible detail, high resolution",
           "rendering of an alien planet, strange plants, strange creatures, surreal",
           "photo of a forest in the fog, sun rays", "aerial photo of a city at night", "macro photo of a dragonfly"]
SEEDS = [420, 421, 422, 423, 424]


def main():
    pipe = N.NativeSDXLPipe(turbo=True, allow_synthetic=True)
    res = {}
    for rep in range(3):
        for piped in (False, True):
            be = BlendingEngine(pipe, do_compile=True, frontier_width=16, verbose=False)
            be.host_frames = True
            be.set_branching(nmb_max_branches=15)
            before = pipe.stats["unet_samples"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            segs = replay.run_multi_transition(be, PROMPTS, SEEDS, None, pipeline_keyframes=piped)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            frames = sum(len(s) for s in segs)
            tag = "warm-up (programs built)" if rep == 0 else ""
            print(f"rep {rep} pipeline_keyframes={piped!s:5}: {frames} frames in {dt * 1e3:8.2f} ms = {frames / dt:7.2f} frames/s, "
                  f"{dt * 250:7.2f} ms per transition, {pipe.stats['unet_samples'] - before} UNet samples {tag}", flush=True)
            if rep:
                res.setdefault(piped, []).append(dt)
    a, b = min(res[False]), min(res[True])
    print(f"best: sequential {a * 250:.2f} ms per transition, pipelined {b * 250:.2f} ms per transition ({a / b:.3f}x)")


if __name__ == "__main__":
    main()
