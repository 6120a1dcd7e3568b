"""A chain of transitions over several prompts on one MI355X (the flow of the reference's
example_multi_trans.py / example_multi_trans_json.py, which also run unchanged with the repo root on
PYTHONPATH: ``diffusers``, ``latentblending`` and ``lunar_tools`` resolve to the facades in this repo).

    python examples/multi_transition.py                 # three prompts, random seeds
    python examples/multi_transition.py movie.json      # replay a movie JSON saved by the UI
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diffusers import AutoPipelineForText2Image
from latentblending.blending_engine import BlendingEngine
from latentblending_amd import replay

pipe = AutoPipelineForText2Image.from_pretrained("stabilityai/sdxl-turbo", torch_dtype=torch.float16, variant="fp16")
pipe.to("cuda")
engine = BlendingEngine(pipe, do_compile=True)
if len(sys.argv) > 1:
    segments = replay.run_movie_json(engine, sys.argv[1], "movie.avi", duration_single_trans=10)
else:
    prompts = ["high resolution ultra 8K image with lake and forest",
               "strange and alien desolate lanscapes 8K",
               "ultra high res psychedelic skyscraper city landscape 8K unreal engine"]
    seeds = np.random.randint(0, np.iinfo(np.int32).max, len(prompts))
    engine.set_negative_prompt("blurry, pale, low-res, lofi")
    segments = replay.run_multi_transition(engine, prompts, seeds, "movie.avi", duration_single_trans=10)
print(f"{len(segments)} segments, {sum(len(s) for s in segments)} key frames -> movie.avi")
