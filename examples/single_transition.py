"""Two-prompt transition on one MI355X through the reference's public API.
The reference's own example_single_trans.py runs unchanged as well (repo root on PYTHONPATH):
``diffusers`` resolves to the facade in ./diffusers, ``latentblending`` to the shim in ./latentblending."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusers import AutoPipelineForText2Image
from latentblending.blending_engine import BlendingEngine

pipe = AutoPipelineForText2Image.from_pretrained("stabilityai/sdxl-turbo", torch_dtype=torch.float16, variant="fp16")
pipe.to("cuda")
engine = BlendingEngine(pipe, do_compile=True, frontier_width=8)
engine.set_prompt1("a lighthouse in a storm, oil painting")
engine.set_prompt2("a quiet harbour at dawn, watercolor")
frames = engine.run_transition(fixed_seeds=[420, 421])
print(f"{len(frames)} frames, fractions {engine.tree_fracts}")
engine.write_movie_transition("transition.avi", duration_transition=4)
