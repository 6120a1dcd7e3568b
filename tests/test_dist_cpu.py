"""World-size-2 `gloo` test of the branch farm on CPU: two ranks run the SPMD engine on the tiny
oracle pipe, split every speculative round between them, and must both end with exactly the tree
the sequential reference produced (tests/golden/tree.json: fractions, similarities, latents, frames)."""
import hashlib
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, run, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    from latentblending_amd.dist import BranchFarm
    set_backend(R.TorchCpuBackend())
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "tree.json")))[run]
    cfgd = c["config"]
    p = OP.StableDiffusionXLPipeline(turbo=c["turbo"], unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    np.random.seed(0)
    farm = BranchFarm()
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=4, farm=farm)
    be.set_dimensions((128, 128))
    if "steps" in cfgd:
        be.set_num_inference_steps(cfgd["steps"])
    if "gs" in cfgd:
        be.set_guidance_scale(cfgd["gs"])
    be.set_branching(depth_strength=cfgd.get("depth"), nmb_max_branches=cfgd["nmb"])
    be.set_prompt1("photo of a reef")
    be.set_prompt2("rendering of an alien planet")
    p.noise.reset()
    p.unet.calls = 0
    imgs = be.run_transition(fixed_seeds=[420, 421])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpu import check_against_golden_run
    check_against_golden_run(be, imgs, c)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).tobytes()).hexdigest()[:16]
    res = {"fracts": [float(f) for f in be.tree_fracts], "idx": [int(i) for i in be.tree_idx_injection],
           "sims": [float(s) for s in be.tree_similarities], "latent_sha": [sha(l[-1]) for l in be.tree_latents],
           "frame_sha": [sha(i) for i in imgs], "unet_calls": p.unet.calls, "collectives": farm.collectives,
           "bytes": farm.bytes_moved, "dropped": be.stats.get("speculation_dropped", 0)}
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("run", [0, 2])
def test_branch_farm_world2_reproduces_sequential_tree(run, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, run, str(tmp_path)), nprocs=2, join=True)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tree.json")))[run]
    r0, r1 = [json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1)]
    for r in (r0, r1):
        assert r["fracts"] == gold["tree_fracts"] and r["idx"] == gold["tree_idx_injection"]
        assert np.allclose(r["sims"], gold["tree_similarities"], rtol=2e-3)
        assert r["collectives"] > 0 and r["bytes"] > 0
    # SPMD: both ranks hold bit-identical trees, latents and frames
    assert r0["sims"] == r1["sims"] and r0["latent_sha"] == r1["latent_sha"] and r0["frame_sha"] == r1["frame_sha"]
    # the work really was split: neither rank ran all UNet forwards of the sequential engine
    assert r0["unet_calls"] < gold["unet_calls"] and r1["unet_calls"] < gold["unet_calls"]
