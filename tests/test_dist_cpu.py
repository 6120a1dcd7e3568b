"""`gloo` tests of the branch farm on CPU: 2 / 3 / 4 ranks run the SPMD engine on the tiny oracle pipe, split every
speculative round between them (unevenly at world 3), and must all end with exactly the tree the sequential
reference produced (tests/golden/tree.json: fractions, similarities, latents, frames); chained transitions with a
recycled anchor (swap_forward + recycle_img1) must equal the farm-less run; ranks holding different plans must
fail loudly instead of hanging."""
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


@pytest.mark.parametrize("run,world", [(0, 2), (2, 2), (0, 3), (1, 4)])
def test_branch_farm_reproduces_sequential_tree(run, world, tmp_path):
    """world 3 splits rounds of 4 branches 2 / 1 / 1 (and later rounds leave ranks without a branch); world 4 runs a
    3-branch tree, so one rank never owns a mid branch."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, run, str(tmp_path)), nprocs=world, join=True)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tree.json")))[run]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    for r in res:
        assert r["fracts"] == gold["tree_fracts"] and r["idx"] == gold["tree_idx_injection"]
        assert np.allclose(r["sims"], gold["tree_similarities"], rtol=2e-3)
        assert r["collectives"] > 0 and r["bytes"] > 0
        # the work really was split: no rank ran all UNet forwards of the sequential engine
        assert r["unet_calls"] < gold["unet_calls"]
    # SPMD: all ranks hold bit-identical trees, latents and frames
    for r in res[1:]:
        assert r["sims"] == res[0]["sims"] and r["latent_sha"] == res[0]["latent_sha"] and r["frame_sha"] == res[0]["frame_sha"]


def _guidance_chain_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    from latentblending_amd.dist import BranchFarm
    from _baseline_cfgs import check_guidance_chain, run_guidance_chain
    set_backend(R.TorchCpuBackend())
    p = OP.StableDiffusionXLPipeline(turbo=False, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    np.random.seed(0)
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=8, farm=BranchFarm())
    p.noise.reset()
    runs = run_guidance_chain(be)
    check_guidance_chain(be, runs, sim_rtol=2e-3, norm_rtol=2e-3, mean_tol=0.25, head_tol=2, ds_tol=0.5)
    json.dump({"guidance": [r[1] for r in runs]}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_branch_farm_leaves_the_last_committed_branchs_guidance(tmp_path):
    """tests/golden/guidance_chain.json under a world-2 farm at frontier 8: both ranks leave 3.25 behind after each of the two
    chained base-model transitions and compute the reference's second transition (round 5 left the last EVALUATED spec's 3.75)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_guidance_chain_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert json.load(open(tmp_path / f"rank{r}.json"))["guidance"] == [3.25, 3.25]


def _cfg4_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    from latentblending_amd.dist import BranchFarm
    from _baseline_cfgs import check_structure_cfg4_plain, gold_configs, setup_cfg4
    set_backend(R.TorchCpuBackend())
    c = gold_configs()["cfg4"]
    p = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    np.random.seed(0)
    farm = BranchFarm()
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=64, farm=farm)
    setup_cfg4(be)
    p.noise.reset()
    p.unet.calls = p.vae.calls = 0
    imgs = be.run_transition(fixed_seeds=[420, 421])
    check_structure_cfg4_plain(be, imgs, c)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).tobytes()).hexdigest()[:16]
    res = {"sims": [float(s) for s in be.tree_similarities], "latent_sha": [sha(l[-1]) for l in be.tree_latents],
           "frame_sha": [sha(i) for i in imgs], "unet_calls": p.unet.calls, "vae_calls": p.vae.calls,
           "collectives": farm.collectives, "bytes": farm.bytes_moved, "rounds": be.stats.get("frontier_rounds", 0)}
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_branch_farm_cfg4_shape_at_world_8(tmp_path):
    """BASELINE.json configs[3]: SDXL-Turbo, 64 branches on one level, sharded over EIGHT ranks (gloo here, RCCL on the
    node): every rank ends with a 66-frame tree of the reference run's structure (tests/golden/configs.json: complete 1/64 grid
    plus one halved gap; the noise tape is consumed in evaluation order), all ranks bit-identical, and no rank did more than its share of the branch work."""
    import torch.multiprocessing as mp
    world = 8
    port = _free_port()
    mp.spawn(_cfg4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "configs.json")))["cfg4"]
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    for r in res:
        assert len(r["frame_sha"]) == 66 and r["collectives"] > 0 and r["bytes"] > 0
        # sequential census: 136 UNet calls / 66 decodes; a rank runs the two anchors plus at most its share of the mids
        assert r["unet_calls"] < gold["unet_calls"] / 2 and r["vae_calls"] < gold["vae_calls"] / 2
    for r in res[1:]:
        assert r["sims"] == res[0]["sims"] and r["latent_sha"] == res[0]["latent_sha"] and r["frame_sha"] == res[0]["frame_sha"]


def _skew_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import math
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    from latentblending_amd.dist import BranchFarm
    set_backend(R.TorchCpuBackend())

    def run(width, farm):
        p = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
        np.random.seed(0)
        be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=width, farm=farm)
        be.pair_metric = lambda a, b, fa, fb: abs(fa - fb) ** 2 * math.exp(3.0 * 0.5 * (fa + fb))
        be.set_dimensions((64, 64))
        be.set_branching(nmb_max_branches=15)
        be.set_prompt1("photo of a reef")
        be.set_prompt2("rendering of an alien planet")
        p.noise.reset()
        be.run_transition(fixed_seeds=[420, 421])
        return be
    farm = BranchFarm()
    spec = run(16, farm)
    seq = run(1, None)                       # (no collectives: every rank repeats the sequential engine for itself)
    res = {"fracts": [float(f) for f in spec.tree_fracts], "seq_fracts": [float(f) for f in seq.tree_fracts],
           "sims": [float(x) for x in spec.tree_similarities], "rounds": spec.stats.get("frontier_rounds", 0),
           "evaluated": spec.stats.get("speculation_evaluated", 0), "collectives": farm.collectives}
    json.dump(res, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_branch_farm_under_a_skewed_metric_walks_the_same_order_on_every_rank(tmp_path):
    """The forward walk of the frontier (exact + predicted distances, ratios learned from exchanged scalars) is part of the
    SPMD plan: under a metric that bends the tree both ranks must pick the same candidates in every round (else the packed
    all-gathers would pair different branches), finish in two rounds and hold the sequential tree."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_skew_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r in res:
        assert r["fracts"] == r["seq_fracts"] and r["fracts"] != [i / 16 for i in range(17)]
        assert r["rounds"] == 2 and r["evaluated"] == 18 and r["collectives"] > 0
    assert res[0]["sims"] == res[1]["sims"] and res[0]["fracts"] == res[1]["fracts"]


def _chain_worker(rank, world, port, out_dir):
    """Two chained transitions (example_multi_trans.py:39-58: swap_forward + recycle_img1) with ancestral noise from a
    tape; world 1 = the farm-less engine in the same frontier mode."""
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
    p = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    np.random.seed(0)
    farm = BranchFarm() if world > 1 else None
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=3, farm=farm)
    be.set_dimensions((128, 128))
    be.set_branching(nmb_max_branches=4)
    be.set_prompt1("photo of a reef")
    be.set_prompt2("rendering of an alien planet")
    p.noise.reset()
    out = []
    imgs = be.run_transition(fixed_seeds=[420, 421])
    out.append({"fracts": [float(f) for f in be.tree_fracts], "sims": [float(s) for s in be.tree_similarities],
                "frames": [int(np.asarray(i).astype(np.int64).sum()) for i in imgs]})
    be.swap_forward()
    be.set_prompt2("a forest in the fog")
    imgs = be.run_transition(recycle_img1=True, fixed_seeds=[421, 999])
    out.append({"fracts": [float(f) for f in be.tree_fracts], "sims": [float(s) for s in be.tree_similarities],
                "frames": [int(np.asarray(i).astype(np.int64).sum()) for i in imgs],
                "last_norm": float(be.tree_latents[-1][-1].float().norm())})
    json.dump(out, open(os.path.join(out_dir, f"chain_rank{rank}_of{world}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_branch_farm_recycled_anchor_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_chain_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_chain_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    solo = json.load(open(tmp_path / "chain_rank0_of1.json"))
    r0, r1 = [json.load(open(tmp_path / f"chain_rank{r}_of2.json")) for r in (0, 1)]
    assert r0 == r1, "ranks diverged"                       # bit-identical on both ranks, both transitions
    for t in (0, 1):
        assert r0[t]["fracts"] == solo[t]["fracts"]
        assert np.allclose(r0[t]["sims"], solo[t]["sims"], rtol=1e-5)
        assert r0[t]["frames"] == solo[t]["frames"]
    assert abs(r0[1]["last_norm"] - solo[1]["last_norm"]) <= 1e-3 * solo[1]["last_norm"]


def _pipelined_chain_worker(rank, world, port, out_dir):
    """Three prompts, replay.run_multi_transition(pipeline_keyframes=True): key frame k is denoised on rank k % world and
    broadcast; ancestral noise from a tape (non-owners skip the draws); world 1 = the farm-less engine."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine, replay
    from latentblending_amd.backend import set_backend
    from latentblending_amd.dist import BranchFarm
    set_backend(R.TorchCpuBackend())
    p = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    np.random.seed(0)
    farm = BranchFarm() if world > 1 else None
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=3, farm=farm)
    be.set_dimensions((128, 128))
    be.set_branching(nmb_max_branches=4)
    p.noise.reset()
    p.unet.calls = 0
    segs = replay.run_multi_transition(be, ["photo of a reef", "rendering of an alien planet", "a forest in the fog"],
                                       [420, 421, 999], None, pipeline_keyframes=True)
    out = {"frames": [[int(np.asarray(i).astype(np.int64).sum()) for i in seg] for seg in segs],
           "fracts": [float(f) for f in be.tree_fracts], "unet_calls": p.unet.calls,
           "last_norm": float(be.tree_latents[-1][-1].float().norm())}
    json.dump(out, open(os.path.join(out_dir, f"pipe_rank{rank}_of{world}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_branch_farm_pipelined_keyframes_match_single_process(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_pipelined_chain_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_pipelined_chain_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    solo = json.load(open(tmp_path / "pipe_rank0_of1.json"))
    r0, r1 = [json.load(open(tmp_path / f"pipe_rank{r}_of2.json")) for r in (0, 1)]
    assert r0["frames"] == r1["frames"] and r0["fracts"] == r1["fracts"], "ranks diverged"
    assert r0["frames"] == solo["frames"] and r0["fracts"] == solo["fracts"]
    assert len(solo["frames"]) == 2 and solo["frames"][0][-1] == solo["frames"][1][0]      # the shared key frame
    assert abs(r0["last_norm"] - solo["last_norm"]) <= 1e-3 * solo["last_norm"]
    # the three key-frame trajectories were split 2 / 1 between the ranks
    assert r0["unet_calls"] + r1["unet_calls"] < 2 * solo["unet_calls"]


def _mismatch_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    from latentblending_amd.dist import BranchFarm
    set_backend(R.TorchCpuBackend())
    p = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False, frontier_width=2, farm=BranchFarm())
    be.set_dimensions((128, 128))
    be.set_branching(nmb_max_branches=3 + rank)             # ranks disagree about the plan
    be.set_prompt1("a")
    be.set_prompt2("b")
    try:
        be.run_transition(fixed_seeds=[1, 2])
        msg = "no error"
    except RuntimeError as exc:
        msg = str(exc)
    json.dump({"msg": msg}, open(os.path.join(out_dir, f"mismatch_rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_branch_farm_detects_diverging_plans(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_mismatch_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        msg = json.load(open(tmp_path / f"mismatch_rank{r}.json"))["msg"]
        assert "different branching plan" in msg, msg


def test_bench_gpus_flag_spawns_the_ranks(tmp_path):
    """The driver's plain command line `python bench.py --gpus N` must BE an N-rank job: bench.py re-executes itself under
    torch.distributed.run when WORLD_SIZE is unset, checks that the world size equals --gpus and prints ONE JSON line from
    rank 0.  --rendezvous-only exercises exactly that entry path on CPU (gloo, no workload); a world / --gpus mismatch must
    fail loudly instead of silently benchmarking one GPU."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rendezvous_only"] is True
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env1, cwd=str(tmp_path))
    assert r.returncode != 0 and "--gpus 2 but 1 rank" in r.stderr
