#!/usr/bin/env python
"""Headline benchmark: transition frames/sec, SDXL-Turbo 512x512, 4 steps, 15 mid branches
(BASELINE.json metric / configs[1]) on N MI355X of one node.

A "step" is one complete ``BlendingEngine.run_transition(fixed_seeds=[420, 421])`` — anchors not
recycled, 2 anchor trajectories + 15 mid branches -> 17 frames, every UNet forward, VAE decode,
slerp, LPIPS and scheduler step executed by the gfx950 kernels of liblbhip.so (recorded launch
programs replayed as hipGraphs; ``--frontier`` gaps evaluated per batched launch).  Weights are
seeded synthetic SDXL-shaped tensors (no checkpoints offline), conditioning is synthetic.

N > 1 (``torchrun``): one process per GPU, ``torch.distributed`` backend nccl (= RCCL over xGMI).  ONE
transition tree is farmed over the ranks (latentblending_amd/dist/farm.py): anchors on ranks 0/1 and
all-gathered, every speculative round's branches split over the ranks, latent stacks + decoded
frames + LPIPS scalars all-gathered, identical greedy commits everywhere.  Per-GPU work is held
fixed (weak scaling): nmb_max_branches = 15 x N  ->  15N + 2 frames per transition; `value` is the
whole job's frames/s.

The JSON line also carries
  roofline      — the dominant kernel family (MFMA GEMM / implicit-GEMM conv): algorithmic FLOPs
                  per transition / its device time, measured live with hipEvents between the ops
                  of an eager replay of the same launch programs, against the 2.5 PFLOP/s dense
                  fp16 MFMA peak (guide: MI355X_MICROARCH.md);
  cpu_baseline  — the CPU fp32 oracle (oracle/, a restatement = "port") timed on this box's host
                  cores on a bounded sample (1 UNet forward + 1 VAE decode at the same shapes,
                  scaled by the transition's census 38 / 17).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GEMM_OPS = ("lb_gemm_f16", "lb_conv3x3_halo_f16")      # one kernel family: MFMA GEMM / implicit-GEMM conv / halo-tile conv
MFMA_F16_PEAK_TFLOPS = 2500.0     # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed transitions")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frontier", type=int, default=16, help="gap children evaluated per batched round (per GPU)")
    ap.add_argument("--branches", type=int, default=15)
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def census(pipe) -> dict:
    return dict(pipe.stats)


def gemm_family_profile(pipe, launches):
    """Eager hipEvent-timed replay of every launch program the transition used.
    launches: {("unet", B, L): count, ("vae", B, L): count}.  Returns totals per transition."""
    tot = {"gemm_flops": 0.0, "gemm_ms": 0.0, "gemm_launches": 0, "attn_flops": 0.0, "attn_ms": 0.0,
           "other_ms": 0.0, "all_ms": 0.0, "gemm_bytes": 0.0}
    for (kind, B, L), count in launches.items():
        if kind == "unet":
            up = pipe.unet_program(B, L)
            progs = [(up.prog_step, count)]
            em = up.em
            n_cond_gemms = sum(1 for n in up.prog_cond.op_names() if n in GEMM_OPS)
            logs = {id(up.prog_step): (em.gemm_log[n_cond_gemms:], em.attn_log)}
        else:
            vp = pipe.vae_program(B, L)
            progs = [(vp.prog, count)]
            logs = {id(vp.prog): (vp.em.gemm_log, vp.em.attn_log)}
        for prog, cnt in progs:
            prog.time_ops()                       # warm
            ms = prog.time_ops()
            names = prog.op_names()
            glog, alog = logs[id(prog)]
            gi = ai = 0
            for n, t in zip(names, ms):
                tot["all_ms"] += t * cnt
                if n in GEMM_OPS:
                    tot["gemm_flops"] += glog[gi]["flops"] * cnt
                    tot["gemm_bytes"] += glog[gi]["bytes"] * cnt
                    tot["gemm_ms"] += t * cnt
                    tot["gemm_launches"] += cnt
                    gi += 1
                elif n == "lb_attn_fwd_d64":
                    tot["attn_flops"] += alog[ai]["flops"] * cnt
                    tot["attn_ms"] += t * cnt
                    ai += 1
                else:
                    tot["other_ms"] += t * cnt
    return tot


def install_launch_counters(pipe, counts):
    """Wrap the launch method of every recorded UNet step / VAE program so that `counts` ends up holding
    {("unet"|"vae", B, L): launches}.  A dict increment per PROGRAM launch (5 per transition): not measurable."""
    for key, prog in pipe._unet_programs.items():
        orig = prog.prog_step.launch

        def counted(stream=None, _o=orig, _k=key):
            counts[("unet",) + _k] = counts.get(("unet",) + _k, 0) + 1
            return _o(stream)
        prog.prog_step.launch = counted
    for key, prog in pipe._vae_programs.items():
        orig = prog.prog.launch

        def counted_v(stream=None, _o=orig, _k=key):
            counts[("vae",) + _k] = counts.get(("vae",) + _k, 0) + 1
            return _o(stream)
        prog.prog.launch = counted_v


def roofline_block(prof, launch_counts):
    achieved = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12 if prof["gemm_ms"] else 0.0
    return {
        "bound": "mfma", "kernel": "gemm_f16_glds_kernel<BM,BN,CONV,GEGLU,S,WMW> (all Linear/Conv of UNet+VAE)",
        "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F16_PEAK_TFLOPS,
        "traffic": _pmc_traffic_per_launch(),
        "traffic_unit": "HBM-side bytes per GEMM launch (rocprofv3 PMC passes committed in profiles/; null if absent)",
        "algorithmic_bytes_per_launch": prof["gemm_bytes"] / max(prof["gemm_launches"], 1),
        "algorithmic_tflop_per_transition_reference": 103.1,
        "per_transition": {"gemm_tflop": prof["gemm_flops"] / 1e12, "gemm_ms": prof["gemm_ms"],
                           "gemm_launches": prof["gemm_launches"],
                           "gemm_avg_us_per_launch": prof["gemm_ms"] * 1e3 / max(prof["gemm_launches"], 1),
                           "gemm_algorithmic_GBs": prof["gemm_bytes"] / (prof["gemm_ms"] * 1e-3) / 1e9 if prof["gemm_ms"] else 0,
                           "attn_tflop": prof["attn_flops"] / 1e12, "attn_ms": prof["attn_ms"],
                           "attn_TFLOPs": prof["attn_flops"] / (prof["attn_ms"] * 1e-3) / 1e12 if prof["attn_ms"] else 0,
                           "other_kernels_ms": prof["other_ms"], "all_program_ms_eager": prof["all_ms"],
                           "program_launches": {"%s_B%d_L%d" % k: v for k, v in launch_counts.items()}},
    }


def _pmc_traffic_per_launch():
    """HBM traffic of the GEMM family from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE),
    per launch like `achieved`; PMC collection cannot run inside the timed bench itself."""
    path = os.path.join(ROOT, "profiles", "r01_rocprof_summary.json")
    try:
        with open(path) as fh:
            return json.load(fh)["gemm_family_hbm_traffic"]["bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(unet_w, vae_w):
    """Bounded CPU sample on this host: one fp32 UNet forward + one fp32 VAE decode of the oracle
    at the benchmark shapes (B=1, 64x64 latent), scaled by the transition census (38 / 17)."""
    from oracle import sdxl_ref as R
    cores = min(os.cpu_count() or 1, 16)     # more threads than this only slows torch's CPU GEMMs down
    torch.set_num_threads(cores)
    ucfg, vcfg = R.UNetCfg(sample_size=64), R.VAECfg()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 64, 64, generator=g).half()
    ctx = torch.randn(1, 77, 2048, generator=g).half()
    te = torch.randn(1, 1280, generator=g).half()
    ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]])
    t0 = time.perf_counter()
    R.unet_forward(ucfg, unet_w, x, torch.tensor(999.0), ctx, te, ids)
    t_unet = time.perf_counter() - t0
    t0 = time.perf_counter()
    R.postprocess_u8(R.vae_decode(vcfg, vae_w, x.float() / vcfg.scaling_factor))
    t_vae = time.perf_counter() - t0
    t_transition = 38 * t_unet + 17 * t_vae
    return {"value": 17.0 / t_transition, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 (torch CPU, {cores} threads): 1 UNet forward B=1 64x64 latent = {t_unet:.2f} s, "
                      f"1 VAE decode = {t_vae:.2f} s; transition = 38 UNet + 17 VAE (census) = {t_transition:.1f} s "
                      f"extrapolated; LPIPS/slerp/host excluded"}


def main():
    # the host layer prints progress lines like the reference does; keep stdout for the ONE JSON line
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    try:
        out = _run()
    finally:
        sys.stdout = real_stdout
    if out is not None:
        print(json.dumps(out), flush=True)


def _run():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import latentblending_amd.native as N
    from latentblending_amd import BlendingEngine

    # seeded synthetic SDXL weights from the product's own provider; rank 0 keeps the fp32 copies so
    # that the cpu_baseline leg can time the oracle on exactly the same parameters
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    t0 = time.perf_counter()
    unet_prov, vae_prov = N.SyntheticProvider(0, keep=want_cpu), N.SyntheticProvider(1, keep=want_cpu)
    pipe = N.NativeSDXLPipe(turbo=True, unet_provider=unet_prov, vae_provider=vae_prov, device=f"cuda:{local_rank}",
                            allow_synthetic=True)      # ("data": "synthetic" in the JSON line)
    t_weights = time.perf_counter() - t0
    unet_w, vae_w = unet_prov.state, vae_prov.state
    farm = None
    if world > 1:
        from latentblending_amd.dist import BranchFarm
        farm = BranchFarm(device=torch.device("cuda", local_rank))
    be = BlendingEngine(pipe, do_compile=not args.no_graphs, frontier_width=args.frontier * world, verbose=False,
                        farm=farm)
    be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
    be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
    be.set_branching(nmb_max_branches=args.branches * world)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    frames = 0
    for _ in range(args.warmup):
        frames = len(be.run_transition(fixed_seeds=[420, 421]))
    for k in pipe.stats:
        pipe.stats[k] = 0
    farm_counts = {}
    if world > 1 and rank == 0 and not args.no_roofline and args.warmup > 0:
        # a farmed transition cannot be repeated by rank 0 alone (collectives), so rank 0 counts its own program
        # launches during the timed transitions (programs exist after the warm-up) and profiles them afterwards
        install_launch_counters(pipe, farm_counts)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frames = len(be.run_transition(fixed_seeds=[420, 421]))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per_transition = {k: v / max(args.steps, 1) for k, v in census(pipe).items()}
    n_runs = max(args.steps + args.warmup, 1)
    per_transition["frontier_rounds"] = be.stats.get("frontier_rounds", 0) / n_runs
    per_transition["speculation_dropped"] = be.stats.get("speculation_dropped", 0) / n_runs

    out = {
        "metric": "transition frames/sec, SDXL-Turbo 512x512 4-step 15-branch",
        "value": frames * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "SDXL-Turbo 512x512, num_inference_steps=4, nmb_max_branches=%d (%d frames/transition), "
                               "fp16, fixed_seeds=[420,421], anchors not recycled" % (args.branches * world, frames),
                   "frontier_width": args.frontier, "hipgraphs": not args.no_graphs,
                   "parallelism": ("branch farm over %d ranks (RCCL all-gather of anchors / branches)" % world)
                   if world > 1 else "single GPU",
                   "farm": None if farm is None else {"collectives": farm.collectives, "bytes_moved": farm.bytes_moved},
                   "census_per_transition": per_transition, "weights_gen_s": round(t_weights, 1)},
    }
    if rank == 0 and world == 1 and not args.no_roofline:      # (needs a solo transition: no collectives)
        # launches of every (program, batch) per transition: one more transition with counting wrappers
        step_launch_counts = {}
        install_launch_counters(pipe, step_launch_counts)
        be.run_transition(fixed_seeds=[420, 421])
        torch.cuda.synchronize()
        out["roofline"] = roofline_block(gemm_family_profile(pipe, step_launch_counts), step_launch_counts)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(unet_w, vae_w)
    if world > 1:
        dist.destroy_process_group()
        if rank == 0 and farm_counts:
            # rank 0's share of the farmed transition (its UNet / VAE batches), same eager hipEvent replay as N=1
            try:
                per_tr = {k: v / max(args.steps, 1) for k, v in farm_counts.items()}
                out["roofline"] = roofline_block(gemm_family_profile(pipe, per_tr), per_tr)
                out["roofline"]["scope"] = "rank 0's programs of the farmed transition"
            except Exception as exc:                      # never lose the throughput line over the profile
                out["roofline"] = None
                out["roofline_error"] = repr(exc)
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
