#!/usr/bin/env python
"""Headline benchmark: transition frames/sec, SDXL-Turbo 512x512, 4 steps, 15 mid branches
(BASELINE.json metric / configs[1]) on N MI355X of one node.

A "step" is one complete ``BlendingEngine.run_transition(fixed_seeds=[420, 421])`` — anchors not
recycled, 2 anchor trajectories + 15 mid branches -> 17 frames, every UNet forward, VAE decode,
slerp, LPIPS and scheduler step executed by the gfx950 kernels of liblbhip.so (recorded launch
programs replayed as hipGraphs; ``--frontier`` gaps evaluated per batched launch).  Weights are
seeded synthetic SDXL-shaped tensors (no checkpoints offline), conditioning is synthetic.

N > 1 (``torchrun``): one process per GPU, ``torch.distributed`` backend nccl (= RCCL over xGMI).  ONE
transition tree is farmed over the ranks (latentblending_amd/dist/farm.py): every rank runs both anchors plus
its share of the mid branches in one wavefront, ONE packed all-gather of (latent stack, frame) per round and one
of the sharded LPIPS scalars, identical greedy commits everywhere.
  --scaling strong (default): the metric's own workload — 15 mid branches (17 frames) whatever N is;
  --scaling weak: per-GPU work fixed, nmb_max_branches = 15 x N -> 15N + 2 frames per transition;
  --branches-total K: K mid branches in total (BASELINE configs[3]: 64 branches on 8 GPUs).
`value` is the whole job's frames/s.  --metric-skew S replaces the perceptual metric by a deliberately skewed
one (distance x exp(S x position)): the greedy order then leaves the balanced tree and the speculative frontier needs
several small rounds — reported through census_per_transition.frontier_rounds / speculation_hit_rate.

The JSON line also carries
  roofline      — the dominant kernel family (MFMA GEMM / implicit-GEMM conv / halo-tile conv): algorithmic
                  FLOPs per transition / its device time, measured live with hipEvents between the ops of an
                  eager replay of the same launch programs, against the 2.5 PFLOP/s dense fp16 MFMA peak;
  rooflines     — the same for every other kernel class on the path: attention (MFMA), GroupNorm / LayerNorm (HBM),
                  the latent-mixing slerp (HBM: the engine's native-size launches and a >= 1 GiB batch timed here),
                  scheduler input scaling + Euler step (HBM);
  cpu_baseline  — the CPU fp32 oracle (oracle/, a restatement = "port") timed on this box's host
                  cores on a bounded sample (UNet forwards + a VAE decode + LPIPS + slerps at the same shapes,
                  scaled by the transition's census 38 / 17 / 30 / 60).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GEMM_OPS = ("lb_gemm_f16", "lb_conv3x3_halo_f16", "lb_upconv2x_halo_f16", "lb_conv3x3_narrow_f16")      # one kernel family: MFMA GEMM / implicit-GEMM conv / halo-tile conv
MFMA_F16_PEAK_TFLOPS = 2500.0     # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed transitions")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frontier", type=int, default=16, help="gap children evaluated per batched round (per GPU)")
    ap.add_argument("--branches", type=int, default=15, help="mid branches of the metric's workload")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--branches-total", type=int, default=0, help="total mid branches (overrides --scaling)")
    ap.add_argument("--metric-skew", type=float, default=0.0, help="skew the perceptual metric by exp(S x position)")
    ap.add_argument("--config", choices=("cfg2", "cfg3"), default="cfg2",
                    help="cfg2 = the metric's workload (SDXL-Turbo 512^2, 4 steps); cfg3 = BASELINE configs[2]: SDXL base "
                         "1024^2, 30 steps, guidance 4.0, depth_strength 0.5, nmb_max_branches=15 (a secondary line)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (cfg 3 and the skewed-metric run) of the N = 1 record")
    ap.add_argument("--torch-baseline", action="store_true",
                    help="opt-in diagnostic instead of the metric: the oracle's PyTorch graph of the UNet step and the VAE decode executed on the "
                         "SAME GPU through the vendor libraries (torch eager under autocast fp16: rocBLAS / hipBLASLt, MIOpen, torch SDPA), "
                         "beside this repo's launch programs on the same weights and inputs")
    ap.add_argument("--no-materialise", action="store_true",
                    help="leave the returned frames in HBM (lazy PIL images) instead of copying them to host PIL images inside the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only with --rendezvous-only)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="entry-path check without a GPU (tests/test_dist_cpu.py): spawn / join the ranks, verify the world size, "
                         "print the JSON line with value null, run no workload")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` started as ONE process (no WORLD_SIZE in the environment): re-execute this very command
    line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one rank per GPU, rendezvous on 127.0.0.1),
    so the driver's plain command and its explicit torchrun command are the same job.  Never returns when it re-executes."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    # --standalone: torchrun itself picks AND HOLDS a free rendezvous port (binding port 0 here and closing the socket
    # before exec left a window in which another process could take the port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"[bench] --gpus {args.gpus} without WORLD_SIZE: re-executing under torch.distributed.run\n")
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def gemm_family_profile(pipe, launches):
    """Eager hipEvent-timed replay of every launch program the transition used.
    launches: {("unet", B, L): count, ("vae", B, L): count}.  Returns totals per transition."""
    tot = {"gemm_flops": 0.0, "gemm_ms": 0.0, "gemm_launches": 0, "attn_flops": 0.0, "attn_ms": 0.0, "attn_launches": 0,
           "other_ms": 0.0, "all_ms": 0.0, "gemm_bytes": 0.0, "halo_ms": 0.0, "halo_flops": 0.0, "halo_launches": 0,
           "gn_ms": 0.0, "gn_bytes": 0.0, "gn_launches": 0, "ln_ms": 0.0, "ln_bytes": 0.0, "ln_launches": 0,
           "attn_self_ms": 0.0, "attn_self_flops": 0.0, "attn_cross_ms": 0.0, "attn_cross_flops": 0.0}
    for (kind, B, L), cnt in launches.items():
        if kind == "unet":
            up = pipe.unet_program(B, L)
            prog, em = up.prog_step, up.em
            n_cond_gemms = sum(1 for n in up.prog_cond.op_names() if n in GEMM_OPS)
            glog, alog, nlog = em.gemm_log[n_cond_gemms:], em.attn_log, em.norm_log
        else:
            vp = pipe.vae_program(B, L)
            prog, glog, alog, nlog = vp.prog, vp.em.gemm_log, vp.em.attn_log, vp.em.norm_log
        prog.time_ops()                       # warm
        ms = prog.time_ops()
        names = prog.op_names()
        gi = ai = ni = 0
        for n, t in zip(names, ms):
            tot["all_ms"] += t * cnt
            if n in GEMM_OPS:
                tot["gemm_flops"] += glog[gi]["flops"] * cnt
                tot["gemm_bytes"] += glog[gi]["bytes"] * cnt
                tot["gemm_ms"] += t * cnt
                tot["gemm_launches"] += cnt
                if n != "lb_gemm_f16":
                    tot["halo_ms"] += t * cnt
                    tot["halo_flops"] += glog[gi]["flops"] * cnt
                    tot["halo_launches"] += cnt
                gi += 1
            elif n in ("lb_attn_fwd_d64", "lb_attn_fwd_d512"):
                a = alog[ai]
                tot["attn_flops"] += a["flops"] * cnt
                tot["attn_ms"] += t * cnt
                tot["attn_launches"] += cnt
                key = "attn_self" if a.get("Skv", 0) == a.get("Sq", -1) else "attn_cross"
                tot[key + "_ms"] += t * cnt
                tot[key + "_flops"] += a["flops"] * cnt
                ai += 1
            elif n in ("lb_groupnorm_nhwc", "lb_groupnorm_from_stats", "lb_layernorm_f16"):
                k = "ln" if n == "lb_layernorm_f16" else "gn"
                while ni < len(nlog) and nlog[ni]["op"] != n:      # (the log is in emission order, both kinds mixed)
                    ni += 1
                tot[k + "_ms"] += t * cnt
                tot[k + "_launches"] += cnt
                tot[k + "_bytes"] += (nlog[ni]["bytes"] if ni < len(nlog) else 0.0) * cnt
                ni += 1
            else:
                tot["other_ms"] += t * cnt
    return tot


def phase_split(pipe, be):
    """SURVEY.md §8(d) "per-phase split": ONE more transition with hipEvent pairs around every UNet step program launch, every
    VAE program launch and every perceptual-distance call (graph replays, in situ), and the host copy timed on the host.
    "mixing_and_gaps" is the remainder: slerp / crossfeed / scale / Euler kernels between the programs plus launch gaps."""
    from latentblending_amd.native.frames import materialise_frames
    spans = {"unet": [], "vae": [], "lpips": []}
    undo = []

    def wrap(obj, name, key):
        orig = getattr(obj, name)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            spans[key].append((e0, e1))
            return r
        setattr(obj, name, timed)
        undo.append((obj, name, orig))
    for prog in pipe._unet_programs.values():
        wrap(prog.prog_step, "launch", "unet")
    for prog in pipe._vae_programs.values():
        wrap(prog.prog, "launch", "vae")
    wrap(pipe, "native_frame_distances", "lpips")
    try:
        host_frames, be.host_frames = be.host_frames, False
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        imgs = be.run_transition(fixed_seeds=[420, 421])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        materialise_frames(imgs)
        t2 = time.perf_counter()
    finally:
        be.host_frames = host_frames
        for obj, name, orig in undo:
            setattr(obj, name, orig)
    ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in spans.items()}
    total = (t1 - t0) * 1e3
    return {"unet_ms": ms["unet"], "vae_ms": ms["vae"], "lpips_ms": ms["lpips"], "host_frames_ms": (t2 - t1) * 1e3,
            "mixing_and_gaps_ms": total - ms["unet"] - ms["vae"] - ms["lpips"], "transition_ms_without_host_copy": total,
            "launches": {k: len(v) for k, v in spans.items()},
            "note": "hipEvents around the program launches (graph replays) of one extra transition; comm = 0 at N = 1"}


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


def _tf(flops, ms):
    return flops / (ms * 1e-3) / 1e12 if ms else 0.0


def _gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9 if ms else 0.0


def _event_time_us(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def mixing_rooflines(device, G=15, L=64):
    """Latent-mixing primitives, timed live on the current stream: the engine's native-size launches (latency-bound)
    and a >= 1 GiB batch of the same kernels against the 8 TB/s HBM roof (6 B / element: read p0, p1, write out;
    Euler-ancestral step 8 B / element: read x, eps, noise, write x)."""
    from latentblending_amd.hip import ops
    from latentblending_amd.hip.lib import api
    n = 4 * L * L
    out = []
    a, b = torch.randn(1, n, device=device).half(), torch.randn(1, n, device=device).half()
    fr = torch.rand(G, device=device, dtype=torch.float64)
    us_native = _event_time_us(lambda: ops.slerp_strided(a, b, fr, n, broadcast0=True, broadcast1=True), iters=50)
    pairs = (1 << 30) // (n * 2 * 3)
    p0, p1 = torch.randn(pairs, n, device=device).half(), torch.randn(pairs, n, device=device).half()
    frb = torch.rand(pairs, device=device, dtype=torch.float64)
    ob = torch.empty_like(p0)
    us_big = _event_time_us(lambda: ops.slerp_strided(p0, p1, frb, n, out=ob), iters=5, warm=2)
    gbs = pairs * n * 6 / us_big / 1e3
    # `achieved` / `frac` are THIS run's measurement (hipEvents on the launch stream around the >= 1 GiB batch): a regression, another box
    # or another commit shows up in them.  The rocprofv3-reported figure of the last committed run of tools/mixing_rocprof.py (kernel
    # durations instead of hipEvents around a host call; usually ~15 % lower) is quoted beside it in its own field, tagged with its profile.
    rocprof = None
    for tag in ("r06", "r05", "r04", "r03"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_mixing_rocprof.json")) as fh:
                k = next(k for k in json.load(fh)["kernels"] if "slerp_strided" in k["name"])
            rocprof = {"GB_per_s": k["GB_per_s"], "frac": k["GB_per_s"] / HBM_PEAK_GBS, "profile": f"profiles/{tag}_mixing_rocprof.json"}
            break
        except Exception:
            continue
    out.append({"kernel": "slerp_strided_kernel (interpolate_spherical: parental mix / crossfeed)", "bound": "hbm",
                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "achieved_source": "hipEvents of this run (launch stream, >= 1 GiB batch)",
                "rocprof_reported": rocprof,
                "algorithmic_bytes_per_element": 6, "batch": {"pairs": pairs, "elements_per_pair": n, "us": us_big},
                "native_launch": {"pairs": G, "elements_per_pair": n, "us": us_native,
                                  "note": "one launch per denoising step of the wavefront: launch-latency bound"}})
    # scheduler: x_in = x / sqrt(sigma^2 + 1) and the Euler-ancestral update on the same >= 1 GiB batch
    params = torch.zeros(pairs, 8, dtype=torch.float32, device=device)
    params[:, 0], params[:, 1], params[:, 2], params[:, 4] = 1.6129, 0.6374, 0.6259, -0.9755
    st = torch.cuda.current_stream().cuda_stream
    us_scale = _event_time_us(lambda: api.lb_scale_model_input_f16(p0.data_ptr(), ob.data_ptr(), params.data_ptr(), n, pairs, 0, st),
                              iters=5, warm=2)
    us_step = _event_time_us(lambda: api.lb_euler_step_f16(p0.data_ptr(), p1.data_ptr(), ob.data_ptr(), ob.data_ptr(),
                                                           params.data_ptr(), n, pairs, 0, 1, st), iters=5, warm=2)
    g1, g2 = pairs * n * 4 / us_scale / 1e3, pairs * n * 8 / us_step / 1e3
    out.append({"kernel": "scale_input_kernel (scheduler.scale_model_input)", "bound": "hbm", "achieved": g1,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g1 / HBM_PEAK_GBS, "algorithmic_bytes_per_element": 4})
    out.append({"kernel": "euler_step_kernel (Euler-ancestral scheduler.step)", "bound": "hbm", "achieved": g2,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g2 / HBM_PEAK_GBS, "algorithmic_bytes_per_element": 8})
    del p0, p1, ob
    torch.cuda.empty_cache()
    return out


def roofline_blocks(prof, launch_counts, device):
    achieved = _tf(prof["gemm_flops"], prof["gemm_ms"])
    dominant = {
        "bound": "mfma", "kernel": "gemm_f16_pp_kernel<GEGLU> + gemm_f16_glds_kernel<BM,BN,CONV,GEGLU,S,WMW,LNA> + conv3x3_halo_kernel<BN,TW> "
                                   "(every Linear / Conv of UNet + VAE)",
        "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_F16_PEAK_TFLOPS,
        "traffic": _pmc_traffic_per_launch()[0],
        "traffic_measured_at": _pmc_traffic_per_launch()[1],
        "measured_by": "hipEvents between the ops of an EAGER replay of the recorded launch programs, in this run (about 2-4 us of event boundary "
                       "per op included: the conservative figure; the rocprofv3 kernel-duration figure of the committed run of the same command is in "
                       "profiles/r06_rocprof_summary.json and is ~5 % higher)",
        "headroom_note": "like-for-like against the vendor library (rocBLAS gemm_ex, no epilogue operands on either side; "
                         "profiles/r06_gemm_bench_final.txt): round 6 runs the 192x128 tile as 8 waves of 48x64 (every SIMD issues the same MFMA "
                         "count) with lean request addressing: M 4352 x N 1280 at K 1280 / 2560 / 5120 = 20.3 / 35.1 / 64.1 us against the "
                         "library's 19.2 / 35.6 / 97.6; ahead or within 3 % on 7 of 10 shapes of the programs, 4-10 % behind on q|k|v, the K = 1280 "
                         "projection and the 640-wide GEGLU (single-partial-round grids: its stream-K tiles balance 230 tiles over 256 CUs); PMC of the halo conv (profiles/r05_halo_pmc_lean_epilogue.json): matrix pipe 51 % busy on "
                         "the VAE's big shapes, 65 % on the UNet's deep-K shape, 0.5 LDS instructions per MFMA, LDS pipe 25 % busy - wave time goes "
                         "to vmcnt / barrier waits; the 3x3 convolutions against MIOpen's best solver on the same operands (torch conv2d, find mode; "
                         "profiles/r06_conv_vs_miopen.txt): the halo-tile kernel is 1.2 - 1.7x the vendor library on every shape of the programs",
        "traffic_unit": "HBM-side bytes per GEMM/conv launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                        "command committed in profiles/; null if absent)",
        "algorithmic_bytes_per_launch": prof["gemm_bytes"] / max(prof["gemm_launches"], 1),
        "algorithmic_tflop_per_transition_reference": 103.1,
        "per_transition": {"gemm_tflop": prof["gemm_flops"] / 1e12, "gemm_ms": prof["gemm_ms"],
                           "gemm_launches": prof["gemm_launches"],
                           "gemm_avg_us_per_launch": prof["gemm_ms"] * 1e3 / max(prof["gemm_launches"], 1),
                           "gemm_algorithmic_GBs": _gbs(prof["gemm_bytes"], prof["gemm_ms"]),
                           "halo_conv_tflop": prof["halo_flops"] / 1e12, "halo_conv_ms": prof["halo_ms"],
                           "halo_conv_TFLOPs": _tf(prof["halo_flops"], prof["halo_ms"]),
                           "attn_ms": prof["attn_ms"], "groupnorm_ms": prof["gn_ms"], "layernorm_ms": prof["ln_ms"],
                           "other_kernels_ms": prof["other_ms"], "all_program_ms_eager": prof["all_ms"],
                           "program_launches": {"%s_B%d_L%d" % k: v for k, v in launch_counts.items()}},
    }
    a = _tf(prof["attn_flops"], prof["attn_ms"])
    gn = _gbs(prof["gn_bytes"], prof["gn_ms"])
    rest = [
        {"kernel": "attn_fwd_d64_stream_kernel<QG> (self-attention, round 6) + attn_fwd_d64_kernel<96,QG,1> (cross-attention, one tile) + "
                   "attn_fwd_d512_kernel (VAE mid block), in situ", "bound": "mfma", "achieved": a,
         "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": a / MFMA_F16_PEAK_TFLOPS,
         "measured_by": "hipEvents between the ops of an eager replay, this run (PMC of the kernels: profiles/r06_attention_pmc.json)",
         "headroom_note": "issue-bound on the softmax's VALU work at 64-wide heads, not on the matrix pipe (PMC); against the vendor's fused attention on "
                          "the same operands (torch SDPA on ROCm, profiles/r06_attn_vs_sdpa.txt): self-attention S = 256 15.0 vs 27.9 us, S = 1024 71.1 vs "
                          "103.4 us, cross-attention 11.3 vs 18.6 us, VAE d = 512 1.00 vs 2.54 ms - 1.4 - 2.6x the library",
         "per_transition": {"tflop": prof["attn_flops"] / 1e12, "ms": prof["attn_ms"], "launches": prof["attn_launches"],
                            "self_TFLOPs": _tf(prof["attn_self_flops"], prof["attn_self_ms"]), "self_ms": prof["attn_self_ms"],
                            "cross_TFLOPs": _tf(prof["attn_cross_flops"], prof["attn_cross_ms"]), "cross_ms": prof["attn_cross_ms"]}},
        {"kernel": "gn_partial_kernel + gn_apply_kernel / gn_fold_stats_kernel + gn_apply_kernel / gn_fused_kernel<NPX> (GroupNorm + SiLU: 6 B per "
                   "element, 4 B where the producing conv's epilogue left the statistics or the one-launch form keeps the slab in registers)", "bound": "hbm",
         "achieved": gn, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gn / HBM_PEAK_GBS,
         "measured_by": "hipEvents between the ops of an eager replay, this run",
         "per_transition": {"ms": prof["gn_ms"], "launches": prof["gn_launches"], "GB": prof["gn_bytes"] / 1e9}},
    ]
    if prof["ln_launches"]:
        ln = _gbs(prof["ln_bytes"], prof["ln_ms"])
        rest.append({"kernel": "layernorm_rows_kernel<NV>", "bound": "hbm", "achieved": ln, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ln / HBM_PEAK_GBS, "measured_by": "hipEvents between the ops of an eager replay, this run", "per_transition": {"ms": prof["ln_ms"], "launches": prof["ln_launches"]}})
    else:
        rest.append({"kernel": "layernorm", "note": "no LayerNorm launch exists: folded into the consuming GEMMs (LB_GEMM_LN_A)"})
    try:
        rest += mixing_rooflines(device)
    except Exception as exc:                              # never lose the throughput line over a side measurement
        rest.append({"kernel": "slerp / scheduler", "error": repr(exc)})
    return dominant, rest


def _pmc_traffic_per_launch():
    """HBM traffic of the GEMM family from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE), per launch
    like `achieved`, and where that was measured (profile file, commit); PMC collection cannot run inside the timed bench."""
    for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.json")
        try:
            with open(path) as fh:
                d = json.load(fh)
            return d["gemm_family_hbm_traffic"]["bytes_per_launch"], {"profile": f"profiles/{tag}_rocprof_summary.json",
                                                                      "commit": d.get("commit", "not recorded")}
        except Exception:
            continue
    return None, None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; falls back to os.cpu_count()."""
    seen, phys, core = set(), None, None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip() and phys is not None and core is not None:
                    seen.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
    except OSError:
        pass
    return len(seen) or (os.cpu_count() or 1)


def same_gpu_torch_baseline(pipe, unet_w, vae_w, device):
    """Opt-in (`--torch-baseline`): what the VENDOR STACK makes of the same arithmetic on the same MI355X.  The oracle's PyTorch graph
    (oracle/sdxl_ref.py: the restatement of the diffusers UNet / VAE the reference calls at diffusers_holder.py:336,135) runs on the GPU,
    eager, under torch.autocast(float16): linear / conv / matmul in fp16 on rocBLAS / hipBLASLt / MIOpen, norms and softmax in fp32 as
    autocast leaves them, attention through torch's fused scaled_dot_product_attention (the oracle's explicit softmax(QK^T)V would
    understate the stack), convolutions in MIOpen's immediate mode (what an untuned diffusers pipeline gets).  Beside it: this repo's
    launch programs (hipGraph replay) on the same weights, batch and resolution.  The oracle is the yardstick here, as in cpu_baseline -
    never part of the product path."""
    import torch.nn.functional as F
    from oracle import sdxl_ref as R
    dev = torch.device(device)
    ucfg, vcfg = R.UNetCfg(sample_size=64), R.VAECfg()

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def sdpa(q, k, v, heads):
        B, Sq, C = q.shape
        d = C // heads
        sp = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        return F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, Sq, C)

    out = {"how": "oracle/sdxl_ref.py graph on the same GPU: torch eager, autocast fp16 (rocBLAS / hipBLASLt GEMMs, MIOpen convs in immediate "
                  "mode, torch SDPA attention; norms / softmax fp32 as autocast keeps them); ours = hipGraph replay of the launch programs"}
    keep_attention = R.attention
    R.attention = sdpa
    try:
        with torch.no_grad():
            wu = {k: v.to(dev) for k, v in unet_w.items()}
            for B in (17, 2):
                g = torch.Generator().manual_seed(B)
                x = torch.randn(B, 4, 64, 64, generator=g).half().to(dev)
                ctx = torch.randn(B, 77, 2048, generator=g).half().to(dev)
                te = torch.randn(B, 1280, generator=g).half().to(dev)
                ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B, device=dev)
                t = torch.tensor(499.0, device=dev)
                with torch.autocast("cuda", dtype=torch.float16):
                    ms_t = timed(lambda: R.unet_forward(ucfg, wu, x, t, ctx, te, ids), 3)
                prog = pipe.unet_program(B, 64)
                prog.set_conditioning(ctx, te, ids)
                prog.forward(x, torch.full((B,), 499.0))
                prog.enable_graphs()
                ms_o = timed(prog.prog_step.launch, 5)
                out[f"unet_step_B{B}_ms"] = {"torch_eager_vendor_stack": round(ms_t, 2), "this_repo": round(ms_o, 2), "ratio": round(ms_t / ms_o, 2)}
            del wu
            torch.cuda.empty_cache()
            wv = {k: v.to(dev) for k, v in vae_w.items()}
            for B in (17,):
                z = torch.randn(B, 4, 64, 64, generator=torch.Generator().manual_seed(3)).half().to(dev)
                with torch.autocast("cuda", dtype=torch.float16):
                    ms_t = timed(lambda: R.vae_decode(vcfg, wv, z), 2)
                ms32 = timed(lambda: R.vae_decode(vcfg, wv, z), 1)      # fp32: what diffusers' force_upcast VAE (SDXL's default) runs
                vp = pipe.vae_program(B, 64)
                vp.decode(z)
                vp.prog.instantiate()
                ms_o = timed(vp.prog.launch, 5)
                out[f"vae_decode_B{B}_ms"] = {"torch_eager_vendor_stack": round(ms_t, 2), "torch_eager_fp32_force_upcast": round(ms32, 2),
                                               "this_repo": round(ms_o, 2), "ratio": round(ms_t / ms_o, 2)}
    finally:
        R.attention = keep_attention
    return out


def cpu_baseline(unet_w, vae_w, census):
    """The CPU fp32 oracle (oracle/, kind "port") on this box's host cores:
    (0) a THREAD SWEEP of one full-size UNet forward (16 / 32 / 64 / physical cores / all hardware threads, whatever the box
        has): the thread count that is fastest here is the one everything below uses, and every timing is in the line;
    (1) the metric's workload (cfg 2) - 38 UNet forwards, 17 decodes, the LPIPS policy, 17 frames - really RUN once end to end
        through the engine's host layer on the oracle pipe when the timed samples predict <= LB_CPU_BASELINE_BUDGET seconds
        (default 180), otherwise extrapolated from the samples x the transition census (the line says which);
    (2) BASELINE.md section 4's cfg-1-scale tree (SDXL-Turbo 256^2, the smallest valid tree), also run."""
    import contextlib
    import io
    from oracle import pipe as OP, sdxl_ref as R
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    ucfg, vcfg = R.UNetCfg(sample_size=64), R.VAECfg()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 64, 64, generator=g).half()
    ctx = torch.randn(1, 77, 2048, generator=g).half()
    te = torch.randn(1, 1280, generator=g).half()
    ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]])
    logical, physical = os.cpu_count() or 1, _physical_cores()
    sweep = {}
    for n in sorted({min(16, logical), min(32, logical), min(64, logical), min(physical, logical), logical}):
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        R.unet_forward(ucfg, unet_w, x, torch.tensor(999.0), ctx, te, ids)
        sweep[n] = time.perf_counter() - t0
        if sweep[n] > 4 * min(sweep.values()) or sum(sweep.values()) > 40:      # (keep the sweep itself bounded)
            break
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    t_unet = sweep[cores]
    t0 = time.perf_counter()
    img = R.vae_decode(vcfg, vae_w, x.float() / vcfg.scaling_factor)
    R.postprocess_u8(img)
    t_vae = time.perf_counter() - t0
    lp = R.OracleLPIPS(7)
    t0 = time.perf_counter()
    lp(img.clamp(-1, 1), img.flip(-1).clamp(-1, 1))
    t_lpips = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(16):
        R.slerp(x, x.flip(-1), k / 16.0)
    t_slerp = (time.perf_counter() - t0) / 16
    n_unet, n_vae = census.get("unet_samples") or 38.0, census.get("vae_decodes") or 17.0
    n_lp, n_sl = census.get("lpips_pairs") or 30.0, census.get("slerps") or 60.0
    t_transition = n_unet * t_unet + n_vae * t_vae + n_lp * t_lpips + n_sl * t_slerp
    host = {"cpu_model": _cpu_model(), "os_cpu_count": logical, "physical_cores": physical, "torch_threads": cores,
            "unet_forward_seconds_by_threads": {str(k): round(v, 3) for k, v in sweep.items()}}
    samples = (f"oracle fp32 (torch CPU, {cores} threads - the fastest of {sorted(sweep)} on {physical} cores / {logical} threads of "
               f"{_cpu_model()}): 1 UNet forward B=1 64x64 latent = {t_unet:.2f} s, 1 VAE decode = {t_vae:.2f} s, 1 LPIPS pair = "
               f"{t_lpips:.2f} s, 1 slerp = {t_slerp * 1e3:.2f} ms; census {n_unet:.0f} UNet + {n_vae:.0f} VAE + {n_lp:.0f} LPIPS + "
               f"{n_sl:.0f} slerp -> {t_transition:.1f} s predicted")
    out = {"value": n_vae / t_transition, "unit": "frames/s", "cores": cores, "kind": "port", "host": host,
           "measured": False, "sample": samples + " (EXTRAPOLATED: the prediction exceeds the budget for a real run)"}
    budget = float(os.environ.get("LB_CPU_BASELINE_BUDGET", "180"))

    def oracle_engine(size, steps, depth, branches):
        o = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=ucfg, vae_cfg=vcfg, weights=unet_w, vae_weights=vae_w)
        BlendingEngine.benchmark_speed, keep = (lambda self: None), BlendingEngine.benchmark_speed   # (its probe forwards are not part of the tree)
        try:
            be = BlendingEngine(o, metric=lp, verbose=False)
        finally:
            BlendingEngine.benchmark_speed = keep
        be.dt_vae = 0.0
        be.set_dimensions((size, size))
        if steps is not None:
            be.set_num_inference_steps(steps)
        be.set_branching(depth_strength=depth, nmb_max_branches=branches)
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
        o.unet.calls = o.vae.calls = 0
        return o, be
    if t_transition <= budget:
        try:    # (1) cfg 2 itself: SDXL-Turbo 512x512, 4 steps, 15 branches - the reference's sequential loop, RUN
            set_backend(R.TorchCpuBackend())
            with contextlib.redirect_stdout(io.StringIO()):
                o, be = oracle_engine(512, None, None, int(round(n_vae)) - 2)
                t0 = time.perf_counter()
                frames = be.run_transition(fixed_seeds=[420, 421])
                dt = time.perf_counter() - t0
            out.update({"value": len(frames) / dt, "measured": True, "seconds": dt, "frames": len(frames),
                        "unet_forwards": o.unet.calls, "vae_decodes": o.vae.calls,
                        "sample": f"cfg 2 RUN once, not extrapolated: {len(frames)} frames in {dt:.1f} s ({o.unet.calls} UNet forwards, "
                                  f"{o.vae.calls} decodes) through THIS REPO's host layer (latentblending_amd.BlendingEngine; the reference's "
                                  f"own host classes cannot run here: /root/reference does not exist on the GPU box) on the CPU fp32 "
                                  f"oracle pipe; " + samples})
        except Exception as exc:
            out["run_error"] = repr(exc)
        finally:
            set_backend(None)
    # (2) the cfg-1-scale tree, really run: engine host layer on the oracle pipe
    try:
        set_backend(R.TorchCpuBackend())
        with contextlib.redirect_stdout(io.StringIO()):
            o, be = oracle_engine(256, 2, 0.5, 3)
            t0 = time.perf_counter()
            frames = be.run_transition(fixed_seeds=[420, 421])
            dt = time.perf_counter() - t0
        out["cfg1_tree"] = {"value": len(frames) / dt, "unit": "frames/s", "seconds": dt, "frames": len(frames),
                            "unet_forwards": o.unet.calls, "vae_decodes": o.vae.calls,
                            "workload": "SDXL-Turbo 256x256, num_inference_steps=2, depth_strength=0.5, nmb_max_branches=3 (BASELINE "
                                        "configs[0] as its smallest valid tree), CPU fp32 oracle pipe under the engine's host layer, RUN not extrapolated"}
    except Exception as exc:                                  # never lose the throughput line over the baseline
        out["cfg1_tree"] = {"error": repr(exc)}
    finally:
        set_backend(None)
    return out


def skewed_metric(be, skew):
    """Policy stress test: the pipe's native LPIPS times exp(skew x mean position of the two frames).  The tree stays
    exact (same greedy rule, same metric on both the sequential and the speculative path), only far less balanced."""
    pipe = be.dh.pipe

    def similarity(a, b, fa, fb):
        d = pipe.native_frame_distances([(a, b)])[0]
        fa, fb = (0.5 if fa is None else fa), (0.5 if fb is None else fb)
        return d * math.exp(skew * 0.5 * (fa + fb))
    return similarity


def secondary_lines(pipe, args, branches):
    """More lines for the driver's record (same process, same resident weights), each best-effort:
    * the metric's workload under a deliberately SKEWED perceptual metric (distance x exp(3 x position)): the greedy order
      leaves the balanced tree, the speculative frontier needs several rounds and drops speculated branches - the other end
      of the range real LPIPS on real images will sit in (the headline's synthetic weights give a balanced tree: 1 round);
    * the opt-in dead-step elision; a chain of transitions with and without pipelined key frames;
    * BASELINE configs[2]: SDXL base 1024^2, 30 steps, guidance 4.0 (CFG), depth_strength 0.5, nmb_max_branches 15."""
    import latentblending_amd.native as N
    from latentblending_amd import BlendingEngine
    from latentblending_amd.native.frames import materialise_frames
    out = []

    def timed(be, steps, warmup):
        for _ in range(warmup):
            be.run_transition(fixed_seeds=[420, 421])
        be.stats.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            imgs = be.run_transition(fixed_seeds=[420, 421])
            materialise_frames(imgs)
        torch.cuda.synchronize()
        return len(imgs), (time.perf_counter() - t0) / steps

    try:
        be = BlendingEngine(pipe, do_compile=not args.no_graphs, frontier_width=args.frontier, verbose=False)
        be.host_frames = True
        be.pair_metric = skewed_metric(be, 3.0)
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
        be.set_branching(nmb_max_branches=branches)
        n, dt = timed(be, 3, 1)
        ev = be.stats.get("speculation_evaluated", 0) / 3
        dr = be.stats.get("speculation_dropped", 0) / 3
        out.append({"name": "cfg2 under a skewed metric (x exp(3 x position))", "value": n / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                    "frontier_rounds": be.stats.get("frontier_rounds", 0) / 3, "speculation_hit_rate": (ev - dr) / ev if ev else None})
        # opt-in prior (BlendingEngine.speculate_from_previous_tree, round 6): the blind first round takes its 15 candidates from the
        # commit order of the PREVIOUS transition on the engine instead of the level order of the binary splitting.  Every branch is
        # still denoised / decoded / scored in the timed transition and the tree is the greedy tree; what is remembered is only which
        # gaps the greedy order asked for last time.  Same skewed metric: ONE round instead of two (the warm-up transition sets the prior).
        be.speculate_from_previous_tree = True
        n, dt = timed(be, 3, 1)
        ev = be.stats.get("speculation_evaluated", 0) / 3
        dr = be.stats.get("speculation_dropped", 0) / 3
        out.append({"name": "cfg2 under a skewed metric (x exp(3 x position)), speculate_from_previous_tree=True (the previous transition's commit "
                            "order as the prior of the blind first round; opt-in)", "value": n / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                    "frontier_rounds": be.stats.get("frontier_rounds", 0) / 3, "speculation_hit_rate": (ev - dr) / ev if ev else None})
        be.speculate_from_previous_tree = False
        # the same two metrics under TWO-STAGE speculation (BlendingEngine.two_stage_speculation: 7 branches with the anchors, then
        # 8 chosen from known distances): two rounds whatever the metric - the robust setting for real checkpoints
        for skew in (3.0, 0.0):
            be.two_stage_speculation = True
            be.pair_metric = skewed_metric(be, skew) if skew else None
            n, dt = timed(be, 3, 1)
            ev = be.stats.get("speculation_evaluated", 0) / 3
            dr = be.stats.get("speculation_dropped", 0) / 3
            out.append({"name": "cfg2, two-stage speculation, " + ("skewed metric (x exp(3 x position))" if skew else "the benchmark's own metric"),
                        "value": n / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "frontier_rounds": be.stats.get("frontier_rounds", 0) / 3,
                        "speculation_hit_rate": (ev - dr) / ev if ev else None})
    except Exception as exc:
        out.append({"name": "cfg2 under a skewed metric", "error": repr(exc)})
    try:    # opt-in engine feature, NOT the metric: the reference performs these forwards, so the headline does too
        be = BlendingEngine(pipe, do_compile=not args.no_graphs, frontier_width=args.frontier, verbose=False)
        be.elide_dead_steps = True
        be.host_frames = True
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
        be.set_branching(nmb_max_branches=branches)
        before = pipe.stats["unet_samples"]
        n, dt = timed(be, 3, 1)
        out.append({"name": "cfg2 with elide_dead_steps=True (SURVEY C15: the mid branches' step at idx_injection is overwritten by the next "
                            "step's crossfeed, coefficient 1.0; bit-identical frames, tests/test_native_gpu.py)", "value": n / dt, "unit": "frames/s",
                    "ms_per_step": dt * 1e3, "unet_samples_per_transition": (pipe.stats["unet_samples"] - before) / 4})
    except Exception as exc:
        out.append({"name": "cfg2 with elide_dead_steps", "error": repr(exc)})
    try:    # a chain of transitions (example_multi_trans.py:39-58) at the metric's settings, the sequential
        # loop vs all key frames denoised ahead of the transitions (replay.run_multi_transition(pipeline_keyframes=True))
        from latentblending_amd import replay
        prompts = ["photo of underwater landscape, fish, und the sea, incredible detail, high resolution",
                   "rendering of an alien planet, strange plants, strange creatures, surreal",
                   "photo of a forest in the fog, sun rays", "aerial photo of a city at night", "macro photo of a dragonfly"]
        seeds = [420, 421, 422, 423, 424]
        chain = {}
        for piped in (False, True, False, True):
            be = BlendingEngine(pipe, do_compile=not args.no_graphs, frontier_width=args.frontier, verbose=False)
            be.host_frames = True
            be.set_branching(nmb_max_branches=branches)
            if not chain:
                replay.run_multi_transition(be, prompts[:3], seeds[:3], None, pipeline_keyframes=True)     # warm-up: programs of every batch size
                replay.run_multi_transition(be, prompts[:3], seeds[:3], None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            segs = replay.run_multi_transition(be, prompts, seeds, None, pipeline_keyframes=piped)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            chain.setdefault(piped, []).append((sum(len(s_) for s_ in segs), dt))
        best = {k: min(v, key=lambda fd: fd[1]) for k, v in chain.items()}
        out.append({"name": "chain of 4 transitions (5 prompts) at the cfg2 settings: sequential loop (recycle_img1) vs key frames pipelined "
                            "(replay.run_multi_transition(pipeline_keyframes=True))", "unit": "frames/s",
                    "sequential": best[False][0] / best[False][1], "value": best[True][0] / best[True][1],
                    "ms_per_transition_sequential": best[False][1] * 250.0, "ms_per_transition_pipelined": best[True][1] * 250.0,
                    "frames": best[True][0]})
    except Exception as exc:
        out.append({"name": "chain of transitions, pipelined key frames", "error": repr(exc)})
    try:
        base_pipe = N.NativeSDXLPipe(turbo=False, unet_native=pipe.unet_native, vae_native=pipe.vae_native, device=str(pipe.device),
                                     allow_synthetic=True)
        be = BlendingEngine(base_pipe, do_compile=not args.no_graphs, frontier_width=args.frontier, verbose=False)
        be.host_frames = True
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
        be.set_branching(depth_strength=0.5, nmb_max_branches=15)
        n, dt = timed(be, 1, 1)
        out.append({"name": "cfg3: SDXL base 1024x1024, 30 steps, guidance 4.0 (CFG), depth_strength 0.5, nmb_max_branches 15",
                    "value": n / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "frames": n,
                    "levels": [int(v) for v in be.list_idx_injection], "stems": [int(v) for v in be.list_nmb_stems]})
        # BASELINE configs[4] at FULL width on this one GPU: example_multi_trans.py:17-58 with 6 prompts on the base model (1024^2, 30 steps,
        # depth_strength 0.5, 15 branches per transition, negative prompt, swap_forward + recycle_img1; LPIPS-driven insertion) - the
        # programs of this shape were recorded by the cfg-3 line above; ONE pass, five transitions
        try:
            from latentblending_amd import replay
            be5 = BlendingEngine(base_pipe, do_compile=not args.no_graphs, frontier_width=args.frontier, verbose=False)
            be5.host_frames = True
            be5.set_negative_prompt("blurry, pale, low-res, lofi")
            be5.set_branching(depth_strength=0.5, nmb_max_branches=15)
            prompts6 = ["lake and forest", "alien desolate landscapes", "psychedelic skyscraper city", "a reef at dawn", "fog over a harbour",
                        "desert under two moons"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            segs = replay.run_multi_transition(be5, prompts6, [420, 421, 977, 12, 90001, 5], None)
            for seg in segs:
                materialise_frames(seg)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            nfr = sum(len(s_) for s_ in segs)
            out.append({"name": "cfg5 at full width on ONE GPU: SDXL base 1024x1024, 30 steps, 6-prompt multi-transition (5 transitions, swap_forward + "
                                "recycle_img1), depth_strength 0.5, 15 branches each, negative prompt", "value": nfr / dt, "unit": "frames/s",
                        "seconds": dt, "frames": nfr, "transitions": len(segs), "ms_per_transition": dt * 1e3 / max(len(segs), 1)})
            del be5
        except Exception as exc:
            out.append({"name": "cfg5 at full width", "error": repr(exc)})
        del be, base_pipe
        torch.cuda.empty_cache()
    except Exception as exc:
        out.append({"name": "cfg3", "error": repr(exc)})
    try:    # BASELINE configs[3] at FULL width on this one GPU: SDXL-Turbo 512^2, 4 steps, 64 branches as ONE frontier of 64 (what an
        # 8-GPU farm shards): 2 x B=2 + 2 x B=66 UNet steps, 66 decodes in batches of keyframe_chunk
        be = BlendingEngine(pipe, do_compile=not args.no_graphs, frontier_width=64, verbose=False)
        be.host_frames = True
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
        be.set_branching(nmb_max_branches=64)
        n, dt = timed(be, 2, 1)
        out.append({"name": "cfg4 at full width on ONE GPU: SDXL-Turbo 512x512, 4 steps, nmb_max_branches 64, frontier 64", "value": n / dt,
                    "unit": "frames/s", "ms_per_step": dt * 1e3, "frames": n, "frontier_rounds": be.stats.get("frontier_rounds", 0) / 2})
        del be
        torch.cuda.empty_cache()
    except Exception as exc:
        out.append({"name": "cfg4 at full width", "error": repr(exc)})
    return out


def main():
    # the host layer prints progress lines like the reference does; keep stdout for the ONE JSON line
    respawn_under_torchrun(parse())
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
    if args.rendezvous_only:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group(args.backend)
            world = dist.get_world_size()
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
        assert world == args.gpus, f"--gpus {args.gpus} but {world} rank(s) joined the rendezvous"
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return {"metric": "transition frames/sec, SDXL-Turbo 512x512 4-step 15-branch", "value": None, "unit": "frames/s",
                "n_gpus": world, "steps": 0, "warmup": 0, "rendezvous_only": True} if rank == 0 else None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()               # the ranks RCCL actually sees
    assert world == args.gpus, (f"--gpus {args.gpus} but the job has {world} rank(s): start it as `python bench.py --gpus N` "
                                f"(re-executes itself under torch.distributed.run) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")

    import latentblending_amd.native as N
    from latentblending_amd import BlendingEngine

    # seeded synthetic SDXL weights from the product's own provider; rank 0 keeps the fp32 copies so
    # that the cpu_baseline leg can time the oracle on exactly the same parameters
    base = args.config == "cfg3"
    want_cpu = rank == 0 and world == 1 and (args.torch_baseline or not args.no_cpu_baseline) and not base
    t0 = time.perf_counter()
    # LB_SYNTH_CACHE=<dir>: keep the generated (seeded, fp16-rounded) tensors in a scratch file between the several
    # processes of a profiling session - the same values, a few seconds instead of a minute of CPU random draws
    cdir = os.environ.get("LB_SYNTH_CACHE")
    cfile = (lambda s: os.path.join(cdir, f"lb_synth_seed{s}.pt")) if cdir else (lambda s: None)
    unet_prov = N.SyntheticProvider(0, keep=want_cpu, cache_file=cfile(0))
    vae_prov = N.SyntheticProvider(1, keep=want_cpu, cache_file=cfile(1))
    pipe = N.NativeSDXLPipe(turbo=not base, unet_provider=unet_prov, vae_provider=vae_prov, device=f"cuda:{local_rank}",
                            allow_synthetic=True)      # ("data": "synthetic" in the JSON line)
    if rank == 0:
        unet_prov.save_cache(); vae_prov.save_cache()
    t_weights = time.perf_counter() - t0
    unet_w, vae_w = unet_prov.state, vae_prov.state
    if args.torch_baseline:
        assert world == 1 and not base, "--torch-baseline: one GPU, cfg 2 model sizes"
        return {"metric": "diagnostic: vendor-stack (torch eager) time of the UNet step / VAE decode on the same GPU, beside this repo's programs",
                "value": None, "unit": "ms", "n_gpus": 1, "same_gpu_torch_baseline": same_gpu_torch_baseline(pipe, unet_w, vae_w, f"cuda:{local_rank}")}
    farm = None
    if world > 1:
        from latentblending_amd.dist import BranchFarm
        farm = BranchFarm(device=torch.device("cuda", local_rank))
    if args.branches_total > 0:
        branches, scaling = args.branches_total, "strong"
    elif args.scaling == "weak":
        branches, scaling = args.branches * world, "weak"
    else:
        branches, scaling = args.branches, "strong"
    be = BlendingEngine(pipe, do_compile=not args.no_graphs, frontier_width=args.frontier * (world if scaling == "weak" else 1),
                        verbose=False, farm=farm)
    be.host_frames = not args.no_materialise   # the metric's frames are host PIL images, as the reference returns them
    if args.metric_skew:
        be.pair_metric = skewed_metric(be, args.metric_skew)
    be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
    be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
    if base:
        be.set_branching(depth_strength=0.5, nmb_max_branches=branches)
    else:
        be.set_branching(nmb_max_branches=branches)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    frames = 0
    cold_start_s = None
    for w in range(args.warmup):
        tw = time.perf_counter()
        frames = len(be.run_transition(fixed_seeds=[420, 421]))
        if w == 0:          # the first transition records the launch programs, captures their graphs and runs them once
            torch.cuda.synchronize()
            cold_start_s = time.perf_counter() - tw
    for k in pipe.stats:
        pipe.stats[k] = 0
    be.stats.clear()
    farm_counts = {}
    if world > 1 and rank == 0 and not args.no_roofline and args.warmup > 0:
        # a farmed transition cannot be repeated by rank 0 alone (collectives), so rank 0 counts its own program
        # launches during the timed transitions (programs exist after the warm-up) and profiles them afterwards
        install_launch_counters(pipe, farm_counts)
    from latentblending_amd.native.frames import materialise_frames
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        imgs = be.run_transition(fixed_seeds=[420, 421])
        if not args.no_materialise:          # the metric's frames are host PIL images, as the reference returns them
            materialise_frames(imgs)
        frames = len(imgs)
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if world > 1:
        mine = torch.tensor([dt / max(args.steps, 1) * 1e3], device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(v.item()) for v in allr]
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_runs = max(args.steps, 1)
    per_transition = {k: v / n_runs for k, v in pipe.stats.items()}
    per_transition["frontier_rounds"] = be.stats.get("frontier_rounds", 0) / n_runs
    per_transition["speculation_dropped"] = be.stats.get("speculation_dropped", 0) / n_runs
    evaluated = be.stats.get("speculation_evaluated", 0) / n_runs
    per_transition["speculation_evaluated"] = evaluated
    per_transition["speculation_hit_rate"] = (evaluated - per_transition["speculation_dropped"]) / evaluated if evaluated else None

    out = {
        "metric": "transition frames/sec, SDXL-Turbo 512x512 4-step 15-branch" if not base else
                  "transition frames/sec, SDXL base 1024x1024 30-step guidance 4.0 nmb_max_branches=15 (BASELINE configs[2], secondary)",
        "value": frames * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": ("SDXL base 1024x1024, num_inference_steps=30, guidance_scale=4.0 (mid-damped), depth_strength=0.5, "
                                "nmb_max_branches=%d (%d frames/transition, levels %s x stems %s), fp16, fixed_seeds=[420,421]"
                                % (branches, frames, list(be.list_idx_injection), list(be.list_nmb_stems))) if base else
                               "SDXL-Turbo 512x512, num_inference_steps=4, nmb_max_branches=%d (%d frames/transition), "
                               "fp16, fixed_seeds=[420,421], anchors not recycled" % (branches, frames),
                   "frontier_width": args.frontier, "hipgraphs": not args.no_graphs, "metric_skew": args.metric_skew,
                   "frames_materialised_in_timed_region": not args.no_materialise,
                   "parallelism": ("branch farm over %d ranks (RCCL: one packed all-gather of branches + one of LPIPS "
                                   "scalars per round; anchors computed by every rank)" % world) if world > 1 else "single GPU",
                   "farm": None if farm is None else {
                       "collectives_per_transition": farm.collectives / max(args.steps + args.warmup, 1),
                       "bytes_moved_per_transition": farm.bytes_moved / max(args.steps + args.warmup, 1),
                       "ms_per_transition_by_rank": per_rank_ms,
                       # LB_FARM_TRACE=1: mean pack / collective / unpack milliseconds per farm call on rank 0 (the device is drained at
                       # every boundary while tracing, so a traced run is slower than an untraced one: read the split, not the total)
                       "trace_rank0": farm.trace_summary() if farm.trace_on else "set LB_FARM_TRACE=1 for the pack / collective / unpack split",
                       "scale_workloads": {"cfg2 (default)": "--gpus N: 15 mid branches over N ranks (strong scaling; the metric's workload)",
                                           "cfg4 (BASELINE configs[3], the config STATED for 8 GPUs)": "--gpus N --branches-total 64 --frontier 64: 64 mid "
                                           "branches in one round, 8 per rank at N = 8 (B = 2 + 8 batches: the anchors' two small steps are 1/5 "
                                           "of a rank's UNet time instead of 2/3 as in cfg 2 at N = 8)"},
                       "expected_strong_scaling": "Amdahl-limited by design: the two B=2 anchor steps (2 x 10.8 ms of a 135 ms transition on one GPU, round 6) "
                                                  "and the anchors' share of the B = 2 + G/N batches do not shard - about 2.3-2.5x at 8 GPUs for cfg 2 "
                                                  "(DESIGN.md section 5); --scaling weak (15 N branches) is the sharded regime"},
                   "census_per_transition": per_transition, "weights_gen_s": round(t_weights, 1),
                   "cold_start_s": None if cold_start_s is None else round(cold_start_s, 2)},
    }
    if rank == 0 and world == 1 and not args.no_roofline:
        try:
            out["phases_per_transition"] = phase_split(pipe, be)
        except Exception as exc:
            out["phases_error"] = repr(exc)
    if rank == 0 and world == 1 and not args.no_roofline:      # (needs a solo transition: no collectives)
        # launches of every (program, batch) per transition: one more transition with counting wrappers
        step_launch_counts = {}
        install_launch_counters(pipe, step_launch_counts)
        be.run_transition(fixed_seeds=[420, 421])
        torch.cuda.synchronize()
        out["roofline"], out["rooflines"] = roofline_blocks(gemm_family_profile(pipe, step_launch_counts), step_launch_counts,
                                                            f"cuda:{local_rank}")
    if rank == 0 and world == 1 and not base and not args.no_secondary and not args.metric_skew:
        out["secondary"] = secondary_lines(pipe, args, branches)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not base:
        out["cpu_baseline"] = cpu_baseline(unet_w, vae_w, per_transition)
    if world > 1:
        dist.destroy_process_group()
        if rank == 0 and farm_counts:
            # rank 0's share of the farmed transition (its UNet / VAE batches), same eager hipEvent replay as N=1
            try:
                per_tr = {k: v / max(args.steps, 1) for k, v in farm_counts.items()}
                out["roofline"], out["rooflines"] = roofline_blocks(gemm_family_profile(pipe, per_tr), per_tr, f"cuda:{local_rank}")
                out["roofline"]["scope"] = "rank 0's programs of the farmed transition"
            except Exception as exc:                      # never lose the throughput line over the profile
                out["roofline"] = None
                out["roofline_error"] = repr(exc)
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
