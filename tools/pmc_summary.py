"""Summarise rocprofv3 outputs of bench.py into profiles/: per-kernel-family time (kernel-trace stats) and
HBM traffic of the GEMM family from the two PMC passes (FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md §HBM,
WRITE_SIZE; both in KiB)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"      # directory holding <tag>_stats/, <tag>_pmc_FETCH_SIZE/, <tag>_pmc_WRITE_SIZE/
bench_args = sys.argv[3] if len(sys.argv) > 3 else "--steps 20 --warmup 5"
commit = sys.argv[4] if len(sys.argv) > 4 else "not recorded"     # the commit the measured tree was built from (the GPU box has no .git)


def family(name):
    if "gemm_f16" in name or "conv3x3_halo" in name or "conv3x3_narrow" in name:
        return "gemm"
    if "splitk_reduce" in name:
        return "gemm_splitk_reduce"
    for k in ("attn_fwd", "layernorm", "gn_apply", "gn_partial", "gn_fold", "cast_f32", "slerp", "lerp", "euler", "lpips", "softmax", "scale_input"):
        if k in name:
            return k
    return "other"


out = {"commit": commit}
stats = glob.glob(f"{src}/{tag}_stats/**/*kernel_stats.csv", recursive=True) or glob.glob(f"{src}/{tag}_stats/*kernel_stats.csv")
if stats:
    fam = collections.defaultdict(lambda: [0.0, 0])
    with open(stats[0]) as fh:
        for row in csv.DictReader(fh):
            f = family(row["Name"])
            fam[f][0] += float(row["TotalDurationNs"])
            fam[f][1] += int(row["Calls"])
    total = sum(v[0] for v in fam.values())
    out["kernel_time_by_family"] = {k: {"total_ms": v[0] / 1e6, "calls": v[1], "avg_us": v[0] / v[1] / 1e3,
                                        "share": v[0] / total} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
    out["kernel_stats_command"] = f"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py {bench_args} --no-cpu-baseline --no-roofline --no-secondary"
pmc = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"{src}/{tag}_pmc_{counter}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    tot, n = 0.0, 0
    with open(files[0]) as fh:
        for row in csv.DictReader(fh):
            if family(row["Kernel_Name"]) == "gemm":
                tot += float(row["Counter_Value"])
                n += 1
    pmc[counter] = {"sum_KiB": tot, "launches": n}
if len(pmc) == 2:
    n = pmc["FETCH_SIZE"]["launches"]
    fetch = pmc["FETCH_SIZE"]["sum_KiB"] * 1024 * 2        # gfx950: FETCH_SIZE counts 128-B requests as 64 B
    write = pmc["WRITE_SIZE"]["sum_KiB"] * 1024
    out["gemm_family_hbm_traffic"] = {"launches": n, "fetch_bytes_corrected": fetch, "write_bytes": write,
                                      "bytes_per_launch": (fetch + write) / n,
                                      "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                                              f"`bench.py {bench_args} --no-graphs --no-cpu-baseline --no-roofline --no-secondary`; FETCH x2 correction "
                                              "(MI355X_MICROARCH.md §HBM); Infinity-Cache hits are counted; GEMM family = "
                                              "gemm_f16_* (ping-pong, direct-to-LDS, register-ring) + conv3x3_halo_kernel + conv3x3_narrow_kernel launches"}
json.dump(out, open(f"profiles/{tag}_rocprof_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
