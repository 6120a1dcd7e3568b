"""rocprof-reported HBM GB/s of the latent-mixing / scheduler kernels on >= 1 GiB batches (north_star: "rocprof-reported HBM
GB/s for slerp / crossfeed").

  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_mixing -- python tools/mixing_rocprof.py run
  python tools/mixing_rocprof.py fold gpurun_out/r03_mixing profiles/r03_mixing_rocprof.json

`run` launches each kernel ITER times on a batch of `pairs` latents of 4 x 64 x 64 fp16 (the benchmark's latent size) and
writes the algorithmic byte counts next to the trace; `fold` divides them by the kernels' average durations in rocprofv3's
kernel_stats.csv (6 B / element for slerp: read p0, p1, write out; 8 B / element for the Euler-ancestral step; 4 B for scale)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ITER = 10


def run():
    import torch
    from latentblending_amd.hip import ops
    from latentblending_amd.hip.lib import api
    dev = "cuda"
    n = 4 * 64 * 64
    pairs = (1 << 30) // (n * 2 * 3)
    p0, p1 = torch.randn(pairs, n, device=dev).half(), torch.randn(pairs, n, device=dev).half()
    fr = torch.rand(pairs, device=dev, dtype=torch.float64)
    out = torch.empty_like(p0)
    params = torch.zeros(pairs, 8, dtype=torch.float32, device=dev)
    params[:, 0], params[:, 1], params[:, 2], params[:, 4] = 1.6129, 0.6374, 0.6259, -0.9755
    st = torch.cuda.current_stream().cuda_stream
    a1, b1 = torch.randn(1, n, device=dev).half(), torch.randn(1, n, device=dev).half()
    fr15 = torch.rand(15, device=dev, dtype=torch.float64)
    for _ in range(ITER):
        ops.slerp_strided(p0, p1, fr, n, out=out)                                           # crossfeed form: G pairs
        ops.slerp_strided(a1, b1, fr15, n, broadcast0=True, broadcast1=True)                 # parental mix at native size (launch-bound)
        api.lb_scale_model_input_f16(p0.data_ptr(), out.data_ptr(), params.data_ptr(), n, pairs, 0, st)
        api.lb_euler_step_f16(p0.data_ptr(), p1.data_ptr(), out.data_ptr(), out.data_ptr(), params.data_ptr(), n, pairs, 0, 1, st)
    torch.cuda.synchronize()
    meta = {"pairs": pairs, "elements_per_pair": n, "iters": ITER,
            "bytes": {"slerp_strided_kernel": pairs * n * 6, "scale_input_kernel": pairs * n * 4, "euler_step_kernel": pairs * n * 8}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(meta, open(os.path.join(ROOT, "gpurun_out", "mixing_rocprof_meta.json"), "w"))


def fold(src, dst):
    meta = json.load(open(os.path.join(ROOT, "gpurun_out", "mixing_rocprof_meta.json")))
    files = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    rows = list(csv.DictReader(open(files[0])))
    out = {"command": "rocprofv3 --kernel-trace --stats -- python tools/mixing_rocprof.py run", "batch": meta, "kernels": []}
    for key, nbytes in meta["bytes"].items():
        for r in rows:
            if key in r["Name"]:
                # (the native-size parental-mix launches share the slerp kernel's name with another VPT: keep the big ones apart)
                avg_ns, mn, calls = float(r["AverageNs"]), float(r["MinNs"]), int(r["Calls"])
                note = "GB/s from the average duration"
                if "slerp" in key:      # the row mixes ITER 1-GiB launches with ITER native-size ones (~MinNs each): take the big ones' average
                    avg_ns = (avg_ns * calls - mn * (calls - meta["iters"])) / meta["iters"]
                    note = "GB/s from the average duration of the 1-GiB launches (the row's total minus the native-size launches at MinNs)"
                gbs = nbytes / avg_ns if avg_ns else 0.0
                out["kernels"].append({"name": r["Name"][:120], "calls": calls, "avg_us_of_the_big_launches": avg_ns / 1e3, "min_us": mn / 1e3,
                                       "algorithmic_bytes": nbytes, "GB_per_s": gbs, "frac_of_8TBs": gbs / 8000.0, "note": note})
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        fold(sys.argv[2], sys.argv[3])
