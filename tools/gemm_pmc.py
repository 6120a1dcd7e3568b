"""Launch the GEMM kernel on a few shapes (eager, one launch each) for `rocprofv3 --pmc ... --kernel-trace`;
`--summarise <dir>` folds the counter CSVs per (kernel, grid)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import ctypes as C
    import torch
    from latentblending_amd.hip import lib
    DEV = "cuda"
    zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
    for (M, N, K, tile) in [(4352, 1280, 1280, 4), (4352, 1280, 5120, 4), (4352, 3840, 1280, 5), (4096, 4096, 4096, 4), (4096, 4096, 4096, 5),
                            (8192, 8192, 8192, 5)]:
        A = torch.randn(M, K, device=DEV).half()
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
        out = torch.empty(M, N, device=DEV, dtype=torch.float16)
        p = lib.LbGemmParams()
        p.A, p.W, p.C, p.lda = A.data_ptr(), W.data_ptr(), out.data_ptr(), K
        p.M, p.N, p.K, p.ldw, p.ldc = M, N, K, K, N
        p.zero_page = zp.data_ptr()
        lib.api.lb_gemm_set_tuning(tile, 0)
        for _ in range(3):
            lib.api.lb_gemm_f16(C.byref(p), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        lib.api.lb_gemm_set_tuning(0, 0)


def summarise(root):
    rows = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "gemm_f16" not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"])
                rows.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "gemm_f16" in r["Kernel_Name"]:
                    key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"])
                    dur.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for (name, grid), ctr in sorted(rows.items()):
        d = {c: sum(v) / len(v) for c, v in sorted(ctr.items())}
        if (name, grid) in dur:
            d["duration_us"] = min(dur[(name, grid)])
        out[f"{name} grid={grid}"] = d
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    if "--summarise" in sys.argv:
        root = sys.argv[sys.argv.index("--summarise") + 1]
        json.dump(summarise(root), open(os.path.join(root, "summary.json"), "w"), indent=1)
    else:
        run()
