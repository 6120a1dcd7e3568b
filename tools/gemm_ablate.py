"""Ablation study of the direct-to-LDS GEMM: where does a K-tile's time go?

Builds three extra copies of liblbhip.so whose gemm_glds.hip is compiled with -DLB_ABLATE=n
(4: -DLB_BURST=1, the older burst-issue loop; 1: no global->LDS requests, 2: requests only (no LDS reads / MFMAs), 3: requests + MFMAs on
register operands (no LDS reads)) and times the same launches with each (results are garbage by
construction; only durations matter).

    python tools/gemm_ablate.py build        # here (hipcc cross-compiles), before gpurun
    python tools/gemm_ablate.py run          # on the GPU box
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "latentblending_amd", "csrc")
OUT = os.path.join(ROOT, "latentblending_amd", "hip", "ablate")


def build():
    from latentblending_amd.csrc import build as B
    B.build_library(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CSRC, "build", s.replace(".hip", ".o")) for s in B.sources() if s != "gemm_glds.hip"]
    for n in (1, 2, 3, 4):
        obj = os.path.join(OUT, f"gemm_glds_ab{n}.o")
        flag = f"-DLB_ABLATE={n}" if n < 4 else "-DLB_BURST=1"
        subprocess.check_call([B._hipcc(), *B.FLAGS, "-w", flag, "-c", os.path.join(CSRC, "gemm_glds.hip"), "-o", obj])
        subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, obj, "-o",
                               os.path.join(OUT, f"liblbhip_ab{n}.so")])
        os.remove(obj)
        print("built", n)


def worker():
    import torch
    from latentblending_amd.hip import lib
    from tools.sweep_gemm import time_variant
    out = {}
    for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (4352, 1280, 5120), (4352, 2560, 1280)]:
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        o = torch.empty(M, N, device="cuda", dtype=torch.float16)
        zp = torch.zeros(64, dtype=torch.uint8, device="cuda")
        p = lib.LbGemmParams()
        p.A, p.W, p.C, p.lda, p.ldw, p.ldc, p.M, p.N, p.K = A.data_ptr(), W.data_ptr(), o.data_ptr(), K, K, N, M, N, K
        p.zero_page = zp.data_ptr()
        for tile, st in [(1, 2), (1, 3), (1, 4), (4, 2), (4, 3), (2, 3)]:
            out[f"M{M}N{N}K{K} t{tile}s{st}"] = time_variant(p, tile, 0, 0, st)
    print("ABLATE_JSON " + json.dumps(out))


def run():
    rows = {}
    for n in (0, 4, 1, 2, 3):
        env = dict(os.environ)
        if n:
            env["LB_HIP_LIBRARY"] = os.path.join(OUT, f"liblbhip_ab{n}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("ABLATE_JSON ")]
        if not line:
            print("ablate", n, "failed:", r.stderr[-2000:])
            continue
        rows[n] = json.loads(line[0][len("ABLATE_JSON "):])
    names = {0: "full", 4: "full(burst issue)", 1: "no-gload", 2: "gload-only", 3: "gload+mfma(reg)"}
    keys = list(rows[0])
    print(f"{'shape / tile':34s}" + "".join(f"{names[n]:>18s}" for n in rows))
    for k in keys:
        print(f"{k:34s}" + "".join(f"{rows[n][k]:15.1f} us" for n in rows))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "gemm_ablate.json"), "w"), indent=1)


if __name__ == "__main__":
    {"build": build, "worker": worker, "run": run}[sys.argv[1]]()
