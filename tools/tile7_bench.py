"""192x128 (6-wave) tile vs the other direct-to-LDS tiles on the plain GEMM shapes of the B = 17 UNet program
(hipGraph-timed, tools/sweep_gemm.time_variant); also correctness of tile 7 against torch on one ragged shape."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from tools.sweep_gemm import time_variant

DEV = "cuda"
rows = []
for (M, N, K) in [(4352, 1280, 1280), (4352, 1280, 5120), (4352, 3840, 1280), (4352, 1280, 2560), (17408, 640, 640),
                  (17408, 640, 2560), (17408, 1920, 640), (17408, 1280, 640), (69632, 320, 640), (2176, 1280, 1280), (8704, 1280, 1280)]:
    A = torch.randn(M, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    p = lib.LbGemmParams()
    p.A, p.W, p.C, p.lda = A.data_ptr(), W.data_ptr(), out.data_ptr(), K
    p.M, p.N, p.K, p.ldw, p.ldc = M, N, K, K, N
    zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
    p.zero_page = zp.data_ptr()
    res = {"shape": f"M{M} N{N} K{K}", "auto": time_variant(p, 0, 0, 0)}
    for tile, st in ((1, 2), (4, 3), (5, 2), (7, 3)):
        res[f"t{tile}"] = time_variant(p, tile, 0, 0, st)
    res["auto2"] = time_variant(p, 0, 0, 0)
    rows.append(res)
    fl = 2.0 * M * N * K
    print(f"{res['shape']:24s} auto {res['auto']:7.1f}/{res['auto2']:7.1f} us | " +
          " ".join(f"t{t}:{res[f't{t}']:7.1f} ({fl / res[f't{t}'] / 1e6:5.0f} TF)" for t in (1, 4, 5, 7)), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/tile7_bench.json", "w"), indent=1)
