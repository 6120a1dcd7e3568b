"""Sweep (tile, ring depth, split-K) of the GEMM kernel over the shapes the SDXL UNet / VAE really
launch, timing each variant as a hipGraph of REP back-to-back launches (no host launch cost)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from latentblending_amd.native.runtime import Program

DEV, REP = "cuda", 20
HALO = "--halo" in sys.argv         # also time the experimental halo-tile 3x3 kernel on the conv shapes
BIG_ONLY = "--big" in sys.argv      # only the big-M shapes, direct-to-LDS variants (256x128 tile study)


def time_variant(p, tile, depth, splitk, glds_stages=0, halo=False):
    lib.api.lb_gemm_set_halo(2 if halo else 0)     # 3x3 convs through csrc/conv3_halo.hip, or never
    lib.api.lb_gemm_set_tuning(tile, splitk)
    lib.api.lb_gemm_set_depth(depth)
    if tile == 0 and depth == 0 and not glds_stages:
        lib.api.lb_gemm_set_variant(-1, 0)          # the library's own choice
    else:
        lib.api.lb_gemm_set_variant(1 if glds_stages else 0, glds_stages)
    prog = Program("sweep")
    try:
        with prog.record():
            for _ in range(REP):
                lib.api.lb_gemm_f16(C.byref(p), 0)
    finally:
        lib.api.lb_gemm_set_tuning(0, 0)
        lib.api.lb_gemm_set_depth(0)
        lib.api.lb_gemm_set_variant(-1, 0)
        lib.api.lb_gemm_set_halo(1)
    prog.instantiate()
    st = torch.cuda.current_stream().cuda_stream
    prog.launch(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        prog.launch(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * REP) * 1e3      # us per launch


def main():
    shapes = []
    for B in (2, 17):
        M3, M2 = 256 * B, 1024 * B
        shapes += [("lin", M3, 1280, 1280), ("lin", M3, 2560, 1280), ("lin", M3, 1280, 5120), ("geglu", M3, 10240, 1280),
                   ("lin", 1280, M3, 1280), ("lin", M2, 640, 640), ("geglu", M2, 5120, 640), ("lin", M2, 640, 2560),
                   ("conv", (B, 16, 1280, 1280)), ("conv", (B, 32, 640, 640)), ("conv", (B, 64, 320, 320)),
                   ("conv", (B, 16, 2560, 1280)), ("conv", (B, 32, 1280, 640))]
    shapes += [("conv", (1, 64, 512, 512)), ("conv", (1, 128, 512, 512)), ("conv", (1, 256, 256, 256)), ("conv", (1, 512, 128, 128)),
               ("conv", (8, 512, 128, 128)), ("lin", 4096, 4096, 4096), ("lin", 8192, 8192, 8192)]
    if "--small" in sys.argv:
        shapes = [sh for sh in shapes if (sh[0] == "conv" and sh[1][0] == 2 and sh[1][1] <= 32) or (sh[0] == "lin" and sh[1] <= 512)]
    if BIG_ONLY:
        shapes = [sh for sh in shapes if (sh[0] == "conv" and sh[1][0] * sh[1][1] ** 2 >= 8192) or (sh[0] != "conv" and sh[1] >= 3840)]
    results = []
    for sh in shapes:
        p = lib.LbGemmParams()
        if sh[0] == "conv":
            B, H, C1, C2 = sh[1]
            x = torch.randn(B, H, H, C1, device=DEV).half()
            w = (torch.randn(C2, 9 * C1, device=DEV) * (9 * C1) ** -0.5).half()
            out = torch.empty(B, H, H, C2, device=DEV, dtype=torch.float16)
            M, N, K = B * H * H, C2, 9 * C1
            p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = 1, H, H, C1, H, H, 3, 3, 1, 1, 0, C1
            p.A, p.W, p.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
            tag = f"conv B{B} {H}x{H} {C1}->{C2}"
        else:
            _, M, N, K = sh
            A = torch.randn(M, K, device=DEV).half()
            W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
            geglu = sh[0] == "geglu"
            out = torch.empty(M, N // 2 if geglu else N, device=DEV, dtype=torch.float16)
            p.A, p.W, p.C, p.lda = A.data_ptr(), W.data_ptr(), out.data_ptr(), K
            p.flags = lib.GEMM_GEGLU if geglu else 0
            tag = f"{sh[0]} M{M} N{N} K{K}"
        p.M, p.N, p.K, p.ldw, p.ldc = M, N, K, K, out.shape[-1]
        ws = torch.empty(min(lib.api.lb_gemm_workspace_bytes(M, N) // 4, 1 << 28), dtype=torch.float32, device=DEV)
        flops = 2.0 * M * N * K
        small = ((M + 63) // 64) * ((N + 63) // 64) <= 640
        best = None
        row = {"shape": tag, "variants": {}}
        zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
        p.zero_page = zp.data_ptr()
        for tile in (1, 2, 3, 4, 5):
            for mode, val in [("d", 1), ("d", 3), ("d", 4), ("g", 2), ("g", 3), ("g", 4)]:
                if tile == 5 and not (mode == "g" and val == 2):
                    continue
                if tile == 4 and (mode == "d" or val == 4):     # 256x128 lives in the direct-to-LDS family, 2-3 stages
                    continue
                if BIG_ONLY and mode == "d":
                    continue
                for sk in ([0] if not small or sh[0] == "geglu" else [1, 0]):
                    p.partial = ws.data_ptr() if (small and sh[0] != "geglu" and sk != 1) else None
                    us = time_variant(p, tile, val if mode == "d" else 0, 0, val if mode == "g" else 0)
                    key = f"t{tile}{mode}{val}k{sk}"
                    row["variants"][key] = us
                    if best is None or us < best[1]:
                        best = (key, us)
        lib.api.lb_gemm_set_depth(0)
        p.partial = ws.data_ptr() if small and sh[0] != "geglu" else None
        auto = time_variant(p, 0, 0, 0)
        if HALO and sh[0] == "conv":
            row["variants"]["halo"] = time_variant(p, 0, 0, 0, halo=True)
            if row["variants"]["halo"] < best[1]:
                best = ("halo", row["variants"]["halo"])
        row.update(best=best[0], best_us=best[1], best_TF=flops / best[1] / 1e6, auto_us=auto, auto_TF=flops / auto / 1e6)
        results.append(row)
        top = sorted(row["variants"].items(), key=lambda kv: kv[1])[:4]
        print(f"{tag:34s} auto {auto:8.1f} us {row['auto_TF']:6.0f} TF | best {best[0]} {best[1]:8.1f} us {row['best_TF']:6.0f} TF | "
              + " ".join(f"{k}:{v:.0f}" for k, v in top), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(results, open("gpurun_out/sweep_gemm.json", "w"), indent=1)


if __name__ == "__main__":
    main()
