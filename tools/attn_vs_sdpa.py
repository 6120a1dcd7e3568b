"""The attention calls of the SDXL UNet / VAE programs: the hand-written kernels (lb_attn_fwd_d64 / _d512) against the VENDOR's fused
attention on the same operands - torch.nn.functional.scaled_dot_product_attention on ROCm (its flash / memory-efficient backends: the
kernels the reference's diffusers AttnProcessor2_0 would run on this GPU).  The attention counterpart of tools/gemm_bench.cpp's rocBLAS
rows and tools/conv_vs_miopen.py: a diagnostic, never linked into the product.  Ours reads q | k | v as column slices of the fused
projection buffer ([B*S][3C], token-major) and writes [B*S][C]; SDPA gets the layout IT prefers for each backend (both [B, H, S, d]
contiguous and the strided view of the same token-major buffer are timed, the faster one counts).  hipEvents, median of 5 rounds of
20 launches.  Usage: python tools/attn_vs_sdpa.py > gpurun_out/attn_vs_sdpa.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from latentblending_amd.hip import ops as o

DEV = "cuda:0"
# (B, heads, Sq, Skv, d, where)
SHAPES = [(17, 20, 256, 256, 64, "UNet self 16^2 (60 / step)"), (17, 10, 1024, 1024, 64, "UNet self 32^2 (10)"),
          (17, 20, 256, 77, 64, "UNet cross 16^2 (60)"), (17, 10, 1024, 77, 64, "UNet cross 32^2 (10)"),
          (2, 20, 256, 256, 64, "UNet self 16^2, B = 2"), (2, 10, 1024, 1024, 64, "UNet self 32^2, B = 2"), (2, 20, 256, 77, 64, "UNet cross 16^2, B = 2"),
          (17, 1, 4096, 4096, 512, "VAE mid block 64^2 (1 / decode)")]


def timed(fn, n=20, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / n * 1e3)
    return sorted(out)[len(out) // 2]


def main():
    print("# fused attention, fp16: hand-written kernel vs torch SDPA (ROCm flash / mem-efficient backends); us per launch; TF/s = 4 B H Sq Skv d / time")
    for (B, H, Sq, Skv, d, what) in SHAPES:
        g = torch.Generator().manual_seed(Sq + Skv)
        C = H * d
        Skp = (Skv + 7) // 8 * 8                         # context rows are padded 77 -> 80 in the programs
        q = torch.randn(B * Sq, C, generator=g).half().to(DEV)
        k = torch.randn(B * Skp, C, generator=g).half().to(DEV)
        v = torch.randn(B * Skp, C, generator=g).half().to(DEV)
        flops = 4.0 * B * H * Sq * Skv * d
        if d == 64:
            ours = lambda out=None: o.attention_d64(q, k, v, B, H, Sq, Skp, skv_valid=Skv, out=out)
        else:
            ours = lambda out=None: o.attention_d512(q, k, v, B, H, Sq, Skp, skv_valid=Skv, out=out)
        got = ours()
        # SDPA operands: the strided [B, H, S, d] views of the token-major buffers, and contiguous copies
        qv = q.view(B, Sq, H, d).transpose(1, 2)
        kv = k.view(B, Skp, H, d)[:, :Skv].transpose(1, 2)
        vv = v.view(B, Skp, H, d)[:, :Skv].transpose(1, 2)
        qc, kc, vc = qv.contiguous(), kv.contiguous(), vv.contiguous()
        ref = F.scaled_dot_product_attention(qc, kc, vc)
        err = float((got.view(B, Sq, H, d).transpose(1, 2).float() - ref.float()).norm() / ref.float().norm())
        t_ours = timed(lambda: ours(got))
        t_view = timed(lambda: F.scaled_dot_product_attention(qv, kv, vv))
        t_cont = timed(lambda: F.scaled_dot_product_attention(qc, kc, vc))
        t_lib = min(t_view, t_cont)
        print(f"B={B:2d} H={H:2d} Sq={Sq:4d} Skv={Skv:4d} d={d:3d}  [{what:32s}]  ours {t_ours:8.1f} us ({flops / t_ours / 1e6:5.0f} TF/s)   "
              f"SDPA {t_lib:8.1f} us ({flops / t_lib / 1e6:5.0f} TF/s; strided {t_view:.1f}, contiguous {t_cont:.1f})   ours / SDPA {t_ours / t_lib:5.2f}   "
              f"rel-L2 {err:.1e}", flush=True)
        del q, k, v, qc, kc, vc, ref, got
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
