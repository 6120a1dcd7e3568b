"""Per-op device-time breakdown of the native launch programs (hipEvents between ops, eager replay).
Usage (GPU box): python tools/profile_programs.py [B ...]   -> gpurun_out/program_profile_*.txt"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N


def table(prog, glog, alog, title, fh):
    prog.time_ops()
    ms = prog.time_ops()
    names = prog.op_names()
    rows, gi, ai = [], 0, 0
    by_kind = defaultdict(float)
    for n, t in zip(names, ms):
        if n in ("lb_gemm_f16", "lb_conv3x3_halo_f16", "lb_upconv2x_halo_f16", "lb_conv3x3_narrow_f16"):
            g = glog[gi]; gi += 1
            halo = n != "lb_gemm_f16"
            rows.append((t, f"gemm{'(halo conv)' if halo else '(conv)' if g['conv'] else ''} M={g['M']} N={g['N']} K={g['K']}", g["flops"]))
            by_kind["conv_halo" if halo else "gemm_conv" if g["conv"] else "gemm"] += t
        elif n in ("lb_attn_fwd_d64", "lb_attn_fwd_d512"):
            a = alog[ai]; ai += 1
            rows.append((t, "attn", a["flops"]))
            by_kind["attn"] += t
        else:
            rows.append((t, n, 0.0))
            by_kind[n] += t
    total = sum(ms)
    print(f"== {title}: {len(ms)} ops, {total:.3f} ms eager (event-to-event)", file=fh)
    for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1]):
        print(f"   {k:28s} {v:9.3f} ms  {100 * v / total:5.1f} %", file=fh)
    agg = defaultdict(lambda: [0.0, 0, 0.0])
    for t, d, f in rows:
        agg[d][0] += t; agg[d][1] += 1; agg[d][2] += f
    print("   -- top shapes --", file=fh)
    for d, (t, c, f) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
        tf = f / (t * 1e-3) / 1e12 if t > 0 and f > 0 else 0
        print(f"   {t:9.3f} ms x{c:4d} avg {1e3 * t / c:8.1f} us  {tf:7.1f} TF/s  {d}", file=fh)
    return total


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [1, 2, 8]
    os.makedirs("gpurun_out", exist_ok=True)
    pipe = N.NativeSDXLPipe(turbo=True, allow_synthetic=True)
    with open("gpurun_out/program_profile.txt", "w") as fh:
        for B in Bs:
            up = pipe.unet_program(B, 64)
            ctx = torch.randn(B, 77, 2048, device="cuda").half()
            up.set_conditioning(ctx, torch.randn(B, 1280, device="cuda").half(), torch.tensor([[512.0, 512, 0, 0, 512, 512]] * B, device="cuda"))
            up.forward(torch.randn(B, 4, 64, 64, device="cuda").half(), torch.full((B,), 499.0))
            nc = sum(1 for n in up.prog_cond.op_names() if n in ("lb_gemm_f16", "lb_conv3x3_halo_f16", "lb_upconv2x_halo_f16", "lb_conv3x3_narrow_f16"))
            table(up.prog_step, up.em.gemm_log[nc:], up.em.attn_log, f"UNet step program B={B} L=64", fh)
            table(up.prog_cond, up.em.gemm_log[:nc], [], f"UNet conditioning program B={B}", fh)
            up.enable_graphs()
            torch.cuda.synchronize()
            for _ in range(3):
                up.prog_step.launch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                up.prog_step.launch()
            torch.cuda.synchronize()
            print(f"   hipGraph replay: {(time.perf_counter() - t0) * 100:.3f} ms per forward (B={B})", file=fh)
            vp = pipe.vae_program(B, 64)
            vp.decode(torch.randn(B, 4, 64, 64, device="cuda").half())
            table(vp.prog, vp.em.gemm_log, vp.em.attn_log, f"VAE decode program B={B} L=64", fh)
            vp.prog.instantiate()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                vp.prog.launch()
            torch.cuda.synchronize()
            print(f"   hipGraph replay: {(time.perf_counter() - t0) * 200:.3f} ms per decode batch (B={B})", file=fh)
    print(open("gpurun_out/program_profile.txt").read())


if __name__ == "__main__":
    main()
