"""In-program A/B of launch-time knobs over the full SDXL UNet step program (hipGraph replay, B = 2 and B = 17 at 512^2): the LDS
ring depth of the direct-to-LDS GEMMs (lb_gemm_set_variant(1, stages)) and the 5-stage attention ring (lb_attn_set_tuning(32)).
The isolated sweeps (profiles/r03_small_m_sweep*.txt) ran with Infinity-Cache-warm weights; inside the program every GEMM
streams its weights from HBM (5.1 GB per forward), so the latency picture differs - this tool measures it where it matters.
Knobs are read when a program is RECORDED.  Prepared at the end of round 3, run at the end of round 4 with the ping-pong policy
added as a knob (profiles/r04_unet_knob_ab.txt): one process, same weights, same box - the cleanest in-situ A/B there is.
Usage: LB_SYNTH_CACHE=/tmp python tools/unet_knob_ab.py > gpurun_out/unet_knob_ab.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N
from latentblending_amd.hip import lib

DEV = "cuda:0"
# round 6: the knobs are this round's kernel changes, each switched back to its round-5 form (rounds 3-4 swept ring depths and the
# ping-pong policy: profiles/r04_unet_knob_ab.txt)
KNOBS = [("default", lambda: None),
         ("attention: r1-5 streaming kernel", lambda: lib.api.lb_attn_set_tuning(256)),
         ("layernorm: r1 kernel", lambda: lib.api.lb_layernorm_set_form(0)),
         ("192x128 tile: 6 waves", lambda: lib.api.lb_gemm_set_t192_waves8(0)),
         ("64x64 K-groups off", lambda: lib.api.lb_gemm_set_kgroups(0)),
         ("groupnorm: statistics + apply launches", lambda: lib.api.lb_groupnorm_set_fused(0)),
         ("attention: block order of r1-5", lambda: lib.api.lb_attn_set_tuning(2048)),
         ("all as in round 5", lambda: (lib.api.lb_attn_set_tuning(256 + 2048), lib.api.lb_layernorm_set_form(0), lib.api.lb_gemm_set_t192_waves8(0),
                                        lib.api.lb_gemm_set_kgroups(0), lib.api.lb_groupnorm_set_fused(0)))]


def reset():
    lib.api.lb_gemm_set_variant(-1, 0)
    lib.api.lb_attn_set_tuning(0)
    lib.api.lb_gemm_set_pp_auto(1)
    lib.api.lb_layernorm_set_form(1)
    lib.api.lb_gemm_set_t192_waves8(1)
    lib.api.lb_gemm_set_kgroups(1)
    lib.api.lb_groupnorm_set_fused(1)


def timed(launch, iters):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        launch()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    cdir = os.environ.get("LB_SYNTH_CACHE")
    prov = N.SyntheticProvider(0, cache_file=os.path.join(cdir, "lb_synth_seed0.pt") if cdir else None)
    net = N.NativeUNet(N.UNetConfig(), prov, DEV)
    prov.save_cache()
    for B in (2, 17):
        g = torch.Generator().manual_seed(B)
        ctx, te = torch.randn(B, 77, 2048, generator=g).half().to(DEV), torch.randn(B, 1280, generator=g).half().to(DEV)
        ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B).to(DEV)
        x = torch.randn(B, 4, 64, 64, generator=g).half().to(DEV)
        ref, best = None, {}
        for rep in range(2):
            for name, setter in KNOBS:
                reset()
                try:
                    setter()
                    prog = net.build(B, 64)
                    prog.set_conditioning(ctx, te, ids)
                    out = prog.forward(x, torch.full((B,), 499.0)).clone()
                    prog.enable_graphs()
                    ms = timed(prog.prog_step.launch, 10 if B == 2 else 5)
                finally:
                    reset()
                if ref is None:
                    ref = out
                err = float((out.float() - ref.float()).norm() / ref.float().norm())
                best[name] = min(best.get(name, 1e9), ms)
                print(f"UNet step B={B:2d} rep {rep} {name:24s}: {ms:8.3f} ms   rel-L2 vs default {err:.2e}", flush=True)
                del prog
        print(f"B={B}: " + ", ".join(f"{k} {v:.3f}" for k, v in best.items()), flush=True)


if __name__ == "__main__":
    main()
