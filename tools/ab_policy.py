"""In-situ A/B of the GEMM tile policy: record the B=17 UNet step program and the B=17 VAE decode program
under several policies (lb_gemm_set_policy masks) and time their hipGraph replays.  Isolated sweeps
(tools/sweep_gemm.py) do not always predict the in-program ranking, so the defaults follow this tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # NEEDS a study build: python -m latentblending_amd.csrc.build --study; LB_HIP_LIBRARY=latentblending_amd/hip/liblbhip_study.so (LB_STUDY_BUILD)
import latentblending_amd.native as N
from latentblending_amd.hip import lib


def replay_ms(launch, reps):
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 17
    masks = [int(a) for a in sys.argv[2:] if not a.startswith("--")] or [0, 2, 4, 8, 1 | 2, 16]
    pipe = N.NativeSDXLPipe(turbo=True)
    for mask in masks:
        lib.api.lb_gemm_set_policy(mask)
        up = pipe.unet_native.build(B, 64)
        up.set_conditioning(torch.randn(B, 77, 2048, device="cuda").half(), torch.randn(B, 1280, device="cuda").half(),
                            torch.tensor([[512.0, 512, 0, 0, 512, 512]] * B, device="cuda"))
        up.forward(torch.randn(B, 4, 64, 64, device="cuda").half(), torch.full((B,), 499.0))
        up.enable_graphs()
        t_u = replay_ms(up.prog_step.launch, 10)
        if "--unet-only" in sys.argv:
            print(f"policy mask {mask:2d}: UNet B={B} {t_u:8.3f} ms", flush=True)
            continue
        vp = pipe.vae_native.build(B, 64)
        vp.decode(torch.randn(B, 4, 64, 64, device="cuda").half())
        vp.prog.instantiate()
        t_v = replay_ms(vp.prog.launch, 5)
        print(f"policy mask {mask:2d}: UNet B={B} {t_u:8.3f} ms   VAE B={B} {t_v:8.3f} ms", flush=True)
        del up, vp
        torch.cuda.empty_cache()
    lib.api.lb_gemm_set_policy(0)


if __name__ == "__main__":
    main()
