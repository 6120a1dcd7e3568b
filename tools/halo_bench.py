"""A/B of the implicit-GEMM conv (library's own tile choice) against the halo-tile 3x3 kernel on the conv
shapes of the benchmark's programs (VAE B=17, UNet B=17 / B=2), each timed as a hipGraph of 20 launches.
Writes gpurun_out/halo_bench.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib
from tools.sweep_gemm import time_variant

DEV = "cuda"


def main():
    shapes = []
    for B in (17, 2):
        shapes += [(B, 64, 512, 512), (B, 128, 512, 512), (B, 256, 512, 256), (B, 256, 256, 256),
                   (B, 512, 256, 128), (B, 512, 128, 128)]                       # VAE decoder
        shapes += [(B, 64, 320, 320), (B, 32, 640, 640), (B, 16, 1280, 1280), (B, 16, 2560, 1280), (B, 16, 1920, 1280),
                   (B, 32, 1920, 640), (B, 32, 1280, 640), (B, 32, 960, 640), (B, 32, 320, 640), (B, 16, 640, 1280),
                   (B, 64, 960, 320), (B, 64, 640, 320)]                         # UNet resnets
    rows = []
    for (B, H, C1, C2) in shapes:
        p = lib.LbGemmParams()
        x = torch.randn(B, H, H, C1, device=DEV).half()
        w = (torch.randn(C2, 9 * C1, device=DEV) * (9 * C1) ** -0.5).half()
        out = torch.empty(B, H, H, C2, device=DEV, dtype=torch.float16)
        M, N, K = B * H * H, C2, 9 * C1
        p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = 1, H, H, C1, H, H, 3, 3, 1, 1, 0, C1
        p.A, p.W, p.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
        p.M, p.N, p.K, p.ldw, p.ldc = M, N, K, K, C2
        zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
        p.zero_page = zp.data_ptr()
        small = ((M + 63) // 64) * ((N + 63) // 64) <= 640
        ws = torch.empty(min(lib.api.lb_gemm_workspace_bytes(M, N) // 4, 1 << 28), dtype=torch.float32, device=DEV) if small else None
        p.partial = ws.data_ptr() if small else None
        flops = 2.0 * M * N * K
        auto = time_variant(p, 0, 0, 0)
        lib.api.lb_conv_halo_set_persistent(0)
        halo1 = time_variant(p, 0, 0, 0, halo=True)
        lib.api.lb_conv_halo_set_persistent(1)
        halo = time_variant(p, 0, 0, 0, halo=True)
        row = {"shape": f"B{B} {H}x{H} {C1}->{C2}", "auto_us": auto, "halo_us": halo, "halo_one_item_us": halo1,
               "auto_TF": flops / auto / 1e6, "halo_TF": flops / halo / 1e6, "halo_one_item_TF": flops / halo1 / 1e6}
        rows.append(row)
        print(f"{row['shape']:24s} implicit GEMM {auto:9.1f} us {row['auto_TF']:6.0f} TF | halo, one item/block {halo1:9.1f} us "
              f"{row['halo_one_item_TF']:6.0f} TF | halo, persistent {halo:9.1f} us {row['halo_TF']:6.0f} TF | x{auto / halo:.2f} x{halo1 / halo:.2f}", flush=True)
        del x, w, out
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/halo_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
