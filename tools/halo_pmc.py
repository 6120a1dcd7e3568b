"""The halo-tile conv (conv3x3_halo_kernel<128,32,3>, 26 % of GPU time) on the VAE's two big shapes and one UNet shape, a few eager
launches each with a residual + bias epilogue, for `rocprofv3 --pmc ... --kernel-trace` (separate passes per counter set; the epilogue
form comes from LB_GEMM_LEAN_EPILOGUE = 0 / 1 in the environment, so a before / after pair is two runs).  tools/pmc_fold.py folds the
CSVs.  Usage: see tools/calls/r05_call4.sh"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib

DEV = "cuda"


def main():
    zp = torch.zeros(64, dtype=torch.uint8, device=DEV)
    for (B, H, C1, C2) in [(17, 512, 128, 128), (17, 256, 256, 256), (17, 32, 1280, 640)]:
        x = torch.randn(B, H, H, C1, device=DEV).half()
        w = (torch.randn(C2, 9 * C1, device=DEV) * (9 * C1) ** -0.5).half()
        res = torch.randn(B, H, H, C2, device=DEV).half()
        bias = torch.randn(C2, device=DEV)
        out = torch.empty(B, H, H, C2, device=DEV, dtype=torch.float16)
        p = lib.LbGemmParams()
        p.conv, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.KH, p.KW, p.stride, p.pad, p.ups, p.ldx = 1, H, H, C1, H, H, 3, 3, 1, 1, 0, C1
        p.A, p.W, p.C, p.bias, p.residual = x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr()
        p.M, p.N, p.K, p.ldw, p.ldc, p.ldr, p.alpha = B * H * H, C2, 9 * C1, 9 * C1, C2, C2, 1.0
        p.zero_page = zp.data_ptr()
        for _ in range(3):
            lib.api.lb_conv3x3_halo_f16(C.byref(p), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        del x, w, res, out


if __name__ == "__main__":
    main()
