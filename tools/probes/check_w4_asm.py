"""Audit of tools/probes/gemm_w4.hip's code object (run here, no GPU): the kernel keeps its 64 accumulator tiles in a[0:255] under fixed
names behind the compiler's back, so the compiler must never touch the accumulator file itself and must never spill.
Checks the hipcc assembly of every instantiation: no v_accvgpr_* outside ;;#ASMSTART / ;;#ASMEND, no scratch access,
exactly 128 in-place MFMAs per K-tile iteration of the main loop.   Usage: python tools/probes/check_w4_asm.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "latentblending_amd", "csrc")
PROBE = os.path.join(ROOT, "tools", "probes", "gemm_w4.hip")


def main():
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-I", SRC, "-S", "--cuda-device-only",
               PROBE, "-o", os.path.join(td, "w4.s")]
        subprocess.check_call(cmd)
        text = open(os.path.join(td, "w4.s")).read()
    kernels = re.findall(r"^(_Z18gemm_f16_w4_kernel\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
    assert kernels, "no gemm_f16_w4_kernel instantiation found"
    ok = True
    for name, body in kernels:
        inside, bad_acc, scratch, mfma_inplace, mfma_other = False, 0, 0, 0, 0
        for line in body.splitlines():
            if "#ASMSTART" in line:
                inside = True
            elif "#ASMEND" in line:
                inside = False
            elif "v_accvgpr" in line and not inside:
                bad_acc += 1
            elif "scratch_" in line:
                scratch += 1
            m = re.search(r"v_mfma_f32_16x16x32_f16 (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], (a\[\d+:\d+\])", line)
            if m:
                if m.group(1) == m.group(2):
                    mfma_inplace += 1
                else:
                    mfma_other += 1
        print(f"{name}: compiler accvgpr ops {bad_acc}, scratch ops {scratch}, in-place MFMAs {mfma_inplace}, other MFMAs {mfma_other}")
        ok &= bad_acc == 0 and scratch == 0 and mfma_inplace == 128 and mfma_other == 0
    print("OK" if ok else "AUDIT FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
