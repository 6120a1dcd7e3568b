"""Build liblbhip.so (gfx950) in-tree with hipcc.  No cmake, no torch headers: the library is a
plain C-ABI shared object loaded with ctypes (include/lb_hip.h is the contract)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "hip")
LIB_PATH = os.path.join(OUT_DIR, "liblbhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
         "-Wno-unused-result", "-I", HERE]


# per-source extra flags (see the header comment of the file for the reason)
_ATTN_FLAGS = ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# gemm_glds.hip (round 6): with the K loop instantiated twice (lean / general requests) hipcc parks the accumulators of the small tiles in
# AGPRs and rotates them through v_accvgpr_read / _mov / _write every K-tile (24-92 extra instructions per tile); the VGPR form keeps them put
EXTRA_FLAGS = {"attn.hip": _ATTN_FLAGS, "attn512.hip": _ATTN_FLAGS, "gemm_glds.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _needs(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, sub="build", extra=()):
    obj = os.path.join(HERE, sub, src.replace(".hip", ".o"))
    headers = [os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".h")]
    if _needs(obj, [os.path.join(HERE, src), os.path.abspath(__file__)] + headers):
        cmd = [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), *extra, "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    return obj


def build_library(force: bool = False, verbose: bool = True, study: bool = False) -> str:
    """``study=True``: liblbhip_study.so with -DLB_STUDY_BUILD - the timing-study switches (lb_slerp_set_study,
    lb_conv_halo_set_study, lb_gemm_set_policy) exist only there; load it with LB_HIP_LIBRARY=<path> (tools/ only)."""
    sub = "build_study" if study else "build"
    lib_path = os.path.join(OUT_DIR, "liblbhip_study.so") if study else LIB_PATH
    os.makedirs(os.path.join(HERE, sub), exist_ok=True)
    if force:
        for f in os.listdir(os.path.join(HERE, sub)):
            os.remove(os.path.join(HERE, sub, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda src: _compile(src, sub, ["-DLB_STUDY_BUILD"] if study else []), sources()))
    LIB_PATH_ = lib_path
    if _needs(LIB_PATH_, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH_]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if verbose:
        print(f"[lb build] {LIB_PATH_} ({os.path.getsize(LIB_PATH_) / 1e6:.1f} MB)")
    return LIB_PATH_


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, study="--study" in sys.argv)
