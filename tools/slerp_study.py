"""Where does the batched slerp's time go?  Product kernel vs the same kernel with (1) lerp weights instead of the float64
sqrt / acos / sin chain and (2) an fp32 weighted sum instead of the float64 one (not exact), on the >= 1 GiB batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # NEEDS a study build: python -m latentblending_amd.csrc.build --study; LB_HIP_LIBRARY=latentblending_amd/hip/liblbhip_study.so (LB_STUDY_BUILD)
from latentblending_amd.hip import lib
from tools.bench_round2 import graph_time

n = 16384
pairs = (1 << 30) // (n * 2 * 3)
p0 = torch.randn(pairs, n, device="cuda").half(); p1 = torch.randn(pairs, n, device="cuda").half()
fr = torch.rand(pairs, device="cuda", dtype=torch.float64)
ob = torch.empty_like(p0)
for study in (0, 1, 2, 0):
    lib.api.lb_slerp_set_study(study)
    us = graph_time(lambda: lib.api.lb_slerp_strided_f16(p0.data_ptr(), n, p1.data_ptr(), n, ob.data_ptr(), fr.data_ptr(), pairs, n, 0), rep=4)
    print(f"study {study}: {us:8.1f} us  {pairs * n * 6 / us / 1e3:7.0f} GB/s", flush=True)
lib.api.lb_slerp_set_study(0)
