"""Dump what ds_read_b64_tr_b16 returns per lane for a few address patterns (LDS holds lds[i] = i as f16)."""
import ctypes as C
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "libtrprobe.so"))
lib.tr_probe.argtypes = [C.c_void_p, C.c_void_p]
lib.tr_probe.restype = C.c_int


def run(addrs):
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
    out = torch.zeros(64 * 4, dtype=torch.float32, device="cuda")
    rc = lib.tr_probe(a.data_ptr(), out.data_ptr())
    assert rc == 0, rc
    return out.cpu().view(64, 4).to(torch.int64).tolist()


patterns = {
    "linear8": [8 * l for l in range(64)],
    "row128_per_lane": [(l & 15) * 128 + (l >> 4) * 8 for l in range(64)],
    "block4x16_stride128": [((l & 15) >> 2) * 128 + (l & 3) * 8 + (l >> 4) * 512 for l in range(64)],
    "block4x16_stride32": [((l & 15) >> 2) * 32 + (l & 3) * 8 + (l >> 4) * 128 for l in range(64)],
}
res = {}
for name, addrs in patterns.items():
    got = run(addrs)
    res[name] = {"addr_elems": [a // 2 for a in addrs], "got": got}
    print("==", name)
    for l in range(64):
        print(f"  lane {l:2d} addr_elem {addrs[l] // 2:5d} -> {got[l]}")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/tr_probe.json", "w"))
