"""Round-2 kernel micro-benchmarks on one MI355X (through gpurun): attention variants on the UNet's shapes, GroupNorm /
LayerNorm on the UNet / VAE shapes, batched slerp.  Every op is timed as a hipGraph of REP back-to-back launches
(no host launch cost), random data.  Writes gpurun_out/bench_round2.json."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from latentblending_amd.hip import lib, ops as o
from latentblending_amd.native.runtime import Program

DEV, REP = "cuda", 20


def graph_time(emit, rep=REP, rounds=3):
    """emit(): issues ONE launch through the C-ABI; returns microseconds per launch."""
    prog = Program("bench")
    with prog.record():
        for _ in range(rep):
            emit()
    prog.instantiate()
    st = torch.cuda.current_stream().cuda_stream
    prog.launch(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(rounds):
        prog.launch(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (rounds * rep) * 1e3


def main():
    out = {}
    zp = o.zero_page(DEV)
    # ---- attention: self (fused [tokens][3C] buffer) and cross (context K|V buffer), UNet shapes at B = 2 and 17
    for B in (17, 2):
        for (H, S, kind) in [(10, 1024, "self"), (20, 256, "self"), (10, 1024, "cross"), (20, 256, "cross")]:
            Cc = H * 64
            if kind == "self":
                qkv = torch.randn(B * S, 3 * Cc, device=DEV).half()
                q, k, v, Skv, valid = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], S, S
            else:
                q = torch.randn(B * S, Cc, device=DEV).half()
                kv = torch.randn(B * 80, 2 * Cc, device=DEV).half()
                k, v, Skv, valid = kv[:, :Cc], kv[:, Cc:], 80, 77
            outb = torch.empty(B * S, Cc, device=DEV, dtype=torch.float16)
            flops = 4.0 * B * H * S * valid * 64
            for force in (0, 1, 2, 17, 18):
                if kind == "self" and force in (17, 18) and S > 96:
                    continue                     # (long sequences always stream)
                lib.api.lb_attn_set_tuning(force)
                try:
                    us = graph_time(lambda: o.attention_d64(q, k, v, B, H, S, Skv, valid, out=outb))
                finally:
                    lib.api.lb_attn_set_tuning(0)
                key = f"attn_{kind}_B{B}_H{H}_S{S}_f{force}"
                out[key] = {"us": us, "TF": flops / us / 1e6}
                print(f"{key:36s} {us:9.1f} us {flops / us / 1e6:8.1f} TF/s", flush=True)
    # ---- GroupNorm (+SiLU): UNet and VAE shapes
    for (B, HW, Cc, f32) in [(17, 4096, 320, 0), (17, 1024, 640, 0), (17, 256, 1280, 0), (17, 256, 2560, 0), (2, 4096, 320, 0),
                             (2, 256, 1280, 0), (17, 4096, 512, 0), (17, 16384, 512, 0), (17, 65536, 256, 0),
                             (17, 262144, 128, 0), (17, 65536, 512, 0), (17, 262144, 256, 0)]:
        x = torch.randn(B, HW, Cc, device=DEV).half()
        y = torch.empty_like(x)
        g = torch.ones(Cc, device=DEV); bt = torch.zeros(Cc, device=DEV)
        ws = torch.empty(lib.api.lb_groupnorm_workspace_bytes(B, 32) // 8, dtype=torch.float64, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        us = graph_time(lambda: lib.api.lb_groupnorm_nhwc(x.data_ptr(), y.data_ptr(), g.data_ptr(), bt.data_ptr(), ws.data_ptr(),
                                                          B, HW, Cc, Cc, Cc, 32, 1e-5, 1, 0, 0), rep=10)
        gb = B * HW * Cc * 2 * 3 / 1e9          # read (stats) + read + write (apply)
        key = f"gn_B{B}_HW{HW}_C{Cc}"
        out[key] = {"us": us, "GBs_3pass": gb / us * 1e6}
        print(f"{key:36s} {us:9.1f} us {gb / us * 1e6:8.0f} GB/s (3 passes of {gb / 3 * 1e3:.0f} MB)", flush=True)
    # ---- LayerNorm
    for (M, Cc) in [(17 * 1024, 640), (17 * 256, 1280), (2 * 1024, 640), (2 * 256, 1280)]:
        x = torch.randn(M, Cc, device=DEV).half()
        y = torch.empty_like(x)
        g = torch.ones(Cc, device=DEV); bt = torch.zeros(Cc, device=DEV)
        us = graph_time(lambda: lib.api.lb_layernorm_f16(x.data_ptr(), y.data_ptr(), g.data_ptr(), bt.data_ptr(), M, Cc, Cc, Cc,
                                                         1e-5, 0))
        key = f"ln_M{M}_C{Cc}"
        out[key] = {"us": us, "GBs": M * Cc * 4 / us / 1e3}
        print(f"{key:36s} {us:9.1f} us {M * Cc * 4 / us / 1e3:8.0f} GB/s", flush=True)
    # ---- batched slerp on a >= 1 GiB problem (6 B / element algorithmic)
    for n in (16384, 32768, 65536):
        pairs = (1 << 30) // (n * 2 * 3) * 2
        p0 = torch.randn(pairs, n, device=DEV).half(); p1 = torch.randn(pairs, n, device=DEV).half()
        fr = torch.rand(pairs, device=DEV, dtype=torch.float64)
        ob = torch.empty_like(p0)
        us = graph_time(lambda: lib.api.lb_slerp_batched_f16(p0.data_ptr(), p1.data_ptr(), ob.data_ptr(), fr.data_ptr(), pairs, n, 0),
                        rep=4)
        out[f"slerp_batched_n{n}"] = {"us": us, "GBs": pairs * n * 6 / us / 1e3, "pairs": pairs}
        print(f"slerp_batched n={n} pairs={pairs}: {us:9.1f} us {pairs * n * 6 / us / 1e3:8.0f} GB/s", flush=True)
        del p0, p1, ob
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/bench_round2.json", "w"), indent=1)


if __name__ == "__main__":
    main()
