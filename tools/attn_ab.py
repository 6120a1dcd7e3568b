"""Attention kernel variants on the UNet's B = 17 shapes (hipEvents over 50 back-to-back launches each, rotating over 4 buffer
sets so that Q / O stream from HBM as they do inside the programs).  lb_attn_set_tuning: bits 0..1 = query groups per wave
(1 / 2; 0 = by shape), bit 6 = the former two-stage form of the one-tile (cross-attention) kernel, bit 7 = 8-byte output stores,
bit 8 (round 6) = the streaming kernel of rounds 1-5 instead of attn_fwd_d64_stream_kernel, bit 9 = the 8-wave ping-pong form,
bit 11 = the block order of rounds 1-5 (query block fastest) instead of the XCD-aware one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from latentblending_amd.hip import ops as o, lib as l
    DEV = "cuda"
    for B in (17, 3, 2):
        for (H, S, kind) in [(10, 1024, "self"), (20, 256, "self"), (10, 1024, "cross"), (20, 256, "cross")]:
            Cc = H * 64
            sets = []
            for _ in range(4):
                if kind == "self":
                    qkv = torch.randn(B * S, 3 * Cc, device=DEV).half()
                    sets.append((qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:], S, S))
                else:
                    q = torch.randn(B * S, Cc, device=DEV).half()
                    kv = torch.randn(B * 80, 2 * Cc, device=DEV).half()
                    sets.append((q, kv[:, :Cc], kv[:, Cc:], 80, 77))
            outb = torch.empty(B * S, Cc, device=DEV, dtype=torch.float16)
            flops = 4.0 * B * H * S * (S if kind == "self" else 77) * 64
            ref, line = None, f"B={B:2d} H={H:2d} S={S:4d} {kind:5s}"
            for force in ([0, 2048, 256, 513, 514, 2048 + 513] if kind == "self" else [0, 2048, 1, 2]):
                l.api.lb_attn_set_tuning(force)
                try:
                    for i in range(8):
                        q, k, v, Skv, valid = sets[i % 4]
                        o.attention_d64(q, k, v, B, H, S, Skv, valid, out=outb)
                    got = outb.clone()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(50):
                        q, k, v, Skv, valid = sets[i % 4]
                        o.attention_d64(q, k, v, B, H, S, Skv, valid, out=outb)
                    e1.record()
                    torch.cuda.synchronize()
                finally:
                    l.api.lb_attn_set_tuning(0)
                us = e0.elapsed_time(e1) * 1e3 / 50
                if ref is None:
                    ref = got
                same = "=" if torch.equal(got, ref) else f"d {float((got.float() - ref.float()).abs().max()):.1e}"
                line += f" | force {force:3d}: {us:6.1f} us {flops / us / 1e6:6.0f} TF/s {same}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
