"""A/B of the VAE mid-block attention (512 channels, one head): the three-launch form per sample (scores GEMM -> row softmax ->
PV GEMM, an S x S fp16 score buffer) vs ONE lb_attn_fwd_d512 launch over the whole batch (VAEConfig.fused_mid_attention).
Prints the kernel alone (B = 17, S = 4096) and the full decode program (hipGraph replays), and compares the frames.
Usage: python tools/vae_attn_ab.py > gpurun_out/vae_attn_ab.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import latentblending_amd.native as N
from latentblending_amd.hip import ops

DEV = "cuda:0"


def _time(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def kernel_alone():
    for B, S in ((17, 4096), (1, 16384), (17, 1024)):
        qkv = (torch.randn(B * S, 1536, generator=torch.Generator().manual_seed(5)) * 1.0).half().to(DEV)
        q, k, v = qkv[:, :512], qkv[:, 512:1024], qkv[:, 1024:]
        out = torch.empty(B * S, 512, dtype=torch.float16, device=DEV)
        ms = _time(lambda: ops.attention_d512(q, k, v, B, 1, S, S, out=out))
        fl = 4.0 * B * S * S * 512
        print(f"lb_attn_fwd_d512 B={B} S={S}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


def main():
    kernel_alone()
    B, L = int(os.environ.get("LB_AB_BATCH", "17")), 64
    z = torch.randn(B, 4, L, L, generator=torch.Generator().manual_seed(3)).half().to(DEV)
    outs, times = {}, {}
    for fused in (False, True, False, True):
        net = N.NativeVAEDecoder(N.VAEConfig(fused_mid_attention=fused), N.SyntheticProvider(1), DEV)
        prog = net.build(B, L)
        prog.decode(z)
        prog.prog.instantiate()
        ms = _time(prog.prog.launch, 5)
        names = prog.prog.op_names()
        times.setdefault(fused, []).append(ms)
        outs[fused] = prog.decode(z).clone()
        print(f"fused_mid_attention={fused!s:5}: {ms:7.3f} ms per decode batch (B={B}, hipGraph), {len(names)} launches", flush=True)
        del prog, net
        torch.cuda.empty_cache()
    d = (outs[True].int() - outs[False].int()).abs()
    print(f"frames: mean |du8| between the two forms {d.float().mean():.4f}, max {int(d.max())}")
    print(f"best: three-launch form {min(times[False]):.3f} ms, fused {min(times[True]):.3f} ms")


if __name__ == "__main__":
    main()
