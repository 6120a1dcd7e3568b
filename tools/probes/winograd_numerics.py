"""CPU numerics study (no GPU, nothing of the product imports it): Winograd F(2x2,3x3) with fp16-rounded transformed inputs / weights and
fp32 accumulation against the direct fp16 conv, on SiLU-shaped activations with and without outlier channels.  Result (DESIGN.md
section 7): rel-L2 5.0-5.5e-4 vs 2.1e-4 - numerically admissible; set aside for LDS-traffic reasons, not for accuracy."""
import torch, math
torch.manual_seed(0)
torch.set_num_threads(16)
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float64)
G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float64)

def rel(a, b): return float((a.double()-b.double()).norm()/b.double().norm())

def study(C, Cout, H, kind):
    x = torch.randn(1, C, H, H)
    if kind == "silu": x = torch.nn.functional.silu(x * 1.5)
    if kind == "outlier": x = torch.nn.functional.silu(x * 1.5); x[:, ::17] *= 12.0
    w = torch.randn(Cout, C, 3, 3) * (9 * C) ** -0.5
    xh, wh = x.half(), w.half()
    ref = torch.nn.functional.conv2d(xh.double(), wh.double(), padding=1)            # exact for the fp16 operands
    direct = torch.nn.functional.conv2d(xh.float(), wh.float(), padding=1).half()     # today's kernel: fp32 accumulate, fp16 store
    # Winograd F(2x2,3x3): tiles of 4x4 inputs at stride 2
    xp = torch.nn.functional.pad(xh.double(), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                         # [1,C,T,T,4,4]
    V = torch.einsum("ij,bctujk,lk->bctuil", Bt, d, Bt)                            # B^T d B (exact in fp32: adds of fp16 values)
    Vh = V.float().half()                                                          # operand rounding
    for wsrc, name in ((w.double(), "weights transformed from fp32 masters"), (wh.double(), "from fp16 weights")):
        U = torch.einsum("ij,ocjk,lk->ocil", G, wsrc, G)                           # G g G^T
        Uh = U.float().half()
        M = torch.einsum("ocil,bctuil->botuil", Uh.double(), Vh.double())          # fp32-accumulate emulated exactly in fp64
        Y = torch.einsum("ij,botujk,lk->botuil", At, M, At)                        # [1,O,T,T,2,2]
        T = Y.shape[2]
        out = Y.permute(0, 1, 2, 4, 3, 5).reshape(1, Cout, 2 * T, 2 * T).float().half()
        print(f"C={C:4d} Cout={Cout:4d} H={H:3d} {kind:8s}: direct fp16 rel-L2 {rel(direct, ref):.2e} | winograd ({name}) {rel(out, ref):.2e}  max|V| {float(Vh.abs().max()):.1f}")

for C, Cout, H, kind in [(128, 128, 64, "silu"), (256, 256, 32, "silu"), (512, 512, 32, "silu"), (320, 320, 32, "silu"), (1280, 1280, 16, "silu"), (128, 128, 64, "outlier"), (512, 512, 32, "outlier")]:
    study(C, Cout, H, kind)
