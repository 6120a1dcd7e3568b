"""Per-kernel parity on a real MI355X: every C-ABI launcher against the CPU oracle
(oracle/sdxl_ref.py) or a plain PyTorch fp32 reference of the same op, on seeded inputs.

Tolerances (stated per test): slerp / lerp / Euler <= 1 fp16 ulp (exact where the reference is
exact); fp16 GEMM / conv / attention / norms: rel-L2 <= 2e-3 and max-abs <= 2^-8 * max|ref|
(+ small absolute floor) against an fp32 reference evaluated on the same fp16-rounded inputs.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import sdxl_ref as R  # noqa: E402  (checker only)

DEV = "cuda"


def ops():
    from latentblending_amd.hip import ops as o
    return o


def lib():
    from latentblending_amd.hip import lib as l
    return l


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def ulp_diff_f16(a, b):
    ai = a.cpu().view(torch.int16).to(torch.int32)
    bi = b.cpu().view(torch.int16).to(torch.int32)
    ai = torch.where(ai < 0, -32768 - ai, ai)
    bi = torch.where(bi < 0, -32768 - bi, bi)
    return int((ai - bi).abs().max())


def check_close(log, name, got, ref, rel=2e-3, frac=2 ** -8, floor=1e-3):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    rl2 = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    bound = frac * ref.abs().max().item() + floor
    log[name] = {"max_abs": err, "rel_l2": rl2, "bound_abs": bound}
    print(f"[parity] {name}: max_abs={err:.3e} (bound {bound:.3e}) rel_l2={rl2:.3e}")
    assert rl2 <= rel, f"{name}: rel-L2 {rl2:.3e} > {rel}"
    assert err <= bound, f"{name}: max-abs {err:.3e} > {bound:.3e}"


# ------------------------------------------------------------------ mixing -------------------
@pytest.mark.parametrize("n", [4096, 16384, 65536])
def test_slerp_f16(n, results_log):
    o = ops()
    p0, p1 = rnd(1, 4, n // 4, seed=1, scale=3.0), rnd(1, 4, n // 4, seed=2, scale=3.0)
    worst = 0
    for f in [0.0, 0.25, 0.37, 0.5, 1.0]:
        got = o.slerp(p0.to(DEV), p1.to(DEV), f)
        ref = R.slerp(p0, p1, f)
        assert got.dtype == torch.float16 and got.shape == p0.shape
        d = ulp_diff_f16(got, ref)
        worst = max(worst, d)
        if f in (0.0, 1.0):
            assert torch.equal(got.cpu(), p0 if f == 0.0 else p1), f"slerp f={f} must be exact"
    results_log[f"slerp_f16_n{n}_max_ulp"] = worst
    assert worst <= 1


def test_slerp_edge_cases(results_log):
    o = ops()
    a = rnd(16384, seed=3)
    same = o.slerp(a.to(DEV), a.to(DEV), 0.3)
    assert ulp_diff_f16(same, R.slerp(a, a, 0.3)) <= 1
    anti = o.slerp(a.to(DEV), (-a).to(DEV), 0.5).cpu()
    ref = R.slerp(a, -a, 0.5)
    assert torch.allclose(anti.float(), ref.float(), atol=1e-3)
    z = torch.zeros(4096, dtype=torch.float16)
    assert torch.isnan(o.slerp(z.to(DEV), a[:4096].to(DEV), 0.5)).all() == torch.isnan(R.slerp(z, a[:4096], 0.5)).all()
    # fp32 / fp64 inputs come back as fp32 (reference utils.py:66-69)
    for dt in (torch.float32, torch.float64):
        x, y = rnd(1001, seed=4, dtype=dt), rnd(1001, seed=5, dtype=dt)
        got = o.slerp(x.to(DEV), y.to(DEV), 0.41)
        ref = R.slerp(x, y, 0.41)
        assert got.dtype == torch.float32
        assert torch.allclose(got.cpu(), ref, rtol=1e-6, atol=1e-6)
    # odd length + misaligned views take the scalar path
    x, y = rnd(4099, seed=6), rnd(4099, seed=7)
    xd, yd = x.to(DEV), y.to(DEV)
    assert ulp_diff_f16(o.slerp(xd[1:], yd[1:], 0.6), R.slerp(x[1:], y[1:], 0.6)) <= 1
    # several pairs in one launch (more than the 16-pair kernarg block)
    ps = [rnd(8192, seed=10 + i) for i in range(20)]
    qs = [rnd(8192, seed=40 + i) for i in range(20)]
    fr = [i / 19 for i in range(20)]
    outs = o.slerp_pairs([p.to(DEV) for p in ps], [q.to(DEV) for q in qs], fr)
    assert max(ulp_diff_f16(g, R.slerp(p, q, f)) for g, p, q, f in zip(outs, ps, qs, fr)) <= 1


def test_slerp_batched(results_log):
    o = ops()
    for n in (16384, 65536, 131072):     # LDS-staged and re-read variants
        p0, p1 = rnd(6, n, seed=8, scale=2.0), rnd(6, n, seed=9, scale=2.0)
        fr = torch.tensor([0.0, 0.1, 0.5, 0.77, 1.0, 0.33], dtype=torch.float64)
        got = o.slerp_batched(p0.to(DEV), p1.to(DEV), fr.to(DEV)).cpu()
        for i in range(6):
            assert ulp_diff_f16(got[i], R.slerp(p0[i], p1[i], float(fr[i]))) <= 1


@pytest.mark.parametrize("n", [4096, 16384, 32768, 8 * 37])
def test_slerp_strided(n, results_log):
    """One launch for G pairs with device-side fractions: contiguous batches, a broadcast pair (the parental mix of
    two anchors at many fractions) and the fract = 0 / 1 end points, bit-identical to the per-pair kernel."""
    o = ops()
    G = 9
    a, b = rnd(G, n, seed=71), rnd(G, n, seed=72)
    fr = [0.0, 1.0, 0.5, 0.25, 0.37, 0.8, 0.1234, 0.9, 0.6]
    frd = torch.tensor(fr, dtype=torch.float64, device=DEV)
    got = o.slerp_strided(a.to(DEV), b.to(DEV), frd, n).cpu()
    ref = torch.stack([R.slerp(a[g], b[g], fr[g]) for g in range(G)])
    per_pair = torch.stack([t.cpu() for t in o.slerp_pairs([a[g].to(DEV) for g in range(G)], [b[g].to(DEV) for g in range(G)], fr)])
    assert torch.equal(got, per_pair)
    assert torch.equal(got[0], a[0]) and torch.equal(got[1], b[1])
    u = ulp_diff_f16(got, ref)
    got_b = o.slerp_strided(a[0].to(DEV), b[0].to(DEV), frd, n, broadcast0=True, broadcast1=True).cpu()
    ref_b = torch.stack([R.slerp(a[0], b[0], f) for f in fr])
    ub = ulp_diff_f16(got_b, ref_b)
    results_log[f"slerp_strided_n{n}"] = {"ulp": u, "ulp_broadcast": ub}
    assert u <= 1 and ub <= 1


def test_frame_inbetweening_on_device_matches_reference_golden(results_log):
    """lb_frames_lerp_u8 through add_frames_linear_interp on DeviceImage key frames: every output frame equals the
    unchanged reference's (tests/golden/frames.json), byte for byte; plus a 512x512 case against the host path."""
    import hashlib
    import json
    import numpy as np
    from latentblending_amd import utils
    from latentblending_amd.native.frames import DeviceImage
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(np.asarray(a)).tobytes()).hexdigest()[:16]
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frames.json")))
    for c in gold:
        rng = np.random.RandomState(c["seed"])
        imgs = [rng.randint(0, 256, size=(c["h"], c["w"], 3)).astype(np.uint8) for _ in range(c["n"])]
        dev = [DeviceImage(torch.from_numpy(i).to(DEV)) for i in imgs]
        assert utils._device_frame_stack(dev) is not None or c["target"] <= c["n"]
        np.random.seed(c["rng_seed"])
        out = utils.add_frames_linear_interp(dev, nmb_frames_target=c["target"])
        assert len(out) == c["count"]
        assert [sha(o) for o in out] == c["sha"], f"case seed {c['seed']}"
    rng = np.random.RandomState(3)
    imgs = [rng.randint(0, 256, size=(512, 512, 3)).astype(np.uint8) for _ in range(5)]
    np.random.seed(9)
    host = utils.add_frames_linear_interp(imgs, nmb_frames_target=40)
    np.random.seed(9)
    devo = utils.add_frames_linear_interp([DeviceImage(torch.from_numpy(i).to(DEV)) for i in imgs], nmb_frames_target=40)
    assert len(host) == len(devo) == 40 and all(np.array_equal(a, b) for a, b in zip(host, devo))
    results_log["frames_lerp_u8"] = {"golden_cases": len(gold), "full_size_frames": 40}


def test_lerp_bit_exact(results_log):
    o = ops()
    a, b = rnd(1, 77, 2048, seed=11), rnd(1, 77, 2048, seed=12)
    for f in (0.0, 0.125, 0.5, 0.7321, 1.0):
        got = o.lerp(a.to(DEV), b.to(DEV), f).cpu()
        ref = R.lerp(a, b, f)
        assert got.dtype == ref.dtype and torch.equal(got, ref), f"lerp f={f}"
    a32, b32 = rnd(1, 1280, seed=13, dtype=torch.float32), rnd(1, 1280, seed=14, dtype=torch.float32)
    assert torch.equal(o.lerp(a32.to(DEV), b32.to(DEV), 0.3).cpu(), R.lerp(a32, b32, 0.3))


@pytest.mark.parametrize("ancestral", [True, False])
def test_euler_step_matches_oracle(ancestral, results_log):
    o = ops()
    sched = R.EulerScheduler(ancestral=ancestral)
    sched.set_timesteps(4 if ancestral else 30)
    x = rnd(2, 4, 64, 64, seed=15, scale=5.0)
    eps = rnd(2, 4, 64, 64, seed=16)
    noise = rnd(2, 4, 64, 64, seed=17)
    sched.noise_source = lambda shape: noise[:1]
    worst = 0
    for i in range(len(sched.timesteps) - (0 if not ancestral else 0)):
        t = sched.timesteps[i]
        sched._step_index = None
        scaled_ref = sched.scale_model_input(x[:1], t)
        sched._step_index = None
        ref = sched.step(eps[:1], t, x[:1])[0]
        s_from, s_to = float(sched.sigmas[i]), float(sched.sigmas[i + 1])
        if ancestral:
            s_up, s_down = R.ancestral_sigmas(s_from, s_to)
            row = (s_from, s_down, s_up, 0.0, s_down - s_from)
        else:
            row = (s_from, s_to, 0.0, 0.0, s_to - s_from)
        params = o.step_params([row, row], DEV)
        got_scaled = o.scale_model_input(x.to(DEV), params)
        assert ulp_diff_f16(got_scaled[:1], scaled_ref) <= 1
        got = o.euler_step(x.to(DEV), eps.to(DEV), params, noise=noise[:1].expand(2, -1, -1, -1).contiguous().to(DEV),
                           ancestral=ancestral)
        worst = max(worst, ulp_diff_f16(got[:1], ref))
    results_log[f"euler_{'anc' if ancestral else 'plain'}_max_ulp"] = worst
    assert worst <= 1


@pytest.mark.parametrize("cfg", [False, True])
def test_ddim_step_matches_oracle(cfg, results_log):
    """lb_ddim_step_f16 (eta = 0) against the oracle restatement of diffusers' DDIMScheduler.step on fp16 tensors - which, like
    diffusers, rounds every tensor operation to fp16 - at the first, a middle and the last step (prev_timestep < 0), with and
    without the fp16 CFG combine of diffusers_holder.py:347-349: <= 1 fp16 ulp (the kernel rounds in the same six places)."""
    from latentblending_amd.native.scheduler import NativeDDIMScheduler
    o = ops()
    sched, ref = NativeDDIMScheduler(device=DEV), R.DDIMScheduler()
    sched.set_timesteps(30); ref.set_timesteps(30)
    B, g = 3, 4.0
    x = rnd(B, 4, 64, 64, seed=401, scale=3.0)
    eps = rnd(2 * B if cfg else B, 4, 64, 64, seed=402)
    worst = 0
    for i in (0, 13, 29):
        t = int(ref.timesteps[i])
        e = eps
        if cfg:             # noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond), fp16 tensor arithmetic
            eu, et = eps[:B], eps[B:]
            e = eu + g * (et - eu)
        want = ref.step(e, t, x)[0]
        assert want.dtype == torch.float16
        params = o.step_params([sched.step_row(i, g)] * B, DEV)
        got = o.ddim_step(x.to(DEV), eps.to(DEV), params, cfg=cfg)
        worst = max(worst, ulp_diff_f16(got, want))
        # and through the diffusers-style API of the native scheduler (no CFG there: the caller combines)
        if not cfg:
            sched._step_index = None
            assert torch.equal(sched.step(eps.to(DEV), float(t), x.to(DEV))[0], got)
    results_log[f"ddim_step_cfg{int(cfg)}_max_ulp"] = worst
    print(f"[parity] ddim step cfg={cfg}: max ulp {worst}")
    assert worst <= 1
    # odd sizes (no 16-byte vectors)
    xs, es = rnd(2, 4, 5, 7, seed=403), rnd(2, 4, 5, 7, seed=404)
    params = o.step_params([sched.step_row(5)] * 2, DEV)
    assert ulp_diff_f16(o.ddim_step(xs.to(DEV), es.to(DEV), params), ref.step(es, int(ref.timesteps[5]), xs)[0]) <= 1


def test_euler_cfg_combine(results_log):
    o = ops()
    x = rnd(1, 4, 32, 32, seed=18, scale=4.0)
    eu, et = rnd(1, 4, 32, 32, seed=19), rnd(1, 4, 32, 32, seed=20)
    g, s_from, s_to = 3.5, 2.0, 1.5
    eps = eu + g * (et - eu)                       # fp16 tensor arithmetic, as diffusers_holder.py:349
    sched_x = x.float()
    x0 = sched_x - s_from * eps.float()
    ref = (sched_x + ((sched_x - x0) / s_from) * (s_to - s_from)).half()
    params = o.step_params([(s_from, s_to, 0.0, g, s_to - s_from)], DEV)
    got = o.euler_step(x.to(DEV), torch.cat([eu, et]).to(DEV), params, cfg=True)
    assert ulp_diff_f16(got, ref) <= 1
    dup = o.scale_model_input(x.to(DEV), params, dup_for_cfg=True)
    assert dup.shape[0] == 2 and torch.equal(dup[0], dup[1])


# ------------------------------------------------------------------ GEMM ---------------------
GEMM_SHAPES = [(256, 1280, 1280), (1024, 3840, 640), (100, 64, 72), (4096, 320, 2880), (77, 1280, 2048),
               (2, 1280, 320), (333, 132, 200)]


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_gemm_plain(shape, tile, results_log):
    o, l = ops(), lib()
    M, N, K = shape
    A, W = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5)
    bias = rnd(N, seed=23, dtype=torch.float32)
    res = rnd(M, N, seed=24)
    ref = A.float() @ W.float().t() + bias + res.float()
    l.api.lb_gemm_set_tuning(tile, 0)
    try:
        got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), residual=res.to(DEV))
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
    check_close(results_log, f"gemm_{M}x{N}x{K}_tile{tile}", got, ref)


def test_gemm_transpose_detecting(results_log):
    """A = I with an asymmetric W: a swapped fragment layout cannot pass."""
    o = ops()
    n = 128
    A = torch.eye(n, dtype=torch.float16)
    W = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 / 251).half()
    got = o.gemm(A.to(DEV), W.to(DEV))
    assert torch.equal(got.cpu(), W.t().contiguous())


@pytest.mark.parametrize("splitk", [2, 5, 16])
def test_gemm_splitk(splitk, results_log):
    o, l = ops(), lib()
    M, N, K = 256, 1280, 5120
    A, W = rnd(M, K, seed=25), rnd(N, K, seed=26, scale=K ** -0.5)
    bias, res = rnd(N, seed=27, dtype=torch.float32), rnd(M, N, seed=28)
    ref = A.float() @ W.float().t() + bias + res.float()
    l.api.lb_gemm_set_tuning(3, splitk)
    try:
        got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), residual=res.to(DEV))
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
    check_close(results_log, f"gemm_splitk{splitk}", got, ref)
    got_auto = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), residual=res.to(DEV))
    check_close(results_log, "gemm_splitk_auto", got_auto, ref)


@pytest.mark.parametrize("tile", [1, 2, 3])
def test_gemm_geglu(tile, results_log):
    o, l = ops(), lib()
    M, C = 300, 640
    A = rnd(M, C, seed=29)
    W = rnd(8 * C, C, seed=30, scale=C ** -0.5)
    bias = rnd(8 * C, seed=31, dtype=torch.float32, scale=0.1)
    proj = A.float() @ W.float().t() + bias
    h, gate = proj.chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    l.api.lb_gemm_set_tuning(tile, 0)
    try:
        got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), flags=l.GEMM_GEGLU)
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
    assert got.shape == (M, 4 * C)
    check_close(results_log, f"gemm_geglu_tile{tile}", got, ref)


def test_gemm_epilogues(results_log):
    o, l = ops(), lib()
    M, N, K = 512, 256, 320
    A, W = rnd(M, K, seed=32), rnd(N, K, seed=33, scale=K ** -0.5)
    bias = rnd(N, seed=34, dtype=torch.float32)
    base = A.float() @ W.float().t()
    # transposed store
    got = o.gemm(A.to(DEV), W.to(DEV), flags=l.GEMM_TRANS_OUT)
    check_close(results_log, "gemm_trans_out", got, base.t())
    # fp32 output with fp32 residual, alpha
    res32 = rnd(M, N, seed=35, dtype=torch.float32, scale=100.0)
    got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), residual=res32.to(DEV), alpha=0.5,
                 flags=l.GEMM_OUT_F32 | l.GEMM_RES_F32)
    assert got.dtype == torch.float32
    check_close(results_log, "gemm_f32_out", got, 0.5 * base + bias + res32, rel=1e-4, frac=2 ** -12)
    # per-sample row vector (time-embedding add): 4 samples x 128 rows
    rv = rnd(4, N, seed=36)
    got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), rowvec=rv.to(DEV), rows_per_batch=128)
    check_close(results_log, "gemm_rowvec", got, base + bias + rv.float().repeat_interleave(128, 0))
    # activations
    got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), flags=l.GEMM_SILU)
    check_close(results_log, "gemm_silu", got, F.silu(base + bias))
    got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), flags=l.GEMM_RELU)
    check_close(results_log, "gemm_relu", got, F.relu(base + bias))


# ------------------------------------------------------------------ conv ---------------------
CONV_CASES = [
    # (B, H, W, Cin, Cout, k, stride, pad, ups)
    (2, 16, 16, 64, 128, 3, 1, 1, 0),
    (1, 32, 32, 320, 320, 3, 1, 1, 0),
    (2, 16, 16, 64, 64, 3, 2, 1, 0),      # UNet downsampler
    (1, 8, 8, 128, 128, 3, 1, 1, 1),      # fused nearest-2x upsample + conv
    (2, 16, 16, 4, 64, 3, 1, 1, 0),       # conv_in: Cin 4 padded to 8
    (1, 16, 16, 192, 64, 1, 1, 0, 0),     # 1x1 shortcut
    (1, 16, 16, 320, 4, 3, 1, 1, 0),      # conv_out: tiny N
    (1, 64, 64, 3, 64, 11, 4, 2, 0),      # LPIPS conv1
    (1, 15, 15, 64, 192, 5, 1, 2, 0),     # LPIPS conv2
    (1, 10, 14, 72, 96, 3, 1, 1, 0),      # non-square, Cin not a multiple of 64
]


@pytest.mark.parametrize("tile", [0, 1, 3])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_implicit_gemm(case, tile, results_log):
    o, l = ops(), lib()
    B, H, Wd, Cin, Cout, k, st, pad, ups = case
    x = rnd(B, Cin, H, Wd, seed=37)
    w = rnd(Cout, Cin, k, k, seed=38, scale=(Cin * k * k) ** -0.5)
    bias = rnd(Cout, seed=39, dtype=torch.float32)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), bias, stride=st, padding=pad).permute(0, 2, 3, 1)
    cin_p = (Cin + 7) // 8 * 8
    cout_p = (Cout + 3) // 4 * 4
    x_nhwc = torch.zeros(B, H, Wd, cin_p, dtype=torch.float16)
    x_nhwc[..., :Cin] = x.permute(0, 2, 3, 1)
    wp = torch.zeros(cout_p, k * k * cin_p, dtype=torch.float16)
    wp[:Cout] = o.pack_conv_weight(w, cin_p)
    bp = torch.zeros(cout_p, dtype=torch.float32)
    bp[:Cout] = bias
    l.api.lb_gemm_set_tuning(tile, 0)
    try:
        got = o.gemm(x_nhwc.to(DEV), wp.to(DEV), bias=bp.to(DEV),
                     conv=dict(KH=k, KW=k, stride=st, pad=pad, ups=ups))
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
    check_close(results_log, f"conv_{'_'.join(map(str, case))}_tile{tile}", got[..., :Cout], ref)


def test_conv_resnet_epilogue(results_log):
    """conv + bias + per-sample time-embedding vector + residual, the ResnetBlock2D epilogue."""
    o = ops()
    B, H, Wd, C = 2, 16, 16, 128
    x, w = rnd(B, C, H, Wd, seed=40), rnd(C, C, 3, 3, seed=41, scale=(9 * C) ** -0.5)
    bias, temb, res = rnd(C, seed=42, dtype=torch.float32), rnd(B, C, seed=43), rnd(B, H, Wd, C, seed=44)
    ref = (F.conv2d(x.float(), w.float(), bias, padding=1) + temb.float()[:, :, None, None]).permute(0, 2, 3, 1) + res.float()
    got = o.gemm(x.permute(0, 2, 3, 1).contiguous().to(DEV), o.pack_conv_weight(w).to(DEV), bias=bias.to(DEV),
                 rowvec=temb.to(DEV), rows_per_batch=H * Wd, residual=res.to(DEV),
                 conv=dict(KH=3, KW=3, stride=1, pad=1))
    check_close(results_log, "conv_resnet_epilogue", got, ref)


# ------------------------------------------------------------------ norms --------------------
@pytest.mark.parametrize("case", [(1, 4096, 320, False), (2, 256, 2560, False), (1, 1024, 1920, False),
                                  (1, 16384, 128, True), (2, 64, 32, False), (1, 4096, 512, True),
                                  # round 6, one-launch form (slab in registers): the UNet's 32^2 / 16^2 levels, a ragged pixel count
                                  (3, 1024, 640, False), (2, 1024, 320, False), (3, 256, 1280, False), (2, 256, 1920, False),
                                  (2, 1024, 1280, False), (2, 250, 640, False)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm(case, silu, results_log):
    """fp32 torch reference; where the one-launch form applies it must also agree with the statistics + apply launches to rounding
    (same float64 E[x^2] - E[x]^2, different fp32 partial sums: <= 2 fp16 ulps of the largest output)."""
    from latentblending_amd.hip.lib import api
    o = ops()
    B, HW, C, f32_in = case
    x = rnd(B, HW, C, seed=45, scale=2.0, dtype=torch.float32 if f32_in else torch.float16)
    x = x + 0.5
    gamma, beta = rnd(C, seed=46, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=47, dtype=torch.float32) * 0.1
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    got = o.groupnorm_nhwc(x.to(DEV), gamma.to(DEV), beta.to(DEV), 32, 1e-5, silu)
    check_close(results_log, f"groupnorm_{B}_{HW}_{C}_{int(f32_in)}_{int(silu)}", got, ref, floor=2e-3)
    api.lb_groupnorm_set_fused(0)
    try:
        two = o.groupnorm_nhwc(x.to(DEV), gamma.to(DEV), beta.to(DEV), 32, 1e-5, silu)
    finally:
        api.lb_groupnorm_set_fused(1)
    assert float((got.float() - two.float()).abs().max()) <= 2.0 ** -9 * max(1.0, float(two.float().abs().max()))


@pytest.mark.parametrize("shape", [(1024, 640), (256, 1280), (77, 2048), (5, 64), (4352, 1280), (131, 768), (9, 1032)])
def test_layernorm(shape, results_log):
    """fp32 torch reference; the round-6 kernel (every load of a row in flight, permlane / DPP reductions) must also equal the
    round-1 kernel bit for bit (same per-lane summation order, same butterfly)."""
    from latentblending_amd.hip.lib import api
    o = ops()
    M, C = shape
    x = rnd(M, C, seed=48, scale=3.0) + 1
    gamma, beta = rnd(C, seed=49, dtype=torch.float32) * 0.1 + 1, rnd(C, seed=50, dtype=torch.float32) * 0.1
    ref = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    got = o.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV))
    check_close(results_log, f"layernorm_{M}_{C}", got, ref, floor=2e-3)
    api.lb_layernorm_set_form(0)
    try:
        old = o.layernorm(x.to(DEV), gamma.to(DEV), beta.to(DEV))
    finally:
        api.lb_layernorm_set_form(1)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), old.view(torch.int16))


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 11])
@pytest.mark.parametrize("case", [(512, 1280, 1280, False), (4352, 3840, 1280, False), (1024, 5120, 640, True),
                                  (4352, 10240, 1280, True), (300, 192, 64, False)])
def test_gemm_layernorm_fused(case, tile, results_log):
    """LB_GEMM_LN_A: LayerNorm folded into the consuming GEMM (row statistics from the A fragments, affine fix in the
    epilogue) vs torch LayerNorm -> Linear (-> GEGLU) in fp32, on rows with a large common offset (mean >> std is the
    hard case of the E[x^2] - E[x]^2 form) and non-trivial gamma / beta."""
    o, l = ops(), lib()
    M, N, K, geglu = case
    x = rnd(M, K, seed=81) * 1.5 + rnd(M, 1, seed=82) * 4.0          # per-row offsets up to ~3 sigma of the row spread
    w = rnd(N, K, seed=83, scale=K ** -0.5)
    b = rnd(N, seed=84, dtype=torch.float32)
    gamma = 1.0 + 0.2 * rnd(K, seed=85, dtype=torch.float32)
    beta = 0.1 * rnd(K, seed=86, dtype=torch.float32)
    y = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5)
    ref = y @ w.float().t() + b
    if geglu:
        h, gt = ref.chunk(2, dim=-1)
        ref = h * F.gelu(gt)
    wf, colsum, b2 = o.fold_layernorm(w, b, gamma, beta)
    l.api.lb_gemm_set_tuning(tile, 0)
    try:
        got = o.gemm(x.to(DEV), wf.to(DEV), bias=b2.to(DEV), flags=l.GEMM_GEGLU if geglu else 0,
                     ln=(colsum.to(DEV), 1e-5))
        again = o.gemm(x.to(DEV), wf.to(DEV), bias=b2.to(DEV), flags=l.GEMM_GEGLU if geglu else 0,
                       ln=(colsum.to(DEV), 1e-5))
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
    check_close(results_log, f"gemm_ln_fused_{'_'.join(map(str, case))}_tile{tile}", got, ref, rel=3e-3, frac=2 ** -7)
    assert torch.equal(got, again)


# ------------------------------------------------------------------ attention ----------------
@pytest.mark.parametrize("case", [(1, 10, 1024, 1024, 1024), (2, 20, 256, 256, 256), (2, 5, 100, 80, 77),
                                  (1, 2, 64, 64, 64), (1, 10, 4096, 80, 77), (3, 4, 200, 200, 200)])
def test_attention_d64(case, results_log):
    o = ops()
    B, H, Sq, Skv, valid = case
    C = H * 64
    q, k, v = rnd(B, Sq, C, seed=51), rnd(B, Skv, C, seed=52), rnd(B, Skv, C, seed=53)
    ref = R.attention(q.float(), k.float()[:, :valid], v.float()[:, :valid], H)
    got = o.attention_d64(q.reshape(B * Sq, C).to(DEV), k.reshape(B * Skv, C).to(DEV), v.reshape(B * Skv, C).to(DEV),
                          B, H, Sq, Skv, valid)
    check_close(results_log, f"attn_{'_'.join(map(str, case))}", got.reshape(B, Sq, C), ref, floor=2e-3)


# (49 / 50 = bit 5: the 5-stage-ring A/B form added at the end of round 3 without a GPU run: opt in with LB_TEST_EXPERIMENTAL=1)
# (bit 6 = the former two-stage form of the one-tile kernel, bit 7 = 8-byte output stores instead of the paired 16-byte ones,
#  bit 8 = the streaming kernel of rounds 1-5; without it streamed shapes run attn_fwd_d64_stream_kernel, round 6;
#  bit 9 = the 8-wave ping-pong form attn_fwd_d64_pp_kernel, round 6; bit 11 = the block order of rounds 1-5)
@pytest.mark.parametrize("force", [1, 2, 17, 18, 65, 66, 129, 130, 257, 258, 273, 513, 514, 529, 530, 2049, 2066] + ([49, 50] if os.environ.get("LB_TEST_EXPERIMENTAL") == "1" else []))
@pytest.mark.parametrize("case", [(2, 3, 300, 300, 300), (2, 2, 130, 80, 77), (1, 2, 70, 96, 90), (1, 1, 16, 8, 5), (3, 7, 200, 200, 200)])
def test_attention_d64_variants(case, force, results_log):
    """Every kernel variant (1 / 2 query groups per wave, single 96-key tile / streamed 64-key tiles) on ragged shapes,
    with Q, K and V read as column slices of ONE fused [tokens][3C] buffer (the UNet's layout)."""
    o, l = ops(), lib()
    B, H, Sq, Skv, valid = case
    C = H * 64
    q, k, v = rnd(B, Sq, C, seed=61), rnd(B, Skv, C, seed=62), rnd(B, Skv, C, seed=63)
    ref = R.attention(q.float(), k.float()[:, :valid], v.float()[:, :valid], H)
    if Sq == Skv:
        qkv = torch.cat([q, k, v], dim=-1).reshape(B * Sq, 3 * C).to(DEV)
        qd, kd, vd = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        kv = torch.cat([k, v], dim=-1).reshape(B * Skv, 2 * C).to(DEV)
        qd, kd, vd = q.reshape(B * Sq, C).to(DEV), kv[:, :C], kv[:, C:]
    l.api.lb_attn_set_tuning(force)
    try:
        got = o.attention_d64(qd, kd, vd, B, H, Sq, Skv, valid)
    finally:
        l.api.lb_attn_set_tuning(0)
    check_close(results_log, f"attn_f{force}_{'_'.join(map(str, case))}", got.reshape(B, Sq, C), ref, floor=2e-3)
    if force & 128:                 # the paired 16-byte stores (default) hold the same values (the 8-byte form rounds two of four
        l.api.lb_attn_set_tuning(force & ~128)     # halves through v_fma_mixlo_f16, i.e. once instead of twice: <= 1 fp16 ulp apart)
        try:
            wide = o.attention_d64(qd, kd, vd, B, H, Sq, Skv, valid)
        finally:
            l.api.lb_attn_set_tuning(0)
        assert float((wide.float() - got.float()).abs().max()) <= 2.0 ** -10 * max(1.0, float(got.float().abs().max()))


@pytest.mark.parametrize("force", [0, 1, 2, 17, 513, 530])
@pytest.mark.parametrize("case", [(2, 12, 77), (1, 3, 200), (2, 2, 64)])
def test_attention_causal(case, force, results_log):
    """Causal mask of the CLIP text towers (key k visible to query q iff k <= q), single-tile and streamed forms."""
    o, l = ops(), lib()
    B, H, S = case
    C = H * 64
    q, k, v = rnd(B, S, C, seed=64), rnd(B, S, C, seed=65), rnd(B, S, C, seed=66)
    qh, kh, vh = [t.float().view(B, S, H, 64).transpose(1, 2) for t in (q, k, v)]
    ref = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(1, 2).reshape(B, S, C)
    l.api.lb_attn_set_tuning(force)
    try:
        got = o.attention_d64(q.reshape(B * S, C).to(DEV), k.reshape(B * S, C).to(DEV), v.reshape(B * S, C).to(DEV), B, H, S, S,
                              causal=True)
    finally:
        l.api.lb_attn_set_tuning(0)
    check_close(results_log, f"attn_causal_f{force}_{'_'.join(map(str, case))}", got.reshape(B, S, C), ref, floor=2e-3)


def test_gemm_gelu_epilogues(results_log):
    """Non-gated activations of the CLIP MLPs in the GEMM epilogue: quick-GELU (x sigmoid(1.702 x)) and erf-GELU."""
    o, l = ops(), lib()
    x, w, b = rnd(154, 768, seed=87), rnd(3072, 768, seed=88, scale=768 ** -0.5), rnd(3072, seed=89, dtype=torch.float32)
    y = x.float() @ w.float().t() + b
    got_q = o.gemm(x.to(DEV), w.to(DEV), bias=b.to(DEV), flags=l.GEMM_QUICK_GELU)
    check_close(results_log, "gemm_quick_gelu", got_q, y * torch.sigmoid(1.702 * y))
    got_g = o.gemm(x.to(DEV), w.to(DEV), bias=b.to(DEV), flags=l.GEMM_GELU)
    check_close(results_log, "gemm_gelu_erf", got_g, F.gelu(y))


@pytest.mark.parametrize("force", [0, 256, 513])
def test_attention_spiked_scores(force, results_log):
    """Force large running-max jumps between KV tiles (online-softmax rescale path): streaming kernel (round 6), the rounds 1-5
    kernel and the ping-pong form."""
    o, l = ops(), lib()
    B, H, S = 1, 2, 256
    C = H * 64
    q, k, v = rnd(B, S, C, seed=54), rnd(B, S, C, seed=55), rnd(B, S, C, seed=56)
    k[0, 200] = q[0, 3] * 6.0          # one key far above the rest, in the last tile
    k[0, 70] = q[0, 100] * 4.0
    k[0, :64] = -q[0, 5:6] * 5.0       # query 5: every score of the FIRST tile far below zero (the first tile sets the running maximum
    #                                    whatever its sign; the later tiles then raise it by ~2^60)
    ref = R.attention(q.float(), k.float(), v.float(), H)
    l.api.lb_attn_set_tuning(force)
    try:
        got = o.attention_d64(q.reshape(S, C).to(DEV), k.reshape(S, C).to(DEV), v.reshape(S, C).to(DEV), B, H, S, S)
    finally:
        l.api.lb_attn_set_tuning(0)
    check_close(results_log, f"attn_spiked_f{force}", got.reshape(B, S, C), ref, floor=2e-3)


@pytest.mark.parametrize("case", [(2, 1, 1024, 1024, 1024), (1, 1, 4096, 4096, 4096), (1, 2, 200, 200, 200), (2, 1, 100, 77, 70),
                                  (1, 1, 64, 32, 32), (3, 1, 130, 33, 33)])
def test_attention_d512(case, results_log):
    """lb_attn_fwd_d512 (the VAE mid-block attention as one launch) vs fp32 softmax attention: full tiles, ragged query and key
    counts, masked key tail, several heads of 512."""
    o = ops()
    B, H, Sq, Skv, valid = case
    C = H * 512
    q, k, v = rnd(B, Sq, C, seed=151), rnd(B, Skv, C, seed=152), rnd(B, Skv, C, seed=153)
    ref = R.attention(q.float(), k.float()[:, :valid], v.float()[:, :valid], H)
    got = o.attention_d512(q.reshape(B * Sq, C).to(DEV), k.reshape(B * Skv, C).to(DEV), v.reshape(B * Skv, C).to(DEV),
                           B, H, Sq, Skv, valid)
    check_close(results_log, f"attn512_{'_'.join(map(str, case))}", got.reshape(B, Sq, C), ref, floor=2e-3)


def test_attention_d512_fused_qkv_and_spikes(results_log):
    """Operands as column slices of one [tokens][3 * 512] projection (the VAE program's layout), with keys far above the rest
    placed in late tiles (deferred-rescale path) and a query row scaled up (large scores: the fp32 score path)."""
    o = ops()
    B, S, C = 2, 512, 512
    qkv = rnd(B, S, 3 * C, seed=154)
    qkv[0, 300, C:2 * C] = qkv[0, 5, :C] * 3.0
    qkv[1, 40, C:2 * C] = qkv[1, 400, :C] * 2.0
    qkv[1, 7, :C] *= 4.0
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ref = R.attention(q.float(), k.float(), v.float(), 1)
    d = qkv.reshape(B * S, 3 * C).to(DEV)
    got = o.attention_d512(d[:, :C], d[:, C:2 * C], d[:, 2 * C:], B, 1, S, S)
    check_close(results_log, "attn512_fused_qkv_spiked", got.reshape(B, S, C), ref, floor=2e-3)


def test_softmax_rows(results_log):
    o = ops()
    x = rnd(300, 4096, seed=57, scale=4.0)
    ref = torch.softmax(x.float() * 0.3, dim=-1)
    got = o.softmax_rows_(x.to(DEV).clone(), 0.3)
    check_close(results_log, "softmax_rows", got, ref, floor=1e-4)


# ------------------------------------------------------------------ small kernels ------------
def test_small_kernels(results_log):
    o = ops()
    # sinusoid == oracle Timesteps
    vals = torch.tensor([[999.0], [249.0], [1.0]])
    ref = R.sinusoid(vals.reshape(-1), 320)
    got = o.sinusoid(vals.to(DEV), 320)
    check_close(results_log, "sinusoid_t", got, ref, floor=2e-3)
    ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * 2)
    ref = R.sinusoid(ids.reshape(-1), 256).reshape(2, -1)
    out = torch.zeros(2, 1280 + 6 * 256, dtype=torch.float16, device=DEV)
    o.sinusoid(ids.to(DEV), 256, out=out, col_off=1280)
    check_close(results_log, "sinusoid_ids", out[:, 1280:], ref, floor=2e-3)
    assert (out[:, :1280] == 0).all()
    # concat copy
    a, b = rnd(50, 64, seed=58), rnd(50, 32, seed=59)
    dst = torch.zeros(50, 96, dtype=torch.float16, device=DEV)
    o.copy_cols(a.to(DEV), dst, 0)
    o.copy_cols(b.to(DEV), dst, 64)
    assert torch.equal(dst.cpu(), torch.cat([a, b], dim=1))
    # layout converts
    z = rnd(2, 4, 8, 8, seed=60)
    nh = o.nchw_to_nhwc(z.to(DEV), 8, mul=1 / 0.13025)
    ref = (z.float() / 0.13025)
    check_close(results_log, "nchw_to_nhwc", nh[..., :4].permute(0, 3, 1, 2), ref)
    assert (nh[..., 4:] == 0).all()
    assert torch.equal(o.nhwc_to_nchw(o.nchw_to_nhwc(z.to(DEV), 8), 4).cpu(), z)
    # uint8 quantisation == VaeImageProcessor.postprocess
    img = rnd(1, 3, 16, 16, seed=61, dtype=torch.float32)
    x4 = torch.zeros(1, 16, 16, 4)
    x4[..., :3] = img.permute(0, 2, 3, 1)
    got = o.postprocess_u8(x4.to(DEV)).cpu().numpy()
    assert np.array_equal(got, R.postprocess_u8(img))
    # max-pool
    f = rnd(2, 64, 15, 15, seed=62)
    ref = F.max_pool2d(f.float(), 3, 2).permute(0, 2, 3, 1)
    got = o.maxpool3s2(f.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert torch.equal(got.cpu().float(), ref)


# ------------------------------------------------------------------ direct-to-LDS GEMM variant
@pytest.mark.parametrize("stages", [2, 3, 4])
@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 7, 10, 11])
def test_gemm_glds_variant(tile, stages, results_log):
    """gemm_glds.hip (global_load_lds staging, S-stage LDS ring) against the same references.  Tile 10 (round 6) = the 192x128 tile
    as 8 waves of 48 x 64 (three 16-row MFMA tiles per wave)."""
    if tile in (7, 10, 11) and stages != 3:
        pytest.skip("the 192x128 tiles and the two-K-group 64x64 tile have one ring depth (3)")
    o, l = ops(), lib()
    l.api.lb_gemm_set_variant(1, stages)
    l.api.lb_gemm_set_tuning(tile, 0)
    try:
        for (M, N, K) in [(256, 1280, 1280), (100, 64, 72), (333, 132, 200), (2048, 640, 2560), (77, 1280, 2048)]:
            A, W = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=K ** -0.5)
            bias, res = rnd(N, seed=73, dtype=torch.float32), rnd(M, N, seed=74)
            ref = A.float() @ W.float().t() + bias + res.float()
            got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), residual=res.to(DEV))
            check_close(results_log, f"glds_gemm_{M}x{N}x{K}_t{tile}s{stages}", got, ref)
        # GEGLU
        M, C = 300, 640
        A, W = rnd(M, C, seed=75), rnd(8 * C, C, seed=76, scale=C ** -0.5)
        bias = rnd(8 * C, seed=77, dtype=torch.float32, scale=0.1)
        h, gate = (A.float() @ W.float().t() + bias).chunk(2, dim=-1)
        got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), flags=l.GEMM_GEGLU)
        check_close(results_log, f"glds_geglu_t{tile}s{stages}", got, h * F.gelu(gate))
        # split-K
        l.api.lb_gemm_set_tuning(tile, 5)
        A, W = rnd(256, 5120, seed=78), rnd(1280, 5120, seed=79, scale=5120 ** -0.5)
        got = o.gemm(A.to(DEV), W.to(DEV))
        check_close(results_log, f"glds_splitk_t{tile}s{stages}", got, A.float() @ W.float().t())
        l.api.lb_gemm_set_tuning(tile, 0)
        # convolutions incl. padding / stride / upsample / tiny Cin
        for case in CONV_CASES:
            B, H, Wd, Cin, Cout, k, st, pad, ups = case
            x = rnd(B, Cin, H, Wd, seed=80)
            w = rnd(Cout, Cin, k, k, seed=81, scale=(Cin * k * k) ** -0.5)
            b = rnd(Cout, seed=82, dtype=torch.float32)
            xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
            ref = F.conv2d(xin, w.float(), b, stride=st, padding=pad).permute(0, 2, 3, 1)
            cin_p, cout_p = (Cin + 7) // 8 * 8, (Cout + 3) // 4 * 4
            xn = torch.zeros(B, H, Wd, cin_p, dtype=torch.float16)
            xn[..., :Cin] = x.permute(0, 2, 3, 1)
            wp = torch.zeros(cout_p, k * k * cin_p, dtype=torch.float16)
            wp[:Cout] = o.pack_conv_weight(w, cin_p)
            bp = torch.zeros(cout_p, dtype=torch.float32)
            bp[:Cout] = b
            got = o.gemm(xn.to(DEV), wp.to(DEV), bias=bp.to(DEV), conv=dict(KH=k, KW=k, stride=st, pad=pad, ups=ups))
            check_close(results_log, f"glds_conv_{'_'.join(map(str, case))}_t{tile}s{stages}", got[..., :Cout], ref)
    finally:
        l.api.lb_gemm_set_variant(-1, 0)
        l.api.lb_gemm_set_tuning(0, 0)


@pytest.mark.parametrize("variant", [0, 1])
def test_subpixel_upsample_conv(variant, results_log):
    """upsample(nearest 2x) + conv3x3 computed as four 2x2 sub-pixel convs scattered into the output."""
    o, l = ops(), lib()
    import ctypes as C
    B, H, Wd, Cc = 2, 16, 12, 64
    x, w = rnd(B, Cc, H, Wd, seed=90), rnd(Cc, Cc, 3, 3, seed=91, scale=(9 * Cc) ** -0.5)
    bias = rnd(Cc, seed=92, dtype=torch.float32)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.zeros(B, 2 * H, 2 * Wd, Cc, dtype=torch.float16, device=DEV)
    l.api.lb_gemm_set_variant(variant, 0)
    try:
        for (py, px), k in o.subpixel_upsample_weights(w).items():
            o.gemm(xn, k.to(DEV), bias=bias.to(DEV), out=out, conv=dict(KH=2, KW=2, stride=1, pad=0, parity=(py, px)))
    finally:
        l.api.lb_gemm_set_variant(1, 0)
    check_close(results_log, f"subpixel_upconv_v{variant}", out, ref, rel=3e-3)


@pytest.mark.parametrize("B,H,Wd,Cin,Cout", [(2, 16, 16, 64, 64), (1, 8, 32, 128, 320), (3, 32, 32, 192, 128), (2, 16, 48, 64, 200)])
def test_subpixel_upsample_conv_one_launch(B, H, Wd, Cin, Cout, results_log):
    """The same upsample+conv as ONE launch of the halo-tile kernel's 2x2 form (scatter = 2, stacked weights): must equal
    both the fp32 torch reference and, bit for bit in fp16, nothing less than the four-launch implicit-GEMM form's tolerance."""
    o, l = ops(), lib()
    x, w = rnd(B, Cin, H, Wd, seed=190), rnd(Cout, Cin, 3, 3, seed=191, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=192, dtype=torch.float32)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    subs = o.subpixel_upsample_weights(w)
    w4 = torch.stack([subs[(0, 0)], subs[(0, 1)], subs[(1, 0)], subs[(1, 1)]]).to(DEV).contiguous()
    out = torch.full((B, 2 * H, 2 * Wd, Cout), float("nan"), dtype=torch.float16, device=DEV)
    o.gemm(xn, w4[0], bias=bias.to(DEV), out=out, conv=dict(KH=2, KW=2, stride=1, pad=0, parity="all"))
    assert torch.isfinite(out).all(), "one-launch sub-pixel conv left output pixels unwritten"
    check_close(results_log, f"subpixel_upconv_halo_{B}x{H}x{Wd}x{Cin}x{Cout}", out, ref, rel=3e-3)
    four = torch.zeros_like(out)
    for (py, px), k in subs.items():
        o.gemm(xn, k.to(DEV), bias=bias.to(DEV), out=four, conv=dict(KH=2, KW=2, stride=1, pad=0, parity=(py, px)))
    assert (out.float() - four.float()).abs().max().item() <= 2e-2 * ref.abs().max().item()


# ------------------------------------------------------------------ halo-tile 3x3 conv
@pytest.mark.parametrize("case", [(2, 32, 32, 64, 128), (1, 64, 64, 128, 320), (3, 16, 16, 192, 132), (1, 8, 96, 64, 64),
                                  (2, 48, 16, 128, 256)])
def test_conv3x3_halo_against_conv2d(case, results_log):
    """3x3 / stride 1 / pad 1 conv from the LDS-resident halo tile vs F.conv2d, incl. bias + residual, ragged N,
    TW = 32 and TW = 16 tilings, multi-chunk Cin, image borders on every side of a tile."""
    o, l = ops(), lib()
    B, H, Wd, Cin, Cout = case
    x = rnd(B, Cin, H, Wd, seed=90)
    w = rnd(Cout, Cin, 3, 3, seed=91, scale=(Cin * 9) ** -0.5)
    b = rnd(Cout, seed=92, dtype=torch.float32)
    res = rnd(B, H, Wd, Cout, seed=93)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1) + res.float()
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = o.pack_conv_weight(w, Cin)
    got = o.gemm(xn.to(DEV), wp.to(DEV), bias=b.to(DEV), residual=res.to(DEV),
                 conv=dict(KH=3, KW=3, stride=1, pad=1, halo=True))
    check_close(results_log, f"halo_conv_{'_'.join(map(str, case))}", got, ref)
    # and through the lb_gemm_f16 router
    l.api.lb_gemm_set_halo(2)
    try:
        got2 = o.gemm(xn.to(DEV), wp.to(DEV), bias=b.to(DEV), residual=res.to(DEV), conv=dict(KH=3, KW=3, stride=1, pad=1))
    finally:
        l.api.lb_gemm_set_halo(1)
    check_close(results_log, f"halo_conv_routed_{'_'.join(map(str, case))}", got2, ref)


@pytest.mark.parametrize("case", [(2, 32, 128, 3, True), (1, 64, 320, 4, False), (3, 16, 64, 7, True)])
def test_conv3x3_narrow_output(case, results_log):
    """conv3_narrow.hip: the conv_out layers (VAE 128 -> 3 with fp32 output, UNet 320 -> 4 fp16) against F.conv2d; the
    launcher must route them there by itself (lb_gemm_plan tile code 8)."""
    import ctypes as C
    o, l = ops(), lib()
    B, H, Cin, Cout, f32 = case
    x = rnd(B, Cin, H, H, seed=201)
    w = rnd(Cout, Cin, 3, 3, seed=202, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=203, dtype=torch.float32)
    ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1)
    cout_p = (Cout + 3) // 4 * 4
    wp = torch.zeros(cout_p, 9 * Cin, dtype=torch.float16)
    wp[:Cout] = o.pack_conv_weight(w, Cin)
    bp = torch.zeros(cout_p, dtype=torch.float32)
    bp[:Cout] = b
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    p = l.LbGemmParams()
    p.M, p.N, p.K, p.conv, p.flags = B * H * H, cout_p, 9 * Cin, 1, (l.GEMM_OUT_F32 if f32 else 0)
    p.Hin = p.Win = p.Hout = p.Wout = H
    p.Cin, p.KH, p.KW, p.stride, p.pad, p.ldx = Cin, 3, 3, 1, 1, Cin
    p.zero_page = o.zero_page(DEV).data_ptr()
    t = C.c_int()
    l.api.lb_gemm_plan(C.byref(p), C.byref(t), None, None)
    assert t.value == 8
    got = o.gemm(xn, wp.to(DEV), bias=bp.to(DEV), flags=l.GEMM_OUT_F32 if f32 else 0, conv=dict(KH=3, KW=3, stride=1, pad=1))
    assert got.dtype == (torch.float32 if f32 else torch.float16)
    check_close(results_log, f"conv3x3_narrow_{'_'.join(map(str, case))}", got[..., :Cout], ref)
    assert float(got[..., Cout:].abs().max()) == 0 if cout_p > Cout else True


@pytest.mark.parametrize("case", [(2, 32, 128, 128, False, False), (3, 16, 64, 320, True, False), (2, 32, 256, 128, False, True)])
def test_conv_channel_stats_and_groupnorm_from_them(case, results_log):
    """LB_GEMM_CH_STATS: the halo-tile conv's epilogue leaves, per (64-pixel row block, channel), (sum, sum of squares) of the
    values it STORES (bias, alpha, residual included; fp16-rounded unless the output is fp32); lb_groupnorm_from_stats folds
    them and applies GroupNorm (+SiLU) with ONE pass over the activation.  Checked: the raw statistics against torch on the
    stored tensor, and the normalised output against the two-pass kernel and against torch GroupNorm."""
    o, l = ops(), lib()
    B, H, Cin, Cout, with_res, f32 = case
    x = rnd(B, Cin, H, H, seed=211)
    w = rnd(Cout, Cin, 3, 3, seed=212, scale=(9 * Cin) ** -0.5)
    b = rnd(Cout, seed=213, dtype=torch.float32)
    res = rnd(B, H, H, Cout, seed=214) if with_res else None
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = o.pack_conv_weight(w, Cin).to(DEV)
    kind, tw, items, grid = o.conv_halo_plan(B, H, H, Cin, Cout)
    assert kind == 3
    flags = l.GEMM_OUT_F32 if f32 else 0
    l.api.lb_gemm_set_halo(2)
    try:
        rows = o.conv_ch_stat_rows(B, H, H, Cin, Cout)
        assert rows == items // ((Cout + 127) // 128) // B * 4 > 0
        st = torch.full((Cout, B * rows, 2), float("nan"), dtype=torch.float32, device=DEV)          # channel-major
        with pytest.raises(RuntimeError):                       # a buffer of any other size is refused before anything is launched
            o.gemm(xn, wp, bias=b.to(DEV), flags=flags, conv=dict(KH=3, KW=3, stride=1, pad=1), ch_stats=st[:, :-1])
        y = o.gemm(xn, wp, bias=b.to(DEV), residual=None if res is None else res.to(DEV), flags=flags, alpha=0.5,
                   conv=dict(KH=3, KW=3, stride=1, pad=1), ch_stats=st)
        y_plain = o.gemm(xn, wp, bias=b.to(DEV), residual=None if res is None else res.to(DEV), flags=flags, alpha=0.5,
                         conv=dict(KH=3, KW=3, stride=1, pad=1))
    finally:
        l.api.lb_gemm_set_halo(1)
    assert torch.equal(y, y_plain), "the statistics epilogue must not change what the conv stores"
    assert torch.isfinite(st).all(), "every (row block, channel) slot must be written"
    yf = y.float().reshape(B, H * H, Cout)
    tot = st.reshape(Cout, B, rows, 2).double().sum(dim=2).permute(1, 0, 2).cpu()
    want_s, want_q = yf.double().sum(dim=1).cpu(), (yf.double() ** 2).sum(dim=1).cpu()
    assert torch.allclose(tot[..., 0], want_s, rtol=1e-4, atol=1e-2) and torch.allclose(tot[..., 1], want_q, rtol=1e-4, atol=1e-2)
    gamma, beta = (1 + 0.1 * rnd(Cout, seed=215, dtype=torch.float32)).to(DEV), (0.1 * rnd(Cout, seed=216, dtype=torch.float32)).to(DEV)
    got = o.groupnorm_from_stats(y, gamma, beta, 32, 1e-6, True, st, rows)
    two_pass = o.groupnorm_nhwc(y, gamma, beta, 32, 1e-6, True)
    ref = F.silu(F.group_norm(y.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-6)).permute(0, 2, 3, 1)
    check_close(results_log, f"groupnorm_from_conv_stats_{'_'.join(map(str, case))}", got, ref)
    assert (got.float() - two_pass.float()).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------ ping-pong 256x256 GEMM (gemm_pp.hip, tile code 9)
PP_SHAPES = [(4352, 1280, 1280), (512, 768, 640), (300, 260, 128), (1000, 3840, 64), (256, 256, 192), (4352, 512, 5120)]


@pytest.mark.parametrize("group", [8, 0, 3])
@pytest.mark.parametrize("shape", PP_SHAPES)
def test_gemm_pingpong(shape, group, results_log):
    """gemm_pp.hip against the fp32 reference AND bit for bit against the lock-step tiles (same K order per accumulator);
    group = tile order inside an XCD's run (8 block rows per group = the default, 0 = strips, 3 = a width that does not divide the
    block rows); ragged M / N (rows beyond the edge re-read the last valid row), K = 64 (one K-tile: prologue + drain only), an odd number of K-tiles,
    bias / residual epilogue.  Repeated launches: a staging race would show as run-to-run differences."""
    o, l = ops(), lib()
    M, N, K = shape
    A, W = rnd(M, K, seed=91), rnd(N, K, seed=92, scale=K ** -0.5)
    bias, res = rnd(N, seed=93, dtype=torch.float32), rnd(M, N, seed=94)
    ref = A.float() @ W.float().t() + bias + res.float()
    Ad, Wd, bd, rd = A.to(DEV), W.to(DEV), bias.to(DEV), res.to(DEV)
    l.api.lb_gemm_set_tuning(1, 1)
    try:
        lock_step = o.gemm(Ad, Wd, bias=bd, residual=rd)
        l.api.lb_gemm_set_tuning(9, 1)
        l.api.lb_gemm_pp_set_group(group)
        runs = [o.gemm(Ad, Wd, bias=bd, residual=rd) for _ in range(6)]
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
        l.api.lb_gemm_pp_set_group(8)
    check_close(results_log, f"pp_gemm_{M}x{N}x{K}_group{group}", runs[0], ref)
    assert torch.equal(runs[0], lock_step)
    for r in runs[1:]:
        assert torch.equal(r, runs[0])


def test_gemm_pingpong_geglu_and_splitk(results_log):
    o, l = ops(), lib()
    l.api.lb_gemm_set_tuning(9, 1)
    try:
        for (M, C, mult) in [(4352, 1280, 8), (300, 640, 8), (520, 320, 2)]:
            A, W = rnd(M, C, seed=95), rnd(mult * C, C, seed=96, scale=C ** -0.5)
            bias = rnd(mult * C, seed=97, dtype=torch.float32, scale=0.1)
            h, gate = (A.float() @ W.float().t() + bias).chunk(2, dim=-1)
            got = o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), flags=l.GEMM_GEGLU)
            assert got.shape == (M, mult * C // 2)
            check_close(results_log, f"pp_geglu_{M}x{C}x{mult}", got, h * F.gelu(gate))
            l.api.lb_gemm_set_tuning(1, 1)
            assert torch.equal(o.gemm(A.to(DEV), W.to(DEV), bias=bias.to(DEV), flags=l.GEMM_GEGLU), got)
            l.api.lb_gemm_set_tuning(9, 1)
        l.api.lb_gemm_set_tuning(9, 3)                       # split-K slabs through the ping-pong loop (uneven slices: 20 K-tiles / 3)
        A, W = rnd(512, 1280, seed=98), rnd(768, 1280, seed=99, scale=1280 ** -0.5)
        got = o.gemm(A.to(DEV), W.to(DEV))
        check_close(results_log, "pp_splitk3", got, A.float() @ W.float().t())
    finally:
        l.api.lb_gemm_set_tuning(0, 0)


@pytest.mark.parametrize("shape", [(4352, 1280, 1280), (1000, 384, 256), (300, 260, 128)])
def test_gemm_192x128_eight_waves_bit_identical_to_six(shape):
    """Round 6: the 192x128 tile as 8 waves of 48 x 64 (tile code 10, what lb_gemm_plan now takes) accumulates every output in the
    same K order as the 6-wave form of rounds 2-5 (tile code 7): identical bits, with bias + residual and as a GEGLU."""
    o, l = ops(), lib()
    M, N, K = shape
    A, W = rnd(M, K, seed=311).to(DEV), rnd(N, K, seed=312, scale=K ** -0.5).to(DEV)
    bias, res = rnd(N, seed=313, dtype=torch.float32).to(DEV), rnd(M, N, seed=314).to(DEV)
    outs = {}
    try:
        for tile in (7, 10):
            l.api.lb_gemm_set_tuning(tile, 1)
            outs[tile] = (o.gemm(A, W, bias=bias, residual=res), o.gemm(A, W[: N // 8 * 8], bias=bias[: N // 8 * 8], flags=l.GEMM_GEGLU))
    finally:
        l.api.lb_gemm_set_tuning(0, 0)
    assert torch.equal(outs[7][0], outs[10][0]) and torch.equal(outs[7][1], outs[10][1])


# ------------------------------------------------------------------ one-round-trip tile epilogue (lb_gemm.h, round 5)
def _lean_ab(fn):
    """fn() with the per-row epilogue everywhere, then with the one-round-trip form: both results."""
    l = lib()
    l.api.lb_gemm_set_lean_epilogue(0)
    try:
        a = fn()
    finally:
        l.api.lb_gemm_set_lean_epilogue(1)
    return a, fn()


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 7, 9, 10])
@pytest.mark.parametrize("shape", [(4352, 1280, 640), (1000, 384, 256), (300, 260, 128), (512, 768, 192)])
def test_lean_epilogue_gemm_bit_identical(shape, tile, results_log):
    """Every tile family, whole and ragged wave tiles (M = 1000 / 300: masked rows; N = 260 / 384: wave tiles that overhang N fall back
    to the general form), alpha, bias, fp16 residual - also IN PLACE (residual == output), and a time-embedding row vector whose
    samples are 250 rows long (wave tiles inside one sample take the lean form, tiles across a boundary the general one): the
    stored bits must not depend on the epilogue form, and must match the fp32 reference."""
    o, l = ops(), lib()
    M, N, K = shape
    A, W = rnd(M, K, seed=301).to(DEV), rnd(N, K, seed=302, scale=K ** -0.5).to(DEV)
    bias, res = rnd(N, seed=303, dtype=torch.float32).to(DEV), rnd(M, N, seed=304).to(DEV)
    rpb = 250 if M % 250 == 0 else M // 4
    rv = rnd(M // rpb, N, seed=305).to(DEV)
    l.api.lb_gemm_set_tuning(tile, 1 if tile else 0)
    try:
        a, b = _lean_ab(lambda: o.gemm(A, W, bias=bias, residual=res, alpha=0.5))
        assert torch.equal(a, b)
        check_close(results_log, f"lean_epi_gemm_{M}x{N}x{K}_t{tile}", b, 0.5 * (A.float() @ W.float().t()) + bias + res.float())
        a, b = _lean_ab(lambda: o.gemm(A, W, bias=bias, rowvec=rv, rows_per_batch=rpb))
        assert torch.equal(a, b)
        check_close(results_log, f"lean_epi_gemm_rowvec_{M}x{N}x{K}_t{tile}", b,
                    A.float() @ W.float().t() + bias + rv.float().repeat_interleave(rpb, 0))
        a, b = _lean_ab(lambda: o.gemm(A, W))
        assert torch.equal(a, b)

        def in_place():
            buf = res.clone()
            o.gemm(A, W, bias=bias, residual=buf, out=buf)
            return buf
        a, b = _lean_ab(in_place)
        assert torch.equal(a, b)
        assert torch.equal(b, o.gemm(A, W, bias=bias, residual=res))
    finally:
        l.api.lb_gemm_set_tuning(0, 0)


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 128, 32), (1, 64, 64, 128, 320, 32), (3, 16, 16, 192, 132, 16), (2, 48, 16, 128, 256, 16)])
def test_lean_epilogue_halo_conv_bit_identical(case, results_log):
    """The persistent halo conv (TW = 32 and 16, several tiles per block, N = 132 / 320: a last channel block that overhangs N takes
    the general form) with bias + residual, bias + time-embedding vector, alpha, and the channel statistics: same stored bits and
    same statistics under both epilogue forms."""
    o, l = ops(), lib()
    B, H, Wd, Cin, Cout, tw = case
    x = rnd(B, H, Wd, Cin, seed=311).to(DEV)
    w = rnd(Cout, Cin, 3, 3, seed=312, scale=(Cin * 9) ** -0.5)
    wp = o.pack_conv_weight(w, Cin).to(DEV)
    b = rnd(Cout, seed=313, dtype=torch.float32).to(DEV)
    res, temb = rnd(B, H, Wd, Cout, seed=314).to(DEV), rnd(B, Cout, seed=315).to(DEV)
    conv = dict(KH=3, KW=3, stride=1, pad=1, halo=True)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w.float(), b.cpu(), padding=1).permute(0, 2, 3, 1)
    a, c = _lean_ab(lambda: o.gemm(x, wp, bias=b, residual=res, alpha=0.25, conv=conv))
    assert torch.equal(a, c)
    check_close(results_log, f"lean_epi_halo_res_{'_'.join(map(str, case))}", c, 0.25 * (ref - b.cpu()) + b.cpu() + res.float().cpu())
    a, c = _lean_ab(lambda: o.gemm(x, wp, bias=b, rowvec=temb, rows_per_batch=H * Wd, conv=conv))
    assert torch.equal(a, c)
    check_close(results_log, f"lean_epi_halo_temb_{'_'.join(map(str, case))}", c, ref + temb.float().cpu()[:, None, None, :])
    if Cout % 128 == 0:
        l.api.lb_gemm_set_halo(2)
        try:
            rows = o.conv_ch_stat_rows(B, H, Wd, Cin, Cout)

            def with_stats():
                st = torch.full((Cout, B * rows, 2), float("nan"), dtype=torch.float32, device=DEV)
                y = o.gemm(x, wp, bias=b, residual=res, conv=dict(KH=3, KW=3, stride=1, pad=1), ch_stats=st)
                return torch.cat([y.float().reshape(-1), st.reshape(-1)])
            a, c = _lean_ab(with_stats)
        finally:
            l.api.lb_gemm_set_halo(1)
        assert torch.isfinite(c).all() and torch.equal(a, c)


@pytest.mark.parametrize("B,H,Wd,Cin,Cout", [(2, 16, 32, 64, 128), (1, 32, 32, 128, 256), (2, 16, 16, 64, 200)])
def test_lean_epilogue_subpixel_upconv_bit_identical(B, H, Wd, Cin, Cout, results_log):
    """The 2x2 sub-pixel form (all four parities in one launch): the lean epilogue computes the scattered output rows itself."""
    o = ops()
    x, w = rnd(B, Cin, H, Wd, seed=321), rnd(Cout, Cin, 3, 3, seed=322, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=323, dtype=torch.float32).to(DEV)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), bias.cpu(), padding=1).permute(0, 2, 3, 1)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    subs = o.subpixel_upsample_weights(w)
    w4 = torch.stack([subs[(0, 0)], subs[(0, 1)], subs[(1, 0)], subs[(1, 1)]]).to(DEV).contiguous()

    def run():
        out = torch.full((B, 2 * H, 2 * Wd, Cout), float("nan"), dtype=torch.float16, device=DEV)
        o.gemm(xn, w4[0], bias=bias, out=out, alpha=0.5, conv=dict(KH=2, KW=2, stride=1, pad=0, parity="all"))
        return out
    a, c = _lean_ab(run)
    assert torch.isfinite(c).all() and torch.equal(a, c)
    check_close(results_log, f"lean_epi_upconv_{B}x{H}x{Wd}x{Cin}x{Cout}", c, 0.5 * (ref - bias.cpu()) + bias.cpu(), rel=3e-3)
