"""Parity of the EXACT launch programs the benchmark times (BASELINE configs[1]: SDXL-Turbo 512^2, 4 steps,
15 mid branches) against the CPU fp32 oracle with the same seeded weights: full-size UNet at B = 2 and B = 17,
full-size VAE decoder at B = 17, one whole cfg-2 transition (17 frames) end to end, the VAE precision plan on
large activations, the reference's slerp golden vectors fed straight to the device kernel, and the
real-checkpoint path (HF-layout safetensors -> from_safetensors -> diffusers facade).

Tolerances: UNet forward rel-L2 <= 1e-2 per sample; frames mean |du8| <= 2 and >= 99 % of pixels
within +-4; tree (fractions, order, injection indices) identical; slerp <= 1 fp16 ulp vs the reference's bits.
The oracle is only the checker here (tests may import oracle/).
"""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

from oracle import pipe as OP  # noqa: E402  (checker only)
from oracle import sdxl_ref as R  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def native():
    import latentblending_amd.native as n
    return n


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _threads():
    torch.set_num_threads(min(os.cpu_count() or 1, 16))


@pytest.fixture(scope="module")
def full_models():
    """Full SDXL UNet (2.57 B params, seed 0) and VAE decoder (seed 1): fp32 oracle weights on the host and
    the packed native modules on the GPU, shared by every test of this module."""
    n = native()
    _threads()
    ucfg, vcfg = R.UNetCfg(sample_size=64), R.VAECfg()
    uw = R.make_weights(R.unet_spec(ucfg), 0)
    vw = R.make_weights(R.vae_decoder_spec(vcfg), 1)
    unet = n.NativeUNet(n.UNetConfig(**dataclasses.asdict(ucfg)), n.DictProvider(uw), DEV)
    vae = n.NativeVAEDecoder(n.VAEConfig(**dataclasses.asdict(vcfg)), n.DictProvider(vw), DEV)
    return dict(ucfg=ucfg, vcfg=vcfg, uw=uw, vw=vw, unet=unet, vae=vae)


# ------------------------------------------------------------------ UNet at the benchmark's batch sizes
@pytest.mark.parametrize("B", [2, 17])
def test_full_unet_batched_matches_oracle(B, full_models, results_log):
    """unet_B2_L64 / unet_B17_L64: the programs (tile policy, split-K, halo convs) the timed transition replays;
    every sample has its own latent / context / pooled embedding; eager and hipGraph replays must agree bitwise."""
    m = full_models
    g = torch.Generator().manual_seed(1000 + B)
    x = torch.randn(B, 4, 64, 64, generator=g).half()
    ctx = torch.randn(B, 77, 2048, generator=g).half()
    te = torch.randn(B, 1280, generator=g).half()
    ids = torch.tensor([[512.0, 512.0, 0.0, 0.0, 512.0, 512.0]] * B)
    t = 749.0 if B == 17 else 999.0
    prog = m["unet"].build(B, 64)
    prog.set_conditioning(ctx.to(DEV), te.to(DEV), ids.to(DEV))
    got = prog.forward(x.to(DEV), torch.full((B,), t)).clone()
    prog.enable_graphs()
    got_graph = prog.forward(x.to(DEV), torch.full((B,), t)).clone()
    assert torch.equal(got, got_graph), "hipGraph replay differs from the eager replay"
    worst, refmax = 0.0, 0.0
    for b in range(B):                                  # oracle sample by sample (bounded host memory)
        ref = R.unet_forward(m["ucfg"], m["uw"], x[b:b + 1], torch.tensor(t), ctx[b:b + 1], te[b:b + 1], ids[b:b + 1])
        worst = max(worst, rel_l2(got[b:b + 1], ref))
        refmax = max(refmax, float(ref.abs().max()))
    results_log[f"unet_full_B{B}_L64_worst_rel_l2"] = worst
    print(f"[parity] FULL SDXL UNet B={B} 512^2: worst per-sample rel_l2={worst:.3e} max|ref|={refmax:.3f} "
          f"ops={prog.prog_step.num_ops}")
    assert torch.isfinite(got).all() and worst <= 1e-2
    del prog
    torch.cuda.empty_cache()


def test_full_vae_B17_matches_oracle(full_models, results_log):
    """vae_B17_L64: the decode batch of the timed transition (fp16 x 2^-4 residual stream, halo convs)."""
    m = full_models
    g = torch.Generator().manual_seed(77)
    z = torch.randn(17, 4, 64, 64, generator=g).half()
    prog = m["vae"].build(17, 64)
    got_u8 = prog.decode(z.to(DEV)).cpu().numpy()
    got_f = prog.image_f32[..., :3].permute(0, 3, 1, 2).float().cpu()
    worst_rel, dsum, within, n = 0.0, 0.0, 0.0, 0
    for b in range(17):
        ref_img = R.vae_decode(m["vcfg"], m["vw"], z[b:b + 1].float() / m["vcfg"].scaling_factor)
        ref_u8 = R.postprocess_u8(ref_img)
        worst_rel = max(worst_rel, rel_l2(got_f[b:b + 1], ref_img))
        d = np.abs(got_u8[b:b + 1].astype(np.int32) - ref_u8.astype(np.int32))
        dsum += float(d.sum()); within += float((d <= 4).sum()); n += d.size
    results_log["vae_full_B17_L64"] = {"worst_rel_l2": worst_rel, "mean_abs_u8": dsum / n, "frac_within_4": within / n}
    print(f"[parity] FULL SDXL VAE B=17 512^2: worst rel_l2={worst_rel:.3e} mean|du8|={dsum / n:.3f} within4={within / n:.4f}")
    assert worst_rel <= 1e-2 and dsum / n <= 2 and within / n >= 0.99
    del prog
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ BASELINE configs[2] shapes: SDXL base, 1024^2
def test_full_unet_and_vae_at_1024_match_oracle(full_models, results_log):
    """SDXL base 1.0 at 1024x1024 (latent 128x128): the CFG batch of one denoising step (B = 2: [uncond | text]) through
    the full UNet - self-attention over 4096 tokens, 16x the attention work of the 512^2 programs - and one VAE decode
    (mid-block attention over 16384 tokens), against the fp32 oracle."""
    m = full_models
    g = torch.Generator().manual_seed(3128)
    B, L = 2, 128
    x = torch.randn(B, 4, L, L, generator=g).half()
    ctx = torch.randn(B, 77, 2048, generator=g).half()
    te = torch.randn(B, 1280, generator=g).half()                    # (sample 0 plays the negative branch: the encoded "",
    #                                                                   a dense embedding like any other - NOT zeros)
    ids = torch.tensor([[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]] * B)
    prog = m["unet"].build(B, L)
    prog.set_conditioning(ctx.to(DEV), te.to(DEV), ids.to(DEV))
    got = prog.forward(x.to(DEV), torch.full((B,), 958.0)).clone()
    worst = 0.0
    for b in range(B):
        ref = R.unet_forward(m["ucfg"], m["uw"], x[b:b + 1], torch.tensor(958.0), ctx[b:b + 1], te[b:b + 1], ids[b:b + 1])
        worst = max(worst, rel_l2(got[b:b + 1], ref))
    results_log["unet_full_B2_L128_worst_rel_l2"] = worst
    print(f"[parity] FULL SDXL UNet B=2 1024^2: worst per-sample rel_l2={worst:.3e}")
    assert torch.isfinite(got).all() and worst <= 1e-2
    del prog
    torch.cuda.empty_cache()
    z = torch.randn(1, 4, L, L, generator=g).half()
    vprog = m["vae"].build(1, L)
    got_u8 = vprog.decode(z.to(DEV)).cpu().numpy()
    ref_img = R.vae_decode(m["vcfg"], m["vw"], z.float() / m["vcfg"].scaling_factor)
    rv = rel_l2(vprog.image_f32[..., :3].permute(0, 3, 1, 2), ref_img)
    d = np.abs(got_u8.astype(np.int32) - R.postprocess_u8(ref_img).astype(np.int32))
    results_log["vae_full_L128"] = {"rel_l2": rv, "mean_abs_u8": float(d.mean()), "frac_within_4": float((d <= 4).mean())}
    print(f"[parity] FULL SDXL VAE 1024^2: rel_l2={rv:.3e} mean|du8|={d.mean():.3f} within4={(d <= 4).mean():.4f}")
    assert rv <= 1e-2 and d.mean() <= 2 and (d <= 4).mean() >= 0.99
    del vprog
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ one whole benchmark transition
def test_cfg2_transition_matches_oracle(full_models, results_log):
    """BASELINE configs[1] end to end: SDXL-Turbo 512^2, 4 steps, 15 mid branches -> 17 frames.  Native engine
    exactly as bench.py runs it (hipGraphs, speculative frontier 16, wavefront-fused anchors) against the same
    engine driving the CPU fp32 oracle pipe strictly sequentially (38 UNet forwards + 17 decodes on the host)."""
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    n, m = native(), full_models
    _threads()
    o = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=m["ucfg"], vae_cfg=m["vcfg"], weights=m["uw"], vae_weights=m["vw"])
    p = n.NativeSDXLPipe(turbo=True, unet_native=m["unet"], vae_native=m["vae"])
    tape = OP.NoiseTape(12345)
    p.scheduler.noise_source = tape
    np.random.seed(0)
    set_backend(R.TorchCpuBackend())
    be_o = BlendingEngine(o, metric=R.OracleLPIPS(7), verbose=False)
    set_backend(None)
    be_p = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=16)
    for be in (be_o, be_p):
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
        be.set_branching(nmb_max_branches=15)
    tape.reset()
    imgs_p = be_p.run_transition(fixed_seeds=[420, 421])
    set_backend(R.TorchCpuBackend())
    try:
        o.noise.reset()
        imgs_o = be_o.run_transition(fixed_seeds=[420, 421])
    finally:
        set_backend(None)
    assert len(imgs_o) == len(imgs_p) == 17
    same_tree = be_o.tree_fracts == be_p.tree_fracts and be_o.tree_idx_injection == be_p.tree_idx_injection
    lat_err = max(rel_l2(a[-1], b[-1]) for a, b in zip(be_p.tree_latents, be_o.tree_latents)) if same_tree else float("nan")
    d = np.stack([np.abs(np.asarray(a).astype(np.int32) - np.asarray(b).astype(np.int32)) for a, b in zip(imgs_p, imgs_o)]) \
        if same_tree else np.zeros(1)
    so, sp = [float(s) for s in be_o.tree_similarities], [float(s) for s in be_p.tree_similarities]
    results_log["transition_cfg2_full"] = {
        "frames": len(imgs_p), "same_tree": bool(same_tree), "final_latent_rel_l2": lat_err,
        "mean_abs_u8": float(d.mean()), "frac_within_4": float((d <= 4).mean()),
        "fracts_native": be_p.tree_fracts, "fracts_oracle": be_o.tree_fracts, "sims_native": sp, "sims_oracle": so,
        "frontier_rounds": be_p.stats.get("frontier_rounds"), "speculation_dropped": be_p.stats.get("speculation_dropped")}
    print(f"[parity] cfg-2 transition (17 frames, 512^2): same_tree={same_tree} latent rel_l2={lat_err:.3e} "
          f"mean|du8|={d.mean():.3f} within4={(d <= 4).mean():.4f}")
    assert same_tree, (be_o.tree_fracts, be_p.tree_fracts)
    assert np.allclose(sp, so, rtol=5e-2, atol=1e-3)
    assert lat_err <= 3e-2
    assert d.mean() <= 2 and (d <= 4).mean() >= 0.99


# ------------------------------------------------------------------ SDXL base: CFG, multi-level tree, full width
def test_base_multilevel_transition_matches_oracle(full_models, results_log):
    """BASELINE configs[2] in reduced length at FULL WIDTH (the real SDXL channel widths, 2.57 B parameters): SDXL base,
    classifier-free guidance 4.0 (mid-damped, negative prompt = the ENCODED ""), Euler scheduler, 6 steps,
    depth_strength 0.5, nmb_max_branches 6 -> four levels [3, 3, 4, 5] x 1 stem, i.e. parents taken from different
    injection levels (/root/reference/latentblending/blending_engine.py:550-561).  Native engine (hipGraphs, frontier) vs
    the same engine driving the CPU fp32 oracle pipe sequentially: identical tree, frames mean |du8| <= 2.
    Rendered at LB_TEST_BASE_SIDE^2 (default 512: 21 CFG forwards + 6 decodes of the oracle cost ~100 s of host time;
    1024 = the config's own size costs ~7 min; the 1024^2 kernels themselves are covered by
    test_full_unet_and_vae_at_1024_match_oracle)."""
    from latentblending_amd import BlendingEngine
    from latentblending_amd.backend import set_backend
    n, m = native(), full_models
    _threads()
    side = int(os.environ.get("LB_TEST_BASE_SIDE", "512"))
    o = OP.StableDiffusionXLPipeline(turbo=False, unet_cfg=m["ucfg"], vae_cfg=m["vcfg"], weights=m["uw"], vae_weights=m["vw"])
    p = n.NativeSDXLPipe(turbo=False, unet_native=m["unet"], vae_native=m["vae"], allow_synthetic=True)
    np.random.seed(0)
    set_backend(R.TorchCpuBackend())
    be_o = BlendingEngine(o, metric=R.OracleLPIPS(7), verbose=False)
    set_backend(None)
    be_p = BlendingEngine(p, verbose=False, do_compile=True, frontier_width=4)
    for be in (be_o, be_p):
        be.set_dimensions((side, side))
        be.set_num_inference_steps(6)
        be.set_guidance_scale(4.0)
        be.set_branching(depth_strength=0.5, nmb_max_branches=6)
        be.set_prompt1("photo of underwater landscape, fish, und the sea, incredible detail, high resolution")
        be.set_prompt2("rendering of an alien planet, strange plants, strange creatures, surreal")
    assert [int(i) for i in be_p.list_idx_injection] == [3, 3, 4, 5] and be_p.text_embedding1[1] is not None
    assert float(be_p.text_embedding1[1].abs().max()) > 0, "the default negative prompt \"\" is encoded, not zeroed"
    imgs_p = be_p.run_transition(fixed_seeds=[420, 421])
    set_backend(R.TorchCpuBackend())
    try:
        imgs_o = be_o.run_transition(fixed_seeds=[420, 421])
    finally:
        set_backend(None)
    same_tree = be_o.tree_fracts == be_p.tree_fracts and be_o.tree_idx_injection == be_p.tree_idx_injection
    assert len(set(be_p.tree_idx_injection)) >= 3, "the tree must be multi-level"
    lat_err = max(rel_l2(a[-1], b[-1]) for a, b in zip(be_p.tree_latents, be_o.tree_latents)) if same_tree else float("nan")
    d = np.stack([np.abs(np.asarray(a).astype(np.int32) - np.asarray(b).astype(np.int32)) for a, b in zip(imgs_p, imgs_o)]) \
        if same_tree else np.zeros(1)
    results_log["transition_base_multilevel_full_width"] = {
        "side": side, "frames": len(imgs_p), "same_tree": bool(same_tree), "final_latent_rel_l2": lat_err,
        "mean_abs_u8": float(d.mean()), "frac_within_4": float((d <= 4).mean()), "fracts": be_p.tree_fracts,
        "idx_injection": [int(i) for i in be_p.tree_idx_injection],
        "sims_native": [float(s) for s in be_p.tree_similarities], "sims_oracle": [float(s) for s in be_o.tree_similarities]}
    print(f"[parity] SDXL-base multi-level transition at full width ({side}^2, CFG 4.0, 6 steps): same_tree={same_tree} "
          f"idx={be_p.tree_idx_injection} latent rel_l2={lat_err:.3e} mean|du8|={d.mean():.3f} within4={(d <= 4).mean():.4f}")
    assert same_tree, (be_o.tree_fracts, be_p.tree_fracts, be_o.tree_idx_injection, be_p.tree_idx_injection)
    assert lat_err <= 3e-2 and d.mean() <= 2 and (d <= 4).mean() >= 0.99


# ------------------------------------------------------------------ VAE precision plan on large activations
@pytest.mark.parametrize("where", ["conv_in", "late"])
@pytest.mark.parametrize("scaled_stream", [True, False])
def test_vae_large_activation_stream(where, scaled_stream, results_log):
    """The real SDXL VAE overflows fp16 in its residual stream (that is why the reference upcasts it to fp32,
    diffusers_holder.py:129-139).  Synthetic N(0, 1/fan_in) weights never get there, so force it: weights scaled
    by 2^12 / 2^15 put the residual stream at 1e4..1e5 (beyond the fp16 maximum of 65504) either from conv_in on or
    from the middle of the decoder.  Both stream formats (fp16 x 2^-4, fp32) must still match the fp32 oracle."""
    n = native()
    vcfg = R.tiny_vae_cfg()
    vw = R.make_weights(R.vae_decoder_spec(vcfg), 5)
    nm, mul = ("decoder.conv_in", 2.0 ** 12) if where == "conv_in" else ("decoder.up_blocks.1.resnets.1.conv2", 2.0 ** 15)
    vw[nm + ".weight"] = vw[nm + ".weight"] * mul          # (powers of two: still exactly fp16-representable)
    vw[nm + ".bias"] = vw[nm + ".bias"] * mul
    g = torch.Generator().manual_seed(9)
    z = torch.randn(2, 4, 16, 16, generator=g).half()
    taps = {}
    ref_img = R.vae_decode(vcfg, vw, z.float() / vcfg.scaling_factor, taps=taps)
    peak = max(float(t.abs().max()) for t in taps.values())
    cfg = n.VAEConfig(**{**dataclasses.asdict(vcfg), "stream_fp16_scaled": scaled_stream})
    vnet = n.NativeVAEDecoder(cfg, n.DictProvider(vw), DEV)
    prog = vnet.build(2, 16)
    got_u8 = prog.decode(z.to(DEV)).cpu().numpy()
    got = prog.image_f32[..., :3].permute(0, 3, 1, 2)
    r = rel_l2(got, ref_img)
    d = np.abs(got_u8.astype(np.int32) - R.postprocess_u8(ref_img).astype(np.int32))
    key = f"vae_large_act_{where}_{'f16s' if scaled_stream else 'f32'}"
    results_log[key] = {"peak_stream": peak, "rel_l2": r, "mean_abs_u8": float(d.mean()), "frac_within_4": float((d <= 4).mean())}
    print(f"[parity] {key}: peak |stream|={peak:.3e} rel_l2={r:.3e} mean|du8|={d.mean():.3f} within4={(d <= 4).mean():.4f}")
    assert peak >= 1e4, "the test must reach the regime the 2^-4 stream scale exists for"
    assert torch.isfinite(got).all()
    assert r <= 1e-2 and d.mean() <= 2 and (d <= 4).mean() >= 0.99


# ------------------------------------------------------------------ the reference's own slerp vectors on the device
def test_slerp_kernel_matches_reference_golden(results_log):
    """tests/golden/slerp.json holds outputs of the UNCHANGED reference interpolate_spherical
    (latentblending/utils.py:29-71) as bit patterns; feed the same inputs to lb_slerp_pairs_* on the GPU."""
    from latentblending_amd.hip import ops
    with open(os.path.join(ROOT, "tests", "golden", "slerp.json")) as fh:
        cases = json.load(fh)["slerp"]
    DT = {"torch.float16": torch.float16, "torch.float32": torch.float32, "torch.float64": torch.float64}

    def seeded(nn, seed, dtype=torch.float16, scale=1.0):
        gg = torch.Generator().manual_seed(seed)
        return (torch.randn(nn, generator=gg) * scale).to(dtype)

    def ordered(bits):
        b = bits.to(torch.int32)
        return torch.where(b < 0, -32768 - b, b)

    worst, exact, total = 0, 0, 0
    for c in cases:
        dt = DT[c["in_dtype"]]
        if c["name"] == "identical":
            p0 = seeded(c["n"], c["seed0"]); p1 = p0.clone()
        elif c["name"] == "antipodal":
            p0 = seeded(c["n"], c["seed0"]); p1 = -p0
        elif c["name"] == "zero_norm":
            p0 = torch.zeros(c["n"], dtype=torch.float16); p1 = seeded(c["n"], c["seed0"])
        else:
            p0, p1 = seeded(c["n"], c["seed0"], dt, c["scale"]), seeded(c["n"], c["seed1"], dt, c["scale"])
        out = ops.slerp(p0.to(DEV), p1.to(DEV), c["fract"]).cpu()
        assert str(out.dtype) == c["out_dtype"], c["name"]
        if c["nan"]:
            assert torch.isnan(out).all(), c["name"]
        elif out.dtype == torch.float16:
            want = torch.tensor(c["out_bits"], dtype=torch.int16)
            ulp = int((ordered(out.view(torch.int16)) - ordered(want)).abs().max())
            worst = max(worst, ulp)
            exact += int((out.view(torch.int16) == want).sum()); total += want.numel()
            if c["fract"] in (0.0, 1.0):
                assert ulp == 0, (c["name"], c["fract"])
            assert ulp <= 1, (c["name"], c["fract"], ulp)
        else:
            want = torch.tensor(c["out_f32"], dtype=torch.float32)
            assert torch.allclose(out, want, rtol=2e-6, atol=1e-7), c["name"]
    results_log["slerp_gpu_vs_reference_golden"] = {"worst_ulp_f16": worst, "bit_exact_fraction": exact / max(total, 1)}
    print(f"[parity] slerp kernel vs reference golden bits: worst {worst} ulp, {exact}/{total} elements bit-exact")


# ------------------------------------------------------------------ real-checkpoint path
def test_safetensors_checkpoint_roundtrip_through_facade(tmp_path, monkeypatch, results_log):
    """example_single_trans.py:11 loads a checkpoint with AutoPipelineForText2Image.from_pretrained.  Write an
    HF-layout checkpoint (unet/ + vae/ *.fp16.safetensors, diffusers key names, OIHW conv weights) of a tiny
    SDXL-shaped model, load it through the facade (native.from_safetensors -> DictProvider) and compare UNet and
    VAE outputs with the CPU oracle evaluated on the very tensors that were written."""
    import importlib
    import sys
    from safetensors.torch import save_file
    ucfg, vcfg = R.tiny_unet_cfg(), R.tiny_vae_cfg()
    uw = R.make_weights(R.unet_spec(ucfg), 21)
    vw = R.make_weights(R.vae_decoder_spec(vcfg), 22)
    (tmp_path / "unet").mkdir(); (tmp_path / "vae").mkdir()
    save_file({k: v.half().contiguous() for k, v in uw.items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    save_file({k: v.half().contiguous() for k, v in vw.items()}, str(tmp_path / "vae" / "diffusion_pytorch_model.fp16.safetensors"))
    monkeypatch.setenv("LB_WEIGHTS_DIR", str(tmp_path))
    monkeypatch.setenv("LB_TINY_MODEL", "1")
    for mname in [k for k in sys.modules if k == "diffusers" or k.startswith("diffusers.")]:
        del sys.modules[mname]
    sys.path.insert(0, ROOT)
    diffusers = importlib.import_module("diffusers")
    pipe = diffusers.AutoPipelineForText2Image.from_pretrained("stabilityai/sdxl-turbo", torch_dtype=torch.float16, variant="fp16")
    pipe.to("cuda")
    assert dataclasses.asdict(pipe.unet_cfg) == dataclasses.asdict(native().UNetConfig(**dataclasses.asdict(ucfg)))
    g = torch.Generator().manual_seed(3)
    L = 16
    x = torch.randn(2, 4, L, L, generator=g).half()
    ctx = torch.randn(2, 77, ucfg.cross_dim, generator=g).half()
    te = torch.randn(2, ucfg.pooled_dim, generator=g).half()
    ids = torch.tensor([[128.0, 128.0, 0.0, 0.0, 128.0, 128.0]] * 2)
    ref = R.unet_forward(ucfg, uw, x, torch.tensor(499.0), ctx, te, ids)
    got = pipe.unet(x.to(DEV), 499.0, encoder_hidden_states=ctx.to(DEV),
                    added_cond_kwargs={"text_embeds": te.to(DEV), "time_ids": ids.to(DEV)})[0]
    ru = rel_l2(got, ref)
    z = torch.randn(1, 4, L, L, generator=g).half()
    ref_u8 = R.postprocess_u8(R.vae_decode(vcfg, vw, z.float() / vcfg.scaling_factor))
    frame = pipe.native_latent2image(z.to(DEV))
    d = np.abs(np.asarray(frame).astype(np.int32) - ref_u8[0].astype(np.int32))
    results_log["safetensors_roundtrip"] = {"unet_rel_l2": ru, "vae_mean_abs_u8": float(d.mean())}
    print(f"[parity] safetensors round trip: UNet rel_l2={ru:.3e}, VAE mean|du8|={d.mean():.3f}")
    assert ru <= 1e-2 and d.mean() <= 2
