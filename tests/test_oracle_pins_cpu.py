"""Pins of the ORACLE's third-party half against independent implementations that exist in this image.

`diffusers` and `lpips` are not installed, so `oracle/sdxl_ref.py` restates their UNet / VAE / LPIPS forward
passes from the published architecture.  CLIP is pinned against the real `transformers` towers (tests/test_native_gpu.py).
This file adds the VAE decoder: `transformers` ships the decoder of the latent-diffusion / taming-transformers autoencoder -
the very module `diffusers.AutoencoderKL.decoder` was ported from - as `JanusVQVAEDecoder`
(transformers/models/janus/modeling_janus.py: conv_in -> mid (resnet, single-head attention, resnet) -> per level
`num_res_blocks + 1` resnets [+ nearest-2x upsample + 3x3 conv] -> GroupNorm(32, eps 1e-6) -> swish -> conv_out).  With the SDXL
VAE's configuration (base 128, multipliers (1, 2, 4, 4), 2 + 1 resnets per level, 4 latent channels) the only structural
difference is that the Janus decoder ALSO puts attention blocks behind the resnets of its lowest-resolution level; the SD
autoencoders were trained with `attn_resolutions = []`, i.e. without them (diffusers: `UpDecoderBlock2D` has no attention), so
the test empties that ModuleList - the forward code then skips them - and leaves everything else of the third-party module as
it is.  The oracle's weights (HF `AutoencoderKL` key names) are copied into the module's own parameters by name; diffusers'
`post_quant_conv` (a 1x1 conv in front of the decoder, not part of the LDM decoder class) is applied with torch before it.

What this pins: the oracle's resnet block (norm -> swish -> conv, twice, 1x1 shortcut), its mid-block attention (GroupNorm,
q / k / v / out projections, softmax(QK^T / sqrt(C)) V over H*W tokens, residual), the block / upsampler order, GroupNorm
epsilon and group count, the output head - against code its author never saw.  What it does not pin: that diffusers'
AutoencoderKL is configured as stated (block_out_channels (128, 256, 512, 512), layers_per_block 2: from its config.json).
The UNet, the schedulers and LPIPS stay restatements without an independent implementation in this image.
"""
import pytest
import torch

from oracle import sdxl_ref as R

janus = pytest.importorskip("transformers.models.janus.modeling_janus")
from transformers.models.janus.configuration_janus import JanusVQVAEConfig  # noqa: E402


def third_party_decoder(cfg: R.VAECfg, w):
    base = cfg.block_channels[0]
    assert all(c % base == 0 for c in cfg.block_channels)
    jc = JanusVQVAEConfig(base_channels=base, channel_multiplier=[c // base for c in cfg.block_channels],
                          num_res_blocks=cfg.layers_per_block, latent_channels=cfg.latent_channels,
                          out_channels=cfg.out_channels, dropout=0.0)
    dec = janus.JanusVQVAEDecoder(jc).float().eval()
    dec.up[0].attn = torch.nn.ModuleList()          # attn_resolutions = [] (see the module docstring)
    assert [len(u.block) for u in dec.up] == [cfg.layers_per_block + 1] * len(cfg.block_channels)

    def put(param, value):
        assert tuple(param.shape) == tuple(value.shape), (tuple(param.shape), tuple(value.shape))
        param.data.copy_(value.float())

    def conv(mod, key):
        put(mod.weight, w[key + ".weight"])
        put(mod.bias, w[key + ".bias"])

    def lin_as_conv(mod, key):                       # the oracle's Linear [C, C] is the module's 1x1 conv [C, C, 1, 1]
        put(mod.weight, w[key + ".weight"][:, :, None, None])
        put(mod.bias, w[key + ".bias"])

    def resnet(mod, key):
        for name in ("norm1", "conv1", "norm2", "conv2"):
            conv(getattr(mod, name), f"{key}.{name}")
        if key + ".conv_shortcut.weight" in w:
            conv(mod.nin_shortcut, key + ".conv_shortcut")
        else:
            assert not hasattr(mod, "nin_shortcut")

    conv(dec.conv_in, "decoder.conv_in")
    resnet(dec.mid.block_1, "decoder.mid_block.resnets.0")
    a = "decoder.mid_block.attentions.0"
    conv(dec.mid.attn_1.norm, a + ".group_norm")
    for mine, theirs in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "proj_out")):
        lin_as_conv(getattr(dec.mid.attn_1, theirs), f"{a}.{mine}")
    resnet(dec.mid.block_2, "decoder.mid_block.resnets.1")
    for ui, up in enumerate(dec.up):
        for li, blk in enumerate(up.block):
            resnet(blk, f"decoder.up_blocks.{ui}.resnets.{li}")
        if ui < len(dec.up) - 1:
            conv(up.upsample.conv, f"decoder.up_blocks.{ui}.upsamplers.0.conv")
        else:
            assert not hasattr(up, "upsample")
    conv(dec.norm_out, "decoder.conv_norm_out")
    conv(dec.conv_out, "decoder.conv_out")
    used = {k for k in w if not k.startswith("post_quant_conv")}
    n_theirs = sum(p.numel() for p in dec.parameters())
    assert n_theirs == sum(w[k].numel() for k in used), "every oracle tensor has exactly one home in the third-party module"
    return dec


@pytest.mark.parametrize("which,latent", [("tiny", 16), ("sdxl", 8)])
def test_oracle_vae_decoder_equals_the_third_party_ldm_decoder(which, latent):
    """Same weights, same latent: the oracle's `vae_decode` and transformers' LDM decoder agree to fp32 round-off - at the tiny
    width the other tests use and at the SDXL VAE's full width (83.7 M decoder parameters, 8 x 8 latent -> 64 x 64 image)."""
    cfg = R.tiny_vae_cfg() if which == "tiny" else R.VAECfg()
    w = {k: v.float() for k, v in R.make_weights(R.vae_decoder_spec(cfg), seed=1).items()}
    dec = third_party_decoder(cfg, w)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2, cfg.latent_channels, latent, latent, generator=g)
    with torch.no_grad():
        mine = R.vae_decode(cfg, w, z)
        pq = torch.nn.functional.conv2d(z, w["post_quant_conv.weight"], w["post_quant_conv.bias"])
        theirs = dec(pq)
    assert mine.shape == theirs.shape == (2, cfg.out_channels, 8 * latent, 8 * latent)
    err = float((mine - theirs).norm() / theirs.norm())
    assert err < 2e-5, err                      # (fp32 on both sides; different summation orders in bmm vs matmul attention)
    assert float((mine - theirs).abs().max()) < 1e-3 * float(theirs.abs().max())


def test_oracle_attention_equals_torch_sdpa():
    """The multi-head attention every transformer block of the oracle's UNet (and the VAE's mid block, heads = 1) goes through,
    against torch's own `scaled_dot_product_attention` (math backend on CPU): self- and cross-attention shapes of the SDXL UNet."""
    g = torch.Generator().manual_seed(9)
    for (B, Sq, Sk, C, heads) in ((2, 256, 256, 1280, 20), (2, 1024, 77, 640, 10), (1, 64, 64, 512, 1)):
        q, k, v = (torch.randn(B, S, C, generator=g) for S in (Sq, Sk, Sk))
        mine = R.attention(q, k, v, heads)
        d = C // heads
        split = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
        theirs = torch.nn.functional.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(B, Sq, C)
        assert float((mine - theirs).norm() / theirs.norm()) < 1e-5
