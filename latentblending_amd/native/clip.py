"""Native CLIP text towers for SDXL prompt conditioning (gfx950).

What ``pipe.encode_prompt`` does in the reference's stack (/root/reference/latentblending/diffusers_holder.py:79-96 ->
diffusers ``StableDiffusionXLPipeline.encode_prompt`` -> ``transformers.CLIPTextModel`` /
``CLIPTextModelWithProjection``): the prompt is tokenised twice (77 tokens, BOS ... EOS, padded), run through
CLIP-L/14 (12 layers, width 768, quick-GELU) and OpenCLIP-bigG/14 (32 layers, width 1280, GELU) with a CAUSAL mask;
the PENULTIMATE hidden states of both towers are concatenated to ``prompt_embeds [1, 77, 2048]`` and the bigG tower's
projected EOS token is ``pooled_prompt_embeds [1, 1280]``.

Here a tower is one recorded launch program on the kernels the UNet already uses: token + position embedding gather,
LayerNorm, fused QKV projection (bias in the GEMM epilogue), the d = 64 attention kernel with its causal mask (77 keys
= ONE 96-key tile), output projection + residual in the epilogue, MLP with the activation in the epilogue.  Parameter
names are the HF ``transformers`` state-dict keys, so ``DictProvider`` / ``from_safetensors`` load real checkpoints
(``text_encoder/model.safetensors``, ``text_encoder_2/...``) unchanged.

Tokenisers: ``transformers.CLIPTokenizer`` when ``vocab.json`` / ``merges.txt`` are on disk; there is no vocabulary
offline, so ``HashTokenizer`` (deterministic word -> id hashing) stands in and says so (``allow_synthetic``).
"""
from __future__ import annotations

import os
import re
import warnings
import zlib
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..hip import lib
from ..hip.lib import api
from .runtime import Arena, Emitter, Program, F16, F32, _stream

MAX_TOKENS = 77


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    projection_dim: Optional[int] = None       # None: CLIPTextModel (no text_projection)
    bos_token_id: int = 49406
    eos_token_id: int = 49407
    pad_token_id: int = 49407

    @staticmethod
    def clip_l() -> "CLIPTextConfig":          # SDXL text_encoder (openai/clip-vit-large-patch14 text tower)
        return CLIPTextConfig()

    @staticmethod
    def openclip_bigg() -> "CLIPTextConfig":   # SDXL text_encoder_2 (laion/CLIP-ViT-bigG-14 text tower)
        return CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                              hidden_act="gelu", projection_dim=1280, pad_token_id=0)


class NativeCLIPText:
    """One CLIP text tower.  ``forward(ids)`` -> (penultimate hidden states [B, 77, C] fp16,
    projected EOS embedding [B, P] fp16 or None)."""

    def __init__(self, cfg: CLIPTextConfig, provider, device="cuda", prefix: Optional[str] = None):
        assert cfg.hidden_size % 64 == 0 and cfg.hidden_size // cfg.num_attention_heads == 64, "head_dim must be 64"
        assert cfg.hidden_act in ("quick_gelu", "gelu")
        self.cfg, self.device = cfg, torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        if prefix is None:      # HF checkpoints nest the tower under "text_model." or not, depending on class / version
            state = getattr(provider, "state", None) or {}
            base = getattr(provider, "prefix", "")
            prefix = "" if (base + "embeddings.token_embedding.weight") in state else "text_model."
        c, pv, p = cfg.hidden_size, provider, prefix

        def dev(t, dt):
            return t.to(device=self.device, dtype=dt).contiguous()

        self.w["tok"] = dev(pv.weight(p + "embeddings.token_embedding.weight", (cfg.vocab_size, c), c, c ** 0.5 * 0.02), F16)
        self.w["pos"] = dev(pv.weight(p + "embeddings.position_embedding.weight", (cfg.max_position_embeddings, c), c, c ** 0.5 * 0.01), F16)
        for i in range(cfg.num_hidden_layers):
            L = f"{p}encoder.layers.{i}."
            q = [pv.weight(L + f"self_attn.{n}_proj.weight", (c, c), c) for n in ("q", "k", "v")]
            qb = [pv.bias(L + f"self_attn.{n}_proj.bias", c) for n in ("q", "k", "v")]
            self.w[f"{i}.qkv.w"] = dev(torch.cat(q, 0), F16)                       # one [3C, C] projection
            self.w[f"{i}.qkv.b"] = dev(torch.cat(qb, 0), F32)
            self.w[f"{i}.out.w"] = dev(pv.weight(L + "self_attn.out_proj.weight", (c, c), c, 0.5), F16)
            self.w[f"{i}.out.b"] = dev(pv.bias(L + "self_attn.out_proj.bias", c), F32)
            for n in ("layer_norm1", "layer_norm2"):
                self.w[f"{i}.{n}.w"] = dev(pv.norm_weight(L + n + ".weight", c), F32)
                self.w[f"{i}.{n}.b"] = dev(pv.bias(L + n + ".bias", c), F32)
            self.w[f"{i}.fc1.w"] = dev(pv.weight(L + "mlp.fc1.weight", (cfg.intermediate_size, c), c), F16)
            self.w[f"{i}.fc1.b"] = dev(pv.bias(L + "mlp.fc1.bias", cfg.intermediate_size), F32)
            self.w[f"{i}.fc2.w"] = dev(pv.weight(L + "mlp.fc2.weight", (c, cfg.intermediate_size), cfg.intermediate_size, 0.5), F16)
            self.w[f"{i}.fc2.b"] = dev(pv.bias(L + "mlp.fc2.bias", c), F32)
        self.w["final.w"] = dev(pv.norm_weight(p + "final_layer_norm.weight", c), F32)
        self.w["final.b"] = dev(pv.bias(p + "final_layer_norm.bias", c), F32)
        if cfg.projection_dim:
            root = prefix[:-len("text_model.")] if prefix.endswith("text_model.") else prefix
            self.w["proj"] = dev(pv.weight(root + "text_projection.weight", (cfg.projection_dim, c), c), F16)
        self._programs: Dict[int, "_CLIPProgram"] = {}

    def forward(self, ids: torch.Tensor) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """ids: [B, 77] integer token ids (any device)."""
        B, S = ids.shape
        assert S == self.cfg.max_position_embeddings, f"expected {self.cfg.max_position_embeddings} tokens, got {S}"
        prog = self._programs.get(B)
        if prog is None:
            prog = self._programs[B] = _CLIPProgram(self, B)
        return prog.run(ids)


class _CLIPProgram:
    def __init__(self, net: NativeCLIPText, B: int):
        cfg, dev = net.cfg, net.device
        self.net, self.B = net, B
        c, S, M = cfg.hidden_size, cfg.max_position_embeddings, B * cfg.max_position_embeddings
        self.arena = Arena(dev)
        self.em = Emitter(self.arena)
        self.ids = torch.zeros(M, dtype=torch.int32, device=dev)
        self.eos_rows = torch.zeros(B, dtype=torch.int32, device=dev)
        self.penultimate = torch.zeros(B, S, c, dtype=F16, device=dev)
        self.pooled = torch.zeros(B, cfg.projection_dim, dtype=F16, device=dev) if cfg.projection_dim else None
        self.prog = Program("clip-text")
        with self.prog.record():
            self._emit()

    def _emit(self):
        net, cfg, em, ar, w = self.net, self.net.cfg, self.em, self.arena, self.net.w
        c, S, B = cfg.hidden_size, cfg.max_position_embeddings, self.B
        M, H = B * S, cfg.num_attention_heads
        act = lib.GEMM_QUICK_GELU if cfg.hidden_act == "quick_gelu" else lib.GEMM_GELU
        h = ar.alloc((M, c))
        api.lb_embed_tokens_f16(self.ids.data_ptr(), w["tok"].data_ptr(), w["pos"].data_ptr(), h.data_ptr(), M, S, c,
                                cfg.vocab_size, _stream())
        for i in range(cfg.num_hidden_layers):
            if i == cfg.num_hidden_layers - 1:          # hidden_states[-2]: the input of the last layer
                em.copy(self.penultimate, h)
            ln = ar.alloc((M, c))
            em.layernorm(h, ln, w[f"{i}.layer_norm1.w"], w[f"{i}.layer_norm1.b"], M=M, C_=c, eps=cfg.layer_norm_eps)
            qkv = ar.alloc((M, 3 * c))
            em.gemm(ln, w[f"{i}.qkv.w"], qkv, M=M, bias=w[f"{i}.qkv.b"])
            ar.release(ln)
            a = ar.alloc((M, c))
            em.attention(qkv.data_ptr(), qkv.data_ptr() + c * 2, qkv.data_ptr() + 2 * c * 2, a, B=B, H=H, Sq=S, Skv=S,
                         valid=S, ldq=3 * c, ldk=3 * c, ldv=3 * c, ldo=c, causal=True)
            ar.release(qkv)
            em.gemm(a, w[f"{i}.out.w"], h, M=M, bias=w[f"{i}.out.b"], residual=h)
            ar.release(a)
            ln = ar.alloc((M, c))
            em.layernorm(h, ln, w[f"{i}.layer_norm2.w"], w[f"{i}.layer_norm2.b"], M=M, C_=c, eps=cfg.layer_norm_eps)
            ff = ar.alloc((M, cfg.intermediate_size))
            em.gemm(ln, w[f"{i}.fc1.w"], ff, M=M, bias=w[f"{i}.fc1.b"], flags=act)
            ar.release(ln)
            em.gemm(ff, w[f"{i}.fc2.w"], h, M=M, bias=w[f"{i}.fc2.b"], residual=h)
            ar.release(ff)
        if self.pooled is not None:                     # text_projection(final_layer_norm(last_hidden)[EOS row])
            eos = ar.alloc((B, c))
            api.lb_gather_rows_f16(h.data_ptr(), self.eos_rows.data_ptr(), eos.data_ptr(), B, c, c, _stream())
            fin = ar.alloc((B, c))
            em.layernorm(eos, fin, w["final.w"], w["final.b"], M=B, C_=c, eps=cfg.layer_norm_eps)
            em.gemm(fin, w["proj"], self.pooled, M=B)
            ar.release(eos)
            ar.release(fin)
        ar.release(h)

    def run(self, ids: torch.Tensor):
        cfg = self.net.cfg
        ids = ids.to(torch.int64).cpu()
        # pooled token: transformers' rule (legacy configs with eos_token_id == 2 take argmax, others the first EOS)
        if cfg.eos_token_id == 2:
            pos = ids.argmax(dim=-1)
        else:
            pos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
        S = cfg.max_position_embeddings
        self.ids.copy_(ids.reshape(-1).to(torch.int32))
        self.eos_rows.copy_((torch.arange(self.B) * S + pos).to(torch.int32))
        self.prog.launch()
        return self.penultimate, self.pooled


class HashTokenizer:
    """Stand-in tokenizer (no CLIP vocabulary offline): lower-cased words / punctuation are hashed into the id range
    below BOS, wrapped in BOS ... EOS and padded to 77.  Deterministic, prompt-sensitive, NOT the CLIP BPE."""

    def __init__(self, cfg: CLIPTextConfig):
        self.cfg = cfg

    def __call__(self, text: str) -> torch.Tensor:
        c = self.cfg
        words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", text.lower())[: MAX_TOKENS - 2]
        ids = [c.bos_token_id] + [1 + zlib.crc32(w.encode()) % (c.bos_token_id - 1) for w in words] + [c.eos_token_id]
        ids += [c.pad_token_id] * (MAX_TOKENS - len(ids))
        return torch.tensor([ids], dtype=torch.int64)


def _hf_tokenizer(path: str):
    from transformers import CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(path)

    def call(text: str) -> torch.Tensor:
        return tok(text, padding="max_length", max_length=MAX_TOKENS, truncation=True, return_tensors="pt").input_ids
    return call


class NativeTextEncoders:
    """Both SDXL text towers + tokenisers; ``encode`` is a ``text_encoder_fn`` for ``NativeSDXLPipe``."""

    def __init__(self, enc1: NativeCLIPText, enc2: NativeCLIPText, tokenizer1=None, tokenizer2=None,
                 allow_synthetic: bool = False):
        assert enc2.cfg.projection_dim, "the second tower must be a CLIPTextModelWithProjection"
        self.enc1, self.enc2 = enc1, enc2
        synthetic = tokenizer1 is None or tokenizer2 is None
        self.tok1 = tokenizer1 or HashTokenizer(enc1.cfg)
        self.tok2 = tokenizer2 or HashTokenizer(enc2.cfg)
        if synthetic and not allow_synthetic and os.environ.get("LB_ALLOW_SYNTHETIC") != "1":
            warnings.warn("NativeTextEncoders: no CLIP vocabulary (tokenizer/vocab.json + merges.txt) - using the HASH "
                          "tokenizer stand-in; token ids are not CLIP's", UserWarning, stacklevel=2)

    @classmethod
    def from_dir(cls, root: str, device="cuda", **kw) -> "NativeTextEncoders":
        """HF layout: ``root/text_encoder``, ``root/text_encoder_2`` (safetensors), ``root/tokenizer``, ``root/tokenizer_2``."""
        from .weights import from_safetensors
        e1 = NativeCLIPText(CLIPTextConfig.clip_l(), from_safetensors(os.path.join(root, "text_encoder")), device)
        e2 = NativeCLIPText(CLIPTextConfig.openclip_bigg(), from_safetensors(os.path.join(root, "text_encoder_2")), device)
        toks = []
        for name in ("tokenizer", "tokenizer_2"):
            d = os.path.join(root, name)
            toks.append(_hf_tokenizer(d) if os.path.isfile(os.path.join(d, "vocab.json")) else None)
        return cls(e1, e2, toks[0], toks[1], **kw)

    @torch.no_grad()
    def encode(self, text: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (prompt_embeds [1, 77, C1 + C2] fp16, pooled [1, P] fp16), the pair diffusers' encode_prompt builds."""
        h1, _ = self.enc1.forward(self.tok1(text))
        h2, pooled = self.enc2.forward(self.tok2(text))
        return torch.cat([h1, h2], dim=-1).clone(), pooled.clone()
