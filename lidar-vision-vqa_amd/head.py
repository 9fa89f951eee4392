"""Prefix assembly + stand-in language head.

The reference concatenates `[E(<vision_start>), prefix_vision*s, E(<vision_end>), E(<lidar_start>),
prefix_lidar*s, E(<lidar_end>), E(prompt), E(answer)]` and calls
`base(inputs_embeds=..., attention_mask=ones, labels=...)` on a LoRA-wrapped `Qwen/Qwen2.5-0.5B`
(encoder-decoder/training/core/validation.py:105-158, trainer.py:568-675,
inference/inference_engine.py:139-227).  That checkpoint is fetched by name -> unreachable offline, so the
head here is a random-init decoder of the SAME architecture (transformers' Qwen2ForCausalLM: RMSNorm,
rotary GQA self-attention with q/k/v bias, SwiGLU, tied lm_head, shifted cross-entropy) with the SAME
state_dict keys, pinned against transformers in tests/golden/head_prefix.npz.  LoRA-B is zero at init,
so a LoRA-wrapped base equals the base for an untrained-weights parity test.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _ffi as F
from . import ops
from .fusion import _HipModule, _f32


def assemble_prefix(prefix_vision: Optional[torch.Tensor], prefix_lidar: Optional[torch.Tensor], e_special: torch.Tensor,
                    e_prompt: torch.Tensor, e_answer: torch.Tensor, answer_ids: torch.Tensor, prefix_scale: float = 0.2
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Eval order VISION -> LIDAR -> TEXT -> ANSWER, prompt appended once (validation.py:125-148).
    e_special rows: <vision_start>, <vision_end>, <lidar_start>, <lidar_end>.
    Returns inputs_embeds [B,L,d], attention_mask [B,L] (ones), labels [B,L] (-100 except the answer span)."""
    B = e_prompt.shape[0]
    pieces = []

    def scaled(p):
        return ops.scale_add_rows(_f32(p).view(-1, p.shape[-1]), None, prefix_scale).view_as(p)
    if prefix_vision is not None:
        pieces += [e_special[0].expand(B, 1, -1), scaled(prefix_vision), e_special[1].expand(B, 1, -1)]
    if prefix_lidar is not None:
        pieces += [e_special[2].expand(B, 1, -1), scaled(prefix_lidar), e_special[3].expand(B, 1, -1)]
    pieces.append(e_prompt)
    inp = torch.cat(pieces + [e_answer], dim=1).contiguous()
    L = inp.shape[1]
    labels = torch.full((B, L), -100, dtype=torch.long, device=inp.device)
    labels[:, -answer_ids.shape[1]:] = answer_ids
    attn = torch.ones((B, L), dtype=torch.long, device=inp.device)
    return inp, attn, labels


class _Weight(nn.Module):
    """RMSNorm parameter container (key `<name>.weight`)."""

    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))


class _Attn(nn.Module):
    def __init__(self, d, dkv):
        super().__init__()
        self.q_proj = nn.Linear(d, d, bias=True)
        self.k_proj = nn.Linear(d, dkv, bias=True)
        self.v_proj = nn.Linear(d, dkv, bias=True)
        self.o_proj = nn.Linear(d, d, bias=False)


class _Mlp(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.gate_proj = nn.Linear(d, inter, bias=False)
        self.up_proj = nn.Linear(d, inter, bias=False)
        self.down_proj = nn.Linear(inter, d, bias=False)


class _Layer(nn.Module):
    def __init__(self, d, dkv, inter):
        super().__init__()
        self.self_attn = _Attn(d, dkv)
        self.mlp = _Mlp(d, inter)
        self.input_layernorm = _Weight(d)
        self.post_attention_layernorm = _Weight(d)


class _Model(nn.Module):
    def __init__(self, vocab, d, dkv, inter, n_layers):
        super().__init__()
        self.embed_tokens = nn.Embedding(vocab, d)
        self.layers = nn.ModuleList([_Layer(d, dkv, inter) for _ in range(n_layers)])
        self.norm = _Weight(d)


class HeadOutput:
    def __init__(self, logits, loss):
        self.logits, self.loss = logits, loss


class StandInHead(_HipModule):
    """Qwen2-architecture causal LM; call like the reference's `base(inputs_embeds=, attention_mask=, labels=)`."""

    def __init__(self, vocab: int, d: int, inter: int, n_heads: int, n_kv_heads: int, n_layers: int, rms_eps: float = 1e-6,
                 rope_theta: float = 1000000.0):
        super().__init__()
        assert d % n_heads == 0 and n_heads % n_kv_heads == 0
        self.dh = d // n_heads
        assert self.dh % 16 == 0 and self.dh <= 128, "stand-in head uses the fused attention kernel (head_dim multiple of 16, <= 128)"
        self.cfg = dict(vocab=vocab, d=d, inter=inter, n_heads=n_heads, n_kv_heads=n_kv_heads, n_layers=n_layers, rms_eps=rms_eps,
                        rope_theta=rope_theta)
        dkv = self.dh * n_kv_heads
        self.model = _Model(vocab, d, dkv, inter, n_layers)
        self.lm_head = nn.Linear(d, vocab, bias=False)
        self.lm_head.weight = self.model.embed_tokens.weight      # tie_word_embeddings=True
        object.__setattr__(self, "_packed", {})

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        """E(ids): a row gather of the embedding table (byte movement)."""
        return self.model.embed_tokens.weight.detach()[ids]

    def _pack(self, key, params, biases=None):
        """Concatenated projection weights (q|k|v, gate|up) as one GEMM operand; rebuilt when a member changes."""
        split = self._split()
        ver = tuple((p.data_ptr(), p._version) for p in params) + (split,)
        hit = self._packed.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        w = ops.cast(torch.cat([p.detach().float() for p in params], dim=0).contiguous(), split)
        b = torch.cat([p.detach().float() for p in biases]).contiguous() if biases else None
        self._packed[key] = (ver, w, b)
        return w, b

    def forward(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None) -> HeadOutput:
        self._guard(inputs_embeds)
        c = self.cfg
        B, L, d = inputs_embeds.shape
        H, Hk, dh = c["n_heads"], c["n_kv_heads"], self.dh
        dkv = dh * Hk
        split = self._split()
        x = _f32(inputs_embeds).view(B * L, d)
        ld = d + 2 * dkv
        for i, layer in enumerate(self.model.layers):
            a = layer.self_attn
            h = ops.rmsnorm(x, layer.input_layernorm.weight, c["rms_eps"], split)
            wqkv, bqkv = self._pack(("qkv", i), (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight),
                                    (a.q_proj.bias, a.k_proj.bias, a.v_proj.bias))
            _, qkv = ops.linear(h, wqkv, bqkv, out_bf=True)
            sl = lambda t, c0: (t[0][:, c0:], None if t[1] is None else t[1][:, c0:])
            ops.rope_inplace(sl(qkv, 0), B * L, L, H, dh, ld, c["rope_theta"])
            ops.rope_inplace(sl(qkv, d), B * L, L, Hk, dh, ld, c["rope_theta"])
            st = (L * ld, ld, dh)
            o = ops.attention(sl(qkv, 0), sl(qkv, d), sl(qkv, d + dkv), batch=B, n_heads=H, n_kv_heads=Hk, nq=L, nkv=L, dh=dh,
                              q_strides=st, k_strides=st, v_strides=st, scale=1.0 / math.sqrt(dh), causal=True)
            x, _ = ops.linear(o, self._w(a.o_proj.weight), None, residual=x, out_f32=True)
            h = ops.rmsnorm(x, layer.post_attention_layernorm.weight, c["rms_eps"], split)
            wgu, _ = self._pack(("gu", i), (layer.mlp.gate_proj.weight, layer.mlp.up_proj.weight))
            gu, _ = ops.linear(h, wgu, None, out_f32=True)
            act = ops.swiglu(gu, split)
            x, _ = ops.linear(act, self._w(layer.mlp.down_proj.weight), None, residual=x, out_f32=True)
        hf = ops.rmsnorm(x, self.model.norm.weight, c["rms_eps"], split)
        logits, _ = ops.linear(hf, self._w(self.model.embed_tokens.weight), None, out_f32=True)
        loss = None
        if labels is not None:
            shifted = torch.full_like(labels, -100)
            shifted[:, :-1] = labels[:, 1:]
            loss = ops.cross_entropy(logits, shifted.reshape(-1).contiguous())
        return HeadOutput(logits.view(B, L, -1), loss)
