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
import os
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


import ctypes as _ct


class _Qwen2LayerPtrs(_ct.Structure):
    """include/lvq.h: lvq_qwen2_layer (device pointers of one decoder layer)."""
    _fields_ = [(n, _ct.c_void_p) for n in ("ln1", "ln2", "wqkv", "wqkv_lo", "bqkv", "wo", "wo_lo", "wgu", "wgu_lo", "wdown", "wdown_lo",
                                             "k_cache", "k_cache_lo", "v_cache", "v_cache_lo")]


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

    def _layers(self, x: torch.Tensor, B: int, Lq: int, pos0: int = 0, cache=None) -> torch.Tensor:
        """The decoder stack on rows x [B*Lq, d] = positions pos0 .. pos0+Lq-1 of every sequence.  cache = None: plain causal
        self-attention over these rows (training-eval path).  cache = list of per-layer ((k_hi, k_lo), (v_hi, v_lo)) buffers
        [B, Lmax, dkv]: the new keys / values are appended at pos0 and attention runs over positions 0 .. pos0+Lq-1."""
        c = self.cfg
        d = c["d"]
        H, Hk, dh = c["n_heads"], c["n_kv_heads"], self.dh
        dkv = dh * Hk
        split = self._split()
        ld = d + 2 * dkv
        sl = lambda t, c0: (t[0][:, c0:], None if t[1] is None else t[1][:, c0:])
        for i, layer in enumerate(self.model.layers):
            a = layer.self_attn
            h = ops.rmsnorm(x, layer.input_layernorm.weight, c["rms_eps"], split)
            wqkv, bqkv = self._pack(("qkv", i), (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight),
                                    (a.q_proj.bias, a.k_proj.bias, a.v_proj.bias))
            _, qkv = ops.linear(h, wqkv, bqkv, out_bf=True)
            ops.rope_inplace(sl(qkv, 0), B * Lq, Lq, H, dh, ld, c["rope_theta"], pos0)
            ops.rope_inplace(sl(qkv, d), B * Lq, Lq, Hk, dh, ld, c["rope_theta"], pos0)
            st = (Lq * ld, ld, dh)
            if cache is None:
                o = ops.attention(sl(qkv, 0), sl(qkv, d), sl(qkv, d + dkv), batch=B, n_heads=H, n_kv_heads=Hk, nq=Lq, nkv=Lq, dh=dh,
                                  q_strides=st, k_strides=st, v_strides=st, scale=1.0 / math.sqrt(dh), causal=True)
            else:
                kc, vc = cache[i]
                lmax = kc[0].shape[1]
                for part in (0, 1):                       # hi (and lo in the bf16x3 mode): byte movement into the cache
                    if qkv[part] is None:
                        continue
                    rows = qkv[part].view(B, Lq, ld)
                    kc[part][:, pos0:pos0 + Lq].copy_(rows[:, :, d:d + dkv])
                    vc[part][:, pos0:pos0 + Lq].copy_(rows[:, :, d + dkv:])
                cs = (lmax * dkv, dkv, dh)
                o = ops.attention(sl(qkv, 0), kc, vc, batch=B, n_heads=H, n_kv_heads=Hk, nq=Lq, nkv=pos0 + Lq, dh=dh,
                                  q_strides=st, k_strides=cs, v_strides=cs, scale=1.0 / math.sqrt(dh), causal=Lq > 1)
            x, _ = ops.linear(o, self._w(a.o_proj.weight), None, residual=x, out_f32=True)
            h = ops.rmsnorm(x, layer.post_attention_layernorm.weight, c["rms_eps"], split)
            wgu, _ = self._pack(("gu", i), (layer.mlp.gate_proj.weight, layer.mlp.up_proj.weight))
            gu, _ = ops.linear(h, wgu, None, out_f32=True)
            act = ops.swiglu(gu, split)
            x, _ = ops.linear(act, self._w(layer.mlp.down_proj.weight), None, residual=x, out_f32=True)
        return x

    def _logits(self, x: torch.Tensor) -> torch.Tensor:
        hf = ops.rmsnorm(x, self.model.norm.weight, self.cfg["rms_eps"], self._split())
        logits, _ = ops.linear(hf, self._w(self.model.embed_tokens.weight), None, out_f32=True)
        return logits

    def forward(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None) -> HeadOutput:
        self._guard(inputs_embeds)
        B, L, d = inputs_embeds.shape
        x = self._layers(_f32(inputs_embeds).view(B * L, d), B, L)
        logits = self._logits(x)
        loss = None
        if labels is not None:
            shifted = torch.full_like(labels, -100)
            shifted[:, :-1] = labels[:, 1:]
            loss = ops.cross_entropy(logits, shifted.reshape(-1).contiguous())
        return HeadOutput(logits.view(B, L, -1), loss)

    def _native_layers(self, cache):
        """Host array of lvq_qwen2_layer structs for lvq_qwen2_decode_step + the tensors that must stay alive behind it."""
        keep, arr = [], (_Qwen2LayerPtrs * len(self.model.layers))()
        p = lambda t: None if t is None else t.data_ptr()
        for i, layer in enumerate(self.model.layers):
            a = layer.self_attn
            wqkv, bqkv = self._pack(("qkv", i), (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight),
                                    (a.q_proj.bias, a.k_proj.bias, a.v_proj.bias))
            wo = self._w(a.o_proj.weight)
            wgu, _ = self._pack(("gu", i), (layer.mlp.gate_proj.weight, layer.mlp.up_proj.weight))
            wd = self._w(layer.mlp.down_proj.weight)
            ln1 = layer.input_layernorm.weight.detach().float().contiguous()
            ln2 = layer.post_attention_layernorm.weight.detach().float().contiguous()
            (kh, kl), (vh, vl) = cache[i]
            keep += [wqkv, bqkv, wo, wgu, wd, ln1, ln2]
            arr[i] = _Qwen2LayerPtrs(p(ln1), p(ln2), p(wqkv[0]), p(wqkv[1]), p(bqkv), p(wo[0]), p(wo[1]), p(wgu[0]), p(wgu[1]),
                                     p(wd[0]), p(wd[1]), p(kh), p(kl), p(vh), p(vl))
        return arr, keep

    @torch.no_grad()
    def generate(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, max_new_tokens: int = 64,
                 do_sample: bool = False, num_beams: int = 1, pad_token_id: Optional[int] = None,
                 eos_token_id: Optional[int] = None, output_scores: bool = False, temperature: float = 1.0, top_k: Optional[int] = 50,
                 top_p: float = 1.0, generator: Optional[torch.Generator] = None, **unused):
        """`base_model.generate(inputs_embeds=, attention_mask=, max_new_tokens=, temperature=, top_p=, top_k=, do_sample=,
        num_beams=1, pad_token_id=, eos_token_id=)` as inference_engine.py:283-296 calls it, with a per-layer KV cache.
        do_sample=False: greedy (first maximum).  do_sample=True (the reference's default, temperature 0.7 / top_k 50 / top_p 0.9):
        every step draws from transformers' warped distribution (temperature -> top-k -> top-p) with lvq_sample_rows; the uniforms
        come from torch's RNG (`generator` or torch.manual_seed), so runs are reproducible but -- like any two sampling
        implementations -- not stream-identical to transformers' multinomial.  Returns the NEW token ids [B, n] (what transformers
        returns when only inputs_embeds is given); with output_scores=True also the per-step raw logits [B, n, V].  Beam search
        is not built."""
        self._guard(inputs_embeds)
        if num_beams != 1:
            raise F.LvqError("StandInHead.generate: beam search is not implemented (num_beams must be 1)")
        if do_sample and not (temperature > 0.0 and 0.0 < top_p <= 1.0):
            raise ValueError("temperature must be > 0 and top_p in (0, 1]")
        if attention_mask is not None and not bool((attention_mask == 1).all()):
            raise F.LvqError("StandInHead.generate expects an all-ones attention_mask (the reference builds exactly that)")
        c = self.cfg
        B, L, d = inputs_embeds.shape
        dkv = self.dh * c["n_kv_heads"]
        dev = inputs_embeds.device
        lmax = L + max_new_tokens
        split = self._split()
        mk = lambda: (torch.empty((B, lmax, dkv), dtype=torch.bfloat16, device=dev),
                      torch.empty((B, lmax, dkv), dtype=torch.bfloat16, device=dev) if split else None)
        cache = [(mk(), mk()) for _ in self.model.layers]
        x = self._layers(_f32(inputs_embeds).view(B * L, d), B, L, 0, cache)          # prefill
        native = not os.environ.get("LVQ_DECODE_PYTHON")       # decode steps: one native call per token (csrc/decoder.hip)
        if native:
            lib = F.lib()
            layers_arr, keep = self._native_layers(cache)
            prec = 3 if split else 1
            nbytes = lib.lvq_qwen2_decode_workspace_bytes(F.cint(B), F.cint(d), F.cint(c["n_heads"]), F.cint(c["n_kv_heads"]),
                                                          F.cint(c["inter"]), F.cint(lmax), F.cint(prec))
            step_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        step_logits = self._logits(x.view(B, L, d)[:, -1].contiguous())                 # [B, V]
        pad = 0 if pad_token_id is None else int(pad_token_id)
        unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        ids, scores = [], []
        for t in range(max_new_tokens):
            nxt = ops.sample_rows(step_logits, temperature, top_k, top_p, generator) if do_sample else ops.argmax_rows(step_logits)
            if eos_token_id is not None:
                nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
            ids.append(nxt)
            if output_scores:
                scores.append(step_logits)
            if eos_token_id is not None:
                unfinished = unfinished & (nxt != int(eos_token_id))
                if not bool(unfinished.any()):                                          # host sync, as in transformers' loop
                    break
            if t + 1 == max_new_tokens:
                break
            x = self.embed(nxt).float().contiguous()
            if native:
                rc = lib.lvq_qwen2_decode_step(layers_arr, F.cint(len(self.model.layers)), F.ptr(x), F.cint(B), F.cint(d),
                                               F.cint(c["n_heads"]), F.cint(c["n_kv_heads"]), F.cint(c["inter"]), F.cint(L + t), F.cint(lmax),
                                               F.cfloat(c["rms_eps"]), F.cfloat(c["rope_theta"]), F.cint(prec), F.ptr(step_ws),
                                               F.csize(step_ws.numel()), F.stream_ptr(dev))
                F.check(rc, "lvq_qwen2_decode_step")
            else:
                x = self._layers(x, B, 1, L + t, cache)
            step_logits = self._logits(x)
        out = torch.stack(ids, dim=1)
        return (out, torch.stack(scores, dim=1)) if output_scores else out
