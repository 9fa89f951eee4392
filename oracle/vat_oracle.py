"""oracle/vat_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain torch-CPU fp32 restatement of the fusion half of the hot path (VATBlock, VATLiDAR, VATVision,
VisionAdapter, sdp_attention, the DeepEncoder fuse-projector, prefix assembly and a Qwen2-style
stand-in language head).  Functional style: every function takes the reference module's
`state_dict()` (same key names), so the same weights drive the reference,
this oracle and the HIP modules.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py may import it.

Pinned against the UNMODIFIED reference classes imported in the build container
(tools/make_goldens.py -> tests/golden/vat_*.npz; tests/test_oracle_vat.py).
All citations relative to /root/reference/src/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
NUM_VIEWS = 6  # encoder-decoder/training/models/vat_lidar.py:39


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def sdp_attention(q, k, v, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """deepencoder/clip_sdpa.py:50-66 (== sam_vary_sdpa.py:27-42): [B,H,S,D] -> [B,H,S,D]."""
    s = (q @ k.transpose(-2, -1)) / math.sqrt(q.shape[-1])
    if attn_mask is not None:
        s = s + attn_mask
    return torch.softmax(s, dim=-1) @ v


def mha(xq: torch.Tensor, xkv: torch.Tensor, sd: SD, p: str, n_heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True, eval) with packed in_proj."""
    d = xq.shape[-1]
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = xq @ w[0:d].t() + b[0:d]
    k = xkv @ w[d:2 * d].t() + b[d:2 * d]
    v = xkv @ w[2 * d:3 * d].t() + b[2 * d:3 * d]
    B, nq, _ = q.shape
    nk = k.shape[1]
    dh = d // n_heads
    q = q.view(B, nq, n_heads, dh).transpose(1, 2)
    k = k.view(B, nk, n_heads, dh).transpose(1, 2)
    v = v.view(B, nk, n_heads, dh).transpose(1, 2)
    o = sdp_attention(q, k, v).transpose(1, 2).reshape(B, nq, d)
    return o @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]


def vat_block(q: torch.Tensor, kv: torch.Tensor, sd: SD, p: str, n_heads: int) -> torch.Tensor:
    """encoder-decoder/training/models/vat_blocks.py:36-47 (eval: dropout = identity)."""
    qn = layer_norm(q, sd[p + "sa_ln.weight"], sd[p + "sa_ln.bias"])
    q = q + mha(qn, qn, sd, p + "sa.", n_heads)
    q = q + mha(layer_norm(q, sd[p + "ca_ln.weight"], sd[p + "ca_ln.bias"]), kv, sd, p + "ca.", n_heads)
    h = layer_norm(q, sd[p + "mlp_ln.weight"], sd[p + "mlp_ln.bias"])
    h = gelu(h @ sd[p + "mlp.0.weight"].t() + sd[p + "mlp.0.bias"])
    return q + (h @ sd[p + "mlp.3.weight"].t() + sd[p + "mlp.3.bias"])


def lidar_grid(H: int, W: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """vat_lidar.py:127-185: geom [HW,5] = (x, y, r, sin t, cos t), sid [HW] in 0..5."""
    yv, xv = torch.meshgrid(torch.linspace(-1.0, 1.0, H), torch.linspace(-1.0, 1.0, W), indexing="ij")
    r = torch.clamp((xv ** 2 + yv ** 2).sqrt(), 0.0, 1.0)
    th = torch.atan2(yv, xv)
    geom = torch.stack((xv, yv, r, torch.sin(th), torch.cos(th)), dim=-1).view(H * W, 5)
    ft = th.reshape(-1)
    pi = math.pi
    sid = torch.empty(H * W, dtype=torch.long)
    sid[(ft >= pi / 3) & (ft < 2 * pi / 3)] = 0
    sid[(ft >= 0.0) & (ft < pi / 3)] = 1
    sid[(ft >= 2 * pi / 3) & (ft <= pi)] = 2
    sid[(ft >= -2 * pi / 3) & (ft < -pi / 3)] = 3
    sid[(ft >= -pi / 3) & (ft < 0.0)] = 4
    sid[(ft >= -pi) & (ft < -2 * pi / 3)] = 5
    return geom, sid


def vat_lidar_tokens(bev: torch.Tensor, sd: SD) -> torch.Tensor:
    """vat_lidar.py:212-248: BEV [B,C,H,W] -> K/V tokens [B,HW,d]."""
    B, C, H, W = bev.shape
    x = F.conv2d(bev, sd["refine.0.weight"], sd["refine.0.bias"], padding=1, groups=C)
    x = gelu(x)
    d = sd["proj.weight"].shape[0]
    x = torch.einsum("bchw,dc->bhwd", x, sd["proj.weight"].view(d, C)) + sd["proj.bias"]
    x = x.reshape(B, H * W, d)
    x = layer_norm(x, sd["norm_tokens.weight"], sd["norm_tokens.bias"])
    geom, sid = lidar_grid(H, W)
    pe = gelu(geom @ sd["geo_mlp.0.weight"].t() + sd["geo_mlp.0.bias"]) @ sd["geo_mlp.2.weight"].t() + sd["geo_mlp.2.bias"]
    x = x + pe.unsqueeze(0)
    x = x + sd["view_embed"][sid].unsqueeze(0)
    return x


def vat_lidar(bev: torch.Tensor, sd: SD, n_heads: int) -> torch.Tensor:
    """vat_lidar.py:187-304."""
    B = bev.shape[0]
    x = vat_lidar_tokens(bev, sd)
    nq = sd["query"].shape[0]
    per = nq // NUM_VIEWS
    q = sd["query"] + sd["view_embed"].repeat_interleave(per, dim=0)
    q = q.unsqueeze(0).expand(B, -1, -1)
    n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(n_layers):
        q = vat_block(q, x, sd, f"blocks.{i}.", n_heads)
    q = layer_norm(q, sd["final_ln.weight"], sd["final_ln.bias"])
    h = layer_norm(q, sd["post.0.weight"], sd["post.0.bias"])
    h = gelu(h @ sd["post.1.weight"].t() + sd["post.1.bias"])
    return h @ sd["post.4.weight"].t() + sd["post.4.bias"]


def vat_vision(kv: torch.Tensor, sd: SD, n_heads: int) -> torch.Tensor:
    """vat_vision.py:140-235."""
    B = kv.shape[0]
    q = sd["query"]
    if "view_query_embed" in sd:
        per = q.shape[0] // NUM_VIEWS
        q = q + sd["view_query_embed"].repeat_interleave(per, dim=0)
    q = q.unsqueeze(0).expand(B, -1, -1)
    n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(n_layers):
        q = vat_block(q, kv, sd, f"blocks.{i}.", n_heads)
    q = layer_norm(q, sd["final_ln.weight"], sd["final_ln.bias"])
    h = layer_norm(q, sd["post.0.weight"], sd["post.0.bias"])
    h = gelu(h @ sd["post.1.weight"].t() + sd["post.1.bias"])
    q = h @ sd["post.4.weight"].t() + sd["post.4.bias"]
    h = layer_norm(q, sd["proj.0.weight"], sd["proj.0.bias"])
    h = gelu(h @ sd["proj.1.weight"].t() + sd["proj.1.bias"])
    h = h @ sd["proj.4.weight"].t() + sd["proj.4.bias"]
    return layer_norm(h, sd["proj.5.weight"], sd["proj.5.bias"])


def vision_adapter(views: Sequence[torch.Tensor], sd: SD) -> torch.Tensor:
    """vision_adapter.py:68-145 (eval): LN(t + view_embed[v]) per view, concatenated."""
    if len(views) != NUM_VIEWS:
        raise ValueError(f"Expected {NUM_VIEWS} views, got {len(views)}")
    out = [layer_norm(t + sd["view_embed"][v].unsqueeze(0), sd["norm.weight"], sd["norm.bias"]) for v, t in enumerate(views)]
    return torch.cat(out, dim=0)


def deepencoder_fuse(clip_tokens: torch.Tensor, sam_feat: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """deepencoder/deepencoder_infer.py:505-511 + build_linear.py:18-19: cat(clip[:,1:], sam) -> Linear."""
    sam = sam_feat.flatten(2).permute(0, 2, 1)
    return torch.cat((clip_tokens[:, 1:], sam), dim=-1) @ w.t() + b


# --------------------------------------------------------------------------------------
# a15: prefix assembly (encoder-decoder/training/core/validation.py:105-158 eval order) and a
# Qwen2-style stand-in head (the real head is `Qwen/Qwen2.5-0.5B`, fetched by name ->
# unavailable offline).  The stand-in follows transformers' Qwen2ForCausalLM forward
# (RMSNorm, rotary GQA self-attention with q/k/v bias, SwiGLU, tied lm_head, shifted CE loss) and is
# pinned against transformers.Qwen2ForCausalLM (random init) in tools/make_goldens.py.
# --------------------------------------------------------------------------------------
def assemble_prefix(prefix_vision, prefix_lidar, e_special: torch.Tensor, e_prompt: torch.Tensor,
                    e_answer: torch.Tensor, answer_ids: torch.Tensor, prefix_scale: float = 0.2):
    """e_special rows: <vision_start>, <vision_end>, <lidar_start>, <lidar_end>.
    Returns inputs_embeds [B,L,d], attention_mask [B,L] (ones), labels [B,L] (-100 except answer)."""
    pieces = []
    B = e_prompt.shape[0]
    if prefix_vision is not None:
        pieces += [e_special[0].expand(B, 1, -1), prefix_vision * prefix_scale, e_special[1].expand(B, 1, -1)]
    if prefix_lidar is not None:
        pieces += [e_special[2].expand(B, 1, -1), prefix_lidar * prefix_scale, e_special[3].expand(B, 1, -1)]
    pieces.append(e_prompt)
    inp = torch.cat(pieces + [e_answer], dim=1)
    L = inp.shape[1]
    labels = torch.full((B, L), -100, dtype=torch.long)
    labels[:, -answer_ids.shape[1]:] = answer_ids
    attn = torch.ones((B, L), dtype=torch.long)
    return inp, attn, labels


def rms_norm(x, w, eps):
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def _rope(L: int, dh: int, theta: float):
    inv = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    fr = torch.outer(torch.arange(L, dtype=torch.float32), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def qwen2_head(inputs_embeds: torch.Tensor, sd: SD, cfg: dict, labels: Optional[torch.Tensor] = None):
    """cfg: n_layers, n_heads, n_kv_heads, rms_eps, rope_theta.  Returns (logits [B,L,V], loss|None)."""
    x = inputs_embeds
    B, L, d = x.shape
    H, Hk = cfg["n_heads"], cfg["n_kv_heads"]
    dh = d // H
    cos, sin = _rope(L, dh, cfg.get("rope_theta", 10000.0))
    causal = torch.full((L, L), float("-inf")).triu(1)
    for i in range(cfg["n_layers"]):
        p = f"model.layers.{i}."
        h = rms_norm(x, sd[p + "input_layernorm.weight"], cfg["rms_eps"])
        q = (h @ sd[p + "self_attn.q_proj.weight"].t() + sd[p + "self_attn.q_proj.bias"]).view(B, L, H, dh).transpose(1, 2)
        k = (h @ sd[p + "self_attn.k_proj.weight"].t() + sd[p + "self_attn.k_proj.bias"]).view(B, L, Hk, dh).transpose(1, 2)
        v = (h @ sd[p + "self_attn.v_proj.weight"].t() + sd[p + "self_attn.v_proj.bias"]).view(B, L, Hk, dh).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        k = k.repeat_interleave(H // Hk, dim=1)
        v = v.repeat_interleave(H // Hk, dim=1)
        o = sdp_attention(q, k, v, causal).transpose(1, 2).reshape(B, L, d)
        x = x + o @ sd[p + "self_attn.o_proj.weight"].t()
        h = rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg["rms_eps"])
        g = F.silu(h @ sd[p + "mlp.gate_proj.weight"].t()) * (h @ sd[p + "mlp.up_proj.weight"].t())
        x = x + g @ sd[p + "mlp.down_proj.weight"].t()
    x = rms_norm(x, sd["model.norm.weight"], cfg["rms_eps"])
    logits = x @ sd["model.embed_tokens.weight"].t()
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), labels[:, 1:].reshape(-1),
                               ignore_index=-100)
    return logits, loss


def qwen2_generate(inputs_embeds: torch.Tensor, sd: SD, cfg: dict, max_new_tokens: int, eos_token_id: Optional[int] = None,
                   pad_token_id: int = 0):
    """Greedy decoding as `base_model.generate(inputs_embeds=, do_sample=False, num_beams=1)` runs it
    (inference/inference_engine.py:283-296): returns (new token ids [B, n], per-step logits [B, n, V]).  Every step
    recomputes the whole sequence (no KV cache -- the oracle is the checker, not the thing measured).  A finished
    sequence emits pad_token_id; decoding stops when every sequence has emitted eos_token_id."""
    x = inputs_embeds
    B = x.shape[0]
    E = sd["model.embed_tokens.weight"]
    ids, scores = [], []
    unfinished = torch.ones(B, dtype=torch.bool)
    for _ in range(max_new_tokens):
        logits, _ = qwen2_head(x, sd, cfg)
        step = logits[:, -1]
        nxt = step.argmax(-1)
        if eos_token_id is not None:
            nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad_token_id))
        ids.append(nxt)
        scores.append(step)
        if eos_token_id is not None:
            unfinished = unfinished & (nxt != eos_token_id)
            if not bool(unfinished.any()):
                break
        x = torch.cat((x, E[nxt].unsqueeze(1)), dim=1)
    return torch.stack(ids, dim=1), torch.stack(scores, dim=1)
