"""Training-side route of the fusion modules (seam 1: `.train()` / autograd are part of the module contract --
"the build may ship inference-only kernels but then must fall back to torch ops when torch.is_grad_enabled()";
reference call sites: training/core/trainer.py:549, 581, 594).

The HIP kernels are forward-only.  A call that needs a gradient (grad mode on and a trainable parameter or input) or runs in
train() mode (dropout active) is therefore evaluated by the torch modules the classes already hold as parameter containers
(nn.LayerNorm / nn.MultiheadAttention / nn.Conv2d / nn.Sequential: same state_dict, same initialisation), so autograd, DDP and
AdamW see ordinary torch graphs.  This is NOT a fallback of the inference path: eval() + no_grad() calls never come here, and they
fail loudly without liblvq_hip.so or with CPU tensors (fusion._HipModule._guard).  Nothing in here touches oracle/.

The route is chosen by what the CALLER asks for, never by what is installed; the first call through it warns once per class,
because an inference loop that merely forgot torch.no_grad() would otherwise lose the kernels silently.
"""
from __future__ import annotations

import warnings
from typing import List

import torch

_warned = set()


def wanted(module: torch.nn.Module, *tensors) -> bool:
    """True when this call must be differentiable or stochastic: train() mode, or grad mode with something trainable in reach."""
    if module.training:
        return True
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


def note(module: torch.nn.Module):
    name = type(module).__name__
    if name not in _warned:
        _warned.add(name)
        warnings.warn(f"{name}: this call runs on the autograd route (torch ops), not on the MI355X kernels, because it is in train() mode or "
                      "needs gradients; wrap inference calls in torch.no_grad() on an eval() module", stacklevel=3)


def vat_block(m, q: torch.Tensor, kv: torch.Tensor) -> torch.Tensor:
    """Pre-LN self-attention, cross-attention onto kv, MLP, each with its residual (vat_blocks.py:36-47)."""
    h = m.sa_ln(q)
    q = q + m.sa(h, h, h, need_weights=False)[0]
    q = q + m.ca(m.ca_ln(q), kv, kv, need_weights=False)[0]
    return q + m.mlp(m.mlp_ln(q))


def vat_lidar(m, bev: torch.Tensor) -> torch.Tensor:
    """BEV [B, C, H, W] -> resampled tokens [B, nq, d] (vat_lidar.py:187-304): depthwise refine, 1x1 projection, token LayerNorm,
    geometric + sector embeddings on the keys; learned queries + their sector embedding through the blocks; final LN and post MLP."""
    B, C, H, W = bev.shape
    x = m.proj(m.refine(bev)).flatten(2).transpose(1, 2)            # [B, HW, d]
    x = m.norm_tokens(x)
    geom, sid = m._grid(H, W, bev.device)
    x = x + m.geo_mlp(geom).unsqueeze(0) + m.view_embed[sid.long()].unsqueeze(0)
    q = (m.query + m.view_embed.repeat_interleave(m.nq_per_view, dim=0)).unsqueeze(0).expand(B, -1, -1)
    for blk in m.blocks:
        q = vat_block(blk, q, x)
    return m.post(m.final_ln(q))


def vat_vision(m, kv_tokens: torch.Tensor) -> torch.Tensor:
    """Image tokens [B, N, d_in] -> [B, N / cf, d_model] (vat_vision.py:140-235)."""
    B = kv_tokens.shape[0]
    q = m.query
    if m.use_per_view_query and m.nq_per_view > 0:
        q = q + m.view_query_embed.repeat_interleave(m.nq_per_view, dim=0)
    q = q.unsqueeze(0).expand(B, -1, -1)
    for blk in m.blocks:
        q = vat_block(blk, q, kv_tokens)
    return m.proj(m.post(m.final_ln(q)))


def vision_adapter(m, views_tokens: List[torch.Tensor]) -> torch.Tensor:
    """dropout(LayerNorm(t + view_embed[v])) per view, concatenated (vision_adapter.py:120-133)."""
    return torch.cat([m.dropout(m.norm(t + m.view_embed[v].unsqueeze(0))) for v, t in enumerate(views_tokens)], dim=0)
