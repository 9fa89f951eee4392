#!/usr/bin/env python3
"""Which bf16 rounding points of the fusion path cost how much of the 1e-3 fused-token tolerance?  (CPU only, no GPU.)

Replays the bench workload (BASELINE configs[1]: one 32 768-point Dist-C scene, 512x512 BEV, d = 768, 12 heads, the bench's seeded
weights) with the oracle's fp32 arithmetic and rounds ONE group of operands to bf16 at a time, exactly where the HIP kernels
round them (GEMM / attention operands; accumulation, LayerNorm / softmax statistics and residuals stay fp32).  Prints the max abs
error of the fused tokens against the unrounded run for every group, for everything rounded (= precision "bf16") and for the
candidate mixed mode (= precision "mixed": plain bf16 on the 262 144-key K|V stream, hi+lo on the 576-row query side).

    python tools/precision_study.py [--scenes 1] [--seed 1100]

The result table is recorded in DESIGN.md section 3.3.  Test infrastructure: imports oracle/.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import lidar_oracle as LO
from oracle import vat_oracle as VO

ACTIVE = set()
F16 = set()          # tags rounded to fp16 (11-bit significand) instead of bf16 (8-bit)


def r(tag: str, x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (RNE) when the tag's group is active; to fp16 when the tag is in F16."""
    if tag in F16:
        return x.half().float()
    return x.bfloat16().float() if tag in ACTIVE else x


def lin(tag_a, a, tag_w, w, b=None):
    y = r(tag_a, a) @ r(tag_w, w).t()
    return y if b is None else y + b


def mha(xq, xkv_proj, sd, p, h, g, kv_tokens=None):
    """g: tag prefix of this attention.  xkv_proj: precomputed (K, V) or None (then kv_tokens are projected here)."""
    d = xq.shape[-1]
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = lin(g + ".q_in", xq, g + ".q_w", w[0:d], b[0:d])
    if xkv_proj is None:
        k = lin(g + ".kv_in", kv_tokens, g + ".kv_w", w[d:2 * d], b[d:2 * d])
        v = lin(g + ".kv_in", kv_tokens, g + ".kv_w", w[2 * d:], b[2 * d:])
    else:
        k, v = xkv_proj
    dh = d // h
    nq, nk = q.shape[0], k.shape[0]
    q = r(g + ".q", q * (1.0 / math.sqrt(dh)))
    k = r(g + ".kv", k)
    v = r(g + ".kv", v)
    out = torch.empty(nq, d)
    for i in range(h):
        s = q[:, i * dh:(i + 1) * dh] @ k[:, i * dh:(i + 1) * dh].t()
        s = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        den = s.sum(dim=-1, keepdim=True)
        out[:, i * dh:(i + 1) * dh] = (r(g + ".p", s) @ v[:, i * dh:(i + 1) * dh]) / den
    return lin(g + ".o", out, g + ".o_w", sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def vat_block(q, kv, sd, p, h, g):
    qn = VO.layer_norm(q, sd[p + "sa_ln.weight"], sd[p + "sa_ln.bias"])
    q = q + mha(qn, None, sd, p + "sa.", h, g + ".sa", kv_tokens=qn)
    qn = VO.layer_norm(q, sd[p + "ca_ln.weight"], sd[p + "ca_ln.bias"])
    q = q + mha(qn, None, sd, p + "ca.", h, g + ".ca", kv_tokens=kv)
    hn = VO.layer_norm(q, sd[p + "mlp_ln.weight"], sd[p + "mlp_ln.bias"])
    h1 = VO.gelu(lin(g + ".mlp", hn, g + ".mlp", sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    return q + lin(g + ".mlp", h1, g + ".mlp", sd[p + "mlp.3.weight"], sd[p + "mlp.3.bias"])


def forward(bev_gelu, pe, sd_l, sd_f, patches, h):
    """bev_gelu [HW, C] = GELU(dwconv(bev)) tokens (fp32, exact); returns fused [nq, d]."""
    d = sd_l["proj.weight"].shape[0]
    C = bev_gelu.shape[1]
    x = lin("lidar.t", bev_gelu, "lidar.w_proj", sd_l["proj.weight"].view(d, C), sd_l["proj.bias"])
    x = VO.layer_norm(x, sd_l["norm_tokens.weight"], sd_l["norm_tokens.bias"]) + pe
    nq = sd_l["query"].shape[0]
    q = sd_l["query"] + sd_l["view_embed"].repeat_interleave(nq // 6, dim=0)
    q = vat_block(q, x, sd_l, "blocks.0.", h, "lidar")
    q = VO.layer_norm(q, sd_l["final_ln.weight"], sd_l["final_ln.bias"])
    hn = VO.layer_norm(q, sd_l["post.0.weight"], sd_l["post.0.bias"])
    h1 = VO.gelu(lin("lidar.post", hn, "lidar.post", sd_l["post.1.weight"], sd_l["post.1.bias"]))
    lt = lin("lidar.post", h1, "lidar.post", sd_l["post.4.weight"], sd_l["post.4.bias"])
    return vat_block(lt, patches, sd_f, "", h, "fuse"), lt


GROUPS = {
    # the 262 144-row K|V stream of VATLiDAR's cross-attention
    "stream: conv tokens t (A of proj)": ["lidar.t"],
    "stream: W_proj": ["lidar.w_proj"],
    "stream: BEV tokens x (A of K|V proj)": ["lidar.ca.kv_in"],
    "stream: W_k|W_v": ["lidar.ca.kv_w"],
    "stream: K, V": ["lidar.ca.kv"],
    "stream: P (softmax numerators)": ["lidar.ca.p"],
    # the 576-row query side
    "query: ca_ln(q), W_q": ["lidar.ca.q_in", "lidar.ca.q_w"],
    "query: Q (scaled)": ["lidar.ca.q"],
    "query: attention out, W_o": ["lidar.ca.o", "lidar.ca.o_w"],
    "query: self-attention (all operands)": ["lidar.sa.q_in", "lidar.sa.q_w", "lidar.sa.kv_in", "lidar.sa.kv_w", "lidar.sa.q", "lidar.sa.kv",
                                             "lidar.sa.p", "lidar.sa.o", "lidar.sa.o_w"],
    "query: MLP": ["lidar.mlp"],
    "query: post head": ["lidar.post"],
    "fusion VATBlock (all operands)": ["fuse.sa.q_in", "fuse.sa.q_w", "fuse.sa.kv_in", "fuse.sa.kv_w", "fuse.sa.q", "fuse.sa.kv", "fuse.sa.p",
                                       "fuse.sa.o", "fuse.sa.o_w", "fuse.ca.q_in", "fuse.ca.q_w", "fuse.ca.kv_in", "fuse.ca.kv_w", "fuse.ca.q",
                                       "fuse.ca.kv", "fuse.ca.p", "fuse.ca.o", "fuse.ca.o_w", "fuse.mlp"],
}
STREAM = [k for k in GROUPS if k.startswith("stream")]
RANDOM_PER_KEY = ["stream: BEV tokens x (A of K|V proj)", "stream: K, V", "stream: P (softmax numerators)"]
MIXED_CANDIDATES = {
    "mixed: x, K, V, P plain bf16; everything else exact": RANDOM_PER_KEY,
    "mixed + plain Q": RANDOM_PER_KEY + ["query: Q (scaled)"],
    "whole stream plain (t, W_proj, x, W_kv, K, V, P); rest exact": STREAM,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1100)
    ap.add_argument("--only-mixed", action="store_true", help="skip the per-group table")
    ap.add_argument("--only-q16", action="store_true", help="only the fp16-Q candidate (stream plain bf16, Q single fp16, K fp16)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    torch.set_grad_enabled(False)
    from lidar_vision_vqa_amd import fusion, lidar, pipeline as P, synth
    cfg = P.PipelineConfig()
    rng = list(cfg.pc_range)
    # same weights as FusionPipeline (seed + module index)
    pv = lidar.PillarVFE(P.Cfg(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=cfg.pillar_filters), 4,
                         list(cfg.voxel_pillar), rng)
    vl = fusion.VATLiDAR(cfg.pillar_filters[-1], cfg.d_model, cfg.n_queries, cfg.n_layers, cfg.n_heads)
    fb = fusion.VATBlock(cfg.d_model, cfg.n_heads, 4 * cfg.d_model, 0.1)
    sds = []
    for i, m in enumerate((pv, vl, fb)):
        synth.load_seeded(m, cfg.weight_seed + i)
        sds.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    sd_p, sd_l, sd_f = sds
    pts = synth.scene_points(cfg.dist, cfg.n_points, args.seed)
    patches = torch.from_numpy(synth.image_patches(cfg.n_patches, cfg.d_model, 2000 + args.seed))
    pts = pts[LO.mask_points_by_range(pts, rng)]
    v, c, n = LO.VoxelGenerator(cfg.voxel_pillar, rng, 4, cfg.t_pillar, cfg.max_pillars).generate(pts)
    bp = LO.collate_batch([dict(voxels=v, voxel_coords=c, voxel_num_points=n)])
    pf = LO.pillar_vfe(bp["voxels"], bp["voxel_num_points"], bp["voxel_coords"], sd_p, cfg.voxel_pillar, rng, cfg.pillar_filters)
    hh, ww = cfg.bev_hw
    bev = LO.pointpillar_scatter(pf, bp["voxel_coords"], ww, hh)
    C = bev.shape[1]
    t = VO.gelu(torch.nn.functional.conv2d(bev, sd_l["refine.0.weight"], sd_l["refine.0.bias"], padding=1, groups=C))
    t = t[0].permute(1, 2, 0).reshape(hh * ww, C).contiguous()
    geom, sid = VO.lidar_grid(hh, ww)
    pe = VO.gelu(geom @ sd_l["geo_mlp.0.weight"].t() + sd_l["geo_mlp.0.bias"]) @ sd_l["geo_mlp.2.weight"].t() + sd_l["geo_mlp.2.bias"]
    pe = pe + sd_l["view_embed"][sid]

    def run(groups):
        ACTIVE.clear()
        for gname in groups:
            ACTIVE.update(GROUPS[gname])
        t0 = time.time()
        out, lt = forward(t, pe, sd_l, sd_f, patches, cfg.n_heads)
        return out, lt, time.time() - t0

    ref, ref_lt, dt = run([])
    if args.only_q16:
        F16.update(["lidar.ca.q"])
        out, lt, dt = run(RANDOM_PER_KEY)
        F16.clear()
        print(f"seed {args.seed}: stream bf16, Q single fp16: fused err {(out - ref).abs().max().item():.3e} (|ref| max {ref.abs().max():.3f})", flush=True)
        out, lt, dt = run(RANDOM_PER_KEY)
        print(f"seed {args.seed}: stream bf16, Q exact (= mixed):  fused err {(out - ref).abs().max().item():.3e}", flush=True)
        return
    print(f"reference: fused absmax {ref.abs().max():.4f}, lidar tokens absmax {ref_lt.abs().max():.4f}  ({dt:.1f} s)", flush=True)
    rows = []
    for gname in ([] if args.only_mixed else GROUPS):
        out, lt, dt = run([gname])
        rows.append((gname, (out - ref).abs().max().item(), (lt - ref_lt).abs().max().item()))
        print(f"{gname:60s} fused err {rows[-1][1]:.3e}   lidar-token err {rows[-1][2]:.3e}  ({dt:.1f} s)", flush=True)
    for name, groups in MIXED_CANDIDATES.items():
        out, lt, dt = run(groups)
        print(f"{name:60s} fused err {(out - ref).abs().max().item():.3e}   lidar-token err {(lt - ref_lt).abs().max().item():.3e}", flush=True)
    # fp16 forms (11-bit significand, same MFMA rate as bf16, ONE product instead of the two of hi + lo): which operands can take it?
    F16_CANDIDATES = {
        "x, P bf16; Q, K, V fp16; rest exact": (["lidar.ca.q", "lidar.ca.kv"], ["stream: BEV tokens x (A of K|V proj)", "stream: P (softmax numerators)"]),
        "P, V... bf16 stream, Q fp16 (single), rest exact": (["lidar.ca.q"], RANDOM_PER_KEY),
        "x, K, V, P bf16; W_k|W_v fp16 (single); rest exact": (["lidar.ca.kv_w"], RANDOM_PER_KEY),
        "x, W_k|W_v, K, Q fp16; V, P bf16; rest exact": (["lidar.ca.kv_in", "lidar.ca.kv_w", "lidar.ca.q"], ["stream: K, V", "stream: P (softmax numerators)"]),
        "t, W_proj, x, W_k|W_v, Q fp16; K, V, P bf16; rest exact": (["lidar.t", "lidar.w_proj", "lidar.ca.kv_in", "lidar.ca.kv_w", "lidar.ca.q"],
                                                                   ["stream: K, V", "stream: P (softmax numerators)"]),
        "t, W_proj, x, W_k|W_v, K, V, Q fp16; P bf16; rest exact": (["lidar.t", "lidar.w_proj", "lidar.ca.kv_in", "lidar.ca.kv_w", "lidar.ca.q", "lidar.ca.kv"],
                                                                   ["stream: P (softmax numerators)"]),
    }
    for name, (f16, groups) in F16_CANDIDATES.items():
        F16.update(f16)
        out, lt, dt = run(groups)
        F16.clear()
        print(f"{name:60s} fused err {(out - ref).abs().max().item():.3e}   lidar-token err {(lt - ref_lt).abs().max().item():.3e}", flush=True)
    out, lt, dt = run(list(GROUPS))
    print(f"{'everything bf16':60s} fused err {(out - ref).abs().max().item():.3e}   lidar-token err {(lt - ref_lt).abs().max().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
