#!/usr/bin/env python3
"""Kernel micro-benchmarks on the MI355X (HIP events, random data, interleaved rounds in ONE process --
cdna_hip_programming.md rules 24/25).  Usage on the GPU box:

    python tools/bench_kernels.py gemm attn norm vox
"""
from __future__ import annotations

import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from lidar_vision_vqa_amd import lidar, ops, synth  # noqa: E402

DEV = torch.device("cuda:0")
torch.set_grad_enabled(False)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return min(ts), float(np.median(ts))


def bench_gemm():
    shapes = [(4 * 262144, 1536, 768, "VATLiDAR K|V proj (S=4)"), (262144, 1536, 768, "K|V proj (S=1)"),
              (32768, 768, 768, "headline Q / out proj"), (4 * 262144, 768, 64, "1x1 conv proj (S=4)"),
              (8192, 8192, 8192, "square 8k"), (4096, 4096, 4096, "square 4k"), (2304, 3072, 768, "MLP up (S=4,nq=576)"),
              (2304, 768, 3072, "MLP down"), (784, 1536, 768, "patch K|V (S=4)")]
    print(f"{'shape':>28} {'what':>28} {'mode':>7} {'ms(min)':>9} {'TFLOP/s':>9} {'frac':>6}")
    for m, n, k, what in shapes:
        a = torch.randn(m, k, device=DEV)
        w = torch.randn(n, k, device=DEV) * 0.05
        bias = torch.randn(n, device=DEV)
        for split in (False, True):
            ab, wb = ops.cast(a, split), ops.cast(w, split)
            t, _ = timeit(lambda: ops.linear(ab, wb, bias, out_bf=True), iters=10)
            fl = 2.0 * m * n * k
            print(f"{str((m, n, k)):>28} {what:>28} {'x3' if split else 'bf16':>7} {t:9.4f} {fl / t / 1e9:9.1f} {fl / t / 1e9 / 2500:6.3f}")
        del a, w


def bench_attn():
    cases = [(4, 12, 576, 262144, 64, "VATLiDAR ca (S=4)"), (1, 12, 32768, 196, 64, "headline 32k x 196"),
             (4, 12, 576, 196, 64, "fusion ca"), (4, 12, 576, 576, 64, "self-attn"), (16, 16, 2048, 2048, 128, "flash ref shape")]
    print(f"{'B,H,Nq,Nkv,dh':>28} {'what':>22} {'mode':>6} {'ms':>9} {'TFLOP/s':>9}")
    for B, H, nq, nkv, dh, what in cases:
        d = H * dh
        q = torch.randn(B * nq, d, device=DEV)
        kv = torch.randn(B * nkv, 2 * d, device=DEV)
        for split in (False, True):
            qb, kvb = ops.cast(q, split), ops.cast(kv, split)
            vsl = (kvb[0][:, d:], None if kvb[1] is None else kvb[1][:, d:])
            fn = lambda: ops.attention(qb, kvb, vsl, batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=nkv, dh=dh,
                                       q_strides=(nq * d, d, dh), k_strides=(nkv * 2 * d, 2 * d, dh),
                                       v_strides=(nkv * 2 * d, 2 * d, dh), scale=1 / math.sqrt(dh))
            t, _ = timeit(fn, iters=5)
            fl = 4.0 * B * nq * nkv * d
            print(f"{str((B, H, nq, nkv, dh)):>28} {what:>22} {'x3' if split else 'bf16':>6} {t:9.4f} {fl / t / 1e9:9.1f}")
        del q, kv


def bench_norm():
    for rows, d in [(4 * 262144, 768), (2304, 768)]:
        x = torch.randn(rows, d, device=DEV)
        g, b = torch.randn(d, device=DEV), torch.randn(d, device=DEV)
        post = torch.randn(262144, d, device=DEV) if rows > 262144 else None
        t, _ = timeit(lambda: ops.layernorm(x, g, b, 1e-5, False, post=post), iters=5)
        by = rows * d * (4 + 2) + (rows * d * 4 if post is not None else 0)
        print(f"layernorm rows={rows} d={d}: {t:.4f} ms  {by / t / 1e6:.0f} GB/s algorithmic")


def bench_projln():
    m, n, k = 4 * 262144, 768, 64
    a = torch.randn(m, k, device=DEV); w = torch.randn(n, k, device=DEV) * 0.1
    bias, g, b = torch.randn(n, device=DEV), torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    post = torch.randn(262144, n, device=DEV)
    for split in (False, True):
        ab, wb = ops.cast(a, split), ops.cast(w, split)
        t, _ = timeit(lambda: ops.linear_ln(ab, wb, bias, g, b, 1e-5, post=post), iters=5)
        def unfused():
            x32, _ = ops.linear(ab, wb, bias, out_f32=True)
            return ops.layernorm(x32, g, b, 1e-5, split, post=post)
        t2, _ = timeit(unfused, iters=5)
        print(f"proj+LN+PE (M={m}, N={n}, K={k}) {'x3' if split else 'bf16'}: fused {t:.3f} ms, unfused {t2:.3f} ms")


def bench_dwconv():
    for S, C, H, W in [(4, 64, 512, 512)]:
        x = torch.randn(S, C, H, W, device=DEV)
        w9 = torch.randn(C, 9, device=DEV) * 0.3
        b = torch.randn(C, device=DEV)
        for split in (False, True):
            t, _ = timeit(lambda: ops.dwconv3x3_gelu(x, w9, b, split), iters=10)
            by = S * C * H * W * (4 + (4 if split else 2))
            print(f"dwconv3x3+gelu S={S} C={C} {H}x{W} {'x3' if split else 'bf16'}: {t * 1e3:.1f} us  {by / t / 1e6:.0f} GB/s algorithmic")


def bench_vox():
    rng = list(synth.PC_RANGE_NUSC)
    for S, n, vs, T, mv, what in [(8, 65536, synth.VOXEL_01, 10, 160000, "cfg-3 0.1m"), (1, 32768, synth.VOXEL_01, 10, 60000, "cfg-2 0.1m"),
                                  (64, 32768, synth.VOXEL_01, 10, 60000, "64 scenes 0.1m"), (8, 65536, synth.VOXEL_PILLAR, 20, 30000, "pillars")]:
        scenes = [synth.scene_points("C", n, 1010 + i) for i in range(min(S, 8))]
        scenes = (scenes * ((S + 7) // 8))[:S]
        pts = torch.from_numpy(np.concatenate(scenes)).to(DEV)
        off = torch.tensor(np.concatenate(([0], np.cumsum([len(s) for s in scenes]))), dtype=torch.int32, device=DEV)
        gen = lidar.VoxelGeneratorWrapper(vs, rng, 4, T, mv)
        out = gen.generate_batch_device(pts, off, S)
        M = int(out[3][-1])
        t, _ = timeit(lambda: gen.generate_batch_device(pts, off, S), iters=10)
        by = 16 * pts.shape[0] + M * (4 * T * 4 + 20)
        print(f"hard voxelise {what}: S={S} N={pts.shape[0]} M={M}: {t * 1e3:.1f} us  {by / t / 1e6:.0f} GB/s algorithmic ({by / t / 1e6 / 8000:.3f} of 8 TB/s)")
        mvfe = lidar.MeanVFE(None, 4)
        t2, _ = timeit(lambda: mvfe.forward_device(out[0], out[2], out[3][S:]), iters=10)
        print(f"   MeanVFE: {t2 * 1e3:.1f} us  {(M * (4 * T * 4 + 4 + 16)) / t2 / 1e6:.0f} GB/s")
        t4, _ = timeit(lambda: gen.generate_mean_device(pts, off, S), iters=10)
        bm = 16 * pts.shape[0] + M * (4 * 4 + 16 + 4)
        print(f"   fused voxelise->mean (no padded tensor): {t4 * 1e3:.1f} us vs {(t + t2) * 1e3:.1f} us for the pair; {bm / t4 / 1e6:.0f} GB/s algorithmic "
              f"({bm / t4 / 1e6 / 8000:.3f} of 8 TB/s)")
        bp = torch.cat((torch.repeat_interleave(torch.arange(S, device=DEV, dtype=torch.float32), torch.tensor([len(s) for s in scenes], device=DEV)).unsqueeze(1), pts), 1).contiguous()
        grid = lidar.grid_size_from(rng, vs)
        ndim = 3 if grid[2] > 1 else 2
        if S * grid[0] * grid[1] * grid[2] >= 2 ** 31:
            continue                      # the dynamic key is int32 (reference quirk): not representable
        t3, _ = timeit(lambda: lidar._dynamic_voxelize(bp, S, rng, vs, grid, ndim), iters=10)
        print(f"   dynamic voxelise: {t3 * 1e3:.1f} us  {(20 * bp.shape[0] + M * 16) / t3 / 1e6:.0f} GB/s algorithmic")


def bench_bev():
    """SURVEY 8f rows f2/f3 at the reference's true BEV grid (C=128, 180x180): HBM-bound, algorithmic bytes per call."""
    from lidar_vision_vqa_amd import bev
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_bev_bridge import synth_sparse
    B = 16
    h16 = torch.randn(B, 128, 180, 180, device=DEV).half()
    out = torch.empty(h16.shape, dtype=torch.float32, device=DEV)
    t, _ = timeit(lambda: bev.f16_to_f32(h16, out), iters=20)
    by = h16.numel() * 6
    print(f"f16->f32 BEV up-cast [{B},128,180,180]: {t * 1e3:.1f} us  {by / t / 1e6:.0f} GB/s ({by / t / 1e6 / 8000:.3f} of 8 TB/s)")
    for batch, m in [(4, 120000), (16, 480000)]:
        feats, idx = synth_sparse(4, batch, 5, 180, 180, m, 128)
        x = bev.SparseTensor(torch.from_numpy(feats).to(DEV), torch.from_numpy(idx).to(DEV), (5, 180, 180), batch)
        y = bev.bev_out(x)
        m1, m2 = x.features.shape[0], y.features.shape[0]
        t, _ = timeit(lambda: bev.bev_out(x), iters=10)
        by = m1 * (16 + 512) + m2 * (12 + 512)            # indices + features in, unique rows + summed features out
        print(f"bev_out z-merge B={batch} rows {m1} -> {m2}, C=128: {t * 1e3:.1f} us  {by / t / 1e6:.0f} GB/s algorithmic "
              f"({by / t / 1e6 / 8000:.3f} of 8 TB/s; one host sync for M2)")
        t, _ = timeit(lambda: y.dense(), iters=10)
        by = batch * 128 * 180 * 180 * 4 + m2 * (12 + 512)
        print(f"   dense() -> [{batch},128,180,180]: {t * 1e3:.1f} us  {by / t / 1e6:.0f} GB/s algorithmic ({by / t / 1e6 / 8000:.3f} of 8 TB/s)")


def bench_ca():
    """Every cross-attention row of SURVEY 8d through VATBlock.cross_attention (ca_ln -> Q / K|V projections -> attention -> out
    projection + residual); FLOPs per scene = 4 Nq d^2 + 4 Nkv d^2 + 4 Nq Nkv d."""
    from lidar_vision_vqa_amd import fusion
    rows = [("32k pts x 196 patches (headline)", 1, 32768, 196, 768, 12), ("resampled LiDAR tokens x patches", 1, 576, 196, 768, 12),
            ("resampled LiDAR tokens x patches", 8, 576, 196, 768, 12), ("cfg-5", 8, 256, 576, 768, 12),
            ("reference-true VATLiDAR (head_dim 112)", 1, 576, 32400, 896, 8), ("reference-true VATVision (head_dim 256)", 1, 768, 1536, 2048, 8)]
    if os.environ.get("CA_ROW"):                      # one row only: the target of `rocprofv3 --kernel-trace --stats`
        rows = [rows[int(os.environ["CA_ROW"])]]
    print(f"{'row':>42} {'B':>3} {'Nq':>6} {'Nkv':>6} {'d':>5} {'h':>3} {'mode':>7} {'ms':>8} {'TFLOP/s':>8} {'frac':>6}")
    for what, B, nq, nkv, d, h in rows:
        blk = synth.load_seeded(fusion.VATBlock(d, h, 4 * d, 0.1).to(DEV).eval(), 5)
        q = torch.randn(B, nq, d, device=DEV)
        kv = torch.randn(B, nkv, d, device=DEV)
        fl = B * (4.0 * nq * d * d + 4.0 * nkv * d * d + 4.0 * nq * nkv * d)
        for prec in ("bf16", "bf16x3"):
            blk.precision = prec
            t, _ = timeit(lambda: blk.cross_attention(q, kv), iters=10)
            print(f"{what:>42} {B:>3} {nq:>6} {nkv:>6} {d:>5} {h:>3} {prec:>7} {t:8.4f} {fl / t / 1e9:8.1f} {fl / t / 1e9 / 2500:6.3f}")
        del blk, q, kv


def bench_decode():
    """SURVEY 8f f4 at the reference's true decoder size (Qwen2.5-0.5B geometry, random weights): prefill of the multimodal prompt
    (2 + 576 + 2 + 258 + 32 prompt positions) and greedy decode steps with the KV cache."""
    from lidar_vision_vqa_amd import head
    L, n_new = 870, 32
    base = head.StandInHead(151936, 896, 4864, 14, 2, 24, 1e-6, 1000000.0).to(DEV).eval()
    for p in base.parameters():
        p.data.normal_(0, 0.02)
    for B in (1, 8):
        for prec in ("bf16", "bf16x3"):
            base.precision = prec
            inp = torch.randn(B, L, 896, device=DEV) * 0.05
            base.generate(inputs_embeds=inp, max_new_tokens=4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            base.generate(inputs_embeds=inp, max_new_tokens=1)
            torch.cuda.synchronize()
            t_pre = time.perf_counter() - t0
            t0 = time.perf_counter()
            base.generate(inputs_embeds=inp, max_new_tokens=n_new)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            per = (t_all - t_pre) / (n_new - 1)
            print(f"decode {prec}: B={B} prefill L={L}: {t_pre * 1e3:.1f} ms; {per * 1e3:.2f} ms/step ({B / per:.0f} tokens/s; "
                  f"one native call per step, 24 layers x 9 dependent launches)")


def bench_attn_one():
    """One shape, few launches: the target of `rocprofv3 --pmc`."""
    B, H, nq, nkv, dh = (int(v) for v in os.environ.get("ATTN_SHAPE", "4,12,576,262144,64").split(","))
    d = H * dh
    q = torch.randn(B * nq, d, device=DEV)
    kv = torch.randn(B * nkv, 2 * d, device=DEV)
    qb, kvb = ops.cast(q, False), ops.cast(kv, False)
    vsl = (kvb[0][:, d:], None)
    fn = lambda: ops.attention(qb, kvb, vsl, batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=nkv, dh=dh, q_strides=(nq * d, d, dh),
                               k_strides=(nkv * 2 * d, 2 * d, dh), v_strides=(nkv * 2 * d, 2 * d, dh), scale=1 / math.sqrt(dh))
    t, _ = timeit(fn, iters=3, warm=1)
    print(f"attn {B,H,nq,nkv,dh}: {t:.4f} ms {4.0 * B * nq * nkv * d / t / 1e9:.1f} TFLOP/s")


def bench_gemm_one():
    m, n, k = (int(v) for v in os.environ.get("GEMM_SHAPE", "1048576,1536,768").split(","))
    a = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * 0.05
    ab, wb = ops.cast(a, False), ops.cast(w, False)
    t, _ = timeit(lambda: ops.linear(ab, wb, None, out_bf=True), iters=3, warm=1)
    print(f"gemm {m,n,k}: {t:.4f} ms {2.0 * m * n * k / t / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "norm", "vox"]
    for w in which:
        print(f"==== {w} ====")
        {"gemm": bench_gemm, "attn": bench_attn, "norm": bench_norm, "vox": bench_vox, "bev": bench_bev, "decode": bench_decode, "ca": bench_ca, "attn1": bench_attn_one,
         "gemm1": bench_gemm_one, "projln": bench_projln, "dwconv": bench_dwconv}[w]()
