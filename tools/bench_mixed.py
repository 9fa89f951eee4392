#!/usr/bin/env python3
"""Micro-benchmarks of the `mixed` precision forms on the bench shapes (4 scenes per launch), interleaved in one process:
K|V projection plain / x2w / x3, stream attention plain / q-split (3 and 2 waves per SIMD) / x3, token kernel plain / x3.

    python tools/bench_mixed.py
"""
from __future__ import annotations

import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from lidar_vision_vqa_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
torch.set_grad_enabled(False)


def timeit(fn, iters=5, warm=2, rounds=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return min(ts)


def main():
    S = int(os.environ.get("SCENES", "4"))
    hw, d, H, nq, dh = 262144, 768, 12, 576, 64
    print(f"--- K|V projection M = {S}*{hw}, N = 1536, K = 768")
    a = torch.randn(S * hw, d, device=DEV)
    w = torch.randn(2 * d, d, device=DEV) * 0.03
    bias = torch.randn(2 * d, device=DEV)
    a1, a3 = ops.cast(a, False), ops.cast(a, True)
    w1, w3 = ops.cast(w, False), ops.cast(w, True)
    del a
    fl = 2.0 * S * hw * 2 * d * d
    for name, fa, fw in (("bf16", a1, w1), ("x2w", a1, w3), ("bf16x3", a3, w3)):
        t = timeit(lambda: ops.linear(fa, fw, bias, out_bf=True))
        print(f"  {name:8s} {t:8.3f} ms   {fl / t / 1e9:8.1f} TFLOP/s algorithmic   ({fl * (1 if name == 'bf16' else 2 if name == 'x2w' else 3) / t / 1e9:8.1f} executed)")
    _, kv1 = ops.linear(a1, w1, bias, out_bf=True)
    _, kv3 = ops.linear(a3, w3, bias, out_bf=True)
    del a1, a3
    print(f"--- stream attention B = {S}, H = 12, nq = 576, nkv = {hw}")
    q = torch.randn(S * nq, d, device=DEV)
    q1, q3 = ops.cast(q, False), ops.cast(q, True)
    fl = 4.0 * S * nq * hw * d
    st = dict(batch=S, n_heads=H, n_kv_heads=H, nq=nq, nkv=hw, dh=dh, q_strides=(nq * d, d, dh), k_strides=(hw * 2 * d, 2 * d, dh),
              v_strides=(hw * 2 * d, 2 * d, dh), scale=1 / math.sqrt(dh))
    v1 = (kv1[0][:, d:], None)
    v3 = (kv3[0][:, d:], kv3[1][:, d:])
    res = {}
    for name, fq, fk, fv in (("bf16", q1, kv1, v1), ("q-split", q3, kv1, v1), ("bf16x3", q3, kv3, v3)):
        t = timeit(lambda: ops.attention(fq, fk, fv, **st), iters=3)
        res[name] = ops.to_f32(ops.attention(fq, fk, fv, **st))
        print(f"  {name:12s} {t:8.3f} ms   {fl / t / 1e9:8.1f} TFLOP/s algorithmic")
    ref = res["bf16x3"]
    for name in ("bf16", "q-split"):
        print(f"  max |{name} - bf16x3| = {(res[name] - ref).abs().max().item():.3e}   (|ref| max {ref.abs().max().item():.3f})")
    del kv1, kv3, res
    print(f"--- token kernel (1x1 conv 64 -> 768 + LayerNorm + table), M = {S}*{hw}")
    t_in = torch.randn(S * hw, 64, device=DEV)
    wp = torch.randn(d, 64, device=DEV) * 0.1
    g, b, pe = torch.ones(d, device=DEV), torch.zeros(d, device=DEV), torch.randn(hw, d, device=DEV)
    for name, split, out_lo in (("bf16", False, None), ("x3 -> plain", True, False), ("x3 -> hi+lo", True, True)):
        ta, tw = ops.cast(t_in, split), ops.cast(wp, split)
        t = timeit(lambda: ops.linear_ln(ta, tw, bias[:d].contiguous(), g, b, 1e-5, post=pe, out_lo=out_lo))
        print(f"  {name:12s} {t:8.3f} ms")


if __name__ == "__main__":
    main()
