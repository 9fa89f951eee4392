#!/usr/bin/env python3
"""Latency experiment for the projection GEMM: same launch with lda = 0 (A always an L2 hit) and/or ldc = 0
(C stores collapse onto one row) to separate HBM-miss latency on A and the store stream from the MFMA loop."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lidar_vision_vqa_amd import _ffi as F, ops

dev = torch.device("cuda:0")
m, n, k = (int(v) for v in os.environ.get("GEMM_SHAPE", "1048576,1536,768").split(","))
a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05
(ah, _), (wh, _) = ops.cast(a, False), ops.cast(w, False)
c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)

def run(lda, ldc, iters=5):
    def f():
        rc = F.lib().lvq_gemm_bf16(F.ptr(ah), None, F.ptr(wh), None, None, None, None, F.i64(0), F.cfloat(1.0), F.cint(0), F.i64(m), F.cint(n),
                                   F.cint(k), F.i64(lda), F.i64(k), F.i64(ldc), F.cint(1), F.i64(0), F.i64(0), F.i64(0), None, F.ptr(c), None, F.stream_ptr(dev))
        F.check(rc, "gemm")
    for _ in range(2): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record()
        for _ in range(iters): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best

for lda, ldc, what in [(k, n, "normal"), (0, n, "A rows all alias row 0 (L2 hits)"), (k, 0, "C rows alias row 0"), (0, 0, "both")]:
    t = run(lda, ldc)
    print(f"{what:40s}: {t:.4f} ms  {2.0 * m * n * k / t / 1e9:.1f} TFLOP/s")
