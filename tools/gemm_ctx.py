#!/usr/bin/env python3
"""K|V-projection GEMM (16 scenes) timed in different contexts: with / without bias, right after a 6.4 GB fill (dirty lines in
L2 / Infinity Cache, as after the token kernel), after torch.cuda.empty_cache() (fresh pages).  Used for same-box A/B runs."""
import os, sys, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lidar_vision_vqa_amd import ops
DEV = torch.device("cuda:0")
torch.set_grad_enabled(False)
S = 16
m, n, k = S * 262144, 1536, 768
a = torch.randn(m, k, device=DEV); w = torch.randn(n, k, device=DEV) * 0.05; bias = torch.randn(n, device=DEV)
ab, wb = ops.cast(a, False), ops.cast(w, False)
del a
def ev(): return torch.cuda.Event(enable_timing=True)
def timed(fn, pre=None, iters=4):
    ts = []
    for _ in range(iters + 1):
        if pre: pre()
        s, e = ev(), ev()
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts[1:]), sum(ts[1:]) / iters
print("gemm no bias        :", timed(lambda: ops.linear(ab, wb, None, out_bf=True)))
print("gemm + bias         :", timed(lambda: ops.linear(ab, wb, bias, out_bf=True)))
# preceded by a big memory-bound kernel that writes A-sized data (like the token kernel)
x = torch.empty(m, k, dtype=torch.bfloat16, device=DEV)
print("after 6.4 GB fill   :", timed(lambda: ops.linear(ab, wb, bias, out_bf=True), pre=lambda: x.fill_(1.0)))
# A produced by a kernel right before (fresh dirty lines)
def pre2():
    ab[0].copy_(x.view(torch.int16)) if False else None
# fresh output allocation each time vs reused
outs = []
def fresh():
    torch.cuda.empty_cache()
print("after empty_cache   :", timed(lambda: ops.linear(ab, wb, bias, out_bf=True), pre=fresh))
