#!/usr/bin/env python3
"""Tiled stream attention: where does the time go?  Same shape as the bench (B scenes x 12 heads x 576 queries x 4096 tiles), q hi + lo,
piece sources (a) all from the table, (b) all live and contiguous per scene, (c) bench-like (39 % live in (tile, scene, piece) order),
(d) the dense (untiled) call."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_vision_vqa_amd import ops
DEV = torch.device("cuda:0"); torch.set_grad_enabled(False)
B = int(os.environ.get("SCENES", "8")); H, nq, nt, dh = 12, 576, 4096, 64
d = H * dh
def timeit(fn, iters=3):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
q = ops.cast(torch.randn(B * nq, d, device=DEV), True)
table = torch.randn(nt * 64, 2 * d, device=DEV).to(torch.bfloat16)
live = torch.randn(B * nt * 64, 2 * d, device=DEV).to(torch.bfloat16)
t_idx = torch.arange(nt, device=DEV).view(1, nt, 1); p_idx = torch.arange(8, device=DEV).view(1, 1, 8); b_idx = torch.arange(B, device=DEV).view(B, 1, 1)
tab_src = (~(64 * t_idx + 8 * p_idx)).expand(B, nt, 8)
cases = {"all table": tab_src.clone(),
         "all live, scene-contiguous": ((b_idx * nt + t_idx) * 64 + 8 * p_idx).expand(B, nt, 8).clone()}
g = torch.Generator(device=DEV).manual_seed(1)
is_live = torch.rand(B, nt, 8, device=DEV, generator=g) < 0.39
order = is_live.permute(1, 0, 2).reshape(-1)                       # (tile, scene, piece) order
k = torch.cumsum(order.int(), 0) - 1
src = torch.where(order, 8 * k, torch.zeros_like(k)).view(nt, B, 8).permute(1, 0, 2)
cases["39 % live, (tile, scene, piece) order"] = torch.where(is_live, src, tab_src).contiguous()
fl = 4.0 * B * nq * nt * 64 * d
for name, s in cases.items():
    s32 = s.to(torch.int32).contiguous().view(-1)
    t = timeit(lambda: ops.attention_tiled(q, live, table, s32, batch=B, n_heads=H, nq=nq, n_tiles=nt, dh=dh, scale=1 / math.sqrt(dh)))
    print(f"{name:45s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s", flush=True)
kv = live[:B * nt * 64]
t = timeit(lambda: ops.attention(q, (kv, None), (kv[:, d:], None), batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=nt * 64, dh=dh, q_strides=(nq * d, d, dh),
                                 k_strides=(nt * 64 * 2 * d, 2 * d, dh), v_strides=(nt * 64 * 2 * d, 2 * d, dh), scale=1 / math.sqrt(dh)))
print(f"{'dense (untiled) call':45s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s")
