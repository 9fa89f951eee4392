#!/usr/bin/env python3
"""Tiled (row-source) stream attention against the dense call on the same shape as the bench (B scenes x 12 heads x 576 queries x 4096
tiles), q hi + lo: (a) every key from the table rows, (b) every key a computed row, scene-contiguous, (c) 27.6 % computed rows in
(tile, scene) order, (d) the dense (untiled) kernel, (e) the signed pair stream over the 27.6 %."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_vision_vqa_amd import ops
DEV = torch.device("cuda:0"); torch.set_grad_enabled(False)
B = int(os.environ.get("SCENES", "8")); H, nq, nt, dh = 12, 576, 4096, 64
d = H * dh; hw = nt * 64
def timeit(fn, iters=3):
    fn(); fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
q1 = ops.cast(torch.randn(nq, d, device=DEV), True)
q = (q1[0].repeat(B, 1), q1[1].repeat(B, 1))
kv = torch.randn(hw + B * hw, 2 * d, device=DEV).to(torch.bfloat16)
e_idx = torch.arange(hw, device=DEV, dtype=torch.int32).view(1, hw)
cases = {"all table": e_idx.expand(B, hw).contiguous(),
         "all computed, scene-contiguous": (hw + torch.arange(B, device=DEV, dtype=torch.int32).view(B, 1) * hw + e_idx).contiguous()}
g = torch.Generator(device=DEV).manual_seed(1)
dirty = torch.rand(B, hw, device=DEV, generator=g) < 0.276
order = dirty.view(B, nt, 64).permute(1, 0, 2).reshape(-1)                       # (tile, scene, cell) order
num = (torch.cumsum(order.int(), 0) - 1).view(nt, B, 64).permute(1, 0, 2).reshape(B, hw)
cases["27.6 % computed, (tile, scene) order"] = torch.where(dirty, hw + num, e_idx.expand(B, hw)).to(torch.int32).contiguous()
fl = 4.0 * B * nq * hw * d
for name, s in cases.items():
    t = timeit(lambda: ops.attention_tiled(q, kv, s, batch=B, n_heads=H, nq=nq, n_tiles=nt, dh=dh, scale=1 / math.sqrt(dh)))
    print(f"{name:45s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s (algorithmic)", flush=True)
dk = kv[hw:]
t = timeit(lambda: ops.attention(q, (dk, None), (dk[:, d:], None), batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=hw, dh=dh, q_strides=(nq * d, d, dh),
                                 k_strides=(hw * 2 * d, 2 * d, dh), v_strides=(hw * 2 * d, 2 * d, dh), scale=1 / math.sqrt(dh)))
print(f"{'dense (untiled) call':45s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s")
s = cases["27.6 % computed, (tile, scene) order"]
pair_src, pair_info = ops.bev_scene_pairs(s, B, nt, hw)
tot = ops.attention_stream_totals(q1, kv[:hw], n_heads=H, nq=nq, nkv=hw, dh=dh, scale=1 / math.sqrt(dh))
t = timeit(lambda: ops.attention_tiled_signed(q1, kv, s, pair_src, pair_info, tot, batch=B, n_heads=H, nq=nq, n_tiles=nt, dh=dh,
                                              scale=1 / math.sqrt(dh), shared_q=True))
print(f"{'signed pair stream over the 27.6 %':45s} {t:8.3f} ms  {fl / t / 1e9:7.1f} TFLOP/s (algorithmic)")
