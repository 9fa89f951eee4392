"""bf16x3 long-stream attention (4, 12, 576, 262144, 64) timed for LVQ_ATTN_NW sweeps."""
import os, sys, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lidar_vision_vqa_amd import ops
DEV = torch.device("cuda:0")
B, H, nq, nkv, dh = 4, 12, 576, 262144, 64
d = H * dh
q = torch.randn(B * nq, d, device=DEV); kv = torch.randn(B * nkv, 2 * d, device=DEV)
qb, kvb = ops.cast(q, True), ops.cast(kv, True)
vsl = (kvb[0][:, d:], kvb[1][:, d:])
fn = lambda: ops.attention(qb, kvb, vsl, batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=nkv, dh=dh, q_strides=(nq * d, d, dh),
                           k_strides=(nkv * 2 * d, 2 * d, dh), v_strides=(nkv * 2 * d, 2 * d, dh), scale=1 / math.sqrt(dh))
for _ in range(2): fn()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    s.record()
    for _ in range(3): fn()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 3)
print(f"x3 attn NW={os.environ.get('LVQ_ATTN_NW','default')}: {best:.3f} ms")
