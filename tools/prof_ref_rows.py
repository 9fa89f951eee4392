#!/usr/bin/env python3
"""The two reference-geometry cross-attention rows of SURVEY 8d (VATLiDAR: 576 q x 32400 keys, d = 896, 8 heads; VATVision: 768 q x 1536 keys,
d = 2048, 8 heads) for `rocprofv3 --kernel-trace --stats`: python3 tools/prof_ref_rows.py <lidar|vision> <mode> [iters]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidar_vision_vqa_amd import fusion, synth
which, mode = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B, nq, nkv, d, h = (1, 576, 32400, 896, 8) if which == "lidar" else (1, 768, 1536, 2048, 8)
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
blk = fusion.VATBlock(d, h, 4 * d, 0.1).to(dev).eval()
synth.load_seeded(blk, 401)
blk.precision = mode
q, kv = torch.randn(B, nq, d, device=dev), torch.randn(B, nkv, d, device=dev)
for _ in range(3):
    blk.cross_attention(q, kv)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    blk.cross_attention(q, kv)
e.record()
torch.cuda.synchronize()
print(f"{which} {mode}: {s.elapsed_time(e) / iters:.4f} ms per call")
