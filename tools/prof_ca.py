#!/usr/bin/env python3
"""Profiling target: the fused cross-attention at the headline shape, `python tools/prof_ca.py [mode] [iters]` (rocprofv3 -- python3 tools/prof_ca.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lidar_vision_vqa_amd import fusion, synth
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
blk = fusion.VATBlock(768, 12, 3072, 0.1).to(dev).eval()
synth.load_seeded(blk, 401)
blk.precision = mode
q, kv = torch.randn(1, 32768, 768, device=dev), torch.randn(1, 196, 768, device=dev)
for _ in range(iters):
    blk.cross_attention(q, kv)
torch.cuda.synchronize()
