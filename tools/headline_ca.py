#!/usr/bin/env python3
"""The literal headline shape (B, Nq, Nkv, d, h) = (1, 32768, 196, 768, 12) through VATBlock.cross_attention, timed with HIP
events; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidar_vision_vqa_amd import pipeline as P
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = P.PipelineConfig()
pipe = P.FusionPipeline(cfg, dev, precision=os.environ.get("PREC", "bf16"))
d = cfg.d_model
q = torch.randn(1, 32768, d, device=dev)
kv = torch.randn(1, cfg.n_patches, d, device=dev)
for _ in range(5):
    pipe.fuse.cross_attention(q, kv)
torch.cuda.synchronize()
n = int(os.environ.get("ITERS", "20"))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
s.record()
for _ in range(n):
    pipe.fuse.cross_attention(q, kv)
e.record()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
flops = 4.0 * 32768 * d * d + 4.0 * cfg.n_patches * d * d + 4.0 * 32768 * cfg.n_patches * d
print(f"cross_attention 32768x196: {ms:.4f} ms/call GPU, {t_issue / n * 1e3:.4f} ms/call host issue, {flops / ms / 1e9:.1f} TFLOP/s")
