#!/usr/bin/env python3
"""Time the cross-attention sub-path `q + ca(ca_ln(q), kv, kv)` (fused kernel vs the unfused chain) at the SURVEY 8d shapes.
    python tools/bench_ca.py [--shapes headline,resampled,...] [--iters 20]"""
import argparse, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lidar_vision_vqa_amd import fusion, synth

SHAPES = {"headline": (1, 32768, 196), "resampled": (1, 576, 196), "resampled_b8": (8, 576, 196), "pipeline_b32": (32, 576, 196), "rows4096": (1, 4096, 196)}


def ev(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="headline,resampled_b8,pipeline_b32")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", default="mixed,bf16")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.set_grad_enabled(False)
    d, h = 768, 12
    blk = fusion.VATBlock(d, h, 4 * d, 0.1).to(dev).eval()
    synth.load_seeded(blk, 401)
    for name in args.shapes.split(","):
        B, nq, nkv = SHAPES[name]
        q, kv = torch.randn(B, nq, d, device=dev), torch.randn(B, nkv, d, device=dev)
        flops = B * (4.0 * nq * d * d + 4.0 * nkv * d * d + 4.0 * nq * nkv * d)
        row = {"shape": [B, nq, nkv, d, h], "gflop": round(flops / 1e9, 2)}
        for mode in args.modes.split(","):
            blk.precision = mode
            ms = ev(lambda: blk.cross_attention(q, kv), args.iters)
            os.environ["LVQ_NO_FUSED_CA"] = "1"
            ms0 = ev(lambda: blk.cross_attention(q, kv), args.iters)
            del os.environ["LVQ_NO_FUSED_CA"]
            row[mode] = {"fused_ms": round(ms, 4), "unfused_ms": round(ms0, 4), "fused_tflops": round(flops / ms / 1e9, 1),
                         "frac_of_2500": round(flops / ms / 1e9 / 2500.0, 4)}
        print(json.dumps({name: row}), flush=True)


if __name__ == "__main__":
    main()
