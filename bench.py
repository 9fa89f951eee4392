#!/usr/bin/env python3
"""bench.py -- throughput of the LiDAR-vision fusion hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path (lidar_vision_vqa_amd.pipeline.FusionPipeline) over one batch of
`--scenes` synthetic scenes per GPU (default 32): points and image-patch tokens are already resident in HBM when the
timed region starts.  Workload = BASELINE.json configs[1]: 32 768-point scenes, 0.1 m
voxel grid, 196 ViT-B/16 patches, d=768, 12 heads, bf16 MFMA.  Scenes shard one batch per rank with no
data-path collective; the only exchange is one RCCL all-reduce(SUM) per step of a fused fp32 buffer
[token-sum (d) | scene count] -> "scaling": "weak".

Rank 0 prints ONE JSON line: metric/value (fused tokens/s, whole job), `roofline` for the dominant
kernel (the bf16 MFMA GEMM that projects the BEV tokens to K|V for the cross-attention: algorithmic
FLOPs / HIP-event time measured live on the launch stream), `cpu_baseline` (the CPU restatement in
oracle/ timed on the host cores on a bounded sample) and the headline cross-attention shape
(1, 32768, 196, 768, 12) timed in the same run.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (never the 2:1-sparse figure)
PEAK_HBM_GBS = 8000.0


def avg_ms(pairs):
    return float(np.mean([s.elapsed_time(e) for s, e in pairs])) if pairs else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=32,
                    help="scenes per GPU per step (same box: 16 -> 396 k, 24 -> 392 k, 32 -> 408 k, 48 -> 402 k, 64 -> 409 k fused tokens/s; 4 -> 359 k, 8 -> 373 k)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (bf16x3, headline cross-attn)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)      # inference path (the reference's eval / no_grad mode)

    from lidar_vision_vqa_amd import dist as D
    from lidar_vision_vqa_amd import ops, pipeline as P

    D.init_dist_if_needed()
    cfg = P.PipelineConfig()
    pipe = P.FusionPipeline(cfg, dev, precision=args.precision)
    S = args.scenes
    # scene i of rank r: seed 1100 + 1000*r + i (SURVEY 8d cfg-4 convention)
    pts, off, patches, pts_np, patches_np = P.synthetic_batch(cfg, S, 1100 + 1000 * rank, dev)
    red = torch.zeros(cfg.d_model + 1, dtype=torch.float32, device=dev)

    def step():
        out = pipe(pts, off, patches)
        D.reduce_step(out["fused"], red)       # fused [token-sum | n_scenes] buffer, one all-reduce when world > 1
        return out

    ops.EVENTS = {}                 # event recording is on during warm-up too (first-use costs stay out of the timed region)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ops.EVENTS = {}
    # a full CPython cycle collection (~30 ms with torch's module graph alive) otherwise lands inside the timed steps every
    # dozen iterations and stalls the launch stream: collect now, keep the survivors out of later scans
    gc.collect()
    gc.freeze()
    gc.disable()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    events, ops.EVENTS = ops.EVENTS, None
    dt = D.max_over_ranks(dt, dev)
    tokens_per_step = S * cfg.n_queries
    value = world * tokens_per_step * args.steps / dt

    if rank != 0:
        gc.enable()
        D.finalize()
        return

    # ---- roofline of the dominant kernel: BEV-token K|V projection GEMM inside VATLiDAR's cross-attention ----
    h, w = cfg.bev_hw
    d = cfg.d_model
    kv_pairs = [p for p in events.get("ca_kv_proj", [])]
    # launches per step: n_layers (VATLiDAR, M = S*HW) + 1 (fusion block, M = S*196); keep the big ones
    per_step = cfg.n_layers + 1
    big = [p for i, p in enumerate(kv_pairs) if (i % per_step) < cfg.n_layers]
    kv_ms = avg_ms(big)
    kv_flops = 2.0 * (S * h * w) * (2 * d) * d           # = 4*Nkv*d^2 per scene, x S scenes per launch
    # HBM traffic of that launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of the
    # same kernel at the same shape, profiles/r01_pmc_traffic.json); counters cannot be read from inside this process
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            for k, v in json.load(f).items():
                if k.startswith(f"k_gemm_256 M={S * h * w} N={2 * d} K={d} "):      # measured for 4 and 16 scenes per launch
                    traffic = v["traffic_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = None
    if kv_ms:
        ach = kv_flops / (kv_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "k_gemm_256 (256x256 tile, LDS-DMA, A ring 3 / W ring 2; VATLiDAR.ca K|V projection, M=S*HW, N=2d, K=d)",
                    "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "avg_launch_ms": round(kv_ms, 4), "flops_per_launch": kv_flops}

    result = {
        "metric": "fused tokens/sec/GPU + cross-attn MFMA-roofline % (32k pts x 196 patches)",
        "value": round(value, 1), "unit": "fused tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: " + cfg.describe(), "scenes_per_gpu_per_step": S,
                   "fused_tokens_per_scene": cfg.n_queries, "parallelism": f"scene-parallel x{world}"},
        "roofline": roofline,
    }

    if not args.no_extras:
        # ---- per-stage event times of the cross-attention sub-path in the timed region ----
        stages = {k: {"mean": round(avg_ms(v), 4), "max": round(max(a.elapsed_time(b) for a, b in v), 4),
                      "min": round(min(a.elapsed_time(b) for a, b in v), 4), "launches_per_step": len(v) // args.steps}
                  for k, v in events.items()}
        result["stage_ms"] = stages
        # ---- the literal headline shape: (B,Nq,Nkv,d,h) = (1,32768,196,768,12), ~97.5 GFLOP ----
        q = torch.randn(1, 32768, d, device=dev)
        kv = torch.randn(1, cfg.n_patches, d, device=dev)
        flops = 4.0 * 32768 * d * d + 4.0 * cfg.n_patches * d * d + 4.0 * 32768 * cfg.n_patches * d
        hl = {}
        for prec in ("bf16", "bf16x3"):
            pipe.fuse.precision = prec
            for _ in range(3):
                pipe.fuse.cross_attention(q, kv)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                pipe.fuse.cross_attention(q, kv)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 10
            hl[prec] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 2), "frac_of_bf16_peak": round(flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
        result["cross_attn_32768x196"] = {"flops": flops, **hl}
        pipe.set_precision(args.precision)
        # ---- HBM-bound side of the path: hard voxelisation at BASELINE cfg-3 (8 scenes x 65 536 points, 0.1 m grid) ----
        # algorithmic bytes: 16 N (points in) + M (4 T C + 12 + 4) (padded voxels, coords, counts out)
        from lidar_vision_vqa_amd import lidar as LD, synth as SY
        vscenes = [SY.scene_points("C", 65536, 1010 + i) for i in range(8)]
        vpts = torch.from_numpy(np.concatenate(vscenes)).to(dev)
        voff = torch.tensor(np.concatenate(([0], np.cumsum([len(x) for x in vscenes]))), dtype=torch.int32, device=dev)
        gen = LD.VoxelGeneratorWrapper(SY.VOXEL_01, list(SY.PC_RANGE_NUSC), 4, 10, 160000)
        vout = gen.generate_batch_device(vpts, voff, 8)
        m_vox = int(vout[3][-1])
        for _ in range(3):
            gen.generate_batch_device(vpts, voff, 8)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            gen.generate_batch_device(vpts, voff, 8)
        e.record()
        torch.cuda.synchronize()
        vus = s.elapsed_time(e) / 20 * 1e3
        vbytes = 16.0 * vpts.shape[0] + m_vox * (4.0 * 10 * 4 + 16)
        vtraffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                for k, v in json.load(f).items():
                    if k.startswith("lvq_voxelize_hard (voxel_hashed.hip) 8 x 65536 points"):
                        vtraffic = v["traffic_bytes_per_launch"]        # separate rocprofv3 --pmc passes of the same call (see the file)
        except Exception:
            vtraffic = None
        result["voxelise_cfg3"] = {"bound": "hbm", "traffic": vtraffic, "kernels": "k_bin + k_slab + k_words + k_place (voxel_hashed.hip), whole call", "us": round(vus, 1),
                                   "points": int(vpts.shape[0]), "voxels": m_vox, "achieved": round(vbytes / vus / 1e3, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(vbytes / vus / 1e3 / 8000.0, 4), "algorithmic_bytes": vbytes}
        del vpts, vout
        # ---- the parity-exact mode (bf16x3) on the same workload ----
        if args.precision != "bf16x3":
            pipe.set_precision("bf16x3")
            for _ in range(2):
                pipe(pts, off, patches)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n3 = max(2, args.steps // 2)
            for _ in range(n3):
                pipe(pts, off, patches)
            torch.cuda.synchronize()
            result["value_bf16x3"] = round(tokens_per_step * n3 / (time.perf_counter() - t1), 1)
            pipe.set_precision(args.precision)

    # ---- CPU baseline: the oracle restatement on the host cores, bounded sample (1 scene) ----
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pipeline_oracle as PO
        cores = os.cpu_count() or 1
        torch.set_num_threads(cores)
        sd = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        sds = (sd(pipe.pillar_vfe), sd(pipe.vat_lidar), sd(pipe.fuse))
        t1 = time.perf_counter()
        ref = PO.run(cfg, pts_np[:1], patches_np[:1], *sds)
        cpu_s = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": round(cfg.n_queries / cpu_s, 2), "unit": "fused tokens/s", "cores": torch.get_num_threads(),
                                  "kind": "port", "sample": f"1 scene of the same workload end to end ({cpu_s:.1f} s, torch fp32 "
                                  f"{torch.get_num_threads()} threads; voxeliser single-threaded C like spconv's CPU generator)"}
        out = pipe(pts, off, patches)
        err = (out["fused"][0].cpu() - ref["fused"][0]).abs().max().item()
        result["parity_vs_cpu"] = {"dtype": args.precision, "fused_max_abs_err": round(err, 6),
                                   "fused_ref_absmax": round(ref["fused"].abs().max().item(), 4)}
    gc.enable()
    print(json.dumps(result), flush=True)
    D.finalize()


if __name__ == "__main__":
    main()
