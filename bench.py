#!/usr/bin/env python3
"""bench.py -- throughput of the LiDAR-vision fusion hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path (lidar_vision_vqa_amd.pipeline.FusionPipeline) over one batch of `--scenes` synthetic
scenes per GPU (default 32): points and image-patch tokens are already resident in HBM when the timed region starts.
Workload = BASELINE.json configs[1]: 32 768-point Dist-C scenes, 0.1 m voxel grid, 196 ViT-B/16 patches,
d = 768, 12 heads, bf16 MFMA.  Scenes shard one batch per rank with no data-path collective; the only exchange is one RCCL
all-reduce(SUM) per step of a fused fp32 buffer [token-sum (d) | scene count] -> "scaling": "weak".

`value` is measured in the precision mode `--precision` (default "mixed": bf16 MFMA tiles everywhere, plain bf16 operands on the
262 144-key K/V stream, hi + lo bf16 operands elsewhere -- the fastest mode that meets the north-star 1e-3 tolerance, see
`parity_vs_cpu`, which is measured for EVERY mode timed in this run against the CPU oracle).  Rank 0 prints ONE JSON line with
`roofline` (dominant kernel: the stream attention or the K|V projection GEMM, whichever took longer; algorithmic FLOPs / HIP-event time on the launch stream inside the timed
region), `roofline_headline` (the literal "32k pts x 196 patches" cross-attention sub-path), every cross-attention row of
SURVEY 8d, the HBM-bound voxelisers at cfg-3, the same workload on Dist-U scenes, and `cpu_baseline` (the CPU restatement in
oracle/ timed on the host cores on a bounded sample).
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
TOL = 1e-3                  # north_star: fused tokens within 1e-3 of the fp32 CPU path
MODES = ("mixed", "mixed16", "bf16", "bf16x3")


def avg_ms(pairs):
    return float(np.mean([s.elapsed_time(e) for s, e in pairs])) if pairs else None


def event_ms(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def profile_traffic(prefixes):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic*.json; counters cannot be read from
    inside this process).  Returns (bytes | None, source file | None): a constant of the committed profile, not a live reading."""
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
        if "pmc_traffic" not in name or not name.endswith(".json"):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                for k, v in json.load(f).items():
                    if any(k.startswith(p) for p in prefixes):
                        return v["traffic_bytes_per_launch"], "profiles/" + name
        except Exception:
            continue
    return None, None


def cross_attention_rows(P, fusion, dev, rows_filter=None):
    """Every cross-attention row of SURVEY 8d: the whole `q + ca(ca_ln(q), kv, kv)` sub-path, all modes."""
    rows = [("32k pts x 196 patches (headline)", 1, 32768, 196, 768, 12), ("resampled tokens x patches", 1, 576, 196, 768, 12),
            ("resampled tokens x patches, batch 8", 8, 576, 196, 768, 12), ("cfg-5", 8, 256, 576, 768, 12),
            ("reference-true VATLiDAR", 1, 576, 32400, 896, 8), ("reference-true VATVision", 1, 768, 1536, 2048, 8)]
    out = {}
    for label, B, nq, nkv, d, h in rows:
        if rows_filter and label not in rows_filter:
            continue
        blk = fusion.VATBlock(d, h, 4 * d, 0.1).to(dev).eval()
        q, kv = torch.randn(B, nq, d, device=dev), torch.randn(B, nkv, d, device=dev)
        flops = B * (4.0 * nq * d * d + 4.0 * nkv * d * d + 4.0 * nq * nkv * d)
        entry = {"shape": [B, nq, nkv, d, h], "flops": flops}
        for mode in MODES:
            blk.precision = mode
            ms = event_ms(lambda: blk.cross_attention(q, kv), iters=10)
            entry[mode] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1), "frac_of_bf16_peak": round(flops / ms / 1e9 / PEAK_BF16_TFLOPS, 4)}
        out[label] = entry
        del blk, q, kv
    return out


def voxel_blocks(dev):
    """HBM-bound side at BASELINE cfg-3 (8 scenes x 65 536 points, 0.1 m grid): hard and dynamic voxelisers, whole call.
    Algorithmic bytes: hard 16 N + M (4 T C + 16); dynamic 20 N + M * 16 + 4 N (unq_inv materialised)."""
    from lidar_vision_vqa_amd import lidar as LD, synth as SY
    scenes = [SY.scene_points("C", 65536, 1010 + i) for i in range(8)]
    lens = [len(x) for x in scenes]
    pts = torch.from_numpy(np.concatenate(scenes)).to(dev)
    off = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device=dev)
    rng = list(SY.PC_RANGE_NUSC)
    gen = LD.VoxelGeneratorWrapper(SY.VOXEL_01, rng, 4, 10, 160000)
    m_vox = int(gen.generate_batch_device(pts, off, 8)[3][-1])
    us = event_ms(lambda: gen.generate_batch_device(pts, off, 8), iters=20) * 1e3
    nbytes = 16.0 * pts.shape[0] + m_vox * (4.0 * 10 * 4 + 16)
    tr, src = profile_traffic(["lvq_voxelize_hard (voxel_hashed.hip) 8 x 65536 points"])
    hard = {"bound": "hbm", "kernels": "lvq_voxelize_hard, whole call", "us": round(us, 1), "points": int(pts.shape[0]), "voxels": m_vox,
            "achieved": round(nbytes / us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(nbytes / us / 1e3 / PEAK_HBM_GBS, 4),
            "frac_of_6300_achievable": round(nbytes / us / 1e3 / 6300.0, 4), "algorithmic_bytes": nbytes, "traffic": tr, "traffic_source": src}
    us_m = event_ms(lambda: gen.generate_mean_device(pts, off, 8), iters=20) * 1e3
    nb_m = 16.0 * pts.shape[0] + m_vox * (4.0 * 4 + 16)
    mean = {"bound": "hbm", "kernels": "lvq_voxelize_mean (voxelise -> MeanVFE fused), whole call", "us": round(us_m, 1),
            "achieved": round(nb_m / us_m / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(nb_m / us_m / 1e3 / PEAK_HBM_GBS, 4),
            "algorithmic_bytes": nb_m}
    bidx = torch.repeat_interleave(torch.arange(8, device=dev, dtype=torch.float32), torch.tensor(lens, device=dev))
    bp = torch.cat((bidx[:, None], pts), 1).contiguous()
    grid = LD.grid_size_from(rng, SY.VOXEL_01)
    m_dyn = int(LD._dynamic_voxelize(bp, 8, rng, SY.VOXEL_01, grid, 3)["counts"][0])
    us_d = event_ms(lambda: LD._dynamic_voxelize(bp, 8, rng, SY.VOXEL_01, grid, 3), iters=20) * 1e3
    nb_d = 20.0 * bp.shape[0] + 4.0 * bp.shape[0] + m_dyn * (4.0 + 4.0 + 16.0)       # points in, unq_inv, (key, count, coords) per voxel
    tr, src = profile_traffic(["lvq_voxelize_dynamic 8 x 65536 points"])
    dyn = {"bound": "hbm", "kernels": "lvq_voxelize_dynamic (unique keys ascending + inverse + counts + coords), whole call",
           "us": round(us_d, 1), "points": int(bp.shape[0]), "voxels": m_dyn, "achieved": round(nb_d / us_d / 1e3, 1), "peak": PEAK_HBM_GBS,
           "unit": "GB/s", "frac": round(nb_d / us_d / 1e3 / PEAK_HBM_GBS, 4), "algorithmic_bytes": nb_d, "traffic": tr, "traffic_source": src}
    return hard, mean, dyn, scenes


def _vox_worker(args):
    """One scene through the CPU voxeliser restatement (process pool worker: the scene-parallel CPU variant of SURVEY 8d)."""
    pts, vs, rng, T, mv = args
    from oracle import lidar_oracle as LO
    g = LO.VoxelGenerator(vs, rng, 4, T, mv)
    t0 = time.perf_counter()
    g.generate(pts)
    return time.perf_counter() - t0


def cpu_voxel_baseline(scenes):
    """oracle/voxel_oracle.c on cfg-3's 8 scenes: single-threaded (what spconv's CPU generator is) and scene-parallel (one
    process per scene), median of 5 after a warm-up."""
    from lidar_vision_vqa_amd import synth as SY
    from oracle import lidar_oracle as LO
    rng = list(SY.PC_RANGE_NUSC)
    g = LO.VoxelGenerator(SY.VOXEL_01, rng, 4, 10, 160000)
    g.generate(scenes[0])
    single = []
    for _ in range(5):
        t0 = time.perf_counter()
        for s in scenes:
            g.generate(s)
        single.append(time.perf_counter() - t0)
    par = None
    try:
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        with ctx.Pool(len(scenes)) as pool:
            jobs = [(s, SY.VOXEL_01, rng, 10, 160000) for s in scenes]
            pool.map(_vox_worker, jobs)                         # warm-up: imports, the 168 MB lookup table per worker
            runs = []
            for _ in range(5):
                t0 = time.perf_counter()
                pool.map(_vox_worker, jobs)
                runs.append(time.perf_counter() - t0)
        par = float(np.median(runs))
    except Exception as err:                                    # a box without fork/spawn headroom: report the single-thread figure only
        par = None
        print(f"[bench] scene-parallel CPU voxeliser skipped: {err}", file=sys.stderr)
    n = sum(len(s) for s in scenes)
    return {"points": n, "single_thread_ms": round(float(np.median(single)) * 1e3, 2), "single_thread_mpts_per_s": round(n / np.median(single) / 1e6, 2),
            "scene_parallel_processes": len(scenes), "scene_parallel_ms": None if par is None else round(par * 1e3, 2),
            "sample": "cfg-3, 8 x 65 536 points, 0.1 m grid, T=10, 160 000 voxels: median of 5 after warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=32, help="scenes per GPU per step")
    ap.add_argument("--precision", default="mixed", choices=list(MODES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (other modes, cross-attention rows, voxelisers, Dist-U)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one rank per GPU.  With fewer devices than ranks (a rehearsal of the N > 1 path on a one-GPU box) the ranks share devices and the
    # process group falls back to gloo -- RCCL refuses two ranks on one device; such a line says so in config.parallelism
    ndev = torch.cuda.device_count()
    rehearsal = world > ndev
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)      # inference path (the reference's eval / no_grad mode)

    from lidar_vision_vqa_amd import dist as D
    from lidar_vision_vqa_amd import fusion, ops, pipeline as P

    D.init_dist_if_needed("gloo" if rehearsal else None)
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

    # ---- roofline of the two big kernels of VATLiDAR's cross-attention; `roofline` is whichever took longer per launch ----
    h, w = cfg.bev_hw
    d = cfg.d_model
    per_step = cfg.n_layers + 1        # launches per step: n_layers (VATLiDAR, the S*HW-key stream) + 1 (fusion block, 196 keys); keep the big ones
    big = lambda tag: [p for i, p in enumerate(events.get(tag, [])) if (i % per_step) < cfg.n_layers]
    kv_ms, at_ms = avg_ms(big("ca_kv_proj")), avg_ms(big("ca_attn"))
    # rows the GEMM actually projects: the tiled key stream computes K|V for the DIRTY cells only (a pillar in the 3 x 3 neighbourhood; clean
    # cells come from the per-model table) -> EXECUTED rows, read back once after the timed region; the dense routes project every BEV cell
    tc = getattr(pipe.vat_lidar, "_last_tile_counts", None)
    live_rows = int(tc[2]) if tc is not None else S * h * w
    kv_rows = (live_rows + 255) // 256 * 256             # whole 256-row tiles run
    kv_flops = 2.0 * kv_rows * (2 * d) * d               # 4 d^2 per projected key
    at_flops = 4.0 * S * cfg.n_queries * (h * w) * d     # QK^T + PV over EVERY key (4 nq nkv d per scene)
    prec = args.precision
    roofline = roofline_kv = roofline_attn = None
    if kv_ms:
        ach = kv_flops / (kv_ms * 1e-3) / 1e12
        form = {"bf16": "plain operands: 1 MFMA pass", "mixed": "A plain, W hi+lo: 2 MFMA passes (executed FLOPs = 2x algorithmic)",
                "mixed16": "A plain, W hi+lo: 2 MFMA passes (executed FLOPs = 2x algorithmic)",
                "bf16x3": "A and W hi+lo: 3 MFMA passes (executed FLOPs = 3x algorithmic)"}[prec]
        ex = {"bf16": 1, "mixed": 2, "mixed16": 2, "bf16x3": 3}[prec]
        tr, src = profile_traffic([f"k_gemm_256 M={kv_rows} N={2 * d} K={d} {prec}"])
        roofline_kv = {"bound": "mfma", "kernel": "k_gemm_256 (256x256 tile, LDS-DMA, A ring 3 / W ring 2; VATLiDAR.ca K|V projection over the dirty BEV cells, "
                       f"M = {kv_rows} of {S * h * w} BEV cells, N=2d, K=d); " + form,
                       # achieved / frac / flops_per_launch = EXECUTED MFMA FLOPs (rows actually projected x passes actually issued)
                       "achieved": round(ach * ex, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach * ex / PEAK_BF16_TFLOPS, 4),
                       "traffic": tr, "traffic_source": src, "avg_launch_ms": round(kv_ms, 4), "flops_per_launch": kv_flops * ex,
                       "rows_projected": kv_rows, "rows_dense": S * h * w, "live_fraction": round(live_rows / float(S * h * w), 4),
                       "mfma_passes": ex, "achieved_one_pass": round(ach, 2),
                       "dense_formula_flops": 2.0 * S * h * w * (2 * d) * d,
                       "dense_formula_rate": round(2.0 * S * h * w * (2 * d) * d / (kv_ms * 1e-3) / 1e12, 2)}
    if at_ms:
        ach = at_flops / (at_ms * 1e-3) / 1e12
        form = {"bf16": "Q, K, P, V plain: 1 MFMA pass", "mixed": "Q hi+lo (2 MFMA passes over QK^T), K, P, V plain: executed FLOPs = 1.5x algorithmic",
                "mixed16": "Q one fp16 operand against fp16 K (1 MFMA pass over QK^T), P, V plain bf16",
                "bf16x3": "all operands hi+lo: executed FLOPs = 3x algorithmic"}[prec]
        ex = {"bf16": 1.0, "mixed": 1.5, "mixed16": 1.0, "bf16x3": 3.0}[prec]
        # keys the launch actually streams: block 0's queries are scene-independent, so a scene streams only its dirty cells, twice
        # (computed rows added, the table rows of the same cells subtracted from the per-model totals) -- read back after the timed region
        pi = getattr(pipe.vat_lidar.blocks[0], "_last_pair_info", None)
        keys_streamed = S * h * w
        if pi is not None:
            pin = pi.cpu().numpy()
            keys_streamed = int(sum(int(t) * 64 if int(u) else h * w for t, u in pin))
            form += f"; signed pair stream: {keys_streamed} of {S * h * w} keys streamed ({sum(int(u) for _, u in pin)} of {S} scenes signed)"
        ex *= keys_streamed / float(S * h * w)
        piped = (prec in ("mixed", "mixed16") or os.environ.get("LVQ_ATTN_PIPE")) and not os.environ.get("LVQ_ATTN_NO_PIPE")
        ring = ("software-pipelined stream (next block's score MFMAs between this block's exponentials), LDS-DMA ring of 4 K|V tiles, 2 waves per SIMD"
                if piped else "LDS-DMA ring of 3 K|V tiles, 3 waves per SIMD")
        tr, src = profile_traffic([f"k_attn32 S={S} nq={cfg.n_queries} nkv={h * w} {prec}"])
        roofline_attn = {"bound": "mfma", "kernel": f"k_attn32 (VATLiDAR.ca: {S} scenes x {cfg.n_heads} heads x {cfg.n_queries} queries over the {h * w}-key "
                         "BEV stream, head_dim 64; 32x32x16 MFMA, " + ring + ", fixed softmax reference); " + form,
                         # achieved / frac / flops_per_launch = EXECUTED MFMA FLOPs (keys actually streamed x passes actually issued);
                         # the dense formula of SURVEY 8d (every key of every scene) is reported beside it, not as the roofline number
                         "achieved": round(ach * ex, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach * ex / PEAK_BF16_TFLOPS, 4),
                         "traffic": tr, "traffic_source": src, "avg_launch_ms": round(at_ms, 4), "flops_per_launch": at_flops * ex,
                         "keys_streamed": keys_streamed, "keys_dense": S * h * w,
                         "dense_formula_flops": at_flops, "dense_formula_rate": round(ach, 2), "dense_formula_frac": round(ach / PEAK_BF16_TFLOPS, 4)}
    # the fused K|V kernel (default route): refine conv + folded LayerNorm / K|V projection for the dirty cells, HBM-side kernel
    roofline_bev_kv = None
    bk_ms = avg_ms(events.get("bev_kv", []))
    if bk_ms:
        # K|V rows out (bf16) + the table T once (fp32) + index map + the (t hi, t lo, rstd, key) rows written and read between the two launches
        nbytes = live_rows * (2 * 2 * d) + (h * w) * (2 * d) * 4 + S * h * w * 4 + 2 * live_rows * (256 + 8)
        ex_flops = 2.0 * live_rows * (2 * d + 64) * 64 * {"bf16": 1, "mixed": 3, "mixed16": 3, "bf16x3": 3}[prec]
        tr, src = profile_traffic([f"k_tile_kv S={S} {prec}"])
        roofline_bev_kv = {"bound": "hbm", "kernel": "lvq_bev_tile_kv = k_conv_rows + k_kv_rows (pillar halo gather + depthwise 3x3 + GELU -> 64-channel token t -> K|V = rstd (M t + m0) + T[key] "
                           f"for the {live_rows} dirty cells of {S * h * w}; LayerNorm and the 768-deep K|V projection folded into a 64-deep one)",
                           "achieved": round(nbytes / (bk_ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": round(nbytes / (bk_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "algorithmic_bytes": nbytes, "traffic": tr, "traffic_source": src,
                           "avg_launch_ms": round(bk_ms, 4), "rows": live_rows, "rows_dense": S * h * w,
                           "mfma_flops_executed": ex_flops, "mfma_rate": round(ex_flops / (bk_ms * 1e-3) / 1e12, 1),
                           "replaces": "token kernel + K|V GEMM over the dirty rows (4 d^2 FLOPs per key, SURVEY 8d): 4.1 + 10.0 ms at the same row count"}
    if roofline_kv or roofline_attn:
        roofline = roofline_attn if (at_ms or 0) >= (kv_ms or 0) else roofline_kv

    result = {
        "metric": "fused tokens/sec/GPU + cross-attn MFMA-roofline % (32k pts x 196 patches)",
        "value": round(value, 1), "unit": "fused tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: " + cfg.describe(), "scenes_per_gpu_per_step": S,
                   "fused_tokens_per_scene": cfg.n_queries, "parallelism": f"scene-parallel x{world}" + (f" (REHEARSAL: {world} ranks on {ndev} device(s), gloo)" if rehearsal else ""),
                   "precision_mode": args.precision + " (bf16 MFMA tiles throughout; see parity_vs_cpu for the error of every mode)"},
        "roofline": roofline, "roofline_kv_proj": roofline_kv, "roofline_bev_kv": roofline_bev_kv, "roofline_attention": roofline_attn,
    }

    mode_values = {args.precision: round(value, 1)}
    if not args.no_extras and world == 1:                 # secondary measurements: single-GPU runs only (N > 1 prints the scaling line and leaves)
        # ---- per-stage event times of the cross-attention sub-path in the timed region ----
        result["stage_ms"] = {k: {"mean": round(avg_ms(v), 4), "max": round(max(a.elapsed_time(b) for a, b in v), 4),
                                  "min": round(min(a.elapsed_time(b) for a, b in v), 4), "launches_per_step": len(v) // args.steps}
                              for k, v in events.items()}
        # ---- the same workload in the other precision modes ----
        for mode in MODES:
            if mode == args.precision:
                continue
            pipe.set_precision(mode)
            n_it = max(2, args.steps // 2)
            ms = event_ms(lambda: pipe(pts, off, patches), iters=n_it, warm=2)
            mode_values[mode] = round(tokens_per_step / ms * 1e3, 1)
        pipe.set_precision(args.precision)
        result["value_by_mode"] = mode_values
        # ---- Dist-U scenes (SURVEY 8d cfg-2 asks for both distributions): ~1 point per voxel, worst case for the voxelisers ----
        cfg_u = P.PipelineConfig(dist="U")
        pu, ou, pau, _, _ = P.synthetic_batch(cfg_u, S, 1002, dev)
        ms_u = event_ms(lambda: pipe(pu, ou, pau), iters=max(2, args.steps // 2), warm=2)
        result["value_dist_u"] = {"value": round(tokens_per_step / ms_u * 1e3, 1), "ms_per_step": round(ms_u, 3),
                                  "note": "same pipeline and mode on Dist-U scenes (seed 1002+i); `value` is Dist-C"}
        del pu, ou, pau
        # ---- every cross-attention row of SURVEY 8d, all modes; the first row is the literal headline shape ----
        ca = cross_attention_rows(P, fusion, dev)
        result["cross_attention_rows"] = ca
        hl = ca["32k pts x 196 patches (headline)"]
        tr, src = profile_traffic(["cross_attn_32768x196 " + args.precision])
        result["roofline_headline"] = {"bound": "mfma", "kernel": "ca sub-path at (B,Nq,Nkv,d,h) = (1,32768,196,768,12): ca_ln, Q / K|V / out projections, attention, residual",
                                       "mode": args.precision, "achieved": hl[args.precision]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                       "frac": hl[args.precision]["frac_of_bf16_peak"], "flops": hl["flops"], "ms": hl[args.precision]["ms"],
                                       "traffic": tr, "traffic_source": src}
        # context: the sub-path is five launches over a 32768 x 768 activation -- as built it moves 24 B (bf16) / 36 B (hi + lo modes) per
        # activation element (fp32 in, bf16 LN out, Q, O, fp32 residual in, fp32 out), i.e. its HBM floor is ABOVE its MFMA floor
        b_el = 24.0 if args.precision == "bf16" else 36.0
        hb = b_el * 32768 * 768 + 2.0 * 196 * 768 * 4
        result["roofline_headline"].update({"hbm_bytes_as_built": hb, "hbm_floor_ms": round(hb / (PEAK_HBM_GBS * 1e9) * 1e3, 4),
                                            "mfma_floor_ms": round(hl["flops"] / (PEAK_BF16_TFLOPS * 1e12) * 1e3, 4),
                                            "hbm_frac": round(hb / (hl[args.precision]["ms"] * 1e-3) / (PEAK_HBM_GBS * 1e9), 4)})
        result["cross_attn_32768x196"] = {"flops": hl["flops"], **{m: hl[m] for m in MODES}}
        # ---- HBM-bound side: hard / fused-mean / dynamic voxelisers at cfg-3 ----
        hard, mean, dyn, vscenes = voxel_blocks(dev)
        result["voxelise_cfg3"] = hard
        result["voxelise_mean_cfg3"] = mean
        result["voxelise_dynamic_cfg3"] = dyn
    else:
        vscenes = None

    # ---- CPU baseline: the oracle restatement on the host cores, bounded sample; parity of every timed mode against it ----
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
                                  "kind": "port", "sample": f"1 scene of the same workload end to end, one run ({cpu_s:.1f} s: a second run would "
                                  f"double the bounded sample; torch fp32 {torch.get_num_threads()} threads; voxeliser single-threaded C like spconv's CPU generator)"}
        if vscenes is not None:
            result["cpu_baseline"]["voxelise_cfg3_cpu"] = cpu_voxel_baseline(vscenes)
        parity = {}
        for mode in (MODES if not args.no_extras else (args.precision,)):
            pipe.set_precision(mode)
            out = pipe(pts, off, patches)
            err = (out["fused"][0].cpu() - ref["fused"][0]).abs().max().item()
            parity[mode] = {"fused_max_abs_err": round(err, 6), "meets_1e-3": bool(err <= TOL), "value": mode_values.get(mode)}
        pipe.set_precision(args.precision)
        result["parity_vs_cpu"] = {"dtype": args.precision, "fused_max_abs_err": parity[args.precision]["fused_max_abs_err"],
                                   "fused_ref_absmax": round(ref["fused"].abs().max().item(), 4), "tolerance": TOL,
                                   "value_meets_tolerance": parity[args.precision]["meets_1e-3"], "modes": parity}
    gc.enable()
    print(json.dumps(result), flush=True)
    D.finalize()


if __name__ == "__main__":
    main()
