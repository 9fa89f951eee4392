#!/usr/bin/env python3
"""Voxeliser timings at BASELINE cfg-3 (8 scenes x 65 536 points): hard / fused-mean / dynamic, both grids, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lidar_vision_vqa_amd import lidar, synth
DEV = torch.device("cuda:0")
rng = list(synth.PC_RANGE_NUSC)
S, n = 8, 65536
for dist in ("C", "U"):
    scenes = [synth.scene_points(dist, n, 1010 + i) for i in range(S)]
    pts = torch.from_numpy(np.concatenate(scenes)).to(DEV)
    off = torch.tensor(np.concatenate(([0], np.cumsum([len(s) for s in scenes]))), dtype=torch.int32, device=DEV)
    bp = torch.cat((torch.repeat_interleave(torch.arange(S, device=DEV, dtype=torch.float32), torch.tensor([len(s) for s in scenes], device=DEV)).unsqueeze(1), pts), 1).contiguous()
    def t(fn, it=50):
        for _ in range(5): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / it * 1e3
    for vs, T, mv, nd in [(synth.VOXEL_01, 10, 160000, 3), (synth.VOXEL_PILLAR, 20, 30000, 2)]:
        gen = lidar.VoxelGeneratorWrapper(vs, rng, 4, T, mv)
        grid = lidar.grid_size_from(rng, vs)
        print(f"Dist-{dist} vs={vs}: hard {t(lambda: gen.generate_batch_device(pts, off, S)):.1f} us, mean {t(lambda: gen.generate_mean_device(pts, off, S)):.1f} us, "
              f"dynamic {t(lambda: lidar._dynamic_voxelize(bp, S, rng, vs, grid, nd)):.1f} us", flush=True)
