"""Hard voxeliser only (cfg-3 sizes, both grids): the target of `rocprofv3 --kernel-trace` for the per-kernel split."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lidar_vision_vqa_amd import lidar, synth
DEV = torch.device("cuda:0")
rng = list(synth.PC_RANGE_NUSC)
S, n = 8, 65536
scenes = [synth.scene_points("C", n, 1010 + i) for i in range(S)]
pts = torch.from_numpy(np.concatenate(scenes)).to(DEV)
off = torch.tensor(np.concatenate(([0], np.cumsum([len(s) for s in scenes]))), dtype=torch.int32, device=DEV)
for vs, T, mv in [(synth.VOXEL_01, 10, 160000), (synth.VOXEL_PILLAR, 20, 30000)]:
    gen = lidar.VoxelGeneratorWrapper(vs, rng, 4, T, mv)
    for _ in range(5):
        out = gen.generate_batch_device(pts, off, S)
    torch.cuda.synchronize()
