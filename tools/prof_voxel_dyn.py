"""Dynamic voxeliser only (cfg-3: 8 x 65 536 points, 0.1 m grid, 3-D keys): the target of `rocprofv3 --kernel-trace [--pmc ...]`."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lidar_vision_vqa_amd import lidar, synth
DEV = torch.device("cuda:0")
rng = list(synth.PC_RANGE_NUSC)
S, n = 8, 65536
scenes = [synth.scene_points("C", n, 1010 + i) for i in range(S)]
pts = torch.from_numpy(np.concatenate(scenes)).to(DEV)
bp = torch.cat((torch.repeat_interleave(torch.arange(S, device=DEV, dtype=torch.float32), torch.tensor([len(s) for s in scenes], device=DEV)).unsqueeze(1), pts), 1).contiguous()
grid = lidar.grid_size_from(rng, synth.VOXEL_01)
torch.cuda.synchronize()
for _ in range(10):
    lidar._dynamic_voxelize(bp, S, rng, synth.VOXEL_01, grid, 3)
torch.cuda.synchronize()
