"""oracle/pipeline_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the end-to-end hot path (the same composition as
lidar_vision_vqa_amd.pipeline.FusionPipeline) built from oracle/lidar_oracle.py
and oracle/vat_oracle.py.  It is the checker of tests/test_gpu_pipeline.py and smoke(), and the thing
timed by bench.py's cpu_baseline leg ("port": the reference itself cannot travel to the GPU box).
Follows the reference order: mask_points_by_range -> voxelise -> VFE -> scatter -> VATLiDAR -> VATBlock.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import lidar_oracle as LO
from . import vat_oracle as VO


@torch.no_grad()
def run(cfg, scenes: List[np.ndarray], patches: np.ndarray, sd_pillar: Dict[str, torch.Tensor], sd_lidar: Dict[str, torch.Tensor],
        sd_fuse: Dict[str, torch.Tensor], do_3d: bool = True) -> Dict[str, object]:
    rng = list(cfg.pc_range)
    out3, outp = [], []
    g3 = LO.VoxelGenerator(cfg.voxel_3d, rng, 4, cfg.t_3d, cfg.max_voxels_3d) if do_3d else None
    gp = LO.VoxelGenerator(cfg.voxel_pillar, rng, 4, cfg.t_pillar, cfg.max_pillars)
    for pts in scenes:
        pts = pts[LO.mask_points_by_range(pts, rng)]
        if do_3d:
            v, c, n = g3.generate(pts)
            out3.append(dict(voxels=v, voxel_coords=c, voxel_num_points=n))
        v, c, n = gp.generate(pts)
        outp.append(dict(voxels=v, voxel_coords=c, voxel_num_points=n))
    res: Dict[str, object] = {}
    if do_3d:
        b3 = LO.collate_batch(out3)
        res.update(voxel_coords=b3["voxel_coords"], voxel_num_points=b3["voxel_num_points"],
                   voxel_features=LO.mean_vfe(b3["voxels"], b3["voxel_num_points"]))
    bp = LO.collate_batch(outp)
    pf = LO.pillar_vfe(bp["voxels"], bp["voxel_num_points"], bp["voxel_coords"], sd_pillar, cfg.voxel_pillar, rng, cfg.pillar_filters)
    h, w = cfg.bev_hw
    bev = LO.pointpillar_scatter(pf, bp["voxel_coords"], w, h)
    if bev.shape[0] < len(scenes):   # trailing scenes without pillars
        bev = torch.cat((bev, torch.zeros((len(scenes) - bev.shape[0],) + tuple(bev.shape[1:]))), dim=0)
    lt = VO.vat_lidar(bev, sd_lidar, cfg.n_heads)
    fused = VO.vat_block(lt, torch.as_tensor(patches), sd_fuse, "", cfg.n_heads)
    res.update(pillar_features=pf, pillar_coords=bp["voxel_coords"], lidar_tokens=lt, fused=fused)
    return res
