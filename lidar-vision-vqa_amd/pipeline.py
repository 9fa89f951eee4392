"""The hot path end to end, device-resident and free of host synchronisation:

    points [S x N, 4] + image-patch tokens [S, P, d]  ->  fused LiDAR+image tokens [S, nq, d]

composed ONLY of the reference's own modules at their own interfaces:

    range mask (fused, see below) -> hard voxelise 0.1 m grid (T=10, 60 000) -> MeanVFE          [3-D branch, a1/a3/a5]
                                  -> hard voxelise pillars 0.2 m (T=20, 30 000) -> PillarVFE[64]
                                  -> PointPillarScatter [S,64,512,512]                           [a3/a6/a8]
                                  -> VATLiDAR(c_in=64, d, nq, L, h)  -> LiDAR tokens [S,nq,d]    [a9/a10]
    patches -> VATBlock(q = LiDAR tokens, kv = patches)              -> fused tokens [S,nq,d]    [a10]

mask_points_by_range (common_utils.py:78-81) is algebraically fused into the voxeliser: every point it
removes (x or y outside the inclusive range) also fails the voxeliser's `c < 0 || c >= grid` test, and
removing points that the voxeliser drops anyway changes neither voxel order nor slots, so the outputs are
bit-identical to mask-then-voxelise (tests/test_gpu_pipeline.py checks this against the oracle, which
masks first like the reference does).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import fusion, lidar, ops, synth


@dataclass
class PipelineConfig:
    n_points: int = 32768
    dist: str = "C"
    d_model: int = 768
    n_heads: int = 12
    n_queries: int = 576
    n_layers: int = 1
    n_patches: int = 196
    pillar_filters: List[int] = field(default_factory=lambda: [64])
    pc_range: tuple = synth.PC_RANGE_NUSC
    voxel_3d: tuple = synth.VOXEL_01
    t_3d: int = 10
    max_voxels_3d: int = 60000
    voxel_pillar: tuple = synth.VOXEL_PILLAR
    t_pillar: int = 20
    max_pillars: int = 30000
    weight_seed: int = 2024

    @property
    def bev_hw(self):
        g = lidar.grid_size_from(self.pc_range, self.voxel_pillar)
        return int(g[1]), int(g[0])

    def describe(self) -> str:
        h, w = self.bev_hw
        return (f"{self.n_points}-pt scenes (Dist-{self.dist}) -> hard voxelise {self.voxel_3d} T={self.t_3d} max={self.max_voxels_3d} + MeanVFE; "
                f"pillars {self.voxel_pillar} T={self.t_pillar} max={self.max_pillars} -> PillarVFE{self.pillar_filters} -> "
                f"PointPillarScatter {h}x{w} (as an index map; sparse key stream: K|V computed only for the cells with a pillar in their 3x3 neighbourhood, by LayerNorm and the K|V projection folded onto the 64-channel conv token, every other key from the per-model table; block 0 attends to the computed rows only, signed against the table's softmax totals) -> VATLiDAR(c_in={self.pillar_filters[-1]},d={self.d_model},nq={self.n_queries},"
                f"L={self.n_layers},h={self.n_heads}) -> VATBlock(q=LiDAR tokens, kv={self.n_patches} ViT-B/16 patches)")


class Cfg(dict):
    __getattr__ = dict.__getitem__


class FusionPipeline(torch.nn.Module):
    def __init__(self, cfg: PipelineConfig, device, precision: Optional[str] = None, dense_bev: bool = False):
        super().__init__()
        self.cfg = cfg
        self.dense_bev = dense_bev          # True: materialise the BEV canvas as the reference does (returned as out["bev"])
        self._side = None                   # side stream of the 3-D voxel branch (created on first use)
        rng = list(cfg.pc_range)
        self.gen3d = lidar.VoxelGeneratorWrapper(cfg.voxel_3d, rng, 4, cfg.t_3d, cfg.max_voxels_3d)
        self.gen3d.ws_tag = "vox3"           # runs on the side stream, concurrently with the pillar voxeliser
        self.genp = lidar.VoxelGeneratorWrapper(cfg.voxel_pillar, rng, 4, cfg.t_pillar, cfg.max_pillars)
        gp = lidar.grid_size_from(rng, cfg.voxel_pillar)
        self.mean_vfe = lidar.MeanVFE(Cfg(), 4)
        self.pillar_vfe = lidar.PillarVFE(Cfg(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=cfg.pillar_filters),
                                          4, list(cfg.voxel_pillar), rng)
        self.scatter = lidar.PointPillarScatter(Cfg(NUM_BEV_FEATURES=cfg.pillar_filters[-1]), gp)
        self.vat_lidar = fusion.VATLiDAR(cfg.pillar_filters[-1], cfg.d_model, cfg.n_queries, cfg.n_layers, cfg.n_heads)
        self.fuse = fusion.VATBlock(cfg.d_model, cfg.n_heads, 4 * cfg.d_model, 0.1)
        for i, m in enumerate((self.pillar_vfe, self.vat_lidar, self.fuse)):
            synth.load_seeded(m, cfg.weight_seed + i)
        self.to(device).eval()
        self.set_precision(precision)

    def set_precision(self, p: Optional[str]):
        self.vat_lidar.precision = p
        self.fuse.precision = p

    @torch.no_grad()
    def forward(self, points: torch.Tensor, scene_off: torch.Tensor, patches: torch.Tensor) -> Dict[str, torch.Tensor]:
        """points [sum N, 4] fp32, scene_off [S+1] int32, patches [S, P, d] fp32 -- all device resident."""
        S = patches.shape[0]
        # 3-D branch: "32k-point cloud voxelised to a 0.1 m grid" + per-voxel mean.  It shares nothing but the input points with the
        # pillar branch and consists of ~10 launch-latency-bound kernels, so it runs on a side stream under the pillar / fusion
        # kernels and is joined before returning (its own workspace: tag "vox3").
        main = torch.cuda.current_stream(points.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=points.device)
        side = main if os.environ.get("LVQ_NO_SIDE_STREAM") else self._side
        side.wait_stream(main) if side is not main else None
        with torch.cuda.stream(side):
            if os.environ.get("LVQ_NO_FUSED_MEAN"):
                # reference dataflow: padded [M,T,4] voxels, then MeanVFE over them
                vox3, co3, num3, svo3 = self.gen3d.generate_batch_device(points, scene_off, S)
                feat3 = self.mean_vfe.forward_device(vox3, num3, svo3[S:])
            else:
                # fused voxelise -> mean: bit-identical features, the padded tensor is never written
                vox3 = None
                feat3, co3, num3, svo3 = self.gen3d.generate_mean_device(points, scene_off, S)
        # pillar branch -> BEV
        voxp, cop, nump, svop = self.genp.generate_batch_device(points, scene_off, S)
        pf = self.pillar_vfe.forward_device(voxp, nump, cop, svop[S:])
        if self.dense_bev:
            # reference dataflow: PointPillarScatter materialises the [S, C, H, W] canvas, VATLiDAR consumes it
            bev = self.scatter.forward_device(pf, cop, S, svop[S:])
            lidar_tokens = self.vat_lidar(bev)                              # [S, nq, d]
        else:
            # sparse BEV bridge (default): scatter + VATLiDAR's refine conv as one gather over an index map -- same tokens bit
            # for bit (tests/test_gpu_pipeline.py), no 268 MB canvas write + read per 4 scenes
            bev = None
            h, w = self.cfg.bev_hw
            lidar_tokens = self.vat_lidar.forward_pillars(pf, cop, svop[S:], S, h, w)
        fused = self.fuse(lidar_tokens, patches)                            # [S, nq, d]
        if side is not main:
            main.wait_stream(side)
        for t_ in (vox3, co3, num3, svo3, feat3):
            if t_ is not None:
                t_.record_stream(main)
        return dict(fused=fused, lidar_tokens=lidar_tokens, voxel_features=feat3, voxel_coords=co3, voxel_num_points=num3,
                    scene_voxel_off=svo3, pillar_features=pf, pillar_coords=cop, scene_pillar_off=svop, bev=bev)


def synthetic_batch(cfg: PipelineConfig, n_scenes: int, seed0: int, device):
    """Device-resident synthetic inputs of SURVEY 8d: seeds seed0+i for points, 2000+seed0+i for patches."""
    pts = [synth.scene_points(cfg.dist, cfg.n_points, seed0 + i) for i in range(n_scenes)]
    off = np.concatenate(([0], np.cumsum([len(p) for p in pts]))).astype(np.int32)
    patches = np.stack([synth.image_patches(cfg.n_patches, cfg.d_model, 2000 + seed0 + i) for i in range(n_scenes)])
    return (torch.from_numpy(np.concatenate(pts)).to(device), torch.from_numpy(off).to(device),
            torch.from_numpy(patches).to(device), pts, patches)
