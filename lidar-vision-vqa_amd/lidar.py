"""LiDAR side of the hot path with the reference's (OpenPCDet) plugin interfaces, MI355X-native.

Mirrors, name for name (citations relative to /root/reference/src/lidar-encoder/pcdet/):
  mask_points_by_range            utils/common_utils.py:78-81
  VoxelGeneratorWrapper.generate  datasets/processor/data_processor.py:16-61   (spconv hard voxeliser)
  voxelize_batch                  = per-scene generate() + collate_batch (datasets/dataset.py:230-244), fused
  MeanVFE / PillarVFE             models/backbones_3d/vfe/{mean_vfe,pillar_vfe}.py
  DynamicMeanVFE / DynamicPillarVFE / DynamicPillarVFESimple2D / DynamicVoxelVFE
                                  models/backbones_3d/vfe/dynamic_*.py
  PointPillarScatter              models/backbones_2d/map_to_bev/pointpillar_scatter.py:5-37
  __all__ / map_to_bev_all        the `vfe.__all__[NAME](model_cfg=..., num_point_features=..., ...)` registries
                                  (vfe/__init__.py:9-18, detector3d_template.py:52-66,85-95)

Same constructor keywords, same `forward(batch_dict) -> batch_dict` protocol, same state_dict keys;
the arithmetic runs in liblvq_hip.so.  Inference only (eval-mode BatchNorm is folded); calling a
module in train() mode raises.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _ffi as F

_WS: Dict[Tuple[int, str], torch.Tensor] = {}


def workspace(nbytes: int, device: torch.device, tag: str = "vox") -> torch.Tensor:
    """Grow-only per-device scratch buffer handed to the C ABI (the library never allocates)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def grid_size_from(point_cloud_range, voxel_size) -> np.ndarray:
    """data_processor.py:135-136."""
    pcr = np.asarray(point_cloud_range, dtype=np.float32)
    return np.round((pcr[3:6] - pcr[0:3]) / np.array(voxel_size)).astype(np.int64)


def mask_points_by_range(points: torch.Tensor, limit_range) -> torch.Tensor:
    F.require_cuda(points)
    n, c = points.shape
    keep = torch.empty(n, dtype=torch.uint8, device=points.device)
    rc = F.lib().lvq_mask_points_by_range(F.ptr(points), F.i64(n), F.cint(c), F.f32x(list(limit_range)), F.ptr(keep),
                                          F.stream_ptr(points.device))
    F.check(rc, "lvq_mask_points_by_range")
    return keep.bool()


class VoxelGeneratorWrapper:
    """Same constructor keywords as data_processor.py:17; generate(points) -> (voxels, coordinates(z,y,x),
    num_points).  numpy in -> numpy out (drop-in for the CPU dataloader); CUDA tensor in -> CUDA tensors out."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels,
                 break_on_cap: bool = False, device: Optional[torch.device] = None):
        self.vsize = [float(np.float32(v)) for v in vsize_xyz]
        self.range = [float(np.float32(v)) for v in coors_range_xyz]
        r = np.asarray(coors_range_xyz, dtype=np.float32)
        self.grid = np.round((r[3:] - r[:3]) / np.asarray(vsize_xyz, dtype=np.float32)).astype(np.int32)
        self.c = int(num_point_features)
        self.t = int(max_num_points_per_voxel)
        self.max_voxels = int(max_num_voxels)
        self.break_on_cap = bool(break_on_cap)
        self.device = device

    # -- batched, device-resident form (a3 + a4 fused) --------------------------------------------
    def generate_batch_device(self, points: torch.Tensor, scene_off: torch.Tensor, n_scenes: int):
        """points [N,c] fp32 CUDA (scenes back to back), scene_off [S+1] int32 CUDA.
        Returns (voxels [cap,T,c], coords_bzyx [cap,4] i32, num_points [cap] i32, scene_voxel_off [S+1] i32);
        rows >= scene_voxel_off[S] are unspecified.  No host synchronisation."""
        F.require_cuda(points, scene_off)
        assert points.dtype == torch.float32 and scene_off.dtype == torch.int32
        n, c = points.shape
        assert c == self.c
        dev = points.device
        cap = max(1, min(n, n_scenes * self.max_voxels))
        voxels = torch.empty((cap, self.t, c), dtype=torch.float32, device=dev)
        coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        num = torch.empty((cap,), dtype=torch.int32, device=dev)
        svo = torch.empty((n_scenes + 1,), dtype=torch.int32, device=dev)
        L = F.lib()
        nbytes = L.lvq_voxelize_hard_workspace_bytes(F.i64(n), F.cint(n_scenes))
        ws = workspace(nbytes, dev, getattr(self, "ws_tag", "vox"))      # generators that run on different streams need their own
        rc = L.lvq_voxelize_hard(F.ptr(points), F.ptr(scene_off), F.i64(n), F.cint(n_scenes), F.cint(c),
                                 F.f32x(self.range), F.f32x(self.vsize), F.i32x(self.grid.tolist()),
                                 F.cint(self.t), F.cint(self.max_voxels), F.cint(int(self.break_on_cap)),
                                 F.i64(cap), F.ptr(voxels), F.ptr(coords), F.ptr(num), F.ptr(svo),
                                 F.ptr(ws), F.csize(ws.numel()), F.stream_ptr(dev))
        F.check(rc, "lvq_voxelize_hard")
        return voxels, coords, num, svo

    def generate_mean_device(self, points: torch.Tensor, scene_off: torch.Tensor, n_scenes: int):
        """transform_points_to_voxels + MeanVFE in one call (SURVEY 8d "fused voxelise -> mean"): returns
        (voxel_features [cap,c], coords_bzyx [cap,4] i32, num_points [cap] i32, scene_voxel_off [S+1] i32), bit-identical to
        generate_batch_device + MeanVFE.forward_device, without the padded [cap,T,c] tensor.  Shapes the fused kernels do
        not take run that pair."""
        F.require_cuda(points, scene_off)
        assert points.dtype == torch.float32 and scene_off.dtype == torch.int32
        n, c = points.shape
        assert c == self.c
        dev = points.device
        L = F.lib()
        if not self.break_on_cap:
            cap = max(1, min(n, n_scenes * self.max_voxels))
            feats = torch.empty((cap, c), dtype=torch.float32, device=dev)
            coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
            num = torch.empty((cap,), dtype=torch.int32, device=dev)
            svo = torch.empty((n_scenes + 1,), dtype=torch.int32, device=dev)
            nbytes = L.lvq_voxelize_hard_workspace_bytes(F.i64(n), F.cint(n_scenes))
            ws = workspace(nbytes, dev, getattr(self, "ws_tag", "vox"))
            rc = L.lvq_voxelize_mean(F.ptr(points), F.ptr(scene_off), F.i64(n), F.cint(n_scenes), F.cint(c), F.f32x(self.range),
                                     F.f32x(self.vsize), F.i32x(self.grid.tolist()), F.cint(self.t), F.cint(self.max_voxels),
                                     F.i64(cap), F.ptr(feats), F.ptr(coords), F.ptr(num), F.ptr(svo), F.ptr(ws), F.csize(ws.numel()),
                                     F.stream_ptr(dev))
            if rc != F.LVQ_EUNSUPPORTED:       # unsupported shape: fall through to the two-call route
                F.check(rc, "lvq_voxelize_mean")
                return feats, coords, num, svo
        voxels, coords, num, svo = self.generate_batch_device(points, scene_off, n_scenes)
        feats = MeanVFE(None, c).forward_device(voxels, num, svo[n_scenes:])
        return feats, coords, num, svo

    def generate(self, points):
        is_np = isinstance(points, np.ndarray)
        dev = self.device or (torch.device("cuda", torch.cuda.current_device()) if is_np else points.device)
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(dev) if is_np else points.contiguous()
        n = pts.shape[0]
        off = torch.tensor([0, n], dtype=torch.int32, device=dev)
        voxels, coords, num, svo = self.generate_batch_device(pts, off, 1)
        m = int(svo[1].item())
        voxels, coords, num = voxels[:m], coords[:m, 1:].contiguous(), num[:m]
        if is_np:
            return voxels.cpu().numpy(), coords.cpu().numpy(), num.cpu().numpy()
        return voxels, coords, num


def voxelize_batch(gen: VoxelGeneratorWrapper, scenes: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transform_points_to_voxels per scene + collate_batch + load_data_to_gpu in one device pass:
    returns the pcdet batch_dict entries `points [sum N,1+c]`, `voxels`, `voxel_coords [sum M,4]` (b,z,y,x),
    `voxel_num_points`, `batch_size`."""
    dev = scenes[0].device
    lens = [int(s.shape[0]) for s in scenes]
    off = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device=dev)
    pts = torch.cat(list(scenes), dim=0).contiguous()
    voxels, coords, num, svo = gen.generate_batch_device(pts, off, len(scenes))
    m = int(svo[-1].item())
    bidx = torch.repeat_interleave(torch.arange(len(scenes), device=dev, dtype=torch.float32),
                                   torch.tensor(lens, device=dev))
    return dict(points=torch.cat((bidx.unsqueeze(1), pts), dim=1), voxels=voxels[:m], voxel_coords=coords[:m],
                voxel_num_points=num[:m], batch_size=len(scenes), scene_voxel_off=svo)


# --------------------------------------------------------------------------------------------------
# VFE modules (batch_dict protocol)
# --------------------------------------------------------------------------------------------------
def _as_i32(t: torch.Tensor) -> torch.Tensor:
    """load_data_to_gpu hands coords / counts over as float32 (pcdet/models/__init__.py:36)."""
    return t if t.dtype == torch.int32 else t.to(torch.int32)


class VFETemplate(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError

    def _inference_only(self):
        if self.training:
            raise F.LvqError(f"{type(self).__name__}: the MI355X path is inference-only (BatchNorm folded); call .eval()")


class MeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward_device(self, voxels: torch.Tensor, num: torch.Tensor, n_live: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sync-free form: rows >= n_live[0] (device int32) are skipped; returns [cap, c]."""
        F.require_cuda(voxels, num)
        m, t, c = voxels.shape
        out = torch.empty((m, c), dtype=torch.float32, device=voxels.device)
        rc = F.lib().lvq_mean_vfe(F.ptr(voxels), F.ptr(num), F.i64(m), F.ptr(n_live), F.cint(t), F.cint(c), F.ptr(out),
                                  F.stream_ptr(voxels.device))
        F.check(rc, "lvq_mean_vfe")
        return out

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        voxels = batch_dict["voxels"].contiguous()
        num = _as_i32(batch_dict["voxel_num_points"]).contiguous()
        batch_dict["voxel_features"] = self.forward_device(voxels, num)
        return batch_dict


class _PFNParams(nn.Module):
    """Parameter container with the reference's PFNLayer / PFNLayerV2 keys
    (pillar_vfe.py:8-26, dynamic_pillar_vfe.py:14-33); never called."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)
        self.cin, self.cout = in_channels, out_channels

    def folded(self):
        """(W, scale, shift) with the eval-mode BatchNorm folded into a per-channel affine.  Input-independent: computed once
        per weights version (parameter / buffer storage + in-place version counters), not per forward."""
        src = [self.linear.weight] + ([self.norm.weight, self.norm.bias, self.norm.running_mean, self.norm.running_var]
                                      if self.use_norm else [self.linear.bias])
        ver = tuple((t.data_ptr(), t._version, t.device) for t in src)
        hit = getattr(self, "_folded_cache", None)
        if hit is not None and hit[0] == ver:
            return hit[1]
        w = self.linear.weight.detach().float().contiguous()
        if self.use_norm:
            scale = self.norm.weight.detach().float() / torch.sqrt(self.norm.running_var.float() + self.norm.eps)
            shift = self.norm.bias.detach().float() - self.norm.running_mean.float() * scale
        else:
            scale = torch.ones(self.cout, device=w.device)
            shift = self.linear.bias.detach().float()
        out = (w, scale.contiguous(), shift.contiguous())
        object.__setattr__(self, "_folded_cache", (ver, out))
        return out


class _PFNStack(VFETemplate):
    def _build(self, num_point_features, extra):
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        num_point_features += extra if self.use_absolute_xyz else extra - 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = self.model_cfg.NUM_FILTERS
        assert len(self.num_filters) > 0
        nf = [num_point_features] + list(self.num_filters)
        self.pfn_layers = nn.ModuleList(
            [_PFNParams(nf[i], nf[i + 1], self.use_norm, last_layer=(i >= len(nf) - 2)) for i in range(len(nf) - 1)])

    def _geom(self, voxel_size, point_cloud_range):
        self.voxel_x, self.voxel_y, self.voxel_z = voxel_size[0], voxel_size[1], voxel_size[2]
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def _abi_layers(self):
        ws, scs, shs = zip(*[l.folded() for l in self.pfn_layers])
        keep = (ws, scs, shs)
        cin = [l.cin for l in self.pfn_layers]
        cout = [l.cout for l in self.pfn_layers]
        flags = (1 if self.use_absolute_xyz else 0) | (2 if self.with_distance else 0)
        return keep, F.ptr_array(ws), F.ptr_array(scs), F.ptr_array(shs), F.i32x(cin), F.i32x(cout), flags, cout[-1]


class PillarVFE(_PFNStack):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self._build(num_point_features, 6)
        self._geom(voxel_size, point_cloud_range)

    @torch.no_grad()
    def forward_device(self, voxels, num, coords, n_live: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sync-free form on capacity-sized buffers (int32 num / coords); rows >= n_live[0] are skipped."""
        self._inference_only()
        F.require_cuda(voxels, num, coords)
        m, t, c = voxels.shape
        keep, wp, sp, hp, cin, cout, flags, c_last = self._abi_layers()
        out = torch.empty((m, c_last), dtype=torch.float32, device=voxels.device)
        rc = F.lib().lvq_pillar_vfe(F.ptr(voxels), F.ptr(num), F.ptr(coords), F.i64(m), F.ptr(n_live), F.cint(t), F.cint(c),
                                    F.cint(len(self.pfn_layers)), wp, sp, hp, cin, cout, F.cint(flags),
                                    F.f32x([self.voxel_x, self.voxel_y, self.voxel_z]),
                                    F.f32x([self.x_offset, self.y_offset, self.z_offset]), F.ptr(out),
                                    F.stream_ptr(voxels.device))
        F.check(rc, "lvq_pillar_vfe")
        del keep
        return out

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        voxels = batch_dict["voxels"].contiguous()
        num = _as_i32(batch_dict["voxel_num_points"]).contiguous()
        coords = _as_i32(batch_dict["voxel_coords"]).contiguous()
        out = self.forward_device(voxels, num, coords)
        batch_dict["pillar_features"] = out.squeeze()   # pillar_vfe.py:121 `features.squeeze()`
        return batch_dict


def _dynamic_voxelize(points: torch.Tensor, batch_size: int, pcr, vsize, grid, ndim: int):
    F.require_cuda(points)
    pts = points.contiguous()
    n, c = pts.shape
    dev = pts.device
    g = [int(v) for v in grid]
    ks = batch_size * g[0] * g[1] * (g[2] if ndim == 3 else 1)
    cap = max(1, min(n, ks))
    inv = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    pc = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
    key = torch.empty((cap,), dtype=torch.int32, device=dev)
    cnt = torch.empty((cap,), dtype=torch.int32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    counts = torch.empty((2,), dtype=torch.int32, device=dev)
    L = F.lib()
    gi = F.i32x(g)
    nbytes = L.lvq_voxelize_dynamic_workspace_bytes(F.i64(n), F.cint(batch_size), gi, F.cint(ndim))
    if nbytes == 0:
        F.check(-4, "lvq_voxelize_dynamic (batch_size * grid cells must stay below 2^31)")
    ws = workspace(nbytes, dev)
    rc = L.lvq_voxelize_dynamic(F.ptr(pts), F.i64(n), F.cint(c), F.cint(batch_size), F.f32x([float(np.float32(v)) for v in pcr]),
                                F.f32x([float(np.float32(v)) for v in vsize]), gi, F.cint(ndim), F.ptr(inv), F.ptr(pc),
                                F.ptr(key), F.ptr(cnt), F.ptr(coords), F.ptr(counts), F.ptr(ws), F.csize(ws.numel()),
                                F.stream_ptr(dev))
    F.check(rc, "lvq_voxelize_dynamic")
    return dict(pts=pts, inv=inv, pt_coords=pc, unq_key=key, unq_cnt=cnt, coords=coords, counts=counts, cap=cap)


def _scatter_mean(dv, col0: int, nc: int) -> torch.Tensor:
    pts = dv["pts"]
    n, c = pts.shape
    sums = torch.zeros((dv["cap"], nc), dtype=torch.float32, device=pts.device)
    rc = F.lib().lvq_scatter_mean(F.ptr(pts), F.i64(n), F.cint(c), F.cint(col0), F.cint(nc), F.ptr(dv["inv"]),
                                  F.ptr(dv["unq_cnt"]), F.i64(dv["cap"]), F.ptr(sums), F.ptr(sums), F.stream_ptr(pts.device))
    F.check(rc, "lvq_scatter_mean")
    return sums


class DynamicMeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.grid_size = [int(v) for v in grid_size]
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        points = batch_dict["points"]
        dv = _dynamic_voxelize(points, int(batch_dict["batch_size"]), self.point_cloud_range, self.voxel_size,
                               self.grid_size, 3)
        feats = _scatter_mean(dv, 1, points.shape[1] - 1)
        m = int(dv["counts"][0].item())
        batch_dict["voxel_features"] = feats[:m].contiguous()
        batch_dict["voxel_coords"] = dv["coords"][:m].contiguous()
        return batch_dict


class _DynamicPFN(_PFNStack):
    KIND = 0
    NDIM = 2
    EXTRA = 6

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self._build(num_point_features, self.EXTRA)
        self._geom(voxel_size, point_cloud_range)
        self.grid_size = [int(v) for v in grid_size]
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)

    def _run(self, batch_dict):
        self._inference_only()
        points = batch_dict["points"]
        bs = int(batch_dict["batch_size"]) if "batch_size" in batch_dict else int(points[:, 0].max().item()) + 1
        dv = _dynamic_voxelize(points, bs, self.point_cloud_range, self.voxel_size, self.grid_size, self.NDIM)
        pts = dv["pts"]
        n, c = pts.shape
        mean = _scatter_mean(dv, 1, 3) if self.KIND != 2 else None
        keep, wp, sp, hp, cin, cout, flags, c_last = self._abi_layers()
        cap = dv["cap"]
        out = torch.zeros((cap, c_last), dtype=torch.float32, device=pts.device)
        nl = len(self.pfn_layers)
        tmp = torch.zeros((cap, self.pfn_layers[0].cout), dtype=torch.float32, device=pts.device) if nl == 2 else None
        rc = F.lib().lvq_dynamic_pfn(F.ptr(pts), F.i64(n), F.cint(c), F.ptr(dv["inv"]), F.ptr(dv["pt_coords"]), F.ptr(mean),
                                     F.cint(self.KIND), F.cint(nl), wp, sp, hp, cin, cout, F.cint(flags),
                                     F.f32x([self.voxel_x, self.voxel_y, self.voxel_z]),
                                     F.f32x([self.x_offset, self.y_offset, self.z_offset]), F.ptr(tmp), F.ptr(out),
                                     F.stream_ptr(pts.device))
        F.check(rc, "lvq_dynamic_pfn")
        del keep
        m = int(dv["counts"][0].item())
        return out[:m].contiguous(), dv["coords"][:m].contiguous()


class DynamicPillarVFE(_DynamicPFN):
    KIND, NDIM, EXTRA = 0, 2, 6

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        feats, coords = self._run(batch_dict)
        batch_dict["voxel_features"] = batch_dict["pillar_features"] = feats
        batch_dict["voxel_coords"] = coords
        return batch_dict


class DynamicVoxelVFE(_DynamicPFN):
    KIND, NDIM, EXTRA = 1, 3, 6

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        feats, coords = self._run(batch_dict)
        batch_dict["pillar_features"] = batch_dict["voxel_features"] = feats
        batch_dict["voxel_coords"] = coords
        return batch_dict


class DynamicPillarVFESimple2D(_DynamicPFN):
    KIND, NDIM, EXTRA = 2, 2, 3   # no f_cluster: +3 only when USE_ABSLOTE_XYZ (dynamic_pillar_vfe.py:152-157)

    def _build(self, num_point_features, extra):
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        if self.use_absolute_xyz:
            num_point_features += 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = self.model_cfg.NUM_FILTERS
        nf = [num_point_features] + list(self.num_filters)
        self.pfn_layers = nn.ModuleList(
            [_PFNParams(nf[i], nf[i + 1], self.use_norm, last_layer=(i >= len(nf) - 2)) for i in range(len(nf) - 1)])

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        feats, coords = self._run(batch_dict)
        batch_dict["pillar_features"] = feats
        batch_dict["pillar_coords"] = coords[:, [0, 2, 3]].contiguous()   # (b, y, x)
        return batch_dict


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = (int(v) for v in grid_size)
        assert self.nz == 1

    @torch.no_grad()
    def forward_device(self, feats, coords, batch_size: int, n_live: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sync-free form: feats [cap,ch] fp32, coords [cap,4] int32 (b,z,y,x), rows >= n_live[0] skipped."""
        F.require_cuda(feats, coords)
        m, ch = feats.shape
        assert ch == self.num_bev_features
        canvas = out if out is not None else torch.empty((batch_size, ch * self.nz, self.ny, self.nx), dtype=torch.float32,
                                                         device=feats.device)
        rc = F.lib().lvq_pillar_scatter(F.ptr(feats), F.ptr(coords), F.i64(m), F.ptr(n_live), F.cint(ch), F.cint(batch_size),
                                        F.cint(self.ny), F.cint(self.nx), F.ptr(canvas), F.stream_ptr(feats.device))
        F.check(rc, "lvq_pillar_scatter")
        return canvas

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        feats = batch_dict["pillar_features"].contiguous()
        coords = _as_i32(batch_dict["voxel_coords"]).contiguous()
        if feats.dim() == 1:
            feats = feats.view(1, -1)
        batch_size = int(batch_dict["batch_size"]) if "batch_size" in batch_dict else int(coords[:, 0].max().item()) + 1
        batch_dict["spatial_features"] = self.forward_device(feats, coords, batch_size)
        return batch_dict


# registries with the reference's NAME strings (vfe/__init__.py:9-18; map_to_bev/__init__.py)
__all__ = {
    "VFETemplate": VFETemplate,
    "MeanVFE": MeanVFE,
    "PillarVFE": PillarVFE,
    "DynMeanVFE": DynamicMeanVFE,
    "DynPillarVFE": DynamicPillarVFE,
    "DynamicPillarVFESimple2D": DynamicPillarVFESimple2D,
    "DynamicVoxelVFE": DynamicVoxelVFE,
}
map_to_bev_all = {"PointPillarScatter": PointPillarScatter}
