"""oracle/lidar_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy for index arithmetic, torch-CPU fp32 for the float ops) of the LiDAR half
of the hot path of Advaith-Sajeev/LiDAR-Vision-VQA.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module; the product package never does.

Pinning:
  * mean_vfe / pillar_vfe / pointpillar_scatter are checked against the UNMODIFIED reference classes
    imported in the build container (tools/make_goldens.py -> tests/golden/lidar_*.npz).
  * voxelize_hard follows un-vendored, un-pinned spconv -> "parity unpinned" for that function
    (see voxel_oracle.c header); pinned by KATs + the hard-vs-dynamic identity.
  * dynamic_* follow dynamic_mean_vfe.py / dynamic_pillar_vfe.py / dynamic_voxel_vfe.py line by
    line; the reference classes cannot run here (ctor calls .cuda(), torch_scatter not installed), so
    they are pinned by the source text, by KATs, and by the shared PFN arithmetic checked through the
    hard PillarVFE goldens.

All file:line citations are relative to /root/reference/src/lidar-encoder/.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvoxel_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.orc_voxelize_hard.restype = ctypes.c_int
        lib.orc_dynamic_keys.restype = ctypes.c_int64
        lib.orc_mean_vfe.restype = None
        _LIB = lib
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------
# a1: range mask  (pcdet/utils/common_utils.py:78-81 via data_processor.py:79-93)
# --------------------------------------------------------------------------------------
def mask_points_by_range(points: np.ndarray, limit_range: Sequence[float]) -> np.ndarray:
    """x and y tested with INCLUSIVE bounds, z not tested at all."""
    lr = np.asarray(limit_range, dtype=np.float32)
    return (points[:, 0] >= lr[0]) & (points[:, 0] <= lr[3]) & (points[:, 1] >= lr[1]) & (points[:, 1] <= lr[4])


# --------------------------------------------------------------------------------------
# a3: grid size + hard voxeliser  (data_processor.py:133-180, 16-61)
# --------------------------------------------------------------------------------------
def grid_size(point_cloud_range: Sequence[float], voxel_size: Sequence[float]) -> np.ndarray:
    """data_processor.py:135-136: range is float32, VOXEL_SIZE a python list (float64)."""
    pcr = np.asarray(point_cloud_range, dtype=np.float32)
    g = (pcr[3:6] - pcr[0:3]) / np.array(voxel_size)
    return np.round(g).astype(np.int64)


class VoxelGenerator:
    """Same constructor keywords as the spconv-2 branch of VoxelGeneratorWrapper
    (data_processor.py:37-43); `generate(points)` returns (voxels, coordinates_zyx, num_points)."""

    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel,
                 max_num_voxels, break_on_cap: bool = False):
        self.vsize = np.asarray(vsize_xyz, dtype=np.float32)
        self.range = np.asarray(coors_range_xyz, dtype=np.float32)
        self.c = int(num_point_features)
        self.t = int(max_num_points_per_voxel)
        self.max_voxels = int(max_num_voxels)
        self.break_on_cap = bool(break_on_cap)
        self.grid = np.round((self.range[3:] - self.range[:3]) / self.vsize).astype(np.int32)
        self._lut = np.full(int(self.grid[0]) * int(self.grid[1]) * int(self.grid[2]), -1, dtype=np.int32)

    def generate(self, points: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        pts = np.ascontiguousarray(points, dtype=np.float32)
        assert pts.ndim == 2 and pts.shape[1] == self.c
        n = pts.shape[0]
        cap = min(self.max_voxels, max(n, 1))
        voxels = np.empty((cap, self.t, self.c), dtype=np.float32)
        coords = np.empty((cap, 3), dtype=np.int32)
        num = np.empty((cap,), dtype=np.int32)
        m = _lib().orc_voxelize_hard(_p(pts), ctypes.c_int64(n), ctypes.c_int(self.c), _p(self.range),
                                     _p(self.vsize), _p(self.grid), ctypes.c_int(self.t),
                                     ctypes.c_int(self.max_voxels), ctypes.c_int(int(self.break_on_cap)),
                                     _p(self._lut), _p(voxels), _p(coords), _p(num))
        return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()


# --------------------------------------------------------------------------------------
# a4: collate_batch (pcdet/datasets/dataset.py:230-244) + load_data_to_gpu's float32 cast
#     (pcdet/models/__init__.py:36)
# --------------------------------------------------------------------------------------
def collate_batch(scenes: List[Dict[str, np.ndarray]]) -> Dict[str, np.ndarray]:
    ret: Dict[str, np.ndarray] = {}
    for key in scenes[0].keys():
        vals = [s[key] for s in scenes]
        if key in ("voxels", "voxel_num_points"):
            ret[key] = np.concatenate(vals, axis=0)
        elif key in ("points", "voxel_coords"):
            ret[key] = np.concatenate(
                [np.pad(v, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, v in enumerate(vals)], axis=0)
        else:
            ret[key] = np.stack(vals, axis=0)
    ret["batch_size"] = len(scenes)
    return ret


# --------------------------------------------------------------------------------------
# a5: MeanVFE  (vfe/mean_vfe.py:25-29)
# --------------------------------------------------------------------------------------
def mean_vfe(voxels: np.ndarray, num_points: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(voxels, dtype=np.float32)
    n = np.ascontiguousarray(num_points, dtype=np.int32)
    m, t, c = v.shape
    out = np.empty((m, c), dtype=np.float32)
    _lib().orc_mean_vfe(_p(v), _p(n), ctypes.c_int64(m), ctypes.c_int(t), ctypes.c_int(c), _p(out))
    return out


# --------------------------------------------------------------------------------------
# a6: PillarVFE + PFNLayer (vfe/pillar_vfe.py:8-49, 94-123), eval-mode BatchNorm
# --------------------------------------------------------------------------------------
def _bn_eval(x: torch.Tensor, w, b, mean, var, eps=1e-3) -> torch.Tensor:
    return (x - mean) / torch.sqrt(var + eps) * w + b


def pfn_layer(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, last: bool, use_norm=True) -> torch.Tensor:
    """x [M,T,Cin] -> [M,1,Cout] (last) or [M,T,2*Cout/..] (pillar_vfe.py:28-49)."""
    y = x @ sd[prefix + "linear.weight"].t()
    if use_norm:
        y = _bn_eval(y, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"],
                     sd[prefix + "norm.running_mean"], sd[prefix + "norm.running_var"])
    else:
        y = y + sd[prefix + "linear.bias"]
    y = torch.relu(y)
    y_max = y.max(dim=1, keepdim=True)[0]
    if last:
        return y_max
    return torch.cat([y, y_max.expand(-1, x.shape[1], -1)], dim=2)


def pillar_vfe(voxels, num_points, coords_bzyx, sd, voxel_size, point_cloud_range, num_filters,
               use_norm=True, with_distance=False, use_absolute_xyz=True) -> torch.Tensor:
    """pillar_vfe.py:94-123.  coords are (b,z,y,x); returns pillar_features [M, num_filters[-1]]."""
    v = torch.as_tensor(voxels, dtype=torch.float32)
    n = torch.as_tensor(num_points).to(torch.float32)
    co = torch.as_tensor(coords_bzyx).to(torch.float32)
    vx, vy, vz = (float(s) for s in voxel_size)
    xo, yo, zo = vx / 2 + point_cloud_range[0], vy / 2 + point_cloud_range[1], vz / 2 + point_cloud_range[2]
    points_mean = v[:, :, :3].sum(dim=1, keepdim=True) / n.view(-1, 1, 1)
    f_cluster = v[:, :, :3] - points_mean
    f_center = torch.zeros_like(v[:, :, :3])
    f_center[:, :, 0] = v[:, :, 0] - (co[:, 3].unsqueeze(1) * vx + xo)
    f_center[:, :, 1] = v[:, :, 1] - (co[:, 2].unsqueeze(1) * vy + yo)
    f_center[:, :, 2] = v[:, :, 2] - (co[:, 1].unsqueeze(1) * vz + zo)
    feats = [v if use_absolute_xyz else v[..., 3:], f_cluster, f_center]
    if with_distance:
        feats.append(torch.norm(v[:, :, :3], 2, 2, keepdim=True))
    f = torch.cat(feats, dim=-1)
    t = f.shape[1]
    mask = (n.int().unsqueeze(1) > torch.arange(t, dtype=torch.int).view(1, -1)).unsqueeze(-1).to(f.dtype)
    f = f * mask
    nl = len(num_filters)
    for i in range(nl):
        f = pfn_layer(f, sd, f"pfn_layers.{i}.", last=(i >= nl - 1), use_norm=use_norm)
    return f.squeeze(1) if f.shape[0] != 1 else f.squeeze()


# --------------------------------------------------------------------------------------
# a7: dynamic voxelisation + DynamicMeanVFE / DynamicPillarVFE / DynamicVoxelVFE / ...Simple2D
#     (vfe/dynamic_mean_vfe.py:37-76, dynamic_pillar_vfe.py:14-46,90-142,219-240,
#      dynamic_voxel_vfe.py:57-106)
# --------------------------------------------------------------------------------------
def dynamic_voxelize(points_bxyzi: np.ndarray, point_cloud_range, voxel_size, grid, ndim: int):
    """Returns dict(keep [N] bool, coords [N',3] i32, unq_key [M] i32, unq_inv [N'] i64, unq_cnt [M] i64).
    `torch.unique(sorted=True)` order == ascending signed int32 key == np.unique order."""
    pts = np.ascontiguousarray(points_bxyzi, dtype=np.float32)
    n, c = pts.shape
    rng = np.asarray(point_cloud_range, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    g = np.asarray(grid, dtype=np.int32)
    keys = np.zeros(n, dtype=np.int32)
    cxyz = np.zeros((n, 3), dtype=np.int32)
    valid = np.zeros(n, dtype=np.uint8)
    _lib().orc_dynamic_keys(_p(pts), ctypes.c_int64(n), ctypes.c_int(c), _p(rng), _p(vs), _p(g),
                            ctypes.c_int(ndim), _p(keys), _p(cxyz), _p(valid))
    keep = valid.astype(bool)
    k = keys[keep]
    unq, inv, cnt = np.unique(k, return_inverse=True, return_counts=True)
    return dict(keep=keep, coords=cxyz[keep], unq_key=unq.astype(np.int32), unq_inv=inv.astype(np.int64).reshape(-1),
                unq_cnt=cnt.astype(np.int64))


def scatter_mean(x: torch.Tensor, inv: torch.Tensor, m: int) -> torch.Tensor:
    """torch_scatter.scatter_mean(x, inv, dim=0): segment sum / clamp(count, 1)."""
    out = torch.zeros((m, x.shape[1]), dtype=x.dtype)
    out.index_add_(0, inv, x)
    cnt = torch.bincount(inv, minlength=m).clamp(min=1).to(x.dtype).view(-1, 1)
    return out / cnt


def scatter_max(x: torch.Tensor, inv: torch.Tensor, m: int) -> torch.Tensor:
    out = torch.full((m, x.shape[1]), float("-inf"), dtype=x.dtype)
    return out.scatter_reduce(0, inv.view(-1, 1).expand(-1, x.shape[1]), x, reduce="amax", include_self=True)


def pfn_layer_v2(x, inv, m, sd, prefix, last, use_norm=True):
    """dynamic_pillar_vfe.py:35-46 (BatchNorm1d on [N,C], eval mode)."""
    y = x @ sd[prefix + "linear.weight"].t()
    if use_norm:
        y = _bn_eval(y, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"],
                     sd[prefix + "norm.running_mean"], sd[prefix + "norm.running_var"])
    else:
        y = y + sd[prefix + "linear.bias"]
    y = torch.relu(y)
    y_max = scatter_max(y, inv, m)
    if last:
        return y_max
    return torch.cat([y, y_max[inv, :]], dim=1)


def _decode_coords(unq: np.ndarray, grid, ndim: int) -> np.ndarray:
    """key -> (b, z, y, x) int32, with the reference's int32 floor-division (torch `//` on int tensors
    floors) -- dynamic_mean_vfe.py:66-71 / dynamic_pillar_vfe.py:129-135."""
    u = unq.astype(np.int32)
    gx, gy, gz = (int(v) for v in grid)
    if ndim == 3:
        sxyz, syz, sz = np.int32(gx * gy * gz), np.int32(gy * gz), np.int32(gz)
        vc = np.stack((u // sxyz, (u % sxyz) // syz, (u % syz) // sz, u % sz), axis=1)
    else:
        sxy, sy = np.int32(gx * gy), np.int32(gy)
        vc = np.stack((u // sxy, (u % sxy) // sy, u % sy, np.zeros_like(u)), axis=1)
    return np.ascontiguousarray(vc[:, [0, 3, 2, 1]].astype(np.int32))


def dynamic_mean_vfe(points_bxyzi, point_cloud_range, voxel_size, grid):
    dv = dynamic_voxelize(points_bxyzi, point_cloud_range, voxel_size, grid, 3)
    pts = torch.as_tensor(np.asarray(points_bxyzi, dtype=np.float32)[dv["keep"]])
    inv = torch.as_tensor(dv["unq_inv"])
    m = len(dv["unq_key"])
    feats = scatter_mean(pts[:, 1:].contiguous(), inv, m)
    return dict(voxel_features=feats, voxel_coords=_decode_coords(dv["unq_key"], grid, 3), **dv)


def dynamic_pfn_vfe(points_bxyzi, point_cloud_range, voxel_size, grid, sd, num_filters, kind: str,
                    use_norm=True, with_distance=False, use_absolute_xyz=True):
    """kind in {"pillar" (DynamicPillarVFE), "voxel" (DynamicVoxelVFE), "simple2d" (DynamicPillarVFESimple2D)}."""
    ndim = 3 if kind == "voxel" else 2
    dv = dynamic_voxelize(points_bxyzi, point_cloud_range, voxel_size, grid, ndim)
    pts = torch.as_tensor(np.asarray(points_bxyzi, dtype=np.float32)[dv["keep"]])
    pc = torch.as_tensor(dv["coords"]).to(torch.float32)
    inv = torch.as_tensor(dv["unq_inv"])
    m = len(dv["unq_key"])
    vx, vy, vz = (float(s) for s in voxel_size)
    xo, yo, zo = vx / 2 + point_cloud_range[0], vy / 2 + point_cloud_range[1], vz / 2 + point_cloud_range[2]
    xyz = pts[:, 1:4].contiguous()
    f_center = torch.zeros_like(xyz)
    f_center[:, 0] = xyz[:, 0] - (pc[:, 0] * vx + xo)
    f_center[:, 1] = xyz[:, 1] - (pc[:, 1] * vy + yo)
    f_center[:, 2] = xyz[:, 2] - ((pc[:, 2] * vz + zo) if kind == "voxel" else zo)
    if kind == "simple2d":
        feats = [f_center, pts[:, 1:] if use_absolute_xyz else pts[:, 4:]]
    else:
        mean = scatter_mean(xyz, inv, m)
        f_cluster = xyz - mean[inv, :]
        feats = [pts[:, 1:] if use_absolute_xyz else pts[:, 4:], f_cluster, f_center]
    if with_distance:
        feats.append(torch.norm(pts[:, 1:4], 2, dim=1, keepdim=True))
    f = torch.cat(feats, dim=-1)
    nl = len(num_filters)
    for i in range(nl):
        f = pfn_layer_v2(f, inv, m, sd, f"pfn_layers.{i}.", last=(i >= nl - 1), use_norm=use_norm)
    coords = _decode_coords(dv["unq_key"], grid, 2)
    if kind == "voxel":
        coords = _decode_coords(dv["unq_key"], grid, 3)
    if kind == "simple2d":
        coords = np.ascontiguousarray(coords[:, [0, 2, 3]])  # (b, y, x): dynamic_pillar_vfe.py:232-236
    return dict(features=f, voxel_coords=coords, **dv)


# --------------------------------------------------------------------------------------
# a8: PointPillarScatter (backbones_2d/map_to_bev/pointpillar_scatter.py:14-37)
# --------------------------------------------------------------------------------------
def pointpillar_scatter(pillar_features: torch.Tensor, coords_bzyx, nx: int, ny: int, nz: int = 1) -> torch.Tensor:
    assert nz == 1
    pf = torch.as_tensor(pillar_features, dtype=torch.float32)
    co = torch.as_tensor(np.asarray(coords_bzyx)).to(torch.int64)
    c = pf.shape[1]
    bsz = int(co[:, 0].max().item()) + 1
    out = torch.zeros((bsz, c, nz * nx * ny), dtype=pf.dtype)
    for b in range(bsz):
        msk = co[:, 0] == b
        tc = co[msk]
        idx = tc[:, 1] + tc[:, 2] * nx + tc[:, 3]
        out[b][:, idx] = pf[msk].t()
    return out.view(bsz, c * nz, ny, nx)
