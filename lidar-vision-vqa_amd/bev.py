"""Data formats either side of the sparse backbone: the reference's module / function API on
the HIP kernels of csrc/bev_bridge.hip.

  collect_feature_tokens   training/data/utils.py:24-49         token -> .npy path index (recursive glob, first hit wins)
  save_bev_feature         get-data/precompute_bev_features.py:391-395   np.save(path, bev.astype(float16))
  BevFeatureStore          training/data/dataset.py:139-146     np.load + .float(): fp16 file -> pinned host -> async H2D ->
                                                                lvq_f16_to_f32 on the stream (exact)
  SparseTensor             the four attributes of spconv.SparseConvTensor the path touches (features, indices,
                           spatial_shape, batch_size) + dense(); spconv itself is not a dependency
  bev_out                  spconv_backbone_voxelnext.py:149-164 z-merge: unique (b, y, x) rows + index_add_
  HeightCompression        map_to_bev/height_compression.py:10-26
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _ffi as F
from .lidar import workspace


# --------------------------------------------------------------------------------------------------
# f2: on-disk BEV features
# --------------------------------------------------------------------------------------------------
def _npy_files(root: str):
    """Every `*.npy` below `root`, depth first, directories and files in name order (os.walk; deterministic on any filesystem)."""
    for base, dirs, files in os.walk(root):
        dirs.sort()
        for name in sorted(files):
            if name.endswith(".npy"):
                yield os.path.join(base, name)


def collect_feature_tokens(feature_dirs: List[str]) -> Dict[str, str]:
    """Index of the reference's on-disk BEV store (training/data/utils.py:24-49): sample_token = file stem -> path over every
    `.npy` below the given roots (the writer puts them in split sub-folders, precompute_bev_features.py:391-395).  A token found
    under several roots resolves to the EARLIEST root in `feature_dirs`; inside one root the first hit in name order wins
    (the reference takes glob order there, which the filesystem decides).  Roots that do not exist are skipped with one
    notice per job (rank 0)."""
    index: Dict[str, str] = {}
    absent = [d for d in feature_dirs if not os.path.isdir(d)]
    if absent and os.environ.get("RANK", "0") == "0":
        print("[bev] feature roots not found, skipped: " + ", ".join(map(str, absent)))
    for root in feature_dirs:
        if root in absent:
            continue
        for path in _npy_files(str(root)):
            index.setdefault(os.path.splitext(os.path.basename(path))[0], path)
    return index


def save_bev_feature(path, bev) -> None:
    """The writer side of the format (precompute_bev_features.py:391-395): one fp16 [C,H,W] array per sample token."""
    arr = bev.detach().cpu().numpy() if isinstance(bev, torch.Tensor) else np.asarray(bev)
    np.save(path, arr.astype(np.float16, copy=False))


def f16_to_f32(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    F.require_cuda(src)
    assert src.dtype == torch.float16
    out = out if out is not None else torch.empty(src.shape, dtype=torch.float32, device=src.device)
    F.check(F.lib().lvq_f16_to_f32(F.ptr(src), F.ptr(out), F.i64(src.numel()), F.stream_ptr(src.device)), "lvq_f16_to_f32")
    return out


class BevFeatureStore:
    """`bev = torch.from_numpy(np.load(token2path[tok])).float()` per sample + the collate stack (dataset.py:139-146),
    as one device-side batch: the fp16 files are memory-mapped, copied into ONE pinned staging buffer, sent with one
    asynchronous H2D copy and up-cast by the HIP kernel -- half the PCIe bytes of shipping fp32."""

    def __init__(self, feature_dirs: Sequence[str], device):
        self.token2path = collect_feature_tokens(list(feature_dirs))
        self.device = torch.device(device)
        self._pinned: Optional[torch.Tensor] = None
        self._copied: Optional[torch.cuda.Event] = None      # the staging buffer is reused: wait for the previous H2D first

    def __contains__(self, token: str) -> bool:
        return token in self.token2path

    def __len__(self) -> int:
        return len(self.token2path)

    def load(self, tokens: Sequence[str]) -> torch.Tensor:
        """[B, C, H, W] fp32 on the device, in the order of `tokens` (KeyError for an unknown token, as the reference)."""
        arrs = [np.load(self.token2path[t], mmap_mode="r") for t in tokens]
        shape = arrs[0].shape
        for a in arrs:
            if a.dtype != np.float16 or a.shape != shape:
                raise F.LvqError(f"BEV features must be fp16 arrays of one shape, got {a.dtype} {a.shape} vs {shape}")
        n = len(arrs) * int(np.prod(shape))
        if self._copied is not None:
            self._copied.synchronize()
        if self._pinned is None or self._pinned.numel() < n:
            self._pinned = torch.empty(n, dtype=torch.float16).pin_memory()
        host = self._pinned[:n].view(len(arrs), *shape)
        hv = host.numpy()
        for i, a in enumerate(arrs):
            hv[i] = a
        dev16 = host.to(self.device, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record(torch.cuda.current_stream(self.device))
        return f16_to_f32(dev16)


# --------------------------------------------------------------------------------------------------
# f3: sparse BEV bridge
# --------------------------------------------------------------------------------------------------
class SparseTensor:
    """features [M,C] fp32, indices [M,1+ndim] int32 (batch first), spatial_shape (D,H,W) or (H,W), batch_size."""

    def __init__(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(v) for v in spatial_shape]
        self.batch_size = int(batch_size)

    def replace_feature(self, features: torch.Tensor) -> "SparseTensor":
        return SparseTensor(features, self.indices, self.spatial_shape, self.batch_size)

    @torch.no_grad()
    def dense(self) -> torch.Tensor:
        """[N, C, D, H, W] (3-D) or [N, C, H, W] (2-D), zeros where no index points."""
        feats = self.features.contiguous()
        idx = self.indices.to(torch.int32).contiguous()
        F.require_cuda(feats, idx)
        m, c = feats.shape
        nd = len(self.spatial_shape)
        assert idx.shape[1] == nd + 1 and nd in (2, 3)
        d = self.spatial_shape[0] if nd == 3 else 1
        h, w = self.spatial_shape[-2], self.spatial_shape[-1]
        out = torch.empty((self.batch_size, c * d, h, w), dtype=torch.float32, device=feats.device)
        L = F.lib()
        ws = workspace(L.lvq_sparse_to_dense_workspace_bytes(F.cint(self.batch_size), F.cint(d), F.cint(h), F.cint(w)), feats.device, "dense")
        rc = L.lvq_sparse_to_dense(F.ptr(feats), F.ptr(idx), F.cint(nd + 1), F.i64(m), F.ptr(None), F.cint(c),
                                   F.cint(self.batch_size), F.cint(d), F.cint(h), F.cint(w), F.ptr(out), F.ptr(ws), F.csize(ws.numel()),
                                   F.stream_ptr(feats.device))
        F.check(rc, "lvq_sparse_to_dense")
        return out.view(self.batch_size, c, d, h, w) if nd == 3 else out


@torch.no_grad()
def bev_out(x_conv: SparseTensor) -> SparseTensor:
    """VoxelResBackBone8xVoxelNeXt.bev_out: merge the z axis of a sparse 3-D tensor into a sparse 2-D one."""
    feats = x_conv.features.contiguous()
    idx = x_conv.indices.to(torch.int32).contiguous()
    F.require_cuda(feats, idx)
    m, c = feats.shape
    ny, nx = x_conv.spatial_shape[1], x_conv.spatial_shape[2]
    dev = feats.device
    L = F.lib()
    nbytes = L.lvq_sparse_bev_merge_workspace_bytes(F.i64(m), F.cint(x_conv.batch_size), F.cint(ny), F.cint(nx))
    if nbytes == 0:
        raise F.LvqError("lvq_sparse_bev_merge: batch * ny * nx must stay below 2^31")
    ws = workspace(nbytes, dev, "bevmerge")
    cap = max(1, min(m, x_conv.batch_size * ny * nx))
    out_idx = torch.empty((cap, 3), dtype=torch.int32, device=dev)
    out_feats = torch.empty((cap, c), dtype=torch.float32, device=dev)
    inv = torch.empty((max(m, 1),), dtype=torch.int32, device=dev)
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    rc = L.lvq_sparse_bev_merge(F.ptr(idx), F.ptr(feats), F.i64(m), F.cint(c), F.cint(x_conv.batch_size), F.cint(ny), F.cint(nx),
                                F.ptr(out_idx), F.ptr(out_feats), F.ptr(inv), F.ptr(counts), F.ptr(ws), F.csize(ws.numel()),
                                F.stream_ptr(dev))
    F.check(rc, "lvq_sparse_bev_merge")
    m2 = int(counts[0].item())            # the reference's torch.unique synchronises here as well
    return SparseTensor(out_feats[:m2], out_idx[:m2], [ny, nx], x_conv.batch_size)


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES

    @torch.no_grad()
    def forward(self, batch_dict):
        encoded_spconv_tensor = batch_dict["encoded_spconv_tensor"]
        spatial_features = encoded_spconv_tensor.dense()
        if spatial_features.dim() == 5:
            N, C, D, H, W = spatial_features.shape
            spatial_features = spatial_features.view(N, C * D, H, W)
        batch_dict["spatial_features"] = spatial_features
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict
