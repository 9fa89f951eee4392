"""CPU restatement (numpy) of the data-format rows either side of the sparse backbone -- TEST INFRASTRUCTURE ONLY
(imported by tests/, never by the product path).

  load_bev        training/data/dataset.py:139-146   np.load(path) -> float32
  bev_out         spconv_backbone_voxelnext.py:149-164  unique over (b, y, x) rows + index_add_
  dense           spconv SparseConvTensor.dense() as HeightCompression uses it (height_compression.py:18-21)

Pinned by tests/test_bev_bridge.py against the reference's own three torch lines (torch.unique(dim=0, return_inverse) +
new_zeros + index_add_) run on the CPU, and against torch's fp16 -> fp32 cast.
"""
import numpy as np


def load_bev(path) -> np.ndarray:
    return np.load(path).astype(np.float32)


def bev_out(features: np.ndarray, indices_bzyx: np.ndarray):
    """-> (features_unique [M2,C] fp32, indices_unique [M2,3] (b,y,x), inverse [M])."""
    cat = indices_bzyx[:, [0, 2, 3]]
    uniq, inv = np.unique(cat, axis=0, return_inverse=True)      # rows in ascending lexicographic order, like torch.unique(dim=0)
    inv = np.asarray(inv).reshape(-1)
    out = np.zeros((uniq.shape[0], features.shape[1]), dtype=np.float32)
    np.add.at(out, inv, features.astype(np.float32))             # sequential fp32 accumulation in row order (CPU index_add_)
    return out, uniq.astype(np.int32), inv.astype(np.int64)


def dense(features: np.ndarray, indices: np.ndarray, spatial_shape, batch_size: int) -> np.ndarray:
    c = features.shape[1]
    out = np.zeros((batch_size, c, *spatial_shape), dtype=np.float32)
    if len(spatial_shape) == 3:
        out[indices[:, 0], :, indices[:, 1], indices[:, 2], indices[:, 3]] = features
    else:
        out[indices[:, 0], :, indices[:, 1], indices[:, 2]] = features
    return out


def height_compression(features, indices, spatial_shape, batch_size):
    d = dense(features, indices, spatial_shape, batch_size)
    if d.ndim == 5:
        n, c, dd, h, w = d.shape
        d = d.reshape(n, c * dd, h, w)
    return d
