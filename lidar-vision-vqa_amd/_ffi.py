"""ctypes binding of the C-ABI library liblvq_hip.so (include/lvq.h).

PyTorch is plumbing here: it owns device memory (`tensor.data_ptr()`), the current HIP stream and
`torch.distributed`; every byte of arithmetic on the hot path happens inside the library.  There is
NO fallback: if the library is missing or a call returns an error code, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblvq_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lvq.h")

LVQ_OK = 0
LVQ_EUNSUPPORTED = -5      # include/lvq.h: shape outside what a kernel family implements (callers may pick another route)


class LvqError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None


def declared_symbols() -> List[str]:
    """Every function name declared in include/lvq.h (used by the export test and INTEGRATION.md)."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lvq_[a-z0-9_]+)\s*\(", txt)))


def lib() -> ctypes.CDLL:
    """Load liblvq_hip.so (built by __graft_entry__.build() / `make -C lidar-vision-vqa_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LvqError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        L.lvq_version.restype = ctypes.c_char_p
        L.lvq_strerror.restype = ctypes.c_char_p
        L.lvq_strerror.argtypes = [ctypes.c_int]
        for name in ("lvq_voxelize_hard_workspace_bytes", "lvq_voxelize_dynamic_workspace_bytes", "lvq_pillar_dwconv_workspace_bytes",
                     "lvq_sparse_bev_merge_workspace_bytes", "lvq_sparse_to_dense_workspace_bytes", "lvq_qwen2_decode_workspace_bytes",
                     "lvq_bev_tiles_workspace_bytes", "lvq_attention_workspace_bytes", "lvq_colsum_workspace_bytes",
                     "lvq_attention_stream_totals_workspace_bytes", "lvq_attention_tiled_signed_workspace_bytes",
                     "lvq_bev_tile_kv_workspace_bytes", "lvq_ca_fused_packed_bytes", "lvq_ca_fused_workspace_bytes"):
            getattr(L, name).restype = ctypes.c_size_t
        _lib = L
    return _lib


def check(rc: int, what: str):
    if rc != LVQ_OK:
        raise LvqError(f"{what} failed: {lib().lvq_strerror(rc).decode()} ({rc})")


def stream_ptr(device=None) -> ctypes.c_void_p:
    """Current torch HIP stream as the ABI's lvq_stream_t."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def require_cuda(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise LvqError("lidar_vision_vqa_amd ops run on an MI355X only: got a CPU tensor "
                           "(there is deliberately no CPU fallback; the CPU checker lives in oracle/)")
        if t is not None and not t.is_contiguous():
            raise LvqError("tensor must be contiguous")


def f32x(vals: Sequence[float]):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def i32x(vals: Sequence[int]):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def ptr_array(tensors: Sequence[torch.Tensor]):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


i64 = ctypes.c_int64
cint = ctypes.c_int
cfloat = ctypes.c_float
csize = ctypes.c_size_t
