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
LIB_PATH = os.environ.get("LVQ_LIB_PATH") or os.path.join(_HERE, "liblvq_hip.so")     # LVQ_LIB_PATH: another build of the same ABI (A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lvq.h")

LVQ_OK = 0
LVQ_EUNSUPPORTED = -5      # include/lvq.h: shape outside what a kernel family implements (callers may pick another route)


class LvqError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None


class Tuning(ctypes.Structure):
    """include/lvq.h: lvq_tuning -- the kernel-family choices the library takes from its caller (it never reads the environment)."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "attn_no32", "attn32_nw", "attn_pipe", "attn_nsplit", "attn_qt", "attn_nw", "gemm_no_gemv", "gemm_stream_c_mb", "gemm_no256",
        "gemm_no256x256", "gemm_256x256_min_tiles", "gemm_ln_tiles", "pillar_vfe_generic", "voxel_path", "pairs_one_wg",
        "ca_fused_variant")] + [("ca_fused_stamps", ctypes.c_uint64), ("reserved", ctypes.c_int32 * 8)]


TUNING_FIELDS = tuple(n for n, _ in Tuning._fields_ if n != "reserved")


def tuning_from_env(env=None) -> dict:
    """The LVQ_* variables of INTEGRATION.md as an lvq_tuning record.  They are a convenience of THIS mirror, read once when the
    library is loaded (and again by set_tuning()); a C / C++ caller fills the struct itself."""
    e = os.environ if env is None else env

    def flag(name):
        return 1 if e.get(name) else 0

    def num(name):
        return int(e[name]) if e.get(name) else 0

    return {
        "attn_no32": flag("LVQ_ATTN_NO32"), "attn32_nw": num("LVQ_ATTN32_NW"),
        "attn_pipe": -1 if e.get("LVQ_ATTN_NO_PIPE") else flag("LVQ_ATTN_PIPE"),
        "attn_nsplit": num("LVQ_ATTN_NSPLIT"), "attn_qt": num("LVQ_ATTN_QT"), "attn_nw": num("LVQ_ATTN_NW"),
        "gemm_no_gemv": flag("LVQ_GEMM_NO_GEMV"),
        "gemm_stream_c_mb": -1 if e.get("LVQ_GEMM_NO_STREAM_C") else num("LVQ_GEMM_STREAM_C_MB"),
        "gemm_no256": flag("LVQ_GEMM_NO256"), "gemm_no256x256": flag("LVQ_GEMM_NO256X256"),
        "gemm_256x256_min_tiles": num("LVQ_GEMM_256X256_MIN_TILES"), "gemm_ln_tiles": flag("LVQ_GEMM_LN_TILES"),
        "pillar_vfe_generic": flag("LVQ_PILLAR_VFE_GENERIC"),
        "voxel_path": 2 if e.get("LVQ_VOXEL_LEGACY") else flag("LVQ_VOXEL_BINNED"),
        "pairs_one_wg": flag("LVQ_PAIRS_ONE_WG"), "ca_fused_variant": num("LVQ_CA_DBG"), "ca_fused_stamps": 0,
    }


def set_tuning(**fields) -> dict:
    """lvq_set_tuning: the record implied by the environment, overridden by `fields` (names of include/lvq.h: lvq_tuning).
    set_tuning() with no arguments restores the environment's choices.  Returns the record now in force."""
    rec = tuning_from_env()
    for k, v in fields.items():
        if k not in TUNING_FIELDS:
            raise LvqError(f"lvq_tuning has no field {k!r}")
        rec[k] = int(v)
    t = Tuning()
    for k, v in rec.items():
        setattr(t, k, v)
    check(lib().lvq_set_tuning(ctypes.byref(t)), "lvq_set_tuning")
    return rec


def get_tuning() -> dict:
    t = Tuning()
    check(lib().lvq_get_tuning(ctypes.byref(t)), "lvq_get_tuning")
    return {k: int(getattr(t, k)) for k in TUNING_FIELDS}


class tuning:
    """`with tuning(attn_nsplit=1): ...` -- a kernel-family choice for the calls inside (process-wide: not for concurrent launch threads)."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.old = get_tuning()
        cur = dict(self.old)
        cur.update({k: int(v) for k, v in self.fields.items()})
        _push(cur)
        return self

    def __exit__(self, *exc):
        _push(self.old)
        return False


def _push(rec: dict):
    t = Tuning()
    for k, v in rec.items():
        if k not in TUNING_FIELDS:
            raise LvqError(f"lvq_tuning has no field {k!r}")
        setattr(t, k, v)
    check(lib().lvq_set_tuning(ctypes.byref(t)), "lvq_set_tuning")


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
        L.lvq_tuning_defaults.restype = None
        _lib = L
        _push(tuning_from_env())               # the library itself never reads the environment (include/lvq.h: lvq_tuning)
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
