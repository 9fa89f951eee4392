"""Thin Python wrappers over the fusion-side C ABI (include/lvq.h): argument marshalling only.

A "BF" value is a pair (hi, lo) of torch.bfloat16 tensors; lo is None for a plain-bf16 operand and the
residual x - hi for a split one (see include/lvq.h "Precision modes").  Modes:
  bf16    every operand plain (2^-9 operand rounding; fastest, ~3.5e-2 from the fp32 CPU reference on the bench workload)
  bf16x3  every operand hi + lo, every product hi*hi + hi*lo + lo*hi (meets the 1e-3 bar everywhere; 2.7x slower)
  mixed   bf16x3 everywhere EXCEPT the tensors whose rounding is independent per key of a long K/V stream and therefore
          averages out in the softmax-weighted sum (BEV tokens x, K, V, P of VATLiDAR's cross-attention: 262 144 keys): those
          stay plain.  Operands whose rounding is COMMON to all keys stay split: weights (W_proj, W_k|W_v), the conv tokens
          (93 % of the cells hold the same constant) and the query side.  tools/precision_study.py measures each group.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _ffi as F

BF = Tuple[torch.Tensor, Optional[torch.Tensor]]

PRECISIONS = ("bf16", "bf16x3", "mixed", "mixed16")
_default_precision = os.environ.get("LVQ_PRECISION", "bf16x3")
assert _default_precision in PRECISIONS


def default_precision() -> str:
    return _default_precision


def set_default_precision(p: str):
    global _default_precision
    if p not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}")
    _default_precision = p


# Optional per-launch timing with HIP events ON THE LAUNCH STREAM (bench.py's roofline leg): when EVENTS is a
# dict, every tagged launch appends (start, end) torch.cuda.Event pairs under its tag.
EVENTS = None


class _Region:
    def __init__(self, tag, device):
        self.tag, self.device = tag, device

    def __enter__(self):
        if EVENTS is not None and self.tag:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        if EVENTS is not None and self.tag:
            self.e.record(torch.cuda.current_stream(self.device))
            EVENTS.setdefault(self.tag, []).append((self.s, self.e))
        return False


def region(tag, device):
    return _Region(tag, device)


def _bf_empty(shape, dev, split: bool) -> BF:
    hi = torch.empty(shape, dtype=torch.bfloat16, device=dev)
    return hi, (torch.empty(shape, dtype=torch.bfloat16, device=dev) if split else None)


def cast(x: torch.Tensor, split: bool) -> BF:
    F.require_cuda(x)
    assert x.dtype == torch.float32
    hi, lo = _bf_empty(x.shape, x.device, split)
    F.check(F.lib().lvq_cast_bf16(F.ptr(x), F.i64(x.numel()), F.ptr(hi), F.ptr(lo), F.stream_ptr(x.device)), "lvq_cast_bf16")
    return hi, lo


def to_f32(x: BF, alpha: float = 1.0) -> torch.Tensor:
    hi, lo = x
    out = torch.empty(hi.shape, dtype=torch.float32, device=hi.device)
    F.check(F.lib().lvq_bf16_to_f32(F.ptr(hi), F.ptr(lo), F.i64(hi.numel()), F.cfloat(alpha), F.ptr(out),
                                    F.stream_ptr(hi.device)), "lvq_bf16_to_f32")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float, split: bool,
              want_f32: bool = False, want_bf: bool = True, add: Optional[torch.Tensor] = None, add_group: int = 1,
              post: Optional[torch.Tensor] = None):
    """x [rows, d] fp32 -> (y_f32 | None, BF | None)."""
    F.require_cuda(x, gamma, beta, add, post)
    rows, d = x.shape
    y32 = torch.empty_like(x) if want_f32 else None
    hi, lo = _bf_empty(x.shape, x.device, split) if want_bf else (None, None)
    rc = F.lib().lvq_layernorm(F.ptr(x), F.ptr(add), F.cint(add.shape[0] if add is not None else 0), F.cint(add_group),
                               F.ptr(gamma), F.ptr(beta), F.cfloat(eps), F.i64(rows), F.cint(d), F.ptr(post),
                               F.i64(post.shape[0] if post is not None else 0), F.ptr(y32), F.ptr(hi), F.ptr(lo),
                               F.stream_ptr(x.device))
    F.check(rc, "lvq_layernorm")
    return y32, ((hi, lo) if want_bf else None)


def rmsnorm(x: torch.Tensor, gamma: torch.Tensor, eps: float, split: bool) -> BF:
    rows, d = x.shape
    hi, lo = _bf_empty(x.shape, x.device, split)
    rc = F.lib().lvq_rmsnorm(F.ptr(x), F.ptr(gamma), F.cfloat(eps), F.i64(rows), F.cint(d), F.ptr(None), F.ptr(hi), F.ptr(lo),
                             F.stream_ptr(x.device))
    F.check(rc, "lvq_rmsnorm")
    return hi, lo


def linear(a: BF, w: BF, bias: Optional[torch.Tensor] = None, *, gelu: bool = False, alpha: float = 1.0,
           residual: Optional[torch.Tensor] = None, rowtab: Optional[torch.Tensor] = None, out_f32: bool = False,
           out_bf: bool = False, w_rows: Optional[Tuple[int, int]] = None, tag: Optional[str] = None):
    """y = a @ w[r0:r1].T (+bias[r0:r1]) ... ; a hi [M,K], w hi [N,K].  Returns (y_f32 | None, BF | None)."""
    ah, al = a
    wh, wl = w
    m, k = ah.shape
    r0, r1 = w_rows if w_rows is not None else (0, wh.shape[0])
    n = r1 - r0
    assert wh.shape[1] == k and (al is None or wl is not None)      # plain | bf16x3 | a plain, w split ("x2w")
    dev = ah.device
    split = wl is not None
    c32 = torch.empty((m, n), dtype=torch.float32, device=dev) if out_f32 else None
    ch, cl = _bf_empty((m, n), dev, al is not None) if out_bf else (None, None)
    wo = r0 * k * 2  # byte offset of the weight row slice
    bo = r0 * 4
    import ctypes
    with region(tag, dev):
        rc = F.lib().lvq_gemm_bf16(
            F.ptr(ah), F.ptr(al), ctypes.c_void_p(wh.data_ptr() + wo), ctypes.c_void_p(wl.data_ptr() + wo if split else 0),
            ctypes.c_void_p(bias.data_ptr() + bo if bias is not None else 0), F.ptr(residual), F.ptr(rowtab),
            F.i64(rowtab.shape[0] if rowtab is not None else 0), F.cfloat(alpha), F.cint(1 if gelu else 0), F.i64(m), F.cint(n),
            F.cint(k), F.i64(k), F.i64(k), F.i64(n), F.cint(1), F.i64(0), F.i64(0), F.i64(0), F.ptr(c32), F.ptr(ch), F.ptr(cl),
            F.stream_ptr(dev))
    F.check(rc, f"lvq_gemm_bf16 (m={m}, n={n}, k={k})")
    return c32, ((ch, cl) if out_bf else None)


_ATT_WS = {}


def attention(q: BF, k: BF, v: BF, *, batch: int, n_heads: int, n_kv_heads: int, nq: int, nkv: int, dh: int,
              q_strides, k_strides, v_strides, scale: float, bias: Optional[torch.Tensor] = None, causal: bool = False,
              tag: Optional[str] = None) -> BF:
    """q/k/v: BF views (possibly column slices of packed projections); *_strides = (batch, row, head) in elements.
    Returns BF [batch*nq, n_heads*dh]."""
    import ctypes
    qh, ql = q
    dev = qh.device
    split = ql is not None                       # q split, k / v plain = the "mixed" stream form (lvq_attention_stream_ok shapes)
    oh, ol = _bf_empty((batch * nq, n_heads * dh), dev, split)
    L = F.lib()
    L.lvq_attention_workspace_bytes.restype = ctypes.c_size_t
    nbytes = L.lvq_attention_workspace_bytes(F.cint(batch), F.cint(n_heads), F.cint(nq), F.cint(nkv), F.cint(dh),
                                             F.cint(3 if k[1] is not None else 1))
    key = (dev.index, "attn")
    ws = _ATT_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ATT_WS[key] = ws
    d = n_heads * dh
    with region(tag, dev):
        rc = L.lvq_attention_bf16(
            F.ptr(qh), F.ptr(ql), F.ptr(k[0]), F.ptr(k[1]), F.ptr(v[0]), F.ptr(v[1]), F.ptr(bias), F.cint(batch), F.cint(n_heads),
            F.cint(n_kv_heads), F.cint(nq), F.cint(nkv), F.cint(dh),
            F.i64(q_strides[0]), F.i64(q_strides[1]), F.i64(q_strides[2]),
            F.i64(k_strides[0]), F.i64(k_strides[1]), F.i64(k_strides[2]),
            F.i64(v_strides[0]), F.i64(v_strides[1]), F.i64(v_strides[2]),
            F.i64(nq * d), F.i64(d), F.i64(dh), F.cfloat(scale), F.cint(1 if causal else 0), F.ptr(oh), F.ptr(ol),
            F.ptr(ws), F.csize(ws.numel()), F.stream_ptr(dev))
    F.check(rc, f"lvq_attention_bf16 (B={batch}, H={n_heads}, nq={nq}, nkv={nkv}, dh={dh})")
    return oh, ol


def ca_fused_ok(batch: int, nq: int, nkv: int, d: int, n_heads: int) -> bool:
    """Shapes of the fused short-K/V cross-attention kernel (include/lvq.h: lvq_ca_fused_ok)."""
    return bool(F.lib().lvq_ca_fused_ok(F.cint(batch), F.cint(nq), F.cint(nkv), F.cint(d), F.cint(n_heads)))


def ca_fused_pack(ln_w: torch.Tensor, ln_b: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor, out_w: torch.Tensor, out_b: torch.Tensor,
                  n_heads: int, f16: bool) -> torch.Tensor:
    """ca_ln + ca weights -> the fragment-major 16-bit blob of lvq_ca_fused (once per weights version)."""
    F.require_cuda(ln_w, ln_b, in_w, in_b, out_w, out_b)
    d = out_w.shape[0]
    L = F.lib()
    nbytes = int(L.lvq_ca_fused_packed_bytes(F.cint(d), F.cint(n_heads)))
    if nbytes == 0:
        raise F.LvqError(f"lvq_ca_fused: d={d}, heads={n_heads} is not a shape of the fused kernel")
    blob = torch.empty(nbytes, dtype=torch.uint8, device=out_w.device)
    rc = L.lvq_ca_fused_pack(F.ptr(ln_w), F.ptr(ln_b), F.ptr(in_w), F.ptr(in_b), F.ptr(out_w), F.ptr(out_b), F.cint(d), F.cint(n_heads),
                             F.cint(1 if f16 else 0), F.ptr(blob), F.csize(nbytes), F.stream_ptr(out_w.device))
    F.check(rc, "lvq_ca_fused_pack")
    return blob


def ca_fused(q: torch.Tensor, kv: torch.Tensor, blob: torch.Tensor, eps: float, n_heads: int, f16: bool, tag: Optional[str] = None) -> torch.Tensor:
    """out = q + ca(ca_ln(q), kv, kv) for q [B, nq, d], kv [B, nkv, d] fp32 (vat_blocks.py:41-42) in two launches."""
    F.require_cuda(q, kv, blob)
    B, nq, d = q.shape
    nkv = kv.shape[1]
    dev = q.device
    L = F.lib()
    nbytes = int(L.lvq_ca_fused_workspace_bytes(F.cint(B), F.cint(nq), F.cint(nkv), F.cint(d), F.cint(n_heads)))
    if nbytes == 0:
        raise F.LvqError(f"lvq_ca_fused: (B={B}, nq={nq}, nkv={nkv}, d={d}, heads={n_heads}) is not a shape of the fused kernel")
    key = (dev.index, "ca_fused")
    ws = _ATT_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ATT_WS[key] = ws
    out = torch.empty_like(q)
    with region(tag, dev):
        rc = L.lvq_ca_fused(F.ptr(q), F.ptr(kv), F.ptr(blob), F.cfloat(eps), F.cint(B), F.cint(nq), F.cint(nkv), F.cint(d), F.cint(n_heads),
                            F.cint(1 if f16 else 0), F.ptr(out), F.ptr(ws), F.csize(ws.numel()), F.stream_ptr(dev))
    F.check(rc, f"lvq_ca_fused (B={B}, nq={nq}, nkv={nkv})")
    return out


def attention_stream_ok(nq: int, nkv: int, dh: int) -> bool:
    """True when lvq_attention_bf16 takes (q split, k / v plain) for this shape: the long-stream kernel's shapes."""
    return bool(F.lib().lvq_attention_stream_ok(F.cint(nq), F.cint(nkv), F.cint(dh)))


def bev_occupied_cells(bev: torch.Tensor, cap: Optional[int] = None):
    """Dense canvas [B, C, H, W] fp32 -> its occupied cells as pillars: feats [cap, C] fp32, coords [cap, 4] int32 (b, 0, y, x), n [1] int32 on the
    device (rows >= n are not written; cap defaults to every cell).  lvq_bev_occupied_cells: the inverse of PointPillarScatter."""
    F.require_cuda(bev)
    B, C, H, W = bev.shape
    cap = B * H * W if cap is None else int(cap)
    feats = torch.empty((cap, C), dtype=torch.float32, device=bev.device)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=bev.device)
    n = torch.empty(1, dtype=torch.int32, device=bev.device)
    rc = F.lib().lvq_bev_occupied_cells(F.ptr(bev), F.cint(B), F.cint(C), F.cint(H), F.cint(W), F.i64(cap), F.ptr(feats), F.ptr(coords), F.ptr(n),
                                        F.stream_ptr(bev.device))
    F.check(rc, "lvq_bev_occupied_cells")
    return feats, coords, n


def dwconv3x3_gelu(bev: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], split: bool) -> BF:
    """bev [B,C,H,W] fp32 -> tokens BF [B*H*W, C]."""
    F.require_cuda(bev, w, b)
    B, C, H, W = bev.shape
    hi, lo = _bf_empty((B * H * W, C), bev.device, split)
    rc = F.lib().lvq_dwconv3x3_gelu(F.ptr(bev), F.ptr(w), F.ptr(b), F.cint(B), F.cint(C), F.cint(H), F.cint(W), F.ptr(hi),
                                    F.ptr(lo), F.stream_ptr(bev.device))
    F.check(rc, "lvq_dwconv3x3_gelu")
    return hi, lo


_PD_WS = {}


def pillar_dwconv3x3_gelu(feat: torch.Tensor, coords: torch.Tensor, n_live: Optional[torch.Tensor], batch: int, ny: int, nx: int,
                          w: torch.Tensor, b: Optional[torch.Tensor], split: bool) -> BF:
    """Sparse BEV bridge: PointPillarScatter + depthwise 3x3 + GELU -> tokens BF [batch*ny*nx, C] without the dense canvas
    (bit-identical to pillar_scatter + dwconv3x3_gelu).  feat [M, C] fp32, coords [M, 4] int32 (b, z, y, x), rows >= n_live[0] skipped."""
    F.require_cuda(feat, coords, w, b, n_live)
    m, C = feat.shape
    hi, lo = _bf_empty((batch * ny * nx, C), feat.device, split)
    L = F.lib()
    nbytes = int(L.lvq_pillar_dwconv_workspace_bytes(F.cint(batch), F.cint(ny), F.cint(nx)))
    key = feat.device.index if feat.device.index is not None else torch.cuda.current_device()
    ws = _PD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=feat.device)
        _PD_WS[key] = ws
    import ctypes
    rc = L.lvq_pillar_dwconv3x3_gelu(F.ptr(feat), F.ptr(coords), F.i64(m), F.ptr(n_live), F.cint(C), F.cint(batch), F.cint(ny), F.cint(nx),
                                     F.ptr(w), F.ptr(b), F.ptr(hi), F.ptr(lo), F.ptr(ws), ctypes.c_size_t(ws.numel()), F.stream_ptr(feat.device))
    F.check(rc, "lvq_pillar_dwconv3x3_gelu")
    return hi, lo


# ---- sparse (tiled) BEV key stream: include/lvq.h "sparse BEV key stream of VATLiDAR" ----
def pillar_index_map(coords: torch.Tensor, n_live: Optional[torch.Tensor], batch: int, ny: int, nx: int) -> torch.Tensor:
    """[batch, ny, nx] int32: pillar row or -1 (PointPillarScatter as an index map)."""
    F.require_cuda(coords, n_live)
    idx = torch.empty((batch, ny, nx), dtype=torch.int32, device=coords.device)
    rc = F.lib().lvq_pillar_index_map(F.ptr(coords), F.i64(coords.shape[0]), F.ptr(n_live), F.cint(batch), F.cint(ny), F.cint(nx), F.ptr(idx),
                                      F.stream_ptr(coords.device))
    F.check(rc, "lvq_pillar_index_map")
    return idx


_TILE_WS = {}


def bev_tiles(idx: Optional[torch.Tensor], batch: int, ny: int, nx: int, device, row_base: int, force_all: bool = False):
    """Piece / row bookkeeping of the tiled key stream (8 x 8-cell tiles of eight 2 x 4-cell pieces; a cell is dirty when its 3 x 3
    neighbourhood holds a pillar) -> (live_list [batch*nt*8] i32, piece_dirty [batch*nt*8, 2] i32, row_src [batch, ny*nx] i32 absolute rows
    of the K|V buffer (dirty: row_base + number, clean: table row = key index), counts [3] i32 = live pieces, their rows, dirty rows)."""
    nt = (ny // 8) * (nx // 8)
    live = torch.empty((batch * nt * 8,), dtype=torch.int32, device=device)
    dirty = torch.empty((batch * nt * 8, 2), dtype=torch.int32, device=device)
    src = torch.empty((batch, ny * nx), dtype=torch.int32, device=device)
    counts = torch.empty((3,), dtype=torch.int32, device=device)
    L = F.lib()
    nbytes = int(L.lvq_bev_tiles_workspace_bytes(F.cint(batch), F.cint(ny), F.cint(nx)))
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    ws = _TILE_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _TILE_WS[key] = ws
    rc = L.lvq_bev_tiles(F.ptr(idx), F.cint(batch), F.cint(ny), F.cint(nx), F.cint(1 if force_all else 0), F.cint(row_base), F.ptr(live), F.ptr(dirty),
                         F.ptr(src), F.ptr(counts), F.ptr(ws), F.csize(ws.numel()), F.stream_ptr(device))
    F.check(rc, "lvq_bev_tiles")
    return live, dirty, src, counts


def bev_tile_tokens(feat: torch.Tensor, idx: torch.Tensor, live: torch.Tensor, dirty: torch.Tensor, counts: torch.Tensor, cap_rows: int, batch: int, ny: int, nx: int,
                    w9: torch.Tensor, b9: Optional[torch.Tensor], w: BF, bias, gamma, beta, eps: float, pe_tiled: torch.Tensor,
                    out_lo: bool, tag: Optional[str] = None) -> BF:
    """Fused refine conv -> proj -> LayerNorm -> + positional table over the live pieces; the DIRTY cells' rows are stored compactly
    -> x BF [cap_rows, n] (rows 0 .. counts[2]-1 written)."""
    F.require_cuda(feat, idx, live, dirty, counts, w9, b9, pe_tiled)
    n = w[0].shape[0]
    xh, xl = _bf_empty((cap_rows, n), feat.device, out_lo)
    with region(tag, feat.device):
        rc = F.lib().lvq_bev_tile_tokens(F.ptr(feat), F.ptr(idx), F.ptr(live), F.ptr(dirty), F.ptr(counts), F.i64(batch * (ny // 8) * (nx // 8)), F.cint(batch),
                                         F.cint(ny), F.cint(nx), F.cint(feat.shape[1]), F.ptr(w9), F.ptr(b9), F.ptr(w[0]), F.ptr(w[1]), F.ptr(bias),
                                         F.ptr(gamma), F.ptr(beta), F.cfloat(eps), F.ptr(pe_tiled), F.cint(n), F.ptr(xh), F.ptr(xl),
                                         F.stream_ptr(feat.device))
    F.check(rc, "lvq_bev_tile_tokens")
    return xh, xl


def bev_tile_kv(feat: torch.Tensor, idx: torch.Tensor, live: torch.Tensor, dirty: torch.Tensor, counts: torch.Tensor, cap_rows: int, batch: int, ny: int,
                nx: int, w9: torch.Tensor, b9: Optional[torch.Tensor], m: BF, m0: torch.Tensor, r: BF, r0: torch.Tensor, c0: float, d_ln: int, eps: float,
                t_tiled: torch.Tensor, out: torch.Tensor, tag: Optional[str] = None, split_launch: bool = True, k_fp16: bool = False) -> torch.Tensor:
    """K|V rows of the dirty cells straight from the pillar features (lvq_bev_tile_kv: LayerNorm and the K|V projection folded onto the
    64-channel conv token) -> `out` [>= cap_rows, 2n] plain bf16, rows 0 .. counts[2]-1 written."""
    F.require_cuda(feat, idx, live, dirty, counts, w9, b9, m0, r0, t_tiled, out)
    n2 = m[0].shape[0]
    if out.dtype != torch.bfloat16 or not out.is_contiguous() or out.shape[0] < cap_rows or out.shape[1] != n2 or t_tiled.shape[1] != n2:
        raise F.LvqError("bev_tile_kv: `out` must be a contiguous bf16 [>= cap_rows, 2n] buffer and the table [HW, 2n]")
    t_f16 = t_tiled.dtype == torch.float16                       # fp16 table: two-launch form only
    if t_tiled.dtype not in (torch.float16, torch.float32) or (t_f16 and not split_launch):
        raise F.LvqError("bev_tile_kv: the table is fp32, or fp16 with the two-launch form")
    L = F.lib()
    cap_tiles = batch * (ny // 8) * (nx // 8)
    ws = None
    if split_launch:
        nbytes = int(L.lvq_bev_tile_kv_workspace_bytes(F.i64(cap_tiles)))
        key = (feat.device.index, "tile_kv", torch.cuda.current_stream(feat.device).cuda_stream)
        ws = _TILE_WS.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=feat.device)
            _TILE_WS[key] = ws
    with region(tag, feat.device):
        rc = L.lvq_bev_tile_kv(F.ptr(feat), F.ptr(idx), F.ptr(live), F.ptr(dirty), F.ptr(counts), F.i64(cap_tiles), F.cint(batch),
                               F.cint(ny), F.cint(nx), F.cint(feat.shape[1]), F.ptr(w9), F.ptr(b9), F.ptr(m[0]), F.ptr(m[1]), F.ptr(m0), F.ptr(r[0]),
                               F.ptr(r[1]), F.ptr(r0), F.cfloat(c0), F.cint(d_ln), F.cfloat(eps), F.ptr(t_tiled), F.cint(1 if t_f16 else 0), F.cint(n2 // 2),
                               F.cint(1 if k_fp16 else 0), F.ptr(out),
                               F.ptr(ws), F.csize(ws.numel() if ws is not None else 0), F.stream_ptr(feat.device))
    F.check(rc, "lvq_bev_tile_kv")
    return out


def linear_live_rows(a: BF, w: BF, bias: Optional[torch.Tensor], rows_dev: torch.Tensor, w_rows, tag: Optional[str] = None,
                     out: Optional[torch.Tensor] = None) -> BF:
    """a [cap, k] @ w[r0:r1].T + bias over the first *rows_dev rows (device-side count) -> BF [cap, n] (hi only unless a is split).
    `out` (plain bf16 [>= cap, n], contiguous): write there instead of a fresh buffer."""
    import ctypes
    ah, al = a
    wh, wl = w
    cap, k = ah.shape
    r0, r1 = w_rows
    n = r1 - r0
    if out is not None:
        if al is not None or out.dtype != torch.bfloat16 or not out.is_contiguous() or out.shape[0] < cap or out.shape[1] != n:
            raise F.LvqError("linear_live_rows: `out` must be a contiguous plain bf16 [>= cap, n] buffer")
        ch, cl = out, None
    else:
        ch, cl = _bf_empty((cap, n), ah.device, al is not None)
    wo, bo = r0 * k * 2, r0 * 4
    with region(tag, ah.device):
        rc = F.lib().lvq_gemm_bf16_live_rows(F.ptr(ah), F.ptr(al), ctypes.c_void_p(wh.data_ptr() + wo),
                                             ctypes.c_void_p(wl.data_ptr() + wo if wl is not None else 0),
                                             ctypes.c_void_p(bias.data_ptr() + bo if bias is not None else 0), F.i64(cap), F.ptr(rows_dev), F.cint(n),
                                             F.cint(k), F.i64(k), F.i64(k), F.i64(n), F.ptr(ch), F.ptr(cl), F.stream_ptr(ah.device))
    F.check(rc, f"lvq_gemm_bf16_live_rows (cap={cap}, n={n}, k={k})")
    return ch, cl


def attention_tiled(q: BF, kv: torch.Tensor, row_src: torch.Tensor, *, batch: int, n_heads: int, nq: int, n_tiles: int, dh: int, scale: float,
                    tag: Optional[str] = None, k_fp16: bool = False) -> BF:
    """q BF [batch*nq, d]; kv [rows, 2d] plain bf16 (K | V packed): the per-model table rows and every batch's computed rows in ONE buffer;
    row_src [batch, n_tiles*64] i32 = the row of every key slot -> BF [batch*nq, d]."""
    qh, ql = q
    dev = qh.device
    d = n_heads * dh
    oh, ol = _bf_empty((batch * nq, d), dev, ql is not None)
    L = F.lib()
    nbytes = int(L.lvq_attention_workspace_bytes(F.cint(batch), F.cint(n_heads), F.cint(nq), F.cint(n_tiles * 64), F.cint(dh), F.cint(1)))
    ws = _att_ws(dev, nbytes)
    with region(tag, dev):
        rc = L.lvq_attention_bf16_tiled(F.ptr(qh), F.ptr(ql), F.ptr(kv), F.ptr(kv[:, d:]), F.ptr(row_src), F.cint(batch), F.cint(n_heads), F.cint(nq),
                                        F.cint(n_tiles), F.cint(dh), F.i64(nq * d), F.i64(d), F.i64(dh), F.i64(2 * d), F.i64(dh), F.i64(nq * d), F.i64(d),
                                        F.i64(dh), F.cfloat(scale), F.cint(1 if k_fp16 else 0), F.ptr(oh), F.ptr(ol), F.ptr(ws), F.csize(ws.numel()),
                                        F.stream_ptr(dev))
    F.check(rc, f"lvq_attention_bf16_tiled (B={batch}, H={n_heads}, nq={nq}, tiles={n_tiles})")
    return oh, ol


def _att_ws(dev, nbytes: int) -> torch.Tensor:
    key = (dev.index, "attn")
    ws = _ATT_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ATT_WS[key] = ws
    return ws


def attention_stream_totals(q: BF, kv: torch.Tensor, *, n_heads: int, nq: int, nkv: int, dh: int, scale: float, k_fp16: int = 0) -> torch.Tensor:
    """Unnormalised softmax sums (O | m | l) of ONE batch of queries q BF [nq, d] over the dense key stream kv [nkv, 2d] (K | V packed,
    plain bf16) -> fp32 [n_heads, nq, dh + 2]: the per-model totals of attention_tiled_signed."""
    qh, ql = q
    dev = qh.device
    d = n_heads * dh
    L = F.lib()
    nbytes = int(L.lvq_attention_stream_totals_workspace_bytes(F.cint(n_heads), F.cint(nq), F.cint(nkv), F.cint(dh)))
    if nbytes == 0:
        raise F.LvqError(f"attention_stream_totals: shape (nq={nq}, nkv={nkv}, dh={dh}) is not a long-stream shape")
    ws = _att_ws(dev, nbytes)
    tot = torch.empty((n_heads, nq, dh + 2), dtype=torch.float32, device=dev)
    rc = L.lvq_attention_bf16_stream_totals(F.ptr(qh), F.ptr(ql), F.ptr(kv), F.ptr(kv[:, d:]), F.cint(n_heads), F.cint(nq), F.cint(nkv), F.cint(dh),
                                            F.i64(d), F.i64(dh), F.i64(2 * d), F.i64(dh), F.cfloat(scale), F.cint(int(k_fp16)), F.ptr(tot), F.ptr(ws), F.csize(ws.numel()),
                                            F.stream_ptr(dev))
    F.check(rc, f"lvq_attention_bf16_stream_totals (H={n_heads}, nq={nq}, nkv={nkv})")
    return tot


def bev_scene_pairs(row_src: torch.Tensor, batch: int, n_tiles: int, row_base: int):
    """Per-scene pair lists of the signed stream -> (pair_src [batch, n_tiles, 64] i32, pair_info [batch, 2] i32 = pair tiles, use flag)."""
    dev = row_src.device
    pair_src = torch.empty((batch, n_tiles, 64), dtype=torch.int32, device=dev)
    pair_info = torch.empty((batch, 2), dtype=torch.int32, device=dev)
    rc = F.lib().lvq_bev_scene_pairs(F.ptr(row_src), F.cint(batch), F.cint(n_tiles), F.cint(row_base), F.cint(n_tiles), F.ptr(pair_src),
                                     F.ptr(pair_info), F.stream_ptr(dev))
    F.check(rc, "lvq_bev_scene_pairs")
    return pair_src, pair_info


def attention_tiled_signed(q: BF, kv: torch.Tensor, row_src: torch.Tensor, pair_src: torch.Tensor, pair_info: torch.Tensor, totals: torch.Tensor, *, batch: int, n_heads: int, nq: int, n_tiles: int, dh: int, scale: float,
                           shared_q: bool, tag: Optional[str] = None, k_fp16: bool = False, stats: Optional[torch.Tensor] = None) -> BF:
    """attention_tiled over the dirty rows only (queries independent of the batch; `totals` from attention_stream_totals with the
    SAME q over the table rows kv[:n_tiles*64]).  q BF [nq, d] when shared_q else [batch*nq, d] (identical per batch) -> BF [batch*nq, d]."""
    qh, ql = q
    dev = qh.device
    d = n_heads * dh
    oh, ol = _bf_empty((batch * nq, d), dev, ql is not None)
    L = F.lib()
    nbytes = int(L.lvq_attention_tiled_signed_workspace_bytes(F.cint(batch), F.cint(n_heads), F.cint(nq), F.cint(n_tiles), F.cint(dh)))
    if nbytes == 0:
        raise F.LvqError(f"attention_tiled_signed: shape (nq={nq}, tiles={n_tiles}, dh={dh}) is not a long-stream shape")
    ws = _att_ws(dev, nbytes)
    with region(tag, dev):
        rc = L.lvq_attention_bf16_tiled_signed(F.ptr(qh), F.ptr(ql), F.ptr(kv), F.ptr(kv[:, d:]), F.ptr(row_src), F.ptr(pair_src), F.ptr(pair_info), F.cint(pair_src.shape[1]), F.ptr(totals),
                                               F.cint(batch), F.cint(n_heads), F.cint(nq), F.cint(n_tiles), F.cint(dh),
                                               F.i64(0 if shared_q else nq * d), F.i64(d), F.i64(dh), F.i64(2 * d), F.i64(dh), F.i64(nq * d), F.i64(d),
                                               F.i64(dh), F.cfloat(scale), F.cint(1 if k_fp16 else 0), F.ptr(oh), F.ptr(ol), F.ptr(stats), F.ptr(ws), F.csize(ws.numel()), F.stream_ptr(dev))
    F.check(rc, f"lvq_attention_bf16_tiled_signed (B={batch}, H={n_heads}, nq={nq}, tiles={n_tiles})")
    return oh, ol


def stream_guard(q_hi: torch.Tensor, k_rows: torch.Tensor, n_heads: int, scale: float) -> torch.Tensor:
    """Per-model half of the plain-stream guard (include/lvq.h: lvq_stream_guard): q_hi [nq, n_heads * 64] bf16, k_rows [nkv, >= n_heads * 64]
    bf16 (row stride = k_rows.stride(0)) -> g [n_heads, nq] fp32, g = (1 + max |score|) / sqrt(N_eff) per (head, query)."""
    F.require_cuda(q_hi)
    assert q_hi.dtype == torch.bfloat16 and k_rows.dtype == torch.bfloat16 and k_rows.stride(1) == 1 and k_rows.is_cuda
    nq, nkv = q_hi.shape[0], k_rows.shape[0]
    dev = q_hi.device
    L = F.lib()
    L.lvq_stream_guard_workspace_bytes.restype = F.csize
    nbytes = int(L.lvq_stream_guard_workspace_bytes(F.cint(nq), F.i64(nkv)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)          # once per weights version: not cached
    g = torch.empty((n_heads, nq), dtype=torch.float32, device=dev)
    rc = L.lvq_stream_guard(F.ptr(q_hi), F.ptr(k_rows), F.cint(n_heads), F.cint(nq), F.i64(nkv), F.i64(q_hi.stride(0)), F.i64(k_rows.stride(0)),
                            F.cfloat(scale), F.ptr(g), F.ptr(ws), F.csize(nbytes), F.stream_ptr(dev))
    F.check(rc, "lvq_stream_guard")
    return g


def scale_add_rows(x: torch.Tensor, add: Optional[torch.Tensor], alpha: float = 1.0) -> torch.Tensor:
    F.require_cuda(x, add)
    rows, d = x.shape
    out = torch.empty_like(x)
    rc = F.lib().lvq_scale_add_rows(F.ptr(x), F.ptr(add), F.i64(add.shape[0] if add is not None else 0), F.cfloat(alpha),
                                    F.i64(rows), F.cint(d), F.ptr(out), F.stream_ptr(x.device))
    F.check(rc, "lvq_scale_add_rows")
    return out


def rope_inplace(x: BF, rows: int, seq_len: int, n_heads: int, dh: int, ld: int, theta: float, pos0: int = 0):
    """Rotary embedding applied in place to a (column slice of a) packed projection; position = pos0 + row % seq_len."""
    hi, lo = x
    if pos0:
        rc = F.lib().lvq_rope_inplace_at(F.ptr(hi), F.ptr(lo), F.i64(rows), F.cint(seq_len), F.cint(pos0), F.cint(n_heads), F.cint(dh),
                                         F.i64(ld), F.cfloat(theta), F.stream_ptr(hi.device))
        F.check(rc, "lvq_rope_inplace_at")
        return
    rc = F.lib().lvq_rope_inplace(F.ptr(hi), F.ptr(lo), F.i64(rows), F.cint(seq_len), F.cint(n_heads), F.cint(dh), F.i64(ld),
                                  F.cfloat(theta), F.stream_ptr(hi.device))
    F.check(rc, "lvq_rope_inplace")


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    """[rows, n] fp32 -> [rows] int64, first maximum of every row (greedy decoding)."""
    F.require_cuda(x)
    rows, n = x.shape
    out = torch.empty((rows,), dtype=torch.int64, device=x.device)
    F.check(F.lib().lvq_argmax_rows(F.ptr(x), F.i64(rows), F.cint(n), F.ptr(out), F.stream_ptr(x.device)), "lvq_argmax_rows")
    return out


def sample_rows(logits: torch.Tensor, temperature: float, top_k: int, top_p: float, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """[rows, vocab] fp32 -> [rows] int64 drawn like transformers' do_sample step (temperature -> top-k -> top-p -> multinomial,
    lvq_sample_rows).  The uniforms come from torch's RNG on the logits' device (seedable through `generator` / torch.manual_seed)."""
    F.require_cuda(logits)
    rows, n = logits.shape
    u = torch.rand((rows,), dtype=torch.float32, device=logits.device, generator=generator)
    out = torch.empty((rows,), dtype=torch.int64, device=logits.device)
    rc = F.lib().lvq_sample_rows(F.ptr(logits), F.i64(rows), F.cint(n), F.cfloat(temperature), F.cint(int(top_k or 0)), F.cfloat(top_p),
                                 F.ptr(u), F.ptr(out), F.stream_ptr(logits.device))
    F.check(rc, f"lvq_sample_rows (vocab={n}, top_k={top_k}: the top-k filter may only be disabled for vocab <= 1024)")
    return out


def swiglu(gate_up: torch.Tensor, split: bool) -> BF:
    rows, two_i = gate_up.shape
    hi, lo = _bf_empty((rows, two_i // 2), gate_up.device, split)
    rc = F.lib().lvq_swiglu(F.ptr(gate_up), F.i64(rows), F.cint(two_i // 2), F.ptr(hi), F.ptr(lo), F.stream_ptr(gate_up.device))
    F.check(rc, "lvq_swiglu")
    return hi, lo


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """Mean CE over labels >= 0 (ignore_index -100): returns a 0-dim device tensor, no host sync."""
    rows, vocab = logits.shape
    acc = torch.zeros(2, dtype=torch.float32, device=logits.device)
    rc = F.lib().lvq_cross_entropy(F.ptr(logits), F.ptr(labels), F.i64(rows), F.cint(vocab), F.ptr(acc), F.stream_ptr(logits.device))
    F.check(rc, "lvq_cross_entropy")
    return acc[0] / acc[1]


GEMM_LN_WIDTHS = (256, 512, 768, 896, 1024)


def linear_ln_supported(n: int, k: int) -> bool:
    return n in GEMM_LN_WIDTHS and k % 32 == 0 and k <= 256


def linear_ln(a: BF, w: BF, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: Optional[torch.Tensor], eps: float,
              post: Optional[torch.Tensor] = None, tag: Optional[str] = None, out_lo: Optional[bool] = None) -> BF:
    """LayerNorm(a @ w.T + bias) * gamma + beta + post[row % rows] -> BF, no fp32 [M,N] intermediate (lvq_gemm_ln_bf16).
    out_lo=False: split operands, plain result (the "mixed" mode's BEV tokens)."""
    ah, al = a
    wh, wl = w
    m, k = ah.shape
    n = wh.shape[0]
    split = al is not None
    yh, yl = _bf_empty((m, n), ah.device, split if out_lo is None else (out_lo and split))
    with region(tag, ah.device):
        rc = F.lib().lvq_gemm_ln_bf16(F.ptr(ah), F.ptr(al), F.ptr(wh), F.ptr(wl), F.ptr(bias), F.ptr(gamma), F.ptr(beta), F.cfloat(eps),
                                      F.ptr(post), F.i64(post.shape[0] if post is not None else 0), F.i64(m), F.cint(n), F.cint(k),
                                      F.i64(k), F.i64(k), F.ptr(yh), F.ptr(yl), F.stream_ptr(ah.device))
    F.check(rc, f"lvq_gemm_ln_bf16 (m={m}, n={n}, k={k})")
    return yh, yl
