"""GPU tests of the sparse (tiled) BEV key stream of VATLiDAR (csrc/bev_tiles.hip; include/lvq.h "sparse BEV key stream"):
tile bookkeeping against numpy, the fused token kernel against the unfused kernels, the live-row GEMM, the tiled attention
against the dense call on the gathered rows (bit for bit), and the whole route against its all-tiles-live form (bit for bit)
and against the CPU oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lidar_vision_vqa_amd import pipeline as P  # noqa: E402
from lidar_vision_vqa_amd import synth  # noqa: E402
from oracle import pipeline_oracle as PO  # noqa: E402

DEV = "cuda:0"


def ops():
    from lidar_vision_vqa_amd import ops as o
    return o


def random_pillars(B, H, W, M, seed, C=64):
    g = torch.Generator().manual_seed(seed)
    cells = torch.randperm(B * H * W, generator=g)[:M]
    cells[:4] = torch.tensor([0, W - 1, (H - 1) * W, B * H * W - 1])            # image corners
    coords = torch.stack((cells // (H * W), torch.zeros_like(cells), (cells // W) % H, cells % W), 1).to(torch.int32)
    feat = torch.from_numpy(synth.randn((M, C), seed + 1))
    return coords.to(DEV).contiguous(), feat.to(DEV)


def _np_bookkeeping(occ, row_base):
    """numpy restatement of lvq_bev_tiles: occ [B, H, W] bool -> (live codes, piece_dirty, row_src [B, HW], counts)."""
    B, H, W = occ.shape
    tw = W // 8
    nt = (H // 8) * tw
    pad = np.pad(occ, ((0, 0), (1, 1), (1, 1)))
    dirty = np.zeros((B, H, W), bool)                                             # a pillar in the 3 x 3 neighbourhood
    for dy in range(3):
        for dx in range(3):
            dirty |= pad[:, dy:dy + H, dx:dx + W]
    mask = np.zeros((B, nt, 8), np.int64)                                         # bit j = 4 cy + cx of piece p (rows 2 (p >> 1) .., columns 4 (p & 1) ..)
    for t in range(nt):
        for p in range(8):
            y0, x0 = (t // tw) * 8 + (p >> 1) * 2, (t % tw) * 8 + (p & 1) * 4
            for j in range(8):
                mask[:, t, p] |= dirty[:, y0 + (j >> 2), x0 + (j & 3)].astype(np.int64) << j
    codes, pdirty = [], []
    row_src = np.empty((B, nt * 64), np.int64)
    nd = 0
    for t in range(nt):
        for s in range(B):
            for p in range(8):
                m = int(mask[s, t, p])
                if m:
                    codes.append((t * B + s) * 8 + p)
                    pdirty.append((nd, m))
                for j in range(8):
                    e = 64 * t + 8 * p + j
                    if (m >> j) & 1:
                        row_src[s, e] = row_base + nd
                        nd += 1
                    else:
                        row_src[s, e] = e
    return codes, pdirty, row_src, (len(codes), 8 * len(codes), nd)


@pytest.mark.parametrize("B,H,W,M", [(3, 64, 64, 40), (1, 128, 96, 900), (2, 32, 32, 0), (2, 16, 24, 5000)])
def test_tile_bookkeeping_vs_numpy(B, H, W, M):
    o = ops()
    M = min(M, B * H * W)
    coords, _ = random_pillars(B, H, W, max(M, 4), 11)
    n_live = torch.tensor([M], dtype=torch.int32, device=DEV)
    idx = o.pillar_index_map(coords, n_live, B, H, W)
    occ = np.zeros((B, H, W), bool)
    c = coords.cpu().numpy()[:M]
    occ[c[:, 0], c[:, 2], c[:, 3]] = True
    assert np.array_equal(idx.cpu().numpy() >= 0, occ)
    nt = (H // 8) * (W // 8)
    base = H * W
    codes, pdirty, row_src, cnt = _np_bookkeeping(occ, base)
    live, dirty, src, counts = o.bev_tiles(idx, B, H, W, DEV, base)
    assert counts.cpu().numpy().tolist() == list(cnt)
    n = cnt[0]
    assert live.cpu().numpy()[:n].tolist() == codes
    assert dirty.cpu().numpy()[:n].tolist() == [list(x) for x in pdirty]
    assert np.array_equal(src.cpu().numpy(), row_src.astype(np.int32))
    live2, dirty2, src2, counts2 = o.bev_tiles(idx, B, H, W, DEV, 0, force_all=True)
    assert counts2.cpu().numpy().tolist() == [B * nt * 8, B * nt * 64, B * nt * 64] and bool((dirty2[:, 1] == 255).all())
    if B == 1:                                                                    # the table build: every row in key order
        assert np.array_equal(src2.cpu().numpy()[0], np.arange(H * W))


def _tile_major(rows_hw: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """[.., H*W, d] row-major cells -> the tiled stream's key order: tile (8 x 8) major, piece (2 x 4) next, cell inside the piece last."""
    lead, d = rows_hw.shape[:-2], rows_hw.shape[-1]
    n = len(lead)
    v = rows_hw.view(*lead, H // 8, 4, 2, W // 8, 2, 4, d)                      # (ty, pr, r, tx, pc, c, d)
    return v.permute(*range(n), n + 0, n + 3, n + 1, n + 4, n + 2, n + 5, n + 6).reshape(*lead, H * W, d)


@pytest.mark.parametrize("n,split", [(256, False), (768, False), (768, True), (512, True), (1024, False)])
def test_tile_tokens_vs_unfused_kernels(n, split):
    """lvq_bev_tile_tokens (conv + proj + LayerNorm + table in one kernel, live tiles only) against lvq_pillar_dwconv3x3_gelu +
    lvq_gemm_ln_bf16 on every cell: the conv tokens are the same numbers, the product the same MFMA chain; LayerNorm statistics are
    summed in another order, so rows agree to an ulp of the bf16 result."""
    o = ops()
    B, H, W, M, C = 2, 64, 48, 300, 64
    coords, feat = random_pillars(B, H, W, M, 21)
    n_live = torch.tensor([M - 3], dtype=torch.int32, device=DEV)                 # trailing rows ignored
    w9 = torch.from_numpy(synth.randn((C, 9), 22, 0.3)).to(DEV)
    b9 = torch.from_numpy(synth.randn((C,), 23)).to(DEV)
    wp = torch.from_numpy(synth.randn((n, C), 24, 0.15)).to(DEV)
    bias = torch.from_numpy(synth.randn((n,), 25)).to(DEV)
    gam = torch.from_numpy(synth.randn((n,), 26)).to(DEV) + 1.0
    bet = torch.from_numpy(synth.randn((n,), 27)).to(DEV)
    pe = torch.from_numpy(synth.randn((H * W, n), 28)).to(DEV)
    pe_t = _tile_major(pe, H, W).contiguous()
    t = o.pillar_dwconv3x3_gelu(feat, coords, n_live, B, H, W, w9, b9, split)
    ref = o.to_f32(o.linear_ln(t, o.cast(wp, split), bias, gam, bet, 1e-5, post=pe)).view(B, H * W, n)
    ref_t = _tile_major(ref, H, W)
    idx = o.pillar_index_map(coords, n_live, B, H, W)
    nt = (H // 8) * (W // 8)
    for force in (True, False):
        live, dirty, src, counts = o.bev_tiles(idx, B, H, W, DEV, 0, force_all=force)
        x = o.bev_tile_tokens(feat, idx, live, dirty, counts, B * nt * 64, B, H, W, w9, b9, o.cast(wp, split), bias, gam, bet, 1e-5, pe_t, out_lo=split)
        got = o.to_f32(x)
        nl, nd = int(counts[0]), int(counts[2])
        assert (nl == B * nt * 8 and nd == B * nt * 64) if force else (0 < nl < B * nt * 8 and 0 < nd < 8 * nl)
        srcn = src.cpu().numpy()
        for s in range(B):                                                        # row row_src[s, e] (when computed) holds key e of scene s
            e = np.nonzero(srcn[s] != np.arange(H * W))[0] if not force else np.arange(H * W)
            rows = torch.from_numpy(srcn[s][e].astype(np.int64)).to(DEV)
            want = ref_t[s][torch.from_numpy(e).to(DEV)]
            assert float((got[rows] - want).abs().max()) < (3e-4 if split else 2.0 ** -7 * float(want.abs().max())), (force, s)
    # a clean cell equals the all-empty-scene value of that cell (what the per-model table holds), bit for bit
    empty = torch.full_like(idx[:1], -1)
    live_e, dirty_e, src_e, counts_e = o.bev_tiles(empty, 1, H, W, DEV, 0, force_all=True)
    xe = o.bev_tile_tokens(feat[:1] * 0, empty, live_e, dirty_e, counts_e, nt * 64, 1, H, W, w9, b9, o.cast(wp, split), bias, gam, bet, 1e-5, pe_t,
                           out_lo=split)
    live_f, dirty_f, src_f, counts_f = o.bev_tiles(idx, B, H, W, DEV, 0, force_all=True)
    xf = o.bev_tile_tokens(feat, idx, live_f, dirty_f, counts_f, B * nt * 64, B, H, W, w9, b9, o.cast(wp, split), bias, gam, bet, 1e-5, pe_t, out_lo=split)
    srcs = o.bev_tiles(idx, B, H, W, DEV, H * W)[2].cpu().numpy()
    srcf = src_f.cpu().numpy()
    clean = [(s, e) for s in range(B) for e in range(0, H * W, 7) if srcs[s, e] == e][:200]
    assert clean
    for s, e in clean:
        assert torch.equal(xf[0][srcf[s, e]], xe[0][e])


def test_gemm_live_rows_device_count():
    o = ops()
    cap, n, k = 4096, 512, 128
    a = torch.from_numpy(synth.randn((cap, k), 31)).to(DEV)
    w = torch.from_numpy(synth.randn((n + 64, k), 32, 0.1)).to(DEV)
    bias = torch.from_numpy(synth.randn((n + 64,), 33)).to(DEV)
    for rows in (0, 64, 1000, 4096):
        for a_split, w_split in ((False, False), (False, True), (True, True)):
            ab, wb = o.cast(a, a_split), o.cast(w, w_split)
            ref = o.to_f32(o.linear(ab, wb, bias, out_bf=True, w_rows=(64, 64 + n))[1])
            rd = torch.tensor([rows], dtype=torch.int32, device=DEV)
            canary = 7.0
            got = o.linear_live_rows(ab, wb, bias, rd, (64, 64 + n))
            assert (got[1] is not None) == a_split
            g32 = o.to_f32(got)
            if rows:
                assert float((g32[:rows] - ref[:rows]).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()), (rows, a_split, w_split)
            # rows past the last whole 256-row tile of the live count are never written
            up = (rows + 255) // 256 * 256
            if up < cap:
                got[0][up:].fill_(canary)
                again = o.linear_live_rows(ab, wb, bias, rd, (64, 64 + n))
                del again
            del canary


def _row_case(B, H, n_tiles, dirty_frac, seed, k_new_scale=1.0, k_tab_scale=1.0, shuffle=True):
    """Random row-granular stream: ONE K|V buffer [table rows (n_tiles*64) | computed rows], row_src [B, n_tiles*64] (key e of batch b is
    a computed row with probability dirty_frac[b], its table row e otherwise) and the dirty flags."""
    d = H * 64
    hw = n_tiles * 64
    g = torch.Generator().manual_seed(seed)
    fr = torch.tensor(dirty_frac if isinstance(dirty_frac, (list, tuple)) else [dirty_frac] * B).view(B, 1)
    is_dirty = torch.rand(B, hw, generator=g) < fr
    if float(fr[0]) > 0:
        is_dirty[0, :5] = torch.tensor([True, False, True, True, False])
    n_d = int(is_dirty.sum())
    kv = torch.randn(hw + n_d + 7, 2 * d, generator=g)
    kv[:hw, :d] *= k_tab_scale
    kv[hw:, :d] *= k_new_scale
    num = torch.randperm(n_d, generator=g) if shuffle else torch.arange(n_d)         # any numbering of the computed rows is legal
    src = torch.arange(hw).repeat(B, 1)
    src[is_dirty] = hw + num
    return kv.to(torch.bfloat16).to(DEV), src.to(torch.int32), is_dirty


@pytest.mark.parametrize("B,H,nq,n_tiles,qsplit", [(2, 2, 120, 64, False), (3, 4, 576, 128, True), (1, 12, 576, 256, True)])
def test_attention_tiled_equals_dense_on_gathered_rows(B, H, nq, n_tiles, qsplit):
    """Every key slot through its own source row of the one K|V buffer == the dense call on the same rows gathered into a
    [B, 64 n_tiles, 2d] buffer: same kernel, same key order -> bit-identical outputs."""
    o = ops()
    dh = 64
    d = H * dh
    kv, src, _ = _row_case(B, H, n_tiles, 0.4, 5)
    dense = kv[src.to(DEV).long().view(-1)].view(B * n_tiles * 64, 2 * d).contiguous()
    q = torch.randn(B * nq, d, generator=torch.Generator().manual_seed(6)).to(DEV)
    qb = o.cast(q, qsplit)
    got = o.attention_tiled(qb, kv, src.to(DEV).contiguous(), batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=dh, scale=1.0 / math.sqrt(dh))
    ref = o.attention(qb, (dense, None), (dense[:, d:], None), batch=B, n_heads=H, n_kv_heads=H, nq=nq, nkv=n_tiles * 64, dh=dh,
                      q_strides=(nq * d, d, dh), k_strides=(n_tiles * 64 * 2 * d, 2 * d, dh), v_strides=(n_tiles * 64 * 2 * d, 2 * d, dh),
                      scale=1.0 / math.sqrt(dh))
    assert torch.equal(got[0], ref[0])
    if qsplit:
        assert torch.equal(got[1], ref[1])


@pytest.mark.parametrize("one_wg", [False, True])
@pytest.mark.parametrize("B,nt,fr", [(4, 96, [0.3, 0.0, 0.7, 0.05]), (3, 1100, [0.27, 0.49, 0.002])])
def test_scene_pairs_vs_numpy(B, nt, fr, one_wg, tune):
    """Pair lists from the chunked two-kernel form (many workgroups per scene, the default) and from the one-workgroup-per-scene form."""
    tune(pairs_one_wg=int(one_wg))
    o = ops()
    hw = nt * 64
    _, src, _ = _row_case(B, 2, nt, fr, 3)
    pair_src, pair_info = o.bev_scene_pairs(src.to(DEV).contiguous(), B, nt, hw)
    ps, pi = pair_src.cpu().numpy(), pair_info.cpu().numpy()
    s_np = src.numpy()
    for b in range(B):
        lv = [(int(s_np[b, e]), e) for e in range(hw) if s_np[b, e] >= hw]
        n_pt = (len(lv) + 31) // 32
        use = n_pt < nt
        assert pi[b, 1] == int(use) and pi[b, 0] == (n_pt if use else 0)
        if not use:
            continue
        for j, (row, e) in enumerate(lv):
            assert ps[b, j // 32, j % 32] == row and ps[b, j // 32, 32 + j % 32] == e
        for j in range(len(lv), 32 * n_pt):                        # padding: table row 0 in both halves
            assert ps[b, j // 32, j % 32] == 0 and ps[b, j // 32, 32 + j % 32] == 0


def _shared_q(o, H, nq, seed):
    return o.cast(torch.randn(nq, H * 64, generator=torch.Generator().manual_seed(seed)).to(DEV), True)


def _full_reference(o, qb, kv, src, B, H, nq, n_tiles):
    d = H * 64
    qh = qb[0].unsqueeze(0).expand(B, nq, d).reshape(B * nq, d).contiguous()
    ql = qb[1].unsqueeze(0).expand(B, nq, d).reshape(B * nq, d).contiguous()
    return o.attention_tiled((qh, ql), kv, src.to(DEV).contiguous(), batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=64, scale=1.0 / 8.0)


def _signed(o, qb, kv, src, B, H, nq, n_tiles):
    hw = n_tiles * 64
    srcd = src.to(DEV).contiguous()
    pair_src, pair_info = o.bev_scene_pairs(srcd, B, n_tiles, hw)
    tot = o.attention_stream_totals(qb, kv[:hw], n_heads=H, nq=nq, nkv=hw, dh=64, scale=1.0 / 8.0)
    out = o.attention_tiled_signed(qb, kv, srcd, pair_src, pair_info, tot, batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=64, scale=1.0 / 8.0,
                                   shared_q=True)
    return out, pair_info


@pytest.mark.parametrize("B,H,nq,n_tiles,fr", [(3, 2, 120, 64, 0.35), (2, 4, 576, 128, [0.4, 0.9]), (4, 12, 576, 256, [0.0, 0.2, 0.45, 0.6]),
                                               (1, 2, 120, 67, 0.3)])
def test_attention_tiled_signed_matches_full_stream(B, H, nq, n_tiles, fr):
    """TOTALS(table) - table rows of the dirty cells + their computed rows == the full stream, up to fp32 accumulation order; batches
    whose pair list is not shorter (dirty > 50 %) run their full list inside the same launch."""
    o = ops()
    kv, src, is_dirty = _row_case(B, H, n_tiles, fr, 17)
    qb = _shared_q(o, H, nq, 18)
    ref = _full_reference(o, qb, kv, src, B, H, nq, n_tiles)
    (oh, ol), pair_info = _signed(o, qb, kv, src, B, H, nq, n_tiles)
    pi = pair_info.cpu().numpy()
    want_use = [int((int(is_dirty[b].sum()) + 31) // 32 < n_tiles) for b in range(B)]
    assert pi[:, 1].tolist() == want_use
    got, want = o.to_f32((oh, ol)).view(B, nq, -1), o.to_f32(ref).view(B, nq, -1)
    scale = want.abs().max().item()
    for b in range(B):                                            # (full-list batches: same stream, the signed row sum takes another path)
        assert (got[b] - want[b]).abs().max().item() < (2e-5 if want_use[b] else 2e-6) * max(scale, 1.0), (b, (got[b] - want[b]).abs().max().item())


@pytest.mark.parametrize("kind", ["cancel", "overflow"])
def test_attention_tiled_signed_falls_back_when_unusable(kind):
    """(a) The computed keys score far below the table keys they replace: what is left after the subtraction is < 1/16 of the table total
    -> the (batch, head) is flagged and redone over its full stream (then bit-identical to the full stream).  (b) computed scores beyond
    2^128: the signed sums are not finite -> same re-run (which itself falls back to the classic softmax form)."""
    o = ops()
    B, H, nq, n_tiles = 2, 2, 120, 64
    hw = n_tiles * 64
    if kind == "cancel":
        kv, src, is_dirty = _row_case(B, H, n_tiles, [0.45, 0.3], 23, k_new_scale=0.01, k_tab_scale=6.0)
        # the cells that stay clean in some batch must hold almost no mass: shrink their table keys (cells dirty in BOTH keep the large ones)
        clean_somewhere = (~is_dirty).any(0).to(DEV)
        t32 = kv.float()
        tab = t32[:hw]
        tab[clean_somewhere, :H * 64] *= 0.002
        kv = t32.to(torch.bfloat16)
    else:
        kv, src, _ = _row_case(B, H, n_tiles, [0.45, 0.3], 23, k_new_scale=400.0)
    qb = _shared_q(o, H, nq, 24)
    ref = _full_reference(o, qb, kv, src, B, H, nq, n_tiles)
    (oh, ol), _ = _signed(o, qb, kv, src, B, H, nq, n_tiles)
    assert torch.isfinite(o.to_f32((oh, ol))).all()
    assert torch.equal(oh, ref[0]) and torch.equal(ol, ref[1])


@pytest.mark.parametrize("B,H,nq,n_tiles,fr,qsplit", [(2, 2, 120, 64, 0.35, True), (3, 4, 576, 128, [0.4, 0.9, 0.0], True), (2, 12, 576, 256, [0.2, 0.45], True),
                                                      (1, 2, 120, 67, 0.3, True), (2, 2, 240, 72, 0.3, False)])
def test_pipelined_stream_is_bit_identical(B, H, nq, n_tiles, fr, qsplit, tune):
    """The software-pipelined form of the long-stream kernel (4 LDS slots, loader-only waves for the query padding; the default for
    split and fp16 queries, lvq_tuning.attn_pipe = 1 forces it for plain ones) against the plain form (attn_pipe = -1).  Same arithmetic in
    the same order: with the KV split pinned, the tiled stream, the per-model totals and the signed pair stream (pair lists of any
    length, full-list batches, 1 .. n tiles per split) are equal bit for bit."""
    o = ops()
    kv, src, _ = _row_case(B, H, n_tiles, fr, 31)
    srcd = src.to(DEV).contiguous()
    hw = n_tiles * 64
    d = H * 64
    res = {}
    for pipe in (False, True):
        tune(attn_pipe=1 if pipe else -1)
        out = {}
        for ns in ("1", "3", "8"):
            if 8 * int(ns) > n_tiles:
                continue
            tune(attn_nsplit=int(ns))
            qb = o.cast(torch.randn(nq, d, generator=torch.Generator().manual_seed(32)).to(DEV), qsplit)
            qfull = tuple(None if x is None else x.unsqueeze(0).expand(B, nq, d).reshape(B * nq, d).contiguous() for x in qb)
            out["tiled", ns] = o.attention_tiled(qfull, kv, srcd, batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=64, scale=0.125)
            if qsplit:
                pair_src, pair_info = o.bev_scene_pairs(srcd, B, n_tiles, hw)
                tot = o.attention_stream_totals(qb, kv[:hw], n_heads=H, nq=nq, nkv=hw, dh=64, scale=0.125)
                out["totals", ns] = (tot,)
                out["signed", ns] = o.attention_tiled_signed(qb, kv, srcd, pair_src, pair_info, tot, batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=64,
                                                             scale=0.125, shared_q=True)
                if ns == "3":                                     # the fp16 Q K^T form (K half of the buffer as IEEE fp16 bit patterns)
                    kv_h = kv.clone(); kv_h[:, :d] = kv[:, :d].float().half().view(torch.bfloat16)
                    tot16 = o.attention_stream_totals(qb, kv_h[:hw], n_heads=H, nq=nq, nkv=hw, dh=64, scale=0.125, k_fp16=True)
                    out["totals16", ns] = (tot16,)
                    out["signed16", ns] = o.attention_tiled_signed(qb, kv_h, srcd, pair_src, pair_info, tot16, batch=B, n_heads=H, nq=nq, n_tiles=n_tiles,
                                                                   dh=64, scale=0.125, shared_q=True, k_fp16=True)
        res[pipe] = out
    tune(attn_nsplit=0, attn_pipe=0)
    assert res[False].keys() == res[True].keys() and len(res[True]) > 0
    for k in res[False]:
        for a, b in zip(res[False][k], res[True][k]):
            if a is None:
                assert b is None
                continue
            va = a.view(torch.int16) if a.dtype == torch.bfloat16 else a.view(torch.int32)
            vb = b.view(torch.int16) if b.dtype == torch.bfloat16 else b.view(torch.int32)
            assert torch.equal(va, vb), k


def tiled_cfg(**kw):
    base = dict(n_points=8192, d_model=256, n_heads=4, n_queries=120, n_layers=2, n_patches=196, voxel_pillar=(0.8, 0.8, 8.0), max_pillars=30000)
    base.update(kw)
    return P.PipelineConfig(**base)


@pytest.mark.parametrize("prec", ["bf16", "mixed"])
def test_tiled_route_is_bit_identical_to_all_tiles_live(prec, monkeypatch):
    """The sparse key stream (clean tiles from the per-model table) == the same route with every tile computed per scene, bit for
    bit, incl. an EMPTY scene (every tile clean) and two layers; and it is the route actually taken (live tiles < all tiles)."""
    cfg = tiled_cfg()
    pipe = P.FusionPipeline(cfg, DEV, precision=prec)
    pts, off, patches, pts_np, _ = P.synthetic_batch(cfg, 3, 1001, DEV)
    n0 = len(pts_np[0])
    off2 = torch.tensor([0, n0, n0, pts.shape[0]], dtype=torch.int32, device=DEV)   # scene 1 is empty
    h, w = cfg.bev_hw
    assert pipe.vat_lidar._tiled_route_ok(64, h, w)
    signed = pipe(pts, off2, patches)                              # default: block 0 streams the live pieces only (signed pair stream)
    monkeypatch.setenv("LVQ_NO_SIGNED_STREAM", "1")
    a = pipe(pts, off2, patches)
    # same operand roundings, different fp32 accumulation order in block 0's attention
    assert (signed["lidar_tokens"] - a["lidar_tokens"]).abs().max().item() < (2e-2 if prec == "bf16" else 2e-4)
    nl = int(pipe.vat_lidar._last_tile_counts[0])
    assert 0 < nl < 3 * (h // 8) * (w // 8) * 8
    orig = pipe.vat_lidar.forward_pillars
    monkeypatch.setattr(pipe.vat_lidar, "forward_pillars", lambda *x, **k: orig(*x, all_tiles_live=True, **k))
    b = pipe(pts, off2, patches)
    assert int(pipe.vat_lidar._last_tile_counts[0]) == 3 * (h // 8) * (w // 8) * 8
    assert torch.equal(a["lidar_tokens"], b["lidar_tokens"]) and torch.equal(a["fused"], b["fused"])
    monkeypatch.undo()
    # the older (cell-order, untiled) route agrees to the precision mode's tolerance
    monkeypatch.setenv("LVQ_NO_TILED_STREAM", "1")
    c = pipe(pts, off2, patches)
    tol = 5e-2 if prec == "bf16" else 2e-3
    assert (a["lidar_tokens"] - c["lidar_tokens"]).abs().max().item() < tol


@pytest.mark.parametrize("prec", ["bf16", "mixed"])
def test_fused_kv_kernel_matches_token_kernel_plus_gemm(prec, monkeypatch):
    """lvq_bev_tile_kv (LayerNorm and the K|V projection folded onto the 64-channel conv token: K|V = rstd (M t + m0) + T[key]) against
    the unfused pair lvq_bev_tile_tokens -> lvq_gemm_bf16_live_rows on the same scenes: the K|V rows of the dirty cells and of the
    per-model table agree to the rounding of the bf16 result (the fused form skips the bf16 rounding of the d-wide token), and so do
    the LiDAR tokens.  The two-launch form equals the single-kernel form bit for bit (same arithmetic), and the fp16 table
    option (LVQ_KV_T_FP16) moves a row by at most one bf16 ulp against the fp32 table."""
    cfg = tiled_cfg()
    pipe = P.FusionPipeline(cfg, DEV, precision=prec)
    pts, off, patches, _, _ = P.synthetic_batch(cfg, 2, 1001, DEV)
    h, w = cfg.bev_hw
    vl = pipe.vat_lidar
    key = ("kv_buffer", h, w, torch.device(DEV))
    monkeypatch.setenv("LVQ_KV_T_FP16", "1")
    a16 = pipe(pts, off, patches)                                 # two launches, fp16 table (opt-in)
    kv16 = [b.clone() for b in vl._pe_cache[key][1]]
    nd = int(vl._last_tile_counts[2])
    assert 0 < nd < 2 * h * w
    monkeypatch.delenv("LVQ_KV_T_FP16")
    a = pipe(pts, off, patches)                                   # default: two launches, fp32 table
    kv_f = [b.clone() for b in vl._pe_cache[key][1]]
    for x16, x32 in zip(kv16, kv_f):
        f16, f32 = x16[:h * w + nd].float(), x32[:h * w + nd].float()
        assert float((f16 - f32).abs().max()) <= 2.0 ** -7 * float(f32.abs().max())                   # one ulp of the largest row entry
        assert float((f16 != f32).float().mean()) < 0.2                                               # most entries do not move at all
    assert float((a16["lidar_tokens"] - a["lidar_tokens"]).abs().max()) < (5e-2 if prec == "bf16" else 2e-4)
    monkeypatch.setenv("LVQ_KV_ONE_LAUNCH", "1")
    a1 = pipe(pts, off, patches)                                  # single kernel, fp32 table: the same arithmetic
    for f, o1 in zip(kv_f, vl._pe_cache[key][1]):
        assert torch.equal(f[:h * w + nd], o1[:h * w + nd])
    assert torch.equal(a["lidar_tokens"], a1["lidar_tokens"])
    monkeypatch.delenv("LVQ_KV_ONE_LAUNCH")
    monkeypatch.setenv("LVQ_NO_FUSED_KV", "1")
    b = pipe(pts, off, patches)
    kv_u = vl._pe_cache[key][1]
    assert int(vl._last_tile_counts[2]) == nd
    for f, u in zip(kv_f, kv_u):
        f32, u32 = f[:h * w + nd].float(), u[:h * w + nd].float()
        scale = float(u32.abs().max())
        # bf16 results of two different roundings of the same number differ by at most one ulp (2^-8 relative) + the token rounding of the unfused route
        assert float((f32 - u32).abs().max()) < (2.0 ** -6 if prec == "bf16" else 2.0 ** -7) * scale
        assert float((f32 - u32).abs().mean()) < 2.0 ** -10 * scale
    tol = 5e-2 if prec == "bf16" else 5e-4
    assert float((a["lidar_tokens"] - b["lidar_tokens"]).abs().max()) < tol


def test_attention_tiled_signed_fp16_qk():
    """k_fp16: the K half of the buffer holds IEEE fp16, q is rounded once to fp16 and Q K^T is one fp16 MFMA pass -- against the Q-split
    bf16 form on the same (fp16-representable) keys: equal up to the 2^-11 rounding of q."""
    o = ops()
    B, H, nq, n_tiles = 3, 4, 576, 128
    d = H * 64
    hw = n_tiles * 64
    kv, src, _ = _row_case(B, H, n_tiles, [0.3, 0.45, 0.8], 29)
    k16 = kv[:, :d].float().half()                                # keys that both formats hold exactly: fp16 values with 8-bit significands
    kq = k16.float().bfloat16()
    kv_b = kv.clone(); kv_b[:, :d] = kq                           # bf16 view of the same numbers
    kv_h = kv.clone(); kv_h[:, :d] = kq.float().half().view(torch.bfloat16)      # ... and their fp16 bit patterns
    qb = _shared_q(o, H, nq, 30)
    ref, _ = _signed(o, qb, kv_b, src, B, H, nq, n_tiles)
    srcd = src.to(DEV).contiguous()
    pair_src, pair_info = o.bev_scene_pairs(srcd, B, n_tiles, hw)
    tot = o.attention_stream_totals(qb, kv_h[:hw], n_heads=H, nq=nq, nkv=hw, dh=64, scale=1.0 / 8.0, k_fp16=True)
    got = o.attention_tiled_signed(qb, kv_h, srcd, pair_src, pair_info, tot, batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=64, scale=1.0 / 8.0,
                                   shared_q=True, k_fp16=True)
    g32, r32 = o.to_f32(got), o.to_f32(ref)
    err = float((g32 - r32).abs().max())
    assert 0 < err < 3e-3 * max(float(r32.abs().max()), 1.0), err
    full = o.attention_tiled((qb[0].repeat(B, 1), qb[1].repeat(B, 1)), kv_h, srcd, batch=B, n_heads=H, nq=nq, n_tiles=n_tiles, dh=64, scale=1.0 / 8.0,
                             k_fp16=True)
    assert float((o.to_f32(full) - g32).abs().max()) < 2e-5 * max(float(r32.abs().max()), 1.0)


@pytest.mark.parametrize("prec,tol", [("mixed", 1e-3), ("mixed16", 1e-3), ("bf16", None)])
def test_tiled_route_vs_oracle(prec, tol):
    """Small-grid pipeline through the tiled route against the CPU oracle at the north-star 1e-3.  16 384 keys average the per-key
    roundings 4x less than the 262 144 of the bench workload: the stream guard (VATLiDAR.stream_guard) decides per model whether the
    plain key stream is still good for the bar and runs hi + lo operands otherwise (tests/test_gpu_mixed_guard.py)."""
    cfg = tiled_cfg(dist="C")
    pipe = P.FusionPipeline(cfg, DEV, precision=prec)
    pts, off, patches, pts_np, patches_np = P.synthetic_batch(cfg, 2, 1001, DEV)
    out = pipe(pts, off, patches)
    sd = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = PO.run(cfg, pts_np, patches_np, sd(pipe.pillar_vfe), sd(pipe.vat_lidar), sd(pipe.fuse), do_3d=False)
    err = (out["fused"].cpu() - ref["fused"]).abs().max().item()
    if tol is None:
        tol = 2e-2 * ref["fused"].abs().max().item()
    assert err < tol, err
