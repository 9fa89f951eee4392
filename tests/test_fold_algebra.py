"""CPU check of the algebra behind lvq_bev_tile_kv (csrc/bev_tiles.hip: k_conv_rows + k_kv_rows; fusion.VATLiDAR._kv_fold):
LayerNorm of a linear map of the conv token t is a per-row rescale of another linear map of t, so the reference's chain
    x = LayerNorm(Wp t + bp) * gamma + beta + PE[key]          (vat_lidar.py:222-248)
    K|V = W_kv x + b_kv                                          (vat_blocks.py:42, in_proj rows d .. 3d)
equals  rstd (M t + m0) + T[key]  with the factors folded exactly as the host code folds them."""
import math

import torch


def _fold(wp, bp, gam, bet, pe, wkv, bkv):
    d, c = wp.shape
    wc, bc = wp - wp.mean(0, keepdim=True), bp - bp.mean()
    r = torch.linalg.qr(torch.cat((wc, bc[:, None]), 1), mode="r").R               # |Wc t + bc|^2 = |R [t; 1]|^2
    return dict(rt=r[:c, :c], r0=r[:c, c], c0=r[c, c] ** 2, m=wkv @ (gam[:, None] * wc), m0=wkv @ (gam * bc), t=(bet + pe) @ wkv.T + bkv)


def test_folded_kv_equals_layernorm_then_projection():
    g = torch.Generator().manual_seed(3)
    d, c, n, eps = 96, 16, 200, 1e-5
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    wp, bp, gam, bet = rnd(d, c) * 0.3, rnd(d), rnd(d) + 1.0, rnd(d)
    wkv, bkv, pe, t = rnd(2 * d, d) * 0.1, rnd(2 * d), rnd(n, d), rnd(n, c)
    ref = (torch.nn.functional.layer_norm(t @ wp.T + bp, (d,), gam, bet, eps) + pe) @ wkv.T + bkv
    f = _fold(wp, bp, gam, bet, pe, wkv, bkv)
    var = (((t @ f["rt"].T + f["r0"]) ** 2).sum(1) + f["c0"]) / d
    got = (1.0 / torch.sqrt(var + eps))[:, None] * (t @ f["m"].T + f["m0"]) + f["t"]
    assert float((got - ref).abs().max()) < 1e-12 * float(ref.abs().max())


def test_static_bound_on_k_holds():
    """|LayerNorm(y)_n| <= sqrt(d - 1) for any input, hence every K entry is bounded by its weight row (the guard of the mixed16 mode)."""
    g = torch.Generator().manual_seed(4)
    d, c = 64, 8
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    wp, bp, gam, bet = rnd(d, c), rnd(d), rnd(d) + 1.0, rnd(d)
    wk, bk, pe = rnd(d, d) * 0.2, rnd(d), rnd(50, d)
    xmax = math.sqrt(d) * gam.abs() + bet.abs() + pe.abs().amax(0)
    bound = float((wk.abs() @ xmax + bk.abs()).max())
    for scale in (1e-3, 1.0, 1e3, 1e6):                                          # the bound does not depend on the size of the input
        t = rnd(50, c) * scale
        x = torch.nn.functional.layer_norm(t @ wp.T + bp, (d,), gam, bet, 1e-5) + pe
        assert float((x @ wk.T + bk).abs().max()) <= bound
