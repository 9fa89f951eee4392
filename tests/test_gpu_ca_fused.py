"""GPU parity tests of the fused short-K/V cross-attention sub-path (csrc/cross_fused.hip, include/lvq.h: lvq_ca_fused*) through the
C ABI against the CPU oracle (oracle/vat_oracle.py, itself pinned by the goldens of the unmodified reference VATBlock) and against
the goldens of the reference module.

    out = q + ca(ca_ln(q), kv, kv)            encoder-decoder/training/models/vat_blocks.py:41-42

Tolerances (north_star: fused tokens within 1e-3 of the fp32 CPU path): the fp16-operand form must meet 1e-3 max abs outright;
the bf16-operand form is bounded by its operand rounding (2e-2 of max|ref|, as every plain-bf16 kernel chain of this repository)."""
import numpy as np
import pytest
import torch

from conftest import golden, golden_error

pytestmark = pytest.mark.gpu

from lidar_vision_vqa_amd import synth  # noqa: E402
from oracle import vat_oracle as VO  # noqa: E402

DEV = "cuda:0"
TOL = 1e-3
REL_BF16 = 2e-2
D, H = 768, 12


def fusion():
    from lidar_vision_vqa_amd import fusion as f
    return f


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def block(seed):
    m = fusion().VATBlock(D, H, 4 * D, 0.1).to(DEV).eval()
    synth.load_seeded(m, seed)
    return m, {k: v.detach().cpu() for k, v in m.state_dict().items()}


def oracle_ca(q, kv, sd):
    return q + VO.mha(VO.layer_norm(q, sd["ca_ln.weight"], sd["ca_ln.bias"]), kv, sd, "ca.", H)


def run(m, q, kv, prec):
    m.precision = prec
    with torch.no_grad():
        return m.cross_attention(q, kv)


def run_kernel(m, q, kv, f16):
    """lvq_ca_fused itself, whatever route the precision mode would pick for this shape."""
    B, nq, d = q.shape
    with torch.no_grad():
        return m._cross_attn_fused(q.reshape(B * nq, d), kv, B, nq, f16).view(B, nq, d)


def tol16(nkv):
    """fp16 operands: 1e-3 outright from 128 keys on (the shapes the parity-true modes route here: VATBlock._ca_fused_mode); with fewer
    keys the softmax averages fewer rounded V rows and the attention branch itself is larger (one key: out = q + W_o V), so the bound
    is the operand rounding: 2^-11 per stage, ~3 stages deep, on a branch of magnitude <= 3."""
    return TOL if nkv >= 128 else 2.5e-3


# (B, nq, nkv): the resampled-token shape (nq = 576 leaves half a 128-row tile per batch element), a slice of the headline shape, ragged
# key counts (193 = six full blocks + 1 key; 224 = every block full), rows = one wave
SHAPES = [(1, 256, 196), (2, 576, 196), (1, 4096, 196), (1, 128, 193), (3, 64, 224), (1, 32, 200), (2, 160, 211)]


@pytest.mark.parametrize("B,nq,nkv", SHAPES)
def test_fused_ca_matches_oracle(B, nq, nkv):
    from lidar_vision_vqa_amd import ops
    assert ops.ca_fused_ok(B, nq, nkv, D, H)
    m, sd = block(501)
    q, kv = synth.randn((B, nq, D), 502), synth.randn((B, nkv, D), 503)
    ref = oracle_ca(torch.from_numpy(q), torch.from_numpy(kv), sd).numpy()
    out16 = run_kernel(m, dev(q), dev(kv), True).cpu().numpy()
    e16 = np.abs(out16 - ref).max()
    assert e16 <= tol16(nkv), f"fp16 operands: {e16:.3e}"
    if nkv >= 128:                     # the route the parity-true mode takes for this shape IS the kernel
        assert np.array_equal(run(m, dev(q), dev(kv), "mixed").cpu().numpy(), out16)
    outb = run_kernel(m, dev(q), dev(kv), False).cpu().numpy()
    eb = np.abs(outb - ref).max()
    assert eb <= REL_BF16 * np.abs(ref).max(), f"bf16 operands: {eb:.3e}"
    assert np.isfinite(out16).all() and np.isfinite(outb).all()


def test_fused_ca_is_the_route_taken():
    """The parity-true mode and the bf16 mode go through lvq_ca_fused (no silent detour through the unfused chain): the result
    differs from the unfused chain's bit pattern and LVQ_NO_FUSED_CA restores that chain."""
    import os
    m, _ = block(511)
    q, kv = dev(synth.randn((1, 256, D), 512)), dev(synth.randn((1, 196, D), 513))
    a = run(m, q, kv, "bf16")
    os.environ["LVQ_NO_FUSED_CA"] = "1"
    try:
        b = run(m, q, kv, "bf16")
    finally:
        del os.environ["LVQ_NO_FUSED_CA"]
    assert not torch.equal(a, b)
    assert (a - b).abs().max().item() < 2e-2 * b.abs().max().item()


def test_fused_ca_row_offsets_and_scales():
    """LayerNorm inside the kernel is a one-pass shifted form: rows with a large common offset (mean >> spread), rows with tiny and
    with large spread, and a constant row (variance 0 -> eps) must come out like the two-pass fp32 LayerNorm of the oracle."""
    m, sd = block(521)
    q = synth.randn((1, 128, D), 522)
    q[0, 0:16] += 1000.0
    q[0, 16:32] -= 250.0
    q[0, 32:48] *= 1e-3
    q[0, 48:64] *= 300.0
    q[0, 64] = 3.0
    q[0, 65] = 0.0
    kv = synth.randn((1, 196, D), 523)
    ref = oracle_ca(torch.from_numpy(q), torch.from_numpy(kv), sd).numpy()
    out = run(m, dev(q), dev(kv), "mixed").cpu().numpy()
    # the residual itself is O(1000) in the offset rows: compare the attention branch (out - q), whose scale is O(1)
    err = np.abs((out - q) - (ref - q))
    # fp32 cancellation of (x + q) - q at |q| ~ 1000 is ~6e-5; rows scaled by 300 carry a 300x residual as well
    assert err[0, 0:48].max() <= 2e-3 and err[0, 64:].max() <= TOL and err[0, 48:64].max() <= 3e-2, (err[0, 0:16].max(), err[0, 16:32].max(),
                                                                                                 err[0, 32:48].max(), err[0, 48:64].max(), err[0, 64:].max())
    assert np.isfinite(out).all()


def test_fused_ca_fp16_range_guards():
    """kv tokens far outside fp16's range must not produce inf / NaN (operands are clamped where a value is not bounded by
    construction); kv tokens 100x larger (scores 100x larger: a one-hot softmax, whose argmax a 2^-11 rounding of K can move) stay
    within a few per cent of the output scale."""
    m, sd = block(531)
    q = synth.randn((1, 64, D), 532)
    kv = synth.randn((1, 196, D), 533)
    big = kv.copy()
    big[0, 5] *= 1e6
    out = run_kernel(m, dev(q), dev(big), True)
    assert bool(torch.isfinite(out).all())
    kv100 = kv * 100.0
    ref = oracle_ca(torch.from_numpy(q), torch.from_numpy(kv100), sd).numpy()
    out = run_kernel(m, dev(q), dev(kv100), True).cpu().numpy()
    assert np.abs(out - ref).max() <= 5e-2 * np.abs(ref).max()


def test_fused_ca_properties_at_the_headline_shape():
    """(B, Nq, Nkv, d, h) = (1, 32768, 196, 768, 12) in the parity-true mode, size-independent properties: (i) a 640-row slice equals the
    same rows computed alone, bit for bit (rows are independent); (ii) row-permutation equivariance, bit for bit; (iii) invariance to a
    permutation of the kv tokens up to accumulation order; (iv) 512 rows spread over the range against the CPU oracle at 1e-3."""
    m, sd = block(541)
    q, kv = dev(synth.randn((1, 32768, D), 542)), dev(synth.randn((1, 196, D), 543))
    full = run(m, q, kv, "mixed")
    assert bool(torch.isfinite(full).all())
    assert torch.equal(run(m, q[:, 1024:1664].contiguous(), kv, "mixed"), full[:, 1024:1664])
    perm = torch.randperm(32768, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    assert torch.equal(run(m, q[:, perm].contiguous(), kv, "mixed"), full[:, perm])
    kperm = torch.randperm(196, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    assert (run(m, q, kv[:, kperm].contiguous(), "mixed") - full).abs().max().item() < 5e-4
    rows = torch.arange(0, 32768, 64)
    ref = oracle_ca(q[:, rows].cpu(), kv.cpu(), sd)
    assert (full[:, rows].cpu() - ref).abs().max().item() <= TOL


@pytest.mark.parametrize("prec", ["mixed", "bf16"])
def test_fused_ca_inside_the_reference_block_golden(prec):
    """The whole VATBlock (self-attention, fused cross-attention, MLP) against the golden of the unmodified reference class at the
    BASELINE shape (256 queries x 196 patches, d = 768, h = 12): the parity-true mode stays within 1e-3 with the fp16 cross-attention
    inside; the bf16 mode within its operand-rounding bound."""
    import cases
    c = cases.VAT_BLOCK_CASES["baseline_256x196"]
    m = fusion().VATBlock(c["d"], c["h"], c["dff"], 0.1).to(DEV).eval()
    synth.load_seeded(m, c["seed"])
    q, kv = dev(synth.randn((c["B"], c["Nq"], c["d"]), c["seed"] + 1000)), dev(synth.randn((c["B"], c["Nk"], c["d"]), c["seed"] + 2000))
    m.precision = prec
    m.fused_ca_in_block = True
    with torch.no_grad():
        out = m(q, kv)
    g = golden("vat_block_baseline_256x196")
    err = golden_error(out, g)
    assert err <= (TOL if prec == "mixed" else REL_BF16 * float(np.abs(g["out"]).max())), err
