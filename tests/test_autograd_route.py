"""The training-side route of the fusion modules (lidar_vision_vqa_amd/autograd_route.py; SURVEY 8b: `.train()` and autograd are part of
the module contract -- reference call sites trainer.py:549, 581, 594): torch ops over the modules' own parameter containers.  Checked on
the CPU against the oracle (the same restatement the kernels are held to), with the reference's own test criteria for this seam: gradient
presence on every parameter, train-mode stochasticity, ValueError paths (training-test/models/test_vat_block.py:87-203,
test_vat_lidar.py:165-290).  The inference route (eval + no_grad) is NOT reachable without the GPU: that is checked too."""
import warnings

import numpy as np
import pytest
import torch

from lidar_vision_vqa_amd import _ffi, fusion, synth
from oracle import vat_oracle as VO


def sd_of(m):
    return {k: v.detach() for k, v in m.state_dict().items()}


@pytest.fixture(autouse=True)
def _quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        yield


def test_vat_block_matches_oracle_and_has_gradients():
    m = fusion.VATBlock(96, 4, 384, 0.1).eval()
    synth.load_seeded(m, 11)
    q = torch.from_numpy(synth.randn((2, 10, 96), 12)).requires_grad_(True)
    kv = torch.from_numpy(synth.randn((2, 7, 96), 13))
    y = m(q, kv)
    ref = VO.vat_block(q.detach(), kv, sd_of(m), "", 4)
    assert (y.detach() - ref).abs().max().item() < 1e-5
    y.square().mean().backward()
    assert q.grad is not None and all(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in m.parameters())


def test_vat_lidar_matches_oracle_and_has_gradients():
    m = fusion.VATLiDAR(16, 96, n_queries=12, n_layers=2, n_heads=4).eval()
    synth.load_seeded(m, 21)
    bev = torch.from_numpy(synth.randn((2, 16, 8, 8), 22))
    y = m(bev)
    assert tuple(y.shape) == (2, 12, 96)
    ref = VO.vat_lidar(bev, sd_of(m), 4)
    assert (y.detach() - ref).abs().max().item() < 2e-5
    y.sum().backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None]
    assert not missing, missing


def test_vat_vision_and_adapter_match_oracle():
    m = fusion.VATVision(64, 96, n_input_tokens=24, compression_factor=2, n_layers=1, n_heads=4, use_per_view_query=True).eval()
    synth.load_seeded(m, 31)
    kv = torch.from_numpy(synth.randn((1, 24, 64), 32))
    y = m(kv)
    assert (y.detach() - VO.vat_vision(kv, sd_of(m), 4)).abs().max().item() < 2e-5
    y.sum().backward()
    assert m.query.grad is not None and m.view_query_embed.grad is not None
    a = fusion.VisionAdapter(64, 0.1).eval()
    synth.load_seeded(a, 33)
    views = [torch.from_numpy(synth.randn((5, 64), 40 + i)) for i in range(6)]
    out = a(views)
    assert (out.detach() - VO.vision_adapter(views, sd_of(a))).abs().max().item() < 1e-5
    with pytest.raises(ValueError):
        a(views[:5])


def test_train_mode_is_stochastic_and_eval_is_not():
    m = fusion.VATBlock(96, 4, 384, 0.5)
    synth.load_seeded(m, 51)
    q, kv = torch.from_numpy(synth.randn((1, 6, 96), 52)), torch.from_numpy(synth.randn((1, 5, 96), 53))
    m.train()
    assert not torch.equal(m(q, kv), m(q, kv))
    m.eval()
    assert torch.equal(m(q, kv), m(q, kv))


def test_first_call_warns_and_inference_route_stays_kernel_only():
    from lidar_vision_vqa_amd import autograd_route as AG
    AG._warned.discard("MlpProjectorLinear")
    p = fusion.MlpProjectorLinear(8, 4).eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)
        with pytest.raises(UserWarning, match="autograd route"):
            p(torch.zeros(2, 8))
    m = fusion.VATBlock(96, 4, 384, 0.1).eval()
    with torch.no_grad(), pytest.raises(_ffi.LvqError):            # eval + no_grad = the kernels, and they do not run on CPU tensors
        m(torch.zeros(1, 4, 96), torch.zeros(1, 5, 96))
    for p_ in m.parameters():
        p_.requires_grad_(False)
    with pytest.raises(_ffi.LvqError):                             # grad mode but nothing trainable in reach: still the kernels
        m(torch.zeros(1, 4, 96), torch.zeros(1, 5, 96))
