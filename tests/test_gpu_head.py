"""a15: prefix assembly + stand-in head on the GPU vs the goldens produced by replaying
validation.py:105-158 with the imported reference VAT modules and transformers' Qwen2ForCausalLM."""
import numpy as np
import pytest
import torch

import cases
from conftest import golden

pytestmark = pytest.mark.gpu

from lidar_vision_vqa_amd import synth  # noqa: E402

DEV = "cuda:0"


def build(hc, prec):
    from lidar_vision_vqa_amd import fusion, head
    d = hc["d"]
    base = head.StandInHead(hc["vocab"], d, hc["inter"], hc["n_heads"], hc["n_kv_heads"], hc["n_layers"], hc["rms_eps"],
                            hc["rope_theta"]).to(DEV).eval()
    sd = {k: torch.from_numpy(synth.seeded_array(k, tuple(v.shape), hc["seed"])) for k, v in base.state_dict().items() if k != "lm_head.weight"}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    base.load_state_dict(sd)
    vl = synth.load_seeded(fusion.VATLiDAR(16, d, hc["nq_lidar"], 1, 4).to(DEV).eval(), hc["seed"] + 1)
    va = synth.load_seeded(fusion.VisionAdapter(64, 0.1).to(DEV).eval(), hc["seed"] + 2)
    vv = synth.load_seeded(fusion.VATVision(64, d, 48, 2, 1, 4).to(DEV).eval(), hc["seed"] + 3)
    for m in (base, vl, vv):
        m.precision = prec
    return base, vl, va, vv


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-3), ("bf16", None)])   # None: 2e-2 * max|ref| (operand-rounding bound)
def test_prefix_assembly_and_answer_logits(prec, tol):
    from lidar_vision_vqa_amd import head
    hc = cases.HEAD_CASE
    g = golden("head_prefix")
    B = hc["B"]
    base, vl, va, vv = build(hc, prec)
    assert set(base.state_dict()) >= {"model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.bias", "model.norm.weight", "lm_head.weight"}
    with torch.no_grad():
        bev = torch.from_numpy(synth.randn((B, 16, 10, 10), hc["seed"] + 4)).to(DEV)
        kv = torch.stack([va([torch.from_numpy(synth.randn((8, 64), hc["seed"] + 10 + 6 * b + v)).to(DEV) for v in range(6)]) for b in range(B)])
        pl, pv = vl(bev), vv(kv)
        p_ids, a_ids = torch.from_numpy(g["p_ids"]).to(DEV), torch.from_numpy(g["a_ids"]).to(DEV)
        E = base.embed(torch.arange(4, device=DEV))
        inp, attn, labels = head.assemble_prefix(pv, pl, E, base.embed(p_ids), base.embed(a_ids), a_ids, 0.2)
        out = base(inputs_embeds=inp, attention_mask=attn, labels=labels)
    assert np.array_equal(labels.cpu().numpy(), g["labels"])                                  # integer: exact
    assert tuple(inp.shape) == g["inputs_embeds"].shape
    assert np.abs(inp.cpu().numpy() - g["inputs_embeds"]).max() < (1e-3 if prec == "bf16x3" else 2e-2)
    ans = out.logits[:, -hc["n_answer"]:].cpu().numpy()
    err = np.abs(ans - g["answer_logits"]).max()
    if tol is None:
        tol = 2e-2 * float(np.abs(g["answer_logits"]).max())
    assert err < tol, err                                                                     # north_star: answer logits within 1e-3
    assert abs(float(out.loss) - float(g["loss"])) < tol
