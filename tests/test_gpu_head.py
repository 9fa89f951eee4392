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


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-3), ("bf16", None)])
def test_head_reference_geometry(prec, tol):
    """The reference decoder's geometry (Qwen2.5-0.5B: d = 896, 14 / 2 heads, head_dim 64, inter 4864; 4 layers, vocab 8192) on
    BASELINE configs[4]'s sequence -- 576 vision + 256 LiDAR prefix tokens, prompt, 32 answer positions (L = 880): answer logits
    within 1e-3 of transformers' Qwen2 (golden head_ref_prefix: head slice elementwise, log-sum-exp and L2 norm of every row,
    arg-max ids), labels exact, loss."""
    from lidar_vision_vqa_amd import head
    hc = cases.HEAD_REF_CASE
    g = golden("head_ref_prefix")
    B, d = hc["B"], hc["d"]
    base = head.StandInHead(hc["vocab"], d, hc["inter"], hc["n_heads"], hc["n_kv_heads"], hc["n_layers"], hc["rms_eps"],
                            hc["rope_theta"]).to(DEV).eval()
    sd = {k: torch.from_numpy(synth.seeded_array(k, tuple(v.shape), hc["seed"])) for k, v in base.state_dict().items() if k != "lm_head.weight"}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    base.load_state_dict(sd)
    base.precision = prec
    pl = torch.from_numpy(synth.randn((B, hc["nq_lidar"], d), hc["seed"] + 1)).to(DEV)
    pv = torch.from_numpy(synth.randn((B, hc["nq_vision"], d), hc["seed"] + 2)).to(DEV)
    p_ids, a_ids = torch.from_numpy(g["p_ids"]).to(DEV), torch.from_numpy(g["a_ids"]).to(DEV)
    with torch.no_grad():
        E = base.embed(torch.arange(4, device=DEV))
        inp, attn, labels = head.assemble_prefix(pv, pl, E, base.embed(p_ids), base.embed(a_ids), a_ids, 0.2)
        out = base(inputs_embeds=inp, attention_mask=attn, labels=labels)
    assert inp.shape[1] == 880 and np.array_equal(labels.cpu().numpy(), g["labels"])
    assert np.abs(inp.double().sum(-1).cpu().numpy() - g["inputs_embeds_sum"]).max() < 1e-3
    al = out.logits[:, -hc["n_answer"]:].cpu()
    if tol is None:
        tol = 2e-2 * float(np.abs(g["answer_logits_head"]).max())
    assert np.abs(al[:, :, :512].numpy() - g["answer_logits_head"]).max() < tol
    assert np.abs(torch.logsumexp(al.double(), -1).numpy() - g["answer_lse"]).max() < tol
    assert np.abs(al.double().pow(2).sum(-1).sqrt().numpy() - g["answer_row_norm"]).max() < tol * np.sqrt(hc["vocab"])
    if prec == "bf16x3":
        assert np.array_equal(al.argmax(-1).numpy().astype(np.int32), g["answer_argmax"])
    assert abs(float(out.loss) - float(g["loss"])) < tol


@pytest.mark.parametrize("prec,tol", [("bf16x3", 1e-3), ("bf16", None)])
def test_greedy_generate_vs_transformers(prec, tol):
    """SURVEY 8f row f4 (inference_engine.py:283-296): greedy decoding with the KV cache == transformers'
    `generate(inputs_embeds=, do_sample=False)` on the seeded Qwen2 stand-in: token ids exactly (the golden top-1 / top-2
    margin is 0.11), per-step logits within the precision mode's bound, EOS / pad handling, argmax kernel."""
    from lidar_vision_vqa_amd import head, ops, _ffi
    hc = cases.HEAD_CASE
    g, gp = golden("head_generate"), golden("head_prefix")
    base = build(hc, prec)[0]
    inp = torch.from_numpy(gp["inputs_embeds"])[:, :-hc["n_answer"]].contiguous().to(DEV)
    attn = torch.ones(inp.shape[:2], dtype=torch.long, device=DEV)
    n = g["ids"].shape[1]
    ids, scores = base.generate(inputs_embeds=inp, attention_mask=attn, max_new_tokens=n, do_sample=False, num_beams=1,
                                pad_token_id=0, eos_token_id=None, output_scores=True)
    assert ids.dtype == torch.int64 and tuple(ids.shape) == g["ids"].shape
    if tol is None:
        tol = 2e-2 * float(np.abs(g["scores"]).max())
    assert np.abs(scores.cpu().numpy() - g["scores"]).max() < tol
    assert np.array_equal(ids.cpu().numpy(), g["ids"])
    # the cached decode steps agree with a full forward over prompt + generated tokens (no cache)
    with torch.no_grad():
        full = torch.cat((inp, base.embed(ids[:, :-1])), dim=1)
        ref = base(inputs_embeds=full).logits[:, inp.shape[1] - 1:]
    assert (ref - scores).abs().max().item() < (1e-3 if prec == "bf16x3" else tol)
    # native decode-step runtime (default) == the Python-driven loop, bit for bit
    import os
    os.environ["LVQ_DECODE_PYTHON"] = "1"
    try:
        ids_py, scores_py = base.generate(inputs_embeds=inp, attention_mask=attn, max_new_tokens=n, do_sample=False, output_scores=True)
    finally:
        del os.environ["LVQ_DECODE_PYTHON"]
    assert torch.equal(ids_py, ids) and torch.equal(scores_py, scores)
    ids2 = base.generate(inputs_embeds=inp, attention_mask=attn, max_new_tokens=n, do_sample=False, pad_token_id=0,
                         eos_token_id=int(g["eos"]))
    assert np.array_equal(ids2.cpu().numpy(), g["ids_eos"])
    with pytest.raises(_ffi.LvqError):
        base.generate(inputs_embeds=inp, num_beams=2)                      # beam search is not built
    x = torch.tensor([[1.0, 5.0, 5.0, -2.0], [-3.0, -3.0, -7.0, -3.0]], device=DEV)
    assert ops.argmax_rows(x).cpu().tolist() == [1, 0]                     # first maximum
    big = torch.randn(7, 151936, device=DEV)
    assert torch.equal(ops.argmax_rows(big), big.argmax(-1))               # Qwen2.5 vocabulary width


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_inference_engine_vs_unmodified_reference(prec, tmp_path):
    """engine.InferenceEngine (format_prompt / process_lidar / build_inputs_embeds / generate / generate_batch)
    against goldens produced by the UNMODIFIED reference InferenceEngine class on CPU (tools/make_engine_golden.py)."""
    from lidar_vision_vqa_amd import engine, bev as BV
    hc = cases.HEAD_CASE
    g = golden("engine")
    base, vl, va, vv = build(hc, prec)
    tok = synth.DummyTokenizer(hc["vocab"])
    question, system = "How many cars are ahead of the ego vehicle?", "You are a driving assistant."
    tol = 1e-3 if prec == "bf16x3" else 2e-2
    views = [torch.from_numpy(synth.randn((8, 64), hc["seed"] + 50 + v)) for v in range(6)]
    bev = synth.randn((16, 10, 10), hc["seed"] + 40)
    for tag, use_vision in (("l", False), ("v", True)):
        models = dict(tokenizer=tok, base_model=base, vat_lidar=vl, vat_vision=vv if use_vision else None, vision_adapter=va,
                      multiview_tokens_fn=lambda sample_token: views, device=torch.device(DEV), d_model=hc["d"],
                      config=dict(use_vision=use_vision, prefix_scale=0.2, system_prompt=system if use_vision else ""))
        eng = engine.InferenceEngine(models)
        prompt = eng.format_prompt(question, include_vision=use_vision)
        assert tok.encode(prompt) == g[f"{tag}_prompt_ids"].tolist()
        lp = eng.process_lidar(torch.from_numpy(bev))
        vp = eng.process_vision("sample-token") if use_vision else None
        emb, attn = eng.build_inputs_embeds(prompt, lp, vp)
        assert tuple(emb.shape) == g[f"{tag}_inputs_embeds"].shape and bool((attn == 1).all())
        assert np.abs(emb.cpu().numpy() - g[f"{tag}_inputs_embeds"]).max() < tol
        n = g[f"{tag}_ids"].shape[1]
        want = tok.decode(g[f"{tag}_ids"][0]).strip()
        ans = eng.generate(question, bev, sample_token="sample-token" if use_vision else None, max_new_tokens=n, do_sample=False)
        assert ans == want and len(ans) == n          # the reference itself returns "" here (slice quirk, see engine.py)
    # stored fp16 BEV by path (np.load + .float() in the reference), and the batch form
    p = tmp_path / "tok.npy"
    BV.save_bev_feature(p, bev)
    eng = engine.InferenceEngine(dict(tokenizer=tok, base_model=base, vat_lidar=vl, device=torch.device(DEV), d_model=hc["d"],
                                      config=dict(use_vision=False, prefix_scale=0.2)))
    a = eng.generate_batch([question, question], [str(p), bev.astype(np.float16)], max_new_tokens=4, do_sample=False)
    assert len(a) == 2 and a[0] == a[1] and len(a[0]) == 4
    assert int(g["l_reference_answer_is_empty"]) == 1


@pytest.mark.parametrize("dh,nkv,H,Hk", [(64, 300, 4, 2), (64, 1, 2, 2), (128, 1237, 6, 2), (64, 4099, 14, 2)])
@pytest.mark.parametrize("split", [False, True])
def test_decode_attention_one_query(dh, nkv, H, Hk, split):
    """lvq_attention_bf16 with one query per sequence (the decode-step kernel, head_dim 64 / 128, grouped KV heads, strided cache
    layout) against softmax(q k^T / sqrt(dh)) v in fp32 on the operands the kernel sees."""
    from lidar_vision_vqa_amd import ops
    B, lmax = 2, nkv + 5
    g = torch.Generator().manual_seed(dh * 1000 + nkv)
    q = torch.randn(B, H * dh, generator=g)
    kc = torch.randn(B, lmax, Hk * dh, generator=g)
    vc = torch.randn(B, lmax, Hk * dh, generator=g)
    rnd = (lambda t: t) if split else (lambda t: t.to(torch.bfloat16).float())
    qd, kd, vd = (ops.cast(t.reshape(-1, t.shape[-1]).contiguous().to(DEV), split) for t in (q, kc, vc))
    out = ops.attention(qd, kd, vd, batch=B, n_heads=H, n_kv_heads=Hk, nq=1, nkv=nkv, dh=dh, q_strides=(H * dh, H * dh, dh),
                        k_strides=(lmax * Hk * dh, Hk * dh, dh), v_strides=(lmax * Hk * dh, Hk * dh, dh), scale=1.0 / dh ** 0.5)
    got = out[0].float() + (out[1].float() if split else 0)
    qq = rnd(q).view(B, H, 1, dh)
    kk = rnd(kc)[:, :nkv].view(B, nkv, Hk, dh).permute(0, 2, 1, 3).repeat_interleave(H // Hk, dim=1)
    vv = rnd(vc)[:, :nkv].view(B, nkv, Hk, dh).permute(0, 2, 1, 3).repeat_interleave(H // Hk, dim=1)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) / dh ** 0.5, dim=-1) @ vv).reshape(B, H * dh)
    assert (got.cpu() - ref).abs().max().item() < (2e-5 if split else 8e-3)


@pytest.mark.parametrize("m,n,k", [(1, 896, 896), (2, 1152, 128), (3, 100, 64), (5, 4864, 896), (8, 9728, 256), (7, 65, 4864)])
@pytest.mark.parametrize("split", [False, True])
def test_skinny_gemv_matches_tile_gemm(m, n, k, split, tune):
    """lvq_gemm_bf16 with M <= 8 (k_gemv: one pass over W, fp32 FMAs) against the MFMA tile kernels on the same operands: every
    epilogue option (bias, GELU, alpha, residual, row table; fp32 and bf16 / lo outputs), narrow and wide N (1 or 4 rows per wave),
    N not a multiple of the rows per wave."""
    from lidar_vision_vqa_amd import ops, _ffi as F
    import ctypes
    g = torch.Generator().manual_seed(m * 100000 + n + k)
    a = ops.cast((torch.randn(m, k, generator=g)).to(DEV), split)
    w = ops.cast((torch.randn(n, k, generator=g) * 0.1).to(DEV), split)
    bias = torch.randn(n, generator=g).to(DEV)
    res = torch.randn(m, n, generator=g).to(DEV)
    tab = torch.randn(3, n, generator=g).to(DEV)

    def run(gelu, alpha, use_res, use_tab):
        c32 = torch.empty((m, n), dtype=torch.float32, device=DEV)
        ch = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
        cl = torch.empty((m, n), dtype=torch.bfloat16, device=DEV) if split else None
        rc = F.lib().lvq_gemm_bf16(F.ptr(a[0]), F.ptr(a[1]), F.ptr(w[0]), F.ptr(w[1]), F.ptr(bias), F.ptr(res if use_res else None),
                                   F.ptr(tab if use_tab else None), F.i64(3 if use_tab else 0), F.cfloat(alpha), F.cint(1 if gelu else 0),
                                   F.i64(m), F.cint(n), F.cint(k), F.i64(k), F.i64(k), F.i64(n), F.cint(1), F.i64(0), F.i64(0), F.i64(0),
                                   F.ptr(c32), F.ptr(ch), F.ptr(cl), F.stream_ptr(torch.device(DEV)))
        F.check(rc, "lvq_gemm_bf16")
        torch.cuda.synchronize()
        return c32.cpu(), ch.float().cpu() + (cl.float().cpu() if split else 0)

    for gelu, alpha, use_res, use_tab in [(False, 1.0, False, False), (True, 0.5, True, True), (False, 2.0, True, False)]:
        got32, got16 = run(gelu, alpha, use_res, use_tab)
        tune(gemm_no_gemv=1)
        ref32, ref16 = run(gelu, alpha, use_res, use_tab)
        tune(gemm_no_gemv=0)
        scale = float(ref32.abs().max()) + 1e-6
        assert (got32 - ref32).abs().max().item() < 2e-5 * scale * max(1.0, k / 256)        # same products, different summation order
        assert (got16 - ref16).abs().max().item() < (2e-5 if split else 8e-3) * scale


def _hf_warped_probs(row: torch.Tensor, temperature, top_k, top_p) -> torch.Tensor:
    """The distribution transformers samples from: its own logits processors applied to one row (CPU)."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    s = row[None].clone()
    ids = torch.zeros((1, 1), dtype=torch.long)
    s = TemperatureLogitsWarper(temperature)(ids, s)
    if top_k and top_k > 0:
        s = TopKLogitsWarper(top_k=top_k)(ids, s)
    if top_p < 1.0:
        s = TopPLogitsWarper(top_p=top_p)(ids, s)
    return torch.softmax(s, dim=-1)[0]


@pytest.mark.parametrize("vocab,temperature,top_k,top_p", [(400, 0.7, 50, 0.9), (1000, 1.3, 7, 1.0), (300, 0.7, 0, 0.5), (64, 1.0, 64, 0.95),
                                                           (5000, 0.25, 1000, 0.999)])
def test_sample_rows_distribution(vocab, temperature, top_k, top_p):
    """lvq_sample_rows draws from the distribution transformers' warpers define (the reference's default call: temperature 0.7,
    top_k 50, top_p 0.9, inference_engine.py:236-240): support exactly inside the kept set, frequencies within 5 sigma of the
    warped probabilities over 120 000 draws, and as close to them as torch.multinomial's own draws of the same size are."""
    from lidar_vision_vqa_amd import ops
    n = 120000
    g = torch.Generator().manual_seed(vocab + top_k)
    row = torch.randn(vocab, generator=g) * 2.0
    p = _hf_warped_probs(row, temperature, top_k, top_p).double()
    logits = row.to(DEV)[None].expand(n, vocab).contiguous()
    gen = torch.Generator(device=DEV).manual_seed(1234)
    ids = ops.sample_rows(logits, temperature, top_k, top_p, gen).cpu()
    assert ids.dtype == torch.int64 and int(ids.min()) >= 0 and int(ids.max()) < vocab
    freq = torch.bincount(ids, minlength=vocab).double() / n
    assert bool((freq[p == 0] == 0).all()), "a token outside transformers' kept set was drawn"
    sigma = torch.sqrt(p * (1 - p) / n)
    assert float(((freq - p).abs() / (sigma + 1.0 / n)).max()) < 5.5
    ref = torch.bincount(torch.multinomial(p.float(), n, replacement=True, generator=g), minlength=vocab).double() / n
    tv_ours, tv_torch = 0.5 * float((freq - p).abs().sum()), 0.5 * float((ref - p).abs().sum())
    assert tv_ours < 2.0 * tv_torch + 1e-3, (tv_ours, tv_torch)
    # same seed -> same draws; u -> id is monotone in the rank order (inverse CDF)
    assert torch.equal(ops.sample_rows(logits, temperature, top_k, top_p, torch.Generator(device=DEV).manual_seed(1234)).cpu(), ids)


def test_sample_rows_edge_cases():
    from lidar_vision_vqa_amd import ops, _ffi
    # top_k = 1 is greedy; ties at the k-th value are all kept (transformers removes only scores < the k-th largest)
    x = torch.randn(9, 151936, device=DEV)                             # Qwen2.5 vocabulary width
    assert torch.equal(ops.sample_rows(x, 0.7, 1, 0.9), x.argmax(-1))
    ids = ops.sample_rows(x, 0.7, 50, 0.9)
    top50 = x.topk(50, dim=-1).indices
    assert bool((ids[:, None] == top50).any(-1).all())
    flat = torch.zeros(20000, 300, device=DEV)                          # every logit ties: top_k = 5 keeps all 300
    ids = ops.sample_rows(flat, 1.0, 5, 1.0, torch.Generator(device=DEV).manual_seed(3)).cpu()
    cnt = torch.bincount(ids, minlength=300)
    assert int(cnt.min()) > 20 and int(cnt.max()) < 140
    # more survivors than the kernel's 2048 candidate slots (a constant row of 5000 logits: every logit ties at the k-th value, plus three
    # larger ones): the larger logits are never displaced by ties, the ties are kept in ascending index order (ids 0 .. 2044), and the draw
    # is reproducible under a fixed generator (collection order used to be the atomics' arrival order)
    wide = torch.zeros(4000, 5000, device=DEV)
    wide[:, [4100, 4500, 4999]] = 1.0
    g1 = ops.sample_rows(wide, 1.0, 50, 1.0, torch.Generator(device=DEV).manual_seed(11)).cpu()
    g2 = ops.sample_rows(wide, 1.0, 50, 1.0, torch.Generator(device=DEV).manual_seed(11)).cpu()
    assert torch.equal(g1, g2)
    big = (g1 == 4100) | (g1 == 4500) | (g1 == 4999)
    assert bool(((g1 < 2045) | big).all()) and int(big.sum()) > 0 and int((g1[~big]).max()) > 1900
    frac = float(big.float().mean())                                    # 3 e / (3 e + 2045) = 0.398 %
    assert 0.001 < frac < 0.01, frac
    with pytest.raises(_ffi.LvqError):
        ops.sample_rows(x, 0.7, 0, 0.9)                                 # top-k disabled on a 151 936-wide row: not supported
    with pytest.raises(_ffi.LvqError):
        ops.sample_rows(x, 0.0, 50, 0.9)


def test_engine_default_call_samples():
    """The reference's DEFAULT `engine.generate(question, bev)` (do_sample=True, temperature 0.7, top_k 50, top_p 0.9) runs on
    the drop-in, is reproducible under a seeded generator, and collapses to the greedy answer at top_k = 1."""
    from lidar_vision_vqa_amd import engine
    hc = cases.HEAD_CASE
    base, vl, va, vv = build(hc, "bf16x3")
    tok = synth.DummyTokenizer(hc["vocab"])
    eng = engine.InferenceEngine(dict(tokenizer=tok, base_model=base, vat_lidar=vl, device=torch.device(DEV), d_model=hc["d"],
                                      config=dict(use_vision=False, prefix_scale=0.2)))
    bev = synth.randn((16, 10, 10), hc["seed"] + 40)
    q = "How many cars are ahead of the ego vehicle?"
    a = eng.generate(q, bev, max_new_tokens=12, generator=torch.Generator(device=DEV).manual_seed(7))
    b = eng.generate(q, bev, max_new_tokens=12, generator=torch.Generator(device=DEV).manual_seed(7))
    assert isinstance(a, str) and a == b
    outs = {eng.generate(q, bev, max_new_tokens=12, generator=torch.Generator(device=DEV).manual_seed(s)) for s in range(6)}
    assert len(outs) > 1                                                # it does sample
    greedy = eng.generate(q, bev, max_new_tokens=12, do_sample=False)
    assert eng.generate(q, bev, max_new_tokens=12, top_k=1) == greedy
    assert len(eng.generate_batch([q, q], [bev, bev], max_new_tokens=3)) == 2
    with pytest.raises(Exception):
        eng.generate(q, bev, num_beams=4)
