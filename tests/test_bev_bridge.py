"""CPU tests of the data-format rows: the numpy oracle against the reference's own torch lines run on
the CPU, and the host-side index / writer logic.  No GPU, no compute calls into the HIP library."""
import numpy as np
import pytest
import torch

from oracle import bev_oracle as BO


def synth_sparse(seed, batch, d, h, w, m, c, dup=0.5):
    """Sparse 3-D tensor rows like x_conv4 (+) 2*x_conv5 (+) 4*x_conv6 (spconv_backbone_voxelnext.py:188-191): unique
    (b,z,y,x) rows, many of which share (b,y,x)."""
    rng = np.random.default_rng(seed)
    cells = rng.choice(batch * h * w, size=max(1, int(m * (1 - dup))), replace=False)
    pick = rng.choice(cells, size=m)
    z = rng.integers(0, d, size=m)
    idx = np.stack((pick // (h * w), z, (pick // w) % h, pick % w), axis=1).astype(np.int32)
    idx = np.unique(idx, axis=0)
    idx = idx[rng.permutation(len(idx))]
    feats = rng.standard_normal((len(idx), c)).astype(np.float32)
    return feats, idx


@pytest.mark.parametrize("seed,batch,d,h,w,m,c", [(1, 2, 5, 18, 18, 400, 16), (2, 1, 2, 7, 9, 60, 3), (3, 3, 5, 45, 45, 5000, 128)])
def test_oracle_bev_out_matches_reference_torch_lines(seed, batch, d, h, w, m, c):
    feats, idx = synth_sparse(seed, batch, d, h, w, m, c)
    # spconv_backbone_voxelnext.py:150-157, verbatim semantics on CPU tensors
    features_cat = torch.from_numpy(feats)
    indices_cat = torch.from_numpy(idx)[:, [0, 2, 3]]
    indices_unique, _inv = torch.unique(indices_cat, dim=0, return_inverse=True)
    features_unique = features_cat.new_zeros((indices_unique.shape[0], features_cat.shape[1]))
    features_unique.index_add_(0, _inv, features_cat)
    of, oi, oinv = BO.bev_out(feats, idx)
    assert np.array_equal(oi, indices_unique.numpy())
    assert np.array_equal(oinv, _inv.numpy())
    assert np.allclose(of, features_unique.numpy(), rtol=1e-6, atol=1e-6)
    assert len(oi) < len(idx)                      # the z-merge really merges


def test_oracle_dense_and_height_compression():
    feats, idx = synth_sparse(5, 2, 3, 6, 7, 80, 4)
    dn = BO.dense(feats, idx, (3, 6, 7), 2)
    assert dn.shape == (2, 4, 3, 6, 7)
    for r in (0, len(idx) // 2, len(idx) - 1):
        b, z, y, x = idx[r]
        assert np.array_equal(dn[b, :, z, y, x], feats[r])
    assert np.count_nonzero(dn) <= feats.size
    hc = BO.height_compression(feats, idx, (3, 6, 7), 2)
    assert hc.shape == (2, 12, 6, 7)
    b, z, y, x = idx[0]
    assert np.array_equal(hc[b, np.arange(4) * 3 + z, y, x], feats[0])      # view(N, C*D, H, W): channel c*D + z


def test_collect_feature_tokens_and_writer(tmp_path, capsys):
    from lidar_vision_vqa_amd import bev as B
    r1, r2 = tmp_path / "a", tmp_path / "b"
    (r1 / "train").mkdir(parents=True)
    (r1 / "val" / "deep").mkdir(parents=True)
    r2.mkdir()
    rng = np.random.default_rng(0)
    arrs = {k: rng.standard_normal((4, 5, 6)).astype(np.float32) for k in ("tokA", "tokB", "tokC", "tokD")}
    B.save_bev_feature(r1 / "train" / "tokA.npy", arrs["tokA"])
    B.save_bev_feature(r1 / "val" / "deep" / "tokB.npy", torch.from_numpy(arrs["tokB"]))
    B.save_bev_feature(r2 / "tokC.npy", arrs["tokC"])
    B.save_bev_feature(r2 / "tokA.npy", arrs["tokD"])                 # duplicate token in a later root: the first root wins
    (r1 / "train" / "notes.txt").write_text("x")
    t2p = B.collect_feature_tokens([str(r1), str(tmp_path / "missing"), str(r2)])
    assert "missing" in capsys.readouterr().out                     # the absent root is named once, then skipped
    assert sorted(t2p) == ["tokA", "tokB", "tokC"]
    assert t2p["tokA"] == str(r1 / "train" / "tokA.npy")
    stored = np.load(t2p["tokB"])
    assert stored.dtype == np.float16 and stored.shape == (4, 5, 6)
    # the load side (dataset.py:139-146): torch.from_numpy(np.load(path)).float() == the oracle's restatement
    assert np.array_equal(BO.load_bev(t2p["tokB"]), torch.from_numpy(np.load(t2p["tokB"])).float().numpy())
    assert np.array_equal(BO.load_bev(t2p["tokA"]), arrs["tokA"].astype(np.float16).astype(np.float32))


def test_engine_prompt_format_and_tokenizer_match_reference_golden():
    """Host-side half of the inference engine (no GPU): prompt strings -> ids identical to what the unmodified reference
    InferenceEngine produced with the same DummyTokenizer (tests/golden/engine.npz)."""
    import os
    from lidar_vision_vqa_amd import engine, synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "engine.npz"))
    tok = synth.DummyTokenizer(512)
    q, system = "How many cars are ahead of the ego vehicle?", "You are a driving assistant."
    for tag, use_vision in (("l", False), ("v", True)):
        eng = engine.InferenceEngine(dict(tokenizer=tok, base_model=None, vat_lidar=None, vat_vision=object() if use_vision else None,
                                          device="cpu", d_model=128,
                                          config=dict(use_vision=use_vision, prefix_scale=0.2, system_prompt=system if use_vision else "")))
        assert tok.encode(eng.format_prompt(q, include_vision=use_vision)) == g[f"{tag}_prompt_ids"].tolist()
    assert tok.decode([0, 4 + 11, 3, 4 + 12]) == chr(32 + 11) + chr(32 + 12) and tok.convert_tokens_to_ids("<lidar_end>") == 3
