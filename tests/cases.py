"""Shared case tables: tools/make_goldens.py (build container, imports the reference) and the parity
tests (oracle on CPU, HIP path on the GPU) iterate the same lists.  Inputs and weights are
regenerated from seeds (lidar_vision_vqa_amd.synth); tests/golden/*.npz hold reference OUTPUTS.

The sweeps mirror the reference's own test parametrisation
(src/encoder-decoder/training-test/models/test_vat_block.py:153-164, test_vat_lidar.py:247-253)
at sizes that keep the fixtures small, plus the BASELINE.json shapes (d=768, h=12).
"""

# name -> dict(d, h, dff, B, Nq, Nk, seed)
VAT_BLOCK_CASES = {
    "tiny_d96":        dict(d=96,  h=4,  dff=384,  B=2, Nq=7,   Nk=11,  seed=11),
    "d256_h8":         dict(d=256, h=8,  dff=1024, B=1, Nq=64,  Nk=256, seed=12),
    "d512_h8_b2":      dict(d=512, h=8,  dff=2048, B=2, Nq=128, Nk=512, seed=13),
    "baseline_256x196": dict(d=768, h=12, dff=3072, B=1, Nq=256, Nk=196, seed=14),
    "cfg5_64x576_b2":  dict(d=768, h=12, dff=3072, B=2, Nq=64,  Nk=576, seed=15),
    "hd112_d224_h2":   dict(d=224, h=2,  dff=896,  B=1, Nq=12,  Nk=50,  seed=16),   # head_dim 112 (896/8)
    "hd448_d896_h2":   dict(d=896, h=2,  dff=3584, B=1, Nq=12,  Nk=100, seed=17),   # reference default: vat_heads=2
}

# name -> dict(c_in, d, nq, L, h, B, H, W, seed)
VAT_LIDAR_CASES = {
    "tiny":      dict(c_in=16,  d=96,  nq=12,  L=1, h=4,  B=2, H=10, W=10, seed=21),
    "odd_hw":    dict(c_in=16,  d=96,  nq=12,  L=2, h=4,  B=1, H=9,  W=11, seed=22),   # centre pixel -> sector 1
    "c128_d256": dict(c_in=128, d=256, nq=384, L=2, h=8,  B=1, H=50, W=50, seed=23),
    "d768_h12":  dict(c_in=64,  d=768, nq=96,  L=1, h=12, B=1, H=32, W=32, seed=24),
    # the reference's own geometry: BEV [128,180,180] (precompute_bev_features.py:231-291), d = 896 (Qwen2.5-0.5B hidden)
    "ref_default": dict(c_in=128, d=896, nq=12,  L=1, h=2, B=1, H=180, W=180, seed=25),   # default_config.py:37-42 (head_dim 448)
    "ref_medium":  dict(c_in=128, d=896, nq=576, L=4, h=8, B=1, H=180, W=180, seed=26, sliced=True),   # vat_lidar.py:67-69 (head_dim 112)
}

# name -> dict(d_in, d_model, n_in, cf, L, h, B, per_view, seed)
VAT_VISION_CASES = {
    "tiny":       dict(d_in=128, d_model=96,  n_in=48, cf=2, L=1, h=4, B=2, per_view=False, seed=31),
    "per_view":   dict(d_in=128, d_model=96,  n_in=48, cf=2, L=1, h=4, B=1, per_view=True,  seed=32),
    "d256_l2":    dict(d_in=256, d_model=128, n_in=96, cf=2, L=2, h=8, B=1, per_view=True,  seed=33),
    # the reference's own geometry: 6 x 256 DeepEncoder tokens of width 2048 -> d = 896
    "ref_default": dict(d_in=2048, d_model=896, n_in=1536, cf=128, L=1, h=2, B=1, per_view=True, seed=34),   # default_config.py:45-52 (head_dim 1024)
    "ref_cf2":     dict(d_in=2048, d_model=896, n_in=1536, cf=2, L=2, h=8, B=1, per_view=True, seed=35, sliced=True),   # nq = 768, head_dim 256
}

VISION_ADAPTER_CASES = {
    "d128_hw16": dict(d_in=128, hw=16, seed=41),
    "d2048_hw8": dict(d_in=2048, hw=8, seed=42),
}

# LiDAR side: name -> dict(dist, n, seed, vsize, T, max_voxels, filters)
PILLAR_CASES = {
    "pp64_C4k":    dict(dist="C", n=4096, seed=51, T=20, max_voxels=30000, filters=[64], wseed=61),
    "pp32_64_U2k": dict(dist="U", n=2048, seed=52, T=20, max_voxels=30000, filters=[32, 64], wseed=62),
}
MEAN_CASES = {
    "mean_C8k": dict(dist="C", n=8192, seed=53, T=10, max_voxels=60000),
}

SDPA_CASES = {
    "nomask": dict(B=2, H=4, S=50, D=64, mask=False, seed=71),
    "bias":   dict(B=1, H=12, S=49, D=64, mask=True, seed=72),
}

# large outputs are stored as (first rows, every 16th row, per-row L2 norm and sum of ALL rows) -- SURVEY 8c
SLICE_ROWS = 32
SLICE_STRIDE = 16


def slice_rows(out2d):
    """out2d [rows, d] numpy/torch -> dict(head, strided, norm, rsum) covering every row."""
    import numpy as np
    a = np.asarray(out2d, dtype=np.float32)
    return dict(head=a[:SLICE_ROWS], strided=a[::SLICE_STRIDE], norm=np.sqrt((a.astype(np.float64) ** 2).sum(-1)).astype(np.float32),
                rsum=a.astype(np.float64).sum(-1).astype(np.float32))


# stand-in head (transformers.Qwen2ForCausalLM random init): SURVEY 8c
HEAD_CASE = dict(vocab=512, d=128, inter=256, n_heads=4, n_kv_heads=2, n_layers=2, rms_eps=1e-6,
                 rope_theta=1000000.0, seed=81, nq_vision=24, nq_lidar=12, n_prompt=9, n_answer=32, B=2)

# the reference decoder's geometry (Qwen2.5-0.5B: d = 896, 14 / 2 heads, inter 4864; 4 of its 24 layers, vocab cut to 8192 so the
# fixture stays small) at BASELINE configs[4]'s sequence: 576 vision + 256 LiDAR prefix tokens + prompt + 32 answer positions
HEAD_REF_CASE = dict(vocab=8192, d=896, inter=4864, n_heads=14, n_kv_heads=2, n_layers=4, rms_eps=1e-6, rope_theta=1000000.0,
                     seed=85, nq_vision=576, nq_lidar=256, n_prompt=12, n_answer=32, B=1)
