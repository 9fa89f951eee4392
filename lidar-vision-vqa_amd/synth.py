"""Synthetic scenes, image-patch tokens and seeded weights.

There is no network for nuScenes or checkpoints, so every test, golden fixture and bench run uses
these generators.  numpy's PCG64 `default_rng(seed)` streams are platform-stable, so the build
container (where the reference is imported to make the goldens) and the GPU box regenerate
identical inputs and weights from a seed; only outputs are stored in tests/golden/.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

# nuScenes ranges / voxel sizes used by the reference yaml configs
PC_RANGE_NUSC = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)   # dataset_configs/nuscenes_dataset.yaml:20
VOXEL_01 = (0.1, 0.1, 0.2)                               # cbgs_voxel01_res3d_centerpoint.yaml (0.1 m grid)
VOXEL_PILLAR = (0.2, 0.2, 8.0)                           # cbgs_pp_multihead.yaml:18-24


def scene_points(dist: str, n: int, seed: int) -> np.ndarray:
    """[n,4] float32 (x, y, z, intensity).

    "U": x,y ~ U(-51.2,51.2), z ~ U(-5,3) -- ~1 point per 0.1 m voxel, worst case for hashing.
    "C": 70 % ground ring (r ~ |N(0,15 m)| clipped to 50, theta ~ U, z ~ N(-1.5,0.3)), 30 % in 64
         Gaussian clusters sigma=(0.8,0.8,0.6) m centred uniformly in range -- multi-point voxels, cap hits.
    """
    rng = np.random.default_rng(seed)
    if dist == "U":
        xy = rng.uniform(-51.2, 51.2, size=(n, 2))
        z = rng.uniform(-5.0, 3.0, size=(n, 1))
    elif dist == "C":
        ng = int(round(0.7 * n))
        r = np.minimum(np.abs(rng.normal(0.0, 15.0, size=ng)), 50.0)
        th = rng.uniform(-np.pi, np.pi, size=ng)
        g = np.stack((r * np.cos(th), r * np.sin(th), rng.normal(-1.5, 0.3, size=ng)), axis=1)
        nc = n - ng
        centres = np.stack((rng.uniform(-51.2, 51.2, 64), rng.uniform(-51.2, 51.2, 64), rng.uniform(-5.0, 3.0, 64)), axis=1)
        which = rng.integers(0, 64, size=nc)
        c = centres[which] + rng.normal(0.0, 1.0, size=(nc, 3)) * np.array([0.8, 0.8, 0.6])
        pts = np.concatenate((g, c), axis=0)
        pts = pts[rng.permutation(n)]  # shuffle_points is upstream of the path (data_processor.py:95-105)
        xy, z = pts[:, :2], pts[:, 2:3]
    else:
        raise ValueError(f"unknown distribution {dist!r}")
    inten = rng.uniform(0.0, 1.0, size=(n, 1))
    return np.concatenate((xy, z, inten), axis=1).astype(np.float32)


def image_patches(n_tokens: int, d: int, seed: int) -> np.ndarray:
    """[n_tokens, d] float32 ~ N(0,1) stand-in for ViT patch tokens."""
    return np.random.default_rng(seed).standard_normal((n_tokens, d)).astype(np.float32)


def randn(shape: Tuple[int, ...], seed: int, scale: float = 1.0) -> np.ndarray:
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def seeded_array(name: str, shape: Tuple[int, ...], seed: int, dtype=np.float32) -> np.ndarray:
    """One parameter / buffer from (seed, crc32(name)) -- independent of registration order."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith("running_var"):
        return (0.5 + rng.random(shape)).astype(np.float32)
    if name.endswith("running_mean"):
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if name.endswith("weight"):          # LayerNorm / BatchNorm gamma
        return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    return (0.02 * rng.standard_normal(shape)).astype(np.float32)   # biases


def seeded_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, np.ndarray]:
    return {k: seeded_array(k, tuple(s), seed) for k, s in shapes}


def load_seeded(module, seed: int):
    """Fill a torch module (reference class or ours) in place from the seed recipe; returns module."""
    import torch
    sd = module.state_dict()
    new = {k: torch.from_numpy(seeded_array(k, tuple(v.shape), seed)).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)
    return module


class DummyTokenizer:
    """Character-level stand-in for the Qwen tokenizer (fetched by name in the reference, unavailable offline) with the four
    members inference_engine.py touches: __call__(prompt, return_tensors="pt", add_special_tokens=False)["input_ids"],
    convert_tokens_to_ids, decode(ids, skip_special_tokens=True), pad_token_id / eos_token_id.  Ids 0..3 are the special
    tokens <vision_start> <vision_end> <lidar_start> <lidar_end> (the rows assemble_prefix uses); other characters map
    to 4 + ord(c) % (vocab - 4)."""
    SPECIALS = ("<vision_start>", "<vision_end>", "<lidar_start>", "<lidar_end>")

    def __init__(self, vocab: int, eos_token_id=None, pad_token_id: int = 0):
        self.vocab = int(vocab)
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id

    def convert_tokens_to_ids(self, tok: str) -> int:
        return self.SPECIALS.index(tok)

    def encode(self, text: str):
        ids, i = [], 0
        while i < len(text):
            for k, sp in enumerate(self.SPECIALS):
                if text.startswith(sp, i):
                    ids.append(k)
                    i += len(sp)
                    break
            else:
                ids.append(4 + ord(text[i]) % (self.vocab - 4))
                i += 1
        return ids

    def __call__(self, text: str, return_tensors: str = "pt", add_special_tokens: bool = False):
        import torch
        return {"input_ids": torch.tensor([self.encode(text)], dtype=torch.long)}

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        out = []
        for t in [int(v) for v in ids]:
            if t < 4:
                if not skip_special_tokens:
                    out.append(self.SPECIALS[t])
            else:
                out.append(chr(32 + (t - 4) % 95))          # printable ASCII
        return "".join(out)
