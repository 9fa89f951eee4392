"""GPU parity tests for (1) the voxeliser configuration the reference actually runs -- VoxelNeXt,
tools/cfgs/nuscenes_models/cbgs_voxel0075_voxelnext.yaml:6,60-66 selected by get-data/precompute_bev_features.py:40:
range +-54 m, VOXEL_SIZE (0.075, 0.075, 0.2) (0.075 is not an fp32 number), grid 1440 x 1440 x 40, T = 10, 120 000 / 160 000
voxels -- incl. points ON cell boundaries and batches large enough to cross the hashed -> slab-binned -> global-hash key-space
fallbacks, and (2) workspace / output guard bands: the C ABI must not touch a byte outside the sizes it reports."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lidar_vision_vqa_amd import _ffi as F  # noqa: E402
from lidar_vision_vqa_amd import synth  # noqa: E402
from oracle import lidar_oracle as LO  # noqa: E402

DEV = "cuda:0"
RNG_VN = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]           # cbgs_voxel0075_voxelnext.yaml:6
VS_VN = (0.075, 0.075, 0.2)                               # :61
RNG_NUSC = list(synth.PC_RANGE_NUSC)


def L():
    from lidar_vision_vqa_amd import lidar
    return lidar


def voxelnext_scene(n, seed, boundary=0):
    """Dist-C cloud stretched to +-54 m; `boundary` extra points sit exactly on cell edges lo + k * 0.075 (as fp32 computes
    them), one ulp below and one ulp above, and on the range limits themselves."""
    pts = synth.scene_points("C", n, seed)
    pts[:, :2] *= np.float32(54.0 / 51.2)
    if boundary:
        rng = np.random.default_rng(seed + 7)
        k = rng.integers(0, 1441, size=(boundary, 2))
        lo, vs = np.float32(-54.0), np.float32(0.075)
        edge = (lo + k.astype(np.float32) * vs).astype(np.float32)
        # the same edges as the double-precision product rounds them (what a sensor driver would emit)
        edge2 = (-54.0 + k * 0.075).astype(np.float32)
        variants = [edge, np.nextafter(edge, np.float32(-np.inf)), np.nextafter(edge, np.float32(np.inf)), edge2]
        xy = np.concatenate(variants)
        z = rng.uniform(-5.0, 3.0, size=(len(xy), 1)).astype(np.float32)
        zk = (np.float32(-5.0) + rng.integers(0, 41, size=(len(xy), 1)).astype(np.float32) * np.float32(0.2)).astype(np.float32)
        z[::3] = zk[::3]
        extra = np.concatenate((xy, z, rng.random((len(xy), 1)).astype(np.float32)), axis=1).astype(np.float32)
        pts = np.concatenate((pts, extra))
        pts = pts[rng.permutation(len(pts))]
    return np.ascontiguousarray(pts, dtype=np.float32)


@pytest.mark.parametrize("path", ["hashed", "binned", "legacy"])
@pytest.mark.parametrize("n,mv", [(34000, 120000), (120000, 160000), (300000, 120000)])
def test_hard_voxelnext_config_vs_oracle(n, mv, path, tune):
    """Single scene, every implementation, bit for bit (n = 300 000 > 120 000 occupied cells: the max_voxels cap is hit)."""
    tune(voxel_path={"hashed": 0, "binned": 1, "legacy": 2}[path])
    pts = voxelnext_scene(n, 4000 + n, boundary=2000)
    assert L().grid_size_from(RNG_VN, VS_VN).tolist() == [1440, 1440, 40]
    ov, oc, on = LO.VoxelGenerator(VS_VN, RNG_VN, 4, 10, mv).generate(pts)
    gv, gc, gn = L().VoxelGeneratorWrapper(VS_VN, RNG_VN, 4, 10, mv).generate(pts)
    assert gc.shape == oc.shape and np.array_equal(gc, oc)
    assert np.array_equal(gn, on)
    assert np.array_equal(gv.view(np.uint32), ov.view(np.uint32))
    if n == 300000:
        assert len(on) == mv


@pytest.mark.parametrize("n_scenes", [3, 25, 26, 40])
def test_hard_voxelnext_batches_cross_the_keyspace_fallbacks(n_scenes):
    """1440 * 1440 * 40 = 82.9 M cells per scene: 25 scenes stay below the hashed path's 2^31 key space, 26 and 40 cross it
    (slab-binned / global-hash fallbacks) -- same per-scene oracle result either way, incl. an empty scene and a one-point one."""
    lid = L()
    sizes = [9000 if s % 5 else 20000 for s in range(n_scenes)]
    sizes[1], sizes[2] = 0, 1
    scenes = [voxelnext_scene(sz, 5000 + s, boundary=50 if sz > 1 else 0) if sz else np.zeros((0, 4), np.float32)
              for s, sz in enumerate(sizes)]
    gen = lid.VoxelGeneratorWrapper(VS_VN, RNG_VN, 4, 10, 120000)
    bd = lid.voxelize_batch(gen, [torch.from_numpy(x).to(DEV) for x in scenes])
    svo = bd["scene_voxel_off"].cpu().numpy()
    co, num, vox = bd["voxel_coords"].cpu().numpy(), bd["voxel_num_points"].cpu().numpy(), bd["voxels"].cpu().numpy()
    og = LO.VoxelGenerator(VS_VN, RNG_VN, 4, 10, 120000)
    for s, sc in enumerate(scenes):
        ov, oc, on = og.generate(sc)
        a, b = svo[s], svo[s + 1]
        assert b - a == len(on), (s, b - a, len(on))
        assert np.array_equal(co[a:b, 1:], oc) and (co[a:b, 0] == s).all(), s
        assert np.array_equal(num[a:b], on), s
        assert np.array_equal(vox[a:b].view(np.uint32), ov.view(np.uint32)), s


@pytest.mark.parametrize("bs,n", [(1, 34000), (4, 120000), (25, 12000)])
def test_dynamic_voxelnext_config_vs_oracle(bs, n):
    """DynamicMeanVFE's unique step (dynamic_mean_vfe.py:53-72) on the 0.075 m grid; 25 scenes = keys up to 2.07e9 (< 2^31)."""
    lid = L()
    per = [voxelnext_scene(n, 6000 + s, boundary=200) for s in range(bs)]
    bpts = np.concatenate([np.pad(p, ((0, 0), (1, 0)), constant_values=s) for s, p in enumerate(per)]).astype(np.float32)
    grid = LO.grid_size(RNG_VN, VS_VN)
    o = LO.dynamic_voxelize(bpts, RNG_VN, VS_VN, grid, 3)
    dv = lid._dynamic_voxelize(torch.from_numpy(bpts).to(DEV), bs, RNG_VN, VS_VN, grid, 3)
    m, nvalid = dv["counts"].cpu().tolist()
    assert m == len(o["unq_key"]) and nvalid == int(o["keep"].sum())
    assert np.array_equal(dv["unq_key"][:m].cpu().numpy(), o["unq_key"])
    assert np.array_equal(dv["unq_cnt"][:m].cpu().numpy().astype(np.int64), o["unq_cnt"])
    inv = dv["inv"].cpu().numpy()
    assert np.array_equal(inv >= 0, o["keep"])
    assert np.array_equal(inv[o["keep"]].astype(np.int64), o["unq_inv"])
    assert np.array_equal(dv["coords"][:m].cpu().numpy(), LO._decode_coords(o["unq_key"], grid, 3))
    if bs == 25:
        assert int(o["unq_key"].max()) > 2 ** 30
    # hard-vs-dynamic identity: same occupied cells per scene, counts = min(count, T) when no voxel cap triggers
    hard = lid.voxelize_batch(lid.VoxelGeneratorWrapper(VS_VN, RNG_VN, 4, 10, 10 ** 6), [torch.from_numpy(p).to(DEV) for p in per])
    assert hard["voxels"].shape[0] == m
    hc = hard["voxel_coords"].cpu().numpy().astype(np.int64)
    hkey = ((hc[:, 0] * 1440 + hc[:, 3]) * 1440 + hc[:, 2]) * 40 + hc[:, 1]
    order = np.argsort(hkey)
    assert np.array_equal(hkey[order].astype(np.int32), o["unq_key"])
    assert np.array_equal(hard["voxel_num_points"].cpu().numpy()[order], np.minimum(o["unq_cnt"], 10))


def test_dynamic_voxelnext_26_scenes_is_refused():
    """26 * 82.9 M cells >= 2^31: the reference wraps int32 silently; the ABI returns LVQ_EOVERFLOW."""
    with pytest.raises(F.LvqError):
        L()._dynamic_voxelize(torch.zeros((8, 5), device=DEV), 26, RNG_VN, VS_VN, [1440, 1440, 40], 3)


# ---------------------------------------------------------------------------------------------------------------------------
# guard bands: every buffer handed to the ABI is allocated at EXACTLY the size the ABI asks for, with a patterned band in
# front of and behind it; after the call the bands must be untouched
# ---------------------------------------------------------------------------------------------------------------------------
BAND = 1 << 16
PATTERN = 0xA5


class Banded:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.buf = torch.full((self.nbytes + 2 * BAND,), PATTERN, dtype=torch.uint8, device=DEV)

    @property
    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + BAND)

    def view(self, dtype, shape):
        return self.buf[BAND:BAND + self.nbytes].view(dtype).view(*shape)

    def intact(self):
        return bool((self.buf[:BAND] == PATTERN).all()) and bool((self.buf[BAND + self.nbytes:] == PATTERN).all())


def _hot_scene(seed):
    """Ordinary cloud + cells with 1500 / 2600 / 5000 points: slab regions overflow -> overflow list + global ranking."""
    rng = np.random.default_rng(seed)
    base = synth.scene_points("C", 30000, seed)
    hot = []
    for (cx, cy), k in zip([(3.03, -7.01), (-20.55, 11.11), (40.0, 40.0)], [1500, 2600, 5000]):
        hot.append(np.concatenate((cx + rng.uniform(0.0, 0.09, (k, 1)), cy + rng.uniform(0.0, 0.09, (k, 1)),
                                   rng.uniform(-1.19, -1.01, (k, 1)), rng.random((k, 1))), axis=1).astype(np.float32))
    pts = np.concatenate([base] + hot)
    return np.ascontiguousarray(pts[rng.permutation(len(pts))], dtype=np.float32)


def _one_cell_scene():
    rng = np.random.default_rng(3)
    return np.concatenate((rng.uniform(0.0, 0.09, (20000, 2)), rng.uniform(-0.19, -0.01, (20000, 1)), rng.random((20000, 1))),
                          axis=1).astype(np.float32)


GUARD_CASES = {
    "ordinary": lambda: [synth.scene_points("C", 32768, 1003), synth.scene_points("U", 4097, 11)],
    "hot_cells": lambda: [_hot_scene(77), np.zeros((0, 4), np.float32), _hot_scene(78)],
    "one_cell": lambda: [_one_cell_scene()],
    "tiny": lambda: [synth.scene_points("U", 1, 5), synth.scene_points("U", 63, 6)],
}


def _call_hard(scenes, vs, rng, T, mv, mean, ws_delta=0, path_env=None):
    lib = F.lib()
    S = len(scenes)
    pts = torch.from_numpy(np.concatenate(scenes)).to(DEV)
    n = pts.shape[0]
    off = torch.tensor(np.concatenate(([0], np.cumsum([len(x) for x in scenes]))), dtype=torch.int32, device=DEV)
    grid = L().grid_size_from(rng, vs).astype(np.int32).tolist()
    cap = max(1, min(n, S * mv))
    vox = Banded(cap * (1 if mean else T) * 4 * 4)
    co, num, svo = Banded(cap * 16), Banded(cap * 4), Banded((S + 1) * 4)
    ws = Banded(lib.lvq_voxelize_hard_workspace_bytes(F.i64(n), F.cint(S)) + ws_delta)
    st = F.stream_ptr(torch.device(DEV))
    rngf, vsf = F.f32x([float(np.float32(v)) for v in rng]), F.f32x([float(np.float32(v)) for v in vs])
    if mean:
        rc = lib.lvq_voxelize_mean(F.ptr(pts), F.ptr(off), F.i64(n), F.cint(S), F.cint(4), rngf, vsf, F.i32x(grid), F.cint(T), F.cint(mv),
                                   F.i64(cap), vox.ptr, co.ptr, num.ptr, svo.ptr, ws.ptr, F.csize(ws.nbytes), st)
    else:
        rc = lib.lvq_voxelize_hard(F.ptr(pts), F.ptr(off), F.i64(n), F.cint(S), F.cint(4), rngf, vsf, F.i32x(grid), F.cint(T), F.cint(mv),
                                   F.cint(0), F.i64(cap), vox.ptr, co.ptr, num.ptr, svo.ptr, ws.ptr, F.csize(ws.nbytes), st)
    torch.cuda.synchronize()
    return rc, dict(vox=vox, co=co, num=num, svo=svo, ws=ws), cap


@pytest.mark.parametrize("path", ["hashed", "binned", "legacy"])
@pytest.mark.parametrize("case", list(GUARD_CASES))
@pytest.mark.parametrize("vs,T,mv", [(synth.VOXEL_01, 10, 60000), (synth.VOXEL_PILLAR, 20, 30000)])
def test_hard_voxelizer_guard_bands(case, vs, T, mv, path, tune):
    tune(voxel_path={"hashed": 0, "binned": 1, "legacy": 2}[path])
    scenes = GUARD_CASES[case]()
    rc, bufs, cap = _call_hard(scenes, vs, RNG_NUSC, T, mv, mean=False)
    assert rc == 0, F.lib().lvq_strerror(rc)
    for name, b in bufs.items():
        assert b.intact(), f"{name}: bytes outside the reported size were written ({case}, {path})"
    # and the exact-size call still computes the right thing
    svo = bufs["svo"].view(torch.int32, (len(scenes) + 1,)).cpu().numpy()
    num = bufs["num"].view(torch.int32, (cap,)).cpu().numpy()
    a = 0
    for s, sc in enumerate(scenes):
        _, oc, on = LO.VoxelGenerator(vs, RNG_NUSC, 4, T, mv).generate(sc)
        assert svo[s + 1] - svo[s] == len(on)
        assert np.array_equal(num[svo[s]:svo[s + 1]], on)


@pytest.mark.parametrize("case", list(GUARD_CASES))
@pytest.mark.parametrize("vs,T,mv", [(synth.VOXEL_01, 10, 60000), (synth.VOXEL_PILLAR, 20, 30000)])
def test_voxelize_mean_guard_bands(case, vs, T, mv):
    scenes = GUARD_CASES[case]()
    rc, bufs, cap = _call_hard(scenes, vs, RNG_NUSC, T, mv, mean=True)
    assert rc == 0, F.lib().lvq_strerror(rc)
    for name, b in bufs.items():
        assert b.intact(), f"{name}: bytes outside the reported size were written ({case})"


@pytest.mark.parametrize("mean", [False, True])
def test_hard_voxelizer_workspace_one_byte_short_is_refused(mean):
    rc, bufs, _ = _call_hard(GUARD_CASES["ordinary"](), synth.VOXEL_01, RNG_NUSC, 10, 60000, mean=mean, ws_delta=-1)
    assert rc == -2                                              # LVQ_EWORKSPACE, nothing launched
    for name, b in bufs.items():
        assert b.intact(), name
    assert bool((bufs["svo"].view(torch.uint8, (-1,)) == PATTERN).all())


@pytest.mark.parametrize("path", ["binned", "legacy"])
@pytest.mark.parametrize("case", list(GUARD_CASES))
@pytest.mark.parametrize("ndim,vs", [(3, synth.VOXEL_01), (2, synth.VOXEL_PILLAR)])
def test_dynamic_voxelizer_guard_bands(case, ndim, vs, path, tune):
    legacy = path
    tune(voxel_path={"hashed": 0, "binned": 1, "legacy": 2}[path])
    lib = F.lib()
    scenes = [s for s in GUARD_CASES[case]()]
    bs = len(scenes)
    bpts = np.concatenate([np.pad(p, ((0, 0), (1, 0)), constant_values=s) for s, p in enumerate(scenes)]).astype(np.float32)
    pts = torch.from_numpy(bpts).to(DEV)
    n = pts.shape[0]
    grid = [int(v) for v in LO.grid_size(RNG_NUSC, vs)]
    ks = bs * grid[0] * grid[1] * (grid[2] if ndim == 3 else 1)
    cap = max(1, min(n, ks))
    inv, pc = Banded(max(n, 1) * 4), Banded(max(n, 1) * 12)
    key, cnt, co, counts = Banded(cap * 4), Banded(cap * 4), Banded(cap * 16), Banded(8)
    gi = F.i32x(grid)
    ws = Banded(lib.lvq_voxelize_dynamic_workspace_bytes(F.i64(n), F.cint(bs), gi, F.cint(ndim)))
    rc = lib.lvq_voxelize_dynamic(F.ptr(pts), F.i64(n), F.cint(5), F.cint(bs), F.f32x(RNG_NUSC), F.f32x([float(np.float32(v)) for v in vs]),
                                  gi, F.cint(ndim), inv.ptr, pc.ptr, key.ptr, cnt.ptr, co.ptr, counts.ptr, ws.ptr, F.csize(ws.nbytes),
                                  F.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert rc == 0, lib.lvq_strerror(rc)
    for name, b in dict(inv=inv, pc=pc, key=key, cnt=cnt, co=co, counts=counts, ws=ws).items():
        assert b.intact(), f"{name}: bytes outside the reported size were written ({case}, ndim={ndim}, legacy={legacy})"
    o = LO.dynamic_voxelize(bpts, RNG_NUSC, vs, grid, ndim)
    m = int(counts.view(torch.int32, (2,))[0])
    assert m == len(o["unq_key"])
    assert np.array_equal(key.view(torch.int32, (cap,))[:m].cpu().numpy(), o["unq_key"])
    # one byte short
    ws2 = Banded(ws.nbytes - 1)
    rc = lib.lvq_voxelize_dynamic(F.ptr(pts), F.i64(n), F.cint(5), F.cint(bs), F.f32x(RNG_NUSC), F.f32x([float(np.float32(v)) for v in vs]),
                                  gi, F.cint(ndim), inv.ptr, pc.ptr, key.ptr, cnt.ptr, co.ptr, counts.ptr, ws2.ptr, F.csize(ws2.nbytes),
                                  F.stream_ptr(torch.device(DEV)))
    torch.cuda.synchronize()
    assert rc == -2 and ws2.intact()
