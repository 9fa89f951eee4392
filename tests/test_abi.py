"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/lvq.h declares; pure-host entry points behave; the Python host mirrors the reference's
plugin interfaces (names, constructor keywords, state_dict keys)."""
import ctypes
import os

import pytest
import torch

from lidar_vision_vqa_amd import _ffi


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _ffi.lib()


def test_exports_every_declared_symbol(lib):
    names = _ffi.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_integration_md_lists_every_entry_point():
    """INTEGRATION.md's appendix (tools/gen_abi_table.py) names every function the header declares, with the reference interface it cites."""
    import os
    txt = open(os.path.join(os.path.dirname(_ffi.HEADER_PATH), "..", "INTEGRATION.md")).read()
    table = txt[txt.index("abi-table:begin"):]
    missing = [n for n in _ffi.declared_symbols() if f"| `{n}` |" not in table]
    assert not missing, f"run tools/gen_abi_table.py: {missing}"


def test_host_only_entry_points(lib):
    assert b"gfx950" in lib.lvq_version()
    assert lib.lvq_strerror(0) == b"ok" and b"workspace" in lib.lvq_strerror(-2)
    n = lib.lvq_voxelize_hard_workspace_bytes(ctypes.c_int64(32768), ctypes.c_int(1))
    assert 1 << 20 < n < 1 << 24
    g = (ctypes.c_int32 * 3)(1024, 1024, 40)
    assert lib.lvq_voxelize_dynamic_workspace_bytes(ctypes.c_int64(32768), ctypes.c_int(8), g, ctypes.c_int(3)) > 0
    # 64 scenes x 1440x1440x40 cells >= 2^31 keys: refused (the reference wraps int32 silently)
    g2 = (ctypes.c_int32 * 3)(1440, 1440, 40)
    assert lib.lvq_voxelize_dynamic_workspace_bytes(ctypes.c_int64(10), ctypes.c_int(64), g2, ctypes.c_int(3)) == 0


def test_tuning_record_is_the_only_routing_input(lib, monkeypatch):
    """Kernel-family choices enter through lvq_set_tuning (include/lvq.h: lvq_tuning), not through the environment: the library does not
    import getenv at all, the record round-trips through the ABI, and the Python mirror maps the LVQ_* variables onto it."""
    import subprocess
    syms = subprocess.run(["nm", "-D", "--undefined-only", _ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    assert ctypes.sizeof(_ffi.Tuning) == 16 * 4 + 8 + 8 * 4
    t = _ffi.Tuning()
    t.attn_nsplit = 7
    lib.lvq_tuning_defaults(ctypes.byref(t))
    assert t.attn_nsplit == 0
    try:
        rec = _ffi.set_tuning(attn_nsplit=3, voxel_path=2, ca_fused_stamps=1 << 40)
        got = _ffi.get_tuning()
        assert got == rec and got["attn_nsplit"] == 3 and got["voxel_path"] == 2 and got["ca_fused_stamps"] == 1 << 40
        with _ffi.tuning(attn_pipe=-1):
            assert _ffi.get_tuning()["attn_pipe"] == -1 and _ffi.get_tuning()["attn_nsplit"] == 3
        assert _ffi.get_tuning()["attn_pipe"] == 0
        assert lib.lvq_set_tuning(None) == 0 and not any(_ffi.get_tuning().values())
        with pytest.raises(_ffi.LvqError):
            _ffi.set_tuning(no_such_field=1)
        monkeypatch.setenv("LVQ_VOXEL_BINNED", "1")
        monkeypatch.setenv("LVQ_ATTN_NO_PIPE", "1")
        monkeypatch.setenv("LVQ_GEMM_STREAM_C_MB", "64")
        rec = _ffi.set_tuning()
        assert rec["voxel_path"] == 1 and rec["attn_pipe"] == -1 and rec["gemm_stream_c_mb"] == 64 and _ffi.get_tuning() == rec
    finally:
        monkeypatch.undo()
        _ffi.set_tuning()


def test_no_cpu_fallback():
    """The product path fails loudly on CPU tensors instead of silently computing elsewhere."""
    from lidar_vision_vqa_amd import fusion, lidar
    m = fusion.VATBlock(96, 4, 384, 0.1).eval()
    with pytest.raises(_ffi.LvqError), torch.no_grad():
        m(torch.zeros(1, 4, 96), torch.zeros(1, 5, 96))
    with pytest.raises(_ffi.LvqError):
        lidar.MeanVFE(None, 4)(dict(voxels=torch.zeros(2, 3, 4), voxel_num_points=torch.ones(2)))


def test_module_api_mirrors_reference():
    from lidar_vision_vqa_amd import fusion, lidar
    # registries with the reference's NAME strings (vfe/__init__.py:9-18)
    assert set(lidar.__all__) >= {"MeanVFE", "PillarVFE", "DynMeanVFE", "DynPillarVFE", "DynamicPillarVFESimple2D", "DynamicVoxelVFE"}

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    p = lidar.__all__["PillarVFE"](model_cfg=Cfg(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64]),
                                   num_point_features=4, voxel_size=[0.2, 0.2, 8], point_cloud_range=[-51.2, -51.2, -5, 51.2, 51.2, 3],
                                   grid_size=[512, 512, 1], depth_downsample_factor=None)
    assert p.get_output_feature_dim() == 64
    assert set(p.state_dict()) == {"pfn_layers.0.linear.weight", "pfn_layers.0.norm.weight", "pfn_layers.0.norm.bias",
                                   "pfn_layers.0.norm.running_mean", "pfn_layers.0.norm.running_var",
                                   "pfn_layers.0.norm.num_batches_tracked"}
    assert tuple(p.state_dict()["pfn_layers.0.linear.weight"].shape) == (64, 10)
    v = fusion.VATLiDAR(c_in=128, d_model=96, n_queries=12, n_layers=2, n_heads=4)
    keys = set(v.state_dict())
    for k in ("view_embed", "query", "refine.0.weight", "refine.0.bias", "proj.weight", "proj.bias", "norm_tokens.weight",
              "geo_mlp.0.weight", "geo_mlp.2.bias", "final_ln.weight", "post.0.weight", "post.1.weight", "post.4.bias",
              "blocks.1.ca.in_proj_weight", "blocks.0.sa.out_proj.bias", "blocks.1.mlp.3.weight"):
        assert k in keys, k
    assert tuple(v.state_dict()["blocks.0.ca.in_proj_weight"].shape) == (288, 96)
    with pytest.raises(AssertionError):
        fusion.VATLiDAR(c_in=16, d_model=96, n_queries=256)        # vat_lidar.py:75: n_queries % 6 == 0
    vv = fusion.VATVision(d_in=128, d_model=96, n_input_tokens=48, compression_factor=2, n_layers=1, n_heads=4, use_per_view_query=True)
    assert "view_query_embed" in vv.state_dict() and vv.n_queries == 24
    with pytest.raises(ValueError):
        fusion.VATVision(d_in=128, d_model=96, n_input_tokens=50, compression_factor=2, use_per_view_query=True, strict_per_view=True)
    va = fusion.VisionAdapter(128)
    assert set(va.state_dict()) == {"view_embed", "norm.weight", "norm.bias"}
    with pytest.raises(ValueError):
        va([torch.zeros(4, 128)] * 5)


def test_default_init_matches_reference_rng_order():
    """Same torch.manual_seed -> same default-initialised weights as torch's own containers built in the
    reference's order (vat_blocks.py:17-34): checkpoints AND fresh inits are interchangeable."""
    from lidar_vision_vqa_amd import fusion
    import torch.nn as nn
    torch.manual_seed(123)
    ours = fusion.VATBlock(64, 4, 128, 0.1)
    torch.manual_seed(123)
    ln1 = nn.LayerNorm(64); sa = nn.MultiheadAttention(64, 4, dropout=0.1, batch_first=True)
    assert torch.equal(ours.sa.in_proj_weight, sa.in_proj_weight)


def _fracs(node, path=""):
    """Every (path, value) of a key that starts with `frac` anywhere in the line."""
    if isinstance(node, dict):
        for k, v in node.items():
            if isinstance(v, (int, float)) and k.startswith("frac"):
                yield path + "/" + k, v
            else:
                yield from _fracs(v, path + "/" + k)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _fracs(v, f"{path}[{i}]")


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r*_bench_final.json (the newest) is the line `python bench.py` printed on the MI355X for the committed build: the fields
    the driver and the judge read must all be there (bench.py contract: metric / value / roofline / cpu_baseline ...), the roofline entry
    is the metric's own shape, and no fraction of a roofline anywhere in the line exceeds 1 (a fraction above 1 means a kernel is
    credited with work it does not do: VERDICT r2, `roofline_kv_proj`)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_final.json")))[-1]
    with open(path) as f:
        d = json.load(f)
    with open(os.path.join(root, "BASELINE.json")) as f:
        base = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].replace("x", "×") == base["metric"].replace("x", "×") or d["metric"][:30] == base["metric"][:30]
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert abs(d["value"] - d["config"]["scenes_per_gpu_per_step"] * d["config"]["fused_tokens_per_scene"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    fr = list(_fracs(d))
    assert fr and all(0.0 <= v <= 1.0 for _, v in fr), [x for x in fr if not 0.0 <= x[1] <= 1.0]
    if os.path.basename(path) >= "r03":
        assert r.get("shape") == [1, 32768, 196, 768, 12]                      # the metric's own shape (first row)
        assert "roofline_kv_proj" not in d
        assert d["parity_vs_cpu"]["scenes"] >= 3 and d["parity_vs_cpu"]["value_meets_tolerance"] is True
