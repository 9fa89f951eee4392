#!/usr/bin/env python3
"""Copy the summaries of one round's profiling runs from gpurun_out/ (scratch) into profiles/ (tracked):
    python tools/collect_profiles.py r03
expects the outputs of `tools/profile_ca.sh <tag> mixed`, `tools/profile_ca.sh <tag>bf16 bf16`, `tools/profile_round.sh <tag>` and
`python bench.py > gpurun_out/bench_final.json`.  Writes profiles/<tag>_{bench_final.json, bench_kernel_stats.csv, voxel_kernel_stats.csv,
headline_{mixed,bf16}_kernel_stats.csv, pmc_summary.json, pmc_headline_summary.json, pmc_traffic.json} -- bench.py reads the last one."""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src_glob, dst):
    src = glob.glob(os.path.join(G, src_glob))
    if src:
        shutil.copy(src[0], os.path.join(P, dst))
    else:
        print("missing:", src_glob)


cp("bench_final.json", f"{tag}_bench_final.json")
cp(f"{tag}_prof/bench/*kernel_stats.csv", f"{tag}_bench_kernel_stats.csv")
cp(f"{tag}_prof/vox/*kernel_stats.csv", f"{tag}_voxel_kernel_stats.csv")
cp(f"{tag}_prof/pmc_summary.json", f"{tag}_pmc_summary.json")
cp(f"{tag}_ca_prof/kernel_stats.csv", f"{tag}_headline_mixed_kernel_stats.csv")
cp(f"{tag}bf16_ca_prof/kernel_stats.csv", f"{tag}_headline_bf16_kernel_stats.csv")
ca = {}
for t, mode in ((tag, "mixed"), (tag + "bf16", "bf16")):
    with open(os.path.join(G, f"{t}_ca_prof", "pmc_summary.json")) as f:
        ca[mode] = {k: v for k, v in json.load(f).items() if "k_ca_" in k}
with open(os.path.join(P, f"{tag}_pmc_headline_summary.json"), "w") as f:
    json.dump({"note": "tools/profile_ca.sh: rocprofv3 --pmc passes (SQ set / FETCH_SIZE / WRITE_SIZE, separate passes with --kernel-trace only) of "
                       "`python3 tools/prof_ca.py <mode> 12`: lvq_ca_fused at (1, 32768, 196, 768, 12); averages per launch over 12 launches", **ca}, f, indent=1)


def headline(mode, f16):
    kf = [v for k, v in ca[mode].items() if "k_ca_fused" in k][0]
    kp = [v for k, v in ca[mode].items() if "k_ca_kvproj" in k][0]
    return {"kernels": f"k_ca_kvproj<{f16}> + k_ca_fused<{f16}, 7, 0> (the two launches of lvq_ca_fused)",
            "traffic_bytes_per_launch": round(kf["traffic_bytes_per_launch"] + kp["traffic_bytes_per_launch"]),
            "k_ca_fused": {"traffic_bytes_per_launch": round(kf["traffic_bytes_per_launch"]), "hbm_read_bytes_corrected": round(kf["hbm_read_bytes_corrected"]),
                           "hbm_write_bytes": round(kf["hbm_write_bytes"]), "FETCH_SIZE_KiB_raw": kf["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": kf["WRITE_SIZE"],
                           "avg_ms_under_pmc": round(kf["avg_ns_under_pmc"] / 1e6, 4), "vgpr": kf["vgpr"], "SQ_INSTS_MFMA": kf["SQ_INSTS_MFMA"],
                           "SQ_INSTS_VALU": kf["SQ_INSTS_VALU"], "SQ_LDS_BANK_CONFLICT": kf["SQ_LDS_BANK_CONFLICT"]},
            "k_ca_kvproj": {"traffic_bytes_per_launch": round(kp["traffic_bytes_per_launch"]), "avg_ms_under_pmc": round(kp["avg_ns_under_pmc"] / 1e6, 4)},
            "algorithmic_bytes": "x read 100.7 MB + out written 100.7 MB + kv 0.6 MB + weights 4.7 MB = 206.6 MB (SURVEY 8d); measured ~352 MB: the fp32 residual x is "
                                 "read a second time in the out-projection epilogue (+100.7 MB; 384 KB per 128-query tile does not fit the 160 KB LDS next to the weight "
                                 "ring) and each of the 256 workgroups streams the 3 MB of packed weights, K and V^T from L2 / MALL (FETCH_SIZE counts L2 misses to the "
                                 "fabric, MALL hits included)"}


with open(os.path.join(G, f"{tag}_prof", "pmc_summary.json")) as f:
    r = json.load(f)
prev = {}
for name in sorted(os.listdir(P)):
    if "pmc_traffic" in name and name.endswith(".json") and not name.startswith(tag):
        with open(os.path.join(P, name)) as f:
            prev = json.load(f)
out = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, --kernel-trace only; FETCH_SIZE doubled: gfx950 tallies "
               "128-B requests at 64 B; WRITE_SIZE exact) on MI355X.  bench.py looks entries up by key prefix.  Headline entries: tools/profile_ca.sh "
               f"(profiles/{tag}_pmc_headline_summary.json); step entries: tools/profile_round.sh {tag} (profiles/{tag}_pmc_summary.json); written by tools/collect_profiles.py.",
       "lvq_ca_fused 1x32768x196 mixed": headline("mixed", "true"), "lvq_ca_fused 1x32768x196 mixed16": headline("mixed", "true"),
       "lvq_ca_fused 1x32768x196 bf16": headline("bf16", "false")}
a = r["k_attn32<4, 1, 2, true>"]
out["k_attn32 S=32 nq=576 nkv=262144 mixed"] = {
    "kernel": "k_attn32<4, 1, 2, true> (pipelined form, 4 KV splits)", "traffic_bytes_per_launch": round(a["traffic_bytes_per_launch"]),
    "hbm_read_bytes_corrected": round(a["hbm_read_bytes_corrected"]), "hbm_write_bytes": round(a["hbm_write_bytes"]),
    "avg_ms_under_pmc": round(a["avg_ns_under_pmc"] / 1e6, 3), "mfma_busy_frac_of_simd_cycles": round(a["mfma_busy_frac_of_simd_cycles"], 4),
    "algorithmic_bytes": prev.get("k_attn32 S=32 nq=576 nkv=262144 mixed", {}).get("algorithmic_bytes")}
k, c = r["bt::k_kv_rows<6, true, false>"], r["bt::k_conv_rows<true>"]
out["k_tile_kv S=32 mixed"] = {"kernels": "k_conv_rows<true> + k_kv_rows<6, true> (the two launches of lvq_bev_tile_kv)",
                               "traffic_bytes_per_launch": round(k["traffic_bytes_per_launch"] + c["traffic_bytes_per_launch"]),
                               "k_kv_rows": {"traffic_bytes_per_launch": round(k["traffic_bytes_per_launch"]), "avg_ms_under_pmc": round(k["avg_ns_under_pmc"] / 1e6, 3)},
                               "k_conv_rows": {"traffic_bytes_per_launch": round(c["traffic_bytes_per_launch"]), "avg_ms_under_pmc": round(c["avg_ns_under_pmc"] / 1e6, 3)}}
dyn = os.path.join(G, f"{tag}_dyn_prof", "pmc_summary.json")           # tools/profile_voxel_dyn.sh (optional)
if os.path.exists(dyn):
    with open(dyn) as f:
        d = json.load(f)
    parts = {k.split("(")[0].replace("vb::", "").replace("__amd_rocclr_fillBufferAligned", "memset"): v for k, v in d.items()
             if isinstance(v, dict) and ("vb::" in k or "fillBuffer" in k)}
    rd = sum(v.get("hbm_read_bytes_corrected", 0) for v in parts.values())
    wr = sum(v.get("hbm_write_bytes", 0) for v in parts.values())
    out["lvq_voxelize_dynamic 8 x 65536 points, 0.1 m grid, 3-D keys: whole call = memset + k_bin_hist + k_bin_scatter + k_dyn_slab_count + k_dyn_slab_write"] = {
        "command": "tools/profile_voxel_dyn.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- python3 tools/prof_voxel_dyn.py (10 calls; averages per launch)",
        "hbm_read_bytes_corrected": round(rd), "hbm_write_bytes": round(wr), "traffic_bytes_per_launch": round(rd + wr), "algorithmic_bytes_per_launch": 24212208,
        "per_kernel": {k: {"read_bytes": round(v.get("hbm_read_bytes_corrected", 0)), "write_bytes": round(v.get("hbm_write_bytes", 0)),
                           "us_under_pmc": round(v.get("avg_ns_under_pmc", 0) / 1e3, 1)} for k, v in parts.items()},
        "note": "2.7x the algorithmic bytes, as in round 2: the two binning passes each read the points and write 6 MB of keys / entries; the slab writer stores 22.7 MB for "
                "13.7 MB of outputs (the per-point inverse map leaves in slab order: 4-byte stores scattered over the original point order).  Round 3 changed the time "
                "of that kernel (53 -> 28 us: unique outputs by rank), not its traffic"}
    cp(f"{tag}_dyn_prof/kernel_stats.csv", f"{tag}_voxel_dynamic_kernel_stats.csv")
with open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print("profiles/ updated for", tag)
