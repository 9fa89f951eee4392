#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes into one JSON of per-kernel, per-launch averages.

    python tools/pmc_summary.py OUT.json DIR [DIR ...]      # every *_counter_collection.csv below the DIRs

Counter values are summed over the rows of one dispatch (rocprofv3 emits one row per counter and dispatch, already reduced over
XCDs / SEs) and averaged over the dispatches of a kernel whose grid size is the largest seen for that kernel name (= the bench-size
launches; warm-up and small-shape launches of the same kernel are dropped).  FETCH_SIZE / WRITE_SIZE are reported raw (KiB units)
and as bytes with the gfx950 correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE counts exactly half of a 16-byte-per-lane
coalesced read stream (x2), WRITE_SIZE is exact for 16-byte stores.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:120]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    per = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [(grid, value)]
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            disp = defaultdict(float)
            meta = {}
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    key = (r["Dispatch_Id"], r["Counter_Name"])
                    disp[key] += float(r["Counter_Value"])
                    meta[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r.get("VGPR_Count") or 0),
                                              int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for (did, cname), v in disp.items():
                k, grid, vg, ns = meta[did]
                per[k][cname].append((grid, v, ns, vg))
    res = {"note": __doc__.strip().split("\n\n")[1].replace("\n", " ")}
    for k, cs in sorted(per.items()):
        gmax = max(g for vals in cs.values() for g, _, _, _ in vals)
        entry = {"grid_size": gmax}
        for cname, vals in sorted(cs.items()):
            sel = [(v, ns, vg) for g, v, ns, vg in vals if g == gmax]
            entry[cname] = sum(v for v, _, _ in sel) / len(sel)
            entry.setdefault("launches", {})[cname] = len(sel)
            entry["avg_ns_under_pmc"] = sum(ns for _, ns, _ in sel) / len(sel)
            entry["vgpr"] = sel[0][2]
        if "FETCH_SIZE" in entry:
            entry["hbm_read_bytes_corrected"] = entry["FETCH_SIZE"] * 1024.0 * 2.0
        if "WRITE_SIZE" in entry:
            entry["hbm_write_bytes"] = entry["WRITE_SIZE"] * 1024.0
        if "FETCH_SIZE" in entry and "WRITE_SIZE" in entry:
            entry["traffic_bytes_per_launch"] = entry["hbm_read_bytes_corrected"] + entry["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in entry and "GRBM_GUI_ACTIVE" in entry and entry["GRBM_GUI_ACTIVE"] > 0:
            # MFMA busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
            entry["mfma_busy_frac_of_simd_cycles"] = entry["SQ_VALU_MFMA_BUSY_CYCLES"] / (entry["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        res[k] = entry
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(f"{len(res) - 1} kernels -> {out}")


if __name__ == "__main__":
    main()
