#!/usr/bin/env python3
"""Effective shader clock and issue utilisation of k_merkle_layer from one rocprofv3 counter pass over tools/merkle_shapes.py:

  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \\
            --output-format csv -d <dir> -- python3 tools/merkle_shapes.py
  python3 tools/merkle_clock.py <dir> [kernel name substring = k_merkle_layer]

Per launch shape (grid size): duration, effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, "DVFS give-back"),
VALU instructions per SIMD-cycle at that clock (SQ_INSTS_VALU x 4 cycles per wave64 instruction / (cycles x 1024 SIMDs)), and how the
waves' lifetime splits into issuing / parked on s_waitcnt / issue-stalled. The 39.3 T lane-ops/s peak that bench.py prices against assumes
2.4 GHz; this table says how much of the gap to it is clock and how much is stall."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "k_merkle_layer"
    per = defaultdict(lambda: defaultdict(float))       # dispatch id -> counter -> value
    meta = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel not in r["Kernel_Name"]:
                    continue
                k = r["Dispatch_Id"]
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
                short = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bf::", "")
                meta[k] = {"grid": int(r["Grid_Size"]), "start": int(r.get("Start_Timestamp") or 0), "end": int(r.get("End_Timestamp") or 0), "name": short}
    if any(m["end"] == 0 for m in meta.values()):
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    k = r.get("Dispatch_Id")
                    if k in meta:
                        meta[k]["start"], meta[k]["end"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    shapes = defaultdict(list)
    for k, c in per.items():
        shapes[(meta[k]["name"], meta[k]["grid"])].append((meta[k]["end"] - meta[k]["start"], c))
    out = []
    for (kname, grid), launches in sorted(shapes.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        launches = launches[1:] if len(launches) > 2 else launches      # first launch of a shape: cold
        n = len(launches)
        dur = sum(l[0] for l in launches) / n
        c = {name: sum(l[1][name] for l in launches) / n for name in launches[0][1]}
        if dur <= 0 or "GRBM_GUI_ACTIVE" not in c:
            continue
        cycles = c["GRBM_GUI_ACTIVE"] / 8
        row = {"kernel": kname, "grid_lanes": grid, "launches": n, "avg_us": round(dur / 1e3, 1), "effective_clock_GHz": round(cycles / dur, 3)}
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_INSTS_LDS"):
            row["lds_bank_conflict_cycles_per_lds_inst"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"], 3)
        if "SQ_INSTS_VALU" in c:
            row["valu_insts_per_simd_cycle_x4"] = round(c["SQ_INSTS_VALU"] * 4 / (cycles * 1024), 4)
            row["valu_lane_ops_per_s_T"] = round(c["SQ_INSTS_VALU"] * 64 / dur / 1e3, 2)
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            w = c["SQ_WAVE_CYCLES"]
            for name, key in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_ACTIVE_INST_VALU", "issuing_valu"), ("SQ_WAIT_ANY", "parked_waitcnt"), ("SQ_WAIT_INST_ANY", "issue_stalled")):
                if name in c:
                    row[f"wave_time_{key}"] = round(c[name] / w, 4)
            row["waves_in_flight_per_simd"] = round(w * 4 / (cycles * 1024), 2)     # SQ_WAVE_CYCLES counts quad-cycles
        out.append(row)
    json.dump({"_doc": __doc__.strip().split("\n\n")[0], "kernel": kernel, "shapes": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
