#!/usr/bin/env python3
"""Summarise the passes of tools/valu_pmc.sh: per-launch averages of the SQ counters of the dominant fused kernel,
VALU instructions per cell-evaluation and the fraction of time the SIMDs issue VALU work.

usage: valu_from_pmc.py gpurun_out/valu_<tag> <tag>"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELLS, SIMDS, XCDS = 1024 * 256 * 256, 1024, 8
out = {"round": 1, "tag": tag, "workload": "c2 (1024 x 256x256), bench.py --steps 41 --warmup 7",
       "note": "rocprofv3 --kernel-trace --pmc <two counters per pass>; per-launch averages over the dispatches of the "
               "binary16->binary16 fused kernel; one launch = 2 steps = 2 * cells / 64 wave-cell-evaluations"}
for prec in ("fast", "exact"):
    vals = {}
    kernel = None
    for path in glob.glob(os.path.join(src, f"{prec}_*", "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(path)):
            if "fused2" in r["Kernel_Name"] and "DF16_DF16_" in r["Kernel_Name"]:
                kernel = r["Kernel_Name"].split("(")[0]
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            vals[k] = sum(v) / len(v)
            vals[k + "_dispatches"] = len(v)
    if not vals:
        continue
    wave_evals = 2.0 * CELLS / 64.0
    d = {}
    if "SQ_INSTS_VALU" in vals:
        d["valu_instr_per_cell_eval"] = vals["SQ_INSTS_VALU"] / wave_evals
    if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        # ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the XCDs
        d["valu_busy_fraction"] = vals["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (vals["GRBM_GUI_ACTIVE"] / XCDS)
    out[prec] = {"kernel": kernel, **vals, "derived": d}
path = os.path.join(root, "profiles", f"{tag}_valu_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print(path, {k: v.get("derived") for k, v in out.items() if isinstance(v, dict)})
