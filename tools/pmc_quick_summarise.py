#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_quick.sh: counter value per launch of the dominant
(most dispatched `step_stream`) kernel and per wave-cell-evaluation (fused launches: 2 * cells / 64)."""
import csv
import glob
import os
import sys

out, B, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cells = B * G * G
arms = {}
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    arm = os.path.relpath(path, out).split(os.sep)[0].split("_SQ_")[0].split("_GRBM")[0]
    arm = os.path.relpath(path, out).split(os.sep)[0]
    name = arm.split("_", 1)[0]
    per = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "step_stream" not in k:
            continue
        per.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in per.items():
        arms.setdefault(name, {}).setdefault(k, {})[c] = (sum(v) / len(v), len(v))
for name, kernels in sorted(arms.items()):
    for k, cs in sorted(kernels.items(), key=lambda kv: -max(n for _, n in kv[1].values())):
        steps = 2 if "fused2" in k else 1
        evals = steps * cells / 64.0
        n = max(n for _, n in cs.values())
        if n < 5:
            continue
        line = f"{name:10s} {k[:60]:60s} launches={n:4d}"
        for c, (v, _) in sorted(cs.items()):
            line += f" {c}={v / evals:.3f}/eval"
        if "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
            line += f" valu_busy={cs['SQ_ACTIVE_INST_VALU'][0] * 4 / 1024 / (cs['GRBM_GUI_ACTIVE'][0] / 8):.3f}"
        print(line)
