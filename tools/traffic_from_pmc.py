#!/usr/bin/env python3
"""Turn the PMC passes of tools/profile_round.sh into profiles/traffic_<tag>_c2_<precision>.json.

HBM bytes per launch = FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, both in KB,
averaged over the dispatches of the dominant step kernel; the kernel's average duration comes from the
--stats pass of the same command.

usage: traffic_from_pmc.py gpurun_out/prof_<tag> <tag>"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELLS = 1024 * 256 * 256


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern} under {src}")
    return hits[0]


def counter_avg(path, want):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == want:
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


for prec in ("fast", "exact"):
    fetch = counter_avg(find(f"pmc_{prec}_FETCH_SIZE/**/*counter_collection.csv"), "FETCH_SIZE")
    write = counter_avg(find(f"pmc_{prec}_WRITE_SIZE/**/*counter_collection.csv"), "WRITE_SIZE")
    stats = {r["Name"]: r for r in csv.DictReader(open(find(f"stats_{prec}/**/*kernel_stats.csv")))}
    fused = [k for k in fetch if "fused2" in k]
    single = [k for k in fetch if "step_stream" in k and "fused2" not in k]
    out = {"workload": "c2", "precision": prec, "round": 1, "tag": tag}
    for label, names, steps in (("fused", fused, 2), ("single", single, 1)):
        if not names:
            continue
        k = max(names, key=lambda n: fetch[n][1])
        f_kb, n = fetch[k]
        w_kb = write[k][0]
        hbm = (2.0 * f_kb + w_kb) * 1024.0
        entry = {"kernel": k.split("(")[0], "steps_per_launch": steps, "dispatches": n, "FETCH_SIZE_KB_avg": f_kb,
                 "WRITE_SIZE_KB_avg": w_kb, "hbm_bytes_per_launch": hbm,
                 "algorithmic_bytes_per_launch": 16 * CELLS * steps,
                 "hbm_bytes_per_cell_update": hbm / (CELLS * steps)}
        if k in stats:
            entry["rocprofv3_kernel_avg_ns"] = float(stats[k]["AverageNs"])
        if len(names) > 1:                                  # e.g. the float32 / binary16 plane variants of a run
            entry["variants"] = [{"kernel": n_.split("(")[0], "dispatches": fetch[n_][1],
                                  "hbm_bytes_per_cell_update": (2.0 * fetch[n_][0] + write[n_][0]) * 1024.0 / (CELLS * steps)}
                                 for n_ in sorted(names, key=lambda n_: -fetch[n_][1])]
        out[label] = entry
    main = out.get("fused") or out["single"]
    out.update({"kernel": main["kernel"], "steps_per_launch": main["steps_per_launch"],
                "hbm_bytes_per_launch": main["hbm_bytes_per_launch"],
                "algorithmic_bytes_per_launch": main["algorithmic_bytes_per_launch"],
                "hbm_bytes_per_cell_update": main["hbm_bytes_per_cell_update"],
                "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; "
                        "gfx950 correction FETCH_SIZE x2; fused kernels: one launch = two steps"})
    path = os.path.join(root, "profiles", f"traffic_{tag}_c2_{prec}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, f"{out['hbm_bytes_per_cell_update']:.3f} B/cell-update", main["kernel"])
