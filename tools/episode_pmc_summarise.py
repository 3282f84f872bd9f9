#!/usr/bin/env python3
"""Summary of tools/episode_pmc.sh: per episode kernel the average launch duration (kernel statistics pass) and every
collected SQ counter per launch, per step and per world-step (launch = 64 steps of B worlds)."""
import csv
import glob
import os
import sys

out = sys.argv[1]
SHAPES = {"c4dim8": (1000, 64), "es": (2048, 64)}               # worlds, steps per launch (the tools' chunk)
for tag, (B, K) in SHAPES.items():
    dur = {}
    for path in glob.glob(os.path.join(out, f"{tag}_stats*", "**", "*kernel_stats.csv"), recursive=True):
        mode = os.path.basename(os.path.dirname(os.path.dirname(path))).replace(f"{tag}_stats", "").strip("_") or "exact"
        for r in csv.DictReader(open(path)):
            if "episode" in r["Name"]:
                dur[(mode, r["Name"].split("(")[0])] = (float(r["AverageNs"]), int(r["Calls"]))
    for (mode, k), (ns, calls) in sorted(dur.items()):
        print(f"{tag} {mode:5s} {k:45s} {calls:5d} launches, {ns / 1e3:9.1f} us per launch = {ns / 1e3 / K:6.3f} us per step "
              f"({B} worlds)")
    per = {}
    for path in glob.glob(os.path.join(out, f"{tag}_pmc*", "**", "*counter_collection.csv"), recursive=True):
        d = os.path.relpath(path, out).split(os.sep)[0]
        mode = "fast" if "_fast_" in d else "exact"
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "episode" not in k:
                continue
            per.setdefault((mode, k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (mode, k, c), v in sorted(per.items()):
        full = [x for x in v if x >= 0.5 * max(v)]                # (the last chunk of an ES generation is shorter)
        a = sum(full) / len(full)
        print(f"{tag} {mode:5s} {k:45s} {c:22s} {a:14.0f} per launch, {a / K:12.1f} per step, {a / K / B:9.2f} per world-step")
