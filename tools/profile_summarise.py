#!/usr/bin/env python3
"""Turn the passes of tools/profile_round.sh into the files under profiles/ that bench.py and DESIGN.md cite:

  profiles/<tag>_<workload>_<precision>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied)
  profiles/traffic_<tag>_<workload>_<precision>.json       HBM bytes per launch of the fused kernel:
        FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, both in KB, averaged over its
        dispatches; separate --pmc passes; the kernel's average duration from the --stats pass
  profiles/<tag>_<workload>_valu_pmc.json                  SQ counters of the fused kernel per launch: VALU
        instructions per cell-evaluation, SIMD busy fraction

usage: profile_summarise.py gpurun_out/prof_<tag>_<workload> <tag> <workload>"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, XCDS = 1024, 8


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


def bench_line(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"no JSON line in {path}")


def counter_avg(path, want):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == want:
            per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


valu_out = None
for prec in ("exact", "fast"):
    line = bench_line(os.path.join(src, f"stats_{prec}.json"))
    cfg = line["config"]
    cells = cfg["worlds_per_gpu"] * cfg["grid"][0] * cfg["grid"][1]
    desc = f"{wl} ({cfg['worlds_per_gpu']} x {cfg['grid'][0]}x{cfg['grid'][1]})"
    stats_csv = find(f"stats_{prec}/**/*kernel_stats.csv")
    stats = {r["Name"]: r for r in csv.DictReader(open(stats_csv))}
    shutil.copy(stats_csv, os.path.join(root, "profiles", f"{tag}_{wl}_{prec}_kernel_stats.csv"))
    json.dump(line, open(os.path.join(root, "profiles", f"{tag}_bench_{wl}_{prec}.json"), "w"))
    fetch = counter_avg(find(f"pmc_{prec}_FETCH_SIZE/**/*counter_collection.csv"), "FETCH_SIZE")
    write = counter_avg(find(f"pmc_{prec}_WRITE_SIZE/**/*counter_collection.csv"), "WRITE_SIZE")
    out = {"workload": wl, "precision": prec, "round": int(tag[1:3]) if tag[1:3].isdigit() else 0, "tag": tag, "plane_elem_bytes": 2, "shape": desc,
           "worlds_per_gpu": cfg["worlds_per_gpu"], "grid": cfg["grid"],
           "library_build_id": cfg.get("library_build_id"),         # bench.py flags the derived fields stale when
                                                                      # the live library was built from other sources
           "command": f"bench.py --workload {wl} --precision {prec} (tools/profile_round.sh)"}
    for label, names, steps in (("fused", [k for k in fetch if "fused2" in k], 2),
                                ("single", [k for k in fetch if "step_stream" in k and "fused2" not in k], 1)):
        if not names:
            continue
        k = max(names, key=lambda n: fetch[n][1])
        f_kb, n = fetch[k]
        w_kb = write[k][0]
        hbm = (2.0 * f_kb + w_kb) * 1024.0
        entry = {"kernel": k.split("(")[0], "steps_per_launch": steps, "dispatches": n, "FETCH_SIZE_KB_avg": f_kb,
                 "WRITE_SIZE_KB_avg": w_kb, "hbm_bytes_per_launch": hbm,
                 "algorithmic_bytes_per_launch": 8 * cells * steps,
                 "hbm_bytes_per_cell_update": hbm / (cells * steps),
                 "read_bytes_per_cell_update": 2.0 * f_kb * 1024.0 / (cells * steps),
                 "write_bytes_per_cell_update": w_kb * 1024.0 / (cells * steps)}
        if k in stats:
            entry["rocprofv3_kernel_avg_ns"] = float(stats[k]["AverageNs"])
            entry["rocprofv3_kernel_calls"] = int(stats[k]["Calls"])
        out[label] = entry
    main = out.get("fused") or out["single"]
    out.update({"kernel": main["kernel"], "steps_per_launch": main["steps_per_launch"],
                "hbm_bytes_per_launch": main["hbm_bytes_per_launch"],
                "algorithmic_bytes_per_launch": main["algorithmic_bytes_per_launch"],
                "hbm_bytes_per_cell_update": main["hbm_bytes_per_cell_update"],
                "calibration": "the guide's FETCH_SIZE x2 correction is stated for 16-byte-per-lane reads; these kernels read 8 "
                               "bytes per lane (four binary16 cells).  The SINGLE-step kernel is the calibration on a known byte "
                               "count in this access pattern: it must read 4 B x 66/64 rows (+ 0.016 B of halo columns on W > 256) "
                               "= 4.125-4.14 B and write exactly 4 B per cell-update - compare single.read_/write_bytes_per_cell_update",
                "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace only; gfx950 "
                        "correction FETCH_SIZE x2; fused kernels: one launch = two steps; algorithmic bytes = 8 B per "
                        "cell-update (binary16 planes: 2 read + 2 written)"})
    path = os.path.join(root, "profiles", f"traffic_{tag}_{wl}_{prec}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, f"{out['hbm_bytes_per_cell_update']:.3f} B/cell-update", main["kernel"],
          f"avg {main.get('rocprofv3_kernel_avg_ns', 0) / 1e6:.4f} ms")
    # ---- SQ counters of the fused kernel
    if valu_out is None:
        valu_out = {"round": int(tag[1:3]) if tag[1:3].isdigit() else 0, "tag": tag, "workload": desc, "plane_elem_bytes": 2,
                    "library_build_id": cfg.get("library_build_id"), "worlds_per_gpu": cfg["worlds_per_gpu"],
                    "grid": cfg["grid"],
                    "note": "rocprofv3 --kernel-trace --pmc <two counters per pass>; per-launch averages over the dispatches "
                            "of the fused step-pair kernel; one launch = 2 steps = 2 * cells / 64 wave-cell-evaluations"}
    vals, kernel = {}, None
    for path in glob.glob(os.path.join(src, f"pmc_{prec}_S*", "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(path)):
            if "fused2" in r["Kernel_Name"]:
                kernel = r["Kernel_Name"].split("(")[0]
                per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in per.items():
            vals[k] = sum(v) / len(v)
            vals[k + "_dispatches"] = len(v)
    if vals:
        wave_evals = 2.0 * cells / 64.0
        d = {}
        if "SQ_INSTS_VALU" in vals:
            d["valu_instr_per_cell_eval"] = vals["SQ_INSTS_VALU"] / wave_evals
        if "SQ_INSTS_VALU_TRANS_F32" in vals:
            d["trans_instr_per_cell_eval"] = vals["SQ_INSTS_VALU_TRANS_F32"] / wave_evals
        if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
            # ACTIVE_INST_VALU counts quad-cycles summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the XCDs
            d["valu_busy_fraction"] = vals["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS / (vals["GRBM_GUI_ACTIVE"] / XCDS)
        if "GRBM_GUI_ACTIVE" in vals and kernel:
            # effective clock under this load (MI355X_MICROARCH.md, DVFS give-back): GRBM_GUI_ACTIVE is summed over the 8
            # XCDs; the kernel's wall time from the --stats pass - a different run of the same command, so +-5 %
            ns = next((float(r["AverageNs"]) for n_, r in stats.items() if n_.split("(")[0] == kernel), None)
            if ns:
                d["effective_clock_ghz"] = vals["GRBM_GUI_ACTIVE"] / XCDS / ns
        valu_out[prec] = {"kernel": kernel, **vals, "derived": d}
if valu_out:
    path = os.path.join(root, "profiles", f"{tag}_{wl}_valu_pmc.json")
    json.dump(valu_out, open(path, "w"), indent=1)
    print(path, {k: v.get("derived") for k, v in valu_out.items() if isinstance(v, dict)})
