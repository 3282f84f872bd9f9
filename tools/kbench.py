#!/usr/bin/env python3
"""Kernel micro-bench (tuning aid): ms/step of the step kernel, interleaved A/B over several builds
of the library in ONE process (cdna_hip_programming.md §5.4 rule 24: boxes and even consecutive
processes on one box differ by up to 10 %).

usage: kbench.py B G precision[,precision] [--libs name=path[@ENV=V[;ENV=V]],...] [--steps N] [--rounds R]
       env DW_* overrides are read by the library at handle creation; an arm's @ENV=V pairs are set only while
       its handle is created (path may be empty: the default build)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("B", type=int)
ap.add_argument("G", type=int)
ap.add_argument("precisions", nargs="?", default="fast")
ap.add_argument("--libs", default="")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--agents", type=int, default=0)
a = ap.parse_args()
libs = [("default", None)]
if a.libs:
    libs = [tuple(x.split("=", 1)) for x in a.libs.split(",")]
arms = []
for prec in a.precisions.split(","):
    for name, spec in libs:
        path, _, envs = (spec or "").partition("@")
        over = dict(kv.split("=", 1) for kv in envs.split(";") if kv)
        saved = {k: os.environ.get(k) for k in over}
        os.environ.update(over)
        p = amd.default_params(a.B, a.G, a.G, a.agents)
        p.precision = _ffi.PRECISION[prec]
        eng = amd.Engine(p, lib_path=path or None)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        eng.init_random(42)
        L = eng.step_n(220, 0.75, 0.75 / 512, 0.75, 1.5)      # developed state, L ~ 1.07
        arms.append((f"{name}/{prec}", eng, L, []))
for r in range(a.rounds):
    for label, eng, L, times in arms:
        eng.timer_start()
        eng.step_n(a.steps, L, 0.0, 0.75, 1.5)                 # constant luminosity: steady workload
        times.append(eng.timer_stop() / a.steps)
cells = a.B * a.G * a.G
env = {k[3:]: v for k, v in os.environ.items() if k.startswith("DW_") and k != "DW_LIB"}
for label, eng, L, times in arms:
    med, mn = statistics.median(times), min(times)
    print(f"{label:22s} B={a.B} G={a.G} ms/step med={med:.4f} min={mn:.4f} max={max(times):.4f} "
          f"GB/s(med)={8 * cells / med / 1e6:.0f} frac={8 * cells / med / 1e6 / 8000:.3f} "   # binary16 planes: 8 B per cell-update
          f"fixups={eng.last_fixup_count()} env={env} :: {eng.kernel_info()[:44]}")
