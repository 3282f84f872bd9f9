#!/usr/bin/env python3
"""Time of dw_init_random (Philox draw + per-world statistics, one kernel since round 4) and of the statistics pass after an
upload (stats_only) on one shape.  usage: init_bench.py B G [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402

B, G = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = amd.Engine(amd.default_params(B, G, G, 0))
for quantised in (False, True):
    ts = []
    for r in range(reps):
        eng.timer_start()
        eng.init_random(7 + r, quantised=quantised)
        ts.append(eng.timer_stop())
    cells = B * G * G
    byt = cells * (4 if quantised else 8)
    print(f"init_random(quantised={quantised}) B={B} G={G}: min {min(ts):.3f} ms, median {sorted(ts)[len(ts) // 2]:.3f} ms "
          f"= {byt / min(ts) / 1e6:.0f} GB/s written; stats {eng.reduce()['max_k'][:3]}", flush=True)
eng.close()
