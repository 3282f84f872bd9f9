#!/usr/bin/env python3
"""Large-scale exactness soak: the exact mode (fused step pairs) against the float64 C oracle on the
host's cores over the whole luminosity ramp, planes compared bit for bit at checkpoints.

usage: soak_exact.py [worlds=256] [grid=256] [steps=512] [checkpoints=4] [seed=7]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402
from oracle import c_oracle  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 512
nchk = int(sys.argv[4]) if len(sys.argv) > 4 else 4
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 7
c_oracle.build()
c_oracle.set_threads(min(c_oracle.max_threads(), c_oracle.usable_cpus()))
p = amd.default_params(B, G, G, 0)
p.precision = _ffi.PRECISION["exact"]
eng = amd.Engine(p)
eng.init_random(seed)
light, dark = eng.download_planes()
Lg = Lo = 0.75
dL = 0.75 / 512
per = steps // nchk
mismatch, t_gpu, t_cpu, fix = 0, 0.0, 0.0, 0
for c in range(nchk):
    t0 = time.perf_counter()
    Lg = eng.step_n(per, Lg, dL, 0.75, 1.5)
    gl, gd = eng.download_planes()
    t_gpu += time.perf_counter() - t0
    fix += eng.last_fixup_count()
    t0 = time.perf_counter()
    Lo = c_oracle.step_n(light, dark, Lo, dL, per)
    t_cpu += time.perf_counter() - t0
    kl, kd = np.rint(gl * 1000), np.rint(gd * 1000)
    ol, od = np.rint(light * 1000), np.rint(dark * 1000)
    mismatch += int(np.count_nonzero(kl != ol) + np.count_nonzero(kd != od))
    print(f"checkpoint {c + 1}/{nchk}: step {(c + 1) * per}, L {Lg:.6f} (oracle {Lo:.6f}), mismatching cell values so far "
          f"{mismatch}, mean light {ol.mean() / 1000:.4f} dark {od.mean() / 1000:.4f}", flush=True)
print(json.dumps({"soak": "exact mode vs float64 C oracle", "worlds": B, "grid": G, "steps": per * nchk,
                  "cell_updates": B * G * G * per * nchk, "mismatching_cell_values": mismatch, "L_equal": Lg == Lo,
                  "gpu_wall_s_incl_downloads": t_gpu, "oracle_wall_s": t_cpu, "kernel": eng.kernel_info()[:60]}))
eng.close()
sys.exit(0 if mismatch == 0 and Lg == Lo else 1)
