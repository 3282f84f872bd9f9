#!/usr/bin/env python3
"""Kernel micro-bench: ms/step of the step kernel for one configuration (tuning aid).
usage: kbench.py B G [precision] [steps]   (env DW_* overrides are read by the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd
from therldaisyworld_amd import _ffi
B, G = int(sys.argv[1]), int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "fast"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
p = amd.default_params(B, G, G, 0)
p.precision = _ffi.PRECISION[prec]
eng = amd.Engine(p)
eng.init_random(42)
L = eng.step_n(220, 0.75, 0.75 / 512, 0.75, 1.5)     # reach a developed state (L ~ 1.07)
best = []
for rep in range(5):
    eng.timer_start()
    L2 = eng.step_n(steps, L, 0.0, 0.75, 1.5)         # constant luminosity: steady state workload
    best.append(eng.timer_stop() / steps)
ms = sorted(best)[len(best) // 2]
cells = B * G * G
env = {k[3:]: v for k, v in os.environ.items() if k.startswith('DW_') and k != 'DW_LIB'}
print(f"{prec:5s} B={B} G={G} ms/step={ms:.4f} min={min(best):.4f} GB/s={16 * cells / ms / 1e6:.0f} "
      f"frac={16 * cells / ms / 1e6 / 8000:.3f} fixups={eng.last_fixup_count()} env={env} :: {eng.kernel_info()[:40]}")
