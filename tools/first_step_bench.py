#!/usr/bin/env python3
"""Time of the FIRST step of an episode (from the un-quantised Philox state) per arithmetic mode, and the fraction of
cells the exact mode's bounded float32 pass sends to float64.  usage: first_step_bench.py B G [B G ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402

args = [int(x) for x in sys.argv[1:]] or [1024, 256]
for B, G in zip(args[0::2], args[1::2]):
    for prec, env in (("exact", {}), ("fast", {}), ("exact", {"DW_FIRST_GENERIC": "1"}), ("fast", {"DW_FIRST_GENERIC": "1"})):
        os.environ.update(env)
        p = amd.default_params(B, G, G, 0)
        p.precision = _ffi.PRECISION[prec]
        eng = amd.Engine(p)
        times = []
        for rep in range(3):
            eng.init_random(42 + rep)
            eng.sync()
            eng.timer_start()
            eng.step(0.75)
            times.append(eng.timer_stop())
        fix = eng.last_fixup_count()
        later = []                                               # an ordinary single step from the quantised state, for scale
        for rep in range(3):
            eng.timer_start()
            eng.step(0.76)
            later.append(eng.timer_stop())
        eng.close()
        for k in env:                                            # (read by the library at every launch)
            os.environ.pop(k)
        print(f"{B} x {G}^2 {prec:5s} {'one thread per cell' if env else 'default':19s} first step: {min(times):9.3f} ms (min of 3), "
              f"float64 cells {fix} = {100.0 * fix / (B * G * G):.4f} %; an ordinary single step: {min(later):.3f} ms "
              f"(ratio {min(times) / min(later):.2f})", flush=True)
