#!/usr/bin/env python3
"""Where a dw_run_episode call's wall time goes on the small-world episode path (1000 worlds of 8x8, 4 greedy agents, as
bench.py's c4_dim8): per call of K steps - the Python wrapper, the C call, the kernel (HIP events on the library's stream).

usage: episode_overhead.py [precision=exact] [calls=200] [K,K,...=16,64,256] [grid=8] [worlds=1000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "exact"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
Ks = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else [16, 64, 256]
G = int(sys.argv[4]) if len(sys.argv) > 4 else 8
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
N = 4
p = amd.default_params(B, G, G, N)
p.precision = _ffi.PRECISION[precision]
eng = amd.Engine(p)
eng.init_random(3)
eng.policy_greedy(argmin=False)
eng.step_device_actions(0.9)
c_time = [0.0]
inner = eng._lib.dw_run_episode


def timed(*a):
    t = time.perf_counter()
    rc = inner(*a)
    c_time[0] += time.perf_counter() - t
    return rc


class _Lib:                                   # the engine's library with dw_run_episode timed
    def __init__(self, lib):
        self.__dict__["_l"] = lib

    def __getattr__(self, k):
        return timed if k == "dw_run_episode" else getattr(self._l, k)


eng._lib = _Lib(eng._lib)
for K in Ks:
    Ls = np.full(K, 0.9)
    for _ in range(20):
        eng.run_episode(Ls, _ffi.POLICY_ARGMAX, reuse_buffers=True)
    eng.sync()
    c_time[0] = 0.0
    ev = 0.0
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.run_episode(Ls, _ffi.POLICY_ARGMAX, reuse_buffers=True)
    wall = time.perf_counter() - t0
    for _ in range(20):                        # the same calls bracketed by events (the bracket itself costs host time)
        eng.timer_start()
        eng.run_episode(Ls, _ffi.POLICY_ARGMAX, reuse_buffers=True)
        ev += eng.timer_stop()
    print(f"{precision} {B} x {G}x{G} K={K}: wall {wall / calls * 1e6:.1f} us per call = {wall / calls / K * 1e6:.3f} us/step; C call "
          f"{c_time[0] / calls * 1e6:.1f} us; stream time (copy up, kernel, copy down) {ev / 20 * 1e3:.1f} us; "
          f"python wrapper {(wall - c_time[0]) / calls * 1e6:.1f} us", flush=True)
eng.close()
