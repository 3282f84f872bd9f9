#!/usr/bin/env python3
"""The float32-only mode has no oracle to be bit-identical to - but every kernel family evaluates the same
correctly rounded operations in the same order, so all of them must agree with each other bit for bit.
Random shapes and step counts; the same quantised state stepped by the default kernel selection and by
every alternative the environment switches can force (generic / tiled / wave-strip single steps, fused
pairs, packed mode, short strips, LDS-resident
episodes).

usage: fuzz_fast_consistency.py [cases=40] [seed=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402

VARIANTS = [{}, {"DW_NO_FUSE": "1"}, {"DW_KERNEL": "tiled"}, {"DW_NO_PACK": "1"}, {"DW_PACK_MIN_STRIPS": "1"},
            {"DW_STRIP_ROWS": "8"}, {"DW_NO_EPISODE_KERNEL": "1"}, {"DW_NO_EPISODE_KERNEL": "1", "DW_PACK_MIN_STRIPS": "1"},
            {"DW_NO_EPISODE_KERNEL": "1", "DW_PACK_MIN_STRIPS": "1", "DW_NO_FUSE": "1"},
            {"DW_KERNEL": "tiled", "DW_NO_EPISODE_KERNEL": "1"}]
KEYS = sorted({k for v in VARIANTS for k in v})

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for i in range(cases):
    rng = np.random.RandomState(seed * 10000 + i)
    W = int(rng.choice([8, 16, 32, 64, 96, 128, 192, 256, 260, 320, 512]))
    H = int(rng.randint(3, 100))
    B = int(rng.randint(1, 40 if W * H < 20000 else 5))
    steps = int(rng.randint(1, 14))
    L0, dL = float(rng.uniform(0.8, 1.4)), float(rng.uniform(-0.005, 0.02))
    light = np.floor(rng.rand(B, H, W) * 400) * (rng.rand(B, H, W) < 0.6)
    dark = np.floor(rng.rand(B, H, W) * 400) * (rng.rand(B, H, W) < 0.6)
    outs, kernels = [], []
    for var in VARIANTS:
        for key in KEYS:
            os.environ.pop(key, None)
        os.environ.update(var)
        p = amd.default_params(B, H, W, 0)
        p.precision = _ffi.PRECISION["fast"]
        eng = amd.Engine(p)
        eng.upload_state_f32((light / 1000).astype(np.float32), (dark / 1000).astype(np.float32), quantised=True)
        eng.step_n(steps, L0, dL, 0.6, 1.9)
        gl, gd = eng.download_planes()
        s = eng.reduce()
        outs.append((np.rint(gl * 1000), np.rint(gd * 1000), s["sum_light_k"].copy(), s["max_k"].copy()))
        kernels.append(eng.kernel_info()[:24])
        eng.close()
    same = all(all(np.array_equal(x, y) for x, y in zip(outs[0], o)) for o in outs[1:])
    bad += not same
    print(("ok  " if same else "FAIL"), seed * 10000 + i, f"B={B} H={H} W={W} steps={steps} kernels={sorted(set(kernels))}", flush=True)
for key in KEYS:
    os.environ.pop(key, None)
print(f"{cases - bad}/{cases} cases: every kernel selection gives bit-identical float32 results")
sys.exit(1 if bad else 0)
