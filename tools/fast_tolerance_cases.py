#!/usr/bin/env python3
"""Calibration of the float32-only mode's test tolerances: the numbers tests/test_gpu_parity.py asserts at
(measured x 2).  Prints, for the synthetic random states of test_single_step_vs_oracle (every shape x luminosity)
the worst fraction of differing cells after ONE step, and for the 3-step run of
test_fused_fast_trajectory_vs_oracle_tolerance the fraction and the largest difference.

usage (GPU box): python tools/fast_tolerance_cases.py [--out profiles/r02_fast_tolerance_cases.json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DW_PACK_MIN_STRIPS", "1")
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402
from oracle import c_oracle  # noqa: E402
from tests.test_gpu_parity import SHAPES, _random_quantised, _k  # noqa: E402


def engine(B, H, W, precision):
    p = amd.default_params(B, H, W, 0)
    p.precision = _ffi.PRECISION[precision]
    return amd.Engine(p)


ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
a = ap.parse_args()
rows = []
for (B, H, W) in SHAPES:
    for L in (0.75, 1.0, 1.31):
        rng = np.random.RandomState(B * 1000 + H + W)
        light, dark = _random_quantised(rng, B, H, W)
        ref = c_oracle.forward(light, dark, L)
        eng = engine(B, H, W, "fast")
        eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=True)
        eng.step(L)
        gl, gd = eng.download_planes()
        dl, dd = np.abs(_k(gl) - _k(ref[:, 1])), np.abs(_k(gd) - _k(ref[:, 2]))
        rows.append({"shape": [B, H, W], "L": L, "cells": int(dl.size),
                     "frac_diff": (np.count_nonzero(dl) + np.count_nonzero(dd)) / (2.0 * dl.size),
                     "max_diff": int(max(dl.max(), dd.max()))})
        eng.close()
big = [r for r in rows if r["cells"] >= 20000]
worst = max(rows, key=lambda r: r["frac_diff"])
worst_big = max(big, key=lambda r: r["frac_diff"])
print("synthetic random states, one step: worst frac_diff", worst, "| worst among >= 20000 cells", worst_big)
B, H, W = 2, 256, 256
eng = engine(B, H, W, "fast")
eng.init_random(3)
eng.step_n(60, 0.9, 0.002, 0.75, 1.5)
light, dark = eng.download_planes()
eng.step_n(3, 1.02, 0.002, 0.75, 1.5)
c_oracle.step_n(light, dark, 1.02, 0.002, 3)
gl, gd = eng.download_planes()
dl, dd = np.abs(_k(gl) - _k(light)), np.abs(_k(gd) - _k(dark))
three = {"frac_diff": (np.count_nonzero(dl) + np.count_nonzero(dd)) / (2.0 * dl.size), "max_diff": int(max(dl.max(), dd.max()))}
print("developed state, 3 steps (fused pair + single):", three)
eng.close()
if a.out:
    json.dump({"single_step_synthetic": rows, "worst": worst, "worst_ge_20000_cells": worst_big, "three_steps_developed": three},
              open(a.out, "w"), indent=1)
