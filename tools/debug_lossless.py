#!/usr/bin/env python3
"""debug aid: where does an exact-mode step differ from the C oracle on the synthetic full-range state of
tests/test_gpu_parity.py::test_binary16_planes_are_lossless?"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd
from therldaisyworld_amd import _ffi
from oracle import c_oracle


def k(x):
    return np.rint(np.asarray(x) * 1000.0).astype(np.int64)


for (B, H, W) in [(2, 100, 256), (1, 70, 320), (9, 40, 64), (3, 12, 12)]:
    rng = np.random.RandomState(H)
    light = (np.arange(B * H * W).reshape(B, H, W) % 1001).astype(np.float64)
    dark = np.minimum(1000.0 - light, rng.randint(0, 1001, size=(B, H, W)).astype(np.float64))
    ref = c_oracle.forward(light / 1000.0, dark / 1000.0, 1.0)
    res = {}
    for prec in ("exact", "f64", "fast"):
        p = amd.default_params(B, H, W, 0)
        p.precision = _ffi.PRECISION[prec]
        eng = amd.Engine(p)
        eng.upload_state_f32((light / 1000.0).astype(np.float32), (dark / 1000.0).astype(np.float32), quantised=True)
        if prec == "exact":
            print((B, H, W), "audit (max err quanta, max err/bound, flagged, audited):", eng.audit_tie_bound(1.0))
        eng.step(1.0)
        res[prec] = tuple(k(x) for x in eng.download_planes())
        print((B, H, W), prec, eng.kernel_info()[:40], "fixups", eng.last_fixup_count())
        eng.close()
    for prec, (kl, kd) in res.items():
        bl, bd = kl != k(ref[:, 1]), kd != k(ref[:, 2])
        print("   ", prec, "differing cells light/dark:", int(bl.sum()), int(bd.sum()))
        for pos in list(zip(*np.nonzero(bl)))[:6]:
            b, r, c = pos
            print("       light", pos, "in l,d =", light[pos], dark[pos], "ours", kl[pos], "oracle", k(ref[:, 1])[pos],
                  "f64-mode", res["f64"][0][pos], "oracle un-rounded pre-clip?", ref[b, 1, r, c])
        for pos in list(zip(*np.nonzero(bd)))[:6]:
            b, r, c = pos
            print("       dark ", pos, "in l,d =", light[pos], dark[pos], "ours", kd[pos], "oracle", k(ref[:, 2])[pos],
                  "f64-mode", res["f64"][1][pos])
