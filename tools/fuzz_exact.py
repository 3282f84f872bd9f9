#!/usr/bin/env python3
"""Fuzz of the exact mode against the float64 C oracle: random shapes (every kernel family: generic,
tiled, wave-strip rotate / halo / general / packed, fused pairs), random physics constants in the ranges
callers use, random luminosity schedules, in a fifth of the cases with the LDS queue / mismatch list shrunk so that
the overflow fallbacks run; planes and reductions compared bit for bit after every run.
The audit of the tie bound (dw_audit_tie_bound) is evaluated along the way.

usage: fuzz_exact.py [cases=100] [seed=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

os.environ.setdefault("DW_PACK_MIN_STRIPS", "1")
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402
from oracle import c_oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c_oracle.build()
bad, worst = 0, 0.0
for i in range(cases):
    rng = np.random.RandomState(seed * 100000 + i)
    W = int(rng.choice([8, 12, 16, 20, 32, 64, 96, 128, 192, 256, 258, 260, 320, 512, 516, 1024]))
    H = int(rng.randint(3, 140))
    B = int(rng.randint(1, 8 if W * H > 20000 else 40))
    over = {}
    if rng.rand() < 0.7:
        al = float(rng.uniform(0.5, 0.95))
        ad = float(rng.uniform(0.05, 0.5))
        over.update(albedo_light=al, albedo_dark=ad)
    if rng.rand() < 0.3:
        over.update(albedo_bare=float(rng.uniform(0.4, 0.6)))
    if rng.rand() < 0.5:
        over.update(q2=float(rng.choice([0.0, 1.0, 0.5, 2.0])) * (0.2 * 1000.0 / 5.67e-8) / 8.0)
    if rng.rand() < 0.5:
        over.update(dt=float(rng.choice([0.25, 0.5, 1.0, 2.0])))
    if rng.rand() < 0.4:
        over.update(gamma=float(rng.uniform(0.1, 0.4)), g=float(rng.uniform(0.002, 0.005)),
                    temp_optimal=float(rng.uniform(285.0, 305.0)))
    steps = int(rng.randint(1, 16))
    L0, dL = float(rng.uniform(0.7, 1.5)), float(rng.uniform(-0.01, 0.02))
    p = amd.default_params(B, H, W, 0)
    p.precision = _ffi.PRECISION["exact"]
    caps = {}
    if rng.rand() < 0.2:                                # shrink the LDS queue / mismatch list: overflow fallbacks
        caps = {"DW_TEST_QUEUE_CAP": str(int(rng.choice([1, 4, 16]))), "DW_TEST_MISMATCH_CAP": str(int(rng.choice([0, 1, 2])))}
    for k in ("DW_TEST_QUEUE_CAP", "DW_TEST_MISMATCH_CAP"):
        os.environ.pop(k, None)
    os.environ.update(caps)
    for k, v in over.items():
        setattr(p, k, v)
    eng = amd.Engine(p)
    eng.init_random(i + 7)
    light, dark = eng.download_planes()
    Lg = eng.step_n(steps, L0, dL, 0.6, 1.8)
    Lo = c_oracle.step_n(light, dark, L0, dL, steps, 0.6, 1.8, params=c_oracle.OracleParams.defaults(**over))
    gl, gd = eng.download_planes()
    kl, kd, ol, od = np.rint(gl * 1000), np.rint(gd * 1000), np.rint(light * 1000), np.rint(dark * 1000)
    s = eng.reduce()
    same = (Lg == Lo and np.array_equal(kl, ol) and np.array_equal(kd, od)
            and np.array_equal(s["sum_light_k"], kl.sum(axis=(1, 2)).astype(np.uint64))
            and np.array_equal(s["max_k"], np.maximum(kl.max(axis=(1, 2)), kd.max(axis=(1, 2))).astype(np.uint32)))
    ratio = eng.audit_tie_bound(Lg)[1]                  # max float32 error / tie bound over the current state
    worst = max(worst, ratio)
    bad += not same
    print(("ok  " if same else "FAIL"), i, f"B={B} H={H} W={W} steps={steps} L0={L0:.3f} dL={dL:+.4f} {over} {caps} "
          f"err/bound={ratio:.3f} :: {eng.kernel_info()[:36]}", flush=True)
    eng.close()
print(f"{cases - bad}/{cases} cases bit-identical to the float64 oracle; worst float32 error / tie bound = {worst:.3f}")
sys.exit(1 if bad else 0)
