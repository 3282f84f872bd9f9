#!/usr/bin/env python3
"""fast_tolerance.py — what the float32-only mode (DW_PRECISION_FAST) deviates from the float64 reference by.

SURVEY.md 8(c) states the tolerance of a float32 path as: from IDENTICAL states, every cell within one
quantum (1e-3) and >= 99.95 % of the cells exact after one step; population means of a trajectory within
1e-3.  This tool measures both on DEVELOPED states over a whole luminosity ramp (not only on synthetic
random states):

  per-step   an exact-mode engine walks the ramp (its states are the float64 reference's, bit for bit:
             tests + soak, and re-checked here against the C oracle at a few steps); every `--every`
             steps its state is uploaded into a fast-mode engine, both take the same step, and the
             differing cells are counted (light and dark separately);
  trajectory a fast-mode engine walks the same ramp on its own from the same initial state; per step the
             ensemble means of light and dark are compared with the exact engine's.

Several builds of the library can be compared in one process (--libs name=path,...): the default build and
e.g. the DW_FAST_SPLIT=1 variant (hi/lo coefficient chain in the float32-only mode too).

    python tools/fast_tolerance.py --worlds 64 --grid 256 --steps 512 --every 4 --out profiles/r02_fast_tolerance.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402


def k(x):
    return np.rint(np.asarray(x) * 1000.0).astype(np.int32)


def engine(B, G, precision, lib=None):
    p = amd.default_params(B, G, G, 0)
    p.precision = _ffi.PRECISION[precision]
    return amd.Engine(p, lib_path=lib)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=64)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--every", type=int, default=4)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--libs", default="", help="name=path,... (default: the in-tree library)")
    ap.add_argument("--oracle-checks", type=int, default=4, help="steps at which the exact engine is re-checked against the C oracle")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    B, G = a.worlds, a.grid
    libs = [("default", None)]
    if a.libs:
        libs = [tuple(x.split("=", 1)) for x in a.libs.split(",")]
    min_L, max_L, dL = 0.75, 1.5, 0.75 / 512

    ex = engine(B, G, "exact")
    ex.init_random(a.seed)
    light0, dark0 = ex.download_planes()
    fast_step = {name: engine(B, G, "fast", path) for name, path in libs}      # re-seeded from the exact state
    fast_traj = {name: engine(B, G, "fast", path) for name, path in libs}      # walks the ramp on its own
    for e in fast_traj.values():
        e.init_random(a.seed)
    per_step = {name: [] for name, _ in libs}
    drift = {name: [] for name, _ in libs}
    oracle_at = set(np.linspace(a.every, a.steps - 1, a.oracle_checks).astype(int) // a.every * a.every) if a.oracle_checks else set()
    oracle_ok = []
    L = min_L
    for t in range(a.steps):
        sample = t > 0 and t % a.every == 0                   # t = 0: the state is not quantised yet
        if sample:
            sl, sd = ex.download_planes()
            for name, e in fast_step.items():
                e.upload_state_f32(sl.astype(np.float32), sd.astype(np.float32), quantised=True)
        ex.step(L)
        if sample:
            rl, rd = (k(x) for x in ex.download_planes())
            if t in oracle_at:
                from oracle import c_oracle
                ref = c_oracle.forward(sl, sd, L)
                oracle_ok.append(bool(np.array_equal(k(ref[:, 1]), rl) and np.array_equal(k(ref[:, 2]), rd)))
            for name, e in fast_step.items():
                e.step(L)
                fl, fd = (k(x) for x in e.download_planes())
                dl, dd = np.abs(fl - rl), np.abs(fd - rd)
                per_step[name].append({"t": t, "L": L, "frac_diff_light": float(np.count_nonzero(dl)) / dl.size,
                                       "frac_diff_dark": float(np.count_nonzero(dd)) / dd.size,
                                       "max_diff_quanta": int(max(dl.max(), dd.max())),
                                       "mean_light": float(rl.mean()) / 1000.0, "mean_dark": float(rd.mean()) / 1000.0})
        s_ex = ex.reduce()
        for name, e in fast_traj.items():
            e.step(L)
            s = e.reduce()
            n = float(B) * G * G * 1000.0
            drift[name].append((abs(float(s["sum_light_k"].sum()) - float(s_ex["sum_light_k"].sum())) / n,
                                abs(float(s["sum_dark_k"].sum()) - float(s_ex["sum_dark_k"].sum())) / n,
                                # the worst single world of the ensemble
                                float(np.abs(s["sum_light_k"].astype(np.int64) - s_ex["sum_light_k"].astype(np.int64)).max()) / (G * G * 1000.0),
                                float(np.abs(s["sum_dark_k"].astype(np.int64) - s_ex["sum_dark_k"].astype(np.int64)).max()) / (G * G * 1000.0)))
        L = min(max(L + dL, min_L), max_L)
        if t % 64 == 0:
            print(f"step {t}/{a.steps}", flush=True)
    out = {"worlds": B, "grid": G, "steps": a.steps, "every": a.every, "seed": a.seed,
           "exact_engine_equals_c_oracle_at_checked_steps": oracle_ok, "builds": {}}
    for name, _ in libs:
        ps = per_step[name]
        fr = np.array([0.5 * (p["frac_diff_light"] + p["frac_diff_dark"]) for p in ps])
        d = np.array(drift[name])
        out["builds"][name] = {
            "per_step_frac_cells_differing": {"mean": float(fr.mean()), "median": float(np.median(fr)),
                                              "p95": float(np.percentile(fr, 95)), "max": float(fr.max()),
                                              "argmax_t": int(ps[int(fr.argmax())]["t"])},
            "per_step_frac_identical_min": float(1.0 - fr.max()),
            "per_step_max_diff_quanta": int(max(p["max_diff_quanta"] for p in ps)),
            "trajectory_ensemble_mean_drift": {"light_max": float(d[:, 0].max()), "dark_max": float(d[:, 1].max())},
            "trajectory_single_world_mean_drift": {"light_max": float(d[:, 2].max()), "dark_max": float(d[:, 3].max())},
            "samples": ps[:: max(1, len(ps) // 32)],
        }
        b = out["builds"][name]
        print(f"{name}: per-step differing cells mean {fr.mean():.3e} p95 {np.percentile(fr, 95):.3e} max {fr.max():.3e} "
              f"(t={b['per_step_frac_cells_differing']['argmax_t']}), max |diff| {b['per_step_max_diff_quanta']} quanta; "
              f"trajectory drift of the ensemble mean: light {d[:, 0].max():.2e} dark {d[:, 1].max():.2e}; worst single world: "
              f"light {d[:, 2].max():.2e} dark {d[:, 3].max():.2e}")
    print("exact engine == C oracle at checked steps:", oracle_ok)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
