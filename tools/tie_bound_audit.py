#!/usr/bin/env python3
"""Audit of the exact mode's tie bound: for every cell of developed states over whole luminosity
ramps, |gq_float32 - gq_float64| against the per-cell bound eps the kernels test with (dw_audit_tie_bound), for
the default constants and mutated sets, on random and on adversarially dense states.

    python tools/tie_bound_audit.py [--out profiles/r03_tie_bound_audit.json] [--worlds 8] [--dim 256] [--steps 256]

Prints / writes per case: worst error (quanta), worst error / eps (must stay < 1; the tests assert < 0.5), the
fraction of cell values the tie test flags."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402

CASES = [("default", {}), ("neutral albedos", dict(albedo_light=0.5, albedo_dark=0.5)), ("no microclimate", dict(q2=0.0)),
         ("dt=2, albedos 0.8/0.2", dict(dt=2.0, albedo_light=0.8, albedo_dark=0.2)),
         ("gamma=0.3, To=290, g=0.004", dict(gamma=0.3, temp_optimal=290.0, g=0.004)), ("dt=0.5", dict(dt=0.5)),
         ("asymmetric albedos 0.7/0.5/0.2", dict(albedo_light=0.7, albedo_dark=0.2)),
         ("p=0.9", dict(p=0.9)), ("dt=-1", dict(dt=-1.0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--worlds", type=int, default=8)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--steps", type=int, default=256)
    a = ap.parse_args()
    rows = []
    rng = np.random.RandomState(5)
    for name, over in CASES:
        for state in ("philox", "dense"):
            p = amd.default_params(a.worlds, a.dim, a.dim, 0)
            p.precision = _ffi.PRECISION["exact"]
            for k, v in over.items():
                setattr(p, k, v)
            eng = amd.Engine(p)
            if state == "philox":
                eng.init_random(17)
            else:       # covers up to 1 in BOTH species: total density up to 2, bare fraction down to -1
                light = np.rint(rng.rand(a.worlds, a.dim, a.dim) * 1000) / 1000 * (rng.rand(a.worlds, a.dim, a.dim) > 0.2)
                dark = np.rint(rng.rand(a.worlds, a.dim, a.dim) * 1000) / 1000 * (rng.rand(a.worlds, a.dim, a.dim) > 0.2)
                eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=True)
            L, dL = 0.75, 0.75 / a.steps
            L = eng.step_n(1, L, dL, 0.75, 1.5)
            worst_ratio = worst_err = 0.0
            flagged = total = 0
            for _ in range(a.steps):
                err, ratio, nf, n = eng.audit_tie_bound(L)
                worst_ratio, worst_err = max(worst_ratio, ratio), max(worst_err, err)
                flagged += nf
                total += n
                L = eng.step_n(1, L, dL, 0.75, 1.5)
            eng.close()
            row = {"constants": name, "state": state, "worst_error_quanta": worst_err, "worst_error_over_eps": worst_ratio,
                   "flagged_fraction": flagged / total, "cell_values_audited": total}
            rows.append(row)
            print(f"{name:32s} {state:7s} worst err {worst_err:.3e} quanta, err/eps {worst_ratio:.3f}, flagged "
                  f"{100 * flagged / total:.4f} % of {total:.2e} cell values", flush=True)
    out = {"tool": "tools/tie_bound_audit.py", "library_build_id": _ffi.load().dw_build_id().decode(),
           "worlds": a.worlds, "dim": a.dim, "steps": a.steps, "rows": rows,
           "worst_error_over_eps": max(r["worst_error_over_eps"] for r in rows)}
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}))


if __name__ == "__main__":
    main()
