#!/usr/bin/env python3
"""Fuzz of the exact mode against the float64 C oracle: random shapes (every kernel family: generic,
tiled, wave-strip rotate / halo / general / packed, fused pairs), random physics constants in the ranges
callers use, random un-quantised initial states in all three upload formats (the first step: step_first_stream /
step_generic), random luminosity schedules, in a fifth of the cases with the LDS queue / mismatch list shrunk so that
the overflow fallbacks run; planes and reductions compared bit for bit after every run.
The audit of the tie bound (dw_audit_tie_bound) is evaluated along the way.

usage: fuzz_exact.py [cases=100] [seed=1] [only=<case index>]"""
import os

os.environ.setdefault("DW_TEST_HOOKS", "1")     # the DW_TEST_* queue caps below are honoured only under it
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

os.environ.setdefault("DW_PACK_MIN_STRIPS", "1")
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402
from oracle import c_oracle  # noqa: E402


def run_case(seed, i, log=None, explain=False):
    """One case; returns (identical, worst float32 error / tie bound).  Appends its description to `log`."""
    c_oracle.build()
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
    # a side generator (the case itself stays what it was): a third of the cases keep the library's own kernel selection,
    # i.e. small batches of narrow worlds take the tiled / generic kernels instead of the packed wave-strips
    if np.random.RandomState((seed * 7919 + i * 104729 + 1) % (2 ** 32)).rand() < 0.33:
        caps = dict(caps, DW_PACK_MIN_STRIPS="512")
    saved = {k: os.environ.pop(k, None) for k in ("DW_TEST_QUEUE_CAP", "DW_TEST_MISMATCH_CAP")}
    saved["DW_PACK_MIN_STRIPS"] = os.environ.get("DW_PACK_MIN_STRIPS")
    os.environ.update(caps)
    try:
        for k, v in over.items():
            setattr(p, k, v)
        eng = amd.Engine(p)
    finally:
        for k, v in saved.items():                      # (read by the library at handle creation)
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    fmt = str(rng.choice(["philox", "f64", "f32"], p=[0.6, 0.2, 0.2]))   # the un-quantised state the first step reads
    if fmt == "philox":
        eng.init_random(i + 7)
    else:
        cover = float(rng.choice([0.2, 0.6, 1.0]))              # sparse like the reference's initial grid ... dense
        l0 = cover * rng.rand(B, H, W) * (rng.rand(B, H, W) < rng.uniform(0.2, 0.9))
        d0 = cover * rng.rand(B, H, W) * (rng.rand(B, H, W) < rng.uniform(0.2, 0.9))
        if fmt == "f64":
            eng.upload_state(l0, d0)
        else:
            eng.upload_state_f32(l0.astype(np.float32), d0.astype(np.float32), quantised=False)
    light, dark = eng.download_planes()
    Lg = eng.step_n(steps, L0, dL, 0.6, 1.8)
    Lo = c_oracle.step_n(light, dark, L0, dL, steps, 0.6, 1.8, params=c_oracle.OracleParams.defaults(**over))
    gl, gd = eng.download_planes()
    kl, kd, ol, od = np.rint(gl * 1000), np.rint(gd * 1000), np.rint(light * 1000), np.rint(dark * 1000)
    s = eng.reduce()
    el, ed = kl.sum(axis=(1, 2)).astype(np.uint64), kd.sum(axis=(1, 2)).astype(np.uint64)
    em = np.maximum(kl.max(axis=(1, 2)), kd.max(axis=(1, 2))).astype(np.uint32)
    same = bool(Lg == Lo and np.array_equal(kl, ol) and np.array_equal(kd, od) and np.array_equal(s["sum_light_k"], el)
                and np.array_equal(s["sum_dark_k"], ed) and np.array_equal(s["max_k"], em))
    ratio = eng.audit_tie_bound(Lg)[1]                  # max float32 error / tie bound over the current state
    desc = (f"{'ok  ' if same else 'FAIL'} {i} B={B} H={H} W={W} {fmt} steps={steps} L0={L0:.3f} dL={dL:+.4f} {over} {caps} "
            f"err/bound={ratio:.3f} :: {eng.kernel_info()[:36]}")
    if log is not None:
        log.append(desc)
    if not same and explain:                            # say where it differs
        dl, dd = np.argwhere(kl != ol), np.argwhere(kd != od)
        print("L", Lg, Lo, "light cells differing", len(dl), dl[:12].tolist(), "dark", len(dd), dd[:12].tolist())
        for b in range(B):
            if s["sum_light_k"][b] != el[b] or s["sum_dark_k"][b] != ed[b] or s["max_k"][b] != em[b]:
                print("  world", b, "sum_light", int(s["sum_light_k"][b]), int(el[b]), "sum_dark", int(s["sum_dark_k"][b]), int(ed[b]),
                      "max", int(s["max_k"][b]), int(em[b]))
    eng.close()
    return same, ratio


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    bad, worst = 0, 0.0
    for i in range(cases):
        if only >= 0 and i != only:
            continue
        log = []
        same, ratio = run_case(seed, i, log, explain=only >= 0)
        worst = max(worst, ratio)
        bad += not same
        print(log[-1], flush=True)
    print(f"{cases - bad}/{cases} cases bit-identical to the float64 oracle; worst float32 error / tie bound = {worst:.3f}")
    sys.exit(1 if bad else 0)
