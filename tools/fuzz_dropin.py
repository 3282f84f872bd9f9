#!/usr/bin/env python3
"""Fuzz of the drop-in class against the NumPy oracle environment: random sequences of what callers do
to an RLDaisyWorld - step with full / sub-shaped / None / float actions, in-place edits and assignments
of grid / agent_states / agent_indices, get_obs on other positions, update_agents and forward as plain
methods, attribute changes followed by reset - compared after every operation (observations, rewards, done
flags, grid, agents, L, step_count), with the legacy NumPy RNG stream kept in lock step.

usage: fuzz_dropin.py [cases=40] [seed=1]"""
import os

os.environ.setdefault("DW_TEST_HOOKS", "1")     # the DW_TEST_* queue caps below are honoured only under it
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import therldaisyworld_amd as amd  # noqa: E402
from oracle import daisy_oracle as O  # noqa: E402


class Oracle:
    """OracleDaisyWorld behind the attribute names of the reference class."""

    def __init__(self, **kw):
        self.e = O.OracleDaisyWorld.like_reference_ctor(**kw)

    def __getattr__(self, k):
        e = object.__getattribute__(self, "e")
        return getattr(e, k) if hasattr(e, k) else getattr(e.P, k)

    def __setattr__(self, k, v):
        if k == "e":
            return object.__setattr__(self, k, v)
        e = self.e
        if k in ("grid", "agent_indices", "agent_states", "L", "dL", "step_count"):
            setattr(e, k, v)
        else:
            setattr(e.P, k, v)


def same(a, b, exact=True, rtol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    # the float64 caches (beta, growth) cross zero: a relative tolerance alone would flag 1e-14 differences
    return np.array_equal(a, b) if exact else np.allclose(a, b, rtol=rtol, atol=1e-300 if rtol <= 1e-12 else 1e-11)


def _run_case(seed, log):
    rng = np.random.RandomState(seed)
    dim = int(rng.choice([5, 8, 12, 16, 20, 64, 256]))
    N = int(rng.randint(0, 5))
    kw = dict(grid_dimension=dim, n_agents=N)
    if rng.rand() < 0.25:
        kw["collision_mode"] = 1                               # RNG-coupled collision pass (ref :220-242)
    if rng.rand() < 0.3:
        kw["neighborhood_mode"] = str(rng.choice(["moore", "circular", "von_neumann"]))
    envs = []
    for make in (lambda: Oracle(**kw), lambda: amd.RLDaisyWorld(**kw)):
        np.random.seed(seed)
        envs.append(make())
    ora, dev = envs
    states = [None, None]                                  # each env's private copy of the global RNG state
    np.random.seed(seed + 1)
    states[0] = states[1] = np.random.get_state()
    quantised = False                                      # un-rounded temperature channels after reset: FFT noise
    stepped = False                                        # the side-effect caches are those of the last physics pass

    def both(fn, exact=True, what="", rtol=1e-12):
        outs = []
        for i, e in enumerate((ora, dev)):
            np.random.set_state(states[i])
            outs.append(fn(e))
            states[i] = np.random.get_state()
        flags = [same(x, y, exact, rtol) for x, y in zip(outs[0], outs[1])]
        rng_ok = np.array_equal(states[0][1], states[1][1]) and states[0][2] == states[1][2]
        ok = all(flags) and rng_ok
        if not ok:
            detail = []
            for i, (x, y) in enumerate(zip(outs[0], outs[1])):
                if not flags[i]:
                    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
                    if x.shape != y.shape:
                        detail.append(f"out[{i}] shapes {x.shape} vs {y.shape}")
                    else:
                        d = np.abs(x - y)
                        w = np.unravel_index(np.argmax(d), d.shape) if d.size else ()
                        detail.append(f"out[{i}] max|diff| {d.max() if d.size else 0:.3g} at {w} ({np.count_nonzero(d)} differ)")
            log.append(f"seed {seed}: mismatch after {what} exact={exact} rng_ok={rng_ok}: " + "; ".join(detail))
        return ok

    def snapshot(e):
        return (e.grid.copy(), np.asarray(e.agent_indices).copy(), np.asarray(e.agent_states).copy(),
                np.float64(e.L), np.float64(e.dL), np.int64(e.step_count))

    B = int(rng.choice([1, 3, 6])) if dim <= 64 else int(rng.choice([1, 2]))

    def do_reset(e):
        e.batch_size = B
        obs = e.reset()
        return (obs, *snapshot(e))

    if not both(do_reset, exact=False, what="reset"):
        return False
    for step in range(int(rng.randint(6, 16))):
        op = rng.choice(["step", "step", "step", "step_none", "step_sub", "edit_grid", "assign_grid", "edit_agents",
                         "get_obs", "update_agents", "forward", "mutate_reset", "read_caches", "read_caches",
                         "mutate_live", "reset_plain"])
        n_now = int(dev.n_agents)
        if op == "step" or (op in ("step_sub", "edit_agents", "get_obs", "update_agents") and n_now == 0):
            a = rng.randint(9, size=(B, n_now, 1)).astype(float if rng.rand() < 0.3 else int) if n_now else None
            fn = lambda e, a=a: (*e.step(a)[:3], *snapshot(e))
        elif op == "step_none":
            fn = lambda e: (*e.step(None)[:3], *snapshot(e))
        elif op == "step_sub":
            a = rng.randint(9, size=(int(rng.randint(1, B + 1)), int(rng.randint(1, n_now + 1)), 1))
            fn = lambda e, a=a: (*e.step(a)[:3], *snapshot(e))
        elif op == "edit_grid":
            f = float(rng.uniform(0.3, 0.9))
            def fn(e, f=f):
                e.grid[:, 1] *= f
                e.grid[:, 2, ::2] = 0.0
                return snapshot(e)
        elif op == "assign_grid":
            def fn(e):
                g = e.grid.copy()
                g[:, 1:3] = np.round(g[:, 1:3] * 0.75, 3)
                e.grid = g
                return snapshot(e)
        elif op == "edit_agents":
            v = float(rng.uniform(0.02, 0.9))
            pos = rng.randint(dim, size=2)
            def fn(e, v=v, pos=pos):
                e.agent_states[0, 0, 0] = v
                e.agent_indices[-1, -1] = pos
                return snapshot(e)
        elif op == "get_obs":
            idx = rng.randint(dim, size=(B, n_now, 2))
            fn = lambda e, idx=idx: (e.get_obs(idx), *snapshot(e))
        elif op == "update_agents":
            a = rng.randint(9, size=(B, n_now, 1))
            def fn(e, a=a):
                e.update_agents(a)
                return snapshot(e)
        elif op == "forward":
            def fn(e):
                new = e.forward(e.grid)
                return (new, *snapshot(e))
        elif op == "mutate_live":                           # constants the next physics pass must already see
            dt, ag = float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.02, 0.05, 0.1]))
            micro, newL = bool(rng.randint(2)), float(rng.uniform(0.8, 1.4))
            def fn(e, dt=dt, ag=ag, micro=micro, newL=newL):
                e.dt, e.agent_gamma = dt, ag
                e.q2 = e.q / 8.0 if micro else 0.0          # ref set_use_microclimate :85-92
                e.L = newL
                return snapshot(e)
        elif op == "reset_plain":
            def fn(e):
                obs = e.reset()
                return (obs, *snapshot(e))
        elif op == "read_caches":                           # what notebook_helpers.py:45-55 reads after a step
            if not stepped:
                continue
            def fn(e):
                return (e.temp, e.temp_light, e.temp_dark, e.beta, e.beta_l, e.beta_d, e.growth,
                        np.asarray(e.dead_temp, dtype=np.float64).reshape(-1)[:1])
        else:                                              # attribute changes take effect at the next reset
            nn = int(rng.randint(0, 5))
            al, ad = float(rng.uniform(0.55, 0.9)), float(rng.uniform(0.1, 0.45))
            def fn(e, nn=nn, al=al, ad=ad):
                e.n_agents = nn
                e.albedo_light, e.albedo_dark = al, ad
                e.min_L, e.max_L, e.ramp_period = 0.8, 1.4, 40
                e.batch_size = B
                obs = e.reset()
                return (obs, *snapshot(e))
        exact = quantised and op not in ("mutate_reset", "reset_plain")
        if op in ("step", "step_none", "step_sub"):
            exact = True                                   # a step rounds every channel it returns
        if op == "read_caches":
            if not both(fn, exact=False, what=f"{op} (operation {step})", rtol=1e-9):
                return False
            continue
        if not both(fn, exact=exact, what=f"{op} (operation {step})"):
            return False
        stepped = op in ("step", "step_none", "step_sub", "forward")
        if op in ("step", "step_none", "step_sub", "assign_grid"):
            quantised = op != "assign_grid" or quantised
        if op in ("mutate_reset", "reset_plain"):
            quantised = False
    dev.close()
    return True


def run_case(seed, log):
    """_run_case under a randomly shrunk repair queue / mismatch list in a third of the cases (a side generator: the
    case itself is unchanged): the exact mode's overflow fallbacks must give the same results."""
    crng = np.random.RandomState((int(seed) * 2654435761 + 12345) % (2 ** 32))
    caps = {}
    if crng.rand() < 0.33:
        caps = {"DW_TEST_QUEUE_CAP": str(int(crng.choice([1, 4, 16]))), "DW_TEST_MISMATCH_CAP": str(int(crng.choice([0, 1, 2])))}
    saved = {k: os.environ.pop(k, None) for k in ("DW_TEST_QUEUE_CAP", "DW_TEST_MISMATCH_CAP")}
    os.environ.update(caps)                              # (read by the library whenever a handle is created)
    try:
        out = _run_case(seed, log)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    if caps and log:
        log[-1] = f"{log[-1]} {caps}"
    return out


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    log, bad = [], 0
    for i in range(cases):
        ok = run_case(seed * 10000 + i, log)
        bad += not ok
        print("ok  " if ok else "FAIL", seed * 10000 + i, flush=True)
    for line in log:
        print(line)
    print(f"{cases - bad}/{cases} cases identical")
    sys.exit(1 if bad else 0)
