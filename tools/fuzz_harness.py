#!/usr/bin/env python3
"""Fuzz of the lifespan harness (device-resident chunks, snapshots, step pairs with world flags) against
the notebook's loop on the NumPy oracle environment: random grid sizes (LDS-resident, packed, tiled,
wave-strip), agent counts, policies, luminosity ramps and chunk sizes; per-world and per-agent lifespans, the
state the environment is left in and the legacy RNG stream must all be identical.

usage: fuzz_harness.py [cases=30] [seed=1]"""
import os

os.environ.setdefault("DW_TEST_HOOKS", "1")     # the DW_TEST_* queue caps below are honoured only under it
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

os.environ.setdefault("DW_PACK_MIN_STRIPS", "1")
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd.harness import simulate_lifespan  # noqa: E402
from oracle import daisy_oracle as O  # noqa: E402


def _run_case(seed, log):
    rng = np.random.RandomState(seed)
    dim = int(rng.choice([8, 12, 16, 17, 20, 24, 31, 32, 64, 72, 128, 256]))
    N = int(rng.randint(1, 5))
    B = int(rng.choice([1, 3, 7])) if dim <= 72 else int(rng.choice([1, 2]))
    eps = float(rng.choice([0.0, 0.5, 1.0]))
    greedy = bool(rng.randint(2))
    no_agent = rng.rand() < 0.15
    ramp = int(rng.randint(20, 60))
    max_L = float(rng.uniform(1.7, 2.2))
    neutral = rng.rand() < 0.2
    chunk = int(rng.choice([5, 8, 32]))
    prec = "exact"
    outs = []
    for which in ("oracle", "device"):
        np.random.seed(seed)
        if which == "oracle":
            env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=dim, n_agents=N)
            P = env.P
            agent = None if no_agent else O.OracleGreedy(epsilon=eps, greedy=greedy)
        else:
            env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=N, precision=prec)
            P = env
            agent = None if no_agent else amd.Greedy(epsilon=eps, greedy=greedy)
        P.batch_size = B
        env.batch_size = B
        P.ramp_period, P.max_L, P.min_L = ramp, max_L, 0.9
        if neutral:
            P.albedo_light = P.albedo_dark = 0.5
        if which == "oracle":
            d, a = O.simulate_lifespan(env, agent)
        else:
            d, a = simulate_lifespan(env, agent, chunk=chunk)
        outs.append((d, a, env.grid.copy(), np.asarray(env.agent_indices).copy(), np.asarray(env.agent_states).copy(),
                     np.float64(env.L), np.int64(env.step_count), np.float64(np.random.rand())))
        if which == "device":
            env.close()
    names = ("done_at", "agents_done_at", "grid", "agent_indices", "agent_states", "L", "step_count", "next rand()")
    bad = [n for n, x, y in zip(names, outs[0], outs[1]) if not np.array_equal(np.asarray(x), np.asarray(y))]
    info = (f"dim={dim} B={B} N={N} eps={eps} greedy={greedy} no_agent={no_agent} ramp={ramp} max_L={max_L:.2f} "
            f"neutral={neutral} chunk={chunk} steps={int(outs[0][6])}")
    if bad:
        log.append(f"seed {seed}: {info}: differs in {bad}")
    return not bad, info


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
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    log, nbad = [], 0
    for i in range(cases):
        ok, info = run_case(seed * 10000 + i, log)
        nbad += not ok
        print("ok  " if ok else "FAIL", seed * 10000 + i, info, flush=True)
    for line in log:
        print(line)
    print(f"{cases - nbad}/{cases} cases identical")
    sys.exit(1 if nbad else 0)
