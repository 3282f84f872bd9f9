#!/usr/bin/env python3
"""Fuzz of the ES fitness harness (device-resident MLP episodes in chunks, early stop replay) against the
reference's get_fitness arithmetic (daisy/evo/sges.py:144-181) run on the NumPy oracle environment with
OracleMLP policies: random grids, agent counts, hunger rates (agents starve inside the run), chunk sizes
and step limits; fitness, counters and the state the environment is left in must be identical.

usage: fuzz_fitness.py [cases=30] [seed=1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd.harness import get_fitness  # noqa: E402
from oracle import daisy_oracle as O  # noqa: E402


def oracle_get_fitness(env, agent, adversary, max_steps):
    """ref sges.py:144-181 verbatim on the oracle environment."""
    obs = env.reset()
    B, N = obs.shape[:2]
    half = N // 2
    done_at = np.zeros((B, N, 1), dtype=int)
    total_steps, sum_reward, all_done = 0, 0.0, False
    while not all_done and env.step_count < max_steps:
        a = np.append(agent.get_action(obs[:, :half]), adversary.get_action(obs[:, half:]), axis=1)
        obs, reward, done, _ = env.step(a)
        all_done = (np.ones_like(done).sum() - done.sum()) == 0
        done_at += (1 - 1 * done)
        sum_reward += (reward[:, :half]).mean()
        total_steps = total_steps + (1 - 1 * done)
    return sum_reward / (B * N), total_steps, done_at


def run_case(seed, log):
    rng = np.random.RandomState(seed)
    dim = int(rng.choice([8, 16, 24, 64]))
    N = int(rng.choice([2, 4, 6]))
    B = int(rng.choice([1, 4, 9]))
    gamma_a = float(rng.choice([0.05, 0.1, 0.25]))          # larger: the agents starve, the episode stops early
    max_steps = int(rng.choice([17, 33, 64, 100]))
    chunk = int(rng.choice([1, 4, 16, 64]))
    pa = rng.randn(1808) * float(rng.choice([0.0, 0.3, 1.0]))   # 0: action 0 forever (never grazes)
    pb = rng.randn(1808) * 0.5
    outs = []
    for which in ("oracle", "device"):
        np.random.seed(seed)
        if which == "oracle":
            env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=dim, n_agents=N)
            env.P.batch_size, env.P.agent_gamma = B, gamma_a
            f, total, done_at = oracle_get_fitness(env, O.OracleMLP(pa), O.OracleMLP(pb), max_steps)
        else:
            env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=N)
            env.batch_size, env.agent_gamma = B, gamma_a
            a, b = amd.MLP(), amd.MLP()
            a.set_parameters(pa)
            b.set_parameters(pb)
            np.random.seed(seed)                             # MLP() drew Glorot weights: rewind for the reset
            env2 = amd.RLDaisyWorld(grid_dimension=dim, n_agents=N)
            env.close()
            env = env2
            env.batch_size, env.agent_gamma = B, gamma_a
            f, total, done_at = get_fitness(env, a, b, max_steps=max_steps, chunk=chunk)
        outs.append((np.float64(f), np.asarray(total), np.asarray(done_at), env.grid.copy(),
                     np.asarray(env.agent_indices).copy(), np.asarray(env.agent_states).copy(), np.float64(env.L),
                     np.int64(env.step_count)))
        if which == "device":
            env.close()
    names = ("fitness", "total_steps", "done_at", "grid", "agent_indices", "agent_states", "L", "step_count")
    bad = [n for n, x, y in zip(names, outs[0], outs[1]) if not np.array_equal(x, y)]
    info = f"dim={dim} B={B} N={N} agent_gamma={gamma_a} max_steps={max_steps} chunk={chunk} steps={int(outs[0][7])}"
    if bad:
        log.append(f"seed {seed}: {info}: differs in {bad}")
    return not bad, info


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
