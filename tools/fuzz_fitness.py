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


def run_population_case(seed, log):
    """get_fitness_population (a whole ES population as one ensemble) against the same arithmetic per member
    on the oracle environment: members stop at different steps, the ensemble runs on until the last one."""
    from therldaisyworld_amd.harness import get_fitness_population
    rng = np.random.RandomState(seed)
    dim = int(rng.choice([8, 16, 24]))
    N = int(rng.choice([2, 4]))
    P = int(rng.randint(2, 6))
    wpm = int(rng.randint(1, 5))
    gamma_a = float(rng.choice([0.05, 0.15, 0.3]))
    max_steps = int(rng.choice([20, 45, 80]))
    chunk = int(rng.choice([1, 6, 64]))
    params = rng.randn(P, 1808) * rng.choice([0.0, 0.3, 1.0], size=(P, 1))
    adversary_of = rng.randint(P, size=P)
    half = N // 2
    np.random.seed(seed)
    env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=N)
    env.agent_gamma = gamma_a
    res = get_fitness_population(env, list(params), adversary_of, worlds_per_member=wpm, max_steps=max_steps, chunk=chunk)
    dev_final = (env.grid.copy(), np.asarray(env.agent_indices).copy(), np.int64(env.step_count))
    env.close()
    np.random.seed(seed)
    ref = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=dim, n_agents=N)
    ref.P.batch_size, ref.P.agent_gamma = P * wpm, gamma_a
    obs = ref.reset()
    nets = [O.OracleMLP(w) for w in params]
    sum_reward = np.zeros(P)
    done_at = np.zeros((P * wpm, N, 1), dtype=int)
    running = np.ones(P, dtype=bool)
    while running.any() and ref.step_count < max_steps:
        acts = []
        for m in range(P):
            o = obs[m * wpm:(m + 1) * wpm]
            acts.append(np.append(nets[m].get_action(o[:, :half]), nets[adversary_of[m]].get_action(o[:, half:]), axis=1))
        obs, reward, done, _ = ref.step(np.concatenate(acts, axis=0))
        live = np.repeat(running, wpm)[:, None, None]
        done_at += live * (1 - 1 * done)
        for m in range(P):
            if running[m]:
                sum_reward[m] += reward[m * wpm:(m + 1) * wpm, :half].mean()
                running[m] = not done[m * wpm:(m + 1) * wpm].all()
    bad = []
    for m in range(P):
        if res[m][0] != sum_reward[m] / (wpm * N):
            bad.append(f"fitness[{m}]")
        if not np.array_equal(np.array(res[m][2]), done_at[m * wpm:(m + 1) * wpm]):
            bad.append(f"done_at[{m}]")
    if not np.array_equal(dev_final[0], ref.grid):
        bad.append("grid")
    if not np.array_equal(dev_final[1], ref.agent_indices):
        bad.append("agent_indices")
    if dev_final[2] != ref.step_count:
        bad.append("step_count")
    info = f"population dim={dim} P={P} wpm={wpm} N={N} agent_gamma={gamma_a} max_steps={max_steps} chunk={chunk} steps={ref.step_count}"
    if bad:
        log.append(f"seed {seed}: {info}: differs in {bad}")
    return not bad, info


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    log, nbad = [], 0
    for i in range(cases):
        ok, info = (run_population_case if i % 3 == 2 else run_case)(seed * 10000 + i, log)
        nbad += not ok
        print("ok  " if ok else "FAIL", seed * 10000 + i, info, flush=True)
    for line in log:
        print(line)
    print(f"{cases - nbad}/{cases} cases identical")
    sys.exit(1 if nbad else 0)
