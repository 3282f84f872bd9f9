#!/usr/bin/env python3
"""Evolution-strategy fitness evaluation (ref daisy/evo/sges.py:144-181, :314-349) as one batched ensemble:
P random MLP policies x `wpm` worlds each, device-resident in chunks (dw_run_episode_mlp).

usage: es_fitness_bench.py [P] [wpm] [dim] [max_steps] [chunk]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd.harness import get_fitness_population  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wpm = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 16
max_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 768
chunks = [int(c) for c in sys.argv[5].split(",")] if len(sys.argv) > 5 else [64, 1]
np.random.seed(7)
pop = [amd.MLP() for _ in range(P)]
for chunk in chunks + chunks[:1]:
    np.random.seed(11)
    env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=4)
    t0 = time.perf_counter()
    res = get_fitness_population(env, pop, worlds_per_member=wpm, max_steps=max_steps, chunk=chunk)
    dt = time.perf_counter() - t0
    steps = env.step_count
    # later generations on the same environment, as in an ES loop: no device handle to (re)create, kernels loaded
    later = []
    for _ in range(3):
        t0 = time.perf_counter()
        res2 = get_fitness_population(env, pop, worlds_per_member=wpm, max_steps=max_steps, chunk=chunk)
        later.append((time.perf_counter() - t0, env.step_count))
    dt2, steps2 = min(later)
    print(f"P={P} x {wpm} worlds of {dim}x{dim}, 4 agents, chunk={chunk}: first generation (creates the device handle) "
          f"{steps} steps in {dt:.3f} s = {dt / steps * 1e6:.0f} us/step; later generations (best of 3) {steps2} steps in "
          f"{dt2:.3f} s = {dt2 / steps2 * 1e6:.0f} us/step, {P * wpm * 4 * steps2 / dt2 / 1e6:.2f} M agent-steps/s "
          f"(all three: {', '.join(f'{t * 1e3:.1f} ms' for t, _ in later)}); best fitness {max(r[0] for r in res):.4f}", flush=True)
    env.close()
