#!/usr/bin/env python3
"""cProfile of the lifespan harness on the README's grid (1000 worlds of 8x8, 4 agents): where the host time of one
policy's sweep goes.  usage: sweep_profile.py [policy=antigreedy|greedy|random|half_random|none] [dim=8] [worlds=1000]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd.harness import simulate_lifespan  # noqa: E402

policy = sys.argv[1] if len(sys.argv) > 1 else "antigreedy"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 8
worlds = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
agent = {"greedy": lambda: amd.Greedy(epsilon=0.0), "antigreedy": lambda: amd.Greedy(epsilon=0.0, greedy=False),
         "random": lambda: amd.Greedy(epsilon=1.0), "half_random": lambda: amd.Greedy(epsilon=0.5), "none": lambda: None}[policy]()
np.random.seed(13)
env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=4)
env.batch_size = worlds
simulate_lifespan(env, agent)                                  # warm: device handle, kernels
for only in (False, True):
    t0 = time.perf_counter()
    done_at, _ = simulate_lifespan(env, agent, final_state=not only)
    dt = time.perf_counter() - t0
    print(f"{policy} dim {dim} x {worlds} worlds, lifespans_only={only}: {dt * 1e3:.2f} ms, {env.step_count} steps, "
          f"{dt / env.step_count * 1e6:.1f} us/step, mean lifespan {done_at.mean():.3f}")
pr = cProfile.Profile()
pr.enable()
simulate_lifespan(env, agent)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
env.close()
