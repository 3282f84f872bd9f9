#!/usr/bin/env python3
"""cProfile of the drop-in's per-step loop `obs, r, d, _ = env.step(agent(obs))` (the way the reference's notebooks and
agents/greedy.py drive the environment).  usage: dropin_profile.py [dim=8] [B=1000] [N=4] [steps=300]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
np.random.seed(0)
env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=N)
env.batch_size = B
obs = env.reset()
agent = amd.Greedy()
for _ in range(10):
    obs, r, d, _ = env.step(agent(obs))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    obs, r, d, _ = env.step(agent(obs))
pr.disable()
print(f"dim={dim} B={B} N={N}: {steps} steps")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
env.close()
