#!/usr/bin/env python3
"""Per-call latency of the drop-in class at the reference's default shapes (host overheads included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import therldaisyworld_amd as amd
for dim, B, N in ((16, 32, 4), (64, 32, 4), (8, 1000, 4), (256, 32, 4)):
    np.random.seed(0)
    env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=N)
    env.batch_size = B
    obs = env.reset()
    agent = amd.Greedy()
    for _ in range(5):
        obs, r, d, _ = env.step(agent(obs))
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        obs, r, d, _ = env.step(agent(obs))
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(20):
        g = env.grid; env._grid_m = None
    dg = (time.perf_counter() - t0) / 20
    print(f"dim={dim} B={B} N={N}: env.step(agent(obs)) {dt*1e6:.0f} us/step = {B*dim*dim/dt/1e6:.1f} Mcell-updates/s; env.grid {dg*1e6:.0f} us")
    env.close()
