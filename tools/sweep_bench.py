#!/usr/bin/env python3
"""README lifespan sweep shape (B worlds of 8x8, 4 agents): device-resident episode loop vs per-step loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import therldaisyworld_amd as amd
from therldaisyworld_amd.harness import simulate_lifespan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for policy, agent in (("greedy", amd.Greedy(epsilon=0.0)), ("half_random", amd.Greedy(epsilon=0.5)), ("no", None)):
    for dev in (True, False):
        np.random.seed(13)
        env = amd.RLDaisyWorld(grid_dimension=8)
        env.batch_size = B
        env.reset()
        t0 = time.perf_counter()
        d, a = simulate_lifespan(env, agent, use_device_loop=dev)
        dt = time.perf_counter() - t0
        steps = env.step_count
        print(f"{policy:12s} device_loop={dev!s:5s} B={B} steps={steps} wall={dt:.3f}s  "
              f"{B * 64 * steps / dt / 1e6:.2f} Mcell-updates/s  {B * 4 * steps / dt / 1e3:.0f} k agent-steps/s  "
              f"biosphere {d.mean():.3f} agents {a.mean():.3f}")
        env.close()
