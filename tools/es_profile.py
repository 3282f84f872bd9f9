#!/usr/bin/env python3
"""cProfile of one generation of tools/es_fitness_bench.py's workload (64 x 32 worlds of 16x16, 4 agents): where the host
time of get_fitness_population goes.  usage: es_profile.py [P] [wpm] [dim] [chunk]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd.harness import get_fitness_population  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wpm = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 16
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 64
np.random.seed(7)
pop = [amd.MLP() for _ in range(P)]
np.random.seed(11)
env = amd.RLDaisyWorld(grid_dimension=dim, n_agents=4)
get_fitness_population(env, pop, worlds_per_member=wpm, max_steps=768, chunk=chunk)      # warm: handle, kernels
pr = cProfile.Profile()
pr.enable()
get_fitness_population(env, pop, worlds_per_member=wpm, max_steps=768, chunk=chunk)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
env.close()
