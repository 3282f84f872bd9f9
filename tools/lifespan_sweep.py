#!/usr/bin/env python3
"""The README's lifespan sweep (ref notebooks/greedy_longevity_abatement.ipynb cell 2; README.md:59-80;
BASELINE configs[3]) as an ensemble shard over the GPUs of a node.

Every rank runs `--worlds` worlds per policy on its own GPU through the drop-in environment and the
lifespan harness (`therldaisyworld_amd.harness.simulate_lifespan`), with the reference's seeding
convention shifted per rank (seed + rank); there is no communication while the episodes run.  At the end
the per-world lifespans are gathered once (RCCL all-gather of a few kB) and rank 0 prints the table.

    python tools/lifespan_sweep.py --worlds 1000 --dim 8                       # one GPU
    python tools/lifespan_sweep.py --gpus 8 --worlds 1000 --dim 256            # C4: 8 x 1000 worlds of 256x256
                                                                               # (starts and supervises its 8 ranks:
                                                                               # ensemble.launch_ranks; torchrun works too)

The step loop is device-resident in chunks (`dw_run_episode`): one launch per chunk for dim*dim <= 4096,
back-to-back launches without host round trips for larger worlds.  Measured on one MI355X, 1000 worlds,
4 agents, run to the death of every biosphere: dim 8: 0.015-0.1 s per policy; dim 256 (the C4 shard):
0.074-0.083 s per policy exact (0.068-0.078 with --lifespans-only), 0.056-0.065 s with --precision fast (~468 steps, step pairs in one fused
launch with the agents' step patched in) plus 1.0 s for the reference-compatible host-RNG reset
(`--init philox` draws the initial state on the device instead: 3 ms).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def selftest(args, rank, world):
    """Every rank fabricates the lifespans of its block of worlds from their GLOBAL ids; the gathered table
    must list them in rank order, exactly once (tests/test_bench_spawn.py)."""
    from therldaisyworld_amd import ensemble
    dist = ensemble.init_process_group("gloo") if world > 1 else None
    B, N = args.worlds, args.agents
    ids = rank * B + np.arange(B)
    done_at = (400 + ids % 7).astype(int)
    agents_done_at = (100 + (ids[:, None] + np.arange(N)[None]) % 5).astype(int)[..., None]
    wall = 0.01 * (rank + 1)
    if dist is not None:
        done_at = ensemble.gather_per_world(done_at)
        agents_done_at = ensemble.gather_per_world(agents_done_at)
        wall = ensemble.max_over_ranks(wall)
    if rank == 0:
        expect = 400 + np.arange(world * B) % 7
        print(json.dumps({"sweep": "selftest", "n_gpus": world, "worlds": int(done_at.shape[0]),
                          "in_rank_order": bool(np.array_equal(done_at, expect)),
                          "biosphere_lifespan_mean": float(done_at.mean()),
                          "agent_lifespan_mean": float(agents_done_at.mean()), "wall_s": wall}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, default=1000, help="worlds per rank and policy")
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--seed", type=int, default=13)
    ap.add_argument("--precision", default="exact", choices=["exact", "fast"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--init", default="numpy", choices=["numpy", "philox"],
                    help="numpy: the reference's legacy-RNG reset on the host (same worlds as the reference for "
                         "the same seed; 2 x B x 2 x dim^2 uniforms, ~1 s for 1000 worlds of 256x256); philox: "
                         "device-side initial state with the same distribution (reset_synthetic)")
    ap.add_argument("--albedos", default="default", choices=["default", "neutral"],
                    help="neutral: albedo_light = albedo_dark = 0.5 (the notebook's control case)")
    ap.add_argument("--lifespans-only", action="store_true",
                    help="simulate_lifespan(final_state=False): no snapshots, the last chunk is not replayed - the same "
                         "lifespans, the environments are discarded anyway")
    ap.add_argument("--gpus", type=int, default=0,
                    help="start this many ranks (one per GPU) and supervise them; 0: run as the single process / the "
                         "torchrun rank this process already is")
    ap.add_argument("--rank-timeout-s", type=float, default=900.0)
    ap.add_argument("--selftest", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (launch, rendezvous, per-world gather in rank order, "
                         "max-over-ranks, rank-0 table) with synthetic lifespans: no GPU is touched")
    args = ap.parse_args()

    from therldaisyworld_amd import ensemble
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(ensemble.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus,
                                               rank_timeout_s=args.rank_timeout_s))
    rank, local_rank, world = ensemble.rank_info()
    if args.gpus > 1 and args.gpus != world:
        raise SystemExit(f"lifespan_sweep.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest:
        return selftest(args, rank, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("lifespan_sweep.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("DW_BENCH_ALL_RANKS_ON_DEVICE0"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = ensemble.init_process_group(args.backend, device=local_rank) if world > 1 else None

    import therldaisyworld_amd as amd
    from therldaisyworld_amd.harness import simulate_lifespan

    # (torch's import leaves ~10^6 collector-tracked objects: a full collection is 40-50 ms and would land inside one
    # policy's timed episode - whichever reaches the allocation count; see bench.py)
    import gc
    gc.collect()
    gc.freeze()

    policies = [("greedy", lambda: amd.Greedy(epsilon=0.0)),
                ("antigreedy", lambda: amd.Greedy(epsilon=0.0, greedy=False)),
                ("random", lambda: amd.Greedy(epsilon=1.0)),
                ("half_random", lambda: amd.Greedy(epsilon=0.5)),
                ("no_agents_act", lambda: None)]
    rows = []
    for name, make in policies:
        np.random.seed(args.seed + rank)
        env = amd.RLDaisyWorld(grid_dimension=args.dim, n_agents=args.agents, device=local_rank,
                               precision=args.precision, world_offset=rank * args.worlds)
        env.batch_size = args.worlds
        if args.albedos == "neutral":
            env.albedo_light = env.albedo_dark = 0.5
        t0 = time.perf_counter()
        if args.init == "philox":
            env.reset_synthetic(args.seed + rank)
            obs = env.get_obs()
        else:
            obs = env.reset()
        t_reset = time.perf_counter() - t0
        done_at, agents_done_at = simulate_lifespan(env, make(), obs=obs, final_state=not args.lifespans_only)
        wall = time.perf_counter() - t0
        steps = env.step_count
        env.close()
        if dist is not None:
            done_at = ensemble.gather_per_world(done_at)
            agents_done_at = ensemble.gather_per_world(agents_done_at)
            wall = ensemble.max_over_ranks(wall)
        rows.append({"policy": name, "worlds": int(done_at.shape[0]), "steps_run": int(steps),
                     "biosphere_lifespan_mean": float(done_at.mean()), "biosphere_lifespan_std": float(done_at.std()),
                     "agent_lifespan_mean": float(agents_done_at.mean()),
                     "agent_lifespan_std": float(agents_done_at.std()), "wall_s": wall, "reset_s": t_reset,
                     "cell_updates_per_s_episode": done_at.shape[0] * args.dim * args.dim * steps
                                                   / max(wall - t_reset, 1e-9)})
        if rank == 0:
            # mean +/- standard error, as README.md:59-80 reports them
            se_b = done_at.std() / np.sqrt(done_at.size)
            se_a = agents_done_at.std() / np.sqrt(agents_done_at.size)
            print(f"{name:14s} worlds={done_at.shape[0]:6d} biosphere {done_at.mean():8.3f} +/- {se_b:5.3f}"
                  f"   agents {agents_done_at.mean():8.3f} +/- {se_a:5.3f}   {steps} steps in "
                  f"{wall:.3f} s (reset {t_reset:.3f} s)", flush=True)
    if rank == 0:
        print(json.dumps({"sweep": "README lifespan sweep", "dim": args.dim, "agents": args.agents,
                          "n_gpus": world, "precision": args.precision, "albedos": args.albedos, "rows": rows}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
