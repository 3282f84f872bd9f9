#!/usr/bin/env python3
"""bench.py — cell-updates/s of the fused RLDaisyWorld step on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5|c1|target] [--precision fast|exact]

A "step" is one pass of the hot path (ref RLDaisyWorld.step, daisy_world_rl.py:475-497) over the
whole batch of synthetic worlds: update_agents (if the workload has agents) + the fused
stencil/reaction kernel + per-world reductions.  State is resident in HBM before the timed region
(device-side Philox initial state).  One process per GPU; worlds are independent, so
ranks share nothing in the data path (weak scaling: every rank steps its own `worlds` worlds) and
RCCL is used only for the barrier / max-over-ranks timing and a final gather of per-world statistics.

Arithmetic modes: `fast` (default here) is float32 arithmetic — every cell within one quantum (1e-3) of
the float64 reference per step, >= 99.5 % identical (tests/test_gpu_parity.py) — and on wide grids
without agents dw_step_n runs TWO steps per launch (temporal blocking: step-1 rows live only in
registers) and keeps the states between its launches as binary16 planes (lossless for the quantised
state), so its algorithmic GB/s can exceed the HBM peak; `exact` (the drop-in class's default) is
float32 plus a float64 re-evaluation of every near-tie cell and is bit-identical to the float64 reference.
Both are measured; the one not chosen by --precision is reported under "modes".

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     algorithmic 16 B per cell-update (float32 light+dark read once, written once) x the
               cells of one launch / the step kernel's average launch duration measured here with HIP
               events on the kernel's own stream, against the 8 TB/s HBM3E peak.
  cpu_baseline the oracle's C restatement (oracle/daisy_oracle.c, "port") timed on this host's cores
               on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_CELL_UPDATE = 16       # 2 x float32 read + 2 x float32 written

WORKLOADS = {
    # name: (worlds per GPU, grid, agents per world, description)
    "c1": (1, 64, 0, "BASELINE configs[0]: 1 world, 64x64, no agent (the reference's CPU-runnable case)"),
    "c2": (1024, 256, 0, "BASELINE configs[1]: 1024 worlds, 256x256, no agent, ramped luminosity"),
    "c3": (256, 1024, 1, "BASELINE configs[2]: 256 worlds, 1024x1024, 1 greedy agent per world"),
    "c5": (8, 8192, 16, "BASELINE configs[4] per-GPU shard: 8 worlds, 8192x8192, 16 mixed-policy agents"),
    "target": (1024, 4096, 0, "north-star target: 1024 worlds of 4096x4096, no agent (256 GiB of float32 ping-pong "
                              "state in the 288 GB of one MI355X; --worlds N for a smaller ensemble)"),
    "c4": (1000, 8, 4, "BASELINE configs[3] per-GPU shard at the README's grid: 1000 worlds, 8x8, 4 greedy agents, "
                       "device-resident episode loop (dw_run_episode)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512,
                    help="timed steps (default 512: the full luminosity ramp BASELINE configs[1] is defined on)")
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--worlds", type=int, default=0, help="override worlds per GPU")
    ap.add_argument("--precision", default="fast", choices=["exact", "fast", "f64"],
                    help="arithmetic mode of the headline number (default fast = float32, the tolerance the "
                         "north star states; the other of exact/fast is measured too and reported under 'modes')")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--no-modes", action="store_true", help="skip the extra run in the other arithmetic mode")
    return ap.parse_args()


def cpu_baseline(grid: int, params_obj, budget_s: float = 15.0):
    """Time the oracle's C restatement on this host: all cores (OpenMP over worlds) on a
    bounded sample of the workload's grid, one thread on one world of it, and BASELINE configs[0] (C1:
    1 world, 64x64, 500 steps) exactly, single thread.  Three repeats each, median."""
    import statistics
    from oracle import c_oracle
    c_oracle.build()
    cores = max(1, min(c_oracle.max_threads(), len(os.sched_getaffinity(0))))
    g = min(grid, 256)                               # bounded sample of the workload's grid
    worlds = 2 * cores
    rng = np.random.RandomState(0)

    def fresh(n, dim):
        light = (rng.rand(n, dim, dim) < 0.33) * 0.2 * rng.rand(n, dim, dim)
        dark = (rng.rand(n, dim, dim) < 0.33) * 0.2 * rng.rand(n, dim, dim)
        return np.ascontiguousarray(light), np.ascontiguousarray(dark)

    def timed(n, dim, steps, repeats=3):
        rates = []
        for _ in range(repeats):
            light, dark = fresh(n, dim)
            t0 = time.perf_counter()
            c_oracle.step_n(light, dark, 0.75, 0.75 / 512, steps)
            rates.append(n * dim * dim * steps / (time.perf_counter() - t0))
        return statistics.median(rates)

    c_oracle.set_threads(cores)
    light, dark = fresh(worlds, g)
    c_oracle.step_n(light, dark, 0.75, 0.75 / 512, 1)                      # warm-up (thread pool, pages)
    t0 = time.perf_counter()
    c_oracle.step_n(light, dark, 0.75, 0.75 / 512, 2)
    per_step = (time.perf_counter() - t0) / 2
    steps = int(max(4, min(200, budget_s * 0.8 / 3 / max(per_step, 1e-6))))
    rate_all = timed(worlds, g, steps)
    c_oracle.set_threads(1)
    s1 = max(2, steps // 2)
    rate_1 = timed(1, g, s1)
    rate_c1 = timed(1, 64, 500)                                             # BASELINE configs[0], exactly
    c_oracle.set_threads(cores)
    return {
        "value": rate_all, "unit": "cell-updates/s", "cores": cores, "kind": "port",
        "sample": f"oracle/daisy_oracle.c (float64, reference staging), median of 3: {worlds} worlds x {g}x{g} x "
                  f"{steps} steps, OpenMP over worlds on {cores} threads; single thread, 1 world x {g}x{g} x {s1} steps: "
                  f"{rate_1:.3e}; C1 (1 world x 64x64 x 500 steps) single thread: {rate_c1:.3e} cell-updates/s "
                  f"(the Python reference itself: 1.5-2.2e6 on C1, SURVEY.md 6)",
        "single_thread_value": rate_1, "c1_single_thread_value": rate_c1,
    }


def load_traffic(workload: str, precision: str, steps_per_launch: int = 0):
    """HBM bytes per launch of the dominant kernel from the PMC profile of the same command, if one has
    been committed under profiles/ (collected per the guide: separate --pmc passes, FETCH_SIZE x2 on
    gfx950).  `steps_per_launch` (when given) must match the profile's."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("precision") == precision:
            if steps_per_launch and d.get("steps_per_launch", 1) != steps_per_launch:
                part = d.get("single" if steps_per_launch == 1 else "fused")
                if not part:
                    continue
                d = dict(d, hbm_bytes_per_launch=part["hbm_bytes_per_launch"])
            best = d
    return None if best is None else best.get("hbm_bytes_per_launch")


def main():
    args = parse()
    from therldaisyworld_amd import ensemble
    rank, local_rank, world = ensemble.rank_info()
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    n_gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("DW_BENCH_ALL_RANKS_ON_DEVICE0"):     # rehearsal of the N > 1 path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = ensemble.init_process_group(args.backend) if world > 1 else None

    import therldaisyworld_amd as amd
    from therldaisyworld_amd import _ffi

    B, G, N, desc = WORKLOADS[args.workload]
    if args.worlds:
        B = args.worlds
    min_L, max_L, dL = 0.75, 1.5, 0.75 / 512
    cells = B * G * G

    def measure(precision, steps, warmup):
        """One timed run of `steps` steps in the given arithmetic mode; returns the numbers of the JSON."""
        p = amd.default_params(B, G, G, N)
        p.device = local_rank
        p.precision = _ffi.PRECISION[precision]
        p.world_offset = rank * B                       # global world ids: the ensemble is one sweep
        eng = amd.Engine(p)
        eng.init_random(args.seed)
        # per-agent policy for the agent workloads: greedy (c3) or greedy/antigreedy/random/half-random by
        # agent index (c5).  Random actions are drawn on the host and uploaded.
        rng = np.random.RandomState(args.seed + rank)

        def run(nsteps, L):
            if N == 0:
                return eng.step_n(nsteps, L, dL, min_L, max_L)
            if args.workload == "c4":                       # small worlds: whole chunks of steps in one launch
                if L == min_L:                              # first step from the un-quantised state
                    eng.policy_greedy(argmin=False)
                    eng.step_device_actions(L)
                    L = min(max(L + dL, min_L), max_L)
                    nsteps -= 1
                while nsteps > 0:
                    k = min(nsteps, 64)
                    Ls = []
                    for _ in range(k):
                        Ls.append(L)
                        L = min(max(L + dL, min_L), max_L)
                    eng.run_episode(Ls, _ffi.POLICY_ARGMAX)
                    nsteps -= k
                return L
            # agent workloads on wide grids (c3, c5): the episode loop stays on the device in chunks
            # (dw_run_episode without per-step world flags: step pairs in one fused launch, the agents'
            # in-between step patched in).  c3: every agent greedy.  c5: agents 0-3 greedy, 4-7 antigreedy,
            # 8-11 random, 12-15 half-random (one coin per step for the batch, as Greedy does); random
            # actions are drawn on the host into the int8 table, -1 / -2 stand for the greedy / anti-greedy
            # choice evaluated on the device; nothing is downloaded but the (K,B,N) agent flags.
            if L == min_L:                                  # first step from the un-quantised state
                eng.policy_greedy(argmin=False)
                eng.step_device_actions(L)
                L = min(max(L + dL, min_L), max_L)
                nsteps -= 1
            while nsteps > 0:
                k = min(nsteps, 64)
                Ls = []
                for _ in range(k):
                    Ls.append(L)
                    L = min(max(L + dL, min_L), max_L)
                if args.workload == "c5":
                    table = np.empty((k, B, 16), dtype=np.int8)
                    table[:, :, 0:4] = -1
                    table[:, :, 4:8] = -2
                    table[:, :, 8:12] = rng.randint(9, size=(k, B, 4))
                    coin = rng.rand(k) > 0.5
                    table[:, :, 12:16] = np.where(coin[:, None, None], -1, rng.randint(9, size=(k, B, 4)))
                    eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table, world_flags=False)
                else:
                    eng.run_episode(Ls, _ffi.POLICY_ARGMAX, world_flags=False)
                nsteps -= k
            return L

        L = run(warmup, min_L)
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        eng.timer_start()
        L = run(steps, L)
        ev_ms = eng.timer_stop()                        # HIP events on the kernel's stream (synchronises)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = ensemble.max_over_ranks(time.perf_counter() - t0)
        ev_ms = ensemble.max_over_ranks(ev_ms)
        kernel_ms = ev_ms / steps
        achieved = BYTES_PER_CELL_UPDATE * cells / (kernel_ms * 1e-3) / 1e9
        stats = eng.reduce()
        info = eng.kernel_info()
        # dw_step_n runs wide agent-free grids as fused step PAIRS: the dominant kernel's launch = 2 steps
        # (agent workloads c3 / c5 too: dw_run_episode pairs the steps and patches the agents' step in)
        paired = N == 0 or (args.workload in ("c3", "c5") and not os.environ.get("DW_NO_AGENT_FUSE"))
        spl = 2 if (paired and "fuses step pairs" in info and not os.environ.get("DW_NO_FUSE")) else 1
        res = {"value": cells * steps * n_gpus / elapsed, "ms_per_step": elapsed / steps * 1e3,
               "kernel_ms": kernel_ms, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
               "fixups": eng.last_fixup_count(), "kernel": info, "stats": stats, "steps_per_launch": spl}
        eng.close()
        return res

    m = measure(args.precision, args.steps, args.warmup)
    value, kernel_ms, achieved, fixups, info, stats = (m["value"], m["kernel_ms"], m["achieved"], m["fixups"],
                                                       m["kernel"], m["stats"])
    elapsed_ms_per_step = m["ms_per_step"]
    all_stats = ensemble.gather_per_world(stats) if dist is not None else stats   # end-of-run gather (RCCL)

    out = {
        "metric": "cell-updates/sec (grid x batch), fused stencil+growth step",
        "value": value,
        "unit": "cell-updates/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"exact": "f32 (+ f64 re-evaluation of near-tie cells: bit-identical to f64)", "fast": "f32",
                  "f64": "f64"}[args.precision],
        "data": "synthetic (device Philox initial state with the distribution of initialize_grid; ramped luminosity)",
        "config": {"workload": f"{args.workload}: {desc}", "worlds_per_gpu": B, "grid": [G, G], "agents_per_world": N,
                   "precision": args.precision, "kernel": info, "total_worlds": int(all_stats.shape[0]),
                   "parallelism": f"ensemble shard x{n_gpus} (no data-path collective)"},
        # per launch of the dominant kernel: algorithmic bytes = 16 B x cell-updates of one launch; achieved =
        # that / the launch duration (HIP events on the kernel's stream / launches); traffic = PMC HBM bytes
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": load_traffic(args.workload, args.precision, m["steps_per_launch"]),
                     "bytes_per_cell_update": BYTES_PER_CELL_UPDATE, "steps_per_launch": m["steps_per_launch"],
                     "cell_updates_per_launch": cells * m["steps_per_launch"],
                     "algorithmic_bytes_per_launch": BYTES_PER_CELL_UPDATE * cells * m["steps_per_launch"],
                     "launch_ms": kernel_ms * m["steps_per_launch"], "kernel_ms_per_step": kernel_ms,
                     "f64_fixups_last_step": fixups,
                     "note": ("two steps share one HBM round trip (temporal blocking in registers) and the states "
                              "between the launches of a run are binary16 planes (lossless: integers <= 1000): measured "
                              "traffic is about a quarter of the algorithmic bytes, so frac may exceed 1; the fused "
                              "kernels are VALU-issue-bound (83-85 % busy, profiles/r01k_valu_pmc.json)")
                     if m["steps_per_launch"] == 2 else "single-step kernel: HBM-bound"},
    }
    if not args.no_modes:
        # the other arithmetic mode on the same workload, for the record (shorter run).  In "fast" mode
        # dw_step_n fuses pairs of steps into one launch on wide grids without agents (temporal blocking),
        # so its algorithmic GB/s may exceed the HBM peak; measured HBM bytes are in profiles/.
        other = "fast" if args.precision == "exact" else "exact"
        o = measure(other, max(10, args.steps // 2), max(4, args.warmup // 2))
        out["modes"] = {other: {"value": o["value"], "ms_per_step": o["ms_per_step"], "kernel_ms": o["kernel_ms"],
                                "achieved_GBps": o["achieved"], "frac": o["frac"], "kernel": o["kernel"],
                                "steps_per_launch": o["steps_per_launch"],
                                "traffic": load_traffic(args.workload, other, o["steps_per_launch"])}}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(G, None)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
