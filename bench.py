#!/usr/bin/env python3
"""bench.py — cell-updates/s of the fused RLDaisyWorld step on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|c2|c3|c4|c5|c1] [--precision exact|fast]

A "step" is one pass of the hot path (ref RLDaisyWorld.step, daisy_world_rl.py:475-497) over the whole
batch of synthetic worlds: update_agents (if the workload has agents) + the fused stencil/reaction
kernel + per-world reductions.  State is resident in HBM before the timed region (device-side Philox
initial state).  One process per GPU; worlds are independent, so ranks share nothing in the
data path (weak scaling: every rank steps its own `worlds` worlds) and RCCL is used only for the barrier /
max-over-ranks timing and a final gather of per-world statistics.  `--gpus N` without a torchrun
environment launches the N ranks itself (children only: the parent never touches the GPU).

Default workload: the north-star shape, 1024 worlds of 4096x4096 on ONE GPU (128 GiB of binary16 ping-pong
state; if the allocation fails the world count is halved until it fits and the line says so).  The other
BASELINE configs - c2 (1024 x 256^2 over the whole ramp), c3 (256 x 1024^2, a greedy agent), c4 (the 1000 x 256^2
shard with 4 greedy agents and per-step biosphere flags; also at the README's 8x8 grid) and c5 (the 8 x 8192^2
shard, 16 mixed-policy agents) - are measured in the same invocation, both modes, and reported under "workloads".  Arithmetic modes: the headline is `fast` - float32 arithmetic, the "stated fp32 tolerance" of the
north star: from identical states every cell within one quantum (1e-3) and >= 99.98 % of the cell values
identical per step (measured >= 99.994 % on developed states over the whole luminosity ramp,
profiles/r02_fast_tolerance.json; asserted by tests/test_gpu_parity.py); `exact` - float32 plus a float64
re-evaluation of every near-tie cell, bit-identical to the float64 reference and the default of the drop-in
class - is measured in the same invocation and reported under "modes".

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline     per launch of the dominant kernel (the fused step-pair kernel: one launch = 2 steps):
               `achieved` = ALGORITHMIC bytes of a launch / its measured duration, where algorithmic bytes
               follow SURVEY 8(d): 4 x sizeof(plane element) per cell-update in the storage format of that
               launch (binary16 planes: 8 B) - never the 16 B of float32 storage; `launch_ms` comes from
               HIP events recorded on the kernel's own stream around the run of fused launches inside the
               timed region (dw_last_step_n_timing).  `traffic` = HBM bytes per launch from the PMC profile
               committed under profiles/ (collected separately, as the guide prescribes), `measured_frac` =
               traffic / launch time / peak; `valu` = VALU instructions per cell-evaluation and SIMD busy
               fraction from the committed SQ counters; `bound` is what those counters say.  `copy_ceiling` =
               what a plain device copy reaches on this card, measured live (the practical ceiling
               beside the spec peak), `measured_over_copy_ceiling` = the kernel's measured HBM rate / that.
  power        {"board_w", "limit_w", "sclk_mhz", "samples"}: the board's power and engine clock read with rocm-smi beside an
               UNTIMED extra pass of the same launches after the timed region (N = 1, data-parallel workloads; null when
               rocm-smi cannot be read).  At the north-star shape the step-pair kernels hold the card at its package power
               limit and the engine clock is what the power manager leaves (profiles/r04_power_trace.txt).
  cpu_baseline the oracle's C restatement (oracle/daisy_oracle.c, "port") timed on this host's cores
               on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (worlds per GPU, grid, agents per world, description)
    "target": (1024, 4096, 0, "north-star target: 1024 worlds of 4096x4096, no agent, ramped luminosity (128 GiB of "
                              "binary16 ping-pong state on one MI355X)"),
    "c1": (1, 64, 0, "BASELINE configs[0]: 1 world, 64x64, no agent (the reference's CPU-runnable case)"),
    "c2": (1024, 256, 0, "BASELINE configs[1]: 1024 worlds, 256x256, no agent, ramped luminosity"),
    "c3": (256, 1024, 1, "BASELINE configs[2]: 256 worlds, 1024x1024, 1 greedy agent per world"),
    "c5": (8, 8192, 16, "BASELINE configs[4] per-GPU shard: 8 worlds, 8192x8192, 16 mixed-policy agents"),
    "c4": (1000, 256, 4, "BASELINE configs[3] per-GPU shard: 1000 worlds, 256x256, 4 greedy agents, device-resident "
                         "episode loop with per-step biosphere flags (dw_run_episode, as the lifespan sweep runs it)"),
    "c4_dim8": (1000, 8, 4, "BASELINE configs[3] per-GPU shard at the README's own grid: 1000 worlds, 8x8, 4 greedy agents, "
                            "LDS-resident episode kernel (dw_run_episode)"),
}
# what the default invocation measures beside the headline workload: (timed steps, warm-up steps, pre-heat seconds)
# (c4_dim8 is a latency-bound launch of 1000 waves: half a second of pre-heat, or the timed region measures the clock ramp)
EXTRA_WORKLOADS = {"c2": (512, 64, 0.5), "c3": (64, 8, 0.3), "c4": (64, 8, 0.3), "c4_dim8": (512, 64, 0.5), "c5": (32, 8, 0.3)}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed steps")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--worlds", type=int, default=0, help="override worlds per GPU")
    ap.add_argument("--precision", default="fast", choices=["exact", "fast", "f64"],
                    help="arithmetic mode of the headline number (default fast = float32 within the stated, tested "
                         "tolerance; the other of exact/fast is measured too and reported under 'modes')")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--preheat-s", type=float, default=2.0,
                    help="seconds of untimed stepping before the warm-up steps, so that a short timed region does "
                         "not measure a GPU whose clocks are still rising (reported as preheat_s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--rank-timeout-s", type=float, default=300.0,
                    help="self-launched N > 1 runs: kill every rank and exit 124 when the run takes longer")
    ap.add_argument("--rank-log-dir", default="", help="self-launched N > 1 runs: keep the per-rank logs here")
    ap.add_argument("--no-modes", action="store_true", help="skip the extra run in the other arithmetic mode")
    ap.add_argument("--no-power", action="store_true", help="skip the untimed extra pass that reads board power / clock")
    ap.add_argument("--no-workloads", action="store_true", help="skip the extra measurements of c2 ... c5")
    ap.add_argument("--selftest-spawn", action="store_true",
                    help="CPU rehearsal of the N > 1 plumbing (rank launch, rendezvous, barrier, max-over-ranks, "
                         "gather, rank-0 relay) without touching a GPU: prints the JSON skeleton")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N > 1 without torchrun: launch the ranks ourselves
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    """Parent of a self-launched multi-rank run: `--gpus N` fresh child processes of this script, one per GPU,
    supervised by therldaisyworld_amd.ensemble.launch_ranks (torchrun environment; every child polled; the first
    failing rank terminates the others; `--rank-timeout-s` bounds the whole run; rank logs relayed on failure).
    The parent imports neither torch nor the HIP library: a process that has initialised the GPU never
    replaces itself or forks GPU work."""
    from therldaisyworld_amd.ensemble import launch_ranks
    return launch_ranks(os.path.abspath(__file__), argv, args.gpus, rank_timeout_s=args.rank_timeout_s,
                        log_dir=args.rank_log_dir or None)


def cpu_baseline(grid: int, budget_s: float = 15.0):
    """Time the oracle's C restatement on this host: all cores (OpenMP over worlds) on a
    bounded sample of the workload's grid, one thread on one world of it, and BASELINE configs[0] (C1:
    1 world, 64x64, 500 steps) exactly, single thread.  Three repeats each, median."""
    import statistics
    import numpy as np
    from oracle import c_oracle
    c_oracle.build()
    cores = max(1, min(c_oracle.max_threads(), c_oracle.usable_cpus()))   # affinity mask capped by the cgroup quota
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


def load_profile(kind: str, workload: str, precision: str):
    """Newest committed PMC summary for (workload, precision): kind 'traffic' -> profiles/traffic_*.json (HBM bytes
    per fused launch: separate --pmc passes, FETCH_SIZE x2 on gfx950), kind 'valu' -> profiles/*_valu_pmc.json
    (SQ counters of the fused kernel).  Returns (dict, file name, meta) or (None, None, None); meta = the build id
    and shape the profile was taken on (absent in the profiles of rounds 1-2)."""
    pat = "traffic_*.json" if kind == "traffic" else "*_valu_pmc.json"
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), key=lambda q: (json_round(q), q)):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        meta = {"library_build_id": d.get("library_build_id"), "worlds_per_gpu": d.get("worlds_per_gpu"), "grid": d.get("grid")}
        if kind == "traffic":
            if d.get("workload") == workload and d.get("precision") == precision and d.get("plane_elem_bytes", 4) == 2:
                best = (d.get("fused") or d, os.path.basename(path), meta)
        else:
            if str(d.get("workload", "")).split()[0] == workload and d.get("plane_elem_bytes", 4) == 2 and precision in d:
                best = (d[precision], os.path.basename(path), meta)
    return best if best else (None, None, None)


def json_round(path):
    try:
        return int(json.load(open(path)).get("round", 0))
    except Exception:
        return 0


def selftest_spawn(args):
    """The N > 1 plumbing on CPU (gloo): what tests/test_bench_spawn.py runs."""
    import numpy as np
    from therldaisyworld_amd import ensemble
    rank, _, world = ensemble.rank_info()
    # failure rehearsals (tests/test_bench_spawn.py): a rank that dies before the rendezvous / never finishes
    if os.environ.get("DW_SELFTEST_FAIL_RANK") == str(rank):
        sys.stderr.write(f"selftest: rank {rank} exits 3 before the rendezvous\n")
        raise SystemExit(3)
    if os.environ.get("DW_SELFTEST_HANG_RANK") == str(rank):
        time.sleep(3600)
    dist = ensemble.init_process_group("gloo", timeout_s=float(os.environ.get("DW_SELFTEST_DIST_TIMEOUT_S", "120"))) if world > 1 else None
    B = args.worlds or 3
    # rehearsal of the DW_ENOMEM path of make_engine: one rank could only allocate half the worlds - every rank must
    # end up with that rank's count (ensemble.agree_on_worlds, the call measure() makes)
    B_here = max(1, B // 2) if os.environ.get("DW_SELFTEST_HALVE_RANK") == str(rank) else B
    B = ensemble.agree_on_worlds(B_here) if dist is not None else B_here
    local = np.arange(B, dtype=np.int64) + rank * B
    if dist is not None:
        dist.barrier()
    elapsed = ensemble.max_over_ranks(0.001 * (rank + 1))
    allw = ensemble.gather_per_world(local) if dist is not None else local
    group = ensemble.describe_group()
    per_rank = ensemble.gather_scalars(float(rank + 1))
    if rank == 0:
        print(json.dumps({"selftest": "spawn", "n_gpus": world, "total_worlds": int(allw.shape[0]),
                          "worlds": [int(x) for x in allw], "max_elapsed": elapsed, "worlds_per_rank": B,
                          "rccl": group, "per_rank_value": per_rank}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    if args.selftest_spawn:
        return selftest_spawn(args)

    import numpy as np
    from therldaisyworld_amd import ensemble
    rank, local_rank, world = ensemble.rank_info()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = world

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if os.environ.get("DW_BENCH_ALL_RANKS_ON_DEVICE0"):     # rehearsal of the N > 1 path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # (DW_DIST_FORCE_COLLECTIVES=1: a one-rank group anyway - every collective of the run on RCCL, on a 1-GPU box)
    force_group = os.environ.get("DW_DIST_FORCE_COLLECTIVES", "0") == "1"
    dist = ensemble.init_process_group(args.backend, device=local_rank) if (world > 1 or force_group) else None

    import therldaisyworld_amd as amd
    from therldaisyworld_amd import _ffi, telemetry

    # torch's import leaves ~10^6 objects tracked by the cyclic collector: a full collection costs 40-50 ms and is triggered by
    # allocation COUNTS, i.e. it lands in whichever timed region reaches the count (seen as +41 ms in ONE policy of
    # tools/lifespan_sweep.py, a different one when the run was instrumented).  Collect now, then take everything alive out
    # of the collector's reach: later collections only look at what the measurement itself allocates.
    import gc
    gc.collect()
    gc.freeze()

    min_L, max_L, dL = 0.75, 1.5, 0.75 / 512

    def make_engine(B, G, N, precision):
        """Engine for B worlds; the world count is halved while the device cannot hold the state."""
        while True:
            p = amd.default_params(B, G, G, N)
            p.device = local_rank
            p.precision = _ffi.PRECISION[precision]
            p.world_offset = rank * B                       # global world ids: the ensemble is one sweep
            eng = None
            try:
                eng = amd.Engine(p)
                try:
                    eng.init_random(args.seed)              # un-quantised like the reference's initialize_grid
                except amd.DaisyHipError as e:
                    if e.code != _ffi.DW_ENOMEM:
                        raise
                    eng.init_random(args.seed, quantised=True)   # no room for the float32 staging: rounded draw
                return eng, B
            except amd.DaisyHipError as e:
                if eng is not None:
                    eng.close()
                if e.code != _ffi.DW_ENOMEM or B == 1:
                    raise
                B //= 2

    def measure(workload, precision, steps, warmup, preheat_s, worlds=0, power=False):
        """One timed run of `steps` steps of `workload` in the given arithmetic mode."""
        B, G, N, desc = WORKLOADS[workload]
        if worlds:
            B = worlds
        eng, B = make_engine(B, G, N, precision)
        if dist is not None:
            # every rank must step the same number of worlds: the smallest any rank could allocate.  This MIN is also
            # the first collective after the allocation / initial draw (up to 128 GiB + 0.04 s per rank, more when a
            # rank had to halve): behind it the 120 s collective timeout no longer spans that work.
            eng.sync()
            B_all = ensemble.agree_on_worlds(B)
            if B_all != B:
                eng.close()
                eng, B = make_engine(B_all, G, N, precision)
                if ensemble.agree_on_worlds(B) != B:
                    raise SystemExit(f"rank {rank}: could not allocate the {B_all} worlds the ranks agreed on")
        cells = B * G * G
        rng = np.random.RandomState(args.seed + rank)

        def run(nsteps, L, ramp=dL):
            if N == 0:
                return eng.step_n(nsteps, L, ramp, min_L, max_L)
            first = not getattr(run, "started", False)
            run.started = True
            if first:                                       # first step from the un-quantised state
                eng.policy_greedy(argmin=False)
                eng.step_device_actions(L)
                L = min(max(L + ramp, min_L), max_L)
                nsteps -= 1
            # agent workloads: the episode loop stays on the device in chunks (dw_run_episode: one launch per
            # chunk for the 8x8 worlds of c4; step pairs in one fused launch with the agents' in-between step
            # patched in on the wide grids of c3 / c5).  c3 / c4: every agent greedy.  c5: agents 0-3 greedy,
            # 4-7 antigreedy, 8-11 random, 12-15 half-random (one coin per step for the batch, as Greedy does);
            # random actions are drawn on the host into the int8 table, -1 / -2 stand for the greedy /
            # anti-greedy choice evaluated on the device; nothing is downloaded but the (K,B,N) agent flags.
            while nsteps > 0:
                k = min(nsteps, 64)
                # update_L's recurrence (ref :463-473: repeated addition, clamped) for k steps in one call: the running sum
                # adds in the same order, and with ramp >= 0 a clamped value stays clamped
                acc = np.add.accumulate(np.concatenate(([L], np.full(k, ramp))))
                Ls = np.clip(acc[:k], min_L, max_L)
                L = float(min(max(acc[k], min_L), max_L))
                if workload == "c5":
                    table = np.empty((k, B, 16), dtype=np.int8)
                    table[:, :, 0:4] = -1
                    table[:, :, 4:8] = -2
                    table[:, :, 8:12] = rng.randint(9, size=(k, B, 4))
                    coin = rng.rand(k) > 0.5
                    table[:, :, 12:16] = np.where(coin[:, None, None], -1, rng.randint(9, size=(k, B, 4)))
                    eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table, world_flags=False, reuse_buffers=True)
                elif workload in ("c4", "c4_dim8"):
                    eng.run_episode(Ls, _ffi.POLICY_ARGMAX, reuse_buffers=True)     # with the per-step world flags of the sweep
                else:
                    eng.run_episode(Ls, _ffi.POLICY_ARGMAX, world_flags=False, reuse_buffers=True)
                nsteps -= k
            return L

        # pre-heat at constant luminosity (develops the daisies, does not advance the ramp), then warm-up
        L = min_L
        t_pre = time.perf_counter()
        pre_steps = 0
        chunk = 64                                          # 31 fused launches + 2 single steps per call
        while time.perf_counter() - t_pre < preheat_s:
            L = run(chunk, L, 0.0)
            eng.sync()
            pre_steps += chunk
        preheat = time.perf_counter() - t_pre
        if warmup:
            L = run(warmup, L)
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        eng.timer_start()
        L = run(steps, L)
        ev_ms = eng.timer_stop()                        # HIP events on the kernel's stream (synchronises)
        torch.cuda.synchronize()
        own_elapsed = time.perf_counter() - t0              # this rank's own K steps (before it waits for the others)
        if dist is not None:
            dist.barrier()
        elapsed = ensemble.max_over_ranks(time.perf_counter() - t0)
        fused_ms, fused_n, elem_bytes = eng.last_step_n_timing() if N == 0 else (0.0, 0, 2)
        stats = eng.reduce()
        info = eng.kernel_info()
        fixups = eng.last_fixup_count()
        pw = None
        if power and N == 0:
            # ~2 s more of the same launches at the luminosity reached, untimed (everything the line reports was read
            # above), with the board's power and engine clock read beside them
            extra = int(min(max(2.0 / max(elapsed / steps, 1e-6), 8), 20000))
            pw = telemetry.board_power_while(lambda: (run(extra + (extra & 1), L, 0.0), eng.sync()),   # (dw_step_n only enqueues)
                                              device=local_rank)
        res = {"workload": workload, "power": pw, "desc": desc, "precision": precision, "B": B, "G": G, "N": N, "cells": cells,
               "value": cells * steps * n_gpus / elapsed, "ms_per_step": elapsed / steps * 1e3,
               "rank_value": cells * steps / own_elapsed,
               "event_ms_per_step": ev_ms / steps, "fused_ms": fused_ms, "fused_launches": fused_n,
               "plane_elem_bytes": elem_bytes, "fixups": fixups, "kernel": info, "stats": stats,
               "preheat_s": preheat, "preheat_steps": pre_steps}
        eng.close()
        return res

    def copy_ceiling(nbytes=1 << 30, reps=20):
        """what a plain device copy reaches on THIS card (read nbytes + write nbytes per pass), the
        practical ceiling beside the 8 TB/s spec: the better of hipMemcpyDtoD (Tensor.copy_) and an elementwise
        kernel (torch.add, float32), timed with events on torch's stream after the engines are closed."""
        a = torch.ones(nbytes // 4, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        best = {}
        for name, op in (("memcpy_dtod", lambda: b.copy_(a)), ("elementwise_add", lambda: torch.add(a, 1.0, out=b))):
            for _ in range(3):
                op()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                op()
            e1.record()
            torch.cuda.synchronize()
            best[name] = 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        torch.cuda.empty_cache()
        return {"GB/s": max(best.values()), "frac_of_peak": max(best.values()) / HBM_PEAK_GBS, "by_method": best,
                "bytes_per_pass": 2 * nbytes}

    def roofline(m):
        """SURVEY 8(d) accounting for the dominant kernel of run `m` (see the module docstring)."""
        bpc = 4 * m["plane_elem_bytes"]                 # 2 planes read + 2 written, in the launch's storage format
        if m["fused_launches"] > 0:
            spl, launch_ms, timed = 2, m["fused_ms"] / m["fused_launches"], m["fused_launches"]
            src = "HIP events around the fused launches of the timed dw_step_n call (dw_last_step_n_timing)"
        else:                                           # single-step launches only (agent workloads, tiny runs)
            spl, launch_ms, timed = 1, m["event_ms_per_step"], 0
            src = "HIP events over the whole timed region / steps (includes the agent and policy kernels)"
        alg = bpc * m["cells"] * spl
        achieved = alg / (launch_ms * 1e-3) / 1e9
        tr, tr_file, tr_meta = load_profile("traffic", m["workload"], m["precision"])
        va, va_file, va_meta = load_profile("valu", m["workload"], m["precision"])
        live_id = _ffi.load().dw_build_id().decode()

        def provenance(meta, fname):
            """Where a committed counter profile came from, and whether it describes THIS library and shape."""
            same_build = meta.get("library_build_id") == live_id
            same_shape = meta.get("worlds_per_gpu") == m["B"] and meta.get("grid") == [m["G"], m["G"]]
            return {"file": f"profiles/{fname}", "library_build_id": meta.get("library_build_id"),
                    "worlds_per_gpu": meta.get("worlds_per_gpu"), "grid": meta.get("grid"),
                    "stale": not same_build, "same_shape": same_shape}
        # the committed profile may be of another world count: scale its bytes per cell-update to this run's launch
        traffic = tr["hbm_bytes_per_cell_update"] * m["cells"] * spl if tr and spl == 2 else None
        measured = traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None
        valu = None
        if va and spl == 2:
            valu = {"instr_per_cell_eval": va["derived"]["valu_instr_per_cell_eval"],
                    "busy_frac": va["derived"]["valu_busy_fraction"], "profile": provenance(va_meta, va_file)}
        bound = "valu" if (valu and valu["busy_frac"] >= 0.7 and (measured is None or measured < 0.6)) else "hbm"
        return {"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # two fractions, named for what they are: `frac` = ALGORITHMIC bytes / time / peak - the
                # task's accounting, a throughput figure in byte units; `hbm_frac_measured` = bytes the memory system
                # actually moved (PMC) / time / peak - how busy HBM is.  A fused launch moves about half its
                # algorithmic bytes, so the first is about twice the second; the kernel is VALU-bound (`bound`).
                "frac": achieved / HBM_PEAK_GBS, "frac_is": "algorithmic bytes / launch time / peak",
                "traffic": traffic, "measured_frac": measured, "hbm_frac_measured": measured,
                "traffic_profile": provenance(tr_meta, tr_file) if traffic else None,
                "traffic_source": f"profiles/{tr_file} (PMC passes of the same command, collected separately)" if traffic else None,
                "valu": valu, "bytes_per_cell_update": bpc, "plane_elem_bytes": m["plane_elem_bytes"],
                "steps_per_launch": spl, "cell_updates_per_launch": m["cells"] * spl,
                "algorithmic_bytes_per_launch": alg, "launch_ms": launch_ms, "launches_timed": timed,
                "launch_ms_source": src, "f64_fixups_last_step": m["fixups"]}

    def brief(m):
        r = roofline(m)
        return {"value": m["value"], "ms_per_step": m["ms_per_step"], "worlds_per_gpu": m["B"], "grid": [m["G"], m["G"]],
                "kernel": m["kernel"], "roofline": {k: r[k] for k in ("bound", "achieved", "frac", "traffic", "measured_frac",
                                                                     "bytes_per_cell_update", "launch_ms", "launches_timed",
                                                                     "steps_per_launch")}}

    m = measure(args.workload, args.precision, args.steps, args.warmup, args.preheat_s, args.worlds,
                power=rank == 0 and n_gpus == 1 and not args.no_power)
    all_stats = ensemble.gather_per_world(m["stats"]) if dist is not None else m["stats"]   # end-of-run gather (RCCL)
    group = ensemble.describe_group()                       # backend, world size, RCCL version, ranks that answered
    per_rank = ensemble.gather_scalars(m["rank_value"])     # each rank's own cell-updates/s over ITS elapsed time
    B, G, N = m["B"], m["G"], m["N"]
    out = {
        "metric": "cell-updates/sec (grid x batch), fused stencil+growth step",
        "value": m["value"],
        "unit": "cell-updates/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,                             # BASELINE.md publishes no throughput for this path
        "dtype": {"exact": "f32", "fast": "f32", "f64": "f64"}[args.precision],
        "data": "synthetic (device Philox initial state with the distribution of initialize_grid; ramped luminosity)",
        "config": {"workload": f"{args.workload}: {m['desc']}", "worlds_per_gpu": B, "grid": [G, G], "agents_per_world": N,
                   "precision": args.precision + {
                       "exact": " (float32 + float64 re-evaluation of near-tie cells: bit-identical to the float64 reference)",
                       "fast": " (float32: per step every cell within 1e-3 and >= 99.98 % of the cell values identical to the "
                               "float64 reference, profiles/r02_fast_tolerance.json)"}.get(args.precision, ""),
                   "plane_format": "binary16 per-mille (lossless for the quantised state)", "kernel": m["kernel"],
                   "library_build_id": _ffi.load().dw_build_id().decode(),   # content hash of csrc + header + flags
                   "total_worlds": int(all_stats.shape[0]),
                   "parallelism": f"ensemble shard x{n_gpus} (no data-path collective)"},
        "roofline": roofline(m),
        "preheat_s": round(m["preheat_s"], 3),
        "power": m["power"],                             # board W / limit / engine clock while the kernels run (or null)
        "rccl": group,
        "per_rank_value": per_rank,
    }
    if WORKLOADS[args.workload][0] != B and not args.worlds:
        out["config"]["note"] = f"{WORKLOADS[args.workload][0]} worlds did not fit this device: measured {B}"
    if not args.no_modes and args.precision in ("exact", "fast"):
        other = "fast" if args.precision == "exact" else "exact"
        out["modes"] = {other: brief(measure(args.workload, other, args.steps, args.warmup, min(args.preheat_s, 1.0), B))}
    if not args.no_workloads and args.workload == "target":
        # every BASELINE config in the same invocation, both arithmetic modes: c2 over the whole luminosity ramp
        # (512 steps after 64 of warm-up), the agent workloads c3 / c4 / c5 as short runs of their per-GPU shard
        out["workloads"] = {w: {p: brief(measure(w, p, k, wu, ph)) for p in ("exact", "fast")}
                            for w, (k, wu, ph) in EXTRA_WORKLOADS.items()}
    if rank == 0:
        cc = copy_ceiling()
        out["roofline"]["copy_ceiling"] = cc
        if out["roofline"]["traffic"]:
            out["roofline"]["measured_over_copy_ceiling"] = out["roofline"]["measured_frac"] * HBM_PEAK_GBS / cc["GB/s"]
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(G)
        out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
