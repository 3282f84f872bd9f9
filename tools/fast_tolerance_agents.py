#!/usr/bin/env python3
"""Calibration of the float32-only mode's tolerance ON AGENT WORKLOADS: what tests/test_gpu_configs.py asserts at
(measured x 2).  From identical, developed, quantised states the engine (`fast`) and the oracle run the same K-step
chunk with the ORACLE's actions (explicit table: the float32 planes are not bit-identical, so a device policy could
choose differently); measured per chunk: the largest plane deviation in quanta, the fraction of cell values that
differ, agent positions, the largest agent-state deviation.

  c3   2 x 1024 x 1024, one greedy agent per world      (BASELINE configs[2]'s grid: the four-wave ring)
  c5   1 x 2048 x 2048, 16 agents in C5's policy mix    (BASELINE configs[4]'s agents: overlapped strips)

usage (GPU box): python tools/fast_tolerance_agents.py [--out profiles/r03_fast_tolerance_agents.json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402
from tests.test_gpu_configs import _c5_table, _engine, _k, _oracle_like, _oracle_step, _resolve_codes  # noqa: E402


def chunk_deviation(case, develop, K, seed):
    B, G, N = (2, 1024, 1) if case == "c3" else (1, 2048, 16)
    eng = _engine(amd, B, G, G, N, "fast")
    eng.init_random(seed)
    dL = 0.75 / 512
    L = eng.step_n(develop, 0.75, dL, 0.75, 1.5)            # a developed, quantised state (no agents acting yet)
    env = _oracle_like(eng, G, L)
    Ls = [min(L + i * dL, 1.5) for i in range(K)]
    rng = np.random.RandomState(seed)
    codes = np.full((K, B, N), -1, dtype=np.int8) if case == "c3" else _c5_table(rng, K, B)
    table = np.zeros((K, B, N), dtype=np.int8)
    for t in range(K):
        a = _resolve_codes(env, codes[t])
        table[t] = a[..., 0]
        _oracle_step(env, Ls[t], a)
    eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table, world_flags=False)
    gl, gd = eng.download_planes()
    dl, dd = np.abs(_k(gl) - _k(env.grid[:, 1])), np.abs(_k(gd) - _k(env.grid[:, 2]))
    idx, st = eng.download_agents()
    row = {"case": case, "developed_steps": develop, "L": L, "K": K, "seed": seed,
           "max_deviation_quanta": int(max(dl.max(), dd.max())),
           "differing_fraction": (np.count_nonzero(dl) + np.count_nonzero(dd)) / (2.0 * dl.size),
           "positions_equal": bool(np.array_equal(idx, env.agent_indices)),
           "max_agent_state_deviation": float(np.abs(st[..., None] - env.agent_states).max())}
    eng.close()
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rows = []
    for case in ("c3", "c5"):
        for develop in (40, 200, 360):
            for K in (6, 12):
                row = chunk_deviation(case, develop, K, 42 if case == "c3" else 7)
                rows.append(row)
                print(row, flush=True)
    worst = {c: {"max_deviation_quanta": max(r["max_deviation_quanta"] for r in rows if r["case"] == c),
                 "differing_fraction_per_step": max(r["differing_fraction"] / r["K"] for r in rows if r["case"] == c),
                 "differing_fraction_K6": max(r["differing_fraction"] for r in rows if r["case"] == c and r["K"] == 6),
                 "max_agent_state_deviation": max(r["max_agent_state_deviation"] for r in rows if r["case"] == c),
                 "positions_equal": all(r["positions_equal"] for r in rows if r["case"] == c)} for c in ("c3", "c5")}
    out = {"tool": "tools/fast_tolerance_agents.py", "library_build_id": _ffi.load().dw_build_id().decode(),
           "rows": rows, "worst": worst,
           "asserted_by": "tests/test_gpu_configs.py::test_c3_grid_agents_fast_vs_oracle_tolerance / "
                          "test_c5_agent_mix_fast_vs_oracle_tolerance at (measured x 2)"}
    print(json.dumps(worst))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
