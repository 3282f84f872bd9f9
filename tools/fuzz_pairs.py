#!/usr/bin/env python3
"""Differential fuzz of the paired-step paths: random shapes / agent counts / policies / luminosity
schedules, dw_run_episode with step pairs (fused launch + look-ahead patch, with and without world flags)
against one launch per step (DW_NO_AGENT_FUSE=1), everything compared bit for bit.

usage: fuzz_pairs.py [cases=60] [seed=1]"""
import os

os.environ.setdefault("DW_TEST_HOOKS", "1")     # the DW_TEST_* queue caps below are honoured only under it
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def run_case(case_seed):
    import therldaisyworld_amd as amd
    from therldaisyworld_amd import _ffi
    rng = np.random.RandomState(case_seed)
    W = int(rng.choice([8, 16, 32, 64, 128, 256, 260, 320, 512, 516]))
    H = int(rng.randint(3, 90))
    B = int(rng.randint(1, 40 if W < 256 else 6))
    N = int(rng.randint(1, 9))
    prec = str(rng.choice(["exact", "fast"]))
    K = int(rng.randint(3, 24))
    flags = bool(rng.randint(2))
    mode = int(rng.choice([_ffi.POLICY_ARGMAX, _ffi.POLICY_ARGMIN, _ffi.POLICY_ZEROS, _ffi.POLICY_TABLE]))
    use = (rng.rand(K) < 0.3).astype(np.uint8) if mode in (_ffi.POLICY_ARGMAX, _ffi.POLICY_ARGMIN) else None
    table = rng.randint(-2, 9, size=(K, B, N)).astype(np.int8) if (mode == _ffi.POLICY_TABLE or use is not None) else None
    L0, dL = float(rng.uniform(0.8, 1.3)), float(rng.uniform(-0.01, 0.03))
    p = amd.default_params(B, H, W, N)
    p.precision = _ffi.PRECISION[prec]
    p.agent_gamma = float(rng.choice([0.05, 0.2]))           # some agents starve inside the run
    caps = {}
    if rng.rand() < 0.3:                                     # shrunk repair queue / mismatch list: the overflow fallbacks
        caps = {"DW_TEST_QUEUE_CAP": str(int(rng.choice([1, 4, 16]))), "DW_TEST_MISMATCH_CAP": str(int(rng.choice([0, 1, 2])))}
    saved = {k: os.environ.pop(k, None) for k in ("DW_TEST_QUEUE_CAP", "DW_TEST_MISMATCH_CAP")}
    os.environ.update(caps)
    try:
        eng = amd.Engine(p)                                  # (the library reads them at handle creation)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    eng.init_random(case_seed)
    eng.step(L0, np.zeros((B, N, 1), dtype=int))
    Ls = [min(max(L0 + dL * (i + 1), 0.6), 2.2) for i in range(K)]
    alive, ok = eng.run_episode(Ls, mode, use, table, world_flags=flags)
    out = [ok, *eng.download_planes(), *eng.download_planes(1), *eng.download_agents(), eng.reduce().tobytes(),
           eng.get_obs()]
    if flags:
        out.append(alive)
    info = f"B={B} H={H} W={W} N={N} {prec} K={K} mode={mode} flags={flags} {caps} :: {eng.kernel_info()[:40]}"
    eng.close()
    return info, out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        seeds = [int(v) for v in sys.argv[3:]]
        np.savez(sys.argv[2], **{f"s{s}_{i}": np.asarray(a) for s in seeds for i, a in enumerate(run_case(s)[1])})
        sys.exit(0)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seeds = [seed * 1000 + i for i in range(cases)]
    os.environ["DW_PACK_MIN_STRIPS"] = "1"
    # the reference run (no pairs) in a child process: the library reads its environment at handle creation
    ref_path = "/tmp/fuzz_pairs_ref.npz"
    env = dict(os.environ, DW_NO_AGENT_FUSE="1")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", ref_path, *map(str, seeds)], env=env)
    ref = np.load(ref_path)
    bad = 0
    for s in seeds:
        info, out = run_case(s)
        same = all(np.array_equal(np.asarray(a), ref[f"s{s}_{i}"]) for i, a in enumerate(out))
        bad += not same
        print(("ok  " if same else "FAIL"), s, info, flush=True)
    print(f"{cases - bad}/{cases} cases identical")
    sys.exit(1 if bad else 0)
