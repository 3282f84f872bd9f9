"""bench.py --gpus N launches its own ranks when it is not started by torchrun (the reference's own launcher
spawns its workers itself too, daisy/evo/sges.py:215-245).  CPU rehearsal of that plumbing over gloo: rank
launch with the torchrun environment, rendezvous on 127.0.0.1, barrier, max-over-ranks, per-world gather and the
relay of rank 0's JSON line - no GPU is touched (`--selftest-spawn`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-spawn", "--backend", "gloo", *extra],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    d = _run(["--gpus", "2", "--worlds", "3"])
    assert d["n_gpus"] == 2 and d["total_worlds"] == 6
    assert d["worlds"] == [0, 1, 2, 3, 4, 5]                 # rank order, contiguous blocks
    assert abs(d["max_elapsed"] - 0.002) < 1e-12             # the MAX over ranks


def test_bench_single_rank_needs_no_launcher():
    d = _run(["--gpus", "1", "--worlds", "4"])
    assert d["n_gpus"] == 1 and d["worlds"] == [0, 1, 2, 3]


def test_bench_refuses_a_mismatched_torchrun_environment():
    env = {k: v for k, v in os.environ.items()}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
