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


# ---- supervision of the self-launched ranks (ensemble.launch_ranks) -----------------------------------------
def _spawn(extra, env_extra, timeout=120):
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-spawn", "--backend", "gloo", *extra],
                       capture_output=True, text=True, env=env, timeout=timeout)
    return p, time.monotonic() - t0


def test_a_rank_that_dies_before_the_rendezvous_ends_the_run_promptly():
    # rank 1 exits 3 at once; rank 0 would otherwise wait in the rendezvous for its whole timeout
    p, took = _spawn(["--gpus", "2", "--rank-timeout-s", "200"], {"DW_SELFTEST_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert took < 30, f"the parent took {took:.1f} s"
    assert "(1, 3)" in p.stderr and "rank 1 exits 3" in p.stderr      # who failed, and its own stderr tail
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_rank_zero_failing_is_reported_too():
    p, took = _spawn(["--gpus", "2", "--rank-timeout-s", "200"], {"DW_SELFTEST_FAIL_RANK": "0"})
    assert p.returncode != 0 and took < 30 and "(0, 3)" in p.stderr


def test_a_rank_that_never_finishes_is_killed_at_the_deadline():
    # rank 1 sleeps for an hour; rank 0's rendezvous would time out only after 60 s: the deadline comes first
    p, took = _spawn(["--gpus", "2", "--rank-timeout-s", "6"],
                     {"DW_SELFTEST_HANG_RANK": "1", "DW_SELFTEST_DIST_TIMEOUT_S": "60"})
    assert p.returncode == 124, (p.returncode, p.stderr[-1500:])
    assert 5 < took < 40, f"the parent took {took:.1f} s"
    assert "--rank-timeout-s" in p.stderr


def test_launch_ranks_keeps_logs_on_request(tmp_path):
    p, _ = _spawn(["--gpus", "2", "--worlds", "2", "--rank-log-dir", str(tmp_path)], {})
    assert p.returncode == 0, p.stderr[-2000:]
    assert sorted(os.listdir(tmp_path)) == ["rank0.err", "rank0.out", "rank1.err", "rank1.out"]
    line = [ln for ln in (tmp_path / "rank0.out").read_text().splitlines() if ln.startswith("{")][0]
    assert json.loads(line)["total_worlds"] == 4


def test_lifespan_sweep_launches_and_gathers_its_own_ranks():
    # C4's 8-GPU half as one command (tools/lifespan_sweep.py --gpus N); here 2 ranks over gloo, no GPU
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lifespan_sweep.py"), "--selftest", "--gpus", "2",
                        "--worlds", "5", "--agents", "3"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["worlds"] == 10 and d["in_rank_order"]
    assert abs(d["wall_s"] - 0.02) < 1e-12


def test_a_signal_to_the_parent_takes_every_rank_down(tmp_path):
    """The ranks run in sessions of their own, so an outer `timeout` / scheduler / closed terminal that signals the parent
    does not reach them by itself: the parent turns SIGTERM (and SIGHUP) into a clean stop of every rank.  Here rank 1
    sleeps for an hour; the parent gets SIGTERM after the ranks are up; afterwards no process of the run is alive."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DW_SELFTEST_HANG_RANK="1", DW_SELFTEST_DIST_TIMEOUT_S="600")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-spawn", "--backend", "gloo", "--gpus", "2",
                          "--rank-timeout-s", "600", "--rank-log-dir", str(tmp_path)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def rank_pids():
        me, out = str(p.pid), []
        for d in os.listdir("/proc"):
            if d.isdigit():
                try:
                    with open(f"/proc/{d}/stat") as f:
                        fields = f.read().rsplit(")", 1)[1].split()
                    if fields[1] == me:                       # ppid
                        out.append(int(d))
                except OSError:
                    pass
        return out

    t_end = time.monotonic() + 60
    kids = []
    while time.monotonic() < t_end and len(kids) < 2:         # both ranks started
        kids = rank_pids()
        time.sleep(0.1)
    assert len(kids) == 2, kids
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    try:
        p.wait(timeout=30)
    except subprocess.TimeoutExpired:
        p.kill()
        raise AssertionError("the parent did not exit after SIGTERM")
    assert p.returncode != 0
    t_end = time.monotonic() + 10
    while time.monotonic() < t_end and any(os.path.exists(f"/proc/{k}") for k in kids):
        time.sleep(0.1)
    assert not [k for k in kids if os.path.exists(f"/proc/{k}")], "a rank survived its parent's SIGTERM"


def test_a_rank_that_could_only_allocate_half_the_worlds_sets_the_count_for_all():
    """bench.py `make_engine` halves a rank's world count while its device cannot hold the state (DW_ENOMEM); weak
    scaling then needs EVERY rank to step that smaller count (ensemble.agree_on_worlds: one MIN all-reduce, also the
    first collective after the allocations).  The line also says which backend ran, with how many ranks, and lists
    every rank's own throughput."""
    d = _run(["--gpus", "2", "--worlds", "8"], {"DW_SELFTEST_HALVE_RANK": "1"})
    assert d["worlds_per_rank"] == 4 and d["total_worlds"] == 8
    assert d["worlds"] == [0, 1, 2, 3, 4, 5, 6, 7]
    assert d["rccl"]["backend"] == "gloo" and d["rccl"]["world_size"] == 2 and d["rccl"]["ranks_reporting"] == [0, 1]
    assert d["per_rank_value"] == [1.0, 2.0]
    d1 = _run(["--gpus", "1", "--worlds", "4"])
    assert d1["rccl"] == {"backend": None, "world_size": 1, "nccl_version": None, "ranks_reporting": [0]}
