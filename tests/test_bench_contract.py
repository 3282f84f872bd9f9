"""bench.py's output contract on a GPU box (`-m gpu`): ONE JSON line with the keys the driver parses, the roofline
object computed per SURVEY 8(d) (4 x sizeof(plane element) bytes per cell-update, never above the peak) and the CPU
baseline beside it - on a small ensemble so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_follows_the_contract():
    d = _bench("--workload", "target", "--worlds", "8", "--steps", "12", "--warmup", "4", "--preheat-s", "0.2")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["unit"] == "cell-updates/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["worlds_per_gpu"] == 8
    from therldaisyworld_amd import build
    assert d["config"]["library_build_id"] == build.source_id()      # the line names the sources it was measured on
    assert abs(d["value"] - 8 * 4096 * 4096 * 12 / (d["ms_per_step"] * 12e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bytes_per_cell_update"] == 4 * r["plane_elem_bytes"] == 8 and r["steps_per_launch"] == 2
    assert r["launches_timed"] == 5                        # 12 steps = 5 fused pairs + 2 single steps
    assert r["algorithmic_bytes_per_launch"] == 8 * 2 * 8 * 4096 * 4096
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-9
    assert 0.2 < r["frac"] <= 1.0 and r["peak"] == 8000.0 and r["unit"] == "GB/s" and r["bound"] in ("hbm", "valu")
    assert r["measured_frac"] is None or r["measured_frac"] <= r["frac"]
    assert 2000.0 < r["copy_ceiling"]["GB/s"] <= 8000.0     # a device copy, live: below the spec peak, well above PCIe
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e6 and c["unit"] == "cell-updates/s" and c["sample"]
    for mode in d["modes"].values():
        assert mode["roofline"]["frac"] <= 1.0
    assert set(d["workloads"]["c2"]) == {"exact", "fast"}
    # what ran: a single process has no collective backend; its own throughput is the whole job's
    assert d["rccl"] == {"backend": None, "world_size": 1, "nccl_version": None, "ranks_reporting": [0]}
    assert len(d["per_rank_value"]) == 1 and d["per_rank_value"][0] >= d["value"] * 0.999
    # board power / engine clock read beside an untimed extra pass of the same launches (null when rocm-smi cannot be read)
    pw = d["power"]
    assert pw["board_w"] is None or (50.0 < pw["board_w"] < 2000.0 and 100 <= pw["sclk_mhz"] <= 3000 and pw["samples"])


def test_bench_agent_workload_and_self_launched_ranks():
    d = _bench("--workload", "c3", "--worlds", "4", "--steps", "9", "--warmup", "3", "--preheat-s", "0.1", "--no-cpu-baseline",
               "--no-modes")
    assert d["config"]["agents_per_world"] == 1 and d["roofline"]["steps_per_launch"] == 1 and d["value"] > 0
    env = dict(os.environ, DW_BENCH_ALL_RANKS_ON_DEVICE0="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "c2",
                        "--worlds", "64", "--steps", "16", "--warmup", "4", "--preheat-s", "0.1", "--no-cpu-baseline",
                        "--no-modes"], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in env.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert p.returncode == 0, p.stderr[-3000:]
    d2 = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d2["config"]["total_worlds"] == 128 and d2["cpu_baseline"] is None
    assert d2["rccl"]["backend"] == "gloo" and d2["rccl"]["world_size"] == 2 and d2["rccl"]["ranks_reporting"] == [0, 1]
    assert len(d2["per_rank_value"]) == 2 and sum(d2["per_rank_value"]) >= d2["value"] * 0.999


def test_bench_through_a_one_rank_rccl_group():
    """The N > 1 path's collectives on RCCL itself: a 1-GPU box cannot hold two nccl ranks, but a ONE-rank nccl group can be
    made to run every collective of a run (DW_DIST_FORCE_COLLECTIVES=1: rendezvous, communicator bound to the device,
    the MIN / MAX all-reduces on float64, the all-gathers of int64 / float64 / padded uint8 per-world statistics,
    barriers, destroy) - device tensors, dtypes and the library's version string as the 8-GPU run will use them."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(DW_DIST_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "nccl", "--workload", "c2", "--worlds", "64",
                        "--steps", "16", "--warmup", "4", "--preheat-s", "0.1", "--no-cpu-baseline", "--no-modes",
                        "--no-workloads"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["total_worlds"] == 64
    assert d["rccl"]["backend"] == "nccl" and d["rccl"]["world_size"] == 1 and d["rccl"]["ranks_reporting"] == [0]
    assert d["rccl"]["nccl_version"]                        # RCCL answered for itself
    assert len(d["per_rank_value"]) == 1 and d["per_rank_value"][0] >= d["value"] * 0.999
