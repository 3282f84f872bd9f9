"""The N>1 path on CPU: world sharding and the end-of-run gather over torch.distributed (gloo,
world_size 2).  The data path itself has no collective (worlds are independent)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_worlds_tiles_the_range():
    from therldaisyworld_amd.ensemble import shard_worlds
    for total in (1, 7, 8, 1000, 8000, 1024):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_worlds(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0
            for (o0, c0), (o1, c1) in zip(blocks, blocks[1:]):
                assert o0 + c0 == o1
            assert blocks[-1][0] + blocks[-1][1] == total
            counts = [c for _, c in blocks]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_worlds(10, 2, 2)


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["DW_ROOT"])
from therldaisyworld_amd import ensemble
dist = ensemble.init_process_group("gloo")
rank, _, world = ensemble.rank_info()
total = 7                                    # ragged: 4 + 3 worlds
off, cnt = ensemble.shard_worlds(total, rank, world)
# per-world "lifespans" that encode the global world id, as a rank would produce them
done_at = (np.arange(off, off + cnt) * 10 + 1).astype(np.int32)
agents = np.stack([np.arange(off, off + cnt) * 100 + n for n in range(4)], axis=1).astype(np.int32)[..., None]
stats = np.zeros(cnt, dtype=[("max_k", "<u4"), ("reserved", "<u4"), ("sum_light_k", "<u8"), ("sum_dark_k", "<u8")])
stats["sum_light_k"] = np.arange(off, off + cnt) + 5
g_done = ensemble.gather_per_world(done_at, counts=[4, 3])
g_agents = ensemble.gather_per_world(agents)
g_stats = ensemble.gather_per_world(stats)
assert g_done.tolist() == [i * 10 + 1 for i in range(total)], g_done
assert g_agents.shape == (total, 4, 1) and g_agents[:, 2, 0].tolist() == [i * 100 + 2 for i in range(total)]
assert g_stats["sum_light_k"].tolist() == [i + 5 for i in range(total)]
t = ensemble.max_over_ranks(1.0 + rank)
assert t == float(world)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_gather_over_gloo_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DW_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} ok" in out


ONE_RANK = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["DW_ROOT"])
from therldaisyworld_amd import ensemble
dist = ensemble.init_process_group("gloo")
assert dist is not None and dist.get_world_size() == 1
x = (np.arange(12, dtype=np.int32).reshape(3, 4) * 7)
assert np.array_equal(ensemble.gather_per_world(x, counts=[3]), x)
assert ensemble.max_over_ranks(2.5) == 2.5 and ensemble.agree_on_worlds(37) == 37
assert ensemble.gather_scalars(1.25) == [1.25]
g = ensemble.describe_group()
assert g["backend"] == "gloo" and g["world_size"] == 1 and g["ranks_reporting"] == [0], g
dist.barrier()
dist.destroy_process_group()
print("one-rank ok")
"""


def test_one_rank_group_runs_the_collectives_when_forced():
    """DW_DIST_FORCE_COLLECTIVES=1 (the GPU box's rehearsal of the RCCL path, tests/test_bench_contract.py): a group of one
    rank is created and every helper goes through its collective instead of short-circuiting."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(DW_ROOT=ROOT, DW_DIST_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, "-c", ONE_RANK], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "one-rank ok" in p.stdout, p.stderr[-2000:]
    # without the switch a single process makes no group at all
    env.pop("DW_DIST_FORCE_COLLECTIVES")
    p = subprocess.run([sys.executable, "-c", "import os, sys; sys.path.insert(0, os.environ['DW_ROOT']); "
                        "from therldaisyworld_amd import ensemble; assert ensemble.init_process_group('gloo') is None; "
                        "assert ensemble.describe_group()['backend'] is None"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
