"""Ensemble sharding: independent worlds split over the GPUs of a node, one process per GPU.

Worlds never interact (every operation of the path is per world; the only shared quantity is the
scalar luminosity, a deterministic function of the step count — SURVEY.md §8e), so the data path has
NO collective: each rank owns a contiguous block of worlds, keyed for the device RNG by its global
world offset, and steps it independently.  The only communication is a gather of small per-world
results (lifespans, reductions) at the end of a run, over ``torch.distributed`` (backend "nccl" =
RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
"""
from __future__ import annotations

import os

import numpy as np


def rank_info():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_worlds(total_worlds: int, rank: int, world_size: int):
    """Contiguous block [offset, offset+count) of `total_worlds` for `rank`; the first
    ``total_worlds % world_size`` ranks get one extra world.  Blocks tile the range exactly."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, extra = divmod(int(total_worlds), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def init_process_group(backend: str | None = None, device: int | None = None, timeout_s: float = 120.0):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process).  `device`: the
    GPU of this rank when it is not LOCAL_RANK (rehearsals with several ranks on one GPU).  `timeout_s` bounds the
    rendezvous and every later collective: a rank that never arrives makes the others fail after that long
    instead of sitting in the store for torch's default 10-30 minutes.  With the nccl (= RCCL) backend the
    process group is bound to this rank's GPU at creation (`device_id`), so a wrong device ordinal fails here,
    before any collective."""
    import datetime

    import torch
    import torch.distributed as dist
    rank, local_rank, world = rank_info()
    if world == 1 and os.environ.get("DW_DIST_FORCE_COLLECTIVES", "0") != "1":
        return None
    if world == 1:                                            # the one-rank rehearsal (see _no_group)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {"timeout": datetime.timedelta(seconds=float(timeout_s))}
        if backend == "nccl":
            dev = local_rank if device is None else device
            ndev = torch.cuda.device_count()
            if not (0 <= dev < ndev):
                raise RuntimeError(f"rank {rank}: GPU {dev} requested but this process sees {ndev} device(s)")
            torch.cuda.set_device(dev)
            kwargs["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return dist


# ------------------------------------------------------------------------------------------------
# self-launched ranks (the reference's launcher analogue: daisy/evo/sges.py:215-245, 401-412)
# ------------------------------------------------------------------------------------------------
def _tail(path: str, nbytes: int = 3000) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            f.seek(max(0, f.tell() - nbytes))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def launch_ranks(script: str, argv, n_ranks: int, rank_timeout_s: float = 300.0, log_dir: str | None = None,
                 extra_env=None, poll_s: float = 0.1) -> int:
    """Start `n_ranks` fresh child processes of `script` (one per GPU) with the torchrun environment set
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), SUPERVISE them, relay rank 0's
    stdout, and return the exit code for the parent (0 only if every rank exited 0).

    * every child is polled; the first non-zero exit terminates the others (SIGTERM, SIGKILL after 5 s) - a rank
      that dies before the rendezvous must not leave the rest waiting in it;
    * `rank_timeout_s` bounds the whole run: past it every child is killed and the parent returns 124;
    * all ranks log to files (`log_dir`, default a fresh temporary directory): stdout of rank 0 is relayed when
      the run ends, the stderr tail of every failed rank is relayed to the parent's stderr;
    * a rendezvous port that was taken between its selection and rank 0's bind ("address already in use") gets
      ONE fresh attempt on another port.

    * the ranks run in sessions of their own, so a signal to the parent's process group does not reach them: while
      they run, SIGTERM / SIGHUP to the parent (an outer `timeout`, a scheduler, a closed terminal) are turned into
      SystemExit, so that the `finally` below stops every rank; and every rank asks the kernel to send it SIGKILL when
      its parent dies (PR_SET_PDEATHSIG, set between fork and exec - before anything touches a GPU) for the cases no
      handler sees (SIGKILL of the parent).

    The parent never imports torch or the HIP library and nothing is re-exec'd: children are fresh processes,
    and only those are ever signalled (by the exact process group each was started in)."""
    import signal
    import socket
    import subprocess
    import sys
    import tempfile
    import time

    import threading

    own_dir = log_dir is None
    if own_dir:
        log_dir = tempfile.mkdtemp(prefix="dw_ranks_")
    os.makedirs(log_dir, exist_ok=True)

    def die_with_parent():                                      # runs in the child between fork and exec
        try:
            import ctypes
            ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)     # PR_SET_PDEATHSIG
        except Exception:
            pass

    def on_signal(signum, frame):
        raise SystemExit(128 + signum)

    restore = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGHUP):
            restore[sig] = signal.signal(sig, on_signal)

    def stop(procs):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)            # the child's own session: it and what it started
                except (ProcessLookupError, PermissionError):
                    pass
        t_end = time.monotonic() + 5.0
        for p in procs:
            while p.poll() is None and time.monotonic() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                p.wait()

    def attempt(tag):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs, files = [], []
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.update(extra_env or {})
            out = open(os.path.join(log_dir, f"rank{r}{tag}.out"), "wb")
            err = open(os.path.join(log_dir, f"rank{r}{tag}.err"), "wb")
            files += [out, err]
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(script), *argv], env=env, stdout=out,
                                          stderr=err, start_new_session=True, preexec_fn=die_with_parent))
        deadline = time.monotonic() + float(rank_timeout_s)
        code, why = 0, ""
        try:
            while True:
                rcs = [p.poll() for p in procs]
                bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
                if bad:
                    code, why = 1, f"ranks failed (rank, exit code): {bad}; the other ranks were terminated"
                    break
                if all(rc == 0 for rc in rcs):
                    break
                if time.monotonic() > deadline:
                    code = 124
                    why = (f"ranks still running after --rank-timeout-s {rank_timeout_s:g}: "
                           f"{[r for r, rc in enumerate(rcs) if rc is None]}; all ranks were killed")
                    break
                time.sleep(poll_s)
        finally:
            stop(procs)
            for f in files:
                f.close()
        return code, why, [p.returncode for p in procs]

    tag = ""
    try:
        code, why, rcs = attempt(tag)
        if code == 1 and any("ddress already in use" in _tail(os.path.join(log_dir, f"rank{r}.err")) for r in range(n_ranks)):
            tag = ".retry"
            code, why, rcs = attempt(tag)
    finally:
        for sig, old in restore.items():
            signal.signal(sig, old)
    sys.stdout.write(_tail(os.path.join(log_dir, f"rank0{tag}.out"), 1 << 22))
    sys.stdout.flush()
    if code:
        sys.stderr.write(f"launch_ranks: {why}\n")
        for r, rc in enumerate(rcs):
            if rc != 0:
                sys.stderr.write(f"--- rank {r} (exit {rc}) stderr tail [{log_dir}/rank{r}{tag}.err] ---\n")
                sys.stderr.write(_tail(os.path.join(log_dir, f"rank{r}{tag}.err")) + "\n")
        sys.stderr.flush()
    elif own_dir:
        import shutil
        shutil.rmtree(log_dir, ignore_errors=True)
    return code


def _no_group() -> bool:
    """True when there is nothing to communicate with: no process group, or a group of one rank - unless
    DW_DIST_FORCE_COLLECTIVES=1 asks for the collectives anyway (the GPU rehearsal of the RCCL path on a 1-GPU box:
    every collective of a run on a one-rank nccl group, tests/test_gpu_round4.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get("DW_DIST_FORCE_COLLECTIVES", "0") != "1"


def gather_per_world(local: np.ndarray, counts=None, device=None) -> np.ndarray:
    """Concatenate per-world arrays (leading axis = this rank's worlds) from all ranks, in rank
    order, on every rank.  Ragged shards are handled by padding to the largest shard."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local)
    if _no_group():
        return local.copy()
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(ns, n_local)
    ns = [int(n.item()) for n in ns]
    if counts is not None and list(counts) != ns:
        raise RuntimeError(f"shard sizes {ns} do not match the expected {list(counts)}")
    nmax = max(ns)
    flat = local.reshape(local.shape[0], -1)
    as_bytes = np.ascontiguousarray(flat).view(np.uint8).reshape(local.shape[0], -1)
    pad = np.zeros((nmax, as_bytes.shape[1]), dtype=np.uint8)
    pad[: local.shape[0]] = as_bytes
    t = torch.from_numpy(pad).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    parts = []
    for n, o in zip(ns, outs):
        b = o.cpu().numpy()[:n]
        parts.append(np.ascontiguousarray(b).view(local.dtype).reshape((n,) + local.shape[1:]))
    return np.concatenate(parts, axis=0)


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if _no_group():
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def agree_on_worlds(worlds_here: int, device=None) -> int:
    """Weak scaling needs every rank to step the SAME number of worlds: the smallest any rank could allocate (a rank whose
    device could not hold the requested ensemble halves its count, bench.py `make_engine`).  One MIN all-reduce; it is
    also the first collective after the ranks' allocations and initial draws, i.e. the barrier behind which the
    collective timeout no longer spans that work."""
    return -int(max_over_ranks(-float(worlds_here), device))


def describe_group(device=None) -> dict:
    """What actually ran, for the bench line: backend, world size, the collective library's version and - from an
    all-gather of the rank ids over that backend - which ranks answered.  Single process: backend None."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "nccl_version": None, "ranks_reporting": [0]}
    backend = dist.get_backend()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else "cpu"
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())     # RCCL reports itself through this call
        except Exception:
            version = None
    return {"backend": backend, "world_size": dist.get_world_size(), "nccl_version": version,
            "ranks_reporting": sorted(int(t.item()) for t in got)}


def gather_scalars(value: float, device=None) -> list:
    """One float per rank, in rank order, on every rank."""
    import torch
    import torch.distributed as dist
    if _no_group():
        return [float(value)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    got = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(got, mine)
    return [float(t.item()) for t in got]

