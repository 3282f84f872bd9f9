#!/usr/bin/env python3
"""State-machine fuzz of the C ABI (through the Engine wrapper) against the NumPy oracle environment:
random interleavings of dw_step (with / without actions), dw_step_n (fused pairs, LDS-resident episodes),
dw_policy_greedy + dw_step_device_actions, dw_run_episode (LDS kernel, per-step launches, step pairs with
and without world flags), snapshots / restores and uploads, on every kernel family - after each operation
the planes, agents, per-world reductions, observations and the materialised 7-channel grid are compared
with the model.

usage: fuzz_engine.py [cases=30] [seed=1]"""
import copy
import os

os.environ.setdefault("DW_TEST_HOOKS", "1")     # the DW_TEST_* queue caps below are honoured only under it
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import therldaisyworld_amd as amd  # noqa: E402
from therldaisyworld_amd import _ffi  # noqa: E402
from oracle import daisy_oracle as O  # noqa: E402


def k(x):
    return np.rint(np.asarray(x) * 1000.0)


class Model:
    """The oracle environment driven like the engine: luminosity given per call, actions optional."""

    def __init__(self, dim, B, N):
        self.e = O.OracleDaisyWorld(grid_dimension=dim, n_agents=N, batch_size=B)
        self.B, self.N, self.dim = B, N, dim

    def load(self, light, dark, idx, st, L):
        e = self.e
        e.L = L
        e.set_initial_cover(light.copy(), dark.copy())
        e.agent_indices = idx.astype(np.int64).copy()
        e.agent_states = st.reshape(self.B, self.N, 1).astype(np.float64).copy()

    def step(self, L, action):
        e = self.e
        e.L = L
        if action is not None and self.N:
            e.update_agents(np.asarray(action).reshape(self.B, self.N, 1))
        e.grid = e.forward(e.grid)

    def greedy(self, argmin):
        obs = self.e.get_obs(self.e.agent_indices)
        return O.OracleGreedy(epsilon=0.0, greedy=not argmin)(obs) if self.N else None


def check(eng, m, what, log, seed, quantised):
    e = m.e
    gl, gd = eng.download_planes()
    bad = []
    if not (np.array_equal(k(gl), k(e.grid[:, 1])) and np.array_equal(k(gd), k(e.grid[:, 2]))):
        bad.append("planes")
    if m.N:
        idx, st = eng.download_agents()
        if not np.array_equal(idx, e.agent_indices):
            bad.append("agent_indices")
        if not np.array_equal(st[..., None], e.agent_states):
            bad.append("agent_states")
    if quantised:
        s = eng.reduce()
        if not (np.array_equal(s["sum_light_k"], k(e.grid[:, 1]).sum(axis=(1, 2)).astype(np.uint64))
                and np.array_equal(s["max_k"], np.maximum(k(e.grid[:, 1]).max(axis=(1, 2)),
                                                           k(e.grid[:, 2]).max(axis=(1, 2))).astype(np.uint32))):
            bad.append("reductions")
        if not np.array_equal(eng.download_grid(e.L), e.grid):
            bad.append("materialised grid")
        if m.N and not np.array_equal(eng.get_obs(e.L), e.get_obs(e.agent_indices)):
            bad.append("observations")
    if bad:
        log.append(f"seed {seed}: after {what}: {bad}")
    return not bad


def _run_case(seed, log):
    rng = np.random.RandomState(seed)
    dim = int(rng.choice([16, 32, 64, 96, 256, 260, 320]))
    B = int(rng.randint(1, 7 if dim <= 96 else 3))
    N = int(rng.randint(0, 5))
    if rng.rand() < 0.5:
        os.environ["DW_PACK_MIN_STRIPS"] = "1"
    else:
        os.environ.pop("DW_PACK_MIN_STRIPS", None)
    p = amd.default_params(B, dim, dim, N)
    p.precision = _ffi.PRECISION["exact"]
    eng = amd.Engine(p)
    eng.init_random(seed)
    light, dark = eng.download_planes()
    idx, st = eng.download_agents() if N else (np.zeros((B, 0, 2), np.int64), np.zeros((B, 0)))
    m = Model(dim, B, N)
    L = float(rng.uniform(0.8, 1.2))
    m.load(light, dark, idx, st, L)
    quantised = False
    saved = None
    ops = []
    ok = True
    for _ in range(int(rng.randint(5, 14))):
        op = rng.choice(["step_actions", "step_plain", "step_n", "policy_step", "episode", "episode_noflags", "snapshot",
                         "restore", "upload"])
        dL = float(rng.uniform(-0.005, 0.02))
        if op == "step_actions" and N:
            a = rng.randint(9, size=(B, N, 1))
            eng.step(L, a)
            m.step(L, a)
            quantised = True
        elif op in ("step_plain", "step_actions"):
            eng.step(L)
            m.step(L, None)
            quantised = True
        elif op == "step_n":
            n = int(rng.randint(1, 9))
            L2 = eng.step_n(n, L, dL, 0.6, 1.9)
            for _i in range(n):
                m.step(L, None)
                L = min(max(L + dL, 0.6), 1.9)
            assert L2 == L
            quantised = True
        elif op == "policy_step" and N and quantised:
            argmin = bool(rng.randint(2))
            a = m.greedy(argmin)
            eng.policy_greedy(argmin=argmin)
            eng.step_device_actions(L)
            m.step(L, a)
        elif op in ("episode", "episode_noflags") and N and quantised:
            K = int(rng.randint(2, 10))
            Ls = []
            for _i in range(K):
                L = min(max(L + dL, 0.6), 1.9)
                Ls.append(L)
            table = rng.randint(-2, 9, size=(K, B, N)).astype(np.int8)
            alive, okf = eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table, world_flags=(op == "episode"))
            for t in range(K):
                a = table[t].astype(np.int64)[..., None]
                g1, g2 = m.greedy(False), m.greedy(True)
                a = np.where(a == -1, g1, np.where(a == -2, g2, a))
                m.step(Ls[t], a)
                r = m.e.agent_states * (m.e.agent_states > 0)
                if not np.array_equal(okf[t][..., None], ~(r < 0.1)):
                    log.append(f"seed {seed}: agent flags of episode step {t} differ")
                    ok = False
                if alive is not None:
                    mx = np.maximum(k(m.e.grid[:, 1]).max(axis=(1, 2)), k(m.e.grid[:, 2]).max(axis=(1, 2)))
                    if not np.array_equal(alive[t], mx > 5):
                        log.append(f"seed {seed}: world flags of episode step {t} differ")
                        ok = False
        elif op == "snapshot" and quantised:
            eng.snapshot_save()
            saved = (copy.deepcopy(m.e.grid), m.e.agent_indices.copy(), m.e.agent_states.copy(), L)
        elif op == "restore" and saved is not None:
            eng.snapshot_restore()
            m.e.grid, m.e.agent_indices, m.e.agent_states = copy.deepcopy(saved[0]), saved[1].copy(), saved[2].copy()
            L = saved[3]
            m.e.L = L
        elif op == "upload":
            light = np.round(rng.rand(B, dim, dim) * 0.4 * (rng.rand(B, dim, dim) < 0.5), 3)
            dark = np.round(rng.rand(B, dim, dim) * 0.4 * (rng.rand(B, dim, dim) < 0.5), 3)
            eng.upload_state(light, dark)
            idx = rng.randint(dim, size=(B, N, 2))
            st = np.round(rng.rand(B, N), 3)
            if N:
                eng.upload_agents(idx, st)
            m.load(light, dark, idx, st, L)
            quantised = False
            saved = None
        else:
            continue
        ops.append(op)
        L = min(max(L + 0.003, 0.6), 1.9)
        if op in ("restore", "upload", "snapshot"):
            continue                                         # nothing stepped: the retained previous state is the model's business
        ok = check(eng, m, " > ".join(ops[-4:]), log, seed, quantised) and ok
        if not ok:
            break
    info = f"dim={dim} B={B} N={N} pack_min={os.environ.get('DW_PACK_MIN_STRIPS')} ops={len(ops)} :: {eng.kernel_info()[:34]}"
    eng.close()
    return ok, info


def run_case(seed, log):
    """_run_case under a randomly shrunk repair queue / mismatch list in a third of the cases (a side generator: the
    case itself is unchanged): the exact mode's overflow fallbacks must give the same results."""
    crng = np.random.RandomState((int(seed) * 2654435761 + 12345) % (2 ** 32))
    caps = {}
    if crng.rand() < 0.33:
        caps = {"DW_TEST_QUEUE_CAP": str(int(crng.choice([1, 4, 16]))), "DW_TEST_MISMATCH_CAP": str(int(crng.choice([0, 1, 2])))}
    saved = {k: os.environ.pop(k, None) for k in ("DW_TEST_QUEUE_CAP", "DW_TEST_MISMATCH_CAP")}
    os.environ.update(caps)                              # (read by the library whenever a handle is created)
    try:
        out = _run_case(seed, log)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    if caps and log:
        log[-1] = f"{log[-1]} {caps}"
    return out


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    log, nbad = [], 0
    for i in range(cases):
        ok, info = run_case(seed * 10000 + i, log)
        nbad += not ok
        print("ok  " if ok else "FAIL", seed * 10000 + i, info, flush=True)
    for line in log:
        print(line)
    print(f"{cases - nbad}/{cases} cases identical")
    sys.exit(1 if nbad else 0)
