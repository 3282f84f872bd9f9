"""GPU parity tests at the BASELINE.json workloads with agents on wide grids (run with ``-m gpu``):

  C3  256 worlds x 1024x1024, 1 greedy agent per world      (ref update_agents daisy_world_rl.py:181-244,
  C5  8 worlds x 8192x8192, 16 mixed-policy agents per world  Greedy agents/greedy.py:14-36)

W = 1024 runs as five overlapped 248-column wave-strips, W = 2048 / 8192 as 9 / 34, with the agents' step
of every pair patched into the fused result (dw_agents_fused.hpp) - these tests compare that path with the
ORACLE (NumPy environment, physics pass by the C restatement: oracle.OracleDaisyWorldC + OracleGreedy),
not with the HIP path itself: planes, agent positions and states, observations, rewards, done flags,
per-step agent / world flags after every chunk.  Full-size runs (256 x 1024^2, 8 x 8192^2 x 16) check the
step pairs against one launch per step plus the reductions' checksums, and one full-size 8192^2 world
with its 16 agents against the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import daisy_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import therldaisyworld_amd as t
    return t


def _k(x):
    return np.rint(np.asarray(x) * 1000.0).astype(np.int64)


def _engine(amd, B, H, W, N, precision):
    from therldaisyworld_amd import _ffi
    p = amd.default_params(B, H, W, N)
    p.precision = _ffi.PRECISION[precision]
    return amd.Engine(p)


def _oracle_like(eng, dim, L):
    """Oracle environment holding the engine's current (downloaded) state."""
    light, dark = eng.download_planes()
    idx, st = eng.download_agents()
    env = O.OracleDaisyWorldC(grid_dimension=dim, n_agents=eng.N, batch_size=eng.B)
    env.L = L
    env.set_initial_cover(light, dark)
    env.agent_indices = idx.astype(np.int64)
    env.agent_states = st.reshape(eng.B, eng.N, 1).copy()
    return env


def _oracle_step(env, L, action):
    """One reference step at luminosity L; returns (reward, done) of that step (ref :486-492)."""
    env.L = L
    _, reward, done, _ = env.step(np.asarray(action).reshape(env.P.batch_size, env.P.n_agents, 1).astype(np.int64))
    return reward, done


def _resolve_codes(env, codes):
    """Table codes -> actions on the oracle: -1 / -2 are the greedy / anti-greedy choice of that agent
    (ref Greedy.__call__, agents/greedy.py:25-30, epsilon = 0)."""
    obs = env.get_obs(env.agent_indices)
    g1 = O.OracleGreedy(epsilon=0.0, greedy=True)(obs)
    g2 = O.OracleGreedy(epsilon=0.0, greedy=False)(obs)
    c = codes.astype(np.int64)[..., None]
    return np.where(c == -1, g1, np.where(c == -2, g2, c))


def _c5_table(rng, K, B):
    """The C5 policy mix by agent index: 0-3 greedy, 4-7 anti-greedy, 8-11 random, 12-15
    half-random (one coin per step for the whole batch, as Greedy draws it)."""
    table = np.empty((K, B, 16), dtype=np.int8)
    table[:, :, 0:4] = -1
    table[:, :, 4:8] = -2
    table[:, :, 8:12] = rng.randint(9, size=(K, B, 4))
    coin = rng.rand(K) > 0.5
    table[:, :, 12:16] = np.where(coin[:, None, None], -1, rng.randint(9, size=(K, B, 4)))
    return table


def _compare_exact(eng, env, L_last, what):
    gl, gd = eng.download_planes()
    assert np.array_equal(_k(gl), _k(env.grid[:, 1])), f"{what}: light plane"
    assert np.array_equal(_k(gd), _k(env.grid[:, 2])), f"{what}: dark plane"
    idx, st = eng.download_agents()
    assert np.array_equal(idx, env.agent_indices), f"{what}: agent positions"
    assert np.array_equal(st[..., None], env.agent_states), f"{what}: agent states"
    assert np.array_equal(eng.get_obs(L_last), env.get_obs(env.agent_indices)), f"{what}: observations"
    s = eng.reduce()
    assert np.array_equal(s["sum_light_k"], _k(env.grid[:, 1]).sum(axis=(1, 2)).astype(np.uint64)), f"{what}: sums"
    assert np.array_equal(s["max_k"], np.maximum(_k(env.grid[:, 1]).max(axis=(1, 2)),
                                                  _k(env.grid[:, 2]).max(axis=(1, 2))).astype(np.uint32)), f"{what}: max"


def _run_chunks_exact(amd, eng, env, L, dL, chunks, policy, rng, world_flags):
    """Drive the engine's device-resident episode loop and the oracle through the same chunks; compare
    everything after every chunk.  policy: "greedy" (device policy, every agent) or "c5" (table codes)."""
    from therldaisyworld_amd import _ffi
    B, N = eng.B, eng.N
    for ci, K in enumerate(chunks):
        Ls = [L + i * dL for i in range(K)]
        L += K * dL
        if policy == "greedy":
            table = None
            alive, ok = eng.run_episode(Ls, _ffi.POLICY_ARGMAX, world_flags=world_flags)
        else:
            table = _c5_table(rng, K, B)
            alive, ok = eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table, world_flags=world_flags)
        for t in range(K):
            codes = np.full((B, N), -1, dtype=np.int8) if table is None else table[t]
            reward, done = _oracle_step(env, Ls[t], _resolve_codes(env, codes))
            assert np.array_equal(ok[t][..., None], ~done), f"chunk {ci} step {t}: agent flags"
            if world_flags:
                mx = np.maximum(_k(env.grid[:, 1]).max(axis=(1, 2)), _k(env.grid[:, 2]).max(axis=(1, 2)))
                assert np.array_equal(alive[t], mx > 5), f"chunk {ci} step {t}: world flags"
        r_dev, d_dev = eng.reward_done()
        assert np.array_equal(r_dev, reward) and np.array_equal(d_dev, done), f"chunk {ci}: reward / done"
        _compare_exact(eng, env, Ls[-1], f"chunk {ci} (K={K}, flags={world_flags})")
    return L


# ---------------------------------------------------------------------------------------------
# C3's grid (W = 1024: five overlapped strips) with its greedy agent, against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world_flags", [False, True])
def test_c3_grid_greedy_agent_exact_vs_oracle(amd, world_flags):
    """2 x 1024 x 1024, one Greedy(epsilon=0) agent per world, 12 steps through dw_run_episode (chunks of
    5 and 7: step pairs with binary16 planes in between + closing single steps), exact mode: bit-identical
    to OracleDaisyWorldC + OracleGreedy after every chunk."""
    B, G, N = 2, 1024, 1
    eng = _engine(amd, B, G, G, N, "exact")
    assert "step_stream" in eng.kernel_info()
    eng.init_random(42)
    L, dL = 0.95, 0.75 / 512
    env = _oracle_like(eng, G, L)
    zeros = np.zeros((B, N, 1), dtype=np.int64)
    eng.step(L, zeros)                                      # first step from the un-quantised state
    _oracle_step(env, L, zeros)
    L += dL
    _compare_exact(eng, env, L - dL, "first step")
    _run_chunks_exact(amd, eng, env, L, dL, [5, 7], "greedy", None, world_flags)
    eng.close()


# measured by tools/fast_tolerance_agents.py (profiles/r03_fast_tolerance_agents.json; states developed for 40 / 200 /
# 360 steps, 6-step chunks): largest plane deviation 3 quanta, at most 0.092 % (c3) / 0.087 % (c5) of the cell values
# differ, agent positions equal, agent states equal.  Asserted at (measured x 2); agent states within 2 quanta.
FAST_AGENTS_K, FAST_AGENTS_MAX_QUANTA, FAST_AGENTS_MAX_DIFFERING = 6, 6, 1.9e-3


def _fast_chunk_vs_oracle(amd, B, G, N, seed, develop, codes_of):
    """From a developed, quantised state the engine (float32-only mode) and the oracle run the same 6-step chunk with
    the ORACLE's actions (explicit table: the float32 planes are not bit-identical, so a device policy might choose
    differently)."""
    from therldaisyworld_amd import _ffi
    K = FAST_AGENTS_K
    eng = _engine(amd, B, G, G, N, "fast")
    eng.init_random(seed)
    dL = 0.75 / 512
    L = eng.step_n(develop, 0.75, dL, 0.75, 1.5)
    env = _oracle_like(eng, G, L)
    Ls = [L + i * dL for i in range(K)]
    codes = codes_of(K, B)
    table = np.zeros((K, B, N), dtype=np.int8)
    for t in range(K):
        a = _resolve_codes(env, codes[t])
        table[t] = a[..., 0]
        _oracle_step(env, Ls[t], a)
    eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table, world_flags=False)
    gl, gd = eng.download_planes()
    dl, dd = np.abs(_k(gl) - _k(env.grid[:, 1])), np.abs(_k(gd) - _k(env.grid[:, 2]))
    assert max(dl.max(), dd.max()) <= FAST_AGENTS_MAX_QUANTA
    assert (np.count_nonzero(dl) + np.count_nonzero(dd)) / (2.0 * dl.size) <= FAST_AGENTS_MAX_DIFFERING
    idx, st = eng.download_agents()
    assert np.array_equal(idx, env.agent_indices)
    assert np.abs(st[..., None] - env.agent_states).max() <= 2e-3 + 1e-12
    eng.close()


@pytest.mark.parametrize("develop", [40, 200])
def test_c3_grid_agents_fast_vs_oracle_tolerance(amd, develop):
    """C3's grid (2 x 1024^2, one greedy agent per world: the four-wave ring) in the float32-only mode."""
    _fast_chunk_vs_oracle(amd, 2, 1024, 1, 42, develop, lambda K, B: np.full((K, B, 1), -1, dtype=np.int8))


@pytest.mark.parametrize("develop", [40, 200])
def test_c5_agent_mix_fast_vs_oracle_tolerance(amd, develop):
    """C5's policy mix (1 x 2048^2, 16 agents: greedy / anti-greedy / random / half-random by agent index, overlapped
    strips) in the float32-only mode - round 2 compared this workload's `fast` runs only with themselves."""
    rng = np.random.RandomState(7)
    _fast_chunk_vs_oracle(amd, 1, 2048, 16, 7, develop, lambda K, B: _c5_table(rng, K, B))


# ---------------------------------------------------------------------------------------------
# C5's agents (16 per world, mixed policies by agent index) on a 2048-wide world, against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world_flags", [False, True])
def test_c5_agent_mix_exact_vs_oracle(amd, world_flags):
    """1 x 2048 x 2048, 16 agents: 0-3 greedy, 4-7 anti-greedy, 8-11 random, 12-15 half-random, from an
    explicit code table (-1 / -2 evaluated on the device) - 11 steps in chunks of 4 and 7, exact mode."""
    B, G, N = 1, 2048, 16
    eng = _engine(amd, B, G, G, N, "exact")
    eng.init_random(7)
    L, dL = 1.0, 0.75 / 512
    env = _oracle_like(eng, G, L)
    zeros = np.zeros((B, N, 1), dtype=np.int64)
    eng.step(L, zeros)
    _oracle_step(env, L, zeros)
    L += dL
    _run_chunks_exact(amd, eng, env, L, dL, [4, 7], "c5", np.random.RandomState(5), world_flags)
    eng.close()


def test_c5_agents_crowded_exact_vs_oracle(amd):
    """The same mix with all 16 agents started inside one 3x3 block of a 1 x 1040 x 1040 world (W % 256 != 0,
    the block straddles the boundary of two column strips): agents meet on cells, eat each other's targets
    and their patched 3x3 blocks overlap."""
    B, G, N = 1, 1040, 16
    eng = _engine(amd, B, G, G, N, "exact")
    eng.init_random(3)
    L, dL = 1.0, 0.004
    L = eng.step_n(30, L, dL, 0.75, 1.5)                    # something to graze
    rng = np.random.RandomState(11)
    idx = np.stack([100 + rng.randint(3, size=(B, N)), 246 + rng.randint(3, size=(B, N))], axis=-1)
    eng.upload_agents(idx, np.ones((B, N)))
    env = _oracle_like(eng, G, L)
    _run_chunks_exact(amd, eng, env, L, dL, [9, 4], "c5", rng, True)
    eng.close()


# ---------------------------------------------------------------------------------------------
# full-size runs
# ---------------------------------------------------------------------------------------------
def _episode_outputs(amd, B, G, N, precision, seed, K, table_rng, policy, fuse, monkeypatch):
    from therldaisyworld_amd import _ffi
    if fuse:
        monkeypatch.delenv("DW_NO_AGENT_FUSE", raising=False)
    else:
        monkeypatch.setenv("DW_NO_AGENT_FUSE", "1")
    eng = _engine(amd, B, G, G, N, precision)
    eng.init_random(seed)
    L, dL = 1.0, 0.75 / 512
    eng.step(L, np.zeros((B, N, 1), dtype=np.int64))
    Ls = [L + (i + 1) * dL for i in range(K)]
    if policy == "greedy":
        _, ok = eng.run_episode(Ls, _ffi.POLICY_ARGMAX, world_flags=False)
    else:
        _, ok = eng.run_episode(Ls, _ffi.POLICY_TABLE, None, _c5_table(table_rng, K, B), world_flags=False)
    stats = eng.reduce()
    gl, gd = eng.download_planes()
    kl, kd = np.rint(gl * 1000.0).astype(np.uint16), np.rint(gd * 1000.0).astype(np.uint16)
    del gl, gd
    out = (ok, kl, kd, *eng.download_agents(), stats)
    eng.close()
    return out


def _check_pairs_vs_singles(a, b):
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    for f in ("max_k", "sum_light_k", "sum_dark_k"):
        assert np.array_equal(a[5][f], b[5][f])
    # the fused reductions are the checksums of the planes
    kl, kd, s = a[1], a[2], a[5]
    assert np.array_equal(s["sum_light_k"], kl.sum(axis=(1, 2), dtype=np.uint64))
    assert np.array_equal(s["sum_dark_k"], kd.sum(axis=(1, 2), dtype=np.uint64))
    assert np.array_equal(s["max_k"], np.maximum(kl.max(axis=(1, 2)), kd.max(axis=(1, 2))))


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_c3_full_size_step_pairs_equal_single_steps(amd, monkeypatch, precision):
    """C3 at its full size (256 worlds x 1024^2, one greedy agent each): 7 steps as step pairs with the
    agents' step patched in == one launch per step, and the reductions are the planes' checksums."""
    outs = [_episode_outputs(amd, 256, 1024, 1, precision, 42, 7, None, "greedy", fuse, monkeypatch) for fuse in (True, False)]
    _check_pairs_vs_singles(*outs)


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_c5_full_size_step_pairs_equal_single_steps(amd, monkeypatch, precision):
    """C5's per-GPU shard at its full size (8 worlds x 8192^2, 16 mixed-policy agents each): the same property."""
    outs = [_episode_outputs(amd, 8, 8192, 16, precision, 42, 7, np.random.RandomState(2), "c5", fuse, monkeypatch)
            for fuse in (True, False)]
    _check_pairs_vs_singles(*outs)


def test_c5_full_size_world_vs_oracle(amd):
    """One full-size C5 world (8192 x 8192, 16 mixed-policy agents: 34 overlapped strips per row band) for 5
    steps in exact mode against the oracle (C physics shared out by rows over the host's cores)."""
    B, G, N = 1, 8192, 16
    eng = _engine(amd, B, G, G, N, "exact")
    eng.init_random(9)
    L, dL = 1.05, 0.75 / 512
    env = _oracle_like(eng, G, L)
    zeros = np.zeros((B, N, 1), dtype=np.int64)
    eng.step(L, zeros)
    _oracle_step(env, L, zeros)
    L += dL
    _run_chunks_exact(amd, eng, env, L, dL, [5], "c5", np.random.RandomState(13), False)
    eng.close()


# ---------------------------------------------------------------------------------------------
# C4's per-GPU shard at its full size: 1000 worlds x 256x256, 4 greedy agents, per-step biosphere flags
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_c4_full_size_device_loop_equals_per_step_loop(amd, monkeypatch, precision):
    """BASELINE configs[3]'s shard as the lifespan sweep runs it (dw_run_episode with world flags: step pairs in one
    fused launch that also reduces what both steps' flags need, the agents' in-between step patched in) against one
    launch per step, late in the ramp where biospheres die: per-step world flags and agent flags, planes, agents and
    reductions identical; the reductions are the planes' checksums; some, not all, worlds die inside the window."""
    from therldaisyworld_amd import _ffi
    B, G, N, K = 1000, 256, 4, 24
    outs = []
    for fuse in (True, False):
        if fuse:
            monkeypatch.delenv("DW_NO_AGENT_FUSE", raising=False)
        else:
            monkeypatch.setenv("DW_NO_AGENT_FUSE", "1")
        eng = _engine(amd, B, G, G, N, precision)
        eng.init_random(13)
        dL = 0.75 / 512
        L = eng.step_n(452, 0.75, dL, 0.75, 1.5)             # no grazing up to here: the worlds are about to die of heat
        Ls = [min(L + i * dL, 1.5) for i in range(K)]
        alive, ok = eng.run_episode(Ls, _ffi.POLICY_ARGMAX, threshold_k=5, world_flags=True)
        stats = eng.reduce()
        gl, gd = eng.download_planes()
        kl, kd = np.rint(gl * 1000.0).astype(np.uint16), np.rint(gd * 1000.0).astype(np.uint16)
        del gl, gd
        outs.append((ok, kl, kd, *eng.download_agents(), stats, alive))
        eng.close()
    _check_pairs_vs_singles(outs[0][:6], outs[1][:6])
    assert np.array_equal(outs[0][6], outs[1][6])               # (K, B) biosphere flags of every step
    alive = outs[0][6]
    assert alive[0].any() and not alive[-1].all(), (alive[0].sum(), alive[-1].sum())
    # the last step's flags are the predicate of the lifespan harness on the final reductions
    assert np.array_equal(alive[-1], outs[0][5]["max_k"] > 5)
