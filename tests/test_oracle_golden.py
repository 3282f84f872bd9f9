"""The CPU oracle (oracle/daisy_oracle.py) against the golden vectors generated from the reference
(tests/golden/make_golden.py).  This is what pins the oracle: SURVEY.md §8(c) G1-G9."""
import numpy as np
import pytest

from oracle import daisy_oracle as O


def _assert_initial_grid(got, ref):
    """The initial grid is NOT quantised (ref :304-324): covers are exact, the three temperature
    channels carry the reference's ~1e-13 FFT noise."""
    assert np.array_equal(got[:, :3], ref[:, :3])
    np.testing.assert_allclose(got[:, 3:6], ref[:, 3:6], rtol=1e-12, atol=0)
    assert np.array_equal(got[:, 6], ref[:, 6])


def _env(dim, n_agents, batch, **kw):
    return O.OracleDaisyWorld(grid_dimension=dim, n_agents=n_agents, batch_size=batch, **kw)


def test_g6_direct_stencil_equals_ft_convolve(golden):
    g = golden("G6_ft_convolve")
    k = g["kernel"]
    for i in range(int(g["n"])):
        y = O.toroidal_conv3x3(g[f"x_{i}"], k[0, 0])
        np.testing.assert_allclose(y, g[f"y_{i}"], rtol=0, atol=5e-15)


def test_g6_neighborhood_masks(golden):
    g = golden("G6_ft_convolve")
    for mode in ("moore", "von_neumann", "circular", "asdf"):
        for r in (1, 2, 3, 4):
            assert np.array_equal(O.neighborhood_mask(r, mode), g[f"nbhd_{mode}_{r}"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g1_forward(golden, tag):
    g = golden("G1_forward")
    env = _env(16, 2, 2)
    env.L = float(g[f"{tag}_L"])
    env.agent_indices = g[f"{tag}_agent_indices"].copy()
    env.agent_states = g[f"{tag}_agent_states"].copy()
    out = env.forward(g[f"{tag}_grid_in"].copy())
    ref = g[f"{tag}_grid_out"]
    # light/dark/bare: bit-exact after the 3-decimal quantiser
    assert np.array_equal(out[:, :3], ref[:, :3])
    # rounded temperature channels: identical up to one quantum on a measure-zero set; exact here
    assert np.array_equal(out[:, 3:], ref[:, 3:])
    for name in ("temp", "temp_light", "temp_dark", "temp_effective", "beta", "beta_l", "beta_d",
                 "growth", "dead_temp"):
        np.testing.assert_allclose(getattr(env, name), g[f"{tag}_{name}"], rtol=1e-12, atol=1e-12)


def test_g2_c1_trajectory(golden):
    g = golden("G2_c1_trajectory")
    env = _env(64, 0, 1)
    env.set_initial_cover(g["light0"], g["dark0"])
    env.agent_indices = np.zeros((1, 0, 2), dtype=np.int64)
    env.agent_states = np.ones((1, 0, 1))
    snaps = set(int(s) for s in g["snap_steps"])
    for t in range(1, 501):
        assert env.L == g["L_used"][t - 1]
        obs, reward, done, _ = env.step()
        assert env.grid[:, 1].mean() == g["mean_light"][t - 1]
        assert env.grid[:, 2].mean() == g["mean_dark"][t - 1]
        assert abs(env.temp.mean() - g["mean_temp"][t - 1]) < 1e-9
        assert env.dead_temp[0] == pytest.approx(g["dead_temp"][t - 1], rel=1e-14)
        assert np.array_equal(reward, g["reward"][t - 1])
        assert np.array_equal(done, g["done"][t - 1])
        if t in snaps:
            assert np.array_equal(np.rint(env.grid[:, 1] * 1000).astype(np.uint16), g[f"light_k_{t}"])
            assert np.array_equal(np.rint(env.grid[:, 2] * 1000).astype(np.uint16), g[f"dark_k_{t}"])
            assert np.array_equal(env.grid[:, 3:6], g[f"temp3_{t}"])
            assert np.array_equal(env.grid[:, 0], g[f"bare_{t}"])
    assert env.L == float(g["final_L"])
    assert tuple(obs.shape) == tuple(g["obs_shape"])


def _g3_actions(g):
    acts = []
    for t in range(int(g["n_steps"])):
        acts.append(None if f"action_{t}_none" in g.files else g[f"action_{t}"])
    return acts


def test_g3_agents(golden):
    g = golden("G3_agents")
    env = _env(8, 4, 4)
    env.L = float(g["L0"])
    env.set_initial_cover(g["light0"], g["dark0"])
    env.L = float(g["L0"])
    env.agent_indices = g["agent_indices0"].copy()
    env.agent_states = g["agent_states0"].copy()
    for t, a in enumerate(_g3_actions(g)):
        obs, reward, done, _ = env.step(a)
        assert np.array_equal(env.agent_indices, g["agent_indices"][t]), t
        assert np.array_equal(env.agent_states, g["agent_states"][t]), t
        assert np.array_equal(env.grid[:, 1], g["light"][t]), t
        assert np.array_equal(env.grid[:, 2], g["dark"][t]), t
        assert np.array_equal(obs, g["obs"][t]), t
        assert np.array_equal(reward, g["reward"][t]), t
        assert np.array_equal(done, g["done"][t]), t
        assert env.L == g["L_after"][t]
    assert np.array_equal(env.grid, g["grid_final"])


def test_g4_greedy(golden):
    g = golden("G4_greedy")
    obs = g["obs"]
    assert np.array_equal(O.OracleGreedy()(obs), g["greedy"])
    assert np.array_equal(O.OracleGreedy(greedy=False)(obs), g["antigreedy"])
    for eps in (0.0, 0.5, 1.0):
        np.random.seed(1234)
        agent = O.OracleGreedy(epsilon=eps)
        seq = np.array([agent(obs) for _ in range(12)])
        assert np.array_equal(seq, g[f"eps_{eps}_seq"])


@pytest.mark.parametrize("agent_status,daisy_status", [
    ("greedy", "light_and_dark"), ("antigreedy", "light_and_dark"), ("random", "neutral_albedo"),
    ("half_random", "light_and_dark"), ("no", "light_and_dark")])
def test_g5_lifespans(golden, agent_status, daisy_status):
    g = golden("G5_lifespans")
    B, seed = int(g["B"]), int(g["seed"])
    np.random.seed(seed)
    env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=8)
    env.P.batch_size = B
    if daisy_status == "neutral_albedo":
        env.P.albedo_dark = env.P.albedo_light = env.P.albedo_bare
    agent = {"greedy": O.OracleGreedy(0.0), "antigreedy": O.OracleGreedy(0.0, greedy=False),
             "random": O.OracleGreedy(1.0), "half_random": O.OracleGreedy(0.5), "no": None}[agent_status]
    env.reset()
    done_at, agents_done_at = O.simulate_lifespan(env, agent)
    key = f"{agent_status}_{daisy_status}"
    assert np.array_equal(done_at, g[key + "_done_at"])
    assert np.array_equal(agents_done_at, g[key + "_agents_done_at"])


def test_g7_no_agents(golden):
    g = golden("G7_no_agents")
    env = _env(12, 0, 3)
    env.set_initial_cover(g["light0"], g["dark0"])
    _assert_initial_grid(env.grid, g["grid0"])
    env.agent_indices = np.zeros((3, 0, 2), dtype=np.int64)
    env.agent_states = np.ones((3, 0, 1))
    for t in range(6):
        obs, reward, done, _ = env.step()
        assert reward.dtype == np.bool_ and str(g["reward_dtype"]) == "bool"
        assert np.array_equal(reward, g["reward"][t])
        assert np.array_equal(done, g["done"][t])
        assert np.array_equal(env.grid, g["grids"][t])
    assert tuple(obs.shape) == tuple(g["obs_shape"]) == (3, 0, 7, 3, 3)
    assert env.L == float(g["L_final"]) and env.step_count == int(g["step_count"])
    env.grid[1, 1:3] = 0.0
    obs, reward, done, _ = env.step()
    assert np.array_equal(reward, g["dead_reward"]) and np.array_equal(done, g["dead_done"])


def test_g8_collisions(golden):
    g = golden("G8_collisions")
    env = _env(5, 4, 3, collision_mode=1)
    env.set_initial_cover(g["light0"], g["dark0"])
    env.agent_indices = g["agent_indices0"].copy()
    env.agent_states = g["agent_states0"].copy()
    np.random.seed(int(g["jitter_seed"]))
    for t in range(6):
        env.step(np.full((3, 4, 1), 8))
        assert np.array_equal(env.agent_indices, g["agent_indices"][t])
        np.testing.assert_array_equal(env.agent_states, g["agent_states"][t])
    assert np.array_equal(env.grid, g["grid_final"])


def test_g9_ctor_rng_order(golden):
    g = golden("G9_ctor_rng_order")
    np.random.seed(int(g["seed"]))
    env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=16)
    _assert_initial_grid(env.grid, g["ctor_grid"])
    assert np.array_equal(env.agent_indices, g["ctor_agent_indices"])
    assert env.dL == float(g["ctor_dL"]) and env.L == float(g["ctor_L"])
    env.P.batch_size, env.P.n_agents = 5, 3
    env.P.albedo_light, env.P.min_L, env.P.ramp_period = 0.7, 0.8, 100
    obs = env.reset()
    _assert_initial_grid(env.grid, g["reset_grid"])
    assert np.array_equal(env.agent_indices, g["reset_agent_indices"])
    np.testing.assert_allclose(obs, g["reset_obs"], rtol=1e-12, atol=0)
    assert env.dL == float(g["reset_dL"]) and env.L == float(g["reset_L"])
    obs, reward, done, _ = env.step(np.random.randint(9, size=(5, 3, 1)))
    assert np.array_equal(obs, g["step_obs"])
    assert np.array_equal(reward, g["step_reward"]) and np.array_equal(done, g["step_done"])
    assert np.array_equal(env.grid, g["step_grid"])
    assert env.L == float(g["step_L"])


# ---- the C restatement (oracle/daisy_oracle.c) ------------------------------------------------
def test_c_oracle_g1_forward(golden):
    from oracle import c_oracle
    g = golden("G1_forward")
    for tag in "abc":
        gi = g[f"{tag}_grid_in"]
        out, caches = c_oracle.forward(gi[:, 1], gi[:, 2], float(g[f"{tag}_L"]), want_caches=True)
        ref = g[f"{tag}_grid_out"].copy()
        # the fixture has agent states written into channel 4 at agent cells (ref :454-459)
        idx = g[f"{tag}_agent_indices"]
        for bb in range(idx.shape[0]):
            for nn in range(idx.shape[1]):
                out[bb, 4, idx[bb, nn, 0], idx[bb, nn, 1]] = g[f"{tag}_agent_states"][bb, nn, 0]
        assert np.array_equal(out, ref)
        np.testing.assert_allclose(caches[:, 0], g[f"{tag}_temp"][:, 0], rtol=1e-12)
        np.testing.assert_allclose(caches[:, 5:7], g[f"{tag}_growth"], rtol=1e-9, atol=1e-15)


def test_c_oracle_g2_trajectory(golden):
    from oracle import c_oracle
    g = golden("G2_c1_trajectory")
    light, dark = g["light0"].copy(), g["dark0"].copy()
    L, dL, t = 0.75, 0.75 / 512, 0
    for snap in (int(s) for s in g["snap_steps"]):
        L = c_oracle.step_n(light, dark, L, dL, snap - t)
        t = snap
        assert np.array_equal(np.rint(light * 1000).astype(np.uint16), g[f"light_k_{t}"])
        assert np.array_equal(np.rint(dark * 1000).astype(np.uint16), g[f"dark_k_{t}"])
    assert L == float(g["final_L"])


def test_c_oracle_matches_numpy_oracle_multiworld():
    from oracle import c_oracle
    rng = np.random.RandomState(0)
    B, H, W = 5, 9, 12
    light = np.rint(rng.rand(B, H, W) * 400) / 1000
    dark = np.rint(rng.rand(B, H, W) * 400) / 1000
    env = O.OracleDaisyWorld(grid_dimension=H, n_agents=0, batch_size=B)
    env.L = 1.03
    grid = np.zeros((B, 7, H, W))
    grid[:, 1], grid[:, 2] = light, dark
    ref = env.forward(grid)
    out = c_oracle.forward(light, dark, 1.03)
    assert np.array_equal(out, ref)


def test_c_oracle_row_parallel_mode_equals_world_parallel_mode():
    """Few big worlds are shared out by rows instead of by worlds (oracle/daisy_oracle.c, few_big_worlds):
    the same cells by the same expressions."""
    from oracle import c_oracle
    rng = np.random.RandomState(4)
    B, H, W = 2, 512, 1024                              # H*W = 2^19 and B < threads: the row-parallel path
    light = np.rint(rng.rand(B, H, W) * 400) / 1000
    dark = np.rint(rng.rand(B, H, W) * 400) / 1000
    nthreads = c_oracle.max_threads()
    if nthreads < 3:
        pytest.skip("needs > 2 OpenMP threads to take the row-parallel path")
    out_rows, caches_rows = c_oracle.forward(light, dark, 1.07, want_caches=True)
    l1, d1 = light.copy(), dark.copy()
    La = c_oracle.step_n(l1, d1, 1.07, 0.01, 3)
    c_oracle.set_threads(1)                             # B >= threads: the world-parallel path
    try:
        out_w, caches_w = c_oracle.forward(light, dark, 1.07, want_caches=True)
        l2, d2 = light.copy(), dark.copy()
        Lb = c_oracle.step_n(l2, d2, 1.07, 0.01, 3)
    finally:
        c_oracle.set_threads(nthreads)
    assert np.array_equal(out_rows, out_w) and np.array_equal(caches_rows, caches_w)
    assert La == Lb and np.array_equal(l1, l2) and np.array_equal(d1, d2)


def test_c_backed_environment_equals_numpy_environment(golden):
    """OracleDaisyWorldC (physics pass in C, everything else the Python restatement) against the NumPy
    environment on an agent episode, and against the reference fixture G3 itself."""
    rng = np.random.RandomState(9)
    envs = []
    for cls in (O.OracleDaisyWorld, O.OracleDaisyWorldC):
        np.random.seed(21)
        env = cls(grid_dimension=12, n_agents=3, batch_size=4)
        obs = env.reset()
        envs.append((env, obs))
    np.testing.assert_allclose(envs[0][1], envs[1][1], rtol=1e-12, atol=0)   # un-rounded initial temperatures
    for t in range(25):
        a = rng.randint(9, size=(4, 3, 1))
        ra, rb = envs[0][0].step(a.copy()), envs[1][0].step(a.copy())
        for x, y in zip(ra[:3], rb[:3]):
            assert np.array_equal(x, y), t
        assert np.array_equal(envs[0][0].grid, envs[1][0].grid)
        # un-rounded caches: pow() against ** 0.25 and another summation order (a few 1e-16 relative)
        np.testing.assert_allclose(envs[0][0].growth, envs[1][0].growth, rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(envs[0][0].temp, envs[1][0].temp, rtol=1e-12)
    g = golden("G3_agents")
    env = O.OracleDaisyWorldC(grid_dimension=8, n_agents=4, batch_size=4)
    env.L = float(g["L0"])
    env.set_initial_cover(g["light0"], g["dark0"])
    env.L = float(g["L0"])
    env.agent_indices = g["agent_indices0"].copy()
    env.agent_states = g["agent_states0"].copy()
    for t, a in enumerate(_g3_actions(g)):
        obs, reward, done, _ = env.step(a)
        assert np.array_equal(env.agent_indices, g["agent_indices"][t]), t
        assert np.array_equal(env.agent_states, g["agent_states"][t]), t
        assert np.array_equal(obs, g["obs"][t]), t
        assert np.array_equal(reward, g["reward"][t]) and np.array_equal(done, g["done"][t]), t
        assert env.L == g["L_after"][t]
    assert np.array_equal(env.grid, g["grid_final"])


def test_g10_mlp_policy(golden):
    g = golden("G10_mlp")
    agent, adversary = O.OracleMLP(g["params_agent"]), O.OracleMLP(g["params_adversary"])
    assert np.array_equal(agent(g["obs0"]), g["action0"])
    env = _env(8, 4, 6)
    env.set_initial_cover(g["light0"], g["dark0"])
    env.agent_indices = g["agent_indices0"].copy()
    env.agent_states = np.ones((6, 4, 1))
    obs = env.get_obs(env.agent_indices)
    np.testing.assert_allclose(obs, g["obs0"], rtol=1e-12, atol=0)
    half, sum_reward = 2, 0.0
    for t in range(40):
        a = np.append(agent.get_action(obs[:, :half]), adversary.get_action(obs[:, half:]), axis=1)
        assert np.array_equal(a, g["actions"][t]), t
        obs, reward, done, _ = env.step(a)
        assert np.array_equal(obs, g["obs"][t]) and np.array_equal(reward, g["rewards"][t])
        sum_reward += reward[:, :half].mean()
    assert sum_reward == float(g["sum_reward"])
    assert np.array_equal(env.grid, g["grid_final"])


def test_g11_trained_mlp_rollout(golden):
    """The trained policy the reference ships (CMA-ES generation 127), 160 steps on 8 default 16x16 worlds."""
    g = golden("G11_trained_mlp")
    agent = O.OracleMLP(g["parameters"])
    env = _env(16, 4, 8)
    env.set_initial_cover(g["light0"], g["dark0"])
    env.agent_indices = g["agent_indices0"].copy()
    env.agent_states = np.ones((8, 4, 1))
    obs = env.get_obs(env.agent_indices)
    np.testing.assert_allclose(obs, g["obs0"], rtol=1e-12, atol=0)
    for t in range(160):
        a = agent(obs)
        assert np.array_equal(a, g["actions"][t]), t
        obs, reward, done, _ = env.step(a)
        assert np.array_equal(reward, g["rewards"][t]) and np.array_equal(done, g["dones"][t]), t
        assert np.array_equal(env.grid[:, 1].mean(axis=(-2, -1)), g["mean_light"][t]), t
        assert np.array_equal(env.grid[:, 2].mean(axis=(-2, -1)), g["mean_dark"][t]), t
    assert np.array_equal(env.grid, g["grid_final"])
    assert np.array_equal(env.agent_indices, g["agent_indices_final"])
    assert env.L == float(g["L_final"])


def _g12_checks(env, g, get=lambda e, k: getattr(e, k), P=None):
    """Shared by the oracle test here and the drop-in test on the GPU: the G12 protocol on `env`, whose
    ramp parameters live on `P` (oracle: env.P, drop-in: the environment itself)."""
    P = env if P is None else P
    P.ramp_up_down, P.ramp_period = True, 12
    P.min_L, P.max_L, P.ddL = 0.9, 1.2, 0.01
    env.batch_size = 3
    if P is not env:
        P.batch_size = 3
    obs = env.reset()
    assert np.array_equal(env.grid[:, 1], g["light0"]) and np.array_equal(env.agent_indices, g["agent_indices0"])
    assert env.L == float(g["L0"]) and env.dL == float(g["dL0"])
    actions = np.random.randint(9, size=(60, 3, 2, 1))
    assert np.array_equal(actions, g["actions"])
    for t in range(60):
        obs, reward, done, _ = env.step(actions[t])
        assert env.L == g["L"][t] and env.dL == g["dL"][t], t
        assert P.min_L == g["min_L"][t] and P.max_L == g["max_L"][t] and env.step_count == g["step_count"][t], t
        assert np.array_equal(reward, g["rewards"][t]), t
    assert np.array_equal(env.grid, g["grid_final"]) and np.array_equal(obs, g["obs_final"])
    env.reset()
    assert env.L == float(g["reset2_L"]) and env.dL == float(g["reset2_dL"]) and env.step_count == 0
    assert P.min_L == float(g["reset2_min_L"]) and P.max_L == float(g["reset2_max_L"])
    assert np.array_equal(env.grid[:, 1], g["reset2_light"])
    assert np.array_equal(env.agent_indices, g["reset2_agent_indices"])
    a2 = np.random.randint(9, size=(5, 3, 2, 1))
    for t in range(5):
        env.step(a2[t])
        assert env.L == g["L2"][t]
    assert np.array_equal(env.grid, g["grid_final2"])


def test_g12_triangle_ramp(golden):
    """update_L with ramp_up_down (ref :463-473) over five ramp periods, and the second reset()."""
    g = golden("G12_ramp_up_down")
    np.random.seed(99)
    env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=8, n_agents=2)
    _g12_checks(env, g, P=env.P)


def test_harness_luminosity_schedule_matches_reference_ramp_g12(golden):
    """The chunked harnesses precompute the luminosity of the next K steps without touching the
    environment: same schedule as the reference's update_L, triangle ramp included."""
    from types import SimpleNamespace
    from therldaisyworld_amd.harness import _luminosity_schedule
    g = golden("G12_ramp_up_down")
    env = SimpleNamespace(L=float(g["L0"]), dL=float(g["dL0"]), step_count=0, min_L=0.9, max_L=1.2, ddL=0.01,
                          ramp_up_down=True, ramp_period=12)
    Ls = _luminosity_schedule(env, 61)
    assert Ls[0] == float(g["L0"]) and Ls[1:] == [float(v) for v in g["L"]]
    assert (env.L, env.dL, env.step_count, env.min_L, env.max_L) == (float(g["L0"]), float(g["dL0"]), 0, 0.9, 1.2)


def test_g13_config_roundtrip(golden):
    """The oracle environment fed the reference's saved config reproduces the run after restore_config."""
    import json
    g = golden("G13_config_roundtrip")
    cfg = json.loads(str(g["config_json"]))
    assert sorted(cfg.keys()) == [str(k) for k in g["config_keys"]] and len(cfg) == 20
    np.random.seed(6)
    env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=8, n_agents=4)
    for k, v in cfg.items():                      # ref _apply_config :132-152 (dL is overwritten by reset)
        if k not in ("dL", "initial_L"):
            setattr(env.P, k, v)
    env.P.batch_size = 4
    obs = env.reset()
    actions = np.random.randint(9, size=(6, 4, 3, 1))
    assert np.array_equal(actions, g["actions"])
    for t in range(6):
        obs, reward, done, _ = env.step(actions[t])
    assert np.array_equal(obs, g["obs_final"]) and np.array_equal(reward, g["reward_final"])
    assert np.array_equal(env.grid, g["grid_final"])
    assert env.L == float(g["L_final"]) and env.dL == float(g["dL_final"])


def _g14_variants():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "g14_variants", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_variants.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.G14_VARIANTS


def _g14_checks(env, g, name, P=None):
    """G14 protocol on `env` (oracle: parameters on env.P; drop-in: on the environment itself)."""
    P = env if P is None else P
    spec = _g14_variants()[name]
    P.batch_size = 2
    env.batch_size = 2
    if "call" in spec:
        if hasattr(env, spec["call"]):
            getattr(env, spec["call"])(*spec["args"])
        else:                                       # oracle: set_use_microclimate(False) == q2 = 0 (ref :85-92)
            P.q2 = P.q / 8.0 if spec["args"][0] else 0.0
    for k, v in spec.get("attrs", {}).items():
        setattr(P, k, v)
    P.min_L, P.max_L, P.ramp_period = 0.8, 1.45, 14
    obs = env.reset()
    actions = np.random.randint(9, size=(14, 2, 2, 1))
    assert np.array_equal(actions, g[f"{name}_actions"])
    for t in range(14):
        obs, reward, done, _ = env.step(actions[t])
        assert np.array_equal(reward, g[f"{name}_rewards"][t]), t
    assert P.q2 == float(g[f"{name}_q2"])
    assert np.array_equal(obs, g[f"{name}_obs"])
    assert np.array_equal(env.grid, g[f"{name}_grid"])
    np.testing.assert_allclose(env.temp, g[f"{name}_temp"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(env.growth, g[f"{name}_growth"], rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("name", ["no_microclimate", "slow_time", "other_physics", "wide_albedo"])
def test_g14_attribute_mutations(golden, name):
    g = golden("G14_attribute_mutations")
    np.random.seed(314)
    env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=16, n_agents=2)
    _g14_checks(env, g, name, P=env.P)


def _g15_checks(env, g, exact_channels=True):
    """G15 protocol: the calls and edits callers make besides reset()/step()."""
    env.batch_size = 2
    if hasattr(env, "P"):
        env.P.batch_size = 2
    obs = env.reset()
    np.testing.assert_allclose(obs, g["obs0"], rtol=1e-12, atol=0)   # un-rounded initial temperatures: FFT noise
    idx2 = np.random.randint(8, size=(2, 3, 2))
    assert np.array_equal(idx2, g["idx2"])
    np.testing.assert_allclose(env.get_obs(idx2), g["obs_at_idx2"], rtol=1e-12, atol=0)
    a = np.random.randint(5, 9, size=(2, 3, 1))
    env.update_agents(a)
    assert np.array_equal(env.agent_indices, g["after_update_indices"])
    assert np.array_equal(env.agent_states, g["after_update_states"])
    np.testing.assert_allclose(env.grid, g["after_update_grid"], rtol=1e-12, atol=0)
    new = env.forward(env.grid)
    assert np.array_equal(new, g["forward_new"])
    np.testing.assert_allclose(env.grid, g["grid_after_forward_call"], rtol=1e-12, atol=0)
    env.grid = new
    obs3, r3, d3, _ = env.step(a)
    assert np.array_equal(obs3, g["obs3"]) and np.array_equal(r3, g["r3"]) and np.array_equal(d3, g["d3"])
    assert np.array_equal(env.grid, g["grid3"])
    env.grid[:, 1] *= 0.5
    env.agent_states[0, 0, 0] = 0.05
    env.agent_indices[1, 2] = [3, 4]
    obs4, r4, d4, _ = env.step(None)
    assert np.array_equal(obs4, g["obs4"]) and np.array_equal(r4, g["r4"]) and np.array_equal(d4, g["d4"])
    assert np.array_equal(env.grid, g["grid4"])
    assert np.array_equal(env.agent_indices, g["indices4"]) and np.array_equal(env.agent_states, g["states4"])
    assert env.L == float(g["L4"])


def test_g15_direct_method_calls(golden):
    g = golden("G15_direct_method_calls")
    np.random.seed(77)
    env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=8, n_agents=3)
    _g15_checks(env, g)
