#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE implementation.

Run only in the build container (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``/root/reference/daisy`` (never copied into this repo), drives
``RLDaisyWorld`` / ``Greedy`` / ``ft_convolve`` on seeded inputs and stores *data only*
(inputs and outputs) as small ``.npz`` files.  The fixture ids follow SURVEY.md §8(c):

  G1 forward()      G2 C1 trajectory      G3 scripted agents     G4 Greedy policy
  G5 lifespan sweep G6 ft_convolve pin    G7 no-agent path       G8 collision_mode=1
  G9 constructor / reset RNG order      G10 MLP policy (seeded Glorot)   G11 the shipped trained MLP
  G12 triangle luminosity ramp (ramp_up_down)      G13 save_config / restore_config round trip
  G14 attribute mutations (microclimate off, dt, agent_gamma, q2, temp_optimal, albedos)
  G15 direct method calls (get_obs(idx), update_agents, forward(grid), grid assignment, in-place edits)
"""
import os
import sys
import warnings

import numpy as np

REF = os.environ.get("DAISY_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.simplefilter("ignore", DeprecationWarning)

from daisy.daisy_world_rl import RLDaisyWorld  # noqa: E402
from daisy.agents.greedy import Greedy  # noqa: E402
from daisy.nn.functional import ft_convolve, make_neighborhood  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={sorted(arrays)}")


def k1000(x):
    """light/dark planes after np.round(.,3) are exactly k/1000: store k as uint16."""
    k = np.rint(np.asarray(x) * 1000.0)
    assert np.array_equal(k / 1000.0, x), "state is not an exact multiple of 1e-3"
    return k.astype(np.uint16)


# ------------------------------------------------------------------------------------------
def g1_forward():
    np.random.seed(101)
    env = RLDaisyWorld(grid_dimension=16)
    env.batch_size = 2
    env.n_agents = 2
    env.reset()
    out = {}
    cases = [("a", 0.75, 0), ("b", 1.1, 5), ("c", 1.45, 50)]
    for tag, L, presteps in cases:
        for _ in range(presteps):
            env.step(np.random.randint(9, size=(2, 2, 1)))
        env.L = L
        grid_in = env.grid.copy()
        new = env.forward(env.grid.copy())
        out[f"{tag}_L"] = np.float64(L)
        out[f"{tag}_grid_in"] = grid_in
        out[f"{tag}_agent_indices"] = env.agent_indices.copy()
        out[f"{tag}_agent_states"] = env.agent_states.copy()
        out[f"{tag}_grid_out"] = new.copy()
        out[f"{tag}_temp"] = env.temp.copy()
        out[f"{tag}_temp_light"] = env.temp_light.copy()
        out[f"{tag}_temp_dark"] = env.temp_dark.copy()
        out[f"{tag}_temp_effective"] = env.temp_effective.copy()
        out[f"{tag}_dead_temp"] = env.dead_temp.copy()
        out[f"{tag}_beta"] = env.beta.copy()
        out[f"{tag}_beta_l"] = env.beta_l.copy()
        out[f"{tag}_beta_d"] = env.beta_d.copy()
        out[f"{tag}_growth"] = env.growth.copy()
        env.grid = new
    save("G1_forward", **out)


def g2_c1_trajectory():
    """BASELINE config 1: seed 42, B=1, 64x64, no agents, 500 steps."""
    np.random.seed(42)
    env = RLDaisyWorld(grid_dimension=64, n_agents=0)
    env.batch_size = 1
    env.reset()
    steps = 500
    snaps_at = [1, 10, 100, 400, 457, 500]
    out = {"light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "snap_steps": np.array(snaps_at)}
    mean_l, mean_d, mean_T, Ls, dead = [], [], [], [], []
    rewards, dones = [], []
    for t in range(1, steps + 1):
        L_used = env.L
        obs, reward, done, _ = env.step()
        mean_l.append(env.grid[:, 1].mean())
        mean_d.append(env.grid[:, 2].mean())
        mean_T.append(env.temp.mean())
        Ls.append(L_used)
        dead.append(env.dead_temp[0])
        rewards.append(reward.copy())
        dones.append(done.copy())
        if t in snaps_at:
            out[f"light_k_{t}"] = k1000(env.grid[:, 1])
            out[f"dark_k_{t}"] = k1000(env.grid[:, 2])
            out[f"temp3_{t}"] = env.grid[:, 3:6].copy()
            out[f"bare_{t}"] = env.grid[:, 0].copy()
    out.update(mean_light=np.array(mean_l), mean_dark=np.array(mean_d), mean_temp=np.array(mean_T),
               L_used=np.array(Ls), dead_temp=np.array(dead), reward=np.array(rewards),
               done=np.array(dones), final_L=np.float64(env.L), obs_shape=np.array(obs.shape))
    save("G2_c1_trajectory", **out)


def g3_agents():
    """B=4, 8x8, N=4; scripted + random actions; same-cell arrivals; a dying agent;
    a sub-shaped action (1,1,1)."""
    np.random.seed(7)
    env = RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.batch_size = 4
    env.reset()
    # world 0: agents 0 and 1 will both arrive on (2,3) with a grazing action; world 1: agents
    # 2,3 start on the same cell and stay+graze (action 8); world 2: agent 1 is nearly starved.
    env.agent_indices[0] = np.array([[2, 4], [1, 3], [6, 6], [0, 0]])
    env.agent_indices[1] = np.array([[5, 5], [7, 0], [3, 3], [3, 3]])
    env.agent_states[2, 1, 0] = 0.04
    env.agent_states[0, :, 0] = 0.3
    env.L = 1.0   # warm enough that daisies are growing
    out = {"light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "agent_indices0": env.agent_indices.copy(), "agent_states0": env.agent_states.copy(),
           "L0": np.float64(env.L), "dL": np.float64(env.dL)}
    scripted = [
        np.array([[[4], [7], [0], [1]], [[2], [3], [8], [8]], [[5], [5], [6], [7]], [[8], [4], [0], [5]]]),
        np.array([[[5], [6], [7], [8]], [[1], [2], [3], [4]], [[0], [1], [2], [3]], [[4], [5], [6], [7]]]),
    ]
    rng = np.random.RandomState(99)
    actions = list(scripted) + [rng.randint(9, size=(4, 4, 1)) for _ in range(30)]
    actions.insert(5, np.array([[[6]]]))          # sub-shaped (1,1,1) action, ref tests :23
    actions.insert(9, None)                       # default action path (ref :477-478)
    actions.insert(12, rng.randint(9, size=(4, 4, 1)).astype(np.float64))  # float actions
    rec = {k: [] for k in ("agent_indices", "agent_states", "obs", "reward", "done", "light",
                           "dark", "L_after")}
    for t, a in enumerate(actions):
        obs, reward, done, _ = env.step(a)
        rec["agent_indices"].append(env.agent_indices.copy())
        rec["agent_states"].append(env.agent_states.copy())
        rec["obs"].append(obs.copy())
        rec["reward"].append(reward.copy())
        rec["done"].append(done.copy())
        rec["light"].append(env.grid[:, 1].copy())
        rec["dark"].append(env.grid[:, 2].copy())
        rec["L_after"].append(env.L)
        if a is None:
            out[f"action_{t}_none"] = np.array(1)
        else:
            out[f"action_{t}"] = a
    out["n_steps"] = np.array(len(actions))
    out["grid_final"] = env.grid.copy()
    for k, v in rec.items():
        out[k] = np.array(v)
    save("G3_agents", **out)


def g4_greedy():
    rng = np.random.RandomState(5)
    obs = rng.rand(6, 3, 7, 3, 3)
    obs[0, 0, 1:3] = 0.0                        # all-zero tie -> first candidate
    obs[1, 1, 1:3] = 0.25                       # all-equal tie
    obs[2, 2, 1] = np.array([[0, .1, 0], [.3, 0, .3], [0, .1, 0]])   # tie between idx 3 and 5
    obs[2, 2, 2] = 0.0
    obs = obs * make_neighborhood(1, "von_neumann")
    out = {"obs": obs}
    out["greedy"] = Greedy()(obs)
    out["antigreedy"] = Greedy(greedy=False)(obs)
    for eps in (0.0, 0.5, 1.0):
        np.random.seed(1234)
        agent = Greedy(epsilon=eps)
        out[f"eps_{eps}_seq"] = np.array([agent(obs) for _ in range(12)])
    save("G4_greedy", **out)


def simulate_lifespan(env, agent):
    """The notebook's harness (greedy_longevity_abatement.ipynb cell 2:28-57), driven on the
    reference objects."""
    obs = env.reset()
    done_at = np.zeros((*obs.shape[:1],), dtype=int)
    agents_done_at = np.zeros((*obs.shape[:2], 1), dtype=int)
    while True:
        action = agent(obs) if agent is not None else None
        obs, reward, done, info = env.step(action)
        grid_done = env.grid[:, 1:3].max(axis=(1, 2, 3)) <= 0.005
        done_at += (1 - 1 * grid_done)
        agents_done_at += (1 - 1 * done)
        if grid_done.mean() == 1.0:
            break
    return done_at, agents_done_at


def g5_lifespans(B=100, seed=13):
    """dim=8, N=4, seed 13: 5 policies x 2 albedo settings.  Protocol: seed -> ctor ->
    batch_size=B -> (albedo override) -> reset() x1 -> simulate_lifespan (resets again)."""
    out = {"B": np.array(B), "seed": np.array(seed)}
    for agent_status in ["greedy", "antigreedy", "random", "half_random", "no"]:
        for daisy_status in ["light_and_dark", "neutral_albedo"]:
            np.random.seed(seed)
            env = RLDaisyWorld(grid_dimension=8)
            env.batch_size = B
            if daisy_status == "neutral_albedo":
                env.albedo_dark = env.albedo_bare
                env.albedo_light = env.albedo_bare
            agent = {"greedy": lambda: Greedy(epsilon=0.0),
                     "antigreedy": lambda: Greedy(epsilon=0.0, greedy=False),
                     "random": lambda: Greedy(epsilon=1.0),
                     "half_random": lambda: Greedy(epsilon=0.5),
                     "no": lambda: None}[agent_status]()
            env.reset()
            done_at, agents_done_at = simulate_lifespan(env, agent)
            key = f"{agent_status}_{daisy_status}"
            out[key + "_done_at"] = done_at
            out[key + "_agents_done_at"] = agents_done_at
            print(f"  {key}: biosphere {done_at.mean():.3f}  agents {agents_done_at.mean():.3f}")
    save("G5_lifespans", **out)


def g6_ft_convolve():
    rng = np.random.RandomState(3)
    k = rng.rand(1, 1, 3, 3)   # asymmetric: pins orientation
    out = {"kernel": k}
    for i, (h, w) in enumerate([(5, 5), (7, 7), (8, 8), (16, 16), (17, 16), (9, 12), (64, 64)]):
        x = rng.rand(2, 1, h, w)
        out[f"x_{i}"] = x
        out[f"y_{i}"] = ft_convolve(x, k)
    out["n"] = np.array(7)
    for mode in ("moore", "von_neumann", "circular", "asdf"):
        for r in (1, 2, 3, 4):
            out[f"nbhd_{mode}_{r}"] = make_neighborhood(radius=r, mode=mode)
    save("G6_ft_convolve", **out)


def g7_no_agents():
    np.random.seed(21)
    env = RLDaisyWorld(grid_dimension=12, n_agents=0)
    env.batch_size = 3
    obs0 = env.reset()
    out = {"light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "obs0_shape": np.array(obs0.shape), "grid0": env.grid.copy()}
    rs, ds, gs = [], [], []
    for t in range(6):
        obs, reward, done, info = env.step()
        rs.append(reward.copy()); ds.append(done.copy()); gs.append(env.grid.copy())
    out.update(reward=np.array(rs), done=np.array(ds), grids=np.array(gs),
               obs_shape=np.array(obs.shape), reward_dtype=np.array(str(reward.dtype)),
               L_final=np.float64(env.L), step_count=np.array(env.step_count))
    # a dead world: all daisies removed -> reward False, done True
    env.grid[1, 1:3] = 0.0
    obs, reward, done, info = env.step()
    out["dead_reward"] = reward.copy()
    out["dead_done"] = done.copy()
    save("G7_no_agents", **out)


def g8_collisions():
    np.random.seed(31)
    env = RLDaisyWorld(grid_dimension=5, n_agents=4, collision_mode=1)
    env.batch_size = 3
    env.reset()
    env.agent_indices[0] = np.array([[2, 2], [2, 2], [1, 1], [4, 4]])
    env.agent_indices[1] = np.array([[0, 0], [0, 0], [0, 0], [3, 3]])
    env.agent_states[0, :, 0] = np.array([0.6, 0.4, 0.9, 0.2])
    env.agent_states[1, :, 0] = np.array([0.5, 0.7, 0.3, 0.8])
    out = {"light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "agent_indices0": env.agent_indices.copy(), "agent_states0": env.agent_states.copy()}
    np.random.seed(77)     # pins the jitter draws consumed inside update_agents
    states, idxs = [], []
    for t in range(6):
        a = np.full((3, 4, 1), 8)   # everyone stays and grazes -> collisions persist
        env.step(a)
        states.append(env.agent_states.copy()); idxs.append(env.agent_indices.copy())
    out.update(agent_states=np.array(states), agent_indices=np.array(idxs), jitter_seed=np.array(77),
               grid_final=env.grid.copy())
    save("G8_collisions", **out)


def g9_ctor_rng_order():
    np.random.seed(2024)
    env = RLDaisyWorld(grid_dimension=16)
    out = {"seed": np.array(2024), "ctor_grid": env.grid.copy(),
           "ctor_agent_indices": env.agent_indices.copy(), "ctor_dL": np.float64(env.dL),
           "ctor_L": np.float64(env.L)}
    env.batch_size = 5
    env.n_agents = 3
    env.albedo_light = 0.7
    env.min_L = 0.8
    env.ramp_period = 100
    obs = env.reset()
    out.update(reset_grid=env.grid.copy(), reset_agent_indices=env.agent_indices.copy(),
               reset_agent_states=env.agent_states.copy(), reset_obs=obs.copy(),
               reset_dL=np.float64(env.dL), reset_L=np.float64(env.L))
    obs, reward, done, _ = env.step(np.random.randint(9, size=(5, 3, 1)))
    out.update(step_obs=obs.copy(), step_reward=reward.copy(), step_done=done.copy(),
               step_grid=env.grid.copy(), step_L=np.float64(env.L))
    cfg = env.make_config()
    out["config_keys"] = np.array(sorted(cfg.keys()))
    out["config_vals"] = np.array([float(cfg[k]) for k in sorted(cfg.keys())])
    save("G9_ctor_rng_order", **out)


def g10_mlp():
    """MLP policy (daisy/agents/mlp.py): seeded Glorot parameters, actions on real observations, and a
    get_fitness-style rollout (first half of the agents driven by one net, second half by another;
    daisy/evo/sges.py:144-181)."""
    from daisy.agents.mlp import MLP
    np.random.seed(4242)
    agent, adversary = MLP(), MLP()
    pa, pb = agent.get_parameters(), adversary.get_parameters()
    env = RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.batch_size = 6
    obs = env.reset()
    out = {"params_agent": pa, "params_adversary": pb, "obs0": obs.copy(), "action0": agent(obs),
           "light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "agent_indices0": env.agent_indices.copy()}
    half = obs.shape[1] // 2
    acts, rewards, dones, obss = [], [], [], []
    sum_reward = 0.0
    for t in range(40):
        a = np.append(agent.get_action(obs[:, :half]), adversary.get_action(obs[:, half:]), axis=1)
        obs, reward, done, _ = env.step(a)
        acts.append(a.copy()); rewards.append(reward.copy()); dones.append(done.copy()); obss.append(obs.copy())
        sum_reward += (reward[:, :half]).mean()
    out.update(actions=np.array(acts), rewards=np.array(rewards), dones=np.array(dones), obs=np.array(obss),
               sum_reward=np.float64(sum_reward), grid_final=env.grid.copy())
    save("G10_mlp", **out)


def g11_trained_mlp():
    """The trained policy the reference ships as a result file
    (results/cmaes_exp_002/cmaes_exp_002_seed11_best_agent_gen127.json: 1808 weights + the 4 config keys of
    MLP.make_config) loaded with MLP.set_parameters (the reference's restore_config raises, mlp.py:41) and
    rolled out on the default 16x16 world.
    Stored: the config (data), actions/rewards/dones per step, population means and the final grid."""
    import json
    from daisy.agents.mlp import MLP
    path = os.path.join(REF, "results", "cmaes_exp_002", "cmaes_exp_002_seed11_best_agent_gen127.json")
    with open(path) as f:
        cfg = json.load(f)
    np.random.seed(11)
    agent = MLP()
    agent.set_parameters(np.array(cfg["parameters"]))
    env = RLDaisyWorld(grid_dimension=16, n_agents=4)
    env.batch_size = 8
    obs = env.reset()
    out = {"in_dim": np.array(cfg["in_dim"]), "out_dim": np.array(cfg["out_dim"]),
           "h_dim": np.array(cfg["h_dim"]), "act_name": np.array(cfg["act_name"]),
           "parameters": np.array(cfg["parameters"], dtype=np.float64),
           "restored_parameters": agent.get_parameters(),
           "light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "agent_indices0": env.agent_indices.copy(), "obs0": obs.copy()}
    acts, rewards, dones, ml, md = [], [], [], [], []
    for t in range(160):
        a = agent(obs)
        obs, reward, done, _ = env.step(a)
        acts.append(a.astype(np.int8)); rewards.append(reward.copy()); dones.append(done.copy())
        ml.append(env.grid[:, 1].mean(axis=(-2, -1))); md.append(env.grid[:, 2].mean(axis=(-2, -1)))
    out.update(actions=np.array(acts), rewards=np.array(rewards), dones=np.array(dones),
               mean_light=np.array(ml), mean_dark=np.array(md), grid_final=env.grid.copy(),
               agent_indices_final=env.agent_indices.copy(), L_final=np.float64(env.L))
    save("G11_trained_mlp", **out)


def g12_ramp_up_down():
    """update_L with the triangle ramp (daisy_world_rl.py:463-473: ramp_up_down, ramp_period, ddL): the
    luminosity schedule over several ramp periods with 2 agents acting, and what a second reset() does to
    dL / min_L / max_L afterwards."""
    np.random.seed(99)
    env = RLDaisyWorld(grid_dimension=8, n_agents=2)
    env.batch_size = 3
    env.ramp_up_down = True
    env.ramp_period = 12
    env.min_L, env.max_L, env.ddL = 0.9, 1.2, 0.01
    obs = env.reset()
    out = {"light0": env.grid[:, 1].copy(), "dark0": env.grid[:, 2].copy(),
           "agent_indices0": env.agent_indices.copy(), "L0": np.float64(env.L), "dL0": np.float64(env.dL)}
    actions = np.random.randint(9, size=(60, 3, 2, 1))
    Ls, dLs, mins, maxs, counts, rewards = [], [], [], [], [], []
    for t in range(60):
        obs, reward, done, _ = env.step(actions[t])
        Ls.append(env.L); dLs.append(env.dL); mins.append(env.min_L); maxs.append(env.max_L)
        counts.append(env.step_count); rewards.append(reward.copy())
    out.update(actions=actions, L=np.array(Ls), dL=np.array(dLs), min_L=np.array(mins), max_L=np.array(maxs),
               step_count=np.array(counts), rewards=np.array(rewards), grid_final=env.grid.copy(),
               obs_final=obs.copy())
    obs = env.reset()
    out.update(reset2_L=np.float64(env.L), reset2_dL=np.float64(env.dL), reset2_min_L=np.float64(env.min_L),
               reset2_max_L=np.float64(env.max_L), reset2_step_count=np.array(env.step_count),
               reset2_light=env.grid[:, 1].copy(), reset2_agent_indices=env.agent_indices.copy())
    a2 = np.random.randint(9, size=(5, 3, 2, 1))
    L2 = []
    for t in range(5):
        env.step(a2[t])
        L2.append(env.L)
    out.update(actions2=a2, L2=np.array(L2), grid_final2=env.grid.copy())
    save("G12_ramp_up_down", **out)


def g13_config_roundtrip():
    """save_config / restore_config (daisy_world_rl.py:94-171): a modified environment's 20-key config
    as the JSON text the reference writes, restored into a fresh environment which then runs 6 steps."""
    import json
    import tempfile
    np.random.seed(5)
    env = RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.albedo_light, env.albedo_dark = 0.7, 0.3
    env.n_agents, env.agent_gamma, env.gamma = 3, 0.04, 0.27
    env.min_L, env.max_L, env.ramp_period = 0.8, 1.4, 64
    env.reset()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "cfg.json")
        env.save_config(path)
        text = open(path).read()
        np.random.seed(6)
        env2 = RLDaisyWorld(grid_dimension=8, n_agents=4)
        env2.restore_config(path)
    env2.batch_size = 4
    obs = env2.reset()
    actions = np.random.randint(9, size=(6, 4, 3, 1))
    for t in range(6):
        obs, reward, done, _ = env2.step(actions[t])
    cfg2 = env2.make_config()
    keys = sorted(cfg2.keys())
    save("G13_config_roundtrip", config_json=np.array(text), config_keys=np.array(keys),
         config_after=np.array([float(cfg2[k]) for k in keys]), actions=actions, obs_final=obs.copy(),
         reward_final=reward.copy(), grid_final=env2.grid.copy(), L_final=np.float64(env2.L),
         dL_final=np.float64(env2.dL))


from g14_variants import G14_VARIANTS  # noqa: E402  (pure data, shared with the tests)


def g14_attribute_mutations():
    """The constants callers change on a constructed environment (notebooks: set_use_microclimate, dt,
    agent_gamma, q2, temp_optimal, albedos ...) followed by reset() and 14 steps with 2 agents."""
    out = {}
    for name, spec in G14_VARIANTS.items():
        np.random.seed(314)
        env = RLDaisyWorld(grid_dimension=16, n_agents=2)
        env.batch_size = 2
        if "call" in spec:
            getattr(env, spec["call"])(*spec["args"])
        for k, v in spec.get("attrs", {}).items():
            setattr(env, k, v)
        env.min_L, env.max_L, env.ramp_period = 0.8, 1.45, 14
        obs = env.reset()
        actions = np.random.randint(9, size=(14, 2, 2, 1))
        rewards = []
        for t in range(14):
            obs, reward, done, _ = env.step(actions[t])
            rewards.append(reward.copy())
        out.update({f"{name}_actions": actions, f"{name}_rewards": np.array(rewards), f"{name}_obs": obs.copy(),
                    f"{name}_grid": env.grid.copy(), f"{name}_q2": np.float64(env.q2),
                    f"{name}_temp": env.temp.copy(), f"{name}_growth": env.growth.copy()})
    save("G14_attribute_mutations", **out)


def g15_direct_method_calls():
    """The methods and attribute edits callers use besides reset()/step(): get_obs on caller-supplied
    positions, update_agents alone, forward(grid) as a function, grid assignment, in-place edits of
    env.grid and env.agent_states between steps."""
    np.random.seed(77)
    env = RLDaisyWorld(grid_dimension=8, n_agents=3)
    env.batch_size = 2
    obs = env.reset()
    out = {"obs0": obs.copy()}
    idx2 = np.random.randint(8, size=(2, 3, 2))
    out.update(idx2=idx2, obs_at_idx2=env.get_obs(idx2).copy())
    a = np.random.randint(5, 9, size=(2, 3, 1))                       # grazing moves
    env.update_agents(a)
    out.update(a=a, after_update_indices=env.agent_indices.copy(), after_update_states=env.agent_states.copy(),
               after_update_grid=env.grid.copy())
    new = env.forward(env.grid)
    out.update(forward_new=new.copy(), grid_after_forward_call=env.grid.copy())
    env.grid = new
    obs3, r3, d3, _ = env.step(a)
    out.update(obs3=obs3.copy(), r3=r3.copy(), d3=d3.copy(), grid3=env.grid.copy())
    env.grid[:, 1] *= 0.5                                              # in-place edits between steps
    env.agent_states[0, 0, 0] = 0.05
    env.agent_indices[1, 2] = [3, 4]
    obs4, r4, d4, _ = env.step(None)
    out.update(obs4=obs4.copy(), r4=r4.copy(), d4=d4.copy(), grid4=env.grid.copy(),
               indices4=env.agent_indices.copy(), states4=env.agent_states.copy(), L4=np.float64(env.L))
    save("G15_direct_method_calls", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14",
                             "g15"]
    fns = {"g1": g1_forward, "g2": g2_c1_trajectory, "g3": g3_agents, "g4": g4_greedy,
           "g5": g5_lifespans, "g6": g6_ft_convolve, "g7": g7_no_agents, "g8": g8_collisions,
           "g9": g9_ctor_rng_order, "g10": g10_mlp, "g11": g11_trained_mlp,
           "g12": g12_ramp_up_down, "g13": g13_config_roundtrip,
           "g14": g14_attribute_mutations, "g15": g15_direct_method_calls}
    for w in which:
        fns[w]()
