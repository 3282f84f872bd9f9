"""CPU-side tests (no GPU needed): the C-ABI library loads and exports every symbol that
include/daisyworld_hip.h declares, fails loudly without a device, and the host-side mirrors
(RNG call order, policy, masks, config) match the reference fixtures."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from therldaisyworld_amd import build
    return build.build_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "daisyworld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dw_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(built):
    from therldaisyworld_amd import _ffi
    names = _declared_symbols()
    assert len(names) >= 30
    lib = C.CDLL(built)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported by the library"
    assert sorted(_ffi.SIGNATURES) == names, "ctypes binding and header disagree"
    assert _ffi.load().dw_abi_version() == _ffi.DW_ABI_VERSION


def test_struct_layouts_match_header(built):
    from therldaisyworld_amd import _ffi
    assert C.sizeof(_ffi.DwWorldStats) == 24 == _ffi.STATS_DTYPE.itemsize
    assert C.sizeof(_ffi.DwParams) == 10 * 4 + 8 + 18 * 8
    p = _ffi.DwParams()
    assert _ffi.load().dw_default_params(C.byref(p), 7, 8, 12, 3) == 0
    assert (p.batch, p.height, p.width, p.n_agents) == (7, 8, 12, 3)
    assert p.abi_version == _ffi.DW_ABI_VERSION and p.obs_mask == 0x0BA and p.precision == 0
    assert p.q == 0.2 * 1000.0 / 5.67e-8 and p.q2 == p.q / 8.0
    assert (p.albedo_bare, p.albedo_light, p.albedo_dark, p.temp_optimal) == (0.5, 0.75, 0.25, 295.5)
    assert (p.agent_gamma, p.light_proportion, p.initial_al) == (0.05, 0.33, 0.2)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu(built):
    import therldaisyworld_amd as t
    from therldaisyworld_amd._ffi import DaisyHipError, DW_ENODEVICE
    with pytest.raises(DaisyHipError) as e:
        t.Engine(t.default_params(1, 8, 8, 0))
    assert e.value.code == DW_ENODEVICE
    with pytest.raises(DaisyHipError):
        t.RLDaisyWorld(grid_dimension=8)


def test_bad_arguments_are_rejected(built):
    from therldaisyworld_amd import _ffi
    lib = _ffi.load()
    p = _ffi.DwParams()
    lib.dw_default_params(C.byref(p), 1, 2, 8, 0)      # height 2 < 3
    h = C.c_void_p()
    assert lib.dw_create(C.byref(p), C.byref(h)) == _ffi.DW_EINVAL
    assert b"3x3" in lib.dw_last_error()
    lib.dw_default_params(C.byref(p), 1, 50000, 50000, 0)   # H*W overflows the kernels' int32 cell index
    assert lib.dw_create(C.byref(p), C.byref(h)) == _ffi.DW_EINVAL
    assert b"2^31-1" in lib.dw_last_error()
    lib.dw_default_params(C.byref(p), 1, 8, 8, 0)
    p.abi_version = 99
    assert lib.dw_create(C.byref(p), C.byref(h)) == _ffi.DW_EINVAL
    assert lib.dw_step(None, None, 0, 0, 1.0) == _ffi.DW_EINVAL
    assert lib.dw_destroy(None) == 0


def test_host_greedy_matches_reference_fixture_g4(golden):
    from therldaisyworld_amd import Greedy
    g = golden("G4_greedy")
    obs = g["obs"]
    assert np.array_equal(Greedy()(obs), g["greedy"])
    assert np.array_equal(Greedy(greedy=False)(obs), g["antigreedy"])
    for eps in (0.0, 0.5, 1.0):
        np.random.seed(1234)
        agent = Greedy(epsilon=eps)
        seq = np.array([agent(obs) for _ in range(12)])
        assert np.array_equal(seq, g[f"eps_{eps}_seq"])


def test_neighborhood_masks_match_reference_fixture_g6(golden, capsys):
    from therldaisyworld_amd.nn.functional import make_neighborhood
    from therldaisyworld_amd.engine import mask_bits, VON_NEUMANN_MASK, MOORE_MASK
    g = golden("G6_ft_convolve")
    for mode in ("moore", "von_neumann", "circular", "asdf"):
        for r in (1, 2, 3, 4):
            assert np.array_equal(make_neighborhood(radius=r, mode=mode), g[f"nbhd_{mode}_{r}"])
    assert "not recognized" in capsys.readouterr().out
    assert mask_bits(make_neighborhood(1, "von_neumann")) == VON_NEUMANN_MASK
    assert mask_bits(make_neighborhood(1, "moore")) == MOORE_MASK
    with pytest.raises(ValueError):
        mask_bits(make_neighborhood(2, "moore"))


def _bare_env(**attrs):
    """The drop-in without its constructor (which needs a GPU): host logic only."""
    from therldaisyworld_amd import RLDaisyWorld
    env = RLDaisyWorld.__new__(RLDaisyWorld)
    env.batch_size, env.dim, env.n_agents = 32, 16, 4
    env.initial_al = env.initial_ad = 0.2
    env.light_proportion = env.dark_proportion = 0.33
    env._idx_m = env._st_m = None
    env._agents_on_device = False
    for k, v in attrs.items():
        setattr(env, k, v)
    return env


def test_host_rng_call_order_matches_reference_g9(golden):
    """Constructor + reset draw from np.random in the reference's order: randint(agents),
    rand(dark), rand(light), randint(agents) (ref daisy_world_rl.py:81-83, :287-297, :175)."""
    g = golden("G9_ctor_rng_order")
    np.random.seed(int(g["seed"]))
    env = _bare_env()
    env.initialize_agents()
    light, dark = env.draw_initial_cover()
    env.initialize_agents()
    assert np.array_equal(light, g["ctor_grid"][:, 1]) and np.array_equal(dark, g["ctor_grid"][:, 2])
    assert np.array_equal(env._idx_m.array, g["ctor_agent_indices"])
    env.batch_size, env.n_agents = 5, 3
    light, dark = env.draw_initial_cover()
    env.initialize_agents()
    assert np.array_equal(light, g["reset_grid"][:, 1]) and np.array_equal(dark, g["reset_grid"][:, 2])
    assert np.array_equal(env._idx_m.array, g["reset_agent_indices"])
    assert np.array_equal(env._st_m.array, g["reset_agent_states"])


def test_update_L_and_config_host_logic(tmp_path):
    env = _bare_env(step_count=0, ramp_up_down=False, ramp_period=512, min_L=0.75, max_L=1.5, ddL=0.0)
    env.dL = (env.max_L - env.min_L) / env.ramp_period
    L = env.min_L
    for _ in range(600):
        L = env.update_L(L)
    assert L == 1.5 and env.step_count == 600
    # triangle ramp (ref :466-469)
    env = _bare_env(step_count=0, ramp_up_down=True, ramp_period=4, min_L=0.0, max_L=1.0, ddL=0.0, dL=0.25)
    seq, L = [], 0.0
    for _ in range(9):
        L = env.update_L(L)
        seq.append(L)
    from oracle import daisy_oracle as O
    o = O.OracleDaisyWorld(grid_dimension=8, n_agents=0, batch_size=1, ramp_period=4)
    o.P.ramp_up_down, o.P.min_L, o.P.max_L, o.dL = True, 0.0, 1.0, 0.25
    ref, Lo = [], 0.0
    for _ in range(9):
        Lo = o.update_L(Lo)
        ref.append(Lo)
    assert seq == ref == [0.25, 0.5, 0.75, 0.5, 0.25, 0.0, 0.0, 0.25, 0.5]
    # config round trip (ref :94-171)
    from therldaisyworld_amd import RLDaisyWorld
    env = _bare_env()
    for k in RLDaisyWorld._CONFIG_KEYS:
        setattr(env, k, 1.5)
    env.n_agents = 7
    path = tmp_path / "cfg.json"
    env.save_config(str(path))
    env2 = _bare_env()
    env2.restore_config(str(path))
    assert env2.make_config() == env.make_config() and len(env.make_config()) == 20


def test_action_marshalling():
    from therldaisyworld_amd.engine import Engine
    a = Engine._actions(np.array([[[3.0], [8.0]]]))
    assert a.dtype == np.int32 and a.shape == (1, 2) and a.tolist() == [[3, 8]]
    with pytest.raises(ValueError):
        Engine._actions(np.array([[[0.5]]]))
    with pytest.raises(ValueError):
        Engine._actions(np.zeros(3))


def test_host_mlp_mirror_matches_reference_fixture_g10(golden):
    """Host mirror of the MLP policy: same Glorot draws (np.random.randn order) and same actions."""
    from therldaisyworld_amd import MLP
    g = golden("G10_mlp")
    np.random.seed(4242)
    agent, adversary = MLP(), MLP()
    assert np.array_equal(agent.get_parameters(), g["params_agent"])
    assert np.array_equal(adversary.get_parameters(), g["params_adversary"])
    assert np.array_equal(agent(g["obs0"]), g["action0"])
    half = 2
    for t in range(1, 40):
        obs = g["obs"][t - 1]
        a = np.append(agent.get_action(obs[:, :half]), adversary.get_action(obs[:, half:]), axis=1)
        assert np.array_equal(a, g["actions"][t])
    clone = MLP()
    clone._apply_config(agent.make_config())
    assert np.array_equal(clone.get_parameters(), agent.get_parameters())


def test_host_mlp_restores_the_shipped_result_file_g11(golden, tmp_path):
    """A result file in the reference's format (in_dim/out_dim/h_dim/act_name/parameters) restores into
    the mirror, survives a save/restore round trip, and reproduces the reference's first actions."""
    import json
    from therldaisyworld_amd import MLP
    g = golden("G11_trained_mlp")
    cfg = {"in_dim": int(g["in_dim"]), "out_dim": int(g["out_dim"]), "h_dim": [int(v) for v in g["h_dim"]],
           "act_name": str(g["act_name"]), "parameters": [float(v) for v in g["parameters"]]}
    path = tmp_path / "best_agent.json"
    path.write_text(json.dumps(cfg))
    agent = MLP()
    agent.restore_config(str(path))
    assert np.array_equal(agent.get_parameters(), g["restored_parameters"])
    assert np.array_equal(agent(g["obs0"])[..., 0], g["actions"][0][..., 0])
    again = tmp_path / "again.json"
    agent.save_config(str(again))
    clone = MLP()
    clone.restore_config(str(again))
    assert np.array_equal(clone.get_parameters(), agent.get_parameters())
    assert sorted(clone.load_config(str(again)).keys()) == ["act_name", "h_dim", "in_dim", "out_dim", "parameters"]


def test_tools_and_entry_points_compile():
    """Every script the docs point at at least parses (they need a GPU to run)."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, n) for n in ("bench.py", "__graft_entry__.py")]
    assert len(files) >= 12
    for f in files:
        with open(f) as src:
            compile(src.read(), f, "exec")


def _random_episode(rng, K, B, N, p_done):
    """Rewards / done flags of K steps whose agents die for good, one after the other."""
    dead = np.zeros((B, N, 1), dtype=bool)
    rewards, dones = np.zeros((K, B, N, 1)), np.zeros((K, B, N, 1), dtype=bool)
    for t in range(K):
        dead |= rng.rand(B, N, 1) < p_done
        r = np.round(rng.rand(B, N, 1), 3) * ~dead
        rewards[t], dones[t] = r, r < 0.1
    return rewards, dones


@pytest.mark.parametrize("chunk", [1, 5, 64])
@pytest.mark.parametrize("p_done", [0.0, 0.02, 0.2])
def test_fitness_bookkeeping_per_chunk_equals_the_reference_loop(chunk, p_done):
    """harness._fitness_chunk / _population_chunk account a chunk of steps at once; the result (float64 sums
    accumulated in step order, integer counters, stopping step) equals the reference's per-step statements
    (daisy/evo/sges.py:160-176) whatever the chunking."""
    from therldaisyworld_amd import harness
    rng = np.random.RandomState(int(chunk * 100 + p_done * 1000))
    K, P, wpm, N = 40, 3, 4, 4
    B, half = P * wpm, N // 2
    rewards, dones = _random_episode(rng, K, B, N, p_done)
    # --- reference loop, one world block (get_fitness) ---
    done_at, total, sum_reward, steps = np.zeros((B, N, 1), dtype=int), 0, 0.0, 0
    for t in range(K):
        reward, done = rewards[t], dones[t]
        all_done = (np.ones_like(done).sum() - done.sum()) == 0
        done_at += (1 - 1 * done)
        sum_reward += (reward[:, :half]).mean()
        total = total + (1 - 1 * done)
        steps += 1
        if all_done:
            break
    acc = {"done_at": np.zeros((B, N, 1), dtype=int), "total_steps": 0, "sum_reward": 0.0}
    got, t0 = 0, 0
    while t0 < K:
        e, fin = harness._fitness_chunk(acc, rewards[t0:t0 + chunk], dones[t0:t0 + chunk], half)
        got += e
        if fin:
            break
        t0 += chunk
    assert got == steps and acc["sum_reward"] == sum_reward
    assert np.array_equal(acc["done_at"], done_at) and np.array_equal(acc["total_steps"], total)
    # --- reference loop per member on its own block of worlds (get_fitness_population) ---
    want = []
    for m in range(P):
        sl = slice(m * wpm, (m + 1) * wpm)
        d_at, tot, s, n = np.zeros((wpm, N, 1), dtype=int), np.zeros((wpm, N, 1), dtype=int), 0.0, 0
        for t in range(K):
            reward, done = rewards[t][sl], dones[t][sl]
            d_at += (1 - 1 * done)
            tot += (1 - 1 * done)
            s += reward[:, :half].mean()
            n += 1
            if done.all():
                break
        want.append((d_at, tot, s, n))
    acc = {"done_at": np.zeros((B, N, 1), dtype=int), "total_steps": np.zeros((B, N, 1), dtype=int),
           "sum_reward": np.zeros(P), "running": np.ones(P, dtype=bool)}
    got, t0 = 0, 0
    while t0 < K:
        e, fin = harness._population_chunk(acc, rewards[t0:t0 + chunk], dones[t0:t0 + chunk], half, wpm)
        got += e
        if fin:
            break
        t0 += chunk
    assert got == max(w[3] for w in want)
    for m, (d_at, tot, s, n) in enumerate(want):
        sl = slice(m * wpm, (m + 1) * wpm)
        assert np.array_equal(acc["done_at"][sl], d_at) and np.array_equal(acc["total_steps"][sl], tot)
        assert acc["sum_reward"][m] == s


def test_pair_allocation_is_all_or_nothing(tmp_path):
    """when the second of two plane allocations fails, neither pointer may stay set - the next call
    must report the failure again (or retry), never find one plane and launch on a null second one.  The helper
    has no HIP types (csrc/dw_host_util.hpp), so it is exercised here with a mock allocator under g++."""
    import shutil
    import subprocess
    src = tmp_path / "pair.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include "dw_host_util.hpp"
static int n_alloc = 0, n_free = 0, fail_on = 0;
int main() {
    auto alloc = [](void** p, size_t n) { if (++n_alloc == fail_on) { *p = (void*)0x1; return 7; } *p = std::malloc(n); return *p ? 0 : 1; };
    auto release = [](void* p) { ++n_free; std::free(p); return 0; };
    void *a = nullptr, *b = nullptr;
    fail_on = 2;                                            // the SECOND allocation fails
    int rc = dw::alloc_pair_all_or_nothing(&a, &b, 64, alloc, release);
    if (rc != 7 || a || b || n_free != 1) { std::printf("after failure: rc=%d a=%p b=%p frees=%d\n", rc, a, b, n_free); return 1; }
    fail_on = 3;                                            // the retry fails at once: still nothing held
    rc = dw::alloc_pair_all_or_nothing(&a, &b, 64, alloc, release);
    if (rc != 7 || a || b) return 2;
    fail_on = 0;
    rc = dw::alloc_pair_all_or_nothing(&a, &b, 64, alloc, release);
    if (rc != 0 || !a || !b) return 3;
    void *a0 = a, *b0 = b;
    const int before = n_alloc;
    rc = dw::alloc_pair_all_or_nothing(&a, &b, 64, alloc, release);   // both present: nothing happens
    if (rc != 0 || a != a0 || b != b0 || n_alloc != before) return 4;
    void* lone = std::malloc(8);                            // a half-allocated pair left behind by older code
    void* none = nullptr;
    const int frees = n_free;
    rc = dw::alloc_pair_all_or_nothing(&lone, &none, 64, alloc, release);
    if (rc != 0 || !lone || !none || n_free != frees + 1) return 5;
    std::puts("ok");
    return 0;
}
''')
    gxx = shutil.which("g++")
    assert gxx
    exe = tmp_path / "pair"
    subprocess.check_call([gxx, "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "therldaisyworld_amd", "csrc"), str(src),
                           "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout)


def test_division_free_permille_quotient_is_exact_below_2_pow_22(tmp_path):
    """dw_div1000 / dw_permille_to_natural (csrc/dw_physics.hpp): k * 0.001 with one Newton correction is the correctly
    rounded k / 1000.0 for every integer |k| < 2^22 - per-mille covers (<= 1000) and rounded temperatures in
    milli-kelvin (<= 4194 K).  Checked exhaustively by the same two fma operations compiled for the host."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    src = tmp_path / "chk.c"
    src.write_text(
        "#include <stdio.h>\n#include <math.h>\n"
        "int main(void){ long bad = 0; for (long k = -(1L<<22) + 1; k < (1L<<22); ++k) { double kk = (double)k, r = 0.001,"
        " q = kk * r, v = fma(fma(-q, 1000.0, kk), r, q); if (v != kk / 1000.0) ++bad; } printf(\"%ld\\n\", bad); return 0; }\n")
    exe = tmp_path / "chk"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    assert subprocess.check_output([str(exe)]).strip() == b"0"


def test_ft_convolve_refuses_what_it_does_not_implement_before_touching_a_device():
    """therldaisyworld_amd.nn.functional.ft_convolve (ref daisy/nn/functional.py:12-49): argument checks run on the host,
    before an engine is created - a grid that is not (B, C, H, W), a kernel that is not 3x3 (the reference's un-padded
    branch), a grid below 3x3.  (The device results are pinned by fixture G6 in tests/test_gpu_round4.py.)"""
    from therldaisyworld_amd.nn.functional import ft_convolve, make_neighborhood
    k = np.ones((1, 1, 3, 3))
    with pytest.raises(ValueError):
        ft_convolve(np.zeros((4, 8, 8)), k)
    with pytest.raises(ValueError):
        ft_convolve(np.zeros((2, 1, 8, 8)), np.ones((1, 1, 8, 8)))
    with pytest.raises(ValueError):
        ft_convolve(np.zeros((2, 1, 2, 8)), k)
    assert make_neighborhood(1, "von_neumann").tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]


def test_host_helper_draws_numpys_legacy_stream_bit_for_bit():
    """include/daisyworld_host.h: dw_mt19937_random_sample = np.random.rand on the global legacy generator - the same
    doubles AND the same state afterwards (later rand / randn / randint calls continue identically), from any position in
    the 624-word state, with a cached Gaussian pending, across block boundaries; RLDaisyWorld._legacy_rand falls back to
    NumPy for small draws and gives the same numbers either way."""
    import re
    import subprocess
    from therldaisyworld_amd import _ffi, build
    from therldaisyworld_amd.daisy_world_rl import RLDaisyWorld
    build.build_host_library()
    host = _ffi.load_host()
    assert host is not None and host.dw_host_abi_version() == _ffi.DW_HOST_ABI_VERSION
    # every function the header declares is exported
    header = open(os.path.join(ROOT, "include", "daisyworld_host.h")).read()
    declared = set(re.findall(r"^int\s+(dw_\w+)\s*\(", header, flags=re.M))
    exported = subprocess.run(["nm", "-D", "--defined-only", build.HOST_LIB], capture_output=True, text=True).stdout
    assert declared == {"dw_mt19937_random_sample", "dw_mt19937_randint", "dw_mt19937_greedy_draws", "dw_host_abi_version"}
    for name in declared:
        assert re.search(rf"\bT {name}\b", exported), name
    rng = np.random.RandomState(5)
    saved = np.random.get_state()
    try:
        for trial in range(120):
            np.random.seed(int(rng.randint(2 ** 31)))
            pre = int(rng.randint(0, 1500))
            if pre:
                np.random.rand(pre)
            if rng.rand() < 0.3:
                np.random.randn(1)                           # leaves a cached Gaussian in the state
            shape = [(4096,), (4097,), (5000,), (8, 2, 16, 16), (3, 2, 40, 41), (100003,), (623 * 8,), (624 * 8 + 1,)][trial % 8]
            st = np.random.get_state()
            a, a2, g, r = np.random.rand(*shape), np.random.rand(7), np.random.randn(3), np.random.randint(9, size=5)
            np.random.set_state(st)
            b = RLDaisyWorld._legacy_rand(*shape)
            b2, g2, r2 = np.random.rand(7), np.random.randn(3), np.random.randint(9, size=5)
            assert b.shape == a.shape and np.array_equal(a, b), (trial, shape)
            assert np.array_equal(a2, b2) and np.array_equal(g, g2) and np.array_equal(r, r2), (trial, shape)
        # randint on the legacy masked-rejection path, and a chunk of the Greedy policy's draws (coin + randint(9) per step)
        from therldaisyworld_amd.agents.greedy import Greedy
        u32, i32p, i64p = C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        for trial in range(120):
            np.random.seed(int(rng.randint(2 ** 31)))
            pre = int(rng.randint(0, 1300))
            if pre:
                np.random.rand(pre)
            n = int(rng.choice([1, 5, 623, 624, 625, 4000, 20001]))
            low = int(rng.choice([0, 0, -3, 5]))
            high = low + int(rng.choice([2, 3, 9, 16, 17, 100, 65537]))
            st = np.random.get_state()
            a, a2 = np.random.randint(low, high, size=n), np.random.rand(3)
            key, pos, b = np.array(st[1], dtype=np.uint32), C.c_int32(int(st[2])), np.empty(n, dtype=np.int64)
            assert host.dw_mt19937_randint(key.ctypes.data_as(u32), C.byref(pos), low, high - 1 - low, b.ctypes.data_as(i64p), n) == 0
            np.random.set_state((st[0], key, pos.value, st[3], st[4]))
            assert a.dtype == b.dtype and np.array_equal(a, b) and np.array_equal(a2, np.random.rand(3)), (trial, n, low, high)
            eps, K, B, N = float(rng.choice([0.0, 0.5, 1.0, 0.3])), int(rng.randint(1, 40)), int(rng.randint(1, 200)), int(rng.randint(1, 6))
            agent = Greedy(epsilon=eps)
            st = np.random.get_state()
            ut, tb = np.zeros(K, dtype=np.uint8), np.zeros((K, B, N), dtype=np.int8)
            for t in range(K):                               # the harness's NumPy path (ref greedy.py:23-32 per step)
                if not agent.draw_branch():
                    ut[t] = 1
                    tb[t] = agent.draw_random_actions(B, N)[..., 0]
            after = np.random.rand(3)
            key, pos = np.array(st[1], dtype=np.uint32), C.c_int32(int(st[2]))
            ut2, tb2 = np.zeros(K, dtype=np.uint8), np.zeros((K, B, N), dtype=np.int8)
            assert host.dw_mt19937_greedy_draws(key.ctypes.data_as(u32), C.byref(pos), eps, K, B * N, ut2.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                tb2.ctypes.data_as(C.POINTER(C.c_int8))) == 0
            np.random.set_state((st[0], key, pos.value, st[3], st[4]))
            assert np.array_equal(ut, ut2) and np.array_equal(tb, tb2) and np.array_equal(after, np.random.rand(3)), (trial, eps, K, B, N)
        # bad arguments are refused without touching anything
        key = np.zeros(624, dtype=np.uint32)
        pos = C.c_int32(700)
        out = np.zeros(4)
        assert host.dw_mt19937_random_sample(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                             out.ctypes.data_as(C.POINTER(C.c_double)), 4) == -1
        assert pos.value == 700 and not out.any()
    finally:
        np.random.set_state(saved)


def test_board_power_sampler_parses_rocm_smi_and_reports_failures(tmp_path, monkeypatch):
    """therldaisyworld_amd.telemetry.board_power_while (bench.py's "power" object) against a stand-in rocm-smi that prints the
    tool's own format: maximum of the power samples, median clock, the limit; a missing tool, unparsable output, work that
    raises or ends before the first sample all come back as {"board_w": None, "error": ...}."""
    import stat
    import time
    from therldaisyworld_amd import telemetry
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\ncat <<'X'\n"
                    "============================ ROCm System Management Interface ============================\n"
                    "GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\n"
                    "GPU[0]\t\t: sclk clock level: 1: (2140Mhz)\n"
                    "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\n"
                    "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1389.0\nX\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    pw = telemetry.board_power_while(lambda: time.sleep(0.6), settle_s=0.05, samples=2)
    assert pw["board_w"] == 1389.0 and pw["limit_w"] == 1400.0 and pw["sclk_mhz"] == 2140 and len(pw["samples"]) == 2
    # the work is over before the first sample
    pw = telemetry.board_power_while(lambda: None, settle_s=0.05)
    assert pw["board_w"] is None and "no sample" in pw["error"]
    # the work raises: reported, not thrown
    def boom():
        raise RuntimeError("kernel fault")
    pw = telemetry.board_power_while(boom, settle_s=0.05)
    assert pw["board_w"] is None and "kernel fault" in pw["error"]
    # unparsable output
    fake.write_text("#!/bin/sh\necho nothing useful\n")
    pw = telemetry.board_power_while(lambda: time.sleep(0.3), settle_s=0.05)
    assert pw["board_w"] is None


def test_member_means_equal_the_reference_call_on_each_block():
    """harness._member_means (per step and member: the mean reward of the first half of the agents over the member's worlds)
    == `reward[:, :half].mean()` on that member's (wpm, N, 1) block, as daisy/evo/sges.py:170 forms it - bit for bit: the
    16-byte gather + contiguous reduction of the N = 4 configuration and the strided reduction of every other shape, for blocks
    below and above NumPy's pairwise-summation block size."""
    from therldaisyworld_amd.harness import _member_means
    rng = np.random.RandomState(2)
    for trial in range(200):
        K, P, wpm = int(rng.randint(1, 6)), int(rng.randint(1, 5)), int(rng.choice([1, 3, 4, 32, 33, 70, 300]))
        N = 4 if trial % 2 else int(rng.randint(1, 9))
        half = max(1, N // 2)
        r = rng.rand(K, P * wpm, N, 1) * float(rng.choice([1.0, 1e-3, 1e3]))
        want = np.array([[r[t, m * wpm:(m + 1) * wpm][:, :half].mean() for m in range(P)] for t in range(K)])
        assert np.array_equal(_member_means(r, P, wpm, half), want), (trial, K, P, wpm, N)
