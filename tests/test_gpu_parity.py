"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle and the golden vectors generated from the reference.

Bars:
  * DW_PRECISION_F64 and DW_PRECISION_EXACT: light/dark planes BIT-EXACT against the float64 oracle
    (integers k = 1000*cover compared), single steps and whole trajectories;
  * DW_PRECISION_FAST: every cell within one quantum (1e-3) and >= 99.98 % of the cell values identical after
    one step from the same state (measured: >= 99.991 % on synthetic random states, >= 99.994 % on developed
    states over the whole luminosity ramp; profiles/r02_fast_tolerance*.json);
  * agents / observations / rewards: exact equality with the reference fixtures (float64).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
from oracle import daisy_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import therldaisyworld_amd as t
    return t


@pytest.fixture(autouse=True)
def _packed_kernels_for_small_ensembles(request, monkeypatch):
    """The packed wave-strip kernels normally take over only for ensembles with >= 512 wave-strips (small
    batches stay on the lower-latency tiled / generic kernels).  The tests run small batches, so they
    lower the threshold - except those marked `default_pack_threshold`, which cover the small-batch paths."""
    if "default_pack_threshold" not in request.keywords:
        monkeypatch.setenv("DW_PACK_MIN_STRIPS", "1")


def _engine(amd, B, H, W, N=0, precision="exact", **over):
    from therldaisyworld_amd import _ffi
    p = amd.default_params(B, H, W, N)
    p.precision = _ffi.PRECISION[precision]
    for k, v in over.items():
        setattr(p, k, v)
    return amd.Engine(p)


def _random_quantised(rng, B, H, W, hi=400, sparsity=0.3):
    light = np.rint(rng.rand(B, H, W) * hi) * (rng.rand(B, H, W) > sparsity)
    dark = np.rint(rng.rand(B, H, W) * hi) * (rng.rand(B, H, W) > sparsity)
    tot = light + dark
    scale = np.where(tot > 1000, 1000.0 / np.maximum(tot, 1), 1.0)
    return np.floor(light * scale) / 1000.0, np.floor(dark * scale) / 1000.0


def _k(x):
    return np.rint(np.asarray(x) * 1000.0).astype(np.int64)


def _oracle_params(**over):
    return c_oracle.OracleParams.defaults(**over)


# ---------------------------------------------------------------------------------------------
# forward() on caller data: fixture G1 (generated from the reference)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_forward_matches_reference_fixture_g1(amd, golden, tag):
    g = golden("G1_forward")
    gi = g[f"{tag}_grid_in"]
    eng = _engine(amd, 2, 16, 16, 2)
    eng.upload_agents(g[f"{tag}_agent_indices"], g[f"{tag}_agent_states"][..., 0])
    grid, t, b, gr, e = eng.forward(gi[:, 1], gi[:, 2], float(g[f"{tag}_L"]), want_caches=True)
    ref = g[f"{tag}_grid_out"]
    assert np.array_equal(grid[:, :3], ref[:, :3])          # bare/light/dark: bit-exact
    assert np.array_equal(grid[:, 3:], ref[:, 3:])          # rounded temps + agent stamps + ch6
    np.testing.assert_allclose(t[:, 0:1], g[f"{tag}_temp"], rtol=1e-12)
    np.testing.assert_allclose(t[:, 1:2], g[f"{tag}_temp_light"], rtol=1e-12)
    np.testing.assert_allclose(t[:, 2:3], g[f"{tag}_temp_dark"], rtol=1e-12)
    np.testing.assert_allclose(e, g[f"{tag}_temp_effective"], rtol=1e-12)
    np.testing.assert_allclose(b[:, 0:1], g[f"{tag}_beta"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(b[:, 1:2], g[f"{tag}_beta_l"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gr, g[f"{tag}_growth"], rtol=1e-9, atol=1e-14)
    eng.close()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_dropin_stage_methods_g1(amd, golden, tag):
    """calculate_albedo / _daisy_density / _temperature / _growth_rate / _growth as stand-alone methods of the
    drop-in (ref :340-432), chained like forward() chains them, against the side-effect caches the reference left
    after forward() on the same grid (fixture G1) and stage by stage against the oracle's restatement.  The
    stencil stages convolve on the device (direct 9-tap form of the reference's FFT convolution: ~1e-15)."""
    g = golden("G1_forward")
    gi = g[f"{tag}_grid_in"].copy()
    env = amd.RLDaisyWorld(grid_dimension=16, n_agents=2)
    env.batch_size = 2
    env.reset()
    env.L = float(g[f"{tag}_L"])
    P = O.Params(grid_dimension=16, n_agents=2, batch_size=2)
    Al, A = env.calculate_albedo(gi[:, :3])
    oAl, oA = O.calculate_albedo(P, g[f"{tag}_grid_in"][:, :3].copy())
    np.testing.assert_allclose(Al, oAl, rtol=1e-14, atol=0)
    np.testing.assert_allclose(A, oA, rtol=1e-13, atol=0)
    assert np.array_equal(gi[:, 0], env.p - gi[:, 1] - gi[:, 2])      # channel 0 rewritten in place (ref :381)
    dens = env.calculate_daisy_density(gi[:, 1:3])
    np.testing.assert_allclose(dens, O.calculate_daisy_density(P, g[f"{tag}_grid_in"][:, 1:3]), rtol=1e-13, atol=1e-300)
    t, tl, td = env.calculate_temperature(Al, A)
    for ours, name in ((t, "temp"), (tl, "temp_light"), (td, "temp_dark"), (env.temp_effective, "temp_effective"),
                       (env.dead_temp, "dead_temp"), (env.temp, "temp")):
        np.testing.assert_allclose(ours, g[f"{tag}_{name}"], rtol=1e-12, atol=0)
    b, bl, bd = env.calculate_growth_rate(t, tl, td)
    for ours, name in ((b, "beta"), (bl, "beta_l"), (bd, "beta_d"), (env.beta_l, "beta_l")):
        np.testing.assert_allclose(ours, g[f"{tag}_{name}"], rtol=1e-9, atol=1e-12)
    gr = env.calculate_growth(b, bl, bd, dens)
    np.testing.assert_allclose(gr, g[f"{tag}_growth"], rtol=1e-9, atol=1e-14)
    assert env.growth is gr or np.array_equal(env.growth, gr)
    # the new covers forward() would produce from these stages equal the reference's (ref :449-452)
    new = np.round(np.clip(gi[:, 1:3] + env.dt * gr, 0, 1), 3)
    assert np.array_equal(new, g[f"{tag}_grid_out"][:, 1:3])
    env.close()


# ---------------------------------------------------------------------------------------------
# one step from a quantised state: all three precisions x kernel shapes
# ---------------------------------------------------------------------------------------------
SHAPES = [
    (3, 8, 8),        # generic kernel, tiny
    (2, 9, 13),       # generic kernel, odd non-square
    (4, 16, 16),      # packed wave-strip: 16 worlds per wave row, only 4 present
    (33, 8, 8),       # packed: 32 worlds per wave row, ragged second group (the README sweep's grid)
    (19, 24, 32),     # packed: 8 worlds per wave row, ragged
    (3, 64, 64),      # packed: 4 worlds per wave row (C1's shape)
    (2, 50, 64),      # packed, H < one strip
    (2, 96, 128),     # packed: 2 worlds per wave row, two row strips
    (5, 130, 128),    # packed, three row strips, ragged group
    (2, 40, 96),      # packed, W does not divide 256: 2 worlds on 48 lanes, 16 lanes idle
    (2, 50, 192),     # packed: one world on 48 lanes
    (5, 33, 100),     # packed: 2 worlds of 25 lanes, ragged group, odd group size (gathered reductions)
    (2, 40, 132),     # 33 lanes of 64 would be used: left to the tiled kernel
    (2, 256, 256),    # wave-strip, wrap inside the wave (C2's shape)
    (1, 70, 320),     # wave-strip, general halo
    (1, 33, 516),     # wave-strip, general halo, W % 256 = 4
    (1, 40, 258),     # W % 4 != 0 -> generic
    (3, 12, 12),      # generic (W = 12 does not divide 256)
    (2, 3, 256),      # streaming kernel, strip shorter than one row block
    (1, 5, 512),      # streaming kernel, dpp-old halo, odd tiny height
    (1, 65, 260),     # streaming kernel, general halo: second strip has a single active lane; H = SR + 1
]


@pytest.mark.parametrize("B,H,W", SHAPES)
@pytest.mark.parametrize("precision", ["f64", "exact", "fast"])
@pytest.mark.parametrize("L", [0.75, 1.0, 1.31])
def test_single_step_vs_oracle(amd, B, H, W, precision, L):
    rng = np.random.RandomState(B * 1000 + H + W)
    light, dark = _random_quantised(rng, B, H, W)
    ref = c_oracle.forward(light, dark, L)
    eng = _engine(amd, B, H, W, 0, precision)
    eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=True)
    eng.step(L)
    gl, gd = eng.download_planes()
    kl, kd, rl, rd = _k(gl), _k(gd), _k(ref[:, 1]), _k(ref[:, 2])
    if precision == "fast":
        # synthetic random states: measured worst over these shapes x luminosities 0.0089 % of the cell values
        # differing, all by one quantum (profiles/r02_fast_tolerance_cases.json); asserted at twice that (and one
        # flip always allowed, for the tiny grids)
        dl, dd = np.abs(kl - rl), np.abs(kd - rd)
        assert dl.max() <= 1 and dd.max() <= 1, "fast mode: more than one quantum off"
        n_diff = np.count_nonzero(dl) + np.count_nonzero(dd)
        assert n_diff <= max(1, int(np.ceil(1.8e-4 * 2 * dl.size))), f"fast mode: {n_diff} of {2 * dl.size} values differ"
    else:
        assert np.array_equal(kl, rl) and np.array_equal(kd, rd), f"{precision}: not bit-exact"
    # fused per-world reductions
    s = eng.reduce()
    assert np.array_equal(s["max_k"], np.maximum(kl.max(axis=(1, 2)), kd.max(axis=(1, 2))))
    assert np.array_equal(s["sum_light_k"], kl.sum(axis=(1, 2)))
    assert np.array_equal(s["sum_dark_k"], kd.sum(axis=(1, 2)))
    # previous state is retained un-touched
    pl, pd = eng.download_planes(1)
    assert np.array_equal(_k(pl), _k(light)) and np.array_equal(_k(pd), _k(dark))
    eng.close()


@pytest.mark.default_pack_threshold
@pytest.mark.parametrize("B,H,W,kernel", [(4, 16, 16, "step_generic"), (3, 64, 64, "step_tiled"),
                                          (2, 96, 128, "step_tiled"), (33, 8, 8, "step_generic")])
@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_small_batches_of_narrow_worlds_keep_the_low_latency_kernels(amd, B, H, W, kernel, precision):
    """Default kernel selection: a handful of narrow worlds is not worth a 64-row serial wave-strip; the
    tiled / generic kernels answer (and are still checked against the oracle here)."""
    rng = np.random.RandomState(B + H + W)
    light, dark = _random_quantised(rng, B, H, W)
    ref = c_oracle.forward(light, dark, 1.1)
    eng = _engine(amd, B, H, W, 0, precision)
    assert kernel in eng.kernel_info()
    eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=True)
    eng.step(1.1)
    gl, gd = eng.download_planes()
    if precision == "exact":
        assert np.array_equal(_k(gl), _k(ref[:, 1])) and np.array_equal(_k(gd), _k(ref[:, 2]))
    else:
        assert np.abs(_k(gl) - _k(ref[:, 1])).max() <= 1 and np.abs(_k(gd) - _k(ref[:, 2])).max() <= 1
    s = eng.reduce()
    assert np.array_equal(s["sum_light_k"], _k(gl).sum(axis=(1, 2)))
    eng.close()


@pytest.mark.default_pack_threshold
def test_big_ensembles_of_narrow_worlds_take_the_packed_kernels(amd):
    eng = _engine(amd, 2048, 64, 64, 0, "exact")                # 512 wave-strips of 4 worlds
    assert "halo=packed" in eng.kernel_info() and "fuses step pairs" in eng.kernel_info()
    eng.close()
    eng = _engine(amd, 2044, 64, 64, 0, "exact")                # 511 strips: below the threshold
    assert "step_tiled" in eng.kernel_info()
    eng.close()


def test_exact_mode_uses_float32_path_and_few_fixups(amd):
    """The exact mode must not silently run everything in float64: only near-tie cells are redone."""
    rng = np.random.RandomState(5)
    light, dark = _random_quantised(rng, 4, 256, 256)
    eng = _engine(amd, 4, 256, 256, 0, "exact")
    eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=True)
    eng.step(1.0)
    n = eng.last_fixup_count()
    assert "step_stream" in eng.kernel_info() or "step_tiled" in eng.kernel_info()
    assert 0 < n < 0.02 * light.size, f"{n} float64 fix-ups for {light.size} cells"
    eng.close()


@pytest.mark.parametrize("over", [
    dict(albedo_light=0.5, albedo_dark=0.5),          # neutral albedo (notebook sweep)
    dict(q2=0.0),                                     # set_use_microclimate(False)
    dict(dt=0.5),
    dict(dt=2.0, albedo_light=0.8, albedo_dark=0.2),
    dict(gamma=0.3, temp_optimal=290.0, g=0.004),
    dict(g=0.0),                                      # flat growth curve: beta = 1 (kappa = 1/(sqrt(g)*To) unbounded)
    dict(g=1e-9),
    dict(g=0.05, temp_optimal=310.0),                 # narrow curve: beta << 0 almost everywhere
])
def test_exact_mode_other_constants(amd, over):
    rng = np.random.RandomState(11)
    B, H, W = 2, 64, 128
    light, dark = _random_quantised(rng, B, H, W)
    for L in (0.8, 1.2):
        ref = c_oracle.forward(light, dark, L, _oracle_params(**over))
        eng = _engine(amd, B, H, W, 0, "exact", **over)
        eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=True)
        eng.step(L)
        gl, gd = eng.download_planes()
        assert np.array_equal(_k(gl), _k(ref[:, 1])) and np.array_equal(_k(gd), _k(ref[:, 2]))
        eng.close()


def test_negative_growth_curvature_is_float64_only(amd):
    """g < 0 (a parabola opening upwards) breaks the float32 map's sqrt(g) and the tie bound's om >= 0: the float32
    modes refuse it loudly, DW_PRECISION_F64 evaluates it like the reference's formulas."""
    from therldaisyworld_amd import _ffi
    rng = np.random.RandomState(12)
    B, H, W = 2, 32, 64
    light, dark = _random_quantised(rng, B, H, W)
    for prec in ("exact", "fast"):
        with pytest.raises(amd.DaisyHipError) as e:
            _engine(amd, B, H, W, 0, prec, g=-0.001)
        assert e.value.code == _ffi.DW_EINVAL and "g < 0" in str(e.value)
    eng = _engine(amd, B, H, W, 0, "f64", g=-0.001)
    eng.upload_state(light, dark)
    eng.step(1.1)
    gl, gd = eng.download_planes()
    ref = c_oracle.forward(light, dark, 1.1, _oracle_params(g=-0.001))
    assert np.array_equal(_k(gl), _k(ref[:, 1])) and np.array_equal(_k(gd), _k(ref[:, 2]))
    eng.close()


# ---------------------------------------------------------------------------------------------
# trajectories
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["exact", "f64"])
def test_c1_trajectory_bit_exact_g2(amd, golden, precision):
    """BASELINE config 1 (seed 42, B=1, 64x64, no agents, 500 steps) from the reference's own
    un-quantised float64 initial state: snapshots bit-identical to the reference."""
    g = golden("G2_c1_trajectory")
    eng = _engine(amd, 1, 64, 64, 0, precision)
    eng.upload_state(g["light0"], g["dark0"])
    L, dL, t = 0.75, 0.75 / 512, 0
    for snap in (int(s) for s in g["snap_steps"]):
        L = eng.step_n(snap - t, L, dL, 0.75, 1.5)
        t = snap
        gl, gd = eng.download_planes()
        assert np.array_equal(_k(gl).astype(np.uint16), g[f"light_k_{t}"]), f"light differs at t={t}"
        assert np.array_equal(_k(gd).astype(np.uint16), g[f"dark_k_{t}"]), f"dark differs at t={t}"
        grid = eng.download_grid()
        assert np.array_equal(grid[:, 3:6], g[f"temp3_{t}"])
        assert np.array_equal(grid[:, 0], g[f"bare_{t}"])
    assert L == float(g["final_L"])
    eng.close()


def test_c1_trajectory_fast_mode_statistics(amd, golden):
    """float32-only arithmetic, SINGLE world: the population curves of the 64x64 world of C1 stay within
    5e-3 absolute of the reference at every one of the 500 steps (measured worst 2.6e-3 = the assert / 2).
    SURVEY 8(c) hoped for 1e-3 here; a single small world cannot deliver that in any float32 arithmetic: one
    flipped tie (0.005 % of the cells per step, test below) is amplified by the pattern-forming dynamics.
    What float32 does deliver is the ENSEMBLE statistics - next test.  Cell-wise long-horizon equality is
    not claimed for this mode; it is what the default (exact) mode is for."""
    g = golden("G2_c1_trajectory")
    eng = _engine(amd, 1, 64, 64, 0, "fast")
    eng.upload_state(g["light0"], g["dark0"])
    L, dL = 0.75, 0.75 / 512
    n = 64 * 64
    worst = 0.0
    for t in range(500):
        L = eng.step_n(1, L, dL, 0.75, 1.5)
        s = eng.reduce()
        ml, md = s["sum_light_k"][0] / 1000.0 / n, s["sum_dark_k"][0] / 1000.0 / n
        worst = max(worst, abs(ml - g["mean_light"][t]), abs(md - g["mean_dark"][t]))
    assert worst < 5e-3, worst
    eng.close()


def test_fast_mode_tolerance_on_developed_states_over_the_ramp(amd):
    """The stated float32 tolerance, on DEVELOPED states over the whole luminosity ramp, as measured
    by tools/fast_tolerance.py (profiles/r02_fast_tolerance.json: 64 worlds of 256x256, 512 steps):
      * per step, from identical states: every cell within one quantum and >= 99.99 % of the cells identical
        (measured: at most 0.0056 % differ, at t = 360; SURVEY asks for >= 99.95 %);
      * trajectory: the ensemble means of light and dark stay within 3e-4 of the exact mode's at every step
        (measured 1.5e-4 / 8e-6; SURVEY asks for 1e-3).
    Both asserted at (measured x 2).  The exact engine's states ARE the float64 reference's (soak test)."""
    B, G, steps, every = 64, 256, 512, 8
    ex, fs, ft = (_engine(amd, B, G, G, 0, m) for m in ("exact", "fast", "fast"))
    ex.init_random(42)
    ft.init_random(42)                                       # walks the ramp on its own
    L, dL = 0.75, 0.75 / 512
    n = float(B) * G * G * 1000.0
    worst_frac, worst_q, drift = 0.0, 0, 0.0
    for t in range(steps):
        sample = t > 0 and t % every == 0
        if sample:
            sl, sd = ex.download_planes()
            fs.upload_state_f32(sl.astype(np.float32), sd.astype(np.float32), quantised=True)
        ex.step(L)
        ft.step(L)
        if sample:
            fs.step(L)
            rl, rd = (_k(x) for x in ex.download_planes())
            fl, fd = (_k(x) for x in fs.download_planes())
            dl, dd = np.abs(fl - rl), np.abs(fd - rd)
            worst_q = max(worst_q, int(dl.max()), int(dd.max()))
            worst_frac = max(worst_frac, (np.count_nonzero(dl) + np.count_nonzero(dd)) / (2.0 * dl.size))
        se, sf = ex.reduce(), ft.reduce()
        drift = max(drift, abs(float(sf["sum_light_k"].sum()) - float(se["sum_light_k"].sum())) / n,
                    abs(float(sf["sum_dark_k"].sum()) - float(se["sum_dark_k"].sum())) / n)
        L = min(max(L + dL, 0.75), 1.5)
    assert worst_q <= 1, worst_q
    assert worst_frac <= 1.2e-4, worst_frac                 # >= 99.988 % identical (measured 99.9944 %)
    assert drift <= 3e-4, drift
    for e in (ex, fs, ft):
        e.close()


def test_multiworld_trajectory_vs_c_oracle(amd):
    """256x256 worlds (C2's grid), tiled exact kernel, 40 steps from a device-generated state."""
    B, H, W = 3, 256, 256
    eng = _engine(amd, B, H, W, 0, "exact")
    eng.init_random(1234)
    light, dark = eng.download_planes()
    L, dL = 0.75, 0.75 / 512
    Lo = c_oracle.step_n(light, dark, L, dL, 40)
    Lg = eng.step_n(40, L, dL, 0.75, 1.5)
    gl, gd = eng.download_planes()
    assert Lo == Lg
    assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark))
    eng.close()


def test_distinct_handles_from_distinct_threads(amd):
    """include/daisyworld_hip.h: a handle is not thread-safe, DISTINCT handles may be driven from distinct threads
    (ctypes drops the GIL for the call): four threads, each with its own handle, kernel family and error slot, run a
    trajectory concurrently; every one equals the oracle, and an error raised in one thread stays in that thread."""
    import threading
    from therldaisyworld_amd import _ffi
    shapes = [(3, 256, 256, "exact"), (2, 64, 64, "exact"), (1, 40, 1024, "f64"), (5, 16, 16, "exact")]
    steps, L, dL = 24, 0.75, 0.75 / 512                     # dyadic: L + t*dL is the oracle's repeated sum exactly
    starts, results, errors = [], [None] * len(shapes), [None] * len(shapes)
    for i, (B, H, W, prec) in enumerate(shapes):
        light, dark = _random_quantised(np.random.RandomState(900 + i), B, H, W)
        starts.append((light, dark))
    gate = threading.Barrier(len(shapes))

    def run(i):
        try:
            B, H, W, prec = shapes[i]
            eng = _engine(amd, B, H, W, 0, prec)
            eng.upload_state(*starts[i])
            gate.wait(timeout=120)
            for t in range(steps):                          # single steps and a fused run, interleaved with the others
                if t == 8:
                    eng.step_n(8, L + 8 * dL, dL, 0.75, 1.5)
                elif t < 8 or t >= 16:
                    eng.step(L + t * dL)
            lib = _ffi.load()
            if i == 0:                                      # an error lands in THIS thread's message slot only
                assert lib.dw_upload_state_f64(eng._h, None, None) == _ffi.DW_EINVAL
                assert b"null argument" in lib.dw_last_error()
            gate.wait(timeout=120)
            if i != 0:
                assert b"null argument" not in (lib.dw_last_error() or b"")
            results[i] = eng.download_planes()
            eng.close()
        except BaseException as e:                          # noqa: BLE001 - re-raised in the main thread
            errors[i] = e
            gate.abort()                                    # do not leave the others waiting

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(shapes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    for i, (light, dark) in enumerate(starts):
        l, d = light.copy(), dark.copy()
        c_oracle.step_n(l, d, L, dL, steps)
        assert np.array_equal(_k(results[i][0]), _k(l)) and np.array_equal(_k(results[i][1]), _k(d)), shapes[i]


# ---------------------------------------------------------------------------------------------
# the drop-in class against the reference fixtures
# ---------------------------------------------------------------------------------------------
def _g3_actions(g):
    return [None if f"action_{t}_none" in g.files else g[f"action_{t}"] for t in range(int(g["n_steps"]))]


@pytest.mark.parametrize("precision", ["exact", "f64"])
def test_dropin_agents_g3(amd, golden, precision):
    g = golden("G3_agents")
    np.random.seed(7)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=4, precision=precision)
    env.batch_size = 4
    env.reset()
    # same manual edits as the fixture generator (in-place mutation of the public arrays)
    env.agent_indices[0] = np.array([[2, 4], [1, 3], [6, 6], [0, 0]])
    env.agent_indices[1] = np.array([[5, 5], [7, 0], [3, 3], [3, 3]])
    env.agent_states[2, 1, 0] = 0.04
    env.agent_states[0, :, 0] = 0.3
    env.L = 1.0
    assert np.array_equal(env.grid[:, 1], g["light0"]) and np.array_equal(env.grid[:, 2], g["dark0"])
    assert np.array_equal(env.agent_indices, g["agent_indices0"])
    for t, a in enumerate(_g3_actions(g)):
        obs, reward, done, info = env.step(a)
        assert np.array_equal(env.agent_indices, g["agent_indices"][t]), t
        assert np.array_equal(env.agent_states, g["agent_states"][t]), t
        assert np.array_equal(env.grid[:, 1], g["light"][t]), t
        assert np.array_equal(env.grid[:, 2], g["dark"][t]), t
        assert np.array_equal(obs, g["obs"][t]), t
        assert np.array_equal(reward, g["reward"][t]), t
        assert np.array_equal(done, g["done"][t]) and done.dtype == np.bool_, t
        assert env.L == g["L_after"][t] and info == {}
    assert np.array_equal(env.grid, g["grid_final"])
    env.close()


def test_dropin_ctor_rng_order_g9(amd, golden):
    g = golden("G9_ctor_rng_order")
    np.random.seed(int(g["seed"]))
    env = amd.RLDaisyWorld(grid_dimension=16)
    grid = env.grid
    assert grid.shape == (32, 7, 16, 16)
    assert np.array_equal(grid[:, :3], g["ctor_grid"][:, :3])
    np.testing.assert_allclose(grid[:, 3:6], g["ctor_grid"][:, 3:6], rtol=1e-12)
    assert np.array_equal(env.agent_indices, g["ctor_agent_indices"])
    assert env.dL == float(g["ctor_dL"]) and env.L == float(g["ctor_L"])
    env.batch_size = 5
    env.n_agents = 3
    env.albedo_light = 0.7
    env.min_L = 0.8
    env.ramp_period = 100
    obs = env.reset()
    assert np.array_equal(env.grid[:, :3], g["reset_grid"][:, :3])
    np.testing.assert_allclose(env.grid[:, 3:6], g["reset_grid"][:, 3:6], rtol=1e-12)
    assert np.array_equal(env.agent_indices, g["reset_agent_indices"])
    assert np.array_equal(env.agent_states, g["reset_agent_states"])
    np.testing.assert_allclose(obs, g["reset_obs"], rtol=1e-12, atol=0)
    assert env.dL == float(g["reset_dL"]) and env.L == float(g["reset_L"])
    obs, reward, done, _ = env.step(np.random.randint(9, size=(5, 3, 1)))
    assert np.array_equal(obs, g["step_obs"])
    assert np.array_equal(reward, g["step_reward"]) and np.array_equal(done, g["step_done"])
    assert np.array_equal(env.grid, g["step_grid"])
    assert env.L == float(g["step_L"])
    cfg = env.make_config()
    assert sorted(cfg.keys()) == list(g["config_keys"])
    assert np.allclose([float(cfg[k]) for k in sorted(cfg)], g["config_vals"], rtol=0, atol=0)
    env.close()


def test_dropin_triangle_ramp_g12(amd, golden):
    """ramp_up_down / ramp_period / ddL assigned after construction, five ramp periods with agents acting,
    then a second reset(): luminosity schedule, rewards and grids identical to the reference's."""
    from test_oracle_golden import _g12_checks
    g = golden("G12_ramp_up_down")
    np.random.seed(99)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=2)
    _g12_checks(env, g)
    env.close()


def test_dropin_config_roundtrip_g13(amd, golden, tmp_path):
    """A config file written by the reference's save_config restores into the drop-in (20 keys, n_agents
    and albedos among them); the restored environment then reproduces the reference's 6 steps, and its own
    save_config writes the same JSON object back."""
    import json
    g = golden("G13_config_roundtrip")
    path = tmp_path / "cfg.json"
    path.write_text(str(g["config_json"]))
    np.random.seed(6)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.restore_config(str(path))
    env.batch_size = 4
    obs = env.reset()
    actions = np.random.randint(9, size=(6, 4, 3, 1))
    assert np.array_equal(actions, g["actions"])
    for t in range(6):
        obs, reward, done, _ = env.step(actions[t])
    assert np.array_equal(obs, g["obs_final"]) and np.array_equal(reward, g["reward_final"])
    assert np.array_equal(env.grid, g["grid_final"])
    assert env.L == float(g["L_final"]) and env.dL == float(g["dL_final"])
    cfg = env.make_config()
    assert sorted(cfg.keys()) == [str(k) for k in g["config_keys"]]
    assert [float(cfg[k]) for k in sorted(cfg.keys())] == [float(v) for v in g["config_after"]]
    again = tmp_path / "again.json"
    env.save_config(str(again))
    assert json.loads(again.read_text()) == cfg
    env.close()


@pytest.mark.parametrize("name", ["no_microclimate", "slow_time", "other_physics", "wide_albedo"])
def test_dropin_attribute_mutations_g14(amd, golden, name):
    """The constants notebooks change on a constructed environment (set_use_microclimate, dt, agent_gamma,
    q2, temp_optimal, gamma, g, albedos) reach the device: rewards, observations, grid and the temp /
    growth caches equal the reference's after 14 steps with agents."""
    from test_oracle_golden import _g14_checks
    g = golden("G14_attribute_mutations")
    np.random.seed(314)
    env = amd.RLDaisyWorld(grid_dimension=16, n_agents=2)
    _g14_checks(env, g, name)
    env.close()


def test_dropin_direct_method_calls_g15(amd, golden):
    """get_obs on caller-supplied positions, update_agents alone, forward(grid) as a function, grid
    assignment, in-place edits of env.grid / env.agent_states / env.agent_indices between steps: the
    drop-in's host mirrors notice every one of them, results equal the reference's."""
    from test_oracle_golden import _g15_checks
    g = golden("G15_direct_method_calls")
    np.random.seed(77)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=3)
    _g15_checks(env, g)
    env.close()


@pytest.mark.default_pack_threshold
@pytest.mark.parametrize("seed", list(range(60000, 60016)))
def test_dropin_random_operation_sequences_vs_oracle_environment(amd, seed):
    """A slice of tools/fuzz_dropin.py in the suite: random sequences of steps (full / partial / None /
    float actions), in-place edits and assignments of grid and agent arrays, get_obs elsewhere,
    update_agents / forward as plain methods, live constant changes, resets, collision and neighbourhood
    modes, cache reads - the drop-in against the NumPy oracle environment after every operation, RNG
    stream in lock step."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_dropin.py")
    spec = importlib.util.spec_from_file_location("fuzz_dropin", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = []
    assert mod.run_case(seed, log), log


@pytest.mark.parametrize("seed", [10005, 10010, 10012, 10033, 10040, 10041, 10042, 10043])
def test_lifespan_harness_random_configurations_vs_notebook_loop_on_oracle(amd, seed):
    """A slice of tools/fuzz_harness.py: the chunked, device-resident lifespan harness against the
    notebook's loop on the NumPy oracle environment (lifespans, final state, RNG stream).  The first four
    seeds are episodes whose last world dies exactly on the last step of a chunk (a case the harness once
    ran past)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_harness.py")
    spec = importlib.util.spec_from_file_location("fuzz_harness", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = []
    ok, info = mod.run_case(seed, log)
    assert ok, log


@pytest.mark.parametrize("seed,case", [(603, 26), (603, 3), (601, 17), (602, 44), (605, 120), (607, 9)])
def test_exact_mode_random_configurations_vs_c_oracle(amd, monkeypatch, seed, case):
    """A slice of tools/fuzz_exact.py: random shape / constants / upload format / luminosity schedule, planes AND
    reductions against the float64 C oracle.  (603, 26) is the case that found a round-3 regression: packed strips
    (W = 128) with the repair queue cut to one entry - a strip swept its queue inside the row loop, sent the sweep's
    sum corrections to the world's counters, overflowed later, was recomputed whole and counted those corrections
    twice (sum of light cover off by one); the corrections of packed strips now wait in LDS until the strip is known
    to finish without an overflow."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_exact.py")
    spec = importlib.util.spec_from_file_location("fuzz_exact", path)
    mod = importlib.util.module_from_spec(spec)
    saved = os.environ.get("DW_PACK_MIN_STRIPS")
    try:
        spec.loader.exec_module(mod)                     # (sets DW_PACK_MIN_STRIPS=1 if unset, like the fixture above)
        log = []
        same, ratio = mod.run_case(seed, case, log)
    finally:
        if saved is None:
            os.environ.pop("DW_PACK_MIN_STRIPS", None)
        else:
            os.environ["DW_PACK_MIN_STRIPS"] = saved
    assert same, log
    assert ratio < 0.6, log                              # the proven bound is never approached


@pytest.mark.default_pack_threshold
@pytest.mark.parametrize("seed", list(range(70000, 70012)))
def test_engine_random_call_sequences_vs_oracle_model(amd, seed):
    """A slice of tools/fuzz_engine.py: random interleavings of the C ABI's state-changing calls against
    the oracle model, everything compared after every operation."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_engine.py")
    spec = importlib.util.spec_from_file_location("fuzz_engine", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = []
    saved = os.environ.get("DW_PACK_MIN_STRIPS")
    try:
        ok, info = mod.run_case(seed, log)
    finally:
        if saved is None:
            os.environ.pop("DW_PACK_MIN_STRIPS", None)
        else:
            os.environ["DW_PACK_MIN_STRIPS"] = saved
    assert ok, log


def test_dropin_no_agents_g7(amd, golden):
    g = golden("G7_no_agents")
    np.random.seed(21)
    env = amd.RLDaisyWorld(grid_dimension=12, n_agents=0)
    env.batch_size = 3
    obs0 = env.reset()
    assert tuple(obs0.shape) == tuple(g["obs0_shape"])
    assert np.array_equal(env.grid[:, :3], g["grid0"][:, :3])
    for t in range(6):
        obs, reward, done, info = env.step()
        assert reward.dtype == np.bool_ and reward.shape == (3, 2)
        assert np.array_equal(reward, g["reward"][t]) and np.array_equal(done, g["done"][t])
        assert np.array_equal(env.grid, g["grids"][t])
    assert tuple(obs.shape) == (3, 0, 7, 3, 3)
    assert env.L == float(g["L_final"]) and env.step_count == int(g["step_count"])
    env.grid[1, 1:3] = 0.0          # in-place edit of the public array, as the fixture generator did
    obs, reward, done, info = env.step()
    assert np.array_equal(reward, g["dead_reward"]) and np.array_equal(done, g["dead_done"])
    env.close()


def test_reference_smoke_tests_on_dropin(amd):
    """The reference's own tests (tests/daisy/test_daisy_world_rl.py:14-68), run on the drop-in."""
    env = amd.RLDaisyWorld()
    a = env.grid
    b = env.forward(a)
    for ii in range(9):
        action = np.array([[[ii]]])
        obs, reward, done, info = env.step(action)
    assert not done.mean()
    assert type(info) == dict
    assert 0.0 <= reward.mean()
    assert a.shape == b.shape
    assert obs.shape[1] == env.n_agents and obs.shape[0] == env.batch_size
    env = amd.RLDaisyWorld()
    for ch in (3, 4, 5):
        assert 0 < env.grid[:, ch].mean()
    env.reset()
    for ch in (3, 4, 5):
        assert 0 < env.grid[:, ch].mean()
    obs, reward, done, info = env.step()
    for ch in (3, 4, 5):
        assert 0 < env.grid[:, ch].mean() and 0 < obs[:, :, ch].mean()
    action = np.random.randint(9, size=(env.batch_size, env.n_agents, 1))
    obs, reward, done, info = env.step(action)
    for ch in (3, 4, 5):
        assert 0 < env.grid[:, ch].mean() and 0 < obs[:, :, ch].mean()
    env.close()


@pytest.mark.parametrize("agent_status,daisy_status", [
    ("greedy", "light_and_dark"), ("antigreedy", "light_and_dark"), ("random", "neutral_albedo"),
    ("half_random", "light_and_dark"), ("no", "light_and_dark")])
def test_dropin_lifespans_g5(amd, golden, agent_status, daisy_status):
    """The README's lifespan sweep protocol (dim 8, N=4, seed 13) on B=100 worlds: per-world
    biosphere and agent lifespans identical to the reference's."""
    g = golden("G5_lifespans")
    B, seed = int(g["B"]), int(g["seed"])
    np.random.seed(seed)
    env = amd.RLDaisyWorld(grid_dimension=8)
    env.batch_size = B
    if daisy_status == "neutral_albedo":
        env.albedo_dark = env.albedo_light = env.albedo_bare
    agent = {"greedy": amd.Greedy(epsilon=0.0), "antigreedy": amd.Greedy(epsilon=0.0, greedy=False),
             "random": amd.Greedy(epsilon=1.0), "half_random": amd.Greedy(epsilon=0.5), "no": None}[agent_status]
    env.reset()
    done_at, agents_done_at = O.simulate_lifespan(env, agent)   # the notebook's harness, any env
    key = f"{agent_status}_{daisy_status}"
    assert np.array_equal(done_at, g[key + "_done_at"])
    assert np.array_equal(agents_done_at, g[key + "_agents_done_at"])
    env.close()


# ---------------------------------------------------------------------------------------------
# device-side pieces
# ---------------------------------------------------------------------------------------------
def test_device_greedy_policy_matches_host_policy(amd):
    np.random.seed(3)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.batch_size = 64
    obs = env.reset()
    for _ in range(5):
        obs, *_ = env.step(np.random.randint(9, size=(64, 4, 1)))
    for argmin in (False, True):
        host = amd.Greedy(epsilon=0.0, greedy=not argmin)(obs)[..., 0]
        env._engine.policy_greedy(argmin=argmin)
        dev = env._engine.download_actions()
        assert np.array_equal(host, dev)
    env.close()


def test_init_random_distribution_and_shard_invariance(amd):
    B, H, W, N = 6, 64, 64, 3
    full = _engine(amd, B, H, W, N, "exact")
    full.init_random(99)
    fl, fd = full.download_planes()
    fi, fs = full.download_agents()
    # same worlds when the ensemble is sharded: world_offset keys the RNG
    for off, nb in ((0, 2), (2, 4)):
        sh = _engine(amd, nb, H, W, N, "exact", world_offset=off)
        sh.init_random(99)
        sl, sd = sh.download_planes()
        si, ss = sh.download_agents()
        assert np.array_equal(sl, fl[off:off + nb]) and np.array_equal(sd, fd[off:off + nb])
        assert np.array_equal(si, fi[off:off + nb]) and np.array_equal(ss, fs[off:off + nb])
        sh.close()
    # distribution of ref initialize_grid :299-302: P(cover>0)=0.33, cover = 0.2*U
    for plane in (fl, fd):
        frac = (plane > 0).mean()
        assert abs(frac - 0.33) < 0.01
        vals = plane[plane > 0]
        assert vals.max() <= 0.2 and abs(vals.mean() - 0.1) < 0.003
    assert not np.array_equal(fl, fd)
    assert fi.min() >= 0 and fi[..., 0].max() < H and fi[..., 1].max() < W and np.all(fs == 1.0)
    # the quantised draw: the same worlds rounded to three decimals (float32 rounding of 1000 * cover), straight
    # into the binary16 planes - steps, snapshots and device planes work at once
    q = _engine(amd, B, H, W, N, "exact")
    q.init_random(99, quantised=True)
    ql, qd = q.download_planes()
    assert np.array_equal(_k(ql), np.rint(np.float32(fl * 1000.0).astype(np.float64))) or np.abs(_k(ql) - _k(fl)).max() <= 1
    assert np.abs(ql - fl).max() <= 5.1e-4 and np.abs(qd - fd).max() <= 5.1e-4
    assert np.array_equal(ql, np.round(ql, 3)) and np.array_equal(q.download_agents()[0], fi)
    q.device_planes()
    q.snapshot_save()
    q.step(0.9)
    ref = c_oracle.forward(ql, qd, 0.9)
    kl, kd = (_k(x) for x in q.download_planes())
    assert np.array_equal(kl, _k(ref[:, 1])) and np.array_equal(kd, _k(ref[:, 2]))
    q.close()
    full.close()


def test_lifespan_accumulators(amd):
    np.random.seed(13)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.batch_size = 16
    env.min_L, env.max_L, env.ramp_period = 1.3, 1.6, 16      # hot: worlds die quickly
    obs = env.reset()
    eng = env._engine
    eng.lifespan_reset()
    done_at = np.zeros(16, dtype=int)
    agents_done_at = np.zeros((16, 4, 1), dtype=int)
    for _ in range(40):
        obs, reward, done, _ = env.step(np.random.randint(9, size=(16, 4, 1)))
        eng.lifespan_accumulate(5)
        done_at += (1 - 1 * (env.grid[:, 1:3].max(axis=(1, 2, 3)) <= 0.005))
        agents_done_at += (1 - 1 * done)
    d, a, alive = eng.lifespan_download()
    assert np.array_equal(d, done_at) and np.array_equal(a, agents_done_at)
    assert alive == int((env.grid[:, 1:3].max(axis=(1, 2, 3)) > 0.005).sum())
    env.close()


# ---------------------------------------------------------------------------------------------
# device-resident episode loop
# ---------------------------------------------------------------------------------------------
G5_CASES = [(a, d) for a in ("greedy", "antigreedy", "random", "half_random", "no")
            for d in ("light_and_dark", "neutral_albedo")]


def _g5_agent(amd, status):
    return {"greedy": amd.Greedy(epsilon=0.0), "antigreedy": amd.Greedy(epsilon=0.0, greedy=False),
            "random": amd.Greedy(epsilon=1.0), "half_random": amd.Greedy(epsilon=0.5), "no": None}[status]


@pytest.mark.parametrize("agent_status,daisy_status", G5_CASES)
def test_device_episode_loop_lifespans_g5(amd, golden, agent_status, daisy_status):
    """The whole README sweep protocol with the step loop resident on the device: per-world and
    per-agent lifespans identical to the reference's (all 5 policies x 2 albedo settings)."""
    from therldaisyworld_amd.harness import simulate_lifespan
    g = golden("G5_lifespans")
    B, seed = int(g["B"]), int(g["seed"])
    np.random.seed(seed)
    env = amd.RLDaisyWorld(grid_dimension=8)
    env.batch_size = B
    if daisy_status == "neutral_albedo":
        env.albedo_dark = env.albedo_light = env.albedo_bare
    env.reset()
    done_at, agents_done_at = simulate_lifespan(env, _g5_agent(amd, agent_status))
    key = f"{agent_status}_{daisy_status}"
    assert np.array_equal(done_at, g[key + "_done_at"])
    assert np.array_equal(agents_done_at, g[key + "_agents_done_at"])
    env.close()


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_device_episode_loop_leaves_env_like_host_loop(amd, precision):
    """After the harness the environment (grid, agents, L, step_count) and the NumPy RNG stream are
    exactly where the per-step host loop leaves them (16x16 worlds, half-random policy, odd chunk)."""
    from therldaisyworld_amd.harness import simulate_lifespan

    def run(device_loop):
        np.random.seed(5)
        env = amd.RLDaisyWorld(grid_dimension=16, n_agents=3, precision=precision)
        env.batch_size = 12
        env.min_L, env.max_L, env.ramp_period = 1.0, 1.6, 64
        agent = amd.Greedy(epsilon=0.5)
        d, a = simulate_lifespan(env, agent, chunk=7, use_device_loop=device_loop)
        out = (d, a, env.grid.copy(), env.agent_indices.copy(), env.agent_states.copy(), env.L, env.step_count,
               np.random.rand())
        env.close()
        return out

    dev, host = run(True), run(False)
    for x, y in zip(dev, host):
        assert np.array_equal(x, y)


def test_lifespan_harness_on_large_worlds_uses_device_reductions(amd):
    """Worlds above the LDS-resident limit (72x72) go through env.step per step, with the per-world
    "biosphere dead" flag from the step kernel's reductions instead of a full-grid download: same
    lifespans, final state and RNG stream as the notebook's loop verbatim (which reads env.grid)."""
    from therldaisyworld_amd.harness import simulate_lifespan

    def run(verbatim):
        np.random.seed(29)
        env = amd.RLDaisyWorld(grid_dimension=72, n_agents=3)
        env.batch_size = 4
        env.min_L, env.max_L, env.ramp_period = 1.1, 1.7, 48
        agent = amd.Greedy(epsilon=0.5)
        if verbatim:
            d, a = O.simulate_lifespan(env, agent)              # resets the environment itself, like ours
        else:
            d, a = simulate_lifespan(env, agent)
        out = (d, a, env.grid.copy(), env.agent_indices.copy(), env.agent_states.copy(), env.L, env.step_count,
               np.random.rand())
        env.close()
        return out

    ours, notebook = run(False), run(True)
    assert ours[6] > 10
    for x, y in zip(ours, notebook):
        assert np.array_equal(x, y)


def test_episode_table_codes_for_greedy_choices(amd):
    """Table entries -1 / -2 (greedy / anti-greedy choice of that agent at that step) in the LDS-resident
    episode kernel and in the per-step launch path: both equal policy_per_agent + step."""
    from therldaisyworld_amd import _ffi
    B, G, N, K = 7, 16, 4, 9
    rng = np.random.RandomState(3)
    table = rng.randint(-2, 9, size=(K, B, N)).astype(np.int8)
    outs = []
    for mode in ("lds", "launches", "manual"):
        eng = _engine(amd, B, G, G, N, "exact")
        eng.init_random(8)
        eng.step(0.9, np.zeros((B, N, 1), dtype=int))
        Ls = [0.92 + 0.01 * i for i in range(K)]
        if mode == "manual":
            ok = np.zeros((K, B, N), bool)
            for t in range(K):
                for code, argmin in ((-1, False), (-2, True)):          # whole-ensemble greedy pass, then pick
                    eng.policy_greedy(argmin=argmin)
                    g = eng.download_actions()
                    table_t = table[t].astype(np.int32)
                    table[t] = np.where(table_t == code, g, table_t).astype(np.int8)
                eng.upload_actions(table[t].astype(np.int32))
                eng.step_device_actions(Ls[t])
                ok[t] = ~eng.reward_done()[1][..., 0]
        else:
            if mode == "launches":
                import os
                os.environ["DW_NO_EPISODE_KERNEL"] = "1"
            try:
                _, ok = eng.run_episode(Ls, _ffi.POLICY_TABLE, None, table.copy())
            finally:
                import os
                os.environ.pop("DW_NO_EPISODE_KERNEL", None)
        outs.append((ok, *eng.download_planes(), *eng.download_agents()))
        eng.close()
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("B,H,W", [(9, 32, 32), (3, 72, 72), (2, 40, 256), (2, 66, 520),
                                   (5, 17, 17), (6, 20, 24), (4, 31, 33), (3, 16, 64), (3, 64, 16), (130, 24, 24), (7, 3, 300)])
def test_run_episode_matches_stepwise_engine(amd, B, H, W):
    """dw_run_episode == K x (policy + dw_step): planes, agents, flags, reductions, previous state —
    LDS-resident kernels (episode_small: 32x32, 72x72, odd and non-square grids with 256 < H*W <= 1024, more worlds than one
    round of workgroups, a 3-row world) and the back-to-back-launch path of larger worlds (tiled / streaming)."""
    from therldaisyworld_amd import _ffi
    N, K = 5, 11
    outs = []
    for mode in ("episode", "stepwise"):
        eng = _engine(amd, B, H, W, N, "exact")
        eng.init_random(77)
        L, dL = 1.0, 0.01
        eng.step(L, np.zeros((B, N, 1), dtype=int))         # quantise the state
        L += dL
        Ls = [L + i * dL for i in range(K)]
        rng = np.random.RandomState(1)
        use = (rng.rand(K) < 0.4).astype(np.uint8)
        table = rng.randint(9, size=(K, B, N)).astype(np.int8)
        if mode == "episode":
            alive, ok = eng.run_episode(Ls, _ffi.POLICY_ARGMAX, use, table)
        else:
            alive, ok = np.zeros((K, B), bool), np.zeros((K, B, N), bool)
            for t in range(K):
                if use[t]:
                    eng.upload_actions(table[t])
                else:
                    eng.policy_greedy(argmin=False)
                eng.step_device_actions(Ls[t])
                alive[t] = eng.reduce()["max_k"] > 5
                ok[t] = ~eng.reward_done()[1][..., 0]
        outs.append((alive, ok, eng.download_planes(), eng.download_planes(1), eng.download_agents(), eng.reduce(),
                     eng.download_grid(), eng.get_obs(), eng.download_actions()))
        eng.close()
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[8], b[8])                         # the action buffer: the LAST step's codes either way
    for i in (2, 3, 4):
        assert np.array_equal(a[i][0], b[i][0]) and np.array_equal(a[i][1], b[i][1])
    for f in ("max_k", "sum_light_k", "sum_dark_k"):
        assert np.array_equal(a[5][f], b[5][f])
    assert np.array_equal(a[6], b[6]) and np.array_equal(a[7], b[7])


@pytest.mark.parametrize("B,H,W,N", [(3, 64, 256, 5), (2, 130, 520, 3), (6, 32, 64, 4), (33, 8, 8, 6), (1, 70, 320, 16)])
@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("policy", ["argmax", "argmin_with_random_steps", "mixed_table"])
@pytest.mark.parametrize("K", [5, 11])       # 2 / 5 step pairs + the closing single step
def test_agent_step_pairs_equal_single_steps(amd, monkeypatch, B, H, W, N, precision, policy, K):
    """dw_run_episode without per-step world flags runs step PAIRS as one fused launch and patches the
    agents' in-between step in (policy from recomputed step-1 values, grazing in agent order, 3x3 blocks
    around the grazed cells recomputed): planes, agents, per-step agent flags, reductions, observations and
    the retained previous state equal K ordinary steps bit for bit - crowded 8x8 worlds (agents meet on
    cells), packed and wide grids, greedy / anti-greedy / random / per-agent mixes.  Between consecutive pairs
    the planes are binary16 (policy, grazing and the patch read and write them as such)."""
    from therldaisyworld_amd import _ffi
    rng = np.random.RandomState(B * 7 + N)
    use, table, mode = None, None, _ffi.POLICY_ARGMAX
    if policy == "argmin_with_random_steps":
        mode = _ffi.POLICY_ARGMIN
        use = (rng.rand(K) < 0.4).astype(np.uint8)
        table = rng.randint(9, size=(K, B, N)).astype(np.int8)
    elif policy == "mixed_table":                         # per agent and step: action, greedy (-1) or anti-greedy (-2)
        mode = _ffi.POLICY_TABLE
        table = rng.randint(-2, 9, size=(K, B, N)).astype(np.int8)
    outs = []
    for pairs in (True, False):
        if pairs:
            monkeypatch.delenv("DW_NO_AGENT_FUSE", raising=False)
        else:
            monkeypatch.setenv("DW_NO_AGENT_FUSE", "1")
        eng = _engine(amd, B, H, W, N, precision)
        eng.init_random(91)
        L, dL = 0.95, 0.012
        eng.step(L, np.zeros((B, N, 1), dtype=int))         # quantise the state
        Ls = [L + (i + 1) * dL for i in range(K)]
        alive, ok = eng.run_episode(Ls, mode, use, table, world_flags=False)
        assert alive is None
        outs.append((ok, *eng.download_planes(), *eng.download_planes(1), *eng.download_agents(),
                     eng.reduce().tobytes(), eng.download_grid(), eng.get_obs()))
        eng.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("B,H,W,N", [(3, 40, 256, 3), (2, 70, 320, 2), (9, 16, 32, 4), (33, 8, 8, 5)])
@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_agent_step_pairs_with_world_flags_through_the_death_of_the_worlds(amd, monkeypatch, B, H, W, N, precision):
    """With per-step "biosphere alive" flags requested the step pairs use the STATS variants of the fused
    kernels (exact step-1 maximum; count of certain step-2 values above the threshold, with a full scan of
    the patched result for dying worlds): flags of every step, agent flags and final state equal the
    one-launch-per-step path while the luminosity is driven up until every world is dead."""
    from therldaisyworld_amd import _ffi
    K = 45
    outs = []
    for pairs in (True, False):
        if pairs:
            monkeypatch.delenv("DW_NO_AGENT_FUSE", raising=False)
        else:
            monkeypatch.setenv("DW_NO_AGENT_FUSE", "1")
        eng = _engine(amd, B, H, W, N, precision)
        eng.init_random(17)
        L = eng.step_n(60, 0.9, 0.004, 0.75, 1.5)           # grow a biosphere first (agents idle)
        eng.step(L, np.zeros((B, N, 1), dtype=int))
        Ls = [min(L + 0.03 * (i + 1), 2.4) for i in range(K)]   # then overheat it
        alive, ok = eng.run_episode(Ls, _ffi.POLICY_ARGMAX, threshold_k=5)
        outs.append((alive, ok, *eng.download_planes(), *eng.download_agents(), eng.reduce().tobytes()))
        eng.close()
    a, b = outs
    assert a[0][0].all() and not a[0][-1].any()              # alive at first, dead at the end
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("B,H,W,N", [(4, 16, 16, 3), (2, 64, 256, 2)])
def test_snapshot_restore_replays_identically(amd, B, H, W, N):
    """dw_snapshot_save / dw_snapshot_restore: after a restore the same steps give the same state
    (planes, agents, reductions, observations), whatever ran in between."""
    from therldaisyworld_amd import _ffi
    eng = _engine(amd, B, H, W, N, "exact")
    eng.init_random(5)
    eng.step(0.9, np.zeros((B, N, 1), dtype=int))
    with pytest.raises(_ffi.DaisyHipError):
        eng.snapshot_restore()                               # nothing saved yet
    eng.snapshot_save()
    Ls = [0.95 + 0.01 * i for i in range(6)]

    def run():
        alive, ok = eng.run_episode(Ls, _ffi.POLICY_ARGMIN)
        return (alive, ok, *eng.download_planes(), *eng.download_agents(), eng.reduce().tobytes(), eng.get_obs())

    first = run()
    eng.step_n(3, 1.2, 0.0, 0.75, 1.5)                       # wander off
    eng.snapshot_restore()
    second = run()
    for x, y in zip(first, second):
        assert np.array_equal(x, y)
    eng.close()


def test_c1_trajectory_via_episode_kernel_g2(amd, golden):
    """C1 (64x64, no agents) with the step loop resident on the device: bit-exact snapshots."""
    from therldaisyworld_amd import _ffi
    g = golden("G2_c1_trajectory")
    eng = _engine(amd, 1, 64, 64, 0, "exact")
    eng.upload_state(g["light0"], g["dark0"])
    L, dL = 0.75, 0.75 / 512
    L = eng.step_n(1, L, dL, 0.75, 1.5)                      # first step from the un-quantised state
    t = 1
    for snap in (int(s) for s in g["snap_steps"]):
        if snap > t:
            Ls = []
            for _ in range(snap - t):
                Ls.append(L)
                L = min(max(L + dL, 0.75), 1.5)
            eng.run_episode(Ls, _ffi.POLICY_ZEROS)
            t = snap
        gl, gd = eng.download_planes()
        assert np.array_equal(_k(gl).astype(np.uint16), g[f"light_k_{t}"]), t
        assert np.array_equal(_k(gd).astype(np.uint16), g[f"dark_k_{t}"]), t
    assert L == float(g["final_L"])
    eng.close()


# ---------------------------------------------------------------------------------------------
# temporal fusion (two steps per HBM round trip, float32-only mode)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W", [(3, 256, 256), (2, 100, 256), (1, 70, 320), (1, 130, 516), (2, 64, 1024),
                                   (2, 3, 256), (1, 5, 500), (1, 65, 260),
                                   (5, 64, 64), (3, 40, 128), (33, 8, 8), (2, 130, 32), (17, 16, 16),   # packed
                                   (3, 70, 96), (5, 20, 100), (2, 66, 192)])          # packed, W does not divide 256
@pytest.mark.parametrize("nsteps", [3, 5, 8, 13])        # 1, 2, 3, 6 fused launches
@pytest.mark.parametrize("precision", ["fast", "exact"])
def test_fused_step_pairs_equal_single_steps(amd, monkeypatch, B, H, W, nsteps, precision):
    """dw_step_n fuses pairs of steps in one kernel (step_stream_fused2[_exact]): the result must be
    bit-identical to the same number of ordinary steps of the same arithmetic mode (rotate and
    overlapped-strip variants, partial strips, odd counts), the reductions must describe the final
    state and the retained previous state must be the true predecessor."""
    outs = []
    monkeypatch.setenv("DW_NO_EPISODE_KERNEL", "1")     # small packed worlds: keep dw_step_n on the step kernels
    for fuse in (True, False):
        if fuse:
            monkeypatch.delenv("DW_NO_FUSE", raising=False)
        else:
            monkeypatch.setenv("DW_NO_FUSE", "1")
        eng = _engine(amd, B, H, W, 0, precision)
        assert ("fuses step pairs" in eng.kernel_info()) == fuse
        eng.init_random(5)
        L = eng.step_n(nsteps, 0.9, 0.004, 0.75, 1.5)
        outs.append((L, eng.download_planes(), eng.download_planes(1), eng.reduce(), eng.download_grid()))
        eng.close()
    a, b = outs
    assert a[0] == b[0]
    assert np.array_equal(a[1][0], b[1][0]) and np.array_equal(a[1][1], b[1][1])
    assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1])
    for f in ("max_k", "sum_light_k", "sum_dark_k"):
        assert np.array_equal(a[3][f], b[3][f])
    assert np.array_equal(a[4], b[4])


@pytest.mark.parametrize("B,H,W", [(2, 100, 256), (1, 70, 320), (9, 40, 64), (3, 12, 12)])
def test_binary16_planes_are_lossless(amd, B, H, W):
    """The canonical plane format is binary16 per-mille (csrc/dw_common.hpp: every quantised value is an
    integer in [0, 1000], binary16 holds the integers up to 2048 exactly): every value 0..1000 goes through
    upload -> device planes -> download unchanged, as float64 k/1000 bit for bit; and a state stepped in
    exact mode equals the C oracle fed the same values (every kernel reads the planes it wrote)."""
    rng = np.random.RandomState(H)
    # every value 0..1000 at random positions (a REGULAR pattern - e.g. a linear ramp with bare = 0 - makes the
    # float64 pre-rounding values exact rounding ties, x.5, which no two evaluation orders round alike: the
    # reference's own FFT noise decides them; see the header of include/daisyworld_hip.h)
    light = rng.permutation(np.arange(B * H * W) % 1001).reshape(B, H, W).astype(np.float64)
    dark = np.floor(rng.rand(B, H, W) * (1001.0 - light)).astype(np.float64)
    assert set(np.unique(light)) == set(range(1001)) or light.size < 1001
    eng = _engine(amd, B, H, W, 0, "exact")
    eng.upload_state_f32((light / 1000.0).astype(np.float32), (dark / 1000.0).astype(np.float32), quantised=True)
    gl, gd = eng.download_planes()
    assert np.array_equal(gl, light / 1000.0) and np.array_equal(gd, dark / 1000.0)
    s = eng.reduce()
    assert np.array_equal(s["sum_light_k"], light.sum(axis=(1, 2)).astype(np.uint64))
    assert np.array_equal(s["max_k"], np.maximum(light.max(axis=(1, 2)), dark.max(axis=(1, 2))).astype(np.uint32))
    eng.step(1.0)
    ref = c_oracle.forward(light / 1000.0, dark / 1000.0, 1.0)
    kl, kd = (_k(x) for x in eng.download_planes())
    assert np.array_equal(kl, _k(ref[:, 1])) and np.array_equal(kd, _k(ref[:, 2]))
    pl, pd = eng.download_planes(1)
    assert np.array_equal(pl, light / 1000.0) and np.array_equal(pd, dark / 1000.0)
    eng.close()


@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("source", ["f64", "f32", "philox"])
def test_unquantised_initial_state_lives_in_its_upload_format(amd, precision, source):
    """An un-quantised state (the reference's initialize_grid does not round) cannot live in binary16 planes:
    it stays float64 / float32 until the first step has read it, then one more step as the previous state.
    Downloads, reductions, grazing, observations and the greedy policy see it in full precision; the first
    step equals the float64 oracle on exactly those values (exact mode) / is within tolerance (fast)."""
    B, G, N = 3, 48, 2
    rng = np.random.RandomState(17)
    light = (rng.rand(B, G, G) < 0.33) * 0.2 * rng.rand(B, G, G)
    dark = (rng.rand(B, G, G) < 0.33) * 0.2 * rng.rand(B, G, G)
    eng = _engine(amd, B, G, G, N, precision)
    if source == "f64":
        eng.upload_state(light, dark)
    elif source == "f32":
        eng.upload_state_f32(light.astype(np.float32), dark.astype(np.float32), quantised=False)
    else:
        eng.init_random(4)
    idx = rng.randint(G, size=(B, N, 2))
    eng.upload_agents(idx, np.ones((B, N)))
    gl, gd = eng.download_planes()
    if source == "f64":
        assert np.array_equal(gl, light) and np.array_equal(gd, dark)
    elif source == "f32":                                   # per-mille float32 on the device
        np.testing.assert_allclose(gl, light, rtol=3e-7, atol=0)
    assert (gl * 1000 != np.rint(gl * 1000)).any()          # really un-quantised
    from therldaisyworld_amd import _ffi
    with pytest.raises(_ffi.DaisyHipError):
        eng.device_planes()                                 # no binary16 planes before the first step
    with pytest.raises(_ffi.DaisyHipError):
        eng.snapshot_save()
    env = O.OracleDaisyWorld(grid_dimension=G, n_agents=N, batch_size=B)
    env.L = 0.9
    env.set_initial_cover(gl.copy(), gd.copy())
    env.agent_indices = idx.astype(np.int64)
    env.agent_states = np.ones((B, N, 1))
    np.testing.assert_allclose(eng.get_obs(0.9), env.get_obs(env.agent_indices), rtol=1e-12, atol=0)
    a = rng.randint(5, 9, size=(B, N, 1))                   # grazing actions: eat un-quantised covers
    obs, reward, done = eng.env_step(0.9, a)
    robs, rreward, rdone, _ = env.step(a)
    idx2, st2 = eng.download_agents()
    assert np.array_equal(idx2, env.agent_indices)
    kl, kd = (_k(x) for x in eng.download_planes())
    if precision == "exact":
        assert np.array_equal(st2[..., None], env.agent_states)
        assert np.array_equal(kl, _k(env.grid[:, 1])) and np.array_equal(kd, _k(env.grid[:, 2]))
        assert np.array_equal(obs, robs) and np.array_equal(reward, rreward) and np.array_equal(done, rdone)
        pl, pd = eng.download_planes(1)                     # the grazed un-quantised state is the previous state
        assert np.array_equal(pl[np.arange(B)[:, None], env.agent_indices[..., 0], env.agent_indices[..., 1]], np.zeros((B, N)))
    else:
        assert np.abs(kl - _k(env.grid[:, 1])).max() <= 1 and np.abs(kd - _k(env.grid[:, 2])).max() <= 1
    eng.step(0.91, np.zeros((B, N, 1), dtype=int))          # second step: binary16 planes on both sides
    env.L = 0.91
    env.step(np.zeros((B, N, 1), dtype=int))
    if precision == "exact":
        kl, kd = (_k(x) for x in eng.download_planes())
        assert np.array_equal(kl, _k(env.grid[:, 1])) and np.array_equal(kd, _k(env.grid[:, 2]))
        assert np.array_equal(eng.get_obs(), env.get_obs(env.agent_indices))
    eng.device_planes()
    eng.close()


def test_fused_fast_trajectory_vs_oracle_tolerance(amd):
    """Fused float32 stepping against the float64 C oracle from the same developed state: after 3 steps (a
    fused pair + one single step) every cell within 2 quanta and >= 99.94 % identical - measured 0.029 %
    differing, all by one quantum (profiles/r02_fast_tolerance_cases.json); asserted at twice that."""
    B, H, W = 2, 256, 256
    eng = _engine(amd, B, H, W, 0, "fast")
    eng.init_random(3)
    eng.step_n(60, 0.9, 0.002, 0.75, 1.5)            # developed, quantised state
    light, dark = eng.download_planes()
    L0 = 1.02
    L1 = eng.step_n(3, L0, 0.002, 0.75, 1.5)          # fused pair + one single step
    Lo = c_oracle.step_n(light, dark, L0, 0.002, 3)
    gl, gd = eng.download_planes()
    assert L1 == Lo
    dl, dd = np.abs(_k(gl) - _k(light)), np.abs(_k(gd) - _k(dark))
    assert dl.max() <= 2 and dd.max() <= 2
    assert (np.count_nonzero(dl) + np.count_nonzero(dd)) / (2.0 * dl.size) <= 6e-4
    eng.close()


@pytest.mark.parametrize("B,H,W", [(3, 256, 256), (1, 70, 320), (2, 64, 1024), (1, 200, 1024),
                                   (5, 64, 64), (33, 8, 8), (3, 130, 128), (9, 20, 16),                # packed
                                   (3, 70, 96), (5, 20, 100), (2, 66, 192)])
def test_fused_exact_trajectory_bit_exact_vs_oracle(amd, monkeypatch, B, H, W):
    """Exact mode with fused step pairs against the float64 C oracle over 41 steps (20 fused launches +
    the float64 first step from the un-quantised device state): bit-identical planes."""
    monkeypatch.setenv("DW_NO_EPISODE_KERNEL", "1")
    eng = _engine(amd, B, H, W, 0, "exact")
    assert "step_stream_fused2_exact" in eng.kernel_info()
    eng.init_random(11)
    light, dark = eng.download_planes()
    Lg = eng.step_n(41, 0.85, 0.01, 0.75, 1.5)
    Lo = c_oracle.step_n(light, dark, 0.85, 0.01, 41)
    gl, gd = eng.download_planes()
    assert Lg == Lo
    assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark))
    eng.close()


@pytest.mark.parametrize("over", [
    dict(albedo_light=0.5, albedo_dark=0.5), dict(q2=0.0), dict(dt=0.5),
    dict(dt=2.0, albedo_light=0.8, albedo_dark=0.2), dict(gamma=0.3, temp_optimal=290.0, g=0.004)])
@pytest.mark.parametrize("B,H,W", [(2, 64, 256), (1, 40, 264)])
def test_fused_exact_other_constants_vs_oracle(amd, over, B, H, W):
    """Fused exact step pairs with the constants callers change (neutral albedo, no microclimate, dt,
    growth parameters): both steps of a launch share their luminosity-independent coefficients and use
    one hi/lo split scale — still bit-identical to the float64 oracle over a luminosity ramp."""
    eng = _engine(amd, B, H, W, 0, "exact", **over)
    assert "step_stream_fused2_exact" in eng.kernel_info()
    eng.init_random(3)
    light, dark = eng.download_planes()
    Lg = eng.step_n(31, 0.8, 0.02, 0.75, 1.5)
    Lo = c_oracle.step_n(light, dark, 0.8, 0.02, 31, params=_oracle_params(**over))
    gl, gd = eng.download_planes()
    assert Lg == Lo
    assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark))
    eng.close()


@pytest.mark.parametrize("B,H,W,nsteps,kernel", [(2, 256, 256, 7, None), (1, 70, 320, 6, None), (2, 64, 128, 3, None),
                                                 (2, 40, 1024, 7, None),      # the four-wave ring: repairs across waves
                                                 (2, 64, 192, 3, None), (5, 64, 64, 6, None), (9, 16, 32, 5, None),
                                                 (3, 50, 100, 5, None), (2, 64, 128, 3, "tiled"),
                                                 (2, 64, 192, 3, "tiled"), (1, 70, 320, 4, "tiled")])
@pytest.mark.parametrize("qcap,mcap", [(2, 64), (256, 0), (0, 0)])
def test_exact_mode_overflow_fallbacks_are_exact(amd, monkeypatch, B, H, W, nsteps, kernel, qcap, mcap):
    """Shrink the near-tie queues so that they overflow everywhere: the fallbacks (whole strip / tile
    recomputed in float64, single-step and fused kernels, packed and tiled shapes) must still give the
    oracle's result."""
    monkeypatch.setenv("DW_NO_EPISODE_KERNEL", "1")
    if kernel:
        monkeypatch.setenv("DW_KERNEL", kernel)
    monkeypatch.setenv("DW_TEST_QUEUE_CAP", str(qcap))
    monkeypatch.setenv("DW_TEST_MISMATCH_CAP", str(mcap))
    eng = _engine(amd, B, H, W, 0, "exact")
    eng.init_random(23)
    light, dark = eng.download_planes()
    Lg = eng.step_n(nsteps, 0.95, 0.01, 0.75, 1.5)
    Lo = c_oracle.step_n(light, dark, 0.95, 0.01, nsteps)
    gl, gd = eng.download_planes()
    assert Lg == Lo
    assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark))
    s = eng.reduce()
    assert np.array_equal(s["sum_light_k"], _k(gl).sum(axis=(1, 2)))
    assert np.array_equal(s["max_k"], np.maximum(_k(gl).max(axis=(1, 2)), _k(gd).max(axis=(1, 2))))
    eng.close()


# ---------------------------------------------------------------------------------------------
# MLP policy on the device
# ---------------------------------------------------------------------------------------------
def test_mlp_policy_on_device_matches_reference_fixture_g10(amd, golden):
    """The 63-16-32-9 network evaluated by the policy_mlp kernel from device-resident observations:
    same actions as the reference's MLP on every step of a 40-step agent/adversary rollout, same
    rewards, same final grid; the host mirror gives the same actions from NumPy observations."""
    g = golden("G10_mlp")
    np.random.seed(4242)
    agent, adversary = amd.MLP(), amd.MLP()                 # same Glorot draws as the reference's MLP()
    assert np.array_equal(agent.get_parameters(), g["params_agent"])
    assert np.array_equal(adversary.get_parameters(), g["params_adversary"])
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.batch_size = 6
    obs = env.reset()
    np.testing.assert_allclose(obs, g["obs0"], rtol=1e-12, atol=0)
    assert np.array_equal(agent(obs), g["action0"])         # host mirror
    agent.act_on_device(env)
    assert np.array_equal(env._engine.download_actions()[..., None], g["action0"])
    half, sum_reward = 2, 0.0
    for t in range(40):
        agent.act_on_device(env, 0, half)
        adversary.act_on_device(env, half, 4)
        dev_actions = env._engine.download_actions()[..., None]
        assert np.array_equal(dev_actions, g["actions"][t]), t
        obs, reward, done, _ = env.step(dev_actions)
        assert np.array_equal(obs, g["obs"][t]), t
        assert np.array_equal(reward, g["rewards"][t]) and np.array_equal(done, g["dones"][t]), t
        sum_reward += reward[:, :half].mean()
    assert sum_reward == float(g["sum_reward"])
    assert np.array_equal(env.grid, g["grid_final"])
    env.close()


def test_get_fitness_harness_matches_reference_rollout_g10(amd, golden):
    from therldaisyworld_amd.harness import get_fitness
    g = golden("G10_mlp")
    np.random.seed(4242)
    agent, adversary = amd.MLP(), amd.MLP()
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=4)
    env.batch_size = 6
    # get_fitness resets the environment itself: rewind the stream to where the fixture's reset started
    np.random.seed(4242)
    amd.MLP(), amd.MLP()
    env2 = amd.RLDaisyWorld(grid_dimension=8, n_agents=4)   # consumes the constructor's draws again
    env2.batch_size = 6
    fitness, total_steps, done_at = get_fitness(env2, agent, adversary, max_steps=40)
    assert fitness == float(g["sum_reward"]) / (6 * 4)
    assert np.array_equal(np.array(done_at), (1 - 1 * g["dones"]).sum(axis=0))
    assert np.array_equal(total_steps, (1 - 1 * g["dones"]).sum(axis=0))
    assert np.array_equal(env2.grid, g["grid_final"])
    env.close(); env2.close()


@pytest.mark.parametrize("chunk", [64, 7])
def test_get_fitness_early_stop_inside_a_chunk(amd, chunk):
    """Agents that never graze (all-zero network: action 0) starve after ~19 steps, so the episode ends
    inside a device-resident chunk: fitness, counters and the state the environment is left in equal the
    step-by-step loop's (chunk=1), whatever the chunk size."""
    from therldaisyworld_amd.harness import get_fitness

    def run(chunk):
        lazy, other = amd.MLP(), amd.MLP()
        lazy.set_parameters(np.zeros(1808))
        other.set_parameters(np.zeros(1808))
        np.random.seed(31)
        env = amd.RLDaisyWorld(grid_dimension=16, n_agents=4)
        env.batch_size = 5
        f, total, done_at = get_fitness(env, lazy, other, max_steps=100, chunk=chunk)
        out = (f, np.asarray(total), np.asarray(done_at), env.grid.copy(), env.agent_indices.copy(),
               env.agent_states.copy(), env.L, env.step_count)
        env.close()
        return out

    ours, stepwise = run(chunk), run(1)
    assert 10 < ours[7] < 40                                 # all agents done long before max_steps
    for x, y in zip(ours, stepwise):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("seed", [10005, 10023, 10025, 10031, 10048, 10050, 10051, 10052])
def test_get_fitness_random_configurations_vs_reference_arithmetic_on_oracle(amd, seed):
    """A slice of tools/fuzz_fitness.py: the chunked ES fitness harness against sges.get_fitness's loop on the
    oracle environment with OracleMLP policies (fitness, counters, final state).  The first five seeds stop
    early inside a chunk with observation-dependent policies: the replay from the snapshot must see the same
    retained previous state (temperature channels of the observations) as the original run."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_fitness.py")
    spec = importlib.util.spec_from_file_location("fuzz_fitness", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log = []
    ok, info = mod.run_case(seed, log)
    assert ok, log


def test_trained_mlp_rollout_on_device_matches_reference_fixture_g11(amd, golden):
    """The trained policy the reference ships (results/cmaes_exp_002, generation 127) on the default
    16x16 world: policy_mlp on the device + step_device_actions for 160 steps reproduce the reference's
    actions, rewards, per-world populations and final grid bit for bit."""
    g = golden("G11_trained_mlp")
    agent = amd.MLP()
    agent.set_parameters(g["parameters"])
    np.random.seed(11)
    amd.MLP()                                                # the fixture drew one Glorot init before the env
    env = amd.RLDaisyWorld(grid_dimension=16, n_agents=4)
    env.batch_size = 8
    obs = env.reset()
    np.testing.assert_allclose(obs, g["obs0"], rtol=1e-12, atol=0)
    assert np.array_equal(env.agent_indices, g["agent_indices0"])
    eng = env._engine
    for t in range(160):
        agent.act_on_device(env)
        assert np.array_equal(eng.download_actions()[..., None], g["actions"][t]), t
        eng.step_device_actions(env.L)
        env._L_pass = env.L
        env._invalidate()
        reward, done = eng.reward_done()
        assert np.array_equal(reward * (reward > 0), g["rewards"][t]) and np.array_equal(done, g["dones"][t]), t
        env.L = env.update_L(env.L)
        if t % 16 == 15:
            assert np.array_equal(env.grid[:, 1].mean(axis=(-2, -1)), g["mean_light"][t]), t
            assert np.array_equal(env.grid[:, 2].mean(axis=(-2, -1)), g["mean_dark"][t]), t
    assert np.array_equal(env.grid, g["grid_final"])
    assert np.array_equal(env.agent_indices, g["agent_indices_final"])
    assert env.L == float(g["L_final"])
    env.close()


# ---------------------------------------------------------------------------------------------
# the exact mode's error bound, audited
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("over", [
    dict(), dict(albedo_light=0.5, albedo_dark=0.5), dict(q2=0.0), dict(dt=2.0, albedo_light=0.8, albedo_dark=0.2),
    dict(gamma=0.3, temp_optimal=290.0, g=0.004), dict(dt=0.5)])
def test_tie_bound_is_never_approached(amd, over):
    """|gq_f32 - gq_f64| against the per-cell bound eps of the tie test, over whole trajectories
    (luminosity ramp 0.75 -> 1.5, 26 M cell-values per constant set): the float32 error must stay
    below HALF the bound everywhere (the bound already carries a safety factor 2), and the flagged
    fraction must stay small enough for the fix-up to be cheap."""
    B, G = 8, 128
    eng = _engine(amd, B, G, G, 0, "exact", **over)
    eng.init_random(17)
    L, dL = 0.75, 0.75 / 100
    L = eng.step_n(1, L, dL, 0.75, 1.5)
    worst_ratio, worst_err, flagged, total = 0.0, 0.0, 0, 0
    for _ in range(100):
        err, ratio, nf, n = eng.audit_tie_bound(L)
        worst_ratio, worst_err = max(worst_ratio, ratio), max(worst_err, err)
        flagged += nf
        total += n
        L = eng.step_n(1, L, dL, 0.75, 1.5)
    assert worst_ratio < 0.5, (worst_ratio, worst_err)
    assert flagged / total < 0.02, flagged / total
    eng.close()


def test_dropin_collision_mode_g8(amd, golden):
    """collision_mode=1 with its as-implemented semantics (winner gains, losers unchanged, one RNG draw
    per multiply-occupied cell): agent states identical to the reference fixture."""
    g = golden("G8_collisions")
    np.random.seed(31)
    env = amd.RLDaisyWorld(grid_dimension=5, n_agents=4, collision_mode=1)
    env.batch_size = 3
    env.reset()
    env.agent_indices[0] = np.array([[2, 2], [2, 2], [1, 1], [4, 4]])
    env.agent_indices[1] = np.array([[0, 0], [0, 0], [0, 0], [3, 3]])
    env.agent_states[0, :, 0] = np.array([0.6, 0.4, 0.9, 0.2])
    env.agent_states[1, :, 0] = np.array([0.5, 0.7, 0.3, 0.8])
    assert np.array_equal(env.grid[:, 1], g["light0"]) and np.array_equal(env.agent_indices, g["agent_indices0"])
    np.random.seed(int(g["jitter_seed"]))
    for t in range(6):
        env.step(np.full((3, 4, 1), 8))
        assert np.array_equal(env.agent_indices, g["agent_indices"][t])
        assert np.array_equal(env.agent_states, g["agent_states"][t]), t
    assert np.array_equal(env.grid, g["grid_final"])
    env.close()


@pytest.mark.parametrize("mode", ["moore", "circular", "von_neumann"])
def test_dropin_neighbourhood_modes_vs_oracle(amd, mode):
    """Observation masks of the three reference neighbourhood modes at kr=1 (ref nn/functional.py:51-103;
    circular == von Neumann at radius 1): observations, rewards and grids equal the oracle's."""
    np.random.seed(9)
    env = amd.RLDaisyWorld(grid_dimension=10, n_agents=3, neighborhood_mode=mode)
    env.batch_size = 4
    obs = env.reset()
    np.random.seed(9)
    ref = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=10, n_agents=3, neighborhood_mode=mode)
    ref.P.batch_size = 4
    robs = ref.reset()
    np.testing.assert_allclose(obs, robs, rtol=1e-12, atol=0)
    assert (obs[:, :, :, 0, 0] != 0).any() == (mode == "moore")
    rng = np.random.RandomState(2)
    for _ in range(8):
        a = rng.randint(9, size=(4, 3, 1))
        obs, reward, done, _ = env.step(a)
        robs, rreward, rdone, _ = ref.step(a)
        assert np.array_equal(obs, robs) and np.array_equal(reward, rreward) and np.array_equal(done, rdone)
    assert np.array_equal(env.grid, ref.grid)
    with pytest.raises(ValueError):
        amd.RLDaisyWorld(grid_dimension=10, kr=2)        # the reference itself cannot observe with kr != 1 (:258)
    env.close()


def test_population_fitness_as_one_ensemble(amd):
    """A population of MLP policies evaluated as one batched ensemble (dw_policy_mlp_population): each
    member's fitness / done_at equal what the single-member get_fitness path computes on the same worlds."""
    from therldaisyworld_amd.harness import get_fitness_population
    np.random.seed(77)
    pop = [amd.MLP() for _ in range(3)]
    adversary_of = [1, 2, 0]
    wpm, N, G = 5, 4, 8
    np.random.seed(123)
    env = amd.RLDaisyWorld(grid_dimension=G, n_agents=N)
    res = get_fitness_population(env, pop, adversary_of, worlds_per_member=wpm, max_steps=30)
    # reference arithmetic per member on ITS block of worlds: oracle environment fed the same initial state
    np.random.seed(123)
    ref_env = O.OracleDaisyWorld.like_reference_ctor(grid_dimension=G, n_agents=N)
    ref_env.P.batch_size = 3 * wpm
    obs = ref_env.reset()
    nets = [O.OracleMLP(m.get_parameters()) for m in pop]
    half = N // 2
    sum_reward = np.zeros(3)
    done_at = np.zeros((3 * wpm, N, 1), dtype=int)
    running = np.ones(3, dtype=bool)
    while running.any() and ref_env.step_count < 30:
        acts = []
        for m in range(3):
            o = obs[m * wpm:(m + 1) * wpm]
            acts.append(np.append(nets[m].get_action(o[:, :half]), nets[adversary_of[m]].get_action(o[:, half:]), axis=1))
        obs, reward, done, _ = ref_env.step(np.concatenate(acts, axis=0))
        live = np.repeat(running, wpm)[:, None, None]
        done_at += live * (1 - 1 * done)
        for m in range(3):
            if running[m]:
                sum_reward[m] += reward[m * wpm:(m + 1) * wpm, :half].mean()
                running[m] = not done[m * wpm:(m + 1) * wpm].all()
    for m in range(3):
        assert res[m][0] == sum_reward[m] / (wpm * N)
        assert np.array_equal(np.array(res[m][2]), done_at[m * wpm:(m + 1) * wpm])
    assert np.array_equal(env.grid, ref_env.grid)
    env.close()


def test_policy_per_agent_modes(amd):
    from therldaisyworld_amd import _ffi
    np.random.seed(3)
    env = amd.RLDaisyWorld(grid_dimension=8, n_agents=6)
    env.batch_size = 16
    env.reset()
    for _ in range(4):
        obs, *_ = env.step(np.random.randint(9, size=(16, 6, 1)))
    eng = env._engine
    table = np.random.randint(9, size=(16, 6)).astype(np.int32)
    eng.upload_actions(table)
    eng.policy_per_agent([_ffi.POLICY_ARGMAX, _ffi.POLICY_ARGMIN, _ffi.POLICY_TABLE, _ffi.POLICY_ARGMAX,
                          _ffi.POLICY_TABLE, _ffi.POLICY_ARGMIN])
    got = eng.download_actions()
    gmax = amd.Greedy(epsilon=0.0)(obs)[..., 0]
    gmin = amd.Greedy(epsilon=0.0, greedy=False)(obs)[..., 0]
    want = np.stack([gmax[:, 0], gmin[:, 1], table[:, 2], gmax[:, 3], table[:, 4], gmin[:, 5]], axis=1)
    assert np.array_equal(got, want)
    env.close()


@pytest.mark.parametrize("B,H,W", [(1024, 256, 256), (2, 4096, 4096), (3000, 64, 64)])
@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_full_size_translation_invariance_and_checksums(amd, B, H, W, precision):
    """Size-independent properties at BASELINE's full sizes (C2; 4096-wide worlds; a packed ensemble), where
    the CPU oracle is too slow: the map commutes with translations of the torus (every strip, halo and
    world-group boundary is exercised, in both modes bit for bit), fused step pairs included, and the
    per-world reductions equal the checksums of the downloaded planes."""
    eng = _engine(amd, B, H, W, 0, precision)
    eng.init_random(123)
    eng.step(0.9)                                            # quantised start
    l0, d0 = eng.download_planes()
    L = eng.step_n(7, 0.91, 0.01, 0.75, 1.5)
    l1, d1 = eng.download_planes()
    s = eng.reduce()
    assert np.array_equal(s["sum_light_k"], np.rint(l1 * 1000).sum(axis=(1, 2)).astype(np.uint64))
    assert np.array_equal(s["sum_dark_k"], np.rint(d1 * 1000).sum(axis=(1, 2)).astype(np.uint64))
    assert np.array_equal(s["max_k"], np.rint(np.maximum(l1.max(axis=(1, 2)), d1.max(axis=(1, 2))) * 1000))
    dr, dc = 37 % H, 101 % W
    eng.upload_state_f32(np.roll(l0, (dr, dc), axis=(1, 2)).astype(np.float32),
                         np.roll(d0, (dr, dc), axis=(1, 2)).astype(np.float32), quantised=True)
    L2 = eng.step_n(7, 0.91, 0.01, 0.75, 1.5)
    l2, d2 = eng.download_planes()
    assert L2 == L
    assert np.array_equal(l2, np.roll(l1, (dr, dc), axis=(1, 2)))
    assert np.array_equal(d2, np.roll(d1, (dr, dc), axis=(1, 2)))
    eng.close()


def test_exact_mode_full_ramp_soak_vs_oracle(amd):
    """The whole luminosity ramp (512 steps: growth, pattern formation, die-off) of 8 worlds of 256x256
    in exact mode with fused step pairs against the float64 C oracle: 2.7e8 cell-updates, bit-identical
    planes at four checkpoints (any single tie resolved differently would show up and be amplified)."""
    B, G = 8, 256
    eng = _engine(amd, B, G, G, 0, "exact")
    eng.init_random(2026)
    light, dark = eng.download_planes()
    Lg = Lo = 0.75
    dL = 0.75 / 512
    for _ in range(4):
        Lg = eng.step_n(128, Lg, dL, 0.75, 1.5)
        Lo = c_oracle.step_n(light, dark, Lo, dL, 128)
        gl, gd = eng.download_planes()
        assert Lg == Lo
        assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark))
    s = eng.reduce()
    assert np.array_equal(s["sum_light_k"], _k(light).sum(axis=(1, 2)))
    eng.close()


def test_dropin_side_effect_caches_match_reference_g2(amd, golden):
    """env.temp / env.dead_temp / population means read through the drop-in's lazy attributes after each
    step equal the reference's (fixture G2: seed 42, 64x64, no agents, first 40 steps)."""
    g = golden("G2_c1_trajectory")
    np.random.seed(42)
    env = amd.RLDaisyWorld(grid_dimension=64, n_agents=0)
    env.batch_size = 1
    env.reset()
    assert np.array_equal(env.grid[:, 1], g["light0"]) and np.array_equal(env.grid[:, 2], g["dark0"])
    for t in range(40):
        assert env.L == g["L_used"][t]
        obs, reward, done, _ = env.step()
        assert abs(env.temp.mean() - g["mean_temp"][t]) < 1e-9
        assert env.dead_temp[0] == pytest.approx(g["dead_temp"][t], rel=1e-14)
        assert env.grid[:, 1].mean() == g["mean_light"][t] and env.grid[:, 2].mean() == g["mean_dark"][t]
        assert np.array_equal(reward, g["reward"][t]) and np.array_equal(done, g["done"][t])
        assert env.beta.shape == (1, 1, 64, 64) and env.growth.shape == (1, 2, 64, 64)
    env.close()


def test_external_stream_and_event_timer(amd):
    """dw_set_stream: the handle runs on a caller-owned HIP stream, results unchanged.  (The stream is made
    with the HIP runtime the library itself links, through ctypes; torch's current stream works the same
    way when torch initialised the device first, as bench.py does.)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    outs = []
    for use_external in (False, True):
        eng = _engine(amd, 2, 64, 128, 0, "exact")
        if use_external:
            stream = ctypes.c_void_p()
            assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
            eng.set_stream(stream.value)
        eng.init_random(4)
        eng.timer_start()
        L = eng.step_n(9, 0.9, 0.01, 0.75, 1.5)
        ms = eng.timer_stop()
        assert ms > 0.0
        outs.append((L, eng.download_planes()))
        eng.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1][0], outs[1][1][0]) and np.array_equal(outs[0][1][1], outs[1][1][1])
