"""Round-4 GPU tests (``-m gpu``): what changed in the kernels this round, against the CPU oracle through the C ABI.

  * the wave-strip single-step kernels keep their per-world maximum as an integer maximum over ALL float32 values and
    settle near-tie cells afterwards (A / T / E, dw_step_stream.hpp): the per-world reductions must equal NumPy's on the
    oracle's planes on every strip geometry, also when every strip takes the re-scan path.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402


@pytest.fixture(scope="module")
def amd():
    import therldaisyworld_amd as t
    return t


def _engine(amd, B, H, W, N=0, precision="exact", **over):
    from therldaisyworld_amd import _ffi
    p = amd.default_params(B, H, W, N)
    p.precision = _ffi.PRECISION[precision]
    for k, v in over.items():
        setattr(p, k, v)
    return amd.Engine(p)


def _k(x):
    return np.rint(np.asarray(x) * 1000.0).astype(np.int64)


@pytest.mark.parametrize("B,H,W", [(3, 256, 256), (2, 100, 256), (2, 64, 1024), (1, 130, 516), (1, 70, 320), (1, 128, 4096)])
@pytest.mark.parametrize("force", [False, True])
def test_single_exact_steps_strip_maximum_with_and_without_rescan(amd, monkeypatch, B, H, W, force):
    """Single exact steps (no step pairs) on developed states: planes and the three per-world reductions equal the
    oracle's after every step - with the ordinary maximum logic and with every strip forced through `rescan_max`
    (the path a strip takes when a near-tie cell holds its largest float32 value)."""
    monkeypatch.setenv("DW_NO_FUSE", "1")
    monkeypatch.setenv("DW_NO_EPISODE_KERNEL", "1")
    if force:
        monkeypatch.setenv("DW_TEST_FORCE_RESCAN", "1")
    eng = _engine(amd, B, H, W, 0, "exact")
    assert "step_stream_exact" in eng.kernel_info()
    eng.init_random(5)
    light, dark = eng.download_planes()
    L = 0.9
    for chunk in (30, 1, 1, 1, 2):
        Lg = eng.step_n(chunk, L, 0.004, 0.75, 1.5)
        Lo = c_oracle.step_n(light, dark, L, 0.004, chunk)
        assert Lg == Lo
        L = Lg
        gl, gd = eng.download_planes()
        assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark))
        s = eng.reduce()
        assert np.array_equal(s["max_k"], np.maximum(_k(light).max(axis=(1, 2)), _k(dark).max(axis=(1, 2))))
        assert np.array_equal(s["sum_light_k"], _k(light).sum(axis=(1, 2)))
        assert np.array_equal(s["sum_dark_k"], _k(dark).sum(axis=(1, 2)))
    eng.close()


@pytest.mark.parametrize("B,H,W", [(3, 256, 256), (2, 64, 1024), (1, 70, 320)])
def test_single_fast_steps_reductions_are_the_checksums_of_the_planes(amd, monkeypatch, B, H, W):
    """float32-only single steps: the integer maximum and the packed sums are those of the planes the kernel wrote."""
    monkeypatch.setenv("DW_NO_FUSE", "1")
    monkeypatch.setenv("DW_NO_EPISODE_KERNEL", "1")
    eng = _engine(amd, B, H, W, 0, "fast")
    eng.init_random(6)
    L = 0.95
    for chunk in (25, 1, 1):
        L = eng.step_n(chunk, L, 0.004, 0.75, 1.5)
        gl, gd = eng.download_planes()
        s = eng.reduce()
        assert np.array_equal(s["max_k"], np.maximum(_k(gl).max(axis=(1, 2)), _k(gd).max(axis=(1, 2))))
        assert np.array_equal(s["sum_light_k"], _k(gl).sum(axis=(1, 2)))
        assert np.array_equal(s["sum_dark_k"], _k(gd).sum(axis=(1, 2)))
    eng.close()


def test_ft_convolve_on_device_matches_reference_fixture_g6(golden):
    """The drop-in's module-level ft_convolve (ref daisy/nn/functional.py:12-49; device twin dw_conv3x3_f64) against
    the reference's own outputs for an ASYMMETRIC kernel (fixture G6: orientation pin) on dims 5, 7, 8, 16, 17x16,
    9x12 and 64; plus a non-square multi-channel batch against the oracle's direct stencil."""
    from oracle import daisy_oracle as O
    from therldaisyworld_amd.nn.functional import ft_convolve
    g = golden("G6_ft_convolve")
    k = g["kernel"]
    assert k.shape == (1, 1, 3, 3) and not np.allclose(k[0, 0], k[0, 0].T)       # really asymmetric
    for i in range(int(g["n"])):
        y = ft_convolve(g[f"x_{i}"], k)
        assert y.shape == g[f"y_{i}"].shape and y.dtype == np.float64
        np.testing.assert_allclose(y, g[f"y_{i}"], rtol=0, atol=5e-15)
    rng = np.random.RandomState(3)
    x = rng.rand(3, 2, 10, 24)
    y = ft_convolve(x, k)
    np.testing.assert_allclose(y, O.toroidal_conv3x3(x, k[0, 0]), rtol=0, atol=5e-15)
    with pytest.raises(ValueError):
        ft_convolve(x, np.ones((1, 1, 10, 24)))                               # the reference's un-padded branch
    with pytest.raises(ValueError):
        ft_convolve(x[0], k)


_FAMILIES = {                                       # kernel family -> (B, H, W, extra environment)
    "rotate": (3, 100, 256, {}),                    # W == 256: the wrap is a rotation inside the wave
    "dpp_old": (2, 70, 512, {}),                    # W a multiple of 256
    "ring": (2, 40, 1024, {}),                      # W == 1024: step pairs by the four-wave ring
    "general": (2, 66, 516, {}),                    # any other multiple of 4 above 256 (narrow last strip, overlapped pairs)
    "packed": (9, 50, 64, {"DW_PACK_MIN_STRIPS": "1"}),
    "packed_odd_group": (5, 40, 96, {"DW_PACK_MIN_STRIPS": "1"}),      # 24 lanes per world: gathering lane groups
    "tiled": (3, 64, 128, {"DW_PACK_MIN_STRIPS": "100000"}),
}


@pytest.mark.parametrize("family", sorted(_FAMILIES))
@pytest.mark.parametrize("qcap,mcap", [(1, 0), (1, 2), (4, 1), (16, 0), (16, 2), (40, 1)])
def test_shrunk_repair_queues_in_every_kernel_family(amd, monkeypatch, family, qcap, mcap):
    """the sweep / overflow / reduction interplay of the exact mode is its fragile spot (a round-3
    regression lived there) - six shrunk-queue cases PER KERNEL FAMILY stay in the GPU suite: from the un-quantised
    Philox state (first-step kernel), a first single step, then 9 more steps through dw_step_n (four step pairs + a
    closing single step on the wave-strip families), queues cut to 1 .. 40 entries and mismatch lists to 0 .. 2: planes
    and all three reductions equal the float64 C oracle's."""
    B, H, W, env = _FAMILIES[family]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("DW_NO_EPISODE_KERNEL", "1")
    monkeypatch.setenv("DW_TEST_QUEUE_CAP", str(qcap))
    monkeypatch.setenv("DW_TEST_MISMATCH_CAP", str(mcap))
    eng = _engine(amd, B, H, W, 0, "exact")
    assert f"DW_TEST_QUEUE_CAP={qcap}" in eng.kernel_info()      # the handle reports the switches it was created under
    eng.init_random(31 + qcap)
    light, dark = eng.download_planes()
    L = 1.05
    for chunk in (1, 9):
        Lg = eng.step_n(chunk, L, 0.004, 0.75, 1.5)
        Lo = c_oracle.step_n(light, dark, L, 0.004, chunk)
        assert Lg == Lo
        L = Lg
        gl, gd = eng.download_planes()
        assert np.array_equal(_k(gl), _k(light)) and np.array_equal(_k(gd), _k(dark)), (family, qcap, mcap, chunk)
        s = eng.reduce()
        assert np.array_equal(s["max_k"], np.maximum(_k(light).max(axis=(1, 2)), _k(dark).max(axis=(1, 2))))
        assert np.array_equal(s["sum_light_k"], _k(light).sum(axis=(1, 2)))
        assert np.array_equal(s["sum_dark_k"], _k(dark).sum(axis=(1, 2)))
    eng.close()


@pytest.mark.parametrize("B,H,W,N", [(4, 16, 16, 3), (1, 5, 7, 2), (3, 64, 260, 0)])
def test_two_snapshot_slots_are_independent(amd, B, H, W, N):
    """dw_snapshot_save_slot / dw_snapshot_restore_slot (ABI 5): two states saved at different steps come back as they
    were, in either order, whatever ran in between (one copy launch per save; 1 x 5 x 7: planes of 70 bytes, the byte
    tail of the copy kernel)."""
    from therldaisyworld_amd import _ffi
    p = amd.default_params(B, H, W, N)
    p.precision = _ffi.PRECISION["exact"]
    eng = amd.Engine(p)
    eng.init_random(11)
    act = np.zeros((B, N, 1), dtype=int) if N else None
    eng.step(0.9, act)

    def state():
        out = [*eng.download_planes(), eng.reduce().tobytes()]
        if N:
            out += [*eng.download_agents(), eng.get_obs()]
        return out

    with pytest.raises(_ffi.DaisyHipError):
        eng.snapshot_restore(1)                              # nothing saved in that slot yet
    with pytest.raises(_ffi.DaisyHipError):
        eng.snapshot_save(2)                                 # two slots
    a = state()
    eng.snapshot_save(0)
    eng.step(0.95, act)
    eng.step(1.0, act)
    b = state()
    eng.snapshot_save(1)
    eng.step_n(3, 1.2, 0.0, 0.75, 1.5) if not N else [eng.step(1.2, act) for _ in range(3)]
    eng.snapshot_restore(0)
    for x, y in zip(a, state()):
        assert np.array_equal(x, y)
    eng.step(1.3, act)
    eng.snapshot_restore(1)
    for x, y in zip(b, state()):
        assert np.array_equal(x, y)
    eng.snapshot_restore(0)                                  # still there
    for x, y in zip(a, state()):
        assert np.array_equal(x, y)
    eng.close()
