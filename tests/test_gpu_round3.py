"""GPU tests added in round 3 (run with ``-m gpu`` on an MI355X, through the C ABI)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


# ---------------------------------------------------------------------------------------------
# a failed second plane allocation must leave the handle retryable, never half-allocated
# ---------------------------------------------------------------------------------------------
_ALLOC_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import therldaisyworld_amd as amd
from therldaisyworld_amd import _ffi
p = amd.default_params(3, 64, 64, 0)
eng = amd.Engine(p)
for attempt, call in enumerate((lambda: eng.init_random(7),
                                lambda: eng.upload_state(np.zeros((3, 64, 64)), np.zeros((3, 64, 64))))):
    try:
        call()
    except amd.DaisyHipError as e:
        assert e.code == _ffi.DW_ENOMEM, e
    else:
        raise SystemExit(f"injected allocation failure {attempt} was not reported")
eng.init_random(7)                       # the hook is spent: the same handle allocates both planes now
eng.step_n(3, 0.9, 0.001, 0.75, 1.5)
a = eng.download_planes()
ref = amd.Engine(p)
ref.init_random(7)
ref.step_n(3, 0.9, 0.001, 0.75, 1.5)
b = ref.download_planes()
assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
eng.upload_state(b[0], b[1])             # float64 pair: allocated all right as well
print("ok")
"""


def test_failed_plane_pair_allocation_is_reported_again_and_retryable():
    env = dict(os.environ, DW_TEST_FAIL_PAIR_ALLOC="2")
    p = subprocess.run([sys.executable, "-c", _ALLOC_SCRIPT, ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.returncode, p.stdout[-500:], p.stderr[-2000:])


# ---------------------------------------------------------------------------------------------
# repair-queue overflow no longer collapses the exact mode (adaptive strip height)
# ---------------------------------------------------------------------------------------------
def _strip_rows(eng):
    import re
    return int(re.search(r"wave-strip=(\d+)x256", eng.kernel_info()).group(1))


def test_exact_mode_adapts_its_strip_height_to_queue_overflows(amd, monkeypatch):
    """A repair queue that is too small for the state's tie density (forced here: 40 entries per wave against
    ~65 near-tie cells per 64-row strip pair) made every strip fall back to whole-strip float64: 4-6x the time.
    Now the overflow is reported to the host, which halves the strip height of the following launches: same
    results bit for bit, and after the adaptation at most 1.3x the time of the un-forced run."""
    B, G, warm, timed = 256, 256, 220, 64

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = _engine(amd, B, G, G, 0, "exact")
        for k in env:
            monkeypatch.delenv(k)
        eng.init_random(42)
        L = eng.step_n(warm, 0.75, 0.75 / 512, 0.75, 1.5)
        eng.step_n(64, L, 0.0, 0.75, 1.5)                        # adaptation happens here at the latest
        rows = _strip_rows(eng)
        eng.timer_start()
        eng.step_n(timed, L, 0.0, 0.75, 1.5)
        ms = eng.timer_stop()
        planes = eng.download_planes()
        eng.close()
        return ms, rows, planes

    ms_ref, rows_ref, ref = run({})
    ms_cap, rows_cap, got = run({"DW_TEST_QUEUE_CAP": "40"})
    ms_cliff, rows_cliff, cliff = run({"DW_TEST_QUEUE_CAP": "40", "DW_NO_ADAPT": "1"})
    for a, b in ((ref, got), (ref, cliff)):
        assert np.array_equal(_k(a[0]), _k(b[0])) and np.array_equal(_k(a[1]), _k(b[1]))
    assert rows_ref == 64 and rows_cliff == 64 and rows_cap < 64, (rows_ref, rows_cap, rows_cliff)
    assert ms_cap <= 1.3 * ms_ref, (ms_ref, ms_cap, ms_cliff)
    assert ms_cliff > 1.5 * ms_ref, (ms_ref, ms_cliff)          # what the adaptation avoids


def test_strip_height_recovers_after_clean_launches(amd, monkeypatch):
    """After enough launches without an overflow the strips grow back to their default height."""
    monkeypatch.setenv("DW_TEST_QUEUE_CAP", "40")
    eng = _engine(amd, 64, 256, 256, 0, "exact")
    monkeypatch.delenv("DW_TEST_QUEUE_CAP")
    eng.init_random(42)
    L = eng.step_n(260, 0.75, 0.75 / 512, 0.75, 1.5)              # developed: overflows, strips shrink
    assert _strip_rows(eng) < 64
    light, dark = eng.download_planes()
    eng.upload_state_f32(np.zeros_like(light, dtype=np.float32), np.zeros_like(dark, dtype=np.float32), quantised=True)
    eng.step_n(2000, L, 0.0, 0.75, 1.5)                          # a dead planet: no ties at all
    assert _strip_rows(eng) == 64
    eng.close()
