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
