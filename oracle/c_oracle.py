"""ctypes front end of oracle/libdaisy_oracle.so (the C restatement in daisy_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of daisy_oracle.c.  Imported by tests/, smoke() and the
cpu_baseline leg of bench.py; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdaisy_oracle.so")


class OracleParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "p", "g", "S", "sigma", "gamma", "q", "q2", "dt",
        "albedo_bare", "albedo_light", "albedo_dark", "temp_optimal")]

    @classmethod
    def defaults(cls, **over):
        S, sigma = 1000.0, 5.67e-8
        q = 0.2 * S / sigma
        d = dict(p=1.0, g=0.003265, S=S, sigma=sigma, gamma=0.25, q=q, q2=q / 8.0, dt=1.0,
                 albedo_bare=0.5, albedo_light=0.75, albedo_dark=0.25, temp_optimal=295.5)
        d.update(over)
        return cls(**d)

    @classmethod
    def from_obj(cls, P):
        """From any object carrying the reference's attribute names (oracle Params, the shim...)."""
        return cls(**{n: float(getattr(P, n)) for n, _ in cls._fields_})


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "daisy_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdaisy_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a
    job a share of its host cores - 128 visible, ~16 usable: OpenMP teams sized by the visible count thrash)."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse:
                quota, period = parse(txt)
            else:
                quota, period = txt, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        _lib.oracle_forward.argtypes = [C.POINTER(OracleParams), C.c_double, C.c_int, C.c_int,
                                        C.c_int, dp, dp, dp, dp]
        _lib.oracle_forward.restype = C.c_int
        _lib.oracle_step_n.argtypes = [C.POINTER(OracleParams), dp, C.c_double, C.c_double,
                                       C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp]
        _lib.oracle_step_n.restype = C.c_int
        _lib.oracle_set_threads.argtypes = [C.c_int]
        assert _lib.oracle_sizeof_params() == C.sizeof(OracleParams)
        _lib.oracle_set_threads(min(int(_lib.oracle_max_threads()), usable_cpus()))
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def forward(light, dark, L, params: OracleParams | None = None, want_caches=False):
    """light, dark: (B,H,W) float64 -> out7 (B,7,H,W) [, caches (B,7,H,W)]."""
    P = params or OracleParams.defaults()
    light = np.ascontiguousarray(light, dtype=np.float64)
    dark = np.ascontiguousarray(dark, dtype=np.float64)
    B, H, W = light.shape
    out = np.empty((B, 7, H, W))
    caches = np.empty((B, 7, H, W)) if want_caches else None
    rc = lib().oracle_forward(C.byref(P), float(L), B, H, W, _dp(light), _dp(dark), _dp(out),
                              _dp(caches) if want_caches else None)
    assert rc == 0
    return (out, caches) if want_caches else out


def step_n(light, dark, L, dL, steps, min_L=0.75, max_L=1.5, params: OracleParams | None = None):
    """In-place `steps` no-agent steps on (B,H,W) float64 planes; returns the advanced L."""
    P = params or OracleParams.defaults()
    assert light.dtype == np.float64 and light.flags.c_contiguous
    assert dark.dtype == np.float64 and dark.flags.c_contiguous
    B, H, W = light.shape
    Lc = C.c_double(float(L))
    rc = lib().oracle_step_n(C.byref(P), C.byref(Lc), float(dL), float(min_L), float(max_L),
                             int(steps), B, H, W, _dp(light), _dp(dark))
    assert rc == 0
    return Lc.value


def set_threads(n: int):
    lib().oracle_set_threads(int(n))


def max_threads() -> int:
    return int(lib().oracle_max_threads())
