"""ctypes binding of include/daisyworld_hip.h (libdaisyworld_hip.so).

There is deliberately no fallback: if the shared library is missing, or no gfx950 device is
usable, the calls raise.  Build the library with ``python -m therldaisyworld_amd.build`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DW_LIB", os.path.join(_HERE, "libdaisyworld_hip.so"))   # DW_LIB: tuning builds

DW_ABI_VERSION = 5
DW_OK, DW_EINVAL, DW_ENODEVICE, DW_ENOMEM, DW_EHIP, DW_ESTATE = 0, -1, -2, -3, -4, -5
PRECISION = {"exact": 0, "fast": 1, "f64": 2}
STATE_CURRENT, STATE_PREVIOUS = 0, 1
POLICY_ARGMAX, POLICY_ARGMIN, POLICY_ZEROS, POLICY_TABLE = 0, 1, 2, 3
STAGE_ALBEDO, STAGE_DENSITY, STAGE_TEMPERATURE, STAGE_GROWTH_RATE, STAGE_GROWTH = 1, 2, 3, 4, 5
STAGE_IO = {1: (3, 3), 2: (2, 2), 3: (2, 4), 4: (3, 3), 5: (4, 2)}          # stage -> (planes in, planes out)


class DaisyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdaisyworld_hip error {code}: {msg}")
        self.code = code


class DwParams(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("n_agents", C.c_int32), ("device", C.c_int32), ("precision", C.c_int32), ("obs_mask", C.c_int32),
        ("collision_mode", C.c_int32), ("reserved0", C.c_int32), ("world_offset", C.c_int64),
    ] + [(n, C.c_double) for n in (
        "p", "g", "S", "sigma", "gamma", "q", "q2", "dt",
        "albedo_bare", "albedo_light", "albedo_dark", "temp_optimal",
        "agent_gamma", "food_chain_penalty",
        "initial_al", "initial_ad", "light_proportion", "dark_proportion")]


class DwWorldStats(C.Structure):
    _fields_ = [("max_k", C.c_uint32), ("reserved", C.c_uint32), ("sum_light_k", C.c_uint64),
                ("sum_dark_k", C.c_uint64)]


STATS_DTYPE = np.dtype([("max_k", "<u4"), ("reserved", "<u4"), ("sum_light_k", "<u8"), ("sum_dark_k", "<u8")])

_vp, _i32, _i64, _u32, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double
_pd, _pi, _pf, _pu8 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)

# name -> (restype, argtypes); every symbol include/daisyworld_hip.h declares
SIGNATURES = {
    "dw_default_params": (C.c_int, [C.POINTER(DwParams), _i32, _i32, _i32, _i32]),
    "dw_create": (C.c_int, [C.POINTER(DwParams), C.POINTER(_vp)]),
    "dw_destroy": (C.c_int, [_vp]),
    "dw_set_params": (C.c_int, [_vp, C.POINTER(DwParams)]),
    "dw_get_params": (C.c_int, [_vp, C.POINTER(DwParams)]),
    "dw_last_error": (C.c_char_p, []),
    "dw_abi_version": (C.c_int, []),
    "dw_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "dw_pinned_free": (C.c_int, [C.c_void_p]),
    "dw_build_id": (C.c_char_p, []),
    "dw_upload_state_f64": (C.c_int, [_vp, _pd, _pd]),
    "dw_upload_state_f32": (C.c_int, [_vp, _pf, _pf, C.c_int]),
    "dw_upload_agents": (C.c_int, [_vp, _pi, _pd]),
    "dw_download_agents": (C.c_int, [_vp, _pi, _pd]),
    "dw_init_random": (C.c_int, [_vp, _u64]),
    "dw_init_random_quantised": (C.c_int, [_vp, _u64]),
    "dw_download_planes": (C.c_int, [_vp, C.c_int, _pd, _pd]),
    "dw_download_grid": (C.c_int, [_vp, _dbl, _pd]),
    "dw_download_caches": (C.c_int, [_vp, _dbl, _pd, _pd, _pd, _pd]),
    "dw_step": (C.c_int, [_vp, _pi, _i32, _i32, _dbl]),
    "dw_step_device_actions": (C.c_int, [_vp, _dbl]),
    "dw_upload_actions": (C.c_int, [_vp, _pi]),
    "dw_download_actions": (C.c_int, [_vp, _pi]),
    "dw_step_n": (C.c_int, [_vp, _i32, _pd, _dbl, _dbl, _dbl, C.c_int]),
    "dw_last_step_n_timing": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(_i32), C.POINTER(_i32)]),
    "dw_update_agents": (C.c_int, [_vp, _pi, _i32, _i32]),
    "dw_forward_f64": (C.c_int, [_vp, _pd, _pd, _dbl, _pd, _pd, _pd, _pd, _pd]),
    "dw_conv3x3_f64": (C.c_int, [_vp, _pd, _pd, _pd]),
    "dw_stage_f64": (C.c_int, [_vp, C.c_int, _pd, _pd, _dbl, _pd]),
    "dw_get_obs": (C.c_int, [_vp, _dbl, _pd]),
    "dw_get_reward_done": (C.c_int, [_vp, _pd, _pu8]),
    "dw_reduce": (C.c_int, [_vp, C.POINTER(DwWorldStats)]),
    "dw_policy_greedy": (C.c_int, [_vp, C.c_int]),
    "dw_policy_per_agent": (C.c_int, [_vp, _pi]),
    "dw_policy_mlp": (C.c_int, [_vp, _pd, _i32, _i32, _i32, _dbl]),
    "dw_policy_mlp_population": (C.c_int, [_vp, _pd, _i32, _pi, _i32, _i32, _dbl]),
    "dw_env_step": (C.c_int, [_vp, _pi, _i32, _i32, _dbl, _pd, _pd, _pu8]),
    "dw_snapshot_save": (C.c_int, [_vp]),
    "dw_snapshot_restore": (C.c_int, [_vp]),
    "dw_snapshot_save_slot": (C.c_int, [_vp, C.c_int32]),
    "dw_snapshot_restore_slot": (C.c_int, [_vp, C.c_int32]),
    "dw_lifespan_reset": (C.c_int, [_vp]),
    "dw_lifespan_accumulate": (C.c_int, [_vp, _u32]),
    "dw_lifespan_download": (C.c_int, [_vp, _pi, _pi, _pi]),
    "dw_run_episode_mlp": (C.c_int, [_vp, _i32, _pd, _pd, _i32, _pi, _pi, _i32, _dbl, _pd, _pu8]),
    "dw_run_episode": (C.c_int, [_vp, _i32, _pd, C.c_int, _pu8, C.POINTER(C.c_int8), _u32, _pu8, _pu8]),
    "dw_set_stream": (C.c_int, [_vp, _vp]),
    "dw_sync": (C.c_int, [_vp]),
    "dw_timer_start": (C.c_int, [_vp]),
    "dw_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "dw_device_planes": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "dw_kernel_info": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "dw_last_fixup_count": (C.c_int, [_vp, C.POINTER(_u64)]),
    "dw_audit_tie_bound": (C.c_int, [_vp, _dbl, _pd]),
}

_libs = {}


def load(path=None):
    """Load the shared library and declare every prototype.  Raises if it has not been built.
    `path` selects another build of the same ABI (tuning A/B runs); default = the in-tree library."""
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build the HIP extension first "
                "(python -m therldaisyworld_amd.build); there is no CPU fallback")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if path == LIB_PATH:    # the product library must export every symbol of the header
                    raise
                continue                # an older tuning build (tools/kbench.py A/B): bind what it has
            fn.restype = res
            fn.argtypes = args
        if lib.dw_abi_version() != DW_ABI_VERSION:
            raise ImportError("libdaisyworld_hip.so ABI version mismatch; rebuild it")
        _libs[path] = lib
    return _libs[path]


HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdaisyworld_host.so")
DW_HOST_ABI_VERSION = 1
_host = []


def load_host():
    """The optional host helper (include/daisyworld_host.h: NumPy's legacy random stream in bulk), or None when it has not
    been built - its callers then use NumPy itself, with identical results."""
    if not _host:
        lib = None
        if os.path.exists(HOST_LIB_PATH):
            try:
                lib = C.CDLL(HOST_LIB_PATH)
                lib.dw_host_abi_version.restype = C.c_int
                lib.dw_host_abi_version.argtypes = []
                lib.dw_mt19937_random_sample.restype = C.c_int
                lib.dw_mt19937_random_sample.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                                         C.c_size_t]
                lib.dw_mt19937_randint.restype = C.c_int
                lib.dw_mt19937_randint.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_int64, C.c_uint64,
                                                   C.POINTER(C.c_int64), C.c_size_t]
                lib.dw_mt19937_greedy_draws.restype = C.c_int
                lib.dw_mt19937_greedy_draws.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_double, C.c_int32,
                                                        C.c_size_t, C.POINTER(C.c_uint8), C.POINTER(C.c_int8)]
                if lib.dw_host_abi_version() != DW_HOST_ABI_VERSION:
                    lib = None
            except (OSError, AttributeError):
                lib = None
        _host.append(lib)
    return _host[0]


def check(rc, lib=None):
    """Raise on a non-zero return code, with the message of the library that produced it."""
    if rc != DW_OK:
        raise DaisyHipError(rc, (lib or load()).dw_last_error().decode("utf-8", "replace"))


def ptr_d(a):
    return None if a is None else a.ctypes.data_as(_pd)


def ptr_i(a):
    return None if a is None else a.ctypes.data_as(_pi)


def ptr_f(a):
    return None if a is None else a.ctypes.data_as(_pf)


def ptr_u8(a):
    return None if a is None else a.ctypes.data_as(_pu8)
