"""`Engine`: a thin NumPy-in / NumPy-out object over the C ABI (include/daisyworld_hip.h).

It owns one ``dw_handle`` (one HIP device, one stream).  All compute happens in the HIP library; this
class only marshals arrays and raises ``DaisyHipError`` on any failure.  The gym-style drop-in
``therldaisyworld_amd.RLDaisyWorld`` is built on it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import DwParams, DwWorldStats

VON_NEUMANN_MASK = 0x0BA
MOORE_MASK = 0x1FF


def mask_bits(neighborhood: np.ndarray) -> int:
    """3x3 0/1 array -> the 9-bit row-major mask of dw_params.obs_mask."""
    nb = np.asarray(neighborhood)
    if nb.shape != (3, 3):
        raise ValueError("only kr=1 (3x3) observation neighbourhoods are supported "
                         "(the reference hard-codes 3, daisy_world_rl.py:258)")
    bits = 0
    for i, v in enumerate(nb.ravel()):
        if v:
            bits |= 1 << i
    return bits


def default_params(batch, height, width, n_agents) -> DwParams:
    p = DwParams()
    _ffi.check(_ffi.load().dw_default_params(C.byref(p), batch, height, width, n_agents))
    return p


class Engine:
    def __init__(self, params: DwParams, lib_path=None):
        self._lib = _ffi.load(lib_path)
        self._h = C.c_void_p()
        self.params = DwParams()
        C.memmove(C.byref(self.params), C.byref(params), C.sizeof(DwParams))
        self._check(self._lib.dw_create(C.byref(self.params), C.byref(self._h)))
        p = self.params
        self.B, self.H, self.W, self.N = p.batch, p.height, p.width, p.n_agents

    def _check(self, rc):
        _ffi.check(rc, self._lib)                           # the message of THIS engine's library build

    # -- life cycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.dw_destroy(self._h)
            self._h = C.c_void_p()
        for blk in (getattr(self, "_pinned", None) or {}).values():   # (arrays handed out with reuse_buffers die with them)
            self._lib.dw_pinned_free(blk[0])
        self._pinned = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params: DwParams):
        self._check(self._lib.dw_set_params(self._h, C.byref(params)))
        C.memmove(C.byref(self.params), C.byref(params), C.sizeof(DwParams))

    # -- state --------------------------------------------------------------------------------
    def _plane(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.shape != (self.B, self.H, self.W):
            raise ValueError(f"plane shape {a.shape} != {(self.B, self.H, self.W)}")
        return a

    def upload_state(self, light, dark):
        light, dark = self._plane(light, np.float64), self._plane(dark, np.float64)
        self._check(self._lib.dw_upload_state_f64(self._h, _ffi.ptr_d(light), _ffi.ptr_d(dark)))

    def upload_state_f32(self, light, dark, quantised=False):
        light, dark = self._plane(light, np.float32), self._plane(dark, np.float32)
        self._check(self._lib.dw_upload_state_f32(self._h, _ffi.ptr_f(light), _ffi.ptr_f(dark), int(quantised)))

    def upload_agents(self, indices, states):
        idx = np.ascontiguousarray(indices, dtype=np.int32).reshape(self.B, self.N, 2)
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(self.B, self.N)
        self._check(self._lib.dw_upload_agents(self._h, _ffi.ptr_i(idx), _ffi.ptr_d(st)))

    def download_agents(self):
        idx = np.empty((self.B, self.N, 2), dtype=np.int32)
        st = np.empty((self.B, self.N), dtype=np.float64)
        self._check(self._lib.dw_download_agents(self._h, _ffi.ptr_i(idx), _ffi.ptr_d(st)))
        return idx, st

    def init_random(self, seed: int, quantised: bool = False):
        """Device Philox initial state; quantised=True: covers rounded to 3 decimals, straight into the binary16
        planes (no float32 staging of the un-quantised state, no float64 first step)."""
        fn = self._lib.dw_init_random_quantised if quantised else self._lib.dw_init_random
        self._check(fn(self._h, C.c_uint64(seed & (2 ** 64 - 1))))

    def download_planes(self, which=_ffi.STATE_CURRENT):
        light = np.empty((self.B, self.H, self.W))
        dark = np.empty((self.B, self.H, self.W))
        self._check(self._lib.dw_download_planes(self._h, which, _ffi.ptr_d(light), _ffi.ptr_d(dark)))
        return light, dark

    def download_grid(self, L_init=0.75):
        grid = np.empty((self.B, 7, self.H, self.W))
        self._check(self._lib.dw_download_grid(self._h, float(L_init), _ffi.ptr_d(grid)))
        return grid

    def download_caches(self, L, temps=True, betas=True, growth=True, temp_effective=True):
        B, H, W = self.B, self.H, self.W
        t = np.empty((B, 3, H, W)) if temps else None
        b = np.empty((B, 3, H, W)) if betas else None
        g = np.empty((B, 2, H, W)) if growth else None
        e = np.empty((B, 1, H, W)) if temp_effective else None
        self._check(self._lib.dw_download_caches(self._h, float(L), _ffi.ptr_d(t), _ffi.ptr_d(b), _ffi.ptr_d(g),
                                           _ffi.ptr_d(e)))
        return t, b, g, e

    # -- hot path -----------------------------------------------------------------------------
    @staticmethod
    def _actions(action):
        """(b, n, 1) or (b, n) integer-valued array -> C-contiguous int32 (b, n)."""
        a = np.asarray(action)
        if a.ndim == 3:
            a = a[..., 0]
        if a.ndim != 2:
            raise ValueError(f"action must have shape (b, n, 1); got {np.shape(action)}")
        if not np.issubdtype(a.dtype, np.integer):
            ai = np.rint(a)
            if not np.array_equal(ai, a):
                raise ValueError("non-integral action codes are not supported")
            a = ai
        return np.ascontiguousarray(a, dtype=np.int32)

    def step(self, L, action=None):
        if action is None:
            self._check(self._lib.dw_step(self._h, None, 0, 0, float(L)))
        else:
            a = self._actions(action)
            self._check(self._lib.dw_step(self._h, _ffi.ptr_i(a), a.shape[0], a.shape[1], float(L)))

    def env_step(self, L, action=None):
        """step + get_obs + reward/done in one call with one synchronisation.  Returns
        (obs (B,N,7,3,3), reward (B,N,1), done (B,N,1) bool)."""
        obs = np.zeros((self.B, self.N, 7, 3, 3))
        reward = np.zeros((self.B, self.N, 1))
        done = np.zeros((self.B, self.N, 1), dtype=np.uint8)
        if action is None:
            a, ab, an = None, 0, 0
        else:
            a = self._actions(action)
            ab, an = a.shape
        self._check(self._lib.dw_env_step(self._h, _ffi.ptr_i(a), ab, an, float(L), _ffi.ptr_d(obs), _ffi.ptr_d(reward),
                                    _ffi.ptr_u8(done)))
        return obs, reward, done.astype(bool)

    def step_device_actions(self, L):
        self._check(self._lib.dw_step_device_actions(self._h, float(L)))

    def step_n(self, nsteps, L, dL, min_L, max_L, use_device_actions=False):
        Lc = C.c_double(float(L))
        self._check(self._lib.dw_step_n(self._h, int(nsteps), C.byref(Lc), float(dL), float(min_L), float(max_L),
                                  int(bool(use_device_actions))))
        return Lc.value

    def last_step_n_timing(self):
        """(ms spent in the fused step-pair launches of the last step_n call, their number, plane element bytes)."""
        ms, n, eb = C.c_float(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.dw_last_step_n_timing(self._h, C.byref(ms), C.byref(n), C.byref(eb)))
        return ms.value, n.value, eb.value

    def update_agents(self, action):
        a = self._actions(action)
        self._check(self._lib.dw_update_agents(self._h, _ffi.ptr_i(a), a.shape[0], a.shape[1]))

    def upload_actions(self, action):
        a = self._actions(action)
        if a.shape != (self.B, self.N):
            raise ValueError(f"device action buffer needs shape {(self.B, self.N)}")
        self._check(self._lib.dw_upload_actions(self._h, _ffi.ptr_i(a)))

    def download_actions(self):
        a = np.empty((self.B, self.N), dtype=np.int32)
        self._check(self._lib.dw_download_actions(self._h, _ffi.ptr_i(a)))
        return a

    def forward(self, light, dark, L, want_caches=False):
        light, dark = self._plane(light, np.float64), self._plane(dark, np.float64)
        B, H, W = self.B, self.H, self.W
        grid = np.empty((B, 7, H, W))
        if want_caches:
            t, b, g, e = np.empty((B, 3, H, W)), np.empty((B, 3, H, W)), np.empty((B, 2, H, W)), np.empty((B, 1, H, W))
        else:
            t = b = g = e = None
        self._check(self._lib.dw_forward_f64(self._h, _ffi.ptr_d(light), _ffi.ptr_d(dark), float(L), _ffi.ptr_d(grid),
                                       _ffi.ptr_d(t), _ffi.ptr_d(b), _ffi.ptr_d(g), _ffi.ptr_d(e)))
        return (grid, t, b, g, e) if want_caches else grid

    def conv3x3(self, plane, kernel):
        """Toroidal 3x3 convolution of a (B,H,W) float64 plane on the device (ref ft_convolve)."""
        x = self._plane(plane, np.float64)
        k = np.ascontiguousarray(kernel, dtype=np.float64).reshape(9)
        out = np.empty_like(x)
        self._check(self._lib.dw_conv3x3_f64(self._h, _ffi.ptr_d(x), _ffi.ptr_d(k), _ffi.ptr_d(out)))
        return out

    def stage(self, stage, planes, L=0.0, kernel=None):
        """One stage of forward() on caller data, on the device (dw_stage_f64): `planes` = the stage's input
        planes, each (B,H,W); returns the list of its output planes."""
        n_in, n_out = _ffi.STAGE_IO[stage]
        if len(planes) != n_in:
            raise ValueError(f"stage {stage} takes {n_in} planes, got {len(planes)}")
        x = np.ascontiguousarray(np.stack([self._plane(a, np.float64) for a in planes]))
        out = np.empty((n_out, self.B, self.H, self.W))
        k = None if kernel is None else np.ascontiguousarray(kernel, dtype=np.float64).reshape(9)
        self._check(self._lib.dw_stage_f64(self._h, int(stage), _ffi.ptr_d(x), _ffi.ptr_d(out), float(L),
                                           _ffi.ptr_d(k) if k is not None else None))
        return list(out)

    def get_obs(self, L_init=0.75):
        obs = np.zeros((self.B, self.N, 7, 3, 3))
        if self.B * self.N:
            self._check(self._lib.dw_get_obs(self._h, float(L_init), _ffi.ptr_d(obs)))
        return obs

    def reward_done(self):
        reward = np.zeros((self.B, self.N, 1))
        done = np.zeros((self.B, self.N, 1), dtype=np.uint8)
        if self.B * self.N:
            self._check(self._lib.dw_get_reward_done(self._h, _ffi.ptr_d(reward), _ffi.ptr_u8(done)))
        return reward, done.astype(bool)

    def reduce(self):
        out = np.zeros(self.B, dtype=_ffi.STATS_DTYPE)
        self._check(self._lib.dw_reduce(self._h, out.ctypes.data_as(C.POINTER(DwWorldStats))))
        return out

    def policy_greedy(self, argmin=False):
        self._check(self._lib.dw_policy_greedy(self._h, _ffi.POLICY_ARGMIN if argmin else _ffi.POLICY_ARGMAX))

    def policy_per_agent(self, agent_mode):
        """agent_mode (N,) of POLICY_ARGMAX / POLICY_ARGMIN / POLICY_TABLE (keep the uploaded action)."""
        m = np.ascontiguousarray(agent_mode, dtype=np.int32)
        if m.shape != (self.N,):
            raise ValueError(f"agent_mode must have shape {(self.N,)}")
        self._check(self._lib.dw_policy_per_agent(self._h, _ffi.ptr_i(m)))

    def policy_mlp(self, params, agent_begin=0, agent_end=None, L_init=0.75):
        w = np.ascontiguousarray(params, dtype=np.float64).ravel()
        end = self.N if agent_end is None else int(agent_end)
        self._check(self._lib.dw_policy_mlp(self._h, _ffi.ptr_d(w), int(w.size), int(agent_begin), end, float(L_init)))

    def policy_mlp_population(self, params, world_member, agent_begin=0, agent_end=None, L_init=0.75):
        """params (P,1808); world_member (B,) int: which parameter set each world's agents use."""
        w = np.ascontiguousarray(params, dtype=np.float64)
        if w.ndim != 2 or w.shape[1] != 1808:
            raise ValueError("params must have shape (n_members, 1808)")
        m = np.ascontiguousarray(world_member, dtype=np.int32)
        if m.shape != (self.B,):
            raise ValueError(f"world_member must have shape {(self.B,)}")
        end = self.N if agent_end is None else int(agent_end)
        self._check(self._lib.dw_policy_mlp_population(self._h, _ffi.ptr_d(w), int(w.shape[0]), _ffi.ptr_i(m),
                                                 int(agent_begin), end, float(L_init)))

    def lifespan_reset(self):
        self._check(self._lib.dw_lifespan_reset(self._h))

    def lifespan_accumulate(self, threshold_k=5):
        self._check(self._lib.dw_lifespan_accumulate(self._h, int(threshold_k)))

    def lifespan_download(self):
        done_at = np.zeros(self.B, dtype=np.int32)
        agents = np.zeros((self.B, self.N, 1), dtype=np.int32)
        alive = C.c_int32(0)
        self._check(self._lib.dw_lifespan_download(self._h, _ffi.ptr_i(done_at), _ffi.ptr_i(agents) if self.N else None,
                                             C.byref(alive)))
        return done_at, agents, alive.value

    def snapshot_save(self, slot=0):
        """Device-side copy of the current state (planes, the retained previous state, agents, reductions) into one of
        the two slots."""
        self._check(self._lib.dw_snapshot_save_slot(self._h, int(slot)))

    def snapshot_restore(self, slot=0):
        self._check(self._lib.dw_snapshot_restore_slot(self._h, int(slot)))

    def run_episode(self, L_schedule, policy_mode, use_table=None, table=None, threshold_k=5, world_flags=True,
                    reuse_buffers=False):
        """K device-resident steps (one launch for H*W <= 4096, back-to-back launches otherwise).
        Returns (world_alive (K,B) bool, agent_ok (K,B,N) bool).  `table` entries: 0..8 an action, -1 / -2 the
        greedy / anti-greedy choice of that agent at that step.  world_flags=False: the per-step world
        reductions are not needed (world_alive is returned as None), which lets wide grids run step PAIRS in
        one fused launch with the agents' in-between step patched in (dw_agents_fused.hpp)."""
        Ls = np.ascontiguousarray(L_schedule, dtype=np.float64)
        K = Ls.shape[0]
        ut = None if use_table is None else np.ascontiguousarray(use_table, dtype=np.uint8)
        tb = None if table is None else np.ascontiguousarray(table, dtype=np.int8)
        if ut is not None and ut.shape != (K,):
            raise ValueError("use_table must have shape (K,)")
        if tb is not None and tb.shape != (K, self.B, self.N):
            raise ValueError(f"table must have shape {(K, self.B, self.N)}")
        # (the library writes every flag as 0 / 1: the uint8 arrays are returned as bool VIEWS, no second pass)
        # reuse_buffers=True: the flag arrays are this engine's own and are OVERWRITTEN by the next such call (a fresh
        # 256 KB array per chunk is a fresh mmap: ~60 page faults while the library copies into it) - for loops that
        # consume a chunk's flags before they run the next chunk
        if reuse_buffers:
            cache = self.__dict__.setdefault("_flag_buffers", {})
            key = (K, bool(world_flags))
            if key not in cache:
                cache.clear()
                cache[key] = (np.empty((K, self.B), dtype=np.uint8) if world_flags else None,
                              np.empty((K, self.B, self.N), dtype=np.uint8))
            alive, ok = cache[key]
        else:
            alive = np.empty((K, self.B), dtype=np.uint8) if world_flags else None
            ok = np.empty((K, self.B, self.N), dtype=np.uint8)
        self._check(self._lib.dw_run_episode(
            self._h, K, _ffi.ptr_d(Ls), int(policy_mode), _ffi.ptr_u8(ut),
            None if tb is None else tb.ctypes.data_as(C.POINTER(C.c_int8)), int(threshold_k), _ffi.ptr_u8(alive),
            _ffi.ptr_u8(ok) if self.N else None))
        return (None if alive is None else alive.view(np.bool_)), ok.view(np.bool_)

    def run_episode_mlp(self, L_schedule, params, member_a=None, member_b=None, split=None, L_init=0.75,
                        reuse_buffers=False, n_members=None):
        """K device-resident steps with MLP policies: agents [0, split) of world b use parameter set
        member_a[b], agents [split, N) member_b[b].  Returns (reward (K,B,N,1) float64, done (K,B,N,1) bool)
        exactly as K calls of env.step would (ref step :486-492).

        `params=None`: the parameter sets of the last call that passed them are still on the device and are used again
        (pass `n_members`).  `reuse_buffers=True`: the returned arrays are views of page-locked buffers owned by this
        engine (no staging copy on the way back) and are OVERWRITTEN by the next call with reuse_buffers=True - for
        harnesses that consume a chunk's rewards before they run the next one.  `reuse_buffers=1` selects a SECOND such
        buffer (True is buffer 0): a harness that runs chunk c + 1 while it still reads chunk c's rewards alternates."""
        Ls = np.ascontiguousarray(L_schedule, dtype=np.float64)
        K = Ls.shape[0]
        if params is None:
            if n_members is None:
                raise ValueError("params=None needs n_members (the sets uploaded by an earlier call)")
            w, nm = None, int(n_members)
        else:
            w = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, 1808)
            nm = w.shape[0]
        ma = None if member_a is None else np.ascontiguousarray(member_a, dtype=np.int32)
        mb = None if member_b is None else np.ascontiguousarray(member_b, dtype=np.int32)
        for m in (ma, mb):
            if m is not None and m.shape != (self.B,):
                raise ValueError(f"member maps must have shape {(self.B,)}")
        split = self.N // 2 if split is None else int(split)
        n = K * self.B * self.N
        if reuse_buffers is not False and reuse_buffers is not None and n:
            buf = self._pinned_block(9 * n, 0 if reuse_buffers is True else int(reuse_buffers))
            reward = np.frombuffer(buf, dtype=np.float64, count=n).reshape(K, self.B, self.N, 1)
            done = np.frombuffer(buf, dtype=np.uint8, count=n, offset=8 * n).reshape(K, self.B, self.N, 1)
        else:
            reward = np.empty((K, self.B, self.N, 1))              # (every element is written by the library; done: 0 / 1)
            done = np.empty((K, self.B, self.N, 1), dtype=np.uint8)
        self._check(self._lib.dw_run_episode_mlp(self._h, K, _ffi.ptr_d(Ls), _ffi.ptr_d(w), nm, _ffi.ptr_i(ma),
                                           _ffi.ptr_i(mb), split, float(L_init), _ffi.ptr_d(reward), _ffi.ptr_u8(done)))
        return reward, done.view(np.bool_)

    def _pinned_block(self, nbytes, index=0):
        """Page-locked host buffer number `index` of at least nbytes owned by this engine (grown geometrically, freed by
        close())."""
        blocks = self.__dict__.setdefault("_pinned", {})
        have = blocks.get(index)
        if have is None or have[1] < nbytes:
            if have is not None:
                self._lib.dw_pinned_free(have[0])
                del blocks[index]
            size = max(int(nbytes), 2 * (have[1] if have else 0), 1 << 20)
            ptr = C.c_void_p()
            self._check(self._lib.dw_pinned_alloc(size, C.byref(ptr)))
            blocks[index] = (ptr, size, (C.c_ubyte * size).from_address(ptr.value))
        return blocks[index][2]

    # -- plumbing -----------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr: int):
        self._check(self._lib.dw_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def sync(self):
        self._check(self._lib.dw_sync(self._h))

    def timer_start(self):
        self._check(self._lib.dw_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        self._check(self._lib.dw_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def device_planes(self, which=_ffi.STATE_CURRENT):
        lp, dp = C.c_void_p(), C.c_void_p()
        self._check(self._lib.dw_device_planes(self._h, which, C.byref(lp), C.byref(dp)))
        return lp.value, dp.value

    def kernel_info(self) -> str:
        buf = C.create_string_buffer(512)
        self._check(self._lib.dw_kernel_info(self._h, buf, 512))
        return buf.value.decode()

    def audit_tie_bound(self, L):
        """(max |gq32-gq64| in quanta, max error/bound, flagged cell-values, audited cell-values)."""
        out = np.zeros(4)
        self._check(self._lib.dw_audit_tie_bound(self._h, float(L), _ffi.ptr_d(out)))
        return float(out[0]), float(out[1]), int(out[2]), int(out[3])

    def last_fixup_count(self) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.dw_last_fixup_count(self._h, C.byref(v)))
        return int(v.value)
