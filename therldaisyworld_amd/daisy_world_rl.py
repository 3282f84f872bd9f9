"""Drop-in ``RLDaisyWorld`` whose grid update runs on an MI355X through libdaisyworld_hip.so.

Mirrors the public surface of the reference class (``/root/reference/daisy/daisy_world_rl.py``):
constructor kwargs (:24-29,:44,:62,:79), the mutable attributes callers assign before ``reset()``
(``batch_size``, ``n_agents``, albedos, ``min_L``/``max_L``/``ramp_period``, ``agent_gamma``, ``dt``,
``q2``...), ``reset()`` (:327), ``step(action=None)`` (:475), ``forward(grid)`` (:434),
``get_obs`` (:246), ``update_agents`` (:181), ``update_L`` (:463), ``set_use_microclimate`` (:85) and
the config helpers (:94-171), and the attributes callers read (``grid``, ``temp``, ``beta``, ``growth``,
``dead_temp``, ``agent_indices``, ``agent_states``, ``L``, ``dL``, ``step_count``...).

What stays on the host: the legacy NumPy RNG draws (same global stream, same call order as the
reference, so the same ``np.random.seed`` gives the same worlds), the scalar luminosity recurrence,
and bookkeeping.  Everything per-cell or per-agent runs in HIP kernels - also the stand-alone stage methods
(`calculate_*`: one float64 device call each, dw_stage_f64); there is no CPU compute path, and construction
fails if the HIP library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from . import _ffi
from .engine import Engine, default_params, mask_bits
from .helpers import query_kwargs
from .nn.functional import make_neighborhood


class _Mirror:
    """A host copy of device state handed to the caller, plus a snapshot to detect in-place edits."""

    def __init__(self, array):
        self.array = array
        self.snapshot = array.copy()

    def dirty(self):
        return not np.array_equal(self.array, self.snapshot)


class RLDaisyWorld:

    def __init__(self, **kwargs):
        self.ch = 7
        self.batch_size = 32                                   # ref :20 (kwarg is ignored there too)
        self.kr = query_kwargs("kr", 1, **kwargs)
        self.neighborhood_mode = query_kwargs("neighborhood_mode", "von_neumann", **kwargs)
        self.neighborhood = make_neighborhood(self.kr, self.neighborhood_mode)
        self.dim = kwargs["grid_dimension"] if "grid_dimension" in kwargs.keys() else 16

        self.p = 1.00
        self.g = 0.003265
        self.S = 1000.0
        self.sigma = 5.67e-8
        self.gamma = 0.25
        self.q = 0.2 * self.S / self.sigma
        self.use_microclimate = True
        self.collision_mode = query_kwargs("collision_mode", 0, **kwargs)
        self.q2 = self.q / 8. if self.use_microclimate else 0.0
        self.Toptim = 295.5
        self.dt = 1.0
        self.ddL = 0.
        self.agent_gamma = 0.05
        self.max_L = 1.5
        self.min_L = 0.75
        self.initial_L = self.min_L
        self.ramp_period = kwargs["ramp_period"] if "ramp_period" in kwargs.keys() else 512
        self.ramp_up_down = False
        self.albedo_bare = 0.5
        self.albedo_light = 0.75
        self.albedo_dark = 0.25
        self.temp_optimal = 295.5
        self.food_chain_penalty = 0.5
        self.initial_al = 0.2
        self.initial_ad = 0.2
        self.light_proportion = 0.33
        self.dark_proportion = 0.33
        self.n_agents = query_kwargs("n_agents", 4, **kwargs)

        # extensions (unknown to the reference, which ignores unknown kwargs)
        self.precision = query_kwargs("precision", "exact", **kwargs)   # "exact" | "fast" | "f64"
        self.device = query_kwargs("device", int(os.environ.get("LOCAL_RANK", "0")), **kwargs)
        self.world_offset = query_kwargs("world_offset", 0, **kwargs)

        self._engine = None
        self._shape = None
        self._grid_m = None          # _Mirror of the materialised 7-channel grid
        self._idx_m = None           # _Mirror of agent_indices (B,N,2) int64
        self._st_m = None            # _Mirror of agent_states (B,N,1) float64
        self._agents_on_device = False
        self._caches = {}
        self._cache_src = None       # None -> device state; or (light, dark, L) of a forward(grid) call
        self._L_pass = self.min_L    # luminosity of the most recent physics pass

        self.initialize_neighborhood()
        self.initialize_agents()
        self.reset()

    # ------------------------------------------------------------------------------------------
    # configuration helpers (ref :85-171) — host-only
    # ------------------------------------------------------------------------------------------
    def set_use_microclimate(self, use_microclimate=True):
        self.use_microclimate = use_microclimate
        self.q2 = self.q / 8. if self.use_microclimate else 0.0

    _CONFIG_KEYS = ("max_L", "min_L", "initial_L", "ramp_period", "dL", "p", "g", "S", "sigma", "gamma",
                    "albedo_bare", "albedo_light", "albedo_dark", "temp_optimal", "light_proportion",
                    "dark_proportion", "initial_al", "initial_ad", "n_agents", "agent_gamma")

    def make_config(self):
        return {k: getattr(self, k) for k in self._CONFIG_KEYS}

    def save_config(self, filepath=None):
        if filepath is None:
            filepath = os.path.join("results", "default_model_config.json")
        with open(filepath, "w") as f:
            json.dump(self.make_config(), f)

    def _apply_config(self, config):
        for k in self._CONFIG_KEYS:
            setattr(self, k, config[k])

    def load_config(self, filepath=None):
        if filepath is None:
            filepath = os.path.join("results", "default_model_config.json")
        with open(filepath, "r") as f:
            return json.load(f)

    def restore_config(self, filepath=None):
        self._apply_config(self.load_config(filepath))

    def initialize_neighborhood(self):
        """ref :265-283.  The convolution kernels live in the HIP library (dw_api.hip make_f64);
        the arrays are kept here only because they are public attributes of the reference."""
        self.n_daisies = 2
        k = np.ones((1, 1, 3, 3)) * np.exp(-1)
        k[:, :, 1, 1] = 1.0
        k[:, :, 0::2, 0::2] = np.exp(-2)
        self.daisy_kernel = k / k.sum()
        self.local_albedo_kernel = np.zeros((1, 1, 3, 3))
        self.local_albedo_kernel[:, :, 1, 1] = 1.0
        self.adjacent_albedo_kernel = np.ones((1, 1, 3, 3)) / 8.
        self.adjacent_albedo_kernel[:, :, 1, 1] = 0.

    # ------------------------------------------------------------------------------------------
    # engine management
    # ------------------------------------------------------------------------------------------
    _PARAM_FLOATS = ("p", "g", "S", "sigma", "gamma", "q", "q2", "dt", "albedo_bare", "albedo_light", "albedo_dark",
                     "temp_optimal", "agent_gamma", "food_chain_penalty", "initial_al", "initial_ad", "light_proportion",
                     "dark_proportion")

    def _param_key(self):
        """Everything _params() reads, as one tuple: the reference reads its attributes live on every call, so the class
        looks at them on every call too - but builds and pushes a dw_params only when one of them changed."""
        return (self.batch_size, self.dim, self.n_agents, self.device, self.precision, self.collision_mode, self.world_offset,
                self.neighborhood.tobytes(), *[getattr(self, name) for name in self._PARAM_FLOATS])

    def _params(self):
        p = default_params(int(self.batch_size), int(self.dim), int(self.dim), int(self.n_agents))
        p.device = int(self.device)
        p.precision = _ffi.PRECISION[self.precision]
        p.obs_mask = mask_bits(self.neighborhood)
        p.collision_mode = int(self.collision_mode)
        p.world_offset = int(self.world_offset)
        for name in ("p", "g", "S", "sigma", "gamma", "q", "q2", "dt", "albedo_bare", "albedo_light",
                     "albedo_dark", "temp_optimal", "agent_gamma", "food_chain_penalty", "initial_al",
                     "initial_ad", "light_proportion", "dark_proportion"):
            setattr(p, name, float(getattr(self, name)))
        return p

    def _ensure_engine(self):
        """(Re)create the device handle when the caller changed batch_size / dim / n_agents, and push
        the current constants (the reference reads its attributes live on every call)."""
        key = self._param_key()
        if self._engine is not None and key == getattr(self, "_pushed_key", None):
            return self._engine                              # nothing changed since the last push
        shape = (int(self.batch_size), int(self.dim), int(self.n_agents), int(self.device))
        p = self._params()
        self._pushed_key = key
        if self._engine is None or shape != self._shape:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(p)
            self._shape = shape
            self._agents_on_device = False
        else:
            self._engine.set_params(p)
        return self._engine

    def _sync_to_device(self):
        """Push host-side edits (assignment or in-place mutation of grid / agent arrays)."""
        eng = self._engine
        if self._grid_m is not None and self._grid_m.dirty():
            g = self._grid_m.array
            eng.upload_state(g[:, 1], g[:, 2])
            self._grid_m.snapshot = g.copy()          # the edited array stays the environment's grid (as in the
            self._caches = {}                         # reference, where it IS the state) until the next step
        if self._idx_m is not None and (not self._agents_on_device or self._idx_m.dirty() or self._st_m.dirty()):
            eng.upload_agents(self._idx_m.array, self._st_m.array[..., 0])
            self._idx_m.snapshot = self._idx_m.array.copy()
            self._st_m.snapshot = self._st_m.array.copy()
            self._agents_on_device = True

    def _invalidate(self):
        self._grid_m = None
        self._caches = {}
        self._cache_src = None
        self._idx_m = None
        self._st_m = None

    # ------------------------------------------------------------------------------------------
    # public state attributes
    # ------------------------------------------------------------------------------------------
    @property
    def grid(self):
        if self._grid_m is None:
            self._sync_to_device()
            self._grid_m = _Mirror(self._engine.download_grid(self._L_pass))
        return self._grid_m.array

    @grid.setter
    def grid(self, value):
        # the reference simply rebinds the attribute: reads return the assigned array (all 7 channels as
        # given) and later in-place edits of it are edits of the environment's state
        if not (isinstance(value, np.ndarray) and value.dtype == np.float64):
            value = np.array(value, dtype=np.float64)
        self._ensure_engine().upload_state(value[:, 1], value[:, 2])
        self._grid_m = _Mirror(value)
        self._caches = {}
        self._cache_src = None

    def _pull_agents(self):
        if self._idx_m is None:
            idx, st = self._engine.download_agents()
            self._idx_m = _Mirror(idx.astype(np.int64))
            self._st_m = _Mirror(st[..., None].copy())

    @property
    def agent_indices(self):
        self._pull_agents()
        return self._idx_m.array

    @agent_indices.setter
    def agent_indices(self, value):
        self._pull_agents()
        self._idx_m.array = np.array(value, dtype=np.int64)
        self._agents_on_device = False

    @property
    def agent_states(self):
        self._pull_agents()
        return self._st_m.array

    @agent_states.setter
    def agent_states(self, value):
        self._pull_agents()
        self._st_m.array = np.array(value, dtype=np.float64)
        self._agents_on_device = False

    # side-effect caches of the last physics pass (ref :345-347,373,404,415-419)
    def _cache(self, name):
        if not self._caches:
            if self._cache_src is None:
                self._sync_to_device()
                t, b, g, e = self._engine.download_caches(self._L_pass)
            else:
                light, dark, L = self._cache_src
                _, t, b, g, e = self._engine.forward(light, dark, L, want_caches=True)
            self._caches = {"temp": t[:, 0:1], "temp_light": t[:, 1:2], "temp_dark": t[:, 2:3],
                            "beta": b[:, 0:1], "beta_l": b[:, 1:2], "beta_d": b[:, 2:3], "growth": g,
                            "temp_effective": e}
        return self._caches[name]

    temp = property(lambda self: self._cache("temp"))
    temp_light = property(lambda self: self._cache("temp_light"))
    temp_dark = property(lambda self: self._cache("temp_dark"))
    temp_effective = property(lambda self: self._cache("temp_effective"))
    beta = property(lambda self: self._cache("beta"))
    beta_l = property(lambda self: self._cache("beta_l"))
    beta_d = property(lambda self: self._cache("beta_d"))
    growth = property(lambda self: self._cache("growth"))

    @property
    def dead_temp(self):
        """ref :407-408,:416 — temperature of a lifeless planet at the luminosity of the last pass."""
        return np.array([((self.S * self._L_pass * (1 - self.albedo_bare)) / self.sigma) ** (1 / 4)])

    # ------------------------------------------------------------------------------------------
    # initialisation (ref :173-179, :285-338): host RNG in the reference's call order
    # ------------------------------------------------------------------------------------------
    def initialize_agents(self):
        idx = np.random.randint(self.dim, size=(self.batch_size, self.n_agents, 2))
        st = np.ones((self.batch_size, self.n_agents, 1))
        self._idx_m = _Mirror(idx.astype(np.int64))
        self._st_m = _Mirror(st)
        self._agents_on_device = False

    @staticmethod
    def _legacy_rand(*shape):
        """`np.random.rand(*shape)` on the GLOBAL legacy stream - through the host helper when it is built
        (include/daisyworld_host.h: the same numbers, the generator left in the same state, 3-4x faster: the draw is the
        largest single item of a reset of a big ensemble), through NumPy otherwise."""
        n = int(np.prod(shape))
        host = _ffi.load_host() if n >= 4096 else None
        if host is None:
            return np.random.rand(*shape)
        state = np.random.get_state()
        if state[0] != "MT19937":
            return np.random.rand(*shape)
        key = np.array(state[1], dtype=np.uint32)            # (a copy: the state tuple's array is NumPy's)
        pos = C.c_int32(int(state[2]))
        out = np.empty(n)
        rc = host.dw_mt19937_random_sample(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                           out.ctypes.data_as(C.POINTER(C.c_double)), n)
        if rc != 0:                                          # (nothing consumed: the global state was not touched)
            return np.random.rand(*shape)
        np.random.set_state((state[0], key, pos.value, state[3], state[4]))
        return out.reshape(shape)

    def draw_initial_cover(self):
        """The RNG half of ref initialize_grid :287-302 (dark drawn first, then light)."""
        B, d = self.batch_size, self.dim
        dark_probability = self._legacy_rand(B, 2, d, d)
        light_probability = self._legacy_rand(B, 2, d, d)
        # ref: 1.0 * (p[:, 0] < proportion) * initial * p[:, 1] - the same products (multiplication commutes, the mask is
        # exactly 0 or 1) in two passes instead of four
        dark = np.multiply(dark_probability[:, 1], self.initial_ad)
        dark *= dark_probability[:, 0] < self.dark_proportion
        light = np.multiply(light_probability[:, 1], self.initial_al)
        light *= light_probability[:, 0] < self.light_proportion
        return light, dark

    def initialize_grid(self):
        light, dark = self.draw_initial_cover()
        eng = self._ensure_engine()
        eng.upload_state(light, dark)
        self._L_pass = self.L
        self._grid_m = None
        self._caches = {}
        self._cache_src = None

    def reset(self):
        self.L = self.min_L
        self.dL = (self.max_L - self.min_L) / self.ramp_period
        self.step_count = 0
        self.initialize_grid()
        self.initialize_agents()
        return self.get_obs(self.agent_indices)

    def reset_synthetic(self, seed=0):
        """Extension: device-side initial state (Philox keyed by seed / global world id / cell) with
        the distribution of initialize_grid / initialize_agents — for grids too large to draw and
        upload from the host.  Not stream-compatible with np.random."""
        self.L = self.min_L
        self.dL = (self.max_L - self.min_L) / self.ramp_period
        self.step_count = 0
        eng = self._ensure_engine()
        eng.init_random(seed)
        self._L_pass = self.L
        self._invalidate()
        self._agents_on_device = True
        return None

    # ------------------------------------------------------------------------------------------
    # the path
    # ------------------------------------------------------------------------------------------
    def get_obs(self, agent_indices=None):
        """ref :246-263: the 3x3 wrap-around patches of ALL channels of self.grid around the given positions,
        times the neighbourhood mask.  While the device state is the only copy (no grid handed out since the
        last step / reset) and the positions are the environment's own, the observe kernel builds them
        directly; otherwise they are what the reference computes: slices of the host grid, whose channels
        may be stale or edited by the caller."""
        eng = self._ensure_engine()
        own = agent_indices is None or (self._idx_m is not None and (
            agent_indices is self._idx_m.array or np.array_equal(agent_indices, self._idx_m.array)))
        if own and self._grid_m is None:
            self._sync_to_device()
            return eng.get_obs(self._L_pass)
        grid = self.grid
        idx = np.asarray(self.agent_indices if agent_indices is None else agent_indices).astype(np.int64)
        B, N = idx.shape[:2]
        off = np.arange(-1, 2)
        rows = (idx[..., 0, None] + off) % self.dim                       # (B,N,3)
        cols = (idx[..., 1, None] + off) % self.dim
        obs = grid[np.arange(B)[:, None, None, None, None], np.arange(self.ch)[None, None, :, None, None],
                   rows[:, :, None, :, None], cols[:, :, None, None, :]]
        return obs * self.neighborhood

    def update_agents(self, action):
        """ref :181-244.  Movement and grazing run on the device; with collision_mode == 1 the
        collision pass (:220-242) follows on the host because it consumes the legacy RNG stream."""
        grid = self.grid                # the reference mutates self.grid in place (:214-216): only the covers
        self._update_agents_on_device(action)   # of grazed cells change, other channels keep their stale values
        light, dark = self._engine.download_planes()
        grid[:, 1] = light
        grid[:, 2] = dark
        self._grid_m = _Mirror(grid)

    def _update_agents_on_device(self, action):
        eng = self._ensure_engine()
        self._sync_to_device()
        eng.update_agents(action)
        self._invalidate()
        if self.collision_mode == 1:
            self._collision_pass()

    def _collision_pass(self):
        """ref :220-244 as implemented: per world, per cell in row-major order, if more than one agent
        sits on the cell, the resident with the largest (state + 0.01*U) gains food_chain_penalty times
        the summed states of the other residents; the losers keep their state (the reference zeroes a
        copy); one np.random.rand(1,N,1) draw per multiply-occupied cell; then clip to [0,1]."""
        idx, st = self._engine.download_agents()
        st = st[..., None].copy()                              # (B,N,1) float64, before the clip
        for bb in range(idx.shape[0]):
            cells = idx[bb, :, 0].astype(np.int64) * self.dim + idx[bb, :, 1]
            uniq, counts = np.unique(cells, return_counts=True)
            for cell in uniq[counts > 1]:                      # np.unique sorts: row-major scan order
                residents = (cells == cell)[None, :, None]
                temp_values = 1.0 * st[bb:bb + 1] + 0.01 * np.random.rand(*st[bb:bb + 1].shape)
                winner_value = np.max(temp_values[residents])
                winner_index = temp_values == winner_value
                eat = st[bb:bb + 1][residents][temp_values[residents] != winner_value].sum()
                st[bb:bb + 1][winner_index] += self.food_chain_penalty * eat
        st = np.clip(st, 0., 1.)
        self._engine.upload_agents(idx, st[..., 0])
        self._invalidate()

    def forward(self, grid):
        """ref :434-461 — stateless w.r.t. the covers: evaluates the physics pass on `grid` in float64
        on the device and returns the new 7-channel grid; like the reference it rewrites channel 0
        of its argument in place (:381) and refreshes temp/beta/growth."""
        eng = self._ensure_engine()
        self._sync_to_device()
        grid[:, 0] = self.p - grid[:, 1] - grid[:, 2]
        light = np.ascontiguousarray(grid[:, 1], dtype=np.float64)
        dark = np.ascontiguousarray(grid[:, 2], dtype=np.float64)
        new_grid = eng.forward(light, dark, self.L)
        self._L_pass = self.L
        self._caches = {}
        self._cache_src = (light, dark, self.L)
        return new_grid

    # ------------------------------------------------------------------------------------------
    # the stages of forward() as stand-alone methods (ref :340-432).  forward() / step() never call these (the
    # device evaluates the fused map); each is ONE device call on the caller's arrays (dw_stage_f64: float64, the
    # reference's staging), and like the reference's stages they overwrite the temp / beta / growth attributes.
    # ------------------------------------------------------------------------------------------
    def _stage(self, stage, planes, kernel=None):
        """Run one stage on the device: `planes` are (B,H,W) views; results come back as (B,1,H,W) arrays."""
        outs = self._ensure_engine().stage(stage, [np.asarray(x, dtype=np.float64) for x in planes], self.L,
                                           None if kernel is None else np.asarray(kernel)[0, 0])
        return [o[:, None] for o in outs]

    @staticmethod
    def _channels(x, n):
        """The first n channels of a (B,>=n,H,W) array, or n separate (B,1,H,W) / (B,H,W) arrays, as (B,H,W) views."""
        x = np.asarray(x)
        return [x[:, c] for c in range(n)]

    def calculate_albedo(self, groundcover):
        """ref :377-394 -> (local_albedo, adjacent_albedo), (B,1,H,W) each; channel 0 of the argument is rewritten
        with the bare fraction the pass works with (ref :381)."""
        local, adjacent, bare = self._stage(_ffi.STAGE_ALBEDO, self._channels(groundcover, 3), self.adjacent_albedo_kernel)
        groundcover[:, 0] = bare[:, 0]
        return local, adjacent

    def calculate_daisy_density(self, local_daisies):
        """ref :423-432 -> (B,2,H,W): light and dark cover under the daisy kernel."""
        return np.concatenate(self._stage(_ffi.STAGE_DENSITY, self._channels(local_daisies, 2), self.daisy_kernel), axis=1)

    def calculate_temperature(self, local_albedo, adjacent_albedo):
        """ref :396-421 -> (temp, temp_light, temp_dark); also sets temp_effective and moves dead_temp to this
        pass's luminosity."""
        planes = [np.asarray(local_albedo)[:, 0], np.asarray(adjacent_albedo)[:, 0]]
        effective, temp, temp_light, temp_dark = self._stage(_ffi.STAGE_TEMPERATURE, planes)
        self._L_pass = self.L
        self._stage_caches(temp_effective=effective, temp=temp, temp_light=temp_light, temp_dark=temp_dark)
        return temp, temp_light, temp_dark

    def calculate_growth_rate(self, temp, temp_l, temp_d):
        """ref :340-348 -> (beta, beta_l, beta_d)."""
        betas = self._stage(_ffi.STAGE_GROWTH_RATE, [np.asarray(t)[:, 0] for t in (temp, temp_l, temp_d)])
        self._stage_caches(beta=betas[0], beta_l=betas[1], beta_d=betas[2])
        return tuple(betas)

    def calculate_growth(self, beta, beta_l, beta_d, daisy_density):
        """ref :350-375 -> (B,2,H,W) growth of light and dark; `beta` is accepted and unused, as in the reference
        (the branch that reads it is dead code there, :362-364)."""
        planes = [np.asarray(beta_l)[:, 0], np.asarray(beta_d)[:, 0], *self._channels(daisy_density, 2)]
        growth = np.concatenate(self._stage(_ffi.STAGE_GROWTH, planes), axis=1)
        self._stage_caches(growth=growth)
        return growth

    def _stage_caches(self, **values):
        """The stand-alone stages overwrite the side-effect caches one by one, as the reference's do: the others
        keep the values of the last physics pass (materialised first), or stay unset when there has been none."""
        if not self._caches:
            try:
                self._cache("temp")
            except _ffi.DaisyHipError as e:
                if e.code != _ffi.DW_ESTATE:                  # "no state yet" is the only expected reason
                    raise
                self._caches = {}
        self._caches = dict(self._caches, **values)

    def update_L(self, L):
        """ref :463-473 (host float64 scalar)."""
        self.step_count += 1
        if self.ramp_up_down and self.step_count % self.ramp_period == 0:
            self.dL *= -1
            self.min_L -= self.ddL
            self.max_L += self.ddL
        L += self.dL
        return max([min([L, self.max_L]), self.min_L])

    def step(self, action=None):
        """ref :475-497."""
        eng = self._ensure_engine()
        self._sync_to_device()
        if action is None and self.n_agents:
            action = np.zeros((self.batch_size, self.n_agents, 1))
        if self.collision_mode == 1 and action is not None:
            self._update_agents_on_device(action)             # device move/graze + host collision pass
            action = None
        if self.n_agents:
            obs, reward, _ = eng.env_step(self.L, action)        # one call, one synchronisation
            self._L_pass = self.L
            self._invalidate()
        else:
            eng.step(self.L, action)
            self._L_pass = self.L
            self._invalidate()
            obs = eng.get_obs(self._L_pass)
            s = eng.reduce()
            reward = np.stack([s["sum_light_k"] > 0, s["sum_dark_k"] > 0], axis=-1)
        reward = reward * (reward > 0)
        done = reward < 0.1
        info = {}
        self.L = self.update_L(self.L)
        return obs, reward, done, info

    def __call__(self, grid):
        pass

    # extension: release the device handle explicitly
    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None
