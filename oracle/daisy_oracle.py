"""CPU oracle (NumPy, float64) for the RLDaisyWorld grid-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or
executed by the product package ``therldaisyworld_amd``; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and there only as the checker.

What this is
------------
A literal restatement, in plain NumPy float64, of the algorithm of the reference
``/root/reference/daisy/daisy_world_rl.py`` (class ``RLDaisyWorld``) and of the
two helpers it leans on (``daisy/nn/functional.py``: ``ft_convolve`` and
``make_neighborhood``) plus the scripted policy ``daisy/agents/greedy.py``.
It keeps the reference's *structure* (five separate 3x3 toroidal convolutions,
separate albedo / temperature / growth stages, the NCHW 7-channel float64 grid,
the python-ordered agent loop) so that it checks the fused algebra used by the
HIP kernels instead of sharing it.  The only deliberate difference is that the
FFT-based circular convolution is replaced by the mathematically identical
direct 9-tap toroidal stencil (``np.roll``); the 3-decimal quantiser at the end
of ``forward`` absorbs the ~1e-15 FFT noise, so light/dark planes come out
bit-identical (pinned by tests/golden, see below).

Pinning
-------
Parity is pinned by golden vectors generated in the build container by importing
the reference itself (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``):
``tests/test_oracle_golden.py`` checks every function here against them.

Every function cites the reference lines it follows as ``ref: file:line``.
"""
from __future__ import annotations

import numpy as np

# channel indices of the reference grid (ref: daisy_world_rl.py:18,310-312,446-450)
CH_BARE, CH_LIGHT, CH_DARK, CH_TEMP, CH_TEMP_LIGHT, CH_TEMP_DARK, CH_UNUSED = range(7)
N_CH = 7


# ----------------------------------------------------------------------------------------------
# neighbourhood masks  (ref: daisy/nn/functional.py:51-103)
# ----------------------------------------------------------------------------------------------
def neighborhood_mask(radius: int = 1, mode: str = "von_neumann") -> np.ndarray:
    """(2r+1)x(2r+1) 0/1 mask.  ref: nn/functional.py:51-63 (L1 ball), :65-77 (L-inf ball),
    :79-90 (L2 ball), :93-103 (dispatch; unknown mode -> von Neumann)."""
    ax = np.arange(-radius, radius + 1)
    cc, rr = np.meshgrid(ax, ax)
    if mode == "moore":
        dist = np.maximum(np.abs(cc), np.abs(rr))
    elif mode == "circular":
        dist = np.sqrt(cc ** 2 + rr ** 2)
    else:  # "von_neumann" and anything unknown
        dist = np.abs(cc) + np.abs(rr)
    return (dist <= radius).astype(np.float64)


# ----------------------------------------------------------------------------------------------
# toroidal 3x3 convolution  (ref: daisy/nn/functional.py:12-49)
# ----------------------------------------------------------------------------------------------
def toroidal_conv3x3(x: np.ndarray, k: np.ndarray) -> np.ndarray:
    """True (flipped-kernel) circular convolution over the last two axes:
    ``out[i,j] = sum_{a,b} k[a,b] * x[i-(a-1), j-(b-1)]``.

    ref: ``ft_convolve`` nn/functional.py:12-49 computes exactly this through
    fftshift/fft2/ifft2 (verified numerically by tests/golden G6 with an asymmetric kernel).
    ``k`` is the trailing 3x3 of the reference's (1,1,3,3) kernels."""
    k = np.asarray(k, dtype=np.float64).reshape(3, 3)
    out = np.zeros_like(x, dtype=np.float64)
    for a in range(3):
        for b in range(3):
            if k[a, b] != 0.0:
                out += k[a, b] * np.roll(x, shift=(a - 1, b - 1), axis=(-2, -1))
    return out


def daisy_kernel() -> np.ndarray:
    """ref: daisy_world_rl.py:270-273 — {1 centre, e^-1 edges, e^-2 corners}, normalised."""
    k = np.ones((3, 3)) * np.exp(-1)
    k[1, 1] = 1.0
    k[0::2, 0::2] = np.exp(-2)
    return k / k.sum()


def adjacent_albedo_kernel() -> np.ndarray:
    """ref: daisy_world_rl.py:280-281 — 1/8 on the 8 Moore neighbours, 0 at the centre."""
    k = np.ones((3, 3)) / 8.0
    k[1, 1] = 0.0
    return k


# ----------------------------------------------------------------------------------------------
# parameters  (ref: daisy_world_rl.py:15-83)
# ----------------------------------------------------------------------------------------------
class Params:
    """Plain bag of the reference's constructor state (ref: daisy_world_rl.py:18-79)."""

    def __init__(self, grid_dimension=16, n_agents=4, batch_size=32, ramp_period=512,
                 kr=1, neighborhood_mode="von_neumann", collision_mode=0):
        self.ch = N_CH
        self.batch_size = batch_size            # ref :20 (hard-wired 32 in the reference ctor)
        self.kr = kr
        self.neighborhood_mode = neighborhood_mode
        self.dim = grid_dimension               # ref :29
        self.p = 1.0
        self.g = 0.003265
        self.S = 1000.0
        self.sigma = 5.67e-8
        self.gamma = 0.25
        self.q = 0.2 * self.S / self.sigma      # ref :38
        self.q2 = self.q / 8.0                  # ref :46-49 (use_microclimate=True)
        self.collision_mode = collision_mode
        self.dt = 1.0
        self.ddL = 0.0
        self.agent_gamma = 0.05
        self.max_L = 1.5
        self.min_L = 0.75
        self.ramp_period = ramp_period
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
        self.n_agents = n_agents


# ----------------------------------------------------------------------------------------------
# physics stages (A1, A2, A4, A5, A6 of SURVEY.md §8a)
# ----------------------------------------------------------------------------------------------
def calculate_albedo(P: Params, groundcover: np.ndarray):
    """ref: daisy_world_rl.py:377-394.  ``groundcover`` is a (B,3,H,W) *view*; channel 0 is
    rewritten in place with p - light - dark (:381).  Returns (local, adjacent), each (B,1,H,W)."""
    groundcover[:, 0] = P.p - groundcover[:, 1] - groundcover[:, 2]
    shape = (groundcover.shape[0], 1) + groundcover.shape[-2:]
    local = np.zeros(shape)
    adjacent = np.zeros(shape)
    k_adj = adjacent_albedo_kernel()
    for ii, alb in enumerate((P.albedo_bare, P.albedo_light, P.albedo_dark)):
        local += alb * groundcover[:, ii:ii + 1]
        adjacent += alb * toroidal_conv3x3(groundcover[:, ii:ii + 1], k_adj)
    return local, adjacent


def calculate_daisy_density(P: Params, local_daisies: np.ndarray) -> np.ndarray:
    """ref: daisy_world_rl.py:423-432 — each of light, dark convolved with the daisy kernel."""
    dens = np.zeros_like(local_daisies, dtype=np.float64)
    k = daisy_kernel()
    for jj in range(2):
        dens[:, jj:jj + 1] = toroidal_conv3x3(local_daisies[:, jj:jj + 1], k)
    return dens


def calculate_temperature(P: Params, L: float, local_albedo, adjacent_albedo):
    """ref: daisy_world_rl.py:396-421.  Returns (temp, temp_light, temp_dark, temp_effective,
    dead_temp)."""
    Al = local_albedo
    A = adjacent_albedo
    temp_effective = ((P.S * L * (1 - A)) / P.sigma) ** (1 / 4)
    dead_effective = ((P.S * L * (1 - P.albedo_bare)) / P.sigma) ** (1 / 4)
    temp = (P.q * (A - Al) + temp_effective ** 4) ** (1 / 4)
    light_temp = (P.q2 * (Al - P.albedo_light) + temp ** 4) ** (1 / 4)
    dark_temp = (P.q2 * (Al - P.albedo_dark) + temp ** 4) ** (1 / 4)
    return temp, light_temp, dark_temp, temp_effective, np.array([dead_effective])


def calculate_growth_rate(P: Params, temp, temp_l, temp_d):
    """ref: daisy_world_rl.py:340-348 — beta = 1 - g (T_opt - T)^2, unclipped."""
    beta = 1 - P.g * (P.temp_optimal - temp) ** 2
    beta_l = 1 - P.g * (P.temp_optimal - temp_l) ** 2
    beta_d = 1 - P.g * (P.temp_optimal - temp_d) ** 2
    return beta, beta_l, beta_d


def calculate_growth(P: Params, beta, beta_l, beta_d, daisy_density):
    """ref: daisy_world_rl.py:350-375 (the live branch :366-367)."""
    a_l = daisy_density[:, 0]
    a_d = daisy_density[:, 1]
    a_b = P.p - a_l - a_d
    growth = np.zeros_like(daisy_density)
    growth[:, 0] = a_l * (a_b * beta_l[:, 0] - P.gamma)
    growth[:, 1] = a_d * (a_b * beta_d[:, 0] - P.gamma)
    return growth


# ----------------------------------------------------------------------------------------------
# the environment
# ----------------------------------------------------------------------------------------------
class OracleDaisyWorld:
    """Restatement of ``RLDaisyWorld`` (ref: daisy_world_rl.py:13-501).

    Unlike the reference constructor it takes ``batch_size`` directly and does not draw from the
    RNG until :meth:`reset` is called; use :meth:`like_reference_ctor` to reproduce the
    reference's constructor-time RNG consumption (ref :81-83)."""

    def __init__(self, **kw):
        self.P = Params(**kw)
        self.neighborhood = neighborhood_mask(self.P.kr, self.P.neighborhood_mode)
        self.grid = None
        self.agent_indices = None
        self.agent_states = None
        self.L = self.P.min_L
        self.dL = (self.P.max_L - self.P.min_L) / self.P.ramp_period
        self.step_count = 0
        # side-effect caches (ref :345-347,373,404,415-419)
        self.temp = self.temp_light = self.temp_dark = self.temp_effective = None
        self.dead_temp = None
        self.beta = self.beta_l = self.beta_d = None
        self.growth = None

    @classmethod
    def like_reference_ctor(cls, **kw):
        """Consume the legacy global RNG exactly as ``RLDaisyWorld.__init__`` does
        (ref :81-83: initialize_agents(); reset()) with the hard-wired batch_size 32 (:20)."""
        kw = dict(kw)
        kw["batch_size"] = 32
        env = cls(**kw)
        env.initialize_agents()
        env.reset()
        return env

    # -- initialisation ------------------------------------------------------------------
    def initialize_agents(self):
        """ref: daisy_world_rl.py:173-179."""
        P = self.P
        self.agent_indices = np.random.randint(P.dim, size=(P.batch_size, P.n_agents, 2))
        self.agent_states = np.ones((P.batch_size, P.n_agents, 1))

    def initialize_grid(self):
        """ref: daisy_world_rl.py:285-324 — dark drawn first (:287), then light (:293)."""
        P = self.P
        dark_prob = np.random.rand(P.batch_size, 2, P.dim, P.dim)
        light_prob = np.random.rand(P.batch_size, 2, P.dim, P.dim)
        dark = 1.0 * (dark_prob[:, 0] < P.dark_proportion) * P.initial_ad * dark_prob[:, 1]
        light = 1.0 * (light_prob[:, 0] < P.light_proportion) * P.initial_al * light_prob[:, 1]
        self.set_initial_cover(light, dark)

    def set_initial_cover(self, light, dark):
        """Second half of ref :304-324: assemble the 7-channel grid from given covers and fill the
        three temperature channels with one physics pass at the current L (un-rounded)."""
        P = self.P
        grid = np.zeros((P.batch_size, P.ch, P.dim, P.dim))
        grid[:, CH_BARE] = P.p - light - dark
        grid[:, CH_LIGHT] = light
        grid[:, CH_DARK] = dark
        temp, temp_l, temp_d, growth = self._physics(grid)
        grid[:, 3:4] = temp
        grid[:, 4:5] = temp_l
        grid[:, 5:6] = temp_d
        self.grid = grid

    def reset(self):
        """ref: daisy_world_rl.py:327-338."""
        P = self.P
        self.L = P.min_L
        self.dL = (P.max_L - P.min_L) / P.ramp_period
        self.step_count = 0
        self.initialize_grid()
        self.initialize_agents()
        return self.get_obs(self.agent_indices)

    # -- physics -------------------------------------------------------------------------
    def _physics(self, grid):
        """The common chain of ref :314-319 and :436-443; fills the side-effect caches."""
        P = self.P
        local_albedo, adjacent_albedo = calculate_albedo(P, grid[:, :3])
        density = calculate_daisy_density(P, grid[:, 1:3])
        temp, temp_l, temp_d, temp_eff, dead = calculate_temperature(
            P, self.L, local_albedo, adjacent_albedo)
        beta, beta_l, beta_d = calculate_growth_rate(P, temp, temp_l, temp_d)
        growth = calculate_growth(P, beta, beta_l, beta_d, density)
        self.temp, self.temp_light, self.temp_dark = temp, temp_l, temp_d
        self.temp_effective, self.dead_temp = temp_eff, dead
        self.beta, self.beta_l, self.beta_d = beta, beta_l, beta_d
        self.growth = growth
        return temp, temp_l, temp_d, growth

    def forward(self, grid):
        """ref: daisy_world_rl.py:434-461."""
        P = self.P
        temp, temp_l, temp_d, growth = self._physics(grid)
        new_grid = 0.0 * grid
        new_grid[:, 3:4] = temp
        new_grid[:, 4:5] = temp_l
        new_grid[:, 5:6] = temp_d
        new_grid[:, 1:3] = np.clip(grid[:, 1:3] + P.dt * growth, 0, 1)
        new_grid[:, 0] = P.p - new_grid[:, 1] - new_grid[:, 2]
        new_grid = np.round(new_grid, decimals=3)
        if P.n_agents:
            for bb in range(P.batch_size):
                for nn in range(P.n_agents):
                    r, c = self.agent_indices[bb, nn, 0], self.agent_indices[bb, nn, 1]
                    new_grid[bb, CH_TEMP_LIGHT, r, c] = self.agent_states[bb, nn, 0]
        return new_grid

    # -- agents --------------------------------------------------------------------------
    def update_agents(self, action):
        """ref: daisy_world_rl.py:181-244 (collision_mode 0 and 1)."""
        P = self.P
        self.agent_states -= P.agent_gamma
        for bb in range(action.shape[0]):
            for nn in range(action.shape[1]):
                if self.agent_states[bb, nn, 0] > 0.0:
                    a = action[bb, nn, 0]
                    if a == 8:
                        pass
                    elif a % 4 == 0:
                        self.agent_indices[bb, nn, 1] -= 1
                    elif a % 4 == 1:
                        self.agent_indices[bb, nn, 0] -= 1
                    elif a % 4 == 2:
                        self.agent_indices[bb, nn, 0] += 1
                    elif a % 4 == 3:
                        self.agent_indices[bb, nn, 1] += 1
                    self.agent_indices[bb, nn] %= P.dim   # ref :208 wraps everything; same effect
                    if a > 4:
                        r, c = self.agent_indices[bb, nn, 0], self.agent_indices[bb, nn, 1]
                        self.agent_states[bb, nn, 0] += self.grid[bb, 1:3, r, c].sum()
                        self.grid[bb, 1:3, r, c] *= 0.0
        if P.collision_mode == 1:
            self._collisions()
        self.agent_states = np.clip(self.agent_states, 0.0, 1.0)

    def _collisions(self):
        """ref: daisy_world_rl.py:220-242 as implemented: per world, per cell in row-major order,
        if more than one agent sits on the cell the one with the largest (state + 0.01*U) gains
        food_chain_penalty * (sum of the others' states); the losers' states are *not* changed
        (the reference's ``*= 0.0`` acts on a fancy-indexed copy, :242).  One ``rand`` draw of the
        whole (1,N,1) block per multiply-occupied cell (:233)."""
        P = self.P
        for bb in range(self.agent_indices.shape[0]):
            for r in range(P.dim):
                for c in range(P.dim):
                    residents = (self.agent_indices[bb] == np.array([r, c])).all(-1)
                    if residents.sum() > 1:
                        jitter = np.random.rand(1, *self.agent_states[bb].shape)[0]
                        tv = 1.0 * self.agent_states[bb] + 0.01 * jitter
                        tvr = tv[residents]
                        winner_value = np.max(tvr)
                        eat = self.agent_states[bb][residents][tvr != winner_value].sum()
                        self.agent_states[bb][tv == winner_value] += P.food_chain_penalty * eat

    def get_obs(self, agent_indices):
        """ref: daisy_world_rl.py:246-263 — 3x3 wrap-around patch of all 7 channels, times mask."""
        P = self.P
        B, N = agent_indices.shape[:2]
        obs = np.zeros((B, N, P.ch, 3, 3))
        off = np.arange(-1, 2)
        for bb in range(B):
            for nn in range(N):
                rows = (agent_indices[bb, nn, 0] + off) % P.dim
                cols = (agent_indices[bb, nn, 1] + off) % P.dim
                obs[bb, nn] = self.grid[bb][:, rows][:, :, cols]
        return obs * self.neighborhood

    # -- stepping ------------------------------------------------------------------------
    def update_L(self, L):
        """ref: daisy_world_rl.py:463-473."""
        P = self.P
        self.step_count += 1
        if P.ramp_up_down and self.step_count % P.ramp_period == 0:
            self.dL *= -1
            P.min_L -= P.ddL
            P.max_L += P.ddL
        L += self.dL
        return max([min([L, P.max_L]), P.min_L])

    def step(self, action=None):
        """ref: daisy_world_rl.py:475-497."""
        P = self.P
        if action is None and P.n_agents:
            action = np.zeros((P.batch_size, P.n_agents, 1))
        if action is not None:
            self.update_agents(action)
        self.grid = self.forward(self.grid)
        obs = self.get_obs(self.agent_indices)
        if P.n_agents:
            reward = 1.0 * self.agent_states
        else:
            reward = self.grid[:, 1:3].sum(axis=(-2, -1)) > 0
        reward = reward * (reward > 0)
        done = reward < 0.1
        self.L = self.update_L(self.L)
        return obs, reward, done, {}


class OracleDaisyWorldC(OracleDaisyWorld):
    """The same environment with the physics pass evaluated by the C restatement (oracle/daisy_oracle.c,
    pinned against this module and the golden vectors by tests/test_oracle_golden.py) - for parity cases
    whose grids are too large for the NumPy stencils (1024^2 and up).  Agents, observations, rewards and
    the luminosity ramp are the Python code above.  The quantised outputs are identical to the NumPy
    environment's; the un-rounded side-effect caches (and the un-rounded temperature channels of the
    initial grid) agree to a few 1e-16 relative (pow() against ** 0.25)."""

    def _c_pass(self, grid):
        from . import c_oracle
        out, caches = c_oracle.forward(np.ascontiguousarray(grid[:, CH_LIGHT]), np.ascontiguousarray(grid[:, CH_DARK]),
                                       self.L, c_oracle.OracleParams.from_obj(self.P), want_caches=True)
        self.temp, self.temp_light, self.temp_dark = caches[:, 0:1], caches[:, 1:2], caches[:, 2:3]
        self.beta_l, self.beta_d = caches[:, 3:4], caches[:, 4:5]
        self.growth = caches[:, 5:7]
        return out

    def _physics(self, grid):
        self._c_pass(grid)
        return self.temp, self.temp_light, self.temp_dark, self.growth

    def forward(self, grid):
        """ref: daisy_world_rl.py:434-461 (physics in C: same staging, same float64 operations)."""
        P = self.P
        out = self._c_pass(grid)
        if P.n_agents:
            for bb in range(P.batch_size):
                for nn in range(P.n_agents):
                    r, c = self.agent_indices[bb, nn, 0], self.agent_indices[bb, nn, 1]
                    out[bb, CH_TEMP_LIGHT, r, c] = self.agent_states[bb, nn, 0]
        return out


# ----------------------------------------------------------------------------------------------
# scripted policy  (ref: daisy/agents/greedy.py:5-36)
# ----------------------------------------------------------------------------------------------
class OracleGreedy:
    """ref: agents/greedy.py:7-36.  One ``np.random.rand()`` per call decides, for the whole batch,
    between the deterministic branch (4 + argmax/argmin over the flat 3x3 indices [3,1,7,5]) and the
    uniformly random branch (``randint(9)`` per agent)."""

    def __init__(self, epsilon=0.0, greedy=True):
        self.epsilon = epsilon
        self.greedy = greedy
        self.move_mask = np.array([3, 1, 7, 5])

    def __call__(self, obs):
        B, N = obs.shape[:2]
        food = (obs[..., 1, :, :] + obs[..., 2, :, :]).reshape(B, N, 9)
        cand = food[:, :, self.move_mask]
        if np.random.rand() > self.epsilon:
            arg = np.argmax(cand, axis=-1) if self.greedy else np.argmin(cand, axis=-1)
            action = 4 + arg
        else:
            action = np.random.randint(9, size=(B, N, 1, 1))
        return action.reshape(B, N, -1)


# ----------------------------------------------------------------------------------------------
# lifespan harness  (ref: notebooks/greedy_longevity_abatement.ipynb cell 2:28-57)
# ----------------------------------------------------------------------------------------------
def simulate_lifespan(env, agent, max_steps=100000):
    """Biosphere lifespan per world and agent lifespan per agent, as the notebook counts them."""
    obs = env.reset()
    done_at = np.zeros(obs.shape[:1], dtype=int)
    agents_done_at = np.zeros((*obs.shape[:2], 1), dtype=int)
    for _ in range(max_steps):
        action = agent(obs) if agent is not None else None
        obs, reward, done, _info = env.step(action)
        grid_done = env.grid[:, 1:3].max(axis=(1, 2, 3)) <= 0.005
        done_at += (1 - 1 * grid_done)
        agents_done_at += (1 - 1 * done)
        if grid_done.mean() == 1.0:
            break
    return done_at, agents_done_at


# ----------------------------------------------------------------------------------------------
# MLP policy  (ref: daisy/agents/mlp.py:12-146) — SURVEY.md §8(f) row N3
# ----------------------------------------------------------------------------------------------
class OracleMLP:
    """63 -> 16 -> 32 -> 9 ReLU network on the flattened (7,3,3) observation, action = argmax of the
    logits (ref mlp.py:97-116).  Parameters are one flat float64 vector: the three weight matrices
    raveled row-major in layer order (ref get_parameters :118-125 / set_parameters :127-144)."""

    IN, H, OUT = 63, (16, 32), 9

    def __init__(self, parameters):
        shapes = [self.IN, *self.H, self.OUT]
        self.layers, start = [], 0
        for a, b in zip(shapes[:-1], shapes[1:]):
            self.layers.append(np.asarray(parameters[start:start + a * b], dtype=np.float64).reshape(a, b))
            start += a * b
        assert start == len(parameters) == 1808

    def forward(self, x):
        for layer in self.layers[:-1]:
            x = np.matmul(x, layer)
            x = x * (x > 0.0)
        return np.matmul(x, self.layers[-1])

    def get_action(self, obs):
        x = obs.reshape(*obs.shape[:-3], self.IN)
        return np.argmax(self.forward(x), axis=-1, keepdims=True)

    __call__ = get_action
