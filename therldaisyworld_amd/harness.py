"""Episode harnesses on top of the drop-in environment.

``simulate_lifespan(env, agent)`` is the reference's lifespan harness
(``notebooks/greedy_longevity_abatement.ipynb`` cell 2:28-57): run an episode until every world's
biosphere is dead and return, per world, the number of steps it was alive and, per agent, the number of
steps its reward stayed >= 0.1.  Results and legacy-NumPy-RNG consumption are identical to running the
notebook's loop on the reference.

For scripted policies (``Greedy`` in any of its modes, or no agent) the loop does not call ``env.step`` per
step: whole chunks of steps run device-resident (``dw_run_episode``: policy, grazing, physics and the
per-step alive flags — one launch with the worlds in LDS for H*W <= 4096, back-to-back launches without
host round trips for larger worlds); the host only draws the policy's random numbers for the chunk in the
reference's order and post-processes the flags.  If the episode ends inside a chunk the chunk is replayed
from a device-side snapshot (``dw_snapshot_save/restore``) for exactly the remaining steps, so the
environment is left in the very state (grid, agents, L, step_count, RNG stream) the reference loop would
leave it in.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ffi
from .agents.greedy import Greedy

LIFESPAN_THRESHOLD_K = 5          # grid_done = max cover <= 0.005 (notebook cell 2:48)


def _count_true(flags):
    """Number of set flags along axis 0 of a (K, ...) array of 0 / 1 bytes (bool or uint8), as intp - `flags.sum(axis=0)`.
    Eight flags at a time: the bytes are summed as uint64 words (no byte can carry for K <= 255), 8x less work than
    NumPy's bool -> int64 reduction (137 -> 17 us for 32 x 1000 x 4 flags: a third of a chunk's host time on 8x8 worlds)."""
    K, rest = flags.shape[0], flags.shape[1:]
    n = int(np.prod(rest, dtype=np.int64))
    if 0 < K <= 255 and n and n % 8 == 0 and flags.flags.c_contiguous and flags.dtype.itemsize == 1:
        words = flags.view(np.uint8).reshape(K, n).view(np.uint64).sum(axis=0)
        return words.view(np.uint8).reshape(rest).astype(np.intp)
    return np.count_nonzero(flags, axis=0)


def _grid_done(env):
    """ref notebook cell 2:48: `env.grid[:, 1:3].max(axis=(1, 2, 3)) <= 0.005` per world.  On the drop-in
    environment the per-world maximum is a by-product of the step kernel (per-mille integer, `dw_reduce`):
    k <= 5 is the same predicate without materialising and downloading the 7-channel float64 grid
    (3.7 GB per step for 1000 worlds of 256x256)."""
    from .daisy_world_rl import RLDaisyWorld
    if isinstance(env, RLDaisyWorld) and (env._grid_m is None or not env._grid_m.dirty()):
        return env._engine.reduce()["max_k"] <= LIFESPAN_THRESHOLD_K
    return env.grid[:, 1:3, :, :].max(axis=(1, 2, 3)) <= 0.005


def _simulate_on_host(env, agent, obs, done_at, agents_done_at):
    """The notebook's loop (any environment / any callable agent), one `env.step` per step."""
    while True:
        action = agent(obs) if agent is not None else None
        obs, reward, done, info = env.step(action)
        grid_done = _grid_done(env)
        done_at += (1 - 1 * grid_done)
        agents_done_at += (1 - 1 * done)
        if grid_done.mean() == 1.0:
            return done_at, agents_done_at


def _policy_mode(agent):
    if agent is None:
        return _ffi.POLICY_ZEROS                      # ref step(None): action 0 for every agent
    if type(agent) is Greedy:
        return _ffi.POLICY_ARGMAX if agent.greedy else _ffi.POLICY_ARGMIN
    return None


def _luminosity_schedule(env, nsteps):
    """L of the next `nsteps` steps, WITHOUT touching env (ref update_L :463-473 replayed on copies)."""
    L, dL, sc, mn, mx = env.L, env.dL, env.step_count, env.min_L, env.max_L
    out = []
    for _ in range(nsteps):
        out.append(L)
        sc += 1
        if env.ramp_up_down and sc % env.ramp_period == 0:
            dL *= -1
            mn -= env.ddL
            mx += env.ddL
        L += dL
        L = max([min([L, mx]), mn])
    return out


def simulate_lifespan(env, agent, chunk=32, use_device_loop=True, obs=None, final_state=True):
    """`obs=None`: reset the environment first, as the notebook's harness does.  Pass the observations of
    a reset the caller has already done (e.g. `env.reset_synthetic(seed); obs = env.get_obs()` — the
    device-side initial state for ensembles too large to draw from the host's legacy RNG).

    `final_state=False`: the caller wants the lifespans only (a sweep that discards the environment).  The chunk in
    which the last biosphere dies is then NOT replayed from a snapshot for exactly the remaining steps, and no
    snapshots are taken: the lifespans are the same numbers, but the environment (grid, agents, L, step_count, and the
    legacy-RNG stream when the policy draws) is left at the END of that chunk, up to `chunk - 1` steps past the point
    the reference loop stops at."""
    if obs is None:
        obs = env.reset()
    B, N = obs.shape[0], obs.shape[1]
    done_at = np.zeros((B,), dtype=int)
    agents_done_at = np.zeros((B, N, 1), dtype=int)
    mode = _policy_mode(agent)
    supported = env.precision != "f64" and env.collision_mode == 0
    if not (use_device_loop and supported and mode is not None):
        return _simulate_on_host(env, agent, obs, done_at, agents_done_at)

    eng = env._engine
    # step 1 through the ordinary path: the initial state is not quantised (ref initialize_grid)
    action = agent(obs) if agent is not None else None
    obs, reward, done, _ = env.step(action)
    alive = eng.reduce()["max_k"] > LIFESPAN_THRESHOLD_K
    done_at += alive
    agents_done_at += (1 - 1 * done)
    if not alive.any():
        return done_at, agents_done_at

    K = int(chunk)

    host = _ffi.load_host() if type(agent) is Greedy else None

    def draws_in_c(state, steps, use_table, table):
        """`steps` calls' worth of the policy's draws (one coin per call, randint(9) for the batch on the random branch) from
        the legacy state `state` by the host helper (include/daisyworld_host.h: the same numbers, 4x faster than two NumPy
        calls per step); the advanced state goes back to NumPy.  False: not available, nothing consumed."""
        if host is None or state[0] != "MT19937":
            return False
        key = np.array(state[1], dtype=np.uint32)
        pos = C.c_int32(int(state[2]))
        rc = host.dw_mt19937_greedy_draws(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), float(agent.epsilon), int(steps),
                                          B * N, None if use_table is None else use_table.ctypes.data_as(C.POINTER(C.c_uint8)),
                                          None if table is None else table.ctypes.data_as(C.POINTER(C.c_int8)))
        if rc != 0:
            return False
        np.random.set_state((state[0], key, pos.value, state[3], state[4]))
        return True

    def draw_chunk():
        """The policy's draws for K steps, in the reference's order, and the generator's state before them (a 2.5 KB
        copy, 34 us: once per chunk - one per step was 16 ms of an 88 ms sweep of 1000 worlds of 256 x 256)."""
        rng_before = np.random.get_state() if agent is not None else None
        use_table = np.zeros(K, dtype=np.uint8)
        table = np.zeros((K, B, N), dtype=np.int8)
        # (a policy that never takes its random branch draws one coin per step: K scalar draws are cheaper than the state
        # exchange with the helper)
        if agent is not None and not (agent.epsilon > 0.0 and draws_in_c(rng_before, K, use_table, table)):
            for t in range(K):
                if not agent.draw_branch():
                    use_table[t] = 1
                    table[t] = agent.draw_random_actions(B, N)[..., 0]
        return rng_before, use_table, table

    def rewind_draws(rng_before, steps):
        """Leave the generator where the reference's loop leaves it: after the draws of `steps` steps of this chunk."""
        if rng_before is None:
            return
        scratch_ut, scratch_tb = np.zeros(max(steps, 1), dtype=np.uint8), np.zeros((max(steps, 1), B, N), dtype=np.int8)
        if agent.epsilon > 0.0 and draws_in_c(rng_before, steps, scratch_ut, scratch_tb):
            return
        np.random.set_state(rng_before)
        for _ in range(steps):
            if not agent.draw_branch():
                agent.draw_random_actions(B, N)

    # The device runs chunk c (a blocking C call that releases the GIL) on a worker thread while this thread draws the
    # policy's random numbers for chunk c + 1: an epsilon-greedy policy's legacy-RNG draws (32 x B x N integers per
    # chunk, 1.2 ms for 1000 worlds) otherwise sit between the chunks - 17 % of the `random` policy's sweep time.
    # The draws do not depend on the device's results; when the episode turns out to end in chunk c, the generator is
    # rewound to its state before chunk c and the draws of the executed steps are repeated.
    # (a policy that never takes its random branch draws one coin per step: nothing to overlap, and the hand-over to
    # the worker costs ~0.2 ms per chunk)
    pool = ThreadPoolExecutor(max_workers=1) if agent is not None and getattr(agent, "epsilon", 1.0) > 0.0 else None
    try:
        ahead = None
        while True:
            rng_before, use_table, table = ahead if ahead is not None else draw_chunk()
            ahead = None
            Ls = _luminosity_schedule(env, K)

            def on_device():
                if final_state:
                    eng.snapshot_save()
                # (the chunk's flags are summed below before the next chunk runs: the engine's own flag buffers are re-used)
                return eng.run_episode(Ls, mode, use_table, table, LIFESPAN_THRESHOLD_K, reuse_buffers=True)

            if pool is not None:
                pending = pool.submit(on_device)
                ahead = draw_chunk()
                alive_k, ok_k = pending.result()
            else:
                alive_k, ok_k = on_device()
            all_dead = ~alive_k.any(axis=1)
            ended = bool(all_dead.any())                       # also when the last world died on the chunk's last step
            executed = int(np.argmax(all_dead)) + 1 if ended else K
            done_at += _count_true(alive_k[:executed])
            agents_done_at += _count_true(ok_k[:executed])[..., None]
            if not final_state:
                executed = K                                   # the environment stays where the chunk ended
            elif executed < K:
                # the episode ended inside the chunk: replay exactly `executed` steps from the snapshot
                eng.snapshot_restore()
                eng.run_episode(Ls[:executed], mode, use_table[:executed], table[:executed], LIFESPAN_THRESHOLD_K)
            if ended:
                rewind_draws(rng_before, executed)
            for _ in range(executed):                          # host scalars of the environment
                env._L_pass = env.L
                env.L = env.update_L(env.L)
            env._invalidate()
            if ended:
                return done_at, agents_done_at
    finally:
        if pool is not None:
            pool.shutdown(wait=True)


def _advance_host_scalars(env, executed):
    """L, step_count and the ramp bookkeeping of `executed` steps the device has already taken."""
    for _ in range(executed):
        env._L_pass = env.L
        env.L = env.update_L(env.L)
    env._invalidate()


def _mlp_chunks(env, params, member_a, member_b, half, max_steps, chunk, after_chunk):
    """The step loop shared by the two fitness harnesses: chunks of steps device-resident
    (``dw_run_episode_mlp``), the reference's per-step float64 bookkeeping done by `after_chunk(rewards,
    dones) -> (executed, finished)` on the downloaded (K,B,N,1) rewards (already `reward * (reward > 0)`) and
    done flags of the chunk: it accounts for the steps up to and including the one that ends the episode, with
    every float64 sum accumulated in step order.  When the episode ends inside a chunk, the chunk is replayed
    from a device-side snapshot for exactly the executed steps, so the environment is left where the
    reference's loop leaves it.

    The device runs chunk c + 1 while this thread accounts for chunk c (a worker thread around the blocking C calls,
    which release the GIL): the bookkeeping - NumPy reductions in the reference's order over 4 MB of rewards per chunk
    for a population of 64 x 32 worlds - costs more than the chunk's kernel.  Chunk c + 1 does not depend on it except
    for "has the episode ended": its rewards go to the engine's second page-locked buffer, the state at its start to
    snapshot slot (c + 1) % 2, and when chunk c turns out to end the episode the speculative chunk is undone from the
    snapshots (slot c % 2 + a replay of the executed steps, or slot (c + 1) % 2 as it is when chunk c ran to its end)."""
    eng = env._engine
    n_members = np.asarray(params).reshape(-1, 1808).shape[0]
    if env.step_count >= max_steps:
        return
    env._sync_to_device()

    def on_device(c, Ls, L_init, first):
        if not first:                                    # (the first step starts from the un-quantised state: nothing to save,
            eng.snapshot_save(c & 1)                     # and it is never undone - it always runs to its end)
        # the parameter sets go up with the first chunk and stay on the device; rewards / done flags come back in the
        # engine's page-locked buffers (consumed by after_chunk before the chunk after the next overwrites them)
        return eng.run_episode_mlp(Ls, params if first else None, member_a, member_b, half, L_init,
                                   reuse_buffers=c & 1, n_members=n_members)

    pool = ThreadPoolExecutor(max_workers=1)
    try:
        c, K = 0, 1                                      # step 1 starts from the un-quantised state: a chunk of its own
        Ls = _luminosity_schedule(env, K)
        L_init = env._L_pass
        pending = pool.submit(on_device, c, Ls, L_init, True)
        while True:
            rewards, dones = pending.result()
            ahead = None
            left = max_steps - (env.step_count + K)
            if left > 0:                                 # chunk c + 1, before chunk c is accounted for
                K_n = min(int(chunk), left)
                sched = _luminosity_schedule(env, K + K_n)
                Ls_n, L_init_n = sched[K:], sched[K - 1]
                ahead = pool.submit(on_device, c + 1, Ls_n, L_init_n, False)
            # (the device already returns reward * (reward > 0), ref step :490: agent states are clipped to [0, 1], so the
            # product is the state itself - a second pass over the (K,B,N,1) array would change no bit of it)
            executed, finished = after_chunk(rewards, dones)
            if executed < K or finished:
                if ahead is not None:
                    ahead.result()                       # the speculative chunk has to leave the device first
                if executed < K:                         # ended inside chunk c: its start + exactly the executed steps
                    eng.snapshot_restore(c & 1)
                    eng.run_episode_mlp(Ls[:executed], None, member_a, member_b, half, L_init, reuse_buffers=c & 1,
                                        n_members=n_members)
                elif ahead is not None:                  # ended with chunk c's last step: the state chunk c + 1 started from
                    eng.snapshot_restore((c + 1) & 1)
                ahead = None
            _advance_host_scalars(env, executed)
            if ahead is None:
                return                                   # the episode ended, or max_steps is reached
            pending, c, K, Ls, L_init = ahead, c + 1, K_n, Ls_n, L_init_n
    finally:
        pool.shutdown(wait=True)


def _fitness_chunk(acc, rewards, dones, half):
    """Bookkeeping of `get_fitness` for one chunk of K steps (rewards (K,B,N,1) = reward * (reward > 0), dones
    (K,B,N,1) bool), equal to the reference's per-step statements (daisy/evo/sges.py:160-176: all_done,
    done_at += 1 - done, sum_reward += mean reward of the agents' half, total_steps += 1 - done; the loop
    leaves after the step in which every agent is done).  Returns (steps accounted for, episode ended)."""
    K = rewards.shape[0]
    ended = np.nonzero(dones.reshape(K, -1).all(axis=1))[0]
    executed = int(ended[0]) + 1 if ended.size else K
    alive = np.count_nonzero(~dones[:executed], axis=0)               # = sum(1 - done): integers, any order
    acc["done_at"] += alive
    acc["total_steps"] = acc["total_steps"] + alive
    means = [rewards[t][:, :half].mean() for t in range(executed)]    # the reference's call, step by step
    acc["sum_reward"] = np.add.accumulate(np.array([acc["sum_reward"]] + means))[-1]   # added in step order
    return executed, ended.size > 0


def _member_means(rewards, P, wpm, half):
    """Per (step, member): `reward[:, :half].mean()` of that member's block of worlds, as the reference forms it (sges.py:170 on
    a (wpm, N, 1) array) - NumPy flattens the non-contiguous slice in C order and sums it pairwise.  rewards: (K, P * wpm, N, 1).
    For the reference's own configuration (N = 4 agents, half = 2) the first two rewards of every world are gathered as ONE
    16-byte item (a complex128 view) and reduced along a CONTIGUOUS axis - the same pairwise sum over the same sequence, 1.7x
    faster than the strided multi-axis reduction; every other shape takes that reduction (both equal the reference's call on
    each block: tests/test_abi_and_host.py)."""
    K, B, N = rewards.shape[0], rewards.shape[1], rewards.shape[2]
    if N == 4 and half == 2 and rewards.flags.c_contiguous and rewards.dtype == np.float64:
        first = np.ascontiguousarray(rewards.reshape(K, B, 4).view(np.complex128)[:, :, 0])       # agents 0, 1 of every world
        return first.view(np.float64).reshape(K, P, wpm * 2).sum(axis=-1) / (wpm * 2)
    return rewards.reshape(K, P, wpm, N, 1)[:, :, :, :half].mean(axis=(2, 3, 4))


def _population_chunk(acc, rewards, dones, half, worlds_per_member):
    """The same for a population evaluated as one ensemble (member m owns worlds [m*wpm, (m+1)*wpm)): while a
    member runs, done_at / total_steps += 1 - done and sum_reward[m] += mean reward of its agents' half; the
    member stops after the step in which all ITS agents are done; the ensemble stops when no member runs.
    acc: done_at, total_steps (B,N,1) int, sum_reward (P,) float64, running (P,) bool - updated in place."""
    K, B, N = rewards.shape[0], rewards.shape[1], rewards.shape[2]
    P = B // worlds_per_member
    running = acc["running"]
    all_done = dones.reshape(K, P, -1).all(axis=2)                                 # (K,P)
    earlier = np.zeros((K, P), dtype=bool)
    earlier[1:] = np.logical_or.accumulate(all_done, axis=0)[:-1]
    run_t = running[None, :] & ~earlier                                            # running DURING step t
    none_left = np.nonzero(~(run_t & ~all_done).any(axis=1))[0]
    executed = int(none_left[0]) + 1 if none_left.size else K
    if run_t[:executed].all():                                                     # the usual chunk: every member still runs
        alive = executed - _count_true(dones[:executed])                           # = sum(1 - done): integers, any order
    else:
        live = np.repeat(run_t[:executed], worlds_per_member, axis=1)[:, :, None, None]
        alive = np.count_nonzero(live & ~dones[:executed], axis=0)                 # = sum(live * (1 - done))
    acc["done_at"][...] += alive
    acc["total_steps"][...] += alive
    # one member's mean of one step = the reference's `reward[:, :half].mean()` on that member's block of worlds
    # (the same pairwise reduction per (step, member); tests G11 + tools/fuzz_fitness.py hold it to that)
    means = _member_means(rewards[:executed], P, worlds_per_member, half)
    acc["sum_reward"][...] = np.add.accumulate(np.concatenate([acc["sum_reward"][None], run_t[:executed] * means]),
                                               axis=0)[-1]                         # added in step order
    running[...] = run_t[executed - 1] & ~all_done[executed - 1]
    return executed, none_left.size > 0


def get_fitness(env, agent, adversary, max_steps=768, chunk=64):
    """ref SimpleGaussianES.get_fitness (daisy/evo/sges.py:144-181): one episode in which the first half
    of every world's agents is driven by `agent` and the second half by `adversary` (both MLP policies);
    returns (fitness, total_steps, done_at) exactly as the reference computes them.  Observations,
    policies, grazing and physics stay on the device for `chunk` steps at a time; only the (K,B,N) rewards
    and done flags come back, for the reference's float64 means in step order."""
    agent.reset()
    obs = env.reset()
    B, N = obs.shape[0], obs.shape[1]
    half = N // 2
    params = np.stack([agent.get_parameters(), adversary.get_parameters()])
    member_a, member_b = np.zeros(B, dtype=np.int32), np.ones(B, dtype=np.int32)
    acc = {"done_at": np.zeros((B, N, 1), dtype=int), "total_steps": 0, "sum_reward": 0.0}
    _mlp_chunks(env, params, member_a, member_b, half, max_steps, chunk,
                lambda rewards, dones: _fitness_chunk(acc, rewards, dones, half))
    fitness = acc["sum_reward"] / (B * N)
    return fitness, acc["total_steps"], acc["done_at"].tolist()


def get_fitness_population(env, population, adversary_of=None, worlds_per_member=32, max_steps=768, chunk=64):
    """All members of an evolution-strategy population evaluated as ONE batched ensemble (SURVEY.md §8f
    N3; the reference runs `get_fitness` once per member and farms members out to MPI workers,
    daisy/evo/sges.py:314-349).  `population` is a list of MLP policies (or a (P,1808) array);
    member m owns worlds [m*worlds_per_member, (m+1)*worlds_per_member); in each of its worlds the
    first half of the agents is driven by member m and the second half by member adversary_of[m]
    (default: itself).  Returns per-member (fitness, total_steps, done_at) computed exactly as
    `get_fitness` computes them for that member's block of worlds.

    Each member's episode ends when all ITS agents are done (or at max_steps), as in the reference; the
    ensemble keeps stepping until every member has finished, and a finished member's statistics are
    frozen at its own stopping step.  (With the legacy-RNG reset the worlds differ from P separate
    `reset()` calls of the reference; the per-member arithmetic is the same.)"""
    params = np.stack([m.get_parameters() if hasattr(m, "get_parameters") else np.asarray(m) for m in population])
    P = params.shape[0]
    adversary_of = np.arange(P) if adversary_of is None else np.asarray(adversary_of, dtype=int)
    env.batch_size = P * worlds_per_member
    obs = env.reset()
    B, N = obs.shape[0], obs.shape[1]
    half = N // 2
    member = np.repeat(np.arange(P), worlds_per_member).astype(np.int32)
    adv_member = adversary_of[member].astype(np.int32)
    done_at = np.zeros((B, N, 1), dtype=int)
    total_steps = np.zeros((B, N, 1), dtype=int)
    sum_reward = np.zeros(P)
    running = np.ones(P, dtype=bool)

    acc = {"done_at": done_at, "total_steps": total_steps, "sum_reward": sum_reward, "running": running}
    _mlp_chunks(env, params, member, adv_member, half, max_steps, chunk,
                lambda rewards, dones: _population_chunk(acc, rewards, dones, half, worlds_per_member))
    fitness = sum_reward / (worlds_per_member * N)
    return [(fitness[m], total_steps[m * worlds_per_member:(m + 1) * worlds_per_member],
             done_at[m * worlds_per_member:(m + 1) * worlds_per_member].tolist()) for m in range(P)]
