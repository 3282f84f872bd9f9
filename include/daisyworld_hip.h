/* daisyworld_hip.h — C ABI of libdaisyworld_hip.so, the MI355X (gfx950) implementation of the
 * RLDaisyWorld grid-update hot path.
 *
 * The reference (riveSunder/therldaisyworld) is pure Python/NumPy and has no FFI layer: the
 * "operator interface" of this path is the method surface of class RLDaisyWorld in
 * daisy/daisy_world_rl.py.  Each entry point below therefore cites the reference method
 * (file:line) whose work it performs.  The Python drop-in therldaisyworld_amd.RLDaisyWorld binds
 * these functions with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C: opaque handle, plain pointers and sizes; no C++/torch types cross the boundary.
 *   - every function returns DW_OK (0) or a negative DW_E* code and never throws; the message of
 *     the last failure on the calling thread is returned by dw_last_error().
 *   - the caller owns every host buffer (C-contiguous); the library owns all device memory inside
 *     the handle and frees it in dw_destroy().
 *   - a handle is bound to one HIP device and one stream; it is not thread-safe, distinct handles
 *     may be used from distinct threads / processes (one process per GPU for ensemble shards).
 *   - dw_step*, dw_policy_* and dw_init_random are asynchronous on the handle's stream; every
 *     function that returns data to the host synchronises that stream first.
 *   - there is NO CPU fallback: on a machine without a usable HIP device dw_create() fails with
 *     DW_ENODEVICE.
 *
 * Layout
 *   grids are row-major [world][row][col]; "row" is the reference's axis -2 (which its code calls
 *   x) and "col" its axis -1 (y).  Device state is two planar BINARY16 arrays per buffer (light,
 *   dark) holding the cover in PER-MILLE units (1000 * cover: exactly the integers 0..1000 once a
 *   step has run, because the reference quantises to 3 decimals, daisy_world_rl.py:452 - and binary16
 *   holds every integer up to 2048 exactly, so the format is lossless: 2 bytes per value, 8 bytes of
 *   HBM traffic per cell-update), kept ping-pong so that the pre-step state stays available for
 *   observations / materialisation.  An UN-quantised state (the reference's initial grid is not
 *   rounded, :285-324) stays in its upload format - float64, or float32 per-mille - until the first
 *   step has consumed it.
 *
 * Numerics of the default mode (DW_PRECISION_EXACT): the light / dark planes after any step, and
 * everything derived from quantised values (observations, rewards, done flags, lifespans, the rounded
 * temperature channels), are bit-identical to the reference's float64 NumPy path.  Two caveats, stated
 * here so that "bit-exact" is not read as a proof: (i) the reference convolves by FFT, this library by
 * the mathematically identical 9-tap stencil, and the float64 repair path forms T_x^4 directly instead
 * of through the chain of fourth roots - both differ from the reference by a few 1e-16 relative BEFORE
 * rounding, the size of the reference's own FFT noise; a cell whose float64 pre-rounding value lies
 * that close to a rounding tie could round differently (never observed: 2e10 cell-updates soaked
 * against the oracle on the round-3 kernels alone, 0 mismatches); (ii) un-quantised outputs - reset() observations, env.grid after
 * reset(), the temp / beta / growth caches - agree with the reference to ~1e-13 relative, not bit for bit.
 */
#ifndef DAISYWORLD_HIP_H
#define DAISYWORLD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DW_ABI_VERSION 5

/* ---- error codes ---------------------------------------------------------------------------- */
enum {
    DW_OK = 0,
    DW_EINVAL = -1,     /* bad argument (null pointer, shape mismatch, unsupported size) */
    DW_ENODEVICE = -2,  /* no usable HIP device / wrong architecture */
    DW_ENOMEM = -3,     /* device or host allocation failed */
    DW_EHIP = -4,       /* a HIP runtime call failed (message has the HIP error string) */
    DW_ESTATE = -5      /* call not valid in the handle's current state (e.g. step before upload) */
};

/* ---- arithmetic modes ----------------------------------------------------------------------- */
enum {
    /* float32 arithmetic with a float64 re-evaluation of every cell whose pre-rounding value lies
     * within a proven error bound of a rounding tie: results are bit-identical to float64
     * evaluation of the reference formulas (the default). */
    DW_PRECISION_EXACT = 0,
    /* float32 arithmetic only: every cell within one quantum (1e-3) of the float64 result and
     * >= 99.98 % of the cell values identical after one step from the same state (measured >= 99.994 % on
     * developed states over the whole luminosity ramp, profiles/r02_fast_tolerance.json); ensemble means
     * of a trajectory within 3e-4.  A single small world's trajectory diverges cell-wise (one flipped tie
     * is amplified by the dynamics): that is what the exact mode is for. */
    DW_PRECISION_FAST = 1,
    /* float64 arithmetic for every cell (slow; the in-library reference the other two are tested
     * against).  (The exact mode's first step from an un-quantised initial state is float32 with an error bound
     * for non-integer inputs, float64 only for the cells that bound cannot decide: the same results.) */
    DW_PRECISION_F64 = 2
};

/* which of the two retained states a download refers to */
enum { DW_STATE_CURRENT = 0, DW_STATE_PREVIOUS = 1 };

/* scripted policy modes for dw_policy_greedy (ref: daisy/agents/greedy.py:14-36) */
enum { DW_POLICY_ARGMAX = 0, DW_POLICY_ARGMIN = 1 };

typedef struct dw_handle dw_handle;

/* Construction-time shape + the physics constants of RLDaisyWorld.__init__
 * (ref: daisy_world_rl.py:18-79).  Shapes are fixed for the life of the handle; the constants can
 * be changed later with dw_set_params (the reference lets callers assign attributes and then call
 * reset(), e.g. notebooks/greedy_longevity_abatement.ipynb cell 2:3-8). */
typedef struct dw_params {
    int32_t abi_version;      /* must be DW_ABI_VERSION */
    int32_t batch;            /* B: worlds in this handle (ref batch_size, :20) */
    int32_t height;           /* H = grid_dimension (:29) */
    int32_t width;            /* W = grid_dimension; H != W is allowed (ft_convolve accepts it) */
    int32_t n_agents;         /* N agents per world (:79); 0 allowed */
    int32_t device;           /* HIP device ordinal */
    int32_t precision;        /* DW_PRECISION_* */
    int32_t obs_mask;         /* 9-bit row-major 3x3 neighbourhood mask applied to observations
                                 (bit 4 = centre); von Neumann = 0x0BA, Moore = 0x1FF
                                 (ref: nn/functional.py:51-103, daisy_world_rl.py:261) */
    int32_t collision_mode;   /* 0 or 1 (ref :44, :220-242) */
    int32_t reserved0;
    int64_t world_offset;     /* global id of world 0 of this shard; keys the device RNG so that an
                                 ensemble sharded over GPUs draws the same worlds as a single run */
    double p, g, S, sigma, gamma, q, q2, dt;                         /* ref :32-52; g < 0 (a growth curve that opens
                                                                        upwards) only with DW_PRECISION_F64: the float32
                                                                        modes return DW_EINVAL for it */
    double albedo_bare, albedo_light, albedo_dark, temp_optimal;     /* ref :65-69 */
    double agent_gamma, food_chain_penalty;                          /* ref :55, :70 */
    double initial_al, initial_ad, light_proportion, dark_proportion;/* ref :73-77 */
} dw_params;

/* Per-world reductions fused into every step: the callers' lifespan test
 * `grid[:,1:3].max((1,2,3)) <= 0.005` (notebook greedy_longevity_abatement cell 2:48), the no-agent
 * reward `grid[:,1:3].sum((-2,-1)) > 0` (ref :489) and the population means.  All integers, in
 * per-mille units, hence exact and order-independent. */
typedef struct dw_world_stats {
    uint32_t max_k;        /* max over light and dark cells of 1000*cover */
    uint32_t reserved;     /* library-internal (a diagnostic count behind dw_last_fixup_count); do not interpret */
    uint64_t sum_light_k;  /* sum over cells of 1000*light */
    uint64_t sum_dark_k;   /* sum over cells of 1000*dark  */
} dw_world_stats;

/* ---- life cycle ----------------------------------------------------------------------------- */

/* Fill *p with the reference's default constants for the given shape (ref :18-79). */
int dw_default_params(dw_params* p, int32_t batch, int32_t height, int32_t width, int32_t n_agents);

/* Create a handle (allocates device state).  ref: RLDaisyWorld.__init__ :15-83 (without the RNG
 * draws, which stay on the host in the shim for same-seed parity, or use dw_init_random). */
int dw_create(const dw_params* p, dw_handle** out);
int dw_destroy(dw_handle* h);

/* Replace the physics constants (shape / device fields must match the handle's). */
int dw_set_params(dw_handle* h, const dw_params* p);
int dw_get_params(const dw_handle* h, dw_params* out);

const char* dw_last_error(void);
int dw_abi_version(void);

/* Page-locked host memory (hipHostMalloc) for buffers the library fills or reads: a download into it runs at the full
 * host-link rate, without the driver's staging copies (a pageable 4 MB reward block of dw_run_episode_mlp: 0.2 ms
 * instead of 0.5-2 ms, box dependent).  No reference counterpart; needs a gfx950 device (DW_ENODEVICE otherwise). */
int dw_pinned_alloc(size_t bytes, void** out);
int dw_pinned_free(void* p);
/* Hash of the sources + compiler flags this library was built from (16 hex digits; "unknown" for a build
 * that did not go through therldaisyworld_amd/build.py).  build.py rebuilds when it differs from the
 * sources in the tree, so a stale binary cannot run under newer host code. */
const char* dw_build_id(void);

/* ---- state in / out ------------------------------------------------------------------------- */

/* Upload an initial cover state, cover fractions in natural units [B][H][W] float64; the state is
 * treated as NOT quantised (ref initialize_grid :285-324 does not round): it is kept as uploaded, and the
 * next step reads these exact values (exact mode: float32 with a bound for non-integer inputs, the undecided
 * cells in float64 from the originals - bit-identical to a float64 evaluation).  Resets the retained
 * "previous" state. */
int dw_upload_state_f64(dw_handle* h, const double* light, const double* dark);

/* Same from float32 natural-unit planes.  quantised != 0 asserts every value is k/1000 (it is rounded to the
 * per-mille integer and goes straight into the binary16 planes); 0: the state is un-quantised, kept as float32
 * per-mille until the first step. */
int dw_upload_state_f32(dw_handle* h, const float* light, const float* dark, int quantised);

/* Agent positions [B][N][2] (row, col) and energy stores [B][N] (ref initialize_agents :173-179). */
int dw_upload_agents(dw_handle* h, const int32_t* indices, const double* states);
int dw_download_agents(dw_handle* h, int32_t* indices, double* states);

/* Synthetic initial state generated on the device with a counter-based RNG (Philox4x32-10 keyed by
 * seed, global world id and cell), same distribution as ref initialize_grid :287-302 and
 * initialize_agents :175-179: cover = [U1 < proportion] * initial_a * U2 per species, agents at
 * uniform cells with state 1.  Used for the large synthetic configurations. */
int dw_init_random(dw_handle* h, uint64_t seed);

/* The same draw with every cover rounded to three decimals (what np.round(., 3) would make of it), written
 * straight into the binary16 planes: a QUANTISED synthetic state.  Needs no float32 staging (dw_init_random holds
 * the un-quantised state in 8 bytes per cell until the second step) and no float64 first step; for ensembles that
 * fill the device.  Same worlds as dw_init_random up to that rounding (same keys). */
int dw_init_random_quantised(dw_handle* h, uint64_t seed);

/* Download light/dark cover of the current or previous state in natural units, [B][H][W] float64
 * (either pointer may be NULL). */
int dw_download_planes(dw_handle* h, int which, double* light, double* dark);

/* Materialise the reference's 7-channel grid `self.grid` [B][7][H][W] float64 for the current
 * state: after a step = exactly ref forward() :445-459 (rounded covers and bare, rounded
 * temperatures of the pre-step state, agent states written into channel 4); after an upload = ref
 * initialize_grid :304-323 (un-rounded).  L_init is used only in the second case. */
int dw_download_grid(dw_handle* h, double L_init, double* grid7);

/* Side-effect caches of the last physics pass, each optional (NULL to skip):
 * temps [B][3][H][W] = temp, temp_light, temp_dark (un-rounded; ref :415-419),
 * betas [B][3][H][W] = beta, beta_l, beta_d (ref :345-347), growth [B][2][H][W] (ref :373),
 * temp_effective [B][H][W] (ref :404).  L is the luminosity of that pass. */
int dw_download_caches(dw_handle* h, double L, double* temps, double* betas, double* growth,
                       double* temp_effective);

/* ---- the hot path --------------------------------------------------------------------------- */

/* One environment step with luminosity L (ref step :475-497 minus update_L, which stays on the
 * host as a float64 scalar recurrence, :463-473):
 *   1. if action != NULL: update_agents (ref :181-244) for the leading action_b x action_n block
 *      of agents, host int32 actions [action_b][action_n] (codes 0..8);
 *   2. grid = forward(grid) (ref :434-461) as one fused stencil + reaction kernel;
 *   3. per-world reductions (dw_world_stats).
 * Asynchronous. */
int dw_step(dw_handle* h, const int32_t* action, int32_t action_b, int32_t action_n, double L);

/* dw_step followed by dw_get_obs and dw_get_reward_done with a single synchronisation: what one call of the
 * reference's step() returns (ref :475-497: obs, reward, done), for latency-bound small batches.  obs /
 * reward / done may be NULL. */
int dw_env_step(dw_handle* h, const int32_t* action, int32_t action_b, int32_t action_n, double L,
                double* obs /* [B][N][7][3][3] */, double* reward /* [B][N] */, uint8_t* done /* [B][N] */);

/* Same, taking the actions from the handle's device action buffer (filled by dw_policy_greedy or
 * dw_upload_actions) — keeps an episode loop free of host round trips. */
int dw_step_device_actions(dw_handle* h, double L);
int dw_upload_actions(dw_handle* h, const int32_t* action /* [B][N] */);
int dw_download_actions(dw_handle* h, int32_t* action /* [B][N] */);

/* `nsteps` consecutive steps with the reference's luminosity recurrence L <- clamp(L + dL)
 * (ref :471-473) evaluated on the host in float64, *L_io updated.  use_device_actions: 0 = no
 * update_agents call at all (ref `action is None` with n_agents == 0), 1 = actions are taken from
 * the device action buffer every step (constant unless a policy refreshes it).
 * The result equals `nsteps` calls of dw_step bit for bit; how the steps are issued is the library's business
 * (on wide grids two steps share one launch: step-1 values live only in registers). */
int dw_step_n(dw_handle* h, int32_t nsteps, double* L_io, double dL, double min_L, double max_L,
              int use_device_actions);

/* Measurement aid (bench.py): duration of the run of fused step-pair launches issued by the LAST
 * dw_step_n call, from HIP events recorded on the handle's stream immediately before the first and after the
 * last of them (synchronises).  fused_launches = 0 (and fused_ms = 0) if that call issued none.
 * plane_elem_bytes = sizeof of the plane element those launches read and write (2: binary16). */
int dw_last_step_n_timing(dw_handle* h, float* fused_ms, int32_t* fused_launches, int32_t* plane_elem_bytes);

/* update_agents alone (ref :181-244) and forward alone on caller data (ref :434-461).  With
 * collision_mode = 1 dw_update_agents stops before the final clip: the reference's collision pass
 * (:220-242) draws from the caller's legacy RNG once per multiply-occupied cell, so the caller applies
 * it to the downloaded agent states, clips and uploads them (the Python drop-in does), then steps
 * without actions; dw_step / dw_run_episode with actions refuse collision_mode = 1.  dw_forward_f64
 * is stateless w.r.t. the cover planes: host covers [B][H][W] float64 in, 7-channel grid out
 * (agent states of the handle are written into channel 4 as the reference does); the optional
 * cache outputs are those of dw_download_caches for that input. */
int dw_update_agents(dw_handle* h, const int32_t* action, int32_t action_b, int32_t action_n);
int dw_forward_f64(dw_handle* h, const double* light, const double* dark, double L, double* grid7,
                   double* temps, double* betas, double* growth, double* temp_effective);

/* One 3x3 toroidal convolution of a caller-supplied float64 plane [B][H][W] with the 9 row-major kernel weights -
 * exactly the operation of ft_convolve (ref daisy/nn/functional.py:12-49, a true convolution on the torus) that
 * calculate_albedo (ref :377-394) and calculate_daisy_density (ref :423-432) are built from.  The drop-in's module-level
 * therldaisyworld_amd.nn.functional.ft_convolve(grid, kernel) calls it (the calculate_* methods use dw_stage_f64, the step
 * kernels fuse their stencils).  Does not touch the handle's state. */
int dw_conv3x3_f64(dw_handle* h, const double* plane, const double kernel[9], double* out);

/* The stages of forward() as stand-alone float64 maps on caller data, evaluated on the device (the drop-in's
 * calculate_* methods; forward() and step() use the fused kernels, never these).  `in` / `out`: host arrays of
 * whole planes, [plane][B][H][W] float64; `L`: the luminosity of the pass (stage 3); `kernel`: the nine row-major
 * weights of the stage's 3x3 toroidal convolution (stages 1, 2; ft_convolve, ref daisy/nn/functional.py:12-49).
 *   DW_STAGE_ALBEDO       ref calculate_albedo :377-394         in  bare (ignored: recomputed as p - l - d, ref :381), light, dark
 *                                                                out local albedo, adjacent albedo, bare
 *   DW_STAGE_DENSITY      ref calculate_daisy_density :423-432  in  light, dark           out density_light, density_dark
 *   DW_STAGE_TEMPERATURE  ref calculate_temperature :396-421    in  local, adjacent albedo out temp_effective, temp, temp_light, temp_dark
 *   DW_STAGE_GROWTH_RATE  ref calculate_growth_rate :340-348    in  temp, temp_light, temp_dark out beta, beta_l, beta_d
 *   DW_STAGE_GROWTH       ref calculate_growth :350-375         in  beta_l, beta_d, density_light, density_dark out growth_light, growth_dark
 * Does not touch the handle's state. */
#define DW_STAGE_ALBEDO 1
#define DW_STAGE_DENSITY 2
#define DW_STAGE_TEMPERATURE 3
#define DW_STAGE_GROWTH_RATE 4
#define DW_STAGE_GROWTH 5
int dw_stage_f64(dw_handle* h, int stage, const double* in, double* out, double L, const double kernel[9]);

/* Observations (ref get_obs :246-263): [B][N][7][3][3] float64 for the handle's agents, taken from
 * the current grid exactly as dw_download_grid would materialise it, times the neighbourhood mask. */
int dw_get_obs(dw_handle* h, double L_init, double* obs);

/* reward [B][N] float64 and done [B][N] uint8 as ref step :486-492 (N > 0). */
int dw_get_reward_done(dw_handle* h, double* reward, uint8_t* done);

/* Per-world reductions of the current state (always up to date after a step). */
int dw_reduce(dw_handle* h, dw_world_stats* per_world /* [B] */);

/* Scripted policy on the device (ref Greedy.__call__ agents/greedy.py:14-36): fills the device
 * action buffer from the current observations.  mode: DW_POLICY_ARGMAX -> 4+argmax of light+dark
 * over the candidates (row,col-1),(row-1,col),(row+1,col),(row,col+1) (= flat 3x3 indices 3,1,7,5),
 * first maximum wins; DW_POLICY_ARGMIN -> 4+argmin.  The epsilon (uniformly random) branch is
 * drawn by the caller (so that the legacy NumPy stream stays on the host) and handed over with
 * dw_upload_actions. */
int dw_policy_greedy(dw_handle* h, int mode);

/* Mixed-policy ensembles (BASELINE configs[4]: policy chosen by agent index): agent_mode[N] per agent
 * index DW_POLICY_ARGMAX, DW_POLICY_ARGMIN, or DW_POLICY_TABLE = keep the action already in the device
 * action buffer (host-drawn random actions put there with dw_upload_actions). */
int dw_policy_per_agent(dw_handle* h, const int32_t* agent_mode);

/* Learned policy on the device (ref MLP.get_action, daisy/agents/mlp.py:97-116):
 * the 63 -> 16 -> 32 -> 9 ReLU network evaluated in float64 on the current observations of agents
 * [agent_begin, agent_end) of every world; argmax of the logits goes to the device action buffer.
 * params: host float64[1808], the three weight matrices raveled row-major in layer order (ref
 * get_parameters :118-125).  Two calls with the two halves of the agents reproduce the agent /
 * adversary split of sges.get_fitness (daisy/evo/sges.py:163-168). */
int dw_policy_mlp(dw_handle* h, const double* params, int32_t n_params, int32_t agent_begin, int32_t agent_end,
                  double L_init);

/* The same for a whole population at once: params is [n_members][1808] and world_member[B] names the
 * parameter set each world's agents use, so that an evolution strategy evaluates all its members as ONE
 * batched ensemble instead of one episode per member (the reference farms members out to MPI workers,
 * daisy/evo/sges.py:314-349). */
int dw_policy_mlp_population(dw_handle* h, const double* params, int32_t n_members, const int32_t* world_member,
                             int32_t agent_begin, int32_t agent_end, double L_init);

/* Device-resident fitness episode of an evolution strategy (ref SimpleGaussianES.get_fitness,
 * daisy/evo/sges.py:144-181, and the population loop :314-349): K consecutive steps in which agents
 * [0, split) of world b act with MLP parameter set member_a[b] and agents [split, N) with member_b[b]
 * (ref: `agent` drives the first half, `adversary` the second, :165-168).  Parameters and member maps go
 * to the device once; per step: observations (ref get_obs :246-263), MLP.get_action (agents/mlp.py:97-116),
 * update_agents, forward; the step's reward = state * (state > 0) and done = reward < 0.1 (ref :486-492)
 * come back as [K][B][N].  member_a / member_b may be NULL when n_members == 1.  L_init: luminosity of the
 * observations' temperature channels if no step has been taken yet.  Worlds of H*W <= 4096 cells run the chunk as
 * ONE launch with the worlds resident in LDS (csrc/dw_episode.hpp: episode_mlp) once the state and the retained
 * previous state are quantised - i.e. from the third step of an episode on; larger worlds and the first two
 * steps take one launch sequence per step.  Same results either way.
 * ABI 4: `params` may be NULL - the parameter sets of the last call on this handle that passed them stay on the device
 * and are used again (n_members must be the same): a fitness harness uploads a population once per generation, not
 * once per chunk.  `reward` / `done` in page-locked memory (dw_pinned_alloc) are filled without a staging copy. */
int dw_run_episode_mlp(dw_handle* h, int32_t nsteps, const double* L_schedule, const double* params /* [n_members][1808] */,
                       int32_t n_members, const int32_t* member_a /* [B] */, const int32_t* member_b /* [B] */,
                       int32_t split, double L_init, double* reward /* [K][B][N] */, uint8_t* done /* [K][B][N] */);

/* Device-side snapshot of the handle's state (current planes, the retained previous state that
 * observations and caches are derived from, agents, per-world reductions) and its restoration: lets an episode harness run chunks of steps ahead (dw_run_episode) and, when the episode
 * turns out to have ended inside a chunk, replay exactly the steps the reference's loop
 * (notebooks/greedy_longevity_abatement.ipynb cell 2:28-57: stop as soon as every world is dead) would
 * have executed — without moving the state over PCIe.  Costs up to two extra copies of the two planes
 * in HBM, allocated at the first save. */
int dw_snapshot_save(dw_handle* h);               /* = slot 0 */
int dw_snapshot_restore(dw_handle* h);
/* Two slots (ABI 5): a harness that lets the device run chunk c + 1 while the host still accounts for chunk c
 * (therldaisyworld_amd/harness.py, the ES fitness episodes: ref daisy/evo/sges.py:144-181) saves the start of chunk c in
 * slot c % 2 - when chunk c turns out to end the episode, the state at its start is still there although chunk c + 1 has
 * saved its own.  A save is one kernel launch for all regions of the state. */
#define DW_SNAPSHOT_SLOTS 2
int dw_snapshot_save_slot(dw_handle* h, int32_t slot);
int dw_snapshot_restore_slot(dw_handle* h, int32_t slot);

/* Device-resident lifespan harness (ref notebooks/greedy_longevity_abatement.ipynb cell 2:28-57):
 * accumulate done_at[b] += (max_k > threshold_k) and agents_done_at[b][n] += !(done) after each
 * step, on the device.  dw_lifespan_reset zeroes them. */
int dw_lifespan_reset(dw_handle* h);
int dw_lifespan_accumulate(dw_handle* h, uint32_t threshold_k);
int dw_lifespan_download(dw_handle* h, int32_t* done_at /* [B] */, int32_t* agents_done_at /* [B][N] */,
                         int32_t* n_worlds_alive);

/* Device-resident episode loop: K consecutive environment steps — policy,
 * update_agents (ref :181-244), forward (ref :434-461), reductions — without a host round trip per
 * step.  Small worlds (H*W <= 4096) run in ONE launch with the worlds held in LDS (H*W <= 256 - the reference's own 8x8
 * and 16x16 worlds - one wave per world with no workgroup barrier in the step: csrc/dw_episode_wave.hpp); larger worlds run the
 * same K steps as back-to-back launches on the handle's stream (policy / table slice, update_agents, the
 * streaming step kernel, flags from its reductions) with one synchronisation at the end.  It is the body of the
 * reference's lifespan harness (notebooks/greedy_longevity_abatement.ipynb cell 2:28-57) and of
 * sges.get_fitness (daisy/evo/sges.py:144-181) for scripted policies.
 *   L_schedule[K]   luminosity of each step (the caller runs the ref update_L recurrence :463-473)
 *   policy_mode     DW_POLICY_ARGMAX / DW_POLICY_ARGMIN (ref Greedy, agents/greedy.py:18-30),
 *                   DW_POLICY_ZEROS (ref step(None): action 0), DW_POLICY_TABLE (all actions given)
 *   use_table[K]    per step: 1 = this step's actions come from `table` (Greedy's epsilon branch,
 *                   drawn by the caller from the legacy NumPy stream); may be NULL (all 0)
 *   table[K][B][N]  int8 codes: 0..8 an action, -1 / -2 the greedy / anti-greedy choice of that agent at
 *                   that step (mixed-policy ensembles, BASELINE configs[4]); may be NULL if never used
 *   world_alive[K][B], agent_ok[K][B][N]   per-step flags out: max cover > threshold_k/1000, and
 *                   reward >= 0.1 (what the harness adds to done_at / agents_done_at)
 * world_alive may be NULL: without per-step world reductions wide grids (and big ensembles of narrow
 * ones) run step PAIRS as one fused launch, with the agents' step in between recomputed around the
 * agents and patched into the result (csrc/dw_agents_fused.hpp) - same results, bit for bit.
 * Needs a quantised current state (take the first step of an episode with dw_step).
 * Afterwards the handle is exactly as after K calls of dw_step (previous state retained; the device action buffer
 * - dw_download_actions - holds the resolved action codes of the LAST step, whichever kernel applied them). */
enum { DW_POLICY_ZEROS = 2, DW_POLICY_TABLE = 3 };
int dw_run_episode(dw_handle* h, int32_t nsteps, const double* L_schedule, int policy_mode,
                   const uint8_t* use_table, const int8_t* table, uint32_t threshold_k, uint8_t* world_alive,
                   uint8_t* agent_ok);

/* ---- plumbing ------------------------------------------------------------------------------- */

/* Use an existing HIP stream (e.g. torch's current stream) instead of the handle's own. */
int dw_set_stream(dw_handle* h, void* hip_stream);
int dw_sync(dw_handle* h);

/* HIP-event timing on the handle's stream: ms between start and stop (stop synchronises). */
int dw_timer_start(dw_handle* h);
int dw_timer_stop(dw_handle* h, float* elapsed_ms);

/* Raw device pointers of the current / previous planes (per-mille binary16, [B][H][W]) for zero-copy interop;
 * DW_ESTATE while the current state is an un-quantised upload (no binary16 planes before the first step). */
int dw_device_planes(dw_handle* h, int which, void** light, void** dark);

/* Name and geometry of the step kernel the handle dispatches for its shape, for bench/profiles:
 * writes a NUL-terminated description into buf. */
int dw_kernel_info(dw_handle* h, char* buf, size_t buflen);

/* Diagnostics for the exact-mode error bound: number of cells re-evaluated in float64 by the last
 * step (summed over worlds). */
int dw_last_fixup_count(dw_handle* h, uint64_t* count);

/* Audit of the exact mode's error bound on the CURRENT (quantised) state at luminosity L: evaluates
 * every cell's per-mille growth in the kernels' float32 arithmetic and in float64 and returns
 *   out[0] max |gq_f32 - gq_f64| in quanta, out[1] max of that error divided by the cell's bound
 *   eps (the bound holds iff < 1), out[2] cells the tie test would flag, out[3] cell-values audited.
 * Used by the tests to show the analytic bound of DESIGN.md is never approached. */
int dw_audit_tie_bound(dw_handle* h, double L, double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* DAISYWORLD_HIP_H */
