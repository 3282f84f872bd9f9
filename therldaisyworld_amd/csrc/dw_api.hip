// dw_api.hip — C ABI (include/daisyworld_hip.h) over the gfx950 kernels in dw_kernels.hpp.
//
// Host-side responsibilities: device state (ping-pong binary16 per-mille planes, agents, reductions),
// per-step derivation of the float32 coefficient set from the float64 constants and the current
// luminosity, kernel selection by grid shape, and the bookkeeping of the un-quantised initial state
// (float64 or float32 buffers that live until the first step has consumed them).  No CPU compute path
// exists here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/daisyworld_hip.h"
#include "dw_kernels.hpp"
#include "dw_host_util.hpp"

using namespace dw;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? DW_ENOMEM : DW_EHIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                      \
    } while (0)

#define NEED(cond, code, ...)                      \
    do {                                           \
        if (!(cond)) return fail(code, __VA_ARGS__); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
// An un-quantised state (the reference's initialize_grid does not round, ref :285-324) cannot live in the
// canonical binary16 planes.  It is held in its upload format - float64 natural units (dw_upload_state_f64) or
// float32 per-mille (dw_init_random, dw_upload_state_f32 with quantised = 0) - and is the CURRENT state until
// the first step has read it, then for one more step the PREVIOUS state (observations and the materialised
// grid derive their temperature channels from the pre-step state).  While it is the current state the
// binary16 planes of the `cur` buffer are undefined.
enum UnqKind { UNQ_F64 = 1, UNQ_F32 = 2 };
enum UnqOwner { OWN_NONE = 0, OWN_CUR = 1, OWN_PREV = 2 };

// Experiment and test switches (environment variables), read ONCE per handle at dw_create - a handle behaves the same for
// its whole life whatever the environment does afterwards - and reported by dw_kernel_info when any is set.  None changes
// results.  The DW_TEST_* hooks (shrunk queues, injected allocation failures, forced fallbacks) are honoured only when
// DW_TEST_HOOKS is set as well: a stray variable in a user's environment cannot inject failures.
struct Switches {
    bool no_sym = false, no_pack = false, no_fuse = false, no_ring = false, no_episode_kernel = false,
         no_episode_wave = false, no_agent_fuse = false, no_agent_preapply = false, first_f64 = false,
         first_generic = false, force_rescan = false, test_hooks = false;
    int pack_min_strips = -1, strip_rows = 0, tile_rpt = 0, queue_cap = -1, mismatch_cap = -1;
    double first_slack = -1.0;
    char kernel[16] = {0};           // DW_KERNEL: "tiled" | "stream"
    char text[256] = {0};            // what was set, for dw_kernel_info
};
static const char* test_hook(const char* name) {              // a DW_TEST_* variable, only under DW_TEST_HOOKS
    return std::getenv("DW_TEST_HOOKS") ? std::getenv(name) : nullptr;
}
static Switches read_switches() {
    Switches w;
    auto note = [&](const char* name, const char* val) {
        const size_t n = std::strlen(w.text);
        snprintf(w.text + n, sizeof(w.text) - n, "%s%s%s%s", n ? " " : "", name, val ? "=" : "", val ? val : "");
    };
    auto flag = [&](const char* name, bool& dst) { if (std::getenv(name)) { dst = true; note(name, nullptr); } };
    auto num = [&](const char* name, int& dst, bool test) {
        if (const char* e = test ? test_hook(name) : std::getenv(name)) { dst = std::atoi(e); note(name, e); }
    };
    w.test_hooks = std::getenv("DW_TEST_HOOKS") != nullptr;
    flag("DW_NO_SYM", w.no_sym); flag("DW_NO_PACK", w.no_pack); flag("DW_NO_FUSE", w.no_fuse); flag("DW_NO_RING", w.no_ring);
    flag("DW_NO_EPISODE_KERNEL", w.no_episode_kernel); flag("DW_NO_EPISODE_WAVE", w.no_episode_wave);
    flag("DW_NO_AGENT_FUSE", w.no_agent_fuse); flag("DW_NO_AGENT_PREAPPLY", w.no_agent_preapply);
    flag("DW_FIRST_STEP_F64", w.first_f64); flag("DW_FIRST_GENERIC", w.first_generic);
    num("DW_PACK_MIN_STRIPS", w.pack_min_strips, false); num("DW_STRIP_ROWS", w.strip_rows, false);
    num("DW_TILE_RPT", w.tile_rpt, false);
    num("DW_TEST_QUEUE_CAP", w.queue_cap, true); num("DW_TEST_MISMATCH_CAP", w.mismatch_cap, true);
    if (test_hook("DW_TEST_FORCE_RESCAN")) { w.force_rescan = true; note("DW_TEST_FORCE_RESCAN", nullptr); }
    if (const char* e = test_hook("DW_TEST_FIRST_SLACK")) { w.first_slack = std::atof(e); note("DW_TEST_FIRST_SLACK", e); }
    if (const char* e = std::getenv("DW_KERNEL")) { snprintf(w.kernel, sizeof(w.kernel), "%s", e); note("DW_KERNEL", e); }
    return w;
}

struct dw_handle {
    dw_params prm;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t cells = 0;                 // B*H*W
    plane_t* L16[2] = {nullptr, nullptr};   // canonical planes (binary16 per-mille integers), ping-pong
    plane_t* D16[2] = {nullptr, nullptr};
    int cur = 0;
    double* L64 = nullptr;            // un-quantised state, float64 natural units (lazily allocated, kept)
    double* D64 = nullptr;
    float* U32L = nullptr;            // un-quantised state, float32 per-mille (lazily allocated; freed again
    float* U32D = nullptr;            //   after use when it is large, see release_unquantised)
    UnqKind unq_kind = UNQ_F64;
    UnqOwner unq = OWN_NONE;
    bool have_state = false;
    bool stepped = false;             // prev/cur form a forward() pair
    double L_last = 0.0;              // luminosity of the last forward()
    int* idx = nullptr;               // [B][N][2]
    double* st = nullptr;             // [B][N]
    int* action = nullptr;            // [B][N]
    int* action_tmp = nullptr;        // staging for host-supplied (possibly sub-shaped) actions
    bool have_agents = false;
    // per-world reductions, double-buffered: each step kernel accumulates into stats2[1-sp] and
    // clears stats2[sp] for the step after it, so the step loop needs no memset launches.
    // Element [B] of each buffer carries the float64 fix-up counter (in sum_l).
    // Each buffer is one allocation: [(B+1) StatsDev][kNumQueues*16 uint queue counters].
    StatsDev* stats2[2] = {nullptr, nullptr};
    size_t stats_bytes = 0;           // bytes of one such buffer
    int sp = 0;                       // buffer holding the CURRENT state's reductions
    uint4* fixq = nullptr;            // exact mode: global near-tie queues [kNumQueues][qcap][3]
    unsigned int qcap = 0;
    int* redo_tiles = nullptr;        // exact mode: tiles to recompute whole (queue overflow)
    // streaming kernel (W >= 256)
    bool use_stream = false;
    Switches sw{};                    // experiment / test switches as they were when the handle was created
    StripGeom sgeom{};
    bool allow_fuse = false;          // wide grids: dw_step_n / dw_run_episode fuse pairs of steps
    bool fused_ring = false;          // W == 1024: the four waves of a workgroup form a ring over the torus row
    bool sym_albedo = false;          // a_dark - a_bare == -(a_light - a_bare) exactly: the exact wave-strip kernels use
                                      // the two-term coefficient chain (growth_t<.., SYM>)
    FusedGeom fgeom{};
    int* done_at = nullptr;           // [B]
    int* agents_done_at = nullptr;    // [B][N]
    int* n_alive = nullptr;
    double* scratch = nullptr;        // device staging for float64 downloads / uploads
    size_t scratch_bytes = 0;
    unsigned char* ep_buf = nullptr;  // device staging of dw_run_episode (schedules, tables, flags)
    size_t ep_bytes = 0;
    double* mlp_w = nullptr;              // parameter sets of the last dw_run_episode_mlp call that passed them
    int mlp_members = 0;
    unsigned char* ep_pinned = nullptr;   // page-locked host image of ep_buf (LDS-resident episode kernels: ONE upload
    size_t ep_pinned_bytes = 0;           // and ONE download per chunk instead of six pageable copies)
    double* reward_d = nullptr;       // [B][N]
    unsigned char* done_d = nullptr;  // [B][N]
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t evf0 = nullptr, evf1 = nullptr;   // around the fused launches of the last dw_step_n call
    int fused_launches = 0;           // ... and how many there were (dw_last_step_n_timing); 0 unless BOTH events
                                      // of that call were recorded (an error return in between leaves 0)
    StatsDev* side_stats = nullptr;   // reductions of dw_forward_f64's side computation (not the handle's)
    unsigned char* pinned = nullptr;  // page-locked host staging of dw_env_step (actions in, obs/reward/done out)
    size_t pinned_bytes = 0;
    // dw_snapshot_save[_slot] / dw_snapshot_restore[_slot]: device copies of the current state (two slots: a harness
    // that runs chunk c + 1 while it still accounts for chunk c keeps the starts of both)
    struct Snapshot {
        plane_t* L = nullptr;
        plane_t* D = nullptr;
        plane_t* PL = nullptr;        // the retained previous state (observations, caches) when there is one
        plane_t* PD = nullptr;
        bool stepped = false;
        double L_last = 0.0;
        UnqOwner unq = OWN_NONE;
        int* idx = nullptr;
        double* st = nullptr;
        unsigned char* stats = nullptr;
        bool valid = false, agents = false;
    } snap[DW_SNAPSHOT_SLOTS];
    // kernel selection
    int tcq = 0, rpt = 0;             // 0 => generic
    Geom geom{};
    size_t tile_lds = 0;
};

static inline bool cur_quantised(const dw_handle* h) { return h->unq != OWN_CUR; }

// Asynchronous copies FROM host memory (the caller's arrays, local vectors) must have finished before that
// memory can go away: a function that issues them declares one of these right after the host buffers, so that
// every early return (HIPCHK / NEED) waits for the stream first.  disarm() after the function's own final
// synchronisation.
struct SyncOnExit {
    hipStream_t stream;
    bool armed = true;
    explicit SyncOnExit(hipStream_t s) : stream(s) {}
    ~SyncOnExit() { if (armed) (void)hipStreamSynchronize(stream); }
    void disarm() { armed = false; }
};

// the un-quantised float32 buffers are as large as all four canonical planes together: give them back once
// nothing refers to them any more if they are big (the north-star shape: 128 GiB)
static void release_unquantised(dw_handle* h) {
    if (h->unq != OWN_NONE || !h->U32L) return;
    for (const auto& sn : h->snap)
        if (sn.valid && sn.unq == OWN_PREV) return;              // a snapshot's previous state lives there
    if (h->cells * 2 * sizeof(float) < ((size_t)1 << 30)) return;
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->U32L); (void)hipFree(h->U32D);
    h->U32L = nullptr; h->U32D = nullptr;
}

// The two planes of an un-quantised state come and go together (dw_host_util.hpp): a failed second allocation
// leaves NEITHER, so a retry on the same handle reports DW_ENOMEM again instead of launching on a null plane.
// DW_TEST_FAIL_PAIR_ALLOC=<n> (tests, under DW_TEST_HOOKS; process-wide countdown): the second allocation of the next n
// pairs fails with out-of-memory.
static int alloc_plane_pair(void** a, void** b, size_t bytes) {
    static int fail_left = [] { const char* e = test_hook("DW_TEST_FAIL_PAIR_ALLOC"); return e ? std::atoi(e) : 0; }();
    int calls = 0;
    hipError_t last = hipSuccess;
    const int rc = alloc_pair_all_or_nothing(
        a, b, bytes,
        [&](void** p, size_t n) {
            if (++calls == 2 && fail_left > 0) { --fail_left; last = hipErrorOutOfMemory; *p = nullptr; return 1; }
            last = hipMalloc(p, n);
            return last == hipSuccess ? 0 : 1;
        },
        [](void* p) { return hipFree(p) == hipSuccess ? 0 : 1; });
    if (rc == 0) return DW_OK;
    (void)hipGetLastError();                                     // the failed hipMalloc must not poison later checks
    return fail(last == hipErrorOutOfMemory ? DW_ENOMEM : DW_EHIP, "allocating two planes of %zu bytes failed: %s",
                bytes, hipGetErrorString(last));
}

static int ensure_u32(dw_handle* h) {
    return alloc_plane_pair(reinterpret_cast<void**>(&h->U32L), reinterpret_cast<void**>(&h->U32D),
                            sizeof(float) * h->cells);
}

static int ensure_f64(dw_handle* h) {
    return alloc_plane_pair(reinterpret_cast<void**>(&h->L64), reinterpret_cast<void**>(&h->D64),
                            sizeof(double) * h->cells);
}

static int run_episode_impl(dw_handle* h, int32_t nsteps, const double* L_schedule, int policy_mode,
                            const uint8_t* use_table, const int8_t* table, uint32_t threshold_k,
                            uint8_t* world_alive, uint8_t* agent_ok);
static bool episode_kernel_applies(const dw_handle* h);
static int observe_into_scratch(dw_handle* h, double L_init, size_t extra_bytes, bool reward_tail = false);

static int ensure_scratch(dw_handle* h, size_t bytes) {
    if (h->scratch_bytes >= bytes) return DW_OK;
    if (h->scratch) HIPCHK(hipFree(h->scratch));
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    HIPCHK(hipMalloc(&h->scratch, bytes));
    h->scratch_bytes = bytes;
    return DW_OK;
}

// episode staging buffer (schedules, tables, per-step flags): grown geometrically from 4 MiB so that a
// longer chunk after a short one does not pay a synchronous hipFree + hipMalloc inside a timed run
static int ensure_ep_buf(dw_handle* h, size_t bytes) {
    if (h->ep_bytes >= bytes) return DW_OK;
    size_t want = h->ep_bytes ? h->ep_bytes * 2 : ((size_t)4 << 20);
    if (want < bytes) want = bytes;
    if (h->ep_buf) HIPCHK(hipFree(h->ep_buf));
    h->ep_buf = nullptr;
    h->ep_bytes = 0;
    HIPCHK(hipMalloc(&h->ep_buf, want));
    h->ep_bytes = want;
    return DW_OK;
}

// near-tie queues of the exact mode: room for 1/64 of all cells (the bound flags ~0.3-0.5 %), at
// least 2048 entries per queue; 48 bytes per entry, i.e. 0.75 B per cell on top of the 16 B of state
static int ensure_fixq(dw_handle* h) {
    const dw_params& p = h->prm;
    if (p.precision != DW_PRECISION_EXACT) return DW_OK;
    if (h->use_stream) return DW_OK;        // the streaming kernel keeps its near-tie queues in LDS
    if (h->fixq || p.width % 4 != 0 || p.width < 64) return DW_OK;
    size_t per_q = (h->cells / 64 + kNumQueues - 1) / kNumQueues;
    if (per_q < 2048) per_q = 2048;
    per_q = (per_q + 255) / 256 * 256;
    h->qcap = (unsigned int)per_q;
    HIPCHK(hipMalloc(&h->fixq, sizeof(uint4) * 3 * per_q * kNumQueues));
    const size_t max_tiles = (size_t)p.batch * ((p.height + 7) / 8) * ((p.width / 4 + 15) / 16);
    HIPCHK(hipMalloc(&h->redo_tiles, sizeof(int) * max_tiles));
    return DW_OK;
}

// ------------------------------------------------------------------------------------------------
// constants: float64 set and the per-step float32 set
// ------------------------------------------------------------------------------------------------
static PhysF64 make_f64(const dw_params& p, double L) {
    PhysF64 P;
    P.p = p.p; P.g = p.g; P.S = p.S; P.sigma = p.sigma; P.gamma = p.gamma; P.q = p.q; P.q2 = p.q2;
    P.dt = p.dt; P.ab = p.albedo_bare; P.al = p.albedo_light; P.ad = p.albedo_dark;
    P.To = p.temp_optimal; P.L = L;
    // ref daisy_world_rl.py:270-273: ones*e^-1, centre 1, corners e^-2, normalised
    const double e1 = std::exp(-1.0), e2 = std::exp(-2.0);
    const double s = 1.0 + 4.0 * e1 + 4.0 * e2;
    P.w0 = 1.0 / s; P.w1 = e1 / s; P.w2 = e2 / s;
    return P;
}

#ifndef DW_TIE_BOUND
#define DW_TIE_BOUND 3
#endif
// kbeta = 1 / sqrt(g To^2) in float64 (g = 0: a denominator so large that w^2 vanishes)
static double cbeta_host(const dw_params& p) {
    const double cbeta = p.g * p.temp_optimal * p.temp_optimal;
    return cbeta > 1e-30 ? 1.0 / std::sqrt(cbeta) : 0x1p60;
}

static void split_hi_lo(double v, double scale, float* hi, float* lo) {
    const double h = std::nearbyint(v * scale) / scale;
    *hi = (float)h;
    *lo = (float)(v - h);
}

// Derivation of the fused float32 coefficients (see dw_physics.hpp, PhysF32) and of the exact-mode
// tie bound (DESIGN.md §"Exact mode").  All in float64, rounded once.
static PhysF32 derive_f32(const dw_params& p, double L, int hb_cap = 40) {
    const double To4 = std::pow(p.temp_optimal, 4);
    const double K = p.S * L / p.sigma;
    const double dal = p.albedo_light - p.albedo_bare, dad = p.albedo_dark - p.albedo_bare;
    const double a1 = (p.q - K) * dal / (8000.0 * To4);
    const double a2 = (p.q - K) * dad / (8000.0 * To4);
    const double a3 = (-p.q + p.q2) * dal / (1000.0 * To4);
    const double a4 = (-p.q + p.q2) * dad / (1000.0 * To4);
    const double e0 = K * (1.0 - p.albedo_bare * p.p) / To4 - 1.0;
    const double c0l = e0 + p.q2 * (p.albedo_bare * p.p - p.albedo_light) / To4;
    const double c0d = e0 + p.q2 * (p.albedo_bare * p.p - p.albedo_dark) / To4;
    // hi parts are multiples of 2^-hb with hb chosen so that every partial sum of the hi chain
    // (|.| <= maxsum) stays below 2^(24-hb): exactly representable in float32 for integer inputs.
    const double kmax = 1000.0;                                  // covers are clipped to [0, 1] whatever p is (ref :449)
    const double maxsum = std::fabs(a1) * 8 * kmax + std::fabs(a2) * 8 * kmax + std::fabs(a3) * kmax +
                          std::fabs(a4) * kmax + std::fmax(std::fabs(c0l), std::fabs(c0d));
    int hb = 23 - (int)std::ceil(std::log2(std::fmax(maxsum, 1e-30)));
    if (hb > hb_cap) hb = hb_cap;                               // a coarser split is always admissible
    if (hb < 0) hb = 0;
    const double scale = std::ldexp(1.0, hb);
    PhysF32 P;
    P.hi_bits = hb;
    split_hi_lo(a1, scale, &P.a1h, &P.a1l);
    split_hi_lo(a2, scale, &P.a2h, &P.a2l);
    split_hi_lo(a3, scale, &P.a3h, &P.a3l);
    split_hi_lo(a4, scale, &P.a4h, &P.a4l);
    split_hi_lo(c0l, scale, &P.c0lh, &P.c0ll);
    float c0dl;
    split_hi_lo(c0d, scale, &P.c0dh, &c0dl);
    P.dc0l = c0dl - P.c0ll;                                      // the lo chain carries light's lo constant
    // float32-only mode: every coefficient rounded once (= fl(hi + lo): the sum of two floats is exact in double)
    P.a1 = (float)((double)P.a1h + P.a1l); P.a2 = (float)((double)P.a2h + P.a2l);
    P.a3 = (float)((double)P.a3h + P.a3l); P.a4 = (float)((double)P.a4h + P.a4l);
    P.c0ls = (float)((double)P.c0lh + P.c0ll); P.c0ds = (float)((double)P.c0dh + c0dl);
    // kbeta = 1 / sqrt(g * To^2) (dw_physics.hpp); g = 0 (beta = 1 everywhere): a denominator so large that w^2
    // vanishes.  g < 0 is refused for the float32 modes by check_params.
    {
        const double cbeta = p.g * p.temp_optimal * p.temp_optimal;
        P.kbeta = cbeta > 1e-30 ? (float)(1.0 / std::sqrt(cbeta)) : 0x1p60f;
    }
    const PhysF64 P64 = make_f64(p, L);
    // dt folded into the density weights (dK = dt * density is what the map needs); the bare fraction then is
    // kb = p - (dKl + dKd) * 0.001 / dt  (dt = 0: no growth at all - weights 0, kb = p)
    P.dw0 = (float)(p.dt * P64.w0); P.dw1 = (float)(p.dt * P64.w1); P.dw2 = (float)(p.dt * P64.w2);
    P.p = (float)p.p; P.ck = p.dt != 0.0 ? (float)(0.001 / p.dt) : 0.f;
    P.ngamma = (float)(-p.gamma);
    // ---- tie bound (per-mille), DESIGN.md "Exact mode":
    //   |gq32 - gq64| <= A0 + eA*|gq| + |dt*K| * (eK0 + eK1*om),   om = 1 - beta >= 0
    const double u = std::ldexp(1.0, -24);
    // absolute error of e: only the lo chain rounds (hi chain exact); lo coefficients <= 2^-(hb+1)
    const double lo_mag = std::ldexp(1.0, -(hb + 1)) * 18.0 * kmax + std::ldexp(1.0, -(hb + 1));
    const double de_abs = 6.0 * u * lo_mag + 1e-9;
    // admissible interval of e over all states with covers in [0, kmax] (signs of the coefficients respected)
    // (light and dark cover are clipped separately: both may be full in the same cell)
    const double pos = (std::fmax(a1, 0.0) + std::fmax(a2, 0.0)) * 8 * kmax + (std::fmax(a3, 0.0) + std::fmax(a4, 0.0)) * kmax;
    const double neg = (std::fmin(a1, 0.0) + std::fmin(a2, 0.0)) * 8 * kmax + (std::fmin(a3, 0.0) + std::fmin(a4, 0.0)) * kmax;
    const double emax = std::fmax(c0l, c0d) + pos, emin = std::fmin(c0l, c0d) + neg;
#if DW_TIE_BOUND == 2
    // round-2 constants (kept for A/B runs: -DDW_TIE_BOUND=2)
    const double dabs = std::fmax(std::fabs(std::pow(std::fmax(1.0 + emax, 1e-6), 0.25) - 1.0),
                                  std::fabs(std::pow(std::fmax(1.0 + emin, 1e-6), 0.25) - 1.0));
    const double cb = p.g * p.temp_optimal * p.temp_optimal;
    const double safety = 1.25;
    const double A0 = safety * (0.25 * kmax * std::fabs(p.dt) * cb * dabs * de_abs + 4.0 * u * kmax * 0.25 + 2e-5);
    const double eA = safety * 9.0 * u, eK0d = safety * 5.0 * u, eK1d = safety * 28.0 * u;
#else
    // Round-3 derivation (every step numbered there).  With v = 1 + e, s = sqrt(v), y = sqrt(s),
    // D = kbeta (y+1)(s+1), w = e/D, om = w^2, the hardware's sqrt / rcp at 1 ulp (relative 2u) and every other
    // operation correctly rounded (relative u):
    //   rel(v)  <= u (1 + re) + de/vmin =: ev                 re = max |e| / (1 + e)
    //   rel(D)  <= 3u + (ev/2 + 2u) sg + (ev/4 + 3u) et       sg = max s/(s+1), et = max y/(y+1)
    //   rel(w)  <= 4u + rel(D) + de/|e| = ew + de/|e|
    //   |beta^ - beta| <= 2 ew om + u (1 + om) + 2 de max(|e| / D^2)
    //   |kb^ - kb| <= pu + 6u Dt + u |kb|       (kb = p - Dt, Dt = ck S the summed densities: S carries 5u, ck u; pu = u|p|
    //                                             unless p is a float) <= pu + 6u|p| + 7u|kb| [kb < 0]
    //   |f^ - f| <= (pu + 6u|p|)(1 + om) + 7u (|f| + |gamma|) + kbmax (2 ew om + u (1 + om) + Ae) + u |gamma| + u |f|
    //               (|beta^| <= 1 + om; |kb beta| = |f + gamma|)
    //   |gq^ - gq| <= |K| |f^ - f| + 5u |gq|        (K carries 4u, the product u)
    // kbmax = max |kb| over states with covers in [0, 1]: the two densities sum to [0, 2].
    const double vmin = 1.0 + emin, vmax = 1.0 + emax;
    const bool admissible = vmin > 0.02;                          // otherwise: flag every cell (float64 everywhere)
    const double vlo = std::fmax(vmin, 0.02);
    const double smax = std::sqrt(std::fmax(vmax, vlo));
    const double ymax = std::sqrt(smax);
    const double sg = smax / (smax + 1.0), et = ymax / (ymax + 1.0);
    const double re = std::fmax(std::fabs(emin) / vlo, std::fabs(emax) / std::fmax(vmax, vlo));
    const double ev = u * (1.0 + re) + de_abs / vlo;
    const double eD = 3.0 * u + (0.5 * ev + 2.0 * u) * sg + (0.25 * ev + 3.0 * u) * et;
    const double ew = 4.0 * u + eD;
    const double kbe = cbeta_host(p);                            // kbeta in float64
    auto e_over_D2 = [&](double e) {
        const double v = std::fmax(1.0 + e, vlo), sv = std::sqrt(v), yv = std::sqrt(sv);
        const double D = kbe * (yv + 1.0) * (sv + 1.0);
        return std::fabs(e) / (D * D);
    };
    const double Ae = 2.0 * de_abs * std::fmax(e_over_D2(emin), e_over_D2(emax));
    const double kbmax = std::fmax(std::fabs(p.p), std::fabs(p.p - 2.0));
    const double safety = 1.0 + 0x1p-7;                           // the second-order terms dropped above (~u relative)
    const double pu = (double)(float)p.p == p.p ? 0.0 : std::fabs(p.p);
    double A0 = safety * (kmax * std::fabs(p.dt) * kbmax * Ae + 2e-6);   // 2e-6: the float32 arithmetic of the threshold
    if (!admissible) A0 = 1.0;                                    // tie_lo < 0: every cell goes to float64
    const double eA = safety * 13.0 * u;
    const double eK0d = safety * (6.0 * std::fabs(p.p) + pu + kbmax + 8.0 * std::fabs(p.gamma)) * u;
    const double eK1d = safety * ((6.0 * std::fabs(p.p) + pu) * u + kbmax * (u + 2.0 * ew));
#endif
    auto up = [](double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; };   // never round a bound DOWN
    P.tie_lo = (float)(0.5 - A0);
    if ((double)P.tie_lo > 0.5 - A0) P.tie_lo = std::nextafterf(P.tie_lo, -INFINITY);
    P.eA = up(eA);
    const float eK0 = up(eK0d), eK1 = up(eK1d);
    P.eK0s = p.dt < 0.0 ? eK0 : -eK0;
    P.eK1s = p.dt < 0.0 ? eK1 : -eK1;
    P.neK1s = -P.eK1s;                                           // the bracket in beta: (eK0s + eK1s) - eK1s*beta
    P.eK01s = (float)((double)P.eK0s + (double)P.eK1s);
    if (std::fabs((double)P.eK01s) < std::fabs((double)P.eK0s + (double)P.eK1s))     // never round the bound DOWN
        P.eK01s = std::nextafterf(P.eK01s, P.eK01s < 0.f ? -1.f : 1.f);
    return P;
}

// Coefficient sets of the two steps of a fused exact launch, split at the SAME scale (the coarser of
// the two) and with the more conservative of the two tie thresholds: they then differ only in the
// luminosity-dependent members (PhysLumF32).
static void derive_f32_pair(const dw_params& p, double L1, double L2, PhysF32* P1, PhysF32* P2) {
    const PhysF32 a = derive_f32(p, L1), b = derive_f32(p, L2);
    const int hb = a.hi_bits < b.hi_bits ? a.hi_bits : b.hi_bits;
    *P1 = derive_f32(p, L1, hb);
    *P2 = derive_f32(p, L2, hb);
    const float tie_lo = P1->tie_lo < P2->tie_lo ? P1->tie_lo : P2->tie_lo;
    P1->tie_lo = tie_lo;
    P2->tie_lo = tie_lo;
}

// Error bound of the float32 map on an UN-quantised state (step_generic<In, 3>, dw_step_generic.hpp; the first step of
// an episode in the exact mode).  Same chain of estimates as derive_f32's round-3 bound with: iota = u for a float64
// state (its values are rounded to float32 on the way in; the float64 re-evaluation reads the originals), 0 for a
// float32 state; stencil sums of non-integers round (2u per sum of four, 3u for the sum of eight); the density
// carries 5u + iota; the coefficient chain is the rounded one (no exact hi part): its absolute error is
// (8u + iota) M + u |c0| with the per-cell M the kernel evaluates.
static FirstStepBound derive_first_bound(const dw_params& p, double L, const PhysF32& P, bool from_f64,
                                         double test_slack = -1.0) {
    const double u = std::ldexp(1.0, -24), iota = from_f64 ? u : 0.0;
    const double kmax = 1000.0;
    const double To4 = std::pow(p.temp_optimal, 4), K = p.S * L / p.sigma;
    const double e0 = K * (1.0 - p.albedo_bare * p.p) / To4 - 1.0;
    const double c0l = e0 + p.q2 * (p.albedo_bare * p.p - p.albedo_light) / To4;
    const double c0d = e0 + p.q2 * (p.albedo_bare * p.p - p.albedo_dark) / To4;
    const double a1 = std::fabs((double)P.a1), a2 = std::fabs((double)P.a2), a3 = std::fabs((double)P.a3), a4 = std::fabs((double)P.a4);
    const double c0m = std::fmax(std::fabs(c0l), std::fabs(c0d));
    // admissible interval of e (both covers may be full in the same cell) and the worst-case absolute error of e
    const double pos = (std::fmax((double)P.a1, 0.0) + std::fmax((double)P.a2, 0.0)) * 8 * kmax +
                       (std::fmax((double)P.a3, 0.0) + std::fmax((double)P.a4, 0.0)) * kmax;
    const double neg = (std::fmin((double)P.a1, 0.0) + std::fmin((double)P.a2, 0.0)) * 8 * kmax +
                       (std::fmin((double)P.a3, 0.0) + std::fmin((double)P.a4, 0.0)) * kmax;
    const double emax = std::fmax(c0l, c0d) + pos, emin = std::fmin(c0l, c0d) + neg;
    const double Mmax = (a1 + a2) * 8 * kmax + (a3 + a4) * kmax;
    const double de_max = (8.0 * u + iota) * Mmax + u * c0m;
    const double vmin = 1.0 + emin, vmax = 1.0 + emax;
    const bool admissible = vmin > 0.02;
    const double vlo = std::fmax(vmin, 0.02);
    const double smax = std::sqrt(std::fmax(vmax, vlo)), ymax = std::sqrt(smax);
    const double sg = smax / (smax + 1.0), et = ymax / (ymax + 1.0);
    const double re = std::fmax(std::fabs(emin) / vlo, std::fabs(emax) / std::fmax(vmax, vlo));
    const double ev = u * (1.0 + re) + de_max / vlo;
    const double eD = 3.0 * u + (0.5 * ev + 2.0 * u) * sg + (0.25 * ev + 3.0 * u) * et;
    const double ew = 4.0 * u + eD;
    const double kbe = cbeta_host(p);
    const double Dmin = kbe * (std::sqrt(std::sqrt(vlo)) + 1.0) * (std::sqrt(vlo) + 1.0);
    const double kbmax = std::fmax(std::fabs(p.p), std::fabs(p.p - 2.0));
    const double pu = (double)(float)p.p == p.p ? 0.0 : std::fabs(p.p);
    const double safety = 1.0 + 0x1p-7;
    const double kK = 5.0 * u + iota;                             // relative error of a density
    auto up = [](double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; };
    FirstStepBound B;
    B.a1 = up(a1); B.a2 = up(a2); B.a3 = up(a3); B.a4 = up(a4);
    B.c_de = up(safety * (8.0 * u + iota) * (1.0 + 8.0 * u));     // (1 + 8u): M itself is a float32 chain
    B.c_c0 = up(safety * u * c0m);
    // |kb^ - kb| <= pu + (kK + 2u)|p| + (kK + 3u)|kb| [kb < 0];  |gq^ - gq| <= |K| |f^ - f| + (kK + u)|gq|
    B.eK0 = up(safety * (pu + (kK + 2.0 * u) * std::fabs(p.p) + kbmax * u + ((kK + 3.0 * u) + u) * std::fabs(p.gamma)));
    B.eK1 = up(safety * (pu + (kK + 2.0 * u) * std::fabs(p.p) + kbmax * (u + 2.0 * ew)));
    B.cW = up(safety * 2.0 * kbmax / Dmin);                       // 2 de |e| / D^2 <= 2 de |w| / Dmin
    B.eA = up(safety * ((kK + 3.0 * u) + u + (kK + u)));
    B.cS = up(safety * u);                                        // the sum k + gq rounds (u |sum|); a float64 k was rounded (u k)
    B.slack = admissible ? 4e-6f : 1.0f;                          // the float32 arithmetic of eps itself; inadmissible: all float64
    if (test_slack > (double)B.slack) B.slack = (float)test_slack;   // tests: a wider bracket flags many cells (list sweeps)
    return B;
}

// ------------------------------------------------------------------------------------------------
// kernel selection
// ------------------------------------------------------------------------------------------------
// strip counts / grid sizes of the wave-strip kernels for strips of `sr` rows
static void set_strip_rows(dw_handle* h, int sr) {
    const dw_params& p = h->prm;
    StripGeom& g = h->sgeom;
    FusedGeom& f = h->fgeom;
    const bool packed = g.wpr > 1 || p.width < 256;
    const long groups = packed ? (p.batch + g.wpr - 1) / g.wpr : p.batch;
    g.SR = f.SR = p.height < sr ? p.height : sr;
    g.nrs = f.nrs = (p.height + g.SR - 1) / g.SR;
    g.nstrips = (int)(groups * g.nrs * g.ncs);
    g.nwg = (g.nstrips + 3) / 4;
    g.chunk = (g.nwg + 7) / 8;
    f.nstrips = (int)(groups * f.nrs * f.ncs);
    f.nwg = h->fused_ring ? f.nstrips : (f.nstrips + 3) / 4;
    f.chunk = (f.nwg + 7) / 8;
}

static void select_kernel(dw_handle* h) {
    const dw_params& p = h->prm;
    h->tcq = 0;
    h->rpt = 0;
    h->use_stream = false;
    h->allow_fuse = false;
    h->sym_albedo = (p.albedo_dark - p.albedo_bare) == -(p.albedo_light - p.albedo_bare) && !h->sw.no_sym;
    if (p.precision == DW_PRECISION_F64) return;
    if (p.width % 4 != 0) return;
    const int Wq = p.width / 4;
    const char* force = h->sw.kernel[0] ? h->sw.kernel : nullptr;   // "tiled" | "stream": A/B experiments
    // packed mode of the wave-strip kernels: narrow worlds whose width divides 256 sit side by side in one
    // 256-column wave row (256/W worlds per wave)
    // (only for ensembles with enough wave-strips to occupy the GPU: a lone strip is a serial march down
    // 64 rows, ~50 us, where the tiled kernel answers in ~9 us; DW_PACK_MIN_STRIPS overrides for tests)
    int pack_min_strips = 512;
    if (h->sw.pack_min_strips >= 0) pack_min_strips = h->sw.pack_min_strips;
    // any width below 256 that is a multiple of 4, provided at least 70 % of the 64 lanes get columns
    // (W = 96: 2 worlds on 48 lanes; W = 132: one world on 33 lanes - left to the tiled kernel)
    const int pk_lpw = p.width / 4, pk_wpr = pk_lpw ? 64 / pk_lpw : 0;
    const bool pack_shape = p.width >= 8 && p.width < 256 && pk_wpr >= 1 && pk_wpr * pk_lpw * 10 >= 64 * 7 &&
                            !h->sw.no_pack;
    const long pack_strips = pack_shape ? (long)((p.batch + pk_wpr - 1) / pk_wpr) * ((p.height + 63) / 64) : 0;
    const bool packable = pack_shape && pack_strips >= pack_min_strips;
    if ((p.width >= 256 || packable) && !(force && std::strcmp(force, "tiled") == 0)) {
        h->use_stream = true;
        StripGeom& g = h->sgeom;
        g.B = p.batch; g.H = p.height; g.W = p.width;
        g.ncs = (p.width + 255) / 256;
        g.lpw = packable ? p.width / 4 : 64;
        g.wpr = packable ? pk_wpr : 1;
        // Strip height: 64 rows (3 % halo re-reads) when that already gives every SIMD two strips; shorter
        // strips for smaller jobs - a strip is a serial march of ~0.8 us per row, so with few strips the
        // launch takes as long as ONE strip and most SIMDs idle.  DW_STRIP_ROWS overrides (experiments).
        {
            const long groups = (long)(packable ? (p.batch + g.wpr - 1) / g.wpr : p.batch) * g.ncs;
            int sr = 64;
            while (sr > 8 && groups * ((p.height + sr - 1) / sr) < 2048) sr >>= 1;    // two strips per SIMD
            if (h->sw.strip_rows >= 1) sr = h->sw.strip_rows;
            g.SR = p.height < sr ? p.height : sr;
        }
        g.qcap = kWaveQueueCap;
        int mcap = kMismatchCap;
        if (h->sw.queue_cap >= 0 && h->sw.queue_cap < kWaveQueueCap) g.qcap = h->sw.queue_cap;   // tests: force the overflow fallbacks
        if (h->sw.mismatch_cap >= 0 && h->sw.mismatch_cap < kMismatchCap) mcap = h->sw.mismatch_cap;
        g.force_rescan = h->sw.force_rescan ? 1 : 0;            // tests: the strip maximum's re-scan path
        h->allow_fuse = !h->sw.no_fuse;
        FusedGeom& f = h->fgeom;
        f.B = p.batch; f.H = p.height; f.W = p.width;
        f.SR = g.SR;
        f.lpw = g.lpw; f.wpr = g.wpr;
        // W == 1024: one WORKGROUP per row strip, its four waves side by side (edge columns through LDS) instead of
        // five overlapped 248-column strips (DW_NO_RING: experiments)
        h->fused_ring = p.width == 1024 && !h->sw.no_ring;
        f.cols_per_strip = p.width <= 256 ? 256 : (h->fused_ring ? 1024 : 248);
        f.ncs = packable ? 1 : (p.width + f.cols_per_strip - 1) / f.cols_per_strip;
        f.qcap = g.qcap;
        f.mcap = mcap;
        f.sure_need = 9 * p.n_agents + 9 * mcap + 1;            // (dw_step_fused.hpp, STATS)
        set_strip_rows(h, g.SR);
        return;
    }
    if (Wq >= 64) {
        h->tcq = 64; h->rpt = 4;
        if (h->sw.tile_rpt == 2 || h->sw.tile_rpt == 4 || h->sw.tile_rpt == 8) h->rpt = h->sw.tile_rpt;   // tuning experiments only
    }
    else if (Wq >= 32) { h->tcq = 32; h->rpt = 4; }
    else if (Wq >= 16) { h->tcq = 16; h->rpt = 2; }
    else return;   // narrow grids: generic kernel
    const int TR = (256 / h->tcq) * h->rpt;
    Geom& g = h->geom;
    g.B = p.batch; g.H = p.height; g.W = p.width; g.Wq = Wq;
    g.tiles_r = (p.height + TR - 1) / TR;
    g.tiles_c = (Wq + h->tcq - 1) / h->tcq;
    g.ntiles = p.batch * g.tiles_r * g.tiles_c;
    g.chunk = (g.ntiles + 7) / 8;
    g.qcap = kMaxFix;
    if (h->sw.queue_cap >= 0 && h->sw.queue_cap < kMaxFix) g.qcap = h->sw.queue_cap;   // tests: force the overflow fallbacks
    h->tile_lds = (size_t)2 * (TR + 2) * (h->tcq + 2) * 4 * sizeof(float);
}

template <int TCQ, int RPT, bool EXACT>
static int launch_tiled(dw_handle* h, const plane_t* iL, const plane_t* iD, plane_t* oL, plane_t* oD,
                        const PhysF32& P, const PhysF64& P64, StatsDev* stats,
                        unsigned long long* fixups, unsigned long long* zero_me, int zero_n, const FixQ& fq) {
    auto kern = step_tiled<TCQ, RPT, EXACT>;
    // only the 32-row tuning tiles exceed the 64 KB a launch may ask for without opting in; the attribute belongs
    // to the CURRENT device's copy of the kernel, so it is set per call (no process-wide flag: one handle per GPU)
    if constexpr (TileCfg<TCQ, RPT>::LDS_BYTES > 64 * 1024)
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(TileCfg<TCQ, RPT>::LDS_BYTES)));
    const unsigned grid = (unsigned)h->geom.chunk * 8u;
    constexpr size_t lds_bytes = TileCfg<TCQ, RPT>::LDS_BYTES;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, h->stream, iL, iD, oL, oD, h->geom, P, stats,
                       fixups, zero_me, zero_n, fq);
    HIPCHK(hipGetLastError());
    if (EXACT) {
        // second kernel of the step: dense float64 re-evaluation of the queued near-tie cells
        const dim3 g((fq.qcap + 255) / 256, kNumQueues);
        hipLaunchKernelGGL(fixup_cells, g, dim3(256), 0, h->stream, oL, oD, h->prm.height, h->prm.width, P64, stats, fq);
        // third: whole tiles whose queue overflowed (normally none: exits immediately)
        hipLaunchKernelGGL(redo_tiles_f64, dim3(256), dim3(256), 0, h->stream, iL, iD, oL, oD, h->geom,
                           TileCfg<TCQ, RPT>::TR, TCQ, P64, stats, fq);
        HIPCHK(hipGetLastError());
    }
    return DW_OK;
}

// forward(): cur -> other buffer, swap.  Assumes agents were already updated.
static int launch_forward(dw_handle* h, double L) {
    const dw_params& p = h->prm;
    NEED(h->have_state, DW_ESTATE, "no state uploaded (call dw_upload_state_* or dw_init_random)");
    const int in = h->cur, out = 1 - h->cur;
    const PhysF32 P = derive_f32(p, L);
    const PhysF64 P64 = make_f64(p, L);
    StatsDev* stats = h->stats2[1 - h->sp];                       // invariant: all zero
    unsigned long long* fixups = &stats[p.batch].sum_l;
    unsigned long long* zero_me = reinterpret_cast<unsigned long long*>(h->stats2[h->sp]);
    const int zero_n = (int)(h->stats_bytes / sizeof(unsigned long long));
    FixQ fq;
    fq.entries = h->fixq;
    fq.counts = reinterpret_cast<unsigned int*>(stats + p.batch + 1);
    fq.qcap = h->qcap;
    fq.redo_tiles = h->redo_tiles;
    const int gcpt = generic_cells_per_thread(p.batch, (long long)p.height * p.width);
    const dim3 ggrid((unsigned)(((long long)p.height * p.width + 256LL * gcpt - 1) / (256LL * gcpt)), (unsigned)p.batch);
    int prec = p.precision;
#ifdef DW_TUNING
    if (const char* e = std::getenv("DW_ABLATE")) {
        if (std::strcmp(e, "copy") == 0) {
            const size_t n4 = h->cells * sizeof(plane_t) / 16;
            hipLaunchKernelGGL(copy_planes, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, h->stream,
                               reinterpret_cast<const float4*>(h->L16[in]), reinterpret_cast<const float4*>(h->D16[in]),
                               reinterpret_cast<float4*>(h->L16[out]), reinterpret_cast<float4*>(h->D16[out]), n4);
            HIPCHK(hipGetLastError());
            h->cur = out; h->sp = 1 - h->sp; h->stepped = true; h->L_last = L;
            return DW_OK;
        }
        const int v = std::strcmp(e, "nomath") == 0 ? 1 : 0;
        HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ablate), &v, sizeof(int), 0, hipMemcpyHostToDevice, h->stream));
    }
#endif
    if (h->unq == OWN_CUR) {
        // first step from an un-quantised state: one thread per cell straight from the upload format, in
        // float64 (exact and f64 modes: bit-identical to the reference's first step) or float32 (fast mode)
#define DW_GEN(T, PR, IL, ID)                                                                                     \
    hipLaunchKernelGGL((step_generic<T, PR>), ggrid, dim3(256), 0, h->stream, IL, ID, h->L16[out], h->D16[out],   \
                       p.height, p.width, P, P64, stats, fixups, zero_me, zero_n, gcpt)
        // exact mode: float32 with the tie bound for non-integer inputs, float64 only for the flagged cells
        // (DW_FIRST_STEP_F64=1: every cell in float64, as in round 2 - experiments)
        const bool first_f64 = h->sw.first_f64;
        const bool f32arith = prec == DW_PRECISION_FAST;
        const bool bounded = prec == DW_PRECISION_EXACT && !first_f64;
#define DW_GEN3(T, IL, ID, FB)                                                                                    \
    hipLaunchKernelGGL((step_generic<T, 3>), ggrid, dim3(256), 0, h->stream, IL, ID, h->L16[out], h->D16[out],    \
                       p.height, p.width, P, P64, stats, fixups, zero_me, zero_n, gcpt, FB)
        // every shape the steady-state wave-strip kernels take (select_kernel: W >= 256 a multiple of 4, or the packed
        // mode of narrower worlds; and any multiple of 256): the wave-strip form of the same arithmetic (dw_step_first.hpp;
        // ~4x fewer vector instructions per cell).  DW_FIRST_GENERIC=1: the one-thread-per-cell kernel (experiments, tests)
        const bool first_stream = (p.width % 256 == 0 || h->use_stream) && (f32arith || bounded) && !h->sw.first_generic;
        if (first_stream) {
            const bool packed = h->use_stream && p.width < 256;
            FirstGeom fg;
            fg.B = p.batch; fg.H = p.height; fg.W = p.width;
            fg.lpw = packed ? h->sgeom.lpw : 64;
            fg.wpr = packed ? h->sgeom.wpr : 1;
            fg.ncs = packed ? 1 : (p.width + 255) / 256;
            const long groups = packed ? (p.batch + fg.wpr - 1) / fg.wpr : p.batch;
            int sr = 64;                                         // shorter strips until every SIMD has four
            while (sr > 8 && groups * fg.ncs * ((p.height + sr - 1) / sr) < 4096) sr >>= 1;
            fg.SR = p.height < sr ? p.height : sr;
            fg.nrs = (p.height + fg.SR - 1) / fg.SR;
            fg.nstrips = (int)(groups * fg.nrs * fg.ncs);
            const int fhalo = packed ? 3 : (p.width == 256 ? 0 : (p.width % 256 == 0 ? 1 : 2));
            const dim3 fgrid((unsigned)((fg.nstrips + 3) / 4));
            const FirstStepBound fb = bounded ? derive_first_bound(p, L, P, h->unq_kind == UNQ_F64, h->sw.first_slack) : FirstStepBound{};
#define DW_FIRST(T, PR, HL, IL, ID)                                                                               \
    hipLaunchKernelGGL((step_first_stream<T, PR, HL>), fgrid, dim3(256), 0, h->stream, IL, ID, h->L16[out],       \
                       h->D16[out], fg, P, P64, stats, fixups, zero_me, zero_n, fb)
#define DW_FIRST_HL(T, PR, IL, ID)                                                                                \
    do { if (fhalo == 0) DW_FIRST(T, PR, 0, IL, ID); else if (fhalo == 1) DW_FIRST(T, PR, 1, IL, ID);           \
         else if (fhalo == 2) DW_FIRST(T, PR, 2, IL, ID); else DW_FIRST(T, PR, 3, IL, ID); } while (0)
            if (h->unq_kind == UNQ_F64) {
                if (f32arith) DW_FIRST_HL(double, 1, h->L64, h->D64);
                else DW_FIRST_HL(double, 3, h->L64, h->D64);
            } else {
                if (f32arith) DW_FIRST_HL(float, 1, h->U32L, h->U32D);
                else DW_FIRST_HL(float, 3, h->U32L, h->U32D);
            }
#undef DW_FIRST_HL
#undef DW_FIRST
        } else if (h->unq_kind == UNQ_F64) {
            if (f32arith) DW_GEN(double, 1, h->L64, h->D64);
            else if (bounded) DW_GEN3(double, h->L64, h->D64, derive_first_bound(p, L, P, true, h->sw.first_slack));
            else DW_GEN(double, 2, h->L64, h->D64);
        } else {
            if (f32arith) DW_GEN(float, 1, h->U32L, h->U32D);
            else if (bounded) DW_GEN3(float, h->U32L, h->U32D, derive_first_bound(p, L, P, false, h->sw.first_slack));
            else DW_GEN(float, 2, h->U32L, h->U32D);
        }
#undef DW_GEN3
        HIPCHK(hipGetLastError());
    } else if (prec == DW_PRECISION_F64 || (h->tcq == 0 && !h->use_stream)) {
        if (prec == DW_PRECISION_F64) DW_GEN(plane_t, 2, h->L16[in], h->D16[in]);
        else if (prec == DW_PRECISION_EXACT) DW_GEN(plane_t, 0, h->L16[in], h->D16[in]);
        else DW_GEN(plane_t, 1, h->L16[in], h->D16[in]);
#undef DW_GEN
        HIPCHK(hipGetLastError());
    } else if (h->use_stream) {
        const bool ex = prec == DW_PRECISION_EXACT;
        const StripGeom& g = h->sgeom;
        const dim3 grid((unsigned)g.chunk * 8u);
        const int halo = p.width < 256 ? 3 : (p.width == 256 ? 0 : (p.width % 256 == 0 ? 1 : 2));
#define DW_STREAM(K, HL)                                                                                \
    hipLaunchKernelGGL((K<HL>), grid, dim3(256), 0, h->stream, h->L16[in], h->D16[in], h->L16[out],     \
                       h->D16[out], g, P, P64, stats, fixups, zero_me, zero_n)
        if (ex) {
            const StreamExactArgs A{h->L16[in], h->D16[in], h->L16[out], h->D16[out], g, P, stats, fixups, zero_me,
                                    zero_n, P64};
#define DW_SX(HL)                                                                                     \
    do {                                                                                              \
        if (h->sym_albedo) hipLaunchKernelGGL((step_stream_exact<HL, true>), grid, dim3(256), 0, h->stream, A); \
        else hipLaunchKernelGGL((step_stream_exact<HL, false>), grid, dim3(256), 0, h->stream, A);     \
    } while (0)
            if (halo == 0) DW_SX(0); else if (halo == 1) DW_SX(1); else if (halo == 2) DW_SX(2); else DW_SX(3);
#undef DW_SX

        } else {
            if (halo == 0) DW_STREAM(step_stream_fast, 0);
            else if (halo == 1) DW_STREAM(step_stream_fast, 1);
            else if (halo == 2) DW_STREAM(step_stream_fast, 2);
            else DW_STREAM(step_stream_fast, 3);
        }
#undef DW_STREAM
        HIPCHK(hipGetLastError());
    } else {
        const bool ex = prec == DW_PRECISION_EXACT;
        int rc;
#define DW_TILED(T, R)                                                                              \
    rc = ex ? launch_tiled<T, R, true>(h, h->L16[in], h->D16[in], h->L16[out], h->D16[out], P, P64,  \
                                       stats, fixups, zero_me, zero_n, fq)                           \
            : launch_tiled<T, R, false>(h, h->L16[in], h->D16[in], h->L16[out], h->D16[out], P, P64, \
                                        stats, fixups, zero_me, zero_n, fq)
        if (h->tcq == 64 && h->rpt == 8) { DW_TILED(64, 8); }
        else if (h->tcq == 64 && h->rpt == 4) { DW_TILED(64, 4); }
        else if (h->tcq == 64) { DW_TILED(64, 2); }
        else if (h->tcq == 32) { DW_TILED(32, 4); }
        else { DW_TILED(16, 2); }
#undef DW_TILED
        if (rc != DW_OK) return rc;
    }
    h->cur = out;
    h->sp = 1 - h->sp;
    h->unq = (h->unq == OWN_CUR) ? OWN_PREV : OWN_NONE;
    h->stepped = true;
    h->L_last = L;
    release_unquantised(h);
    return DW_OK;
}

// Two steps (luminosities L1 then L2) in one launch on wide grids, no agent update in between.  The buffer
// that held the input now holds the state TWO steps back, so the retained "previous state" is not valid
// afterwards; dw_step_n always ends with an ordinary single step.
static int launch_forward_fused2(dw_handle* h, double L1, double L2, unsigned int* pstats = nullptr,
                                 float thr_hi = 0.f) {
    const dw_params& p = h->prm;
    const int in = h->cur, out = 1 - h->cur;
    PhysF32 P1, P2;
    if (p.precision == DW_PRECISION_EXACT) derive_f32_pair(p, L1, L2, &P1, &P2);
    else { P1 = derive_f32(p, L1); P2 = derive_f32(p, L2); }
    unsigned long long* zero_me = reinterpret_cast<unsigned long long*>(h->stats2[h->sp]);
    const int zero_n = (int)(h->stats_bytes / sizeof(unsigned long long));
    const FusedGeom& g = h->fgeom;
    const dim3 grid((unsigned)g.chunk * 8u);
    const bool rot = p.width == 256, pack = p.width < 256, ring = h->fused_ring;
    if (p.precision == DW_PRECISION_EXACT) {
        const FusedExactArgs A{h->L16[in], h->D16[in], h->L16[out], h->D16[out], g, P1, lum_part(P2), zero_me, zero_n,
                               pstats, thr_hi, make_f64(p, L1), L1, L2};
#define DW_FX(R, P, S)                                                                                            \
    do {                                                                                                          \
        if (h->sym_albedo) hipLaunchKernelGGL((step_stream_fused2_exact<R, P, S, true>), grid, dim3(256), 0, h->stream, A); \
        else hipLaunchKernelGGL((step_stream_fused2_exact<R, P, S, false>), grid, dim3(256), 0, h->stream, A);     \
    } while (0)
        if (pstats) {
            if (pack) DW_FX(kFusedRot, true, true); else if (rot) DW_FX(kFusedRot, false, true);
            else if (ring) DW_FX(kFusedRing, false, true); else DW_FX(kFusedOvl, false, true);
        } else {
            if (pack) DW_FX(kFusedRot, true, false); else if (rot) DW_FX(kFusedRot, false, false);
            else if (ring) DW_FX(kFusedRing, false, false); else DW_FX(kFusedOvl, false, false);
        }
#undef DW_FX
    } else {
#define DW_FF(R, P, S)                                                                                           \
    hipLaunchKernelGGL((step_stream_fused2<R, P, S>), grid, dim3(256), 0, h->stream, h->L16[in], h->D16[in],      \
                       h->L16[out], h->D16[out], g, P1, P2, zero_me, zero_n, pstats, thr_hi)
        if (pstats) {
            if (pack) DW_FF(kFusedRot, true, true); else if (rot) DW_FF(kFusedRot, false, true);
            else if (ring) DW_FF(kFusedRing, false, true); else DW_FF(kFusedOvl, false, true);
        } else {
            if (pack) DW_FF(kFusedRot, true, false); else if (rot) DW_FF(kFusedRot, false, false);
            else if (ring) DW_FF(kFusedRing, false, false); else DW_FF(kFusedOvl, false, false);
        }
#undef DW_FF
    }
    HIPCHK(hipGetLastError());
    h->cur = out;
    h->sp = 1 - h->sp;            // the kernel cleared the old buffer; the (untouched, zero) other one is "current"
    h->unq = OWN_NONE;
    h->stepped = false;
    h->L_last = L2;
    release_unquantised(h);
    return DW_OK;
}

static int launch_agents(dw_handle* h, const int* d_action, int act_b, int act_n, bool standalone = false,
                         double* d_reward = nullptr, unsigned char* d_done = nullptr, unsigned char* d_ok = nullptr) {
    const dw_params& p = h->prm;
    if (p.n_agents == 0) return DW_OK;
    NEED(h->have_state, DW_ESTATE, "no state uploaded");
    NEED(h->have_agents, DW_ESTATE, "no agents uploaded (call dw_upload_agents or dw_init_random)");
    NEED(p.collision_mode == 0 || standalone, DW_EINVAL,
         "collision_mode=1: call dw_update_agents, apply the collision pass (it consumes the caller's RNG) to the "
         "downloaded agent states, upload them, then dw_step without actions");
    const int blocks = (p.batch + 63) / 64;
#define DW_AG(T, PL, PD)                                                                                          \
    hipLaunchKernelGGL(agents_update<T>, dim3(blocks), dim3(64), 0, h->stream, PL, PD, h->idx, h->st, d_action,    \
                       act_b, act_n, p.batch, p.n_agents, p.height, p.width, p.agent_gamma,                        \
                       p.collision_mode == 0 ? 1 : 0, d_reward, d_done, d_ok)
    if (h->unq == OWN_CUR) {                       // grazing on the un-quantised state, in its own format
        if (h->unq_kind == UNQ_F64) DW_AG(double, h->L64, h->D64); else DW_AG(float, h->U32L, h->U32D);
    } else {
        DW_AG(plane_t, h->L16[h->cur], h->D16[h->cur]);
    }
#undef DW_AG
    HIPCHK(hipGetLastError());
    return DW_OK;
}

static int stage_host_actions(dw_handle* h, const int32_t* action, int b, int n) {
    const dw_params& p = h->prm;
    NEED(b >= 0 && n >= 0 && b <= p.batch && n <= p.n_agents, DW_EINVAL,
         "action block %dx%d exceeds (B,N)=(%d,%d)", b, n, p.batch, p.n_agents);
    if ((size_t)b * n)
        HIPCHK(hipMemcpyAsync(h->action_tmp, action, sizeof(int) * (size_t)b * n, hipMemcpyHostToDevice,
                              h->stream));
    return DW_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* dw_last_error(void) { return g_err; }
int dw_abi_version(void) { return DW_ABI_VERSION; }

int dw_pinned_alloc(size_t bytes, void** out) {
    NEED(out && bytes > 0, DW_EINVAL, "null argument");
    int ndev = 0;
    NEED(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, DW_ENODEVICE, "no HIP device");
    HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return DW_OK;
}
int dw_pinned_free(void* p) {
    if (p) HIPCHK(hipHostFree(p));
    return DW_OK;
}

#ifndef DW_BUILD_ID
#define DW_BUILD_ID "unknown"
#endif
// the marker makes the id findable in the file without loading it (therldaisyworld_amd/build.py)
static const char kBuildId[] = "DW_BUILD_ID=" DW_BUILD_ID;
const char* dw_build_id(void) { return kBuildId + 12; }

int dw_default_params(dw_params* p, int32_t batch, int32_t height, int32_t width, int32_t n_agents) {
    NEED(p, DW_EINVAL, "null params");
    std::memset(p, 0, sizeof(*p));
    p->abi_version = DW_ABI_VERSION;
    p->batch = batch; p->height = height; p->width = width; p->n_agents = n_agents;
    p->device = 0; p->precision = DW_PRECISION_EXACT; p->obs_mask = 0x0BA; p->collision_mode = 0;
    p->world_offset = 0;
    p->p = 1.0; p->g = 0.003265; p->S = 1000.0; p->sigma = 5.67e-8; p->gamma = 0.25;
    p->q = 0.2 * p->S / p->sigma; p->q2 = p->q / 8.0; p->dt = 1.0;
    p->albedo_bare = 0.5; p->albedo_light = 0.75; p->albedo_dark = 0.25; p->temp_optimal = 295.5;
    p->agent_gamma = 0.05; p->food_chain_penalty = 0.5;
    p->initial_al = 0.2; p->initial_ad = 0.2; p->light_proportion = 0.33; p->dark_proportion = 0.33;
    return DW_OK;
}

static int check_params(const dw_params* p) {
    NEED(p, DW_EINVAL, "null params");
    NEED(p->abi_version == DW_ABI_VERSION, DW_EINVAL, "ABI version %d != %d", p->abi_version, DW_ABI_VERSION);
    NEED(p->batch >= 1 && p->height >= 3 && p->width >= 3, DW_EINVAL,
         "need batch>=1 and grid >= 3x3 (got B=%d H=%d W=%d)", p->batch, p->height, p->width);
    NEED(p->height <= 65535 && p->width <= 65532, DW_EINVAL, "grid dimension too large");
    // cell indices inside one world are int32 in the host code and in every kernel
    NEED((long long)p->height * p->width <= 0x7fffffffLL, DW_EINVAL,
         "a world of %dx%d cells exceeds the 2^31-1 cells per world the kernels index", p->height, p->width);
    NEED(p->n_agents >= 0, DW_EINVAL, "n_agents < 0");
    NEED(p->precision >= 0 && p->precision <= 2, DW_EINVAL, "bad precision %d", p->precision);
    NEED(p->precision == DW_PRECISION_F64 || p->g >= 0.0, DW_EINVAL,
         "g < 0 (a growth curve opening upwards) is evaluated by DW_PRECISION_F64 only");
    NEED((double)p->batch * p->height * p->width < 9.0e18, DW_EINVAL, "too many cells");
    return DW_OK;
}

int dw_create(const dw_params* p, dw_handle** out) {
    NEED(out, DW_EINVAL, "null out");
    *out = nullptr;
    int rc = check_params(p);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DW_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    NEED(p->device >= 0 && p->device < ndev, DW_ENODEVICE, "device %d out of range (%d devices)", p->device, ndev);
    HIPCHK(hipSetDevice(p->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, p->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DW_ENODEVICE, "device %d is %s; this library is built for gfx950 only", p->device,
                    prop.gcnArchName);
    dw_handle* h = new (std::nothrow) dw_handle();
    NEED(h, DW_ENOMEM, "host allocation failed");
    h->prm = *p;
    h->cells = (size_t)p->batch * p->height * p->width;
    auto cleanup = [&](int code) { dw_destroy(h); return code; };
#define TRY(expr)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? DW_ENOMEM : DW_EHIP, "%s failed: %s", #expr, \
                                hipGetErrorString(e_)));                                           \
    } while (0)
    TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    for (int i = 0; i < 2; ++i) {
        TRY(hipMalloc(&h->L16[i], sizeof(plane_t) * h->cells));
        TRY(hipMalloc(&h->D16[i], sizeof(plane_t) * h->cells));
    }
    const size_t bn = (size_t)p->batch * (p->n_agents > 0 ? p->n_agents : 1);
    TRY(hipMalloc(&h->idx, sizeof(int) * bn * 2));
    TRY(hipMalloc(&h->st, sizeof(double) * bn));
    TRY(hipMalloc(&h->action, sizeof(int) * bn));
    TRY(hipMalloc(&h->action_tmp, sizeof(int) * bn));
    TRY(hipMalloc(&h->reward_d, sizeof(double) * bn));
    TRY(hipMalloc(&h->done_d, bn));
    TRY(hipMalloc(&h->agents_done_at, sizeof(int) * bn));
    TRY(hipMalloc(&h->done_at, sizeof(int) * p->batch));
    TRY(hipMalloc(&h->n_alive, sizeof(int)));
    // [(B+1) StatsDev][(kNumQueues+1)*16 uint queue counters]
    h->stats_bytes = sizeof(StatsDev) * (p->batch + 1) + sizeof(unsigned int) * (kNumQueues + 1) * 16;
    for (int i = 0; i < 2; ++i) {
        TRY(hipMalloc(&h->stats2[i], h->stats_bytes));
        TRY(hipMemsetAsync(h->stats2[i], 0, h->stats_bytes, h->stream));
    }
    h->sw = read_switches();                                    // the environment is read here and nowhere else
    select_kernel(h);
    if (int qrc = ensure_fixq(h)) return cleanup(qrc);
    TRY(hipMemsetAsync(h->action, 0, sizeof(int) * bn, h->stream));
    TRY(hipMemsetAsync(h->done_at, 0, sizeof(int) * p->batch, h->stream));
    TRY(hipMemsetAsync(h->agents_done_at, 0, sizeof(int) * bn, h->stream));
    TRY(hipMemsetAsync(h->n_alive, 0, sizeof(int), h->stream));
    TRY(hipMalloc(&h->side_stats, sizeof(StatsDev) * (p->batch + 1)));
    TRY(hipEventCreate(&h->ev0));
    TRY(hipEventCreate(&h->ev1));
    TRY(hipEventCreate(&h->evf0));
    TRY(hipEventCreate(&h->evf1));
#undef TRY
    select_kernel(h);
    *out = h;
    return DW_OK;
}

int dw_destroy(dw_handle* h) {
    if (!h) return DW_OK;
    (void)hipSetDevice(h->prm.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 2; ++i) { (void)hipFree(h->L16[i]); (void)hipFree(h->D16[i]); }
    (void)hipFree(h->L64); (void)hipFree(h->D64);
    (void)hipFree(h->U32L); (void)hipFree(h->U32D); (void)hipFree(h->side_stats);
    (void)hipFree(h->idx); (void)hipFree(h->st); (void)hipFree(h->action); (void)hipFree(h->action_tmp);
    (void)hipFree(h->reward_d); (void)hipFree(h->done_d);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->ep_pinned) (void)hipHostFree(h->ep_pinned);
    (void)hipFree(h->mlp_w);
    for (auto& sn : h->snap) {
        (void)hipFree(sn.L); (void)hipFree(sn.D); (void)hipFree(sn.idx); (void)hipFree(sn.st);
        (void)hipFree(sn.PL); (void)hipFree(sn.PD);
        (void)hipFree(sn.stats);
    }
    (void)hipFree(h->agents_done_at); (void)hipFree(h->done_at); (void)hipFree(h->n_alive);
    (void)hipFree(h->stats2[0]); (void)hipFree(h->stats2[1]); (void)hipFree(h->scratch); (void)hipFree(h->ep_buf); (void)hipFree(h->fixq); (void)hipFree(h->redo_tiles);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->evf0) (void)hipEventDestroy(h->evf0);
    if (h->evf1) (void)hipEventDestroy(h->evf1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return DW_OK;
}

int dw_set_params(dw_handle* h, const dw_params* p) {
    NEED(h, DW_EINVAL, "null handle");
    int rc = check_params(p);
    if (rc) return rc;
    const dw_params& o = h->prm;
    NEED(p->batch == o.batch && p->height == o.height && p->width == o.width && p->n_agents == o.n_agents &&
             p->device == o.device,
         DW_EINVAL, "dw_set_params cannot change shape or device; create a new handle");
    h->prm = *p;
    select_kernel(h);
    return ensure_fixq(h);
}

int dw_get_params(const dw_handle* h, dw_params* out) {
    NEED(h && out, DW_EINVAL, "null argument");
    *out = h->prm;
    return DW_OK;
}

// ---- state in / out ---------------------------------------------------------------------------

// reductions of the current state, whatever its format (after uploads / init, so that dw_reduce is always valid)
static int clear_stats(dw_handle* h) {
    for (int i = 0; i < 2; ++i) HIPCHK(hipMemsetAsync(h->stats2[i], 0, h->stats_bytes, h->stream));
    return DW_OK;
}
static int refresh_stats(dw_handle* h) {
    const dw_params& p = h->prm;
    if (int rc = clear_stats(h)) return rc;
    const int n = p.height * p.width;
    const dim3 g((unsigned)((n + kInitChunk - 1) / kInitChunk), (unsigned)p.batch);
    if (h->unq != OWN_CUR)
        hipLaunchKernelGGL((stats_only<plane_t>), g, dim3(256), 0, h->stream, h->L16[h->cur], h->D16[h->cur], n, h->stats2[h->sp]);
    else if (h->unq_kind == UNQ_F64)
        hipLaunchKernelGGL((stats_only<double>), g, dim3(256), 0, h->stream, h->L64, h->D64, n, h->stats2[h->sp]);
    else
        hipLaunchKernelGGL((stats_only<float>), g, dim3(256), 0, h->stream, h->U32L, h->U32D, n, h->stats2[h->sp]);
    HIPCHK(hipGetLastError());
    return DW_OK;
}

int dw_upload_state_f64(dw_handle* h, const double* light, const double* dark) {
    NEED(h && light && dark, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    if (int arc = ensure_f64(h)) return arc;
    HIPCHK(hipMemcpyAsync(h->L64, light, sizeof(double) * h->cells, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->D64, dark, sizeof(double) * h->cells, hipMemcpyHostToDevice, h->stream));
    h->unq_kind = UNQ_F64;
    h->unq = OWN_CUR;
    h->have_state = true;
    h->stepped = false;
    for (auto& sn : h->snap) sn.valid = false;                      // a snapshot's previous state may have lived in these buffers
    int rc = refresh_stats(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));   // host buffers may be reused by the caller
    return DW_OK;
}

int dw_upload_state_f32(dw_handle* h, const float* light, const float* dark, int quantised) {
    NEED(h && light && dark, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    const unsigned blocks = (unsigned)((h->cells + 255) / 256);
    if (quantised) {
        // natural-unit floats staged in scratch, rounded to the per-mille integers of the canonical planes
        int rc = ensure_scratch(h, sizeof(float) * 2 * h->cells);
        if (rc) return rc;
        float* sL = reinterpret_cast<float*>(h->scratch);
        float* sD = sL + h->cells;
        HIPCHK(hipMemcpyAsync(sL, light, sizeof(float) * h->cells, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(sD, dark, sizeof(float) * h->cells, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(f32nat_to_plane, dim3(blocks), dim3(256), 0, h->stream, sL, h->L16[h->cur], h->cells);
        hipLaunchKernelGGL(f32nat_to_plane, dim3(blocks), dim3(256), 0, h->stream, sD, h->D16[h->cur], h->cells);
        HIPCHK(hipGetLastError());
        h->unq = OWN_NONE;
    } else {
        int rc = ensure_u32(h);
        if (rc) return rc;
        HIPCHK(hipMemcpyAsync(h->U32L, light, sizeof(float) * h->cells, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->U32D, dark, sizeof(float) * h->cells, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(f32nat_to_permille, dim3(blocks), dim3(256), 0, h->stream, h->U32L, h->U32L, h->cells);
        hipLaunchKernelGGL(f32nat_to_permille, dim3(blocks), dim3(256), 0, h->stream, h->U32D, h->U32D, h->cells);
        HIPCHK(hipGetLastError());
        h->unq_kind = UNQ_F32;
        h->unq = OWN_CUR;
    }
    h->have_state = true;
    h->stepped = false;
    for (auto& sn : h->snap) sn.valid = false;
    int rc = refresh_stats(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_upload_agents(dw_handle* h, const int32_t* indices, const double* states) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t bn = (size_t)p.batch * p.n_agents;
    if (bn) {
        NEED(indices && states, DW_EINVAL, "null argument");
        for (size_t i = 0; i < bn; ++i) {
            NEED(indices[2 * i] >= 0 && indices[2 * i] < p.height && indices[2 * i + 1] >= 0 &&
                     indices[2 * i + 1] < p.width,
                 DW_EINVAL, "agent %zu position (%d,%d) outside the %dx%d grid", i, indices[2 * i],
                 indices[2 * i + 1], p.height, p.width);
        }
        HIPCHK(hipMemcpyAsync(h->idx, indices, sizeof(int) * bn * 2, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->st, states, sizeof(double) * bn, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->have_agents = true;
    return DW_OK;
}

int dw_download_agents(dw_handle* h, int32_t* indices, double* states) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t bn = (size_t)p.batch * p.n_agents;
    if (bn) {
        NEED(h->have_agents, DW_ESTATE, "no agents");
        if (indices) HIPCHK(hipMemcpyAsync(indices, h->idx, sizeof(int) * bn * 2, hipMemcpyDeviceToHost, h->stream));
        if (states) HIPCHK(hipMemcpyAsync(states, h->st, sizeof(double) * bn, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_init_random(dw_handle* h, uint64_t seed) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    int rc = ensure_u32(h);                     // the synthetic initial state is un-quantised like the reference's
    if (rc) return rc;
    // the draw reduces its own values into the per-world statistics (no second pass over the planes)
    if (int crc = clear_stats(h)) return crc;
    const int ncell = p.height * p.width;
    const dim3 g((unsigned)((ncell + kInitChunk - 1) / kInitChunk), (unsigned)p.batch);
    hipLaunchKernelGGL((init_random_cells<float>), g, dim3(256), 0, h->stream, h->U32L, h->U32D, ncell,
                       (long long)p.world_offset, (unsigned long long)seed, (float)p.light_proportion,
                       (float)p.dark_proportion, (float)p.initial_al, (float)p.initial_ad, h->stats2[h->sp]);
    HIPCHK(hipGetLastError());
    if (p.n_agents) {
        const int bn = p.batch * p.n_agents;
        hipLaunchKernelGGL(init_random_agents, dim3((bn + 255) / 256), dim3(256), 0, h->stream, h->idx, h->st,
                           p.batch, p.n_agents, p.height, p.width, (long long)p.world_offset,
                           (unsigned long long)seed);
        HIPCHK(hipGetLastError());
    }
    h->have_agents = true;
    h->unq_kind = UNQ_F32;
    h->unq = OWN_CUR;
    h->have_state = true;
    h->stepped = false;
    for (auto& sn : h->snap) sn.valid = false;
    return DW_OK;                               // (statistics: reduced by the draw itself)
}

int dw_init_random_quantised(dw_handle* h, uint64_t seed) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    if (int crc = clear_stats(h)) return crc;
    const int ncell = p.height * p.width;
    const dim3 g((unsigned)((ncell + kInitChunk - 1) / kInitChunk), (unsigned)p.batch);
    hipLaunchKernelGGL((init_random_cells<plane_t>), g, dim3(256), 0, h->stream, h->L16[h->cur], h->D16[h->cur], ncell,
                       (long long)p.world_offset, (unsigned long long)seed, (float)p.light_proportion,
                       (float)p.dark_proportion, (float)p.initial_al, (float)p.initial_ad, h->stats2[h->sp]);
    HIPCHK(hipGetLastError());
    if (p.n_agents) {
        const int bn = p.batch * p.n_agents;
        hipLaunchKernelGGL(init_random_agents, dim3((bn + 255) / 256), dim3(256), 0, h->stream, h->idx, h->st,
                           p.batch, p.n_agents, p.height, p.width, (long long)p.world_offset,
                           (unsigned long long)seed);
        HIPCHK(hipGetLastError());
    }
    h->have_agents = true;
    h->unq = OWN_NONE;
    h->have_state = true;
    h->stepped = false;
    for (auto& sn : h->snap) sn.valid = false;
    release_unquantised(h);
    return DW_OK;                               // (statistics: reduced by the draw itself)
}

int dw_download_planes(dw_handle* h, int which, double* light, double* dark) {
    NEED(h, DW_EINVAL, "null handle");
    NEED(h->have_state, DW_ESTATE, "no state");
    HIPCHK(hipSetDevice(h->prm.device));
    NEED(which == DW_STATE_CURRENT || which == DW_STATE_PREVIOUS, DW_EINVAL, "bad state selector");
    NEED(which == DW_STATE_CURRENT || h->stepped, DW_ESTATE, "no previous state before the first step");
    const int buf = which == DW_STATE_CURRENT ? h->cur : 1 - h->cur;
    const bool unq = (which == DW_STATE_CURRENT && h->unq == OWN_CUR) || (which == DW_STATE_PREVIOUS && h->unq == OWN_PREV);
    const size_t bytes = sizeof(double) * h->cells;
    if (unq && h->unq_kind == UNQ_F64) {
        if (light) HIPCHK(hipMemcpyAsync(light, h->L64, bytes, hipMemcpyDeviceToHost, h->stream));
        if (dark) HIPCHK(hipMemcpyAsync(dark, h->D64, bytes, hipMemcpyDeviceToHost, h->stream));
    } else {
        int rc = ensure_scratch(h, bytes);
        if (rc) return rc;
        const unsigned blocks = (unsigned)((h->cells + 255) / 256);
        for (int pl = 0; pl < 2; ++pl) {
            double* dst = pl == 0 ? light : dark;
            if (!dst) continue;
            if (unq)
                hipLaunchKernelGGL((plane_to_f64<float>), dim3(blocks), dim3(256), 0, h->stream,
                                   pl == 0 ? h->U32L : h->U32D, h->scratch, h->cells);
            else
                hipLaunchKernelGGL((plane_to_f64<plane_t>), dim3(blocks), dim3(256), 0, h->stream,
                                   pl == 0 ? h->L16[buf] : h->D16[buf], h->scratch, h->cells);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(dst, h->scratch, bytes, hipMemcpyDeviceToHost, h->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

// The state observations and the materialised grid derive their temperature channels from: after a step the
// PRE-step state (post = true), before any step the current one (post = false); fmt 0 binary16, 1 float32
// per-mille, 2 float64 natural.
struct DerivedFrom { int fmt; const void* L; const void* D; bool post; };
static DerivedFrom derived_from(const dw_handle* h) {
    const int cur = h->cur, prev = 1 - h->cur;
    if (h->stepped) {
        if (h->unq == OWN_PREV)
            return h->unq_kind == UNQ_F64 ? DerivedFrom{2, h->L64, h->D64, true} : DerivedFrom{1, h->U32L, h->U32D, true};
        return DerivedFrom{0, h->L16[prev], h->D16[prev], true};
    }
    if (h->unq == OWN_CUR)
        return h->unq_kind == UNQ_F64 ? DerivedFrom{2, h->L64, h->D64, false} : DerivedFrom{1, h->U32L, h->U32D, false};
    return DerivedFrom{0, h->L16[cur], h->D16[cur], false};
}
// K<T, POST>(pL, pD, args...) for the format / phase of `src`
#define DW_DISPATCH_DERIVED(K, src, grid, block, ...)                                                               \
    do {                                                                                                            \
        if ((src).fmt == 2) {                                                                                       \
            if ((src).post) hipLaunchKernelGGL((K<double, true>), grid, block, 0, h->stream, (const double*)(src).L, (const double*)(src).D, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<double, false>), grid, block, 0, h->stream, (const double*)(src).L, (const double*)(src).D, __VA_ARGS__); \
        } else if ((src).fmt == 1) {                                                                                \
            if ((src).post) hipLaunchKernelGGL((K<float, true>), grid, block, 0, h->stream, (const float*)(src).L, (const float*)(src).D, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<float, false>), grid, block, 0, h->stream, (const float*)(src).L, (const float*)(src).D, __VA_ARGS__); \
        } else {                                                                                                    \
            if ((src).post) hipLaunchKernelGGL((K<plane_t, true>), grid, block, 0, h->stream, (const plane_t*)(src).L, (const plane_t*)(src).D, __VA_ARGS__); \
            else hipLaunchKernelGGL((K<plane_t, false>), grid, block, 0, h->stream, (const plane_t*)(src).L, (const plane_t*)(src).D, __VA_ARGS__); \
        }                                                                                                           \
    } while (0)

// materialise into device scratch: grid7 and/or caches
static int run_materialise(dw_handle* h, double L, double* d_grid7, double* d_temps, double* d_betas,
                           double* d_growth, double* d_teff) {
    const dw_params& p = h->prm;
    const dim3 g((unsigned)((p.height * p.width + 255) / 256), (unsigned)p.batch);
    const DerivedFrom src = derived_from(h);
    const PhysF64 P = make_f64(p, src.post ? h->L_last : L);
    const plane_t* cL = h->L16[h->cur];            // read by the POST variants only
    const plane_t* cD = h->D16[h->cur];
    DW_DISPATCH_DERIVED(materialise, src, g, dim3(256), cL, cD, p.height, p.width, P, d_grid7, d_temps, d_betas, d_growth,
                        d_teff);
    HIPCHK(hipGetLastError());
    if (src.post && d_grid7 && p.n_agents && h->have_agents) {
        hipLaunchKernelGGL(agents_stamp, dim3((p.batch + 63) / 64), dim3(64), 0, h->stream, d_grid7, h->idx,
                           h->st, p.batch, p.n_agents, p.height, p.width);
        HIPCHK(hipGetLastError());
    }
    return DW_OK;
}

int dw_download_grid(dw_handle* h, double L_init, double* grid7) {
    NEED(h && grid7, DW_EINVAL, "null argument");
    NEED(h->have_state, DW_ESTATE, "no state");
    HIPCHK(hipSetDevice(h->prm.device));
    const size_t bytes = sizeof(double) * 7 * h->cells;
    int rc = ensure_scratch(h, bytes);
    if (rc) return rc;
    rc = run_materialise(h, L_init, h->scratch, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(grid7, h->scratch, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_download_caches(dw_handle* h, double L, double* temps, double* betas, double* growth,
                       double* temp_effective) {
    NEED(h, DW_EINVAL, "null handle");
    NEED(h->have_state, DW_ESTATE, "no state");
    HIPCHK(hipSetDevice(h->prm.device));
    const size_t n = h->cells;
    int rc = ensure_scratch(h, sizeof(double) * 9 * n);
    if (rc) return rc;
    double* d_t = h->scratch;
    double* d_b = d_t + 3 * n;
    double* d_g = d_b + 3 * n;
    double* d_e = d_g + 2 * n;
    rc = run_materialise(h, L, nullptr, temps ? d_t : nullptr, betas ? d_b : nullptr, growth ? d_g : nullptr,
                         temp_effective ? d_e : nullptr);
    if (rc) return rc;
    if (temps) HIPCHK(hipMemcpyAsync(temps, d_t, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream));
    if (betas) HIPCHK(hipMemcpyAsync(betas, d_b, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream));
    if (growth) HIPCHK(hipMemcpyAsync(growth, d_g, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    if (temp_effective) HIPCHK(hipMemcpyAsync(temp_effective, d_e, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

// ---- the hot path -----------------------------------------------------------------------------

int dw_update_agents(dw_handle* h, const int32_t* action, int32_t action_b, int32_t action_n) {
    NEED(h && action, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    int rc = stage_host_actions(h, action, action_b, action_n);
    if (rc) return rc;
    return launch_agents(h, h->action_tmp, action_b, action_n, true);
}

int dw_step(dw_handle* h, const int32_t* action, int32_t action_b, int32_t action_n, double L) {
    NEED(h, DW_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->prm.device));
    if (action) {
        int rc = stage_host_actions(h, action, action_b, action_n);
        if (rc) return rc;
        rc = launch_agents(h, h->action_tmp, action_b, action_n);
        if (rc) return rc;
    }
    return launch_forward(h, L);
}

int dw_step_device_actions(dw_handle* h, double L) {
    NEED(h, DW_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->prm.device));
    int rc = launch_agents(h, h->action, h->prm.batch, h->prm.n_agents);
    if (rc) return rc;
    return launch_forward(h, L);
}

int dw_upload_actions(dw_handle* h, const int32_t* action) {
    NEED(h && action, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    const size_t bn = (size_t)h->prm.batch * h->prm.n_agents;
    if (bn) {
        HIPCHK(hipMemcpyAsync(h->action, action, sizeof(int) * bn, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return DW_OK;
}

int dw_download_actions(dw_handle* h, int32_t* action) {
    NEED(h && action, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    const size_t bn = (size_t)h->prm.batch * h->prm.n_agents;
    if (bn) HIPCHK(hipMemcpyAsync(action, h->action, sizeof(int) * bn, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_step_n(dw_handle* h, int32_t nsteps, double* L_io, double dL, double min_L, double max_L,
              int use_device_actions) {
    NEED(h && L_io, DW_EINVAL, "null argument");
    NEED(nsteps >= 0, DW_EINVAL, "nsteps < 0");
    HIPCHK(hipSetDevice(h->prm.device));
    double L = *L_io;
    auto advance = [&]() { L += dL; L = L > max_L ? max_L : L; L = L < min_L ? min_L : L; };   // ref update_L :471-473
    int s0 = 0;
    h->fused_launches = 0;
    if (!use_device_actions && nsteps >= 1 && h->have_state && !cur_quantised(h)) {
        int rc = launch_forward(h, L);          // the first step reads the un-quantised state in its own format
        if (rc) return rc;
        advance();
        s0 = 1;
    }
    // Small worlds: keep the whole run of steps on the chip (worlds in LDS, one launch per 4096 steps) -
    // unless the ensemble is big enough to fill the GPU with wave-strips, where the packed fused kernel is
    // 1.5-1.8x faster (measured crossover between 2 M and 16 M cells: tools/kbench.py 512 64 / 4096 64).
    const bool big_packed = h->use_stream && h->allow_fuse && h->prm.width < 256 && h->cells >= ((size_t)1 << 23);
    if (!use_device_actions && nsteps - s0 > 1 && h->have_state && h->prm.height * h->prm.width <= 4096 && !big_packed &&
        episode_kernel_applies(h)) {
        std::vector<double> Ls;
        while (s0 < nsteps) {
            const int k = nsteps - s0 < 4096 ? nsteps - s0 : 4096;
            Ls.resize(k);
            for (int i = 0; i < k; ++i) { Ls[i] = L; advance(); }
            int rc = run_episode_impl(h, k, Ls.data(), kPolicySkipAgents, nullptr, nullptr, 5, nullptr, nullptr);
            if (rc) return rc;
            s0 += k;
        }
        *L_io = L;
        return DW_OK;
    }
    if (!use_device_actions && h->allow_fuse && h->use_stream && h->have_state &&
        (h->prm.precision == DW_PRECISION_FAST || h->prm.precision == DW_PRECISION_EXACT) && nsteps - s0 >= 3) {
        // wide grids: pairs of steps share one HBM round trip; the last one or two steps are ordinary launches
        // so that the retained previous state is the true predecessor.  HIP events around the run of fused
        // launches feed dw_last_step_n_timing (the dominant kernel's duration, measured on its own stream).
        // The first fused launch ends the life of an un-quantised PREVIOUS state (after it the retained state is two
        // steps back anyway): drop it here, so that release_unquantised's synchronise + hipFree of 128 GiB at the
        // north-star shape happen in front of the timed window and not inside it.
        if (h->unq == OWN_PREV) { h->unq = OWN_NONE; h->stepped = false; }
        release_unquantised(h);
        HIPCHK(hipEventRecord(h->evf0, h->stream));
        int launched = 0;
        while (nsteps - s0 >= 3) {
            const double L1 = L;
            advance();
            const double L2 = L;
            advance();
            int rc = launch_forward_fused2(h, L1, L2);
            if (rc) return rc;                  // fused_launches stays 0: no timing of a run that was cut short
            s0 += 2;
            launched += 1;
        }
        HIPCHK(hipEventRecord(h->evf1, h->stream));
        h->fused_launches = launched;           // valid only now that evf1 is recorded
    }
    for (int s = s0; s < nsteps; ++s) {
        int rc;
        if (use_device_actions) {
            rc = launch_agents(h, h->action, h->prm.batch, h->prm.n_agents);
            if (rc) return rc;
        }
        rc = launch_forward(h, L);
        if (rc) return rc;
        advance();
    }
    *L_io = L;
    return DW_OK;
}

int dw_last_step_n_timing(dw_handle* h, float* fused_ms, int32_t* fused_launches, int32_t* plane_elem_bytes) {
    NEED(h && fused_ms && fused_launches && plane_elem_bytes, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    *fused_ms = 0.f;
    *fused_launches = h->fused_launches;
    *plane_elem_bytes = (int32_t)sizeof(plane_t);
    if (h->fused_launches > 0) {
        HIPCHK(hipEventSynchronize(h->evf1));
        HIPCHK(hipEventElapsedTime(fused_ms, h->evf0, h->evf1));
    }
    return DW_OK;
}

int dw_forward_f64(dw_handle* h, const double* light, const double* dark, double L, double* grid7,
                   double* temps, double* betas, double* growth, double* temp_effective) {
    NEED(h && light && dark && grid7, DW_EINVAL, "null argument");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t n = h->cells;
    // scratch layout: [in light n][in dark n][grid7 7n][caches 9n] doubles, [new light n][new dark n] binary16
    int rc = ensure_scratch(h, sizeof(double) * 18 * n + sizeof(plane_t) * 2 * n + 16);
    if (rc) return rc;
    double* dL = h->scratch;
    double* dD = dL + n;
    double* dG = dD + n;
    double* d_t = dG + 7 * n;
    double* d_b = d_t + 3 * n;
    double* d_g = d_b + 3 * n;
    double* d_e = d_g + 2 * n;
    plane_t* nL = reinterpret_cast<plane_t*>(d_e + n);
    plane_t* nD = nL + n;
    // the reductions of this side computation must not disturb the handle's per-world stats
    StatsDev* tmp_stats = h->side_stats;
    HIPCHK(hipMemsetAsync(tmp_stats, 0, sizeof(StatsDev) * (p.batch + 1), h->stream));
    HIPCHK(hipMemcpyAsync(dL, light, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    {
        const hipError_t e2 = hipMemcpyAsync(dD, dark, sizeof(double) * n, hipMemcpyHostToDevice, h->stream);
        if (e2 != hipSuccess) {
            (void)hipStreamSynchronize(h->stream);      // the first copy may still be reading `light`
            return fail(DW_EHIP, "dw_forward_f64: %s", hipGetErrorString(e2));
        }
    }
    const PhysF32 P = derive_f32(p, L);
    const PhysF64 P64 = make_f64(p, L);
    const dim3 g((unsigned)((p.height * p.width + 255) / 256), (unsigned)p.batch);
    unsigned long long* tmp_fix = &tmp_stats[p.batch].sum_l;
    hipLaunchKernelGGL((step_generic<double, 2>), g, dim3(256), 0, h->stream, dL, dD, nL, nD, p.height, p.width, P,
                       P64, tmp_stats, tmp_fix, (unsigned long long*)nullptr, 0);
    hipLaunchKernelGGL((materialise<double, true>), g, dim3(256), 0, h->stream, dL, dD, nL, nD, p.height, p.width,
                       P64, dG, temps ? d_t : (double*)nullptr, betas ? d_b : (double*)nullptr,
                       growth ? d_g : (double*)nullptr, temp_effective ? d_e : (double*)nullptr);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess && p.n_agents && h->have_agents) {
        hipLaunchKernelGGL(agents_stamp, dim3((p.batch + 63) / 64), dim3(64), 0, h->stream, dG, h->idx, h->st,
                           p.batch, p.n_agents, p.height, p.width);
        le = hipGetLastError();
    }
    if (le == hipSuccess) le = hipMemcpyAsync(grid7, dG, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, h->stream);
    if (le == hipSuccess && temps) le = hipMemcpyAsync(temps, d_t, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream);
    if (le == hipSuccess && betas) le = hipMemcpyAsync(betas, d_b, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, h->stream);
    if (le == hipSuccess && growth) le = hipMemcpyAsync(growth, d_g, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, h->stream);
    if (le == hipSuccess && temp_effective)
        le = hipMemcpyAsync(temp_effective, d_e, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream);
    // the caller's host buffers are the targets of copies in flight: never return before the stream is idle
    const hipError_t se = hipStreamSynchronize(h->stream);
    if (le == hipSuccess) le = se;
    if (le != hipSuccess) return fail(DW_EHIP, "dw_forward_f64: %s", hipGetErrorString(le));
    return DW_OK;
}

int dw_conv3x3_f64(dw_handle* h, const double* plane, const double kernel[9], double* out) {
    NEED(h && plane && kernel && out, DW_EINVAL, "null argument");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t n = h->cells;
    int rc = ensure_scratch(h, sizeof(double) * 2 * n);
    if (rc) return rc;
    double* d_in = h->scratch;
    double* d_out = d_in + n;
    Kernel9 K;
    for (int i = 0; i < 9; ++i) K.k[i] = kernel[i];
    SyncOnExit guard(h->stream);                              // `plane` / `out` are the caller's
    HIPCHK(hipMemcpyAsync(d_in, plane, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    const dim3 g((unsigned)((p.height * p.width + 255) / 256), (unsigned)p.batch);
    hipLaunchKernelGGL(conv3x3_f64, g, dim3(256), 0, h->stream, d_in, d_out, p.height, p.width, K);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    guard.disarm();
    return DW_OK;
}

int dw_stage_f64(dw_handle* h, int stage, const double* in, double* out, double L, const double kernel[9]) {
    NEED(h && in && out, DW_EINVAL, "null argument");
    NEED(stage >= kStageAlbedo && stage <= kStageGrowth, DW_EINVAL, "bad stage %d", stage);
    NEED(kernel || stage > kStageDensity, DW_EINVAL, "the stencil stages need their 3x3 kernel");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t n = h->cells, nin = (size_t)stage_inputs(stage), nout = (size_t)stage_outputs(stage);
    int rc = ensure_scratch(h, sizeof(double) * (nin + nout) * n);
    if (rc) return rc;
    double* d_in = h->scratch;
    double* d_out = d_in + nin * n;
    Kernel9 K{};
    if (kernel) for (int i = 0; i < 9; ++i) K.k[i] = kernel[i];
    const PhysF64 P = make_f64(p, L);
    SyncOnExit guard(h->stream);                              // `in` / `out` are the caller's
    HIPCHK(hipMemcpyAsync(d_in, in, sizeof(double) * nin * n, hipMemcpyHostToDevice, h->stream));
    const dim3 g((unsigned)((p.height * p.width + 255) / 256), (unsigned)p.batch);
#define DW_STAGE(S) hipLaunchKernelGGL((stage_f64<S>), g, dim3(256), 0, h->stream, d_in, d_out, p.batch, p.height, p.width, P, K)
    switch (stage) {
        case kStageAlbedo: DW_STAGE(kStageAlbedo); break;
        case kStageDensity: DW_STAGE(kStageDensity); break;
        case kStageTemperature: DW_STAGE(kStageTemperature); break;
        case kStageGrowthRate: DW_STAGE(kStageGrowthRate); break;
        default: DW_STAGE(kStageGrowth); break;
    }
#undef DW_STAGE
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(double) * nout * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    guard.disarm();
    return DW_OK;
}

int dw_get_obs(dw_handle* h, double L_init, double* obs) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t bn = (size_t)p.batch * p.n_agents;
    if (bn == 0) return DW_OK;
    NEED(obs, DW_EINVAL, "null obs");
    NEED(h->have_state && h->have_agents, DW_ESTATE, "no state / agents");
    int rc = observe_into_scratch(h, L_init, 0);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(obs, h->scratch, sizeof(double) * bn * 63, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

// step + get_obs + reward/done as one call.  Everything crosses PCIe through one page-locked staging
// buffer (pageable copies are staged and serialised by the runtime, ~20 us each): actions in, the three
// results out, all asynchronous on the handle's stream with ONE synchronisation.
int dw_env_step(dw_handle* h, const int32_t* action, int32_t action_b, int32_t action_n, double L, double* obs,
                double* reward, uint8_t* done) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t bn = (size_t)p.batch * p.n_agents;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const bool big_obs = sizeof(double) * bn * 63 > ((size_t)512 << 10);
    // pinned image: [actions | obs | reward | done] (the last three contiguous, as in the device block; for big
    // observation blocks only reward | done are staged, at o_rew)
    const size_t o_act = 0, o_obs = up(sizeof(int) * bn), o_rew = o_obs + sizeof(double) * bn * 63;
    const size_t total = up(o_rew + sizeof(double) * bn + bn) + 256;
    if (h->pinned_bytes < total) {
        if (h->pinned) HIPCHK(hipHostFree(h->pinned));
        h->pinned = nullptr; h->pinned_bytes = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->pinned), total, hipHostMallocDefault));
        h->pinned_bytes = total;
    }
    if (action) {
        NEED(action_b >= 0 && action_n >= 0 && action_b <= p.batch && action_n <= p.n_agents, DW_EINVAL,
             "action block %dx%d exceeds (B,N)=(%d,%d)", action_b, action_n, p.batch, p.n_agents);
        const size_t na = (size_t)action_b * action_n;
        if (na) {
            std::memcpy(h->pinned + o_act, action, sizeof(int) * na);
            HIPCHK(hipMemcpyAsync(h->action_tmp, h->pinned + o_act, sizeof(int) * na, hipMemcpyHostToDevice, h->stream));
        }
        int rc = launch_agents(h, h->action_tmp, action_b, action_n);
        if (rc) return rc;
    }
    int rc = launch_forward(h, L);
    if (rc) return rc;
    if (bn) {
        NEED(h->have_agents, DW_ESTATE, "no agents");
        // observations, rewards and done flags land in ONE device block [obs | reward | done] and come back in
        // one copy (each extra copy costs its own ~5-10 us of latency on a 90 us step)
        const size_t d_rew = sizeof(double) * bn * 63, d_done = d_rew + sizeof(double) * bn, d_total = d_done + bn;
        rc = observe_into_scratch(h, L, d_total - d_rew + 64, true);     // reward | done written by the same kernel
        if (rc) return rc;
        unsigned char* blk = reinterpret_cast<unsigned char*>(h->scratch);
        if (big_obs) {
            if (obs) HIPCHK(hipMemcpyAsync(obs, blk, d_rew, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipMemcpyAsync(h->pinned + o_rew, blk + d_rew, d_total - d_rew, hipMemcpyDeviceToHost, h->stream));
        } else {
            HIPCHK(hipMemcpyAsync(h->pinned + o_obs, blk, d_total, hipMemcpyDeviceToHost, h->stream));
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        const unsigned char* src = big_obs ? h->pinned + o_rew - d_rew : h->pinned + o_obs;     // base of the block image
        if (obs && !big_obs) std::memcpy(obs, src, d_rew);
        if (reward) std::memcpy(reward, src + d_rew, sizeof(double) * bn);
        if (done) std::memcpy(done, src + d_done, bn);
    }
    return DW_OK;
}

int dw_get_reward_done(dw_handle* h, double* reward, uint8_t* done) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const int bn = p.batch * p.n_agents;
    if (bn == 0) return DW_OK;
    NEED(h->have_agents, DW_ESTATE, "no agents");
    hipLaunchKernelGGL(reward_done, dim3((bn + 255) / 256), dim3(256), 0, h->stream, h->st, h->reward_d, h->done_d, bn);
    HIPCHK(hipGetLastError());
    if (reward) HIPCHK(hipMemcpyAsync(reward, h->reward_d, sizeof(double) * bn, hipMemcpyDeviceToHost, h->stream));
    if (done) HIPCHK(hipMemcpyAsync(done, h->done_d, (size_t)bn, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_reduce(dw_handle* h, dw_world_stats* per_world) {
    NEED(h && per_world, DW_EINVAL, "null argument");
    NEED(h->have_state, DW_ESTATE, "no state");
    static_assert(sizeof(dw_world_stats) == sizeof(StatsDev), "stats layout");
    HIPCHK(hipSetDevice(h->prm.device));
    HIPCHK(hipMemcpyAsync(per_world, h->stats2[h->sp], sizeof(StatsDev) * h->prm.batch, hipMemcpyDeviceToHost,
                          h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

// Greedy / anti-greedy choice of every agent from the CURRENT covers (ref Greedy.__call__, agents/greedy.py:18-30)
// into h->action; agent_mode / codes as in policy_greedy (dw_agents.hpp).
static int launch_policy_greedy(dw_handle* h, int argmin, const int* agent_mode, int codes) {
    const dw_params& p = h->prm;
    const int bn = p.batch * p.n_agents;
    const dim3 g((unsigned)((bn + 255) / 256));
    if (h->unq == OWN_CUR && h->unq_kind == UNQ_F32)
        hipLaunchKernelGGL(policy_greedy<float>, g, dim3(256), 0, h->stream, h->U32L, h->U32D, h->idx, p.batch, p.n_agents,
                           p.height, p.width, p.obs_mask, argmin, agent_mode, h->action, codes);
    else
        hipLaunchKernelGGL(policy_greedy<plane_t>, g, dim3(256), 0, h->stream, h->L16[h->cur], h->D16[h->cur], h->idx,
                           p.batch, p.n_agents, p.height, p.width, p.obs_mask, argmin, agent_mode, h->action, codes);
    HIPCHK(hipGetLastError());
    return DW_OK;
}

int dw_policy_greedy(dw_handle* h, int mode) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const int bn = p.batch * p.n_agents;
    if (bn == 0) return DW_OK;
    NEED(h->have_state && h->have_agents, DW_ESTATE, "no state / agents");
    NEED(mode == DW_POLICY_ARGMAX || mode == DW_POLICY_ARGMIN, DW_EINVAL, "bad policy mode");
    NEED(cur_quantised(h) || h->unq_kind != UNQ_F64, DW_ESTATE,
         "device policy on an exact float64 initial state is not supported; compute the action on the host");
    return launch_policy_greedy(h, mode == DW_POLICY_ARGMIN ? 1 : 0, nullptr, 0);
}

int dw_policy_per_agent(dw_handle* h, const int32_t* agent_mode) {
    NEED(h && agent_mode, DW_EINVAL, "null argument");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const int bn = p.batch * p.n_agents;
    if (bn == 0) return DW_OK;
    NEED(h->have_state && h->have_agents, DW_ESTATE, "no state / agents");
    NEED(cur_quantised(h) || h->unq_kind != UNQ_F64, DW_ESTATE,
         "device policy on an exact float64 initial state is not supported; compute the action on the host");
    for (int n = 0; n < p.n_agents; ++n)
        NEED(agent_mode[n] == DW_POLICY_ARGMAX || agent_mode[n] == DW_POLICY_ARGMIN || agent_mode[n] == DW_POLICY_TABLE,
             DW_EINVAL, "agent %d: bad policy mode %d", n, agent_mode[n]);
    // modes travel in the (otherwise idle) staging buffer of host-supplied actions; 3 -> internal code 2
    std::vector<int> m(p.n_agents);
    for (int n = 0; n < p.n_agents; ++n) m[n] = agent_mode[n] == DW_POLICY_TABLE ? 2 : agent_mode[n];
    HIPCHK(hipMemcpyAsync(h->action_tmp, m.data(), sizeof(int) * p.n_agents, hipMemcpyHostToDevice, h->stream));
    const int prc = launch_policy_greedy(h, 0, h->action_tmp, 0);
    HIPCHK(hipStreamSynchronize(h->stream));      // `m` is a local host buffer: also on the error path
    return prc;
}

// fills h->scratch with the [B][N][63] observations of the current state (device side of dw_get_obs)
// reward_tail: the kernel also writes [reward (B,N) float64 | done (B,N) u8] right behind the observations
static int observe_into_scratch(dw_handle* h, double L_init, size_t extra_bytes, bool reward_tail) {
    const dw_params& p = h->prm;
    const size_t bn = (size_t)p.batch * p.n_agents;
    int rc = ensure_scratch(h, sizeof(double) * bn * 63 + extra_bytes);
    if (rc) return rc;
    double* d_rew = reward_tail ? h->scratch + bn * 63 : nullptr;
    unsigned char* d_done = reward_tail ? reinterpret_cast<unsigned char*>(h->scratch + bn * 64) : nullptr;
    const int threads = (int)(bn * 9);
    const dim3 g((threads + 127) / 128);
    const DerivedFrom src = derived_from(h);
    const PhysF64 P = make_f64(p, src.post ? h->L_last : L_init);
    const plane_t* cL = h->L16[h->cur];            // read by the POST variants only
    const plane_t* cD = h->D16[h->cur];
    DW_DISPATCH_DERIVED(observe, src, g, dim3(128), cL, cD, h->idx, h->st, p.batch, p.n_agents, p.height, p.width, P,
                        p.obs_mask, h->scratch, d_rew, d_done);
    HIPCHK(hipGetLastError());
    return DW_OK;
}

static int policy_mlp_impl(dw_handle* h, const double* params, int32_t n_members, const int32_t* world_member,
                           int32_t agent_begin, int32_t agent_end, double L_init) {
    const dw_params& p = h->prm;
    NEED(agent_begin >= 0 && agent_begin <= agent_end && agent_end <= p.n_agents, DW_EINVAL, "bad agent range");
    NEED(h->have_state && h->have_agents, DW_ESTATE, "no state / agents");
    const size_t bn = (size_t)p.batch * p.n_agents;
    if (bn == 0 || agent_begin == agent_end) return DW_OK;
    if (world_member)
        for (int b = 0; b < p.batch; ++b)
            NEED(world_member[b] >= 0 && world_member[b] < n_members, DW_EINVAL, "world %d: member %d out of range", b,
                 world_member[b]);
    const size_t wbytes = sizeof(double) * 1808 * (size_t)n_members;
    const size_t mbytes = world_member ? sizeof(int) * (size_t)p.batch : 0;
    int rc = observe_into_scratch(h, L_init, wbytes + mbytes + 16);
    if (rc) return rc;
    double* d_w = h->scratch + bn * 63;
    int* d_m = world_member ? reinterpret_cast<int*>(d_w + 1808 * (size_t)n_members) : nullptr;
    SyncOnExit guard(h->stream);                              // params / world_member are the caller's
    HIPCHK(hipMemcpyAsync(d_w, params, wbytes, hipMemcpyHostToDevice, h->stream));
    if (world_member) HIPCHK(hipMemcpyAsync(d_m, world_member, mbytes, hipMemcpyHostToDevice, h->stream));
    const int n = p.batch * (agent_end - agent_begin);
    hipLaunchKernelGGL(policy_mlp, dim3((n + 3) / 4), dim3(64), 0, h->stream, h->scratch, d_w, d_m, p.batch,
                       p.n_agents, agent_begin, agent_end, h->action);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    guard.disarm();
    return DW_OK;
}

int dw_policy_mlp(dw_handle* h, const double* params, int32_t n_params, int32_t agent_begin, int32_t agent_end,
                  double L_init) {
    NEED(h && params, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    NEED(n_params == 63 * 16 + 16 * 32 + 32 * 9, DW_EINVAL, "the MLP policy has 1808 parameters (63-16-32-9), got %d", n_params);
    return policy_mlp_impl(h, params, 1, nullptr, agent_begin, agent_end, L_init);
}

int dw_policy_mlp_population(dw_handle* h, const double* params, int32_t n_members, const int32_t* world_member,
                             int32_t agent_begin, int32_t agent_end, double L_init) {
    NEED(h && params && world_member, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    NEED(n_members >= 1, DW_EINVAL, "n_members < 1");
    return policy_mlp_impl(h, params, n_members, world_member, agent_begin, agent_end, L_init);
}

// K steps with MLP policies without a host round trip: parameters and member maps go to the device once;
// per step: observe (all agents) -> policy_mlp for [0, split) and [split, N) -> update_agents -> step ->
// reward / done of the step into the [K][B][N] device buffers; one download and synchronisation at the end.
int dw_run_episode_mlp(dw_handle* h, int32_t nsteps, const double* L_schedule, const double* params, int32_t n_members,
                       const int32_t* member_a, const int32_t* member_b, int32_t split, double L_init, double* reward,
                       uint8_t* done) {
    NEED(h && L_schedule, DW_EINVAL, "null argument");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    NEED(nsteps >= 1 && nsteps <= 4096, DW_EINVAL, "nsteps must be in 1..4096");
    NEED(n_members >= 1, DW_EINVAL, "n_members < 1");
    NEED(params || (h->mlp_w && h->mlp_members == n_members), DW_ESTATE,
         "params == NULL but no parameter sets of %d members are on the device", n_members);
    NEED(p.collision_mode == 0, DW_EINVAL, "collision_mode=1 is not implemented on the device");
    NEED(split >= 0 && split <= p.n_agents, DW_EINVAL, "split outside 0..n_agents");
    NEED(h->have_state && h->have_agents, DW_ESTATE, "no state / agents");
    const int B = p.batch, N = p.n_agents;
    const size_t K = (size_t)nsteps, bn = (size_t)B * N;
    NEED(bn > 0, DW_EINVAL, "no agents");
    for (int b = 0; b < B; ++b) {
        NEED(!member_a || (member_a[b] >= 0 && member_a[b] < n_members), DW_EINVAL, "world %d: member out of range", b);
        NEED(!member_b || (member_b[b] >= 0 && member_b[b] < n_members), DW_EINVAL, "world %d: member out of range", b);
    }
    NEED(n_members == 1 || (member_a && member_b), DW_EINVAL, "several parameter sets need both member maps");
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t wbytes = sizeof(double) * 1808 * (size_t)n_members;
    const size_t o_ma = 0, o_mb = up(o_ma + sizeof(int) * B), o_r = up(o_mb + sizeof(int) * B);
    const size_t o_d = up(o_r + sizeof(double) * K * bn), o_p32 = up(o_d + K * bn), o_ls = up(o_p32 + sizeof(PhysF32) * K);
    const size_t total = up(o_ls + sizeof(double) * K);
    if (int erc = ensure_ep_buf(h, total)) return erc;
    if (params) {                                               // the sets stay on the device for later calls (params == NULL)
        if (h->mlp_members != n_members) {
            (void)hipFree(h->mlp_w);
            h->mlp_w = nullptr; h->mlp_members = 0;
            HIPCHK(hipMalloc(&h->mlp_w, wbytes));
            h->mlp_members = n_members;
        }
    }
    const double* d_w = h->mlp_w;
    const int* d_ma = member_a ? reinterpret_cast<const int*>(h->ep_buf + o_ma) : nullptr;
    const int* d_mb = member_b ? reinterpret_cast<const int*>(h->ep_buf + o_mb) : nullptr;
    double* d_r = reinterpret_cast<double*>(h->ep_buf + o_r);
    unsigned char* d_d = h->ep_buf + o_d;
    SyncOnExit guard(h->stream);                              // params / member maps are the caller's
    if (params) HIPCHK(hipMemcpyAsync(h->mlp_w, params, wbytes, hipMemcpyHostToDevice, h->stream));
    if (member_a) HIPCHK(hipMemcpyAsync(h->ep_buf + o_ma, member_a, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    if (member_b) HIPCHK(hipMemcpyAsync(h->ep_buf + o_mb, member_b, sizeof(int) * B, hipMemcpyHostToDevice, h->stream));
    // Small worlds with a quantised state and a quantised retained previous state: the rest of the chunk in ONE
    // launch, worlds in LDS (episode_mlp).  Until then - the first two steps of an episode, whose current /
    // previous state is the un-quantised upload - and for large worlds: one launch sequence per step.
    const int Cc = p.height * p.width;
    const int wpb = Cc <= 256 ? 4 : (Cc <= 1024 ? 2 : 1);
    // H*W <= 256 with at most four agents (the ES trainers' own 16x16 x 4): one wave per world (dw_episode_wave.hpp)
    const bool wave_kernel = Cc <= kEwMaxCells && 16 * N <= 64 && !h->sw.no_episode_wave;
    const size_t lds = wave_kernel ? episode_wave_shared_bytes() + episode_mlp_wave_world_bytes(Cc, N) * 4
                                   : episode_mlp_world_bytes(Cc, N) * wpb;
    const bool small = Cc <= 4096 && lds <= 160 * 1024 && p.precision != DW_PRECISION_F64 &&
                       !h->sw.no_episode_kernel;
    std::vector<PhysF32> p32;
    SyncOnExit guard2(h->stream);                             // p32 (filled below) must outlive its upload
    size_t t = 0;
    while (t < K) {
        if (small && cur_quantised(h) && h->stepped && h->unq == OWN_NONE) {
            const size_t Kr = K - t;
            p32.resize(Kr);
            for (size_t i = 0; i < Kr; ++i) p32[i] = derive_f32(p, L_schedule[t + i]);
            HIPCHK(hipMemcpyAsync(h->ep_buf + o_p32, p32.data(), sizeof(PhysF32) * Kr, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->ep_buf + o_ls, L_schedule + t, sizeof(double) * Kr, hipMemcpyHostToDevice, h->stream));
            StatsDev* stats = h->stats2[h->sp];
            if (!wave_kernel) HIPCHK(hipMemsetAsync(stats, 0, sizeof(StatsDev) * (B + 1), h->stream));   // (the wave kernel assigns every record)
            EpisodeMlpIO io;
            const int cur = h->cur, prev = 1 - h->cur;
            io.L = h->L16[cur]; io.D = h->D16[cur]; io.prevL = h->L16[prev]; io.prevD = h->D16[prev];
            io.idx = h->idx; io.st = h->st;
            io.P32 = reinterpret_cast<const PhysF32*>(h->ep_buf + o_p32);
            io.Ls = reinterpret_cast<const double*>(h->ep_buf + o_ls);
            io.weights = d_w; io.member_a = d_ma; io.member_b = d_mb;
            io.reward = d_r + t * bn; io.done = d_d + t * bn;
            io.stats = stats; io.fixups = &stats[B].sum_l;
            io.action = h->action;
            const bool ex = p.precision == DW_PRECISION_EXACT;
            if (wave_kernel) {
                auto kern = ex ? episode_mlp_wave<true> : episode_mlp_wave<false>;
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds));
                const EpisodeMlpWaveArgs A{io, B, N, p.height, p.width, (int)Kr, p.obs_mask, (int)split, p.agent_gamma,
                                           h->L_last, make_f64(p, L_schedule[t])};
                hipLaunchKernelGGL(kern, dim3((unsigned)((B + 3) / 4)), dim3(256), lds, h->stream, A);
            } else {
                auto kern = ex ? episode_mlp<true> : episode_mlp<false>;
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)((B + wpb - 1) / wpb)), dim3(256), lds, h->stream, io, B, N, p.height,
                                   p.width, wpb, (int)Kr, p.obs_mask, p.agent_gamma, make_f64(p, L_schedule[t]), h->L_last,
                                   (int)split);
            }
            HIPCHK(hipGetLastError());
            h->stepped = true;
            h->L_last = L_schedule[K - 1];
            t = K;
            break;
        }
        int rc = observe_into_scratch(h, L_init, 0);
        if (rc) return rc;
        // both halves in one launch (agents [split, N) read member_b), reward / done written by the grazing
        // kernel: 4 instead of 6 launches per step - the loop is bound by the host thread that issues them
        hipLaunchKernelGGL(policy_mlp, dim3((unsigned)((bn + 3) / 4)), dim3(64), 0, h->stream, h->scratch, d_w, d_ma, B, N,
                           0, N, h->action, d_mb, split);
        HIPCHK(hipGetLastError());
        rc = launch_agents(h, h->action, B, N, false, d_r + t * bn, d_d + t * bn);
        if (rc) return rc;
        rc = launch_forward(h, L_schedule[t]);
        if (rc) return rc;
        ++t;
    }
    if (reward) HIPCHK(hipMemcpyAsync(reward, d_r, sizeof(double) * K * bn, hipMemcpyDeviceToHost, h->stream));
    if (done) HIPCHK(hipMemcpyAsync(done, d_d, K * bn, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    guard2.disarm();
    guard.disarm();
    return DW_OK;
}

int dw_lifespan_reset(dw_handle* h) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const size_t bn = (size_t)p.batch * (p.n_agents > 0 ? p.n_agents : 1);
    HIPCHK(hipMemsetAsync(h->done_at, 0, sizeof(int) * p.batch, h->stream));
    HIPCHK(hipMemsetAsync(h->agents_done_at, 0, sizeof(int) * bn, h->stream));
    HIPCHK(hipMemsetAsync(h->n_alive, 0, sizeof(int), h->stream));
    return DW_OK;
}

int dw_lifespan_accumulate(dw_handle* h, uint32_t threshold_k) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    HIPCHK(hipMemsetAsync(h->n_alive, 0, sizeof(int), h->stream));
    const int n = p.batch * (p.n_agents > 0 ? p.n_agents : 1);
    hipLaunchKernelGGL(lifespan_accumulate, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->stats2[h->sp], h->st, p.batch,
                       p.n_agents, threshold_k, h->done_at, h->agents_done_at, h->n_alive);
    HIPCHK(hipGetLastError());
    return DW_OK;
}

int dw_lifespan_download(dw_handle* h, int32_t* done_at, int32_t* agents_done_at, int32_t* n_worlds_alive) {
    NEED(h, DW_EINVAL, "null handle");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    if (done_at) HIPCHK(hipMemcpyAsync(done_at, h->done_at, sizeof(int) * p.batch, hipMemcpyDeviceToHost, h->stream));
    if (agents_done_at && p.n_agents)
        HIPCHK(hipMemcpyAsync(agents_done_at, h->agents_done_at, sizeof(int) * (size_t)p.batch * p.n_agents,
                              hipMemcpyDeviceToHost, h->stream));
    if (n_worlds_alive) HIPCHK(hipMemcpyAsync(n_worlds_alive, h->n_alive, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_run_episode(dw_handle* h, int32_t nsteps, const double* L_schedule, int policy_mode,
                   const uint8_t* use_table, const int8_t* table, uint32_t threshold_k, uint8_t* world_alive,
                   uint8_t* agent_ok) {
    NEED(h && L_schedule, DW_EINVAL, "null argument");
    NEED(policy_mode >= 0 && policy_mode <= 3, DW_EINVAL, "bad policy mode");
    return run_episode_impl(h, nsteps, L_schedule, policy_mode, use_table, table, threshold_k, world_alive, agent_ok);
}

static bool episode_kernel_applies(const dw_handle* h) {
    const dw_params& p = h->prm;
    return p.height * p.width <= 4096 && p.precision != DW_PRECISION_F64 && p.collision_mode == 0 &&
           cur_quantised(h) && !h->sw.no_episode_kernel;
}

// dw_run_episode for worlds that do not fit LDS: the same K steps as K x (policy, dw_step) issued
// back-to-back on the handle's stream - policy kernel or table slice -> update_agents -> step kernel ->
// flags from the step's reductions - with no host round trip in between; one synchronisation at the end.
static int run_episode_stepwise(dw_handle* h, int32_t nsteps, const double* L_schedule, int policy_mode,
                                const uint8_t* use_table, const int8_t* table, uint32_t threshold_k,
                                uint8_t* world_alive, uint8_t* agent_ok) {
    const dw_params& p = h->prm;
    const int N = p.n_agents, B = p.batch;
    const size_t K = (size_t)nsteps, bn = (size_t)B * N;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_tab = 0, o_wa = up(o_tab + K * bn), o_ok = up(o_wa + K * B), o_code = up(o_ok + K * bn);
    const size_t o_ps = up(o_code + bn), total = up(o_ps + sizeof(unsigned int) * 2 * B) + 256;
    if (int erc = ensure_ep_buf(h, total)) return erc;
    SyncOnExit guard(h->stream);                              // `table` is the caller's
    if (table && bn) HIPCHK(hipMemcpyAsync(h->ep_buf + o_tab, table, K * bn, hipMemcpyHostToDevice, h->stream));
    const int nflag = B > (int)bn ? B : (int)bn;
    // Step pairs on wide grids (dw_agents_fused.hpp): policy_t, graze_t, ONE fused launch for forward_t and
    // forward_{t+1}, then the agents' step t+1 recomputed around the agents and patched into the result.
    // Needs no per-step world reductions (the caller passed world_alive == NULL); the last step of the
    // call stays an ordinary step, so the handle ends exactly as after K calls of dw_step.
    const bool may_pair = h->allow_fuse && h->use_stream && bn && N <= kLookaheadMaxAgents &&
                          policy_mode != kPolicySkipAgents && p.precision != DW_PRECISION_F64 &&
                          !h->sw.no_agent_fuse;
    // With per-step world flags the fused launch also reduces what the flags of both steps need (STATS
    // variants: exact step-1 maximum, count of certain step-2 values above the threshold).
    unsigned int* pstats = world_alive ? reinterpret_cast<unsigned int*>(h->ep_buf + o_ps) : nullptr;
    // zeroed once: every agents_lookahead_patch launch leaves its world's words cleared for the next pair
    if (pstats && may_pair) HIPCHK(hipMemsetAsync(pstats, 0, sizeof(unsigned int) * 2 * B, h->stream));
    // action codes of a pair's second step when they come from no table: one byte value for the whole episode
    const int uniform_code = policy_mode == DW_POLICY_ZEROS ? 0 : (policy_mode == DW_POLICY_ARGMIN ? 0xFE : 0xFF);
    if (may_pair && policy_mode != DW_POLICY_TABLE)
        HIPCHK(hipMemsetAsync(h->ep_buf + o_code, uniform_code, bn, h->stream));
    auto greedy = [&](int argmin, int codes) { return launch_policy_greedy(h, argmin, nullptr, codes); };
    // policy + update_agents of the step after a pair run inside that pair's patch kernel (phase E) while the chunk
    // continues: two launches per pair instead of four (DW_NO_AGENT_PREAPPLY: experiments)
    const bool no_preapply = h->sw.no_agent_preapply;
    bool pre_applied = false;
    for (size_t t = 0; t < K; ++t) {
        const bool pair = may_pair && cur_quantised(h) && K - t >= 3;
        if (pre_applied) {
            pre_applied = false;                             // step t's policy and grazing were done by the last patch
        } else if (bn && policy_mode != kPolicySkipAgents) {
            const bool from_table = policy_mode == DW_POLICY_TABLE || (use_table && use_table[t]);
            if (from_table) {
                hipLaunchKernelGGL(actions_from_table, dim3((unsigned)((bn + 255) / 256)), dim3(256), 0, h->stream,
                                   reinterpret_cast<const signed char*>(h->ep_buf + o_tab + t * bn), (int)bn, h->action);
                if (int prc = greedy(0, 1)) return prc;                      // codes -1 / -2: greedy / anti-greedy
            } else if (policy_mode == DW_POLICY_ZEROS) {
                HIPCHK(hipMemsetAsync(h->action, 0, sizeof(int) * bn, h->stream));
            } else {
                if (int prc = greedy(policy_mode == DW_POLICY_ARGMIN ? 1 : 0, 0)) return prc;
            }
            HIPCHK(hipGetLastError());
            // a pair's first step: the agents' ok flags straight from the grazing kernel
            int rc = launch_agents(h, h->action, B, N, false, nullptr, nullptr, pair ? h->ep_buf + o_ok + t * bn : nullptr);
            if (rc) return rc;
        }
        if (pair) {
            const double L1 = L_schedule[t], L2 = L_schedule[t + 1];
            int rc = launch_forward_fused2(h, L1, L2, pstats, (float)threshold_k);   // pstats: zero (see above)
            if (rc) return rc;
            // codes of step t+1: the caller's table slice, or one byte value for the whole ensemble
            const bool tab2 = policy_mode == DW_POLICY_TABLE || (use_table && use_table[t + 1]);
            unsigned char* codes = tab2 ? h->ep_buf + o_tab + (t + 1) * bn : h->ep_buf + o_code;
            LookaheadArgs A;
            A.inL = h->L16[1 - h->cur]; A.inD = h->D16[1 - h->cur];
            A.outL = h->L16[h->cur]; A.outD = h->D16[h->cur];
            A.idx = h->idx; A.st = h->st;
            A.code = reinterpret_cast<const signed char*>(codes);
            A.agent_ok = h->ep_buf + o_ok + (t + 1) * bn;
            A.code_next = nullptr;
            A.agent_ok_next = nullptr;
            A.action_out = h->action;
            if (t + 2 < K && !no_preapply) {                   // the chunk continues with step t+2
                const bool tab3 = policy_mode == DW_POLICY_TABLE || (use_table && use_table[t + 2]);
                A.code_next = reinterpret_cast<const signed char*>(tab3 ? h->ep_buf + o_tab + (t + 2) * bn : h->ep_buf + o_code);
                A.agent_ok_next = h->ep_buf + o_ok + (t + 2) * bn;
                pre_applied = true;
            }
            A.alive_t = pstats ? h->ep_buf + o_wa + t * B : nullptr;
            A.alive_t1 = pstats ? h->ep_buf + o_wa + (t + 1) * B : nullptr;
            A.pstats = pstats; A.thr = threshold_k;
            A.B = B; A.N = N; A.H = p.height; A.W = p.width; A.mask = p.obs_mask;
            A.agent_gamma = p.agent_gamma;
            A.P1 = derive_f32(p, L1); A.P2 = derive_f32(p, L2);
            A.P64 = make_f64(p, L1); A.La = L1; A.Lb = L2;
            if (p.precision == DW_PRECISION_EXACT)
                hipLaunchKernelGGL((agents_lookahead_patch<true>), dim3((unsigned)B), dim3(64), 0, h->stream, A);
            else
                hipLaunchKernelGGL((agents_lookahead_patch<false>), dim3((unsigned)B), dim3(64), 0, h->stream, A);
            HIPCHK(hipGetLastError());
            ++t;                                             // two steps done
            continue;
        }
        int rc = launch_forward(h, L_schedule[t]);
        if (rc) return rc;
        hipLaunchKernelGGL(episode_flags, dim3((unsigned)((nflag + 255) / 256)), dim3(256), 0, h->stream,
                           h->stats2[h->sp], h->st, B, N, threshold_k, h->ep_buf + o_wa + t * B, h->ep_buf + o_ok + t * bn);
        HIPCHK(hipGetLastError());
    }
    if (world_alive) HIPCHK(hipMemcpyAsync(world_alive, h->ep_buf + o_wa, K * B, hipMemcpyDeviceToHost, h->stream));
    if (agent_ok && bn) HIPCHK(hipMemcpyAsync(agent_ok, h->ep_buf + o_ok, K * bn, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    guard.disarm();
    return DW_OK;
}

static int run_episode_impl(dw_handle* h, int32_t nsteps, const double* L_schedule, int policy_mode,
                            const uint8_t* use_table, const int8_t* table, uint32_t threshold_k,
                            uint8_t* world_alive, uint8_t* agent_ok) {
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    NEED(nsteps >= 1 && nsteps <= 4096, DW_EINVAL, "nsteps must be in 1..4096");
    NEED(p.precision != DW_PRECISION_F64, DW_EINVAL, "dw_run_episode supports exact and fast precision");
    NEED(p.collision_mode == 0, DW_EINVAL, "collision_mode=1 is not implemented on the device");
    const int C = p.height * p.width, N = p.n_agents, B = p.batch;
    NEED(h->have_state, DW_ESTATE, "no state uploaded");
    NEED(N == 0 || h->have_agents, DW_ESTATE, "no agents uploaded");
    NEED(cur_quantised(h), DW_ESTATE, "the current state is not quantised yet; take the first step with dw_step");
    NEED(policy_mode != DW_POLICY_TABLE || table, DW_EINVAL, "DW_POLICY_TABLE needs a table");
    if (use_table && !table)
        for (int t = 0; t < nsteps; ++t) NEED(!use_table[t], DW_EINVAL, "use_table set but no table given");
    if (C > 4096 || h->sw.no_episode_kernel)
        return run_episode_stepwise(h, nsteps, L_schedule, policy_mode, use_table, table, threshold_k, world_alive,
                                    agent_ok);
    // H*W <= 256 (the README sweep's 8x8, the ES trainers' 16x16): one wave per world (dw_episode_wave.hpp)
    const bool wave_kernel = C <= kEwMaxCells && N <= 64 && !h->sw.no_episode_wave;
    const int wpb = C <= 256 ? 4 : (C <= 1024 ? 2 : 1);
    const size_t world_bytes = wave_kernel ? episode_wave_world_bytes(C, N) : episode_world_bytes(C, N);
    const size_t lds = world_bytes * wpb + (wave_kernel ? episode_wave_shared_bytes() : 0);
    NEED(lds <= 160 * 1024, DW_EINVAL, "too many agents for the LDS-resident episode kernel");
    // device staging: [P32 K][Ls K][use_table K][table K*B*N] | [world_alive K*B][agent_ok K*B*N]
    const size_t K = (size_t)nsteps, bn = (size_t)B * N;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_p32 = 0, o_ls = up(o_p32 + sizeof(PhysF32) * K), o_ut = up(o_ls + sizeof(double) * K);
    const size_t o_tab = up(o_ut + K), o_wa = up(o_tab + K * bn), o_ok = up(o_wa + K * B), total = up(o_ok + K * bn);
    if (int erc = ensure_ep_buf(h, total)) return erc;
    // The inputs are assembled in a page-locked image of the staging buffer and go up in ONE copy; the flags come back
    // in ONE copy (round 3: four pageable uploads, a memset and two pageable downloads per chunk - 88 us of host time
    // per 64-step chunk of 1000 8x8 worlds against 116 us of kernel).  Beyond 64 MiB: straight from / to the caller's arrays.
    const bool staged = total <= ((size_t)64 << 20);
    if (staged && h->ep_pinned_bytes < total) {
        if (h->ep_pinned) HIPCHK(hipHostFree(h->ep_pinned));
        h->ep_pinned = nullptr; h->ep_pinned_bytes = 0;
        size_t want = h->ep_pinned_bytes ? h->ep_pinned_bytes * 2 : ((size_t)1 << 20);
        if (want < total) want = total;
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->ep_pinned), want, hipHostMallocDefault));
        h->ep_pinned_bytes = want;
    }
    std::vector<PhysF32> p32_own;
    std::vector<unsigned char> ut_own;
    PhysF32* p32 = nullptr;
    unsigned char* ut = nullptr;
    if (staged) {
        p32 = reinterpret_cast<PhysF32*>(h->ep_pinned + o_p32);
        ut = h->ep_pinned + o_ut;
    } else {
        p32_own.resize(K); ut_own.resize(K);
        p32 = p32_own.data(); ut = ut_own.data();
    }
    for (size_t t = 0; t < K; ++t) p32[t] = derive_f32(p, L_schedule[t]);
    if (use_table) std::memcpy(ut, use_table, K); else std::memset(ut, 0, K);
    SyncOnExit guard(h->stream);                              // the images above and the caller's arrays
    if (staged) {
        std::memcpy(h->ep_pinned + o_ls, L_schedule, sizeof(double) * K);
        if (table && bn) std::memcpy(h->ep_pinned + o_tab, table, K * bn);
        HIPCHK(hipMemcpyAsync(h->ep_buf, h->ep_pinned, (table && bn) ? o_tab + K * bn : o_ut + K, hipMemcpyHostToDevice, h->stream));
    } else {
        HIPCHK(hipMemcpyAsync(h->ep_buf + o_p32, p32, sizeof(PhysF32) * K, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->ep_buf + o_ls, L_schedule, sizeof(double) * K, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->ep_buf + o_ut, ut, K, hipMemcpyHostToDevice, h->stream));
        if (table && bn) HIPCHK(hipMemcpyAsync(h->ep_buf + o_tab, table, K * bn, hipMemcpyHostToDevice, h->stream));
    }
    StatsDev* stats = h->stats2[h->sp];
    // episode_wave ASSIGNS every world's whole record (its float64 count in `reserved`) and the counter record behind them:
    // nothing to clear; episode_small accumulates: cleared as before
    if (!wave_kernel) HIPCHK(hipMemsetAsync(stats, 0, sizeof(StatsDev) * (B + 1), h->stream));
    EpisodeIO io;
    const int cur = h->cur, prev = 1 - h->cur;
    io.L = h->L16[cur]; io.D = h->D16[cur]; io.prevL = h->L16[prev]; io.prevD = h->D16[prev];
    io.idx = h->idx; io.st = h->st;
    io.P32 = reinterpret_cast<const PhysF32*>(h->ep_buf + o_p32);
    io.Ls = reinterpret_cast<const double*>(h->ep_buf + o_ls);
    io.use_table = h->ep_buf + o_ut;
    io.table = reinterpret_cast<const signed char*>(h->ep_buf + o_tab);
    io.world_alive = h->ep_buf + o_wa;
    io.agent_ok = h->ep_buf + o_ok;
    io.stats = stats;
    io.fixups = &stats[B].sum_l;
    io.action = (N > 0 && policy_mode != kPolicySkipAgents) ? h->action : nullptr;
    const PhysF64 P64 = make_f64(p, L_schedule[0]);
    const dim3 grid((unsigned)((B + wpb - 1) / wpb));
    const bool ex = p.precision == DW_PRECISION_EXACT;
    if (wave_kernel) {
        io.use_table = use_table ? h->ep_buf + o_ut : nullptr;
        io.table = (table && bn) ? io.table : nullptr;
        auto kern = ex ? episode_wave<true> : episode_wave<false>;
        static size_t lds_set[2] = {0, 0};                      // the attribute only ever has to grow
        if (lds_set[ex] < lds) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            lds_set[ex] = lds;
        }
        const EpisodeWaveArgs A{io, B, N, p.height, p.width, nsteps, policy_mode, p.obs_mask, threshold_k, p.agent_gamma, P64};
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, h->stream, A);
    } else {
        auto kern = ex ? episode_small<true> : episode_small<false>;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, h->stream, io, B, N, p.height, p.width, wpb, nsteps, policy_mode,
                           p.obs_mask, p.agent_gamma, threshold_k, P64);
    }
    HIPCHK(hipGetLastError());
    const bool want_ok = agent_ok && bn;
    if (staged) {
        if (world_alive || want_ok) {
            const size_t lo = world_alive ? o_wa : o_ok, hi = want_ok ? o_ok + K * bn : o_wa + K * B;
            HIPCHK(hipMemcpyAsync(h->ep_pinned + lo, h->ep_buf + lo, hi - lo, hipMemcpyDeviceToHost, h->stream));
        }
    } else {
        if (world_alive) HIPCHK(hipMemcpyAsync(world_alive, h->ep_buf + o_wa, K * B, hipMemcpyDeviceToHost, h->stream));
        if (want_ok) HIPCHK(hipMemcpyAsync(agent_ok, h->ep_buf + o_ok, K * bn, hipMemcpyDeviceToHost, h->stream));
    }
    h->unq = OWN_NONE;
    h->stepped = true;
    h->L_last = L_schedule[K - 1];
    release_unquantised(h);
    HIPCHK(hipStreamSynchronize(h->stream));      // flags are returned
    guard.disarm();
    if (staged) {
        if (world_alive) std::memcpy(world_alive, h->ep_pinned + o_wa, K * B);
        if (want_ok) std::memcpy(agent_ok, h->ep_pinned + o_ok, K * bn);
    }
    return DW_OK;
}

// ---- device-side snapshots of the current state ----------------------------------------------------
// All regions of a snapshot in ONE launch (seven device-to-device copies were seven stream operations: ~55 us in front of
// every chunk of the fitness harness, whose whole chunk kernel takes 0.5 ms)
struct CopyJobs {
    const void* src[8];
    void* dst[8];
    unsigned long long bytes[8];
    int n;
};
__global__ __launch_bounds__(256) void copy_regions(CopyJobs J) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = 0; j < J.n; ++j) {
        const size_t nb = J.bytes[j];
        // (every region starts at a hipMalloc'ed address: 256-byte aligned)
        const uint4* s16 = static_cast<const uint4*>(J.src[j]);
        uint4* d16 = static_cast<uint4*>(J.dst[j]);
        for (size_t i = t0; i < nb / 16; i += stride) d16[i] = s16[i];
        const unsigned char* s1 = static_cast<const unsigned char*>(J.src[j]);
        unsigned char* d1 = static_cast<unsigned char*>(J.dst[j]);
        for (size_t i = (nb & ~(size_t)15) + t0; i < nb; i += stride) d1[i] = s1[i];      // the last < 16 bytes
    }
}
static int launch_copy_regions(dw_handle* h, const CopyJobs& J) {
    size_t most = 0;
    for (int j = 0; j < J.n; ++j) {
        NEED(((reinterpret_cast<uintptr_t>(J.src[j]) | reinterpret_cast<uintptr_t>(J.dst[j])) & 15) == 0, DW_EINVAL,
             "snapshot region not 16-byte aligned");
        most = J.bytes[j] > most ? J.bytes[j] : most;
    }
    if (J.n == 0 || most == 0) return DW_OK;
    size_t blocks = (most / 16 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
    hipLaunchKernelGGL(copy_regions, dim3((unsigned)blocks), dim3(256), 0, h->stream, J);
    HIPCHK(hipGetLastError());
    return DW_OK;
}

int dw_snapshot_save_slot(dw_handle* h, int32_t slot) {
    NEED(h, DW_EINVAL, "null handle");
    NEED(slot >= 0 && slot < DW_SNAPSHOT_SLOTS, DW_EINVAL, "snapshot slot out of range");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    NEED(h->have_state, DW_ESTATE, "no state uploaded");
    NEED(cur_quantised(h), DW_ESTATE, "the current state is an un-quantised upload; take a step first");
    dw_handle::Snapshot& sn = h->snap[slot];
    const size_t bn = (size_t)p.batch * p.n_agents;
    const size_t pb = sizeof(plane_t) * h->cells;
    if (!sn.L) {
        HIPCHK(hipMalloc(&sn.L, pb));
        HIPCHK(hipMalloc(&sn.D, pb));
        HIPCHK(hipMalloc(&sn.stats, h->stats_bytes));
        if (bn) {
            HIPCHK(hipMalloc(&sn.idx, sizeof(int) * 2 * bn));
            HIPCHK(hipMalloc(&sn.st, sizeof(double) * bn));
        }
    }
    CopyJobs J;
    J.n = 0;
    auto job = [&J](void* dst, const void* src, size_t bytes) { J.dst[J.n] = dst; J.src[J.n] = src; J.bytes[J.n] = bytes; ++J.n; };
    job(sn.L, h->L16[h->cur], pb);
    job(sn.D, h->D16[h->cur], pb);
    job(sn.stats, h->stats2[h->sp], h->stats_bytes);
    // the previous state too: observations (temperature channels) and the temp / beta / growth caches are
    // derived from it, so a replay from the snapshot must see the same one.  An un-quantised previous state
    // (the step after an upload) stays where it is: its buffers are not reused before the next upload, which
    // invalidates the snapshot.
    sn.stepped = h->stepped;
    sn.L_last = h->L_last;
    sn.unq = h->unq;
    if (h->stepped && h->unq != OWN_PREV) {
        if (!sn.PL) {
            HIPCHK(hipMalloc(&sn.PL, pb));
            HIPCHK(hipMalloc(&sn.PD, pb));
        }
        job(sn.PL, h->L16[1 - h->cur], pb);
        job(sn.PD, h->D16[1 - h->cur], pb);
    }
    sn.agents = bn && h->have_agents;
    if (sn.agents) {
        job(sn.idx, h->idx, sizeof(int) * 2 * bn);
        job(sn.st, h->st, sizeof(double) * bn);
    }
    if (int rc = launch_copy_regions(h, J)) return rc;
    sn.valid = true;
    return DW_OK;
}

int dw_snapshot_restore_slot(dw_handle* h, int32_t slot) {
    NEED(h, DW_EINVAL, "null handle");
    NEED(slot >= 0 && slot < DW_SNAPSHOT_SLOTS, DW_EINVAL, "snapshot slot out of range");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    const dw_handle::Snapshot& sn = h->snap[slot];
    NEED(sn.valid, DW_ESTATE, "no snapshot saved in this slot (or a later upload / dw_init_random invalidated it)");
    const size_t bn = (size_t)p.batch * p.n_agents;
    const size_t pb = sizeof(plane_t) * h->cells;
    CopyJobs J;
    J.n = 0;
    auto job = [&J](void* dst, const void* src, size_t bytes) { J.dst[J.n] = dst; J.src[J.n] = src; J.bytes[J.n] = bytes; ++J.n; };
    job(h->L16[h->cur], sn.L, pb);
    job(h->D16[h->cur], sn.D, pb);
    job(h->stats2[h->sp], sn.stats, h->stats_bytes);
    if (sn.agents) {
        job(h->idx, sn.idx, sizeof(int) * 2 * bn);
        job(h->st, sn.st, sizeof(double) * bn);
    }
    if (sn.stepped && sn.unq != OWN_PREV) {
        job(h->L16[1 - h->cur], sn.PL, pb);
        job(h->D16[1 - h->cur], sn.PD, pb);
    }
    if (int rc = launch_copy_regions(h, J)) return rc;
    HIPCHK(hipMemsetAsync(h->stats2[1 - h->sp], 0, h->stats_bytes, h->stream));
    h->unq = sn.unq;                            // OWN_PREV: the un-quantised initial state is still in its buffers
    h->stepped = sn.stepped;
    h->L_last = sn.L_last;
    return DW_OK;
}

int dw_snapshot_save(dw_handle* h) { return dw_snapshot_save_slot(h, 0); }
int dw_snapshot_restore(dw_handle* h) { return dw_snapshot_restore_slot(h, 0); }

// ---- plumbing ---------------------------------------------------------------------------------

int dw_set_stream(dw_handle* h, void* hip_stream) {
    NEED(h, DW_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->prm.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream) HIPCHK(hipStreamDestroy(h->stream));
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
    return DW_OK;
}

int dw_sync(dw_handle* h) {
    NEED(h, DW_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->prm.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DW_OK;
}

int dw_timer_start(dw_handle* h) {
    NEED(h, DW_EINVAL, "null handle");
    HIPCHK(hipSetDevice(h->prm.device));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    return DW_OK;
}

int dw_timer_stop(dw_handle* h, float* elapsed_ms) {
    NEED(h && elapsed_ms, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return DW_OK;
}

int dw_device_planes(dw_handle* h, int which, void** light, void** dark) {
    NEED(h && light && dark, DW_EINVAL, "null argument");
    NEED(which == DW_STATE_CURRENT || which == DW_STATE_PREVIOUS, DW_EINVAL, "bad state selector");
    const int buf = which == DW_STATE_CURRENT ? h->cur : 1 - h->cur;
    NEED(which != DW_STATE_CURRENT || cur_quantised(h), DW_ESTATE,
         "the current state is an un-quantised upload: it has no binary16 planes before the first step");
    *light = h->L16[buf];
    *dark = h->D16[buf];
    return DW_OK;
}

int dw_kernel_info(dw_handle* h, char* buf, size_t buflen) {
    NEED(h && buf && buflen, DW_EINVAL, "null argument");
    const dw_params& p = h->prm;
    const char* prec = p.precision == DW_PRECISION_EXACT ? "exact" : (p.precision == DW_PRECISION_FAST ? "fast" : "f64");
    if (h->use_stream && p.precision != DW_PRECISION_F64) {
        const StripGeom& g = h->sgeom;
        snprintf(buf, buflen,
                 "step_stream_%s<halo=%s> wave-strip=%dx256 cells, register window + DPP neighbours, %d-row blocks in "
                 "flight%s, %d strips, grid=%d x 256 threads (4 strips each), XCD-chunked",
                 prec, p.width < 256 ? "packed" : (p.width == 256 ? "rotate" : (p.width % 256 == 0 ? "dpp-old" : "general")),
                 g.SR,
                 p.precision == DW_PRECISION_EXACT ? DW_STREAM_RB_EXACT : DW_STREAM_RB_FAST,
                 p.precision == DW_PRECISION_EXACT ? ", in-wave float64 fix-up from an LDS queue" : "", g.nstrips,
                 g.chunk * 8);
        if (h->allow_fuse) {
            const size_t n = std::strlen(buf);
            snprintf(buf + n, buflen - n, "; dw_step_n fuses step pairs (step_stream_fused2%s)",
                     p.precision == DW_PRECISION_EXACT ? "_exact" : "");
        }
    } else if (h->tcq) {
        const int TR = (256 / h->tcq) * h->rpt;
        snprintf(buf, buflen,
                 "step_tiled<TCQ=%d,RPT=%d,%s> tile=%dx%d cells, %zu B LDS/workgroup, %d tiles, grid=%d x 256 threads, "
                 "XCD-chunked",
                 h->tcq, h->rpt, prec, TR, h->tcq * 4, h->tile_lds, h->geom.ntiles, h->geom.chunk * 8);
    } else {
        snprintf(buf, buflen, "step_generic<%s> one thread per cell, grid=(%d,%d) x 256 threads", prec,
                 (p.height * p.width + 255) / 256, p.batch);
    }
    if (h->sw.text[0]) {                                        // the switches this handle was created under
        const size_t n = std::strlen(buf);
        snprintf(buf + n, buflen - n, "; switches[%s]", h->sw.text);
    }
    return DW_OK;
}

int dw_audit_tie_bound(dw_handle* h, double L, double out[4]) {
    NEED(h && out, DW_EINVAL, "null argument");
    const dw_params& p = h->prm;
    HIPCHK(hipSetDevice(p.device));
    NEED(h->have_state && cur_quantised(h), DW_ESTATE, "the audit needs a quantised current state");
    int rc = ensure_scratch(h, 4 * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(h->scratch);
    HIPCHK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), h->stream));
    const dim3 g((unsigned)((p.height * p.width + 255) / 256), (unsigned)p.batch);
    hipLaunchKernelGGL(tie_audit, g, dim3(256), 0, h->stream, h->L16[h->cur], h->D16[h->cur], p.height, p.width,
                       derive_f32(p, L), make_f64(p, L), d, h->sym_albedo && h->use_stream ? 1 : 0);
    HIPCHK(hipGetLastError());
    unsigned long long r[4];
    HIPCHK(hipMemcpyAsync(r, d, sizeof(r), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::memcpy(&out[0], &r[0], sizeof(double));
    std::memcpy(&out[1], &r[1], sizeof(double));
    out[2] = (double)r[2];
    out[3] = (double)r[3];
    return DW_OK;
}

int dw_last_fixup_count(dw_handle* h, uint64_t* count) {
    NEED(h && count, DW_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->prm.device));
    // the counter behind the per-world records + the per-world counts of the one-wave-per-world episode kernels (`reserved`:
    // zero after every other kernel - the step kernels clear the whole buffer for the step after them)
    std::vector<StatsDev> st((size_t)h->prm.batch + 1);
    HIPCHK(hipMemcpyAsync(st.data(), h->stats2[h->sp], sizeof(StatsDev) * st.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long v = st[h->prm.batch].sum_l;
    for (int b = 0; b < h->prm.batch; ++b) v += st[b].reserved;
    *count = v;
    return DW_OK;
}

}  // extern "C"
