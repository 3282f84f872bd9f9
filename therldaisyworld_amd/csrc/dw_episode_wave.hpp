// dw_episode_wave.hpp — episode_wave: the device-resident episode loop for the SMALLEST worlds (H*W <= 256: the README
// sweep's 8x8 grids, the ES trainers' 16x16), one WAVE per world.
#pragma once
#include "dw_episode.hpp"
#include "dw_step_stream.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// episode_wave — round 4.  Same contract as episode_small (EpisodeIO; SURVEY.md 8(f) row N1: policy -> update_agents ->
// forward -> flags for K steps in one launch; ref notebooks/greedy_longevity_abatement.ipynb cell 2:28-57,
// daisy/agents/greedy.py:14-36, daisy_world_rl.py:181-244, :434-461), rebuilt around ONE WAVE = ONE WORLD:
//
//  * a 64-cell world is exactly one wave, a 256-cell world four cells per lane.  The four waves of a workgroup carry
//    four INDEPENDENT worlds, so nothing in the step loop is a workgroup barrier (episode_small crossed six per step,
//    each coupling four unrelated worlds); a wave's own LDS traffic is ordered by program order.
//  * the step is a chain of dependent latencies - 1000 worlds are 1000 waves on 1024 SIMDs: nothing hides them - so
//    every global access is taken out of it: the luminosity-dependent constants of up to 64 steps (P32[t], Ls[t],
//    use_table[t]) and the wave's slice of the host-drawn action table are copied into LDS once per 64-step segment
//    (the only workgroup barrier), the per-step flags are collected in registers (world: a 64-bit scalar mask over
//    the segment's steps; agents: one mask per agent lane) and written once per segment.
//  * agents are lanes, not a serial loop on lane 0: lane n holds agent n's state (float64, the reference's own
//    operations) and position in registers for the whole launch.  Moving is independent per agent; "the first to land
//    on a cell eats it all" (ref :186-216) is resolved with N wave-uniform v_readlane pairs - agent n gains the cell's
//    cover unless an earlier grazing agent landed on the same cell - then every grazed cell is zeroed.
//  * planes in LDS as interleaved (light, dark) float pairs: one ds_read_b64 per neighbour; the neighbour offsets of a
//    lane's cells are computed once per launch (no integer division in the loop).
//  * exact mode: a near-tie cell is re-evaluated in float64 on the spot by its own lane behind a wave-uniform test
//    (its 3x3 neighbourhood is in registers; 0.13 such cells per step of a 64-cell world).
//  * reductions: the lifespan flag needs "some cover > threshold" - a compare and a lane mask, no reduction; sums and
//    maximum are formed once, after the last step.
// ---------------------------------------------------------------------------------------------
constexpr int kEwSeg = 64;                                      // steps per LDS-resident constants segment
constexpr int kEwMaxCells = 256;
constexpr int kEwSlots = kEwMaxCells / 64;                      // cells per lane

__host__ __device__ constexpr size_t episode_wave_shared_bytes() {      // P32 | Ls | use_table of one segment
    return (size_t)kEwSeg * sizeof(PhysF32) + (size_t)kEwSeg * sizeof(double) + kEwSeg;
}
__host__ __device__ constexpr size_t episode_wave_world_bytes(int C, int N) {
    // planes [2][C] (light, dark) float pairs | action-table slice of a segment | agents' ok masks
    return (size_t)16 * C + ((size_t)kEwSeg * N + 15) / 16 * 16 + ((size_t)8 * N + 15) / 16 * 16;
}

// ---- pieces shared by episode_wave and episode_mlp_wave ----------------------------------------------------------
// this lane's cells (lane + 64 j) and their neighbour offsets, in float2 units inside one (light, dark) plane-pair buffer
struct EwCells {
    int rowU[kEwSlots], rowM[kEwSlots], rowD[kEwSlots], colL[kEwSlots], colM[kEwSlots], colR[kEwSlots];
    bool own[kEwSlots];
};
__device__ __forceinline__ void ew_cells_init(EwCells& G, int lane, int C, int H, int W, bool valid) {
#pragma unroll
    for (int j = 0; j < kEwSlots; ++j) {
        const int c = lane + 64 * j;
        G.own[j] = valid && c < C;
        const int cc0 = c < C ? c : 0;
        const int r = cc0 / W, cc = cc0 - r * W;
        G.rowU[j] = (r == 0 ? H - 1 : r - 1) * W; G.rowM[j] = r * W; G.rowD[j] = (r == H - 1 ? 0 : r + 1) * W;
        G.colL[j] = cc == 0 ? W - 1 : cc - 1; G.colM[j] = cc; G.colR[j] = cc == W - 1 ? 0 : cc + 1;
    }
}

// forward (ref :434-461) of one LDS-resident world by its wave: pc -> pn.  Exact mode: a near-tie cell is re-evaluated
// in float64 by its own lane behind a wave-uniform test, `q64()` supplying the float64 constants of the step only then.
// Returns "this lane wrote a cover above the threshold"; nfix counts this lane's float64 evaluations.
template <bool EXACT, typename Q64>
__device__ __forceinline__ bool ew_forward(const PhysF32& P, const float2* pc, float2* pn, const EwCells& G, int C,
                                           int lane, float thr_f, Q64 q64, unsigned int& nfix) {
    bool alive_here = false;
#pragma unroll
    for (int j = 0; j < kEwSlots; ++j) {
        if (j * 64 >= C) break;                                  // wave-uniform
        float2 nb[9];
        {
            const int rows[3] = {G.rowU[j], G.rowM[j], G.rowD[j]}, cols[3] = {G.colL[j], G.colM[j], G.colR[j]};
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                for (int e3 = 0; e3 < 3; ++e3) nb[a3 * 3 + e3] = pc[rows[a3] + cols[e3]];
        }
        const float li = nb[4].x, di = nb[4].y;
        const float El = (nb[1].x + nb[7].x) + (nb[3].x + nb[5].x);
        const float Cl = (nb[0].x + nb[6].x) + (nb[2].x + nb[8].x);
        const float Ed = (nb[1].y + nb[7].y) + (nb[3].y + nb[5].y);
        const float Cd = (nb[0].y + nb[6].y) + (nb[2].y + nb[8].y);
        const GrowthF32 g = growth_f32<EXACT || kFastSplit>(P, li, di, El, Cl, Ed, Cd);
        float kl, kd;
        if (EXACT) {
            bool tl, td;
            kl = finish_exact(P, li, g.gql, g.dKl, g.oml, tl);
            kd = finish_exact(P, di, g.gqd, g.dKd, g.omd, td);
            const bool tie = G.own[j] && (tl || td);
            if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {       // wave-uniform: rare
                if (tie) {
                    unsigned int wv9[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) wv9[i] = (unsigned)nb[i].x | ((unsigned)nb[i].y << 16);
                    const PhysF64 Q = q64();
                    const NewCoverF64 o64 = cell_f64_lean(Q, wv9);
                    kl = (float)dw_round3_k(o64.nl);
                    kd = (float)dw_round3_k(o64.nd);
                    ++nfix;
                }
            }
        } else {
            kl = finish_fast(li, g.dKl, g.fl);
            kd = finish_fast(di, g.dKd, g.fd);
        }
        if (G.own[j]) {
            pn[lane + 64 * j] = make_float2(kl, kd);
            alive_here = alive_here || fmaxf(kl, kd) > thr_f;
        }
    }
    return alive_here;
}

// the five cells an agent can see or reach, in the reference's candidate order - flat patch indices 3, 1, 7, 5 =
// (r,c-1) (r-1,c) (r+1,c) (r,c+1), which is also the order of the move codes a % 4 = 0, 1, 2, 3 - and its own: their
// covers in natural units (float64, exactly cover_k / 1000.0)
struct EwReach {
    int rU, rD, cL, cR;
    double nat0, nat1, nat2, nat3, nat_own;
};
__device__ __forceinline__ EwReach ew_reach(const float2* pc, int ar, int ac, int H, int W) {
    EwReach R;
    R.rU = (ar == 0 ? H - 1 : ar - 1); R.rD = (ar == H - 1 ? 0 : ar + 1);
    R.cL = (ac == 0 ? W - 1 : ac - 1); R.cR = (ac == W - 1 ? 0 : ac + 1);
    const int rowo = ar * W;
    const float2 cv0 = pc[rowo + R.cL], cv1 = pc[R.rU * W + ac], cv2 = pc[R.rD * W + ac], cv3 = pc[rowo + R.cR];
    const float2 own_cell = pc[rowo + ac];
    R.nat0 = dw_permille_to_natural((double)cv0.x) + dw_permille_to_natural((double)cv0.y);
    R.nat1 = dw_permille_to_natural((double)cv1.x) + dw_permille_to_natural((double)cv1.y);
    R.nat2 = dw_permille_to_natural((double)cv2.x) + dw_permille_to_natural((double)cv2.y);
    R.nat3 = dw_permille_to_natural((double)cv3.x) + dw_permille_to_natural((double)cv3.y);
    R.nat_own = dw_permille_to_natural((double)own_cell.x) + dw_permille_to_natural((double)own_cell.y);
    return R;
}

// update_agents (ref :181-244, collision_mode 0) with the agents as LANES (lane n = agent n; action a in 0..8): decay,
// move, "the first agent to graze a cell eats it all" resolved with one wave-uniform v_readlane + v_cmp (a lane mask) per
// agent, grazed cells zeroed, clip - straight-line select code for every lane (lanes without an agent are masked by
// is_agent).
__device__ __forceinline__ void ew_update_agents(int a, const EwReach& R, bool is_agent, int lane, int N, int W,
                                                 double agent_gamma, double& ast, int& ar, int& ac, float2* pc) {
    const double s0 = ast - agent_gamma;
    const bool alive = is_agent && s0 > 0.0;
    const int m = a & 3;
    const bool stay = a == 8;
    // (one select after the other: a nested conditional on m becomes a switch, i.e. a tree of exec-mask branches)
    // (values first: selecting between loads of R's members gets sunk into one load from a selected address, which
    // keeps R in scratch memory)
    const double n0 = R.nat0, n1 = R.nat1, n2 = R.nat2, n3 = R.nat3, n_own = R.nat_own;
    const int rU = R.rU, rD = R.rD, cL = R.cL, cR = R.cR;
    double gain = n0;
    gain = m == 1 ? n1 : gain;
    gain = m == 2 ? n2 : gain;
    gain = m == 3 ? n3 : gain;
    gain = stay ? n_own : gain;
    int nr = ar, nc = ac;
    nr = m == 1 ? rU : nr;
    nr = m == 2 ? rD : nr;
    nc = m == 0 ? cL : nc;
    nc = m == 3 ? cR : nc;
    nr = stay ? ar : nr;
    nc = stay ? ac : nc;
    ar = alive ? nr : ar;
    ac = alive ? nc : ac;
    const bool graze = alive && a > 4;
    const int o = ar * W + ac;
    // "an earlier agent grazes my cell" as lane masks: one v_readlane + one v_cmp (whose result IS the mask) per agent
    const int key = graze ? o : -1;                              // an agent that does not graze matches no cell
    unsigned long long beaten = 0ull;
    for (int mm = 0; mm < N; ++mm) {                             // wave-uniform
        const int km = __builtin_amdgcn_readlane(key, mm);
        beaten |= __builtin_amdgcn_ballot_w64(km == o) & (~1ull << mm);      // lanes above mm on mm's cell
    }
    const bool first = ((beaten >> lane) & 1ull) == 0ull;
    const double s1 = (graze && first) ? s0 + gain : s0;
    if (is_agent) ast = s1 < 0.0 ? 0.0 : (s1 > 1.0 ? 1.0 : s1);
    if (graze) pc[o] = make_float2(0.f, 0.f);                   // (every read of the step precedes it in program order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ONE argument struct: the float64 constants are needed only by the rare near-tie path and are read from the kernarg
// segment there (as by-value arguments they would sit in 34 SGPRs for the whole launch; `kernarg_struct`, dw_step_stream.hpp)
struct EpisodeWaveArgs {
    EpisodeIO io;
    int B, N, H, W, K, policy_mode, obs_mask;
    unsigned int thr;
    double agent_gamma;
    PhysF64 P64;                                                // cold
};

template <bool EXACT>
__global__ __launch_bounds__(256) void episode_wave(EpisodeWaveArgs A) {
    const EpisodeIO& io = A.io;
    const int B = A.B, N = A.N, H = A.H, W = A.W, K = A.K, policy_mode = A.policy_mode, obs_mask = A.obs_mask;
    const unsigned int thr = A.thr;
    const double agent_gamma = A.agent_gamma;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = H * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x * 4 + wv;
    const bool valid = b < B;                                   // (invalid waves still meet the segment barriers)
    PhysF32* const sP32 = reinterpret_cast<PhysF32*>(smem);
    double* const sLs = reinterpret_cast<double*>(smem + (size_t)kEwSeg * sizeof(PhysF32));
    unsigned char* const sUT = smem + (size_t)kEwSeg * sizeof(PhysF32) + (size_t)kEwSeg * sizeof(double);
    unsigned char* const wbase = smem + episode_wave_shared_bytes() + (size_t)wv * episode_wave_world_bytes(C, N);
    float2* const planes = reinterpret_cast<float2*>(wbase);    // [2][C]
    signed char* const sTab = reinterpret_cast<signed char*>(wbase + (size_t)16 * C);
    unsigned long long* const sOk = reinterpret_cast<unsigned long long*>(wbase + (size_t)16 * C + ((size_t)kEwSeg * N + 15) / 16 * 16);
    const bool with_agents = N > 0 && policy_mode != kPolicySkipAgents;
    const bool any_table = policy_mode == kPolicyTable || (policy_mode != kPolicyZeros && io.use_table != nullptr);

    EwCells G;
    ew_cells_init(G, lane, C, H, W, valid);
    int cur = 0;                                                // planes[cur*C ..]: the current state
    if (valid) {
#pragma unroll
        for (int j = 0; j < kEwSlots; ++j)
            if (G.own[j]) {
                const int c = lane + 64 * j;
                planes[c] = make_float2((float)io.L[(size_t)b * C + c], (float)io.D[(size_t)b * C + c]);
            }
    }
    // agent n lives in lane n
    const bool is_agent = valid && lane < N;
    const int alane = N > 0 ? min(lane, N - 1) : 0;             // (lanes without an agent shadow the last one's table entry)
    double ast = 0.0;
    int ar = 0, ac = 0;
    if (is_agent) {
        ast = io.st[(size_t)b * N + lane];
        ar = io.idx[((size_t)b * N + lane) * 2];
        ac = io.idx[((size_t)b * N + lane) * 2 + 1];
    }
    const float thr_f = (float)thr;
    unsigned int last_fix = 0;                                  // float64 re-evaluations of the last step (this lane)

    for (int t0 = 0; t0 < K; t0 += kEwSeg) {
        const int seg = min(kEwSeg, K - t0);
        // ---- the segment's constants and this wave's slice of the action table into LDS ----
        __syncthreads();                                        // (the previous segment's readers are done)
        for (int i = tid; i < seg * (int)(sizeof(PhysF32) / 4); i += 256)
            reinterpret_cast<unsigned int*>(sP32)[i] = reinterpret_cast<const unsigned int*>(io.P32 + t0)[i];
        for (int i = tid; i < seg; i += 256) {
            sLs[i] = io.Ls[t0 + i];
            sUT[i] = (policy_mode != kPolicyZeros && io.use_table) ? io.use_table[t0 + i] : 0;
        }
        if (valid && with_agents && any_table && io.table)
            for (int i = lane; i < seg * N; i += 64) {
                const int tt = i / N, n = i - tt * N;
                sTab[i] = io.table[((size_t)(t0 + tt) * B + b) * N + n];
            }
        __syncthreads();
        unsigned long long alive_mask = 0ull, ok_mask = 0ull;   // bit i: step t0 + i (world: uniform; agent: this lane's)
        const unsigned long long ut_mask = __builtin_amdgcn_ballot_w64(lane < seg && sUT[lane] != 0);   // steps that take the table

        for (int ts = 0; ts < seg; ++ts) {
            float2* const pc = planes + cur * C;
            float2* const pn = planes + (1 - cur) * C;
            // Everything this step reads from LDS before the physics is issued HERE, in one round trip: the step's
            // constants, the agent's action byte and the five cells an agent can see or reach.
            const PhysF32 P = sP32[ts];
            // ---- policy (ref agents/greedy.py:14-36 or the host-drawn table) + update_agents (ref :181-244) ----
            if (with_agents) {
                // Straight-line, select-based code for every lane (lanes that hold no agent work on agent 0's position and
                // are masked at the end): a lone wave per SIMD pays ~20 cycles for every taken branch, and the branchy
                // form of this block was half of the step's instructions.
                const bool from_table = policy_mode == kPolicyTable || ((ut_mask >> ts) & 1ull);           // wave-uniform
                const int tab = (int)sTab[ts * N + alane];       // 0..8, or -1 / -2: (anti-)greedy choice (unused unless from_table)
                const EwReach R = ew_reach(pc, ar, ac, H, W);
                const bool greedy = from_table ? tab < 0 : policy_mode != kPolicyZeros;
                const bool argmin = (from_table && tab < 0) ? tab == -2 : policy_mode == kPolicyArgmin;
                // first maximum / minimum over the (masked) candidates, as np.argmax / np.argmin
                const double v0 = ((obs_mask >> 3) & 1) ? R.nat0 : 0.0, v1 = ((obs_mask >> 1) & 1) ? R.nat1 : 0.0;
                const double v2 = ((obs_mask >> 7) & 1) ? R.nat2 : 0.0, v3 = ((obs_mask >> 5) & 1) ? R.nat3 : 0.0;
                // (np.argmin = the first maximum of the negated values: one sign flip per candidate - covers are never
                // NaN, and -0.0 compares equal to 0.0 - instead of both comparisons and a select at every level)
                const unsigned long long flip = argmin ? 0x8000000000000000ull : 0ull;
                const double w0 = __longlong_as_double(__double_as_longlong(v0) ^ flip), w1 = __longlong_as_double(__double_as_longlong(v1) ^ flip);
                const double w2 = __longlong_as_double(__double_as_longlong(v2) ^ flip), w3 = __longlong_as_double(__double_as_longlong(v3) ^ flip);
                int best = 0;
                double bestv = w0;
                { const bool bt = w1 > bestv; best = bt ? 1 : best; bestv = bt ? w1 : bestv; }
                { const bool bt = w2 > bestv; best = bt ? 2 : best; bestv = bt ? w2 : bestv; }
                { const bool bt = w3 > bestv; best = bt ? 3 : best; }
                const int a = greedy ? 4 + best : (from_table ? tab : 0);
                if (t0 + ts == K - 1 && is_agent && io.action) io.action[(size_t)b * N + lane] = a;
                ew_update_agents(a, R, is_agent, lane, N, W, agent_gamma, ast, ar, ac, pc);
            }
            // ---- forward (ref :434-461) ----
            const bool last = t0 + ts == K - 1;
            unsigned int nfix = 0;
            const bool alive_here = ew_forward<EXACT>(P, pc, pn, G, C, lane, thr_f, [&]() {
                const EpisodeWaveArgs* cold = &kernarg_struct<EpisodeWaveArgs>();
                asm volatile("" : "+s"(cold));                   // (keeps the 17 scalar loads inside the cold block)
                PhysF64 Q = cold->P64;
                Q.L = sLs[ts];
                return Q;
            }, nfix);
            if (last) last_fix = nfix;
            // ---- per-step flags of the lifespan harness (nb greedy cell 2:46-52) ----
            if (__builtin_amdgcn_ballot_w64(alive_here) != 0ull) alive_mask |= 1ull << ts;
            if (is_agent) {
                const double rw = ast * (ast > 0.0 ? 1.0 : 0.0);
                if (!(rw < 0.1)) ok_mask |= 1ull << ts;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // the new plane is complete before anyone reads it
            __builtin_amdgcn_wave_barrier();
            cur = 1 - cur;
        }
        // ---- the segment's flags ----
        if (valid) {
            for (int i = lane; i < seg; i += 64) io.world_alive[(size_t)(t0 + i) * B + b] = (unsigned char)((alive_mask >> i) & 1ull);
            if (N > 0) {
                if (is_agent) sOk[lane] = ok_mask;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < seg * N; i += 64) {
                    const int tt = i / N, n = i - tt * N;
                    io.agent_ok[((size_t)(t0 + tt) * B + b) * N + n] = (unsigned char)((sOk[n] >> tt) & 1ull);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }

    // ---- back to global memory: planes, the state before the last step (after its grazing), agents, reductions ----
    if (valid) {
        const float2* const pc = planes + cur * C;
        const float2* const pp = planes + (1 - cur) * C;
        float m = 0.f, sl = 0.f, sd = 0.f;
#pragma unroll
        for (int j = 0; j < kEwSlots; ++j)
            if (G.own[j]) {
                const int c = lane + 64 * j;
                const float2 v = pc[c], w = pp[c];
                io.L[(size_t)b * C + c] = (plane_t)v.x;
                io.D[(size_t)b * C + c] = (plane_t)v.y;
                io.prevL[(size_t)b * C + c] = (plane_t)w.x;
                io.prevD[(size_t)b * C + c] = (plane_t)w.y;
                m = fmaxf(m, fmaxf(v.x, v.y));
                sl += v.x;
                sd += v.y;
            }
        if (is_agent) {
            io.st[(size_t)b * N + lane] = ast;
            io.idx[((size_t)b * N + lane) * 2] = ar;
            io.idx[((size_t)b * N + lane) * 2 + 1] = ac;
        }
        m = wave_max(m);
        sl = wave_sum(sl);                                       // integers < 2^24: exact in any order
        sd = wave_sum(sd);
        const unsigned int nf = (unsigned int)wave_sum((float)last_fix);
        if (lane == 0) {                                        // the world's whole record is ASSIGNED: no memset before the launch
            if (b == 0) io.stats[B] = StatsDev{0u, 0u, 0ull, 0ull};   // ... and the counter record behind the worlds'
            io.stats[b].max_k = (unsigned int)m;
            io.stats[b].reserved = EXACT ? nf : 0u;              // float64 re-evaluations of the last step (dw_last_fixup_count sums them)
            io.stats[b].sum_l = (unsigned long long)sl;
            io.stats[b].sum_d = (unsigned long long)sd;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// episode_mlp_wave — episode_mlp's contract (EpisodeMlpIO: K environment steps with MLP policies, the step loop of the ES
// trainers' fitness evaluation, ref SimpleGaussianES.get_fitness daisy/evo/sges.py:144-181 and :314-349) for H*W <= 256
// and 16 N <= 64 - the reference's own configuration, 16x16 worlds with 4 agents - on the one-wave-per-world design:
//   observe   ref get_obs :246-263: lane i < 9 N derives patch cell (agent i / 9, cell i % 9) in float64 from the
//             pre-step planes exactly as the `observe` kernel does; the agents' positions / states are mirrored in LDS
//   policy    ref MLP.get_action (agents/mlp.py:97-116): sixteen lanes per agent - a 64-lane wave holds the four
//             agents' networks side by side - layer after layer with nothing but the wave's own LDS order between
//             them (episode_mlp: three workgroup barriers per layer pass); every dot product accumulated
//             sequentially in index order (the arithmetic of `policy_mlp`)
//   update_agents, reward / done of the step (ref step :486-492), forward: the pieces of episode_wave.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr size_t episode_mlp_wave_world_bytes(int C, int N) {
    // planes [2][C] float pairs | per agent 128 doubles (x 64 | h1 16 | h2 32 | logits 16) | agent states | positions
    return (size_t)16 * C + (size_t)N * 128 * 8 + ((size_t)8 * N + 15) / 16 * 16 + ((size_t)8 * N + 15) / 16 * 16;
}

struct EpisodeMlpWaveArgs {
    EpisodeMlpIO io;
    int B, N, H, W, K, obs_mask, split;
    double agent_gamma, L_prev0;
    PhysF64 P64;
};

template <bool EXACT>
__global__ __launch_bounds__(256) void episode_mlp_wave(EpisodeMlpWaveArgs A) {
    const EpisodeMlpIO& io = A.io;
    const int B = A.B, N = A.N, H = A.H, W = A.W, K = A.K, obs_mask = A.obs_mask, split = A.split;
    const double agent_gamma = A.agent_gamma;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = H * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x * 4 + wv;
    const bool valid = b < B;
    PhysF32* const sP32 = reinterpret_cast<PhysF32*>(smem);
    double* const sLs = reinterpret_cast<double*>(smem + (size_t)kEwSeg * sizeof(PhysF32));
    unsigned char* const wbase = smem + episode_wave_shared_bytes() + (size_t)wv * episode_mlp_wave_world_bytes(C, N);
    float2* const planes = reinterpret_cast<float2*>(wbase);    // [2][C]: current | previous (they swap every step)
    double* const mlp = reinterpret_cast<double*>(wbase + (size_t)16 * C);              // [N][128]
    double* const sSt = mlp + (size_t)N * 128;                                           // [N] agent states
    int* const sIdx = reinterpret_cast<int*>(wbase + (size_t)16 * C + (size_t)N * 1024 + ((size_t)8 * N + 15) / 16 * 16);   // [N][2]

    EwCells G;
    ew_cells_init(G, lane, C, H, W, valid);
    int cur = 0;
    if (valid) {
#pragma unroll
        for (int j = 0; j < kEwSlots; ++j)
            if (G.own[j]) {
                const int c = lane + 64 * j;
                planes[c] = make_float2((float)io.L[(size_t)b * C + c], (float)io.D[(size_t)b * C + c]);
                planes[C + c] = make_float2((float)io.prevL[(size_t)b * C + c], (float)io.prevD[(size_t)b * C + c]);
            }
    }
    const bool is_agent = valid && lane < N;
    double ast = 0.0;
    int ar = 0, ac = 0;
    if (is_agent) {
        ast = io.st[(size_t)b * N + lane];
        ar = io.idx[((size_t)b * N + lane) * 2];
        ac = io.idx[((size_t)b * N + lane) * 2 + 1];
        sSt[lane] = ast; sIdx[2 * lane] = ar; sIdx[2 * lane + 1] = ac;
    }
    const int bc = valid ? b : 0;
    const double* const Wa = io.weights + (io.member_a ? (size_t)io.member_a[bc] * 1808 : 0);
    const double* const Wb = io.weights + (io.member_b ? (size_t)io.member_b[bc] * 1808 : 0);
    // sixteen lanes per agent: lane = 16 * agent + j
    const int gj = lane & 15, gn = lane >> 4;
    const bool mlp_on = valid && gn < N;
    const double* const Wn = (split >= 0 && gn >= split) ? Wb : Wa;
    const double* const W1 = Wn;                 // [63][16]
    const double* const W2 = Wn + 63 * 16;       // [16][32]
    const double* const W3 = W2 + 16 * 32;       // [32][9]
    double* const xg = mlp + (size_t)(mlp_on ? gn : 0) * 128;
    // observation lane i < 9 N: patch cell k of agent n
    const int on_ = lane / 9, ok_ = lane - on_ * 9;
    const bool obs_on = valid && lane < 9 * N;
    double L_prev = A.L_prev0;
    unsigned int last_fix = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (int t0 = 0; t0 < K; t0 += kEwSeg) {
        const int seg = min(kEwSeg, K - t0);
        __syncthreads();
        for (int i = tid; i < seg * (int)(sizeof(PhysF32) / 4); i += 256)
            reinterpret_cast<unsigned int*>(sP32)[i] = reinterpret_cast<const unsigned int*>(io.P32 + t0)[i];
        for (int i = tid; i < seg; i += 256) sLs[i] = io.Ls[t0 + i];
        __syncthreads();

        for (int ts = 0; ts < seg; ++ts) {
            float2* const pc = planes + cur * C;                 // current state
            float2* const pp = planes + (1 - cur) * C;           // the state before the last step taken (becomes the new one)
            const PhysF32 P = sP32[ts];
            const EwReach R = ew_reach(pc, ar, ac, H, W);        // (pre-graze covers: what update_agents needs below)
            // ---- observe (ref get_obs :246-263; the `observe<.., POST = true>` kernel's arithmetic) ----
            if (obs_on) {
                double* x = mlp + (size_t)on_ * 128;
                const int k = ok_;
                if (!((obs_mask >> k) & 1)) {
#pragma unroll
                    for (int ch = 0; ch < 7; ++ch) x[ch * 9 + k] = 0.0;
                } else {
                    PhysF64 Qp = A.P64;
                    Qp.L = L_prev;
                    const int pr = sIdx[2 * on_], pcn = sIdx[2 * on_ + 1];
                    int r = pr + (k / 3 - 1), c = pcn + (k % 3 - 1);
                    r = r < 0 ? r + H : (r >= H ? r - H : r);
                    c = c < 0 ? c + W : (c >= W ? c - W : c);
                    const int ru = r == 0 ? H - 1 : r - 1, rd = r == H - 1 ? 0 : r + 1;
                    const int cl = c == 0 ? W - 1 : c - 1, cr = c == W - 1 ? 0 : c + 1;
                    const int rows[3] = {ru, r, rd}, cols[3] = {cl, c, cr};
                    double l9[9], d9[9];
#pragma unroll
                    for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                        for (int e3 = 0; e3 < 3; ++e3) {
                            const float2 v = pp[rows[a3] * W + cols[e3]];
                            l9[a3 * 3 + e3] = dw_permille_to_natural((double)v.x);      // = (double)k / 1000.0 exactly
                            d9[a3 * 3 + e3] = dw_permille_to_natural((double)v.y);
                        }
                    const CellF64 o = cell_f64(Qp, l9, d9);
                    double v4 = dw_div1000(dw_round3_k(o.Tl));
                    for (int a2 = 0; a2 < N; ++a2)                // ref forward :454-459: agent states stamped, last wins
                        if (sIdx[2 * a2] == r && sIdx[2 * a2 + 1] == c) v4 = sSt[a2];
                    const float2 cc2 = pc[r * W + c];
                    x[0 * 9 + k] = dw_div1000(dw_round3_k(Qp.p - o.nl - o.nd));
                    x[1 * 9 + k] = dw_permille_to_natural((double)cc2.x);
                    x[2 * 9 + k] = dw_permille_to_natural((double)cc2.y);
                    x[3 * 9 + k] = dw_div1000(dw_round3_k(o.T));
                    x[4 * 9 + k] = v4;
                    x[5 * 9 + k] = dw_div1000(dw_round3_k(o.Td));
                    x[6 * 9 + k] = 0.0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- policy: 63 -> 16 -> 32 -> 9, ReLU, float64, sequential dot products ----
            int act = 0;
            {
                double h = 0.0;
                for (int i = 0; i < 63; ++i) h = __builtin_fma(xg[i], W1[i * 16 + gj], h);
                if (mlp_on) xg[64 + gj] = h * (h > 0.0 ? 1.0 : 0.0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double u = 0.0, v = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double hi = xg[64 + i];
                    u = __builtin_fma(hi, W2[i * 32 + gj], u);
                    v = __builtin_fma(hi, W2[i * 32 + gj + 16], v);
                }
                if (mlp_on) {
                    xg[80 + gj] = u * (u > 0.0 ? 1.0 : 0.0);
                    xg[80 + gj + 16] = v * (v > 0.0 ? 1.0 : 0.0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int gj9 = gj < 9 ? gj : 8;
                double o = 0.0;
#pragma unroll
                for (int i = 0; i < 32; ++i) o = __builtin_fma(xg[80 + i], W3[i * 9 + gj9], o);
                if (mlp_on && gj < 9) xg[112 + gj] = o;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // agent lane n: first maximum of its nine logits, as np.argmax
                const double* lg = mlp + (size_t)(is_agent ? lane : 0) * 128 + 112;
                int best = 0;
                double bestv = lg[0];
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    const double ov = lg[k];
                    const bool bt = ov > bestv;
                    best = bt ? k : best;
                    bestv = bt ? ov : bestv;
                }
                act = best;
            }
            // ---- update_agents; the step's reward / done (ref step :486-492) ----
            ew_update_agents(act, R, is_agent, lane, N, W, agent_gamma, ast, ar, ac, pc);
            if (t0 + ts == K - 1 && is_agent && io.action) io.action[(size_t)b * N + lane] = act;
            if (is_agent) {
                sSt[lane] = ast; sIdx[2 * lane] = ar; sIdx[2 * lane + 1] = ac;
                const double rw = ast * (ast > 0.0 ? 1.0 : 0.0);
                io.reward[((size_t)(t0 + ts) * B + b) * N + lane] = rw;
                io.done[((size_t)(t0 + ts) * B + b) * N + lane] = rw < 0.1 ? 1 : 0;
            }
            // ---- forward: the current planes advance into the buffer of the previous state ----
            unsigned int nfix = 0;
            const double Lt = sLs[ts];
            (void)ew_forward<EXACT>(P, pc, pp, G, C, lane, 0.f, [&]() { PhysF64 Q = A.P64; Q.L = Lt; return Q; }, nfix);
            if (t0 + ts == K - 1) last_fix = nfix;
            L_prev = Lt;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            cur = 1 - cur;
        }
    }

    if (valid) {
        const float2* const pc = planes + cur * C;
        const float2* const pp = planes + (1 - cur) * C;
        float m = 0.f, sl = 0.f, sd = 0.f;
#pragma unroll
        for (int j = 0; j < kEwSlots; ++j)
            if (G.own[j]) {
                const int c = lane + 64 * j;
                const float2 v = pc[c], w = pp[c];
                io.L[(size_t)b * C + c] = (plane_t)v.x;
                io.D[(size_t)b * C + c] = (plane_t)v.y;
                io.prevL[(size_t)b * C + c] = (plane_t)w.x;
                io.prevD[(size_t)b * C + c] = (plane_t)w.y;
                m = fmaxf(m, fmaxf(v.x, v.y));
                sl += v.x;
                sd += v.y;
            }
        if (is_agent) {
            io.st[(size_t)b * N + lane] = ast;
            io.idx[((size_t)b * N + lane) * 2] = ar;
            io.idx[((size_t)b * N + lane) * 2 + 1] = ac;
        }
        m = wave_max(m);
        sl = wave_sum(sl);
        sd = wave_sum(sd);
        const unsigned int nf = (unsigned int)wave_sum((float)last_fix);
        if (lane == 0) {
            if (b == 0) io.stats[B] = StatsDev{0u, 0u, 0ull, 0ull};
            io.stats[b].max_k = (unsigned int)m;
            io.stats[b].reserved = EXACT ? nf : 0u;
            io.stats[b].sum_l = (unsigned long long)sl;
            io.stats[b].sum_d = (unsigned long long)sd;
        }
    }
}

}  // namespace dw
