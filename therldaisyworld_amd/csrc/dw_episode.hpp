// dw_episode.hpp — episode_small: device-resident episode loop (policy, grazing, physics, flags for K
// steps in one launch) for worlds that fit LDS (H*W <= 4096).
#pragma once
#include "dw_common.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// episode_small — device-resident episode loop for small worlds (H*W <= 4096: the README sweep's
// 8x8 grids up to C1's 64x64).  SURVEY.md §8(f) row N1.
//
// A workgroup keeps `wpb` whole worlds (both planes, ping-pong) and their agents in LDS and runs
// K consecutive environment steps without leaving the chip: per step
//     policy (ref Greedy.__call__, agents/greedy.py:14-36, or host-drawn random actions)
//   -> update_agents (ref :181-244, one lane per world, agents in order)
//   -> forward (ref :434-461; toroidal 3x3 stencil straight from LDS; exact mode re-evaluates
//      near-tie cells in float64 on the spot)
//   -> per-world reductions + the per-step flags the notebook's lifespan harness counts
//      (world alive: max cover > 0.005; agent alive: reward >= 0.1; cell 2:46-52).
// Luminosity-dependent coefficients of all K steps are precomputed on the host (P32[t], Ls[t]).
// On exit the current planes, the pre-last-step planes (for observations / env.grid), the agents and
// the reductions go back to global memory, so the ordinary entry points continue from there.
// ---------------------------------------------------------------------------------------------
enum { kPolicyArgmax = 0, kPolicyArgmin = 1, kPolicyZeros = 2, kPolicyTable = 3, kPolicySkipAgents = 4 };

struct EpisodeIO {
    plane_t* L;                     // [B][C] current planes (in/out; binary16 in HBM, float32 in LDS)
    plane_t* D;
    plane_t* prevL;                 // [B][C] out: state before the last step (after its grazing)
    plane_t* prevD;
    int* idx;                       // [B][N][2] in/out
    double* st;                     // [B][N] in/out
    const PhysF32* P32;             // [K]
    const double* Ls;               // [K]
    const unsigned char* use_table; // [K]  1 = take this step's actions from `table` (epsilon branch)
    const signed char* table;       // [K][B][N] host-drawn action codes
    unsigned char* world_alive;     // [K][B] out
    unsigned char* agent_ok;        // [K][B][N] out
    StatsDev* stats;                // [B] out: reductions after the last step
    unsigned long long* fixups;     // out: float64 re-evaluations of the last step (summed)
    int* action;                    // [B][N] out: the LAST step's action codes (the handle's action buffer), or null
};

// float64 value of cell c of an LDS-resident world (exact mode's re-evaluation of a near-tie cell)
__device__ inline float ep_cell_f64(const PhysF64& Q, const float* curL, const float* curD, int H, int W, int c,
                                    float& kd) {
    const int r = c / W, cc = c - r * W;
    const int ru = (r == 0 ? H - 1 : r - 1) * W, rm = r * W, rd = (r == H - 1 ? 0 : r + 1) * W;
    const int cl = cc == 0 ? W - 1 : cc - 1, cr = cc == W - 1 ? 0 : cc + 1;
    const int rows[3] = {ru, rm, rd}, cols[3] = {cl, cc, cr};
    unsigned int wv[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 3; ++e)
            wv[a * 3 + e] = (unsigned)curL[rows[a] + cols[e]] | ((unsigned)curD[rows[a] + cols[e]] << 16);
    const NewCoverF64 o = cell_f64_lean(Q, wv);
    kd = (float)dw_round3_k(o.nd);
    return (float)dw_round3_k(o.nl);
}

// LDS bytes of one world: planes [2 buffers][2 species][C] floats | agent states | idx, act | 8 reduction
// words | the step's near-tie cell list (exact mode)
constexpr int kEpFixCap = 240;
__host__ __device__ constexpr size_t episode_world_bytes(int C, int N) {
    return ((size_t)16 * C + (size_t)N * 8 + (size_t)N * 12 + 32 + 2 * kEpFixCap + 15) / 16 * 16;
}

// update_agents (ref :181-244, collision_mode 0) of one LDS-resident world: called by ONE lane, agents in order
// (the first agent to land on a cell eats it all), float64 stores with exactly the reference's operations.
__device__ __forceinline__ void ep_update_agents(double* ast, int* aidx, const int* act, float* curL, float* curD, int N,
                                                 int H, int W, double agent_gamma) {
    for (int n = 0; n < N; ++n) ast[n] -= agent_gamma;
    for (int n = 0; n < N; ++n) {
        double s = ast[n];
        if (s > 0.0) {
            const int a = act[n];
            int r = aidx[2 * n], c = aidx[2 * n + 1];
            if (a != 8) {
                const int m = ((a % 4) + 4) % 4;
                if (m == 0) c -= 1; else if (m == 1) r -= 1; else if (m == 2) r += 1; else c += 1;
            }
            r = ((r % H) + H) % H;
            c = ((c % W) + W) % W;
            aidx[2 * n] = r;
            aidx[2 * n + 1] = c;
            if (a > 4) {
                const int o = r * W + c;
                s += dw_permille_to_natural((double)curL[o]) + dw_permille_to_natural((double)curD[o]);
                curL[o] = 0.f;
                curD[o] = 0.f;
                ast[n] = s;
            }
        }
    }
    for (int n = 0; n < N; ++n) {
        const double s = ast[n];
        ast[n] = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
    }
}

// One physics pass of an LDS-resident world (cur -> nxt; the caller swaps): 3x3 toroidal stencil straight from
// LDS, exact mode with its near-tie list re-evaluated in float64 after the cell loop by all threads of the world,
// per-world reductions into red[0..3].  Called by every thread of the workgroup (it synchronises).
template <bool EXACT>
__device__ __forceinline__ void ep_forward(const PhysF32& P, const PhysF64& Q, const float* curL, const float* curD,
                                           float* nxtL, float* nxtD, int H, int W, int C, int lt, int tpw, int tid,
                                           bool valid, unsigned int* red, unsigned short* fixlist) {
    float tmax = 0.f, tsl = 0.f, tsd = 0.f;
    unsigned int nfix = 0;
    if (valid) {
        for (int c = lt; c < C; c += tpw) {
            const int r = c / W, cc = c - r * W;
            const int ru = (r == 0 ? H - 1 : r - 1) * W, rm = r * W, rd = (r == H - 1 ? 0 : r + 1) * W;
            const int cl = cc == 0 ? W - 1 : cc - 1, cr = cc == W - 1 ? 0 : cc + 1;
            const float li = curL[rm + cc], di = curD[rm + cc];
            const float El = (curL[ru + cc] + curL[rd + cc]) + (curL[rm + cl] + curL[rm + cr]);
            const float Cl = (curL[ru + cl] + curL[rd + cl]) + (curL[ru + cr] + curL[rd + cr]);
            const float Ed = (curD[ru + cc] + curD[rd + cc]) + (curD[rm + cl] + curD[rm + cr]);
            const float Cd = (curD[ru + cl] + curD[rd + cl]) + (curD[ru + cr] + curD[rd + cr]);
            const GrowthF32 g = growth_f32<EXACT || kFastSplit>(P, li, di, El, Cl, Ed, Cd);
            float kl, kd;
            if (EXACT) {
                bool tl, td;
                kl = finish_exact(P, li, g.gql, g.dKl, g.oml, tl);
                kd = finish_exact(P, di, g.gqd, g.dKd, g.omd, td);
                if (tl || td) {
                    // near a tie: re-evaluated in float64 AFTER the loop, by all threads of the world at
                    // once (inline, every such cell would cost its whole wave a float64 evaluation)
                    const unsigned int slot = atomicAdd(&red[4], 1u);
                    if (slot < (unsigned)kEpFixCap) {
                        fixlist[slot] = (unsigned short)c;
                        nxtL[c] = kl;                       // provisional; kept out of the reductions
                        nxtD[c] = kd;
                        continue;
                    }
                    kl = ep_cell_f64(Q, curL, curD, H, W, c, kd);      // list full: on the spot
                    ++nfix;
                }
            } else {
                kl = finish_fast(li, g.dKl, g.fl);
                kd = finish_fast(di, g.dKd, g.fd);
            }
            nxtL[c] = kl;
            nxtD[c] = kd;
            tmax = fmaxf(tmax, fmaxf(kl, kd));
            tsl += kl;
            tsd += kd;
        }
    }
    if (EXACT) {
        __syncthreads();                                     // the list is complete
        if (valid) {
            const unsigned int n = min(red[4], (unsigned)kEpFixCap);
            for (unsigned int e = lt; e < n; e += tpw) {
                const int c = fixlist[e];
                float kd;
                const float kl = ep_cell_f64(Q, curL, curD, H, W, c, kd);
                nxtL[c] = kl;
                nxtD[c] = kd;
                tmax = fmaxf(tmax, fmaxf(kl, kd));
                tsl += kl;
                tsd += kd;
                ++nfix;
            }
        }
        __syncthreads();
        if (valid && lt == 0) red[4] = 0;
    }
    // per-world reductions: wavefront shuffles when a wave belongs to one world, LDS atomics across waves
    {
        const float m = tpw >= 64 ? wave_max(tmax) : tmax;
        const float sl = tpw >= 64 ? wave_sum(tsl) : tsl;
        const float sd = tpw >= 64 ? wave_sum(tsd) : tsd;
        if (valid && (tpw < 64 || (tid & 63) == 0)) {
            atomicMax(&red[0], (unsigned int)m);
            atomicAdd(&red[1], (unsigned int)sl);
            atomicAdd(&red[2], (unsigned int)sd);
        }
        if (EXACT && valid && nfix) atomicAdd(&red[3], nfix);
    }
    __syncthreads();
}

template <bool EXACT>
__global__ __launch_bounds__(256) void episode_small(EpisodeIO io, int B, int N, int H, int W, int wpb, int K,
                                                     int policy_mode, int obs_mask, double agent_gamma,
                                                     unsigned int thr, PhysF64 P64) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = H * W;
    const int tpw = 256 / wpb;                                   // threads per world
    const int tid = threadIdx.x, w = tid / tpw, lt = tid - w * tpw;
    const int b = blockIdx.x * wpb + w;
    const bool valid = b < B;
    // LDS carve-up per world: planes [2 buffers][2 species][C] floats | agent state doubles | idx | act | red | list
    const size_t world_bytes = episode_world_bytes(C, N);
    unsigned char* base = smem + (size_t)w * world_bytes;
    float* planes = reinterpret_cast<float*>(base);
    double* ast = reinterpret_cast<double*>(base + (size_t)16 * C);
    int* aidx = reinterpret_cast<int*>(base + (size_t)16 * C + (size_t)N * 8);
    int* act = aidx + 2 * N;
    unsigned int* red = reinterpret_cast<unsigned int*>(act + N);   // max, sum_l, sum_d, fixups, [4] = list length
    unsigned short* fixlist = reinterpret_cast<unsigned short*>(red + 8);   // cells to re-evaluate in float64
    float* curL = planes;
    float* curD = planes + C;
    float* nxtL = planes + 2 * C;
    float* nxtD = planes + 3 * C;

    if (valid) {
        for (int c = lt; c < C; c += tpw) {
            curL[c] = (float)io.L[(size_t)b * C + c];
            curD[c] = (float)io.D[(size_t)b * C + c];
        }
        for (int n = lt; n < N; n += tpw) {
            ast[n] = io.st[(size_t)b * N + n];
            aidx[2 * n] = io.idx[((size_t)b * N + n) * 2];
            aidx[2 * n + 1] = io.idx[((size_t)b * N + n) * 2 + 1];
        }
        if (lt < 8) red[lt] = 0;
    }
    __syncthreads();

    for (int t = 0; t < K; ++t) {
        // ---- policy: action of each agent for this step, from the state it observes ----
        if (valid && policy_mode != kPolicySkipAgents) {
            for (int n = lt; n < N; n += tpw) {
                int a = 0;
                bool greedy = false, argmin = policy_mode == kPolicyArgmin;
                if (policy_mode == kPolicyTable || (policy_mode != kPolicyZeros && io.use_table[t])) {
                    a = io.table[((size_t)t * B + b) * N + n];          // 0..8, or -1 / -2: (anti-)greedy choice
                    greedy = a < 0;
                    if (greedy) argmin = a == -2;
                } else if (policy_mode != kPolicyZeros) {
                    greedy = true;
                }
                if (greedy) {
                    const int ar = aidx[2 * n], ac = aidx[2 * n + 1];
                    const int cand[4] = {3, 1, 7, 5};             // (r,c-1) (r-1,c) (r+1,c) (r,c+1)
                    int best = 0;
                    double bestv = 0.0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = cand[i];
                        double v = 0.0;
                        if ((obs_mask >> k) & 1) {
                            const int r = (ar + (k / 3 - 1) + H) % H, c = (ac + (k % 3 - 1) + W) % W;
                            v = dw_permille_to_natural((double)curL[r * W + c]) + dw_permille_to_natural((double)curD[r * W + c]);
                        }
                        if (i == 0 || (argmin ? v < bestv : v > bestv)) { best = i; bestv = v; }
                    }
                    a = 4 + best;
                }
                act[n] = a;
                if (t == K - 1 && io.action) io.action[(size_t)b * N + n] = a;
            }
        }
        __syncthreads();
        // ---- update_agents (ref :181-244): one lane per world, agents in order ----
        if (valid && lt == 0 && N > 0 && policy_mode != kPolicySkipAgents)
            ep_update_agents(ast, aidx, act, curL, curD, N, H, W, agent_gamma);
        __syncthreads();
        // ---- forward ----
        const PhysF32 P = io.P32[t];
        PhysF64 Q = P64;
        Q.L = io.Ls[t];
        ep_forward<EXACT>(P, Q, curL, curD, nxtL, nxtD, H, W, C, lt, tpw, tid, valid, red, fixlist);
        { float* x = curL; curL = nxtL; nxtL = x; x = curD; curD = nxtD; nxtD = x; }
        // ---- per-step flags of the lifespan harness; final reductions ----
        if (valid) {
            if (lt == 0) io.world_alive[(size_t)t * B + b] = red[0] > thr ? 1 : 0;
            for (int n = lt; n < N; n += tpw) {
                const double s = ast[n];
                const double rw = s * (s > 0.0 ? 1.0 : 0.0);
                io.agent_ok[((size_t)t * B + b) * N + n] = rw < 0.1 ? 0 : 1;
            }
            if (t == K - 1 && lt == 0) {
                io.stats[b].max_k = red[0];
                io.stats[b].sum_l = red[1];
                io.stats[b].sum_d = red[2];
                if (EXACT && red[3]) atomicAdd(io.fixups, (unsigned long long)red[3]);
            }
        }
        __syncthreads();
        if (valid && lt < 4) red[lt] = 0;
        __syncthreads();
    }

    if (valid) {
        for (int c = lt; c < C; c += tpw) {
            io.L[(size_t)b * C + c] = (plane_t)curL[c];
            io.D[(size_t)b * C + c] = (plane_t)curD[c];
            io.prevL[(size_t)b * C + c] = (plane_t)nxtL[c];
            io.prevD[(size_t)b * C + c] = (plane_t)nxtD[c];
        }
        for (int n = lt; n < N; n += tpw) {
            io.st[(size_t)b * N + n] = ast[n];
            io.idx[((size_t)b * N + n) * 2] = aidx[2 * n];
            io.idx[((size_t)b * N + n) * 2 + 1] = aidx[2 * n + 1];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// episode_mlp — K environment steps with MLP policies without leaving the chip (H*W <= 4096): the step loop of
// the ES trainers' fitness evaluation (ref SimpleGaussianES.get_fitness, daisy/evo/sges.py:144-181, and the
// population loop :314-349).  Per step, for the 1-4 worlds a workgroup keeps in LDS:
//   observe     ref get_obs :246-263 - the 7-channel 3x3 patch of every agent, channel values re-derived in float64
//               from the pre-step planes exactly as the `observe` kernel does (one lane per patch cell)
//   policy      ref MLP.get_action (agents/mlp.py:97-116) - 63 -> 16 -> 32 -> 9, ReLU, float64, sixteen lanes per
//               agent, every dot product accumulated sequentially in index order (the arithmetic of `policy_mlp`);
//               agents [0, split) use parameter set member_a[world], agents [split, N) member_b[world]
//   update_agents, reward / done of the step (ref step :486-492), forward.
// One launch per chunk instead of four launches per step (the per-step loop was bound by the host thread that
// issued them: 118 us issued against 44 us of kernels per step for 64 x 32 worlds of 16x16).
// ---------------------------------------------------------------------------------------------
struct EpisodeMlpIO {
    plane_t* L; plane_t* D;         // [B][C] current planes (in/out)
    plane_t* prevL; plane_t* prevD; // [B][C] in: the state before the last step taken; out: before the last step here
    int* idx; double* st;           // agents in/out
    const PhysF32* P32;             // [K]
    const double* Ls;               // [K]
    const double* weights;          // [n_members][1808]
    const int* member_a;            // [B] or null (one parameter set)
    const int* member_b;            // [B] or null
    double* reward;                 // [K][B][N] out
    unsigned char* done;            // [K][B][N] out
    StatsDev* stats;                // [B] out: reductions after the last step
    unsigned long long* fixups;
    int* action;                    // [B][N] out: the LAST step's action codes (the handle's action buffer), or null
};

__host__ __device__ constexpr size_t episode_mlp_world_bytes(int C, int N) {
    // planes | per agent: state, idx, act, observation (64), hidden 1 (16), hidden 2 (32), logits (16) doubles | red | list
    return ((size_t)16 * C + (size_t)N * (8 + 8 + 4 + 8 * 128) + 32 + 2 * kEpFixCap + 15) / 16 * 16;
}

template <bool EXACT>
__global__ __launch_bounds__(256) void episode_mlp(EpisodeMlpIO io, int B, int N, int H, int W, int wpb, int K,
                                                   int obs_mask, double agent_gamma, PhysF64 P64, double L_prev0,
                                                   int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = H * W;
    const int tpw = 256 / wpb;                                   // threads per world
    const int tid = threadIdx.x, w = tid / tpw, lt = tid - w * tpw;
    const int b = blockIdx.x * wpb + w;
    const bool valid = b < B;
    const size_t world_bytes = episode_mlp_world_bytes(C, N);
    unsigned char* base = smem + (size_t)w * world_bytes;
    float* planes = reinterpret_cast<float*>(base);
    double* ast = reinterpret_cast<double*>(base + (size_t)16 * C);
    double* mlp = ast + N;                                       // [N][128]: x 64 | h1 16 | h2 32 | logits 16
    int* aidx = reinterpret_cast<int*>(mlp + (size_t)N * 128);
    int* act = aidx + 2 * N;
    unsigned int* red = reinterpret_cast<unsigned int*>(act + N);
    unsigned short* fixlist = reinterpret_cast<unsigned short*>(red + 8);
    float* curL = planes;
    float* curD = planes + C;
    float* nxtL = planes + 2 * C;                                // holds the PREVIOUS state between the steps
    float* nxtD = planes + 3 * C;

    if (valid) {
        for (int c = lt; c < C; c += tpw) {
            curL[c] = (float)io.L[(size_t)b * C + c];
            curD[c] = (float)io.D[(size_t)b * C + c];
            nxtL[c] = (float)io.prevL[(size_t)b * C + c];
            nxtD[c] = (float)io.prevD[(size_t)b * C + c];
        }
        for (int n = lt; n < N; n += tpw) {
            ast[n] = io.st[(size_t)b * N + n];
            aidx[2 * n] = io.idx[((size_t)b * N + n) * 2];
            aidx[2 * n + 1] = io.idx[((size_t)b * N + n) * 2 + 1];
        }
        if (lt < 8) red[lt] = 0;
    }
    __syncthreads();
    const int bc = valid ? b : 0;
    const double* Wa = io.weights + (io.member_a ? (size_t)io.member_a[bc] * 1808 : 0);
    const double* Wb = io.weights + (io.member_b ? (size_t)io.member_b[bc] * 1808 : 0);
    double L_prev = L_prev0;

    for (int t = 0; t < K; ++t) {
        // ---- observe (ref get_obs :246-263; the `observe<.., POST = true>` kernel's arithmetic) ----
        if (valid) {
            PhysF64 Qp = P64;
            Qp.L = L_prev;
            for (int i = lt; i < N * 9; i += tpw) {
                const int n = i / 9, k = i - n * 9;
                double* x = mlp + (size_t)n * 128;
                if (!((obs_mask >> k) & 1)) {
#pragma unroll
                    for (int ch = 0; ch < 7; ++ch) x[ch * 9 + k] = 0.0;
                    continue;
                }
                const int r = (aidx[2 * n] + (k / 3 - 1) + H) % H, c = (aidx[2 * n + 1] + (k % 3 - 1) + W) % W;
                const int ru = r == 0 ? H - 1 : r - 1, rd = r == H - 1 ? 0 : r + 1;
                const int cl = c == 0 ? W - 1 : c - 1, cr = c == W - 1 ? 0 : c + 1;
                const int rows[3] = {ru, r, rd}, cols[3] = {cl, c, cr};
                double l9[9], d9[9];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        l9[a * 3 + e] = dw_permille_to_natural((double)nxtL[rows[a] * W + cols[e]]);
                        d9[a * 3 + e] = dw_permille_to_natural((double)nxtD[rows[a] * W + cols[e]]);
                    }
                const CellF64 o = cell_f64(Qp, l9, d9);
                double v4 = dw_div1000(dw_round3_k(o.Tl));
                for (int a = 0; a < N; ++a)                       // ref forward :454-459: agent states stamped, last wins
                    if (aidx[2 * a] == r && aidx[2 * a + 1] == c) v4 = ast[a];
                x[0 * 9 + k] = dw_div1000(dw_round3_k(Qp.p - o.nl - o.nd));
                x[1 * 9 + k] = dw_permille_to_natural((double)curL[r * W + c]);
                x[2 * 9 + k] = dw_permille_to_natural((double)curD[r * W + c]);
                x[3 * 9 + k] = dw_div1000(dw_round3_k(o.T));
                x[4 * 9 + k] = v4;
                x[5 * 9 + k] = dw_div1000(dw_round3_k(o.Td));
                x[6 * 9 + k] = 0.0;
            }
        }
        __syncthreads();
        // ---- policy: sixteen lanes per agent, layer by layer (the workgroup synchronises between the layers) ----
        const int j = lt & 15, g0 = lt >> 4, ngrp = tpw >> 4;
        for (int pass = 0; pass * ngrp < N; ++pass) {              // uniform trip count for the whole workgroup
            const int n = pass * ngrp + g0;
            const bool on = valid && n < N;
            double* x = mlp + (size_t)(on ? n : 0) * 128;
            const double* Wn = (split >= 0 && n >= split) ? Wb : Wa;
            const double* W1 = Wn;                 // [63][16]
            const double* W2 = Wn + 63 * 16;       // [16][32]
            const double* W3 = W2 + 16 * 32;      // [32][9]
            if (on) {
                double h = 0.0;
                for (int i = 0; i < 63; ++i) h = __builtin_fma(x[i], W1[i * 16 + j], h);
                x[64 + j] = h * (h > 0.0 ? 1.0 : 0.0);
            }
            __syncthreads();
            if (on) {
                double u = 0.0, v = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double hi = x[64 + i];
                    u = __builtin_fma(hi, W2[i * 32 + j], u);
                    v = __builtin_fma(hi, W2[i * 32 + j + 16], v);
                }
                x[80 + j] = u * (u > 0.0 ? 1.0 : 0.0);
                x[80 + j + 16] = v * (v > 0.0 ? 1.0 : 0.0);
            }
            __syncthreads();
            if (on && j < 9) {
                double o = 0.0;
#pragma unroll
                for (int i = 0; i < 32; ++i) o = __builtin_fma(x[80 + i], W3[i * 9 + j], o);
                x[112 + j] = o;
            }
            __syncthreads();
            if (on && j == 0) {
                int best = 0;
                double bestv = x[112];
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    const double o = x[112 + k];
                    if (o > bestv) { best = k; bestv = o; }    // first maximum, as np.argmax
                }
                act[n] = best;
                if (t == K - 1 && io.action) io.action[(size_t)b * N + n] = best;
            }
        }
        __syncthreads();
        // ---- update_agents; the step's reward / done (ref step :486-492) ----
        if (valid && lt == 0) {
            ep_update_agents(ast, aidx, act, curL, curD, N, H, W, agent_gamma);
            for (int n = 0; n < N; ++n) {
                const double s = ast[n];
                const double rw = s * (s > 0.0 ? 1.0 : 0.0);
                io.reward[((size_t)t * B + b) * N + n] = rw;
                io.done[((size_t)t * B + b) * N + n] = rw < 0.1 ? 1 : 0;
            }
        }
        __syncthreads();
        // ---- forward ----
        const PhysF32 P = io.P32[t];
        PhysF64 Q = P64;
        Q.L = io.Ls[t];
        ep_forward<EXACT>(P, Q, curL, curD, nxtL, nxtD, H, W, C, lt, tpw, tid, valid, red, fixlist);
        { float* xx = curL; curL = nxtL; nxtL = xx; xx = curD; curD = nxtD; nxtD = xx; }
        L_prev = io.Ls[t];
        if (valid && t == K - 1 && lt == 0) {
            io.stats[b].max_k = red[0];
            io.stats[b].sum_l = red[1];
            io.stats[b].sum_d = red[2];
            if (EXACT && red[3]) atomicAdd(io.fixups, (unsigned long long)red[3]);
        }
        __syncthreads();
        if (valid && lt < 4) red[lt] = 0;
        __syncthreads();
    }

    if (valid) {
        for (int c = lt; c < C; c += tpw) {
            io.L[(size_t)b * C + c] = (plane_t)curL[c];
            io.D[(size_t)b * C + c] = (plane_t)curD[c];
            io.prevL[(size_t)b * C + c] = (plane_t)nxtL[c];
            io.prevD[(size_t)b * C + c] = (plane_t)nxtD[c];
        }
        for (int n = lt; n < N; n += tpw) {
            io.st[(size_t)b * N + n] = ast[n];
            io.idx[((size_t)b * N + n) * 2] = aidx[2 * n];
            io.idx[((size_t)b * N + n) * 2 + 1] = aidx[2 * n + 1];
        }
    }
}

}  // namespace dw
