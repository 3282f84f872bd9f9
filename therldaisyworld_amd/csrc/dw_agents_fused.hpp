// dw_agents_fused.hpp — agents between the two steps of a fused launch.
//
// With agents the reference's episode is  policy_t, graze_t, forward_t, policy_{t+1}, graze_{t+1},
// forward_{t+1}, ...: the agents act on the grid between ANY two physics passes, so step pairs cannot
// simply share an HBM round trip.  But N agents touch N cells.  The pair (t, t+1) is run as
//
//     policy_t, graze_t            (as always, on the state S0 in memory  ->  S0')
//     step_stream_fused2           S2 = F(F(S0'))  - the grazing of step t+1 ignored
//     agents_lookahead_patch       this kernel, one wave per world:
//         A  recompute S1 = F(S0') on the 5 cells each agent can see / reach (from S0', which the fused
//            launch left intact in the other buffer) and evaluate policy_{t+1} (ref Greedy.__call__,
//            agents/greedy.py:14-36, or the caller's table)
//         B  graze_{t+1} (ref update_agents :181-244) for the world's agents in order, on those S1 values
//            (a cell grazed by an earlier agent of the same step reads 0), recording the grazed cells
//         C  recompute S2 on the 3x3 blocks around the grazed cells from S1' (= S1 with the grazed cells
//            zeroed, each S1 value again from S0') and overwrite those <= 9 N cells of the output planes.
//
// Exact mode evaluates A and C in float64 (what the exact kernels' results equal by construction); the
// float32-only mode evaluates them with the same float32 algebra as the step kernels (the one-cell
// instantiation of growth_t: bit-identical to the packed form), so the pair equals two ordinary steps
// bit for bit in both modes.  Inputs are quantised states (integer per-mille values).
#pragma once
#include "dw_step_fused.hpp"

namespace dw {

constexpr int kLookaheadMaxAgents = 64;      // agents per world handled by one wave (lane n = agent n)

// float32 step value of grid cell (r, c) (any integers: wrapped onto the torus) of a quantised state, as
// the step kernels compute it, packed light | dark << 16
template <typename TI>
__device__ inline unsigned int fast1_word(const TI* __restrict__ pL, const TI* __restrict__ pD, int H, int W, int r,
                                          int c, const PhysF32& P) {
    const int rr = wrap_near2(r, H), cc = wrap_near2(c, W);    // (callers pass in-range coordinates +- 2: no division)
    const int ru = rr == 0 ? H - 1 : rr - 1, rd = rr == H - 1 ? 0 : rr + 1;
    const int cl = cc == 0 ? W - 1 : cc - 1, cr = cc == W - 1 ? 0 : cc + 1;
    auto at = [&](const TI* p, int row, int col) -> float { return (float)p[(size_t)row * W + col]; };
    // same association as cells4 (exact anyway: the inputs are integers)
    const float El = (at(pL, rr, cl) + at(pL, rr, cr)) + (at(pL, ru, cc) + at(pL, rd, cc));
    const float Cl = (at(pL, ru, cl) + at(pL, ru, cr)) + (at(pL, rd, cl) + at(pL, rd, cr));
    const float Ed = (at(pD, rr, cl) + at(pD, rr, cr)) + (at(pD, ru, cc) + at(pD, rd, cc));
    const float Cd = (at(pD, ru, cl) + at(pD, ru, cr)) + (at(pD, rd, cl) + at(pD, rd, cr));
    const float li = at(pL, rr, cc), di = at(pD, rr, cc);
    const GrowthF32 g = growth_t<kFastSplit, float>(P, li, di, El, Cl, Ed, Cd);
    return (unsigned int)finish_fast(li, g.dKl, g.fl) | ((unsigned int)finish_fast(di, g.dKd, g.fd) << 16);
}

// the same map on nine already-evaluated (light | dark << 16) words of a 3x3 block, row-major
__device__ inline unsigned int fast_word_from9(const unsigned int* w, const PhysF32& P) {
    float l[9], d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { l[i] = (float)(w[i] & 0xffffu); d[i] = (float)(w[i] >> 16); }
    const float El = (l[3] + l[5]) + (l[1] + l[7]);
    const float Cl = (l[0] + l[2]) + (l[6] + l[8]);
    const float Ed = (d[3] + d[5]) + (d[1] + d[7]);
    const float Cd = (d[0] + d[2]) + (d[6] + d[8]);
    const GrowthF32 g = growth_t<kFastSplit, float>(P, l[4], d[4], El, Cl, Ed, Cd);
    return (unsigned int)finish_fast(l[4], g.dKl, g.fl) | ((unsigned int)finish_fast(d[4], g.dKd, g.fd) << 16);
}

struct LookaheadArgs {
    const plane_t* inL; const plane_t* inD;      // S0': the state the fused launch read
    plane_t* outL; plane_t* outD;                // S2 written by the fused launch, patched here
    int* idx; double* st;                        // agents after step t, updated to step t+1
    const signed char* code;                     // [B][N] for step t+1: 0..8 action, -1 greedy argmax, -2 argmin
    unsigned char* agent_ok;                     // [B][N] reward >= 0.1 after step t+1, or null
    // step t+2's policy + update_agents too (phase E), when the chunk continues: the next pair then starts with its
    // fused launch (two launches per pair instead of four).  Null: the caller runs them as kernels of their own.
    const signed char* code_next;                // [B][N] for step t+2, or null
    unsigned char* agent_ok_next;                // [B][N] reward >= 0.1 after step t+2's update_agents
    int* action_out;                             // [B][N] the handle's action buffer: the action codes of the LAST
                                                 // update_agents this kernel applies (step t+2's if code_next, else t+1's),
                                                 // so that dw_download_actions after a chunk returns the last step's actions
    // per-step "biosphere alive" flags (max cover > thr/1000) of the two steps, or null; pstats[2b], [2b+1]:
    // the fused launch's exact step-1 maximum and its count of certain step-2 row groups above thr
    unsigned char* alive_t; unsigned char* alive_t1;
    unsigned int* pstats; unsigned int thr;      // (the patch kernel clears its world's two words after reading them)
    int B, N, H, W, mask;
    double agent_gamma;
    PhysF32 P1, P2;                              // float32 coefficient sets of steps t, t+1 (fast mode)
    PhysF64 P64; double La, Lb;                  // float64 constants and the two luminosities (exact mode)
};

template <bool EXACT>
__global__ __launch_bounds__(64) void agents_lookahead_patch(LookaheadArgs A) {
    using TI = plane_t;
    using TO = plane_t;
    __shared__ unsigned int s_s1[kLookaheadMaxAgents][5];      // S1 at centre, (r,c-1), (r-1,c), (r+1,c), (r,c+1)
    __shared__ unsigned int s_w[kLookaheadMaxAgents][25];      // S1' on the 5x5 block around a grazed cell, row-major
    __shared__ int s_act[kLookaheadMaxAgents];
    __shared__ int s_ar[kLookaheadMaxAgents], s_ac[kLookaheadMaxAgents];   // the world's agents: loaded once, written back once
    __shared__ double s_st[kLookaheadMaxAgents];
    __shared__ int s_gr[kLookaheadMaxAgents], s_gc[kLookaheadMaxAgents];
    __shared__ int s_ng;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = A.N, H = A.H, W = A.W;
    const size_t woff = (size_t)b * H * W;
    const TI* pL = A.inL + woff;
    const TI* pD = A.inD + woff;
    TO* const oL = A.outL;
    TO* const oD = A.outD;
    PhysF64 Pa = A.P64, Pb = A.P64;
    Pa.L = A.La;
    Pb.L = A.Lb;
    auto step1 = [&](int r, int c) -> unsigned int {
        return EXACT ? exact1_word(pL, pD, H, W, r, c, Pa) : fast1_word(pL, pD, H, W, r, c, A.P1);
    };

    // ---- A: what each agent sees after forward_t (one (agent, cell) evaluation per lane), and its action for
    //         step t+1 ----
    if (lane < N) {
        const int an = b * N + lane;
        s_ar[lane] = A.idx[(size_t)an * 2];
        s_ac[lane] = A.idx[(size_t)an * 2 + 1];
        s_st[lane] = A.st[an];
    }
    __syncthreads();
    {
        const int dr[5] = {0, 0, -1, 1, 0}, dc[5] = {0, -1, 0, 0, 1};      // centre, then Greedy's order 3,1,7,5
        for (int t = lane; t < N * 5; t += 64) {
            const int n = t / 5, i = t - n * 5;
            s_s1[n][i] = step1(s_ar[n] + dr[i], s_ac[n] + dc[i]);
        }
    }
    __syncthreads();
    if (lane < N) {
        const int an = b * N + lane;
        int a = (int)A.code[an];
        if (a < 0) {                                                       // ref Greedy.__call__ :18-30
            const bool argmin = a == -2;
            const int cand[4] = {3, 1, 7, 5};
            int best = 0;
            double bestv = 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned int w = s_s1[lane][i + 1];
                double val = 0.0;
                if ((A.mask >> cand[i]) & 1)
                    val = dw_permille_to_natural((double)(float)(w & 0xffffu)) + dw_permille_to_natural((double)(float)(w >> 16));
                if (i == 0 || (argmin ? val < bestv : val > bestv)) { best = i; bestv = val; }
            }
            a = 4 + best;
        }
        s_act[lane] = a;
        if (A.action_out && !A.code_next) A.action_out[an] = a;
    }
    if (lane == 0) s_ng = 0;
    __syncthreads();

    // ---- B: update_agents of step t+1 (ref :181-244), agents in order (one lane, on the LDS copies) ----
    if (lane == 0) {
        for (int n = 0; n < N; ++n) s_st[n] -= A.agent_gamma;                                  // ref :184
        int ng = 0;
        for (int n = 0; n < N; ++n) {
            double s = s_st[n];
            if (s > 0.0) {                                                                  // ref :189
                const int a = s_act[n];
                int r = s_ar[n], c = s_ac[n];
                int which = 0;                                                              // slot of s_s1 it lands on
                if (a != 8) {                                                               // ref :191-206
                    const int m = ((a % 4) + 4) % 4;
                    if (m == 0) { c -= 1; which = 1; } else if (m == 1) { r -= 1; which = 2; }
                    else if (m == 2) { r += 1; which = 3; } else { c += 1; which = 4; }
                }
                r = wrap_near(r, H);                                                        // ref :208 (in range +- 1)
                c = wrap_near(c, W);
                s_ar[n] = r;
                s_ac[n] = c;
                if (a > 4) {                                                                // ref :210-216
                    bool eaten = false;                       // an earlier agent of this step emptied the cell
                    for (int g = 0; g < ng; ++g) eaten = eaten || (s_gr[g] == r && s_gc[g] == c);
                    if (!eaten) {
                        const unsigned int w = s_s1[n][which];
                        s += dw_permille_to_natural((double)(float)(w & 0xffffu)) + dw_permille_to_natural((double)(float)(w >> 16));
                        s_gr[ng] = r; s_gc[ng] = c; ++ng;
                    }
                    s_st[n] = s;
                }
            }
        }
        s_ng = ng;
    }
    __syncthreads();
    if (lane < N) {                                                                          // ref :244, and the write-back
        const int an = b * N + lane;
        const double s = s_st[lane];
        const double cl = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
        s_st[lane] = cl;                                          // phase E starts from the clipped stores
        if (!A.code_next) {                                       // (phase E writes the agents back itself)
            A.st[an] = cl;
            A.idx[(size_t)an * 2] = s_ar[lane];
            A.idx[(size_t)an * 2 + 1] = s_ac[lane];
        }
        if (A.agent_ok) A.agent_ok[an] = (cl * (cl > 0.0 ? 1.0 : 0.0)) < 0.1 ? 0 : 1;
    }

    // ---- C: S2 on the 3x3 blocks around the grazed cells, from S1' ----
    // two parallel stages: the 5x5 block of S1' values around every grazed cell (one evaluation per lane and pass,
    // each from S0' in memory), then the 9 S2 values of its inner 3x3 block from those (LDS) - instead of nine
    // S1 evaluations one after the other in every S2 lane (the kernel is pure latency: one wave per world)
    const int ng = s_ng;
    for (int p = lane; p < ng * 25; p += 64) {
        const int g = p / 25, t = p - g * 25;
        const int yr = wrap_near2(s_gr[g] + t / 5 - 2, H), yc = wrap_near2(s_gc[g] + t % 5 - 2, W);
        bool grazed = false;
        for (int k = 0; k < ng; ++k) grazed = grazed || (s_gr[k] == yr && s_gc[k] == yc);
        s_w[g][t] = grazed ? 0u : step1(yr, yc);
    }
    __syncthreads();
    for (int p = lane; p < ng * 9; p += 64) {
        const int g = p / 9, t = p - g * 9;
        const int tr = t / 3, tc = t - tr * 3;                   // the S2 cell is block cell (tr + 1, tc + 1) of the 5x5
        const int xr = wrap_near(s_gr[g] + tr - 1, H), xc = wrap_near(s_gc[g] + tc - 1, W);
        unsigned int w2[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 3; ++e) w2[a * 3 + e] = s_w[g][(tr + a) * 5 + (tc + e)];
        unsigned int out;
        if (EXACT) {
            const NewCoverF64 o = cell_f64_lean(Pb, w2);
            out = (unsigned int)dw_round3_k(o.nl) | ((unsigned int)dw_round3_k(o.nd) << 16);
        } else {
            out = fast_word_from9(w2, A.P2);
        }
        const size_t off = woff + (size_t)xr * W + xc;
        oL[off] = (TO)(float)(out & 0xffffu);
        oD[off] = (TO)(float)(out >> 16);
    }

    // ---- D: the two steps' "biosphere alive" flags ----
    if (A.alive_t) {
        const unsigned int max1 = A.pstats[2 * b], sure2 = A.pstats[2 * b + 1];
        __syncthreads();                                          // every lane has read them:
        if (lane == 0) { A.pstats[2 * b] = 0u; A.pstats[2 * b + 1] = 0u; }   // cleared for the next pair's fused launch
        if (lane == 0) A.alive_t[b] = max1 > A.thr ? 1 : 0;                                  // max of S1
        // step t+1: the fused launch counted row groups that certainly hold a value > thr; the patches
        // above touched at most 9 * ng cells, i.e. at most 9 * ng groups.  More groups than that: alive,
        // without looking.  Otherwise (a dying world) look at every cell of the patched result.
        if (sure2 > 9u * (unsigned int)ng) {
            if (lane == 0) A.alive_t1[b] = 1;
        } else {
            __threadfence_block();
            __syncthreads();                                     // the patch stores of this block
            float m = 0.f;
            for (int i = lane; i < H * W; i += 64) m = fmaxf(m, fmaxf((float)oL[woff + i], (float)oD[woff + i]));
            m = wave_max(m);
            if (lane == 0) A.alive_t1[b] = m > (float)A.thr ? 1 : 0;
        }
    }

    // ---- E: step t+2's policy and update_agents on the patched S2 (ref Greedy :18-30, update_agents :181-244) ----
    if (A.code_next) {
        __threadfence_block();
        __syncthreads();                                          // this block's patch stores before its loads of S2
        {
            const int dr[5] = {0, 0, -1, 1, 0}, dc[5] = {0, -1, 0, 0, 1};
            for (int t = lane; t < N * 5; t += 64) {
                const int n = t / 5, i = t - n * 5;
                const int r = wrap_near(s_ar[n] + dr[i], H), c = wrap_near(s_ac[n] + dc[i], W);
                const size_t off = woff + (size_t)r * W + c;
                // (volatile: past this CU's vector cache, which may hold the line from before the patch)
                const TO l = __builtin_bit_cast(TO, *reinterpret_cast<const volatile unsigned short*>(oL + off));
                const TO d = __builtin_bit_cast(TO, *reinterpret_cast<const volatile unsigned short*>(oD + off));
                s_s1[n][i] = (unsigned int)(float)l | ((unsigned int)(float)d << 16);
            }
        }
        __syncthreads();
        if (lane < N) {
            const int an = b * N + lane;
            int a = (int)A.code_next[an];
            if (a < 0) {
                const bool argmin = a == -2;
                const int cand[4] = {3, 1, 7, 5};
                int best = 0;
                double bestv = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned int w = s_s1[lane][i + 1];
                    double val = 0.0;
                    if ((A.mask >> cand[i]) & 1)
                        val = dw_permille_to_natural((double)(float)(w & 0xffffu)) + dw_permille_to_natural((double)(float)(w >> 16));
                    if (i == 0 || (argmin ? val < bestv : val > bestv)) { best = i; bestv = val; }
                }
                a = 4 + best;
            }
            s_act[lane] = a;
            if (A.action_out) A.action_out[an] = a;
        }
        __syncthreads();
        if (lane == 0) {
            for (int n = 0; n < N; ++n) s_st[n] -= A.agent_gamma;
            int ng2 = 0;
            for (int n = 0; n < N; ++n) {
                double s = s_st[n];
                if (s > 0.0) {
                    const int a = s_act[n];
                    int r = s_ar[n], c = s_ac[n];
                    int which = 0;
                    if (a != 8) {
                        const int m = ((a % 4) + 4) % 4;
                        if (m == 0) { c -= 1; which = 1; } else if (m == 1) { r -= 1; which = 2; }
                        else if (m == 2) { r += 1; which = 3; } else { c += 1; which = 4; }
                    }
                    r = wrap_near(r, H);
                    c = wrap_near(c, W);
                    s_ar[n] = r;
                    s_ac[n] = c;
                    if (a > 4) {
                        bool eaten = false;                   // an earlier agent of this step emptied the cell
                        for (int g = 0; g < ng2; ++g) eaten = eaten || (s_gr[g] == r && s_gc[g] == c);
                        if (!eaten) {
                            const unsigned int w = s_s1[n][which];
                            s += dw_permille_to_natural((double)(float)(w & 0xffffu)) + dw_permille_to_natural((double)(float)(w >> 16));
                            s_gr[ng2] = r; s_gc[ng2] = c; ++ng2;
                            const size_t off = woff + (size_t)r * W + c;
                            oL[off] = (TO)0.f;                // the grazed cell (ref :214-216)
                            oD[off] = (TO)0.f;
                        }
                        s_st[n] = s;
                    }
                }
            }
        }
        __syncthreads();
        if (lane < N) {
            const int an = b * N + lane;
            const double s = s_st[lane];
            const double cl = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
            A.st[an] = cl;
            A.idx[(size_t)an * 2] = s_ar[lane];
            A.idx[(size_t)an * 2 + 1] = s_ac[lane];
            if (A.agent_ok_next) A.agent_ok_next[an] = (cl * (cl > 0.0 ? 1.0 : 0.0)) < 0.1 ? 0 : 1;
        }
    }
}

}  // namespace dw
