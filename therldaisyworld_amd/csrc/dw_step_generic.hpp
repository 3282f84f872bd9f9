// dw_step_generic.hpp — step_generic: one thread per cell, any grid shape, float64 or float32 arithmetic: the
// in-library reference, the first step from an un-quantised state (InT = double / float), odd shapes.
// Input: any plane format (dw_common.hpp); output: always the canonical binary16 planes.
#pragma once
#include "dw_common.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// step_generic: grid = (ceil(H*W/(256*cpt)), B), block = 256; a thread evaluates `cpt` cells (256 apart) and the
// wave issues ONE set of reduction atomics for all of them: with one set per 64 cells the first step of the
// north-star shape (2.7e8 waves x 3 atomics on 1024 x 3 addresses) took 0.65 s in either arithmetic.
// generic_cells_per_thread() picks cpt so that the launch still has a few thousand workgroups.
// PREC: 0 exact (quantised input), 1 fast, 2 f64, 3 exact from an UN-quantised input (the first step of an episode):
// float32 with a tie bound for non-integer inputs (FirstStepBound), the flagged cells of a workgroup collected in
// LDS and re-evaluated in float64 from the original inputs by densely packed lanes after the cell loop.
// ---------------------------------------------------------------------------------------------
// Error bound of the float32 map on NON-INTEGER inputs (no exact coefficient chain: every operation rounds).
// Derived like the bound of the quantised case (dw_api.hip derive_f32) with these changes: the
// inputs carry iota (u for a float64 state converted to float32, 0 for a float32 state), stencil sums 2u-3u, the
// density 5u + iota, and the absolute error of e is (8u + iota) * M + u |c0| with the per-cell magnitude
// M = |a1| Sl8 + |a2| Sd8 + |a3| li + |a4| di, which enters the growth curve as 2 de |e| / D^2 <= 2 de |w| / Dmin:
//   eps = |K| (eK0 + eK1 om + cW de |w|) + eA |gq| + cS (|k + gq| + k) + slack
struct FirstStepBound {
    float a1, a2, a3, a4;         // |a_i| of the rounded coefficient set
    float c_de, c_c0;             // de = c_de * M + c_c0
    float eK0, eK1, cW, eA, cS, slack;
};
constexpr int kFirstListCap = 1024;   // flagged cells per workgroup held in LDS (overflow: evaluated in line)

__host__ inline int generic_cells_per_thread(long long batch, long long cells_per_world) {
    long long cpt = batch * cells_per_world / (256LL * 8192);          // big jobs: still >= 8192 workgroups
    const long long few = cells_per_world / (256LL * 512);              // any job: <= 512 atomic sets per world
    cpt = cpt > few ? cpt : few;
    return (int)(cpt < 1 ? 1 : (cpt > 32 ? 32 : cpt));                   // <= 32: float partial sums stay exact
}

template <typename InT, int PREC>
__global__ __launch_bounds__(256) void step_generic(const InT* __restrict__ inL,
                                                    const InT* __restrict__ inD,
                                                    plane_t* __restrict__ outL,
                                                    plane_t* __restrict__ outD, int H, int W,
                                                    PhysF32 P, PhysF64 P64,
                                                    StatsDev* __restrict__ stats,
                                                    unsigned long long* __restrict__ fixups,
                                                    unsigned long long* __restrict__ zero_me,
                                                    int zero_n, int cpt = 1, FirstStepBound FB = FirstStepBound{}) {
    __shared__ unsigned int s_list[PREC == 3 ? kFirstListCap : 1];
    __shared__ unsigned int s_nlist;
    if (PREC == 3) {
        if (threadIdx.x == 0) s_nlist = 0u;
        __syncthreads();
    }
    const int b = blockIdx.y;
    const size_t woff = (size_t)b * H * W;
    float mx = 0.f, suml = 0.f, sumd = 0.f;       // integers <= 1000 * cpt: exact in float32
    unsigned int nfixed = 0;
    if (blockIdx.x == 0 && blockIdx.y == 0)      // clear the reduction buffer of the NEXT step
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_me[i] = 0ull;
    for (int it = 0; it < cpt; ++it) {
    const unsigned int ucell = (blockIdx.x * (unsigned int)cpt + it) * 256u + threadIdx.x;   // H*W < 2^31: no wrap
    const int cell = (int)ucell;
    float kl = 0.f, kd = 0.f;
    bool fixed = false, deferred = false;
    if (ucell < (unsigned int)(H * W)) {
        const int r = cell / W, c = cell - r * W;
        const InT* pl = inL + woff;
        const InT* pd = inD + woff;
        if (PREC == 2) {
            double l9[9], d9[9];
            gather9(pl, H, W, r, c, l9);
            gather9(pd, H, W, r, c, d9);
            const CellF64 o = cell_f64(P64, l9, d9);
            kl = (float)dw_round3_k(o.nl);
            kd = (float)dw_round3_k(o.nd);
        } else {
            const int ru = r == 0 ? H - 1 : r - 1, rd = r == H - 1 ? 0 : r + 1;
            const int cl = c == 0 ? W - 1 : c - 1, cr = c == W - 1 ? 0 : c + 1;
#define DW_AT(p, rr, cc) to_permille((p)[(size_t)(rr) * W + (cc)])
            const float li = DW_AT(pl, r, c), di = DW_AT(pd, r, c);
            const float El = (DW_AT(pl, ru, c) + DW_AT(pl, rd, c)) + (DW_AT(pl, r, cl) + DW_AT(pl, r, cr));
            const float Cl = (DW_AT(pl, ru, cl) + DW_AT(pl, ru, cr)) + (DW_AT(pl, rd, cl) + DW_AT(pl, rd, cr));   // pairs as cells4
            const float Ed = (DW_AT(pd, ru, c) + DW_AT(pd, rd, c)) + (DW_AT(pd, r, cl) + DW_AT(pd, r, cr));
            const float Cd = (DW_AT(pd, ru, cl) + DW_AT(pd, ru, cr)) + (DW_AT(pd, rd, cl) + DW_AT(pd, rd, cr));
            // (neighbour columns from the neighbour LANES by DPP, 6 loads instead of 18: measured 2.1x SLOWER - the
            // edge lanes' divergent loads serialise the wave; the 18 loads of a wave coalesce into 6 cache lines)
#undef DW_AT
            const GrowthF32 g = growth_f32<(PREC != 1 && PREC != 3) || kFastSplit>(P, li, di, El, Cl, Ed, Cd);
            if (PREC == 1) {
                kl = finish_fast(li, g.dKl, g.fl);
                kd = finish_fast(di, g.dKd, g.fd);
            } else if (PREC == 3) {
                // k' = rint(clip(k + gq)) is final unless k + gq lies within the cell's error bound of a rounding tie
                // (or of the clip's corners, which are integers: ties of rint are the only discontinuities)
                const float M = fmaf(FB.a4, di, fmaf(FB.a3, li, fmaf(FB.a2, Ed + Cd, FB.a1 * (El + Cl))));
                const float de = fmaf(FB.c_de, M, FB.c_c0);
                auto settle = [&](float k, float gq, float dK, float om, bool& tie) -> float {
                    const float sum = k + gq;
                    const float r = __builtin_rintf(sum);
                    const float eps = fmaf(fabsf(dK), fmaf(FB.cW * de, __builtin_sqrtf(om), fmaf(FB.eK1, om, FB.eK0)),
                                           fmaf(FB.eA, fabsf(gq), fmaf(FB.cS, fabsf(sum) + k, FB.slack)));
                    tie = !(fabsf(sum - r) <= 0.5f - eps);          // NaN: flagged
                    return fminf(fmaxf(r, 0.0f), 1000.0f);
                };
                bool tl, td;
                kl = settle(li, g.gql, g.dKl, g.oml, tl);
                kd = settle(di, g.gqd, g.dKd, g.omd, td);
                if (tl || td) {
                    const unsigned int slot = atomicAdd(&s_nlist, 1u);
                    if (slot < (unsigned int)kFirstListCap) {
                        s_list[slot] = ucell;                    // evaluated after the loop, by densely packed lanes
                        deferred = true;
                    } else {
                        double l9[9], d9[9];
                        gather9(pl, H, W, r, c, l9);
                        gather9(pd, H, W, r, c, d9);
                        const CellF64 o = cell_f64(P64, l9, d9);
                        kl = (float)dw_round3_k(o.nl);
                        kd = (float)dw_round3_k(o.nd);
                    }
                    fixed = true;
                }
            } else {
                bool tl, td;
                kl = finish_exact(P, li, g.gql, g.dKl, g.oml, tl);
                kd = finish_exact(P, di, g.gqd, g.dKd, g.omd, td);
                if (tl || td) {
                    double l9[9], d9[9];
                    gather9(pl, H, W, r, c, l9);
                    gather9(pd, H, W, r, c, d9);
                    const CellF64 o = cell_f64(P64, l9, d9);
                    kl = (float)dw_round3_k(o.nl);
                    kd = (float)dw_round3_k(o.nd);
                    fixed = true;
                }
            }
        }
        if (PREC == 3 && deferred) { kl = 0.f; kd = 0.f; }      // written and counted by the pass below
        else {
        outL[woff + cell] = (plane_t)kl;
        outD[woff + cell] = (plane_t)kd;
        }
    }
    mx = fmaxf(mx, fmaxf(kl, kd));
    suml += kl;
    sumd += kd;
    nfixed += fixed ? 1u : 0u;
    }
    if (PREC == 3) {
        // the workgroup's flagged cells, one per thread: float64 from the original inputs (ref staging: bit-identical
        // to the reference's own first step), stored, and added to this thread's partial reductions
        __syncthreads();
        const unsigned int n = s_nlist < (unsigned int)kFirstListCap ? s_nlist : (unsigned int)kFirstListCap;
        for (unsigned int i = threadIdx.x; i < n; i += 256) {
            const int cell = (int)s_list[i];
            const int r = cell / W, c = cell - r * W;
            double l9[9], d9[9];
            gather9(inL + woff, H, W, r, c, l9);
            gather9(inD + woff, H, W, r, c, d9);
            const CellF64 o = cell_f64(P64, l9, d9);
            const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
            outL[woff + cell] = (plane_t)kl;
            outD[woff + cell] = (plane_t)kd;
            mx = fmaxf(mx, fmaxf(kl, kd));
            suml += kl;
            sumd += kd;
        }
    }
    // per-world reductions: wave shuffles, the four waves through LDS, then one set of atomics per workgroup
    // (same-address atomics cost ~100 ns each: they, not the arithmetic, bound this kernel on few-world jobs)
    __shared__ float s_red[4][4];
    const int wv = threadIdx.x >> 6;
    const float m = wave_max(mx);
    const float sl = wave_sum(suml), sd = wave_sum(sumd);
    const float nfw = wave_sum((float)nfixed);
    if ((threadIdx.x & 63) == 0) { s_red[wv][0] = m; s_red[wv][1] = sl; s_red[wv][2] = sd; s_red[wv][3] = nfw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bm = fmaxf(fmaxf(s_red[0][0], s_red[1][0]), fmaxf(s_red[2][0], s_red[3][0]));
        const float bl = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);      // <= 256 * 32 * 1000: exact
        const float bd = (s_red[0][2] + s_red[1][2]) + (s_red[2][2] + s_red[3][2]);
        const unsigned long long nf = (unsigned long long)((s_red[0][3] + s_red[1][3]) + (s_red[2][3] + s_red[3][3]));
        atomicMax(&stats[b].max_k, (unsigned int)bm);
        atomicAdd(&stats[b].sum_l, (unsigned long long)bl);
        atomicAdd(&stats[b].sum_d, (unsigned long long)bd);
        if (nf) atomicAdd(fixups, nf);
    }
}

}  // namespace dw
