// dw_step_generic.hpp — step_generic: one thread per cell, any grid shape, float64 or float32 arithmetic: the
// in-library reference, the first step from an un-quantised state (InT = double / float), odd shapes.
// Input: any plane format (dw_common.hpp); output: always the canonical binary16 planes.
#pragma once
#include "dw_common.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// step_generic: grid = (ceil(H*W/256), B), block = 256.  PREC: 0 exact, 1 fast, 2 f64.
// ---------------------------------------------------------------------------------------------
template <typename InT, int PREC>
__global__ __launch_bounds__(256) void step_generic(const InT* __restrict__ inL,
                                                    const InT* __restrict__ inD,
                                                    plane_t* __restrict__ outL,
                                                    plane_t* __restrict__ outD, int H, int W,
                                                    PhysF32 P, PhysF64 P64,
                                                    StatsDev* __restrict__ stats,
                                                    unsigned long long* __restrict__ fixups,
                                                    unsigned long long* __restrict__ zero_me,
                                                    int zero_n) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const size_t woff = (size_t)b * H * W;
    float kl = 0.f, kd = 0.f;
    bool fixed = false;
    if (blockIdx.x == 0 && blockIdx.y == 0)      // clear the reduction buffer of the NEXT step
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_me[i] = 0ull;
    if (cell < H * W) {
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
#undef DW_AT
            const GrowthF32 g = growth_f32<PREC != 1 || kFastSplit>(P, li, di, El, Cl, Ed, Cd);
            if (PREC == 1) {
                kl = finish_fast(li, g.dKl, g.fl);
                kd = finish_fast(di, g.dKd, g.fd);
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
        outL[woff + cell] = (plane_t)kl;
        outD[woff + cell] = (plane_t)kd;
    }
    // per-world reductions: wave shuffles, then one set of atomics per wave
    const float m = wave_max(fmaxf(kl, kd));
    const float sl = wave_sum(kl), sd = wave_sum(kd);
    const unsigned long long nf = __popcll(__ballot(fixed));
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[b].max_k, (unsigned int)m);
        atomicAdd(&stats[b].sum_l, (unsigned long long)sl);
        atomicAdd(&stats[b].sum_d, (unsigned long long)sd);
        if (nf) atomicAdd(fixups, nf);
    }
}

}  // namespace dw
