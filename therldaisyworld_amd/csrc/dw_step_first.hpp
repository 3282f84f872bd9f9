// dw_step_first.hpp — step_first_stream: the FIRST step of an episode (from an un-quantised state: float64 natural
// covers or float32 per-mille ones) as a wave-strip streaming kernel - round 4: for every width the steady-state wave-strip
// kernels take (multiples of 256, any other multiple of 4 above 256, and the packed mode of narrow worlds).
//
// step_generic (one thread per cell, 18 loads and ~140 / ~215 float32 VALU instructions per cell in the
// float32-only / bounded-exact arithmetic) is VALU-bound on this step: SQ_ACTIVE_INST_VALU says the SIMDs issue
// vector instructions in every cycle of its 70 / 94 ms at 1024 x 4096^2 (profiles/r03_first_step_pmc.txt).  This
// kernel does the same arithmetic the way the steady-state kernels do: a lane owns 4 adjacent columns, a wave
// marches down a strip of 256 columns with a 3-row register window (one 16- / 32-byte load per lane, row and plane;
// horizontal neighbours by DPP), and two cells are evaluated per packed float32 instruction.
//
//   PREC 1  float32-only: cells4<false> - bit-identical to step_generic<., 1> (same operations in the same order)
//   PREC 3  exact: the float32 map with the error bound for NON-INTEGER inputs (FirstStepBound, dw_step_generic.hpp);
//           a flagged cell's coordinates go into the wave's LDS list, and after its strip the wave re-evaluates them
//           in float64 from the ORIGINAL inputs (reference staging: bit-identical to the reference's own first
//           step), patches its stores and adds them to its reductions (earlier whenever the list is half full).
#pragma once
#include "dw_common.hpp"
#include "dw_step_generic.hpp"
#include "dw_step_stream.hpp"

namespace dw {

struct FirstGeom {
    int B, H, W;
    int SR;                   // rows per wave-strip (<= 64: a lane's partial sums stay exact in float32)
    int ncs, nrs;             // column (ceil(W / 256); packed: 1) and row strips per world (packed: per world GROUP)
    int nstrips;              // B (packed: world groups) * nrs * ncs
    int lpw, wpr;             // packed mode (W < 256): lanes per world row (W / 4), worlds per wave row (64 / lpw)
};
constexpr int kFirstWaveList = 512;   // flagged cells of a wave held in LDS; swept when more than half full (a row adds <= 256)

// four adjacent cells of an un-quantised plane as per-mille float32
__device__ __forceinline__ float4 first_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 first_load4(const double* p) {
    const double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
    return make_float4(to_permille(a.x), to_permille(a.y), to_permille(b.x), to_permille(b.y));
}

// PREC 3: the float32 map of four cells with the bound for non-integer inputs; the same operations per cell as
// step_generic<., 3>, two cells per packed instruction.  tie[i]: cell i needs float64.
__device__ __forceinline__ void cells4_first(const PhysF32& P, const FirstStepBound& FB, const Row4& upL,
                                             const Row4& miL, const Row4& dnL, const Row4& upD, const Row4& miD,
                                             const Row4& dnD, float* ol, float* od, bool* tie) {
#pragma clang fp contract(off)
    using T = dw_f32x2;
    using V = Lanes<T>;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        auto pr = [&](const float* a) -> T { return V::load(a, i); };
        const T li = pr(miL.x), di = pr(miD.x);
        const T El = pr(miL.h2) + (pr(upL.x) + pr(dnL.x));
        const T Cl = pr(upL.h2) + pr(dnL.h2);
        const T Ed = pr(miD.h2) + (pr(upD.x) + pr(dnD.x));
        const T Cd = pr(upD.h2) + pr(dnD.h2);
        const GrowthT<T> g = growth_t<kFastSplit, T, false>(P, li, di, El, Cl, Ed, Cd);
        const T M = V::fma(T(FB.a4), di, V::fma(T(FB.a3), li, V::fma(T(FB.a2), Ed + Cd, T(FB.a1) * (El + Cl))));
        const T de = V::fma(T(FB.c_de), M, T(FB.c_c0));
        const T cwde = T(FB.cW) * de;
        auto settle = [&](T k, T gq, T dK, T om, bool* t) -> T {
            const T sum = k + gq;
            const T r = V::rint(sum);
            const T eps = V::fma(V::abs(dK), V::fma(cwde, V::sqrt(om), V::fma(T(FB.eK1), om, T(FB.eK0))),
                                 V::fma(T(FB.eA), V::abs(gq), V::fma(T(FB.cS), V::abs(sum) + k, T(FB.slack))));
            const T frac = V::abs(sum - r), thr = T(0.5f) - eps;
            t[0] = !(frac.x <= thr.x);                          // NaN: flagged
            t[1] = !(frac.y <= thr.y);
            return V::clip(r);
        };
        bool tl[2], td[2];
        const T vl = settle(li, g.gql, g.dKl, g.oml, tl);
        const T vd = settle(di, g.gqd, g.dKd, g.omd, td);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ol[i + e] = V::get(vl, e);
            od[i + e] = V::get(vd, e);
            tie[i + e] = tl[e] || td[e];
        }
    }
}

// grid = ceil(nstrips / 4) workgroups of 4 independent waves.  HALO as in step_stream (dw_step_stream.hpp): 0: W == 256
// (the toroidal wrap is a rotation inside the wave), 1: W a multiple of 256 (the columns beside the strip: one extra load
// per row and plane, lanes 0-31 the left one, lanes 32-63 the right one), 2: any other multiple of 4 above 256 (the last
// strip of a row is narrower: its idle lanes shadow the last active one), 3: packed - W < 256, 64 / (W/4) worlds side by
// side in one wave row, the wrap a rotation inside each world's lane group, reductions per lane group.
template <typename InT, int PREC, int HALO>
__global__ __launch_bounds__(256) void step_first_stream(const InT* __restrict__ inL, const InT* __restrict__ inD,
                                                         plane_t* __restrict__ outL, plane_t* __restrict__ outD,
                                                         FirstGeom G, PhysF32 P, PhysF64 P64,
                                                         StatsDev* __restrict__ stats,
                                                         unsigned long long* __restrict__ fixups,
                                                         unsigned long long* __restrict__ zero_me, int zero_n,
                                                         FirstStepBound FB) {
    static_assert(PREC == 1 || PREC == 3, "float32-only or bounded exact");
    constexpr bool PACK = HALO == 3;
    __shared__ unsigned int s_list[PREC == 3 ? 4 * kFirstWaveList : 1];
    if (blockIdx.x == 0)                                        // clear the reduction buffer of the NEXT step
        for (int i = threadIdx.x; i < zero_n; i += 256) zero_me[i] = 0ull;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wv;
    if (strip >= G.nstrips) return;                             // (no workgroup barrier below)
    const int spw = G.nrs * G.ncs;
    const int b = strip / spw, rem = strip - b * spw;           // packed: b is a GROUP of wpr worlds
    const int rs = rem / G.ncs, cs = rem - rs * G.ncs;
    const int r0 = rs * G.SR, nr = min(G.SR, G.H - r0);
    // this lane's world, its four columns, the lanes / columns that hold its left and right neighbours
    const int pw = PACK ? lane / G.lpw : 0, pj = PACK ? lane - pw * G.lpw : 0;
    const int world = PACK ? min(b * G.wpr + min(pw, G.wpr - 1), G.B - 1) : b;      // idle lanes shadow a real world
    const int lsrc = PACK ? (pj == 0 ? lane + G.lpw - 1 : lane - 1) : 0;
    const int rsrc = PACK ? (pj == G.lpw - 1 ? lane - (G.lpw - 1) : lane + 1) : 0;
    const int ncq = PACK ? 64 : min(64, (G.W - cs * 256) >> 2);                     // active lanes of the strip
    const int last_lane = ncq - 1;
    const bool active = PACK ? (pw < G.wpr && b * G.wpr + pw < G.B) : lane < ncq;
    const int c0 = PACK ? 4 * pj : cs * 256 + 4 * min(lane, last_lane);
    const size_t woff = (size_t)world * G.H * G.W;
    const InT* pl = inL + woff;
    const InT* pd = inD + woff;
    int hc = lane < 32 ? cs * 256 - 1 : cs * 256 + 4 * ncq;     // HALO 1 / 2: the column beside the strip for this half-wave
    hc = hc < 0 ? hc + G.W : (hc >= G.W ? hc - G.W : hc);
    unsigned int* list = s_list + (PREC == 3 ? wv * kFirstWaveList : 0);
    unsigned int nlist = 0;                                     // wave-uniform

    struct RawF { float4 l, d; float hl, hd; };
    auto load_raw = [&](int r) -> RawF {                        // r in [-1, H]: wrapped onto the torus
        const int rr = r < 0 ? r + G.H : (r >= G.H ? r - G.H : r);
        const size_t ro = (size_t)rr * G.W;
        RawF o;
        o.l = first_load4(pl + ro + c0);
        o.d = first_load4(pd + ro + c0);
        o.hl = (HALO == 1 || HALO == 2) ? to_permille(pl[ro + hc]) : 0.f;
        o.hd = (HALO == 1 || HALO == 2) ? to_permille(pd[ro + hc]) : 0.f;
        return o;
    };
    auto rows_of = [&](const RawF& w, Row4& L, Row4& D) {
        float ln, rn;
        lr_neighbours<HALO>(w.l, w.hl, lane, last_lane, ln, rn, lsrc, rsrc);
        L = make_row(w.l, ln, rn);
        lr_neighbours<HALO>(w.d, w.hd, lane, last_lane, ln, rn, lsrc, rsrc);
        D = make_row(w.d, ln, rn);
    };

    float mx = 0.f, suml = 0.f, sumd = 0.f;                     // integers <= 64 * 4 * 1000 per lane: exact
    unsigned int nfixed = 0;
    // the listed cells, one per lane: float64 from the original inputs, over the row stores of the loop
    auto sweep_list = [&]() {
        wait_row_stores_before_patching();                      // (also orders the list's LDS writes before the reads)
        __builtin_amdgcn_wave_barrier();
        for (unsigned int i = lane; i < nlist; i += 64) {
            const unsigned int e = list[i];
            const int r = (int)(e >> 16);
            int c = (int)(e & 0xffffu);
            int wi = b;
            if (PACK) {                                         // wave-row column -> (world of the group, column)
                const int pwc = c / G.W;
                c -= pwc * G.W;
                wi = b * G.wpr + pwc;
            }
            const size_t wo = (size_t)wi * G.H * G.W;
            double l9[9], d9[9];
            gather9(inL + wo, G.H, G.W, r, c, l9);
            gather9(inD + wo, G.H, G.W, r, c, d9);
            const CellF64 o = cell_f64(P64, l9, d9);
            const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
            outL[wo + (size_t)r * G.W + c] = (plane_t)kl;
            outD[wo + (size_t)r * G.W + c] = (plane_t)kd;
            if (PACK) {                                         // the entry's world is not this lane's
                atomicMax(&stats[wi].max_k, (unsigned int)fmaxf(kl, kd));
                atomicAdd(&stats[wi].sum_l, (unsigned long long)kl);
                atomicAdd(&stats[wi].sum_d, (unsigned long long)kd);
            } else {
                mx = fmaxf(mx, fmaxf(kl, kd));
                suml += kl;
                sumd += kd;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the list is reused: reads above before later pushes
        __builtin_amdgcn_wave_barrier();
        nlist = 0;
    };

    Row4 upL, upD, miL, miD, dnL, dnD;
    rows_of(load_raw(r0 - 1), upL, upD);
    rows_of(load_raw(r0), miL, miD);
    RawF nxt = load_raw(r0 + 1);
    for (int k = 0; k < nr; ++k) {
        rows_of(nxt, dnL, dnD);
        if (k + 1 < nr) nxt = load_raw(r0 + k + 2);             // in flight while this row is evaluated
        float ol[4], od[4];
        if (PREC == 1) {
            bool unused[4];
            cells4<false, false, bool>(P, upL, miL, dnL, upD, miD, dnD, ol, od, unused);
        } else {
            bool tie[4];
            cells4_first(P, FB, upL, miL, dnL, upD, miD, dnD, ol, od, tie);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool t = tie[i] && (HALO < 2 || active);
                const unsigned long long mask = lane_mask(t);
                if (mask == 0ull) continue;                     // wave-uniform
                if (t) {
                    const unsigned int slot = nlist + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                                __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    // H, W <= 65535; packed: the column inside the wave row (world of the group * W + column)
                    list[slot] = ((unsigned int)(r0 + k) << 16) | (unsigned int)((PACK ? pw * G.W : 0) + c0 + i);
                    ol[i] = 0.f;                                // written and counted by sweep_list()
                    od[i] = 0.f;
                    ++nfixed;
                }
                nlist += (unsigned int)__popcll(mask);
            }
        }
        if (HALO < 2 || active) {
            const size_t off = woff + (size_t)(r0 + k) * G.W + c0;
            stream_store4(outL + off, make_float4(ol[0], ol[1], ol[2], ol[3]));
            stream_store4(outD + off, make_float4(od[0], od[1], od[2], od[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mx = fmaxf(mx, fmaxf(ol[i], od[i]));
                suml += ol[i];
                sumd += od[i];
            }
        }
        upL = miL; upD = miD; miL = dnL; miD = dnD;
        if (PREC == 3 && __builtin_expect(nlist > (unsigned int)(kFirstWaveList - 256), 0)) sweep_list();
    }
    if (PREC == 3 && nlist) sweep_list();
    const float nf = wave_sum((float)nfixed);
    if (lane == 0 && nf > 0.f) atomicAdd(fixups, (unsigned long long)nf);
    if (PACK) {                                                 // per world of the wave row (dw_step_stream.hpp)
        float m = mx, sl = suml, sd = sumd;
        if ((G.lpw & (G.lpw - 1)) == 0) {                       // power-of-two groups: butterfly
            for (int o = G.lpw >> 1; o > 0; o >>= 1) {
                m = fmaxf(m, __shfl_xor(m, o, 64));
                sl += __shfl_xor(sl, o, 64);
                sd += __shfl_xor(sd, o, 64);
            }
        } else {                                                // any group size: the group's first lane gathers
            for (int o = 1; o < G.lpw; ++o) {
                const int src = min(lane + o, 63);
                m = fmaxf(m, __shfl(mx, src, 64));
                sl += __shfl(suml, src, 64);
                sd += __shfl(sumd, src, 64);
            }
        }
        if (pj == 0 && active) {
            atomicMax(&stats[world].max_k, (unsigned int)m);
            atomicAdd(&stats[world].sum_l, (unsigned long long)sl);
            atomicAdd(&stats[world].sum_d, (unsigned long long)sd);
        }
    } else {
        const float m = wave_max(mx);
        const float sl = wave_sum(suml), sd = wave_sum(sumd);   // <= 64 lanes * 256000: exact
        if (lane == 0) {
            atomicMax(&stats[b].max_k, (unsigned int)m);
            atomicAdd(&stats[b].sum_l, (unsigned long long)sl);
            atomicAdd(&stats[b].sum_d, (unsigned long long)sd);
        }
    }
}

}  // namespace dw
