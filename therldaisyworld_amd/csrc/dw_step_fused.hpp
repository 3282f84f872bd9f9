// dw_step_fused.hpp — step_stream_fused2[_exact]: two consecutive steps per HBM round trip (temporal
// blocking inside the wave-strip design; dw_step_n on wide grids without agent updates).
#pragma once
#include <type_traits>
#include "dw_step_stream.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// step_stream_fused2 — TWO consecutive steps per HBM round trip (float32-only mode, no agent
// update between the steps: dw_step_n on wide grids).  Temporal blocking inside the wave-strip
// design: as a wave marches down its strip, every new input row yields one row of step-1 results
// (kept only in a second 3-row register window, never written to memory) and, one row behind it,
// one row of step-2 results, which is stored.  HBM traffic per cell-update drops to 8.25 B
// (measured by PMC, profiles/), and the kernel becomes VALU-bound.
// Planes are binary16 on both sides (plane_t, dw_common.hpp): 4.13 HBM bytes per cell-update measured.
//
// Horizontal neighbours of step-1 results come from adjacent lanes by DPP like the inputs do.
//   ROT (W == 256): the wave spans the whole torus row, all 64 lanes produce output.
//   RING (W == 1024): the FOUR waves of a workgroup span the torus row (wave w owns columns 256 w ... 256 w + 255);
//                   per row they exchange their edge columns - of the input row and of the step-1 row - through 256
//                   bytes of LDS with one workgroup barrier, so all 64 lanes of every wave produce output (the
//                   overlapped strips would need five 248-column strips: 25 % of the lanes redundant).
//   OVL (other W):  strips overlap by one lane (4 columns) on each side: lanes 0 and 63 load and
//                   compute step 1 but only lanes 1..62 (248 columns) produce output; no halo loads.
// Vertically a strip of SR output rows reads SR+4 input rows and computes SR+2 step-1 rows.
// ---------------------------------------------------------------------------------------------
struct FusedGeom {
    int B, H, W;
    int SR;                   // output rows per wave-strip
    int ncs, nrs;             // column / row strips per world
    int nstrips, nwg, chunk;
    int cols_per_strip;       // 256 (ROT) or 248 (OVL)
    int qcap, mcap;           // queue / mismatch-list capacities in use (tests shrink them)
    int lpw, wpr;             // packed mode (W < 256): lanes per world row (W/4), worlds per wave row (64 / lpw)
    int sure_need;            // STATS: sure step-2 row groups after which a wave's count cannot matter any more:
                              // 9 per agent (patched cells) + 9 per possible step-1 mismatch (deducted) + 1
};

// float64 step-1 value of grid cell (r, c) (any integers: wrapped onto the torus) from the input planes,
// as a packed light | dark << 16 word
template <typename TI>
__device__ inline unsigned int exact1_word(const TI* __restrict__ pL, const TI* __restrict__ pD, int H, int W,
                                           int r, int c, const PhysF64& Pa) {
    const int rr = wrap_near2(r, H), cc = wrap_near2(c, W);    // (callers pass in-range coordinates +- 4, H >= 3: no division)
    const int ru = rr == 0 ? H - 1 : rr - 1, rd = rr == H - 1 ? 0 : rr + 1;
    const int cl = cc == 0 ? W - 1 : cc - 1, cr = cc == W - 1 ? 0 : cc + 1;
    const int rows[3] = {ru, rr, rd}, cols[3] = {cl, cc, cr};
    unsigned int w1[9];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            const size_t o = (size_t)rows[x] * W + cols[y];
            w1[x * 3 + y] = (unsigned)(float)pL[o] | ((unsigned)(float)pD[o] << 16);
        }
    const NewCoverF64 s1 = cell_f64_lean(Pa, w1);
    return (unsigned)dw_round3_k(s1.nl) | ((unsigned)dw_round3_k(s1.nd) << 16);
}

// exact two-step value of one cell straight from the input planes, all in float64: nine step-1
// evaluations (luminosity La) feeding one step-2 evaluation (Lb).  Used only to repair the rare
// dependents of a float32 step-1 mismatch and as the overflow fallback.
template <typename TI>
__device__ inline void exact2_cell(const TI* __restrict__ pL, const TI* __restrict__ pD, int H, int W, int r,
                                   int c, const PhysF64& Pa, const PhysF64& Pb, float& kl, float& kd) {
    unsigned int w2[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            int rr = r + a - 1, cc = c + e - 1;
            rr = rr < 0 ? rr + H : (rr >= H ? rr - H : rr);
            cc = cc < 0 ? cc + W : (cc >= W ? cc - W : cc);
            const int ru = rr == 0 ? H - 1 : rr - 1, rd = rr == H - 1 ? 0 : rr + 1;
            const int cl = cc == 0 ? W - 1 : cc - 1, cr = cc == W - 1 ? 0 : cc + 1;
            const int rows[3] = {ru, rr, rd}, cols[3] = {cl, cc, cr};
            unsigned int w1[9];
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    const size_t o = (size_t)rows[x] * W + cols[y];
                    w1[x * 3 + y] = (unsigned)(float)pL[o] | ((unsigned)(float)pD[o] << 16);
                }
            const NewCoverF64 s1 = cell_f64_lean(Pa, w1);
            w2[a * 3 + e] = (unsigned)dw_round3_k(s1.nl) | ((unsigned)dw_round3_k(s1.nd) << 16);
        }
    const NewCoverF64 s2 = cell_f64_lean(Pb, w2);
    kl = (float)dw_round3_k(s2.nl);
    kd = (float)dw_round3_k(s2.nd);
}

constexpr int kMismatchCap = 64;            // float32 step-1 mismatches per wave-strip held in LDS

// EXACT variant (the default mode's dw_step_n on wide grids).  Both steps run in float32 with the
// per-cell tie test; near-tie cells of BOTH steps are queued in the wave's LDS queue with their 3x3
// payload (step 1: the inputs; step 2: the float32 step-1 values).  After the strip the same wave
//   F1  re-evaluates every queued step-1 cell in float64; almost always the float32 value was right
//       (~97 %); a cell where it was not is a MISMATCH,
//   F2  re-evaluates every queued step-2 cell in float64 from its payload and patches the output,
//   F3  for every mismatch recomputes, entirely in float64 from the input planes, the (up to) nine
//       output cells that depend on it, and patches them (rare: ~0.01 % of cells).
// A cell's output is therefore the float64 result whenever any float32 rounding on its dependency
// cone was uncertain.  Queue / mismatch-list overflow: the whole strip is recomputed in float64.
// Fused launches leave the per-world reductions untouched (dw_step_n always ends with a single step,
// which recomputes them), they only keep the double-buffer protocol.
// LAG (all kernels): step 2 runs one row further behind step 1, so that the two row maps of an iteration
// are independent (see the loop below).
// PACK (with ROT): narrow worlds (W | 256) side by side in the wave row, as in step_stream<halo=packed>:
// every lane has its own world, the horizontal wrap is a rotation inside the world's lane group, and a
// local column lc of the wave row decodes to (world, column) = (lc / W, lc % W).
// STATS: what an episode harness needs from a step pair with agents in between (dw_agents_fused.hpp):
//   pstats[2*world]     max over the world's exact step-1 values (the "biosphere alive" flag of step t)
//   pstats[2*world + 1] number of this world's output row groups (4 cells of a lane) holding a step-2
//                       value above `thr_hi` that cannot be an artefact of float32 or be undone by the few
//                       cells patched afterwards - a sound lower bound, see agents_lookahead_patch.
enum { kFusedOvl = 0, kFusedRot = 1, kFusedRing = 2 };
template <int MODE, bool EXACT, bool PACK = false, bool STATS = false, bool SYM = false, typename TI = plane_t,
          typename TO = plane_t>
__device__ __forceinline__ void fused2_body(const TI* __restrict__ inL, const TI* __restrict__ inD,
                                            TO* __restrict__ outL, TO* __restrict__ outD, const FusedGeom& G,
                                            const PhysF32& P1_, const PhysF32& P2_, const PhysF64& P64,
                                            const double& La, const double& Lb,
                                            unsigned long long* __restrict__ zero_me, int zero_n,
                                            unsigned int* __restrict__ pstats = nullptr, float thr_hi = 0.f) {
    // LAG: step 2 runs one row further behind step 1, on results of earlier iterations only (see below).
    // Measured: exact kernels -5...-11 %, packed float32 -6 %, W = 256 float32 -4 %,
    // overlapped strips -1.5 %.  (-DDW_FUSED_LAG=0 builds the dependent order for comparison.)
#ifndef DW_FUSED_LAG
#define DW_FUSED_LAG 1
#endif
    constexpr bool LAG = DW_FUSED_LAG != 0;
    constexpr bool ROT = MODE == kFusedRot, RING = MODE == kFusedRing;
    static_assert(!RING || (LAG && !PACK), "the ring exchange is written for the lagged loop of un-packed worlds");
    __shared__ float s_edge[RING ? 2 * 4 * 8 : 1];               // RING: [parity][wave][.w of lane 63 x4 | .x of lane 0 x4]
    __shared__ uint4 s_queue[EXACT ? 4 * kWaveQueueCap * 3 : 1];
    __shared__ unsigned int s_mm[EXACT ? 4 * kMismatchCap : 1];
    const int bid = blockIdx.x;
    const int wg = (bid & 7) * G.chunk + (bid >> 3);
    if (wg >= G.nwg) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (wg == 0)
        for (int i = tid; i < zero_n; i += 256) zero_me[i] = 0ull;
    // RING: the workgroup is the strip (all four waves stay: they meet at a barrier every row)
    const int s = RING ? wg : __builtin_amdgcn_readfirstlane(wg * 4 + wv);   // wave-uniform: row addressing on the scalar unit
    if (!RING && s >= G.nstrips) return;
    uint4* q = s_queue + (EXACT ? wv * kWaveQueueCap * 3 : 0);
    unsigned int* mm = s_mm + (EXACT ? wv * kMismatchCap : 0);
    const int spw = G.nrs * G.ncs;
    const int b = s / spw;
    const int sw = s - b * spw;
    const int rs = sw / G.ncs, cs = sw - rs * G.ncs;
    const int r0 = rs * G.SR;
    const int nr = min(G.SR, G.H - r0);
    const int pw = PACK ? lane / G.lpw : 0, pj = PACK ? lane - pw * G.lpw : 0;
    const int lsrc = PACK ? (pj == 0 ? lane + G.lpw - 1 : lane - 1) : 0;
    const int rsrc = PACK ? (pj == G.lpw - 1 ? lane - (G.lpw - 1) : lane + 1) : 0;
    // world base of a local column (PACK: the column's own world; lanes of missing worlds shadow the last)
    auto world_off = [&](int lc) -> size_t {
        return (size_t)(PACK ? min(b * G.wpr + min((lc >> 2) / G.lpw, G.wpr - 1), G.B - 1) : b) * G.H * G.W;
    };
    const size_t woff = world_off(4 * lane);
    const int c00 = ROT ? 0 : (RING ? 256 * __builtin_amdgcn_readfirstlane(wv) : cs * 248 - 4);   // grid column of local column 0 (OVL: may be -4)
    int col = PACK ? 4 * pj : c00 + 4 * lane;
    col = col < 0 ? col + G.W : col;
    col = col >= G.W ? col - G.W : col;                         // W >= 256 > 252: one wrap suffices
    const bool writes = PACK ? (pw < G.wpr && b * G.wpr + pw < G.B)
                             : ((ROT || RING) ? true : (lane >= 1 && lane <= 62 && cs * 248 + 4 * (lane - 1) < G.W));
    // which of my four step-1 cells feed an output cell of this wave (exact mode: only their ties matter)
    bool need1[4] = {true, true, true, true};
    if (PACK) { need1[0] = need1[1] = need1[2] = need1[3] = writes; }
    if (EXACT && !ROT && !RING) {
        const bool wl = __builtin_amdgcn_update_dpp(0, writes ? 1 : 0, kDppWaveShr1, 0xf, 0xf, false) != 0;
        const bool wr = __builtin_amdgcn_update_dpp(0, writes ? 1 : 0, kDppWaveShl1, 0xf, 0xf, false) != 0;
        need1[0] = writes || wl;
        need1[1] = writes;
        need1[2] = writes;
        need1[3] = writes || wr;
    }
    // tie flags: lane masks on the scalar unit, except in the packed variants and the STATS variant of the overlapped
    // strips (per-lane bools: they are at their register budget with those and spill inside the row loop otherwise)
    constexpr bool LANE_BOOLS = PACK || (STATS && MODE == kFusedOvl);
    using TieT = std::conditional_t<LANE_BOOLS, bool, unsigned long long>;
    TieT need1m[4], writes_m;
    if constexpr (LANE_BOOLS) {
        for (int i = 0; i < 4; ++i) need1m[i] = need1[i];
        writes_m = writes;
    } else {
        for (int i = 0; i < 4; ++i) need1m[i] = lane_mask(need1[i]);
        writes_m = lane_mask(writes);
    }
    const TI* pL = inL + woff;
    const TI* pD = inD + woff;

    struct RawIn { dw_f16x4 l, d; };                            // one row of both planes as loaded (4 x binary16 each)
    auto load_raw = [&](int rr) -> RawIn {                      // rr in [r0-2, r0+nr+1], clamped + wrapped
        rr = min(rr, r0 + nr + 1);
        rr = rr < 0 ? rr + G.H : rr;
        rr = rr >= G.H ? rr - G.H : rr;
        RawIn w;
        w.l = stream_load4_raw(pL + (size_t)rr * G.W + col);
        w.d = stream_load4_raw(pD + (size_t)rr * G.W + col);
        return w;
    };
    // The edge sums of a prefetched row are wanted HERE: left alone, the compiler sinks the adds (and half of the
    // conversions) past the next iteration's store branch while the cross-lane moves, being convergent, stay, so the
    // v_mov_b32_dpp + v_add_f32 pairs end up in different basic blocks and cannot fold into v_add_f32_dpp
    // (4 instructions per row).  No instruction is emitted for the pin itself.
    auto pin_edge_sums = [&](Row4& L, Row4& D) {
        asm volatile("" : "+v"(L.h2[0]), "+v"(L.h2[3]), "+v"(D.h2[0]), "+v"(D.h2[3]));
    };
    auto nbrs = [&](const float4& v, float& a, float& c) {
        if (PACK) { a = __shfl(v.w, lsrc, 64); c = __shfl(v.x, rsrc, 64); }
        else if (ROT) { a = dpp_mov_nb<kDppWaveRor1>(v.w); c = dpp_mov_nb<kDppWaveRol1>(v.x); }
        else { a = dpp_mov_nb<kDppWaveShr1>(v.w); c = dpp_mov_nb<kDppWaveShl1>(v.x); }   // lanes 0/63: unused
    };
    auto to_rows4 = [&](const float4& l, const float4& d, Row4& L, Row4& D) {
        float a, c;
        nbrs(l, a, c);
        L = make_row(l, a, c);
        nbrs(d, a, c);
        D = make_row(d, a, c);
    };
    // RING: n (<= 4) rows of a plane at once -> Row4s, the neighbours of lanes 0 / 63 from the adjacent waves
    int ring_par = 0;
    auto rows_ring = [&](const float4* v, Row4* const* out, auto NC) {
        constexpr int n = decltype(NC)::value;
        float* mine = s_edge + (ring_par * 4 + wv) * 8;
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < n; ++i) mine[i] = v[i].w;
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < n; ++i) mine[4 + i] = v[i].x;
        }
        __syncthreads();                                        // one barrier per exchange; the parity flip makes the
        const float* lw = s_edge + (ring_par * 4 + ((wv + 3) & 3)) * 8;       // next exchange write the other half
        const float* rw = s_edge + (ring_par * 4 + ((wv + 1) & 3)) * 8 + 4;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const float a = dpp_mov<kDppWaveShr1>(lw[i], v[i].w);   // lane 0 keeps `old` = the left wave's last column
            const float c = dpp_mov<kDppWaveShl1>(rw[i], v[i].x);   // lane 63 keeps the right wave's first column
            *out[i] = make_row(v[i], a, c);                       // straight into its window slot
        }
        ring_par ^= 1;
    };
    unsigned int nq = 0;                                        // queued entries of this wave (uniform)
    // PIN (exact kernels that have the registers): a packed instruction reads at most ONE scalar pair, so the
    // addend constants of the fmas whose multiplier is a scalar pair too - the lo chain's seed c0l, p of the bare
    // fraction, the tie bracket's constant, tie_lo beside eA - were copied into VGPRs again and again (the compiler
    // rematerialises them per row map: ~1.1 v_mov per cell-evaluation).  Held in VGPRs for the whole strip instead.
#ifndef DW_PIN_CONSTS
#define DW_PIN_CONSTS -1                                        // per kernel variant (see PIN below)
#endif
    PhysF32 P1 = P1_, P2 = P2_;
    // bit mask: 1 c0l (both steps), 2 pck, 4 eKb, 8 gt.  What fits without scratch traffic on the row loop's main path
    // (tools/isa_report.py; the queue sweep inside the loop raised the pressure): overlapped strips 14, the ring 8,
    // rotating strips none.
    constexpr int PIN = DW_PIN_CONSTS < 0 ? (MODE == kFusedOvl ? 14 : (MODE == kFusedRing ? 8 : 0)) : DW_PIN_CONSTS;
    if constexpr (EXACT && PIN != 0 && !STATS && !PACK) {
        if constexpr ((PIN & 1) != 0) asm volatile("" : "+v"(P1.c0l), "+v"(P2.c0l));
        if constexpr ((PIN & 2) != 0) { asm volatile("" : "+v"(P1.pck)); P2.pck = P1.pck; }
        if constexpr ((PIN & 4) != 0) { asm volatile("" : "+v"(P1.eKb)); P2.eKb = P1.eKb; }
        if constexpr ((PIN & 8) != 0) { asm volatile("" : "+v"(P1.gt)); P2.gt = P1.gt; }
    }
    // STATS accumulators.  Un-packed worlds (CHEAP): st_m1 = this lane's maximum of the float32 step-1 values, ties
    // included (a near-tie cell's float32 value is within one quantum of the exact one, so a maximum >= thr + 2 proves
    // the biosphere alive whatever the ties were); st_x1 = exact values that decide otherwise (maximum <= thr + 1: a
    // dying world) - the float64 value of every queued step-1 cell, and the NON-tie cells of row groups whose maximum
    // is exactly thr + 1; st_c2w = wave total of output row groups whose float32 step-2 maximum is >= thr + 2 (exact
    // mode; float32-only mode: > thr).  Packed worlds keep one maximum / count per lane (several worlds per wave).
    constexpr bool CHEAP = STATS && !PACK;
    static_assert(LAG || !CHEAP, "the dependent-order loop (DW_FUSED_LAG=0, experiments) keeps the per-lane statistics");
    float st_m1 = 0.f, st_x1 = 0.f;
    unsigned int st_c2 = 0, st_nmm = 0, st_c2w = 0;
    // CHEAP: both results are only ever read as predicates by agents_lookahead_patch - "some step-1 value > thr" and
    // "more sure step-2 row groups than the patches can touch" (9 per grazed cell, 9 per step-1 mismatch deducted
    // below).  Once a wave holds a CERTAIN step-1 value >= thr + 2 and G.sure_need sure step-2 groups, nothing it could
    // still add changes either predicate: the rest of the strip runs in a copy of the row loop without the statistics
    // (a living world with 4 agents: after ~12 of its 64 rows; G.sure_need, host).  st_sure1: wave-uniform.
    bool st_sure1 = false;
    const unsigned long long writes_mask = CHEAP ? lane_mask(writes) : 0ull;
    const float thr_c2 = EXACT ? thr_hi + 1.0f : thr_hi;
    // one row of the map with coefficient set P: (up, mid, down) -> new values; exact mode also queues
    // the near-tie cells (kind 1 = step 1, 2 = step 2; lrow = row index relative to grid row r0-2)
    auto row_map = [&](const PhysF32& P, const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD,
                       const Row4& miD, const Row4& dnD, float4& nl, float4& nd, int kind, int lrow,
                       const TieT* use, float* sure_max = nullptr) {
        float ol[4], od[4];
        TieT tie[4];
        cells4<EXACT, SYM, TieT>(P, upL, miL, dnL, upD, miD, dnD, ol, od, tie);
        if (EXACT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tie[i] = tie[i] & use[i];
        }
        if (STATS && sure_max) {
            if (CHEAP) {                                         // the row group's float32 maximum (four v_max3)
                *sure_max = fmaxf(fmaxf(fmaxf(ol[0], ol[1]), fmaxf(ol[2], ol[3])), fmaxf(fmaxf(od[0], od[1]), fmaxf(od[2], od[3])));
            } else {                                             // max over the cells whose float32 value is certain
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(m, (EXACT && tie_lane(tie[i])) ? 0.f : fmaxf(ol[i], od[i]));
                *sure_max = m;
            }
        }
        nl = make_float4(ol[0], ol[1], ol[2], ol[3]);
        nd = make_float4(od[0], od[1], od[2], od[3]);
        if (EXACT && tie_mask(TieT(tie[0] | tie[1] | tie[2] | tie[3])) != 0ull) {
            queue_tie<0>(tie[0], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<1>(tie[1], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<2>(tie[2], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<3>(tie[3], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
        }
    };
    // the same in two halves (LAG: the cells of both steps first, then their queue pushes)
    auto row_cells = [&](const PhysF32& P, const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD,
                         const Row4& miD, const Row4& dnD, float* ol, float* od, TieT* tie,
                         const TieT* use, float* sure_max = nullptr) {
        cells4<EXACT, SYM, TieT>(P, upL, miL, dnL, upD, miD, dnD, ol, od, tie);
        if (EXACT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tie[i] = tie[i] & use[i];
        }
        if (STATS && sure_max) {
            if (CHEAP) {                                         // the row group's float32 maximum (four v_max3)
                *sure_max = fmaxf(fmaxf(fmaxf(ol[0], ol[1]), fmaxf(ol[2], ol[3])), fmaxf(fmaxf(od[0], od[1]), fmaxf(od[2], od[3])));
            } else {                                             // max over the cells whose float32 value is certain
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(m, (EXACT && tie_lane(tie[i])) ? 0.f : fmaxf(ol[i], od[i]));
                *sure_max = m;
            }
        }
    };
    auto row_queue = [&](const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD, const Row4& miD,
                         const Row4& dnD, const float* ol, const float* od, const TieT* tie, int kind,
                         int lrow) {
        if (EXACT && tie_mask(TieT(tie[0] | tie[1] | tie[2] | tie[3])) != 0ull) {
            queue_tie<0>(tie[0], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<1>(tie[1], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<2>(tie[2], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<3>(tie[3], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
        }
    };
    // STATS bookkeeping of one step-2 / step-1 row map (sm = what row_cells / row_map returned through sure_max)
    auto stats_step2 = [&](float sm) {
        if (CHEAP) st_c2w += (unsigned int)__popcll(__builtin_amdgcn_fcmpf(sm, thr_c2, 2 /* OGT */) & writes_mask);
        else if (writes && sm > thr_hi) st_c2 += 1u;
    };
    auto stats_step1 = [&](float sm, const float* ol, const float* od, const TieT* tie, bool rows_mine) {
        if (!rows_mine) return;                                 // wave-uniform: step-1 rows of MY output cells only
        if (writes) st_m1 = fmaxf(st_m1, sm);
        if (CHEAP) st_sure1 = st_sure1 || (__builtin_amdgcn_fcmpf(sm, thr_hi + 1.0f, 2 /* OGT */) & writes_mask) != 0ull;
        if (CHEAP && EXACT) {
            // a row group whose float32 maximum is exactly thr + 1 decides only through its NON-tie cells (rare)
            const unsigned long long eq = __builtin_amdgcn_fcmpf(sm, thr_hi + 1.0f, 1 /* OEQ */) & writes_mask;
            if (__builtin_expect(eq != 0ull, 0)) {
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) m = fmaxf(m, tie_lane(tie[i]) ? 0.f : fmaxf(ol[i], od[i]));
                st_x1 = fmaxf(st_x1, in_mask(eq) ? m : 0.f);
            }
        }
    };
    const TieT use2[4] = {writes_m, writes_m, writes_m, writes_m};

    // ---- F1 + F2: the float64 sweep over this wave's queue (exact kernels) -------------------------------------
    // Runs when the strip is finished AND whenever the queue is half full (`flush_at`): the queue never overflows
    // on any state the dynamics produce, whatever the strip height (a strip whose queue did overflow lost entries
    // and is recomputed whole in float64 - tens of times slower, which is why it must not happen), and the sweeps
    // run on densely filled lanes.  Step-1 entries whose float32 value was wrong go to the mismatch list, which
    // lives until the end of the strip (F3 needs every output row stored); step-2 entries patch rows this wave has
    // already stored.
    unsigned int nmm = 0;                                       // step-1 mismatches so far (wave-uniform)
    bool redo = false;                                          // queue or mismatch list overflowed: whole strip in float64
    const int flush_at = G.qcap >> 1;
    auto world_of = [&](int lc) -> int {
        return PACK ? min(b * G.wpr + min((lc >> 2) / G.lpw, G.wpr - 1), G.B - 1) : b;
    };
    // grid coordinates of a local (row, column)
    auto grid_rc = [&](int lrow, int lc, int& gr, int& gc) {
        gr = r0 - 2 + lrow;
        gr = gr < 0 ? gr + G.H : (gr >= G.H ? gr - G.H : gr);
        gc = c00 + lc;
        gc = gc < 0 ? gc + G.W : (gc >= G.W ? gc - G.W : gc);
    };
    // world base and grid coordinates of a local (row, column); false if the column's world is missing
    auto locate = [&](int lrow, int lc, size_t& wo, int& gr, int& gc) -> bool {
        grid_rc(lrow, lc, gr, gc);
        wo = world_off(lc);
        if (!PACK) return true;
        const int pwc = (lc >> 2) / G.lpw;
        gc = lc - pwc * G.W;
        return pwc < G.wpr && b * G.wpr + pwc < G.B;
    };
    // is local (row, column) an output cell of this wave?
    auto is_output = [&](int lrow, int lc) -> bool {
        if (lrow < 2 || lrow > nr + 1) return false;
        if (PACK) return (lc >> 2) / G.lpw < G.wpr && b * G.wpr + (lc >> 2) / G.lpw < G.B;
        if (ROT || RING) return true;                        // RING: also the two columns beside the wave's own
        const int ln = lc >> 2;
        return lc >= 4 && lc < 252 && cs * 248 + 4 * (ln - 1) < G.W;
    };
    auto sweep_queue = [&]() {
        if (nq > (unsigned)G.qcap) redo = true;                 // entries were dropped
        if (!redo && nq) {
            wait_row_stores_before_patching();                  // F2 patches rows this wave stored
            // the cold float64 constants come from the kernarg segment HERE (an opaque copy of the reference keeps
            // the loads inside this rarely executed block instead of in scalar registers across the row loop)
            const PhysF64* p64 = &P64;
            const double* pLa = &La;
            const double* pLb = &Lb;
            asm volatile("" : "+s"(p64), "+s"(pLa), "+s"(pLb));
            PhysF64 Pe = *p64;
            const double la = *pLa, lb = *pLb;
            for (unsigned int base = 0; base < nq; base += 64) {
                const unsigned int e = base + lane;
                bool mism = false;
                unsigned int where = 0;
                if (e < nq) {
                    const uint4 e0 = q[e * 3], e1 = q[e * 3 + 1], e2 = q[e * 3 + 2];
                    const unsigned int w[9] = {unpack_ld(e0.z), unpack_ld(e0.w), unpack_ld(e1.x),
                                               unpack_ld(e1.y), unpack_ld(e1.z), unpack_ld(e1.w),
                                               unpack_ld(e2.x), unpack_ld(e2.y), unpack_ld(e2.z)};
                    Pe.L = e0.x == 1u ? la : lb;
                    const NewCoverF64 o = cell_f64_lean(Pe, w);
                    const unsigned int kl = (unsigned)dw_round3_k(o.nl), kd = (unsigned)dw_round3_k(o.nd);
                    where = e0.y;
                    if (e0.x == 1u) {
                        mism = (kl | (kd << 16)) != unpack_ld(e2.w);
                        if (STATS && is_output((int)(where >> 16), (int)(where & 0xffffu))) {   // exact value of a tie cell
                            const unsigned int mx = kl > kd ? kl : kd;
                            if (PACK) atomicMax(&pstats[2 * world_of((int)(where & 0xffffu))], mx);   // any world of the row
                            else st_x1 = fmaxf(st_x1, (float)mx);              // my wave's world: reduced below
                        }
                    } else {
                        int gr, gc;
                        size_t wo;
                        locate((int)(where >> 16), (int)(where & 0xffffu), wo, gr, gc);
                        const size_t off = wo + (size_t)gr * G.W + gc;
                        outL[off] = (TO)(float)kl;
                        outD[off] = (TO)(float)kd;
                    }
                }
                const unsigned long long mask = lane_mask(mism);
                if (mism) {
                    const unsigned int slot = nmm + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (slot < (unsigned)G.mcap) mm[slot] = where;
                }
                nmm += (unsigned)__popcll(mask);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the queue is reused: reads above before later pushes
            __builtin_amdgcn_wave_barrier();
        }
        nq = 0;
    };

    // windows: IL/ID input rows, SL/SD step-1 rows; slot of row index j (0 = row r0-2) is j % 3
    Row4 IL[3], ID[3], SL[3], SD[3];
    {
        const RawIn p0 = load_raw(r0 - 2), p1 = load_raw(r0 - 1), p2 = load_raw(r0);
        if constexpr (RING) {
            const float4 va[4] = {widen4(p0.l), widen4(p0.d), widen4(p1.l), widen4(p1.d)};
            const float4 vb[2] = {widen4(p2.l), widen4(p2.d)};
            Row4* const oa[4] = {&IL[0], &ID[0], &IL[1], &ID[1]};
            Row4* const ob[2] = {&IL[2], &ID[2]};
            rows_ring(va, oa, std::integral_constant<int, 4>{});
            rows_ring(vb, ob, std::integral_constant<int, 2>{});
        } else {
            to_rows4(widen4(p0.l), widen4(p0.d), IL[0], ID[0]);
            to_rows4(widen4(p1.l), widen4(p1.d), IL[1], ID[1]);
            to_rows4(widen4(p2.l), widen4(p2.d), IL[2], ID[2]);
        }
    }
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    if constexpr (LAG) {
        // Software-pipelined by one row: iteration j = 1 .. nr+3 computes
        //   (D1: j <= nr+2)  step-1 row j (grid row r0-2+j) from input rows j-1, j, j+1, and
        //   (D2: j >= 4)     the output row with local index j-2 (grid row r0+j-4) from step-1 rows j-3, j-2, j-1,
        // which are all results of EARLIER iterations: the two row maps of an iteration are independent, so
        // their transcendental chains overlap instead of waiting for each other.  Step-1 row j then replaces
        // step-1 row j-3 in the window.
        auto iter = [&](auto U, auto D1, auto D2, int j, auto ST) {
            constexpr int u = decltype(U)::value;                  // u == j % 3
            constexpr bool do1 = decltype(D1)::value, do2 = decltype(D2)::value;
            constexpr bool st_on = STATS && decltype(ST)::value;   // (the quiet copy of the loop: statistics off)
            RawIn nx;
            if (do1) nx = load_raw(r0 + j);                        // input row j+2, needed by the NEXT iteration
            __builtin_amdgcn_sched_barrier(0);
            float l1[4], d1[4], l2[4], d2[4];
            TieT tie1[4], tie2[4];
            float sm1 = 0.f, sm2 = 0.f;
            if (do2)
                row_cells(P2, SL[u], SL[(u + 1) % 3], SL[(u + 2) % 3], SD[u], SD[(u + 1) % 3], SD[(u + 2) % 3], l2, d2, tie2,
                          use2, st_on ? &sm2 : nullptr);
            if (do1)
                row_cells(P1, IL[(u + 2) % 3], IL[u], IL[(u + 1) % 3], ID[(u + 2) % 3], ID[u], ID[(u + 1) % 3], l1, d1, tie1,
                          need1m, st_on ? &sm1 : nullptr);
            if (do2) {
                row_queue(SL[u], SL[(u + 1) % 3], SL[(u + 2) % 3], SD[u], SD[(u + 1) % 3], SD[(u + 2) % 3], l2, d2, tie2, 2,
                          j - 2);
                if (st_on) stats_step2(sm2);
                if (writes) {
                    const size_t off = woff + (size_t)(r0 + j - 4) * G.W + col;
                    stream_store4(outL + off, make_float4(l2[0], l2[1], l2[2], l2[3]));
                    stream_store4(outD + off, make_float4(d2[0], d2[1], d2[2], d2[3]));
                }
            }
            if (do1) {
                row_queue(IL[(u + 2) % 3], IL[u], IL[(u + 1) % 3], ID[(u + 2) % 3], ID[u], ID[(u + 1) % 3], l1, d1, tie1, 1, j);
                if (st_on) stats_step1(sm1, l1, d1, tie1, j >= 2 && j <= nr + 1);
                if constexpr (RING) {                               // both new rows in ONE exchange (one barrier per iteration)
                    const float4 v[4] = {make_float4(l1[0], l1[1], l1[2], l1[3]), make_float4(d1[0], d1[1], d1[2], d1[3]),
                                         widen4(nx.l), widen4(nx.d)};
                    Row4* const o[4] = {&SL[u], &SD[u], &IL[(u + 2) % 3], &ID[(u + 2) % 3]};
                    rows_ring(v, o, std::integral_constant<int, 4>{});
                } else {
                    to_rows4(make_float4(l1[0], l1[1], l1[2], l1[3]), make_float4(d1[0], d1[1], d1[2], d1[3]), SL[u], SD[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    to_rows4(widen4(nx.l), widen4(nx.d), IL[(u + 2) % 3], ID[(u + 2) % 3]);   // input row j+2 replaces input row j-1
                    if (!PACK) pin_edge_sums(IL[(u + 2) % 3], ID[(u + 2) % 3]);
                }
            }
        };
        using Yes = std::true_type;
        using No = std::false_type;
        iter(U1{}, Yes{}, No{}, 1, Yes{});                          // nr >= 1: rows 1..3 always exist
        iter(U2{}, Yes{}, No{}, 2, Yes{});
        iter(U0{}, Yes{}, No{}, 3, Yes{});
        const int jend = nr + 2;
        int j = 4;
        for (; j + 2 <= jend; j += 3) {                             // j % 3 == 1 at the top
            iter(U1{}, Yes{}, Yes{}, j, Yes{});
            iter(U2{}, Yes{}, Yes{}, j + 1, Yes{});
            iter(U0{}, Yes{}, Yes{}, j + 2, Yes{});
            // wave-uniform and rare (see sweep_queue); not in the STATS variants, which are at their register budget: their
            // live values would spill on the main path (they keep one sweep at the end of the strip)
            if (EXACT && !STATS && __builtin_expect(nq >= (unsigned)flush_at, 0)) sweep_queue();
            if (CHEAP && st_sure1 && st_c2w >= (unsigned int)G.sure_need) { j += 3; break; }   // both predicates are decided: see st_sure1
        }
        if constexpr (CHEAP) {
            // the rest of the strip with the statistics switched off: a second copy of the row loop rather than a
            // branch around the statistics in the first (that one costs the exact kernels registers they do not have)
            for (; j + 2 <= jend; j += 3) {
                iter(U1{}, Yes{}, Yes{}, j, No{});
                iter(U2{}, Yes{}, Yes{}, j + 1, No{});
                iter(U0{}, Yes{}, Yes{}, j + 2, No{});
            }
        }
        if (j <= jend) { iter(U1{}, Yes{}, Yes{}, j, Yes{}); ++j; }   // (the last rows: statistics on again - they only grow)
        if (j <= jend) { iter(U2{}, Yes{}, Yes{}, j, Yes{}); ++j; }
        // j == nr + 3: the last output row
        if (j % 3 == 1) iter(U1{}, No{}, Yes{}, j, Yes{});
        else if (j % 3 == 2) iter(U2{}, No{}, Yes{}, j, Yes{});
        else iter(U0{}, No{}, Yes{}, j, Yes{});
    } else {
        // iteration j = 1 .. nr+2: step-1 row j (grid row r0-2+j) from input rows j-1, j, j+1; then, from j = 3
        // on, output row k = j-3 (local row j-1) from step-1 rows j-2, j-1, j
        auto iter = [&](auto U, int j) {
            constexpr int u = decltype(U)::value;                  // u == j % 3
            const RawIn nx = load_raw(r0 + j);                       // input row j+2, needed by the NEXT iteration
            __builtin_amdgcn_sched_barrier(0);
            float4 l1, d1;
            float sm = 0.f;
            row_map(P1, IL[(u + 2) % 3], IL[u], IL[(u + 1) % 3], ID[(u + 2) % 3], ID[u], ID[(u + 1) % 3], l1, d1, 1, j, need1m,
                    STATS ? &sm : nullptr);
            if (STATS && writes && j >= 2 && j <= nr + 1) st_m1 = fmaxf(st_m1, sm);   // step-1 rows of MY output cells
            to_rows4(l1, d1, SL[u], SD[u]);                        // step-1 row j replaces step-1 row j-3
            if (j >= 3) {
                float4 l2, d2;
                row_map(P2, SL[(u + 1) % 3], SL[(u + 2) % 3], SL[u], SD[(u + 1) % 3], SD[(u + 2) % 3], SD[u], l2, d2, 2, j - 1,
                        use2, STATS ? &sm : nullptr);
                if (STATS && writes && sm > thr_hi) st_c2 += 1u;
                if (writes) {
                    const size_t off = woff + (size_t)(r0 + j - 3) * G.W + col;
                    stream_store4(outL + off, l2);
                    stream_store4(outD + off, d2);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            to_rows4(widen4(nx.l), widen4(nx.d), IL[(u + 2) % 3], ID[(u + 2) % 3]);   // input row j+2 replaces input row j-1
        };
        const int jend = nr + 2;
        int j = 1;
        for (; j + 2 <= jend; j += 3) {                             // j % 3 == 1 at the top
            iter(U1{}, j);
            iter(U2{}, j + 1);
            iter(U0{}, j + 2);
        }
        if (j <= jend) iter(U1{}, j);
        if (j + 1 <= jend) iter(U2{}, j + 1);
    }

    if (EXACT) {
        PhysF64 Pa = P64, Pb = P64;
        Pa.L = La;
        Pb.L = Lb;
        // RING: a mismatch of a step-1 cell in the wave's first / last column also feeds output cells of the ADJACENT
        // wave, and this wave repairs them (F3 / the fallback below): every row store of the workgroup must have
        // completed before any patch store, and every F2 patch (which may rest on a neighbour's mismatched value)
        // before any F3 patch - two workgroup barriers, each behind a wait for the wave's own stores.
        if (RING) {
            wait_row_stores_before_patching();
            __syncthreads();
        }
        sweep_queue();                                          // what the strip's last rows queued
        if (nmm > (unsigned)G.mcap) redo = true;
        if (RING) {
            wait_row_stores_before_patching();
            __syncthreads();
        }
        if (!redo) {
            // F3: everything that depends on a step-1 mismatch, entirely in float64 from the inputs.  Two
            // mismatches per pass: 2 x 25 lanes evaluate step 1 on the 5x5 block around their mismatch
            // (exchanged through the wave's - by now consumed - queue memory), then 2 x 9 lanes evaluate
            // step 2 on the 3x3 block of dependents.  One float64 evaluation per lane and stage instead of
            // ten in a row on nine lanes.
            unsigned int* s1 = reinterpret_cast<unsigned int*>(q);
            for (unsigned int m0 = 0; m0 < nmm; m0 += 2) {
                __builtin_amdgcn_wave_barrier();
                {
                    const unsigned int mi = lane / 25u, t = lane - mi * 25u;
                    if (lane < 50 && m0 + mi < nmm) {
                        const unsigned int where = mm[m0 + mi];
                        const int lrow = (int)(where >> 16) + (int)(t / 5u) - 2;
                        const int lcm = (int)(where & 0xffffu), dx5 = (int)(t % 5u) - 2;
                        if (PACK) {                         // columns wrap inside the mismatch's own world
                            const size_t wo = world_off(lcm);
                            s1[lane] = exact1_word(inL + wo, inD + wo, G.H, G.W, r0 - 2 + lrow,
                                                   lcm - ((lcm >> 2) / G.lpw) * G.W + dx5, Pa);
                        } else {
                            s1[lane] = exact1_word(pL, pD, G.H, G.W, r0 - 2 + lrow, c00 + lcm + dx5, Pa);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
                    const unsigned int mi = lane / 9u, t = lane - mi * 9u;
                    if (lane < 18 && m0 + mi < nmm) {
                        const unsigned int where = mm[m0 + mi];
                        const int dy = (int)(t / 3u) - 1, dx = (int)(t % 3u) - 1;
                        const int lrow = (int)(where >> 16) + dy;
                        int lc = (int)(where & 0xffffu) + dx;
                        if (PACK) {
                            const int lcm = (int)(where & 0xffffu), base = ((lcm >> 2) / G.lpw) * G.W;
                            lc = base + (lcm - base + dx + G.W) % G.W;
                        } else if (ROT) {
                            lc = (lc + 256) & 255;
                        }
                        if (is_output(lrow, lc)) {
                            unsigned int w2[9];
#pragma unroll
                            for (int a = 0; a < 3; ++a)
#pragma unroll
                                for (int e = 0; e < 3; ++e) w2[a * 3 + e] = s1[mi * 25u + (unsigned)((1 + dy + a) * 5 + (1 + dx + e))];
                            const NewCoverF64 o = cell_f64_lean(Pb, w2);
                            int gr, gc;
                            size_t wo;
                            locate(lrow, lc, wo, gr, gc);
                            const size_t off = wo + (size_t)gr * G.W + gc;
                            outL[off] = (TO)(float)dw_round3_k(o.nl);
                            outD[off] = (TO)(float)dw_round3_k(o.nd);
                        }
                    }
                }
            }
        } else {
            // overflow fallback: every output cell of the strip, two float64 steps from the inputs
            // (RING: plus the column on either side, whose step-2 values depend on this wave's - unverified - step-1 edge)
            const int ncol = ROT ? 256 : (RING ? 258 : 248);
            for (int i = lane; i < nr * ncol; i += 64) {
                const int lrow = 2 + i / ncol, lc = (ROT ? 0 : (RING ? -1 : 4)) + i % ncol;
                if (!is_output(lrow, lc)) continue;
                int gr, gc;
                size_t wo;
                locate(lrow, lc, wo, gr, gc);
                float kl, kd;
                exact2_cell(inL + wo, inD + wo, G.H, G.W, gr, gc, Pa, Pb, kl, kd);
                const size_t off = wo + (size_t)gr * G.W + gc;
                outL[off] = (TO)kl;
                outD[off] = (TO)kd;
                if (STATS) {                                     // step-1 maximum from scratch; no step-2 count
                    const unsigned int w1 = exact1_word(inL + wo, inD + wo, G.H, G.W, gr, gc, Pa);
                    atomicMax(&pstats[2 * world_of(lc)], (w1 & 0xffffu) > (w1 >> 16) ? (w1 & 0xffffu) : (w1 >> 16));
                }
            }
            st_m1 = 0.f;
            st_x1 = 0.f;
            st_c2 = 0;
            st_c2w = 0;
        }
        st_nmm = nmm;
    }
    if (STATS) {
        // This wave's contribution.  The counted row groups hold a non-tie step-2 value > thr_hi, i.e. a
        // value whose float32 rounding is the exact one - unless a step-1 input was a mismatch: those cells
        // are the <= 9 per mismatch that F3 rewrote, so 9 * nmm groups are deducted (a sound lower bound).
        if (PACK) {
            float m = st_m1;
            unsigned int c = st_c2;
            if ((G.lpw & (G.lpw - 1)) == 0) {
                for (int o = G.lpw >> 1; o > 0; o >>= 1) {
                    m = fmaxf(m, __shfl_xor(m, o, 64));
                    c += (unsigned int)__shfl_xor((int)c, o, 64);
                }
            } else {
                for (int o = 1; o < G.lpw; ++o) {
                    const int src = min(lane + o, 63);
                    m = fmaxf(m, __shfl(st_m1, src, 64));
                    c += (unsigned int)__shfl((int)st_c2, src, 64);
                }
            }
            c = c > 9u * st_nmm ? c - 9u * st_nmm : 0u;
            if (pj == 0 && writes) {
                atomicMax(&pstats[2 * (b * G.wpr + pw)], (unsigned int)m);
                if (c) atomicAdd(&pstats[2 * (b * G.wpr + pw) + 1], c);
            }
        } else {
            // st_m1 >= thr + 2 somewhere in the strip: alive, whatever the near-ties were; otherwise the exact values
            // decide (float32-only mode: the float32 values ARE the values)
            const float mf = wave_max(st_m1), mx = wave_max(st_x1);
            const float m = (EXACT && !(mf > thr_hi + 1.0f)) ? mx : mf;
            unsigned int c = st_c2w;                             // already the wave's total
            c = c > 9u * st_nmm ? c - 9u * st_nmm : 0u;
            if (lane == 0) {
                atomicMax(&pstats[2 * b], (unsigned int)m);
                if (c) atomicAdd(&pstats[2 * b + 1], c);
            }
        }
    }
}

// Waves per SIMD the float32-only fused kernels are planned for (the register allocator's budget: 128 VGPRs at 4)
#ifndef DW_FUSED_FAST_WAVES
#define DW_FUSED_FAST_WAVES 4
#endif
// (the ring variant that also reduces the per-step world flags would spill inside its row loop at 128: it keeps 3;
// so do the packed variants - 130-134 VGPRs without any scratch, and no measurable difference to 4 waves)
template <int MODE, bool PACK, bool STATS>
constexpr int fused_fast_waves() { return ((MODE == kFusedRing && STATS) || PACK) ? 3 : DW_FUSED_FAST_WAVES; }
template <int MODE, bool PACK = false, bool STATS = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(fused_fast_waves<MODE, PACK, STATS>(), fused_fast_waves<MODE, PACK, STATS>())))
void step_stream_fused2(const plane_t* __restrict__ inL, const plane_t* __restrict__ inD,
                                                          plane_t* __restrict__ outL, plane_t* __restrict__ outD,
                                                          FusedGeom G, PhysF32 P1, PhysF32 P2,
                                                          unsigned long long* __restrict__ zero_me, int zero_n,
                                                          unsigned int* __restrict__ pstats, float thr_hi) {
    const PhysF64 dummy{};
    const double zero = 0.0;
    fused2_body<MODE, false, PACK, STATS>(inL, inD, outL, outD, G, P1, P2, dummy, zero, zero, zero_me, zero_n, pstats, thr_hi);
}

// Waves per SIMD of the exact kernels: the plain variants fit 3 waves/SIMD (151-161 VGPRs), and since the constant
// pairs so do the STATS variants of the overlapped and rotating strips (157 / 168 VGPRs, no scratch in the row loop);
// so does the packed variant without STATS (160); the packed STATS variant and the ring's STATS variant would
// spill inside the loop there and stay at 2.
#ifndef DW_EXACT_PACK_WAVES
#define DW_EXACT_PACK_WAVES 3                                      // tuning builds: 2
#endif
template <int MODE, bool PACK, bool STATS>
constexpr int fused_exact_waves() {
    return (PACK && STATS) || (STATS && MODE == kFusedRing) ? 2 : (PACK ? DW_EXACT_PACK_WAVES : 3);
}
struct FusedExactArgs {
    const plane_t* inL; const plane_t* inD; plane_t* outL; plane_t* outD;
    FusedGeom G;
    PhysF32 P1; PhysLumF32 lum2;                                  // step 2 = P1 with these members replaced:
                                                                  // 15 shared constants instead of 2 x 23 (each
                                                                  // one occupies an SGPR PAIR as a packed operand)
    unsigned long long* zero_me; int zero_n;
    unsigned int* pstats; float thr_hi;                           // STATS variants only
    PhysF64 P64; double La; double Lb;                            // cold (see kernarg_struct)
};

template <int MODE, bool PACK = false, bool STATS = false, bool SYM = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(fused_exact_waves<MODE, PACK, STATS>(), fused_exact_waves<MODE, PACK, STATS>())))
void step_stream_fused2_exact(FusedExactArgs A) {
    const FusedExactArgs& cold = kernarg_struct<FusedExactArgs>();
    const PhysF32 P2 = with_lum(A.P1, A.lum2);
    fused2_body<MODE, true, PACK, STATS, SYM>(A.inL, A.inD, A.outL, A.outD, A.G, A.P1, P2, cold.P64, cold.La, cold.Lb, A.zero_me,
                                             A.zero_n, A.pstats, A.thr_hi);
}

}  // namespace dw
