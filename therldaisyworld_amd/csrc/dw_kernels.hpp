// dw_kernels.hpp — HIP kernels of the RLDaisyWorld hot path for gfx950 (MI355X).
//
//   step_tiled      the hot kernel: fused 3x3 toroidal stencil + albedo/temperature/growth reaction
//                   + 3-decimal quantiser + per-world reductions, LDS-staged tile with halo.
//   step_generic    one thread per cell, any grid shape, float64 or float32: the in-library
//                   reference, the first step from an un-quantised state, odd shapes.
//   agents_update   ref update_agents (daisy_world_rl.py:181-244)
//   observe         ref get_obs (:246-263) + the channel values forward() would have written
//   materialise     ref self.grid after forward() (:445-459) / initialize_grid (:304-323)
//   policy_greedy   ref Greedy.__call__ deterministic branch (agents/greedy.py:14-36)
//   init_random     synthetic initial state (ref :285-302, :173-179) from Philox4x32-10
//
// Wavefront = 64 lanes, 256-thread workgroups (4 waves), no MFMA: the step is HBM-bound
// (16 algorithmic bytes per cell-update, ~90 VALU issue slots per cell).
#pragma once
#include <type_traits>

#include "dw_physics.hpp"

namespace dw {

struct StatsDev {             // mirrors dw_world_stats
    unsigned int max_k;
    unsigned int reserved;
    unsigned long long sum_l;
    unsigned long long sum_d;
};

struct Geom {
    int B, H, W;
    int Wq;                   // W / 4 (tiled kernel only)
    int tiles_r, tiles_c;     // tiles per world
    int ntiles;               // B * tiles_r * tiles_c
    int chunk;                // ceil(ntiles / 8): tiles per XCD
    int qcap;                 // near-tie LDS queue capacity in use (<= kMaxFix; tests shrink it)
};

// ---------------------------------------------------------------------------------------------
// wave / workgroup reductions (wavefront shuffles, 64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// input adaptors: natural-unit float64 planes, or per-mille float32 planes
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double to_natural(double x) { return x; }
__device__ __forceinline__ double to_natural(float k) { return (double)k / 1000.0; }
__device__ __forceinline__ float to_permille(double x) { return (float)(x * 1000.0); }
__device__ __forceinline__ float to_permille(float k) { return k; }

template <typename T>
__device__ __forceinline__ void gather9(const T* __restrict__ plane, int H, int W, int r, int c,
                                        double out[9]) {
    const int ru = r == 0 ? H - 1 : r - 1, rd = r == H - 1 ? 0 : r + 1;
    const int cl = c == 0 ? W - 1 : c - 1, cr = c == W - 1 ? 0 : c + 1;
    const int rows[3] = {ru, r, rd}, cols[3] = {cl, c, cr};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) out[a * 3 + b] = to_natural(plane[(size_t)rows[a] * W + cols[b]]);
}

// ---------------------------------------------------------------------------------------------
// step_generic: grid = (ceil(H*W/256), B), block = 256.  PREC: 0 exact, 1 fast, 2 f64.
// ---------------------------------------------------------------------------------------------
template <typename InT, int PREC>
__global__ __launch_bounds__(256) void step_generic(const InT* __restrict__ inL,
                                                    const InT* __restrict__ inD,
                                                    float* __restrict__ outL,
                                                    float* __restrict__ outD, int H, int W,
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
            const float Cl = (DW_AT(pl, ru, cl) + DW_AT(pl, rd, cl)) + (DW_AT(pl, ru, cr) + DW_AT(pl, rd, cr));
            const float Ed = (DW_AT(pd, ru, c) + DW_AT(pd, rd, c)) + (DW_AT(pd, r, cl) + DW_AT(pd, r, cr));
            const float Cd = (DW_AT(pd, ru, cl) + DW_AT(pd, rd, cl)) + (DW_AT(pd, ru, cr) + DW_AT(pd, rd, cr));
#undef DW_AT
            const GrowthF32 g = growth_f32<PREC != 1>(P, li, di, El, Cl, Ed, Cd);
            if (PREC == 1) {
                kl = finish_fast(li, g.gql);
                kd = finish_fast(di, g.gqd);
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
        outL[woff + cell] = kl;
        outD[woff + cell] = kd;
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

// ---------------------------------------------------------------------------------------------
// step_tiled — the hot kernel.
//
// One 256-thread workgroup updates a tile of TR = (256/TCQ)*RPT rows x 4*TCQ columns of one
// world.  Phase 1 stages the tile plus a one-cell toroidal halo (one row above/below, one 16-byte
// column group left/right; wrap resolved here) of both planes into LDS with coalesced 16-byte
// loads.  Phase 2: each thread owns 4 adjacent columns and walks RPT rows with a 3-row register
// window; per row it needs three ds_read_b128 per plane (its group and the two neighbours).
// Phase 3 (exact mode): cells whose float32 pre-rounding value is within the error bound of a
// rounding tie were queued in LDS; they are re-evaluated in float64 from the LDS tile and patched
// in global memory.  Per-world max/sums are reduced with wavefront shuffles and LDS atomics and
// leave the workgroup as three global atomics.
//
// Workgroup -> tile mapping is XCD-aware: hardware deals consecutive workgroup ids round-robin to
// the 8 XCDs, so id b works on tile (b % 8) * chunk + b / 8: each XCD (and its private L2) gets a
// contiguous run of tiles, and the halo rows shared by vertically adjacent tiles hit in L2.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxFix = 1024;     // per-workgroup LDS queue of near-tie cells
constexpr int kNumQueues = 256;   // global queues (one counter cache line each)

// Global queues of near-tie cells (exact mode).  A workgroup reserves a contiguous run in queue
// (blockIdx % kNumQueues) with ONE atomic and copies its LDS queue there, each entry carrying the
// cell's whole 3x3 neighbourhood (read from the LDS tile), so that the dense `fixup_cells` kernel
// that follows is pure float64 arithmetic with no gathers.  counts[q * 16] is queue q's fill level
// (16 uints = one 64-byte line per counter).  Entry = 3 x uint4 (48 bytes):
//   w0 world, w1 row << 16 | col, w2..w10 the nine (light | dark << 16) per-mille pairs row-major,
//   w11 the float32 result (light' | dark' << 16) that the main kernel stored.
// A tile whose queue overflows (LDS or global) is not patched cell by cell: its id goes to
// `redo_tiles` and `redo_tiles_f64` recomputes the whole tile in float64.
struct FixQ {
    uint4* entries;               // [kNumQueues][qcap][3]
    unsigned int* counts;         // [kNumQueues * 16] then [16]: redo count; zero at kernel start
    unsigned int qcap;
    int* redo_tiles;              // [ntiles]
};

template <int TCQ, int RPT>
struct TileCfg {
    static constexpr int RG = 256 / TCQ;          // row groups per workgroup
    static constexpr int TR = RG * RPT;           // tile rows
    static constexpr int LROWS = TR + 2;          // + halo rows
    static constexpr int LQ = TCQ + 2;            // float4 groups per LDS row (+ halo groups)
    static constexpr int LSTRIDE = LQ * 4;        // floats per LDS row
    static constexpr int PLANE = LROWS * LSTRIDE; // floats per plane
    static constexpr int STAGE_ITERS = (LROWS * LQ + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)2 * PLANE * sizeof(float);
};

struct Row4 {                 // 4 centre values of a row and the horizontal pair sums around them
    float x[4];
    float h2[4];              // h2[i] = x[i-1] + x[i+1]
};

__device__ __forceinline__ Row4 load_row(const float* __restrict__ lds_row_group) {
    // lds_row_group points at the float4 group LEFT of the thread's own group
    const float4 a = *reinterpret_cast<const float4*>(lds_row_group);
    const float4 m = *reinterpret_cast<const float4*>(lds_row_group + 4);
    const float4 c = *reinterpret_cast<const float4*>(lds_row_group + 8);
    Row4 r;
    r.x[0] = m.x; r.x[1] = m.y; r.x[2] = m.z; r.x[3] = m.w;
    r.h2[0] = a.w + m.y;
    r.h2[1] = m.x + m.z;
    r.h2[2] = m.y + m.w;
    r.h2[3] = m.z + c.x;
    return r;
}

// The map on the four cells of one row group: (up, mid, down) rows of both planes -> new values (and,
// in the exact mode, the near-tie flags).  Two cells per packed float32 lane pair (dw_physics.hpp).
template <bool EXACT>
__device__ __forceinline__ void cells4(const PhysF32& P, const Row4& upL, const Row4& miL, const Row4& dnL,
                                       const Row4& upD, const Row4& miD, const Row4& dnD, float* ol, float* od,
                                       bool* tie) {
#pragma clang fp contract(off)
#ifdef DW_SCALAR_CELLS
    using T = float;
#else
    using T = dw_f32x2;
#endif
    constexpr int N = Lanes<T>::N;
#pragma unroll
    for (int i = 0; i < 4; i += N) {
        auto pr = [&](const float* a) -> T { return Lanes<T>::load(a, i); };
        const T li = pr(miL.x), di = pr(miD.x);
        const T El = pr(miL.h2) + (pr(upL.x) + pr(dnL.x));
        const T Cl = pr(upL.h2) + pr(dnL.h2);
        const T Ed = pr(miD.h2) + (pr(upD.x) + pr(dnD.x));
        const T Cd = pr(upD.h2) + pr(dnD.h2);
        const GrowthT<T> g = growth_t<EXACT, T>(P, li, di, El, Cl, Ed, Cd);
        T vl, vd;
        if (EXACT) {
            bool tl[N], td[N];
            vl = finish_exact_t<T>(P, li, g.gql, g.dKl, g.oml, tl);
            vd = finish_exact_t<T>(P, di, g.gqd, g.dKd, g.omd, td);
#pragma unroll
            for (int e = 0; e < N; ++e) tie[i + e] = tl[e] || td[e];
        } else {
            vl = finish_fast_t<T>(li, g.gql);
            vd = finish_fast_t<T>(di, g.gqd);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            ol[i + e] = Lanes<T>::get(vl, e);
            od[i + e] = Lanes<T>::get(vd, e);
        }
    }
}

#ifdef DW_TUNING
// plain streaming copy of both planes: the achievable-bandwidth yardstick for this traffic shape
__global__ __launch_bounds__(256) void copy_planes(const float4* __restrict__ inL, const float4* __restrict__ inD,
                                                   float4* __restrict__ outL, float4* __restrict__ outD, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { outL[i] = inL[i]; outD[i] = inD[i]; }
}
__device__ int g_ablate;   // 0 normal, 1 skip the arithmetic (stage -> LDS -> registers -> store)
#endif

template <int TCQ, int RPT, bool EXACT>
__global__ __launch_bounds__(256) void step_tiled(const float* __restrict__ inL,
                                                  const float* __restrict__ inD,
                                                  float* __restrict__ outL,
                                                  float* __restrict__ outD, Geom G, PhysF32 P,
                                                  StatsDev* __restrict__ stats,
                                                  unsigned long long* __restrict__ fixups,
                                                  unsigned long long* __restrict__ zero_me,
                                                  int zero_n, FixQ fq) {
    using C = TileCfg<TCQ, RPT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ uint2 s_fix[EXACT ? kMaxFix : 1];      // {LDS row << 16 | LDS col, light' | dark' << 16}
    __shared__ unsigned int s_nfix, s_max, s_suml, s_sumd, s_base;

    // ---- XCD-aware tile id ----
    const int bid = blockIdx.x;
    const int t = (bid & 7) * G.chunk + (bid >> 3);
    if (t >= G.ntiles) return;                      // uniform for the whole workgroup
    const int tiles_per_world = G.tiles_r * G.tiles_c;
    const int b = t / tiles_per_world;
    const int tw = t - b * tiles_per_world;
    const int tr = tw / G.tiles_c, tc = tw - tr * G.tiles_c;
    const int r0 = tr * C::TR, q0 = tc * TCQ;
    const int nrows = min(C::TR, G.H - r0);
    const int nq = min(TCQ, G.Wq - q0);
    const int tid = threadIdx.x;
    const size_t woff = (size_t)b * G.H * G.W;

    if (tid == 0) { s_nfix = 0; s_max = 0; s_suml = 0; s_sumd = 0; }
    if (t == 0)                                     // clear the reduction buffer of the NEXT step
        for (int i = tid; i < zero_n; i += 256) zero_me[i] = 0ull;

    // ---- phase 1: stage tile + halo into LDS (all loads issued before the first LDS write) ----
    {
        const int lrows = nrows + 2, lq = nq + 2;
        float4 vl[C::STAGE_ITERS], vd[C::STAGE_ITERS];
#pragma unroll
        for (int it = 0; it < C::STAGE_ITERS; ++it) {
            const int idx = it * 256 + tid;
            // out-of-range slots are clamped onto a valid address (their value is never stored)
            const int j = min(idx / C::LQ, lrows - 1), gq = min(idx % C::LQ, lq - 1);
            int rr = r0 - 1 + j;
            rr = rr < 0 ? rr + G.H : (rr >= G.H ? rr - G.H : rr);
            int gg = q0 - 1 + gq;
            gg = gg < 0 ? gg + G.Wq : (gg >= G.Wq ? gg - G.Wq : gg);
            const size_t off = woff + (size_t)rr * G.W + (size_t)gg * 4;
            vl[it] = *reinterpret_cast<const float4*>(inL + off);
            vd[it] = *reinterpret_cast<const float4*>(inD + off);
        }
#pragma unroll
        for (int it = 0; it < C::STAGE_ITERS; ++it) {
            const int idx = it * 256 + tid;
            const int j = idx / C::LQ, gq = idx - j * C::LQ;
            if (j < lrows && gq < lq) {
                *reinterpret_cast<float4*>(lds + j * C::LSTRIDE + gq * 4) = vl[it];
                *reinterpret_cast<float4*>(lds + C::PLANE + j * C::LSTRIDE + gq * 4) = vd[it];
            }
        }
    }
    __syncthreads();

    // ---- phase 2: 4 columns x RPT rows per thread, 3-row register window ----
    const int cq = tid % TCQ, rg = tid / TCQ;
    const int jr0 = rg * RPT;
    float acc_max = 0.f, acc_l = 0.f, acc_d = 0.f;
    if (cq < nq && jr0 < nrows) {
        const float* baseL = lds + cq * 4;               // group left of mine, LDS row 0
        const float* baseD = baseL + C::PLANE;
        Row4 upL = load_row(baseL + (jr0 + 0) * C::LSTRIDE);
        Row4 upD = load_row(baseD + (jr0 + 0) * C::LSTRIDE);
        Row4 miL = load_row(baseL + (jr0 + 1) * C::LSTRIDE);
        Row4 miD = load_row(baseD + (jr0 + 1) * C::LSTRIDE);
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const int row = jr0 + rr;                    // tile-local row; LDS row = row + 1
            if (row < nrows) {
                const Row4 dnL = load_row(baseL + (row + 2) * C::LSTRIDE);
                const Row4 dnD = load_row(baseD + (row + 2) * C::LSTRIDE);
                float ol[4], od[4];
                bool tie4[4] = {false, false, false, false};
                unsigned int ties = 0;
                cells4<EXACT>(P, upL, miL, dnL, upD, miD, dnD, ol, od, tie4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#ifdef DW_TUNING
                    if (g_ablate == 1) {
                        ol[i] = (miL.h2[i] + upL.x[i] + dnL.x[i]) + (upL.h2[i] + dnL.h2[i]);
                        od[i] = (miD.h2[i] + upD.x[i] + dnD.x[i]) + (upD.h2[i] + dnD.h2[i]);
                        continue;
                    }
#endif
                    if (EXACT) {
                        const bool tie = tie4[i];
                        ties |= (tie ? 1u : 0u) << i;
                        // sums take the float32 value (the fix-up kernel adds the correction);
                        // the max cannot be corrected afterwards, so near-tie cells stay out of it
                        acc_l += ol[i]; acc_d += od[i];
                        acc_max = fmaxf(acc_max, tie ? 0.f : fmaxf(ol[i], od[i]));
                    } else {
                        acc_l += ol[i]; acc_d += od[i];
                        acc_max = fmaxf(acc_max, fmaxf(ol[i], od[i]));
                    }
                }
                const size_t off = woff + (size_t)(r0 + row) * G.W + (size_t)(q0 + cq) * 4;
                *reinterpret_cast<float4*>(outL + off) = make_float4(ol[0], ol[1], ol[2], ol[3]);
                *reinterpret_cast<float4*>(outD + off) = make_float4(od[0], od[1], od[2], od[3]);
                if (EXACT && ties) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (ties & (1u << i)) {
                            const unsigned int slot = atomicAdd(&s_nfix, 1u);
                            if (slot < (unsigned)G.qcap)
                                s_fix[slot] = make_uint2(((unsigned)(row + 1) << 16) | (unsigned)((cq + 1) * 4 + i),
                                                         (unsigned)ol[i] | ((unsigned)od[i] << 16));
                        }
                    }
                }
                upL = miL; upD = miD; miL = dnL; miD = dnD;
            }
        }
    }

    // ---- per-world reductions: wavefront shuffles, then LDS atomics ----
    {
        const float m = wave_max(acc_max);
        const float sl = wave_sum(acc_l), sd = wave_sum(acc_d);
        if ((tid & 63) == 0) {
            atomicMax(&s_max, (unsigned int)m);
            atomicAdd(&s_suml, (unsigned int)sl);
            atomicAdd(&s_sumd, (unsigned int)sd);
        }
    }

    // ---- phase 3 (exact mode): hand the queued near-tie cells to the float64 fix-up kernel ----
    bool redo = false;
    if (EXACT) {
        __syncthreads();   // all pushes done
        const unsigned int n = s_nfix;
        const unsigned int q = (unsigned)bid % (unsigned)kNumQueues;
        if (tid == 0) s_base = (n && n <= (unsigned)G.qcap) ? atomicAdd(&fq.counts[q * 16], n) : 0u;
        __syncthreads();
        const unsigned int base = s_base;
        redo = n > (unsigned)G.qcap || base + n > fq.qcap;        // uniform for the workgroup
        if (!redo) {
            for (unsigned int e = tid; e < n; e += 256) {
                const uint2 ent = s_fix[e];
                const int j = (int)(ent.x >> 16), col = (int)(ent.x & 0xffffu);
                unsigned int w[9];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int o = (j - 1 + a) * C::LSTRIDE + (col - 1 + c);
                        w[a * 3 + c] = (unsigned)lds[o] | ((unsigned)lds[C::PLANE + o] << 16);
                    }
                uint4* dst = fq.entries + ((size_t)q * fq.qcap + base + e) * 3;
                dst[0] = make_uint4((unsigned)b, ((unsigned)(r0 + j - 1) << 16) | (unsigned)(q0 * 4 + col - 4), w[0], w[1]);
                dst[1] = make_uint4(w[2], w[3], w[4], w[5]);
                dst[2] = make_uint4(w[6], w[7], w[8], ent.y);
            }
        } else if (tid == 0) {
            // the reservation (if any) stays in the queue as garbage: mark it so fixup_cells skips it
            if (n <= (unsigned)G.qcap)
                for (unsigned int e = 0; e < n && base + e < fq.qcap; ++e)
                    fq.entries[((size_t)q * fq.qcap + base + e) * 3] = make_uint4(0xffffffffu, 0u, 0u, 0u);
            fq.redo_tiles[atomicAdd(&fq.counts[kNumQueues * 16], 1u)] = t;
        }
    }
    __syncthreads();
    if (tid == 0 && !redo) {
        atomicMax(&stats[b].max_k, s_max);
        atomicAdd(&stats[b].sum_l, (unsigned long long)s_suml);
        atomicAdd(&stats[b].sum_d, (unsigned long long)s_sumd);
        if (EXACT && s_nfix) atomicAdd(fixups, (unsigned long long)s_nfix);
    }
}

// ---------------------------------------------------------------------------------------------
// fixup_cells — exact mode, second kernel of a step: dense float64 re-evaluation of the near-tie
// cells queued by step_tiled.  grid = (ceil(qcap/256), kNumQueues).  Every entry carries its 3x3
// neighbourhood, so this is pure arithmetic: patch the two new planes, correct the per-world sums
// by (float64 result - float32 result) and contribute to the per-world max (the main kernel kept
// near-tie cells out of the max).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fixup_cells(float* __restrict__ outL, float* __restrict__ outD, int H, int W,
                                                   PhysF64 P64, StatsDev* __restrict__ stats, FixQ fq) {
    const unsigned int q = blockIdx.y;
    const unsigned int n = min(fq.counts[q * 16], fq.qcap);
    const unsigned int e = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;                       // uniform per workgroup
    bool active = e < n;
    int world = -1;
    float kl = 0.f, kd = 0.f, dl = 0.f, dd = 0.f;
    if (active) {
        const uint4* src = fq.entries + ((size_t)q * fq.qcap + e) * 3;
        const uint4 e0 = src[0], e1 = src[1], e2 = src[2];
        active = e0.x != 0xffffffffu;                        // slot abandoned by a tile that is redone whole
        if (active) {
            world = (int)e0.x;
            const int r = (int)(e0.y >> 16), c = (int)(e0.y & 0xffffu);
            const unsigned int w[9] = {e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z};
            const NewCoverF64 o = cell_f64_lean(P64, w);
            kl = (float)dw_round3_k(o.nl);
            kd = (float)dw_round3_k(o.nd);
            dl = kl - (float)(e2.w & 0xffffu);
            dd = kd - (float)(e2.w >> 16);
            const size_t off = (size_t)world * H * W + (size_t)r * W + c;
            outL[off] = kl;
            outD[off] = kd;
        }
    }
    // per-world reductions: entries of one tile are contiguous, so a wave holds few distinct worlds
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int w = __shfl(world, leader, 64);
        const bool mine = active && world == w;
        const float m = wave_max(mine ? fmaxf(kl, kd) : 0.f);
        const float sl = wave_sum(mine ? dl : 0.f), sd = wave_sum(mine ? dd : 0.f);
        if ((int)(threadIdx.x & 63) == leader) {
            atomicMax(&stats[w].max_k, (unsigned int)m);
            atomicAdd(&stats[w].sum_l, (unsigned long long)(long long)sl);   // two's complement: may be negative
            atomicAdd(&stats[w].sum_d, (unsigned long long)(long long)sd);
        }
        todo &= ~__ballot(mine);
    }
}

// redo_tiles_f64 — exact mode, third kernel of a step (normally a no-op): whole tiles whose
// near-tie queue overflowed are recomputed in float64 from the pre-step planes.  grid = fixed.
__global__ __launch_bounds__(256) void redo_tiles_f64(const float* __restrict__ inL, const float* __restrict__ inD,
                                                      float* __restrict__ outL, float* __restrict__ outD, Geom G,
                                                      int TR, int TCQ, PhysF64 P64, StatsDev* __restrict__ stats,
                                                      FixQ fq) {
    const unsigned int nredo = fq.counts[kNumQueues * 16];
    __shared__ unsigned int s_max, s_suml, s_sumd;
    for (unsigned int it = blockIdx.x; it < nredo; it += gridDim.x) {
        const int t = fq.redo_tiles[it];
        const int tiles_per_world = G.tiles_r * G.tiles_c;
        const int b = t / tiles_per_world, tw = t - b * tiles_per_world;
        const int tr = tw / G.tiles_c, tc = tw - tr * G.tiles_c;
        const int r0 = tr * TR, c0 = tc * TCQ * 4;
        const int nrows = min(TR, G.H - r0), ncols = min(TCQ * 4, G.W - c0);
        const size_t woff = (size_t)b * G.H * G.W;
        if (threadIdx.x == 0) { s_max = 0; s_suml = 0; s_sumd = 0; }
        __syncthreads();
        for (int i = threadIdx.x; i < nrows * ncols; i += 256) {
            const int r = r0 + i / ncols, c = c0 + i % ncols;
            double l9[9], d9[9];
            gather9(inL + woff, G.H, G.W, r, c, l9);
            gather9(inD + woff, G.H, G.W, r, c, d9);
            const CellF64 o = cell_f64(P64, l9, d9);
            const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
            outL[woff + (size_t)r * G.W + c] = kl;
            outD[woff + (size_t)r * G.W + c] = kd;
            atomicMax(&s_max, (unsigned int)fmaxf(kl, kd));
            atomicAdd(&s_suml, (unsigned int)kl);
            atomicAdd(&s_sumd, (unsigned int)kd);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicMax(&stats[b].max_k, s_max);
            atomicAdd(&stats[b].sum_l, (unsigned long long)s_suml);
            atomicAdd(&stats[b].sum_d, (unsigned long long)s_sumd);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// step_stream — the hot kernel for wide grids (W >= 256): wave-strip streaming.
//
// Every WAVE owns a strip of 256 columns x SR rows of one world and marches down it; a lane owns 4
// adjacent columns.  Rows are loaded straight into registers with one coalesced 16-byte load per
// lane and plane, three rows ahead of their use (the data is touched exactly once, so an LDS round
// trip would be pure overhead — cdna_hip_programming.md §5 "streamed once per block": load straight
// to VGPRs, deep prefetch, late vmcnt).  The 3x3 stencil is a 3-row register window; horizontal
// neighbours come from the adjacent lanes with DPP wavefront shifts (v_mov_b32_dpp wave_shr/shl),
// and the one column to the left / right of the strip arrives either by a wavefront ROTATE (W = 256:
// the toroidal wrap is inside the wave) or with one extra 4-byte load per row and plane in which
// lanes 0-31 fetch the left halo column and lanes 32-63 the right one (the DPP "old" operand then
// drops them into lanes 0 and 63).  No barrier in the loop: waves run independently.
//
// Exact mode needs no second kernel: a near-tie cell's 3x3 neighbourhood is already in the window
// registers, so its 48-byte payload (layout of FixQ) goes into the wave's own LDS queue (slots from
// ballot + mbcnt, no atomics), and when the strip is finished the same wave re-evaluates its queue
// in float64, one entry per lane, patches its own stores and corrects its reductions.  A strip
// whose queue overflows is recomputed whole in float64 at that point.
//
// HALO: 0 = W == 256 (rotate), 1 = W a multiple of 256 (every strip full), 2 = general (W % 4 == 0).
// ---------------------------------------------------------------------------------------------
// rows per block (= rows in flight per wave while a block is computed) and the occupancy the
// register allocator plans for; measured on MI355X (profiles/r01_stream_tuning.md)
#ifndef DW_STREAM_RB_FAST
#define DW_STREAM_RB_FAST 2
#endif
#ifndef DW_STREAM_RB_EXACT
#define DW_STREAM_RB_EXACT 2
#endif
#ifndef DW_STREAM_WAVES_EXACT
#define DW_STREAM_WAVES_EXACT 3
#endif

struct StripGeom {
    int B, H, W;
    int SR;                   // rows per wave-strip
    int ncs, nrs;             // column / row strips per world
    int nstrips;              // B * nrs * ncs
    int nwg;                  // ceil(nstrips / 4) workgroups of 4 waves
    int chunk;                // ceil(nwg / 8): workgroups per XCD
    int qcap;                 // near-tie LDS queue capacity in use (<= kWaveQueueCap; tests shrink it)
};

// streaming accesses of the hot kernel.  The new planes are not read again within the step, so they
// are stored non-temporally; non-temporal LOADS were measured slower (-DDW_NT_LOAD keeps the switch).
typedef float dw_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load4(const float* p) {
#ifdef DW_NT_LOAD
    const dw_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const dw_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void stream_store4(float* p, const float4& v) {
#ifndef DW_NO_NT_STORE      // non-temporal stores: measured -1.5 % (fast) / -6 % (exact) on C2
    dw_f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<dw_f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

struct Raw {                  // one row as loaded: own 4 columns of both planes + the halo column values
    float4 l, d;
    float hl, hd;             // lanes 0-31: column left of the strip; lanes 32-63: column right of it
};

constexpr int kDppWaveShl1 = 0x130, kDppWaveRol1 = 0x134, kDppWaveShr1 = 0x138, kDppWaveRor1 = 0x13C;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float old, float src) {
#ifdef DW_NO_DPP   // tuning experiment: same data movement through ds_bpermute
    const int lane = threadIdx.x & 63;
    if (CTRL == kDppWaveShr1) { const float v = __shfl(src, (lane + 63) & 63, 64); return lane == 0 ? old : v; }
    if (CTRL == kDppWaveShl1) { const float v = __shfl(src, (lane + 1) & 63, 64); return lane == 63 ? old : v; }
    if (CTRL == kDppWaveRor1) return __shfl(src, (lane + 63) & 63, 64);
    return __shfl(src, (lane + 1) & 63, 64);
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
#endif
}

// left / right neighbour values of the lane's 4-column group for one plane
template <int HALO>
__device__ __forceinline__ void lr_neighbours(const float4& v, float halo, int lane, int last_lane, float& lnb,
                                              float& rnb) {
    if (HALO == 0) {                     // toroidal wrap inside the wave
        lnb = dpp_mov<kDppWaveRor1>(0.f, v.w);
        rnb = dpp_mov<kDppWaveRol1>(0.f, v.x);
    } else if (HALO == 1) {              // lane 0 / lane 63 keep `old` = their halo value
        lnb = dpp_mov<kDppWaveShr1>(halo, v.w);
        rnb = dpp_mov<kDppWaveShl1>(halo, v.x);
    } else {
        const float left = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(halo), 0));
        const float right = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(halo), 63));
        lnb = dpp_mov<kDppWaveShr1>(left, v.w);            // lane 0 keeps `old` = left
        const float r = dpp_mov<kDppWaveShl1>(right, v.x);  // lane 63 keeps `old` = right
        rnb = lane == last_lane ? right : r;
    }
}

__device__ __forceinline__ Row4 make_row(const float4& v, float lnb, float rnb) {
    Row4 r;
    r.x[0] = v.x; r.x[1] = v.y; r.x[2] = v.z; r.x[3] = v.w;
    r.h2[0] = lnb + v.y;
    r.h2[1] = v.x + v.z;
    r.h2[2] = v.y + v.w;
    r.h2[3] = v.z + rnb;
    return r;
}

// the three (light | dark << 16) words of columns i-1, i, i+1 of one window row; the column left of
// x[0] is h2[0] - x[1] and the one right of x[3] is h2[3] - x[2] (exact: all values are integers)
// Wave-queue payload word: one (light, dark) pair of per-mille integers in [0, 1000] as the BITS of the
// float light + 1024*dark (exact: < 2^24) - one fma on the producer side, which runs for every lane of a
// wave that holds a near-tie cell; the few consumer lanes decode it back to light | dark << 16.
__device__ __forceinline__ unsigned int pack_ld(float l, float d) { return __float_as_uint(__builtin_fmaf(d, 1024.0f, l)); }
__device__ __forceinline__ unsigned int unpack_ld(unsigned int bits) {
    const unsigned int v = (unsigned int)__uint_as_float(bits);
    return (v & 1023u) | ((v >> 10) << 16);
}

template <int I>
__device__ __forceinline__ void pack3(const Row4& L, const Row4& D, unsigned int& w0, unsigned int& w1,
                                      unsigned int& w2) {
    const float la = I == 0 ? L.h2[0] - L.x[1] : L.x[I == 0 ? 0 : I - 1];
    const float da = I == 0 ? D.h2[0] - D.x[1] : D.x[I == 0 ? 0 : I - 1];
    const float lc = I == 3 ? L.h2[3] - L.x[2] : L.x[I == 3 ? 3 : I + 1];
    const float dc = I == 3 ? D.h2[3] - D.x[2] : D.x[I == 3 ? 3 : I + 1];
    w0 = pack_ld(la, da);
    w1 = pack_ld(L.x[I], D.x[I]);
    w2 = pack_ld(lc, dc);
}

constexpr int kWaveQueueCap = 256;          // near-tie entries per wave-strip held in LDS (48 B each)

template <int I>
__device__ __forceinline__ void queue_tie(bool tie, unsigned int& n, uint4* __restrict__ q, unsigned int cap, int b,
                                          int row, int colq,
                                          const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD,
                                          const Row4& miD, const Row4& dnD, const float* ol, const float* od) {
    const unsigned long long mask = __ballot(tie);
    if (mask == 0ull) return;                                   // wave-uniform
    if (tie) {
        const unsigned int slot = n + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (slot < cap) {
            unsigned int u0, u1, u2, m0, m1, m2, d0, d1, d2;
            pack3<I>(upL, upD, u0, u1, u2);
            pack3<I>(miL, miD, m0, m1, m2);
            pack3<I>(dnL, dnD, d0, d1, d2);
            uint4* dst = q + slot * 3;
            dst[0] = make_uint4((unsigned)b, ((unsigned)row << 16) | (unsigned)(colq + I), u0, u1);
            dst[1] = make_uint4(u2, m0, m1, m2);
            dst[2] = make_uint4(d0, d1, d2, pack_ld(ol[I], od[I]));
        }
    }
    n += (unsigned)__popcll(mask);
}

template <bool EXACT, int HALO, int RB>
__device__ __forceinline__ void stream_body(const float* __restrict__ inL, const float* __restrict__ inD,
                                            float* __restrict__ outL, float* __restrict__ outD, const StripGeom& G,
                                            const PhysF32& P, const PhysF64& P64, StatsDev* __restrict__ stats,
                                            unsigned long long* __restrict__ fixups,
                                            unsigned long long* __restrict__ zero_me, int zero_n) {
    __shared__ uint4 s_queue[EXACT ? 4 * kWaveQueueCap * 3 : 1];
    const int bid = blockIdx.x;
    const int wg = (bid & 7) * G.chunk + (bid >> 3);            // XCD-aware: contiguous run per XCD
    if (wg >= G.nwg) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (wg == 0)
        for (int i = tid; i < zero_n; i += 256) zero_me[i] = 0ull;
    uint4* q = s_queue + (EXACT ? wv * kWaveQueueCap * 3 : 0);
    const int s = wg * 4 + wv;                                  // this wave's strip
    if (s >= G.nstrips) return;                                 // waves are independent: no barrier anywhere
    const int spw = G.nrs * G.ncs;
    const int b = s / spw;
    const int sw = s - b * spw;
    const int rs = sw / G.ncs, cs = sw - rs * G.ncs;
    const int r0 = rs * G.SR, c0 = cs * 256;
    const int nr = min(G.SR, G.H - r0);
    const int ncq = min(64, (G.W - c0) >> 2);               // active lanes (4 columns each)
    const int last_lane = ncq - 1;
    const bool active = lane < ncq;
    const size_t woff = (size_t)b * G.H * G.W;
    const int colq = c0 + 4 * min(lane, last_lane);         // inactive lanes shadow the last active one
    int hcol = lane < 32 ? c0 - 1 : c0 + 4 * ncq;           // halo column of this lane's half-wave
    hcol = hcol < 0 ? hcol + G.W : (hcol >= G.W ? hcol - G.W : hcol);
    const float* pL = inL + woff;
    const float* pD = inD + woff;
    const int last_row = r0 + nr;                           // one past the strip: the bottom halo row
    float acc_max = 0.f, acc_l = 0.f, acc_d = 0.f;
    unsigned int nq = 0;                                    // entries queued by this wave (uniform)

    auto load_raw = [&](int rr) -> Raw {                    // rr in [r0-1, r0+nr], clamped + wrapped
        rr = min(rr, last_row);
        rr = rr < 0 ? rr + G.H : (rr >= G.H ? rr - G.H : rr);
        const float* rl = pL + (size_t)rr * G.W;
        const float* rd = pD + (size_t)rr * G.W;
        Raw w;
        w.l = stream_load4(rl + colq);
        w.d = stream_load4(rd + colq);
        if (HALO != 0) { w.hl = rl[hcol]; w.hd = rd[hcol]; } else { w.hl = 0.f; w.hd = 0.f; }
        return w;
    };
    auto to_rows = [&](const Raw& w, Row4& L, Row4& D) {
        float a, c;
        lr_neighbours<HALO>(w.l, w.hl, lane, last_lane, a, c);
        L = make_row(w.l, a, c);
        lr_neighbours<HALO>(w.d, w.hd, lane, last_lane, a, c);
        D = make_row(w.d, a, c);
    };

    Row4 WL[RB + 2], WD[RB + 2];
    {
        Raw p[RB + 2];
#pragma unroll
        for (int j = 0; j < RB + 2; ++j) p[j] = load_raw(r0 - 1 + j);
#pragma unroll
        for (int j = 0; j < RB + 2; ++j) to_rows(p[j], WL[j], WD[j]);
    }
    auto row_math = [&](const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD, const Row4& miD,
                        const Row4& dnD, int k) {
        float ol[4], od[4];
        bool tie[4];
        cells4<EXACT>(P, upL, miL, dnL, upD, miD, dnD, ol, od, tie);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (EXACT) {
                tie[i] = tie[i] && (HALO != 2 || active);
                acc_max = fmaxf(acc_max, tie[i] ? 0.f : fmaxf(ol[i], od[i]));
            } else {
                acc_max = fmaxf(acc_max, fmaxf(ol[i], od[i]));
            }
            acc_l += ol[i]; acc_d += od[i];
        }
        if (HALO != 2 || active) {                          // HALO 0/1: every lane owns real columns
            const size_t off = woff + (size_t)(r0 + k) * G.W + colq;
            stream_store4(outL + off, make_float4(ol[0], ol[1], ol[2], ol[3]));
            stream_store4(outD + off, make_float4(od[0], od[1], od[2], od[3]));
        }
        if (EXACT) {
            queue_tie<0>(tie[0], nq, q, (unsigned)G.qcap, b, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<1>(tie[1], nq, q, (unsigned)G.qcap, b, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<2>(tie[2], nq, q, (unsigned)G.qcap, b, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<3>(tie[3], nq, q, (unsigned)G.qcap, b, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
        }
    };
    int k = 0;
    for (; k + RB <= nr; k += RB) {
        Raw nx[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) nx[j] = load_raw(r0 + k + RB + 1 + j);
        __builtin_amdgcn_sched_barrier(0);                  // loads first, then the arithmetic
#pragma unroll
        for (int j = 0; j < RB; ++j) row_math(WL[j], WL[j + 1], WL[j + 2], WD[j], WD[j + 1], WD[j + 2], k + j);
        __builtin_amdgcn_sched_barrier(0);
        WL[0] = WL[RB]; WD[0] = WD[RB];
        WL[1] = WL[RB + 1]; WD[1] = WD[RB + 1];
#pragma unroll
        for (int j = 0; j < RB; ++j) to_rows(nx[j], WL[2 + j], WD[2 + j]);
    }
#pragma unroll
    for (int j = 0; j < RB - 1; ++j)                        // tail: < RB rows left, already in the window
        if (k + j < nr) row_math(WL[j], WL[j + 1], WL[j + 2], WD[j], WD[j + 1], WD[j + 2], k + j);
    if (HALO == 2 && !active) { acc_max = 0.f; acc_l = 0.f; acc_d = 0.f; }

    // ---- exact mode: float64 re-evaluation of this wave's queued near-tie cells ----
    if (EXACT) {
        if (nq <= (unsigned)G.qcap) {
            for (unsigned int e = lane; e < nq; e += 64) {
                const uint4 e0 = q[e * 3], e1 = q[e * 3 + 1], e2 = q[e * 3 + 2];
                const unsigned int w[9] = {unpack_ld(e0.z), unpack_ld(e0.w), unpack_ld(e1.x),
                                           unpack_ld(e1.y), unpack_ld(e1.z), unpack_ld(e1.w),
                                           unpack_ld(e2.x), unpack_ld(e2.y), unpack_ld(e2.z)};
                const unsigned int f32v = unpack_ld(e2.w);
                const NewCoverF64 o = cell_f64_lean(P64, w);
                const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
                const size_t off = woff + (size_t)(e0.y >> 16) * G.W + (e0.y & 0xffffu);
                outL[off] = kl;                             // after this wave's own float4 store of the row
                outD[off] = kd;
                acc_l += kl - (float)(f32v & 0xffffu);
                acc_d += kd - (float)(f32v >> 16);
                acc_max = fmaxf(acc_max, fmaxf(kl, kd));
            }
        } else {                                            // queue overflow: the whole strip in float64
            acc_max = 0.f; acc_l = 0.f; acc_d = 0.f;
            const int nc = min(256, G.W - c0);
            for (int i = lane; i < nr * nc; i += 64) {
                const int r = r0 + i / nc, c = c0 + i % nc;
                double l9[9], d9[9];
                gather9(pL, G.H, G.W, r, c, l9);
                gather9(pD, G.H, G.W, r, c, d9);
                const CellF64 o = cell_f64(P64, l9, d9);
                const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
                outL[woff + (size_t)r * G.W + c] = kl;
                outD[woff + (size_t)r * G.W + c] = kd;
                acc_l += kl; acc_d += kd;
                acc_max = fmaxf(acc_max, fmaxf(kl, kd));
            }
        }
    }

    // per-world reductions of this strip: wavefront shuffles, three atomics per strip
    const float m = wave_max(acc_max);
    const float sl = wave_sum(acc_l), sd = wave_sum(acc_d);
    if (lane == 0) {
        atomicMax(&stats[b].max_k, (unsigned int)m);
        atomicAdd(&stats[b].sum_l, (unsigned long long)sl);
        atomicAdd(&stats[b].sum_d, (unsigned long long)sd);
        if (EXACT && nq) atomicAdd(fixups, (unsigned long long)nq);
    }


}

// Two entry points so that each arithmetic mode gets its own register budget: the float32-only
// kernel fits 4 waves per SIMD with 2-row blocks; the exact kernel carries the tie test and the
// fix-up path and is planned for 3 waves per SIMD (<= 168 VGPRs; its 48 KB of LDS queues per
// workgroup allow 3 workgroups per CU as well).
template <int HALO>
__global__ __launch_bounds__(256) void step_stream_fast(const float* __restrict__ inL, const float* __restrict__ inD,
                                                        float* __restrict__ outL, float* __restrict__ outD,
                                                        StripGeom G, PhysF32 P, PhysF64 P64,
                                                        StatsDev* __restrict__ stats,
                                                        unsigned long long* __restrict__ fixups,
                                                        unsigned long long* __restrict__ zero_me, int zero_n) {
    stream_body<false, HALO, DW_STREAM_RB_FAST>(inL, inD, outL, outD, G, P, P64, stats, fixups, zero_me, zero_n);
}

// The exact kernels' float64 constants are needed only by the rare repair code after the strip loop.
// Taken as ordinary by-value arguments they are loaded into SGPRs at kernel entry and stay live through
// the loop, which then runs out of SGPRs (92-190 scalar spills, a v_readlane per use).  So the exact
// kernels take ONE argument struct, and the cold members are read from the kernarg segment at their use.
template <typename A>
__device__ __forceinline__ const A& kernarg_struct() {          // A is the kernel's only argument: offset 0
    return *reinterpret_cast<const A*>((const void*)__builtin_amdgcn_kernarg_segment_ptr());
}

struct StreamExactArgs {
    const float* inL; const float* inD; float* outL; float* outD;
    StripGeom G; PhysF32 P; StatsDev* stats; unsigned long long* fixups; unsigned long long* zero_me; int zero_n;
    PhysF64 P64;                                                  // cold
};

template <int HALO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DW_STREAM_WAVES_EXACT, DW_STREAM_WAVES_EXACT)))
void step_stream_exact(StreamExactArgs A) {
    stream_body<true, HALO, DW_STREAM_RB_EXACT>(A.inL, A.inD, A.outL, A.outD, A.G, A.P,
                                                kernarg_struct<StreamExactArgs>().P64, A.stats, A.fixups, A.zero_me,
                                                A.zero_n);
}

// ---------------------------------------------------------------------------------------------
// step_stream_fused2 — TWO consecutive steps per HBM round trip (float32-only mode, no agent
// update between the steps: dw_step_n on wide grids).  Temporal blocking inside the wave-strip
// design: as a wave marches down its strip, every new input row yields one row of step-1 results
// (kept only in a second 3-row register window, never written to memory) and, one row behind it,
// one row of step-2 results, which is stored.  HBM traffic per cell-update drops to ~8.5 B
// (measured by PMC, profiles/), and the kernel becomes VALU-bound.
//
// Horizontal neighbours of step-1 results come from adjacent lanes by DPP like the inputs do.
//   ROT (W == 256): the wave spans the whole torus row, all 64 lanes produce output.
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
};

// float64 step-1 value of grid cell (r, c) (any integers: wrapped onto the torus) from the input planes,
// as a packed light | dark << 16 word
__device__ inline unsigned int exact1_word(const float* __restrict__ pL, const float* __restrict__ pD, int H, int W,
                                           int r, int c, const PhysF64& Pa) {
    const int rr = ((r % H) + H) % H, cc = ((c % W) + W) % W;
    const int ru = rr == 0 ? H - 1 : rr - 1, rd = rr == H - 1 ? 0 : rr + 1;
    const int cl = cc == 0 ? W - 1 : cc - 1, cr = cc == W - 1 ? 0 : cc + 1;
    const int rows[3] = {ru, rr, rd}, cols[3] = {cl, cc, cr};
    unsigned int w1[9];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int y = 0; y < 3; ++y) {
            const size_t o = (size_t)rows[x] * W + cols[y];
            w1[x * 3 + y] = (unsigned)pL[o] | ((unsigned)pD[o] << 16);
        }
    const NewCoverF64 s1 = cell_f64_lean(Pa, w1);
    return (unsigned)dw_round3_k(s1.nl) | ((unsigned)dw_round3_k(s1.nd) << 16);
}

// exact two-step value of one cell straight from the input planes, all in float64: nine step-1
// evaluations (luminosity La) feeding one step-2 evaluation (Lb).  Used only to repair the rare
// dependents of a float32 step-1 mismatch and as the overflow fallback.
__device__ inline void exact2_cell(const float* __restrict__ pL, const float* __restrict__ pD, int H, int W, int r,
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
                    w1[x * 3 + y] = (unsigned)pL[o] | ((unsigned)pD[o] << 16);
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
template <bool ROT, bool EXACT>
__device__ __forceinline__ void fused2_body(const float* __restrict__ inL, const float* __restrict__ inD,
                                            float* __restrict__ outL, float* __restrict__ outD, const FusedGeom& G,
                                            const PhysF32& P1, const PhysF32& P2, const PhysF64& P64,
                                            const double& La, const double& Lb,
                                            unsigned long long* __restrict__ zero_me, int zero_n) {
    __shared__ uint4 s_queue[EXACT ? 4 * kWaveQueueCap * 3 : 1];
    __shared__ unsigned int s_mm[EXACT ? 4 * kMismatchCap : 1];
    const int bid = blockIdx.x;
    const int wg = (bid & 7) * G.chunk + (bid >> 3);
    if (wg >= G.nwg) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (wg == 0)
        for (int i = tid; i < zero_n; i += 256) zero_me[i] = 0ull;
    const int s = wg * 4 + wv;
    if (s >= G.nstrips) return;
    uint4* q = s_queue + (EXACT ? wv * kWaveQueueCap * 3 : 0);
    unsigned int* mm = s_mm + (EXACT ? wv * kMismatchCap : 0);
    const int spw = G.nrs * G.ncs;
    const int b = s / spw;
    const int sw = s - b * spw;
    const int rs = sw / G.ncs, cs = sw - rs * G.ncs;
    const int r0 = rs * G.SR;
    const int nr = min(G.SR, G.H - r0);
    const size_t woff = (size_t)b * G.H * G.W;
    const int c00 = ROT ? 0 : cs * 248 - 4;                     // grid column of local column 0 (may be -4)
    int col = c00 + 4 * lane;
    col = col < 0 ? col + G.W : col;
    col = col >= G.W ? col - G.W : col;                         // W >= 256 > 252: one wrap suffices
    const bool writes = ROT ? true : (lane >= 1 && lane <= 62 && cs * 248 + 4 * (lane - 1) < G.W);
    // which of my four step-1 cells feed an output cell of this wave (exact mode: only their ties matter)
    bool need1[4] = {true, true, true, true};
    if (EXACT && !ROT) {
        const bool wl = __builtin_amdgcn_update_dpp(0, writes ? 1 : 0, kDppWaveShr1, 0xf, 0xf, false) != 0;
        const bool wr = __builtin_amdgcn_update_dpp(0, writes ? 1 : 0, kDppWaveShl1, 0xf, 0xf, false) != 0;
        need1[0] = writes || wl;
        need1[1] = writes;
        need1[2] = writes;
        need1[3] = writes || wr;
    }
    const float* pL = inL + woff;
    const float* pD = inD + woff;

    auto load_raw = [&](int rr) -> Raw {                        // rr in [r0-2, r0+nr+1], clamped + wrapped
        rr = min(rr, r0 + nr + 1);
        rr = rr < 0 ? rr + G.H : rr;
        rr = rr >= G.H ? rr - G.H : rr;
        Raw w;
        w.l = stream_load4(pL + (size_t)rr * G.W + col);
        w.d = stream_load4(pD + (size_t)rr * G.W + col);
        w.hl = 0.f; w.hd = 0.f;
        return w;
    };
    auto nbrs = [&](const float4& v, float& a, float& c) {
        if (ROT) { a = dpp_mov<kDppWaveRor1>(0.f, v.w); c = dpp_mov<kDppWaveRol1>(0.f, v.x); }
        else { a = dpp_mov<kDppWaveShr1>(0.f, v.w); c = dpp_mov<kDppWaveShl1>(0.f, v.x); }   // lanes 0/63: unused
    };
    auto to_rows4 = [&](const float4& l, const float4& d, Row4& L, Row4& D) {
        float a, c;
        nbrs(l, a, c);
        L = make_row(l, a, c);
        nbrs(d, a, c);
        D = make_row(d, a, c);
    };
    unsigned int nq = 0;                                        // queued entries of this wave (uniform)
    // one row of the map with coefficient set P: (up, mid, down) -> new values; exact mode also queues
    // the near-tie cells (kind 1 = step 1, 2 = step 2; lrow = row index relative to grid row r0-2)
    auto row_map = [&](const PhysF32& P, const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD,
                       const Row4& miD, const Row4& dnD, float4& nl, float4& nd, int kind, int lrow, const bool* use) {
        float ol[4], od[4];
        bool tie[4];
        cells4<EXACT>(P, upL, miL, dnL, upD, miD, dnD, ol, od, tie);
        if (EXACT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tie[i] = tie[i] && use[i];
        }
        nl = make_float4(ol[0], ol[1], ol[2], ol[3]);
        nd = make_float4(od[0], od[1], od[2], od[3]);
        if (EXACT && __ballot(tie[0] || tie[1] || tie[2] || tie[3]) != 0ull) {
            queue_tie<0>(tie[0], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<1>(tie[1], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<2>(tie[2], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<3>(tie[3], nq, q, (unsigned)G.qcap, kind, lrow, 4 * lane, upL, miL, dnL, upD, miD, dnD, ol, od);
        }
    };
    const bool use2[4] = {writes, writes, writes, writes};

    // windows: IL/ID input rows, SL/SD step-1 rows; slot of row index j (0 = row r0-2) is j % 3
    Row4 IL[3], ID[3], SL[3], SD[3];
    {
        const Raw p0 = load_raw(r0 - 2), p1 = load_raw(r0 - 1), p2 = load_raw(r0);
        to_rows4(p0.l, p0.d, IL[0], ID[0]);
        to_rows4(p1.l, p1.d, IL[1], ID[1]);
        to_rows4(p2.l, p2.d, IL[2], ID[2]);
    }
    // iteration j = 1 .. nr+2: step-1 row j (grid row r0-2+j) from input rows j-1, j, j+1; then, from j = 3
    // on, output row k = j-3 (local row j-1) from step-1 rows j-2, j-1, j
    auto iter = [&](auto U, int j) {
        constexpr int u = decltype(U)::value;                  // u == j % 3
        const Raw nx = load_raw(r0 + j);                       // input row j+2, needed by the NEXT iteration
        __builtin_amdgcn_sched_barrier(0);
        float4 l1, d1;
        row_map(P1, IL[(u + 2) % 3], IL[u], IL[(u + 1) % 3], ID[(u + 2) % 3], ID[u], ID[(u + 1) % 3], l1, d1, 1, j, need1);
        to_rows4(l1, d1, SL[u], SD[u]);                        // step-1 row j replaces step-1 row j-3
        if (j >= 3) {
            float4 l2, d2;
            row_map(P2, SL[(u + 1) % 3], SL[(u + 2) % 3], SL[u], SD[(u + 1) % 3], SD[(u + 2) % 3], SD[u], l2, d2, 2, j - 1,
                    use2);
            if (writes) {
                const size_t off = woff + (size_t)(r0 + j - 3) * G.W + col;
                stream_store4(outL + off, l2);
                stream_store4(outD + off, d2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        to_rows4(nx.l, nx.d, IL[(u + 2) % 3], ID[(u + 2) % 3]);   // input row j+2 replaces input row j-1
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;
    using U2 = std::integral_constant<int, 2>;
    const int jend = nr + 2;
    int j = 1;
    for (; j + 2 <= jend; j += 3) {                             // j % 3 == 1 at the top
        iter(U1{}, j);
        iter(U2{}, j + 1);
        iter(U0{}, j + 2);
    }
    if (j <= jend) iter(U1{}, j);
    if (j + 1 <= jend) iter(U2{}, j + 1);

    if (EXACT) {
        PhysF64 Pa = P64, Pb = P64;
        Pa.L = La;
        Pb.L = Lb;
        // grid coordinates of a local (row, column)
        auto grid_rc = [&](int lrow, int lc, int& gr, int& gc) {
            gr = r0 - 2 + lrow;
            gr = gr < 0 ? gr + G.H : (gr >= G.H ? gr - G.H : gr);
            gc = c00 + lc;
            gc = gc < 0 ? gc + G.W : (gc >= G.W ? gc - G.W : gc);
        };
        // is local (row, column) an output cell of this wave?
        auto is_output = [&](int lrow, int lc) -> bool {
            if (lrow < 2 || lrow > nr + 1) return false;
            if (ROT) return true;
            const int ln = lc >> 2;
            return lc >= 4 && lc < 252 && cs * 248 + 4 * (ln - 1) < G.W;
        };
        unsigned int nmm = 0;
        bool redo = nq > (unsigned)G.qcap;
        if (!redo) {
            // F1 + F2 in one sweep over the queue (the entries differ only in the luminosity of their
            // float64 re-evaluation): a step-1 entry whose float32 value was wrong goes to the mismatch
            // list; a step-2 entry (payload = float32 step-1 values) patches the output - cells next to a
            // mismatch are overwritten by F3 below.
            for (unsigned int base = 0; base < nq; base += 64) {
                const unsigned int e = base + lane;
                bool mism = false;
                unsigned int where = 0;
                if (e < nq) {
                    const uint4 e0 = q[e * 3], e1 = q[e * 3 + 1], e2 = q[e * 3 + 2];
                    const unsigned int w[9] = {unpack_ld(e0.z), unpack_ld(e0.w), unpack_ld(e1.x),
                                               unpack_ld(e1.y), unpack_ld(e1.z), unpack_ld(e1.w),
                                               unpack_ld(e2.x), unpack_ld(e2.y), unpack_ld(e2.z)};
                    PhysF64 Pe = Pa;
                    Pe.L = e0.x == 1u ? La : Lb;
                    const NewCoverF64 o = cell_f64_lean(Pe, w);
                    const unsigned int kl = (unsigned)dw_round3_k(o.nl), kd = (unsigned)dw_round3_k(o.nd);
                    where = e0.y;
                    if (e0.x == 1u) {
                        mism = (kl | (kd << 16)) != unpack_ld(e2.w);
                    } else {
                        int gr, gc;
                        grid_rc((int)(where >> 16), (int)(where & 0xffffu), gr, gc);
                        const size_t off = woff + (size_t)gr * G.W + gc;
                        outL[off] = (float)kl;
                        outD[off] = (float)kd;
                    }
                }
                const unsigned long long mask = __ballot(mism);
                if (mism) {
                    const unsigned int slot = nmm + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    if (slot < (unsigned)G.mcap) mm[slot] = where;
                }
                nmm += (unsigned)__popcll(mask);
            }
            redo = nmm > (unsigned)G.mcap;
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
                        const int lc = (int)(where & 0xffffu) + (int)(t % 5u) - 2;
                        s1[lane] = exact1_word(pL, pD, G.H, G.W, r0 - 2 + lrow, c00 + lc, Pa);
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
                        if (ROT) lc = (lc + 256) & 255;
                        if (is_output(lrow, lc)) {
                            unsigned int w2[9];
#pragma unroll
                            for (int a = 0; a < 3; ++a)
#pragma unroll
                                for (int e = 0; e < 3; ++e) w2[a * 3 + e] = s1[mi * 25u + (unsigned)((1 + dy + a) * 5 + (1 + dx + e))];
                            const NewCoverF64 o = cell_f64_lean(Pb, w2);
                            int gr, gc;
                            grid_rc(lrow, lc, gr, gc);
                            const size_t off = woff + (size_t)gr * G.W + gc;
                            outL[off] = (float)dw_round3_k(o.nl);
                            outD[off] = (float)dw_round3_k(o.nd);
                        }
                    }
                }
            }
        } else {
            // overflow fallback: every output cell of the strip, two float64 steps from the inputs
            const int ncol = ROT ? 256 : 248;
            for (int i = lane; i < nr * ncol; i += 64) {
                const int lrow = 2 + i / ncol, lc = (ROT ? 0 : 4) + i % ncol;
                if (!is_output(lrow, lc)) continue;
                int gr, gc;
                grid_rc(lrow, lc, gr, gc);
                float kl, kd;
                exact2_cell(pL, pD, G.H, G.W, gr, gc, Pa, Pb, kl, kd);
                const size_t off = woff + (size_t)gr * G.W + gc;
                outL[off] = kl;
                outD[off] = kd;
            }
        }
    }
}

template <bool ROT>
__global__ __launch_bounds__(256) void step_stream_fused2(const float* __restrict__ inL, const float* __restrict__ inD,
                                                          float* __restrict__ outL, float* __restrict__ outD,
                                                          FusedGeom G, PhysF32 P1, PhysF32 P2,
                                                          unsigned long long* __restrict__ zero_me, int zero_n) {
    const PhysF64 dummy{};
    const double zero = 0.0;
    fused2_body<ROT, false>(inL, inD, outL, outD, G, P1, P2, dummy, zero, zero, zero_me, zero_n);
}

#ifndef DW_FUSED_EXACT_WAVES
#define DW_FUSED_EXACT_WAVES 2
#endif
struct FusedExactArgs {
    const float* inL; const float* inD; float* outL; float* outD;
    FusedGeom G;
    PhysF32 P1; PhysLumF32 lum2;                                  // step 2 = P1 with these members replaced:
                                                                  // 15 shared constants instead of 2 x 23 (each
                                                                  // one occupies an SGPR PAIR as a packed operand)
    unsigned long long* zero_me; int zero_n;
    PhysF64 P64; double La; double Lb;                            // cold (see kernarg_struct)
};

template <bool ROT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DW_FUSED_EXACT_WAVES, DW_FUSED_EXACT_WAVES)))
void step_stream_fused2_exact(FusedExactArgs A) {
    const FusedExactArgs& cold = kernarg_struct<FusedExactArgs>();
    const PhysF32 P2 = with_lum(A.P1, A.lum2);
    fused2_body<ROT, true>(A.inL, A.inD, A.outL, A.outD, A.G, A.P1, P2, cold.P64, cold.La, cold.Lb, A.zero_me,
                           A.zero_n);
}

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
    float* L;                       // [B][C] current planes (in/out)
    float* D;
    float* prevL;                   // [B][C] out: state before the last step (after its grazing)
    float* prevD;
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
};

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
    // LDS carve-up per world: planes [2 buffers][2 species][C] floats | agent state doubles | idx | act | red
    const size_t world_bytes = ((size_t)16 * C + (size_t)N * 8 + (size_t)N * 12 + 16 + 15) / 16 * 16;
    unsigned char* base = smem + (size_t)w * world_bytes;
    float* planes = reinterpret_cast<float*>(base);
    double* ast = reinterpret_cast<double*>(base + (size_t)16 * C);
    int* aidx = reinterpret_cast<int*>(base + (size_t)16 * C + (size_t)N * 8);
    int* act = aidx + 2 * N;
    unsigned int* red = reinterpret_cast<unsigned int*>(act + N);   // max, sum_l, sum_d, fixups
    float* curL = planes;
    float* curD = planes + C;
    float* nxtL = planes + 2 * C;
    float* nxtD = planes + 3 * C;

    if (valid) {
        for (int c = lt; c < C; c += tpw) {
            curL[c] = io.L[(size_t)b * C + c];
            curD[c] = io.D[(size_t)b * C + c];
        }
        for (int n = lt; n < N; n += tpw) {
            ast[n] = io.st[(size_t)b * N + n];
            aidx[2 * n] = io.idx[((size_t)b * N + n) * 2];
            aidx[2 * n + 1] = io.idx[((size_t)b * N + n) * 2 + 1];
        }
        if (lt < 4) red[lt] = 0;
    }
    __syncthreads();

    for (int t = 0; t < K; ++t) {
        // ---- policy: action of each agent for this step, from the state it observes ----
        if (valid && policy_mode != kPolicySkipAgents) {
            for (int n = lt; n < N; n += tpw) {
                int a = 0;
                if (policy_mode == kPolicyTable || (policy_mode != kPolicyZeros && io.use_table[t])) {
                    a = io.table[((size_t)t * B + b) * N + n];
                } else if (policy_mode != kPolicyZeros) {
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
                            v = (double)curL[r * W + c] / 1000.0 + (double)curD[r * W + c] / 1000.0;
                        }
                        if (i == 0 || (policy_mode == kPolicyArgmin ? v < bestv : v > bestv)) { best = i; bestv = v; }
                    }
                    a = 4 + best;
                }
                act[n] = a;
            }
        }
        __syncthreads();
        // ---- update_agents (ref :181-244): one lane per world, agents in order ----
        if (valid && lt == 0 && N > 0 && policy_mode != kPolicySkipAgents) {
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
                        s += (double)curL[o] / 1000.0 + (double)curD[o] / 1000.0;
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
        __syncthreads();
        // ---- forward ----
        const PhysF32 P = io.P32[t];
        PhysF64 Q = P64;
        Q.L = io.Ls[t];
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
                const GrowthF32 g = growth_f32<EXACT>(P, li, di, El, Cl, Ed, Cd);
                float kl, kd;
                if (EXACT) {
                    bool tl, td;
                    kl = finish_exact(P, li, g.gql, g.dKl, g.oml, tl);
                    kd = finish_exact(P, di, g.gqd, g.dKd, g.omd, td);
                    if (tl || td) {
                        const int rows[3] = {ru, rm, rd}, cols[3] = {cl, cc, cr};
                        unsigned int wv[9];
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                wv[a * 3 + e] = (unsigned)curL[rows[a] + cols[e]] | ((unsigned)curD[rows[a] + cols[e]] << 16);
                        const NewCoverF64 o = cell_f64_lean(Q, wv);
                        kl = (float)dw_round3_k(o.nl);
                        kd = (float)dw_round3_k(o.nd);
                        ++nfix;
                    }
                } else {
                    kl = finish_fast(li, g.gql);
                    kd = finish_fast(di, g.gqd);
                }
                nxtL[c] = kl;
                nxtD[c] = kd;
                tmax = fmaxf(tmax, fmaxf(kl, kd));
                tsl += kl;
                tsd += kd;
            }
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
            io.L[(size_t)b * C + c] = curL[c];
            io.D[(size_t)b * C + c] = curD[c];
            io.prevL[(size_t)b * C + c] = nxtL[c];
            io.prevD[(size_t)b * C + c] = nxtD[c];
        }
        for (int n = lt; n < N; n += tpw) {
            io.st[(size_t)b * N + n] = ast[n];
            io.idx[((size_t)b * N + n) * 2] = aidx[2 * n];
            io.idx[((size_t)b * N + n) * 2 + 1] = aidx[2 * n + 1];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// agents_update — ref update_agents (daisy_world_rl.py:181-244), collision_mode 0.
// One thread per world walks its agents IN ORDER (the first agent to land on a cell eats it all).
// Energy stores are float64 and updated with exactly the reference's operations, so alive/dead
// decisions and rewards are bit-identical.  `f64L/f64D` (natural units) are the exact copies of an
// un-quantised initial state when present (else nullptr).
// ---------------------------------------------------------------------------------------------
__global__ void agents_update(float* __restrict__ L32, float* __restrict__ D32,
                              double* __restrict__ f64L, double* __restrict__ f64D,
                              int* __restrict__ idx, double* __restrict__ st,
                              const int* __restrict__ action, int act_b, int act_n, int B, int N,
                              int H, int W, double agent_gamma, int do_clip) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t woff = (size_t)b * H * W;
    for (int n = 0; n < N; ++n) st[(size_t)b * N + n] -= agent_gamma;           // ref :184
    if (b < act_b) {
        for (int n = 0; n < act_n && n < N; ++n) {                               // ref :186-187
            double s = st[(size_t)b * N + n];
            if (s > 0.0) {                                                       // ref :189
                const int a = action[(size_t)b * act_n + n];
                int r = idx[((size_t)b * N + n) * 2 + 0], c = idx[((size_t)b * N + n) * 2 + 1];
                if (a != 8) {                                                    // ref :191-206
                    const int m = ((a % 4) + 4) % 4;
                    if (m == 0) c -= 1; else if (m == 1) r -= 1; else if (m == 2) r += 1; else c += 1;
                }
                r = ((r % H) + H) % H;                                           // ref :208
                c = ((c % W) + W) % W;
                idx[((size_t)b * N + n) * 2 + 0] = r;
                idx[((size_t)b * N + n) * 2 + 1] = c;
                if (a > 4) {                                                     // ref :210-216
                    const size_t o = woff + (size_t)r * W + c;
                    double l, d;
                    if (f64L) { l = f64L[o]; d = f64D[o]; f64L[o] = 0.0; f64D[o] = 0.0; }
                    else { l = (double)L32[o] / 1000.0; d = (double)D32[o] / 1000.0; }
                    s += l + d;
                    L32[o] = 0.f; D32[o] = 0.f;
                    st[(size_t)b * N + n] = s;
                }
            }
        }
    }
    if (do_clip)     // collision_mode 1: the collision pass (host, RNG-coupled) runs before the clip (ref :220-244)
        for (int n = 0; n < N; ++n) {                                            // ref :244
            const double s = st[(size_t)b * N + n];
            st[(size_t)b * N + n] = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
        }
}

// reward / done (ref step :486-492, N > 0):  reward = state * (state > 0); done = reward < 0.1
__global__ void reward_done(const double* __restrict__ st, double* __restrict__ reward,
                            unsigned char* __restrict__ done, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = st[i];
    const double r = s * (s > 0.0 ? 1.0 : 0.0);
    reward[i] = r;
    done[i] = r < 0.1 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// materialise — the reference's 7-channel float64 grid.
//   after a step (POST=true): prev = pre-step covers (post-graze), cur = quantised new covers:
//       ch0 = round3(p - nl - nd) from the un-rounded clipped covers (ref :450,452),
//       ch1/2 = cur/1000, ch3..5 = round3(T, T_light, T_dark) of prev (ref :446-448), ch6 = 0.
//   after an upload (POST=false): prev = the initial covers; ch0..2 un-rounded, ch3..5 un-rounded
//       temperatures (ref :310-323).
// caches (optional): temps[3], betas[3], growth[2], temp_effective — un-rounded (ref :345-419).
// Agent states are written into channel 4 afterwards by agents_stamp (ref :454-459).
// ---------------------------------------------------------------------------------------------
template <typename PrevT, bool POST>
__global__ __launch_bounds__(256) void materialise(const PrevT* __restrict__ pL,
                                                   const PrevT* __restrict__ pD,
                                                   const float* __restrict__ cL,
                                                   const float* __restrict__ cD, int H, int W,
                                                   PhysF64 P, double* __restrict__ grid7,
                                                   double* __restrict__ temps,
                                                   double* __restrict__ betas,
                                                   double* __restrict__ growth,
                                                   double* __restrict__ teff) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H * W) return;
    const size_t n = (size_t)H * W, woff = (size_t)b * n;
    const int r = cell / W, c = cell - r * W;
    double l9[9], d9[9];
    gather9(pL + woff, H, W, r, c, l9);
    gather9(pD + woff, H, W, r, c, d9);
    const CellF64 o = cell_f64(P, l9, d9);
    if (grid7) {
        double* g = grid7 + (size_t)b * 7 * n + cell;
        if (POST) {
            g[0 * n] = dw_round3_k(P.p - o.nl - o.nd) / 1000.0;
            g[1 * n] = (double)cL[woff + cell] / 1000.0;
            g[2 * n] = (double)cD[woff + cell] / 1000.0;
            g[3 * n] = dw_round3_k(o.T) / 1000.0;
            g[4 * n] = dw_round3_k(o.Tl) / 1000.0;
            g[5 * n] = dw_round3_k(o.Td) / 1000.0;
        } else {
            g[0 * n] = P.p - l9[4] - d9[4];
            g[1 * n] = l9[4];
            g[2 * n] = d9[4];
            g[3 * n] = o.T;
            g[4 * n] = o.Tl;
            g[5 * n] = o.Td;
        }
        g[6 * n] = 0.0;
    }
    if (temps) {
        double* t = temps + (size_t)b * 3 * n + cell;
        t[0] = o.T; t[n] = o.Tl; t[2 * n] = o.Td;
    }
    if (betas) {
        double* t = betas + (size_t)b * 3 * n + cell;
        t[0] = o.b; t[n] = o.bl; t[2 * n] = o.bd;
    }
    if (growth) {
        double* t = growth + (size_t)b * 2 * n + cell;
        t[0] = o.gl; t[n] = o.gd;
    }
    if (teff) teff[woff + cell] = o.Te;
}

// ref forward :454-459 — agent states into channel 4 at agent cells, in agent order (last wins)
__global__ void agents_stamp(double* __restrict__ grid7, const int* __restrict__ idx,
                             const double* __restrict__ st, int B, int N, int H, int W) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t n = (size_t)H * W;
    for (int a = 0; a < N; ++a) {
        const int r = idx[((size_t)b * N + a) * 2], c = idx[((size_t)b * N + a) * 2 + 1];
        grid7[(size_t)b * 7 * n + 4 * n + (size_t)r * W + c] = st[(size_t)b * N + a];
    }
}

// ---------------------------------------------------------------------------------------------
// observe — ref get_obs (:246-263): [B][N][7][3][3] float64 = the 3x3 wrap-around patch of the
// 7-channel grid around each agent, times the neighbourhood mask.  One thread per (agent, patch
// cell); the channel values are re-derived in float64 exactly as `materialise` does, so no
// 7-channel grid ever exists in HBM.  Channel 4 shows agent states at agent cells (ref :459).
// ---------------------------------------------------------------------------------------------
template <typename PrevT, bool POST>
__global__ void observe(const PrevT* __restrict__ pL, const PrevT* __restrict__ pD,
                        const float* __restrict__ cL, const float* __restrict__ cD,
                        const int* __restrict__ idx, const double* __restrict__ st, int B, int N,
                        int H, int W, PhysF64 P, int mask, double* __restrict__ obs) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * N * 9) return;
    const int k = gid % 9, an = gid / 9;       // patch cell, flat agent id
    const int b = an / N;
    double* o7 = obs + (size_t)an * 63 + k;    // channel stride 9
    if (!((mask >> k) & 1)) {
#pragma unroll
        for (int ch = 0; ch < 7; ++ch) o7[ch * 9] = 0.0;
        return;
    }
    const int ar = idx[(size_t)an * 2], ac = idx[(size_t)an * 2 + 1];
    const int r = (ar + (k / 3 - 1) + H) % H, c = (ac + (k % 3 - 1) + W) % W;
    const size_t n = (size_t)H * W, woff = (size_t)b * n;
    double l9[9], d9[9];
    gather9(pL + woff, H, W, r, c, l9);
    gather9(pD + woff, H, W, r, c, d9);
    const CellF64 o = cell_f64(P, l9, d9);
    double v[7];
    if (POST) {
        v[0] = dw_round3_k(P.p - o.nl - o.nd) / 1000.0;
        v[1] = (double)cL[woff + (size_t)r * W + c] / 1000.0;
        v[2] = (double)cD[woff + (size_t)r * W + c] / 1000.0;
        v[3] = dw_round3_k(o.T) / 1000.0;
        v[4] = dw_round3_k(o.Tl) / 1000.0;
        v[5] = dw_round3_k(o.Td) / 1000.0;
    } else {
        v[0] = P.p - l9[4] - d9[4]; v[1] = l9[4]; v[2] = d9[4];
        v[3] = o.T; v[4] = o.Tl; v[5] = o.Td;
    }
    v[6] = 0.0;
    if (POST) {   // ref forward :454-459 (reset()'s initial grid carries no agent stamps)
        for (int a = 0; a < N; ++a) {
            const int rr = idx[((size_t)b * N + a) * 2], cc = idx[((size_t)b * N + a) * 2 + 1];
            if (rr == r && cc == c) v[4] = st[(size_t)b * N + a];
        }
    }
#pragma unroll
    for (int ch = 0; ch < 7; ++ch) o7[ch * 9] = v[ch];
}

// ---------------------------------------------------------------------------------------------
// policy_greedy — ref Greedy.__call__ deterministic branch (agents/greedy.py:18-30):
// food = light + dark of the observation patch; candidates are flat 3x3 indices [3,1,7,5];
// action = 4 + argmax (or argmin), first extremum wins.  Reads the CURRENT covers directly
// (ch1+ch2 of the post-step observation are exactly cur/1000).
// ---------------------------------------------------------------------------------------------
// `agent_mode` (optional, [N]): per agent index 0 = argmax, 1 = argmin, 2 = keep the action already in
// the buffer (e.g. host-drawn random actions uploaded earlier) — mixed-policy ensembles (BASELINE C5).
__global__ void policy_greedy(const float* __restrict__ cL, const float* __restrict__ cD,
                              const int* __restrict__ idx, int B, int N, int H, int W, int mask,
                              int argmin, const int* __restrict__ agent_mode, int* __restrict__ action) {
    const int an = blockIdx.x * blockDim.x + threadIdx.x;
    if (an >= B * N) return;
    const int b = an / N;
    if (agent_mode) {
        const int m = agent_mode[an - b * N];
        if (m == 2) return;
        argmin = m == 1;
    }
    const int ar = idx[(size_t)an * 2], ac = idx[(size_t)an * 2 + 1];
    const size_t woff = (size_t)b * H * W;
    const int cand[4] = {3, 1, 7, 5};
    int best = 0;
    double bestv = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = cand[i];
        double v = 0.0;
        if ((mask >> k) & 1) {
            const int r = (ar + (k / 3 - 1) + H) % H, c = (ac + (k % 3 - 1) + W) % W;
            const size_t o = woff + (size_t)r * W + c;
            v = (double)cL[o] / 1000.0 + (double)cD[o] / 1000.0;
        }
        if (i == 0 || (argmin ? v < bestv : v > bestv)) { best = i; bestv = v; }
    }
    action[an] = 4 + best;
}

// ---------------------------------------------------------------------------------------------
// policy_mlp — ref MLP.get_action (daisy/agents/mlp.py:97-116): 63 -> 16 -> 32 -> 9 ReLU network on
// the flattened (7,3,3) observation, action = argmax of the logits (first maximum).  float64 like the
// reference.  Agents [a0, a1) of every world; `obs` is the
// [B][N][63] buffer written by `observe`, `W` the flat parameter vector (three matrices raveled
// row-major in layer order, ref get_parameters :118-125).  SURVEY.md §8(f) row N3.
// ---------------------------------------------------------------------------------------------
// `member` (optional): parameter set of each world — a whole ES population evaluated as one ensemble
// (world b uses W + member[b] * 1808); nullptr = one set for all worlds.
// Sixteen lanes per agent (four agents per wave): lane j owns hidden unit j of layer 1, units j and j+16
// of layer 2 and logit j (< 9); every dot product is accumulated sequentially in index order with fma,
// activations travel through LDS.  ~130 dependent float64 fmas per agent instead of 1808 in one thread.
__global__ __launch_bounds__(64) void policy_mlp(const double* __restrict__ obs, const double* __restrict__ W,
                                                 const int* __restrict__ member, int B, int N, int a0, int a1,
                                                 int* __restrict__ action) {
    __shared__ double s_x[4][64], s_h1[4][16], s_h2[4][32], s_o[4][16];
    const int na = a1 - a0;
    const int g = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int t = blockIdx.x * 4 + g;                       // agent handled by this 16-lane group
    const bool valid = t < B * na;
    const int tc = valid ? t : 0;
    const int b = tc / na, n = a0 + (tc - b * na);
    const double* x = obs + ((size_t)b * N + n) * 63;
    if (member) W += (size_t)member[b] * 1808;
    const double* W1 = W;                 // [63][16]
    const double* W2 = W + 63 * 16;       // [16][32]
    const double* W3 = W2 + 16 * 32;      // [32][9]
    for (int i = j; i < 63; i += 16) s_x[g][i] = x[i];
    __syncthreads();
    double h = 0.0;
    for (int i = 0; i < 63; ++i) h = __builtin_fma(s_x[g][i], W1[i * 16 + j], h);
    s_h1[g][j] = h * (h > 0.0 ? 1.0 : 0.0);
    __syncthreads();
    double u = 0.0, v = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const double hi = s_h1[g][i];
        u = __builtin_fma(hi, W2[i * 32 + j], u);
        v = __builtin_fma(hi, W2[i * 32 + j + 16], v);
    }
    s_h2[g][j] = u * (u > 0.0 ? 1.0 : 0.0);
    s_h2[g][j + 16] = v * (v > 0.0 ? 1.0 : 0.0);
    __syncthreads();
    if (j < 9) {
        double o = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) o = __builtin_fma(s_h2[g][i], W3[i * 9 + j], o);
        s_o[g][j] = o;
    }
    __syncthreads();
    if (j == 0 && valid) {
        int best = 0;
        double bestv = s_o[g][0];
#pragma unroll
        for (int k = 1; k < 9; ++k) {
            const double o = s_o[g][k];
            if (o > bestv) { best = k; bestv = o; }        // first maximum, as np.argmax
        }
        action[(size_t)b * N + n] = best;
    }
}

// ---------------------------------------------------------------------------------------------
// tie_audit — evidence for the exact mode's error bound.  For every cell of a quantised state:
// float32 per-mille growth gq32 (the hot kernels' arithmetic, split coefficient chains) against the
// float64 growth of cell_f64, and the per-cell bound eps the tie test would use.  Reduces
//   out[0] = max |gq32 - gq64| (quanta)      out[1] = max (|gq32 - gq64| / eps)   (< 1 <=> bound holds)
//   out[2] = number of cells the tie test flags   out[3] = number of cells audited
// (both species count).  Non-negative doubles order like their bit patterns: atomicMax on u64.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tie_audit(const float* __restrict__ L, const float* __restrict__ D, int H, int W,
                                                 PhysF32 P, PhysF64 P64, unsigned long long* __restrict__ out) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H * W) return;
    const size_t woff = (size_t)b * H * W;
    const float* pl = L + woff;
    const float* pd = D + woff;
    const int r = cell / W, c = cell - r * W;
    const int ru = r == 0 ? H - 1 : r - 1, rd = r == H - 1 ? 0 : r + 1;
    const int cl = c == 0 ? W - 1 : c - 1, cr = c == W - 1 ? 0 : c + 1;
#define DW_AT(p, rr, cc) (p)[(size_t)(rr) * W + (cc)]
    const float li = DW_AT(pl, r, c), di = DW_AT(pd, r, c);
    const float El = (DW_AT(pl, ru, c) + DW_AT(pl, rd, c)) + (DW_AT(pl, r, cl) + DW_AT(pl, r, cr));
    const float Cl = (DW_AT(pl, ru, cl) + DW_AT(pl, rd, cl)) + (DW_AT(pl, ru, cr) + DW_AT(pl, rd, cr));
    const float Ed = (DW_AT(pd, ru, c) + DW_AT(pd, rd, c)) + (DW_AT(pd, r, cl) + DW_AT(pd, r, cr));
    const float Cd = (DW_AT(pd, ru, cl) + DW_AT(pd, rd, cl)) + (DW_AT(pd, ru, cr) + DW_AT(pd, rd, cr));
#undef DW_AT
    const GrowthF32 g = growth_f32<true>(P, li, di, El, Cl, Ed, Cd);
    double l9[9], d9[9];
    gather9(pl, H, W, r, c, l9);
    gather9(pd, H, W, r, c, d9);
    const CellF64 o = cell_f64(P64, l9, d9);
    const double g64[2] = {P64.dt * o.gl * 1000.0, P64.dt * o.gd * 1000.0};
    const float g32[2] = {g.gql, g.gqd};
    const float dK[2] = {g.dKl, g.dKd};
    const float om[2] = {g.oml, g.omd};
    double max_err = 0.0, max_ratio = 0.0;
    unsigned long long flagged = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float thr = fmaf(-fabsf(dK[k]), fmaf(P.eK1, om[k], P.eK0), fmaf(-P.eA, fabsf(g32[k]), P.tie_lo));
        const double eps = 0.5 - (double)thr;
        const double err = fabs((double)g32[k] - g64[k]);
        max_err = fmax(max_err, err);
        max_ratio = fmax(max_ratio, err / eps);
        const float rr = __builtin_rintf(g32[k]);
        flagged += fabsf(g32[k] - rr) > thr ? 1ull : 0ull;
    }
    atomicMax(&out[0], (unsigned long long)__double_as_longlong(max_err));
    atomicMax(&out[1], (unsigned long long)__double_as_longlong(max_ratio));
    if (flagged) atomicAdd(&out[2], flagged);
    atomicAdd(&out[3], 2ull);
}

// dw_run_episode on worlds that do not fit LDS: one step's flags from the step kernel's reductions
// (same predicates as episode_small), and one step's actions out of the caller's int8 table
__global__ void episode_flags(const StatsDev* __restrict__ stats, const double* __restrict__ st, int B, int N,
                              unsigned int thr, unsigned char* __restrict__ world_alive,
                              unsigned char* __restrict__ agent_ok) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) world_alive[i] = stats[i].max_k > thr ? 1 : 0;
    if (i < B * N) {
        const double s = st[i];
        const double rw = s * (s > 0.0 ? 1.0 : 0.0);
        agent_ok[i] = rw < 0.1 ? 0 : 1;
    }
}
__global__ void actions_from_table(const signed char* __restrict__ table, int n, int* __restrict__ action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) action[i] = (int)table[i];
}

// lifespan counters (ref notebooks/greedy_longevity_abatement.ipynb cell 2:46-52)
__global__ void lifespan_accumulate(const StatsDev* __restrict__ stats, const double* __restrict__ st,
                                    int B, int N, unsigned int thr, int* __restrict__ done_at,
                                    int* __restrict__ agents_done_at, int* __restrict__ n_alive) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        const int alive = stats[i].max_k > thr ? 1 : 0;      // grid_done = max <= 0.005
        done_at[i] += alive;
        if (alive) atomicAdd(n_alive, 1);
    }
    if (i < B * N) {
        const double s = st[i];
        const double r = s * (s > 0.0 ? 1.0 : 0.0);
        agents_done_at[i] += (r < 0.1) ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------------------------
// init_random — ref initialize_grid :287-302 / initialize_agents :175-179 with Philox4x32-10.
// counter = (cell lo, cell hi, world lo, world hi), key = seed.  One call per cell gives the four
// uniforms (U1_dark, U2_dark, U1_light, U2_light); the reference draws dark first.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void init_random_cells(float* __restrict__ L, float* __restrict__ D,
                                                         int H, int W, long long world_offset,
                                                         unsigned long long seed, float light_prop,
                                                         float dark_prop, float ial, float iad) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H * W) return;
    const unsigned long long world = (unsigned long long)(world_offset + b);
    uint32_t r[4];
    philox4x32_10((uint32_t)cell, 0u, (uint32_t)world, (uint32_t)(world >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    const float d = (u01(r[0]) < dark_prop) ? iad * u01(r[1]) : 0.f;
    const float l = (u01(r[2]) < light_prop) ? ial * u01(r[3]) : 0.f;
    const size_t o = (size_t)b * H * W + cell;
    L[o] = l * 1000.f;
    D[o] = d * 1000.f;
}

__global__ void init_random_agents(int* __restrict__ idx, double* __restrict__ st, int B, int N, int H,
                                   int W, long long world_offset, unsigned long long seed) {
    const int an = blockIdx.x * blockDim.x + threadIdx.x;
    if (an >= B * N) return;
    const int b = an / N, a = an - b * N;
    const unsigned long long world = (unsigned long long)(world_offset + b);
    uint32_t r[4];
    philox4x32_10((uint32_t)a, 0x80000000u, (uint32_t)world, (uint32_t)(world >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    idx[(size_t)an * 2 + 0] = (int)(((unsigned long long)r[0] * (unsigned)H) >> 32);
    idx[(size_t)an * 2 + 1] = (int)(((unsigned long long)r[1] * (unsigned)W) >> 32);
    st[an] = 1.0;
}

// plane conversions
__global__ void f64_to_permille(const double* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)(in[i] * 1000.0);
}
__global__ void f32nat_to_permille(const float* __restrict__ in, float* __restrict__ out, size_t n,
                                   int quantise) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float k = in[i] * 1000.f;
        out[i] = quantise ? __builtin_rintf(k) : k;
    }
}
__global__ void permille_to_f64(const float* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)in[i] / 1000.0;
}

// stats of an arbitrary state (used after uploads / init so that dw_reduce is always valid)
constexpr int kStatsChunk = 4096;
template <typename T>
__global__ __launch_bounds__(256) void stats_only(const T* __restrict__ L, const T* __restrict__ D,
                                                  int H, int W, StatsDev* __restrict__ stats) {
    // a workgroup reduces kStatsChunk cells of world blockIdx.y (coalesced, stride 256): three atomics per
    // wave per 4096 cells instead of per 64 (the per-world counters are contended)
    const int b = blockIdx.y;
    const int n = H * W;
    const int base = blockIdx.x * kStatsChunk;
    float m = 0.f, sl = 0.f, sd = 0.f;
    for (int i = threadIdx.x; i < kStatsChunk; i += 256) {
        const int cell = base + i;
        if (cell < n) {
            const float kl = to_permille(L[(size_t)b * n + cell]);
            const float kd = to_permille(D[(size_t)b * n + cell]);
            m = fmaxf(m, fmaxf(kl, kd));
            sl += kl;
            sd += kd;
        }
    }
    m = wave_max(m);
    sl = wave_sum(sl);
    sd = wave_sum(sd);
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[b].max_k, (unsigned int)ceilf(m));
        atomicAdd(&stats[b].sum_l, (unsigned long long)(sl + 0.5f));
        atomicAdd(&stats[b].sum_d, (unsigned long long)(sd + 0.5f));
    }
}

}  // namespace dw
