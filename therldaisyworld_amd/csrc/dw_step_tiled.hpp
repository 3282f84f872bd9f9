// dw_step_tiled.hpp — step_tiled: LDS-staged tile with halo (grids with W < 256, W % 4 == 0), its
// global near-tie queues and the float64 fix-up / tile-redo kernels of the exact mode.  Binary16 planes in
// HBM (8-byte groups of four cells), float32 tile in LDS.
#pragma once
#include "dw_common.hpp"

namespace dw {

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

#ifdef DW_TUNING
// plain streaming copy of both planes: the achievable-bandwidth yardstick for this traffic shape
__global__ __launch_bounds__(256) void copy_planes(const float4* __restrict__ inL, const float4* __restrict__ inD,
                                                   float4* __restrict__ outL, float4* __restrict__ outD, size_t n4) {   // n4: 16-byte groups
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) { outL[i] = inL[i]; outD[i] = inD[i]; }
}
__device__ int g_ablate;   // 0 normal, 1 skip the arithmetic (stage -> LDS -> registers -> store)
#endif

template <int TCQ, int RPT, bool EXACT>
__global__ __launch_bounds__(256) void step_tiled(const plane_t* __restrict__ inL,
                                                  const plane_t* __restrict__ inD,
                                                  plane_t* __restrict__ outL,
                                                  plane_t* __restrict__ outD, Geom G, PhysF32 P,
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
        dw_f16x4 vl[C::STAGE_ITERS], vd[C::STAGE_ITERS];
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
            vl[it] = stream_load4_raw(inL + off);
            vd[it] = stream_load4_raw(inD + off);
        }
#pragma unroll
        for (int it = 0; it < C::STAGE_ITERS; ++it) {
            const int idx = it * 256 + tid;
            const int j = idx / C::LQ, gq = idx - j * C::LQ;
            if (j < lrows && gq < lq) {
                *reinterpret_cast<float4*>(lds + j * C::LSTRIDE + gq * 4) = widen4(vl[it]);
                *reinterpret_cast<float4*>(lds + C::PLANE + j * C::LSTRIDE + gq * 4) = widen4(vd[it]);
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
                stream_store4(outL + off, make_float4(ol[0], ol[1], ol[2], ol[3]));
                stream_store4(outD + off, make_float4(od[0], od[1], od[2], od[3]));
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
__global__ __launch_bounds__(256) void fixup_cells(plane_t* __restrict__ outL, plane_t* __restrict__ outD, int H, int W,
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
            outL[off] = (plane_t)kl;
            outD[off] = (plane_t)kd;
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
__global__ __launch_bounds__(256) void redo_tiles_f64(const plane_t* __restrict__ inL, const plane_t* __restrict__ inD,
                                                      plane_t* __restrict__ outL, plane_t* __restrict__ outD, Geom G,
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
            outL[woff + (size_t)r * G.W + c] = (plane_t)kl;
            outD[woff + (size_t)r * G.W + c] = (plane_t)kd;
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

}  // namespace dw
