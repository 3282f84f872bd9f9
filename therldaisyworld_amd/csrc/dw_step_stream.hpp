// dw_step_stream.hpp — step_stream_{fast,exact}: the wave-strip streaming step kernel for W >= 256
// (register window, DPP neighbours, per-wave LDS queue + in-wave float64 fix-up in exact mode).
#pragma once
#include "dw_common.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// step_stream — the hot kernel for wide grids (W >= 256): wave-strip streaming.
//
// Every WAVE owns a strip of 256 columns x SR rows of one world and marches down it; a lane owns 4
// adjacent columns.  Rows are loaded straight into registers with one coalesced 8-byte load (four
// binary16 cells) per lane and plane, three rows ahead of their use (the data is touched exactly once, so an LDS round
// trip would be pure overhead — cdna_hip_programming.md §5 "streamed once per block": load straight
// to VGPRs, deep prefetch, late vmcnt).  The 3x3 stencil is a 3-row register window; horizontal
// neighbours come from the adjacent lanes with DPP wavefront shifts (v_mov_b32_dpp wave_shr/shl),
// and the one column to the left / right of the strip arrives either by a wavefront ROTATE (W = 256:
// the toroidal wrap is inside the wave) or with one extra 2-byte load per row and plane in which
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
    int lpw, wpr;             // packed mode (W < 256): lanes per world row (W/4), worlds per wave row (64 / lpw)
    int force_rescan;         // tests: every exact strip takes the maximum's re-scan path (see `rescan_max` in stream_body)
};

struct Raw {                  // one row AS LOADED (binary16: 6 VGPRs; widened where it is consumed): own 4 columns
    dw_f16x4 l, d;            // of both planes
    plane_t hl, hd;           // halo column values - lanes 0-31: column left of the strip; lanes 32-63: right of it
};

// the lanes for which a condition holds, as a 64-bit mask.  The condition is the result of vector compares, i.e.
// already a lane mask in a scalar register pair: the intrinsic on the bool itself is that mask and-ed with exec
// (HIP's __ballot(int) first widens the bool to an int: a v_cndmask + v_cmp_ne pair per call in the hot loop).
__device__ __forceinline__ unsigned long long lane_mask(bool c) { return __builtin_amdgcn_ballot_w64(c); }

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

// the same move when no lane needs an `old` value (rotations; shifts whose edge lane is unused): bound_ctrl
// lets the hardware supply 0 there, so no register has to be zeroed for the tied destination
template <int CTRL>
__device__ __forceinline__ float dpp_mov_nb(float src) {
#ifdef DW_NO_DPP
    return dpp_mov<CTRL>(0.f, src);
#else
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xf, 0xf, true));
#endif
}

// left / right neighbour values of the lane's 4-column group for one plane
// HALO: 0 = W == 256 (wrap inside the wave), 1 = W % 256 == 0, 2 = any other W >= 256,
//       3 = packed: W < 256 (a multiple of 4): a wave row holds 64 / (W/4) worlds side by side and the
//           toroidal wrap is a rotation inside each world's group of W/4 lanes (ds_bpermute); widths that do
//           not divide 256 leave the last lanes of the wave idle.
template <int HALO>
__device__ __forceinline__ void lr_neighbours(const float4& v, float halo, int lane, int last_lane, float& lnb,
                                              float& rnb, int lsrc = 0, int rsrc = 0) {
    if (HALO == 3) {
        lnb = __shfl(v.w, lsrc, 64);
        rnb = __shfl(v.x, rsrc, 64);
    } else if (HALO == 0) {              // toroidal wrap inside the wave
        lnb = dpp_mov_nb<kDppWaveRor1>(v.w);
        rnb = dpp_mov_nb<kDppWaveRol1>(v.x);
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

// Ordering of the exact mode's repair stores.  A repaired cell is first written by the strip loop (a 16- or
// 8-byte non-temporal row store of some lane) and later patched by another lane of the SAME wave with a scalar
// store to an address inside that row store.  Both are issued in program order by one wave, but nothing in the
// ISA orders two stores of different lanes and cache policies to overlapping bytes until the first has been
// acknowledged; on gfx9-family parts stores are counted in vmcnt, so `s_waitcnt vmcnt(0)` after the strip loop
// waits until every row store of the wave has been written to L2 before the first patch store is issued.
// Once per strip: free.
__device__ __forceinline__ void wait_row_stores_before_patching() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // compiler: no store moves across
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // gfx9 encoding: vmcnt(0), expcnt / lgkmcnt untouched
}

// Tie flags come in two forms: the wave's 64-bit LANE MASK (wave-uniform: the and with the "cell matters" mask, the
// skip tests, the slot numbering and the count all run on the scalar unit - the form of every kernel that has the
// scalar registers for it) or one bool per lane (the STATS variants, which are at their register budget with it).
__device__ __forceinline__ bool in_mask(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ bool tie_lane(unsigned long long mask) { return in_mask(mask); }    // is THIS lane flagged?
__device__ __forceinline__ bool tie_lane(bool flag) { return flag; }
__device__ __forceinline__ unsigned long long tie_mask(unsigned long long mask) { return mask; }
__device__ __forceinline__ unsigned long long tie_mask(bool flag) { return lane_mask(flag); }

template <int I, typename F>
__device__ __forceinline__ void queue_tie(F tie, unsigned int& n, uint4* __restrict__ q, unsigned int cap,
                                          int b, int row, int colq,
                                          const Row4& upL, const Row4& miL, const Row4& dnL, const Row4& upD,
                                          const Row4& miD, const Row4& dnD, const float* ol, const float* od) {
    const unsigned long long mask = tie_mask(tie);
    if (mask == 0ull) return;                                   // wave-uniform
    if (tie_lane(tie)) {
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

template <bool EXACT, int HALO, int RB, bool SYM = false>
__device__ __forceinline__ void stream_body(const plane_t* __restrict__ inL, const plane_t* __restrict__ inD,
                                            plane_t* __restrict__ outL, plane_t* __restrict__ outD, const StripGeom& G,
                                            const PhysF32& P_, const PhysF64& P64, StatsDev* __restrict__ stats,
                                            unsigned long long* __restrict__ fixups,
                                            unsigned long long* __restrict__ zero_me, int zero_n) {
    __shared__ uint4 s_queue[EXACT ? 4 * kWaveQueueCap * 3 : 1];
    // exact kernels: the addend constants of fmas whose multiplier is a scalar pair too, held in VGPRs for the whole
    // strip (see the fused kernels' PIN; bit mask: 2 pck, 4 eKb, 8 gt)
#ifndef DW_STREAM_PIN
#define DW_STREAM_PIN 0
#endif
    PhysF32 P = P_;
    if constexpr (EXACT && DW_STREAM_PIN != 0) {
        if constexpr ((DW_STREAM_PIN & 2) != 0) asm volatile("" : "+v"(P.pck));
        if constexpr ((DW_STREAM_PIN & 4) != 0) asm volatile("" : "+v"(P.eKb));
        if constexpr ((DW_STREAM_PIN & 8) != 0) asm volatile("" : "+v"(P.gt));
    }
    const int bid = blockIdx.x;
    const int wg = (bid & 7) * G.chunk + (bid >> 3);            // XCD-aware: contiguous run per XCD
    if (wg >= G.nwg) return;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    if (wg == 0)
        for (int i = tid; i < zero_n; i += 256) zero_me[i] = 0ull;
    uint4* q = s_queue + (EXACT ? wv * kWaveQueueCap * 3 : 0);
    const int s = __builtin_amdgcn_readfirstlane(wg * 4 + wv);   // this wave's strip (wave-uniform: row addressing on the scalar unit)
    if (s >= G.nstrips) return;                                 // waves are independent: no barrier anywhere
    const int spw = G.nrs * G.ncs;
    const int b = s / spw;
    const int sw = s - b * spw;
    const int rs = sw / G.ncs, cs = sw - rs * G.ncs;
    const int r0 = rs * G.SR, c0 = cs * 256;
    const int nr = min(G.SR, G.H - r0);
    constexpr bool PACK = HALO == 3;
    // packed mode: `b` is a GROUP of wpr worlds; this lane's world, its column group and the lanes that
    // hold its left / right neighbour columns (rotation inside the world's lane group)
    const int pw = PACK ? lane / G.lpw : 0, pj = PACK ? lane - pw * G.lpw : 0;
    // (widths that do not divide 256 leave the last 64 - wpr*lpw lanes without a world: pw == wpr there)
    const int world = PACK ? min(b * G.wpr + min(pw, G.wpr - 1), G.B - 1) : b;   // idle lanes shadow a real world
    const int lsrc = PACK ? (pj == 0 ? lane + G.lpw - 1 : lane - 1) : 0;
    const int rsrc = PACK ? (pj == G.lpw - 1 ? lane - (G.lpw - 1) : lane + 1) : 0;
    const int ncq = PACK ? 64 : min(64, (G.W - c0) >> 2);   // active lanes (4 columns each)
    const int last_lane = ncq - 1;
    const bool active = PACK ? (pw < G.wpr && b * G.wpr + pw < G.B) : lane < ncq;
    const unsigned long long active_mask = lane_mask(active);
    const size_t woff = (size_t)world * G.H * G.W;
    const int colq = PACK ? 4 * pj : c0 + 4 * min(lane, last_lane);   // inactive lanes shadow the last active one
    int hcol = lane < 32 ? c0 - 1 : c0 + 4 * ncq;           // halo column of this lane's half-wave
    hcol = hcol < 0 ? hcol + G.W : (hcol >= G.W ? hcol - G.W : hcol);
    const plane_t* pL = inL + woff;
    const plane_t* pD = inD + woff;
    const int last_row = r0 + nr;                           // one past the strip: the bottom halo row
    float acc_max = 0.f, acc_l = 0.f, acc_d = 0.f;
    // The strip's maximum.  The new values are non-negative float32 integers, so their BIT PATTERNS order like the values:
    // the row loop keeps an integer maximum (one v_max3_i32 per cell; fmaxf is two more instructions per cell, it quiets a
    // possible signalling NaN in each operand first).  Exact kernels: a near-tie cell's float32 value may be one quantum
    // off, and round 3 kept it out of the maximum with a select per cell.  Now the row loop takes the maximum A over ALL
    // float32 values and the sweeps, which see every near-tie cell's float32 value p and float64 value e, keep T = max p
    // and E = max e: if A > T a cell whose float32 value is final attains A, and if E >= A a near-tie cell's float64
    // value tops every float32 value - either way the strip's maximum is max(A, E).  Only when A == T and E < A (the
    // cells that attain A are all near-tie cells and none keeps the value) the wave re-reads its finished, patched strip
    // (`rescan_max`).  Packed exact strips (per-world maxima inside lane groups) keep the select.
    constexpr bool IMAX = !(EXACT && HALO == 3);
    int amax = 0;
    dw_f32x2 accp_l = dw_f32x2(0.f), accp_d = dw_f32x2(0.f);  // the row loop sums cell PAIRS (one packed add for two cells)
    unsigned int nq = 0;                                    // entries queued by this wave (uniform)
    // packed exact strips: what the queue sweeps change in the SUMS of the wave row's worlds (an entry's world is not
    // the sweeping lane's), per wave in LDS: [world of the wave row][light, dark] (wpr <= 32).  They reach the global
    // counters only when the strip is finished WITHOUT an overflow - a strip that loses entries after an earlier sweep
    // is recomputed whole and adds every cell's full value, so corrections already sent there would count twice.
    __shared__ int s_pfix[(EXACT && PACK) ? 4 * 64 : 1];
    int* const pfix = s_pfix + ((EXACT && PACK) ? wv * 64 : 0);
    if (EXACT && PACK) {
        pfix[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    auto load_raw = [&](int rr) -> Raw {                    // rr in [r0-1, r0+nr], clamped + wrapped
        rr = min(rr, last_row);
        rr = rr < 0 ? rr + G.H : (rr >= G.H ? rr - G.H : rr);
        const plane_t* rl = pL + (size_t)rr * G.W;
        const plane_t* rd = pD + (size_t)rr * G.W;
        Raw w;
        w.l = stream_load4_raw(rl + colq);
        w.d = stream_load4_raw(rd + colq);
        if (HALO == 1 || HALO == 2) { w.hl = rl[hcol]; w.hd = rd[hcol];
        } else { w.hl = (plane_t)0.f; w.hd = (plane_t)0.f; }
        return w;
    };
    auto to_rows = [&](const Raw& w, Row4& L, Row4& D) {
        float a, c;
        const float4 l = widen4(w.l), d = widen4(w.d);
        lr_neighbours<HALO>(l, (float)w.hl, lane, last_lane, a, c, lsrc, rsrc);
        L = make_row(l, a, c);
        lr_neighbours<HALO>(d, (float)w.hd, lane, last_lane, a, c, lsrc, rsrc);
        D = make_row(d, a, c);
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
        unsigned long long tie[4];                          // lane masks (wave-uniform)
        cells4<EXACT, SYM, unsigned long long>(P, upL, miL, dnL, upD, miD, dnD, ol, od, tie);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (EXACT && HALO >= 2) tie[i] &= active_mask;
            if (IMAX) asm("v_max3_i32 %0, %0, %1, %2" : "+v"(amax) : "v"(ol[i]), "v"(od[i]));   // (the compiler builds a tree: 1.5 per cell)
            else acc_max = fmaxf(acc_max, in_mask(tie[i]) ? 0.f : fmaxf(ol[i], od[i]));
        }
        accp_l += dw_f32x2{ol[0], ol[1]} + dw_f32x2{ol[2], ol[3]};      // integers < 2^24: exact in any order
        accp_d += dw_f32x2{od[0], od[1]} + dw_f32x2{od[2], od[3]};
        if (HALO < 2 || active) {                           // HALO 0/1: every lane owns real columns
            const size_t off = woff + (size_t)(r0 + k) * G.W + colq;
            stream_store4(outL + off, make_float4(ol[0], ol[1], ol[2], ol[3]));
            stream_store4(outD + off, make_float4(od[0], od[1], od[2], od[3]));
        }
        if (EXACT) {
            queue_tie<0>(tie[0], nq, q, (unsigned)G.qcap, world, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<1>(tie[1], nq, q, (unsigned)G.qcap, world, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<2>(tie[2], nq, q, (unsigned)G.qcap, world, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
            queue_tie<3>(tie[3], nq, q, (unsigned)G.qcap, world, r0 + k, colq, upL, miL, dnL, upD, miD, dnD, ol, od);
        }
    };
    // One block = RB (2) rows: issue loads, compute the block's two rows from the four-row window, then drop two
    // loaded rows into the window slots that just became free.  In the float32-only kernel the loads run TWO
    // blocks ahead of their use (DEEP: a block loads what the block after next needs and consumes what the
    // previous block loaded; a binary16 row in flight costs 6 VGPRs, and with one block of slack the kernel was
    // bound by its bytes in flight - 4 waves x 2 rows x 1 KB per SIMD: C2 0.135 -> 0.122 ms, 64 x 1024^2 -11 %).
    // The exact kernel is at its 168-VGPR budget (3 waves/SIMD): the extra rows in flight spilled inside the row
    // loop (+12...+17 % time), so it keeps one block of slack.  Window and load buffers are rings: the block
    // alternates between the slot orders (0,1,2,3) / (2,3,0,1) and the buffer pairs (A,B) / (B,A) - written out
    // as two phases - instead of shifting rows from slot to slot (32 register moves per block).
    static_assert(RB == 2, "the ring below is written for two-row blocks");
    constexpr bool DEEP = !EXACT;
    Raw rawA[2], rawB[2];
    if (DEEP) {
        rawA[0] = load_raw(r0 + 3);                         // rows 3, 4 of the strip: consumed by the first block
        rawA[1] = load_raw(r0 + 4);
    }
    auto block = [&](auto PHc, int kk, Raw (&use)[2], Raw (&fill)[2]) {
        constexpr int PH = decltype(PHc)::value;           // physical slot of the block's first window row
        constexpr int s0 = PH, s1 = (PH + 1) & 3, s2 = (PH + 2) & 3, s3 = (PH + 3) & 3;
        if (DEEP) {
            fill[0] = load_raw(r0 + kk + 5);                // for the NEXT block (clamped to the strip's halo row)
            fill[1] = load_raw(r0 + kk + 6);
        } else {
            use[0] = load_raw(r0 + kk + 3);                 // consumed at the end of this block
            use[1] = load_raw(r0 + kk + 4);
        }
        __builtin_amdgcn_sched_barrier(0);                  // loads first, then the arithmetic
        row_math(WL[s0], WL[s1], WL[s2], WD[s0], WD[s1], WD[s2], kk);
        row_math(WL[s1], WL[s2], WL[s3], WD[s1], WD[s2], WD[s3], kk + 1);
        __builtin_amdgcn_sched_barrier(0);
        to_rows(use[0], WL[s0], WD[s0]);                    // rows kk+3, kk+4: window rows 2, 3 of the next block
        to_rows(use[1], WL[s1], WD[s1]);
    };
    // Exact mode: float64 re-evaluation of this wave's queued near-tie cells, one entry per lane - when the strip is
    // finished and whenever the queue is half full, so that it never overflows on states the dynamics produce (a
    // strip whose queue did overflow lost entries and is recomputed whole in float64 below: tens of times slower).
    bool redo = false;
    const int flush_at = G.qcap >> 1;
    float fix_max = 0.f, fix_l = 0.f, fix_d = 0.f;          // what the sweeps change in this lane's reductions (their
                                                            // entries are not the lane's own cells: kept apart from acc_*)
    float fix_tmax = 0.f;                                   // T: the largest float32 value of a swept entry
    unsigned int swept = 0;                                 // entries swept so far (uniform)
    auto sweep_queue = [&]() {
        if (nq > (unsigned)G.qcap) redo = true;
        if (!redo && nq) {
            wait_row_stores_before_patching();                  // patches land inside rows this wave stored
            const PhysF64* p64 = &P64;                          // cold constants: loaded here, not held across the row loop
            asm volatile("" : "+s"(p64));
            const PhysF64 Pe = *p64;
            for (unsigned int e = lane; e < nq; e += 64) {
                const uint4 e0 = q[e * 3], e1 = q[e * 3 + 1], e2 = q[e * 3 + 2];
                const unsigned int w[9] = {unpack_ld(e0.z), unpack_ld(e0.w), unpack_ld(e1.x),
                                           unpack_ld(e1.y), unpack_ld(e1.z), unpack_ld(e1.w),
                                           unpack_ld(e2.x), unpack_ld(e2.y), unpack_ld(e2.z)};
                const unsigned int f32v = unpack_ld(e2.w);
                const NewCoverF64 o = cell_f64_lean(Pe, w);
                const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
                const size_t off = (size_t)e0.x * G.H * G.W + (size_t)(e0.y >> 16) * G.W + (e0.y & 0xffffu);
                outL[off] = (plane_t)kl;                    // after this wave's own row store (see wait_row_stores_before_patching)
                outD[off] = (plane_t)kd;
                if (PACK) {                                 // the entry's world is not this lane's
                    atomicMax(&stats[e0.x].max_k, (unsigned int)fmaxf(kl, kd));   // an exact final value: right in any case
                    const int pwe = (int)e0.x - b * G.wpr;
                    const int dl = (int)(kl - (float)(f32v & 0xffffu)), dd = (int)(kd - (float)(f32v >> 16));
                    if (dl) atomicAdd(&pfix[2 * pwe], dl);
                    if (dd) atomicAdd(&pfix[2 * pwe + 1], dd);
                } else {
                    const float pl = (float)(f32v & 0xffffu), pd = (float)(f32v >> 16);
                    fix_l += kl - pl;
                    fix_d += kd - pd;
                    fix_max = fmaxf(fix_max, fmaxf(kl, kd));
                    fix_tmax = fmaxf(fix_tmax, fmaxf(pl, pd));
                }
            }
            if (lane == 0) atomicAdd(fixups, (unsigned long long)nq);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the queue is reused: reads above before later pushes
            __builtin_amdgcn_wave_barrier();
            swept += nq;
        }
        nq = 0;
    };
    using PH0 = std::integral_constant<int, 0>;
    using PH2 = std::integral_constant<int, 2>;
    int k = 0;
    for (; k + 4 <= nr; k += 4) {
        block(PH0{}, k, rawA, rawB);
        block(PH2{}, k + 2, rawB, rawA);
        if (EXACT && __builtin_expect(nq >= (unsigned)flush_at, 0)) sweep_queue();    // wave-uniform; rare
    }
    bool odd_phase = false;
    if (k + 2 <= nr) { block(PH0{}, k, rawA, rawB); k += 2; odd_phase = true; }
    if (k < nr) {                                           // one row left, already in the window
        if (odd_phase) row_math(WL[2], WL[3], WL[0], WD[2], WD[3], WD[0], k);
        else row_math(WL[0], WL[1], WL[2], WD[0], WD[1], WD[2], k);
    }
    acc_l = accp_l.x + accp_l.y;
    acc_d = accp_d.x + accp_d.y;
    if (IMAX) acc_max = __int_as_float(amax);
    if (HALO >= 2 && !active) { acc_max = 0.f; acc_l = 0.f; acc_d = 0.f; }

    // ---- exact mode: the entries of the strip's last rows; a strip that lost entries: whole in float64 ----
    if (EXACT) {
        sweep_queue();
        if (!redo) {
            acc_l += fix_l;
            acc_d += fix_d;
            if (IMAX) {
                // A / T / E of the comment at `amax`; all three wave-uniform after the shuffles
                const float aw = wave_max(acc_max), tw = wave_max(fix_tmax), ew = wave_max(fix_max);
                // aw > tw: a cell whose float32 value is final attains A; ew >= aw: some near-tie cell's float64 value
                // is at least A, hence at least every final float32 value.  Either way the maximum is max(A, E).
                if (__builtin_expect((swept != 0 && aw == tw && ew < aw) || G.force_rescan, 0)) {
                    // rescan_max: the only cells that attain A are near-tie cells, and none of them keeps the value
                    // (on developed states: well under 1 % of the strips).  Every row store and every patch of this
                    // wave has reached L2 (vmcnt(0)); agent-scope loads read them back past the vector cache, eight
                    // rows in flight at a time.
                    wait_row_stores_before_patching();
                    float m = 0.f;
                    for (int k0 = 0; k0 < nr; k0 += 8) {
                        unsigned long long wl[8], wd[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int kk = min(k0 + j, nr - 1);
                            const size_t off = woff + (size_t)(r0 + kk) * G.W + colq;
                            wl[j] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(outL + off), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                            wd[j] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(outD + off), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float4 fl = widen4(__builtin_bit_cast(dw_f16x4, wl[j])), fd = widen4(__builtin_bit_cast(dw_f16x4, wd[j]));
                            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fl.x, fl.y), fmaxf(fl.z, fl.w)), fmaxf(fmaxf(fd.x, fd.y), fmaxf(fd.z, fd.w))));
                        }
                    }
                    acc_max = (HALO < 2 || active) ? m : 0.f;
                } else {
                    acc_max = fmaxf(aw, ew);
                }
            } else {
                acc_max = fmaxf(acc_max, fix_max);
            }
            if (PACK) {                                     // the sweeps' corrections of the wave row's worlds
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane < G.wpr && b * G.wpr + lane < G.B) {
                    const int dl = pfix[2 * lane], dd = pfix[2 * lane + 1];
                    if (dl) atomicAdd(&stats[b * G.wpr + lane].sum_l, (unsigned long long)(long long)dl);
                    if (dd) atomicAdd(&stats[b * G.wpr + lane].sum_d, (unsigned long long)(long long)dd);
                }
            }
        } else {                                            // queue overflow: the whole strip in float64
            acc_max = 0.f; acc_l = 0.f; acc_d = 0.f;
            const int nc = PACK ? 256 : min(256, G.W - c0);
            for (int i = lane; i < nr * nc; i += 64) {
                const int r = r0 + i / nc;
                int c = c0 + i % nc;
                size_t wo = woff;
                int wi = b;
                if (PACK) {                                 // column i % 256 of the wave row -> (world, column)
                    const int pwc = c / G.W;
                    wi = b * G.wpr + pwc;
                    c -= pwc * G.W;
                    if (pwc >= G.wpr || wi >= G.B) continue;
                    wo = (size_t)wi * G.H * G.W;
                }
                double l9[9], d9[9];
                gather9(inL + wo, G.H, G.W, r, c, l9);
                gather9(inD + wo, G.H, G.W, r, c, d9);
                const CellF64 o = cell_f64(P64, l9, d9);
                const float kl = (float)dw_round3_k(o.nl), kd = (float)dw_round3_k(o.nd);
                outL[wo + (size_t)r * G.W + c] = (plane_t)kl;
                outD[wo + (size_t)r * G.W + c] = (plane_t)kd;
                if (PACK) {
                    atomicMax(&stats[wi].max_k, (unsigned int)fmaxf(kl, kd));
                    atomicAdd(&stats[wi].sum_l, (unsigned long long)kl);
                    atomicAdd(&stats[wi].sum_d, (unsigned long long)kd);
                } else {
                    acc_l += kl; acc_d += kd;
                    acc_max = fmaxf(acc_max, fmaxf(kl, kd));
                }
            }
        }
    }

    // per-world reductions of this strip: wavefront shuffles, three atomics per strip (packed mode: per
    // world of the wave row - a butterfly inside each group of lpw lanes, lpw a power of two)
    if (PACK) {
        float m = acc_max, sl = acc_l, sd = acc_d;
        if ((G.lpw & (G.lpw - 1)) == 0) {                       // power-of-two groups: butterfly
            for (int o = G.lpw >> 1; o > 0; o >>= 1) {
                m = fmaxf(m, __shfl_xor(m, o, 64));
                sl += __shfl_xor(sl, o, 64);
                sd += __shfl_xor(sd, o, 64);
            }
        } else {                                                // any group size: the group's first lane gathers
            for (int o = 1; o < G.lpw; ++o) {
                const int src = min(lane + o, 63);
                m = fmaxf(m, __shfl(acc_max, src, 64));
                sl += __shfl(acc_l, src, 64);
                sd += __shfl(acc_d, src, 64);
            }
        }
        if (pj == 0 && active) {
            atomicMax(&stats[world].max_k, (unsigned int)m);
            atomicAdd(&stats[world].sum_l, (unsigned long long)sl);
            atomicAdd(&stats[world].sum_d, (unsigned long long)sd);
        }
    } else {
        const float m = wave_max(acc_max);
        const float sl = wave_sum(acc_l), sd = wave_sum(acc_d);
        if (lane == 0) {
            atomicMax(&stats[b].max_k, (unsigned int)m);
            atomicAdd(&stats[b].sum_l, (unsigned long long)sl);
            atomicAdd(&stats[b].sum_d, (unsigned long long)sd);
        }
    }


}

// Two entry points so that each arithmetic mode gets its own register budget: the float32-only
// kernel fits 4 waves per SIMD with 2-row blocks; the exact kernel carries the tie test and the
// fix-up path and is planned for 3 waves per SIMD (<= 168 VGPRs; its 48 KB of LDS queues per
// workgroup allow 3 workgroups per CU as well).
template <int HALO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void step_stream_fast(const plane_t* __restrict__ inL, const plane_t* __restrict__ inD,
                                                        plane_t* __restrict__ outL, plane_t* __restrict__ outD,
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
    const plane_t* inL; const plane_t* inD; plane_t* outL; plane_t* outD;
    StripGeom G; PhysF32 P; StatsDev* stats; unsigned long long* fixups; unsigned long long* zero_me; int zero_n;
    PhysF64 P64;                                                  // cold
};

template <int HALO, bool SYM = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DW_STREAM_WAVES_EXACT, DW_STREAM_WAVES_EXACT)))
void step_stream_exact(StreamExactArgs A) {
    const StreamExactArgs& cold = kernarg_struct<StreamExactArgs>();
    stream_body<true, HALO, DW_STREAM_RB_EXACT, SYM>(A.inL, A.inD, A.outL, A.outD, A.G, A.P, cold.P64, A.stats, A.fixups,
                                                     A.zero_me, A.zero_n);
}

}  // namespace dw
