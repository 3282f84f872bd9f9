// dw_common.hpp — shared device helpers of the RLDaisyWorld kernels (gfx950): per-world reduction record,
// wavefront reductions, input adaptors, and the four-cell row group (`Row4`, `cells4`) that the tiled,
// wave-strip and fused step kernels evaluate with packed float32 arithmetic.
#pragma once
#include <type_traits>

#include "dw_physics.hpp"

namespace dw {

struct StatsDev {             // mirrors dw_world_stats
    unsigned int max_k;
    unsigned int reserved;    // the one-wave-per-world episode kernels: float64 re-evaluations of the world's last step
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
// Plane formats.
//   plane_t (binary16)  the canonical format of a QUANTISED state: k = 1000 * cover is an integer in [0, 1000]
//                       after any step (np.round(., 3), ref daisy_world_rl.py:452), and binary16 holds every
//                       integer up to 2048 exactly - lossless, 2 bytes per value, 8 bytes of HBM traffic per
//                       cell-update (2 planes read + 2 written).  Every step kernel reads and writes it.
//   float  (per-mille)  an UN-quantised state from dw_init_random / dw_upload_state_f32(quantised = 0)
//   double (natural)    an un-quantised state from dw_upload_state_f64 (the reference's own initial grid)
// The two un-quantised formats exist only until the first step has consumed them (then one step longer as the
// "previous state" observations are derived from); the adaptors below let the cold kernels read all three.
// ---------------------------------------------------------------------------------------------
typedef _Float16 plane_t;
__device__ __forceinline__ double to_natural(double x) { return x; }
__device__ __forceinline__ double to_natural(float k) { return (double)k / 1000.0; }
// (a binary16 plane holds per-mille INTEGERS: the division-free form is the correctly rounded k / 1000.0, dw_physics.hpp)
__device__ __forceinline__ double to_natural(plane_t k) { return dw_permille_to_natural((double)(float)k); }
__device__ __forceinline__ float to_permille(double x) { return (float)(x * 1000.0); }
__device__ __forceinline__ float to_permille(float k) { return k; }
__device__ __forceinline__ float to_permille(plane_t k) { return (float)k; }

// streaming accesses of four adjacent cells of a binary16 plane (8 bytes).  The new planes are not read again
// within the step, so they are stored non-temporally (measured -1.5 % / -6 % on C2; non-temporal LOADS were
// slower).  One v_cvt_pkrtz_f16_f32 packs two cells (exact whatever its rounding mode: integers <= 1000),
// v_cvt_f32_f16 with an SDWA half-select unpacks one.
typedef _Float16 dw_f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int dw_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dw_f16x4 stream_load4_raw(const plane_t* p) { return *reinterpret_cast<const dw_f16x4*>(p); }
__device__ __forceinline__ float4 widen4(const dw_f16x4& v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
__device__ __forceinline__ float4 stream_load4(const plane_t* p) { return widen4(stream_load4_raw(p)); }
__device__ __forceinline__ void stream_store4(plane_t* p, const float4& v) {
    dw_u32x2 t;
    t.x = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
    t.y = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(v.z, v.w));
    __builtin_nontemporal_store(t, reinterpret_cast<dw_u32x2*>(p));
}

// a coordinate at most one period outside [0, n) back onto the torus (agents move by one cell, stencils reach two)
__device__ __forceinline__ int wrap_near(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
// ... at most TWO periods outside (a repair reaches four rows beyond a strip of a world that may be only three rows tall)
__device__ __forceinline__ int wrap_near2(int v, int n) { return wrap_near(wrap_near(v, n), n); }

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
// `tie`: per-lane bools (bool*), or - the wave-strip kernels - the wave's lane masks (unsigned long long*), which
// stay on the scalar unit from the compare to the queue push.
template <bool EXACT, bool SYM = false, typename F = bool>
__device__ __forceinline__ void cells4(const PhysF32& P, const Row4& upL, const Row4& miL, const Row4& dnL,
                                       const Row4& upD, const Row4& miD, const Row4& dnD, float* ol, float* od,
                                       F* tie) {
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
        const GrowthT<T> g = growth_t<EXACT || kFastSplit, T, EXACT && SYM>(P, li, di, El, Cl, Ed, Cd);
        T vl, vd;
        if (EXACT) {
            F tl[N], td[N];
            if constexpr (DW_TIE_FROM_BETA || sizeof(F) == 8) {
                vl = finish_exact_beta_t<T, F>(P, li, g.gql, g.dKl, g.bl, tl);
                vd = finish_exact_beta_t<T, F>(P, di, g.gqd, g.dKd, g.bd, td);
            } else {
                vl = finish_exact_t<T>(P, li, g.gql, g.dKl, g.oml, tl);
                vd = finish_exact_t<T>(P, di, g.gqd, g.dKd, g.omd, td);
            }
#pragma unroll
            for (int e = 0; e < N; ++e) tie[i + e] = tl[e] | td[e];
        } else {
            vl = finish_fast_t<T>(li, g.dKl, g.fl);
            vd = finish_fast_t<T>(di, g.dKd, g.fd);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            ol[i + e] = Lanes<T>::get(vl, e);
            od[i + e] = Lanes<T>::get(vd, e);
        }
    }
}

}  // namespace dw
