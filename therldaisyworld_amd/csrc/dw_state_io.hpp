// dw_state_io.hpp — state in and out of the device layout: the reference's 7-channel float64 grid and
// its side-effect caches (materialise), the tie-bound audit, Philox initial states, plane conversions and
// the stand-alone per-world reductions.
#pragma once
#include "dw_common.hpp"

namespace dw {

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
                                                   const plane_t* __restrict__ cL,
                                                   const plane_t* __restrict__ cD, int H, int W,
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
            g[0 * n] = dw_div1000(dw_round3_k(P.p - o.nl - o.nd));
            g[1 * n] = to_natural(cL[woff + cell]);
            g[2 * n] = to_natural(cD[woff + cell]);
            g[3 * n] = dw_div1000(dw_round3_k(o.T));
            g[4 * n] = dw_div1000(dw_round3_k(o.Tl));
            g[5 * n] = dw_div1000(dw_round3_k(o.Td));
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
// tie_audit — evidence for the exact mode's error bound.  For every cell of a quantised state:
// float32 per-mille growth gq32 (the hot kernels' arithmetic, split coefficient chains) against the
// float64 growth of cell_f64, and the per-cell bound eps the tie test would use.  Reduces
//   out[0] = max |gq32 - gq64| (quanta)      out[1] = max (|gq32 - gq64| / eps)   (< 1 <=> bound holds)
//   out[2] = number of cells the tie test flags   out[3] = number of cells audited
// (both species count).  Non-negative doubles order like their bit patterns: atomicMax on u64.
// ---------------------------------------------------------------------------------------------
// `sym`: audit the two-term coefficient chain the exact wave-strip kernels use for symmetric albedos (growth_t<.., SYM>)
__global__ __launch_bounds__(256) void tie_audit(const plane_t* __restrict__ L, const plane_t* __restrict__ D, int H, int W,
                                                 PhysF32 P, PhysF64 P64, unsigned long long* __restrict__ out, int sym) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H * W) return;
    const size_t woff = (size_t)b * H * W;
    const plane_t* pl = L + woff;
    const plane_t* pd = D + woff;
    const int r = cell / W, c = cell - r * W;
    const int ru = r == 0 ? H - 1 : r - 1, rd = r == H - 1 ? 0 : r + 1;
    const int cl = c == 0 ? W - 1 : c - 1, cr = c == W - 1 ? 0 : c + 1;
#define DW_AT(p, rr, cc) (float)(p)[(size_t)(rr) * W + (cc)]
    const float li = DW_AT(pl, r, c), di = DW_AT(pd, r, c);
    const float El = (DW_AT(pl, ru, c) + DW_AT(pl, rd, c)) + (DW_AT(pl, r, cl) + DW_AT(pl, r, cr));
    const float Cl = (DW_AT(pl, ru, cl) + DW_AT(pl, rd, cl)) + (DW_AT(pl, ru, cr) + DW_AT(pl, rd, cr));
    const float Ed = (DW_AT(pd, ru, c) + DW_AT(pd, rd, c)) + (DW_AT(pd, r, cl) + DW_AT(pd, r, cr));
    const float Cd = (DW_AT(pd, ru, cl) + DW_AT(pd, rd, cl)) + (DW_AT(pd, ru, cr) + DW_AT(pd, rd, cr));
#undef DW_AT
    const GrowthF32 g = sym ? growth_f32<true, true>(P, li, di, El, Cl, Ed, Cd) : growth_f32<true, false>(P, li, di, El, Cl, Ed, Cd);
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
        const float thr = fmaf(-fabsf(dK[k]), fmaf(fabsf(P.eK1s), om[k], fabsf(P.eK0s)), fmaf(-P.eA, fabsf(g32[k]), P.tie_lo));
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

// ---------------------------------------------------------------------------------------------
// init_random — ref initialize_grid :287-302 / initialize_agents :175-179 with Philox4x32-10.
// counter = (cell lo, cell hi, world lo, world hi), key = seed.  One call per cell gives the four
// uniforms (U1_dark, U2_dark, U1_light, U2_light); the reference draws dark first.
// ---------------------------------------------------------------------------------------------
// Round 4: a workgroup draws kInitChunk consecutive cells of one world, four adjacent cells per thread and pass (one
// 16-byte / 8-byte store per plane), and - the values are in registers - reduces them on the spot into the world's
// statistics: wavefront shuffles, LDS across the four waves, three atomics per workgroup (the separate stats_only pass
// after dw_init_random read the 128 GiB of a 1024 x 4096^2 state back: 81 ms on top of the 37 ms of the draw).
// OutT = float: the un-quantised per-mille state; OutT = plane_t: the same draw rounded to three decimals (np.round(., 3):
// rint of the per-mille value) straight into binary16 planes (dw_init_random_quantised).
constexpr int kInitChunk = 16384;
__device__ __forceinline__ void block_stats_commit(float m, float sl, float sd, StatsDev* __restrict__ st) {
    __shared__ float red[3][4];
    m = wave_max(m);
    sl = wave_sum(sl);
    sd = wave_sum(sd);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = m; red[1][wv] = sl; red[2][wv] = sd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float mm = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        const float a = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float c = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        atomicMax(&st->max_k, (unsigned int)ceilf(mm));
        atomicAdd(&st->sum_l, (unsigned long long)(a + 0.5f));
        atomicAdd(&st->sum_d, (unsigned long long)(c + 0.5f));
    }
}

template <typename OutT>
__global__ __launch_bounds__(256) void init_random_cells(OutT* __restrict__ L, OutT* __restrict__ D, int n,
                                                         long long world_offset, unsigned long long seed,
                                                         float light_prop, float dark_prop, float ial, float iad,
                                                         StatsDev* __restrict__ stats) {
    constexpr bool QUANT = !std::is_same<OutT, float>::value;
    const int b = blockIdx.y;
    const unsigned long long world = (unsigned long long)(world_offset + b);
    const size_t wbase = (size_t)b * n;
    const bool vec = (n & 3) == 0;                              // every group of four cells is whole and aligned
    float m = 0.f, sl = 0.f, sd = 0.f;
    for (int it = 0; it < kInitChunk / 1024; ++it) {
        const int cell0 = blockIdx.x * kInitChunk + it * 1024 + 4 * threadIdx.x;
        if (cell0 >= n) break;
        float l4[4], d4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cell = cell0 + e;
            uint32_t r[4];
            philox4x32_10((uint32_t)cell, 0u, (uint32_t)world, (uint32_t)(world >> 32), (uint32_t)seed,
                          (uint32_t)(seed >> 32), r);
            const float d = (u01(r[0]) < dark_prop) ? iad * u01(r[1]) : 0.f;
            const float l = (u01(r[2]) < light_prop) ? ial * u01(r[3]) : 0.f;
            const bool in = cell < n;
            l4[e] = in ? (QUANT ? __builtin_rintf(l * 1000.f) : l * 1000.f) : 0.f;
            d4[e] = in ? (QUANT ? __builtin_rintf(d * 1000.f) : d * 1000.f) : 0.f;
            m = fmaxf(m, fmaxf(l4[e], d4[e]));
            sl += l4[e];
            sd += d4[e];
        }
        if (vec) {
            if constexpr (QUANT) {
                stream_store4(L + wbase + cell0, make_float4(l4[0], l4[1], l4[2], l4[3]));
                stream_store4(D + wbase + cell0, make_float4(d4[0], d4[1], d4[2], d4[3]));
            } else {
                *reinterpret_cast<float4*>(L + wbase + cell0) = make_float4(l4[0], l4[1], l4[2], l4[3]);
                *reinterpret_cast<float4*>(D + wbase + cell0) = make_float4(d4[0], d4[1], d4[2], d4[3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (cell0 + e < n) { L[wbase + cell0 + e] = (OutT)l4[e]; D[wbase + cell0 + e] = (OutT)d4[e]; }
        }
    }
    block_stats_commit(m, sl, sd, &stats[b]);
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

// conv3x3_f64 — ref ft_convolve (daisy/nn/functional.py:12-49) for a 3x3 kernel: the toroidal TRUE convolution
// out[i][j] = sum_{a,b} k[a][b] * x[i-(a-1)][j-(b-1)] it computes by FFT, evaluated directly in
// float64, row-major taps in ascending (a, b).  Serves the stand-alone calculate_albedo / calculate_daisy_density
// methods of the drop-in; the step kernels never call it (they fuse both stencils).
struct Kernel9 { double k[9]; };
__global__ __launch_bounds__(256) void conv3x3_f64(const double* __restrict__ x, double* __restrict__ out, int H, int W,
                                                   Kernel9 K) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H * W) return;
    const size_t woff = (size_t)b * H * W;
    const int i = cell / W, j = cell - i * W;
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ii = (i - (a - 1) + H) % H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (K.k[a * 3 + c] == 0.0) continue;
            const int jj = (j - (c - 1) + W) % W;
            acc += K.k[a * 3 + c] * x[woff + (size_t)ii * W + jj];
        }
    }
    out[woff + cell] = acc;
}

// stage_f64 — the stages of the reference's forward() as stand-alone float64 maps on CALLER data (the drop-in's
// calculate_* methods; forward() / step() never use them: they run the fused map).  Planes are [plane][B][H][W]
// float64, one thread per cell.  K9 = the 3x3 kernel of the stencil stages, taps in the order of conv3x3_f64.
//   1 albedo       ref :377-394  in  bare (ignored), light, dark   out local albedo, adjacent albedo, bare = p - l - d
//   2 density      ref :423-432  in  light, dark                   out density_light, density_dark
//   3 temperature  ref :396-421  in  local albedo, adjacent albedo out T_effective, T, T_light, T_dark
//   4 growth rate  ref :340-348  in  T, T_light, T_dark            out beta, beta_l, beta_d
//   5 growth       ref :350-375  in  beta_l, beta_d, dens_l, dens_d out growth_light, growth_dark
enum { kStageAlbedo = 1, kStageDensity = 2, kStageTemperature = 3, kStageGrowthRate = 4, kStageGrowth = 5 };
__host__ __device__ constexpr int stage_inputs(int s) { return s == 1 ? 3 : (s == 2 ? 2 : (s == 3 ? 2 : (s == 4 ? 3 : 4))); }
__host__ __device__ constexpr int stage_outputs(int s) { return s == 1 ? 3 : (s == 2 ? 2 : (s == 3 ? 4 : (s == 4 ? 3 : 2))); }

template <typename F>
__device__ __forceinline__ double torus_conv9(const Kernel9& K, int i, int j, int H, int W, F value_at) {
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ii = (i - (a - 1) + H) % H;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (K.k[a * 3 + c] == 0.0) continue;
            const int jj = (j - (c - 1) + W) % W;
            acc += K.k[a * 3 + c] * value_at((size_t)ii * W + jj);
        }
    }
    return acc;
}

template <int STAGE>
__global__ __launch_bounds__(256) void stage_f64(const double* __restrict__ in, double* __restrict__ out, int B, int H,
                                                 int W, PhysF64 P, Kernel9 K) {
    const int b = blockIdx.y;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H * W) return;
    const size_t plane = (size_t)B * H * W, woff = (size_t)b * H * W, o = woff + cell;
    const int i = cell / W, j = cell - i * W;
    if constexpr (STAGE == kStageAlbedo) {
        const double* l = in + plane + woff;
        const double* d = in + 2 * plane + woff;
        const double bare = P.p - l[cell] - d[cell];
        double local = 0.0, adjacent = 0.0;                     // accumulated over bare, light, dark in this order
        local += P.ab * bare;
        adjacent += P.ab * torus_conv9(K, i, j, H, W, [&](size_t q) { return P.p - l[q] - d[q]; });
        local += P.al * l[cell];
        adjacent += P.al * torus_conv9(K, i, j, H, W, [&](size_t q) { return l[q]; });
        local += P.ad * d[cell];
        adjacent += P.ad * torus_conv9(K, i, j, H, W, [&](size_t q) { return d[q]; });
        out[o] = local;
        out[plane + o] = adjacent;
        out[2 * plane + o] = bare;
    } else if constexpr (STAGE == kStageDensity) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double* x = in + c * plane + woff;
            out[c * plane + o] = torus_conv9(K, i, j, H, W, [&](size_t q) { return x[q]; });
        }
    } else if constexpr (STAGE == kStageTemperature) {
        const double Al = in[o], A = in[plane + o];
        const double Te = pow((P.S * P.L * (1.0 - A)) / P.sigma, 0.25);
        const double T = pow(P.q * (A - Al) + dw_pow4(Te), 0.25);
        const double T4 = dw_pow4(T);
        out[o] = Te;
        out[plane + o] = T;
        out[2 * plane + o] = pow(P.q2 * (Al - P.al) + T4, 0.25);
        out[3 * plane + o] = pow(P.q2 * (Al - P.ad) + T4, 0.25);
    } else if constexpr (STAGE == kStageGrowthRate) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double dT = P.To - in[c * plane + o];
            out[c * plane + o] = 1.0 - P.g * (dT * dT);
        }
    } else {
        const double bl = in[o], bd = in[plane + o], kl = in[2 * plane + o], kd = in[3 * plane + o];
        const double kb = P.p - kl - kd;
        out[o] = kl * (kb * bl - P.gamma);
        out[plane + o] = kd * (kb * bd - P.gamma);
    }
}

// plane conversions
// natural-unit float32 upload -> un-quantised per-mille float32 (in place allowed)
__global__ void f32nat_to_permille(const float* __restrict__ in, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1000.f;
}
// natural-unit float32 upload of a QUANTISED state -> canonical binary16 plane
__global__ void f32nat_to_plane(const float* __restrict__ in, plane_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (plane_t)__builtin_rintf(in[i] * 1000.f);
}
// any plane format -> natural-unit float64 (downloads)
template <typename T>
__global__ void plane_to_f64(const T* __restrict__ in, double* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_natural(in[i]);
}

// stats of an arbitrary state (after uploads, so that dw_reduce is always valid): a workgroup reduces kInitChunk
// cells of world blockIdx.y with 16-byte loads per lane and plane, three atomics per workgroup
template <typename T>
__global__ __launch_bounds__(256) void stats_only(const T* __restrict__ L, const T* __restrict__ D, int n,
                                                  StatsDev* __restrict__ stats) {
    constexpr int VEC = 16 / sizeof(T);                         // cells per 16-byte load: 8 binary16, 4 float32, 2 float64
    struct alignas(16) Pack { T v[VEC]; };
    const int b = blockIdx.y;
    const size_t wbase = (size_t)b * n;
    const bool vec = n % VEC == 0;
    float m = 0.f, sl = 0.f, sd = 0.f;
    for (int it = 0; it < kInitChunk / (256 * VEC); ++it) {
        const int cell0 = blockIdx.x * kInitChunk + (it * 256 + threadIdx.x) * VEC;
        if (cell0 >= n) break;
        Pack pl, pd;
        if (vec) {
            pl = *reinterpret_cast<const Pack*>(L + wbase + cell0);
            pd = *reinterpret_cast<const Pack*>(D + wbase + cell0);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const bool in = cell0 + e < n;
                pl.v[e] = in ? L[wbase + cell0 + e] : (T)0;
                pd.v[e] = in ? D[wbase + cell0 + e] : (T)0;
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float kl = to_permille(pl.v[e]), kd = to_permille(pd.v[e]);
            m = fmaxf(m, fmaxf(kl, kd));
            sl += kl;
            sd += kd;
        }
    }
    block_stats_commit(m, sl, sd, &stats[b]);
}

}  // namespace dw
