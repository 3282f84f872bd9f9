// dw_physics.hpp — per-cell arithmetic of the RLDaisyWorld physics pass for gfx950.
//
// Two evaluations of the same map (reference
// daisy/daisy_world_rl.py:340-461):
//
//   cell_f64  float64, staged like the reference (albedo -> temperature -> growth-rate ->
//             growth -> clip).  Used by the DW_PRECISION_F64 kernels, by the materialise /
//             observation kernels, and as the tie "fix-up" of the exact mode.
//   growth_f32 float32, the fused algebra of the hot kernel, working in PER-MILLE units so that
//             every stencil sum of a quantised state is an exact small integer.
//
// Units: the device planes hold k = 1000 * cover.  The reference quantises its state with
// np.round(., 3) (daisy_world_rl.py:452), so after any step k is an integer in [0, 1000].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dw {

// ---------------------------------------------------------------------------------------------
// float64 constants (host fills; passed to kernels by value)
// ---------------------------------------------------------------------------------------------
struct PhysF64 {
    double p, g, S, sigma, gamma, q, q2, dt;
    double ab, al, ad, To;       // albedo bare/light/dark, optimal temperature
    double L;                    // luminosity of this pass
    double w0, w1, w2;           // daisy kernel centre / edge / corner (ref :270-273)
};

// everything the reference's forward() derives for one cell
struct CellF64 {
    double nl, nd;               // clip(cover + dt*growth, 0, 1), NOT yet rounded (ref :449)
    double T, Tl, Td, Te;        // temp, temp_light, temp_dark, temp_effective (ref :404-413)
    double b, bl, bd;            // beta, beta_l, beta_d (ref :342-344)
    double gl, gd;               // growth (ref :366-367)
};

__host__ __device__ inline double dw_root4(double x) { return sqrt(sqrt(x)); }
__host__ __device__ inline double dw_pow4(double x) { double s = x * x; return s * s; }

// np.round(x, 3) = rint(x * 1000) / 1000, ties to even (ref :452).  Returns the integer k.
__host__ __device__ inline double dw_round3_k(double x) { return rint(x * 1000.0); }

// l[9], d[9]: the 3x3 neighbourhood in natural units, row-major, index 4 = the cell itself.
__host__ __device__ inline CellF64 cell_f64(const PhysF64& P, const double* l, const double* d) {
    CellF64 o;
    // ref calculate_albedo :377-394 — bare = p - l - d everywhere, adjacent = mean of the 8
    // Moore neighbours per cover type, weighted by the albedos in the order bare, light, dark.
    double sb = 0.0, sl = 0.0, sd = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i == 4) continue;
        sb += 0.125 * (P.p - l[i] - d[i]);
        sl += 0.125 * l[i];
        sd += 0.125 * d[i];
    }
    const double bare = P.p - l[4] - d[4];
    double Al = 0.0, A = 0.0;
    Al += P.ab * bare; A += P.ab * sb;
    Al += P.al * l[4]; A += P.al * sl;
    Al += P.ad * d[4]; A += P.ad * sd;
    // ref calculate_temperature :396-421
    o.Te = dw_root4((P.S * P.L * (1.0 - A)) / P.sigma);
    o.T = dw_root4(P.q * (A - Al) + dw_pow4(o.Te));
    const double T4 = dw_pow4(o.T);
    o.Tl = dw_root4(P.q2 * (Al - P.al) + T4);
    o.Td = dw_root4(P.q2 * (Al - P.ad) + T4);
    // ref calculate_growth_rate :340-348
    o.b = 1.0 - P.g * (P.To - o.T) * (P.To - o.T);
    o.bl = 1.0 - P.g * (P.To - o.Tl) * (P.To - o.Tl);
    o.bd = 1.0 - P.g * (P.To - o.Td) * (P.To - o.Td);
    // ref calculate_daisy_density :423-432 (symmetric 3-weight kernel)
    const double kl = P.w0 * l[4] + P.w1 * (l[1] + l[3] + l[5] + l[7]) + P.w2 * (l[0] + l[2] + l[6] + l[8]);
    const double kd = P.w0 * d[4] + P.w1 * (d[1] + d[3] + d[5] + d[7]) + P.w2 * (d[0] + d[2] + d[6] + d[8]);
    // ref calculate_growth :350-375
    const double kb = P.p - kl - kd;
    o.gl = kl * (kb * o.bl - P.gamma);
    o.gd = kd * (kb * o.bd - P.gamma);
    // ref forward :449
    double nl = l[4] + P.dt * o.gl, nd = d[4] + P.dt * o.gd;
    o.nl = nl < 0.0 ? 0.0 : (nl > 1.0 ? 1.0 : nl);
    o.nd = nd < 0.0 ? 0.0 : (nd > 1.0 ? 1.0 : nd);
    return o;
}

// k / 1000.0 for an integer k in [0, 65535] WITHOUT a division: one Newton correction of k * 0.001
// is the correctly rounded quotient for every such k (checked exhaustively; tests/test_gpu_parity).
__host__ __device__ inline double dw_permille_to_natural(double k) {
    const double r = 0.001;
    const double q = k * r;
    return fma(fma(-q, 1000.0, k), r, q);
}

// k / 1000.0 for any double k holding an integer (a rounded temperature in milli-kelvin, a per-mille cover): the Newton
// form above is the correctly rounded quotient for every |k| < 2^22 (checked exhaustively: tests/test_abi_and_host.py
// compiles the loop); anything larger - non-physical parameters - takes the division.
__device__ __forceinline__ double dw_div1000(double k) {
    double q = dw_permille_to_natural(k);
    if (__builtin_expect(!(fabs(k) < 4194304.0), 0)) q = k / 1000.0;
    return q;
}

// Lean float64 evaluation for the exact mode's fix-up kernels: only the two new covers, and the
// chain T_eff -> T -> T_x of the reference collapsed to T_x^4 = q2 (A_l - a_x) + q (A - A_l) +
// S L (1 - A) / sigma (the intermediate fourth roots cancel; the difference to the staged form is a
// few 1e-16 relative, the size of the reference's own FFT noise).  w[9]: (light | dark << 16)
// per-mille pairs of the 3x3 neighbourhood, row-major.
struct NewCoverF64 { double nl, nd; };
__device__ inline NewCoverF64 cell_f64_lean(const PhysF64& P, const unsigned int* w) {
    double l[9], d[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        l[i] = dw_permille_to_natural((double)(w[i] & 0xffffu));
        d[i] = dw_permille_to_natural((double)(w[i] >> 16));
    }
    double sb = 0.0, sl = 0.0, sd = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (i == 4) continue;
        sb += 0.125 * (P.p - l[i] - d[i]);
        sl += 0.125 * l[i];
        sd += 0.125 * d[i];
    }
    const double bare = P.p - l[4] - d[4];
    double Al = 0.0, A = 0.0;
    Al += P.ab * bare; A += P.ab * sb;
    Al += P.al * l[4]; A += P.al * sl;
    Al += P.ad * d[4]; A += P.ad * sd;
    const double T4 = P.q * (A - Al) + (P.S * P.L * (1.0 - A)) / P.sigma;
    const double Tl = dw_root4(P.q2 * (Al - P.al) + T4);
    const double Td = dw_root4(P.q2 * (Al - P.ad) + T4);
    const double bl = 1.0 - P.g * (P.To - Tl) * (P.To - Tl);
    const double bd = 1.0 - P.g * (P.To - Td) * (P.To - Td);
    const double kl = P.w0 * l[4] + P.w1 * (l[1] + l[3] + l[5] + l[7]) + P.w2 * (l[0] + l[2] + l[6] + l[8]);
    const double kd = P.w0 * d[4] + P.w1 * (d[1] + d[3] + d[5] + d[7]) + P.w2 * (d[0] + d[2] + d[6] + d[8]);
    const double kb = P.p - kl - kd;
    const double nl = l[4] + P.dt * (kl * (kb * bl - P.gamma));
    const double nd = d[4] + P.dt * (kd * (kb * bd - P.gamma));
    NewCoverF64 o;
    o.nl = nl < 0.0 ? 0.0 : (nl > 1.0 ? 1.0 : nl);
    o.nd = nd < 0.0 ? 0.0 : (nd > 1.0 ? 1.0 : nd);
    return o;
}

// ---------------------------------------------------------------------------------------------
// float32 fused form
// ---------------------------------------------------------------------------------------------
// With li, di the centre values and Sl8, Sd8 the sums over the 8 Moore neighbours (all per-mille),
// K = S*L/sigma and To4 = To^4, the reference's T_x^4 (x = light, dark) is affine in them:
//   e_x := T_x^4/To4 - 1 = c0x + a1*Sl8 + a2*Sd8 + a3*li + a4*di
// (coefficients derived in float64 on the host every step, see derive_f32() in dw_api.hip).
//
// Accuracy devices (they are what makes the exact mode's tie bound small):
//  * every coefficient is split as hi + lo with hi a multiple of 2^-hi_bits chosen so that, for a
//    quantised state (integer li, di, Sl8, Sd8), every product and partial sum of the "hi" chain is
//    exactly representable in float32: that chain carries NO rounding error; the "lo" chain is
//    ~2^-12 times smaller, so its roundings are negligible.  e_x then has one rounding (u*|e_x|).
//  * the fourth root is taken on u = 1 + e as y = sqrt(sqrt(u)), but the deviation from the
//    optimum temperature is formed as  (T_x - To)/To = y - 1 = e / ((y + 1)(y^2 + 1)),  which keeps
//    the RELATIVE accuracy of e instead of cancelling two numbers near 1.  The growth curve only needs
//    w = sqrt(g)*(T_x - To) = (y - 1)/kbeta (beta = 1 - w^2), so kbeta is folded into the denominator:
//    t = fma(s, kbeta, kbeta) = kbeta*(s + 1), D = fma(y, t, t) = kbeta*(y + 1)(s + 1), w = e * rcp(D) -
//    two packed instructions per species fewer than (y+1)*(s+1), e*rcp, cbeta*d, and two roundings fewer.
// Constant PAIRS.  The hot kernels evaluate two cells per lane with packed float32 instructions, whose operands
// are 64-bit: a wave-uniform constant splatted from ONE scalar register would occupy a whole SGPR pair (and an
// instruction may read only one scalar pair, so a second constant had to be copied into VGPRs).  Constants are
// therefore stored two to an 8-byte-aligned pair; the packed form selects the low or the high half for BOTH
// lanes with the instruction's op_sel bits (free), and constants that meet in one instruction share a pair.
// This halves the SGPR footprint of a coefficient set (the exact fused kernel spilled ~27 SGPRs into VGPR
// lanes and paid ~3 v_readlane per cell-evaluation for them).  Each pair is also addressable as two floats.
typedef float dw_f32x2 __attribute__((ext_vector_type(2)));
#define DW_PAIR(lo_name, hi_name, pair_name) \
    union { struct { float lo_name, hi_name; }; dw_f32x2 pair_name; }

struct PhysF32 {
    // e_x = c0x + a1*Sl8 + a2*Sd8 + a3*li + a4*di, coefficients split hi + lo (exact mode) ...
    DW_PAIR(a1h, a2h, a12h);
    DW_PAIR(a3h, a4h, a34h);
    DW_PAIR(a1l, a2l, a12l);
    DW_PAIR(a3l, a4l, a34l);
    DW_PAIR(c0lh, c0ll, c0l);    // constant for light: hi, lo (the lo part seeds the lo chain)
    DW_PAIR(c0dh, dc0l, c0d);    // constant for dark: hi, and (its lo part - light's lo part)
    // ... or rounded once (float32-only mode): a_i = fl(a_ih + a_il), c0x = fl(c0xh + c0xl)
    DW_PAIR(a1, a2, a12);
    DW_PAIR(a3, a4, a34);
    DW_PAIR(c0ls, c0ds, c0s);
    // dt * (daisy kernel weights): dK = dt * density comes straight out of the weighted sum
    DW_PAIR(dw0, dw1, dw01);
    DW_PAIR(dw2, kbeta, dw2kb);  // kbeta = 1 / sqrt(g * To^2):  beta = 1 - (((T-To)/To) / kbeta)^2
    DW_PAIR(p, ck, pck);         // bare fraction kb = p - (dKl + dKd) * ck,  ck = 0.001 / dt
    // exact-mode tie test (per-mille; om = 1 - beta = cbeta*((T-To)/To)^2 >= 0):
    //   |frac(gq)| > tie_lo - eA*|gq| - |dt*K|*(eK0 + eK1*om)   =>  re-evaluate in float64
    DW_PAIR(eK1s, eK0s, eKs);    // -sign(dt) * eK1, eK0: dK * (eK0s + eK1s*om) = -|dK| * (eK0 + eK1*om) (density >= 0)
    DW_PAIR(ngamma, tie_lo, gt); // -gamma; the tie threshold's constant part
    // the same bracket written in beta = 1 - om (the hot kernels have beta, not om):
    //   eK0s + eK1s*om = (eK0s + eK1s) + (-eK1s)*beta
    DW_PAIR(neK1s, eK01s, eKb);
    float eA;                    // used un-packed (|gq| source modifier); eK0 = |eK0s|, eK1 = |eK1s| (host, audit)
    int hi_bits;                 // the hi parts are multiples of 2^-hi_bits (host bookkeeping)
};
static_assert(sizeof(PhysF32) == 32 * sizeof(float), "PhysF32 layout");

// the members of PhysF32 that depend on the luminosity (the rest is shared by the two steps of a fused
// launch when both coefficient sets are split at the same scale, see derive_f32_pair() in dw_api.hip)
struct PhysLumF32 {
    dw_f32x2 a12h, a12l, c0l, c0d, a12, c0s;
};
__host__ __device__ inline PhysLumF32 lum_part(const PhysF32& P) {
    return PhysLumF32{P.a12h, P.a12l, P.c0l, P.c0d, P.a12, P.c0s};
}
__host__ __device__ inline PhysF32 with_lum(PhysF32 P, const PhysLumF32& l) {
    P.a12h = l.a12h; P.a12l = l.a12l; P.c0l = l.c0l; P.c0d = l.c0d; P.a12 = l.a12; P.c0s = l.c0s;
    return P;
}

// The float32 algebra is written once, generic in the lane type T: float (one cell) or dw_f32x2 (two
// horizontally adjacent cells).  On gfx950 the two-cell form compiles to packed float32 instructions
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane per issue slot), which is what
// the wave-strip kernels are bound by; transcendentals, rint and the clamp stay one instruction per cell.
// Floating-point contraction is switched off and every fused multiply-add is spelled out, so both
// instantiations perform the SAME correctly-rounded operations in the same order: all kernels produce
// bit-identical values whichever form they use.

template <typename T> struct Lanes;
template <> struct Lanes<float> {
    static constexpr int N = 1;
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
    static __device__ __forceinline__ float rcp(float v) { return __builtin_amdgcn_rcpf(v); }
    static __device__ __forceinline__ float rint(float v) { return __builtin_rintf(v); }
    static __device__ __forceinline__ float abs(float v) { return __builtin_fabsf(v); }
    static __device__ __forceinline__ float clip(float v) { return fminf(fmaxf(v, 0.0f), 1000.0f); }
    static __device__ __forceinline__ float get(float v, int) { return v; }
    // "a > b or unordered": a NaN anywhere in the float32 chain must flag the cell (float64 decides), exactly as the
    // lane-mask form below does - every kernel variant then agrees on NaN inputs and non-physical parameters
    static __device__ __forceinline__ void gt(float a, float b, bool* out) { out[0] = !(a <= b); }
    // the same comparison as the wave's 64-bit lane mask (v_cmp writes exactly that into a scalar pair): the hot
    // kernels combine, test and count tie flags on the scalar unit, with no per-lane booleans in between
    static __device__ __forceinline__ void gtm(float a, float b, unsigned long long* out) {
        out[0] = __builtin_amdgcn_fcmpf(a, b, 10 /* FCMP_UGT: a NaN is a near-tie too (float64 decides) */);
    }
    static __device__ __forceinline__ float load(const float* a, int i) { return a[i]; }
    static __device__ __forceinline__ float fma_abs(float a, float b, float c) { return __builtin_fmaf(a, __builtin_fabsf(b), c); }
    static __device__ __forceinline__ float lo(dw_f32x2 pair) { return pair.x; }   // the two constants of a pair
    static __device__ __forceinline__ float hi(dw_f32x2 pair) { return pair.y; }
};
template <> struct Lanes<dw_f32x2> {
    static constexpr int N = 2;
    using T = dw_f32x2;
    static __device__ __forceinline__ T fma(T a, T b, T c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ T sqrt(T v) { return T{__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y)}; }
    static __device__ __forceinline__ T rcp(T v) { return T{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
    static __device__ __forceinline__ T rint(T v) { return T{__builtin_rintf(v.x), __builtin_rintf(v.y)}; }
    static __device__ __forceinline__ T abs(T v) { return T{__builtin_fabsf(v.x), __builtin_fabsf(v.y)}; }
    static __device__ __forceinline__ T clip(T v) {
        return T{fminf(fmaxf(v.x, 0.0f), 1000.0f), fminf(fmaxf(v.y, 0.0f), 1000.0f)};
    }
    static __device__ __forceinline__ float get(T v, int i) { return i == 0 ? v.x : v.y; }
    static __device__ __forceinline__ void gt(T a, T b, bool* out) { out[0] = !(a.x <= b.x); out[1] = !(a.y <= b.y); }
    static __device__ __forceinline__ void gtm(T a, T b, unsigned long long* out) {
        out[0] = __builtin_amdgcn_fcmpf(a.x, b.x, 10 /* FCMP_UGT */);
        out[1] = __builtin_amdgcn_fcmpf(a.y, b.y, 10);
    }
    static __device__ __forceinline__ T load(const float* a, int i) { return T{a[i], a[i + 1]}; }
    // a*|b| + c per lane with the scalar VOP3 fma: its |.| source modifier is free, packed ops have none
    static __device__ __forceinline__ T fma_abs(float a, T b, float c) {
        return T{__builtin_fmaf(a, __builtin_fabsf(b.x), c), __builtin_fmaf(a, __builtin_fabsf(b.y), c)};
    }
    // one half of a constant pair for both cells: folds into the packed instruction's op_sel / op_sel_hi bits
    static __device__ __forceinline__ T lo(dw_f32x2 pair) { return __builtin_shufflevector(pair, pair, 0, 0); }
    static __device__ __forceinline__ T hi(dw_f32x2 pair) { return __builtin_shufflevector(pair, pair, 1, 1); }
};

template <typename T>
struct GrowthT {
    T gql, gqd;                  // per-mille growth dt*K*(kb*beta - gamma) for light, dark
    T fl, fd;                    // the factor kb*beta - gamma alone (gq = dK * f)
    T dKl, dKd;                  // dt * density (per-mille), used by the tie bound
    T oml, omd;                  // 1 - beta_l, 1 - beta_d (>= 0), used by the tie bound (audit; kernels that have no beta)
    T bl, bd;                    // beta_l, beta_d: the hot kernels' tie bound reads these instead of om
};
using GrowthF32 = GrowthT<float>;

// Coefficient chain of the float32-only mode: one rounded coefficient each (0, the default: 3 instructions
// fewer per cell) or the exact mode's hi/lo split (1).  Measured per-step deviation from the float64
// reference with both: profiles/r02_fast_tolerance.json (tools/fast_tolerance.py).
#ifndef DW_FAST_SPLIT
#define DW_FAST_SPLIT 0
#endif
constexpr bool kFastSplit = DW_FAST_SPLIT != 0;

// El/Cl: sums of the 4 edge / 4 corner neighbours of light; Ed/Cd of dark; li/di the centre.
// SPLIT = true: hi/lo coefficient chains (exact mode: its tie bound relies on the exact hi chain).
// SPLIT = false: one float32 coefficient each (float32-only mode: 6 instructions fewer per cell).
// SYM (with SPLIT): albedos symmetric about the bare ground's, a_dark - a_bare == -(a_light - a_bare) exactly (the
// reference's defaults 0.25 / 0.5 / 0.75).  Then a2 = -a1 and a4 = -a3 hold exactly, also for the hi and lo parts,
// and e_x = c0x + a1*(Sl8 - Sd8) + a3*(li - di): X = Sl8 - Sd8 and Y = li - di are exact integers (|X| <= 8000 <=
// the 8*kmax the hi grid was sized for), the hi chain a1h*X + a3h*Y is as exact as the four-term one - it is the
// same number - and the lo chain has two roundings instead of four: 6 packed instructions per cell pair instead of
// 8 (the host selects the variant: dw_api.hip, `sym_albedo`).
template <bool SPLIT, typename T, bool SYM = false>
__device__ __forceinline__ GrowthT<T> growth_t(const PhysF32& P, T li, T di, T El, T Cl, T Ed, T Cd) {
#pragma clang fp contract(off)
    using V = Lanes<T>;
    const T one = T(1.0f);
    const T Sl8 = El + Cl, Sd8 = Ed + Cd;
    T el, ed;
    if (SPLIT && SYM) {
        const T X = Sl8 - Sd8, Y = li - di;                  // exact
        T hi = V::lo(P.a12h) * X;                            // exact (see above)
        hi = V::fma(V::lo(P.a34h), Y, hi);
        T lo = V::fma(V::lo(P.a12l), X, V::hi(P.c0l));
        lo = V::fma(V::lo(P.a34l), Y, lo);
        el = (hi + V::lo(P.c0l)) + lo;
        ed = (hi + V::lo(P.c0d)) + (lo + V::hi(P.c0d));
    } else if (SPLIT) {
        T hi = V::lo(P.a12h) * Sl8;                         // exact for integer inputs (see above)
        hi = V::fma(V::hi(P.a12h), Sd8, hi);
        hi = V::fma(V::lo(P.a34h), li, hi);
        hi = V::fma(V::hi(P.a34h), di, hi);
        T lo = V::fma(V::lo(P.a12l), Sl8, V::hi(P.c0l));    // the lo chain starts from light's lo constant
        lo = V::fma(V::hi(P.a12l), Sd8, lo);
        lo = V::fma(V::lo(P.a34l), li, lo);
        lo = V::fma(V::hi(P.a34l), di, lo);
        el = (hi + V::lo(P.c0l)) + lo;
        ed = (hi + V::lo(P.c0d)) + (lo + V::hi(P.c0d));
    } else {
        T base = V::lo(P.a12) * Sl8;
        base = V::fma(V::hi(P.a12), Sd8, base);
        base = V::fma(V::lo(P.a34), li, base);
        base = V::fma(V::hi(P.a34), di, base);
        el = base + V::lo(P.c0s);
        ed = base + V::hi(P.c0s);
    }
    // light
    const T kbe = V::hi(P.dw2kb);
    const T sl = V::sqrt(one + el);
    const T yl = V::sqrt(sl);
    const T tl = V::fma(sl, kbe, kbe);                      // kbeta*(s + 1)
    const T wl = el * V::rcp(V::fma(yl, tl, tl));           // e / (kbeta*(y + 1)(s + 1)) = sqrt(cbeta)*(T_l - To)/To
    const T bl = V::fma(-wl, wl, one);                      // beta_l = 1 - w^2
    // dark
    const T sd = V::sqrt(one + ed);
    const T yd = V::sqrt(sd);
    const T td = V::fma(sd, kbe, kbe);
    const T wd = ed * V::rcp(V::fma(yd, td, td));
    const T bd = V::fma(-wd, wd, one);
    // dt * densities (per-mille) and the bare fraction (natural)
    GrowthT<T> o;
    o.dKl = V::fma(V::lo(P.dw2kb), Cl, V::fma(V::hi(P.dw01), El, V::lo(P.dw01) * li));
    o.dKd = V::fma(V::lo(P.dw2kb), Cd, V::fma(V::hi(P.dw01), Ed, V::lo(P.dw01) * di));
    const T kb = V::fma(-(o.dKl + o.dKd), V::hi(P.pck), V::lo(P.pck));
    o.oml = wl * wl;                                        // only the audit / the cold exact kernels read these
    o.omd = wd * wd;
    o.bl = bl;
    o.bd = bd;
    o.fl = V::fma(kb, bl, V::lo(P.gt));
    o.fd = V::fma(kb, bd, V::lo(P.gt));
    o.gql = o.dKl * o.fl;                                   // the exact finaliser and the audit; the float32-only
    o.gqd = o.dKd * o.fd;                                   // finaliser folds the product into k + gq (one rounding)
    return o;
}

template <bool SPLIT = true, bool SYM = false>
__device__ __forceinline__ GrowthF32 growth_f32(const PhysF32& P, float li, float di, float El, float Cl,
                                                float Ed, float Cd) {
    return growth_t<SPLIT, float, SYM>(P, li, di, El, Cl, Ed, Cd);
}

// FAST finaliser: k' = rint(clip(k + dK*f, 0, 1000)), the sum in one fused multiply-add (one rounding instead of
// two, one instruction fewer) — valid for any (also un-quantised) input.
template <typename T>
__device__ __forceinline__ T finish_fast_t(T k, T dK, T f) {
#pragma clang fp contract(off)
    return Lanes<T>::rint(Lanes<T>::clip(Lanes<T>::fma(dK, f, k)));
}
__device__ __forceinline__ float finish_fast(float k, float dK, float f) { return finish_fast_t<float>(k, dK, f); }

// EXACT finaliser for an integer k: rint(k + gq) = k + rint(gq) unless gq is within the float32
// error bound of a tie, in which case the cell's `tie` flag (one per lane of T) is raised and the caller
// re-evaluates in float64.
template <typename T>
__device__ __forceinline__ T finish_exact_t(const PhysF32& P, T k, T gq, T dK, T om, bool* tie) {
#pragma clang fp contract(off)
    using V = Lanes<T>;
    const T r = V::rint(gq);
    const T frac = V::abs(gq - r);                          // exact (Sterbenz)
    const T thr = V::fma(dK, V::fma(V::lo(P.eKs), om, V::hi(P.eKs)), V::fma_abs(-P.eA, gq, P.tie_lo));
    V::gt(frac, thr, tie);
    return V::clip(k + r);
}
__device__ __forceinline__ float finish_exact(const PhysF32& P, float k, float gq, float dK, float om,
                                              bool& tie) {
    return finish_exact_t<float>(P, k, gq, dK, om, &tie);
}
// The same with the bracket of the threshold written in beta (one packed multiply per cell pair and species fewer:
// om = w*w is not formed).  1 - beta differs from fl(w*w) by at most u*max(1, om), i.e. the threshold moves by
// |dK|*eK1*u*(1 + om) <= 1e-9 quanta: inside the rounding floor of tie_lo.
#ifndef DW_TIE_FROM_BETA
#define DW_TIE_FROM_BETA 1
#endif
template <typename T, typename F>
__device__ __forceinline__ T finish_exact_beta_t(const PhysF32& P, T k, T gq, T dK, T beta, F* tie) {
#pragma clang fp contract(off)
    using V = Lanes<T>;
    const T r = V::rint(gq);
    const T frac = V::abs(gq - r);                          // exact (Sterbenz)
    const T thr = V::fma(dK, V::fma(V::lo(P.eKb), beta, V::hi(P.eKb)), V::fma_abs(-P.eA, gq, P.tie_lo));
    if constexpr (sizeof(F) == 8) V::gtm(frac, thr, tie);   // F = unsigned long long: lane masks
    else V::gt(frac, thr, tie);
    return V::clip(k + r);
}

// Philox4x32-10 (Salmon et al., SC'11) — counter-based RNG for the synthetic initial states.
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in [0,1) with 24 random bits (exactly representable in float32)
__host__ __device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

}  // namespace dw
