/* CPU oracle (plain C, float64) for the RLDaisyWorld physics pass — TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product package links or calls this file.  It is used by tests/ (as the checker
 * for sizes where the NumPy oracle is too slow), by __graft_entry__.smoke() and by the
 * cpu_baseline leg of bench.py.
 *
 * It restates, stage by stage and in float64, RLDaisyWorld.forward() of the reference
 * (/root/reference/daisy/daisy_world_rl.py:434-461) with ft_convolve
 * (/root/reference/daisy/nn/functional.py:12-49) replaced by the identical direct 9-tap
 * toroidal stencil.  It deliberately keeps the reference's staging (three albedo convolutions,
 * two density convolutions, separate temperature / growth-rate / growth stages) rather than the
 * fused algebra the HIP kernels use, so that it checks that algebra.
 *
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_* against the golden vectors generated
 * from the reference (tests/golden/G1, G2) and against oracle/daisy_oracle.py.
 *
 * Build: make -C oracle   ->  oracle/libdaisy_oracle.so
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double p, g, S, sigma, gamma, q, q2, dt;
    double albedo_bare, albedo_light, albedo_dark, temp_optimal;
} oracle_params;

/* ref: daisy_world_rl.py:270-273 */
static void daisy_kernel(double k[3][3]) {
    double s = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double v = exp(-1.0);
            if (a == 1 && b == 1) v = 1.0;
            else if (a != 1 && b != 1) v = exp(-2.0);
            k[a][b] = v;
            s += v;
        }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) k[a][b] /= s;
}

/* out[i][j] = sum_{a,b} k[a][b] * x[i-(a-1)][j-(b-1)], toroidal. ref: nn/functional.py:12-49
 * par != 0: the rows of this ONE world are shared out over the OpenMP threads (few big worlds; every cell
 * is computed by the same expression either way). */
static void conv3x3(const double *x, double *out, int H, int W, const double k[3][3], int par) {
#pragma omp parallel for schedule(static) if (par)
    for (int i = 0; i < H; ++i) {
        for (int j = 0; j < W; ++j) {
            double acc = 0.0;
            for (int a = 0; a < 3; ++a) {
                int ii = (i - (a - 1) + H) % H;
                for (int b = 0; b < 3; ++b) {
                    if (k[a][b] == 0.0) continue;
                    int jj = (j - (b - 1) + W) % W;
                    acc += k[a][b] * x[(size_t)ii * W + jj];
                }
            }
            out[(size_t)i * W + j] = acc;
        }
    }
}

/* np.round(x, 3): rint(x*1000)/1000 (ties to even). ref: daisy_world_rl.py:452 */
static inline double round3(double x) { return rint(x * 1000.0) / 1000.0; }
static inline double clip01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

/* One forward() pass for one world.
 *   light, dark : H*W float64 in
 *   out7        : 7*H*W float64 out (channels bare, light, dark, T, T_light, T_dark, 0), quantised
 *                 exactly like the reference's new_grid BEFORE the agent-state overwrite (:452)
 *   scratch     : >= 8*H*W doubles
 *   caches      : optional (may be NULL) 7*H*W: temp, temp_light, temp_dark, beta_l, beta_d,
 *                 growth_l, growth_d (un-rounded side-effect caches, ref :345-347,373,415-419)
 */
static void forward_world(const oracle_params *P, double L, int H, int W, const double *light,
                          const double *dark, double *out7, double *scratch, double *caches, int par) {
    const size_t n = (size_t)H * W;
    double *bare = scratch, *cb = scratch + n, *cl = scratch + 2 * n, *cd = scratch + 3 * n;
    double *dl = scratch + 4 * n, *dd = scratch + 5 * n;
    double kd[3][3], ka[3][3];
    daisy_kernel(kd);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) ka[a][b] = (a == 1 && b == 1) ? 0.0 : 1.0 / 8.0; /* ref :280-281 */

    for (size_t i = 0; i < n; ++i) bare[i] = P->p - light[i] - dark[i];   /* ref :381 */
    conv3x3(bare, cb, H, W, ka, par);                                     /* ref :391-392 */
    conv3x3(light, cl, H, W, ka, par);
    conv3x3(dark, cd, H, W, ka, par);
    conv3x3(light, dl, H, W, kd, par);                                    /* ref :428-429 */
    conv3x3(dark, dd, H, W, kd, par);

#pragma omp parallel for schedule(static) if (par)
    for (size_t i = 0; i < n; ++i) {
        /* ref :387-392 */
        double Al = 0.0, A = 0.0;
        Al += P->albedo_bare * bare[i];  A += P->albedo_bare * cb[i];
        Al += P->albedo_light * light[i]; A += P->albedo_light * cl[i];
        Al += P->albedo_dark * dark[i];  A += P->albedo_dark * cd[i];
        /* ref :404-413 */
        double te = pow((P->S * L * (1 - A)) / P->sigma, 0.25);
        double T = pow(P->q * (A - Al) + pow(te, 4), 0.25);
        double Tl = pow(P->q2 * (Al - P->albedo_light) + pow(T, 4), 0.25);
        double Td = pow(P->q2 * (Al - P->albedo_dark) + pow(T, 4), 0.25);
        /* ref :343-344 */
        double bl = 1 - P->g * (P->temp_optimal - Tl) * (P->temp_optimal - Tl);
        double bd = 1 - P->g * (P->temp_optimal - Td) * (P->temp_optimal - Td);
        /* ref :355-367 */
        double a_l = dl[i], a_d = dd[i];
        double a_b = P->p - a_l - a_d;
        double gl = a_l * (a_b * bl - P->gamma);
        double gd = a_d * (a_b * bd - P->gamma);
        /* ref :449-452 */
        double nl = clip01(light[i] + P->dt * gl);
        double nd = clip01(dark[i] + P->dt * gd);
        double nb = P->p - nl - nd;
        out7[0 * n + i] = round3(nb);
        out7[1 * n + i] = round3(nl);
        out7[2 * n + i] = round3(nd);
        out7[3 * n + i] = round3(T);
        out7[4 * n + i] = round3(Tl);
        out7[5 * n + i] = round3(Td);
        out7[6 * n + i] = 0.0;
        if (caches) {
            caches[0 * n + i] = T;  caches[1 * n + i] = Tl; caches[2 * n + i] = Td;
            caches[3 * n + i] = bl; caches[4 * n + i] = bd;
            caches[5 * n + i] = gl; caches[6 * n + i] = gd;
        }
    }
}

/* ---- exported entry points (ctypes) ------------------------------------------------------ */

#ifdef _OPENMP
#include <omp.h>
#endif
/* fewer worlds than threads and worlds big enough to share out by rows */
static int few_big_worlds(int B, size_t n) {
#ifdef _OPENMP
    return B < omp_get_max_threads() && n >= (size_t)1 << 19;      /* >= 512k cells: a team per loop pays off */
#else
    (void)B; (void)n;
    return 0;
#endif
}

/* forward() for B worlds.  light/dark: [B][H][W] f64.  out7: [B][7][H][W].  caches optional
 * [B][7][H][W].  Returns 0, or -1 on allocation failure. */
int oracle_forward(const oracle_params *P, double L, int B, int H, int W, const double *light,
                   const double *dark, double *out7, double *caches) {
    const size_t n = (size_t)H * W;
    int rc = 0;
    if (few_big_worlds(B, n)) {                  /* one world at a time, its rows over the threads */
        double *scratch = (double *)malloc(sizeof(double) * 8 * n);
        if (!scratch) return -1;
        for (int b = 0; b < B; ++b)
            forward_world(P, L, H, W, light + b * n, dark + b * n, out7 + (size_t)b * 7 * n, scratch,
                          caches ? caches + (size_t)b * 7 * n : NULL, 1);
        free(scratch);
        return 0;
    }
#pragma omp parallel
    {
        double *scratch = (double *)malloc(sizeof(double) * 8 * n);
        if (!scratch) {
#pragma omp atomic write
            rc = -1;
        } else {
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b)
                forward_world(P, L, H, W, light + b * n, dark + b * n, out7 + (size_t)b * 7 * n,
                              scratch, caches ? caches + (size_t)b * 7 * n : NULL, 0);
            free(scratch);
        }
    }
    return rc;
}

/* `steps` no-agent steps in place on light/dark ([B][H][W] f64): the light/dark recurrence of
 * RLDaisyWorld.step() with N=0 (ref :475-497): forward() then L = clamp(L + dL) (ref :471-473).
 * On return *L_io holds the advanced luminosity.  Used for trajectories and for CPU timing. */
int oracle_step_n(const oracle_params *P, double *L_io, double dL, double min_L, double max_L,
                  int steps, int B, int H, int W, double *light, double *dark) {
    const size_t n = (size_t)H * W;
    int rc = 0;
    double L_end = *L_io;
    if (few_big_worlds(B, n)) {
        double *scratch = (double *)malloc(sizeof(double) * 8 * n);
        double *out7 = (double *)malloc(sizeof(double) * 7 * n);
        if (!scratch || !out7) { free(scratch); free(out7); return -1; }
        for (int b = 0; b < B; ++b) {
            double L = *L_io;
            for (int s = 0; s < steps; ++s) {
                forward_world(P, L, H, W, light + b * n, dark + b * n, out7, scratch, NULL, 1);
                memcpy(light + b * n, out7 + 1 * n, sizeof(double) * n);
                memcpy(dark + b * n, out7 + 2 * n, sizeof(double) * n);
                L += dL;
                L = L > max_L ? max_L : (L < min_L ? min_L : L);
            }
            L_end = L;
        }
        free(scratch);
        free(out7);
        *L_io = L_end;
        return 0;
    }
#pragma omp parallel
    {
        double *scratch = (double *)malloc(sizeof(double) * 8 * n);
        double *out7 = (double *)malloc(sizeof(double) * 7 * n);
        if (!scratch || !out7) {
#pragma omp atomic write
            rc = -1;
        } else {
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b) {
                double L = *L_io;
                for (int s = 0; s < steps; ++s) {
                    forward_world(P, L, H, W, light + b * n, dark + b * n, out7, scratch, NULL, 0);
                    memcpy(light + b * n, out7 + 1 * n, sizeof(double) * n);
                    memcpy(dark + b * n, out7 + 2 * n, sizeof(double) * n);
                    L += dL;
                    L = L > max_L ? max_L : (L < min_L ? min_L : L);
                }
                if (b == 0) {
#pragma omp atomic write
                    L_end = L;
                }
            }
        }
        free(scratch);
        free(out7);
    }
    *L_io = L_end;
    return rc;
}

int oracle_sizeof_params(void) { return (int)sizeof(oracle_params); }

#ifdef _OPENMP
int oracle_max_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int oracle_max_threads(void) { return 1; }
void oracle_set_threads(int n) { (void)n; }
#endif
