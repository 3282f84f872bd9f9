/* Host-side bulk draw from NumPy's legacy generator (MT19937), bit-compatible with np.random.rand / random_sample.
 *
 * The reference's reset() (daisy/daisy_world_rl.py:285-302, initialize_grid) draws np.random.rand(B, 2, d, d) twice from the
 * GLOBAL legacy stream; same-seed parity with the reference needs exactly those numbers and the stream left exactly where
 * NumPy would leave it.  NumPy produces them one at a time (state regeneration and tempering in scalar code, ~4.5 ns per
 * double: 9-10 ms of a 21 ms evolution-strategy generation for 2048 worlds of 16x16).  This file regenerates the state and
 * tempers in loops the compiler vectorises; the drop-in class hands it the state from np.random.get_state() and puts the
 * advanced state back with np.random.set_state().
 *
 * Algorithm: Matsumoto & Nishimura's MT19937 as NumPy ships it (numpy/random/src/mt19937/mt19937.c: mt19937_gen,
 * mt19937_next, mt19937_next_double) - restated, not copied:
 *   regeneration  key[k] = key[(k + 397) mod 624] ^ (y >> 1) ^ (y odd ? 0x9908b0df : 0),  y = (key[k] & 0x80000000) | (key[k+1] & 0x7fffffff)
 *   output        y = key[pos++];  y ^= y >> 11;  y ^= (y << 7) & 0x9d2c5680;  y ^= (y << 15) & 0xefc60000;  y ^= y >> 18
 *   double        a = next >> 5, b = next >> 6;  (a * 67108864.0 + b) / 9007199254740992.0
 * Plain C, built with gcc into libdaisyworld_host.so (therldaisyworld_amd/build.py); no GPU code, no torch types.
 */
#include <stddef.h>
#include <stdint.h>

#include "../../include/daisyworld_host.h"

#ifndef DW_HOST_BUILD_ID
#define DW_HOST_BUILD_ID "unknown"
#endif
/* found in the file as the bytes DW_HOST_BUILD_ID=<hex> (therldaisyworld_amd/build.py decides staleness by content) */
const char dw_host_build_id_[] = "DW_HOST_BUILD_ID=" DW_HOST_BUILD_ID;

#define MT_N 624
#define MT_M 397

#if defined(__GNUC__) && defined(__x86_64__)
#define DW_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define DW_CLONES
#endif

/* one regeneration of the whole state, then all 624 outputs tempered into out[] */
DW_CLONES static void mt_block(uint32_t* key, uint32_t* out) {
    /* k < 227: reads key[k + 397] (not yet rewritten) and key[k + 1] (not yet rewritten): no loop-carried dependence */
    for (int k = 0; k < MT_N - MT_M; ++k) {
        const uint32_t y = (key[k] & 0x80000000u) | (key[k + 1] & 0x7fffffffu);
        key[k] = key[k + MT_M] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    /* 227 <= k < 623: reads key[k - 227], rewritten 227 iterations earlier - in blocks of at most 227 the block's reads of
     * rewritten words all lie before the block */
    for (int k0 = MT_N - MT_M; k0 < MT_N - 1; k0 += MT_N - MT_M) {
        const int k1 = k0 + (MT_N - MT_M) < MT_N - 1 ? k0 + (MT_N - MT_M) : MT_N - 1;
        uint32_t tmp[MT_N - MT_M];
        for (int k = k0; k < k1; ++k) {
            const uint32_t y = (key[k] & 0x80000000u) | (key[k + 1] & 0x7fffffffu);
            tmp[k - k0] = key[k - (MT_N - MT_M)] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        for (int k = k0; k < k1; ++k) key[k] = tmp[k - k0];
    }
    {
        const uint32_t y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    }
    for (int k = 0; k < MT_N; ++k) {
        uint32_t y = key[k];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[k] = y;
    }
}

static inline uint32_t temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

DW_CLONES static void pairs_to_doubles(const uint32_t* w, double* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const int32_t a = (int32_t)(w[2 * i] >> 5), b = (int32_t)(w[2 * i + 1] >> 6);   /* < 2^27: signed conversions vectorise */
        out[i] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
}

/* Fill out[0..n) with the next n doubles of the legacy stream whose state is (key[624], *pos); the state is advanced exactly
 * as n calls of NumPy's random_sample would advance it.  Returns 0, or -1 for a bad argument. */
int dw_mt19937_random_sample(uint32_t* key, int32_t* pos, double* out, size_t n) {
    if (!key || !pos || (!out && n) || *pos < 0 || *pos > MT_N) return -1;
    size_t need = 2 * n;                                    /* 32-bit outputs still to produce */
    size_t done = 0;                                        /* doubles written */
    uint32_t carry = 0;                                     /* the first word of a pair that straddles two blocks */
    int have_carry = 0;
    int p = *pos;
    uint32_t block[MT_N];
    /* what is left of the current state */
    while (need && p < MT_N) {
        const uint32_t y = temper(key[p++]);
        --need;
        if (!have_carry) { carry = y; have_carry = 1; }
        else { out[done++] = ((double)(carry >> 5) * 67108864.0 + (double)(y >> 6)) / 9007199254740992.0; have_carry = 0; }
    }
    /* whole regenerated blocks */
    while (need >= MT_N) {
        mt_block(key, block);
        int k = 0;
        if (have_carry) {
            out[done++] = ((double)(carry >> 5) * 67108864.0 + (double)(block[0] >> 6)) / 9007199254740992.0;
            have_carry = 0;
            k = 1;
        }
        const size_t npairs = (size_t)(MT_N - k) / 2;
        pairs_to_doubles(block + k, out + done, npairs);
        done += npairs;
        k += (int)(2 * npairs);
        if (k < MT_N) { carry = block[k]; have_carry = 1; }
        need -= MT_N;
        p = MT_N;
    }
    /* the beginning of one more block */
    if (need) {
        mt_block(key, block);
        p = 0;
        while (need) {
            const uint32_t y = block[p++];
            --need;
            if (!have_carry) { carry = y; have_carry = 1; }
            else { out[done++] = ((double)(carry >> 5) * 67108864.0 + (double)(y >> 6)) / 9007199254740992.0; have_carry = 0; }
        }
    }
    *pos = p;
    return 0;
}

/* np.random.randint(low, low + rng + 1, size=n) of the legacy generator for a range that fits 32 bits (the epsilon branch of
 * the reference's Greedy policy, daisy/agents/greedy.py:32: randint(9, size=(B, N, 1, 1)) once per step): NumPy's legacy
 * path draws 32-bit words and keeps `word & mask` (mask = the smallest 2^k - 1 >= rng) whenever it is <= rng - masked
 * rejection, one word per attempt (numpy/random/src/distributions/distributions.c, buffered_bounded_masked_uint32 as called by
 * random_bounded_uint64_fill with use_masked: restated).  Returns 0, -1 for a bad argument (rng == 0 or >= 2^32 - 1 are
 * other paths of NumPy's and are refused). */
int dw_mt19937_randint(uint32_t* key, int32_t* pos, int64_t low, uint64_t rng, int64_t* out, size_t n) {
    if (!key || !pos || (!out && n) || *pos < 0 || *pos > MT_N || rng == 0 || rng >= 0xFFFFFFFFull) return -1;
    uint32_t mask = (uint32_t)rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    const uint32_t r = (uint32_t)rng;
    uint32_t block[MT_N];
    int p = *pos;
    size_t done = 0;
    while (done < n && p < MT_N) {                          /* what is left of the current state */
        const uint32_t v = temper(key[p++]) & mask;
        if (v <= r) out[done++] = low + (int64_t)v;
    }
    while (done < n) {
        mt_block(key, block);
        p = 0;
        while (done < n && p < MT_N) {
            const uint32_t v = block[p++] & mask;
            out[done] = low + (int64_t)v;
            done += v <= r;
        }
    }
    *pos = p;
    return 0;
}

/* The draws of K calls of the reference's Greedy policy (daisy/agents/greedy.py:23-32) for a batch of `per_step` agents, in its
 * order: per call ONE np.random.rand() - the call takes its deterministic branch iff that coin > epsilon - and on the other
 * branch np.random.randint(9, size=(B, N, 1, 1)).  use_table[t] = 1 and table[t * per_step ..] = the actions of step t when step t
 * draws its actions, use_table[t] = 0 (row untouched) when it is greedy.  One state exchange per chunk of steps instead of
 * two NumPy calls per step. */
int dw_mt19937_greedy_draws(uint32_t* key, int32_t* pos, double epsilon, int32_t K, size_t per_step, uint8_t* use_table,
                            int8_t* table) {
    if (!key || !pos || *pos < 0 || *pos > MT_N || K < 0 || (K && !use_table) || (K && per_step && !table)) return -1;
    uint32_t block[MT_N];
    int p = *pos;
    for (int k = p; k < MT_N; ++k) block[k] = temper(key[k]);            /* what is left of the current state */
    for (int32_t t = 0; t < K; ++t) {
        if (p == MT_N) { mt_block(key, block); p = 0; }
        const uint32_t a = block[p++] >> 5;
        if (p == MT_N) { mt_block(key, block); p = 0; }
        const uint32_t b = block[p++] >> 6;
        const double coin = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        if (coin > epsilon) { use_table[t] = 0; continue; }
        use_table[t] = 1;
        int8_t* row = table + (size_t)t * per_step;
        size_t done = 0;
        while (done < per_step) {
            if (p == MT_N) { mt_block(key, block); p = 0; }
            while (done < per_step && p < MT_N) {                         /* (branch-free: the acceptance is a coin flip) */
                const uint32_t v = block[p++] & 15u;                      /* randint(9): range 8, mask 15 */
                row[done] = (int8_t)v;
                done += v <= 8u;
            }
        }
    }
    *pos = p;
    return 0;
}

int dw_host_abi_version(void) { return DW_HOST_ABI_VERSION; }
