/* daisyworld_host.h - host-side helper of the drop-in RLDaisyWorld (no GPU code): NumPy's legacy random stream in bulk.
 *
 * libdaisyworld_host.so (gcc, therldaisyworld_amd/csrc/dw_hostrng.c) is OPTIONAL: the drop-in class uses it when it is there
 * and NumPy's own np.random.rand otherwise - the numbers and the generator's final state are identical either way
 * (tests/test_abi_and_host.py), it is only 3-4x faster.  It replaces nothing on the device path; libdaisyworld_hip.so
 * (daisyworld_hip.h) is the product library and has no CPU fallback.
 */
#ifndef DAISYWORLD_HOST_H
#define DAISYWORLD_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DW_HOST_ABI_VERSION 1

/* The draws of the reference's initialize_grid (daisy/daisy_world_rl.py:287-302: `np.random.rand(B, 2, d, d)` twice, dark
 * first) - the next n doubles of NumPy's GLOBAL legacy generator (MT19937), bit for bit: `key` (624 words) and `*pos` are the
 * state as np.random.get_state() returns it (elements 1 and 2) and come back advanced exactly as n calls of random_sample
 * advance it (hand them to np.random.set_state with the untouched has_gauss / cached_gaussian).  Returns 0, -1 for a bad
 * argument (null pointer, pos outside 0..624). */
int dw_mt19937_random_sample(uint32_t* key, int32_t* pos, double* out, size_t n);

/* The epsilon branch of the reference's Greedy policy (daisy/agents/greedy.py:32: `np.random.randint(9, size=(B, N, 1, 1))`,
 * once per step for the whole batch): n integers of np.random.randint(low, low + rng + 1) from the same legacy stream, bit for
 * bit, for 0 < rng < 2^32 - 1 (NumPy's masked-rejection path on 32-bit words).  Same state convention as above. */
int dw_mt19937_randint(uint32_t* key, int32_t* pos, int64_t low, uint64_t rng, int64_t* out, size_t n);

/* The same policy's draws for a CHUNK of K steps in the reference's order (greedy.py:23-32: per call one np.random.rand() coin,
 * the deterministic branch iff coin > epsilon, else randint(9, size=(B, N, 1, 1))): use_table[t] in {0, 1}, table[t][per_step]
 * (int8, rows of greedy steps untouched) - what dw_run_episode takes (daisyworld_hip.h), with one state exchange per chunk. */
int dw_mt19937_greedy_draws(uint32_t* key, int32_t* pos, double epsilon, int32_t K, size_t per_step, uint8_t* use_table,
                            int8_t* table);

int dw_host_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
