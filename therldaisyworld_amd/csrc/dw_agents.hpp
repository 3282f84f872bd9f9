// dw_agents.hpp — the per-agent kernels: update_agents, reward/done, observations, the Greedy and MLP
// policies, per-step episode flags and lifespan counters.
#pragma once
#include "dw_common.hpp"

namespace dw {

// ---------------------------------------------------------------------------------------------
// agents_update — ref update_agents (daisy_world_rl.py:181-244), collision_mode 0.
// One thread per world walks its agents IN ORDER (the first agent to land on a cell eats it all).
// Energy stores are float64 and updated with exactly the reference's operations, so alive/dead
// decisions and rewards are bit-identical.
// ---------------------------------------------------------------------------------------------
// T: the format of the CURRENT state (dw_common.hpp): plane_t, or float / double before the first step.
template <typename T>
__global__ void agents_update(T* __restrict__ L, T* __restrict__ D,
                              int* __restrict__ idx, double* __restrict__ st,
                              const int* __restrict__ action, int act_b, int act_n, int B, int N,
                              int H, int W, double agent_gamma, int do_clip,
                              double* __restrict__ reward = nullptr, unsigned char* __restrict__ done = nullptr,
                              unsigned char* __restrict__ agent_ok = nullptr) {
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
                    const double l = to_natural(L[o]), d = to_natural(D[o]);
                    s += l + d;
                    L[o] = (T)0.f; D[o] = (T)0.f;
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
    if (reward)      // the step's reward / done (ref step :486-492; the physics pass does not touch the stores)
        for (int n = 0; n < N; ++n) {
            const double s = st[(size_t)b * N + n];
            const double r = s * (s > 0.0 ? 1.0 : 0.0);
            reward[(size_t)b * N + n] = r;
            done[(size_t)b * N + n] = r < 0.1 ? 1 : 0;
        }
    if (agent_ok)    // the episode harness's per-step flag (= !done) of a step pair's first step
        for (int n = 0; n < N; ++n) {
            const double s = st[(size_t)b * N + n];
            agent_ok[(size_t)b * N + n] = (s * (s > 0.0 ? 1.0 : 0.0)) < 0.1 ? 0 : 1;
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
// observe — ref get_obs (:246-263): [B][N][7][3][3] float64 = the 3x3 wrap-around patch of the
// 7-channel grid around each agent, times the neighbourhood mask.  One thread per (agent, patch
// cell); the channel values are re-derived in float64 exactly as `materialise` does, so no
// 7-channel grid ever exists in HBM.  Channel 4 shows agent states at agent cells (ref :459).
// ---------------------------------------------------------------------------------------------
template <typename PrevT, bool POST>
__global__ void observe(const PrevT* __restrict__ pL, const PrevT* __restrict__ pD,
                        const plane_t* __restrict__ cL, const plane_t* __restrict__ cD,
                        const int* __restrict__ idx, const double* __restrict__ st, int B, int N,
                        int H, int W, PhysF64 P, int mask, double* __restrict__ obs,
                        double* __restrict__ reward = nullptr, unsigned char* __restrict__ done = nullptr) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * N * 9) return;
    const int k = gid % 9, an = gid / 9;       // patch cell, flat agent id
    const int b = an / N;
    if (reward && k == 0) {                    // the step's reward / done ride along (ref step :486-492)
        const double s = st[an];
        const double r = s * (s > 0.0 ? 1.0 : 0.0);
        reward[an] = r;
        done[an] = r < 0.1 ? 1 : 0;
    }
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
        v[0] = dw_div1000(dw_round3_k(P.p - o.nl - o.nd));
        v[1] = to_natural(cL[woff + (size_t)r * W + c]);
        v[2] = to_natural(cD[woff + (size_t)r * W + c]);
        v[3] = dw_div1000(dw_round3_k(o.T));
        v[4] = dw_div1000(dw_round3_k(o.Tl));
        v[5] = dw_div1000(dw_round3_k(o.Td));
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
template <typename T>
__global__ void policy_greedy(const T* __restrict__ cL, const T* __restrict__ cD,
                              const int* __restrict__ idx, int B, int N, int H, int W, int mask,
                              int argmin, const int* __restrict__ agent_mode, int* __restrict__ action,
                              int codes = 0) {
    const int an = blockIdx.x * blockDim.x + threadIdx.x;
    if (an >= B * N) return;
    const int b = an / N;
    if (agent_mode) {
        const int m = agent_mode[an - b * N];
        if (m == 2) return;
        argmin = m == 1;
    }
    if (codes) {                         // `action` holds table codes: >= 0 an action (kept), -1 argmax, -2 argmin
        const int a = action[an];
        if (a >= 0) return;
        argmin = a == -2;
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
            v = to_natural(cL[o]) + to_natural(cD[o]);
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
                                                 int* __restrict__ action, const int* __restrict__ member_hi = nullptr,
                                                 int split = -1) {
    __shared__ double s_x[4][64], s_h1[4][16], s_h2[4][32], s_o[4][16];
    const int na = a1 - a0;
    const int g = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int t = blockIdx.x * 4 + g;                       // agent handled by this 16-lane group
    const bool valid = t < B * na;
    const int tc = valid ? t : 0;
    const int b = tc / na, n = a0 + (tc - b * na);
    const double* x = obs + ((size_t)b * N + n) * 63;
    // split >= 0: agents [split, N) take their parameter set from member_hi (agent / adversary halves in one launch)
    const int* mm = (split >= 0 && n >= split) ? member_hi : member;
    if (mm) W += (size_t)mm[b] * 1808;
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

}  // namespace dw
