// Device-side pieces of the Solow-v0 / TradeAR1-v0 step shared by the stand-alone step kernels (flat_envs.hip) and the persistent
// flat PAAC rollout (net_flat.hip: one workgroup keeps 64 envs for all T steps): parameter blocks, the per-env Solow step with the
// worker's auto-reset, the shock-tape draw, the TradeAR1 observation transform.  Reference: fed_gym/envs/fed_env.py:161-334,
// fed_gym/agents/paac/emulator_runner.py:48-65.
#pragma once
#include "common.h"
#include "rng.h"

namespace grl {

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void compact_done(bool done, int env, int32_t *done_list, int32_t *done_count) {
    unsigned long long m = __ballot(done);
    if (m == 0) return;
    int lane = threadIdx.x & 63;
    int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(done_count, __popcll(m));
    base = __shfl(base, leader);
    if (done) done_list[base + __popcll(m & ((1ull << lane) - 1ull))] = env;
}

// ------------------------------------------------------------------------------------------ Solow
struct SolowParams {
    float *k, *z, *e, *z0, *tape;
    int32_t *tape_pos, *nhist, *elapsed, *episode;
    const float *actions;
    float *reward;
    uint8_t *done;
    float *obs_raw, *obs, *history;
    int32_t *done_list, *done_count, *err_flag;
    const int32_t *reset_list, *reset_count;
    int E, P, Q, T, rnn, max_steps;
    float rho_z[8], rho_e[8];
    float delta, sigma, kss;
    uint32_t flags, env_off;
    uint64_t seed;
};

// SolowStateProcessor: obs / [100, 1] (state_processors.py:69-71) and the history window as the
// worker builds it (quirk Q11: min(n,rnn) copies of the CURRENT state, zero padded at the end)
__device__ __forceinline__ void solow_write_obs(const SolowParams &S, int env, float k, float zl, int nh) {
    reinterpret_cast<float2 *>(S.obs_raw)[env] = make_float2(k, zl);
    float2 o = make_float2(k / 100.0f, zl);
    reinterpret_cast<float2 *>(S.obs)[env] = o;
    int n = nh < S.rnn ? nh : S.rnn;
    for (int r = 0; r < S.rnn; ++r)
        reinterpret_cast<float2 *>(S.history)[(size_t)env * S.rnn + r] = r < n ? o : make_float2(0.f, 0.f);
}

// SolowEnv._reset without the tape (fed_env.py:242-250): k = k_ss(0.33), e = 0, z ~ N(0, sigma)^p
__device__ __forceinline__ float solow_reset_env(const SolowParams &S, int env) {
    S.k[env] = S.kss;
    for (int i = 0; i < S.Q; ++i) S.e[(size_t)i * S.E + env] = 0.f;
    const bool fixed = S.flags & (GRL_F_RESET_FROM_SNAPSHOT | GRL_F_RESEED_EACH_RESET);
    float zl = 0.f;
    if (S.flags & GRL_F_SOLOW_SS_RESET) {      // SolowSSEnv._reset (fed_env.py:259): self.z = np.array([0.])
        for (int i = 0; i < S.P; ++i) S.z[(size_t)i * S.E + env] = 0.f;
    } else if (fixed && (S.flags & GRL_F_RESET_FROM_SNAPSHOT)) {
        for (int i = 0; i < S.P; ++i) { zl = S.z0[(size_t)i * S.E + env]; S.z[(size_t)i * S.E + env] = zl; }
    } else {
        uint32_t ep = (S.flags & GRL_F_RESEED_EACH_RESET) ? 0u : (uint32_t)S.episode[env];
        for (int i = 0; i < S.P; i += 2) {
            double n0, n1;
            normal_pair(rng_block(S.seed, (uint32_t)env + S.env_off, ep, RS_SOLOW_Z0, i >> 1), n0, n1);
            zl = (float)((double)S.sigma * n0);
            S.z[(size_t)i * S.E + env] = zl;
            if (i + 1 < S.P) { zl = (float)((double)S.sigma * n1); S.z[(size_t)(i + 1) * S.E + env] = zl; }
        }
    }
    S.tape_pos[env] = S.T - 1;
    S.elapsed[env] = 0;
    S.nhist[env] = 1;            // histories[i] = [reset state]  (emulator_runner.py:52)
    S.episode[env] = S.episode[env] + 1;
    return zl;
}

// SolowEnv._step + the worker's auto-reset for ONE env (fed_env.py:201-236, emulator_runner.py:48-65): updates the env's state
// in global memory, writes reward / done / observation / history window, returns what a fused caller keeps in registers.
struct SolowStepOut { float reward, k, z; int nh; bool done; };
__device__ __forceinline__ SolowStepOut solow_step_env(const SolowParams &S, int env, float action) {
    SolowStepOut o;
    float s = fmaxf(1e-3f, action);
    float k = S.k[env];
    float zl = S.z[(size_t)(S.P - 1) * S.E + env];
    float y = expf(zl) * powf(k, 0.33f);
    float kn = (1.0f - S.delta) * k + s * y;
    int pos = S.tape_pos[env];
    float e_t = 0.f;
    if (pos >= 0) e_t = S.tape[(size_t)pos * S.E + env];   // es.pop(): from the end (quirk Q8)
    else atomicAdd(S.err_flag, 1);                          // reference: IndexError, pop from empty list
    S.tape_pos[env] = pos - 1;
    float ar = 0.f, ma = 0.f;
    for (int i = 0; i < S.P; ++i) ar += S.rho_z[i] * S.z[(size_t)i * S.E + env];
    for (int i = 0; i < S.Q; ++i) ma += S.rho_e[i] * S.e[(size_t)i * S.E + env];
    float zn = (ar + ma) + e_t;
    for (int i = 0; i + 1 < S.P; ++i) S.z[(size_t)i * S.E + env] = S.z[(size_t)(i + 1) * S.E + env];
    S.z[(size_t)(S.P - 1) * S.E + env] = zn;
    for (int i = 0; i + 1 < S.Q; ++i) S.e[(size_t)i * S.E + env] = S.e[(size_t)(i + 1) * S.E + env];
    S.e[(size_t)(S.Q - 1) * S.E + env] = e_t;
    S.k[env] = kn;
    o.reward = logf((1.0f - s) * y + 1e-4f);
    S.reward[env] = o.reward;
    int el = S.elapsed[env] + 1;
    o.done = S.max_steps > 0 && el >= S.max_steps;    // Solow itself never ends (fed_env.py:234)
    S.done[env] = o.done ? 1 : 0;
    int nh;
    if (o.done) {   // auto-reset: terminal reward stays, observation is the reset one (quirk Q6)
        zn = solow_reset_env(S, env);
        kn = S.kss;
        nh = 1;
    } else {
        S.elapsed[env] = el;
        nh = S.nhist[env] + 1;
        if (nh > S.rnn + 1) nh = S.rnn + 1;           // list is trimmed to rnn+1 (emulator_runner.py:61)
        S.nhist[env] = nh;
    }
    solow_write_obs(S, env, kn, zn, nh);
    o.k = kn; o.z = zn; o.nh = nh;
    return o;
}

// es = N(0, sigma)^T: pair `pr` of the tape of `env` (fed_env.py:248); episode[env] was already advanced by the reset
__device__ __forceinline__ void solow_tape_pair(const SolowParams &S, int env, int episode_after_reset, int pr) {
    uint32_t ep = (S.flags & GRL_F_RESEED_EACH_RESET) ? 0u : (uint32_t)(episode_after_reset - 1);
    double n0, n1;
    normal_pair(rng_block(S.seed, (uint32_t)env + S.env_off, ep, RS_SOLOW_TAPE, pr), n0, n1);
    S.tape[(size_t)(2 * pr) * S.E + env] = (float)((double)S.sigma * n0);
    S.tape[(size_t)(2 * pr + 1) * S.E + env] = (float)((double)S.sigma * n1);
}

static inline SolowParams solow_params(grl_handle *h) {
    SolowParams S{};
    S.k = h->so.k; S.z = h->so.z; S.e = h->so.e; S.z0 = h->so.z0; S.tape = h->so.tape;
    S.tape_pos = h->so.tape_pos; S.nhist = h->so.nhist; S.elapsed = h->elapsed; S.episode = h->episode;
    S.reward = h->reward; S.done = h->done; S.obs_raw = h->so.obs_raw; S.obs = h->so.obs; S.history = h->so.history;
    S.done_list = h->done_list; S.done_count = h->done_count; S.err_flag = h->err_flag;
    S.E = h->E; S.P = h->so.P; S.Q = h->so.Q; S.T = h->cfg.solow_tape_len; S.rnn = h->cfg.rnn_length;
    S.max_steps = h->cfg.max_episode_steps;
    for (int i = 0; i < 8; ++i) { S.rho_z[i] = h->so.rho_z[i]; S.rho_e[i] = h->so.rho_e[i]; }
    S.delta = (float)h->cfg.solow_delta; S.sigma = (float)h->cfg.solow_sigma;
    S.kss = (float)pow(0.33 / h->cfg.solow_delta, 1.0 / (1.0 - 0.33));   // _k_ss(0.33) (fed_env.py:198-199,243)
    S.flags = h->cfg.flags; S.env_off = (uint32_t)h->cfg.env_id_offset; S.seed = h->cfg.seed;
    return S;
}


// ------------------------------------------------------------------------------------------ TradeAR1
struct TradeParams {
    double *cash, *assets, *q, *p;
    const float *normals;
    uint32_t *nstep;
    int32_t *elapsed, *episode, *nhist;
    int rnn;
    const float *actions;
    float *reward;
    uint8_t *done;
    float *obs_raw, *obs;
    int32_t *done_list, *done_count, *err_flag;
    const int32_t *reset_list, *reset_count;
    int E, n, max_steps;
    double std_e;
    double start;                     // TradeAR1Env.starting_balance: cash = assets = start after a reset (fed_env.py:323-326)
    uint32_t flags, env_off;
    uint64_t seed;
};

// TradeWorker.process_state as it behaves (a3c/worker.py:420-431, quirk Q10):
// [log(cash+1e-4), log(q+1)..., log(p+1)...]; evaluated in float64 like the reference, rounded for the float32 net input
__device__ __forceinline__ float trade_proc(int idx, double v) { return (float)(idx == 0 ? log(v + 1e-4) : log(v + 1.0)); }

static inline TradeParams trade_params(grl_handle *h) {
    TradeParams R{};
    R.cash = h->tr.cash; R.assets = h->tr.assets; R.q = h->tr.q; R.p = h->tr.p; R.normals = h->tr.normals;
    R.nstep = h->tr.nstep; R.nhist = h->tr.nhist; R.rnn = h->cfg.rnn_length; R.elapsed = h->elapsed; R.episode = h->episode; R.reward = h->reward; R.done = h->done;
    R.obs_raw = h->tr.obs_raw; R.obs = h->tr.obs; R.done_list = h->done_list; R.done_count = h->done_count;
    R.err_flag = h->err_flag; R.E = h->E; R.n = h->cfg.n_assets; R.max_steps = h->cfg.max_episode_steps;
    R.std_e = h->tr.std_e; R.start = h->tr.start; R.flags = h->cfg.flags; R.env_off = (uint32_t)h->cfg.env_id_offset; R.seed = h->cfg.seed;
    return R;
}


}  // namespace grl
