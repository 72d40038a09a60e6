// Solow-v0 and TradeAR1-v0 on gfx950 (reference fed_gym/envs/fed_env.py:161-334), with the worker
// loop body of fed_gym/agents/paac/emulator_runner.py:48-65 (step, auto-reset, process_state,
// history window) fused into the same launch.
//
// Mapping: one env per lane, state as float32 struct-of-arrays ([feature][env]) so every load
// and store of a wave is one contiguous 256-B segment.  Both kernels are HBM/latency bound
// (Solow ~53 B, TradeAR1-16 ~481 B per env-step).  Finished episodes are compacted with a wave
// ballot + one atomic per wave into done_list for the follow-up pass that refills Solow's
// shock tape.
#include "common.h"
#include "rng.h"
#include "flat_env_dev.h"

namespace grl {

__global__ __launch_bounds__(256) void solow_step_kernel(SolowParams S) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    bool done = false;
    if (env < S.E) done = solow_step_env(S, env, S.actions[env]).done;
    compact_done(done, env, S.done_list, S.done_count);
}

// explicit reset of the listed envs (gym reset()): light part; the tape is refilled by the next kernel
__global__ __launch_bounds__(256) void solow_reset_kernel(SolowParams S) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= *S.reset_count) return;
    int env = S.reset_list[li];
    float zl = solow_reset_env(S, env);
    S.nhist[env] = 0;   // an explicit reset happens in the learner; the worker's list starts empty (emulator_runner.py:23)
    solow_write_obs(S, env, S.kss, zl, 1);
}

// es = N(0, sigma)^T for every env in the list (fed_env.py:248); episode[env] was already advanced
__global__ __launch_bounds__(256) void solow_tape_kernel(SolowParams S) {
    const int n = *S.reset_count;
    const long long total = (long long)n * (S.T / 2);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        int li = (int)(idx % n);
        int pr = (int)(idx / n);
        int env = S.reset_list[li];
        solow_tape_pair(S, env, S.episode[env], pr);
    }
}

__global__ void solow_observe_kernel(SolowParams S) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= S.E) return;
    solow_write_obs(S, env, S.k[env], S.z[(size_t)(S.P - 1) * S.E + env], S.nhist[env] < 1 ? 1 : S.nhist[env]);
}

template <typename T>
static int dmalloc(grl_handle *h, T **p, size_t n) {
    GRL_HIP(h, hipMalloc((void **)p, n * sizeof(T)));
    h->allocs.push_back(*p);
    GRL_HIP(h, hipMemsetAsync(*p, 0, n * sizeof(T), h->stream));
    return GRL_OK;
}

int solow_alloc(grl_handle *h) {
    const grl_config &c = h->cfg;
    size_t E = h->E;
    // SolowEnv.__init__ (fed_env.py:179-189).  q == 0 behaves exactly like q == 1 with rho_e = [0.5]
    // (e starts empty -> MA term 0, then e = [e_t]); p == 0 cannot even reset in the reference
    // (z[-1] on an empty array) and is rejected by grl_create.
    h->so.P = c.solow_p;
    h->so.Q = c.solow_q > 0 ? c.solow_q : 1;
    double sum = 0;
    for (int i = 1; i <= c.solow_p; ++i) sum += pow(0.5, i);
    for (int i = 0; i < 8; ++i) { h->so.rho_z[i] = 0.f; h->so.rho_e[i] = 0.f; }
    for (int i = 1; i <= c.solow_p; ++i) h->so.rho_z[i - 1] = (float)(pow(0.5, i) / (sum / 0.95));
    for (int i = 1; i <= h->so.Q; ++i) h->so.rho_e[i - 1] = (float)pow(0.5, i);
    int rc;
    if ((rc = dmalloc(h, &h->so.k, E))) return rc;
    if ((rc = dmalloc(h, &h->so.z, E * h->so.P))) return rc;
    if ((rc = dmalloc(h, &h->so.z0, E * h->so.P))) return rc;
    if ((rc = dmalloc(h, &h->so.e, E * h->so.Q))) return rc;
    if ((rc = dmalloc(h, &h->so.tape, E * (size_t)c.solow_tape_len))) return rc;
    if ((rc = dmalloc(h, &h->so.tape_pos, E))) return rc;
    if ((rc = dmalloc(h, &h->so.nhist, E))) return rc;
    if ((rc = dmalloc(h, &h->so.obs_raw, E * 2))) return rc;
    if ((rc = dmalloc(h, &h->so.obs, E * 2))) return rc;
    if ((rc = dmalloc(h, &h->so.history, E * c.rnn_length * 2))) return rc;
    return GRL_OK;
}

static bool solow_needs_tape(const grl_handle *h) {
    return !(h->cfg.flags & GRL_F_RESET_FROM_SNAPSHOT);
}

int solow_launch_step(grl_handle *h, const float *actions_dev) {
    SolowParams S = solow_params(h);
    S.actions = actions_dev;
    GRL_HIP(h, hipMemsetAsync(h->done_count, 0, sizeof(int32_t), h->stream));
    hipLaunchKernelGGL(solow_step_kernel, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, S);
    GRL_HIP(h, hipGetLastError());
    if (solow_needs_tape(h) && h->cfg.max_episode_steps > 0) {
        S.reset_list = h->done_list; S.reset_count = h->done_count;
        hipLaunchKernelGGL(solow_tape_kernel, dim3(512), dim3(256), 0, h->stream, S);
        GRL_HIP(h, hipGetLastError());
    }
    return GRL_OK;
}

int solow_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count, bool) {
    SolowParams S = solow_params(h);
    S.reset_list = list_dev; S.reset_count = count_dev;
    hipLaunchKernelGGL(solow_reset_kernel, dim3((max_count + 255) / 256), dim3(256), 0, h->stream, S);
    GRL_HIP(h, hipGetLastError());
    if (solow_needs_tape(h)) {
        hipLaunchKernelGGL(solow_tape_kernel, dim3(512), dim3(256), 0, h->stream, S);
        GRL_HIP(h, hipGetLastError());
    }
    return GRL_OK;
}

int solow_launch_observe(grl_handle *h) {
    SolowParams S = solow_params(h);
    hipLaunchKernelGGL(solow_observe_kernel, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, S);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

// ------------------------------------------------------------------------------------------ TradeAR1
// The account (cash, assets, quantities, prices) is float64 on the device, as it is in the reference (numpy float64 throughout,
// fed_env.py:300-330): the reward log(assets') - log(assets) is a difference of two nearly equal numbers -- with float32 prices
// it carried ~1e-7 absolute = ~1e-4 relative error.  Observations handed to the net and the reward slot are float32 (runners.py:9).
__global__ __launch_bounds__(256) void trade_step_kernel(TradeParams R) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    bool done = false;
    if (env < R.E) {
        const int n = R.n, S = 1 + 2 * n;
        const size_t E = R.E;
        // TradeAR1Env._step (fed_env.py:300-321)
        double cash = R.cash[env];
        const double assets_old = R.assets[env];
        double cost = 0.0, value = 0.0;
        bool bad = false;
        for (int a = 0; a < n; ++a) {
            const float actf = R.actions[(size_t)env * n + a];
            bad |= !(actf >= -1.0f && actf <= 1.0f);                     // action_space.contains (fed_env.py:301)
            const double act = (double)actf;
            const double p = R.p[a * E + env];
            double q = R.q[a * E + env];
            const double q_add = act > 0.0 ? (act / (double)n) * cash / p : act * q;
            q += q_add;
            cost += q_add * p;
            value += q * p;
            R.q[a * E + env] = q;
        }
        if (bad) atomicAdd(R.err_flag, 1);
        cash = cash + (-cost);
        const double assets = cash + value;
        const bool own_done = assets < 1.0;                              // MIN_CASH (fed_env.py:272,313)
        R.reward[env] = (float)(log(assets + 1e-4) - log(assets_old + 1e-4));
        const int el = R.elapsed[env] + 1;
        done = own_done || (R.max_steps > 0 && el >= R.max_steps);
        R.done[env] = done ? 1 : 0;
        const uint32_t st = R.nstep[env];
        R.nstep[env] = st + 1;
        float *oraw = R.obs_raw + (size_t)env * S, *o = R.obs + (size_t)env * S;
        if (done) {   // auto-reset (emulator_runner.py:50-52) -> TradeAR1Env._reset (fed_env.py:323-330)
            cash = R.start;
            R.assets[env] = R.start;
            R.elapsed[env] = 0;
            R.nhist[env] = 1;            // histories[i] = [reset state]  (emulator_runner.py:52)
            R.episode[env] = R.episode[env] + 1;
            for (int a = 0; a < n; ++a) {
                R.q[a * E + env] = 0.0; R.p[a * E + env] = 1.0;
                oraw[1 + a] = 0.f; oraw[1 + n + a] = 1.f;
                o[1 + a] = trade_proc(1, 0.0); o[1 + n + a] = trade_proc(1, 1.0);
            }
        } else {
            R.assets[env] = assets;
            R.elapsed[env] = el;
            {
                const int nh = R.nhist[env] + 1;
                R.nhist[env] = nh > R.rnn + 1 ? R.rnn + 1 : nh;     // list trimmed to rnn+1 (emulator_runner.py:61)
            }
            const int pairs = (n + 1) / 2;
            for (int a = 0; a < n; a += 2) {
                double z[2];
                if (R.flags & GRL_F_INJECT_NOISE) {
                    z[0] = R.normals[a * E + env];
                    z[1] = a + 1 < n ? R.normals[(a + 1) * E + env] : 0.0;
                } else {
                    normal_pair(rng_block(R.seed, (uint32_t)env + R.env_off, 0u, RS_TRADE_PRICE, st * pairs + (a >> 1)), z[0], z[1]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int ak = a + k;
                    if (ak >= n) break;
                    // _price_transition: p**rho_p * exp(std_e * N(0,1))  (fed_env.py:296-298)
                    const double p = pow(R.p[ak * E + env], 0.9) * exp(R.std_e * z[k]);
                    R.p[ak * E + env] = p;
                    const double q = R.q[ak * E + env];
                    oraw[1 + ak] = (float)q; oraw[1 + n + ak] = (float)p;
                    o[1 + ak] = trade_proc(1, q); o[1 + n + ak] = trade_proc(1, p);
                }
            }
        }
        R.cash[env] = cash;
        oraw[0] = (float)cash;
        o[0] = trade_proc(0, cash);
    }
    compact_done(done, env, R.done_list, R.done_count);
}

__global__ void trade_reset_kernel(TradeParams R) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= *R.reset_count) return;
    const int env = R.reset_list[li];
    const int n = R.n, S = 1 + 2 * n;
    const size_t E = R.E;
    R.cash[env] = R.start; R.assets[env] = R.start; R.elapsed[env] = 0; R.episode[env] = R.episode[env] + 1;
    R.nhist[env] = 0;     // explicit reset: the worker's list starts empty (emulator_runner.py:23)
    float *oraw = R.obs_raw + (size_t)env * S, *o = R.obs + (size_t)env * S;
    oraw[0] = (float)R.start; o[0] = trade_proc(0, R.start);
    for (int a = 0; a < n; ++a) {
        R.q[a * E + env] = 0.0; R.p[a * E + env] = 1.0;
        oraw[1 + a] = 0.f; oraw[1 + n + a] = 1.f;
        o[1 + a] = trade_proc(1, 0.0); o[1 + n + a] = trade_proc(1, 1.0);
    }
}

__global__ void trade_observe_kernel(TradeParams R) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= R.E) return;
    const int n = R.n, S = 1 + 2 * n;
    const size_t E = R.E;
    float *oraw = R.obs_raw + (size_t)env * S, *o = R.obs + (size_t)env * S;
    const double cash = R.cash[env];
    oraw[0] = (float)cash; o[0] = trade_proc(0, cash);
    for (int a = 0; a < n; ++a) {
        const double q = R.q[a * E + env], p = R.p[a * E + env];
        oraw[1 + a] = (float)q; oraw[1 + n + a] = (float)p;
        o[1 + a] = trade_proc(1, q); o[1 + n + a] = trade_proc(1, p);
    }
}

int trade_alloc(grl_handle *h) {
    size_t E = h->E, n = h->cfg.n_assets;
    double sp = h->cfg.trade_std_p;
    h->tr.std_e = sqrt((sp * sp) * (1 - 0.9 * 0.9));   // fed_env.py:277-278
    h->tr.start = h->cfg.trade_starting_balance;
    int rc;
    if ((rc = dmalloc(h, &h->tr.cash, E))) return rc;
    if ((rc = dmalloc(h, &h->tr.assets, E))) return rc;
    if ((rc = dmalloc(h, &h->tr.q, E * n))) return rc;
    if ((rc = dmalloc(h, &h->tr.p, E * n))) return rc;
    if ((rc = dmalloc(h, &h->tr.normals, E * n))) return rc;
    if ((rc = dmalloc(h, &h->tr.nstep, E))) return rc;
    if ((rc = dmalloc(h, &h->tr.nhist, E))) return rc;
    if ((rc = dmalloc(h, &h->tr.obs_raw, E * (1 + 2 * n)))) return rc;
    if ((rc = dmalloc(h, &h->tr.obs, E * (1 + 2 * n)))) return rc;
    return GRL_OK;
}

int trade_launch_step(grl_handle *h, const float *actions_dev) {
    TradeParams R = trade_params(h);
    R.actions = actions_dev;
    GRL_HIP(h, hipMemsetAsync(h->done_count, 0, sizeof(int32_t), h->stream));
    hipLaunchKernelGGL(trade_step_kernel, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, R);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

int trade_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count) {
    TradeParams R = trade_params(h);
    R.reset_list = list_dev; R.reset_count = count_dev;
    hipLaunchKernelGGL(trade_reset_kernel, dim3((max_count + 255) / 256), dim3(256), 0, h->stream, R);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

int trade_launch_observe(grl_handle *h) {
    TradeParams R = trade_params(h);
    hipLaunchKernelGGL(trade_observe_kernel, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, R);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

}  // namespace grl
