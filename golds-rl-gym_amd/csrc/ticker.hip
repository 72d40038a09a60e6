// TickerEnv on gfx950 (reference fed_gym/envs/fed_env.py:89-158) over an OpenCloseSampler price table
// (fed_gym/envs/data/sampler.py:8-41), with the worker loop body of fed_gym/agents/paac/emulator_runner.py:48-65
// (step, auto-reset, process_state) fused into the same launch and TickerTraderStateProcessor.process_state
// (fed_gym/agents/state_processors.py:50-63) as the processed observation.
//
// Mapping: one env per lane.  The account (cash, equity, two positions) is float64 like the reference's numpy scalars
// and evaluated in the reference's operation order (-ffp-contract=off: +, *, / are bit-exact against numpy); the
// table (rows x 4 float64: price, inverse price, volume, volume) is shared by all envs and stays in cache, each env
// owns a WINDOW-row slice starting at its `start`.  ~190 B of state and outputs per env-step: HBM/latency bound.
#include "common.h"
#include "rng.h"

namespace grl {

constexpr int TICKER_WINDOW = 1024;       // fed_env.py:147  self.data.sample(1024)
constexpr double TICKER_SPREAD = 0.006;   // fed_env.py:105
constexpr double TICKER_MIN_CASH = 1.0;   // fed_env.py:96
constexpr double TICKER_START_BALANCE = 10.0;

struct TickerParams {
    double *cash, *assets, *q;            // q (E,2)
    int32_t *idx, *start, *start0, *nhist;
    const double *table;
    int rows;
    int32_t *elapsed, *episode;
    const float *actions;                 // (E,4): choice0, choice1 (0 hold, 1 buy, 2 sell), fraction0, fraction1
    float *reward;
    double *reward64;
    uint8_t *done;
    float *obs_raw, *obs;                 // (E,7)
    int32_t *done_list, *done_count, *err_flag;
    const int32_t *reset_list, *reset_count;
    int E, rnn, max_steps;
    uint32_t flags, env_off;
    uint64_t seed;
};

// [cash, q0, q1, p0, p1, v0, v1] and TickerTraderStateProcessor.process_state of it
__device__ __forceinline__ void ticker_write_obs(const TickerParams &K, int env, double cash, double q0, double q1, const double *row) {
    float *oraw = K.obs_raw + (size_t)env * 7, *o = K.obs + (size_t)env * 7;
    oraw[0] = (float)cash; oraw[1] = (float)q0; oraw[2] = (float)q1;
    oraw[3] = (float)row[0]; oraw[4] = (float)row[1]; oraw[5] = (float)row[2]; oraw[6] = (float)row[3];
    o[0] = (float)log(cash + 1e-4); o[1] = (float)log(q0 + 1.0); o[2] = (float)log(q1 + 1.0);
    o[3] = (float)log(row[0]); o[4] = (float)log(row[1]); o[5] = (float)row[2]; o[6] = (float)row[3];
}

// TickerEnv._reset (fed_env.py:144-156).  The window start is random.randint(0, T - 1024) in the reference
// (sampler.py:38): here a Philox draw keyed by (seed, global env id, episode), or the stored start0 under
// GRL_F_RESET_FROM_SNAPSHOT.
__device__ __forceinline__ int ticker_reset_env(const TickerParams &K, int env) {
    int st;
    if (K.flags & GRL_F_RESET_FROM_SNAPSHOT) {
        st = K.start0[env];
    } else {
        uint32_t ep = (K.flags & GRL_F_RESEED_EACH_RESET) ? 0u : (uint32_t)K.episode[env];
        double u0, u1;
        u01_pair(rng_block(K.seed, (uint32_t)env + K.env_off, ep, RS_TICKER_START, 0u), u0, u1);
        st = (int)(u0 * (double)(K.rows - TICKER_WINDOW + 1));
    }
    st = max(0, min(st, K.rows - TICKER_WINDOW));
    K.start[env] = st;
    K.idx[env] = 0;
    K.cash[env] = TICKER_START_BALANCE;
    K.assets[env] = TICKER_START_BALANCE;
    reinterpret_cast<double2 *>(K.q)[env] = make_double2(0.0, 0.0);
    K.elapsed[env] = 0;
    K.episode[env] = K.episode[env] + 1;
    return st;
}

__global__ __launch_bounds__(256) void ticker_step_kernel(TickerParams K) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    bool done = false;
    if (env < K.E) {
        const float4 act = reinterpret_cast<const float4 *>(K.actions)[env];
        const int d0 = (int)act.x, d1 = (int)act.y;
        double c0 = (double)act.z, c1 = (double)act.w;
        if (!(act.x == 0.f || act.x == 1.f || act.x == 2.f) || !(act.y == 0.f || act.y == 1.f || act.y == 2.f)) atomicAdd(K.err_flag, 1);
        int idx = max(0, min(K.idx[env], TICKER_WINDOW - 1));                    // host-settable fields: keep the row inside the table
        const int st = max(0, min(K.start[env], K.rows - TICKER_WINDOW));
        const double *row = K.table + (size_t)(st + idx) * 4;
        const double p0 = row[0], p1 = row[1];
        const double2 qq = reinterpret_cast<const double2 *>(K.q)[env];
        double cash = K.cash[env], q0 = qq.x, q1 = qq.y;
        const bool b0 = d0 == 1, b1 = d1 == 1, s0 = d0 == 2, s1 = d1 == 2;
        // TickerEnv._step (fed_env.py:110-142), same operation order
        const double bsum = (b0 && b1) ? c0 + c1 : (b0 ? c0 : (b1 ? c1 : 0.0));
        const double denom = fmax(bsum, 1.0);
        if (b0) c0 = c0 / denom;
        if (b1) c1 = c1 / denom;
        const double up = 1.0 + TICKER_SPREAD, dn = 1.0 - TICKER_SPREAD;
        const double a0 = b0 ? c0 * cash / (p0 * up) : (s0 ? -c0 * q0 : 0.0);
        const double a1 = b1 ? c1 * cash / (p1 * up) : (s1 ? -c1 * q1 : 0.0);
        q0 = q0 + a0;
        q1 = q1 + a1;
        const double cb0 = a0 * p0 * up, cb1 = a1 * p1 * up;
        const double cs0 = a0 * (p0 * dn), cs1 = a1 * (p1 * dn);
        const double sb = (b0 && b1) ? cb0 + cb1 : (b0 ? cb0 : (b1 ? cb1 : 0.0));
        const double ss = (s0 && s1) ? cs0 + cs1 : (s0 ? cs0 : (s1 ? cs1 : 0.0));
        cash = cash + (-sb - ss);
        const double old_assets = K.assets[env];
        const double assets = cash + (q0 * p0 + q1 * p1);
        const bool own_done = assets < TICKER_MIN_CASH;
        const double r = log(assets + 1e-4) - log(old_assets + 1e-4);
        K.reward64[env] = r;
        K.reward[env] = (float)r;
        idx += 1;
        int el = K.elapsed[env] + 1;
        done = own_done || (K.max_steps > 0 && el >= K.max_steps);
        if (!done && idx >= TICKER_WINDOW) {       // reference: IndexError on price_vol_data[1024] (no TimeLimit registered)
            atomicAdd(K.err_flag, 1 << 16);
            idx = TICKER_WINDOW - 1;
        }
        K.done[env] = done ? 1 : 0;
        if (done) {   // auto-reset (emulator_runner.py:50-52): the terminal reward stays, the observation is the reset one
            const int nst = ticker_reset_env(K, env);
            K.nhist[env] = 1;
            ticker_write_obs(K, env, TICKER_START_BALANCE, 0.0, 0.0, K.table + (size_t)nst * 4);
        } else {
            K.cash[env] = cash;
            K.assets[env] = assets;
            reinterpret_cast<double2 *>(K.q)[env] = make_double2(q0, q1);
            K.idx[env] = idx;
            K.elapsed[env] = el;
            int nh = K.nhist[env] + 1;
            K.nhist[env] = nh > K.rnn + 1 ? K.rnn + 1 : nh;
            ticker_write_obs(K, env, cash, q0, q1, K.table + (size_t)(st + idx) * 4);
        }
    }
    unsigned long long m = __ballot(done);
    if (m != 0) {
        int lane = threadIdx.x & 63;
        int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(K.done_count, __popcll(m));
        base = __shfl(base, leader);
        if (done) K.done_list[base + __popcll(m & ((1ull << lane) - 1ull))] = env;
    }
}

__global__ void ticker_reset_kernel(TickerParams K) {
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= *K.reset_count) return;
    const int env = K.reset_list[li];
    const int st = ticker_reset_env(K, env);
    K.nhist[env] = 0;     // explicit reset: the worker's list starts empty (emulator_runner.py:23)
    ticker_write_obs(K, env, TICKER_START_BALANCE, 0.0, 0.0, K.table + (size_t)st * 4);
}

__global__ void ticker_observe_kernel(TickerParams K) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= K.E) return;
    int st = max(0, min(K.start[env], K.rows - TICKER_WINDOW));
    int idx = max(0, min(K.idx[env], TICKER_WINDOW - 1));
    const double2 qq = reinterpret_cast<const double2 *>(K.q)[env];
    ticker_write_obs(K, env, K.cash[env], qq.x, qq.y, K.table + (size_t)(st + idx) * 4);
}

static TickerParams ticker_params(grl_handle *h) {
    TickerParams K{};
    K.cash = h->tk.cash; K.assets = h->tk.assets; K.q = h->tk.q; K.idx = h->tk.idx; K.start = h->tk.start; K.start0 = h->tk.start0;
    K.nhist = h->tk.nhist; K.table = h->tk.table; K.rows = h->tk.rows; K.elapsed = h->elapsed; K.episode = h->episode;
    K.reward = h->reward; K.reward64 = h->tk.reward64; K.done = h->done; K.obs_raw = h->tk.obs_raw; K.obs = h->tk.obs;
    K.done_list = h->done_list; K.done_count = h->done_count; K.err_flag = h->err_flag; K.E = h->E; K.rnn = h->cfg.rnn_length;
    K.max_steps = h->cfg.max_episode_steps; K.flags = h->cfg.flags; K.env_off = (uint32_t)h->cfg.env_id_offset; K.seed = h->cfg.seed;
    return K;
}

template <class T>
static int tk_malloc(grl_handle *h, T **p, size_t n) {
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) return hip_fail(h, e, "hipMalloc");
    h->allocs.push_back(*p);
    e = hipMemsetAsync(*p, 0, n * sizeof(T), h->stream);
    return e == hipSuccess ? GRL_OK : hip_fail(h, e, "hipMemsetAsync");
}

int ticker_alloc(grl_handle *h) {
    size_t E = h->E;
    int rc;
    if ((rc = tk_malloc(h, &h->tk.cash, E)) || (rc = tk_malloc(h, &h->tk.assets, E)) || (rc = tk_malloc(h, &h->tk.q, E * 2)) ||
        (rc = tk_malloc(h, &h->tk.reward64, E)) || (rc = tk_malloc(h, &h->tk.idx, E)) || (rc = tk_malloc(h, &h->tk.start, E)) ||
        (rc = tk_malloc(h, &h->tk.start0, E)) || (rc = tk_malloc(h, &h->tk.nhist, E)) || (rc = tk_malloc(h, &h->tk.obs_raw, E * 7)) ||
        (rc = tk_malloc(h, &h->tk.obs, E * 7)))
        return rc;
    h->tk.table = nullptr;
    h->tk.rows = 0;
    return GRL_OK;
}

int ticker_set_table(grl_handle *h, const double *rows_host, int nrows) {
    if (nrows < TICKER_WINDOW) return fail(h, GRL_E_INVALID, "grl_ticker_set_table: the table needs at least 1024 rows (sampler.py:38 samples windows of 1024)");
    for (size_t i = 0; i < (size_t)nrows * 2; ++i) {
        const double p = rows_host[(i >> 1) * 4 + (i & 1)];
        if (!(p > 0.0)) return fail(h, GRL_E_INVALID, "grl_ticker_set_table: prices must be positive");
    }
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    double *t = nullptr;
    GRL_HIP(h, hipMalloc((void **)&t, (size_t)nrows * 4 * sizeof(double)));
    h->allocs.push_back(t);          // an older table stays allocated until grl_destroy (a step using it may be in flight)
    GRL_HIP(h, hipMemcpy(t, rows_host, (size_t)nrows * 4 * sizeof(double), hipMemcpyHostToDevice));
    h->tk.table = t;
    h->tk.rows = nrows;
    return GRL_OK;
}

static int ticker_ready(grl_handle *h) {
    if (!h->tk.table) return fail(h, GRL_E_STATE, "Ticker handle has no price table yet: call grl_ticker_set_table first");
    return GRL_OK;
}

int ticker_launch_step(grl_handle *h, const float *actions_dev) {
    int rc = ticker_ready(h);
    if (rc) return rc;
    TickerParams K = ticker_params(h);
    K.actions = actions_dev;
    GRL_HIP(h, hipMemsetAsync(h->done_count, 0, sizeof(int32_t), h->stream));
    prof_begin(h);
    hipLaunchKernelGGL(ticker_step_kernel, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, K);
    prof_end(h);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

int ticker_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count) {
    int rc = ticker_ready(h);
    if (rc) return rc;
    TickerParams K = ticker_params(h);
    K.reset_list = list_dev; K.reset_count = count_dev;
    hipLaunchKernelGGL(ticker_reset_kernel, dim3((max_count + 255) / 256), dim3(256), 0, h->stream, K);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

int ticker_launch_observe(grl_handle *h) {
    int rc = ticker_ready(h);
    if (rc) return rc;
    TickerParams K = ticker_params(h);
    hipLaunchKernelGGL(ticker_observe_kernel, dim3((h->E + 255) / 256), dim3(256), 0, h->stream, K);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

}  // namespace grl
