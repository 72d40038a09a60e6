// C ABI of libgoldsrl.so (declared in include/goldsrl.h): handle lifecycle, state injection /
// extraction, step / wait / outputs -- the replacement for the reference's Runners +
// EmulatorRunner process pool (fed_gym/agents/paac/runners.py:11-66, emulator_runner.py:38-151).
#include <string.h>

#include <algorithm>
#include <mutex>
#include <set>
#include <cmath>

#include "common.h"
#include "rollout_dev.h"

static thread_local std::string g_create_error;
static std::mutex g_live_mutex;
static std::set<const grl_handle *> g_live_handles;

namespace grl {
bool grl_handle_alive(const grl_handle *h) {
    std::lock_guard<std::mutex> lk(g_live_mutex);
    return g_live_handles.count(h) != 0;
}
void grl_sync_for_destroy(grl_handle *h) {
    if (grl_handle_alive(h)) {
        hipSetDevice(h->cfg.device_id);
        if (h->stream) hipStreamSynchronize(h->stream);
    } else {
        hipDeviceSynchronize();
    }
}
}  // namespace grl

namespace grl {

int fail(grl_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

int hip_fail(grl_handle *h, hipError_t e, const char *what) {
    return fail(h, GRL_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

void prof_begin(grl_handle *h) {
    if (!h->prof_on || h->prof_used + 2 > h->prof_ev.size()) return;
    (void)hipEventRecord(h->prof_ev[h->prof_used], h->stream);
}
void prof_end(grl_handle *h) {
    if (!h->prof_on || h->prof_used + 2 > h->prof_ev.size()) return;
    (void)hipEventRecord(h->prof_ev[h->prof_used + 1], h->stream);
    h->prof_used += 2;
}

struct FieldInfo {
    void *dev;
    size_t elem;      // bytes per element
    size_t inner;     // host layout (E, inner)
    bool feature_major;   // device layout (inner, E) instead of (E, inner)
};

static bool field_info(grl_handle *h, int32_t f, FieldInfo &fi) {
    const int kind = h->cfg.env_kind;
    fi.feature_major = false;
    switch (f) {
        case GRL_FLD_ELAPSED: fi = {h->elapsed, 4, 1, false}; return true;
        case GRL_FLD_EPISODE: fi = {h->episode, 4, 1, false}; return true;
        default: break;
    }
    if (kind == GRL_ENV_SWARM) {
        switch (f) {
            case GRL_FLD_SWARM_X: fi = {h->sw.x, 8, 160, false}; return true;
            case GRL_FLD_SWARM_XA: fi = {h->sw.xa, 8, 20, false}; return true;
            case GRL_FLD_SWARM_PNOISE: fi = {h->sw.pnoise, 8, 160, false}; return true;
            case GRL_FLD_SWARM_ANOISE: fi = {h->sw.anoise, 8, 20, false}; return true;
            case GRL_FLD_RESET_X: fi = {h->sw.rx, 8, 160, false}; return h->sw.rx != nullptr;
            case GRL_FLD_RESET_XA: fi = {h->sw.rxa, 8, 20, false}; return h->sw.rxa != nullptr;
            case GRL_FLD_RESET_PNOISE: fi = {h->sw.rpnoise, 8, 160, false}; return h->sw.rpnoise != nullptr;
            case GRL_FLD_RESET_ANOISE: fi = {h->sw.ranoise, 8, 20, false}; return h->sw.ranoise != nullptr;
            default: return false;
        }
    }
    if (kind == GRL_ENV_SOLOW) {
        switch (f) {
            case GRL_FLD_SOLOW_K: fi = {h->so.k, 4, 1, false}; return true;
            case GRL_FLD_SOLOW_Z: fi = {h->so.z, 4, (size_t)h->so.P, true}; return true;
            case GRL_FLD_SOLOW_Z0: fi = {h->so.z0, 4, (size_t)h->so.P, true}; return true;
            case GRL_FLD_SOLOW_E: fi = {h->so.e, 4, (size_t)h->so.Q, true}; return true;
            case GRL_FLD_SOLOW_TAPE: fi = {h->so.tape, 4, (size_t)h->cfg.solow_tape_len, true}; return true;
            case GRL_FLD_SOLOW_TAPE_POS: fi = {h->so.tape_pos, 4, 1, false}; return true;
            case GRL_FLD_NHIST: fi = {h->so.nhist, 4, 1, false}; return true;
            default: return false;
        }
    }
    if (kind == GRL_ENV_TRADE) {
        size_t n = h->cfg.n_assets;
        switch (f) {
            case GRL_FLD_TRADE_CASH: fi = {h->tr.cash, 8, 1, false}; return true;
            case GRL_FLD_TRADE_ASSETS: fi = {h->tr.assets, 8, 1, false}; return true;
            case GRL_FLD_TRADE_QUANTITY: fi = {h->tr.q, 8, n, true}; return true;
            case GRL_FLD_TRADE_PRICES: fi = {h->tr.p, 8, n, true}; return true;
            case GRL_FLD_TRADE_NORMALS: fi = {h->tr.normals, 4, n, true}; return true;
            case GRL_FLD_NHIST: fi = {h->tr.nhist, 4, 1, false}; return true;
            default: return false;
        }
    }
    if (kind == GRL_ENV_TICKER) {
        switch (f) {
            case GRL_FLD_TICKER_CASH: fi = {h->tk.cash, 8, 1, false}; return true;
            case GRL_FLD_TICKER_ASSETS: fi = {h->tk.assets, 8, 1, false}; return true;
            case GRL_FLD_TICKER_QUANTITY: fi = {h->tk.q, 8, 2, false}; return true;
            case GRL_FLD_TICKER_IDX: fi = {h->tk.idx, 4, 1, false}; return true;
            case GRL_FLD_TICKER_START: fi = {h->tk.start, 4, 1, false}; return true;
            case GRL_FLD_TICKER_START0: fi = {h->tk.start0, 4, 1, false}; return true;
            case GRL_FLD_NHIST: fi = {h->tk.nhist, 4, 1, false}; return true;
            default: return false;
        }
    }
    return false;
}

static int action_cols(const grl_handle *h) {
    switch (h->cfg.env_kind) {
        case GRL_ENV_SWARM: return N_AGENTS * 2;
        case GRL_ENV_SOLOW: return 1;
        case GRL_ENV_TICKER: return 4;
        default: return h->cfg.n_assets;
    }
}

// R6 (paac.py:142-157, 331-349): one lane per env; finished episodes are compacted with a wave ballot + one atomicAdd per wave
__global__ void episodes_account_kernel(const float *__restrict__ reward, const uint8_t *__restrict__ done, int env_base, int E,
                                        double *__restrict__ total, int32_t *__restrict__ len, int64_t *__restrict__ steps,
                                        grl_episode_record *__restrict__ rec, int32_t *__restrict__ count, int cap) {
    const int e = env_base + blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = e < E;
    episodes_account_env(active, e, active ? reward[e] : 0.f, active && done[e] != 0, total, len, steps, rec, count, cap);
}

int episodes_launch_account(grl_handle *h, int env_base, int count) {
    if (!h->ep_total) return GRL_OK;
    if (count < 0) count = h->E - env_base;
    hipLaunchKernelGGL(episodes_account_kernel, dim3((count + 255) / 256), dim3(256), 0, h->stream, h->reward, h->done, env_base,
                       env_base + count, h->ep_total, h->ep_len, h->ep_steps, h->ep_rec, h->ep_count, h->ep_capacity);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

static int reset_list(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count) {
    switch (h->cfg.env_kind) {
        case GRL_ENV_SWARM: return swarm_launch_reset(h, list_dev, count_dev, max_count);
        case GRL_ENV_SOLOW: return solow_launch_reset(h, list_dev, count_dev, max_count, true);
        case GRL_ENV_TICKER: return ticker_launch_reset(h, list_dev, count_dev, max_count);
        default: return trade_launch_reset(h, list_dev, count_dev, max_count);
    }
}

}  // namespace grl

using namespace grl;

extern "C" {

int grl_abi_version(void) { return GRL_ABI_VERSION; }

int grl_config_default(int32_t env_kind, grl_config *cfg) {
    if (!cfg || env_kind < 0 || env_kind > GRL_ENV_TICKER) return GRL_E_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(grl_config);
    cfg->env_kind = env_kind;
    cfg->num_envs = 32;                                   // -ec default (train_paac_conv.py:109)
    cfg->max_episode_steps = env_kind == GRL_ENV_SWARM ? 128 : 1024;   // fed_gym/__init__.py:3-33
    if (env_kind == GRL_ENV_TICKER) cfg->max_episode_steps = 1023;     // not registered in the reference: last row of the 1024-row window
    cfg->grid_size = 84;                                  // --height (train_paac_conv.py:114)
    cfg->n_assets = 2;                                    // fed_env.py:269
    cfg->solow_p = 1; cfg->solow_q = 1;                   // fed_env.py:166
    cfg->solow_tape_len = 2048;                           // fed_env.py:176
    cfg->rnn_length = 5;                                  // --rnn-length
    cfg->seed = 1692;
    cfg->solow_sigma = 0.1; cfg->solow_delta = 0.02;      // fed_env.py:166
    cfg->trade_std_p = 0.05;                              // fed_env.py:269
    cfg->trade_starting_balance = 10.0;                   // fed_env.py:269
    return GRL_OK;
}

int grl_create(const grl_config *cfg, grl_handle **out) {
    if (!cfg || !out) return fail(nullptr, GRL_E_INVALID, "grl_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(grl_config)) return fail(nullptr, GRL_E_INVALID, "grl_create: grl_config size mismatch (ABI drift)");
    if (cfg->env_kind < 0 || cfg->env_kind > GRL_ENV_TICKER) return fail(nullptr, GRL_E_INVALID, "grl_create: unknown env_kind");
    if (cfg->env_kind == GRL_ENV_TICKER && (cfg->n_assets != 2 || cfg->rnn_length < 1 || cfg->rnn_length > 32))
        return fail(nullptr, GRL_E_INVALID, "grl_create: the Ticker table has two price columns (n_assets must be 2) and rnn_length must be in 1..32");
    if (cfg->env_kind == GRL_ENV_TICKER && (cfg->max_episode_steps < 0 || cfg->max_episode_steps > 1023))
        return fail(nullptr, GRL_E_INVALID, "grl_create: Ticker episodes live inside a 1024-row window: max_episode_steps must be in 0..1023 (0: no cap, stepping past the window is an error as in the reference)");
    if (cfg->num_envs <= 0) return fail(nullptr, GRL_E_INVALID, "grl_create: num_envs must be positive");
    if (cfg->env_kind == GRL_ENV_SWARM && (cfg->grid_size < 2 || cfg->grid_size > 254)) return fail(nullptr, GRL_E_INVALID, "grl_create: grid_size must be in 2..254");
    if (cfg->env_kind == GRL_ENV_TRADE && (cfg->n_assets < 1 || cfg->n_assets > 64)) return fail(nullptr, GRL_E_INVALID, "grl_create: n_assets must be in 1..64");
    if (cfg->env_kind == GRL_ENV_TRADE && !(cfg->trade_starting_balance > 0.0 && cfg->trade_starting_balance < 1e300))
        return fail(nullptr, GRL_E_INVALID, "grl_create: trade_starting_balance must be a positive finite number (the observation takes log(cash + 1e-4))");
    if (cfg->env_kind == GRL_ENV_SOLOW) {
        if (cfg->solow_p < 1 || cfg->solow_p > 8) return fail(nullptr, GRL_E_INVALID, "grl_create: solow_p must be in 1..8 (p=0 cannot reset in the reference either: fed_env.py:250 indexes an empty z)");
        if (cfg->solow_q < 0 || cfg->solow_q > 8) return fail(nullptr, GRL_E_INVALID, "grl_create: solow_q must be in 0..8");
        if (cfg->solow_tape_len < 2 || (cfg->solow_tape_len & 1)) return fail(nullptr, GRL_E_INVALID, "grl_create: solow_tape_len must be even and >= 2");
        if (cfg->rnn_length < 1 || cfg->rnn_length > 16) return fail(nullptr, GRL_E_INVALID, "grl_create: rnn_length must be in 1..16");
    }
    if (cfg->env_kind == GRL_ENV_TRADE && (cfg->rnn_length < 1 || cfg->rnn_length > 32))
        return fail(nullptr, GRL_E_INVALID, "grl_create: rnn_length must be in 1..32");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, GRL_E_NO_DEVICE, std::string("grl_create: no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") + "); this library has no CPU path");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, GRL_E_NO_DEVICE, "grl_create: device_id out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, cfg->device_id)) != hipSuccess) return fail(nullptr, GRL_E_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GRL_E_NO_DEVICE, std::string("grl_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
    if ((e = hipSetDevice(cfg->device_id)) != hipSuccess) return fail(nullptr, GRL_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));

    grl_handle *h = new grl_handle();
    {
        std::lock_guard<std::mutex> lk(g_live_mutex);
        g_live_handles.insert(h);
    }
    h->cfg = *cfg;
    h->E = cfg->num_envs;
    h->step_in_flight = false;
    h->prof_on = false; h->prof_used = 0;
    h->ep_total = nullptr; h->ep_len = nullptr; h->ep_steps = nullptr; h->ep_rec = nullptr; h->ep_count = nullptr; h->ep_capacity = 0;
    h->stream = nullptr; h->ev0 = nullptr; h->ev1 = nullptr;
    h->sw = {}; h->so = {}; h->tr = {}; h->tk = {};
    int rc = GRL_OK;
    auto bail = [&](int code) {
        g_create_error = h->err;
        grl_destroy(h);
        return code;
    };
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail(hip_fail(h, e, "hipStreamCreate"));
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess) return bail(hip_fail(h, e, "hipEventCreate"));
    size_t E = h->E;
    auto dm = [&](void **p, size_t bytes) -> int {
        hipError_t ee = hipMalloc(p, bytes);
        if (ee != hipSuccess) return hip_fail(h, ee, "hipMalloc");
        h->allocs.push_back(*p);
        ee = hipMemsetAsync(*p, 0, bytes, h->stream);
        return ee == hipSuccess ? GRL_OK : hip_fail(h, ee, "hipMemsetAsync");
    };
    h->actions_elems = E * action_cols(h);
    if ((rc = dm((void **)&h->elapsed, E * 4)) || (rc = dm((void **)&h->episode, E * 4)) || (rc = dm((void **)&h->reward, E * 4)) ||
        (rc = dm((void **)&h->done, E)) || (rc = dm((void **)&h->done_list, E * 4)) || (rc = dm((void **)&h->done_count, 16 * 4)) ||
        (rc = dm((void **)&h->err_flag, 4)) || (rc = dm((void **)&h->actions, h->actions_elems * 4)))
        return bail(rc);
    switch (cfg->env_kind) {
        case GRL_ENV_SWARM: rc = swarm_alloc(h); break;
        case GRL_ENV_SOLOW: rc = solow_alloc(h); break;
        case GRL_ENV_TICKER: rc = ticker_alloc(h); break;
        default: rc = trade_alloc(h); break;
    }
    if (rc) return bail(rc);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail(hip_fail(h, e, "hipStreamSynchronize"));
    *out = h;
    return GRL_OK;
}

int grl_destroy(grl_handle *h) {
    if (!h) return GRL_OK;
    hipSetDevice(h->cfg.device_id);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) hipFree(p);
    for (void *p : h->user_allocs) hipFree(p);
    for (hipEvent_t ev : h->prof_ev) hipEventDestroy(ev);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    {
        std::lock_guard<std::mutex> lk(g_live_mutex);
        g_live_handles.erase(h);
    }
    delete h;
    return GRL_OK;
}

const char *grl_last_error(const grl_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int grl_reset(grl_handle *h, const int32_t *env_idx, int32_t n) {
    if (!h) return GRL_E_INVALID;
    hipSetDevice(h->cfg.device_id);
    int32_t cnt;
    if (env_idx == nullptr) {
        cnt = h->E;
        int rc = launch_iota(h, h->done_list, h->E);
        if (rc) return rc;
    } else {
        if (n < 0 || n > h->E) return fail(h, GRL_E_INVALID, "grl_reset: n out of range");
        for (int i = 0; i < n; ++i)
            if (env_idx[i] < 0 || env_idx[i] >= h->E) return fail(h, GRL_E_INVALID, "grl_reset: env index out of range");
        cnt = n;
        GRL_HIP(h, hipMemcpyAsync(h->done_list, env_idx, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    }
    GRL_HIP(h, hipMemcpyAsync(h->done_count, &cnt, 4, hipMemcpyHostToDevice, h->stream));
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    if (cnt > 0) {
        int rc = reset_list(h, h->done_list, h->done_count, cnt);
        if (rc) return rc;
    }
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    return GRL_OK;
}

int grl_swarm_reset_injected(grl_handle *h, const double *x0, const double *xa0, const double *random_actions,
                             const double *agent_noise, const double *particle_noise) {
    if (!h || h->cfg.env_kind != GRL_ENV_SWARM) return fail(h, GRL_E_INVALID, "grl_swarm_reset_injected: not a Swarm handle");
    if (!x0 || !xa0 || !random_actions || !agent_noise || !particle_noise) return fail(h, GRL_E_INVALID, "grl_swarm_reset_injected: null argument");
    hipSetDevice(h->cfg.device_id);
    return swarm_reset_injected(h, x0, xa0, random_actions, agent_noise, particle_noise);
}

int grl_ticker_set_table(grl_handle *h, const double *rows, int32_t nrows) {
    if (!h || h->cfg.env_kind != GRL_ENV_TICKER) return fail(h, GRL_E_INVALID, "grl_ticker_set_table: not a Ticker handle");
    if (!rows) return fail(h, GRL_E_INVALID, "grl_ticker_set_table: null argument");
    hipSetDevice(h->cfg.device_id);
    return ticker_set_table(h, rows, nrows);
}

static int copy_field(grl_handle *h, int32_t field, void *host, size_t bytes, bool to_device) {
    if (!h || !host) return fail(h, GRL_E_INVALID, "grl_set/get_state: null argument");
    hipSetDevice(h->cfg.device_id);
    FieldInfo fi;
    if (!field_info(h, field, fi)) return fail(h, GRL_E_INVALID, "grl_set/get_state: field not valid for this env kind/config");
    size_t E = h->E, total = E * fi.inner * fi.elem;
    if (bytes != total) return fail(h, GRL_E_SIZE, "grl_set/get_state: host buffer is " + std::to_string(bytes) + " bytes, field needs " + std::to_string(total));
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    if (!fi.feature_major || fi.inner == 1) {
        if (to_device) GRL_HIP(h, hipMemcpy(fi.dev, host, total, hipMemcpyHostToDevice));
        else GRL_HIP(h, hipMemcpy(host, fi.dev, total, hipMemcpyDeviceToHost));
        return GRL_OK;
    }
    // (E, inner) on the host <-> (inner, E) on the device; feature-major fields are 4-byte (Solow, injected normals) or 8-byte
    // (TradeAR1 quantity / prices)
    auto transpose = [&](auto *hp, auto &tmp) -> int {
        if (to_device) {
            for (size_t e = 0; e < E; ++e)
                for (size_t i = 0; i < fi.inner; ++i) tmp[i * E + e] = hp[e * fi.inner + i];
            GRL_HIP(h, hipMemcpy(fi.dev, tmp.data(), total, hipMemcpyHostToDevice));
        } else {
            GRL_HIP(h, hipMemcpy(tmp.data(), fi.dev, total, hipMemcpyDeviceToHost));
            for (size_t e = 0; e < E; ++e)
                for (size_t i = 0; i < fi.inner; ++i) hp[e * fi.inner + i] = tmp[i * E + e];
        }
        return GRL_OK;
    };
    if (fi.elem == 8) {
        std::vector<uint64_t> tmp(E * fi.inner);
        return transpose((uint64_t *)host, tmp);
    }
    std::vector<uint32_t> tmp(E * fi.inner);
    return transpose((uint32_t *)host, tmp);
}

int grl_set_state(grl_handle *h, int32_t field, const void *host, size_t bytes) { return copy_field(h, field, (void *)host, bytes, true); }
int grl_get_state(grl_handle *h, int32_t field, void *host, size_t bytes) { return copy_field(h, field, host, bytes, false); }

int grl_step_device(grl_handle *h, const float *actions_dev) {
    if (!h || !actions_dev) return fail(h, GRL_E_INVALID, "grl_step_device: null argument");
    hipSetDevice(h->cfg.device_id);
    int rc;
    switch (h->cfg.env_kind) {
        case GRL_ENV_SWARM: rc = swarm_launch_step(h, actions_dev); break;
        case GRL_ENV_SOLOW: rc = solow_launch_step(h, actions_dev); break;
        case GRL_ENV_TICKER: rc = ticker_launch_step(h, actions_dev); break;
        default: rc = trade_launch_step(h, actions_dev); break;
    }
    if (rc == GRL_OK) rc = episodes_launch_account(h);
    if (rc == GRL_OK) h->step_in_flight = true;
    return rc;
}

int grl_step_async(grl_handle *h, const float *actions_host) {
    if (!h || !actions_host) return fail(h, GRL_E_INVALID, "grl_step_async: null argument");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipMemcpyAsync(h->actions, actions_host, h->actions_elems * 4, hipMemcpyHostToDevice, h->stream));
    return grl_step_device(h, h->actions);
}

int grl_swarm_step_f64(grl_handle *h, const double *actions_host) {
    if (!h || !actions_host) return fail(h, GRL_E_INVALID, "grl_swarm_step_f64: null argument");
    if (h->cfg.env_kind != GRL_ENV_SWARM) return fail(h, GRL_E_INVALID, "grl_swarm_step_f64: not a Swarm handle");
    hipSetDevice(h->cfg.device_id);
    const size_t bytes = (size_t)h->E * N_AGENTS * 2 * sizeof(double);
    if (!h->sw.act64) {
        GRL_HIP(h, hipMalloc((void **)&h->sw.act64, bytes));
        h->allocs.push_back(h->sw.act64);
    }
    GRL_HIP(h, hipMemcpyAsync(h->sw.act64, actions_host, bytes, hipMemcpyHostToDevice, h->stream));
    int rc = swarm_launch_step(h, nullptr, h->sw.act64);
    if (rc == GRL_OK) rc = episodes_launch_account(h);
    if (rc == GRL_OK) h->step_in_flight = true;
    return rc;
}

int grl_swarm_step_opts(grl_handle *h, const void *actions_host, int32_t actions_f64, int32_t add_wind) {
    if (!h || !actions_host) return fail(h, GRL_E_INVALID, "grl_swarm_step_opts: null argument");
    if (h->cfg.env_kind != GRL_ENV_SWARM) return fail(h, GRL_E_INVALID, "grl_swarm_step_opts: not a Swarm handle");
    hipSetDevice(h->cfg.device_id);
    int rc;
    if (actions_f64) {
        const size_t bytes = (size_t)h->E * N_AGENTS * 2 * sizeof(double);
        if (!h->sw.act64) {
            GRL_HIP(h, hipMalloc((void **)&h->sw.act64, bytes));
            h->allocs.push_back(h->sw.act64);
        }
        GRL_HIP(h, hipMemcpyAsync(h->sw.act64, actions_host, bytes, hipMemcpyHostToDevice, h->stream));
        rc = swarm_launch_step(h, nullptr, h->sw.act64, add_wind ? 0 : 1);
    } else {
        GRL_HIP(h, hipMemcpyAsync(h->actions, actions_host, h->actions_elems * 4, hipMemcpyHostToDevice, h->stream));
        rc = swarm_launch_step(h, h->actions, nullptr, add_wind ? 0 : 1);
    }
    if (rc == GRL_OK) rc = episodes_launch_account(h);
    if (rc == GRL_OK) h->step_in_flight = true;
    return rc;
}

int grl_device_pci_address(int32_t device_id, char *out, size_t bytes) {
    if (!out || bytes < 16) return GRL_E_INVALID;
    char buf[64] = {0};
    if (hipDeviceGetPCIBusId(buf, (int)sizeof(buf), device_id) != hipSuccess) { (void)hipGetLastError(); return GRL_E_HIP; }
    for (char *c = buf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');      // sysfs spells addresses in lower case
    strncpy(out, buf, bytes - 1);
    out[bytes - 1] = 0;
    return GRL_OK;
}

int grl_wait(grl_handle *h) {
    if (!h) return GRL_E_INVALID;
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    if (!h->step_in_flight) return GRL_OK;
    h->step_in_flight = false;
    if (h->cfg.env_kind != GRL_ENV_SWARM) {
        int32_t flag = 0;
        GRL_HIP(h, hipMemcpy(&flag, h->err_flag, 4, hipMemcpyDeviceToHost));
        if (flag) {
            GRL_HIP(h, hipMemset(h->err_flag, 0, 4));
            if (h->cfg.env_kind == GRL_ENV_TICKER) {
                if (flag & 0xFFFF) return fail(h, GRL_E_ACTION_RANGE, std::to_string(flag & 0xFFFF) + " env(s) received a choice other than 0 (hold), 1 (buy), 2 (sell)");
                return fail(h, GRL_E_STATE, std::to_string(flag >> 16) + " env(s) stepped past row 1023 of their price window (reference: IndexError, fed_env.py:133)");
            }
            if (h->cfg.env_kind == GRL_ENV_TRADE)
                return fail(h, GRL_E_ACTION_RANGE, std::to_string(flag) + " env(s) received an action outside Box(-1, 1)");
            return fail(h, GRL_E_STATE, std::to_string(flag) + " env(s) popped from an empty shock tape (no TimeLimit and more than T steps)");
        }
    }
    return GRL_OK;
}

int grl_outputs(grl_handle *h, grl_out_ptrs *o) {
    if (!h || !o) return GRL_E_INVALID;
    memset(o, 0, sizeof(*o));
    o->reward = h->reward; o->done = h->done; o->elapsed = h->elapsed;
    o->done_list = h->done_list; o->done_count = h->done_count;
    if (h->cfg.env_kind == GRL_ENV_SWARM) {
        o->reward_f64 = h->sw.reward64; o->locust_bins = h->sw.lbins; o->agent_bins = h->sw.abins; o->positions = h->sw.pos;
    } else if (h->cfg.env_kind == GRL_ENV_SOLOW) {
        o->obs_raw = h->so.obs_raw; o->obs = h->so.obs; o->history = h->so.history;
    } else if (h->cfg.env_kind == GRL_ENV_TICKER) {
        o->obs_raw = h->tk.obs_raw; o->obs = h->tk.obs; o->reward_f64 = h->tk.reward64;
    } else {
        o->obs_raw = h->tr.obs_raw; o->obs = h->tr.obs;
    }
    return GRL_OK;
}

int grl_read_output(grl_handle *h, const char *which, void *host, size_t bytes) {
    if (!h || !which || !host) return fail(h, GRL_E_INVALID, "grl_read_output: null argument");
    hipSetDevice(h->cfg.device_id);
    grl_out_ptrs o;
    grl_outputs(h, &o);
    size_t E = h->E, S = h->cfg.env_kind == GRL_ENV_SOLOW ? 2 : (h->cfg.env_kind == GRL_ENV_TICKER ? 7 : (size_t)(1 + 2 * h->cfg.n_assets));
    const void *src = nullptr;
    size_t need = 0;
    std::string w(which);
    if (w == "reward") { src = o.reward; need = E * 4; }
    else if (w == "reward_f64") { src = o.reward_f64; need = E * 8; }
    else if (w == "done") { src = o.done; need = E; }
    else if (w == "elapsed") { src = o.elapsed; need = E * 4; }
    else if (w == "locust_bins") { src = o.locust_bins; need = E * 160; }
    else if (w == "agent_bins") { src = o.agent_bins; need = E * 20; }
    else if (w == "positions") { src = o.positions; need = E * 20; }
    else if (w == "obs_raw") { src = o.obs_raw; need = E * S * 4; }
    else if (w == "obs") { src = o.obs; need = E * S * 4; }
    else if (w == "history") { src = o.history; need = E * (size_t)h->cfg.rnn_length * 2 * 4; }
    else if (w == "done_count") { src = o.done_count; need = 4; }
    else if (w == "done_list") { src = o.done_list; need = bytes; if (bytes > E * 4) need = 0; }
    else return fail(h, GRL_E_INVALID, "grl_read_output: unknown output '" + w + "'");
    if (!src) return fail(h, GRL_E_INVALID, "grl_read_output: '" + w + "' does not exist for this env kind");
    if (need != bytes) return fail(h, GRL_E_SIZE, "grl_read_output: '" + w + "' needs " + std::to_string(need) + " bytes, got " + std::to_string(bytes));
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    GRL_HIP(h, hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost));
    if (w == "done_list") {      // the device compacts finished envs in wave / workgroup arrival order: hand them out sorted
        int32_t cnt = 0;
        GRL_HIP(h, hipMemcpy(&cnt, o.done_count, 4, hipMemcpyDeviceToHost));
        int32_t *ids = (int32_t *)host;
        const size_t n = (size_t)cnt < bytes / 4 ? (size_t)cnt : bytes / 4;
        std::sort(ids, ids + n);
    }
    return GRL_OK;
}

int grl_observe(grl_handle *h) {
    if (!h) return GRL_E_INVALID;
    hipSetDevice(h->cfg.device_id);
    int rc;
    switch (h->cfg.env_kind) {
        case GRL_ENV_SWARM: rc = swarm_launch_observe(h); break;
        case GRL_ENV_SOLOW: rc = solow_launch_observe(h); break;
        case GRL_ENV_TICKER: rc = ticker_launch_observe(h); break;
        default: rc = trade_launch_observe(h); break;
    }
    if (rc) return rc;
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    return GRL_OK;
}

int grl_swarm_materialize_states(grl_handle *h, int32_t first, int32_t count, float *out_host, size_t bytes) {
    if (!h || h->cfg.env_kind != GRL_ENV_SWARM || !out_host) return fail(h, GRL_E_INVALID, "grl_swarm_materialize_states: bad argument");
    if (first < 0 || count <= 0 || first + count > h->E) return fail(h, GRL_E_INVALID, "grl_swarm_materialize_states: range out of bounds");
    hipSetDevice(h->cfg.device_id);
    size_t G = h->cfg.grid_size, need = (size_t)count * N_AGENTS * G * G * 3 * 4;
    if (need != bytes) return fail(h, GRL_E_SIZE, "grl_swarm_materialize_states: need " + std::to_string(need) + " bytes");
    if (2 * G * G * 4 > 64 * 1024) return fail(h, GRL_E_INVALID, "grl_swarm_materialize_states: grid too large for the dense compat path");
    float *d = nullptr;
    GRL_HIP(h, hipMalloc((void **)&d, need));
    int rc = swarm_materialize(h, first, count, d);
    hipError_t e = hipStreamSynchronize(h->stream);
    if (rc == GRL_OK && e == hipSuccess) e = hipMemcpy(out_host, d, need, hipMemcpyDeviceToHost);
    hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(h, e, "grl_swarm_materialize_states");
    return GRL_OK;
}

int grl_transform_actions_device(grl_handle *h, float *actions_dev, int32_t rows) {
    if (!h || !actions_dev || rows < 0) return fail(h, GRL_E_INVALID, "grl_transform_actions_device: bad argument");
    hipSetDevice(h->cfg.device_id);
    int cols = h->cfg.env_kind == GRL_ENV_SWARM ? 2 : action_cols(h);
    if (rows == 0) return GRL_OK;
    return launch_transform(h, h->cfg.env_kind, actions_dev, rows, cols);
}

int grl_transform_actions_host(grl_handle *h, const float *in, float *out, int32_t rows) {
    if (!h || !in || !out || rows < 0) return fail(h, GRL_E_INVALID, "grl_transform_actions_host: bad argument");
    if (rows == 0) return GRL_OK;
    hipSetDevice(h->cfg.device_id);
    int cols = h->cfg.env_kind == GRL_ENV_SWARM ? 2 : action_cols(h);
    size_t bytes = (size_t)rows * cols * 4;
    float *d = nullptr;
    GRL_HIP(h, hipMalloc((void **)&d, bytes));
    hipError_t e = hipMemcpyAsync(d, in, bytes, hipMemcpyHostToDevice, h->stream);
    int rc = GRL_OK;
    if (e == hipSuccess) rc = launch_transform(h, h->cfg.env_kind, d, rows, cols);
    if (e == hipSuccess && rc == GRL_OK) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && rc == GRL_OK) e = hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost);
    hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(h, e, "grl_transform_actions_host");
    return GRL_OK;
}

int grl_returns_device(grl_handle *h, const float *rewards, const float *values, const float *mask, const float *boot,
                       int32_t T, int32_t B, float gamma, float lam, float scale, float clip_lo, float clip_hi,
                       float *y_out, float *adv_out) {
    if (!h || !rewards || !values || !boot || !y_out || !adv_out || T <= 0 || B <= 0 || scale == 0.f)
        return fail(h, GRL_E_INVALID, "grl_returns: bad argument");
    hipSetDevice(h->cfg.device_id);
    return launch_returns(h, rewards, values, mask, boot, T, B, gamma, lam, scale, clip_lo, clip_hi, y_out, adv_out);
}

int grl_returns(grl_handle *h, const float *rewards, const float *values, const float *mask, const float *boot, int32_t T,
                int32_t B, float gamma, float lam, float scale, float clip_lo, float clip_hi, float *y_out,
                float *adv_out) {
    if (!h || !rewards || !values || !boot || !y_out || !adv_out || T <= 0 || B <= 0 || scale == 0.f)
        return fail(h, GRL_E_INVALID, "grl_returns: bad argument");
    hipSetDevice(h->cfg.device_id);
    size_t tb = (size_t)T * B;
    float *d = nullptr;
    GRL_HIP(h, hipMalloc((void **)&d, (5 * tb + B) * 4));
    float *dr = d, *dv = d + tb, *dm = d + 2 * tb, *dy = d + 3 * tb, *da = d + 4 * tb, *db = d + 5 * tb;
    hipError_t e = hipMemcpyAsync(dr, rewards, tb * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dv, values, tb * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess && mask) e = hipMemcpyAsync(dm, mask, tb * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db, boot, (size_t)B * 4, hipMemcpyHostToDevice, h->stream);
    int rc = GRL_OK;
    if (e == hipSuccess) rc = launch_returns(h, dr, dv, mask ? dm : nullptr, db, T, B, gamma, lam, scale, clip_lo, clip_hi, dy, da);
    if (e == hipSuccess && rc == GRL_OK) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && rc == GRL_OK) e = hipMemcpy(y_out, dy, tb * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && rc == GRL_OK) e = hipMemcpy(adv_out, da, tb * 4, hipMemcpyDeviceToHost);
    hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return hip_fail(h, e, "grl_returns");
    return GRL_OK;
}

int grl_dev_alloc(grl_handle *h, size_t bytes, void **out_dev) {
    if (!h || !out_dev || bytes == 0) return fail(h, GRL_E_INVALID, "grl_dev_alloc: bad argument");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipMalloc(out_dev, bytes));
    h->user_allocs.push_back(*out_dev);
    GRL_HIP(h, hipMemsetAsync(*out_dev, 0, bytes, h->stream));
    return GRL_OK;
}

int grl_dev_free(grl_handle *h, void *dev) {
    if (!h) return GRL_E_INVALID;
    auto it = std::find(h->user_allocs.begin(), h->user_allocs.end(), dev);
    if (it == h->user_allocs.end()) return fail(h, GRL_E_INVALID, "grl_dev_free: pointer was not returned by grl_dev_alloc on this handle");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    h->user_allocs.erase(it);
    GRL_HIP(h, hipFree(dev));
    return GRL_OK;
}

int grl_dev_upload(grl_handle *h, void *dst_dev, const void *src_host, size_t bytes) {
    if (!h || !dst_dev || !src_host) return fail(h, GRL_E_INVALID, "grl_dev_upload: null argument");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, h->stream));
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    return GRL_OK;
}

int grl_dev_download(grl_handle *h, void *dst_host, const void *src_dev, size_t bytes) {
    if (!h || !dst_host || !src_dev) return fail(h, GRL_E_INVALID, "grl_dev_download: null argument");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    GRL_HIP(h, hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return GRL_OK;
}

int grl_dev_copy(grl_handle *h, void *dst_dev, const void *src_dev, size_t bytes) {
    if (!h || !dst_dev || !src_dev) return fail(h, GRL_E_INVALID, "grl_dev_copy: null argument");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, h->stream));
    return GRL_OK;
}

int grl_profile_enable(grl_handle *h, int32_t on) {
    if (!h) return GRL_E_INVALID;
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    if (on && h->prof_ev.empty()) {
        h->prof_ev.resize(8192);
        for (auto &ev : h->prof_ev) GRL_HIP(h, hipEventCreate(&ev));
    }
    h->prof_on = on != 0;
    h->prof_used = 0;
    return GRL_OK;
}

int grl_profile_read(grl_handle *h, int32_t *launches_out, float *total_ms_out) {
    if (!h || !launches_out || !total_ms_out) return GRL_E_INVALID;
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    float total = 0.f;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0.f;
        GRL_HIP(h, hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
        total += ms;
    }
    *launches_out = (int32_t)(h->prof_used / 2);
    *total_ms_out = total;
    return GRL_OK;
}

int grl_episodes_enable(grl_handle *h, int32_t capacity) {
    if (!h || capacity <= 0) return fail(h, GRL_E_INVALID, "grl_episodes_enable: capacity must be positive");
    if (h->ep_total) return fail(h, GRL_E_STATE, "grl_episodes_enable: already enabled");
    hipSetDevice(h->cfg.device_id);
    const size_t E = h->E;
    void *p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t bytes[5] = {E * 8, E * 4, E * 8, (size_t)capacity * sizeof(grl_episode_record), 4};
    for (int i = 0; i < 5; ++i) {
        GRL_HIP(h, hipMalloc(&p[i], bytes[i]));
        h->allocs.push_back(p[i]);
        GRL_HIP(h, hipMemsetAsync(p[i], 0, bytes[i], h->stream));
    }
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    h->ep_len = (int32_t *)p[1]; h->ep_steps = (int64_t *)p[2]; h->ep_rec = (grl_episode_record *)p[3]; h->ep_count = (int32_t *)p[4];
    h->ep_capacity = capacity;
    h->ep_total = (double *)p[0];      // set last: this is the "enabled" flag
    return GRL_OK;
}

int grl_episodes_read(grl_handle *h, grl_episode_record *out, int32_t max_records, int32_t *n_out, int32_t *dropped_out) {
    if (!h || !n_out || max_records < 0 || (max_records > 0 && !out)) return fail(h, GRL_E_INVALID, "grl_episodes_read: bad argument");
    if (!h->ep_total) return fail(h, GRL_E_STATE, "grl_episodes_read: call grl_episodes_enable first");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    int32_t cnt = 0;
    GRL_HIP(h, hipMemcpy(&cnt, h->ep_count, 4, hipMemcpyDeviceToHost));
    const int32_t stored = cnt < h->ep_capacity ? cnt : h->ep_capacity;
    if (stored > max_records) return fail(h, GRL_E_SIZE, "grl_episodes_read: " + std::to_string(stored) + " finished episodes, room for " + std::to_string(max_records));
    if (stored > 0) {
        GRL_HIP(h, hipMemcpy(out, h->ep_rec, (size_t)stored * sizeof(grl_episode_record), hipMemcpyDeviceToHost));
        std::sort(out, out + stored, [](const grl_episode_record &a, const grl_episode_record &b) {
            return a.step_index != b.step_index ? a.step_index < b.step_index : a.env < b.env;
        });
    }
    GRL_HIP(h, hipMemset(h->ep_count, 0, 4));
    *n_out = stored;
    if (dropped_out) *dropped_out = cnt - stored;
    return GRL_OK;
}

int grl_episodes_running(grl_handle *h, double *total_reward_out, int32_t *length_out) {
    if (!h) return GRL_E_INVALID;
    if (!h->ep_total) return fail(h, GRL_E_STATE, "grl_episodes_running: call grl_episodes_enable first");
    hipSetDevice(h->cfg.device_id);
    GRL_HIP(h, hipStreamSynchronize(h->stream));
    if (total_reward_out) GRL_HIP(h, hipMemcpy(total_reward_out, h->ep_total, (size_t)h->E * 8, hipMemcpyDeviceToHost));
    if (length_out) GRL_HIP(h, hipMemcpy(length_out, h->ep_len, (size_t)h->E * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}

int grl_dev_randn(grl_handle *h, float *dst_dev, size_t n, uint32_t stream, uint64_t counter) {
    if (!h || !dst_dev) return fail(h, GRL_E_INVALID, "grl_dev_randn: null argument");
    if (n == 0) return GRL_OK;
    hipSetDevice(h->cfg.device_id);
    return launch_randn(h, dst_dev, n, stream, counter);
}

int grl_stream(grl_handle *h, void **out_stream) {
    if (!h || !out_stream) return GRL_E_INVALID;
    *out_stream = (void *)h->stream;
    return GRL_OK;
}

int grl_timer_start(grl_handle *h) {
    if (!h) return GRL_E_INVALID;
    GRL_HIP(h, hipEventRecord(h->ev0, h->stream));
    return GRL_OK;
}
int grl_timer_stop(grl_handle *h) {
    if (!h) return GRL_E_INVALID;
    GRL_HIP(h, hipEventRecord(h->ev1, h->stream));
    return GRL_OK;
}
int grl_timer_ms(grl_handle *h, float *ms_out) {
    if (!h || !ms_out) return GRL_E_INVALID;
    GRL_HIP(h, hipEventSynchronize(h->ev1));
    GRL_HIP(h, hipEventElapsedTime(ms_out, h->ev0, h->ev1));
    return GRL_OK;
}

}  // extern "C"
