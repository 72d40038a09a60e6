// Internal declarations shared by the translation units of libgoldsrl.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/goldsrl.h"

namespace grl {

constexpr int N_LOCUSTS = 80;   // envs/multiagent.py:8
constexpr int N_AGENTS = 10;    // envs/multiagent.py:9
constexpr int N_POINTS = 90;
constexpr int N_BURN_IN = 10;   // envs/multiagent.py:21
constexpr int SWARM_EPB = 4;    // envs per workgroup: 4*80 = 320 lanes = 5 full waves
constexpr int SWARM_TPB = SWARM_EPB * N_LOCUSTS;

// generator stream ids (ctr[3]); the CPU restatement is oracle/oracle.py:rng_block
enum : uint32_t {
    RS_SWARM_X0 = 0, RS_SWARM_XA0 = 1, RS_SWARM_RANDACT = 2, RS_SWARM_ANOISE = 3, RS_SWARM_PNOISE = 4,
    RS_SOLOW_Z0 = 8, RS_SOLOW_TAPE = 9, RS_TRADE_PRICE = 12, RS_TICKER_START = 16, RS_USER = 64
};

struct SwarmState {
    double *x, *xa, *pnoise, *anoise;              // (E,80,2) (E,10,2) (E,80,2) (E,10,2)
    double *rx, *rxa, *rpnoise, *ranoise;          // reset snapshot (allocated on first use)
    double *reward64;
    double *act64;                                 // staging of grl_swarm_step_f64 (allocated on first use)
    uint8_t *lbins, *abins, *pos;
};

struct SolowState {
    float *k, *z, *e, *z0;      // k (E), z (P,E) feature-major, e (Q,E), z0 (P,E)
    float *tape;                // (T,E) time-major
    int32_t *tape_pos, *nhist;
    float *obs_raw, *obs, *history;   // (E,2) (E,2) (E,rnn,2)
    float rho_z[8], rho_e[8];
    int P, Q;                   // storage orders (>=1)
};

struct TradeState {
    double *cash, *assets, *q, *p;    // float64 like the reference's numpy state (the reward is a difference of logs of two
                                      // nearly equal asset values: float32 prices would leave it at 1e-4 relative); q,p (n,E) asset-major
    float *normals;                   // (n,E) injected
    uint32_t *nstep;                  // generator counter per env
    int32_t *nhist;                   // states in the (PAAC-style) worker history list, as SolowState::nhist
    float *obs_raw, *obs;             // (E,1+2n)
    double std_e;
    double start;                     // starting_balance (fed_env.py:269,323-326)
};

struct TickerState {
    double *cash, *assets, *q;        // q (E,2)
    double *reward64;
    int32_t *idx, *start, *start0;    // row inside the window, first row of the window, start restored under SNAPSHOT
    int32_t *nhist;
    float *obs_raw, *obs;             // (E,7)
    const double *table;              // (rows,4) on the device: price, inverse price, volume, volume
    int rows;
};

}  // namespace grl

struct grl_handle {
    grl_config cfg;
    int E;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    std::string err;
    bool step_in_flight;
    // common per-env
    int32_t *elapsed, *episode;
    float *reward;
    uint8_t *done;
    int32_t *done_list, *done_count;
    int32_t *err_flag;            // device-side deferred error counter (Trade action range)
    float *actions;               // staging for grl_step_async
    size_t actions_elems;
    grl::SwarmState sw;
    grl::SolowState so;
    grl::TradeState tr;
    grl::TickerState tk;
    std::vector<void *> allocs;   // everything hipMalloc'ed by the handle
    std::vector<void *> user_allocs;
    // R6 episode bookkeeping (grl_episodes_*): nullptr until enabled
    double *ep_total;
    int32_t *ep_len;
    int64_t *ep_steps;
    grl_episode_record *ep_rec;
    int32_t *ep_count;
    int32_t ep_capacity;
    // per-kernel profiling (grl_profile_*)
    bool prof_on;
    std::vector<hipEvent_t> prof_ev;   // pairs
    size_t prof_used;
};

namespace grl {
int fail(grl_handle *h, int code, const std::string &msg);
// Is h a handle grl_create returned and grl_destroy has not freed?  A net keeps a pointer to the handle it was created on; a
// host binding's garbage collector may finalise the two in either order, so the grl_*net_destroy functions ask before they
// touch the handle's device id and stream.
bool grl_handle_alive(const grl_handle *h);
// binds the handle's device and drains its stream if the handle is alive, else drains the current device
void grl_sync_for_destroy(grl_handle *h);
// bracket the dominant kernel of a step with an event pair when profiling is on
void prof_begin(grl_handle *h);
void prof_end(grl_handle *h);
int hip_fail(grl_handle *h, hipError_t e, const char *what);
#define GRL_HIP(h, call)                                        \
    do {                                                        \
        hipError_t _e = (call);                                 \
        if (_e != hipSuccess) return grl::hip_fail(h, _e, #call); \
    } while (0)

// api.hip: accounts the step just enqueued on the handle's stream (no-op until grl_episodes_enable)
int episodes_launch_account(grl_handle *h, int env_base = 0, int count = -1);
// swarm.hip
int swarm_alloc(grl_handle *h);
int swarm_launch_step(grl_handle *h, const float *actions_dev, const double *actions64_dev = nullptr, int no_wind = 0);
int swarm_launch_step_range(grl_handle *h, const float *actions_dev, int env_base, int count, int slot, bool counter_zeroed = false);
int swarm_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count);
int swarm_launch_observe(grl_handle *h);
int swarm_reset_injected(grl_handle *h, const double *x0, const double *xa0, const double *ra,
                         const double *an, const double *pn);
int swarm_materialize(grl_handle *h, int first, int count, float *out_dev);
// flat_envs.hip
int solow_alloc(grl_handle *h);
int solow_launch_step(grl_handle *h, const float *actions_dev);
int solow_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count, bool advance_episode);
int solow_launch_observe(grl_handle *h);
int trade_alloc(grl_handle *h);
int trade_launch_step(grl_handle *h, const float *actions_dev);
int trade_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count);
int trade_launch_observe(grl_handle *h);
// ticker.hip
int ticker_alloc(grl_handle *h);
int ticker_set_table(grl_handle *h, const double *rows_host, int nrows);
int ticker_launch_step(grl_handle *h, const float *actions_dev);
int ticker_launch_reset(grl_handle *h, const int32_t *list_dev, const int32_t *count_dev, int max_count);
int ticker_launch_observe(grl_handle *h);
// rollout_math.hip
int launch_returns(grl_handle *h, const float *r, const float *v, const float *mask, const float *boot, int T, int B,
                   float gamma, float lam, float scale, float clip_lo, float clip_hi, float *y, float *adv);
int launch_transform(grl_handle *h, int kind, float *actions_dev, int rows, int cols);
int launch_randn(grl_handle *h, float *dst, size_t n, uint32_t stream, uint64_t counter);
int launch_iota(grl_handle *h, int32_t *dst, int n);
}  // namespace grl
