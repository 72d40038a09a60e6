/* goldsrl.h -- C ABI of libgoldsrl.so: the MI355X-native batched rollout engine that
 * replaces the reference's PAAC actor loop (allentran/golds-rl-gym, fed_gym).
 *
 * Boundary being replaced (paths relative to the reference repo root):
 *   fed_gym/agents/paac/runners.py:11-66        Runners / GridRunners  (start/stop/
 *       get_shared_variables/update_environments/wait_updated)  -> grl_create .. grl_wait
 *   fed_gym/agents/paac/emulator_runner.py:38-151  EmulatorRunner._run bodies (step, auto-reset,
 *       process_state, history window, reward/done slots)        -> grl_step_* + grl_outputs
 *   fed_gym/envs/multiagent.py:30-115, fed_gym/envs/fed_env.py:161-334  env dynamics
 *   fed_gym/agents/state_processors.py:15-42,69-76               observation transforms
 *   fed_gym/agents/paac/paac.py:159-172,351-372                  n-step return / advantage
 *   fed_gym/agents/a3c/worker.py:17-34,232-239,420-442           sigmoid, GAE, Trade transforms
 *   fed_gym/agents/paac/policy_v_network.py:5-80,194-264 + actor_learner.py:31-68
 *       policy/value nets, loss, clip-by-global-norm, Adam       -> grl_net_* (goldsrl_net.h)
 *
 * Conventions: every entry point is extern "C", takes plain pointers and sizes, returns
 * 0 (GRL_OK) or a negative GRL_E_* code; the text of the last error is available from
 * grl_last_error().  Nothing throws, nothing aborts.  The caller owns every HOST buffer it
 * passes; the library owns the opaque handle and the device memory behind it.  A handle is
 * bound to one device and one HIP stream and is NOT thread-safe; distinct handles (one per
 * GPU / per process) are independent.  There is no CPU backend: without a usable HIP device
 * grl_create() fails with GRL_E_NO_DEVICE.
 */
#ifndef GOLDSRL_H
#define GOLDSRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRL_ABI_VERSION 1

/* ---- error codes ------------------------------------------------------------------ */
#define GRL_OK 0
#define GRL_E_INVALID (-1)      /* bad argument / bad config / wrong field for this env kind */
#define GRL_E_NO_DEVICE (-2)    /* no HIP device, or device is not gfx950-class */
#define GRL_E_HIP (-3)          /* a HIP runtime call failed (message has the HIP error string) */
#define GRL_E_SIZE (-4)         /* host buffer size does not match the field */
#define GRL_E_ACTION_RANGE (-5) /* TradeAR1: action outside Box(-1,1) (reference: AssertionError, fed_env.py:301) */
#define GRL_E_STATE (-6)        /* call order violated (e.g. grl_wait without a step in flight) */
#define GRL_E_COMM (-7)         /* RCCL failure */
#define GRL_E_RANGE (-8)        /* conv policy: an activation / gradient left the fp16 range of the matrix-pipe GEMMs (goldsrl_net.h) */

/* ---- environment kinds (gym ids: fed_gym/__init__.py:3-33) -------------------------- */
#define GRL_ENV_SWARM 0 /* Swarm-v0 / Swarm-eval-v0   (envs/multiagent.py) */
#define GRL_ENV_SOLOW 1 /* Solow-v0 / Solow-p-q-*-v0  (envs/fed_env.py:161-250) */
#define GRL_ENV_TRADE 2 /* TradeAR1-v0                (envs/fed_env.py:268-334) */
#define GRL_ENV_TICKER 3 /* TickerEnv over an OpenCloseSampler table (envs/fed_env.py:89-158, envs/data/sampler.py:8-41);
                          * needs grl_ticker_set_table before the first reset.  Actions (E,4) f32: choice0, choice1
                          * (0 hold, 1 BUY_IDX, 2 SELL_IDX), fraction0, fraction1.  Observation (E,7):
                          * [cash, q0, q1, p0, p1, v0, v1]; `obs` is TickerTraderStateProcessor.process_state of it
                          * (agents/state_processors.py:50-63).  The reference registers no TimeLimit for this env and
                          * raises IndexError at row 1024 of its window: grl_config_default sets max_episode_steps = 1023 */

/* ---- config flags ------------------------------------------------------------------- */
/* Reset reproduces the same episode every time, as an env built with seed=<n> does because
 * it calls np.random.seed(n) on every reset (multiagent.py:47-48, fed_env.py:245-246). */
#define GRL_F_RESEED_EACH_RESET 0x1u
/* Reset restores the snapshot stored in the GRL_FLD_RESET_* fields instead of drawing from
 * the device generator (parity tests inject the reference's own reset state here). */
#define GRL_F_RESET_FROM_SNAPSHOT 0x2u
/* TradeAR1: price shocks are read from GRL_FLD_TRADE_NORMALS (host-injected before every
 * step) instead of the device generator. */
#define GRL_F_INJECT_NOISE 0x4u
/* Swarm: one exp() per pair (e^-r = (e^-r/10)^10) and one reciprocal per pair instead of the
 * reference's two exp and two divisions; positions then agree to ~1e-14 instead of ~1e-16. */
#define GRL_F_SWARM_FAST_MATH 0x8u
/* Swarm: do not run the observation (binning) stage inside the step kernel. */
#define GRL_F_SWARM_NO_OBSERVE 0x10u
/* Solow: SolowSSEnv's reset (fed_env.py:257-265; id SolowSS-v0, fed_gym/__init__.py:15-19): z = 0 instead of z ~ N(0, sigma),
 * e = 0, k = k_ss(alpha); the shock tape is still drawn.  Use with solow_p = 1, solow_q = 0, solow_sigma = 0.02. */
#define GRL_F_SOLOW_SS_RESET 0x20u

typedef struct grl_config {
    int32_t struct_size;       /* = sizeof(grl_config); guards against ABI drift */
    int32_t env_kind;          /* GRL_ENV_* */
    int32_t num_envs;          /* E, envs owned by THIS handle (this GPU's shard) */
    int32_t device_id;         /* HIP device ordinal */
    int32_t max_episode_steps; /* gym TimeLimit; 0 = no limit. 128 Swarm, 1024 Solow/Trade */
    int32_t grid_size;         /* Swarm observation bins per axis (train_paac_conv.py --height, 84); <=254 */
    int32_t n_assets;          /* TradeAR1 n (fed_env.py:269); 1..64 */
    int32_t solow_p;           /* AR order of the TFP shock, 0..8 (fed_env.py:166) */
    int32_t solow_q;           /* MA order, 0..8 */
    int32_t solow_tape_len;    /* T, shocks drawn per reset (fed_env.py:176), default 2048 */
    int32_t rnn_length;        /* history window (train_paac_solow.py --rnn-length, 5); 1..16 */
    uint32_t flags;            /* GRL_F_* */
    uint64_t seed;             /* device generator key */
    int64_t env_id_offset;     /* global id of local env 0: generator streams are keyed by global id,
                                  so results do not depend on how envs are sharded over GPUs */
    double solow_sigma;        /* 0.1 */
    double solow_delta;        /* 0.02 */
    double trade_std_p;        /* 0.05 */
    double trade_starting_balance; /* TradeAR1Env(starting_balance=10.) (fed_env.py:269,323-326): cash and assets after every reset; > 0 */
} grl_config;

typedef struct grl_handle grl_handle;

/* ---- state fields for grl_set_state / grl_get_state ----------------------------------
 * Layout is what the host passes/receives (C order); E = num_envs.  "f64"/"f32"/"i32"/"u8". */
enum grl_field {
    /* Swarm (state of envs/multiagent.py:51-57) */
    GRL_FLD_SWARM_X = 0,        /* f64 (E,80,2) locust positions */
    GRL_FLD_SWARM_XA = 1,       /* f64 (E,10,2) agent positions */
    GRL_FLD_SWARM_PNOISE = 2,   /* f64 (E,80,2) particle_noise[t] row in use (row 10: quirk Q1) */
    GRL_FLD_SWARM_ANOISE = 3,   /* f64 (E,10,2) agent_noise[t] row in use */
    GRL_FLD_RESET_X = 4,        /* f64 (E,80,2) snapshot restored on reset (GRL_F_RESET_FROM_SNAPSHOT) */
    GRL_FLD_RESET_XA = 5,       /* f64 (E,10,2) */
    GRL_FLD_RESET_PNOISE = 6,   /* f64 (E,80,2) */
    GRL_FLD_RESET_ANOISE = 7,   /* f64 (E,10,2) */
    /* common */
    GRL_FLD_ELAPSED = 8,        /* i32 (E,)  TimeLimit counter */
    GRL_FLD_EPISODE = 9,        /* i32 (E,)  episodes completed (generator counter) */
    /* Solow (fed_env.py:242-250) */
    GRL_FLD_SOLOW_K = 16,       /* f32 (E,)   capital */
    GRL_FLD_SOLOW_Z = 17,       /* f32 (E,p)  shock history, oldest..newest (p>=1 storage even if p==0) */
    GRL_FLD_SOLOW_E = 18,       /* f32 (E,q)  innovation history */
    GRL_FLD_SOLOW_TAPE = 19,    /* f32 (E,T)  pre-drawn innovations, consumed from the END (quirk Q8) */
    GRL_FLD_SOLOW_TAPE_POS = 20,/* i32 (E,)   index of the next innovation to pop (T-1 after reset) */
    GRL_FLD_SOLOW_Z0 = 21,      /* f32 (E,p)  z restored on reset under RESEED/SNAPSHOT */
    GRL_FLD_NHIST = 22,         /* i32 (E,)   states in the worker's history list (emulator_runner.py:50-63) */
    /* TradeAR1 (fed_env.py:323-330) */
    /* the account is float64 like the reference's numpy state: the reward is log(assets') - log(assets) of two nearly equal
     * values, which float32 prices would leave at ~1e-4 relative */
    GRL_FLD_TRADE_CASH = 32,    /* f64 (E,) */
    GRL_FLD_TRADE_ASSETS = 33,  /* f64 (E,) */
    GRL_FLD_TRADE_QUANTITY = 34,/* f64 (E,n) */
    GRL_FLD_TRADE_PRICES = 35,  /* f64 (E,n) */
    GRL_FLD_TRADE_NORMALS = 36, /* f32 (E,n) N(0,1) draws for the NEXT step (GRL_F_INJECT_NOISE) */
    /* Ticker (account in float64 like the reference's numpy scalars) */
    GRL_FLD_TICKER_CASH = 48,     /* f64 (E,)   cash_balance */
    GRL_FLD_TICKER_ASSETS = 49,   /* f64 (E,)   equity at the last step's prices (the reward's old_assets) */
    GRL_FLD_TICKER_QUANTITY = 50, /* f64 (E,2)  positions */
    GRL_FLD_TICKER_IDX = 51,      /* i32 (E,)   data_idx: row inside the env's 1024-row window */
    GRL_FLD_TICKER_START = 52,    /* i32 (E,)   first table row of the window in use (sampler.py:38 start_idx) */
    GRL_FLD_TICKER_START0 = 53    /* i32 (E,)   start restored by every reset under GRL_F_RESET_FROM_SNAPSHOT */
};

/* ---- step outputs: DEVICE pointers, valid from grl_wait() until grl_destroy() ----------
 * Pointers not applicable to the env kind are NULL. */
typedef struct grl_out_ptrs {
    const float *reward;        /* (E,)  f32, what the learner reads (runners.py shared c_float) */
    const double *reward_f64;   /* (E,)  Swarm only: the float64 value _step returned */
    const uint8_t *done;        /* (E,)  episode_over (own `done` OR TimeLimit) */
    const int32_t *elapsed;     /* (E,) */
    /* Swarm compact observation (process_state, state_processors.py:29-42): */
    const uint8_t *locust_bins; /* (E,80,2) histogram bin of each locust; 255,255 = outside the box */
    const uint8_t *agent_bins;  /* (E,10,2) histogram bin of each agent (density channel 1); 255 = outside */
    const uint8_t *positions;   /* (E,10,2) np.digitize indices clamped to grid-1 (one-hot channel, quirk Q2) */
    /* Solow / Trade observation: */
    const float *obs_raw;       /* (E,S)  what env.step returned: Solow S=2 [k, z]; Trade S=1+2n */
    const float *obs;           /* (E,S)  processed state fed to the net */
    const float *history;       /* (E,rnn,S) Solow: window fed to the GRU (quirk Q11) */
    /* episode-done compaction of the last step: */
    const int32_t *done_list;   /* (<=E,) local env ids that finished */
    const int32_t *done_count;  /* (1,) */
} grl_out_ptrs;

/* ---- lifecycle ------------------------------------------------------------------------ */
int grl_abi_version(void);
int grl_config_default(int32_t env_kind, grl_config *cfg); /* reference defaults for the kind */
int grl_create(const grl_config *cfg, grl_handle **out);
int grl_destroy(grl_handle *h);
const char *grl_last_error(const grl_handle *h /* NULL: error of a failed grl_create */);

/* gym.Env.reset() for env_idx[0..n) (NULL => all envs). Synchronous. */
int grl_reset(grl_handle *h, const int32_t *env_idx, int32_t n);
/* Swarm: run SwarmEnv._reset's burn-in (multiagent.py:58-61) on caller-supplied draws, all envs:
 * x0 (E,80,2) xa0 (E,10,2) random_actions (E,10,10,2) agent_noise (E,11,10,2) particle_noise (E,11,80,2), f64. */
int grl_swarm_reset_injected(grl_handle *h, const double *x0, const double *xa0, const double *random_actions,
                             const double *agent_noise, const double *particle_noise);

/* Ticker: uploads the price table every env samples its 1024-row windows from -- the `data_matrix` of
 * OpenCloseSampler (envs/data/sampler.py:11-28): rows (nrows,4) f64 C-order = [price, inverse price, volume, volume],
 * nrows >= 1024, prices > 0.  Replaces TickerEnv.__init__'s `self.data = sampler.OpenCloseSampler(...)`
 * (envs/fed_env.py:107).  Must be called before the first grl_reset. */
int grl_ticker_set_table(grl_handle *h, const double *rows, int32_t nrows);
int grl_set_state(grl_handle *h, int32_t field, const void *host, size_t bytes);
int grl_get_state(grl_handle *h, int32_t field, void *host, size_t bytes);

/* ---- the step: Runners.update_environments() / wait_updated() (runners.py:45-54) ------------
 * actions: already transformed for the env (what the learner wrote into shared_actions):
 *   Swarm (E,10,2) f32, Solow (E,1) f32 in (0,1), Trade (E,n) f32 in [-1,1].
 * grl_step_async copies from HOST memory; grl_step_device takes a DEVICE pointer (no copy, the
 * rollout path).  Both only enqueue work on the handle's stream. */
int grl_step_async(grl_handle *h, const float *actions_host);
int grl_step_device(grl_handle *h, const float *actions_dev);
/* Swarm only: the step with FLOAT64 host actions (E,10,2), for callers that hand SwarmEnv.step a float64 array directly -- the
 * eval monitor does (policy_monitor.py:173-178: mu + sigma * np.random.normal is float64), while the learner's actions pass
 * through the float32 shared array (quirk Q7) and use grl_step_async.  The dynamics are chaotic (~1.15x per step), so a
 * float32-rounded action shows up as 4e-4 in the reward after 128 steps. */
int grl_swarm_step_f64(grl_handle *h, const double *actions_host);
/* Swarm only: SwarmEnv._step(v_action, add_wind) with both of its knobs (multiagent.py:30-36): actions_host is (E,10,2) float64
 * (actions_f64 != 0) or float32; add_wind = 0 skips `v_action[:, 0] += WIND_SPEED` for the agents (the locusts' U term of
 * v_calculate, multiagent.py:39, does not depend on it).  add_wind = 1 is grl_step_async / grl_swarm_step_f64. */
int grl_swarm_step_opts(grl_handle *h, const void *actions_host, int32_t actions_f64, int32_t add_wind);
int grl_wait(grl_handle *h);                       /* hipStreamSynchronize + deferred error checks */
/* PCI address "dddd:bb:dd.f" of HIP device `device_id` (hipDeviceGetPCIBusId), for callers that place their host thread next to
 * the GPU (goldsrl/affinity.py; the reference leaves its worker processes' placement to the OS, runners.py:34-36).  `bytes` >= 16. */
int grl_device_pci_address(int32_t device_id, char *out, size_t bytes);
int grl_outputs(grl_handle *h, grl_out_ptrs *out); /* device pointers */
/* Copy one output to the host: which = name of a grl_out_ptrs member, e.g. "reward". */
int grl_read_output(grl_handle *h, const char *which, void *host, size_t bytes);
/* Observation of the CURRENT state without stepping (initial states: paac.py:86,247-251). */
int grl_observe(grl_handle *h);
/* Swarm: dense get_local_states() (emulator_runner.py:98-111) for envs [first, first+count):
 * out (count,10,G,G,3) f32 on the host.  Compat/debug path only; the rollout never builds it. */
int grl_swarm_materialize_states(grl_handle *h, int32_t first, int32_t count, float *out_host, size_t bytes);

/* ---- action transforms (emulator_runner.py:77-79,113-118; a3c/worker.py:440-442) ------------
 * In place on n rows of DEVICE memory, enqueued on the handle's stream. kind: GRL_ENV_*. */
int grl_transform_actions_device(grl_handle *h, float *actions_dev, int32_t rows);
/* Host convenience used by the Python statics: copies in, transforms on the GPU, copies out. */
int grl_transform_actions_host(grl_handle *h, const float *in, float *out, int32_t rows);

/* ---- n-step return / advantage / GAE (paac.py:159-172,360-372; a3c/worker.py:232-239,284-294) --
 * All pointers HOST, (T,B) C-order f32 except boot (B,).  mask = 1-done or NULL (Swarm form).
 * clip_lo<clip_hi: rewards are clipped first (actor_learner.py:91-97).  lam==1 gives the PAAC
 * n-step advantage; lam<1 gives A3C's GAE.  adv is divided by `scale` (paac.py:177,371) and
 * y is the critic target.  Synchronous. */
int grl_returns(grl_handle *h, const float *rewards, const float *values, const float *mask,
                const float *boot, int32_t T, int32_t B, float gamma, float lam, float scale,
                float clip_lo, float clip_hi, float *y_out, float *adv_out);
/* Same on DEVICE pointers, enqueued on the handle's stream (the rollout path). */
int grl_returns_device(grl_handle *h, const float *rewards, const float *values, const float *mask,
                       const float *boot, int32_t T, int32_t B, float gamma, float lam, float scale,
                       float clip_lo, float clip_hi, float *y_out, float *adv_out);

/* ---- device memory helpers for callers that keep rollout buffers on the GPU ---------------- */
int grl_dev_alloc(grl_handle *h, size_t bytes, void **out_dev);
int grl_dev_free(grl_handle *h, void *dev);
int grl_dev_upload(grl_handle *h, void *dst_dev, const void *src_host, size_t bytes);
int grl_dev_download(grl_handle *h, void *dst_host, const void *src_dev, size_t bytes);
int grl_dev_copy(grl_handle *h, void *dst_dev, const void *src_dev, size_t bytes); /* D2D, enqueued on the stream */
/* Fill n float32 with N(0,1) from the handle's generator (stream id `stream`, counter base). */
int grl_dev_randn(grl_handle *h, float *dst_dev, size_t n, uint32_t stream, uint64_t counter);
/* Raw HIP stream of the handle (hipStream_t as void*), so callers can order their own work. */
int grl_stream(grl_handle *h, void **out_stream);
/* Time between two points on the handle's stream with HIP events:
 * grl_timer_start / grl_timer_stop enqueue events; grl_timer_ms syncs and returns the ms. */
int grl_timer_start(grl_handle *h);
int grl_timer_stop(grl_handle *h);
int grl_timer_ms(grl_handle *h, float *ms_out);

/* Per-kernel timing for the roofline line of bench.py: while enabled, every launch of the env
 * step kernel is bracketed by a HIP event pair on the handle's stream (up to 4096 launches).
 * grl_profile_read syncs and returns the number of bracketed launches and their summed duration. */
int grl_profile_enable(grl_handle *h, int32_t on);
int grl_profile_read(grl_handle *h, int32_t *launches_out, float *total_ms_out);

/* ---- R6: the learner's per-env bookkeeping (fed_gym/agents/paac/paac.py:142-157 flat, :331-349 grid) kept on the device next
 * to the env state, because the actor loop runs there.  For every env step taken through the handle after grl_episodes_enable
 * (grl_step_*, grl_net_rollout, grl_fnet_rollout), per env e:
 *     total_episode_rewards[e] += reward[e]   (the float32 the learner reads, summed in float64 as numpy 1.13 does for
 *                                              `0 + np.float32`; unclipped: the clip of :145 applies to the stored reward only)
 *     emulator_steps[e] += 1
 *     if episode_over[e]: append {step_index, e, emulator_steps[e], total_episode_rewards[e]}; both are zeroed.
 * step_index counts the steps the env has taken since enable (1-based; all envs of a handle step in lockstep), so the
 * reference's global_step at the moment of the record (global_step += 1 per env in e order, :149/:341) is
 *     global_step_at_enable + (step_index - 1) * num_envs + env + 1
 * and `rl/reward` = total_reward, total_rewards.append(total_reward / length) (:150-155, :342-347).
 * grl_episodes_read waits for the stream, returns the finished episodes ordered by (step_index, env) -- the reference's append
 * order -- and empties the list.  *dropped_out = records lost because more than `capacity` episodes finished between two reads. */
typedef struct grl_episode_record {
    int64_t step_index;
    int32_t env;      /* local env index of the handle */
    int32_t length;   /* emulator_steps[e] at the end of the episode */
    double total_reward;
} grl_episode_record;
int grl_episodes_enable(grl_handle *h, int32_t capacity);
int grl_episodes_read(grl_handle *h, grl_episode_record *out, int32_t max_records, int32_t *n_out, int32_t *dropped_out);
/* the running accumulators (host arrays of num_envs entries; either may be NULL) */
int grl_episodes_running(grl_handle *h, double *total_reward_out, int32_t *length_out);

#ifdef __cplusplus
}
#endif
#endif /* GOLDSRL_H */
