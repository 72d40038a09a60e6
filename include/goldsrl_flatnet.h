/* goldsrl_flatnet.h -- C ABI of FlatPolicyVNetwork (GRU + MLP) on the device.
 *
 * Replaces (paths relative to the reference repo root):
 *   fed_gym/agents/paac/policy_v_network.py:194-264   FlatPolicyVNetwork (+ predict())
 *   fed_gym/agents/a3c/estimators.py:5-28             make_cell / true_length / rnn_graph_lstm (GRU trunk)
 *   fed_gym/agents/paac/paac.py:23-38,119-187         choose_next_actions, PAACLearner.train inner loop
 *   fed_gym/agents/paac/actor_learner.py:31-68,91-97  Adam, clip_by_global_norm, reward clip
 * Flat parameter vector, tf.trainable_variables() creation order (D = temporal_size, S0 = static_size,
 * H = rnn hidden = 32, S = static hidden = 32, A = num_actions):
 *   gru_gates_w[D+H,2H] gru_gates_b[2H] gru_cand_w[D+H,H] gru_cand_b[H] temporal_w[H,2H] temporal_b
 *   static1_w[S0,2H] static1_b static2_w[2H,H] static2_b mu1_w[3H,2S] mu1_b mu2_w[2S,S] mu2_b mu3_w[S,A] mu3_b
 *   sig1_w sig1_b sig2_w sig2_b sig3_w sig3_b v1_w[3H,2S] v1_b v2_w[2S,1] v2_b
 * Conventions as in goldsrl.h.
 */
#ifndef GOLDSRL_FLATNET_H
#define GOLDSRL_FLATNET_H

#include "goldsrl.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grl_fnet_config {
    int32_t struct_size;
    int32_t static_size;    /* conf['static_size']   (--static-size, 2; TradeAR1: 1+2n) */
    int32_t temporal_size;  /* conf['temporal_size'] (--temporal-size, 2) */
    int32_t rnn_length;     /* history rows fed to the GRU (--rnn-length, 5); <= 32 */
    int32_t num_actions;    /* 1 for Solow; <= 16 */
    int32_t rnn_hidden;     /* conf['rnn_hidden_size'], must be 32 (--temporal-hidden-size default) */
    int32_t static_hidden;  /* conf['static_hidden_size'], must be 32 */
    int32_t max_samples;    /* largest n of one predict/train call (workspace size) */
    float scale;            /* conf['scale'] (train_paac_solow.py --scale, 100) */
    float clip_norm;        /* 40, 'global'; <= 0: 'ignore' */
    float gamma;            /* 0.99 */
    float mu_bound;         /* ub = -lb = 5 (policy_v_network.py:207-208) */
    float gae_lambda;       /* 1: PAAC's clipped, masked n-step return (paac.py:145,159-177).  In (0,1): the A3C worker's
                             * GAE on the raw rewards (a3c/worker.py:232-294, `_lambda` = 0.96 at :87): delta_t = r_t +
                             * gamma V_{t+1} - V_t, advantage = discounted sum with gamma*lambda, target = advantage + V_t,
                             * bootstrap 0 behind a finished episode (done_penalty, :232) */
} grl_fnet_config;

typedef struct grl_fnet grl_fnet;

int grl_fnet_config_default(grl_fnet_config *cfg);
int grl_fnet_create(grl_handle *h, const grl_fnet_config *cfg, grl_fnet **out);
int grl_fnet_destroy(grl_fnet *net);
const char *grl_fnet_last_error(const grl_fnet *net);
int64_t grl_fnet_num_params(const grl_fnet *net);
int grl_fnet_set_params(grl_fnet *net, const float *host, int64_t n);
int grl_fnet_get_params(grl_fnet *net, float *host, int64_t n);
int grl_fnet_get_grads(grl_fnet *net, float *host, int64_t n);
/* Adam moments + update count (flat-weights checkpoint, as grl_net_get/set_optimizer_state) */
int grl_fnet_get_optimizer_state(grl_fnet *net, float *m_host, float *v_host, int64_t n, int64_t *step_out);
int grl_fnet_set_optimizer_state(grl_fnet *net, const float *m_host, const float *v_host, int64_t n, int64_t step);
/* draw counter of grl_fnet_rollout's action noise (as grl_net_get/set_action_counter) */
int grl_fnet_get_action_counter(grl_fnet *net, uint64_t *out);
int grl_fnet_set_action_counter(grl_fnet *net, uint64_t value);

/* network.predict(states, histories) + the value head, on HOST arrays: states (n,S0), history (n,T,D),
 * outputs mu (n,A) sigma (n,A) vs (n,) (any may be NULL).  Synchronous. */
int grl_fnet_predict(grl_fnet *net, int32_t n, const float *states, const float *history, float *mu, float *sigma, float *vs);
/* Same on the CURRENT observation/history of the (Solow) handle the net was created on: n = num_envs. */
int grl_fnet_predict_env(grl_fnet *net, float *mu, float *sigma, float *vs);
/* One gradient step on HOST samples: loss of policy_v_network.py:228-251 (no entropy term), backward through
 * the length-masked GRU, clip_by_global_norm, Adam.  advantages are ALREADY divided by scale (paac.py:177).
 * stats_host: {loss, policy_loss, critic_loss_mean, global_norm}.  apply_update=0: gradients only. */
int grl_fnet_train(grl_fnet *net, int32_t n, const float *states, const float *history, const float *actions,
                   const float *advantages, const float *critic_target, float lr, int32_t apply_update, float *stats_host);
/* Device-resident PAAC rollout on a Solow or TradeAR1 handle (paac.py:119-172): T x [forward, a = mu + sigma*N(0,1),
 * sigmoid (Solow) / tanh (TradeAR1), env step/auto-reset/observe/history], bootstrap, reward clip +-2, MASKED n-step
 * returns.  TradeAR1 (BASELINE config 5; the reference has no PAAC runner for it) uses the net with
 * static_size = temporal_size = 1+2n, num_actions = n and the worker-style history window (quirk Q11). Async. */
int grl_fnet_rollout(grl_fnet *net, int32_t T);
/* on != 0: the rollouts that follow (persistent-kernel form) also fill the net's training workspace with every layer's
 * activations of their T x num_envs forwards, and grl_fnet_train_rollout / _grads on such a rollout start at the backward pass --
 * what the reference's train step recomputes (policy_v_network.py:246-264 on the batch paac.py:178-186 stacks) is the same
 * parameters on the same inputs.  Bit-identical gradients; the rollout pays the stores (+14 % at rnn 5 and 20; off by default: a rollout nobody
 * trains on should not).  Anything that moves the parameters or overwrites the workspace in between falls back to the recomputation. */
int grl_fnet_set_keep_activations(grl_fnet *net, int32_t on);
/* Gradient step on the last rollout: [all-reduce over ranks if a communicator is attached], clip, Adam. */
int grl_fnet_train_rollout(grl_fnet *net, float lr, float *stats_host);
/* The same in two halves (as grl_net_train_rollout_grads / set_grads / apply_grads in goldsrl_net.h), for callers that sum the
 * gradient over ranks themselves: _grads leaves the LOCAL mean gradient in the net (grl_fnet_get_grads) and updates nothing;
 * grl_fnet_apply_grads: clip_by_global_norm(grad_scale * grads) + Adam(lr); grad_scale = 1/world after a sum over ranks. */
int grl_fnet_train_rollout_grads(grl_fnet *net, float *stats_host);
int grl_fnet_set_grads(grl_fnet *net, const float *host, int64_t n);
int grl_fnet_apply_grads(grl_fnet *net, float lr, float grad_scale, float *stats_host);
/* "actions" (T,E,A) "values" (T,E) "rewards" (T,E) raw "masks" (T,E) "y" (T,E) "adv" (T,E) "boot" (E,)
 * "states" (T,E,S0); Solow: "histories" (T,E,rnn,2); TradeAR1: "nhist" (T,E) int32 rows of the window */
int grl_fnet_read_rollout(grl_fnet *net, const char *which, void *host, size_t bytes);
/* Profiling aid for the persistent rollout kernel: the first call attaches a timestamp buffer (n_out = 0); after the next
 * grl_fnet_rollout a call returns the constant-clock (100 MHz) ticks workgroup 0 took at every barrier of the kernel. */
int grl_fnet_rollout_stage_times(grl_fnet *net, int64_t *out, int32_t max, int32_t *n_out);

/* ---- multi-GPU (BASELINE config 5: 65 536 TradeAR1 envs over 8 GPUs): one process per GPU owning a contiguous env block, one
 * RCCL all-reduce (sum, fp32, ~31k floats: latency-bound) of the flat gradient per rollout; no counterpart in the reference
 * (single device, actor_learner.py:70-75).  The loss is a batch mean (policy_v_network.py:246-251): the summed gradient is scaled
 * by 1/world, clip_by_global_norm is applied after the reduction, Adam runs replicated.  Unique id: grl_comm_unique_id()
 * (goldsrl_net.h), shipped to the ranks by the caller. */
int grl_fnet_comm_init(grl_fnet *net, const void *unique_id, size_t bytes, int32_t rank, int32_t world_size);
int grl_fnet_comm_broadcast_params(grl_fnet *net, int32_t root);
int grl_fnet_comm_destroy(grl_fnet *net);
/* ncclCommCount / ncclCommUserRank of the attached communicator and the all-reduce timing, as grl_net_comm_info (goldsrl_net.h) */
int grl_fnet_comm_info(grl_fnet *net, int32_t *count_out, int32_t *user_rank_out, int64_t *allreduce_calls_out,
                       double *allreduce_ms_total_out, float *allreduce_ms_last_out);

#ifdef __cplusplus
}
#endif
#endif /* GOLDSRL_FLATNET_H */
