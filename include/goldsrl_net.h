/* goldsrl_net.h -- C ABI of the policy/value estimator side of libgoldsrl.so.
 *
 * Replaces (paths relative to the reference repo root):
 *   fed_gym/agents/paac/policy_v_network.py:5-80     ConvSingleAgentPolicyNetwork (+ predict())
 *   fed_gym/agents/paac/networks.py:100-167          placeholder contract (states/actions/advantages/critic_target)
 *   fed_gym/agents/paac/paac.py:408-419              choose_next_actions: a = mu + sigma*N(0,1)
 *   fed_gym/agents/paac/actor_learner.py:31-68,115-119  Adam, clip_by_global_norm, lr anneal
 * Conventions as in goldsrl.h.  A net is bound to the device/stream of the grl_handle it was
 * created on.  Parameters are one flat float32 vector in tf.trainable_variables() creation order:
 *   conv1_w[8,8,3,32] conv1_b conv2_w[4,4,32,64] conv2_b conv3_w[3,3,64,64] conv3_b dense1_w[3136,512]
 *   dense1_b dense2_w[512,256] dense2_b pol1_w[256,512] pol1_b mu_w[512,A] mu_b[A] sigma_w[512,A] sigma_b[A]
 *   v1_w[256,512] v1_b v2_w[512,256] v2_b v3_w[256,1] v3_b          (2 210 213 floats at A = num_actions = 2)
 * Arithmetic: float32 in, float32 out, float32 accumulation, as the reference's TF graph.  conv2, conv3 and the dense layers run
 * on the fp16 matrix pipe with every fp32 operand split into two fp16 terms (22-23 significand bits; csrc/net_gemm.h), which
 * measures closer to a float64 evaluation than a plain float32 one.  Range contract of that form: activations and weights
 * below 65 504 in magnitude (since round 4 the weights of conv2, conv3, pol1 and v1 below 32: the GEMM instances that read them keep one
 * accumulator set, whose third weight plane holds h_w * 2^11; a larger weight takes the same way out as any other violation).  Nothing is clamped: every GEMM checks its output tile, and the first call that synchronises
 * after a violation (predict, train_*, apply_grads) does not return results computed from inf operands.  By default it switches the
 * net to the fp32 form of the same GEMM kernels (v_mfma_f32_16x16x4_f32, no range beyond float32's, an update of the headline configuration takes 0.50 s instead of 0.33: 1.5x, round 4; the net
 * stays there until four updates in a row stayed well inside the range -- grl_net_set_range_return below -- or grl_net_set_gemm_f32(net, 0)) and runs the work again: predict and train_obs in full; a gradient step over a
 * rollout runs again when only its backward pass overflowed, and is given up (GRL_OK, NaN statistics, update_skipped = 1 in
 * grl_net_range_info; parameters and Adam moments untouched) when the rollout's own forward passes did -- the next rollout is valid.
 * GRL_NET_RANGE_FALLBACK=off in the environment keeps the hard failure (GRL_E_RANGE); GRL_NET_GEMM=f32 starts on the fp32 form.
 * Gradient passes pick an exact power-of-two loss scale per call, so advantages / targets of any float32 magnitude are fine.  GRL_NET_LOSS_SCALE=off in the environment disables that scale (diagnostic only).
 */
#ifndef GOLDSRL_NET_H
#define GOLDSRL_NET_H

#include "goldsrl.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GRL_NET_CONV_SINGLE_AGENT 0 /* policy_v_network.py:5-66; Swarm handles only */
/* Evaluate the whole net per agent image, as the reference does, instead of the default shared evaluation (conv1-conv3
 * once per env + exact per-agent corrections, dense1 split into a per-env part and a 5x5 patch part: csrc/net_shared.inc,
 * csrc/net_patch.inc).  Same sums, different association; kept as the A/B reference of the optimisation. */
#define GRL_NET_F_PER_AGENT_TRUNK 0x1
/* Recompute conv3 and the dense stack in the gradient step instead of keeping the rollout's activations resident
 * in HBM (22 KB per agent-sample and step; chosen automatically when that buffer does not fit).  Bit-identical. */
#define GRL_NET_F_RECOMPUTE_FORWARD 0x2
/* Enqueue every chunk on the handle's stream instead of dealing independent chunks round-robin to four streams (the
 * default: memory-bound helper kernels of one chunk overlap the MFMA GEMMs of the others; GRL_NET_LANES=1..8 in the
 * environment overrides the count).  Same results up to the order in which per-lane gradient sums are added. */
#define GRL_NET_F_SINGLE_STREAM 0x4

typedef struct grl_net_config {
    int32_t struct_size;
    int32_t kind;              /* GRL_NET_* */
    int32_t max_chunk_samples; /* agent-samples processed per pass; activations are sized for this (multiple of 10) */
    int32_t reserved;          /* flags: GRL_NET_F_* */
    float scale;               /* conf['scale'] (train_paac_conv.py --scale, 1000) */
    float entropy_beta;        /* conf['entropy_regularisation_strength'] (0.02) */
    float clip_norm;           /* --clip_norm (40), clip_norm_type 'global'; <= 0 means 'ignore' */
    float gamma;               /* --gamma (0.99) */
    int32_t num_actions;       /* conf['num_actions'] (policy_v_network.py:10,40-43): width A of the mu / sigma heads, 1..4; default 2
                                * (SwarmEnvironmentCreator.num_actions).  predict / train on supplied samples work for any A;
                                * grl_net_rollout needs A = 2 because SwarmEnv.step takes (10, 2) actions. */
    int32_t reserved2;
} grl_net_config;

typedef struct grl_net grl_net;

int grl_net_config_default(int32_t kind, grl_net_config *cfg);
int grl_net_create(grl_handle *h, const grl_net_config *cfg, grl_net **out);
int grl_net_destroy(grl_net *net);
const char *grl_net_last_error(const grl_net *net);
int64_t grl_net_num_params(const grl_net *net);
int grl_net_set_params(grl_net *net, const float *host, int64_t n);
int grl_net_get_params(grl_net *net, float *host, int64_t n);
int grl_net_get_grads(grl_net *net, float *host, int64_t n);   /* flat gradient of the last grl_net_train call (before clipping) */
/* Adam moments (flat, parameter order) and the number of updates applied so far: with the parameters this is the whole
 * training state of the estimator (flat-weights checkpoint; the reference's tf.train.Saver path, actor_learner.py:70-89,
 * is disabled in its scripts). */
int grl_net_get_optimizer_state(grl_net *net, float *m_host, float *v_host, int64_t n, int64_t *step_out);
int grl_net_set_optimizer_state(grl_net *net, const float *m_host, const float *v_host, int64_t n, int64_t step);
/* Counter of the action-noise draws of grl_net_rollout (one per rollout step; the generator is keyed by (seed, env id, counter)):
 * part of a checkpoint, so that a resumed run continues the noise stream instead of replaying it from 0. */
int grl_net_get_action_counter(grl_net *net, uint64_t *out);
int grl_net_set_action_counter(grl_net *net, uint64_t value);

/* network.predict(states) (policy_v_network.py:69-80) on the CURRENT observation of the Swarm
 * handle: B = 10*num_envs agent-samples in env-major order.  Outputs are HOST arrays (may be NULL):
 * mu (B,A) sigma (B,A) vs (B,).  Synchronous. */
int grl_net_predict(grl_net *net, float *mu_host, float *sigma_host, float *vs_host);
/* Same on caller-supplied compact observations (host): locust_bins (n_envs,80,2) agent_bins (n_envs,10,2)
 * positions (n_envs,10,2), all uint8 as in grl_out_ptrs. */
int grl_net_predict_obs(grl_net *net, int32_t n_envs, const uint8_t *locust_bins, const uint8_t *agent_bins,
                        const uint8_t *positions, float *mu_host, float *sigma_host, float *vs_host);

/* One PAAC rollout entirely on the device (GridPAACLearner.train inner loop, paac.py:302-372):
 * T x [forward, a = mu + sigma*N(0,1), SwarmRunner norm clip, env step/auto-reset/observe], bootstrap
 * forward, n-step returns/advantages.  Buffers stay on the device for grl_net_train_rollout().
 * reward_layout: 0 = every agent column receives its env's reward ("broadcast"),
 *                1 = the reference's indexing (paac.py:331-338, quirk Q4).
 * Asynchronous: enqueues on the handle's stream; grl_wait(h) joins. */
int grl_net_rollout(grl_net *net, int32_t T, int32_t reward_layout);
/* Gradient step on the last rollout's T*B samples: loss (policy_v_network.py:45-66), backward,
 * [all-reduce over ranks if a communicator is attached], clip_by_global_norm, Adam(lr).
 * stats_host (may be NULL) receives {loss, policy_loss, critic_loss_mean, global_norm}. Synchronous. */
int grl_net_train_rollout(grl_net *net, float lr, float *stats_host);
/* The same in two halves, for callers that exchange or post-process gradients themselves (bench.py's fallback when no
 * RCCL communicator can be formed sums them over ranks on the host, goldsrl/distributed.py):
 * _grads: loss + backward over the last rollout only -- the LOCAL mean gradient is left in the net (grl_net_get_grads), stats
 *         are the local loss terms and the local gradient norm, nothing is updated;
 * grl_net_set_grads: upload a flat gradient;  grl_net_apply_grads: clip_by_global_norm(grad_scale * grads) + Adam(lr). */
int grl_net_train_rollout_grads(grl_net *net, float *stats_host);
int grl_net_set_grads(grl_net *net, const float *host, int64_t n);
int grl_net_apply_grads(grl_net *net, float lr, float grad_scale, float *stats_host);
/* Gradient step on caller-supplied samples (tests; network.loss feed of paac.py:374-387):
 * compact observations for n_envs envs (n = 10*n_envs samples), actions (n,A), advantages (n,)
 * ALREADY divided by scale, critic_target (n,).  apply_update=0 only computes gradients. */
int grl_net_train_obs(grl_net *net, int32_t n_envs, const uint8_t *locust_bins, const uint8_t *agent_bins,
                      const uint8_t *positions, const float *actions, const float *advantages,
                      const float *critic_target, float lr, int32_t apply_update, float *stats_host);
/* Copy one rollout buffer to the host: "actions" (T,B,2) raw sampled actions, "values" (T,B),
 * "rewards" (T,B), "y" (T,B), "adv" (T,B), "boot" (B,). */
int grl_net_read_rollout(grl_net *net, const char *which, void *host, size_t bytes);
/* Debug/test access to a forward activation of the last chunk: "a1" (n,20,20,32) "a2" (n,9,9,64)
 * "a3" (n,7,7,64) "d1" (n,512) "d2" (n,256) "p1" (n,512) "v1" (n,512) "v2" (n,256); "a1" only with
 * GRL_NET_F_PER_AGENT_TRUNK, otherwise the per-env "a1sh"/"sraw" (n/10,20,20,32).  In the default shared evaluation the
 * per-agent "a2"/"a3" are not materialised by the forward pass; they are expanded on demand by this call. */
int grl_net_read_activation(grl_net *net, const char *which, float *host, size_t bytes);

/* ---- multi-GPU: one process per GPU, one RCCL all-reduce (sum, fp32) of the flat gradient per
 * rollout over xGMI (no counterpart in the reference, which is single-device: actor_learner.py:70-75).
 * Rank 0 calls grl_comm_unique_id and ships the bytes to the other ranks by any means (goldsrl/distributed.py uses its own
 * TCP store on MASTER_ADDR); every rank then calls grl_net_comm_init.  Afterwards grl_net_train_* all-reduces gradients before
 * clip+Adam, so parameters stay replicated; the GEMMs' range flag is max-reduced in the same group, so a pass that left the fp16
 * range is repeated on the fp32 form, given up, or (GRL_NET_RANGE_FALLBACK=off) failed on EVERY rank alike, and no replica is updated
 * from it. */
size_t grl_comm_unique_id_bytes(void);
int grl_comm_unique_id(void *out, size_t bytes);
int grl_net_comm_init(grl_net *net, const void *unique_id, size_t bytes, int32_t rank, int32_t world_size);
int grl_net_comm_broadcast_params(grl_net *net, int32_t root);
int grl_net_comm_destroy(grl_net *net);
/* Arithmetic form of the GEMMs (see the header comment): *gemm_f32_out = 1 when the net computes on the fp32 form, *fallbacks_out = how
 * often a range violation switched it there, *update_skipped_out = 1 when the LAST train_rollout* call gave its update up.  Any pointer
 * may be NULL. */
int grl_net_range_info(grl_net *net, int32_t *gemm_f32_out, int32_t *fallbacks_out, int32_t *update_skipped_out);
/* 1: compute every GEMM on the fp32 form (and stay there); 0: back to the three-product fp16 form. */
int grl_net_set_gemm_f32(grl_net *net, int32_t on);
/* The way back from a range fallback (round 5).  The reference's float32 graph (policy_v_network.py:14-59) has no operand range,
 * so an activation spike costs it nothing; here it used to cost 1.5x for the rest of the run.  While a net computes on the fp32 form
 * BECAUSE of a violation, every GEMM tile records the largest |value| it hands on; at the end of every applied update
 * (grl_net_train_rollout, grl_net_train_obs with apply_update, grl_net_apply_grads) that maximum -- max-reduced over the ranks of an
 * attached communicator, like the flag -- is compared with 65 504 / 4, and after `clean_passes` such updates in a row the net returns
 * to the fp16 form (default 4; GRL_NET_RANGE_RETURN=n|off in the environment; 0 = never, the behaviour until round 4).  A form
 * chosen with grl_net_set_gemm_f32 or GRL_NET_GEMM=f32 is never left on its own.
 * grl_net_range_return_info: *returns_out = how often the net went back, *clean_passes_out = clean updates counted so far,
 * *needed_out = how many it takes, *absmax_last_out = the maximum the last decision saw.  Any pointer may be NULL. */
int grl_net_range_return_info(grl_net *net, int32_t *returns_out, int32_t *clean_passes_out, int32_t *needed_out, float *absmax_last_out);
int grl_net_set_range_return(grl_net *net, int32_t clean_passes);

/* What RCCL itself says about the attached communicator: ncclCommCount / ncclCommUserRank (0 / -1 without one), and the
 * gradient all-reduces so far: calls, summed and last duration in ms (HIP events around the collective on the handle's stream).
 * Any out pointer may be NULL. */
int grl_net_comm_info(grl_net *net, int32_t *count_out, int32_t *user_rank_out, int64_t *allreduce_calls_out,
                      double *allreduce_ms_total_out, float *allreduce_ms_last_out);

/* Host-side cost of the actor loop (the reference's learner thread spends its time in Python between session.run calls,
 * paac.py:302-387; here the host only enqueues): wall-clock ms this thread spent inside grl_net_rollout (asynchronous: all of it is
 * enqueue work) and inside grl_net_train_rollout* split at its first blocking call -- `train_enqueue` up to the stream
 * synchronisation that ends the update, `train_wait` inside it -- summed over `updates_out` gradient steps since the net was created.
 * With one rank per GPU and eight ranks per host this is the number that says whether a rank is bound by its host thread
 * (enqueue ~ the whole update) or by its GPU (enqueue << update).  When the launch queues fill up, enqueue calls block on the
 * device and the figure is an upper bound of the host's own work.  Any out pointer may be NULL. */
int grl_net_host_times(grl_net *net, int64_t *rollouts_out, int64_t *updates_out, double *rollout_enqueue_ms_out,
                       double *train_enqueue_ms_out, double *train_wait_ms_out);

/* Per-kernel timing of the GEMM kernels for bench.py's roofline (HIP events around every launch
 * of gemm_rowk / gemm_tn while enabled): returns launches, summed ms and summed FLOPs. */
int grl_net_profile_enable(grl_net *net, int32_t on);
int grl_net_profile_read(grl_net *net, int32_t *launches_out, float *total_ms_out, double *flops_out);
/* The same per GEMM family (arrays of ntags >= 15 entries): 0 other, 1-3 small dense layers forward / data gradient / weight
 * gradient, 4-6 dense1 patch GEMMs forward / data / weight, 7-9 once-per-env GEMMs forward / data / weight, 10-12 slot GEMMs
 * (conv3 products / data / weight), 13 conv2 class corrections, 14 per-agent evaluation mode. */
int grl_net_profile_read_tags(grl_net *net, int32_t ntags, int32_t *launches, float *ms, double *flops);

#ifdef __cplusplus
}
#endif
#endif /* GOLDSRL_NET_H */
