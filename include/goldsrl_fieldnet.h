/* goldsrl_fieldnet.h -- C ABI of ConvPolicyVFieldNetwork on the device.
 *
 * Replaces (paths relative to the reference repo root):
 *   fed_gym/agents/paac/policy_v_network.py:83-191   ConvPolicyVFieldNetwork (+ predict())
 *   fed_gym/agents/paac/networks.py:170-190          ConvFieldNetwork placeholders (states, agent_positions, actions, ...)
 *   fed_gym/agents/paac/actor_learner.py:31-68       Adam, clip_by_global_norm
 * No script of the reference builds this net (only tests/estimators_tests.py:152-215, a shape test at 32x32x3 with 5 filters,
 * 2 conv layers and 3 actions); it is the last estimator of agents/paac.  `use_rnn` is False in the reference (:88), so the
 * history placeholder is never consumed and has no counterpart here.
 *
 * Graph: conv_layers x [Conv2D(filters, 3x3, 'same', relu) -> MaxPool 2x2/2] -> flatten (h, w, f) -> Dense 64 relu -> Dense 32
 * relu = processed_state; policy: Dense 64 relu -> Dense 2*H*W*A relu -> mus = Dense H*W*A tanh, sigmas = Dense H*W*A sigmoid,
 * both reshaped (N,H,W,A) and gathered at the agent's (height_idx, width_idx) -- so only A columns of either head are
 * evaluated per sample, with identical results; value: Dense 64 relu -> Dense 32 relu -> -scale * softplus(Dense 1).
 * Loss as ConvSingleAgentPolicyNetwork (:154-173).
 *
 * Flat parameter vector, tf.trainable_variables() creation order (F = filters, D0 = (H/2^L)*(W/2^L)*F, HWA = H*W*A):
 *   conv0_w[3,3,C,F] conv0_b ... conv{L-1}_w[3,3,F,F] conv{L-1}_b dense1_w[D0,64] dense1_b dense2_w[64,32] dense2_b
 *   pol1_w[32,64] pol1_b pol2_w[64,2*HWA] pol2_b mu_w[2*HWA,HWA] mu_b sigma_w[2*HWA,HWA] sigma_b
 *   v1_w[32,64] v1_b v2_w[64,32] v2_b v3_w[32,1] v3_b
 * Conventions as in goldsrl.h.  All sums run in a fixed order (bitwise reproducible).
 */
#ifndef GOLDSRL_FIELDNET_H
#define GOLDSRL_FIELDNET_H

#include "goldsrl.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grl_fieldnet_config {
    int32_t struct_size;
    int32_t height, width;      /* conf['height'], conf['width']: multiples of 2^conv_layers, <= 128 */
    int32_t channels;           /* conf['channels'], 1..8 */
    int32_t filters;            /* conf['filters'], 1..32 */
    int32_t conv_layers;        /* conf['conv_layers'], 1..3 */
    int32_t num_actions;        /* conf['num_actions'], 1..4 */
    int32_t max_samples;        /* largest n of one predict / train call */
    float scale;                /* conf['scale'] */
    float entropy_beta;         /* conf['entropy_regularisation_strength'] */
    float clip_norm;            /* 'global' clip; <= 0: 'ignore' */
} grl_fieldnet_config;

typedef struct grl_fieldnet grl_fieldnet;

int grl_fieldnet_config_default(grl_fieldnet_config *cfg);      /* the reference test's geometry: 32x32x3, 5 filters, 2 layers, 3 actions */
/* the net lives on the device / stream of `h` (any env kind) */
int grl_fieldnet_create(grl_handle *h, const grl_fieldnet_config *cfg, grl_fieldnet **out);
int grl_fieldnet_destroy(grl_fieldnet *net);
const char *grl_fieldnet_last_error(const grl_fieldnet *net);
int64_t grl_fieldnet_num_params(const grl_fieldnet *net);
int grl_fieldnet_set_params(grl_fieldnet *net, const float *host, int64_t n);
int grl_fieldnet_get_params(grl_fieldnet *net, float *host, int64_t n);
int grl_fieldnet_get_grads(grl_fieldnet *net, float *host, int64_t n);

/* network.predict(states, histories, positions) (policy_v_network.py:175-191) on HOST arrays: states (n,H,W,C) float32,
 * positions (n,2) int32 [(height_idx, width_idx)], outputs mu (n,A) sigma (n,A) vs (n,) (any may be NULL).  Synchronous. */
int grl_fieldnet_predict(grl_fieldnet *net, int32_t n, const float *states, const int32_t *positions, float *mu, float *sigma, float *vs);
/* One gradient step on HOST samples: loss, backward, clip_by_global_norm, Adam(lr).  advantages as fed to the placeholder.
 * stats_host: {loss, policy_loss, critic_loss_mean, global_norm}.  apply_update = 0: gradients only. */
int grl_fieldnet_train(grl_fieldnet *net, int32_t n, const float *states, const int32_t *positions, const float *actions,
                    const float *advantages, const float *critic_target, float lr, int32_t apply_update, float *stats_host);

#ifdef __cplusplus
}
#endif
#endif /* GOLDSRL_FIELDNET_H */
