// ConvPolicyVFieldNetwork on gfx950 (reference fed_gym/agents/paac/policy_v_network.py:83-191; placeholders networks.py:170-190):
// the field-output estimator of agents/paac -- 3x3 'same' convolutions with 2x2 max-pools, a small dense stack, and mu / sigma
// heads that are Dense(H*W*A) layers gathered at the agent's grid position.  No script of the reference builds it (only a shape
// test, tests/estimators_tests.py:152-215), so it is far off the throughput path: the kernels below are plain fp32, one lane per
// output element (every layer is tiny at the reference's geometry -- 32x32x3 images, 5 filters, 64/32 hidden units -- and the
// Dense(H*W*A) heads, 18.9 M weights each, are touched in A columns per sample only), with every sum in a fixed order.
//   forward:  field_conv_pool x L -> field_dense x 4 (dense1, dense2, pol1, pol2) + x 2 (v1, v2) -> field_heads_fwd
//   backward: field_heads_bwd (loss terms, dZ of the heads, dP2, dV2) -> field_heads_wgrad -> field_dense_{wgrad,dgrad} ->
//             field_pool_bwd / field_conv_wgrad / field_conv_dgrad x L -> sumsq / clip / Adam
#include <string.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/goldsrl_fieldnet.h"
#include "common.h"

namespace grl {

constexpr int FIELD_FC = 32;          // self.fc_hidden (policy_v_network.py:86)
constexpr int FIELD_MAX_LAYERS = 3;
constexpr int FIELD_MAX_A = 4;
constexpr float FIELD_LOG_2PI = 1.8378770664093453f;

struct FieldOff {
    long cw[FIELD_MAX_LAYERS], cb[FIELD_MAX_LAYERS];
    long d1w, d1b, d2w, d2b, p1w, p1b, p2w, p2b, muw, mub, sgw, sgb, v1w, v1b, v2w, v2b, v3w, v3b, total;
};

// relu(conv3x3 'same') then 2x2 max-pool: one lane per pooled output (n, y2, x2, f); idx = position of the FIRST maximum in the
// window (dy*2 + dx), the one the pooling gradient goes to
__global__ void field_conv_pool_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b, long total,
                                       int H, int W, int Cin, int F, float *__restrict__ pooled, uint8_t *__restrict__ idx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % F);
    long r = i / F;
    const int x2 = (int)(r % (W / 2)); r /= (W / 2);
    const int y2 = (int)(r % (H / 2));
    const long n = r / (H / 2);
    float best = 0.f;
    int bi = 0;
    for (int d = 0; d < 4; ++d) {
        const int yy = 2 * y2 + (d >> 1), xx = 2 * x2 + (d & 1);
        float acc = b[f];
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yy + ky - 1;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = xx + kx - 1;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float *xp = x + ((n * H + iy) * W + ix) * Cin;
                const float *wp = w + (long)((ky * 3 + kx) * Cin) * F + f;
                for (int ci = 0; ci < Cin; ++ci) acc += xp[ci] * wp[(long)ci * F];
            }
        }
        const float v = fmaxf(acc, 0.f);
        if (d == 0 || v > best) { best = v; bi = d; }
    }
    pooled[i] = best;
    idx[i] = (uint8_t)bi;
}

// out[n][j] = act(sum_k in[n][k] W[k][j] + b[j]); act: 0 none, 1 relu
__global__ void field_dense_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ b, long N, int K, int J,
                                   int act, float *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * J) return;
    const long n = i / J;
    const int j = (int)(i - n * J);
    float acc = b[j];
    const float *ip = in + n * K;
    for (int k = 0; k < K; ++k) acc += ip[k] * w[(long)k * J + j];
    out[i] = act ? fmaxf(acc, 0.f) : acc;
}

__device__ __forceinline__ float field_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// mu = tanh(p2 . mu_w[:, col..col+A) + b), sigma = sigmoid(...), col = (pos_y * W + pos_x) * A -- tf.gather_nd of the (N,H,W,A)
// fields (policy_v_network.py:140-152); vs = -scale * softplus(v2 . v3_w + b).  One wave per sample.
template <int A>
__global__ __launch_bounds__(256) void field_heads_fwd_kernel(const float *__restrict__ p2, const float *__restrict__ v2, const int32_t *__restrict__ pos,
                                                              const float *__restrict__ P, FieldOff o, int N, int P2, int HWA, int W, float scale,
                                                              float *__restrict__ mu, float *__restrict__ sigma, float *__restrict__ vs) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const long col = ((long)pos[n * 2] * W + pos[n * 2 + 1]) * A;
    const float *muw = P + o.muw + col, *sgw = P + o.sgw + col;
    float m[A], s[A], zv = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) { m[a] = 0.f; s[a] = 0.f; }
    for (int k = lane; k < P2; k += 64) {
        const float xv = p2[(long)n * P2 + k];
#pragma unroll
        for (int a = 0; a < A; ++a) { m[a] += xv * muw[(long)k * HWA + a]; s[a] += xv * sgw[(long)k * HWA + a]; }
    }
    if (lane < FIELD_FC) zv = v2[(long)n * FIELD_FC + lane] * P[o.v3w + lane];
#pragma unroll
    for (int a = 0; a < A; ++a) { m[a] = field_wave_sum(m[a]); s[a] = field_wave_sum(s[a]); }
    zv = field_wave_sum(zv);
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            mu[(long)n * A + a] = tanhf(m[a] + P[o.mub + col + a]);
            sigma[(long)n * A + a] = 1.0f / (1.0f + expf(-(s[a] + P[o.sgb + col + a])));
        }
        zv += P[o.v3b];
        vs[n] = -scale * (zv > 20.f ? zv : log1pf(expf(zv)));
    }
}

// loss terms and dZ of the heads per sample (dzh row: [dz_mu(A) | dz_sigma(A) | dz_v | policy term | critic term]), then
// dp2 = relu'(p2) . (mu_w[:, cols] dz_mu + sigma_w[:, cols] dz_sigma) and dv2 = relu'(v2) . v3_w dz_v.  One wave per sample.
template <int A>
__global__ __launch_bounds__(256) void field_heads_bwd_kernel(const float *__restrict__ p2, const float *__restrict__ v2, const int32_t *__restrict__ pos,
                                                              const float *__restrict__ mu, const float *__restrict__ sigma, const float *__restrict__ vs,
                                                              const float *__restrict__ actions, const float *__restrict__ adv,
                                                              const float *__restrict__ y, const float *__restrict__ P, FieldOff o, int N, int P2,
                                                              int HWA, int W, float scale, float beta, float inv_n, float *__restrict__ dzh,
                                                              float *__restrict__ dp2, float *__restrict__ dv2) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float Adv = adv[n], v = vs[n], tgt = y[n];
    const float dlogp = -Adv * inv_n, dent = -beta * inv_n;
    float zm[A], zs[A], logp = 0.f, ent = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const float m = mu[(long)n * A + a], s = sigma[(long)n * A + a], d = actions[(long)n * A + a] - m;
        logp += -0.5f * (d / s) * (d / s) - logf(s) - 0.5f * FIELD_LOG_2PI;      // Normal.log_prob, summed over the actions (:158-160)
        ent += 0.5f + 0.5f * FIELD_LOG_2PI + logf(s);
        const float dmu = dlogp * d / (s * s);
        const float dsg = dlogp * (d * d / (s * s * s) - 1.0f / s) + dent / s;
        zm[a] = dmu * (1.0f - m * m);
        zs[a] = dsg * s * (1.0f - s);
    }
    const float dvs = 0.5f * (v - tgt) / scale * inv_n;
    const float zv = dvs * (-scale) * (-expm1f(v / scale));      // d(-scale softplus(z))/dz = -scale sigmoid(z), sigmoid(z) = -expm1(vs/scale)
    const int dz = 2 * A + 3;
    if (lane == 0) {
        float *q = dzh + (long)n * dz;
#pragma unroll
        for (int a = 0; a < A; ++a) { q[a] = zm[a]; q[A + a] = zs[a]; }
        q[2 * A] = zv;
        q[2 * A + 1] = -(logp * Adv + beta * ent);
        q[2 * A + 2] = 0.25f * (v - tgt) * (v - tgt) / scale;
    }
    const long col = ((long)pos[n * 2] * W + pos[n * 2 + 1]) * A;
    const float *muw = P + o.muw + col, *sgw = P + o.sgw + col;
    for (int k = lane; k < P2; k += 64) {
        float g = 0.f;
#pragma unroll
        for (int a = 0; a < A; ++a) g += zm[a] * muw[(long)k * HWA + a];
#pragma unroll
        for (int a = 0; a < A; ++a) g += zs[a] * sgw[(long)k * HWA + a];
        dp2[(long)n * P2 + k] = p2[(long)n * P2 + k] > 0.f ? g : 0.f;
    }
    if (lane < FIELD_FC) dv2[(long)n * FIELD_FC + lane] = v2[(long)n * FIELD_FC + lane] > 0.f ? zv * P[o.v3w + lane] : 0.f;
}

// gradients of the gathered heads into the (zeroed) flat gradient: lane k owns row k of mu_w / sigma_w and walks the samples in
// order (two agents on one pixel add into the same columns); lanes < A walk the biases; v3 and the loss sums likewise
template <int A>
__global__ void field_heads_wgrad_kernel(const float *__restrict__ p2, const float *__restrict__ v2, const int32_t *__restrict__ pos,
                                         const float *__restrict__ dzh, FieldOff o, int N, int P2, int HWA, int W, float *__restrict__ G,
                                         double *__restrict__ stats64) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int dz = 2 * A + 3;
    if (k < P2) {
        for (int n = 0; n < N; ++n) {
            const long col = ((long)pos[n * 2] * W + pos[n * 2 + 1]) * A;
            const float pv = p2[(long)n * P2 + k];
            const float *q = dzh + (long)n * dz;
#pragma unroll
            for (int a = 0; a < A; ++a) {
                G[o.muw + (long)k * HWA + col + a] += pv * q[a];
                G[o.sgw + (long)k * HWA + col + a] += pv * q[A + a];
            }
        }
    }
    if (k < A) {
        for (int n = 0; n < N; ++n) {
            const long col = ((long)pos[n * 2] * W + pos[n * 2 + 1]) * A;
            G[o.mub + col + k] += dzh[(long)n * dz + k];
            G[o.sgb + col + k] += dzh[(long)n * dz + A + k];
        }
    }
    if (k < FIELD_FC) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += v2[(long)n * FIELD_FC + k] * dzh[(long)n * dz + 2 * A];
        G[o.v3w + k] = s;
    }
    if (k == FIELD_FC) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dzh[(long)n * dz + 2 * A];
        G[o.v3b] = s;
    }
    if (k == FIELD_FC + 1 || k == FIELD_FC + 2) {
        double s = 0.0;
        for (int n = 0; n < N; ++n) s += (double)dzh[(long)n * dz + 2 * A + (k - FIELD_FC)];
        stats64[k - FIELD_FC - 1] = s;
    }
}

// dW[k][j] = sum_n in[n][k] dout[n][j] (samples in order); lanes with k == K write db[j] = sum_n dout[n][j]
__global__ void field_dense_wgrad_kernel(const float *__restrict__ in, const float *__restrict__ dout, long N, int K, int J, float *__restrict__ gw,
                                         float *__restrict__ gb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(K + 1) * J) return;
    const int k = (int)(i / J), j = (int)(i - (long)k * J);
    float s = 0.f;
    if (k < K) {
        for (long n = 0; n < N; ++n) s += in[n * K + k] * dout[n * J + j];
        gw[i] = s;
    } else {
        for (long n = 0; n < N; ++n) s += dout[n * J + j];
        gb[j] = s;
    }
}

// din[n][k] (+)= relu'(fwd[n][k]) . sum_j dout[n][j] W[k][j]      (fwd = the forward activation of the input layer, or nullptr)
__global__ void field_dense_dgrad_kernel(const float *__restrict__ dout, const float *__restrict__ w, const float *__restrict__ fwd, long N, int K,
                                         int J, int accumulate, float *__restrict__ din) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    const long n = i / K;
    const int k = (int)(i - n * K);
    float s = 0.f;
    const float *dp = dout + n * J, *wp = w + (long)k * J;
    for (int j = 0; j < J; ++j) s += dp[j] * wp[j];
    if (fwd && !(fwd[i] > 0.f)) s = 0.f;
    din[i] = accumulate ? din[i] + s : s;
}

__global__ void field_relu_mask_kernel(const float *__restrict__ fwd, long n, float *__restrict__ d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(fwd[i] > 0.f)) d[i] = 0.f;
}

// dz[n][y][x][f] = dpooled[n][y/2][x/2][f] where (y, x) is the window's argmax and the pooled value is positive (ReLU), else 0
__global__ void field_pool_bwd_kernel(const float *__restrict__ dpooled, const float *__restrict__ pooled, const uint8_t *__restrict__ idx, long total,
                                      int H, int W, int F, float *__restrict__ dz) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % F);
    long r = i / F;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const long n = r / H;
    const long pi = ((n * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)) * F + f;
    const int d = (y & 1) * 2 + (x & 1);
    dz[i] = (idx[pi] == d && pooled[pi] > 0.f) ? dpooled[pi] : 0.f;
}

// per-sample partial weight gradient of a 3x3 'same' conv: slab[n][t], t = ((ky*3+kx)*Cin+ci)*F+f, then F bias entries
__global__ void field_conv_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dz, int H, int W, int Cin, int F,
                                        float *__restrict__ slab) {
    const long n = blockIdx.x;
    const int T = 9 * Cin * F + F;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float s = 0.f;
        if (t < 9 * Cin * F) {
            const int f = t % F, ci = (t / F) % Cin, tap = t / (F * Cin), ky = tap / 3, kx = tap - ky * 3;
            for (int yy = 0; yy < H; ++yy) {
                const int iy = yy + ky - 1;
                if ((unsigned)iy >= (unsigned)H) continue;
                for (int xx = 0; xx < W; ++xx) {
                    const int ix = xx + kx - 1;
                    if ((unsigned)ix >= (unsigned)W) continue;
                    s += x[((n * H + iy) * W + ix) * Cin + ci] * dz[((n * H + yy) * W + xx) * F + f];
                }
            }
        } else {
            const int f = t - 9 * Cin * F;
            for (int p = 0; p < H * W; ++p) s += dz[(n * H * W + p) * F + f];
        }
        slab[n * T + t] = s;
    }
}

__global__ void field_slab_reduce_kernel(const float *__restrict__ slab, long N, int T, int split, float *__restrict__ gw, float *__restrict__ gb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float s = 0.f;
    for (long n = 0; n < N; ++n) s += slab[n * T + t];
    if (t < split) gw[t] = s; else gb[t - split] = s;
}

// dx[n][y][x][ci] = sum_{ky,kx,f} dz[n][y-ky+1][x-kx+1][f] w[ky][kx][ci][f]
__global__ void field_conv_dgrad_kernel(const float *__restrict__ dz, const float *__restrict__ w, long total, int H, int W, int Cin, int F,
                                        float *__restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % Cin);
    long r = i / Cin;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const long n = r / H;
    float s = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int oy = y - ky + 1;
        if ((unsigned)oy >= (unsigned)H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ox = x - kx + 1;
            if ((unsigned)ox >= (unsigned)W) continue;
            const float *dp = dz + ((n * H + oy) * W + ox) * F;
            const float *wp = w + (long)((ky * 3 + kx) * Cin + ci) * F;
            for (int f = 0; f < F; ++f) s += dp[f] * wp[f];
        }
    }
    dx[i] = s;
}

// sum of squares in a fixed order: 256 strided partial sums per workgroup, then one workgroup over the partials
__global__ __launch_bounds__(256) void field_sumsq_kernel(const float *__restrict__ g, long n, double *__restrict__ part) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void field_finalize_kernel(const double *__restrict__ part, int nparts, const double *__restrict__ stats64, float inv_n, float clip_norm,
                                      float *__restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += part[i];
    const float norm = (float)sqrt(s);
    const float pl = (float)(stats64[0] * (double)inv_n), cl = (float)(stats64[1] * (double)inv_n);
    stats[0] = pl; stats[1] = cl; stats[2] = pl + cl; stats[3] = norm;
    stats[4] = clip_norm > 0.f ? clip_norm / fmaxf(norm, clip_norm) : 1.0f;      // tf.clip_by_global_norm (actor_learner.py:52-57)
}

__global__ void field_adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
                                  const float *__restrict__ stats, float lr_t) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * stats[4];
    const float mi = 0.9f * m[i] + 0.1f * gi;
    const float vi = 0.999f * v[i] + 0.001f * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + 1e-8f);
}

}  // namespace grl

struct grl_fieldnet {
    grl_handle *h;
    grl_fieldnet_config cfg;
    std::string err;
    grl::FieldOff off;
    int L, F, A, P2, HWA, D0;
    int lh[grl::FIELD_MAX_LAYERS + 1], lw[grl::FIELD_MAX_LAYERS + 1], lc[grl::FIELD_MAX_LAYERS + 1];      // input geometry of layer l (l = L: the flattened map)
    float *params, *grads, *adam_m, *adam_v;
    long adam_t;
    float *x[grl::FIELD_MAX_LAYERS + 1];          // x[0] = states, x[l+1] = pooled output of layer l
    uint8_t *idx[grl::FIELD_MAX_LAYERS];
    float *dxl[grl::FIELD_MAX_LAYERS + 1];        // gradient of x[l] (l >= 1)
    float *dzc;                                   // dense conv-output gradient of the layer at hand
    float *cslab;
    float *d1, *d2, *p1, *p2, *v1, *v2, *mu, *sigma, *vs;
    float *dd1, *dd2, *dp1, *dp2, *dv1, *dv2, *dzh;
    float *d_act, *d_adv, *d_y, *stats;
    int32_t *d_pos;
    double *stats64;
    std::vector<void *> allocs;
};

namespace grl {

static int xfail(grl_fieldnet *n, int code, const std::string &msg) {
    if (n) n->err = msg;
    return code;
}
#define FLD_HIP(n, call)                                                                                   \
    do {                                                                                                   \
        hipError_t _e = (call);                                                                            \
        if (_e != hipSuccess) return xfail(n, GRL_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T>
static int xalloc(grl_fieldnet *n, T **p, size_t count) {
    FLD_HIP(n, hipMalloc((void **)p, count * sizeof(T)));
    n->allocs.push_back(*p);
    FLD_HIP(n, hipMemsetAsync(*p, 0, count * sizeof(T), n->h->stream));
    return GRL_OK;
}

static inline unsigned nb(long total) { return (unsigned)((total + 255) / 256); }

#define FIELD_DISPATCH(A_, CALL)                       \
    switch (A_) {                                      \
        case 1: { constexpr int kA = 1; CALL; } break; \
        case 2: { constexpr int kA = 2; CALL; } break; \
        case 3: { constexpr int kA = 3; CALL; } break; \
        default: { constexpr int kA = 4; CALL; } break; \
    }

static int field_forward(grl_fieldnet *net, int n) {
    hipStream_t st = net->h->stream;
    const float *P = net->params;
    const FieldOff &o = net->off;
    for (int l = 0; l < net->L; ++l) {
        const long total = (long)n * (net->lh[l] / 2) * (net->lw[l] / 2) * net->F;
        hipLaunchKernelGGL(field_conv_pool_kernel, dim3(nb(total)), dim3(256), 0, st, net->x[l], P + o.cw[l], P + o.cb[l], total, net->lh[l],
                           net->lw[l], net->lc[l], net->F, net->x[l + 1], net->idx[l]);
    }
    auto dense = [&](const float *in, long w, long b, int K, int J, float *out) {
        hipLaunchKernelGGL(field_dense_kernel, dim3(nb((long)n * J)), dim3(256), 0, st, in, P + w, P + b, (long)n, K, J, 1, out);
    };
    dense(net->x[net->L], o.d1w, o.d1b, net->D0, 2 * FIELD_FC, net->d1);
    dense(net->d1, o.d2w, o.d2b, 2 * FIELD_FC, FIELD_FC, net->d2);
    dense(net->d2, o.p1w, o.p1b, FIELD_FC, 2 * FIELD_FC, net->p1);
    dense(net->p1, o.p2w, o.p2b, 2 * FIELD_FC, net->P2, net->p2);
    dense(net->d2, o.v1w, o.v1b, FIELD_FC, 2 * FIELD_FC, net->v1);
    dense(net->v1, o.v2w, o.v2b, 2 * FIELD_FC, FIELD_FC, net->v2);
    FIELD_DISPATCH(net->A, hipLaunchKernelGGL(field_heads_fwd_kernel<kA>, dim3((n + 3) / 4), dim3(256), 0, st, net->p2, net->v2, net->d_pos, P, o, n,
                                              net->P2, net->HWA, net->cfg.width, net->cfg.scale, net->mu, net->sigma, net->vs));
    FLD_HIP(net, hipGetLastError());
    return GRL_OK;
}

static int field_upload(grl_fieldnet *net, int n, const float *states, const int32_t *positions) {
    const grl_fieldnet_config &c = net->cfg;
    for (int i = 0; i < n; ++i)
        if (positions[2 * i] < 0 || positions[2 * i] >= c.height || positions[2 * i + 1] < 0 || positions[2 * i + 1] >= c.width)
            return xfail(net, GRL_E_INVALID, "agent position outside the " + std::to_string(c.height) + "x" + std::to_string(c.width) + " field (tf.gather_nd would raise)");
    hipStream_t st = net->h->stream;
    FLD_HIP(net, hipMemcpyAsync(net->x[0], states, (size_t)n * c.height * c.width * c.channels * 4, hipMemcpyHostToDevice, st));
    FLD_HIP(net, hipMemcpyAsync(net->d_pos, positions, (size_t)n * 8, hipMemcpyHostToDevice, st));
    return GRL_OK;
}

}  // namespace grl

using namespace grl;

extern "C" {

int grl_fieldnet_config_default(grl_fieldnet_config *cfg) {
    if (!cfg) return GRL_E_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(grl_fieldnet_config);
    cfg->height = 32; cfg->width = 32; cfg->channels = 3; cfg->filters = 5; cfg->conv_layers = 2; cfg->num_actions = 3;      // tests/estimators_tests.py:157-176
    cfg->max_samples = 256;
    cfg->scale = 1.f; cfg->entropy_beta = 0.f; cfg->clip_norm = 40.f;
    return GRL_OK;
}

int grl_fieldnet_create(grl_handle *h, const grl_fieldnet_config *cfg, grl_fieldnet **out) {
    if (!h || !cfg || !out) return GRL_E_INVALID;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(grl_fieldnet_config)) return fail(h, GRL_E_INVALID, "grl_fieldnet_create: config size mismatch");
    const int L = cfg->conv_layers;
    if (L < 1 || L > FIELD_MAX_LAYERS || cfg->height < 2 || cfg->width < 2 || cfg->height > 128 || cfg->width > 128 || cfg->height % (1 << L) ||
        cfg->width % (1 << L) || cfg->channels < 1 || cfg->channels > 8 || cfg->filters < 1 || cfg->filters > 32 || cfg->num_actions < 1 ||
        cfg->num_actions > FIELD_MAX_A || cfg->max_samples < 1)
        return fail(h, GRL_E_INVALID, "grl_fieldnet_create: size out of range (height/width <= 128 and multiples of 2^conv_layers, conv_layers 1..3, "
                                      "channels 1..8, filters 1..32, num_actions 1..4)");
    hipSetDevice(h->cfg.device_id);
    grl_fieldnet *n = new grl_fieldnet();
    n->h = h; n->cfg = *cfg; n->adam_t = 0;
    n->L = L; n->F = cfg->filters; n->A = cfg->num_actions;
    n->HWA = cfg->height * cfg->width * cfg->num_actions; n->P2 = 2 * n->HWA;
    n->lh[0] = cfg->height; n->lw[0] = cfg->width; n->lc[0] = cfg->channels;
    for (int l = 0; l < L; ++l) { n->lh[l + 1] = n->lh[l] / 2; n->lw[l + 1] = n->lw[l] / 2; n->lc[l + 1] = n->F; }
    n->D0 = n->lh[L] * n->lw[L] * n->F;
    long p = 0;
    auto take = [&](long cnt) { long r = p; p += cnt; return r; };
    FieldOff &o = n->off;
    for (int l = 0; l < L; ++l) { o.cw[l] = take(9L * n->lc[l] * n->F); o.cb[l] = take(n->F); }
    o.d1w = take((long)n->D0 * 2 * FIELD_FC); o.d1b = take(2 * FIELD_FC); o.d2w = take(2 * FIELD_FC * FIELD_FC); o.d2b = take(FIELD_FC);
    o.p1w = take(FIELD_FC * 2 * FIELD_FC); o.p1b = take(2 * FIELD_FC); o.p2w = take(2L * FIELD_FC * n->P2); o.p2b = take(n->P2);
    o.muw = take((long)n->P2 * n->HWA); o.mub = take(n->HWA); o.sgw = take((long)n->P2 * n->HWA); o.sgb = take(n->HWA);
    o.v1w = take(FIELD_FC * 2 * FIELD_FC); o.v1b = take(2 * FIELD_FC); o.v2w = take(2 * FIELD_FC * FIELD_FC); o.v2b = take(FIELD_FC);
    o.v3w = take(FIELD_FC); o.v3b = take(1);
    o.total = p;
    const size_t ms = cfg->max_samples;
    int rc = GRL_OK;
    auto Al = [&](float **q, size_t cnt) { if (rc == GRL_OK) rc = xalloc(n, q, cnt); };
    Al(&n->params, o.total); Al(&n->grads, o.total); Al(&n->adam_m, o.total); Al(&n->adam_v, o.total);
    size_t maxconv = 0, maxslab = 0;
    for (int l = 0; l <= L; ++l) {
        const size_t e = ms * n->lh[l] * n->lw[l] * n->lc[l];
        Al(&n->x[l], e);
        if (l >= 1) { Al(&n->dxl[l], e); if (rc == GRL_OK) rc = xalloc(n, &n->idx[l - 1], e); }
        if (l < L) {
            const size_t conv = ms * n->lh[l] * n->lw[l] * n->F, T = 9 * (size_t)n->lc[l] * n->F + n->F;
            if (conv > maxconv) maxconv = conv;
            if (ms * T > maxslab) maxslab = ms * T;
        }
    }
    n->dxl[0] = nullptr;
    Al(&n->dzc, maxconv); Al(&n->cslab, maxslab);
    Al(&n->d1, ms * 64); Al(&n->d2, ms * 32); Al(&n->p1, ms * 64); Al(&n->p2, ms * n->P2); Al(&n->v1, ms * 64); Al(&n->v2, ms * 32);
    Al(&n->mu, ms * n->A); Al(&n->sigma, ms * n->A); Al(&n->vs, ms);
    Al(&n->dd1, ms * 64); Al(&n->dd2, ms * 32); Al(&n->dp1, ms * 64); Al(&n->dp2, ms * n->P2); Al(&n->dv1, ms * 64); Al(&n->dv2, ms * 32);
    Al(&n->dzh, ms * (2 * n->A + 3)); Al(&n->d_act, ms * n->A); Al(&n->d_adv, ms); Al(&n->d_y, ms); Al(&n->stats, 8);
    if (rc == GRL_OK) rc = xalloc(n, &n->d_pos, ms * 2);
    if (rc == GRL_OK) rc = xalloc(n, &n->stats64, 8 + 1024);
    if (rc == GRL_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = xfail(n, GRL_E_HIP, "hipStreamSynchronize");
    if (rc != GRL_OK) {
        fail(h, rc, "grl_fieldnet_create: " + n->err);
        grl_fieldnet_destroy(n);
        return rc;
    }
    *out = n;
    return GRL_OK;
}

int grl_fieldnet_destroy(grl_fieldnet *n) {
    if (!n) return GRL_OK;
    grl_sync_for_destroy(n->h);      // the handle may have been destroyed first (finaliser order of a host binding)
    for (void *p : n->allocs) hipFree(p);
    delete n;
    return GRL_OK;
}

const char *grl_fieldnet_last_error(const grl_fieldnet *n) { return n ? n->err.c_str() : ""; }
int64_t grl_fieldnet_num_params(const grl_fieldnet *n) { return n ? n->off.total : 0; }

static int field_copy_flat(grl_fieldnet *n, float *dev, float *host, int64_t cnt, bool to_dev) {
    if (!n || !host) return GRL_E_INVALID;
    if (cnt != n->off.total) return xfail(n, GRL_E_SIZE, "expected " + std::to_string(n->off.total) + " floats");
    hipSetDevice(n->h->cfg.device_id);
    FLD_HIP(n, hipStreamSynchronize(n->h->stream));
    if (to_dev) FLD_HIP(n, hipMemcpy(dev, host, cnt * 4, hipMemcpyHostToDevice));
    else FLD_HIP(n, hipMemcpy(host, dev, cnt * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}
int grl_fieldnet_set_params(grl_fieldnet *n, const float *host, int64_t cnt) { return field_copy_flat(n, n ? n->params : nullptr, (float *)host, cnt, true); }
int grl_fieldnet_get_params(grl_fieldnet *n, float *host, int64_t cnt) { return field_copy_flat(n, n ? n->params : nullptr, host, cnt, false); }
int grl_fieldnet_get_grads(grl_fieldnet *n, float *host, int64_t cnt) { return field_copy_flat(n, n ? n->grads : nullptr, host, cnt, false); }

int grl_fieldnet_predict(grl_fieldnet *net, int32_t n, const float *states, const int32_t *positions, float *mu, float *sigma, float *vs) {
    if (!net || n <= 0 || !states || !positions) return xfail(net, GRL_E_INVALID, "grl_fieldnet_predict: bad argument");
    if (n > net->cfg.max_samples) return xfail(net, GRL_E_SIZE, "grl_fieldnet_predict: n exceeds max_samples");
    hipSetDevice(net->h->cfg.device_id);
    int rc = field_upload(net, n, states, positions);
    if (rc == GRL_OK) rc = field_forward(net, n);
    if (rc) return rc;
    FLD_HIP(net, hipStreamSynchronize(net->h->stream));
    if (mu) FLD_HIP(net, hipMemcpy(mu, net->mu, (size_t)n * net->A * 4, hipMemcpyDeviceToHost));
    if (sigma) FLD_HIP(net, hipMemcpy(sigma, net->sigma, (size_t)n * net->A * 4, hipMemcpyDeviceToHost));
    if (vs) FLD_HIP(net, hipMemcpy(vs, net->vs, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}

int grl_fieldnet_train(grl_fieldnet *net, int32_t n, const float *states, const int32_t *positions, const float *actions, const float *advantages,
                    const float *critic_target, float lr, int32_t apply_update, float *stats_host) {
    if (!net || n <= 0 || !states || !positions || !actions || !advantages || !critic_target) return xfail(net, GRL_E_INVALID, "grl_fieldnet_train: bad argument");
    if (n > net->cfg.max_samples) return xfail(net, GRL_E_SIZE, "grl_fieldnet_train: n exceeds max_samples");
    hipSetDevice(net->h->cfg.device_id);
    hipStream_t st = net->h->stream;
    int rc = field_upload(net, n, states, positions);
    if (rc) return rc;
    FLD_HIP(net, hipMemcpyAsync(net->d_act, actions, (size_t)n * net->A * 4, hipMemcpyHostToDevice, st));
    FLD_HIP(net, hipMemcpyAsync(net->d_adv, advantages, (size_t)n * 4, hipMemcpyHostToDevice, st));
    FLD_HIP(net, hipMemcpyAsync(net->d_y, critic_target, (size_t)n * 4, hipMemcpyHostToDevice, st));
    if ((rc = field_forward(net, n))) return rc;
    const float *P = net->params;
    float *G = net->grads;
    const FieldOff &o = net->off;
    const int P2 = net->P2, HWA = net->HWA, W = net->cfg.width;
    // the gathered heads touch A columns per sample: the rest of their gradient is zero
    FLD_HIP(net, hipMemsetAsync(G + o.muw, 0, (size_t)(o.v1w - o.muw) * 4, st));
    FIELD_DISPATCH(net->A, hipLaunchKernelGGL(field_heads_bwd_kernel<kA>, dim3((n + 3) / 4), dim3(256), 0, st, net->p2, net->v2, net->d_pos, net->mu,
                                              net->sigma, net->vs, net->d_act, net->d_adv, net->d_y, P, o, n, P2, HWA, W, net->cfg.scale,
                                              net->cfg.entropy_beta, 1.0f / (float)n, net->dzh, net->dp2, net->dv2));
    FIELD_DISPATCH(net->A, hipLaunchKernelGGL(field_heads_wgrad_kernel<kA>, dim3(nb(P2 > 64 ? P2 : 64)), dim3(256), 0, st, net->p2, net->v2, net->d_pos,
                                              net->dzh, o, n, P2, HWA, W, G, net->stats64));
    auto wgrad = [&](const float *in, const float *dout, int K, int J, long w, long b) {
        hipLaunchKernelGGL(field_dense_wgrad_kernel, dim3(nb((long)(K + 1) * J)), dim3(256), 0, st, in, dout, (long)n, K, J, G + w, G + b);
    };
    auto dgrad = [&](const float *dout, long w, const float *fwd, int K, int J, int acc, float *din) {
        hipLaunchKernelGGL(field_dense_dgrad_kernel, dim3(nb((long)n * K)), dim3(256), 0, st, dout, P + w, fwd, (long)n, K, J, acc, din);
    };
    wgrad(net->p1, net->dp2, 2 * FIELD_FC, P2, o.p2w, o.p2b);
    dgrad(net->dp2, o.p2w, net->p1, 2 * FIELD_FC, P2, 0, net->dp1);
    wgrad(net->d2, net->dp1, FIELD_FC, 2 * FIELD_FC, o.p1w, o.p1b);
    wgrad(net->v1, net->dv2, 2 * FIELD_FC, FIELD_FC, o.v2w, o.v2b);
    dgrad(net->dv2, o.v2w, net->v1, 2 * FIELD_FC, FIELD_FC, 0, net->dv1);
    wgrad(net->d2, net->dv1, FIELD_FC, 2 * FIELD_FC, o.v1w, o.v1b);
    // dd2 = relu'(d2) . (dp1 W_p1^T + dv1 W_v1^T): both branches of processed_state first, then the mask
    dgrad(net->dp1, o.p1w, nullptr, FIELD_FC, 2 * FIELD_FC, 0, net->dd2);
    dgrad(net->dv1, o.v1w, nullptr, FIELD_FC, 2 * FIELD_FC, 1, net->dd2);
    hipLaunchKernelGGL(field_relu_mask_kernel, dim3(nb((long)n * FIELD_FC)), dim3(256), 0, st, net->d2, (long)n * FIELD_FC, net->dd2);
    wgrad(net->d1, net->dd2, 2 * FIELD_FC, FIELD_FC, o.d2w, o.d2b);
    dgrad(net->dd2, o.d2w, net->d1, 2 * FIELD_FC, FIELD_FC, 0, net->dd1);
    wgrad(net->x[net->L], net->dd1, net->D0, 2 * FIELD_FC, o.d1w, o.d1b);
    dgrad(net->dd1, o.d1w, nullptr, net->D0, 2 * FIELD_FC, 0, net->dxl[net->L]);   // the ReLU of the pooled map is applied by pool_bwd
    for (int l = net->L - 1; l >= 0; --l) {
        const int H = net->lh[l], Wd = net->lw[l], Cin = net->lc[l], F = net->F;
        const long conv = (long)n * H * Wd * F;
        const int T = 9 * Cin * F + F;
        hipLaunchKernelGGL(field_pool_bwd_kernel, dim3(nb(conv)), dim3(256), 0, st, net->dxl[l + 1], net->x[l + 1], net->idx[l], conv, H, Wd, F, net->dzc);
        hipLaunchKernelGGL(field_conv_wgrad_kernel, dim3(n), dim3(256), 0, st, net->x[l], net->dzc, H, Wd, Cin, F, net->cslab);
        hipLaunchKernelGGL(field_slab_reduce_kernel, dim3(nb(T)), dim3(256), 0, st, net->cslab, (long)n, T, 9 * Cin * F, G + o.cw[l], G + o.cb[l]);
        if (l > 0) {
            const long tin = (long)n * H * Wd * Cin;
            hipLaunchKernelGGL(field_conv_dgrad_kernel, dim3(nb(tin)), dim3(256), 0, st, net->dzc, P + o.cw[l], tin, H, Wd, Cin, F, net->dxl[l]);
        }
    }
    const int nparts = 1024;
    hipLaunchKernelGGL(field_sumsq_kernel, dim3(nparts), dim3(256), 0, st, G, o.total, net->stats64 + 8);
    hipLaunchKernelGGL(field_finalize_kernel, dim3(1), dim3(64), 0, st, net->stats64 + 8, nparts, net->stats64, 1.0f / (float)n, net->cfg.clip_norm,
                       net->stats);
    if (apply_update) {
        net->adam_t += 1;
        const float lr_t = (float)((double)lr * sqrt(1.0 - pow(0.999, (double)net->adam_t)) / (1.0 - pow(0.9, (double)net->adam_t)));
        hipLaunchKernelGGL(field_adam_kernel, dim3(nb(o.total)), dim3(256), 0, st, net->params, G, net->adam_m, net->adam_v, o.total, net->stats, lr_t);
    }
    FLD_HIP(net, hipGetLastError());
    FLD_HIP(net, hipStreamSynchronize(st));
    if (stats_host) {
        float s[5];
        FLD_HIP(net, hipMemcpy(s, net->stats, sizeof(s), hipMemcpyDeviceToHost));
        stats_host[0] = s[2]; stats_host[1] = s[0]; stats_host[2] = s[1]; stats_host[3] = s[3];
    }
    return GRL_OK;
}

}  // extern "C"
