// FlatPolicyVNetwork on gfx950 (reference fed_gym/agents/paac/policy_v_network.py:194-264 with the GRU
// trunk of fed_gym/agents/a3c/estimators.py:5-28), its loss/backward/optimiser and the flat PAAC rollout
// (fed_gym/agents/paac/paac.py:119-187) for Solow handles and, PAAC-style, TradeAR1 handles (BASELINE config 5:
// obs transform of a3c/worker.py:420-431, tanh actions :440-442, hyper-parameters of scripts/train_trade.py:38-40,119).
//
// The net has ~31k parameters and a few tens of kMAC per sample; BASELINE config 2 (4 096 envs) is
// launch/latency bound, so the whole forward (5 GRU steps + 13 dense layers) is ONE launch and the
// whole backward one more.  A workgroup of 4 waves owns 64 samples; activations are staged in LDS as
// [feature][sample] (row stride 65) and every dense layer and every gradient is a small fp32 MFMA GEMM against
// those rows (net_flat_mfma.inc).  Weight gradients are summed per workgroup into private slabs and reduced in
// a fixed order (bitwise reproducible).
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/goldsrl_flatnet.h"
#include "common.h"
#include "rng.h"
#include "flat_env_dev.h"
#include "rollout_dev.h"

namespace grl {

constexpr int FH = 32;       // rnn hidden (and static hidden)
constexpr int LS = 65;       // LDS row stride
constexpr int MAXD = 33;     // temporal_size <= 33 (TradeAR1-16: 1+2n)
constexpr int MAXS0 = 64;
constexpr int MAXA = 16;
enum : uint32_t { RS_FLAT_ACTION = 17 };

struct FOff {
    long gw, gb, cw, cb, tw, tb, s1w, s1b, s2w, s2b, m1w, m1b, m2w, m2b, m3w, m3b, g1w, g1b, g2w, g2b, g3w, g3b, v1w, v1b, v2w, v2b, total;
};

static FOff make_offsets(int D, int S0, int A) {
    FOff o;
    long p = 0;
    auto take = [&](long n) { long r = p; p += n; return r; };
    o.gw = take((long)(D + FH) * 2 * FH); o.gb = take(2 * FH);
    o.cw = take((long)(D + FH) * FH); o.cb = take(FH);
    o.tw = take(FH * 2 * FH); o.tb = take(2 * FH);
    o.s1w = take((long)S0 * 2 * FH); o.s1b = take(2 * FH);
    o.s2w = take(2 * FH * FH); o.s2b = take(FH);
    o.m1w = take(3 * FH * 2 * FH); o.m1b = take(2 * FH); o.m2w = take(2 * FH * FH); o.m2b = take(FH); o.m3w = take((long)FH * A); o.m3b = take(A);
    o.g1w = take(3 * FH * 2 * FH); o.g1b = take(2 * FH); o.g2w = take(2 * FH * FH); o.g2b = take(FH); o.g3w = take((long)FH * A); o.g3b = take(A);
    o.v1w = take(3 * FH * 2 * FH); o.v1b = take(2 * FH); o.v2w = take(2 * FH); o.v2b = take(1);
    o.total = p;
    return o;
}

// workspace features per sample: per GRU step {h_prev, r, u, c} then the dense activations.
// Layout [group of 64 samples][feature][64] (ws_row): the ~3 000 rows a group's backward reads are one contiguous block of 0.8 MB.
// As [feature][n] they lay n * 4 bytes apart -- 655 KB at 8 192 envs x 20 steps, every row of a tile on another page.
__host__ __device__ inline int ws_step(int t) { return t * 4 * FH; }
struct WsOff { int dt, s1, s2, m1, m2, tm, g1, g2, sg, v1, hl, vs, total; };
__host__ __device__ inline WsOff ws_offsets(int T, int A) {
    WsOff w;
    int p = T * 4 * FH;
    w.dt = p; p += 64; w.s1 = p; p += 64; w.s2 = p; p += 32; w.m1 = p; p += 64; w.m2 = p; p += 32; w.tm = p; p += A;
    w.g1 = p; p += 64; w.g2 = p; p += 32; w.sg = p; p += A; w.v1 = p; p += 64; w.hl = p; p += 32; w.vs = p; p += 1;
    w.total = p;
    return w;
}
// row `f` of the group that starts at sample sbase (a multiple of 64): 64 floats
__host__ __device__ inline float *ws_row(float *ws, int f, int sbase, int F) { return ws + ((size_t)(sbase >> 6) * F + f) * 64; }
__host__ __device__ inline const float *ws_row(const float *ws, int f, int sbase, int F) { return ws + ((size_t)(sbase >> 6) * F + f) * 64; }

struct FlatArgs {
    const float *P;
    const float *PT;                // backward only: every [K][N] weight matrix as [N][K] at the same offset (flat_transpose_kernel)
    FOff o;
    int n, S0, D, T, A;
    float scale, bound;
    const float *states, *hist;     // (n,S0) (n,T,D) row-major
    const int32_t *nhist;           // if set: hist is NOT read; row t of sample s is states[s] for t < min(nhist[s],T), else 0
                                    // (the PAAC worker's window, quirk Q11; needs D == S0)
    float *mu, *sigma, *vs;         // (n,A) (n,A) (n)
    float *ws;                      // [feature][n] or nullptr (inference)
    // backward only
    const float *actions, *adv, *y;
    float inv_n;
    float *slab;                    // [blocks][o.total]
    double *stats64;
    long long *ts;                  // stage clock of workgroup 0 (debug, grl_fnet_rollout_stage_times) or nullptr
    int *ts_n;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

#include "net_flat_mfma.inc"
#include "net_flat_fast.inc"
#include "net_flat_bwd_fast.inc"
#include "net_flat_rollout.inc"

// PT: the weight matrices the data gradients multiply with, transposed (block b = matrix b; the GRU kernels: their recurrent rows)
__global__ void flat_transpose_kernel(const float *__restrict__ P, float *__restrict__ PT, FOff o, int D, int A) {
    long off;
    int K, N;
    switch (blockIdx.x) {
        case 0: off = o.m1w; K = 3 * FH; N = 2 * FH; break;
        case 1: off = o.g1w; K = 3 * FH; N = 2 * FH; break;
        case 2: off = o.v1w; K = 3 * FH; N = 2 * FH; break;
        case 3: off = o.m2w; K = 2 * FH; N = FH; break;
        case 4: off = o.g2w; K = 2 * FH; N = FH; break;
        case 5: off = o.m3w; K = FH; N = A; break;
        case 6: off = o.g3w; K = FH; N = A; break;
        case 7: off = o.tw; K = FH; N = 2 * FH; break;
        case 8: off = o.s2w; K = 2 * FH; N = FH; break;
        case 9: off = o.cw + (long)D * FH; K = FH; N = FH; break;
        default: off = o.gw + (long)D * 2 * FH; K = FH; N = 2 * FH; break;
    }
    for (int e = threadIdx.x; e < K * N; e += blockDim.x) {
        const int i = e / N, c = e - i * N;
        PT[off + (long)c * K + i] = P[off + e];
    }
}
constexpr int kFlatTransposed = 11;

// dst[i] = sum over the workgroups' slabs, in a fixed order: 64 parameters per block, thread (q, p) adds the slabs q, q + 16, ..
// for parameter p (16 loads in flight per parameter instead of one thread walking all 256 slabs), the 16 partial sums then in order
__global__ __launch_bounds__(1024) void flat_slab_reduce_kernel(const float *__restrict__ slab, int blocks, long n, float *__restrict__ dst) {
    __shared__ float part[16][64];
    const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = (long)blockIdx.x * 64 + p;
    float s = 0.f;
    if (i < n)
        for (int b = q; b < blocks; b += 16) s += slab[(long)b * n + i];
    part[q][p] = s;
    __syncthreads();
    if (q == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][p];
        dst[i] = t;
    }
}

// sum of squares in float64: kSumsqBlocks partial sums (a contiguous slice each, tree inside the block), added up in order by
// flat_finalize_kernel
constexpr int kSumsqBlocks = 32;
__global__ __launch_bounds__(256) void flat_sumsq_kernel(const float *__restrict__ g, long n, double *__restrict__ out) {
    __shared__ double red[256];
    const long per = (n + kSumsqBlocks - 1) / kSumsqBlocks, lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double s = 0.0;
    for (long i = lo + threadIdx.x; i < hi; i += 256) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// grad_scale: 1/world after a sum over ranks (the loss is a mean over the WHOLE batch, policy_v_network.py:246-251); the norm and
// the clip are those of grad_scale * grads, and the factor is folded into what Adam multiplies the stored gradient by
__global__ void flat_finalize_kernel(const double *__restrict__ sumsq, const double *__restrict__ stats64, float inv_n, float inv_na,
                                     float clip_norm, float grad_scale, float *__restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double ss = 0.0;
    for (int b = 0; b < kSumsqBlocks; ++b) ss += sumsq[b];
    float norm = (float)(sqrt(ss) * (double)grad_scale);
    float pl = (float)(stats64[0] * (double)inv_na), cl = (float)(stats64[1] * (double)inv_n);
    stats[0] = pl; stats[1] = cl; stats[2] = pl + cl; stats[3] = norm;
    stats[4] = grad_scale * (clip_norm > 0.f ? clip_norm / fmaxf(norm, clip_norm) : 1.0f);      // tf.clip_by_global_norm
}

__global__ void flat_adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
                                 const float *__restrict__ stats, float lr_t) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] * stats[4];
    float mi = 0.9f * m[i] + 0.1f * gi;
    float vi = 0.999f * v[i] + 0.001f * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + 1e-8f);
}

// a = mu + sigma*N(0,1) (paac.py:36), SolowRunner.transform_actions_for_env = sigmoid (emulator_runner.py:77-79)
__global__ void flat_sample_kernel(const float *__restrict__ mu, const float *__restrict__ sigma, int n, int A, uint64_t seed,
                                   uint32_t env_off, const uint32_t *__restrict__ counter_base, uint32_t step, int env_kind,
                                   float *__restrict__ raw, float *__restrict__ envact) {
    // the draw counter is read from device memory so that a captured rollout graph can be replayed (base is set per rollout)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * A) return;
    const uint32_t counter = *counter_base + step;
    int e = i / A, k = i - e * A;
    double e0, e1;
    normal_pair(rng_block(seed, (uint32_t)e + env_off, counter, RS_FLAT_ACTION, k >> 1), e0, e1);
    float r = (float)((double)mu[i] + (double)sigma[i] * ((k & 1) ? e1 : e0));
    raw[i] = r;
    if (env_kind == GRL_ENV_SOLOW) {
        float z = expf(-fabsf(r));
        envact[i] = r >= 0.f ? 1.0f / (1.0f + z) : z / (1.0f + z);
    } else {
        envact[i] = tanhf(r);
    }
}

__global__ void flat_mask_kernel(const uint8_t *__restrict__ done, int n, float *__restrict__ mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) mask[i] = 1.0f - (float)done[i];      // episodes_over_masks (paac.py:140)
}

}  // namespace grl

struct grl_fnet {
    grl_handle *h;
    grl_fnet_config cfg;
    std::string err;
    grl::FOff off;
    grl::WsOff wso;
    float *params, *paramsT, *grads, *adam_m, *adam_v;      // paramsT: net_flat_bwd_fast.inc (fb_dx)
    long adam_t;
    float *ws, *slab, *stats;
    double *stats64;
    float *d_states, *d_hist, *d_act, *d_adv, *d_y, *mu, *sigma, *vs;
    int slab_blocks;
    // rollout
    int T;
    float *ro_states, *ro_hist, *ro_act, *ro_envact, *ro_val, *ro_rew, *ro_mask, *ro_y, *ro_adv, *ro_boot;
    int32_t *ro_nhist;
    unsigned long act_counter;
    uint32_t *d_counter;           // act_counter at the start of the rollout in flight
    hipGraphExec_t ro_graph;       // the T-step rollout captured once and replayed (launch-bound at 4 096 envs)
    int ro_graph_T;
    bool ro_graph_ep;              // the captured rollout contains the R6 accounting launches
    int fast_forward;              // 1: synthesized-window forwards use net_flat_fast.inc (GRL_FLAT_FORWARD=layers: the layer-by-layer form)
    int ro_persistent;             // 1: the T-step actor loop is ONE persistent kernel (net_flat_rollout.inc); 0: the hipGraph of launches
    char ro_attr_set[2][3];        // the rollout kernel instance (KEEP x G = 16, 32, 64) has its dynamic-LDS limit raised on this net's device
    int ro_group;                  // envs per workgroup of the persistent rollout; 0: by the env count (GRL_FLAT_GROUP)
    int keep_activations;          // grl_fnet_set_keep_activations: the persistent rollout fills the training workspace
    int ws_resident;               // the workspace holds the forward of the last rollout's T x E samples under the CURRENT parameters
    int arg_slot;                  // this net's slot of g_flat_args (net_flat_fast.inc), -1: none free (graph path)
    long long *d_ts;               // stage timestamps of workgroup 0 of the persistent rollout (debug: grl_fnet_rollout_stage_times)
    int *d_ts_n;
    int last_n;                    // samples of the last gradient pass (grl_fnet_apply_grads normalises the loss sums with it)
    void *comm;                    // ncclComm_t (RCCL): one all-reduce of the flat gradient per rollout, or nullptr
    int comm_world, comm_rank;
    hipEvent_t ar_ev0, ar_ev1;     // bracket the all-reduce on the handle's stream (grl_fnet_comm_info)
    int ar_pending;
    long ar_calls;
    double ar_ms_total;
    float ar_ms_last;
    std::vector<void *> allocs;
};

namespace grl {

static int ffail(grl_fnet *n, int code, const std::string &msg) {
    if (n) n->err = msg;
    return code;
}
#define FNET_HIP(n, call)                                                                                  \
    do {                                                                                                   \
        hipError_t _e = (call);                                                                            \
        if (_e != hipSuccess) return ffail(n, GRL_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

template <typename T>
static int falloc(grl_fnet *n, T **p, size_t count) {
    FNET_HIP(n, hipMalloc((void **)p, count * sizeof(T)));
    n->allocs.push_back(*p);
    FNET_HIP(n, hipMemsetAsync(*p, 0, count * sizeof(T), n->h->stream));
    return GRL_OK;
}

static FlatArgs base_args(grl_fnet *net, int n, const float *states, const float *hist, float *mu, float *sigma, float *vs, bool save,
                          const int32_t *nhist = nullptr) {
    FlatArgs a{};
    a.P = net->params; a.o = net->off; a.n = n; a.S0 = net->cfg.static_size; a.D = net->cfg.temporal_size; a.T = net->cfg.rnn_length;
    a.A = net->cfg.num_actions; a.scale = net->cfg.scale; a.bound = net->cfg.mu_bound; a.states = states; a.hist = hist;
    a.mu = mu; a.sigma = sigma; a.vs = vs; a.ws = save ? net->ws : nullptr; a.nhist = nhist;
    a.ts = net->d_ts; a.ts_n = net->d_ts_n;
    return a;
}

static int launch_forward(grl_fnet *net, int n, const float *states, const float *hist, float *mu, float *sigma, float *vs, bool save,
                          const int32_t *nhist = nullptr) {
    FlatArgs a = base_args(net, n, states, hist, mu, sigma, vs, save, nhist);
    const int groups = (n + 63) / 64;
    if (nhist && net->cfg.static_size == net->cfg.temporal_size && net->fast_forward) {
        // synthesized window (the PAAC worker's: the current state repeated): the 2T + 5 stage form of net_flat_fast.inc
        if (save) hipLaunchKernelGGL(flat_forward_fast_kernel<true>, dim3(groups), dim3(FNT), ff_lds_bytes(net->cfg.static_size), net->h->stream, a);
        else hipLaunchKernelGGL(flat_forward_fast_kernel<false>, dim3(groups), dim3(FNT), ff_lds_bytes(net->cfg.static_size), net->h->stream, a);
        FNET_HIP(net, hipGetLastError());
        return GRL_OK;
    }
    const size_t wbytes = (size_t)(net->cfg.temporal_size + FH) * 3 * FH * sizeof(float);
    static const int wlds_max_groups = getenv("GRL_FLAT_WLDS_GROUPS") ? atoi(getenv("GRL_FLAT_WLDS_GROUPS")) : 256;      // one workgroup per CU; measured (tools/bench_trade_sizes.py): 128 / 256 groups 7.6 -> 5.5 ms, 384 / 512 groups 8.5 -> 10.4 ms
    if (groups <= wlds_max_groups)
        hipLaunchKernelGGL(flat_forward_kernel<true>, dim3(groups), dim3(256), FLAT_LDS_BYTES + wbytes, net->h->stream, a, (int)FLAT_LDS_ROWS);
    else
        hipLaunchKernelGGL(flat_forward_kernel<false>, dim3(groups), dim3(256), FLAT_LDS_BYTES, net->h->stream, a, (int)FLAT_LDS_ROWS);
    FNET_HIP(net, hipGetLastError());
    return GRL_OK;
}

// forward(save) + backward over n device-resident samples; grads <- gradient of the mean loss over THESE n samples
// resident: the workspace already holds this forward (the rollout that produced the samples kept its activations, and the
// parameters have not moved since): the pass starts at the backward
static int train_grads_device(grl_fnet *net, int n, const float *states, const float *hist, const float *actions, const float *adv,
                              const float *y, const int32_t *nhist = nullptr, bool resident = false) {
    hipStream_t st = net->h->stream;
    if (n > net->cfg.max_samples) return ffail(net, GRL_E_SIZE, "train: n exceeds max_samples of the net");
    int rc = GRL_OK;
    if (!resident) {
        net->ws_resident = 0;
        rc = launch_forward(net, n, states, hist, net->mu, net->sigma, net->vs, true, nhist);
    }
    if (rc) return rc;
    int groups = (n + 63) / 64;
    const bool fast = nhist && net->cfg.static_size == net->cfg.temporal_size && net->fast_forward && net->arg_slot >= 0;
    int blocks = groups < net->slab_blocks ? groups : net->slab_blocks;
    if (fast && blocks > 256) blocks = 256;      // the 16-wave form: one workgroup per CU (100 KB of LDS)
    if (!fast) FNET_HIP(net, hipMemsetAsync(net->slab, 0, (size_t)blocks * net->off.total * 4, st));      // the fast backward clears its own
    FNET_HIP(net, hipMemsetAsync(net->stats64, 0, 4 * sizeof(double), st));
    FlatArgs a = base_args(net, n, states, hist, net->mu, net->sigma, net->vs, true, nhist);
    a.actions = actions; a.adv = adv; a.y = y; a.inv_n = 1.0f / (float)n; a.slab = net->slab; a.stats64 = net->stats64;
    if (fast && net->arg_slot >= 0) {
        a.PT = net->paramsT;
        hipLaunchKernelGGL(flat_transpose_kernel, dim3(kFlatTransposed), dim3(256), 0, st, net->params, net->paramsT, net->off,
                           net->cfg.temporal_size, net->cfg.num_actions);
        // the stages of the fast backward are calls that read their arguments from the net's __constant__ slot (net_flat_fast.inc)
        FNET_HIP(net, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_flat_args), &a, sizeof(FlatArgs), (size_t)net->arg_slot * sizeof(FlatArgs),
                                             hipMemcpyHostToDevice, st));
        (void)hipGetLastError();
        hipLaunchKernelGGL(flat_backward_fast_kernel, dim3(blocks), dim3(FNT), FB2_LDS_BYTES, st, net->arg_slot, n, net->cfg.rnn_length, a.ts, a.ts_n);
    } else {
        hipLaunchKernelGGL(flat_backward_kernel, dim3(blocks), dim3(256), FLAT_LDS_BYTES, st, a);
    }
    hipLaunchKernelGGL(flat_slab_reduce_kernel, dim3((unsigned)((net->off.total + 63) / 64)), dim3(1024), 0, st, net->slab, blocks,
                       net->off.total, net->grads);
    FNET_HIP(net, hipGetLastError());
    net->last_n = n;
    return GRL_OK;
}

// One all-reduce (sum, fp32) of the flat gradient per rollout over RCCL/xGMI, the rule of net_train.inc: every rank's
// gradient is the mean over ITS T*E_local samples, the loss is a mean over the whole batch (policy_v_network.py:246-251), so the
// sum is scaled by 1/world (folded into the clip factor); clip after the reduction, Adam replicated.
static int fcomm_allreduce_grads(grl_fnet *net, float *grad_scale_out) {
    *grad_scale_out = 1.0f;
    if (!net->comm) return GRL_OK;
    if (!net->ar_ev0) { FNET_HIP(net, hipEventCreate(&net->ar_ev0)); FNET_HIP(net, hipEventCreate(&net->ar_ev1)); }
    FNET_HIP(net, hipEventRecord(net->ar_ev0, net->h->stream));
    ncclResult_t r = ncclAllReduce(net->grads, net->grads, (size_t)net->off.total, ncclFloat, ncclSum, (ncclComm_t)net->comm, net->h->stream);
    (void)hipGetLastError();   // RCCL probes may leave a stale HIP error on this thread
    if (r != ncclSuccess) return ffail(net, GRL_E_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    FNET_HIP(net, hipEventRecord(net->ar_ev1, net->h->stream));
    net->ar_pending = 1;
    *grad_scale_out = 1.0f / (float)net->comm_world;
    return GRL_OK;
}

// global norm of grad_scale * grads, clip factor, loss statistics of the last gradient pass; then (apply_update) Adam
static int train_apply_device(grl_fnet *net, float lr, int apply_update, float grad_scale, float *stats_host) {
    hipStream_t st = net->h->stream;
    const int n = net->last_n;
    hipLaunchKernelGGL(flat_sumsq_kernel, dim3(kSumsqBlocks), dim3(256), 0, st, net->grads, net->off.total, net->stats64 + 4);
    hipLaunchKernelGGL(flat_finalize_kernel, dim3(1), dim3(64), 0, st, net->stats64 + 4, net->stats64, 1.0f / (float)n,
                       1.0f / ((float)n * (float)net->cfg.num_actions), net->cfg.clip_norm, grad_scale, net->stats);
    if (apply_update) {
        net->ws_resident = 0;      // the parameters move
        net->adam_t += 1;
        float lr_t = (float)((double)lr * sqrt(1.0 - pow(0.999, (double)net->adam_t)) / (1.0 - pow(0.9, (double)net->adam_t)));
        hipLaunchKernelGGL(flat_adam_kernel, dim3((unsigned)((net->off.total + 255) / 256)), dim3(256), 0, st, net->params, net->grads,
                           net->adam_m, net->adam_v, net->off.total, net->stats, lr_t);
    }
    FNET_HIP(net, hipGetLastError());
    FNET_HIP(net, hipStreamSynchronize(st));
    if (net->ar_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, net->ar_ev0, net->ar_ev1) == hipSuccess) { net->ar_ms_last = ms; net->ar_ms_total += ms; net->ar_calls += 1; }
        net->ar_pending = 0;
    }
    if (stats_host) {
        float s[5];
        FNET_HIP(net, hipMemcpy(s, net->stats, sizeof(s), hipMemcpyDeviceToHost));
        stats_host[0] = s[2]; stats_host[1] = s[0]; stats_host[2] = s[1]; stats_host[3] = s[3];
    }
    return GRL_OK;
}

// gradient pass [+ all-reduce over ranks if a communicator is attached and the parameters are to be updated] + clip + Adam
static int train_device(grl_fnet *net, int n, const float *states, const float *hist, const float *actions, const float *adv, const float *y,
                        float lr, int apply_update, float *stats_host, const int32_t *nhist = nullptr, bool resident = false) {
    int rc = train_grads_device(net, n, states, hist, actions, adv, y, nhist, resident);
    if (rc) return rc;
    float grad_scale = 1.0f;
    if (apply_update && (rc = fcomm_allreduce_grads(net, &grad_scale))) return rc;
    return train_apply_device(net, lr, apply_update, grad_scale, stats_host);
}

// the T-step actor loop as stream operations (captured into a graph by grl_fnet_rollout)
static int enqueue_rollout(grl_fnet *net, int T) {
    grl_handle *h = net->h;
    const bool solow = h->cfg.env_kind == GRL_ENV_SOLOW;
    const int E = h->E, R = net->cfg.rnn_length, S0 = net->cfg.static_size, A = net->cfg.num_actions;
    hipStream_t st = h->stream;
    const float *obs = solow ? h->so.obs : h->tr.obs;
    int rc;
    for (int t = 0; t < T; ++t) {
        // states[t] = shared_states, histories[t] = shared_histories (paac.py:132-133).  For TradeAR1 the window is
        // kept as (state, #rows): the worker's history is min(n, rnn) copies of the current state (quirk Q11)
        FNET_HIP(net, hipMemcpyAsync(net->ro_states + (size_t)t * E * S0, obs, (size_t)E * S0 * 4, hipMemcpyDeviceToDevice, st));
        // the window is kept as (state, #rows) for both envs -- the worker's history is min(n, rnn) copies of the current state
        // (quirk Q11); Solow's dense (E, rnn, 2) form is recorded too (grl_fnet_read_rollout "histories")
        const int32_t *nh = solow ? h->so.nhist : h->tr.nhist;
        if (solow) FNET_HIP(net, hipMemcpyAsync(net->ro_hist + (size_t)t * E * R * 2, h->so.history, (size_t)E * R * 8, hipMemcpyDeviceToDevice, st));
        FNET_HIP(net, hipMemcpyAsync(net->ro_nhist + (size_t)t * E, nh, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        rc = launch_forward(net, E, obs, nullptr, net->mu, net->sigma, net->ro_val + (size_t)t * E, false, nh);
        if (rc) return rc;
        hipLaunchKernelGGL(flat_sample_kernel, dim3((E * A + 255) / 256), dim3(256), 0, st, net->mu, net->sigma, E, A, h->cfg.seed,
                           (uint32_t)h->cfg.env_id_offset, (const uint32_t *)net->d_counter, (uint32_t)t, h->cfg.env_kind,
                           net->ro_act + (size_t)t * E * A, net->ro_envact);
        rc = solow ? solow_launch_step(h, net->ro_envact) : trade_launch_step(h, net->ro_envact);
        if (rc) return ffail(net, rc, h->err);
        if ((rc = episodes_launch_account(h))) return ffail(net, rc, h->err);      // R6 (paac.py:142-157), when enabled on the handle
        FNET_HIP(net, hipMemcpyAsync(net->ro_rew + (size_t)t * E, h->reward, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(flat_mask_kernel, dim3((E + 255) / 256), dim3(256), 0, st, h->done, E, net->ro_mask + (size_t)t * E);
    }
    rc = launch_forward(net, E, obs, nullptr, net->mu, net->sigma, net->ro_boot, false, solow ? h->so.nhist : h->tr.nhist);
    if (rc) return rc;
    // rewards clipped to [-2, 2] (paac.py:145), masked n-step return (paac.py:167-172), adv / scale (paac.py:177)
    // gae_lambda < 1: the A3C worker's GAE on the raw rewards (a3c/worker.py:232-294)
    const bool gae = net->cfg.gae_lambda < 1.0f;
    if ((rc = launch_returns(h, net->ro_rew, net->ro_val, net->ro_mask, net->ro_boot, T, E, net->cfg.gamma, gae ? net->cfg.gae_lambda : 1.0f,
                             net->cfg.scale, gae ? 0.f : -2.f, gae ? 0.f : 2.f, net->ro_y, net->ro_adv)))
        return ffail(net, rc, h->err);
    return GRL_OK;
}

// the T-step actor loop as ONE kernel: a workgroup owns 64 envs for the whole rollout (net_flat_rollout.inc).  Returns
// GRL_E_SIZE when the rollout's LDS footprint does not fit a CU (the caller then takes the graph path).
// slots of g_flat_args: one per live net of the process (a net that finds none free keeps the graph path)
static std::mutex g_slot_mu;
static bool g_slot_used[kFlatArgSlots];
static int flat_slot_take() {
    std::lock_guard<std::mutex> lk(g_slot_mu);
    for (int i = 0; i < kFlatArgSlots; ++i)
        if (!g_slot_used[i]) { g_slot_used[i] = true; return i; }
    return -1;
}
static void flat_slot_release(int i) {
    if (i < 0) return;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    g_slot_used[i] = false;
}

static int launch_persistent_rollout(grl_fnet *net, int T) {
    grl_handle *h = net->h;
    if (net->arg_slot < 0) return GRL_E_SIZE;
    const bool solow = h->cfg.env_kind == GRL_ENV_SOLOW;
    RolloutArgs R{};
    R.f = base_args(net, h->E, nullptr, nullptr, nullptr, nullptr, nullptr, false);
    const bool keep = net->keep_activations != 0;
    if (keep) R.f.ws = net->ws;
    R.steps = T; R.env_kind = h->cfg.env_kind; R.slot = net->arg_slot;
    const int n_assets = solow ? 0 : h->cfg.n_assets;
    const size_t lds_bytes = (size_t)rollout_lds_floats(net->cfg.static_size, T, n_assets, &R) * sizeof(float);
    if (lds_bytes > 160 * 1024) return GRL_E_SIZE;
    // envs per workgroup: a workgroup has a CU to itself (LDS), and a group's chain of stages is as long for 16 envs as for 64 with
    // fewer tiles per stage -- so the smallest group that still gives every CU at most one workgroup (GRL_FLAT_GROUP = 64 / 32 / 16
    // fixes it).  Measured at the end of round 5 (tools/flat_group_sweep.py, ms per 20-step rollout, G = 64 / 32 / 16): TradeAR1-16
    // 1 024 envs 1.74 / 1.25 / 1.16, 4 096: 1.76 / 1.25 / 1.17, 8 192: 1.76 / 1.26 / 2.30, 16 384: 1.78 / 2.48 / 4.55; Solow 4 096:
    // 0.79 / 0.51 / 0.48, 8 192: 0.80 / 0.51 / 0.91.  (While every forward call still moved 44 callee-saved registers through
    // scratch, 256 workgroups at once were slower than 128 and groups of 16 paid only up to 1 024 envs.)
    int G = net->ro_group;
    if (!G) G = h->E <= 16 * 256 ? 16 : (h->E <= 32 * 256 ? 32 : 64);
    R.gs = G;
    typedef void (*RoKernel)(int);
    static const RoKernel kernels[2][3] = {{flat_rollout_kernel<16, false>, flat_rollout_kernel<32, false>, flat_rollout_kernel<64, false>},
                                           {flat_rollout_kernel<16, true>, flat_rollout_kernel<32, true>, flat_rollout_kernel<64, true>}};
    const int gi = G == 16 ? 0 : (G == 32 ? 1 : 2);
    const RoKernel kern_fn = kernels[keep ? 1 : 0][gi];
    const void *kern = (const void *)kern_fn;
    // the attribute belongs to the kernel (per device), not to the net: every net raises it to the CU's whole LDS, so that no net's
    // smaller rollout lowers it under another net's larger one
    char &attr_set = net->ro_attr_set[keep ? 1 : 0][gi];
    if (!attr_set) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return GRL_E_SIZE;
        }
        attr_set = 1;
    }
    if (solow) {
        R.so = solow_params(h);
        R.needs_tape = (!(h->cfg.flags & GRL_F_RESET_FROM_SNAPSHOT) && h->cfg.max_episode_steps > 0) ? 1 : 0;
        R.obs0 = h->so.obs;
    } else {
        R.tr = trade_params(h);
        R.obs0 = h->tr.obs;
    }
    R.ro_states = net->ro_states; R.ro_hist = net->ro_hist; R.ro_act = net->ro_act; R.ro_val = net->ro_val; R.ro_rew = net->ro_rew;
    R.ro_mask = net->ro_mask; R.ro_y = net->ro_y; R.ro_adv = net->ro_adv; R.ro_boot = net->ro_boot; R.ro_nhist = net->ro_nhist;
    R.counter_base = net->d_counter; R.seed = h->cfg.seed; R.env_off = (uint32_t)h->cfg.env_id_offset;
    const bool gae = net->cfg.gae_lambda < 1.0f;
    R.gamma = net->cfg.gamma; R.lam = gae ? net->cfg.gae_lambda : 1.0f; R.clip_lo = gae ? 0.f : -2.f; R.clip_hi = gae ? 0.f : 2.f;
    R.ep_total = h->ep_total; R.ep_len = h->ep_len; R.ep_steps = h->ep_steps; R.ep_rec = h->ep_rec; R.ep_count = h->ep_count;
    R.ep_cap = h->ep_capacity;
    hipStream_t st = h->stream;
    R.ts = net->d_ts; R.ts_n = net->d_ts_n;
    if (net->d_ts_n) FNET_HIP(net, hipMemsetAsync(net->d_ts_n, 0, sizeof(int), st));
    FNET_HIP(net, hipMemsetAsync(h->done_count, 0, sizeof(int32_t), st));      // the last step's done list is built inside the kernel
    FNET_HIP(net, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_ro_args), &R, sizeof(RolloutArgs), (size_t)net->arg_slot * sizeof(RolloutArgs),
                                         hipMemcpyHostToDevice, st));      // pageable source: staged before the call returns
    FNET_HIP(net, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_flat_args), &R.f, sizeof(FlatArgs), (size_t)net->arg_slot * sizeof(FlatArgs),
                                         hipMemcpyHostToDevice, st));
    (void)hipGetLastError();      // the symbol lookup may probe other ordinals and leave a stale error on this thread
    const dim3 grid((h->E + G - 1) / G);
    hipLaunchKernelGGL(kern_fn, grid, dim3(FNT), lds_bytes, st, net->arg_slot);
    FNET_HIP(net, hipGetLastError());
    net->ws_resident = keep ? 1 : 0;
    return GRL_OK;
}

}  // namespace grl

using namespace grl;

extern "C" {

int grl_fnet_config_default(grl_fnet_config *cfg) {
    if (!cfg) return GRL_E_INVALID;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(grl_fnet_config);
    cfg->static_size = 2; cfg->temporal_size = 2; cfg->rnn_length = 5; cfg->num_actions = 1;     // train_paac_solow.py:96-129
    cfg->rnn_hidden = 32; cfg->static_hidden = 32; cfg->max_samples = 4096 * 20;
    cfg->scale = 100.f; cfg->clip_norm = 40.f; cfg->gamma = 0.99f; cfg->mu_bound = 5.f; cfg->gae_lambda = 1.f;
    return GRL_OK;
}

int grl_fnet_create(grl_handle *h, const grl_fnet_config *cfg, grl_fnet **out) {
    if (!h || !cfg || !out) return GRL_E_INVALID;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(grl_fnet_config)) return fail(h, GRL_E_INVALID, "grl_fnet_create: config size mismatch");
    if (cfg->rnn_hidden != FH || cfg->static_hidden != FH) return fail(h, GRL_E_INVALID, "grl_fnet_create: hidden sizes must be 32 (the reference defaults)");
    if (cfg->temporal_size < 1 || cfg->temporal_size > MAXD || cfg->static_size < 1 || cfg->static_size > MAXS0 || cfg->num_actions < 1 ||
        cfg->num_actions > MAXA || cfg->rnn_length < 1 || cfg->rnn_length > 32 || cfg->max_samples < 1)
        return fail(h, GRL_E_INVALID, "grl_fnet_create: size out of range");
    if (!(cfg->gae_lambda > 0.f && cfg->gae_lambda <= 1.f)) return fail(h, GRL_E_INVALID, "grl_fnet_create: gae_lambda must be in (0, 1]");
    hipSetDevice(h->cfg.device_id);
    grl_fnet *n = new grl_fnet();
    n->h = h; n->cfg = *cfg;
    n->off = make_offsets(cfg->temporal_size, cfg->static_size, cfg->num_actions);
    n->wso = ws_offsets(cfg->rnn_length, cfg->num_actions);
    n->slab_blocks = 512;      // two workgroups per CU (LDS 75 KB each)
    size_t ms = cfg->max_samples;
    const int A = cfg->num_actions, S0 = cfg->static_size, D = cfg->temporal_size, T = cfg->rnn_length;
    int rc = GRL_OK;
    auto Al = [&](float **p, size_t cnt) { if (rc == GRL_OK) rc = falloc(n, p, cnt); };
    Al(&n->params, n->off.total); Al(&n->paramsT, n->off.total); Al(&n->grads, n->off.total); Al(&n->adam_m, n->off.total); Al(&n->adam_v, n->off.total);
    Al(&n->ws, (ms + 63) / 64 * 64 * n->wso.total); Al(&n->slab, (size_t)n->slab_blocks * n->off.total); Al(&n->stats, 8);
    Al(&n->d_states, ms * S0); Al(&n->d_hist, ms * T * D); Al(&n->d_act, ms * A); Al(&n->d_adv, ms); Al(&n->d_y, ms);
    Al(&n->mu, ms * A); Al(&n->sigma, ms * A); Al(&n->vs, ms);
    if (rc == GRL_OK) rc = falloc(n, &n->stats64, 4 + kSumsqBlocks);      // loss sums (2 used of 4), then the squares' partial sums
    if (rc == GRL_OK) rc = falloc(n, &n->d_counter, 4);
    n->ro_graph = nullptr; n->ro_graph_T = 0; n->ro_graph_ep = false;
    {   // GRL_FLAT_ROLLOUT=graph keeps the launch-per-stage rollout (captured into a hipGraph) for A/B and for the equality tests
        const char *e = getenv("GRL_FLAT_ROLLOUT");
        n->ro_persistent = (e && strcmp(e, "graph") == 0) ? 0 : 1;
        memset(n->ro_attr_set, 0, sizeof(n->ro_attr_set)); n->d_ts = nullptr; n->d_ts_n = nullptr;
        n->keep_activations = 0; n->ws_resident = 0;
        const char *g = getenv("GRL_FLAT_GROUP");
        const int gv = g ? atoi(g) : 0;
        n->ro_group = (gv == 16 || gv == 32 || gv == 64) ? gv : 0;
        n->arg_slot = flat_slot_take();
        const char *f = getenv("GRL_FLAT_FORWARD");
        n->fast_forward = (f && strcmp(f, "layers") == 0) ? 0 : 1;
    }
    n->last_n = 0; n->comm = nullptr; n->comm_world = 1; n->comm_rank = 0;
    n->ar_ev0 = n->ar_ev1 = nullptr; n->ar_pending = 0; n->ar_calls = 0; n->ar_ms_total = 0.0; n->ar_ms_last = 0.f;
    hipError_t e = hipSuccess;
    if (rc == GRL_OK) e = hipFuncSetAttribute((const void *)flat_forward_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLAT_LDS_BYTES);
    if (rc == GRL_OK && e == hipSuccess)
        e = hipFuncSetAttribute((const void *)flat_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(FLAT_LDS_BYTES + (size_t)(MAXD + FH) * 3 * FH * sizeof(float)));
    if (rc == GRL_OK && e == hipSuccess)
        e = hipFuncSetAttribute((const void *)flat_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLAT_LDS_BYTES);
    if (rc == GRL_OK && e == hipSuccess)
        e = hipFuncSetAttribute((const void *)flat_forward_fast_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_lds_bytes(MAXS0));
    if (rc == GRL_OK && e == hipSuccess)
        e = hipFuncSetAttribute((const void *)flat_forward_fast_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ff_lds_bytes(MAXS0));
    if (rc == GRL_OK && e == hipSuccess)
        e = hipFuncSetAttribute((const void *)flat_backward_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FB2_LDS_BYTES);
    if (rc == GRL_OK && e != hipSuccess) rc = ffail(n, GRL_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    if (rc != GRL_OK) {
        fail(h, rc, "grl_fnet_create: " + n->err);
        grl_fnet_destroy(n);
        return rc;
    }
    hipStreamSynchronize(h->stream);
    *out = n;
    return GRL_OK;
}

int grl_fnet_destroy(grl_fnet *n) {
    if (!n) return GRL_OK;
    grl_sync_for_destroy(n->h);      // the handle may have been destroyed first (finaliser order of a host binding)
    if (n->ro_graph) (void)hipGraphExecDestroy(n->ro_graph);
    if (n->comm) {
        ncclCommDestroy((ncclComm_t)n->comm);
        (void)hipGetLastError();   // RCCL teardown may leave a stale HIP error on this thread
    }
    if (n->ar_ev0) { hipEventDestroy(n->ar_ev0); hipEventDestroy(n->ar_ev1); }
    flat_slot_release(n->arg_slot);
    for (void *p : n->allocs) hipFree(p);
    delete n;
    return GRL_OK;
}

const char *grl_fnet_last_error(const grl_fnet *n) { return n ? n->err.c_str() : ""; }
int64_t grl_fnet_num_params(const grl_fnet *n) { return n ? n->off.total : 0; }

static int fcopy_flat(grl_fnet *n, float *dev, float *host, int64_t cnt, bool to_dev) {
    if (!n || !host) return GRL_E_INVALID;
    if (cnt != n->off.total) return ffail(n, GRL_E_SIZE, "expected " + std::to_string(n->off.total) + " floats");
    hipSetDevice(n->h->cfg.device_id);
    FNET_HIP(n, hipStreamSynchronize(n->h->stream));
    if (to_dev) FNET_HIP(n, hipMemcpy(dev, host, cnt * 4, hipMemcpyHostToDevice));
    else FNET_HIP(n, hipMemcpy(host, dev, cnt * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}
int grl_fnet_set_params(grl_fnet *n, const float *host, int64_t cnt) {
    if (n) n->ws_resident = 0;
    return fcopy_flat(n, n ? n->params : nullptr, (float *)host, cnt, true);
}

int grl_fnet_set_keep_activations(grl_fnet *net, int32_t on) {
    if (!net) return GRL_E_INVALID;
    net->keep_activations = on ? 1 : 0;
    return GRL_OK;
}
int grl_fnet_get_params(grl_fnet *n, float *host, int64_t cnt) { return fcopy_flat(n, n ? n->params : nullptr, host, cnt, false); }
int grl_fnet_get_grads(grl_fnet *n, float *host, int64_t cnt) { return fcopy_flat(n, n ? n->grads : nullptr, host, cnt, false); }

int grl_fnet_get_optimizer_state(grl_fnet *n, float *m_host, float *v_host, int64_t cnt, int64_t *step_out) {
    if (!n || !m_host || !v_host || !step_out) return GRL_E_INVALID;
    int rc = fcopy_flat(n, n->adam_m, m_host, cnt, false);
    if (rc == GRL_OK) rc = fcopy_flat(n, n->adam_v, v_host, cnt, false);
    *step_out = n->adam_t;
    return rc;
}

int grl_fnet_set_optimizer_state(grl_fnet *n, const float *m_host, const float *v_host, int64_t cnt, int64_t step) {
    if (!n || !m_host || !v_host || step < 0) return GRL_E_INVALID;
    int rc = fcopy_flat(n, n->adam_m, (float *)m_host, cnt, true);
    if (rc == GRL_OK) rc = fcopy_flat(n, n->adam_v, (float *)v_host, cnt, true);
    if (rc == GRL_OK) n->adam_t = (long)step;
    return rc;
}

int grl_fnet_get_action_counter(grl_fnet *net, uint64_t *out) {
    if (!net || !out) return GRL_E_INVALID;
    *out = (uint64_t)net->act_counter;
    return GRL_OK;
}

int grl_fnet_set_action_counter(grl_fnet *net, uint64_t value) {
    if (!net) return GRL_E_INVALID;
    net->act_counter = (unsigned long)value;
    return GRL_OK;
}

static int fdownload(grl_fnet *net, int n, float *mu, float *sigma, float *vs) {
    const int A = net->cfg.num_actions;
    FNET_HIP(net, hipStreamSynchronize(net->h->stream));
    if (mu) FNET_HIP(net, hipMemcpy(mu, net->mu, (size_t)n * A * 4, hipMemcpyDeviceToHost));
    if (sigma) FNET_HIP(net, hipMemcpy(sigma, net->sigma, (size_t)n * A * 4, hipMemcpyDeviceToHost));
    if (vs) FNET_HIP(net, hipMemcpy(vs, net->vs, (size_t)n * 4, hipMemcpyDeviceToHost));
    return GRL_OK;
}

int grl_fnet_predict(grl_fnet *net, int32_t n, const float *states, const float *history, float *mu, float *sigma, float *vs) {
    if (!net || n <= 0 || !states || !history) return ffail(net, GRL_E_INVALID, "grl_fnet_predict: bad argument");
    if (n > net->cfg.max_samples) return ffail(net, GRL_E_SIZE, "grl_fnet_predict: n exceeds max_samples");
    hipSetDevice(net->h->cfg.device_id);
    const grl_fnet_config &c = net->cfg;
    hipStream_t st = net->h->stream;
    FNET_HIP(net, hipMemcpyAsync(net->d_states, states, (size_t)n * c.static_size * 4, hipMemcpyHostToDevice, st));
    FNET_HIP(net, hipMemcpyAsync(net->d_hist, history, (size_t)n * c.rnn_length * c.temporal_size * 4, hipMemcpyHostToDevice, st));
    int rc = launch_forward(net, n, net->d_states, net->d_hist, net->mu, net->sigma, net->vs, false);
    if (rc) return rc;
    return fdownload(net, n, mu, sigma, vs);
}

static int check_env(grl_fnet *net) {
    grl_handle *h = net->h;
    if (h->cfg.env_kind == GRL_ENV_SOLOW) {
        if (net->cfg.static_size != 2 || net->cfg.temporal_size != 2 || net->cfg.rnn_length != h->cfg.rnn_length || net->cfg.num_actions != 1)
            return ffail(net, GRL_E_INVALID, "a Solow handle needs a net with static=temporal=2, num_actions=1 and the handle's rnn_length");
    } else if (h->cfg.env_kind == GRL_ENV_TRADE) {
        const int S = 1 + 2 * h->cfg.n_assets;      // INPUT_SIZE = TEMPORAL_SIZE = 1+2n, NUM_ACTIONS = n (train_trade.py:38-40)
        if (net->cfg.static_size != S || net->cfg.temporal_size != S || net->cfg.num_actions != h->cfg.n_assets ||
            net->cfg.rnn_length != h->cfg.rnn_length)
            return ffail(net, GRL_E_INVALID, "a TradeAR1 handle needs a net with static=temporal=1+2n, num_actions=n and the handle's rnn_length");
    } else {
        return ffail(net, GRL_E_INVALID, "this call needs a Solow or TradeAR1 handle");
    }
    if (h->E > net->cfg.max_samples) return ffail(net, GRL_E_SIZE, "num_envs exceeds max_samples of the net");
    return GRL_OK;
}

int grl_fnet_predict_env(grl_fnet *net, float *mu, float *sigma, float *vs) {
    if (!net) return GRL_E_INVALID;
    hipSetDevice(net->h->cfg.device_id);
    int rc = check_env(net);
    if (rc) return rc;
    grl_handle *h = net->h;
    if (h->cfg.env_kind == GRL_ENV_SOLOW) rc = launch_forward(net, h->E, h->so.obs, nullptr, net->mu, net->sigma, net->vs, false, h->so.nhist);
    else rc = launch_forward(net, h->E, h->tr.obs, nullptr, net->mu, net->sigma, net->vs, false, h->tr.nhist);
    if (rc) return rc;
    return fdownload(net, h->E, mu, sigma, vs);
}

int grl_fnet_train(grl_fnet *net, int32_t n, const float *states, const float *history, const float *actions, const float *advantages,
                   const float *critic_target, float lr, int32_t apply_update, float *stats_host) {
    if (!net || n <= 0 || !states || !history || !actions || !advantages || !critic_target) return ffail(net, GRL_E_INVALID, "grl_fnet_train: bad argument");
    if (n > net->cfg.max_samples) return ffail(net, GRL_E_SIZE, "grl_fnet_train: n exceeds max_samples");
    hipSetDevice(net->h->cfg.device_id);
    const grl_fnet_config &c = net->cfg;
    hipStream_t st = net->h->stream;
    FNET_HIP(net, hipMemcpyAsync(net->d_states, states, (size_t)n * c.static_size * 4, hipMemcpyHostToDevice, st));
    FNET_HIP(net, hipMemcpyAsync(net->d_hist, history, (size_t)n * c.rnn_length * c.temporal_size * 4, hipMemcpyHostToDevice, st));
    FNET_HIP(net, hipMemcpyAsync(net->d_act, actions, (size_t)n * c.num_actions * 4, hipMemcpyHostToDevice, st));
    FNET_HIP(net, hipMemcpyAsync(net->d_adv, advantages, (size_t)n * 4, hipMemcpyHostToDevice, st));
    FNET_HIP(net, hipMemcpyAsync(net->d_y, critic_target, (size_t)n * 4, hipMemcpyHostToDevice, st));
    return train_device(net, n, net->d_states, net->d_hist, net->d_act, net->d_adv, net->d_y, lr, apply_update, stats_host);
}

int grl_fnet_rollout(grl_fnet *net, int32_t T) {
    if (!net || T <= 0 || T > 1024) return ffail(net, GRL_E_INVALID, "grl_fnet_rollout: bad argument");
    hipSetDevice(net->h->cfg.device_id);
    int rc = check_env(net);
    if (rc) return rc;
    grl_handle *h = net->h;
    const bool solow = h->cfg.env_kind == GRL_ENV_SOLOW;
    const int E = h->E, R = net->cfg.rnn_length, S0 = net->cfg.static_size, A = net->cfg.num_actions;
    if ((long)T * E > net->cfg.max_samples) return ffail(net, GRL_E_SIZE, "grl_fnet_rollout: T*num_envs exceeds max_samples");
    if (!net->ro_states || net->T < T) {
        auto Al = [&](float **p, size_t cnt) { if (rc == GRL_OK) rc = falloc(net, p, cnt); };
        Al(&net->ro_states, (size_t)T * E * S0); Al(&net->ro_act, (size_t)T * E * A);
        Al(&net->ro_envact, (size_t)E * A); Al(&net->ro_val, (size_t)T * E); Al(&net->ro_rew, (size_t)T * E); Al(&net->ro_mask, (size_t)T * E);
        Al(&net->ro_y, (size_t)T * E); Al(&net->ro_adv, (size_t)T * E); Al(&net->ro_boot, E);
        if (solow) Al(&net->ro_hist, (size_t)T * E * R * 2);
        if (rc == GRL_OK) rc = falloc(net, &net->ro_nhist, (size_t)T * E);
        if (rc) return rc;
    }
    net->T = T;
    net->ws_resident = 0;      // set again by a persistent rollout that keeps its activations
    hipStream_t st = h->stream;
    // draw counter of step t = act_counter + t, read by the sample kernel from device memory
    FNET_HIP(net, hipMemsetD32Async((hipDeviceptr_t)net->d_counter, (int)(uint32_t)net->act_counter, 1, st));
    net->act_counter += (unsigned long)T;
    // one workgroup per 64 envs and one workgroup per CU (its LDS): beyond two rounds of workgroups (E > 32 768) the graph of
    // launches takes over.  Measured at the end of round 5, persistent / graph, ms per 20-step rollout: TradeAR1-16 32 768 envs
    // 3.86 / 4.40, 65 536: 7.64 / 7.84; Solow 32 768: 2.01 / 1.93, 65 536: 3.94 / 3.39 (GRL_FLAT_PERSIST_GROUPS moves the limit)
    static const int persist_max_groups = getenv("GRL_FLAT_PERSIST_GROUPS") ? atoi(getenv("GRL_FLAT_PERSIST_GROUPS")) : 512;
    if (net->ro_persistent && (E + 63) / 64 <= persist_max_groups) {
        rc = launch_persistent_rollout(net, T);
        if (rc == GRL_OK) { h->step_in_flight = true; return GRL_OK; }
        if (rc != GRL_E_SIZE) return rc;
        net->ro_persistent = 0;      // does not fit the LDS of a CU (very long rollouts): the graph path from here on
    }
    if (net->ro_graph && net->ro_graph_T == T && net->ro_graph_ep == (h->ep_total != nullptr)) {
        FNET_HIP(net, hipGraphLaunch(net->ro_graph, st));
    } else {
        if (net->ro_graph) { (void)hipGraphExecDestroy(net->ro_graph); net->ro_graph = nullptr; }
        // ~9 tiny operations per step: capture them once into a graph; every argument is a device pointer or a constant
        const bool capturing = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (!capturing) (void)hipGetLastError();
        rc = enqueue_rollout(net, T);
        if (capturing) {
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(st, &graph);
            if (rc == GRL_OK && e == hipSuccess && graph && hipGraphInstantiate(&net->ro_graph, graph, nullptr, nullptr, 0) == hipSuccess) {
                net->ro_graph_T = T;
                net->ro_graph_ep = h->ep_total != nullptr;
            } else {
                net->ro_graph = nullptr;
                (void)hipGetLastError();
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (rc) return rc;
            if (net->ro_graph) FNET_HIP(net, hipGraphLaunch(net->ro_graph, st));
            else if ((rc = enqueue_rollout(net, T))) return rc;      // capture unavailable: plain launches
        } else if (rc) {
            return rc;
        }
    }
    FNET_HIP(net, hipGetLastError());
    h->step_in_flight = true;
    return GRL_OK;
}

// Debug / profiling: attach a timestamp buffer to the persistent rollout and read the stage clock of workgroup 0 after the next
// rollout (100 MHz constant clock ticks at every barrier of the kernel).  n_out = number of stamps written to out[0..max).
int grl_fnet_rollout_stage_times(grl_fnet *net, int64_t *out, int32_t max, int32_t *n_out) {
    if (!net || !n_out) return GRL_E_INVALID;
    hipSetDevice(net->h->cfg.device_id);
    if (!net->d_ts) {
        int rc = falloc(net, &net->d_ts, 4096);
        if (rc == GRL_OK) rc = falloc(net, &net->d_ts_n, 4);
        *n_out = 0;
        return rc;
    }
    FNET_HIP(net, hipStreamSynchronize(net->h->stream));
    int n = 0;
    FNET_HIP(net, hipMemcpy(&n, net->d_ts_n, sizeof(int), hipMemcpyDeviceToHost));
    if (n > max) n = max;
    if (out && n > 0) FNET_HIP(net, hipMemcpy(out, net->d_ts, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    *n_out = n;
    return GRL_OK;
}

int grl_fnet_train_rollout(grl_fnet *net, float lr, float *stats_host) {
    if (!net || !net->ro_states || net->T <= 0) return ffail(net, GRL_E_STATE, "grl_fnet_train_rollout: no rollout to train on");
    hipSetDevice(net->h->cfg.device_id);
    const int n = net->T * net->h->E;
    return train_device(net, n, net->ro_states, nullptr, net->ro_act, net->ro_adv, net->ro_y, lr, 1, stats_host, net->ro_nhist,
                        net->ws_resident != 0);
}

// the gradient step in two halves for callers that exchange gradients themselves (host all-reduce through gloo when no RCCL
// communicator can be formed): _grads leaves the LOCAL mean gradient in the net, nothing is updated
int grl_fnet_train_rollout_grads(grl_fnet *net, float *stats_host) {
    if (!net || !net->ro_states || net->T <= 0) return ffail(net, GRL_E_STATE, "grl_fnet_train_rollout_grads: no rollout to train on");
    hipSetDevice(net->h->cfg.device_id);
    const int n = net->T * net->h->E;
    int rc = train_grads_device(net, n, net->ro_states, nullptr, net->ro_act, net->ro_adv, net->ro_y, net->ro_nhist, net->ws_resident != 0);
    if (rc) return rc;
    return train_apply_device(net, 0.f, 0, 1.0f, stats_host);
}

int grl_fnet_set_grads(grl_fnet *n, const float *host, int64_t cnt) { return fcopy_flat(n, n ? n->grads : nullptr, (float *)host, cnt, true); }

int grl_fnet_apply_grads(grl_fnet *net, float lr, float grad_scale, float *stats_host) {
    if (!net || net->last_n <= 0) return ffail(net, GRL_E_STATE, "grl_fnet_apply_grads: no gradient pass has run yet");
    hipSetDevice(net->h->cfg.device_id);
    return train_apply_device(net, lr, 1, grad_scale, stats_host);
}

int grl_fnet_comm_init(grl_fnet *net, const void *unique_id, size_t bytes, int32_t rank, int32_t world_size) {
    if (!net || !unique_id || bytes != sizeof(ncclUniqueId) || world_size < 1 || rank < 0 || rank >= world_size)
        return ffail(net, GRL_E_INVALID, "grl_fnet_comm_init: bad argument");
    if (net->comm) return ffail(net, GRL_E_STATE, "grl_fnet_comm_init: communicator already attached");
    hipSetDevice(net->h->cfg.device_id);
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, world_size, id, rank);
    (void)hipGetLastError();
    if (r != ncclSuccess) return ffail(net, GRL_E_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    net->comm = (void *)comm; net->comm_world = world_size; net->comm_rank = rank;
    return GRL_OK;
}

int grl_fnet_comm_info(grl_fnet *net, int32_t *count_out, int32_t *user_rank_out, int64_t *allreduce_calls_out,
                       double *allreduce_ms_total_out, float *allreduce_ms_last_out) {
    if (!net) return GRL_E_INVALID;
    int count = 0, urank = -1;
    if (net->comm) {
        ncclResult_t r = ncclCommCount((ncclComm_t)net->comm, &count);
        if (r == ncclSuccess) r = ncclCommUserRank((ncclComm_t)net->comm, &urank);
        (void)hipGetLastError();
        if (r != ncclSuccess) return ffail(net, GRL_E_COMM, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    }
    if (count_out) *count_out = count;
    if (user_rank_out) *user_rank_out = urank;
    if (allreduce_calls_out) *allreduce_calls_out = net->ar_calls;
    if (allreduce_ms_total_out) *allreduce_ms_total_out = net->ar_ms_total;
    if (allreduce_ms_last_out) *allreduce_ms_last_out = net->ar_ms_last;
    return GRL_OK;
}

int grl_fnet_comm_broadcast_params(grl_fnet *net, int32_t root) {
    if (!net || !net->comm) return ffail(net, GRL_E_STATE, "grl_fnet_comm_broadcast_params: no communicator");
    hipSetDevice(net->h->cfg.device_id);
    net->ws_resident = 0;      // the parameters may move (every rank but the root)
    ncclResult_t r = ncclBroadcast(net->params, net->params, (size_t)net->off.total, ncclFloat, root, (ncclComm_t)net->comm, net->h->stream);
    (void)hipGetLastError();
    if (r != ncclSuccess) return ffail(net, GRL_E_COMM, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
    FNET_HIP(net, hipStreamSynchronize(net->h->stream));
    return GRL_OK;
}

int grl_fnet_comm_destroy(grl_fnet *net) {
    if (!net) return GRL_E_INVALID;
    if (net->comm) {
        hipSetDevice(net->h->cfg.device_id);
        hipStreamSynchronize(net->h->stream);
        ncclCommDestroy((ncclComm_t)net->comm);
        (void)hipGetLastError();
        net->comm = nullptr; net->comm_world = 1; net->comm_rank = 0;
    }
    return GRL_OK;
}

int grl_fnet_read_rollout(grl_fnet *net, const char *which, void *host, size_t bytes) {
    if (!net || !which || !host) return GRL_E_INVALID;
    if (!net->ro_states) return ffail(net, GRL_E_STATE, "grl_fnet_read_rollout: no rollout yet");
    hipSetDevice(net->h->cfg.device_id);
    std::string w(which);
    const size_t TE = (size_t)net->T * net->h->E;
    const void *src = nullptr;
    size_t need = TE * 4;
    if (w == "actions") { src = net->ro_act; need = TE * net->cfg.num_actions * 4; }
    else if (w == "values") src = net->ro_val;
    else if (w == "rewards") src = net->ro_rew;
    else if (w == "masks") src = net->ro_mask;
    else if (w == "y") src = net->ro_y;
    else if (w == "adv") src = net->ro_adv;
    else if (w == "boot") { src = net->ro_boot; need = (size_t)net->h->E * 4; }
    else if (w == "states") { src = net->ro_states; need = TE * net->cfg.static_size * 4; }
    else if (w == "histories") { src = net->ro_hist; need = TE * net->cfg.rnn_length * 8; }
    else if (w == "nhist") { src = net->ro_nhist; need = TE * 4; }
    else return ffail(net, GRL_E_INVALID, "grl_fnet_read_rollout: unknown buffer '" + w + "'");
    if (!src) return ffail(net, GRL_E_INVALID, "grl_fnet_read_rollout: '" + w + "' does not exist for this env kind");
    if (need != bytes) return ffail(net, GRL_E_SIZE, "grl_fnet_read_rollout: '" + w + "' needs " + std::to_string(need) + " bytes");
    FNET_HIP(net, hipStreamSynchronize(net->h->stream));
    FNET_HIP(net, hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost));
    return GRL_OK;
}

}  // extern "C"
