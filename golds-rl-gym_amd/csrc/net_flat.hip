// FlatPolicyVNetwork on gfx950 (reference fed_gym/agents/paac/policy_v_network.py:194-264 with the GRU
// trunk of fed_gym/agents/a3c/estimators.py:5-28), its loss/backward/optimiser and the flat PAAC rollout
// (fed_gym/agents/paac/paac.py:119-187) for Solow handles and, PAAC-style, TradeAR1 handles (BASELINE config 5:
// obs transform of a3c/worker.py:420-431, tanh actions :440-442, hyper-parameters of scripts/train_trade.py:38-40,119).
//
// The net has ~31k parameters and a few tens of kMAC per sample; BASELINE config 2 (4 096 envs) is
// launch/latency bound, so the design goal is FEW LAUNCHES, not MFMA: one lane per sample,
// one wave per workgroup, the whole forward (5 GRU steps + 13 dense layers) in ONE launch and the whole
// backward in one more.  Activations of the wave's 64 samples are staged in LDS as [feature][lane]
// (row stride 65 => conflict-free both for a lane reading its own column and for the weight-gradient
// dot products along a row); weights are wave-uniform (scalar loads).  Weight gradients are summed per
// workgroup into private slabs and reduced in a fixed order (bitwise reproducible).
#include <string.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/goldsrl_flatnet.h"
#include "common.h"
#include "rng.h"

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

// workspace features per sample ([feature][n] layout): per GRU step {h_prev, r, u, c} then the dense activations
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

struct FlatArgs {
    const float *P;
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
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc[o] = b[o] + sum_i xin[i][lane] * W[i][o]      (W wave-uniform -> scalar loads)
template <int N>
__device__ __forceinline__ void dense_fwd(const float *__restrict__ W, const float *__restrict__ b, const float *xin, int K, int lane,
                                          float (&acc)[N]) {
#pragma unroll
    for (int o = 0; o < N; ++o) acc[o] = b[o];
    for (int i = 0; i < K; ++i) {
        const float a = xin[i * LS + lane];
        const float *w = W + (long)i * N;
#pragma unroll
        for (int o = 0; o < N; ++o) acc[o] = __builtin_fmaf(a, w[o], acc[o]);
    }
}

__device__ __forceinline__ float dense_fwd1(const float *__restrict__ W, float b, const float *xin, int K, int ldw, int col, int lane) {
    float acc = b;
    for (int i = 0; i < K; ++i) acc = __builtin_fmaf(xin[i * LS + lane], W[(long)i * ldw + col], acc);
    return acc;
}

constexpr int FLAT_LDS_ROWS = 96 + 96 + 65 + 64 + 65;     // bufX, bufC, bufA, bufB, bufT
constexpr size_t FLAT_LDS_BYTES = (size_t)FLAT_LDS_ROWS * LS * sizeof(float);

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(64) void flat_forward_kernel(FlatArgs a) {
    extern __shared__ float lds[];
    float *bufX = lds, *bufA = lds + 192 * LS, *bufB = bufA + 65 * LS;
    const int lane = threadIdx.x, s = blockIdx.x * 64 + lane;
    const bool valid = s < a.n;
    const int ss = valid ? s : 0;
    const float *P = a.P;
    const int D = a.D, T = a.T, A = a.A, n = a.n;
    const WsOff wo = ws_offsets(T, A);
    float *ws = a.ws;
    const bool synth = a.nhist != nullptr;
    const float *hrow = synth ? a.states + (long)ss * a.S0 : a.hist + (long)ss * T * D;
    const int hstep = synth ? 0 : D;      // synthesized window: every valid row is the current state
    int nrows = T;
    if (synth) { int nh = a.nhist[ss]; nh = nh < 1 ? 1 : nh; nrows = nh < T ? nh : T; }
    // true_length (a3c/estimators.py:11-15): number of rows with a non-zero entry
    int len = 0;
    for (int t = 0; t < T; ++t) {
        float m = 0.f;
        for (int i = 0; i < D; ++i) m = fmaxf(m, fabsf(t < nrows ? hrow[t * hstep + i] : 0.f));
        len += m > 0.f ? 1 : 0;
    }
    float h[FH];
#pragma unroll
    for (int i = 0; i < FH; ++i) h[i] = 0.f;
    for (int t = 0; t < T; ++t) {
        // GRUCell (TF 1.4): r,u = sigmoid([x,h] Wg + bg); c = tanh([x, r*h] Wc + bc); h' = u*h + (1-u)*c
        for (int i = 0; i < D; ++i) bufA[i * LS + lane] = t < nrows ? hrow[t * hstep + i] : 0.f;
#pragma unroll
        for (int i = 0; i < FH; ++i) bufA[(D + i) * LS + lane] = h[i];
        float g[2 * FH];
        dense_fwd<2 * FH>(P + a.o.gw, P + a.o.gb, bufA, D + FH, lane, g);
#pragma unroll
        for (int i = 0; i < 2 * FH; ++i) g[i] = sigmoidf_(g[i]);
#pragma unroll
        for (int i = 0; i < FH; ++i) bufA[(D + i) * LS + lane] = g[i] * h[i];
        float c[FH];
        dense_fwd<FH>(P + a.o.cw, P + a.o.cb, bufA, D + FH, lane, c);
        const bool act = t < len;       // dynamic_rnn(sequence_length): the state is copied through past the end
#pragma unroll
        for (int i = 0; i < FH; ++i) {
            c[i] = tanhf(c[i]);
            if (ws && valid) {
                ws[(long)(ws_step(t) + i) * n + s] = h[i];
                ws[(long)(ws_step(t) + FH + i) * n + s] = g[i];
                ws[(long)(ws_step(t) + 2 * FH + i) * n + s] = g[FH + i];
                ws[(long)(ws_step(t) + 3 * FH + i) * n + s] = c[i];
            }
            if (act) h[i] = g[FH + i] * h[i] + (1.0f - g[FH + i]) * c[i];
        }
    }
    // rnn_graph_lstm (a3c/estimators.py:24-28): dense_temporal, dense_static x2, concat
#pragma unroll
    for (int i = 0; i < FH; ++i) {
        bufA[i * LS + lane] = h[i];
        if (ws && valid) ws[(long)(wo.hl + i) * n + s] = h[i];
    }
    {
        float d[2 * FH];
        dense_fwd<2 * FH>(P + a.o.tw, P + a.o.tb, bufA, FH, lane, d);
#pragma unroll
        for (int i = 0; i < 2 * FH; ++i) {
            d[i] = fmaxf(d[i], 0.f);
            bufX[i * LS + lane] = d[i];
            if (ws && valid) ws[(long)(wo.dt + i) * n + s] = d[i];
        }
    }
    for (int i = 0; i < a.S0; ++i) bufB[i * LS + lane] = a.states[(long)ss * a.S0 + i];
    {
        float s1[2 * FH];
        dense_fwd<2 * FH>(P + a.o.s1w, P + a.o.s1b, bufB, a.S0, lane, s1);
#pragma unroll
        for (int i = 0; i < 2 * FH; ++i) {
            s1[i] = fmaxf(s1[i], 0.f);
            bufA[i * LS + lane] = s1[i];
            if (ws && valid) ws[(long)(wo.s1 + i) * n + s] = s1[i];
        }
        float s2[FH];
        dense_fwd<FH>(P + a.o.s2w, P + a.o.s2b, bufA, 2 * FH, lane, s2);
#pragma unroll
        for (int i = 0; i < FH; ++i) {
            s2[i] = fmaxf(s2[i], 0.f);
            bufX[(2 * FH + i) * LS + lane] = s2[i];
            if (ws && valid) ws[(long)(wo.s2 + i) * n + s] = s2[i];
        }
    }
    // mu / sigma heads (policy_v_network.py:214-226) -- two hidden layers each
    for (int head = 0; head < 2; ++head) {
        const long w1 = head ? a.o.g1w : a.o.m1w, b1 = head ? a.o.g1b : a.o.m1b, w2 = head ? a.o.g2w : a.o.m2w,
                   b2 = head ? a.o.g2b : a.o.m2b, w3 = head ? a.o.g3w : a.o.m3w, b3 = head ? a.o.g3b : a.o.m3b;
        const int o1 = head ? wo.g1 : wo.m1, o2 = head ? wo.g2 : wo.m2, o3 = head ? wo.sg : wo.tm;
        float l1[2 * FH];
        dense_fwd<2 * FH>(P + w1, P + b1, bufX, 3 * FH, lane, l1);
#pragma unroll
        for (int i = 0; i < 2 * FH; ++i) {
            l1[i] = fmaxf(l1[i], 0.f);
            bufA[i * LS + lane] = l1[i];
            if (ws && valid) ws[(long)(o1 + i) * n + s] = l1[i];
        }
        float l2[FH];
        dense_fwd<FH>(P + w2, P + b2, bufA, 2 * FH, lane, l2);
#pragma unroll
        for (int i = 0; i < FH; ++i) {
            l2[i] = tanhf(l2[i]);
            bufB[i * LS + lane] = l2[i];
            if (ws && valid) ws[(long)(o2 + i) * n + s] = l2[i];
        }
        for (int k = 0; k < A; ++k) {
            float z = dense_fwd1(P + w3, P[b3 + k], bufB, FH, A, k, lane);
            float v = head ? sigmoidf_(z) : tanhf(z);
            if (ws && valid) ws[(long)(o3 + k) * n + s] = v;
            if (valid) {
                if (head) a.sigma[(long)s * A + k] = v + 1e-3f;      // sigmoid(.) + 1e-3 (:223-226)
                else a.mu[(long)s * A + k] = a.bound * v;           // ((ub-lb)/2)*tanh + (lb+ub)/2 with lb = -ub (:219)
            }
        }
    }
    // value head (:237-244): scale * Dense1(tanh(Dense64(x)))
    {
        float v1[2 * FH];
        dense_fwd<2 * FH>(P + a.o.v1w, P + a.o.v1b, bufX, 3 * FH, lane, v1);
#pragma unroll
        for (int i = 0; i < 2 * FH; ++i) {
            v1[i] = tanhf(v1[i]);
            bufA[i * LS + lane] = v1[i];
            if (ws && valid) ws[(long)(wo.v1 + i) * n + s] = v1[i];
        }
        float z = dense_fwd1(P + a.o.v2w, P[a.o.v2b], bufA, 2 * FH, 1, 0, lane);
        float v = a.scale * z;
        if (valid) a.vs[s] = v;
        if (ws && valid) ws[(long)wo.vs * n + s] = v;
    }
}

// ------------------------------------------------------------------------------------------ backward
// dz rows in dzl[N][LS] (all lanes), forward input rows in xin[K][LS].  Adds this group's contribution
// to the block's private gradient slab; optionally produces dx rows for the lane's own sample.
__device__ __forceinline__ void dense_bwd(const float *__restrict__ W, int K, int N, const float *xin, const float *dzl, float *dxl,
                                          bool accumulate_dx, float *gW, float *gb, int lane) {
    __syncthreads();
    for (int w = lane; w < K * N; w += 64) {
        const int i = w / N, o = w - i * N;
        const float *xr = xin + i * LS, *zr = dzl + o * LS;
        float sum = 0.f;
#pragma unroll 8
        for (int q = 0; q < 64; ++q) sum = __builtin_fmaf(xr[q], zr[q], sum);
        gW[w] += sum;
    }
    for (int o = lane; o < N; o += 64) {
        const float *zr = dzl + o * LS;
        float sum = 0.f;
        for (int q = 0; q < 64; ++q) sum += zr[q];
        gb[o] += sum;
    }
    if (dxl) {
        for (int i = 0; i < K; ++i) {
            const float *w = W + (long)i * N;
            float sum = 0.f;
            for (int o = 0; o < N; ++o) sum = __builtin_fmaf(w[o], dzl[o * LS + lane], sum);
            if (accumulate_dx) dxl[i * LS + lane] += sum;
            else dxl[i * LS + lane] = sum;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(64) void flat_backward_kernel(FlatArgs a) {
    extern __shared__ float lds[];
    float *bufX = lds, *bufC = lds + 96 * LS, *bufA = lds + 192 * LS, *bufB = bufA + 65 * LS, *bufT = bufB + 64 * LS;
    const int lane = threadIdx.x;
    const float *P = a.P;
    const int D = a.D, T = a.T, A = a.A, n = a.n;
    const WsOff wo = ws_offsets(T, A);
    const float *ws = a.ws;
    float *G = a.slab + (long)blockIdx.x * a.o.total;
    const int groups = (n + 63) / 64;
    float loss_p = 0.f, loss_c = 0.f;
    for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
        const int s = grp * 64 + lane;
        const bool valid = s < n;
        const int ss = valid ? s : 0;
        auto W = [&](int f) { return valid ? ws[(long)f * n + ss] : 0.f; };
        __syncthreads();
        // ---- loss terms (policy_v_network.py:228-251): mean over (n, A) of -logp*adv; mean over n of 0.25*(vs-y)^2/scale
        const float advs = valid ? a.adv[ss] : 0.f, tgt = valid ? a.y[ss] : 0.f, vsv = W(wo.vs);
        const float dlogp = -advs * a.inv_n / (float)A;
        const float dvs = valid ? 0.5f * (vsv - tgt) / a.scale * a.inv_n : 0.f;
        if (valid) loss_c += 0.25f * (vsv - tgt) * (vsv - tgt) / a.scale;
        for (int head = 0; head < 2; ++head) {
            const long w1 = head ? a.o.g1w : a.o.m1w, b1 = head ? a.o.g1b : a.o.m1b, w2 = head ? a.o.g2w : a.o.m2w,
                       b2 = head ? a.o.g2b : a.o.m2b, w3 = head ? a.o.g3w : a.o.m3w, b3 = head ? a.o.g3b : a.o.m3b;
            const int o1 = head ? wo.g1 : wo.m1, o2 = head ? wo.g2 : wo.m2;
            for (int k = 0; k < A; ++k) {
                const float tm = W(wo.tm + k), sg = W(wo.sg + k);
                const float mu = a.bound * tm, sigma = sg + 1e-3f;
                const float act = valid ? a.actions[(long)ss * A + k] : mu;
                const float diff = act - mu;
                float dz;
                if (head == 0) {
                    dz = dlogp * diff / (sigma * sigma) * a.bound * (1.0f - tm * tm);
                    if (valid) loss_p += -(-0.5f * (diff / sigma) * (diff / sigma) - logf(sigma) - 0.9189385332046727f) * advs;
                } else {
                    dz = dlogp * (diff * diff / (sigma * sigma * sigma) - 1.0f / sigma) * sg * (1.0f - sg);
                }
                bufB[k * LS + lane] = valid ? dz : 0.f;
            }
            for (int i = 0; i < FH; ++i) bufA[i * LS + lane] = W(o2 + i);
            dense_bwd(P + w3, FH, A, bufA, bufB, bufT, false, G + w3, G + b3, lane);
            for (int i = 0; i < FH; ++i) { float m2 = bufA[i * LS + lane]; bufB[i * LS + lane] = bufT[i * LS + lane] * (1.0f - m2 * m2); }
            for (int i = 0; i < 2 * FH; ++i) bufA[i * LS + lane] = W(o1 + i);
            dense_bwd(P + w2, 2 * FH, FH, bufA, bufB, bufT, false, G + w2, G + b2, lane);
            for (int i = 0; i < 2 * FH; ++i) bufB[i * LS + lane] = bufA[i * LS + lane] > 0.f ? bufT[i * LS + lane] : 0.f;
            if (head == 0) {
                for (int i = 0; i < 2 * FH; ++i) bufX[i * LS + lane] = W(wo.dt + i);
                for (int i = 0; i < FH; ++i) bufX[(2 * FH + i) * LS + lane] = W(wo.s2 + i);
            }
            dense_bwd(P + w1, 3 * FH, 2 * FH, bufX, bufB, bufC, head != 0, G + w1, G + b1, lane);
        }
        // ---- value head
        bufB[lane] = dvs * a.scale;
        for (int i = 0; i < 2 * FH; ++i) bufA[i * LS + lane] = W(wo.v1 + i);
        dense_bwd(P + a.o.v2w, 2 * FH, 1, bufA, bufB, bufT, false, G + a.o.v2w, G + a.o.v2b, lane);
        for (int i = 0; i < 2 * FH; ++i) { float v1 = bufA[i * LS + lane]; bufB[i * LS + lane] = bufT[i * LS + lane] * (1.0f - v1 * v1); }
        dense_bwd(P + a.o.v1w, 3 * FH, 2 * FH, bufX, bufB, bufC, true, G + a.o.v1w, G + a.o.v1b, lane);
        // ---- trunk: bufC holds d x96 = [d dense_temporal (64), d dense_static (32)]
        for (int i = 0; i < FH; ++i) bufB[i * LS + lane] = bufX[(2 * FH + i) * LS + lane] > 0.f ? bufC[(2 * FH + i) * LS + lane] : 0.f;
        for (int i = 0; i < 2 * FH; ++i) bufA[i * LS + lane] = W(wo.s1 + i);
        dense_bwd(P + a.o.s2w, 2 * FH, FH, bufA, bufB, bufT, false, G + a.o.s2w, G + a.o.s2b, lane);
        for (int i = 0; i < 2 * FH; ++i) bufB[i * LS + lane] = bufA[i * LS + lane] > 0.f ? bufT[i * LS + lane] : 0.f;
        __syncthreads();
        for (int i = 0; i < a.S0; ++i) bufA[i * LS + lane] = valid ? a.states[(long)ss * a.S0 + i] : 0.f;
        dense_bwd(P + a.o.s1w, a.S0, 2 * FH, bufA, bufB, nullptr, false, G + a.o.s1w, G + a.o.s1b, lane);
        for (int i = 0; i < 2 * FH; ++i) bufB[i * LS + lane] = bufX[i * LS + lane] > 0.f ? bufC[i * LS + lane] : 0.f;
        for (int i = 0; i < FH; ++i) bufA[i * LS + lane] = W(wo.hl + i);
        dense_bwd(P + a.o.tw, FH, 2 * FH, bufA, bufB, bufT, false, G + a.o.tw, G + a.o.tb, lane);
        float dh[FH], keep[FH];
#pragma unroll
        for (int i = 0; i < FH; ++i) dh[i] = bufT[i * LS + lane];
        // ---- GRU, back through time with the sequence-length mask
        const bool synth = a.nhist != nullptr;
        const float *hrow = synth ? a.states + (long)ss * a.S0 : a.hist + (long)ss * T * D;
        const int hstep = synth ? 0 : D;
        int nrows = T;
        if (synth) { int nh = a.nhist[ss]; nh = nh < 1 ? 1 : nh; nrows = nh < T ? nh : T; }
        int len = 0;
        for (int t = 0; t < T; ++t) {
            float m = 0.f;
            for (int i = 0; i < D; ++i) m = fmaxf(m, fabsf(t < nrows ? hrow[t * hstep + i] : 0.f));
            len += (valid && m > 0.f) ? 1 : 0;
        }
        for (int t = T - 1; t >= 0; --t) {
            const bool act = t < len;
            __syncthreads();
            for (int i = 0; i < D; ++i) bufA[i * LS + lane] = (valid && t < nrows) ? hrow[t * hstep + i] : 0.f;
#pragma unroll
            for (int i = 0; i < FH; ++i) {
                const float hp = W(ws_step(t) + i), r = W(ws_step(t) + FH + i), u = W(ws_step(t) + 2 * FH + i), c = W(ws_step(t) + 3 * FH + i);
                const float dhn = act ? dh[i] : 0.f;
                keep[i] = dhn * u;
                bufB[i * LS + lane] = dhn * (1.0f - u) * (1.0f - c * c);             // dz of the candidate
                bufB[(FH + i) * LS + lane] = dhn * (hp - c) * u * (1.0f - u);         // dz of the update gate (kept for later)
                bufA[(D + i) * LS + lane] = r * hp;
            }
            dense_bwd(P + a.o.cw, D + FH, FH, bufA, bufB, bufT, false, G + a.o.cw, G + a.o.cb, lane);
#pragma unroll
            for (int i = 0; i < FH; ++i) {
                const float hp = W(ws_step(t) + i), r = W(ws_step(t) + FH + i);
                const float drh = bufT[(D + i) * LS + lane];
                keep[i] += drh * r;
                bufB[i * LS + lane] = drh * hp * r * (1.0f - r);                     // dz of the reset gate
                bufA[(D + i) * LS + lane] = hp;
            }
            dense_bwd(P + a.o.gw, D + FH, 2 * FH, bufA, bufB, bufT, false, G + a.o.gw, G + a.o.gb, lane);
#pragma unroll
            for (int i = 0; i < FH; ++i)
                if (act) dh[i] = keep[i] + bufT[(D + i) * LS + lane];
        }
    }
    loss_p = wave_sum_f(loss_p);
    loss_c = wave_sum_f(loss_c);
    if (lane == 0) {
        atomicAdd(&a.stats64[0], (double)loss_p);
        atomicAdd(&a.stats64[1], (double)loss_c);
    }
}

__global__ void flat_slab_reduce_kernel(const float *__restrict__ slab, int blocks, long n, float *__restrict__ dst) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < blocks; ++b) s += slab[(long)b * n + i];
    dst[i] = s;
}

__global__ __launch_bounds__(256) void flat_sumsq_kernel(const float *__restrict__ g, long n, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += (double)g[i] * (double)g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void flat_finalize_kernel(const double *__restrict__ sumsq, const double *__restrict__ stats64, float inv_n, float inv_na,
                                     float clip_norm, float *__restrict__ stats) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float norm = (float)sqrt(sumsq[0]);
    float pl = (float)(stats64[0] * (double)inv_na), cl = (float)(stats64[1] * (double)inv_n);
    stats[0] = pl; stats[1] = cl; stats[2] = pl + cl; stats[3] = norm;
    stats[4] = clip_norm > 0.f ? clip_norm / fmaxf(norm, clip_norm) : 1.0f;      // tf.clip_by_global_norm
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
                                   uint32_t env_off, uint32_t counter, int env_kind, float *__restrict__ raw, float *__restrict__ envact) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * A) return;
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
    float *params, *grads, *adam_m, *adam_v;
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
    return a;
}

static int launch_forward(grl_fnet *net, int n, const float *states, const float *hist, float *mu, float *sigma, float *vs, bool save,
                          const int32_t *nhist = nullptr) {
    FlatArgs a = base_args(net, n, states, hist, mu, sigma, vs, save, nhist);
    hipLaunchKernelGGL(flat_forward_kernel, dim3((n + 63) / 64), dim3(64), FLAT_LDS_BYTES, net->h->stream, a);
    FNET_HIP(net, hipGetLastError());
    return GRL_OK;
}

// forward(save) + backward over n device-resident samples; grads <- mean-loss gradient; optional Adam
static int train_device(grl_fnet *net, int n, const float *states, const float *hist, const float *actions, const float *adv, const float *y,
                        float lr, int apply_update, float *stats_host, const int32_t *nhist = nullptr) {
    hipStream_t st = net->h->stream;
    if (n > net->cfg.max_samples) return ffail(net, GRL_E_SIZE, "train: n exceeds max_samples of the net");
    int rc = launch_forward(net, n, states, hist, net->mu, net->sigma, net->vs, true, nhist);
    if (rc) return rc;
    int groups = (n + 63) / 64;
    int blocks = groups < net->slab_blocks ? groups : net->slab_blocks;
    FNET_HIP(net, hipMemsetAsync(net->slab, 0, (size_t)blocks * net->off.total * 4, st));
    FNET_HIP(net, hipMemsetAsync(net->stats64, 0, 4 * sizeof(double), st));
    FlatArgs a = base_args(net, n, states, hist, net->mu, net->sigma, net->vs, true, nhist);
    a.actions = actions; a.adv = adv; a.y = y; a.inv_n = 1.0f / (float)n; a.slab = net->slab; a.stats64 = net->stats64;
    hipLaunchKernelGGL(flat_backward_kernel, dim3(blocks), dim3(64), FLAT_LDS_BYTES, st, a);
    hipLaunchKernelGGL(flat_slab_reduce_kernel, dim3((unsigned)((net->off.total + 255) / 256)), dim3(256), 0, st, net->slab, blocks,
                       net->off.total, net->grads);
    hipLaunchKernelGGL(flat_sumsq_kernel, dim3(1), dim3(256), 0, st, net->grads, net->off.total, net->stats64 + 2);
    hipLaunchKernelGGL(flat_finalize_kernel, dim3(1), dim3(64), 0, st, net->stats64 + 2, net->stats64, 1.0f / (float)n,
                       1.0f / ((float)n * (float)net->cfg.num_actions), net->cfg.clip_norm, net->stats);
    if (apply_update) {
        net->adam_t += 1;
        float lr_t = (float)((double)lr * sqrt(1.0 - pow(0.999, (double)net->adam_t)) / (1.0 - pow(0.9, (double)net->adam_t)));
        hipLaunchKernelGGL(flat_adam_kernel, dim3((unsigned)((net->off.total + 255) / 256)), dim3(256), 0, st, net->params, net->grads,
                           net->adam_m, net->adam_v, net->off.total, net->stats, lr_t);
    }
    FNET_HIP(net, hipGetLastError());
    FNET_HIP(net, hipStreamSynchronize(st));
    if (stats_host) {
        float s[5];
        FNET_HIP(net, hipMemcpy(s, net->stats, sizeof(s), hipMemcpyDeviceToHost));
        stats_host[0] = s[2]; stats_host[1] = s[0]; stats_host[2] = s[1]; stats_host[3] = s[3];
    }
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
    cfg->scale = 100.f; cfg->clip_norm = 40.f; cfg->gamma = 0.99f; cfg->mu_bound = 5.f;
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
    hipSetDevice(h->cfg.device_id);
    grl_fnet *n = new grl_fnet();
    n->h = h; n->cfg = *cfg;
    n->off = make_offsets(cfg->temporal_size, cfg->static_size, cfg->num_actions);
    n->wso = ws_offsets(cfg->rnn_length, cfg->num_actions);
    n->slab_blocks = 256;
    size_t ms = cfg->max_samples;
    const int A = cfg->num_actions, S0 = cfg->static_size, D = cfg->temporal_size, T = cfg->rnn_length;
    int rc = GRL_OK;
    auto Al = [&](float **p, size_t cnt) { if (rc == GRL_OK) rc = falloc(n, p, cnt); };
    Al(&n->params, n->off.total); Al(&n->grads, n->off.total); Al(&n->adam_m, n->off.total); Al(&n->adam_v, n->off.total);
    Al(&n->ws, ms * n->wso.total); Al(&n->slab, (size_t)n->slab_blocks * n->off.total); Al(&n->stats, 8);
    Al(&n->d_states, ms * S0); Al(&n->d_hist, ms * T * D); Al(&n->d_act, ms * A); Al(&n->d_adv, ms); Al(&n->d_y, ms);
    Al(&n->mu, ms * A); Al(&n->sigma, ms * A); Al(&n->vs, ms);
    if (rc == GRL_OK) rc = falloc(n, &n->stats64, 8);
    hipError_t e = hipSuccess;
    if (rc == GRL_OK) e = hipFuncSetAttribute((const void *)flat_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLAT_LDS_BYTES);
    if (rc == GRL_OK && e == hipSuccess)
        e = hipFuncSetAttribute((const void *)flat_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FLAT_LDS_BYTES);
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
    hipSetDevice(n->h->cfg.device_id);
    hipStreamSynchronize(n->h->stream);
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
int grl_fnet_set_params(grl_fnet *n, const float *host, int64_t cnt) { return fcopy_flat(n, n ? n->params : nullptr, (float *)host, cnt, true); }
int grl_fnet_get_params(grl_fnet *n, float *host, int64_t cnt) { return fcopy_flat(n, n ? n->params : nullptr, host, cnt, false); }
int grl_fnet_get_grads(grl_fnet *n, float *host, int64_t cnt) { return fcopy_flat(n, n ? n->grads : nullptr, host, cnt, false); }

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
    if (h->cfg.env_kind == GRL_ENV_SOLOW) rc = launch_forward(net, h->E, h->so.obs, h->so.history, net->mu, net->sigma, net->vs, false);
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
        else if (rc == GRL_OK) rc = falloc(net, &net->ro_nhist, (size_t)T * E);
        if (rc) return rc;
    }
    net->T = T;
    hipStream_t st = h->stream;
    const float *obs = solow ? h->so.obs : h->tr.obs;
    for (int t = 0; t < T; ++t) {
        // states[t] = shared_states, histories[t] = shared_histories (paac.py:132-133).  For TradeAR1 the window is
        // kept as (state, #rows): the worker's history is min(n, rnn) copies of the current state (quirk Q11)
        FNET_HIP(net, hipMemcpyAsync(net->ro_states + (size_t)t * E * S0, obs, (size_t)E * S0 * 4, hipMemcpyDeviceToDevice, st));
        if (solow) {
            FNET_HIP(net, hipMemcpyAsync(net->ro_hist + (size_t)t * E * R * 2, h->so.history, (size_t)E * R * 8, hipMemcpyDeviceToDevice, st));
            rc = launch_forward(net, E, obs, h->so.history, net->mu, net->sigma, net->ro_val + (size_t)t * E, false);
        } else {
            FNET_HIP(net, hipMemcpyAsync(net->ro_nhist + (size_t)t * E, h->tr.nhist, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
            rc = launch_forward(net, E, obs, nullptr, net->mu, net->sigma, net->ro_val + (size_t)t * E, false, h->tr.nhist);
        }
        if (rc) return rc;
        hipLaunchKernelGGL(flat_sample_kernel, dim3((E * A + 255) / 256), dim3(256), 0, st, net->mu, net->sigma, E, A, h->cfg.seed,
                           (uint32_t)h->cfg.env_id_offset, (uint32_t)net->act_counter, h->cfg.env_kind, net->ro_act + (size_t)t * E * A,
                           net->ro_envact);
        net->act_counter += 1;
        rc = solow ? solow_launch_step(h, net->ro_envact) : trade_launch_step(h, net->ro_envact);
        if (rc) return ffail(net, rc, h->err);
        FNET_HIP(net, hipMemcpyAsync(net->ro_rew + (size_t)t * E, h->reward, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(flat_mask_kernel, dim3((E + 255) / 256), dim3(256), 0, st, h->done, E, net->ro_mask + (size_t)t * E);
    }
    rc = solow ? launch_forward(net, E, obs, h->so.history, net->mu, net->sigma, net->ro_boot, false)
               : launch_forward(net, E, obs, nullptr, net->mu, net->sigma, net->ro_boot, false, h->tr.nhist);
    if (rc) return rc;
    // rewards clipped to [-2, 2] (paac.py:145), masked n-step return (paac.py:167-172), adv / scale (paac.py:177)
    if ((rc = launch_returns(h, net->ro_rew, net->ro_val, net->ro_mask, net->ro_boot, T, E, net->cfg.gamma, 1.0f, net->cfg.scale, -2.f, 2.f,
                             net->ro_y, net->ro_adv)))
        return ffail(net, rc, h->err);
    FNET_HIP(net, hipGetLastError());
    h->step_in_flight = true;
    return GRL_OK;
}

int grl_fnet_train_rollout(grl_fnet *net, float lr, float *stats_host) {
    if (!net || !net->ro_states || net->T <= 0) return ffail(net, GRL_E_STATE, "grl_fnet_train_rollout: no rollout to train on");
    hipSetDevice(net->h->cfg.device_id);
    const int n = net->T * net->h->E;
    if (net->h->cfg.env_kind == GRL_ENV_SOLOW)
        return train_device(net, n, net->ro_states, net->ro_hist, net->ro_act, net->ro_adv, net->ro_y, lr, 1, stats_host);
    return train_device(net, n, net->ro_states, nullptr, net->ro_act, net->ro_adv, net->ro_y, lr, 1, stats_host, net->ro_nhist);
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
