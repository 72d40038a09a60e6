// Device functions of the rollout math shared by the stand-alone kernels (rollout_math.hip, api.hip) and the persistent flat PAAC
// rollout (net_flat.hip).
#pragma once
#include "common.h"

namespace grl {

// One column b of the n-step return / advantage / GAE (paac.py:159-172,360-372; a3c/worker.py:232-239,284-294): walks
// t = T-1..0 with a float64 running return like the reference's float64 numpy arrays, including its one float32 product
// gamma*V(s_T) (the bootstrap value is the net's float32 output; pinned by tests/golden/paac_loop.npz).
// r, v, mask point at the column's t = 0 element, `stride` floats apart per step (mask may be null); y / adv likewise with ostride.
__device__ __forceinline__ void returns_column(const float *r, const float *v, const float *mask, size_t stride, float boot, int T,
                                               float gamma, float lam, float scale, float clip_lo, float clip_hi, float *y, float *adv,
                                               size_t ostride) {
    const bool clip = clip_lo < clip_hi;
    const double g = (double)gamma;
    if (lam == 1.0f) {
        double est = 0.0;
        for (int t = T - 1; t >= 0; --t) {
            const size_t i = (size_t)t * stride, o = (size_t)t * ostride;
            float rew = r[i];
            if (clip) rew = rew > clip_hi ? clip_hi : (rew < clip_lo ? clip_lo : rew);   // rescale_reward
            double ge = (t == T - 1) ? (double)(gamma * boot) : g * est;
            if (mask) ge = ge * (double)mask[i];
            est = (double)rew + ge;
            y[o] = (float)est;
            adv[o] = (float)((est - (double)v[i]) / (double)scale);
        }
    } else {
        // GAE: delta_t = r_t + g V_{t+1} - V_t ; A_t = delta_t + g*lam*A_{t+1} ; target = A_t + V_t
        double run = 0.0, vnext = (double)boot;
        const double gl = g * (double)lam;
        for (int t = T - 1; t >= 0; --t) {
            const size_t i = (size_t)t * stride, o = (size_t)t * ostride;
            float rew = r[i];
            if (clip) rew = rew > clip_hi ? clip_hi : (rew < clip_lo ? clip_lo : rew);
            double m = mask ? (double)mask[i] : 1.0;
            double vt = (double)v[i];
            double delta = (double)rew + g * vnext * m - vt;
            run = delta + gl * m * run;
            y[o] = (float)(run + vt);
            adv[o] = (float)(run / (double)scale);
            vnext = vt;
        }
    }
}

// R6 (paac.py:142-157, 331-349) for one env per lane: finished episodes are compacted with a wave ballot + one atomicAdd per wave.
// Every lane of the wave must call it (inactive lanes with active = false).
__device__ __forceinline__ void episodes_account_env(bool active, int e, float reward, bool done, double *__restrict__ total,
                                                     int32_t *__restrict__ len, int64_t *__restrict__ steps,
                                                     grl_episode_record *__restrict__ rec, int32_t *__restrict__ count, int cap) {
    double t = 0.0;
    int32_t l = 0;
    int64_t s = 0;
    bool fin = false;
    if (active) {
        t = total[e] + (double)reward;
        l = len[e] + 1;
        s = steps[e] + 1;
        fin = done;
    }
    const unsigned long long m = __ballot(fin);
    if (m) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (fin) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            if (slot < cap) {
                grl_episode_record r;
                r.step_index = s; r.env = e; r.length = l; r.total_reward = t;
                rec[slot] = r;
            }
            t = 0.0; l = 0;
        }
    }
    if (active) { total[e] = t; len[e] = l; steps[e] = s; }
}

}  // namespace grl
