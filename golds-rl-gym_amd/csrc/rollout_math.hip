// Rollout-side math of the PAAC learner on gfx950: n-step return / advantage / GAE
// (reference fed_gym/agents/paac/paac.py:159-172,360-372; fed_gym/agents/a3c/worker.py:232-239,
// 284-294), reward clipping (paac/actor_learner.py:91-97), the action transforms
// (paac/emulator_runner.py:77-79,113-118; a3c/worker.py:17-34,440-442) and Gaussian draws.
// All are one-pass HBM-bound kernels: 20 B per (t,b) for the returns.
#include "common.h"
#include "rng.h"
#include "rollout_dev.h"

namespace grl {

__global__ __launch_bounds__(256) void returns_kernel(const float *__restrict__ r, const float *__restrict__ v,
                                                      const float *__restrict__ mask, const float *__restrict__ boot,
                                                      int T, int B, float gamma, float lam, float scale, float clip_lo,
                                                      float clip_hi, float *__restrict__ y, float *__restrict__ adv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    returns_column(r + b, v + b, mask ? mask + b : nullptr, B, boot[b], T, gamma, lam, scale, clip_lo, clip_hi, y + b, adv + b, B);
}

int launch_returns(grl_handle *h, const float *r, const float *v, const float *mask, const float *boot, int T, int B,
                   float gamma, float lam, float scale, float clip_lo, float clip_hi, float *y, float *adv) {
    hipLaunchKernelGGL(returns_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream, r, v, mask, boot, T, B, gamma, lam,
                       scale, clip_lo, clip_hi, y, adv);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

// sigmoid (a3c/worker.py:17-34), tanh (:440-442), Swarm norm clip (emulator_runner.py:113-118)
__global__ void transform_kernel(int kind, float *a, int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == GRL_ENV_SWARM) {
        if (i >= rows) return;
        float2 q = reinterpret_cast<float2 *>(a)[i];
        float d = sqrtf(q.x * q.x + q.y * q.y);          // np.linalg.norm(actions, axis=-1)
        if (d >= 1.0f) { q.x /= d; q.y /= d; }           // MAX_MOVE_NORM = 1
        reinterpret_cast<float2 *>(a)[i] = q;
    } else {
        if (i >= rows * cols) return;
        float x = a[i];
        if (kind == GRL_ENV_SOLOW || kind == GRL_ENV_TICKER) {
            // Ticker rows are [choice0, choice1, fraction0, fraction1]: the choices pass, the fractions get the sigmoid
            // (TickerGatedTraderWorker.transform_raw_action, a3c/worker.py:491-494)
            if (kind == GRL_ENV_TICKER && (i & 3) < 2) return;
            float z = expf(-fabsf(x));
            a[i] = x >= 0.f ? 1.0f / (1.0f + z) : z / (1.0f + z);
        } else {
            a[i] = tanhf(x);
        }
    }
}

int launch_transform(grl_handle *h, int kind, float *actions_dev, int rows, int cols) {
    int n = kind == GRL_ENV_SWARM ? rows : rows * cols;
    hipLaunchKernelGGL(transform_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, kind, actions_dev, rows, cols);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

__global__ void randn_kernel(float *dst, size_t n, uint64_t seed, uint32_t rank_off, uint32_t stream, uint64_t counter) {
    size_t pr = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * pr >= n) return;
    uint64_t c = counter + pr;
    double n0, n1;
    normal_pair(rng_block(seed, rank_off, (uint32_t)(c >> 32), RS_USER + stream, (uint32_t)c), n0, n1);
    dst[2 * pr] = (float)n0;
    if (2 * pr + 1 < n) dst[2 * pr + 1] = (float)n1;
}

int launch_randn(grl_handle *h, float *dst, size_t n, uint32_t stream, uint64_t counter) {
    size_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, h->stream, dst, n, h->cfg.seed,
                       (uint32_t)h->cfg.env_id_offset, stream, counter);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

__global__ void iota_kernel(int32_t *dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = i;
}

int launch_iota(grl_handle *h, int32_t *dst, int n) {
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, dst, n);
    GRL_HIP(h, hipGetLastError());
    return GRL_OK;
}

}  // namespace grl
