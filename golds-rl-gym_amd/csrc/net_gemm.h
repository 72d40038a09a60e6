// fp32 MFMA GEMM building blocks for the policy/value nets (gfx950 only).
//
// All dense contractions of ConvSingleAgentPolicyNetwork (reference
// fed_gym/agents/paac/policy_v_network.py:14-59) -- conv2, conv3 as implicit GEMMs over NHWC
// activations, the dense stack, and their data/weight gradients -- go through two kernels:
//
//   gemm_rowk<BM,BN,WGM,WGN,B_ROWK>   C[M,N] = A[M,K] * B      (A rows gathered, K contiguous in runs)
//        B_ROWK=false: B = W[K,N]  (forward)          B_ROWK=true: B = W[N,K]^T  (data gradient)
//   gemm_tn<BM,BN,WGM,WGN>            C[I,J] = sum_m A[m,I] * B[m,J]   (weight gradient, split over m)
//
// built on v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the fp32 peak of 157 TF; the
// reference computes in float32, so no reduced-precision MFMA is used).  256 threads = 4 waves,
// BK = 32, operands staged through LDS with register prefetch of the next K-tile so the global
// loads fly under the 64-cycle MFMAs.  LDS layouts are chosen so the one-float-per-lane fragment
// reads (lane l: row l&31, k = l>>5) are bank-conflict free: "RowK" tiles are [rows][33], "KRow"
// tiles are [32][rows].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace grl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------- gather descriptors
// Row r of the (virtual) operand = output pixel (n, qy, qx); reduction index k = ((ty*P + tx)*C + c)
// reads src[n][iy0+ty][ix0+tx][c] with (iy0, ix0) = (qy*SY + OY0, qx*SX + OX0) in an NHWC tensor
// [n][H][W][C].  CHECK=true zero-fills taps outside the image (transposed convolutions).
template <int QH, int QW, int SY, int SX, int OY0, int OX0, int S, int P, int C, int H, int W, bool CHECK>
struct ConvGather {
    static constexpr int kPPS = QH * QW;
    static constexpr int kC = C;
    const float *base;
    int rows;
    __device__ __forceinline__ int K() const { return S * P * C; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        int n = r / kPPS;
        int q = r - n * kPPS;
        int qy = q / QW, qx = q - qy * QW;
        iy0 = qy * SY + OY0;
        ix0 = qx * SX + OX0;
        off = (long)n * (H * W * C) + ((long)iy0 * W + ix0) * C;
    }
    // k0: multiple of 32 (or of the column-tile width for gemm_tn) -> offset of that run and its tap
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        int t = k0 / C;
        int c0 = k0 - t * C;
        ty = t / P;
        tx = t - ty * P;
        toff = (ty * W + tx) * C + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int ix0, int ty, int tx) const {
        return !CHECK || ((unsigned)(iy0 + ty) < (unsigned)H && (unsigned)(ix0 + tx) < (unsigned)W);
    }
};

struct DenseRows {   // plain row-major [rows][ld], reduction length k
    const float *base;
    int rows, ld, k;
    __device__ __forceinline__ int K() const { return k; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        off = (long)r * ld;
        iy0 = ix0 = 0;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        toff = k0;
        ty = tx = 0;
    }
    __device__ __forceinline__ bool ok(int, int, int, int) const { return true; }
};

// ---------------------------------------------------------------------------- epilogues
enum { ACT_NONE = 0, ACT_RELU = 1 };

struct EpiBiasAct {   // C[r][c] = act(v + bias[c])
    float *C;
    int ldc;
    const float *bias;
    int act;
    __device__ __forceinline__ void operator()(int r, int c, float v) const {
        v += bias[c];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        C[(long)r * ldc + c] = v;
    }
};

struct EpiGrad {   // dX[r][c] = (v [+ dX[r][c]]) * (fwd[r][c] > 0 if mask)
    float *dX;
    int ld;
    const float *fwd;   // forward activation of the same tensor (post-ReLU), or nullptr
    int accumulate;
    __device__ __forceinline__ void operator()(int r, int c, float v) const {
        long i = (long)r * ld + c;
        if (accumulate) v += dX[i];
        if (fwd) v = fwd[i] > 0.f ? v : 0.f;
        dX[i] = v;
    }
};

// data gradient of conv2 for one parity class (py,px): row (n, yh, xh) -> pixel (2yh+py, 2xh+px) of
// the [n][20][20][32] tensor
struct EpiGradStride2 {
    float *dX;
    const float *fwd;
    int py, px;
    __device__ __forceinline__ void operator()(int r, int c, float v) const {
        int n = r / 100, q = r - n * 100;
        int yh = q / 10, xh = q - yh * 10;
        long i = (((long)n * 20 + 2 * yh + py) * 20 + 2 * xh + px) * 32 + c;
        dX[i] = fwd[i] > 0.f ? v : 0.f;
    }
};

// ---------------------------------------------------------------------------- C = A(rowk) * B
template <int BM, int BN, int WGM, int WGN, bool B_ROWK, class AG, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_rowk(AG ag, const float *__restrict__ Bw, int ldb, int N, Epi epi) {
    constexpr int BK = 32, LDA = 33;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int NA = BM / 32;                       // float4 per thread for the A tile
    constexpr int NB = BN / 32;                       // float4 per thread for the B tile
    static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1, "4 waves");
    __shared__ float As[BM * LDA];
    __shared__ float Bs[B_ROWK ? BN * LDA : BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int M = ag.rows, K = ag.K();

    // A rows owned by this thread: r = (tid>>3) + 32*i, k4 = tid&7
    long aoff[NA];
    int ayx[NA];
    bool arow_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int r = m0 + (tid >> 3) + 32 * i;
        arow_ok[i] = r < M;
        int iy0, ix0;
        ag.row(arow_ok[i] ? r : 0, aoff[i], iy0, ix0);
        ayx[i] = (iy0 << 16) | (ix0 & 0xFFFF);
    }
    float4 ra[NA], rb[NB];
#define GRL_LOAD_TILE(kt_)                                                                                         \
    {                                                                                                              \
        int toff, ty, tx;                                                                                          \
        ag.tap((kt_) * BK, toff, ty, tx);                                                                          \
        _Pragma("clang loop unroll(full)") for (int i = 0; i < NA; ++i) {                                                           \
            int iy0 = ayx[i] >> 16, ix0 = (int)(int16_t)(ayx[i] & 0xFFFF);                                         \
            bool v = arow_ok[i] && ag.ok(iy0, ix0, ty, tx);                                                        \
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
            if (v) t4 = *reinterpret_cast<const float4 *>(ag.base + aoff[i] + toff + (tid & 7) * 4);               \
            ra[i].x = t4.x; ra[i].y = t4.y; ra[i].z = t4.z; ra[i].w = t4.w;                                        \
        }                                                                                                          \
        if (B_ROWK) { /* W[j][k], k contiguous: rows j = n0 + (tid>>3) + 32*i */                                   \
            _Pragma("clang loop unroll(full)") for (int i = 0; i < NB; ++i) {                                                       \
                int j = n0 + (tid >> 3) + 32 * i;                                                                  \
                rb[i] = *reinterpret_cast<const float4 *>(Bw + (long)j * ldb + (kt_) * BK + (tid & 7) * 4);        \
            }                                                                                                      \
        } else { /* W[k][n], n contiguous: BN/4 float4 per k row */                                                \
            _Pragma("clang loop unroll(full)") for (int i = 0; i < NB; ++i) {                                                       \
                int idx = tid + 256 * i;                                                                           \
                int kk = idx / (BN / 4), j4 = idx - kk * (BN / 4);                                                 \
                rb[i] = *reinterpret_cast<const float4 *>(Bw + (long)((kt_) * BK + kk) * ldb + n0 + j4 * 4);       \
            }                                                                                                      \
        }                                                                                                          \
    }
#define GRL_STORE_TILE()                                                                                           \
    {                                                                                                              \
        _Pragma("clang loop unroll(full)") for (int i = 0; i < NA; ++i) {                                                           \
            float *d = As + ((tid >> 3) + 32 * i) * LDA + (tid & 7) * 4;                                           \
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;                                        \
        }                                                                                                          \
        if (B_ROWK) {                                                                                              \
            _Pragma("clang loop unroll(full)") for (int i = 0; i < NB; ++i) {                                                       \
                float *d = Bs + ((tid >> 3) + 32 * i) * LDA + (tid & 7) * 4;                                       \
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;                                    \
            }                                                                                                      \
        } else {                                                                                                   \
            _Pragma("clang loop unroll(full)") for (int i = 0; i < NB; ++i)                                                         \
                *reinterpret_cast<float4 *>(Bs + (tid + 256 * i) * 4) = rb[i];                                     \
        }                                                                                                          \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = K / BK;
    GRL_LOAD_TILE(0)
    const int lr = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        GRL_STORE_TILE()
        __syncthreads();
        if (kt + 1 < nk) GRL_LOAD_TILE(kt + 1)
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = As[(wm * WM + a * 32 + lr) * LDA + kk + lk];
#pragma unroll
            for (int b = 0; b < TN; ++b)
                bf[b] = B_ROWK ? Bs[(wn * WN + b * 32 + lr) * LDA + kk + lk] : Bs[(kk + lk) * BN + wn * WN + b * 32 + lr];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = n0 + wn * WN + b * 32 + lr;
                if (row < M && col < N) epi(row, col, acc[a][b][r]);
            }
#undef GRL_LOAD_TILE
#undef GRL_STORE_TILE
}

// ---------------------------------------------------------------------------- C[I,J] = A^T * B over rows m
// A element (m, i) = gathered patch element i of row m (same descriptors as above, i plays the role
// of k); B = dY[m][J] dense.  Block (bi, bj, chunk) reduces rows [chunk*mc, (chunk+1)*mc) and writes
// its partial tile to slab[chunk][I][J]; a follow-up kernel sums the slabs in a fixed order
// (deterministic, unlike float atomics).
template <int BM, int BN, int WGM, int WGN, class AG>
__global__ __launch_bounds__(256) void gemm_tn(AG ag, const float *__restrict__ dY, int J, int mc, float *__restrict__ slab) {
    constexpr int BK = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr int A4 = BM / 4, B4 = BN / 4;           // float4 per reduction row
    constexpr int NA = BK * A4 / 256, NB = BK * B4 / 256;
    static_assert(WGM * WGN == 4 && NA >= 1 && NB >= 1, "tile too small");
    __shared__ float As[BK * BM];
    __shared__ float Bs[BK * BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int i0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int M = ag.rows, I = ag.K();
    const int mbeg = blockIdx.z * mc;
    const int mend = min(M, mbeg + mc);

    int toff, ty, tx;
    ag.tap(i0, toff, ty, tx);   // the BM-wide column run lies inside one tap row (checked on the host)

    float4 ra[NA], rb[NB];
#define GRL_LOAD_TILE(mt_)                                                                                         \
    {                                                                                                              \
        _Pragma("clang loop unroll(full)") for (int i = 0; i < NA; ++i) {                                                           \
            int idx = tid + 256 * i;                                                                               \
            int kk = idx / A4, c4 = idx - kk * A4;                                                                 \
            int m = (mt_) + kk;                                                                                    \
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                            \
            if (m < mend) {                                                                                        \
                long off; int iy0, ix0;                                                                            \
                ag.row(m, off, iy0, ix0);                                                                          \
                if (ag.ok(iy0, ix0, ty, tx)) v = *reinterpret_cast<const float4 *>(ag.base + off + toff + c4 * 4); \
            }                                                                                                      \
            ra[i] = v;                                                                                             \
        }                                                                                                          \
        _Pragma("clang loop unroll(full)") for (int i = 0; i < NB; ++i) {                                                           \
            int idx = tid + 256 * i;                                                                               \
            int kk = idx / B4, c4 = idx - kk * B4;                                                                 \
            int m = (mt_) + kk;                                                                                    \
            rb[i] = m < mend ? *reinterpret_cast<const float4 *>(dY + (long)m * J + j0 + c4 * 4)                   \
                             : make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        }                                                                                                          \
    }
#define GRL_STORE_TILE()                                                                                           \
    {                                                                                                              \
        _Pragma("clang loop unroll(full)") for (int i = 0; i < NA; ++i) *reinterpret_cast<float4 *>(As + (tid + 256 * i) * 4) = ra[i]; \
        _Pragma("clang loop unroll(full)") for (int i = 0; i < NB; ++i) *reinterpret_cast<float4 *>(Bs + (tid + 256 * i) * 4) = rb[i]; \
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    if (mbeg < mend) {
        GRL_LOAD_TILE(mbeg)
        for (int mt = mbeg; mt < mend; mt += BK) {
            GRL_STORE_TILE()
            __syncthreads();
            if (mt + BK < mend) GRL_LOAD_TILE(mt + BK)
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float af[TM], bf[TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) af[a] = As[(kk + lk) * BM + wm * WM + a * 32 + lr];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[b] = Bs[(kk + lk) * BN + wn * WN + b * 32 + lr];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    float *out = slab + (long)blockIdx.z * I * J;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = i0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = j0 + wn * WN + b * 32 + lr;
                if (row < I && col < J) out[(long)row * J + col] = acc[a][b][r];
            }
#undef GRL_LOAD_TILE
#undef GRL_STORE_TILE
}

// dst[i] (+)= sum_c slab[c][i]   (fixed order -> bitwise reproducible)
__global__ void slab_reduce_kernel(const float *__restrict__ slab, int chunks, long n, float *__restrict__ dst, int accumulate) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? dst[i] : 0.f;
    for (int c = 0; c < chunks; ++c) s += slab[(long)c * n + i];
    dst[i] = s;
}

// column sums of dY[M][J] over row chunks -> slab[chunk][J]   (bias gradients)
// 256 threads: min(J,256) column lanes x (256/J) row lanes, row lanes combined through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ dY, int M, int J, int mc, float *__restrict__ slab) {
    __shared__ float red[256];
    const int jw = J < 256 ? J : 256;            // J is a power of two >= 32 or a multiple of 256
    const int rl = 256 / jw;
    const int jl = threadIdx.x % jw, rsub = threadIdx.x / jw;
    const int j = blockIdx.x * jw + jl;
    const int mbeg = blockIdx.y * mc, mend = min(M, mbeg + mc);
    float s = 0.f;
    if (j < J)
        for (int m = mbeg + rsub; m < mend; m += rl) s += dY[(long)m * J + j];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rsub == 0 && j < J) {
        for (int r = 1; r < rl; ++r) s += red[r * jw + jl];
        slab[(long)blockIdx.y * J + j] = s;
    }
}

}  // namespace grl
