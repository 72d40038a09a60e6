// microbenchmark for VERDICT r3 #5b: the small dense stack's forward as ONE row-tile kernel.
//   d2 = relu(d1 W1 + b1)          d1 [M x 512], W1 [512 x 256]
//   pv = relu(d2 W2 + b2)          W2 [256 x 1024]   (pol1 | v1 side by side, as the product runs them since round 4)
// unfused (what the product does): two launches of the 8-wave 128 x 128 three-product kernel (gemm_rowk's loop, here rowk_w8 with a
// bias + ReLU epilogue); the second reads d2 back from memory (84 MB per 81 920-row chunk) and splits it again.
// fused: one workgroup of 8 waves owns 128 rows.  Phase 1 is the same K loop over 512 for the tile's two 128-column halves of d2; its
// epilogue writes d2 to memory (the gradient step needs it) AND, already split into the two fp16 planes, into LDS in the fragment
// layout (8 K-tiles x 2 planes x 128 rows x 32 halfs = 128 KB).  Phase 2 walks the eight 128-column tiles of pv: per K-tile only the
// WEIGHT tile is staged (global -> split -> LDS), the A fragments come straight from the resident planes.  160 KB of LDS: ONE workgroup
// per CU, two waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 fused_dense.hip -o fused_dense
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void hsplit2(float x, float y, unsigned &h, unsigned &l) {
    const f32x2 xy = {x, y};
    const f16x2 hv = __builtin_convertvector(xy, f16x2);
    const f32x2 r = {(x - (float)hv[0]) * 2048.f, (y - (float)hv[1]) * 2048.f};
    const f16x2 lv = __builtin_convertvector(r, f16x2);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
__device__ __forceinline__ void hsplit4(float4 v, uint2 &h, uint2 &l) {
    hsplit2(v.x, v.y, h.x, l.x);
    hsplit2(v.z, v.w, h.y, l.y);
}
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

constexpr int BK = 32, LDH = 32, BM = 128, BN = 128, WGM = 4, WGN = 2, WM = 32, WN = 64, TM = 2, TN = 4;

// ---------------------------------------------------------------------------- unfused: C = relu(A Bt^T + bias), 8 waves, 128 x 128 tiles
__global__ __launch_bounds__(512, 4) void dense_w8(const float *__restrict__ A, const float *__restrict__ Bt, const float *__restrict__ bias,
                                                   float *__restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * LDH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    {
        const int nx = gridDim.x, L = by_ * nx + bx_, g = L / (8 * nx);
        if ((g + 1) * 8 <= (int)gridDim.y) {
            const int l = L - g * 8 * nx;
            by_ = g * 8 + (l & 7);
            bx_ = l >> 3;
        }
    }
    const int m0 = by_ * BM, n0 = bx_ * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    const int l16 = lane & 15, kg = lane >> 4;
    float4 ra[2], rb[2];
#define LOADT(kt_)                                                                                                                     \
    {                                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)64 * i * K + (kt_) * BK); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * K + (kt_) * BK); \
    }
    LOADT(0)
    f32x4 acc[TM][TN], acl[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = acl[a][b][r] = 0.f;
    const int rofs = (kg ^ swz(l16)) << 3;
    const int aro = (wm * WM + l16) * LDH + rofs, bro = (wn * WN + l16) * LDH + rofs;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = (trow + 64 * i) * LDH + wo;
            uint2 h, l;
            hsplit4(ra[i], h, l);
            *reinterpret_cast<uint2 *>(&As[0][o]) = h;
            *reinterpret_cast<uint2 *>(&As[1][o]) = l;
            hsplit4(rb[i], h, l);
            *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
            *reinterpret_cast<uint2 *>(&Bs[1][o]) = l;
        }
        __syncthreads();
        LOADT(kt + 1 < nk ? kt + 1 : kt)
        __builtin_amdgcn_sched_barrier(0);
        {
            f16x8 af[TM][2];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p) af[a][p] = *reinterpret_cast<const f16x8 *>(&As[p][aro + a * 16 * LDH]);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                f16x8 bf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(&Bs[p][bro + b * 16 * LDH]);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][1], bf[0], acl[a][b], 0, 0, 0);
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[1], acl[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[0], acc[a][b], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn * WN + b * 16 + l16;
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + a * 16 + 4 * kg + r;
                if (row < M) C[(long)row * N + col] = fmaxf(__builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]) + bv, 0.f);
            }
        }
#undef LOADT
}

// ---------------------------------------------------------------------------- fused: both layers of a 128-row tile in one workgroup
constexpr int K1 = 512, N1 = 256, N2 = 1024;      // d1 -> d2 -> pv
constexpr int kFusedLds = (2 * BM * LDH + 2 * BN * LDH + (N1 / BK) * 2 * BM * LDH) * 2;      // bytes: As, Bs, the resident d2 planes
__global__ __launch_bounds__(512, 2) void dense_fused(const float *__restrict__ A, const float *__restrict__ W1t, const float *__restrict__ b1,
                                                      const float *__restrict__ W2t, const float *__restrict__ b2, float *__restrict__ D2,
                                                      float *__restrict__ PV, int M) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    unsigned short *As0 = lds, *As1 = lds + BM * LDH, *Bs0 = lds + 2 * BM * LDH, *Bs1 = Bs0 + BN * LDH;
    unsigned short *Dp = lds + 2 * BM * LDH + 2 * BN * LDH;      // [ktile][plane][BM * LDH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.x * BM;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);
    const int l16 = lane & 15, kg = lane >> 4;
    const int rofs = (kg ^ swz(l16)) << 3;
    const int aro = (wm * WM + l16) * LDH + rofs, bro = (wn * WN + l16) * LDH + rofs;
    float4 ra[2], rb[2];
    f32x4 acc[TM][TN], acl[TM][TN];
#define ZERO_ACC()                                                                                  \
    _Pragma("unroll") for (int a = 0; a < TM; ++a) _Pragma("unroll") for (int b = 0; b < TN; ++b)   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[a][b][r] = acl[a][b][r] = 0.f;
#define MFMA_TILE(AP0_, AP1_)                                                                                                  \
    {                                                                                                                          \
        f16x8 af[TM][2];                                                                                                       \
        _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                                       \
            af[a][0] = *reinterpret_cast<const f16x8 *>(&(AP0_)[aro + a * 16 * LDH]);                                          \
            af[a][1] = *reinterpret_cast<const f16x8 *>(&(AP1_)[aro + a * 16 * LDH]);                                          \
        }                                                                                                                      \
        _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                                       \
            const f16x8 bf0 = *reinterpret_cast<const f16x8 *>(&Bs0[bro + b * 16 * LDH]);                                      \
            const f16x8 bf1 = *reinterpret_cast<const f16x8 *>(&Bs1[bro + b * 16 * LDH]);                                      \
            _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                                   \
                acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][1], bf0, acl[a][b], 0, 0, 0);                         \
                acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf1, acl[a][b], 0, 0, 0);                         \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf0, acc[a][b], 0, 0, 0);                         \
            }                                                                                                                  \
        }                                                                                                                      \
    }
    // ---- phase 1: d2 tile, two halves of 128 columns
    const float *arow = A + (long)(m0 + trow) * K1 + tk4;
    for (int half = 0; half < 2; ++half) {
        const float *brow = W1t + (long)(half * BN + trow) * K1 + tk4;
        ZERO_ACC()
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = *reinterpret_cast<const float4 *>(arow + (long)64 * i * K1);
            rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * K1);
        }
        for (int kt = 0; kt < K1 / BK; ++kt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (trow + 64 * i) * LDH + wo;
                uint2 h, l;
                hsplit4(ra[i], h, l);
                *reinterpret_cast<uint2 *>(&As0[o]) = h;
                *reinterpret_cast<uint2 *>(&As1[o]) = l;
                hsplit4(rb[i], h, l);
                *reinterpret_cast<uint2 *>(&Bs0[o]) = h;
                *reinterpret_cast<uint2 *>(&Bs1[o]) = l;
            }
            __syncthreads();
            const int ktn = kt + 1 < K1 / BK ? kt + 1 : kt;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = *reinterpret_cast<const float4 *>(arow + (long)64 * i * K1 + ktn * BK);
                rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * K1 + ktn * BK);
            }
            __builtin_amdgcn_sched_barrier(0);
            MFMA_TILE(As0, As1)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
        // epilogue: d2 to memory and, split, into the resident planes (fragment layout of phase 2's A operand)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = half * BN + wn * WN + b * 16 + l16;      // = k of phase 2
                const float bv = b1[col];
                const int ktile = col >> 5, kk = col & 31;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = wm * WM + a * 16 + 4 * kg + r;
                    const float v = fmaxf(__builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]) + bv, 0.f);
                    if (m0 + rl < M) D2[(long)(m0 + rl) * N1 + col] = v;
                    const _Float16 h = (_Float16)v, l = (_Float16)((v - (float)h) * 2048.f);
                    const int o = rl * LDH + (((kk >> 3) ^ swz(rl)) << 3) + (kk & 7);
                    Dp[(ktile * 2 + 0) * BM * LDH + o] = __builtin_bit_cast(unsigned short, h);
                    Dp[(ktile * 2 + 1) * BM * LDH + o] = __builtin_bit_cast(unsigned short, l);
                }
            }
    }
    __syncthreads();
    // ---- phase 2: pv tile, eight 128-column tiles; only the weight tile is staged
    for (int nt = 0; nt < N2 / BN; ++nt) {
        const float *brow = W2t + (long)(nt * BN + trow) * N1 + tk4;
        ZERO_ACC()
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * N1);
        for (int kt = 0; kt < N1 / BK; ++kt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int o = (trow + 64 * i) * LDH + wo;
                uint2 h, l;
                hsplit4(rb[i], h, l);
                *reinterpret_cast<uint2 *>(&Bs0[o]) = h;
                *reinterpret_cast<uint2 *>(&Bs1[o]) = l;
            }
            __syncthreads();
            const int ktn = kt + 1 < N1 / BK ? kt + 1 : kt;
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * N1 + ktn * BK);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned short *ap0 = Dp + (kt * 2 + 0) * BM * LDH, *ap1 = Dp + (kt * 2 + 1) * BM * LDH;
            MFMA_TILE(ap0, ap1)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = nt * BN + wn * WN + b * 16 + l16;
                const float bv = b2[col];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * WM + a * 16 + 4 * kg + r;
                    if (row < M) PV[(long)row * N2 + col] = fmaxf(__builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]) + bv, 0.f);
                }
            }
    }
#undef ZERO_ACC
#undef MFMA_TILE
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const int M = 81920;
    std::vector<float> hA((size_t)M * K1), hW1((size_t)N1 * K1), hW2((size_t)N2 * N1), hb1(N1), hb2(N2);
    srand(1);
    auto rnd = [](float s) { return s * ((float)rand() / RAND_MAX * 2.f - 1.f); };
    for (auto &v : hA) v = fmaxf(rnd(1.f), 0.f);      // a post-ReLU activation
    for (auto &v : hW1) v = rnd(0.08f);
    for (auto &v : hW2) v = rnd(0.08f);
    for (auto &v : hb1) v = rnd(0.05f);
    for (auto &v : hb2) v = rnd(0.05f);
    float *A, *W1, *W2, *b1, *b2, *D2a, *PVa, *D2b, *PVb;
    CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&W1, hW1.size() * 4)); CHECK(hipMalloc(&W2, hW2.size() * 4));
    CHECK(hipMalloc(&b1, N1 * 4)); CHECK(hipMalloc(&b2, N2 * 4));
    CHECK(hipMalloc(&D2a, (size_t)M * N1 * 4)); CHECK(hipMalloc(&PVa, (size_t)M * N2 * 4));
    CHECK(hipMalloc(&D2b, (size_t)M * N1 * 4)); CHECK(hipMalloc(&PVb, (size_t)M * N2 * 4));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W1, hW1.data(), hW1.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(W2, hW2.data(), hW2.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b1, hb1.data(), N1 * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b2, hb2.data(), N2 * 4, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void *)dense_fused, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds));
    const double flops = 2.0 * M * ((double)K1 * N1 + (double)N1 * N2);
    const float t1 = time_ms([&] { hipLaunchKernelGGL(dense_w8, dim3(N1 / BN, M / BM), dim3(512), 0, 0, A, W1, b1, D2a, M, N1, K1); }, 20);
    const float t2 = time_ms([&] { hipLaunchKernelGGL(dense_w8, dim3(N2 / BN, M / BM), dim3(512), 0, 0, D2a, W2, b2, PVa, M, N2, N1); }, 20);
    const float tu = time_ms([&] {
        hipLaunchKernelGGL(dense_w8, dim3(N1 / BN, M / BM), dim3(512), 0, 0, A, W1, b1, D2a, M, N1, K1);
        hipLaunchKernelGGL(dense_w8, dim3(N2 / BN, M / BM), dim3(512), 0, 0, D2a, W2, b2, PVa, M, N2, N1);
    }, 20);
    const float tf = time_ms([&] { hipLaunchKernelGGL(dense_fused, dim3(M / BM), dim3(512), kFusedLds, 0, A, W1, b1, W2, b2, D2b, PVb, M); }, 20);
    CHECK(hipDeviceSynchronize());
    printf("M = %d, d1 [M x 512] -> d2 [M x 256] -> pv [M x 1024], %.1f GFLOP of fp32 work\n", M, flops * 1e-9);
    printf("unfused: d2 %.1f us, pv %.1f us, back to back %.1f us = %.1f TFLOP/s\n", t1 * 1e3, t2 * 1e3, tu * 1e3, flops / (tu * 1e-3) * 1e-12);
    printf("fused (one 8-wave workgroup per 128 rows, %d KB of LDS, 1 workgroup per CU): %.1f us = %.1f TFLOP/s\n", kFusedLds / 1024, tf * 1e3,
           flops / (tf * 1e-3) * 1e-12);
    // the two forms compute the same sums in the same order: compare bit for bit, then a few rows against float64
    std::vector<float> pa((size_t)M * N2), pb((size_t)M * N2), da((size_t)M * N1), db((size_t)M * N1);
    CHECK(hipMemcpy(pa.data(), PVa, pa.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(pb.data(), PVb, pb.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(da.data(), D2a, da.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(db.data(), D2b, db.size() * 4, hipMemcpyDeviceToHost));
    size_t diff2 = 0, diffp = 0;
    for (size_t i = 0; i < da.size(); ++i) diff2 += da[i] != db[i];
    for (size_t i = 0; i < pa.size(); ++i) diffp += pa[i] != pb[i];
    printf("fused vs unfused: %zu of %zu d2 values differ, %zu of %zu pv values differ\n", diff2, da.size(), diffp, pa.size());
    double worst = 0;
    for (int row : {0, 1, 127, 128, 40000, M - 1}) {
        std::vector<double> d2(N1);
        for (int j = 0; j < N1; ++j) {
            double s = hb1[j];
            for (int k = 0; k < K1; ++k) s += (double)hA[(size_t)row * K1 + k] * hW1[(size_t)j * K1 + k];
            d2[j] = s > 0 ? s : 0;
        }
        for (int j = 0; j < N2; ++j) {
            double s = hb2[j];
            for (int k = 0; k < N1; ++k) s += d2[k] * hW2[(size_t)j * N1 + k];
            s = s > 0 ? s : 0;
            worst = fmax(worst, fabs(s - pb[(size_t)row * N2 + j]));
        }
    }
    printf("fused pv against float64 on six rows: max abs error %.3g\n", worst);
    return 0;
}
