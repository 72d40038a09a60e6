// microbenchmark: what the fp32-on-bf16-pipe GEMM of net_gemm.h gains when an operand arrives ALREADY split into its three
// bf16 planes (x = h + m + l): the weight operand is constant for a whole rollout + update, so its planes can be made once
// per parameter version; the activation operand could be written pre-split by its producer.
//   ASRC / BSRC = 0   fp32 in global memory, split while staging into LDS (what net_gemm.h does today)
//                 1   three bf16 planes [3][rows][K] in global memory, staged into LDS with 16-byte loads (no VALU split)
//   BSRC        = 2   planes read straight into the MFMA fragment registers (no LDS for B at all)
// Same tile shape, LDS swizzle and MFMA order as gemm_rowk<128,128,2,2>.  Random operands (the matrix pipe is power-limited
// on toggling data, zeros flatter it).
// Build: hipcc -O3 --offload-arch=gfx950 gemm_presplit.hip -o gemm_presplit
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split1(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x);
    const float r = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r);
    l = __float_as_uint(r - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned pack2(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
__device__ __forceinline__ void split4(float4 v, uint2 &h, uint2 &m, uint2 &l) {
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split1(v.x, h0, m0, l0); split1(v.y, h1, m1, l1); split1(v.z, h2, m2, l2); split1(v.w, h3, m3, l3);
    h = make_uint2(pack2(h0, h1), pack2(h2, h3));
    m = make_uint2(pack2(m0, m1), pack2(m2, m3));
    l = make_uint2(pack2(l0, l1), pack2(l2, l3));
}
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// planes[p][i] for a row-major fp32 array of n elements
__global__ void presplit_kernel(const float *__restrict__ src, unsigned short *__restrict__ planes, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    uint2 h, m, l;
    split4(*reinterpret_cast<const float4 *>(src + i), h, m, l);
    *reinterpret_cast<uint2 *>(planes + i) = h;
    *reinterpret_cast<uint2 *>(planes + n + i) = m;
    *reinterpret_cast<uint2 *>(planes + 2 * n + i) = l;
}

template <int BM, int BN, int WGM, int WGN, int ASRC, int BSRC>
__global__ __launch_bounds__(256, 2) void rowk(const float *__restrict__ A, const float *__restrict__ Bt, const unsigned short *__restrict__ Ap,
                                               const unsigned short *__restrict__ Bp, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16, NA = BM / 32, NB = BN / 32;
    constexpr int PA = BM / 64, PB = BN / 64;      // 16-byte loads per thread and plane for a pre-split tile
    __shared__ __attribute__((aligned(16))) unsigned short As[3][BM * LDH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[BSRC == 2 ? 1 : 3][BSRC == 2 ? 8 : BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);
    // pre-split staging map: row prow + 64 i, 16-byte chunk pc (8 consecutive k)
    const int prow = tid >> 2, pc = tid & 3;
    const long MK = (long)M * K, NK = (long)N * K;
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    const unsigned short *aprow = Ap + (long)(m0 + prow) * K + pc * 8;
    const unsigned short *bprow = Bp + (long)(n0 + prow) * K + pc * 8;
    const int l16 = lane & 15, kg = lane >> 4;
    // direct B fragments: lane (l16, kg) of tile b reads 8 consecutive k of row n0 + wn*WN + 16 b + l16
    const unsigned short *bfrow = Bp + (long)(n0 + wn * WN + l16) * K + kg * 8;
    float4 ra[NA], rb[NB];
    // named scalars, not arrays: arrays written in one place and read in another end up in scratch memory
    static_assert(PA <= 2 || ASRC == 0, "PA"); static_assert(PB <= 2 || BSRC != 1, "PB");
    uint4 pa00, pa01, pa10, pa11, pa20, pa21, pb00, pb01, pb10, pb11, pb20, pb21;
    pa00 = pa01 = pa10 = pa11 = pa20 = pa21 = pb00 = pb01 = pb10 = pb11 = pb20 = pb21 = make_uint4(0, 0, 0, 0);
    uint4 fb[TN][3];
#define LOADP(v_, base_, plane_, i_, stride_, kt_) v_ = *reinterpret_cast<const uint4 *>(base_ + (plane_) * (stride_) + (long)64 * (i_) * K + (kt_) * BK);
#define LOAD(kt_)                                                                                                   \
    {                                                                                                               \
        if (ASRC == 0) {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + (kt_) * BK); \
        } else {                                                                                                    \
            LOADP(pa00, aprow, 0, 0, MK, kt_) LOADP(pa10, aprow, 1, 0, MK, kt_) LOADP(pa20, aprow, 2, 0, MK, kt_)   \
            if (PA > 1) { LOADP(pa01, aprow, 0, 1, MK, kt_) LOADP(pa11, aprow, 1, 1, MK, kt_) LOADP(pa21, aprow, 2, 1, MK, kt_) } \
        }                                                                                                           \
        if (BSRC == 0) {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + (kt_) * BK); \
        } else if (BSRC == 1) {                                                                                     \
            LOADP(pb00, bprow, 0, 0, NK, kt_) LOADP(pb10, bprow, 1, 0, NK, kt_) LOADP(pb20, bprow, 2, 0, NK, kt_)   \
            if (PB > 1) { LOADP(pb01, bprow, 0, 1, NK, kt_) LOADP(pb11, bprow, 1, 1, NK, kt_) LOADP(pb21, bprow, 2, 1, NK, kt_) } \
        } else {                                                                                                    \
            _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                          \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) fb[b][p] = *reinterpret_cast<const uint4 *>(bfrow + p * NK + (long)16 * b * K + (kt_) * BK); \
        }                                                                                                           \
    }
    LOAD(0)
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    const int rofs = (kg ^ swz(l16)) << 3;
    const int aro = (wm * WM + l16) * LDH + rofs, bro = (wn * WN + l16) * LDH + rofs;
    const int po = prow * LDH + ((pc ^ swz(prow)) << 3);
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        if (ASRC == 0) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                uint2 h, m, l;
                split4(ra[i], h, m, l);
                const int o = (trow + 32 * i) * LDH + wo;
                *reinterpret_cast<uint2 *>(&As[0][o]) = h;
                *reinterpret_cast<uint2 *>(&As[1][o]) = m;
                *reinterpret_cast<uint2 *>(&As[2][o]) = l;
            }
        } else {
            *reinterpret_cast<uint4 *>(&As[0][po]) = pa00; *reinterpret_cast<uint4 *>(&As[1][po]) = pa10; *reinterpret_cast<uint4 *>(&As[2][po]) = pa20;
            if (PA > 1) {      // swz(row + 64) == swz(row)
                *reinterpret_cast<uint4 *>(&As[0][po + 64 * LDH]) = pa01; *reinterpret_cast<uint4 *>(&As[1][po + 64 * LDH]) = pa11;
                *reinterpret_cast<uint4 *>(&As[2][po + 64 * LDH]) = pa21;
            }
        }
        if (BSRC == 0) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                uint2 h, m, l;
                split4(rb[i], h, m, l);
                const int o = (trow + 32 * i) * LDH + wo;
                *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
                *reinterpret_cast<uint2 *>(&Bs[1][o]) = m;
                *reinterpret_cast<uint2 *>(&Bs[2][o]) = l;
            }
        } else if (BSRC == 1) {
            *reinterpret_cast<uint4 *>(&Bs[0][po]) = pb00; *reinterpret_cast<uint4 *>(&Bs[1][po]) = pb10; *reinterpret_cast<uint4 *>(&Bs[2][po]) = pb20;
            if (PB > 1) {
                *reinterpret_cast<uint4 *>(&Bs[0][po + 64 * LDH]) = pb01; *reinterpret_cast<uint4 *>(&Bs[1][po + 64 * LDH]) = pb11;
                *reinterpret_cast<uint4 *>(&Bs[2][po + 64 * LDH]) = pb21;
            }
        }
        bf16x8 bf[TN][3];
        if (BSRC == 2) {
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[b][p] = __builtin_bit_cast(bf16x8, fb[b][p]);
        }
        __syncthreads();
        LOAD(kt + 1 < nk ? kt + 1 : kt)
        __builtin_amdgcn_sched_barrier(0);
        {
            if (BSRC != 2) {
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8 *>(&Bs[p][bro + b * 16 * LDH]);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8 *>(&As[p][aro + a * 16 * LDH]);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bf[b][0], acc[a][b], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + wm * WM + a * 16 + 4 * kg + r;
                int col = n0 + wn * WN + b * 16 + l16;
                if (row < M && col < N) C[(long)row * N + col] = acc[a][b][r];
            }
}

static double err_vs_double(const std::vector<float> &A, const std::vector<float> &B, const std::vector<float> &C, int M, int N, int K, double *rms_out) {
    double worst = 0., ss = 0.;
    long cnt = 0;
    for (int s = 0; s < 32; ++s) {
        int r = (int)((long)s * 7919 % M);
        for (int c = 0; c < N; c += 5) {
            double ref = 0., mag = 0.;
            for (int k = 0; k < K; ++k) {
                double p = (double)A[(long)r * K + k] * (double)B[(long)c * K + k];
                ref += p; mag += fabs(p);
            }
            double e = fabs((double)C[(long)r * N + c] - ref) / mag;
            worst = fmax(worst, e); ss += e * e; ++cnt;
        }
    }
    *rms_out = sqrt(ss / cnt);
    return worst;
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int BM, int BN, int WGM, int WGN, int ASRC, int BSRC>
static void variant(const char *name, const float *A, const float *B, const unsigned short *Ap, const unsigned short *Bp, float *C, int M, int N, int K,
                    const std::vector<float> &hA, const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL((rowk<BM, BN, WGM, WGN, ASRC, BSRC>), grid, dim3(256), 0, 0, A, B, Ap, Bp, C, M, N, K); }, 20);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-46s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
    fflush(stdout);
}

static void run(int M, int N, int K) {
    std::vector<float> hA((long)M * K), hB((long)N * K), hC((long)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = rnd() * (1.f + 0.37f * rnd());
    for (auto &v : hB) v = 0.05f * rnd() * (1.f + 0.11f * rnd());
    float *A, *B, *C;
    unsigned short *Ap, *Bp;
    (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&B, hB.size() * 4); (void)hipMalloc(&C, hC.size() * 4);
    (void)hipMalloc(&Ap, hA.size() * 6); (void)hipMalloc(&Bp, hB.size() * 6);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((hA.size() / 4 + 255) / 256)), dim3(256), 0, 0, A, Ap, (long)hA.size());
    hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((hB.size() / 4 + 255) / 256)), dim3(256), 0, 0, B, Bp, (long)hB.size());
    float pms = time_ms([&] { hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((hA.size() / 4 + 255) / 256)), dim3(256), 0, 0, A, Ap, (long)hA.size()); }, 5);
    printf("presplit of A (%ld floats): %.3f ms = %.2f TB/s (4 B read + 6 B written per element)\n", (long)hA.size(), pms, hA.size() * 10.0 / pms / 1e9);
    variant<128, 128, 2, 2, 0, 0>("A fp32 split, B fp32 split (today)", A, B, Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 2, 2, 0, 1>("A fp32 split, B planes via LDS", A, B, Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 2, 2, 0, 2>("A fp32 split, B planes direct to registers", A, B, Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 2, 2, 1, 1>("A planes via LDS, B planes via LDS", A, B, Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 2, 2, 1, 2>("A planes via LDS, B planes direct", A, B, Ap, Bp, C, M, N, K, hA, hB, hC);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(Ap); (void)hipFree(Bp);
}

int main() {
    run(40960, 512, 1600);      // dense1 patch forward
    run(40960, 512, 256);       // pol1 / v1 forward
    run(40960, 1664, 512);      // dense1 patch data gradient (N = 1600 padded to the tile)
    return 0;
}
