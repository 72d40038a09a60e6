// microbenchmark (VERDICT r2 #4b): the three-product fp16 GEMM of net_gemm.h with BOTH operands already split into their two fp16
// planes in memory (h = fp16(x), l' = fp16((x - h) * 2^11): 4 bytes per element, the size of the fp32 it replaces) and the K loop
// fed by direct-to-LDS loads (global_load_lds_dwordx4) into a ring of NS stages -- no staging registers, no split instructions, no
// LDS stores by the waves.  One raw s_barrier per K-tile, a counted s_waitcnt vmcnt so that the next tile's loads stay in flight
// across it.  The LDS image is the product's (64-byte rows per plane, 16-byte chunks XOR-swizzled by row group); a wave-instruction
// writes 1 KB = 16 rows x 4 chunks lane-linearly, so the swizzle is applied to the SOURCE address of each lane.
// C[M][N] = A[M][K] . Bt[N][K]^T.  Compared with the same shapes in profiles/r02_f16x3_ubench.txt (adopted loop: 238-266 TF).
// Build: hipcc -O3 --offload-arch=gfx950 gemm_planes_glds.hip -o gemm_planes_glds
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void hsplit2(float x, float y, unsigned &h, unsigned &l) {
    const f32x2 xy = {x, y};
    const f16x2 hv = __builtin_convertvector(xy, f16x2);
    const f32x2 r = {(x - (float)hv[0]) * 2048.f, (y - (float)hv[1]) * 2048.f};
    const f16x2 lv = __builtin_convertvector(r, f16x2);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
__device__ __host__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// planes[0] = h, planes[1] = l' of a row-major fp32 array of n elements
__global__ void hpresplit_kernel(const float *__restrict__ src, unsigned short *__restrict__ planes, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4 *>(src + i);
    uint2 h, l;
    hsplit2(v.x, v.y, h.x, l.x);
    hsplit2(v.z, v.w, h.y, l.y);
    *reinterpret_cast<uint2 *>(planes + i) = h;
    *reinterpret_cast<uint2 *>(planes + n + i) = l;
}

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// BM x BN tile, 8 waves as WGM x WGN, BK = 32, NS ring stages.  MFMAS: 0 = loads and barriers only (what the load path costs
// alone), 1 = the full loop.  LOADS: 0 = no direct-to-LDS loads inside the loop (what the ring hides at best).
template <int BM, int BN, int WGM, int WGN, int NS, int LOADS, int XCD, int STAG = 0>
__global__ __launch_bounds__(512) void planes_glds(const unsigned short *__restrict__ Ap, const unsigned short *__restrict__ Bp, float *__restrict__ C,
                                                   int M, int N, int K) {
    constexpr int BK = 32, LDH = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int STAGE = (2 * BM + 2 * BN) * LDH;              // halves per stage: A planes, then B planes
    constexpr int NA = BM / 64, NB = BN / 64, PER = NA + NB;      // glds wave-instructions per wave and stage (BM / 16 row blocks x 2 planes / 8 waves)
    static_assert(WGM * WGN == 8 && BM % 64 == 0 && BN % 64 == 0, "8 waves");
    extern __shared__ __attribute__((aligned(1024))) unsigned short lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int tn_ = (N + BN - 1) / BN, tm_ = (M + BM - 1) / BM, total = tn_ * tm_;
    int L = blockIdx.x;
    if (XCD && total % 8 == 0) L = (L & 7) * (total >> 3) + (L >> 3);      // the tiles of one M panel meet on one XCD
    const int m0 = (L / tn_) * BM, n0 = (L % tn_) * BN;
    // per-lane source of the wave's glds pieces: piece q of an operand = (plane, 16-row block); lane -> row lane >> 2, physical chunk
    // lane & 3, which holds the row's chunk (lane & 3) ^ swz(row)
    const int prow = lane >> 2, pchunk = (lane & 3) ^ swz(lane >> 2);
    const unsigned short *ga[NA], *gb[NB];
    int la[NA], lb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = wave * NA + i, plane = q / (BM / 16), blk = q % (BM / 16);
        int row = m0 + blk * 16 + prow;
        row = row < M ? row : M - 1;
        ga[i] = Ap + (size_t)plane * M * K + (size_t)row * K + pchunk * 8;
        la[i] = plane * BM * LDH + blk * 16 * LDH;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int q = wave * NB + i, plane = q / (BN / 16), blk = q % (BN / 16);
        int row = n0 + blk * 16 + prow;
        row = row < N ? row : N - 1;
        gb[i] = Bp + (size_t)plane * N * K + (size_t)row * K + pchunk * 8;
        lb[i] = 2 * BM * LDH + plane * BN * LDH + blk * 16 * LDH;
    }
    auto issue = [&](int kt, int buf) {
        unsigned short *st = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(ga[i] + kt * BK), (lptr_t)(st + la[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(gb[i] + kt * BK), (lptr_t)(st + lb[i]), 16, 0, 0);
    };
    f32x4 acc[TM][TN], acl[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = acl[a][b][r] = 0.f;
    const int l16 = lane & 15, kg = lane >> 4;
    const int ro = (kg ^ swz(l16)) << 3;
    const int aro = (wm * WM + l16) * LDH + ro;
    const int bro = 2 * BM * LDH + (wn * WN + l16) * LDH + ro;
    const int KT = K / BK;
    if (LOADS) {
        for (int s = 0; s < NS - 1; ++s)
            if (s < KT) issue(s, s);
    } else {
        for (int s = 0; s < NS; ++s) issue(s < KT ? s : 0, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    f16x8 af[TM][2], bf[TN][2];
    auto rd = [&](int kt) {
        const unsigned short *st = lds + (kt % NS) * STAGE;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int p = 0; p < 2; ++p) af[a][p] = *reinterpret_cast<const f16x8 *>(&st[p * BM * LDH + aro + a * 16 * LDH]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[b][p] = *reinterpret_cast<const f16x8 *>(&st[p * BN * LDH + bro + b * 16 * LDH]);
    };
    auto mm = [&]() {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][1], bf[b][0], acl[a][b], 0, 0, 0);
                acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[b][1], acl[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
            }
    };
    if (STAG == 2) {
        // One barrier per K-tile as below, but the six glds pieces of tile kt + 2 are issued one at a time BETWEEN the MFMAs of tile kt
        // (after tiles 3, 6, 8, 11, 14, 16 of the 16): issued back to back right behind the barrier, all eight waves queue at the
        // address path at once while the matrix pipe idles.
        for (int kt = 0; kt < KT; ++kt) {
            if (LOADS) {
                if (NS > 2 && kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NS > 2 ? NS - 2 : 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            rd(kt);
            const bool more = LOADS && kt + NS - 1 < KT;
            unsigned short *stn = lds + ((kt + NS - 1) % NS) * STAGE;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][1], bf[b][0], acl[a][b], 0, 0, 0);
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[b][1], acl[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                    constexpr int TT = TM * TN;
                    const int t = a * TN + b, p0 = t * PER / TT, p1 = (t + 1) * PER / TT;
                    if (p1 != p0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) {
                            if (p0 < NA) __builtin_amdgcn_global_load_lds((gptr_t)(ga[p0 < NA ? p0 : 0] + (LOADS == 2 ? ((kt + NS - 1) & 1) : (kt + NS - 1)) * BK), (lptr_t)(stn + la[p0 < NA ? p0 : 0]), 16, 0, 0);
                            else __builtin_amdgcn_global_load_lds((gptr_t)(gb[p0 >= NA ? p0 - NA : 0] + (LOADS == 2 ? ((kt + NS - 1) & 1) : (kt + NS - 1)) * BK), (lptr_t)(stn + lb[p0 >= NA ? p0 - NA : 0]), 16, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
    } else if (STAG == 1) {
        // Two wave groups half a K-tile apart (waves w and w + 4 share a SIMD): while one group runs its 48 MFMAs the other reads its
        // fragments, so the LDS pipe and the matrix pipe work at the same time instead of in turns.  Two barriers per K-tile.
        const bool late = wave >= 4;
        for (int kt = 0; kt < KT; ++kt) {
            if (LOADS) {
                if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (LOADS && kt + 2 < KT) issue(kt + 2, (kt + 2) % NS);      // the late group read that buffer one barrier ago
            if (!late) rd(kt);
            else if (kt > 0) mm();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!late) mm();
            else rd(kt);
        }
        if (late) mm();
    } else
    for (int kt = 0; kt < KT; ++kt) {
        if (LOADS) {
            // this wave's pieces of stage kt have landed (those of kt + 1 may still fly), then everybody's have
            if (kt + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < KT) issue(kt + 2, (kt + 2) % NS);      // into the buffer whose reads all waves finished before the barrier
        }
        rd(kt);
        mm();
        if (!LOADS && STAG != 3) {      // STAG 3: the waves run free (static data): what the barriers themselves cost
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * WM + a * 16 + 4 * kg + r, col = n0 + wn * WN + b * 16 + l16;
                if (row < M && col < N) C[(long)row * N + col] = __builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]);
            }
}

static double err_vs_double(const std::vector<float> &A, const std::vector<float> &B, const std::vector<float> &C, int M, int N, int K, double *rms_out) {
    double worst = 0., ss = 0.;
    long cnt = 0;
    for (int s = 0; s < 48; ++s) {
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

template <int BM, int BN, int WGM, int WGN, int NS, int LOADS, int XCD, int STAG = 0>
static void variant(const char *name, const unsigned short *Ap, const unsigned short *Bp, float *C, int M, int N, int K, const std::vector<float> &hA,
                    const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    const int tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM);
    const size_t lds_bytes = (size_t)NS * (2 * BM + 2 * BN) * 32 * 2;
    auto kern = planes_glds<BM, BN, WGM, WGN, NS, LOADS, XCD, STAG>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        printf("%-60s LDS %zu bytes refused\n", name, lds_bytes);
        return;
    }
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds_bytes, 0, Ap, Bp, C, M, N, K); }, 20);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%-60s failed: %s\n", name, hipGetErrorString(e)); return; }
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-60s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
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
    (void)hipMalloc(&Ap, hA.size() * 4); (void)hipMalloc(&Bp, hB.size() * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(hpresplit_kernel, dim3((unsigned)((hA.size() / 4 + 255) / 256)), dim3(256), 0, 0, A, Ap, (long)hA.size());
    hipLaunchKernelGGL(hpresplit_kernel, dim3((unsigned)((hB.size() / 4 + 255) / 256)), dim3(256), 0, 0, B, Bp, (long)hB.size());
    variant<256, 128, 4, 2, 3, 1, 1>("planes + glds ring, 256x128, 64x64 wave tiles, 3 stages", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 3, 1, 1, 1>("  + two wave groups half a K-tile apart", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 256, 2, 4, 3, 1, 1, 1>("  + two wave groups, 128x256", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 4, 2, 3, 1, 1, 1>("  + two wave groups, 128x128 (32x64 wave tiles)", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 3, 0, 1, 1>("  (speed only) two wave groups, no loads in the loop", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 3, 1, 1, 2>("  glds pieces issued between the MFMAs, 256x128", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 256, 2, 4, 3, 1, 1, 2>("  glds pieces issued between the MFMAs, 128x256", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 4, 2, 3, 1, 1, 2>("  glds pieces issued between the MFMAs, 128x128", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 3, 2, 1, 2>("  (speed only) 256x128, every load from K-tiles 0/1 (L2-resident)", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 4, 2, 2, 1, 1, 2>("  same, 128x128 with 2 stages: two workgroups per CU", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 2, 4, 2, 1, 1, 2>("  same, 128x128 (64x32 wave tiles), 2 stages, 2 WG/CU", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 2, 1, 1, 2>("  256x128 with 2 stages", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 256, 2, 4, 3, 1, 1>("planes + glds ring, 128x256, 64x64 wave tiles, 3 stages", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 4, 2, 3, 1, 1>("planes + glds ring, 128x128, 32x64 wave tiles, 3 stages", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 2, 4, 3, 1, 1>("planes + glds ring, 128x128, 64x32 wave tiles, 3 stages", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 3, 0, 1, 3>("(speed only) 256x128, no loads, NO barriers (waves run free)", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 4, 2, 3, 0, 1, 3>("(speed only) 128x128, no loads, NO barriers (waves run free)", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<256, 128, 4, 2, 3, 0, 1>("(speed only) 256x128 without loads in the loop", Ap, Bp, C, M, N, K, hA, hB, hC);
    variant<128, 128, 4, 2, 3, 0, 1>("(speed only) 128x128 without loads in the loop", Ap, Bp, C, M, N, K, hA, hB, hC);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(Ap); (void)hipFree(Bp);
}

int main() {
    run(40960, 512, 1600);      // dense1 patch forward
    run(40960, 512, 256);       // pol1 / v1 forward
    run(40960, 1664, 512);      // dense1 patch data gradient (N = 1600 padded to the tile)
    run(8192, 512, 1600);
    return 0;
}
