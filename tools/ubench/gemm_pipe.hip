// microbenchmark: staging-pipeline variants of the gemm_rowk<128,128,2,2> loop (A 128 rows, B 128 rows, BK = 32 floats).
//   V = 0  single LDS buffer, store -> barrier -> load next -> compute -> barrier      (the kernel as built)
//   V = 1  two LDS buffers, one barrier per K-tile: load t+1 -> compute t -> store t+1 into the other buffer -> barrier
//   V = 2  single buffer, BK = 64: two K-tiles per barrier pair
//   V = 3  as V = 1 but the store of tile t+1 is issued in the middle of compute t (after the second q-step)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 36;
template <int V>
__global__ __launch_bounds__(256, 2) void k(const float *__restrict__ A, const float *__restrict__ B, float *out, int nk, long amask) {
    constexpr int NBUF = (V == 1 || V == 3) ? 2 : 1, KT = V == 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float As[NBUF * KT * 128 * LD], Bs[NBUF * KT * 128 * LD];
    for (int i = threadIdx.x; i < NBUF * KT * 128 * LD; i += 256) { As[i] = (float)(i % 7) * 0.25f; Bs[i] = (float)(i % 5) * 0.5f; }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int aoffL = (wm * 64 + lr) * LD + lk * 16, boffL = (wn * 64 + lr) * LD + lk * 16;
    long aoff[4], boff[4];
    for (int i = 0; i < 4; ++i) {
        aoff[i] = (((long)blockIdx.x * 128 + trow + 32 * i) * 1600L) & amask;
        boff[i] = (long)(trow + 32 * i) * 3136L;
    }
    float4 ra[KT][4], rb[KT][4];
    for (int t = 0; t < KT; ++t) for (int i = 0; i < 4; ++i) { ra[t][i] = make_float4(1, 2, 3, 4); rb[t][i] = make_float4(.1f, .2f, .3f, .4f); }
#define LOADT(kt_, t_)                                                                            \
    {                                                                                             \
        const int toff = ((kt_) % 48) * 32;                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
            ra[t_][i] = *reinterpret_cast<const float4 *>(A + aoff[i] + toff + tk4);              \
            rb[t_][i] = *reinterpret_cast<const float4 *>(B + boff[i] + toff + tk4);              \
        }                                                                                         \
    }
#define STORET(buf_, t_)                                                                          \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
            *reinterpret_cast<float4 *>(As + (buf_) * 128 * LD + (trow + 32 * i) * LD + tk4) = ra[t_][i]; \
            *reinterpret_cast<float4 *>(Bs + (buf_) * 128 * LD + (trow + 32 * i) * LD + tk4) = rb[t_][i]; \
        }                                                                                         \
    }
#define QSTEP(buf_, q_)                                                                           \
    {                                                                                             \
        const float *ap = As + (buf_) * 128 * LD + aoffL, *bp = Bs + (buf_) * 128 * LD + boffL;    \
        float4 af[2], bf[2];                                                                      \
        af[0] = *reinterpret_cast<const float4 *>(ap + (q_) * 4); af[1] = *reinterpret_cast<const float4 *>(ap + 32 * LD + (q_) * 4); \
        bf[0] = *reinterpret_cast<const float4 *>(bp + (q_) * 4); bf[1] = *reinterpret_cast<const float4 *>(bp + 32 * LD + (q_) * 4); \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b) {                             \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);                              \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);                              \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);                              \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);                              \
        }                                                                                         \
    }
    if (V == 0) {
        LOADT(0, 0)
        for (int kt = 0; kt < nk; ++kt) {
            STORET(0, 0)
            __syncthreads();
            LOADT(kt + 1, 0)
            __builtin_amdgcn_sched_barrier(0);
            QSTEP(0, 0) QSTEP(0, 1) QSTEP(0, 2) QSTEP(0, 3)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    } else if (V == 2) {
        LOADT(0, 0) LOADT(1, 1)
        for (int kt = 0; kt < nk; kt += 2) {
            STORET(0, 0) STORET(1, 1)
            __syncthreads();
            LOADT(kt + 2, 0) LOADT(kt + 3, 1)
            __builtin_amdgcn_sched_barrier(0);
            QSTEP(0, 0) QSTEP(0, 1) QSTEP(0, 2) QSTEP(0, 3) QSTEP(1, 0) QSTEP(1, 1) QSTEP(1, 2) QSTEP(1, 3)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    } else {
        LOADT(0, 0)
        STORET(0, 0)
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            LOADT(kt + 1, 0)
            __builtin_amdgcn_sched_barrier(0);
            QSTEP(cur, 0) QSTEP(cur, 1)
            if (V == 3) { __builtin_amdgcn_sched_barrier(0); STORET(cur ^ 1, 0) __builtin_amdgcn_sched_barrier(0); }
            QSTEP(cur, 2) QSTEP(cur, 3)
            __builtin_amdgcn_sched_barrier(0);
            if (V == 1) STORET(cur ^ 1, 0)
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int t = 0; t < KT; ++t) for (int i = 0; i < 4; ++i) s += ra[t][i].x + rb[t][i].y;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V4: 8 waves, 256 x 128 tile (wave tile 64 x 64 as above), single LDS buffer, 2 barriers per K-tile
__global__ __launch_bounds__(512, 1) void k8(const float *__restrict__ A, const float *__restrict__ B, float *out, int nk, long amask) {
    __shared__ __attribute__((aligned(16))) float As[256 * LD], Bs[128 * LD];
    for (int i = threadIdx.x; i < 256 * LD; i += 512) As[i] = (float)(i % 7) * 0.25f;
    for (int i = threadIdx.x; i < 128 * LD; i += 512) Bs[i] = (float)(i % 5) * 0.5f;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;      // 64 rows per pass
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *ap = As + (wm * 64 + lr) * LD + lk * 16, *bp = Bs + (wn * 64 + lr) * LD + lk * 16;
    long aoff[4], boff[2];
    for (int i = 0; i < 4; ++i) aoff[i] = (((long)blockIdx.x * 256 + trow + 64 * i) * 1600L) & amask;
    for (int i = 0; i < 2; ++i) boff[i] = (long)(trow + 64 * i) * 3136L;
    float4 ra[4], rb[2];
    for (int i = 0; i < 4; ++i) ra[i] = make_float4(1, 2, 3, 4);
    for (int i = 0; i < 2; ++i) rb[i] = make_float4(.1f, .2f, .3f, .4f);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(As + (trow + 64 * i) * LD + tk4) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4 *>(Bs + (trow + 64 * i) * LD + tk4) = rb[i];
        __syncthreads();
        const int toff = ((kt + 1) % 48) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const float4 *>(A + aoff[i] + toff + tk4);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const float4 *>(B + boff[i] + toff + tk4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[2], bf[2];
            af[0] = *reinterpret_cast<const float4 *>(ap + q * 4); af[1] = *reinterpret_cast<const float4 *>(ap + 32 * LD + q * 4);
            bf[0] = *reinterpret_cast<const float4 *>(bp + q * 4); bf[1] = *reinterpret_cast<const float4 *>(bp + 32 * LD + q * 4);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += ra[i].x;
    for (int i = 0; i < 2; ++i) s += rb[i].y;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int OCC> void run8(const char *name, const float *A, const float *B, float *out, int blocks, int nk) {
    long amask = ((1L << 28) - 1) & ~3L;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k8, dim3(blocks), dim3(512), 0, 0, A, B, out, nk, amask);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k8, dim3(blocks), dim3(512), 0, 0, A, B, out, nk, amask); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 8 * nk * 64.0 * 4096.0;
    printf("%-70s blocks %6d nk %3d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks, nk, ms, flops / ms / 1e9);
}
template <int V> void run(const char *name, const float *A, const float *B, float *out, int blocks, int nk) {
    long amask = ((1L << 28) - 1) & ~3L;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, A, B, out, nk, amask);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, A, B, out, nk, amask); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * nk * 64.0 * 4096.0;
    printf("%-70s blocks %6d nk %3d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks, nk, ms, flops / ms / 1e9);
}
int main() {
    float *A, *B, *out;
    (void)hipMalloc(&A, (1L << 28) * 4 + (1 << 20)); (void)hipMemset(A, 0, (1L << 28) * 4); (void)hipMalloc(&B, 128 * 3136 * 4 + 8192); (void)hipMemset(B, 0, 128 * 3136 * 4);
    (void)hipMalloc(&out, 16384L * 512 * 4);
    for (int nk : {16, 48}) {
        int blocks = 1312;      // 40960+ rows / 128 x 4 N-tiles
        run<0>("V0 single buffer, 2 barriers per K-tile (as built)", A, B, out, blocks, nk);
        run<1>("V1 double buffer, 1 barrier, store after compute", A, B, out, blocks, nk);
        run<3>("V3 double buffer, 1 barrier, store mid-compute", A, B, out, blocks, nk);
        run<2>("V2 single buffer, BK = 64", A, B, out, blocks, nk);
        run8<1>("V4 8 waves, 256 x 128 tile", A, B, out, blocks / 2, nk);
    }
    return 0;
}
