// microbenchmark: the gemm_rowk<256,64> loop structure with its global loads / LDS stores, to find what
// keeps the real kernel at ~72% of the fp32 MFMA peak.  MODE bits: 1 = global loads, 2 = LDS stores+barriers,
// 4 = loads come from a small (L2-resident) buffer instead of a 2 GB one.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float *__restrict__ A, const float *__restrict__ B, float *out, int nk, long amask) {
    constexpr int LD = 36;
    __shared__ __attribute__((aligned(16))) float As[256 * LD], Bs[64 * LD];
    for (int i = threadIdx.x; i < 256 * LD; i += 256) As[i] = (float)(i % 7) * 0.25f;
    for (int i = threadIdx.x; i < 64 * LD; i += 256) Bs[i] = (float)(i % 5) * 0.5f;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *ap = As + (wave * 64 + lr) * LD + lk * 16, *bp = Bs + lr * LD + lk * 16;
    long aoff[8];
    for (int i = 0; i < 8; ++i) aoff[i] = (((long)blockIdx.x * 256 + trow + 32 * i) * 632L) & amask;   // ~conv2 row pitch
    float4 ra[8], rb0, rb1;
    for (int i = 0; i < 8; ++i) ra[i] = make_float4(1, 2, 3, 4);
    rb0 = rb1 = make_float4(.1f, .2f, .3f, .4f);
    for (int kt = 0; kt < nk; ++kt) {
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4 *>(As + (trow + 32 * i) * LD + tk4) = ra[i];
            *reinterpret_cast<float4 *>(Bs + trow * LD + tk4) = rb0;
            *reinterpret_cast<float4 *>(Bs + (trow + 32) * LD + tk4) = rb1;
            __syncthreads();
        }
        if (MODE & 1) {
            int toff = (kt & 15) * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const float4 *>(A + aoff[i] + toff + tk4);
            rb0 = *reinterpret_cast<const float4 *>(B + (long)trow * 512 + toff + tk4);
            rb1 = *reinterpret_cast<const float4 *>(B + (long)(trow + 32) * 512 + toff + tk4);
        }
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
        if (MODE & 2) __syncthreads();
    }
    float s = ra[0].x + rb0.x + rb1.y;
    for (int i = 1; i < 8; ++i) s += ra[i].y;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, const float *A, const float *B, float *out, int blocks, int nk) {
    long amask = (MODE & 4) ? ((1L << 20) - 1) & ~3L : ((1L << 29) - 1) & ~3L;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, A, B, out, nk, amask);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, A, B, out, nk, amask); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * nk * 64.0 * 4096.0;
    printf("%-44s blocks %6d nk %3d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks, nk, ms, flops / ms / 1e9);
}
int main() {
    float *A, *B, *out;
    (void)hipMalloc(&A, (1L << 29) * 4 + (1 << 20)); (void)hipMemset(A, 0, (1L << 29) * 4); (void)hipMalloc(&B, 64 * 512 * 4 + 4096); (void)hipMemset(B, 0, 64 * 512 * 4);
    (void)hipMalloc(&out, 16384L * 256 * 4);
    for (int nk : {16, 64}) {
        int blocks = 12960;
        run<0>("mfma + ds_read only", A, B, out, blocks, nk);
        run<2>("+ LDS stores + 2 barriers", A, B, out, blocks, nk);
        run<7>("+ global loads (1 MB, L2-resident)", A, B, out, blocks, nk);
        run<3>("+ global loads (2 GB stream)", A, B, out, blocks, nk);
        run<1>("global loads (2 GB) without LDS stores", A, B, out, blocks, nk);
    }
    return 0;
}
