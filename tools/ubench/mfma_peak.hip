// microbenchmark: fp32 MFMA rate with (a) registers only, (b) + LDS fragment reads as in gemm_rowk
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[256 * 36], Bs[64 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += 256) As[i] = (float)(i % 7) * 0.25f;
    for (int i = threadIdx.x; i < 64 * 36; i += 256) Bs[i] = (float)(i % 5) * 0.5f;
    __syncthreads();
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, lk = lane >> 5;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const float *ap = As + (wave * 64 + lr) * 36 + lk * 16, *bp = Bs + lr * 36 + lk * 16;
    float4 af[2] = {make_float4(1, 2, 3, 4), make_float4(2, 3, 4, 5)}, bf[2] = {make_float4(.1f, .2f, .3f, .4f), make_float4(.5f, .6f, .7f, .8f)};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 1) {
                af[0] = *reinterpret_cast<const float4 *>(ap + q * 4); af[1] = *reinterpret_cast<const float4 *>(ap + 32 * 36 + q * 4);
                bf[0] = *reinterpret_cast<const float4 *>(bp + q * 4); bf[1] = *reinterpret_cast<const float4 *>(bp + 32 * 36 + q * 4);
            }
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
        if (MODE == 2) { __syncthreads(); __syncthreads(); }
    }
    float s = 0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int blocks) {
    float *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 64.0 * 4096.0;
    printf("%-28s blocks %5d: %.2f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 768, 2048}) { run<0>("mfma regs only", blocks); run<1>("mfma + ds_read_b128", blocks); run<2>("mfma regs + 2 barriers/iter", blocks); }
    return 0;
}
