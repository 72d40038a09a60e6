// microbenchmark: what each ingredient of the bf16 x6 GEMM loop costs on top of the bare MFMA stream.
// one "tile" = 48 v_mfma_f32_32x32x16_bf16 per wave (128x128x32 block tile, 4 waves).  MODE bits:
//   1: 24 ds_read_b128 fragment reads per tile      2: the split (176 VALU) of 8 float4 per tile
//   4: 24 ds_write_b64 + 2 barriers per tile         8: 8 global_load_dwordx4 per tile
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split4(const float4 v, uint2 &h, uint2 &m, uint2 &l) {
    const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
    h.x = __builtin_amdgcn_perm(x1, x0, 0x07060302u);
    h.y = __builtin_amdgcn_perm(x3, x2, 0x07060302u);
    const float r0 = v.x - __uint_as_float(x0 & 0xffff0000u), r1 = v.y - __uint_as_float(x1 & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(x2 & 0xffff0000u), r3 = v.w - __uint_as_float(x3 & 0xffff0000u);
    const unsigned y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
    m.x = __builtin_amdgcn_perm(y1, y0, 0x07060302u);
    m.y = __builtin_amdgcn_perm(y3, y2, 0x07060302u);
    const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(y2 & 0xffff0000u), s3 = r3 - __uint_as_float(y3 & 0xffff0000u);
    l.x = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    l.y = __builtin_amdgcn_perm(__float_as_uint(s3), __float_as_uint(s2), 0x07060302u);
}
template <int MODE, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const float *__restrict__ G, float *out, int iters) {
    constexpr int LDH = 40;
    __shared__ __attribute__((aligned(16))) unsigned short As[3][128 * LDH], Bs[3][128 * LDH];
    for (int i = threadIdx.x; i < 3 * 128 * LDH; i += 256) { (&As[0][0])[i] = 0x3f80 + (i % 7); (&Bs[0][0])[i] = 0x3c00 + (i % 5); }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1, trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int aro = (wm * 64 + lr) * LDH + lk * 8, bro = (wn * 64 + lr) * LDH + lk * 8;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 af[2][3], bf[2][3];
    for (int a = 0; a < 2; ++a) for (int p = 0; p < 3; ++p) { af[a][p] = *reinterpret_cast<const bf16x8 *>(&As[p][aro + a * 32 * LDH]); bf[a][p] = *reinterpret_cast<const bf16x8 *>(&Bs[p][bro + a * 32 * LDH]); }
    float4 ra[8];
    const float *gp = G + ((long)blockIdx.x * 128 + trow) * 1600 + tk4;
    for (int i = 0; i < 8; ++i) ra[i] = make_float4(1.1f + i, 2.3f, 3.7f, 4.9f + tid);
    for (int it = 0; it < iters; ++it) {
        if (MODE & 6) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint2 h, m, l;
                if (MODE & 2) split4(ra[i], h, m, l);
                else { h = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y)); m = make_uint2(__float_as_uint(ra[i].z), __float_as_uint(ra[i].w)); l = h; }
                if (MODE & 4) {
                    unsigned short(*S)[128 * LDH] = i < 4 ? As : Bs;
                    const int o = (trow + 32 * (i & 3)) * LDH + tk4;
                    *reinterpret_cast<uint2 *>(&S[0][o]) = h;
                    *reinterpret_cast<uint2 *>(&S[1][o]) = m;
                    *reinterpret_cast<uint2 *>(&S[2][o]) = l;
                } else {
                    ra[i].x += __uint_as_float((h.x ^ m.y ^ l.x) & 0x007f0000u);      // keep the split alive
                    ra[i].y += __uint_as_float((h.y ^ m.x ^ l.y) & 0x007f0000u);
                }
            }
            if (MODE & 4) __syncthreads();
        }
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const float4 *>(gp + (long)32 * (i & 3) * 1600 + (it % 50) * 32 + (i >> 2) * 409600L * 100);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (MODE & 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        af[a][p] = *reinterpret_cast<const bf16x8 *>(&As[p][aro + a * 32 * LDH + s * 16]);
                        bf[a][p] = *reinterpret_cast<const bf16x8 *>(&Bs[p][bro + a * 32 * LDH + s * 16]);
                    }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 4) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += ra[i].x + ra[i].y;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int OCC> void run(const char *name, const float *G, int blocks) {
    float *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, OCC>), dim3(blocks), dim3(256), 0, 0, G, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, OCC>), dim3(blocks), dim3(256), 0, 0, G, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_flops = (double)blocks * 4 * iters * 48 * 2.0 * 32 * 32 * 16;
    const double cyc = ms * 1e-3 * 2.4e9 / iters / (blocks / 256.0 / OCC > 1 ? blocks / 256.0 / OCC : 1);
    printf("%-52s occ %d blocks %5d: %7.3f ms  %7.1f TF bf16 (= %6.1f TF fp32-equivalent)  ~%5.0f cyc/tile/residency-slot\n", name, OCC, blocks, ms,
           mfma_flops / ms / 1e9, mfma_flops / 6 / ms / 1e9, cyc);
    (void)hipFree(out);
}
int main() {
    float *G;
    (void)hipMalloc(&G, (size_t)2 * 409600L * 100 * 4 + (size_t)1024 * 128 * 1600 * 4);
    (void)hipMemset(G, 0, (size_t)2 * 409600L * 100 * 4 + (size_t)1024 * 128 * 1600 * 4);
    for (int blocks : {256, 512}) {
        run<0, 2>("mfma only", G, blocks);
        run<1, 2>("+ fragment reads", G, blocks);
        run<2, 2>("+ split", G, blocks);
        run<3, 2>("+ fragment reads + split", G, blocks);
        run<5, 2>("+ fragment reads + LDS writes + barriers", G, blocks);
        run<7, 2>("+ fragment reads + split + LDS writes + barriers", G, blocks);
        run<15, 2>("+ everything incl. global loads", G, blocks);
    }
    return 0;
}
