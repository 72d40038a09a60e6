// microbenchmark: fp32 GEMM on the fp16 matrix pipe with THREE products instead of the six bf16 products of net_gemm.h.
// Each fp32 operand is split into two fp16 terms while its tile is staged into LDS: h = fp16(x), l' = fp16((x - h) * 2^11)
// (22-23 significand bits together); C = sum h_a h_b + 2^-11 * sum (h_a l'_b + l'_a h_b), the l'l' product (2^-22) is dropped.
// The scaled low plane keeps l' in the normal fp16 range whenever h is; two accumulator sets (the second is scaled once at the end).
// Same tile shape (128x128, 2x2 waves), LDS swizzle and loop structure as gemm_rowk; compared with the 6-product bf16 loop in the same file.
// Range: fp16 holds 6.1e-5 .. 65504 with full precision; operands outside need a power-of-two scale at their producer.
// Build: hipcc -O3 --offload-arch=gfx950 gemm_f16x3.hip -o gemm_f16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

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
// fp16 pair split, round-to-nearest: h = fp16(x); l' = fp16((x - h) * 2048) (x - h is exact).  3 VALU instructions per element:
// v_cvt_pk_f16_f32, v_cvt_f32_f16, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

// planes[0] = h, planes[1] = l' of a row-major fp32 array of n elements
__global__ void hpresplit_kernel(const float *__restrict__ src, unsigned short *__restrict__ planes, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    uint2 h, l;
    hsplit4(*reinterpret_cast<const float4 *>(src + i), h, l);
    *reinterpret_cast<uint2 *>(planes + i) = h;
    *reinterpret_cast<uint2 *>(planes + n + i) = l;
}

// MODE 3: as MODE 1 with the B operand (the weights) pre-split into its two planes once, staged into LDS with 16-byte loads.
// MODE 0: six bf16 products (today).  MODE 1: three fp16 products, two accumulator sets.  MODE 2: three fp16 products into ONE
// accumulator (wrong results: the low plane is scaled; only a speed reference for the register cost of the second set)
template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(256, 2) void rowk(const float *__restrict__ A, const float *__restrict__ Bt, const unsigned short *__restrict__ Bp,
                                               float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 32, NP = MODE == 0 ? 3 : 2;
    constexpr bool TWO = MODE == 1 || MODE == 3 || MODE == 6 || MODE == 7;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16, NA = BM / 32, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short As[NP][BM * LDH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[NP][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    int bx_ = blockIdx.x, by_ = blockIdx.y;
    if (MODE == 7) {      // XCD x takes M-tile 8g + x with all its N tiles (gemm_rowk's order)
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
    const int prow = tid >> 2, pc = tid & 3;
    const long NK = (long)N * K;
    const unsigned short *bprow = Bp + (long)(n0 + prow) * K + pc * 8;
    const int po = prow * LDH + ((pc ^ swz(prow)) << 3);
    uint4 pb00, pb01, pb10, pb11;
    pb00 = pb01 = pb10 = pb11 = make_uint4(0, 0, 0, 0);
    static_assert(BN == 128 || MODE != 3, "two 64-row halves");
    float4 ra[NA], rb[NB];
#define LOAD(kt_)                                                                                                                      \
    {                                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + (kt_) * BK); \
        if (MODE != 3) {                                                                                                               \
            _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + (kt_) * BK); \
        } else {                                                                                                                       \
            pb00 = *reinterpret_cast<const uint4 *>(bprow + (kt_) * BK); pb01 = *reinterpret_cast<const uint4 *>(bprow + (long)64 * K + (kt_) * BK); \
            pb10 = *reinterpret_cast<const uint4 *>(bprow + NK + (kt_) * BK); pb11 = *reinterpret_cast<const uint4 *>(bprow + NK + (long)64 * K + (kt_) * BK); \
        }                                                                                                                              \
    }
    LOAD(0)
    f32x4 acc[TM][TN], acl[TWO ? TM : 1][TWO ? TN : 1];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[a][b][r] = 0.f;
                if (TWO) acl[a][b][r] = 0.f;
            }
    const int rofs = (kg ^ swz(l16)) << 3;
    const int aro = (wm * WM + l16) * LDH + rofs, bro = (wn * WN + l16) * LDH + rofs;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int o = (trow + 32 * i) * LDH + wo;
            if (MODE == 0) {
                uint2 h, m, l;
                split4(ra[i], h, m, l);
                *reinterpret_cast<uint2 *>(&As[0][o]) = h;
                *reinterpret_cast<uint2 *>(&As[1][o]) = m;
                *reinterpret_cast<uint2 *>(&As[NP - 1][o]) = l;
            } else {
                uint2 h, l;
                hsplit4(ra[i], h, l);
                *reinterpret_cast<uint2 *>(&As[0][o]) = h;
                *reinterpret_cast<uint2 *>(&As[1][o]) = l;
            }
        }
        if (MODE == 3) {
            *reinterpret_cast<uint4 *>(&Bs[0][po]) = pb00; *reinterpret_cast<uint4 *>(&Bs[0][po + 64 * LDH]) = pb01;
            *reinterpret_cast<uint4 *>(&Bs[1][po]) = pb10; *reinterpret_cast<uint4 *>(&Bs[1][po + 64 * LDH]) = pb11;
        }
#pragma unroll
        for (int i = 0; i < (MODE == 3 ? 0 : NB); ++i) {
            const int o = (trow + 32 * i) * LDH + wo;
            if (MODE == 0) {
                uint2 h, m, l;
                split4(rb[i], h, m, l);
                *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
                *reinterpret_cast<uint2 *>(&Bs[1][o]) = m;
                *reinterpret_cast<uint2 *>(&Bs[NP - 1][o]) = l;
            } else {
                uint2 h, l;
                hsplit4(rb[i], h, l);
                *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
                *reinterpret_cast<uint2 *>(&Bs[1][o]) = l;
            }
        }
        __syncthreads();
        if (MODE != 6) LOAD(kt + 1 < nk ? kt + 1 : kt)
        if (MODE == 6) {      // opaque to the optimiser: the split stays in the loop, only the loads are gone
#pragma unroll
            for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(rb[i].x), "+v"(rb[i].y), "+v"(rb[i].z), "+v"(rb[i].w));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
            bf16x8 bf[TN][3];
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8 *>(&Bs[p < NP ? p : 0][bro + b * 16 * LDH]);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8 *>(&As[p < NP ? p : 0][aro + a * 16 * LDH]);
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
        } else {
            f16x8 bf[TN][2];
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[b][p] = *reinterpret_cast<const f16x8 *>(&Bs[p][bro + b * 16 * LDH]);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f16x8 af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const f16x8 *>(&As[p][aro + a * 16 * LDH]);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    if (TWO) {
                        acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bf[b][0], acl[a][b], 0, 0, 0);
                        acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][1], acl[a][b], 0, 0, 0);
                    } else {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bf[b][0], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][1], acc[a][b], 0, 0, 0);
                    }
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][0], acc[a][b], 0, 0, 0);
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
                float v = acc[a][b][r];
                if (TWO) v = __builtin_fmaf(acl[a][b][r], 1.f / 2048.f, v);
                if (row < M && col < N) C[(long)row * N + col] = v;
            }
}


// ---------------------------------------------------------------------------- software-pipelined form
// Two LDS buffers and ONE barrier per K-tile: while the MFMAs of tile kt run out of buffer kt & 1, the same wave splits the
// registers holding tile kt+1 into buffer (kt+1) & 1 (the VALU work sits in the shadow of the matrix pipe instead of in its own
// phase between two barriers), then issues the global loads of tile kt+2.  PIPE = 1 pins the interleaving with
// sched_group_barrier (one MFMA, then a few VALU / DS instructions), PIPE = 0 leaves it to the compiler.
template <int BM, int BN, int WGM, int WGN, int PIPE>
__global__ __launch_bounds__(256, 2) void rowk_db(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16, NA = BM / 32, NB = BN / 32;
    constexpr int ABUF = 2 * BM * LDH, BBUF = 2 * BN * LDH;      // one buffer = two planes
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    const int l16 = lane & 15, kg = lane >> 4;
    float4 ra[NA], rb[NB];
#define LOADT(kt_)                                                                                                                     \
    {                                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + (kt_) * BK); \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + (kt_) * BK); \
    }
#define STORET(buf_)                                                                                               \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
            const int o = (buf_) * ABUF + (trow + 32 * i) * LDH + wo;                                              \
            uint2 h, l;                                                                                            \
            hsplit4(ra[i], h, l);                                                                                  \
            *reinterpret_cast<uint2 *>(&As[o]) = h;                                                                \
            *reinterpret_cast<uint2 *>(&As[o + BM * LDH]) = l;                                                     \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                           \
            const int o = (buf_) * BBUF + (trow + 32 * i) * LDH + wo;                                              \
            uint2 h, l;                                                                                            \
            hsplit4(rb[i], h, l);                                                                                  \
            *reinterpret_cast<uint2 *>(&Bs[o]) = h;                                                                \
            *reinterpret_cast<uint2 *>(&Bs[o + BN * LDH]) = l;                                                     \
        }                                                                                                          \
    }
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
    LOADT(0)
    STORET(0)
    LOADT(nk > 1 ? 1 : 0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned short *Ac = As + cur * ABUF, *Bc = Bs + cur * BBUF;
        f16x8 bf[TN][2];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[b][p] = *reinterpret_cast<const f16x8 *>(&Bc[p * BN * LDH + bro + b * 16 * LDH]);
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            f16x8 af[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const f16x8 *>(&Ac[p * BM * LDH + aro + a * 16 * LDH]);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bf[b][0], acl[a][b], 0, 0, 0);
                acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][1], acl[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][0], acc[a][b], 0, 0, 0);
            }
            if (a == 1) {      // the staging of tile kt+1 is split over the MFMA block's first half ...
                if (kt + 1 < nk) STORET(cur ^ 1)
            }
            if (a == 2) {      // ... and the loads of tile kt+2 follow as soon as the registers are free
                LOADT(kt + 2 < nk ? kt + 2 : kt)
            }
        }
        if (PIPE == 1) {
            // 48 MFMAs; ~100 VALU, 16 DS writes, 16 DS reads, 8 VMEM reads to place between them
#pragma unroll
            for (int i = 0; i < TM * TN * 3; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // up to three VALU
                __builtin_amdgcn_sched_group_barrier(0x300, 1, 0);      // one DS read or write
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one VMEM read
            }
        }
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
                if (row < M && col < N) C[(long)row * N + col] = __builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]);
            }
#undef LOADT
#undef STORET
}


// ---------------------------------------------------------------------------- A operand straight into the MFMA registers
// With the four waves of a workgroup stacked along M (256 x 64 tile) no two waves share a row of A, so A needs no LDS at all:
// lane (row l16, k group kg) of a wave loads the 8 consecutive k of its row (32 bytes) for each of its four 16-row tiles -- the
// MFMA operand layout -- and splits them in registers.  Only the 64 weight rows go through LDS (written once, read by all four
// waves): 40 KB of LDS traffic per K-tile instead of 96 KB for the same 1 MFLOP.
template <int TN_>
__global__ __launch_bounds__(256, 2) void rowk_adirect(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 32, BM = 256, BN = 16 * TN_, TM = 4, TN = TN_, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM + wave * 64, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);
    const int l16 = lane & 15, kg = lane >> 4;
    const float *arow = A + (long)(m0 + l16) * K + kg * 8;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    float4 ra[TM][2], rb[NB];
#define LOADT(kt_)                                                                                                                     \
    {                                                                                                                                  \
        _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                                               \
            ra[a][0] = *reinterpret_cast<const float4 *>(arow + (long)16 * a * K + (kt_) * BK);                                        \
            ra[a][1] = *reinterpret_cast<const float4 *>(arow + (long)16 * a * K + (kt_) * BK + 4);                                    \
        }                                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + (kt_) * BK); \
    }
    f32x4 acc[TM][TN], acl[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = acl[a][b][r] = 0.f;
    const int bro = l16 * LDH + ((kg ^ swz(l16)) << 3);
    const int nk = K / BK;
    LOADT(0)
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int o = (trow + 32 * i) * LDH + wo;
            uint2 h, l;
            hsplit4(rb[i], h, l);
            *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
            *reinterpret_cast<uint2 *>(&Bs[1][o]) = l;
        }
        f16x8 af[TM][2];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            uint2 h0, l0, h1, l1;
            hsplit4(ra[a][0], h0, l0);
            hsplit4(ra[a][1], h1, l1);
            af[a][0] = __builtin_bit_cast(f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
            af[a][1] = __builtin_bit_cast(f16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
        }
        __syncthreads();
        LOADT(kt + 1 < nk ? kt + 1 : kt)
        __builtin_amdgcn_sched_barrier(0);
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
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + a * 16 + 4 * kg + r;
                int col = n0 + b * 16 + l16;
                if (row < M && col < N) C[(long)row * N + col] = __builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]);
            }
#undef LOADT
}


// ---------------------------------------------------------------------------- eight waves per workgroup
// Same 128 x 128 tile and the same bytes per FLOP from L2, but 512 threads: wave tiles of 32 x 64 (64 accumulator registers instead
// of 128), so four waves fit a SIMD where the 4-wave form fits two -- more waves to cover the waits of the load path.
template <int BM, int BN, int WGM, int WGN, int MINB>
__global__ __launch_bounds__(512, MINB) void rowk_w8(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16, NA = BM / 64, NB = BN / 64;
    static_assert(WGM * WGN == 8, "8 waves");
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
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;      // 64 rows per pass
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    const int l16 = lane & 15, kg = lane >> 4;
    float4 ra[NA], rb[NB];
#define LOADT(kt_)                                                                                                                     \
    {                                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)64 * i * K + (kt_) * BK); \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * K + (kt_) * BK); \
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
        for (int i = 0; i < NA; ++i) {
            const int o = (trow + 64 * i) * LDH + wo;      // swz(row + 64) == swz(row)
            uint2 h, l;
            hsplit4(ra[i], h, l);
            *reinterpret_cast<uint2 *>(&As[0][o]) = h;
            *reinterpret_cast<uint2 *>(&As[1][o]) = l;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int o = (trow + 64 * i) * LDH + wo;
            uint2 h, l;
            hsplit4(rb[i], h, l);
            *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
            *reinterpret_cast<uint2 *>(&Bs[1][o]) = l;
        }
        __syncthreads();
        LOADT(kt + 1 < nk ? kt + 1 : kt)
        __builtin_amdgcn_sched_barrier(0);
        {
            f16x8 bf[TN][2];
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[b][p] = *reinterpret_cast<const f16x8 *>(&Bs[p][bro + b * 16 * LDH]);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                f16x8 af[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const f16x8 *>(&As[p][aro + a * 16 * LDH]);
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1], bf[b][0], acl[a][b], 0, 0, 0);
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][1], acl[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bf[b][0], acc[a][b], 0, 0, 0);
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
                if (row < M && col < N) C[(long)row * N + col] = __builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]);
            }
#undef LOADT
}


// ---------------------------------------------------------------------------- eight waves + two LDS buffers, one barrier per K-tile
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(512, 4) void rowk_w8db(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16, NA = BM / 64, NB = BN / 64;
    constexpr int ABUF = 2 * BM * LDH, BBUF = 2 * BN * LDH;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * ABUF];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * BBUF];
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
    float4 ra[NA], rb[NB];
#define LOADT(kt_)                                                                                                                     \
    {                                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)64 * i * K + (kt_) * BK); \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)64 * i * K + (kt_) * BK); \
    }
#define STORET(buf_)                                                                                               \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
            const int o = (buf_) * ABUF + (trow + 64 * i) * LDH + wo;                                              \
            uint2 h, l;                                                                                            \
            hsplit4(ra[i], h, l);                                                                                  \
            *reinterpret_cast<uint2 *>(&As[o]) = h;                                                                \
            *reinterpret_cast<uint2 *>(&As[o + BM * LDH]) = l;                                                     \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                           \
            const int o = (buf_) * BBUF + (trow + 64 * i) * LDH + wo;                                              \
            uint2 h, l;                                                                                            \
            hsplit4(rb[i], h, l);                                                                                  \
            *reinterpret_cast<uint2 *>(&Bs[o]) = h;                                                                \
            *reinterpret_cast<uint2 *>(&Bs[o + BN * LDH]) = l;                                                     \
        }                                                                                                          \
    }
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
    LOADT(0)
    STORET(0)
    LOADT(nk > 1 ? 1 : 0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const unsigned short *Ac = As + cur * ABUF, *Bc = Bs + cur * BBUF;
        if (kt + 1 < nk) STORET(cur ^ 1)      // tile kt+1 (registers) into the other buffer; nobody reads it before the barrier below
        LOADT(kt + 2 < nk ? kt + 2 : kt)
        {
            f16x8 af[TM][2];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int p = 0; p < 2; ++p) af[a][p] = *reinterpret_cast<const f16x8 *>(&Ac[p * BM * LDH + aro + a * 16 * LDH]);
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                f16x8 bf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(&Bc[p * BN * LDH + bro + b * 16 * LDH]);
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][1], bf[0], acl[a][b], 0, 0, 0);
                    acl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[1], acl[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[0], acc[a][b], 0, 0, 0);
                }
            }
        }
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
                if (row < M && col < N) C[(long)row * N + col] = __builtin_fmaf(acl[a][b][r], 1.f / 2048.f, acc[a][b][r]);
            }
#undef LOADT
#undef STORET
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

// error of a plain fp32 dot in k order against the same double reference, for scale
static double fp32_err(const std::vector<float> &A, const std::vector<float> &B, int M, int N, int K, double *rms_out) {
    double worst = 0., ss = 0.;
    long cnt = 0;
    for (int s = 0; s < 32; ++s) {
        int r = (int)((long)s * 7919 % M);
        for (int c = 0; c < N; c += 5) {
            double ref = 0., mag = 0.;
            float f = 0.f;
            for (int k = 0; k < K; ++k) {
                double p = (double)A[(long)r * K + k] * (double)B[(long)c * K + k];
                ref += p; mag += fabs(p);
                f = fmaf(A[(long)r * K + k], B[(long)c * K + k], f);
            }
            double e = fabs((double)f - ref) / mag;
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

template <int MODE>
static void variant(const char *name, const float *A, const float *B, const unsigned short *Bp, float *C, int M, int N, int K, const std::vector<float> &hA,
                    const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL((rowk<128, 128, 2, 2, MODE>), grid, dim3(256), 0, 0, A, B, Bp, C, M, N, K); }, 20);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-44s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
    fflush(stdout);
}

template <int K_>
static void variant_x(const char *name, const float *A, const float *B, const unsigned short *Bp, float *C, int M, int N, int K, const std::vector<float> &hA,
                      const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    dim3 grid((N + 63) / 64, (M + 255) / 256);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] {
        if (K_ == 0) hipLaunchKernelGGL((rowk<256, 64, 4, 1, 1>), grid, dim3(256), 0, 0, A, B, Bp, C, M, N, K);
        else hipLaunchKernelGGL((rowk_adirect<4>), grid, dim3(256), 0, 0, A, B, C, M, N, K);
    }, 20);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-44s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
    fflush(stdout);
}

template <int BM, int BN, int WGM, int WGN, int MINB>
static void variant_w8(const char *name, const float *A, const float *B, float *C, int M, int N, int K, const std::vector<float> &hA,
                       const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL((rowk_w8<BM, BN, WGM, WGN, MINB>), grid, dim3(512), 0, 0, A, B, C, M, N, K); }, 20);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-44s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
    fflush(stdout);
}

static void variant_w8db(const char *name, const float *A, const float *B, float *C, int M, int N, int K, const std::vector<float> &hA,
                         const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL((rowk_w8db<128, 128, 4, 2>), grid, dim3(512), 0, 0, A, B, C, M, N, K); }, 20);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-44s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
    fflush(stdout);
}

template <int PIPE>
static void variant_db(const char *name, const float *A, const float *B, float *C, int M, int N, int K, const std::vector<float> &hA,
                       const std::vector<float> &hB, std::vector<float> &hC) {
    const double flops = 2.0 * M * N * K;
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    (void)hipMemset(C, 0, (size_t)M * N * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL((rowk_db<128, 128, 2, 2, PIPE>), grid, dim3(256), 0, 0, A, B, C, M, N, K); }, 20);
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
    printf("M %6d N %4d K %4d  %-44s %8.3f ms %7.1f TF   err/sum|ab| worst %.3g rms %.3g\n", M, N, K, name, ms, flops / ms / 1e9, w, rms);
    fflush(stdout);
}

static void run(int M, int N, int K, float amag, float bmag) {
    std::vector<float> hA((long)M * K), hB((long)N * K), hC((long)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = amag * rnd() * (1.f + 0.37f * rnd());
    for (auto &v : hB) v = bmag * rnd() * (1.f + 0.11f * rnd());
    float *A, *B, *C;
    (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&B, hB.size() * 4); (void)hipMalloc(&C, hC.size() * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    double rms, w = fp32_err(hA, hB, M, N, K, &rms);
    printf("operand magnitudes %.3g x %.3g; plain fp32 fma chain: err/sum|ab| worst %.3g rms %.3g\n", amag, bmag, w, rms);
    unsigned short *Bp;
    (void)hipMalloc(&Bp, hB.size() * 4);
    hipLaunchKernelGGL(hpresplit_kernel, dim3((unsigned)((hB.size() / 4 + 255) / 256)), dim3(256), 0, 0, B, Bp, (long)hB.size());
    variant<0>("six bf16 products (today)", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant<1>("three fp16 products, scaled low plane", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant<3>("three fp16 products, weights pre-split", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant<7>("three fp16 products, XCD-aware tile order", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant_w8<128, 128, 4, 2, 4>("8 waves, 128x128, 32x64 wave tiles, 2 WG/CU", A, B, C, M, N, K, hA, hB, hC);
    variant_w8db("8 waves, 128x128, two LDS buffers, one barrier", A, B, C, M, N, K, hA, hB, hC);
    variant_w8<128, 128, 2, 4, 4>("8 waves, 128x128, 64x32 wave tiles, 2 WG/CU", A, B, C, M, N, K, hA, hB, hC);
    variant_w8<256, 128, 4, 2, 1>("8 waves, 256x128, 64x64 wave tiles, 1 WG/CU", A, B, C, M, N, K, hA, hB, hC);
    variant<6>("(speed only) no global loads in the loop", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant_x<0>("three fp16, 256x64 tile, both through LDS", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant_x<1>("three fp16, 256x64 tile, A direct to registers", A, B, Bp, C, M, N, K, hA, hB, hC);
    variant_db<0>("three fp16, two LDS buffers, one barrier", A, B, C, M, N, K, hA, hB, hC);
    variant_db<1>("  + pinned MFMA / VALU interleaving", A, B, C, M, N, K, hA, hB, hC);
    variant<2>("(speed only) three products, one accumulator", A, B, Bp, C, M, N, K, hA, hB, hC);
    (void)hipMemset(A, 0, hA.size() * 4);
    std::vector<float> z(hA.size(), 0.f);
    variant<0>("six bf16 products, A = 0", A, B, Bp, C, M, N, K, z, hB, hC);
    variant<1>("three fp16 products, A = 0", A, B, Bp, C, M, N, K, z, hB, hC);
    (void)hipFree(Bp);
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
}

int main() {
    run(40960, 512, 1600, 1.f, 0.05f);      // dense1 patch forward
    run(40960, 512, 256, 1.f, 0.05f);       // pol1 / v1 forward
    run(40960, 1664, 512, 1.f, 0.05f);      // dense1 patch data gradient (N = 1600 padded to the tile)
    run(40960, 512, 1600, 1e-3f, 0.05f);    // small operands: the scaled low plane keeps the precision ...
    run(8192, 512, 1600, 1e-5f, 0.05f);     // ... h is a subnormal fp16 here (the matrix pipe must not flush it) ...
    run(8192, 512, 1600, 1e-7f, 0.05f);     // ... and here h is 0 or one subnormal step: the low plane carries the value
    return 0;
}
