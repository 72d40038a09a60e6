// microbenchmark: fp32 GEMM C = A * Bt^T computed on the bf16 matrix pipe with no loss of input precision.
// Every fp32 operand is split EXACTLY into three bf16 terms x = h + m + l (8 + 8 + 8 significand bits, by truncation);
// six of the nine partial products (hh, hm, mh, mm, hl, lh -- everything down to 2^-24 |a||b|) are accumulated in fp32
// by the bf16 MFMA, which runs 16x the rate of v_mfma_f32_32x32x2_f32: 6 instructions replace 8.
// Compares speed and error (against a double-precision CPU reference) of
//   rowk_f32      the plain fp32-MFMA loop (the first version of net_gemm.h)
//   rowk_x6       six products on v_mfma_f32_32x32x16_bf16, padded 80-byte LDS rows; MODE bits switch off global loads /
//                 the split / the scheduling fences (what each ingredient costs)
//   rowk_x6p      the same software-pipelined over two LDS buffers, one workgroup per CU (not faster)
//   rowk_x6_16    six products on v_mfma_f32_16x16x32_bf16 (what net_gemm.h uses), padded 96-byte rows or the XOR-swizzled
//                 64-byte rows, optionally with the ReLU-mask epilogue of the data-gradient GEMMs
// Build: hipcc -O3 --offload-arch=gfx950 gemm_x6.hip -o gemm_x6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split4(const float4 v, uint2 &h, uint2 &m, uint2 &l) {
    const unsigned x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
    // upper halves of two floats -> one dword of two bf16 (truncation)
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

template <int BM, int BN, int WGM, int WGN, int MODE = 0>
__global__ __launch_bounds__(256, 2) void rowk_x6(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 40;      // bf16 per LDS row (80 B: the 16 lanes of every ds_read_b128 group hit 16 distinct 4-bank slots)
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32, NA = BM / 32, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short As[3][BM * LDH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    float4 ra[NA], rb[NB], ra2[NA], rb2[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K);
    if (MODE & 8) {
#pragma unroll
    for (int i = 0; i < NA; ++i) ra2[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + BK);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb2[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + BK);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    const int aro = (wm * WM + lr) * LDH + lk * 8, bro = (wn * WN + lr) * LDH + lk * 8;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            uint2 h, m, l;
            if (MODE & 2) { h = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y)); m = make_uint2(__float_as_uint(ra[i].z), __float_as_uint(ra[i].w)); l = h; }
            else split4(ra[i], h, m, l);
            const int o = (trow + 32 * i) * LDH + tk4;
            *reinterpret_cast<uint2 *>(&As[0][o]) = h;
            *reinterpret_cast<uint2 *>(&As[1][o]) = m;
            *reinterpret_cast<uint2 *>(&As[2][o]) = l;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            uint2 h, m, l;
            if (MODE & 2) { h = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y)); m = make_uint2(__float_as_uint(rb[i].z), __float_as_uint(rb[i].w)); l = h; }
            else split4(rb[i], h, m, l);
            const int o = (trow + 32 * i) * LDH + tk4;
            *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
            *reinterpret_cast<uint2 *>(&Bs[1][o]) = m;
            *reinterpret_cast<uint2 *>(&Bs[2][o]) = l;
        }
        __syncthreads();
        const int ktn = (MODE & 8) ? (kt + 2 < nk ? kt + 2 : kt) : (kt + 1 < nk ? kt + 1 : kt);
        if (MODE & 8) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { ra[i] = ra2[i]; ra2[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + ktn * BK); }
#pragma unroll
        for (int i = 0; i < NB; ++i) { rb[i] = rb2[i]; rb2[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + ktn * BK); }
        } else
        if (!(MODE & 1)) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + ktn * BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + ktn * BK);
        }
        if (!(MODE & 4)) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const bf16x8 *>(&As[p][aro + a * 32 * LDH + s * 16]);
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8 *>(&Bs[p][bro + b * 32 * LDH + s * 16]);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);
                }
        }
        if (!(MODE & 4)) __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = n0 + wn * WN + b * 32 + lr;
                if (row < M && col < N) C[(long)row * N + col] = acc[a][b][r];
            }
}


// software-pipelined variant: LDS double-buffered, ONE workgroup per CU; while the MFMAs consume tile kt from buffer kt&1
// the same wave splits tile kt+1 (already in registers) into buffer (kt+1)&1 and tile kt+2 is in flight from global memory.
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
template <int BM, int BN, int WGM, int WGN, int MODE = 0>
__global__ __launch_bounds__(256, 1) void rowk_x6p(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LDH = 40;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32, NA = BM / 32, NB = BN / 32;
    constexpr int ASZ = BM * LDH, BSZ = BN * LDH;
    __shared__ __attribute__((aligned(16))) unsigned short As[2][3][ASZ];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][3][BSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    const int nk = K / BK;
#define LOADSET(RA, RB, kt_)                                                                                                  \
    {                                                                                                                         \
        const int kk = (kt_) < nk ? (kt_) : nk - 1;                                                                           \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) RA[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + kk * BK); \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) RB[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + kk * BK); \
    }
#define SPLITSET(RA, RB, buf_)                                                                                                \
    {                                                                                                                         \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                                      \
            uint2 h, m, l;                                                                                                    \
            split4(RA[i], h, m, l);                                                                                           \
            const int o = (trow + 32 * i) * LDH + tk4;                                                                        \
            *reinterpret_cast<uint2 *>(&As[buf_][0][o]) = h;                                                                  \
            *reinterpret_cast<uint2 *>(&As[buf_][1][o]) = m;                                                                  \
            *reinterpret_cast<uint2 *>(&As[buf_][2][o]) = l;                                                                  \
        }                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                                      \
            uint2 h, m, l;                                                                                                    \
            split4(RB[i], h, m, l);                                                                                           \
            const int o = (trow + 32 * i) * LDH + tk4;                                                                        \
            *reinterpret_cast<uint2 *>(&Bs[buf_][0][o]) = h;                                                                  \
            *reinterpret_cast<uint2 *>(&Bs[buf_][1][o]) = m;                                                                  \
            *reinterpret_cast<uint2 *>(&Bs[buf_][2][o]) = l;                                                                  \
        }                                                                                                                     \
    }
#define MFMASET(buf_)                                                                                                         \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                           \
        bf16x8 af[TM][3], bf[TN][3];                                                                                          \
        _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                        \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const bf16x8 *>(&As[buf_][p][aro + a * 32 * LDH + s * 16]); \
        _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                                        \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8 *>(&Bs[buf_][p][bro + b * 32 * LDH + s * 16]); \
        _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                        \
            _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                                  \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][2], bf[b][0], acc[a][b], 0, 0, 0);                  \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][2], acc[a][b], 0, 0, 0);                  \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][1], acc[a][b], 0, 0, 0);                  \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][1], bf[b][0], acc[a][b], 0, 0, 0);                  \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][1], acc[a][b], 0, 0, 0);                  \
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a][0], bf[b][0], acc[a][b], 0, 0, 0);                  \
            }                                                                                                                 \
    }
#define HINTS()                                                                                                               \
    if (MODE & 1) {                                                                                                           \
        _Pragma("unroll") for (int g = 0; g < 24; ++g) {                                                                      \
            SGB(0x008, 1); SGB(0x100, 1); SGB(0x002, 4); SGB(0x008, 1); SGB(0x200, 1); SGB(0x002, 4);                         \
        }                                                                                                                     \
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    const int aro = (wm * WM + lr) * LDH + lk * 8, bro = (wn * WN + lr) * LDH + lk * 8;
    LOADSET(ra0, rb0, 0)
    LOADSET(ra1, rb1, 1)
    SPLITSET(ra0, rb0, 0)
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {      // nk even
        LOADSET(ra0, rb0, kt + 2)
        MFMASET(0)
        SPLITSET(ra1, rb1, 1)
        HINTS()
        __syncthreads();
        LOADSET(ra1, rb1, kt + 3)
        MFMASET(1)
        SPLITSET(ra0, rb0, 0)
        HINTS()
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = n0 + wn * WN + b * 32 + lr;
                if (row < M && col < N) C[(long)row * N + col] = acc[a][b][r];
            }
}


// variant on v_mfma_f32_16x16x32_bf16 (more FLOP per watt than 32x32x16 on toggling data): wave tile 64x64 = 4x4 tiles of
// 16x16, one K = 32 step per BK tile; LDS rows of 48 bf16 (96 B) make the b128 fragment reads conflict free for this map
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }
template <int BM, int BN, int WGM, int WGN, bool SWZ = false>
__global__ __launch_bounds__(256, 2) void rowk_x6_16(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K, const float *__restrict__ fwd = nullptr) {
    constexpr int BK = 32, LDH = SWZ ? 32 : 48;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16, NA = BM / 32, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short As[3][BM * LDH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    float4 ra[NA], rb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K);
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    const int l16 = lane & 15, kg = lane >> 4;
    const int rofs = SWZ ? ((kg ^ swz(l16)) << 3) : kg * 8;
    const int wo = SWZ ? ((((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4)) : tk4;
    const int aro = (wm * WM + l16) * LDH + rofs, bro = (wn * WN + l16) * LDH + rofs;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            uint2 h, m, l;
            split4(ra[i], h, m, l);
            const int o = (trow + 32 * i) * LDH + wo;
            *reinterpret_cast<uint2 *>(&As[0][o]) = h;
            *reinterpret_cast<uint2 *>(&As[1][o]) = m;
            *reinterpret_cast<uint2 *>(&As[2][o]) = l;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            uint2 h, m, l;
            split4(rb[i], h, m, l);
            const int o = (trow + 32 * i) * LDH + wo;
            *reinterpret_cast<uint2 *>(&Bs[0][o]) = h;
            *reinterpret_cast<uint2 *>(&Bs[1][o]) = m;
            *reinterpret_cast<uint2 *>(&Bs[2][o]) = l;
        }
        __syncthreads();
        const int ktn = kt + 1 < nk ? kt + 1 : kt;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + ktn * BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + ktn * BK);
        __builtin_amdgcn_sched_barrier(0);
        {
            bf16x8 bf[TN][3];
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[b][p] = *reinterpret_cast<const bf16x8 *>(&Bs[p][bro + b * 16 * LDH]);
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
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = m0 + wm * WM + a * 16 + 4 * kg + r;
                int col = n0 + wn * WN + b * 16 + l16;
                if (row < M && col < N) { float v = acc[a][b][r]; if (fwd) v = fwd[(long)row * N + col] > 0.f ? v : 0.f; C[(long)row * N + col] = v; }
            }
}

// the production fp32 loop (net_gemm.h gemm_rowk) on the same dense operands
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void rowk_f32(const float *__restrict__ A, const float *__restrict__ Bt, float *__restrict__ C, int M, int N, int K) {
    constexpr int BK = 32, LD = 36;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32, NA = BM / 32, NB = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const float *arow = A + (long)(m0 + trow) * K + tk4;
    const float *brow = Bt + (long)(n0 + trow) * K + tk4;
    float4 ra[NA], rb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lr = lane & 31, lk = lane >> 5;
    const float *ap = As + (wm * WM + lr) * LD + lk * 16, *bp = Bs + (wn * WN + lr) * LD + lk * 16;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<float4 *>(As + (trow + 32 * i) * LD + tk4) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4 *>(Bs + (trow + 32 * i) * LD + tk4) = rb[i];
        __syncthreads();
        const int ktn = kt + 1 < nk ? kt + 1 : kt;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4 *>(arow + (long)32 * i * K + ktn * BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4 *>(brow + (long)32 * i * K + ktn * BK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float4 *>(ap + a * 32 * LD + q * 4);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float4 *>(bp + b * 32 * LD + q * 4);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
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
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                int col = n0 + wn * WN + b * 32 + lr;
                if (row < M && col < N) C[(long)row * N + col] = acc[a][b][r];
            }
}

static double err_vs_double(const std::vector<float> &A, const std::vector<float> &B, const std::vector<float> &C, int M, int N, int K,
                            double *rms_out) {
    // error of 64 sampled rows relative to sum_k |a||b| (the natural scale of rounding in a dot product)
    double worst = 0., ss = 0.;
    long cnt = 0;
    for (int s = 0; s < 64; ++s) {
        int r = (int)((long)s * 7919 % M);
        for (int c = 0; c < N; c += 3) {
            double ref = 0., mag = 0.;
            for (int k = 0; k < K; ++k) {
                double p = (double)A[(long)r * K + k] * (double)B[(long)c * K + k];
                ref += p;
                mag += fabs(p);
            }
            double e = fabs((double)C[(long)r * N + c] - ref) / mag;
            worst = fmax(worst, e);
            ss += e * e;
            ++cnt;
        }
    }
    *rms_out = sqrt(ss / cnt);
    return worst;
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

static void run(int M, int N, int K, float scale = 1.f) {
    std::vector<float> hA((long)M * K), hB((long)N * K), hC((long)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = scale * rnd() * (1.f + 0.37f * rnd());
    for (auto &v : hB) v = scale * 0.05f * rnd() * (1.f + 0.11f * rnd());
    float *A, *B, *C;
    (void)hipMalloc(&A, hA.size() * 4);
    (void)hipMalloc(&B, hB.size() * 4);
    (void)hipMalloc(&C, hC.size() * 4);
    (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    const double flops = 2.0 * M * N * K;
    {
        dim3 g(N / 128, M / 128);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_f32<128, 128, 2, 2>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  f32 mfma  128x128: %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, ms, flops / ms / 1e9, w, rms);
    }
    {
        dim3 g(N / 128, M / 128);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  bf16 x6   128x128: %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, ms, flops / ms / 1e9, w, rms);
    }
    for (int mode = 1; mode < 8; ++mode) {
        dim3 g(N / 128, M / 128);
        float ms = 0;
        switch (mode) {
        case 1: ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2, 1>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5); break;
        case 2: ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2, 2>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5); break;
        case 3: ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2, 3>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5); break;
        case 4: ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2, 4>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5); break;
        case 5: ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2, 8>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5); break;
        case 6: ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<128, 128, 2, 2, 12>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5); break;
        default: continue;
        }
        printf("   mode %d (1: no global loads, 2: no split, 4: no sched barriers; mode 5 = 2-deep prefetch, 6 = 2-deep + no fences): %8.3f ms %7.1f TF\n", mode, ms, flops / ms / 1e9);
    }
    {
        dim3 g(N / 128, M / 128);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6_16<128, 128, 2, 2>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  x6 16x16x32  128x128: %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, ms, flops / ms / 1e9, w, rms);
    }
    {
        dim3 g(N / 128, M / 128);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6_16<128, 128, 2, 2, true>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  x6 16x16x32 swizzled 128x128: %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, ms, flops / ms / 1e9, w, rms);
    }
    {
        float *F;
        (void)hipMalloc(&F, hC.size() * 4);
        (void)hipMemcpy(F, hA.data(), (hC.size() < hA.size() ? hC.size() : hA.size()) * 4, hipMemcpyHostToDevice);
        dim3 g(N / 128, M / 128);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6_16<128, 128, 2, 2, true>), g, dim3(256), 0, 0, A, B, C, M, N, K, F); }, 5);
        printf("M %6d N %4d K %4d  x6 16x16x32 swizzled 128x128 + ReLU-mask epilogue: %8.3f ms %7.1f TF\n", M, N, K, ms, flops / ms / 1e9);
        (void)hipFree(F);
    }
    {
        dim3 g(N / 64, M / 256);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6_16<256, 64, 4, 1, true>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  x6 16x16x32 swizzled 256x64 : %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, ms, flops / ms / 1e9, w, rms);
    }
    for (int v = 0; v < 2; ++v) {
        dim3 g(N / 128, M / 128);
        float ms = v == 0 ? time_ms([&] { hipLaunchKernelGGL((rowk_x6p<128, 128, 2, 2, 0>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5)
                          : time_ms([&] { hipLaunchKernelGGL((rowk_x6p<128, 128, 2, 2, 1>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  x6 pipelined%s 128x128: %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, v ? " +hints" : "       ", ms, flops / ms / 1e9, w, rms);
    }
    {
        dim3 g(N / 64, M / 256);
        float ms = time_ms([&] { hipLaunchKernelGGL((rowk_x6<256, 64, 4, 1>), g, dim3(256), 0, 0, A, B, C, M, N, K); }, 5);
        (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        double rms, w = err_vs_double(hA, hB, hC, M, N, K, &rms);
        printf("M %6d N %4d K %4d  bf16 x6   256x64 : %8.3f ms %7.1f TF  err/sum|ab| worst %.3g rms %.3g\n", M, N, K, ms, flops / ms / 1e9, w, rms);
    }
    (void)hipFree(A);
    (void)hipFree(B);
    (void)hipFree(C);
}

int main() {
    run(40960, 512, 256);      // dense2 data gradient: K = 256, N = 512
    run(40960, 256, 1024);     // pol1 | v1 pair
    run(40960, 512, 1600);
    return 0;
}
