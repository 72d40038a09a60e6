// fp32 GEMM building blocks for the policy/value nets on the gfx950 matrix cores.
//
// All dense contractions of ConvSingleAgentPolicyNetwork (reference
// fed_gym/agents/paac/policy_v_network.py:14-59) -- conv2, conv3 as implicit GEMMs over NHWC
// activations, the dense stack, and their data/weight gradients -- go through two kernels:
//
//   gemm_rowk<BM,BN,WGM,WGN>   C[M,N] = A[M,K] * Bt[N,K]^T   (A rows gathered; both operands K-contiguous)
//        forward: Bt = W^T (a transposed copy of the layer's kernel kept beside the parameters)
//        data gradient: Bt = W itself (dX = dY * W^T), or a rearranged kernel for the transposed convs
//   gemm_tn<BM,BN,WGM,WGN>     C[I,J] = sum_m A[m,I] * B[m,J]   (weight gradient, split over m into slabs)
//
// Arithmetic.  The reference computes in float32, and so do these kernels -- but on the fp16 matrix pipe, which on
// CDNA4 runs 16x the rate of v_mfma_f32_32x32x2_f32 (2.5 PFLOP/s against 157 TFLOP/s).  While a tile is staged into
// LDS every fp32 operand x is split into two fp16 numbers: h = fp16(x) (round to nearest) and l' = fp16((x - h) * 2^11)
// (x - h is exact in fp32; the scale keeps l' a normal fp16 number whenever h is one), 22-23 significand bits together.
// A product a * b is then h_a h_b + 2^-11 (h_a l'_b + l'_a h_b) + 2^-22 l'_a l'_b: the first three terms run on
// v_mfma_f32_16x16x32_f16 -- hh into one fp32 accumulator set, the two cross terms into a second one that is folded in
// with one fma by 2^-11 after the K loop -- and the last one (below one fp32 ulp of the product) is dropped.  Three
// fp16 MFMAs per K = 32 step; round 1's exact three-way bf16 split needed six (hh hm mh mm hl lh) and was pinned at
// the power-limited rate of the matrix pipe on toggling operands, so halving the MFMA count is worth 1.35-1.47x on the
// dense1 shapes (tools/ubench/gemm_f16x3.hip: 211-238 against 155-161 TFLOP/s of fp32 work, random data) with a
// SMALLER error against a double-precision reference (rms relative to sum|a||b| 0.9-1.1e-8 against 2.0-2.1e-8; a plain
// fp32 fma chain: 2.5e-8).
// Range.  fp16 holds 2^-14 .. 65504 at full precision; the matrix pipe keeps subnormal inputs (checked in the same
// microbenchmark), so smaller operands degrade gracefully (absolute error 2^-36 per element, i.e. full precision for
// every element within 2^-20 of the tensor's largest when that is ~2^6) and an operand above 65504 becomes inf.
// Forward operands (activations, weights) sit inside that range by themselves; the backward pass is linear in the head
// gradients, which are scaled by a power of two chosen per update (net_train.inc: loss_scale_kernel) and unscaled
// exactly when the flat gradient is complete.
//
// 256 or 512 threads (4 or 8 waves, chosen per instance: see gemm_rowk), BK = 32, register prefetch of the next K-tile so the
// global loads fly under the MFMAs; an operand above the fp16 range is caught by the range guard in gemm_rowk's epilogue.  LDS
// holds two fp16 planes per operand in 64-byte rows with an XOR swizzle of the 16-byte chunks (kLdh, swz below): an
// MFMA fragment (8 consecutive k of one row) is one conflict-free ds_read_b128 per plane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

namespace grl {

// an epilogue may ask for the sign bits of what it stores, one 64-bit word per output row and 64-column block (kRowBits)
template <class E, class = void>
struct epi_row_bits : std::false_type {};
template <class E>
struct epi_row_bits<E, std::void_t<decltype(E::kRowBits)>> : std::bool_constant<E::kRowBits> {};

// an epilogue whose elem_aux is ADDED to the accumulator (bias, per-env part, shared pre-activation) says so: the range guard
// checks what is stored, not the bare product (kAddAux)
// kRowAuxN: row_aux_n(r, colbase) instead of row_aux(r), without the sign-bit output of kRowBits
template <class E, class = void>
struct epi_row_auxn : std::false_type {};
template <class E>
struct epi_row_auxn<E, std::void_t<decltype(E::kRowAuxN)>> : std::bool_constant<E::kRowAuxN> {};
template <class E, class = void>
struct epi_add_aux : std::false_type {};
template <class E>
struct epi_add_aux<E, std::void_t<decltype(E::kAddAux)>> : std::bool_constant<E::kAddAux> {};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Scheduling fences around gemm_rowk's MFMA block keep the next tile's global loads ahead of it (FENCE = true: +4-7 % on the dense
// forward / data-gradient instances); the weight-gradient loop (gemm_tn) and the dense1 patch forward are faster when the compiler
// interleaves staging, fragment reads and MFMAs itself (patch weight gradient: -22 %), measured per family on one box.
#ifndef GRL_SCHED_FENCE
#define GRL_SCHED_FENCE __builtin_amdgcn_sched_barrier(0);
#endif

// ---------------------------------------------------------------------------- gather descriptors
// Row r of the (virtual) operand = output pixel (n, qy, qx); reduction index k = ((ty*P + tx)*C + c)
// reads src[n][iy0+ty][ix0+tx][c] with (iy0, ix0) = (qy*SY + OY0, qx*SX + OX0) in an NHWC tensor
// [n][H][W][C].  CHECK=true zero-fills taps outside the image (transposed convolutions).
// RELU=true: the operand is relu(src): conv2's forward and weight gradient read a1sh = relu(sraw) straight from the pre-activation
template <int QH, int QW, int SY, int SX, int OY0, int OX0, int S, int P, int C, int H, int W, bool CHECK, bool RELU = false>
struct ConvGather {
    static constexpr int kPPS = QH * QW;
    static constexpr int kC = C;
    static constexpr bool kRelu = RELU;
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
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int) const { return true; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
    // gemm_tn: row handles are fetched one tile ahead of the data (ahandle / bhandle may load), rowh is arithmetic only
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(m, 0); }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }
    __device__ __forceinline__ int bhandle(int m) const { return m; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const { row(h.x, off, iy0, ix0); }
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const { mbeg = z * mc; mend = min(rows, mbeg + mc); }
};

// A convolution's forward / weight-gradient operand over a ROW LIST (net_shared.inc, trunk_index): only the (env, output) rows whose
// window holds an input pixel that differs from the env's background are computed; every other row of the chunk is the same vector
// (the layer applied to the constant background image), which one extra list entry (-1: the synthetic background image bg) computes
// with the same arithmetic and a fill kernel copies.  OW outputs per side, input image IW x IW x C, stride S, P x P taps.
template <int OW, int IW, int C, int S, int P, bool RELU>
struct ConvRowsList {
    static constexpr bool kRelu = RELU;
    static constexpr int kOut = OW * OW, kImg = IW * IW * C;
    const float *base;          // input [n][IW][IW][C]
    const int *rowlist;         // row r -> env * kOut + output, or -1 for the background row
    const int *rows_dev;        // live entries of rowlist
    const float *bg;            // one background image
    int rows;                   // upper bound
    int nz;                     // gemm_tn: row ranges of the launch
    __device__ __forceinline__ int K() const { return P * P * C; }
    __device__ __forceinline__ long off_of(int q) const {
        const int n = q < 0 ? 0 : q / kOut, o = q < 0 ? 0 : q - n * kOut, qy = o / OW, qx = o - qy * OW;
        return (q < 0 ? (long)(bg - base) : (long)n * kImg) + ((long)(qy * S) * IW + qx * S) * C;
    }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int q = r < *rows_dev ? rowlist[r] : -2;
        iy0 = ix0 = 0;
        off = 0;
        if (q == -2) { iy0 = -1; return; }      // behind the list: zero row, not stored
        off = off_of(q);
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        const int t = k0 / C, c0 = k0 - t * C;
        ty = t / P;
        tx = t - ty * P;
        toff = (ty * IW + tx) * C + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int, int, int) const { return iy0 >= 0; }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 < *rows_dev; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
    // gemm_tn (the layer's weight gradient over the affected rows; rows_dev then excludes the background entry): the live rows are
    // dealt evenly to the nz row ranges of the launch, whatever their number turns out to be on the device
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(rowlist[m], 0); }
    __device__ __forceinline__ bool hvalid(int2 h) const { return h.x >= 0; }
    __device__ __forceinline__ int bhandle(int m) const { return rowlist[m]; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const {
        iy0 = ix0 = 0;
        off = off_of(max(h.x, 0));
    }
    __device__ __forceinline__ void mrange(int z, int, int &mbeg, int &mend) const {
        const int live = *rows_dev, per = ((live + nz - 1) / nz + 31) / 32 * 32;
        mbeg = min(z * per, live);
        mend = min(live, mbeg + per);
    }
};
using GatherConv2ReluRows = ConvRowsList<9, 20, 32, 2, 4, true>;      // relu(sraw)[n][20][20][32] -> rows (n, o2), K = 512
using GatherConv3Rows = ConvRowsList<7, 9, 64, 1, 3, false>;           // a2sh[n][9][9][64] -> rows (n, o3), K = 576

// conv3's transposed convolution (GatherT3) over the list of affected conv2 rows: row r -> rowlist[r] = env * 81 + pixel
struct GatherT3Rows {
    static constexpr bool kRelu = false;
    const float *base;          // dz3[n][7][7][64]
    const int *rowlist, *rows_dev;
    int rows;                   // upper bound
    __device__ __forceinline__ int K() const { return 576; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int q = r < *rows_dev ? rowlist[r] : -1;
        if (q < 0) { iy0 = -64; ix0 = 0; off = 0; return; }      // behind the list: every tap invalid -> a zero row
        const int n = q / 81, o = q - n * 81, qy = o / 9, qx = o - qy * 9;
        iy0 = qy - 2;
        ix0 = qx - 2;
        off = (long)n * 3136 + ((long)iy0 * 7 + ix0) * 64;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        const int t = k0 >> 6, c0 = k0 & 63;
        ty = t / 3;
        tx = t - ty * 3;
        toff = (ty * 7 + tx) * 64 + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int ix0, int ty, int tx) const { return (unsigned)(iy0 + ty) < 7u && (unsigned)(ix0 + tx) < 7u; }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 < *rows_dev; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
};

// conv2's transposed convolution (GatherT2) over a list of touched 2x2 pixel blocks: row r -> blklist[r] = env * 100 + block
struct GatherT2Rows {
    static constexpr bool kRelu = false;
    const float *base;          // dz2[n][9][9][64]
    const int *blklist, *rows_dev;
    int rows;                   // upper bound
    __device__ __forceinline__ int K() const { return 256; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int q = r < *rows_dev ? blklist[r] : -1;
        if (q < 0) { iy0 = -64; ix0 = 0; off = 0; return; }      // behind the list: every tap invalid -> a zero row
        const int n = q / 100, b = q - n * 100, qy = b / 10, qx = b - qy * 10;
        iy0 = qy - 1;
        ix0 = qx - 1;
        off = (long)n * 5184 + ((long)iy0 * 9 + ix0) * 64;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        const int t = k0 >> 6, c0 = k0 & 63;
        ty = t >> 1;
        tx = t & 1;
        toff = (ty * 9 + tx) * 64 + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int ix0, int ty, int tx) const { return (unsigned)(iy0 + ty) < 9u && (unsigned)(ix0 + tx) < 9u; }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 < *rows_dev; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
};

// Pixel-major variant for stride-1 transposed convolutions: row r = q*nsamp + n (pixel q of sample n), so when
// nsamp is a multiple of the M tile every row of a workgroup is the SAME pixel and the validity of a tap is
// block-uniform: K-tiles whose tap falls outside the image are skipped instead of multiplied by zeros
// (conv3's data gradient: 441 of 729 (pixel, tap) combinations are inside).
template <int QH, int QW, int OY0, int OX0, int S, int P, int C, int H, int W>
struct ConvGatherPM {
    static constexpr bool kRelu = false;
    const float *base;
    int rows, nsamp;
    __device__ __forceinline__ int K() const { return S * P * C; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        int q = r / nsamp, n = r - q * nsamp;
        int qy = q / QW, qx = q - qy * QW;
        iy0 = qy + OY0;
        ix0 = qx + OX0;
        off = (long)n * (H * W * C) + ((long)iy0 * W + ix0) * C;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        int t = k0 / C;
        int c0 = k0 - t * C;
        ty = t / P;
        tx = t - ty * P;
        toff = (ty * W + tx) * C + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int ix0, int ty, int tx) const {
        return (unsigned)(iy0 + ty) < (unsigned)H && (unsigned)(ix0 + tx) < (unsigned)W;
    }
    __device__ __forceinline__ bool tile_ok(int m0, int k0) const {
        int q = m0 / nsamp, qy = q / QW, qx = q - qy * QW;
        int t = k0 / C, ty = t / P, tx = t - ty * P;
        return ok(qy + OY0, qx + OX0, ty, tx);
    }
    __device__ __forceinline__ bool tile_active(int) const { return true; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
    // gemm_tn: row handles are fetched one tile ahead of the data (ahandle / bhandle may load), rowh is arithmetic only
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(m, 0); }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }
    __device__ __forceinline__ int bhandle(int m) const { return m; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const { row(h.x, off, iy0, ix0); }
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const { mbeg = z * mc; mend = min(rows, mbeg + mc); }
};

struct DenseRows {   // plain row-major [rows][ld], reduction length k
    static constexpr bool kRelu = false;
    const float *base;
    int rows, ld, k;
    const int *rows_dev = nullptr;      // optional: the live row count on the device (tiles past it exit; rows is the bound)
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
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return !rows_dev || m0 < *rows_dev; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
    // gemm_tn: row handles are fetched one tile ahead of the data (ahandle / bhandle may load), rowh is arithmetic only
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(m, 0); }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }
    __device__ __forceinline__ int bhandle(int m) const { return m; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const { row(h.x, off, iy0, ix0); }
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const { mbeg = z * mc; mend = min(rows, mbeg + mc); }
};

// dense1's per-env GEMMs under the chunk's union mask U (net_shared.inc, trunk_index: bit p = conv3 output p is affected in SOME env
// of the chunk; everywhere else a3sh is the background row in every env).  The 3 136 inputs are 49 pixels x 64 channels, so a
// 64-wide tile of that axis is one pixel: the forward skips the K-tiles of the pixels outside U (their constant contribution rides
// in the bias), the data gradient does not compute their column tiles (only their sum over the envs is needed: closed form), the
// weight gradient writes no slab tile for their rows (rank-1 closed form).
struct DenseRowsKU : DenseRows {
    static constexpr int kMaxK = 64 * 64;      // wg_ctx carries u[0..1]: 64 pixel bits (checked where the instance is launched)
    const unsigned *u;
    __device__ __forceinline__ unsigned long long wg_ctx(int) const { return (unsigned long long)u[0] | ((unsigned long long)u[1] << 32); }
    __device__ __forceinline__ bool tile_ok_c(unsigned long long c, int k0) const { return (c >> (k0 >> 6)) & 1ull; }
    __device__ __forceinline__ int bk_c(unsigned long long, int k0) const { return k0; }
};
struct DenseRowsNU : DenseRows {
    const unsigned *u;
    __device__ __forceinline__ bool n_ok(int, int n0) const { return (u[n0 >> 11] >> ((n0 >> 6) & 31)) & 1u; }
};
struct DenseRowsIU : DenseRows {
    const unsigned *u;
    __device__ __forceinline__ bool i_ok(int, int i0) const { return (u[i0 >> 11] >> ((i0 >> 6) & 31)) & 1u; }
};

// `groups` row-major [npad][ld] operands stacked along M (rows = groups * npad, the first `nlive` rows of each live), every one
// against its own `bnstep`-row block of Bt -- the four conv1-pixel parity classes of the conv2 corrections in ONE launch
// instead of four launch-latency-bound ones.  gemm_tn: grid z = group * cpc + c reduces rows [c*mc, (c+1)*mc) of its group.
struct DenseRowsGroups {
    static constexpr bool kRelu = false;
    const float *base;
    int rows, ld, k, npad, nlive, bnstep, cpc;
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
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 % npad < nlive; }      // padding tiles of a group
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int m0) const { return n0 + (m0 / npad) * bnstep; }
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(m, 0); }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }
    __device__ __forceinline__ int bhandle(int m) const { return m; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const { row(h.x, off, iy0, ix0); }
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const {
        const int g = z / cpc, c = z - g * cpc;
        mbeg = g * npad + c * mc;
        mend = min(g * npad + nlive, mbeg + mc);
    }
};

// The class-major operand of conv2's per-agent corrections WITHOUT its copy in memory (round 3; gather_t2_kernel used to write
// at2[k][m][(ty, tx, co)] = the agent's masked dz2 row at conv2 output o(ty, tx), 335 MB per chunk, read back twice).  The agent's
// rows now sit in a canonical 3 x 3 grid dza[m][sy][sx][64] (cell = conv2 output relative to the agent's first reachable one,
// nothing written where it has none), so a row (class k, agent m) is an affine gather like SlotGatherT3P's: tap (ty, tx) reads cell
// (iy0 + ty, ix0 + tx) if that cell exists, rowdesc = {element offset of tap (0, 0), (iy0 << 16) | nr << 12 | nc << 8 | (ix0 & 0xff)}
// with nr x nc the agent's slot rectangle; invalid rows carry nr = nc = 0.
// Group semantics (one class per npad rows, its own bnstep-row block of Bt, row ranges per class) as DenseRowsGroups.
struct T2SlotGather {
    static constexpr bool kRelu = false;
    const float *base;
    const int2 *rowdesc;
    int rows, npad, nlive, bnstep, cpc;
    __device__ __forceinline__ int K() const { return 256; }
    __device__ __forceinline__ void rowh(int2 d, long &off, int &iy0, int &ix0) const {
        off = d.x;
        iy0 = d.y >> 16;
        ix0 = (int)(int16_t)(d.y & 0xFFFF);
    }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const { rowh(rowdesc[r], off, iy0, ix0); }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        const int t = k0 >> 6, c0 = k0 & 63;
        ty = t >> 1;
        tx = t & 1;
        toff = (ty * 3 + tx) * 64 + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int ixp, int ty, int tx) const {      // ixp: nr << 12 | nc << 8 | (ix0 & 0xff)
        const int ix0 = (int)(int8_t)(ixp & 0xFF), nc = (ixp >> 8) & 15, nr = (ixp >> 12) & 15;
        return (unsigned)(iy0 + ty) < (unsigned)nr && (unsigned)(ix0 + tx) < (unsigned)nc;
    }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 % npad < nlive; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int m0) const { return n0 + (m0 / npad) * bnstep; }
    __device__ __forceinline__ int2 ahandle(int m) const { return rowdesc[m]; }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }      // the offset of tap (0,0) may be negative
    __device__ __forceinline__ int bhandle(int m) const { return m; }
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const {
        const int g = z / cpc, c = z - g * cpc;
        mbeg = g * npad + c * mc;
        mend = min(g * npad + nlive, mbeg + mc);
    }
};

// Two row-major [rows][ld] operands side by side along K (k < half from base, k >= half from base + delta), multiplied against
// two weight matrices stacked the same way (Bt's k offset jumps by bdelta at k = half): dX = [dY1 | dY2] . [W1 | W2]^T in one
// pass instead of two accumulating ones (pol1 / v1 both feed dense2's output).
struct DenseRowsPair {
    static constexpr bool kRelu = false;
    const float *base;
    int rows, ld, half;      // K() = 2 * half
    int delta, bdelta;       // float offsets of the second operand / second weight matrix relative to the first
    __device__ __forceinline__ int K() const { return 2 * half; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        off = (long)r * ld;
        iy0 = ix0 = 0;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        toff = k0 < half ? k0 : delta + (k0 - half);
        ty = tx = 0;
    }
    __device__ __forceinline__ bool ok(int, int, int, int) const { return true; }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int) const { return true; }
    __device__ __forceinline__ int bk(int k0, int) const { return k0 < half ? k0 : bdelta + (k0 - half); }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
    // gemm_tn: row handles are fetched one tile ahead of the data (ahandle / bhandle may load), rowh is arithmetic only
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(m, 0); }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }
    __device__ __forceinline__ int bhandle(int m) const { return m; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const { row(h.x, off, iy0, ix0); }
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const { mbeg = z * mc; mend = min(rows, mbeg + mc); }
};

// The compact slot rows in TAP-CLASS order (net_shared.inc, slot_sort): a touched conv2 pixel u reaches conv3 outputs u - t, t in
// 0..2, only inside 0..6, so a slot on the rim of the 9x9 map has 1, 2, 3, 4 or 6 live taps of the 9 (44 % are live on the bench
// workload).  perm[s] = compact row (or -1 behind the live count), tmask[s >> 8] = the taps ANY of the 256 rows of a tile has live.
// Forward (prod = d2s . W3f, N = 9 taps x 64): column tiles of dead taps are neither computed nor stored (n_ok); expand_conv3_patch
// reads live taps only.
struct SlotRowsP {
    static constexpr bool kRelu = false;
    const float *base;
    const int *perm;
    int rows, ld, k;
    const int *rows_dev;
    const unsigned *tmask;
    __device__ __forceinline__ int K() const { return k; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int m = perm[r];
        iy0 = m < 0 ? -1 : 0;
        ix0 = 0;
        off = (long)(m < 0 ? 0 : m) * ld;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        toff = k0;
        ty = tx = 0;
    }
    __device__ __forceinline__ bool ok(int iy0, int, int, int) const { return iy0 >= 0; }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 < *rows_dev; }
    __device__ __forceinline__ bool n_ok(int m0, int n0) const { return !tmask || ((tmask[m0 >> 8] >> (n0 >> 6)) & 1u); }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
};

// conv3's transposed convolution evaluated only at the <= 9 conv2 pixels ("slots") an agent's one-hot can reach: row r is one
// slot with pixel u = ulist[r] of the 9x9 map; k = (ty, tx, co) reads dz3[n][uy-2+ty][ux-2+tx][co] like
// ConvGather<9,9,1,1,-2,-2,3,3,64,7,7,true>.  The source is the patch-compact per-agent dz3 (net_patch.inc):
// base[n][py][px][co] (5x5 window with origin org[n] = oy*3+ox);
// every pixel u - 2 + t inside the 7x7 map lies inside the agent's window by construction.  Rows are the COMPACT slot
// rows (only the touched pixels an agent really has, 6.25 on average instead of 9): rowagent[r] names the sample.
struct SlotGatherT3P {
    static constexpr bool kRelu = false;
    const float *base;
    const int2 *rowdesc;         // per compact slot row: {element offset of tap (0,0) in base, (iy0 << 16) | (ix0 & 0xffff)}; dead rows
                                 // carry iy0 = ix0 = -16 (every tap outside).  Built by slot_rowdesc_kernel (net_shared.inc)
                                 // so that the gather needs ONE index load per row instead of the chain row -> sample -> origin
    int rows;                    // upper bound (9 per sample); the live count is *rows_dev
    const int *rows_dev;
    // Rows come in TAP-CLASS order (slot_sort, net_shared.inc): rowdesc[s] describes compact row perm[s]; tmask[s >> 8] = the taps
    // any row of the 256-row tile has live (tile_ok skips the K-tiles of the others: their gathers are all zeros); zmask[z] = the
    // same union over row range z of gemm_tn (i_ok: no slab tile; the slab reduction knows the same masks).  nullptr: no skipping.
    const int *perm;
    const unsigned *tmask, *zmask;
    __device__ __forceinline__ int K() const { return 576; }
    __device__ __forceinline__ void rowh(int2 d, long &off, int &iy0, int &ix0) const {
        off = d.x;
        iy0 = d.y >> 16;
        ix0 = (int)(int16_t)(d.y & 0xFFFF);
    }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const { rowh(rowdesc[r], off, iy0, ix0); }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        int t = k0 >> 6;
        int c0 = k0 & 63;
        ty = t / 3;
        tx = t - ty * 3;
        toff = (ty * 5 + tx) * 64 + c0;
    }
    __device__ __forceinline__ bool ok(int iy0, int ix0, int ty, int tx) const {
        return (unsigned)(iy0 + ty) < 7u && (unsigned)(ix0 + tx) < 7u;
    }
    // the masks are in the forward's tap numbering (output u - t); this gather's tap (ty, tx) reads output u - 2 + t: tap 8 - k there
    __device__ __forceinline__ unsigned wg_ctx(int m0) const { return tmask ? tmask[m0 >> 8] : 0x1FFu; }
    __device__ __forceinline__ bool tile_ok_c(unsigned c, int k0) const { return (c >> (8 - (k0 >> 6))) & 1u; }
    __device__ __forceinline__ int bk_c(unsigned, int k0) const { return k0; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 < *rows_dev; }
    __device__ __forceinline__ int live_rows() const { return *rows_dev; }      // gemm_rowk, XCD_ORDER = 2
    __device__ __forceinline__ bool i_ok(int z, int i0) const { return !zmask || ((zmask[z] >> (8 - (i0 >> 6))) & 1u); }
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
    // gemm_tn: row handles are fetched one tile ahead of the data (ahandle / bhandle may load), rowh is arithmetic only
    __device__ __forceinline__ int2 ahandle(int m) const { return rowdesc[m]; }
    __device__ __forceinline__ bool hvalid(int2) const { return true; }      // the offset of tap (0,0) may be negative
    __device__ __forceinline__ int bhandle(int m) const { return perm ? perm[m] : m; }      // the B operand's rows are the compact ones
    __device__ __forceinline__ void mrange(int z, int mc, int &mbeg, int &mend) const {
        mbeg = z * mc;
        mend = min(*rows_dev, mbeg + mc);
    }
};


// SlotGatherT3P for gemm_tn with 128-wide I tiles made of TWO LIVE taps of the row range (round 5; PatchRowsPair's idea on conv3's
// slot weight gradient): the B tile (d2s rows) is staged once for two taps, half the workgroups and slab tiles.  I tile index i =
// tap in this gather's numbering = bit 8 - i of zmask[z].
struct SlotGatherT3PPair : SlotGatherT3P {
    static constexpr bool kPairI = true;
    __device__ __forceinline__ int2 pair(int z, int t) const {
        unsigned zm = zmask ? zmask[z] & 0x1FFu : 0x1FFu, rev = 0;
        for (int i = 0; i < 9; ++i) rev |= ((zm >> (8 - i)) & 1u) << i;
        for (int s = 0; s < 2 * t; ++s) rev &= rev - 1u;
        if (!rev) return make_int2(-1, -1);
        const int p0 = __ffs(rev) - 1;
        rev &= rev - 1u;
        return make_int2(p0, rev ? __ffs(rev) - 1 : -1);
    }
};

// conv3's per-agent corrections GATHERED at the patch pixels (round 4; replaces the slot-product tensor and its expansion kernel in
// the forward pass).  Row = one (sample, patch pixel q) item some dense1 GEMM may read (net_patch.inc, wmask); k = (tap, ci):
//     A[row][t * 64 + ci] = (a2_a - a2sh)[u = r + t][ci]   where the agent has touched conv2 pixel u, else 0,
// against conv3's own kernel W3[t][ci][co]: the accumulator IS the agent's correction of conv3's pre-activation at pixel r, and the
// epilogue (EpiPatchExpand) finishes what expand_conv3_patch_kernel did.  The agent's touched pixels are read from the CANONICAL
// 3 x 3 cell block d2c[sample][jy][jx][64] (jy = H - uy: conv2_prep's canonical grid), so that the address of tap t is linear in t:
// offset of tap (0, 0) + toff, toff = -(ty * 3 + tx) * 64, the same for every row of a tile; a cell exists where cy0 - ty lies in
// the agent's valid cell range [lo, hi] (per axis; packed with cy0 + 8 into the row's 16-bit iy0 / ix0).  Rows come sorted by patch
// pixel (all samples' pixel q together, samples in patch-sort order), so a 256-row tile's union of live taps (tmask) is tight and
// tile_ok skips the K-tiles of dead taps: the executed FLOPs are those of the slot-product form, without its 0.45-3.4 GB round trip
// through memory per chunk.
struct SlotsToPatch {
    static constexpr bool kRelu = false;
    const float *base;               // d2c: canonical cell blocks in the row order of the conv2 correction GEMM (EpiConv2Corr)
    const int4 *desc;                // per sorted row: {code (EpiPatchExpand), element offset of tap (0,0) in d2c, live kernel rows << 3 | columns, spare}
    int rows;                        // upper bound (25 per sample); the live count is *rows_dev
    const int *rows_dev;
    const unsigned short *tmask;     // per 128-row tile: union of the rows' live taps (bit t = ty * 3 + tx)
    __device__ __forceinline__ int K() const { return 576; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int4 d = desc[r];
        off = d.y;
        iy0 = d.z >> 3;      // the kernel rows that reach one of the agent's cells from this pixel (3 bits) ...
        ix0 = d.z & 7;       // ... and the columns
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        const int t = k0 >> 6;
        ty = t / 3;
        tx = t - ty * 3;
        toff = -(ty * 3 + tx) * 64 + (k0 & 63);      // tap (ty, tx) reads the cell (ty, tx) steps before the one of tap (0,0)
    }
    __device__ __forceinline__ bool ok(int iy0, int ix0, int ty, int tx) const { return ((iy0 >> ty) & (ix0 >> tx) & 1) != 0; }
    __device__ __forceinline__ unsigned wg_ctx(int m0) const { return tmask ? (unsigned)tmask[m0 >> 7] : 0x1FFu; }      // gemm_rowk, BM = 128
    __device__ __forceinline__ bool tile_ok_c(unsigned c, int k0) const { return (c >> (k0 >> 6)) & 1u; }
    __device__ __forceinline__ int bk_c(unsigned, int k0) const { return k0; }
    __device__ __forceinline__ bool tile_active(int m0) const { return m0 < *rows_dev; }
    __device__ __forceinline__ int live_rows() const { return *rows_dev; }      // gemm_rowk, XCD_ORDER = 2
    __device__ __forceinline__ int bn(int n0, int) const { return n0; }
};

// dense1 on the shared a3 (net_patch.inc): an agent's a3 differs from its env's a3sh only inside a 5x5 window ("patch",
// origin (oy, ox) in {0,1,2}^2 = group g) of the 7x7 map.  Samples are sorted by group into 256-row tiles (perm[slot] =
// sample or -1 for padding, tilegroup[slot >> 8] = g or -1), so that one workgroup multiplies against ONE group's 1600
// rows of the dense1 kernel: patch element (py, px, c) <-> dense1 input ((oy+py)*7 + ox+px)*64 + c.
//   mode 0: the reduction runs over the patch (forward, weight gradient): B's k offset is remapped per 320-float strip
//   mode 1: the output columns are the patch (data gradient): B's row base is remapped per 64-column tile (one pixel)
struct PatchRows {
    static constexpr bool kRelu = false;
    const float *base;
    const int *perm;
    const signed char *tilegroup;
    int rows, ld, k, mode;
    const int *sbeg, *send;      // gemm_tn only: grid z = slice of <= 1024 sorted rows inside one group (empty: sbeg == send)
    // Support masks (net_patch.inc, patch_masks_kernel): inside a group the samples are sorted by the shape of what conv3 can make
    // non-zero in their patch, and tmask[t] says for the 128 sorted rows of tile t which of the 25 patch pixels ANY of them can have
    // non-zero (bit py * 5 + px).  Everything outside is an exact zero of d3, so the work on it is skipped: K-tiles of the forward
    // (tile_ok), pixel columns of the data gradient (n_ok), pixel rows of the weight gradient (i_ok).  nullptr: no skipping.
    const unsigned *tmask;
    const unsigned *zmask;      // the same union per weight-gradient slice (rows [sbeg[z], send[z]))
    __device__ __forceinline__ bool n_ok(int m0, int n0) const {      // gemm_rowk, BM = 256, mode 1: column tile n0 = one patch pixel
        return mode != 1 || !tmask || (((tmask[m0 >> 7] | tmask[(m0 >> 7) + 1]) >> (n0 >> 6)) & 1u);
    }
    __device__ __forceinline__ bool i_ok(int z, int i0) const {      // gemm_tn: slice z, I tile i0 = one patch pixel
        return !zmask || ((zmask[z] >> (i0 >> 6)) & 1u);
    }
    __device__ __forceinline__ int2 ahandle(int m) const { return make_int2(perm[m], 0); }      // gemm_tn: fetched one tile ahead
    __device__ __forceinline__ bool hvalid(int2 h) const { return h.x >= 0; }                     // padding rows of the sorted layout
    __device__ __forceinline__ int bhandle(int m) const { return perm[m]; }
    __device__ __forceinline__ void rowh(int2 h, long &off, int &iy0, int &ix0) const {
        iy0 = 0;
        ix0 = 0;
        off = (long)h.x * ld;
    }
    __device__ __forceinline__ void mrange(int z, int, int &mbeg, int &mend) const {
        mbeg = sbeg[z];
        mend = send[z];
    }
    __device__ __forceinline__ int K() const { return k; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int m = perm[r];
        iy0 = m < 0 ? -1 : 0;
        ix0 = 0;
        off = (long)(m < 0 ? 0 : m) * ld;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        toff = k0;
        ty = tx = 0;
    }
    __device__ __forceinline__ bool ok(int iy0, int, int, int) const { return iy0 >= 0; }
    static constexpr int kMaxK0 = 32 * 64;    // mode 0: a 32-bit support word, one bit per 64-wide pixel (checked where launched)
    struct Ctx { unsigned tm; int b0; };      // the tile's support union; dense1 row of the group's patch pixel (0,0)
    __device__ __forceinline__ Ctx wg_ctx(int m0) const {      // gemm_rowk, BM = 128
        Ctx c{0xFFFFFFFFu, 0};
        if (mode == 0) {
            if (tmask) c.tm = tmask[m0 >> 7];
            const int g = tilegroup[m0 >> 8], oy = g / 3, ox = g - oy * 3;
            c.b0 = (oy * 7 + ox) * 64;
        }
        return c;
    }
    __device__ __forceinline__ bool tile_ok_c(const Ctx &c, int k0) const {      // mode 0: K-tile k0 lies in one patch pixel
        return mode != 0 || ((c.tm >> ((k0 >> 6) & 31)) & 1u);                      // (other modes: K is not the patch, no shift by k0)
    }
    __device__ __forceinline__ int bk_c(const Ctx &c, int k0) const {
        if (mode != 0) return k0;
        const int py = k0 / 320;
        return c.b0 + py * 128 + k0;      // ((oy + py) * 7 + ox) * 64 + (k0 - py * 320)
    }
    __device__ __forceinline__ bool tile_active(int m0) const { return tilegroup[m0 >> 8] >= 0; }
    __device__ __forceinline__ int bn(int n0, int m0) const {
        if (mode != 1) return n0;
        const int g = tilegroup[m0 >> 8], oy = g / 3, ox = g - oy * 3;
        const int j = n0 >> 6, py = j / 5, px = j - py * 5;
        return ((oy + py) * 7 + ox + px) * 64 + (n0 & 63);
    }
};

// PatchRows for gemm_tn with 128-wide I tiles made of TWO LIVE patch pixels (round 5).  The weight gradient's I axis is the patch
// (25 pixels x 64 channels) and a slice's support union zmask[z] says which pixels any of its rows can have non-zero; the 64 x 128
// tiles of round 3 skipped the dead pixels one by one.  Here grid x = t takes the (2t)-th and (2t+1)-th SET bit of zmask[z]: a
// 128 x 128 tile whose two 64-column halves are two arbitrary live pixels (the last tile of an odd count has one) -- the tile shape
// of the dense layers' weight gradients (half the B re-reads and LDS fragment reads per MFMA of the 64-wide form) on exactly the live
// pixels, in both support regimes.  Every output element is the same sum over the same K-tiles of 32 rows: bit-identical slabs.
struct PatchRowsPair : PatchRows {
    static constexpr bool kPairI = true;
    // the tile's two pixels (second: -1 if the slice has an odd number of live pixels and this is its last tile; first: -1 = no tile)
    __device__ __forceinline__ int2 pair(int z, int t) const {
        unsigned zm = zmask ? zmask[z] : 0x1FFFFFFu;
        for (int s = 0; s < 2 * t; ++s) zm &= zm - 1u;
        if (!zm) return make_int2(-1, -1);
        const int p0 = __ffs(zm) - 1;
        zm &= zm - 1u;
        return make_int2(p0, zm ? __ffs(zm) - 1 : -1);
    }
};

// The same idea on gemm_rowk's N axis (dense1's patch data gradient, mode 1: the output columns are the patch): grid x = t takes the
// (2t)-th and (2t+1)-th set bit of the 128-row tile's support union, a 128 x 128 tile on eight waves (wave column wn = pixel half)
// instead of 256 x 64 per pixel.  Same K-tiles in the same order per output element: bit-identical g3p.
struct PatchRowsPairN : PatchRows {
    static constexpr bool kPairN = true;
    __device__ __forceinline__ int2 npair(int m0, int t) const {      // BM = 128
        unsigned zm = tmask ? tmask[m0 >> 7] : 0x1FFFFFFu;
        for (int s = 0; s < 2 * t; ++s) zm &= zm - 1u;
        if (!zm) return make_int2(-1, -1);
        const int p0 = __ffs(zm) - 1;
        zm &= zm - 1u;
        return make_int2(p0, zm ? __ffs(zm) - 1 : -1);
    }
};

// conv2's per-agent corrections as ONE GEMM (net_shared.inc, forward): row = agent (sorted by the parity class of its conv1 window,
// perm[slot] = sample or -1, tilegroup[slot >> 8] = class or -1), K = 4 conv1 candidates x 32 channels of (a1_a - a1_sh), N = 9
// canonical conv2 outputs x 64 channels against the class's own 576-row block of Bt.
struct CorrRows {
    static constexpr bool kRelu = false;
    const float *base;
    const int *perm;
    const signed char *tilegroup;
    int rows;
    __device__ __forceinline__ int K() const { return 128; }
    __device__ __forceinline__ void row(int r, long &off, int &iy0, int &ix0) const {
        const int m = perm[r];
        iy0 = m < 0 ? -1 : 0;
        ix0 = 0;
        off = (long)(m < 0 ? 0 : m) * 128;
    }
    __device__ __forceinline__ void tap(int k0, int &toff, int &ty, int &tx) const {
        toff = k0;
        ty = tx = 0;
    }
    __device__ __forceinline__ bool ok(int iy0, int, int, int) const { return iy0 >= 0; }
    __device__ __forceinline__ bool tile_ok(int, int) const { return true; }
    __device__ __forceinline__ bool tile_active(int m0) const { return tilegroup[m0 >> 8] >= 0; }
    // A column tile is one canonical output j = (ry, rx).  A class whose window starts on an odd conv1 row (py = 1) reaches canonical
    // rows 0..1 only, an even one rows 0..2 (ry = qy + (cy == 1 && py == 0): net_shared.inc), likewise the columns: 4, 6, 6 or 9 of the
    // nine outputs exist for the class, the other blocks of its B are zeros and no agent of the class has a slot there -- those tiles
    // are neither computed nor stored (round 5: 31 % of the launch, in every support regime).
    __device__ __forceinline__ bool n_ok(int m0, int n0) const {
        const int g = tilegroup[m0 >> 8], j = n0 >> 6, ry = j / 3, rx = j - ry * 3;
        return ry < 3 - (g >> 1) && rx < 3 - (g & 1);
    }
    __device__ __forceinline__ int bk(int k0, int) const { return k0; }
    __device__ __forceinline__ int bn(int n0, int m0) const { return n0 + tilegroup[m0 >> 8] * 576; }
};

// ---------------------------------------------------------------------------- epilogues
// An epilogue is evaluated in two passes so that every load it needs is in flight before the first store is issued
// (stores and loads through unrelated pointers cannot be reordered by the compiler: one load -> select -> store chain per
// element serialises 64 memory round trips per lane, which cost the K = 256 data-gradient GEMMs 40 % of their time):
//   row_aux(r)            per output row: index indirections (sorted row -> sample)
//   elem_aux(r, c, ra)    per output element: ReLU mask source, per-env term, bias
//   store(r, c, v, ra, ea)
enum { ACT_NONE = 0, ACT_RELU = 1 };
// kColSum = true: the epilogue also wants the column sums of what it stores (bias gradient of the layer whose dY it
// produces) and defines value(v, ea) and csum


struct EpiBiasAct {
    static constexpr bool kColSum = false, kAddAux = true;   // C[r][c] = act(v + bias[c])
    float *C;
    int ldc;
    const float *bias;
    int act;
    __device__ __forceinline__ int row_aux(int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int, int c, int) const { return bias[c]; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const {
        v += ea;
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        C[(long)r * ldc + c] = v;
    }
};

// EpiBiasAct (ReLU) that also keeps the SIGN of what it stores: one word of 64 column bits per row and 64-column block
// (bits[r * wpr + (c >> 6)], through the kRowBits ballots).  The data gradient of the layer above needs of this tensor only
// relu' = (value > 0): 8 bytes per output row and lane instead of 32 floats (EpiGradSumBits) -- 250 MB less per chunk of the
// gradient step for d2 and v1 (1.5 us more per forward GEMM).  d1 keeps its fp32 mask: its producer's epilogue (EpiPatchFwd, rows
// scattered through the group sort) paid 9 us for the ballots, more than the data gradient gained.
struct EpiBiasActBits {
    static constexpr bool kColSum = false, kAddAux = true;
    static constexpr bool kRowBits = true;
    float *C;
    int ldc;
    const float *bias;
    unsigned long long *bits;
    int wpr;
    __device__ __forceinline__ int row_aux_n(int, int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int, int c, int) const { return bias[c]; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const { C[(long)r * ldc + c] = fmaxf(v + ea, 0.f); }
    __device__ __forceinline__ bool bit(float v, float ea) const { return v + ea > 0.f; }
    __device__ __forceinline__ void store_bits(int r, int colbase, unsigned long long w, int) const { bits[(long)r * wpr + (colbase >> 6)] = w; }
};

// Two dense layers that read the SAME input as one launch (pol1 and v1 both read dense2's output d2, policy_v_network.py:40-52):
// columns [0, half) are the first layer's (C0, bias0), columns [half, 2 half) the second's (C1, bias1, and the sign bits of what is
// stored, as EpiBiasActBits keeps them).  The A tile is staged once for both and one launch is saved per forward pass.
struct EpiBiasActSplit {
    static constexpr bool kColSum = false, kAddAux = true;
    static constexpr bool kRowBits = true;
    float *C0, *C1;
    int ldc, half;
    const float *bias0, *bias1;
    unsigned long long *bits;
    int wpr;
    __device__ __forceinline__ int row_aux_n(int, int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int, int c, int) const { return c < half ? bias0[c] : bias1[c - half]; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const {
        if (c < half) C0[(long)r * ldc + c] = fmaxf(v + ea, 0.f);
        else C1[(long)r * ldc + c - half] = fmaxf(v + ea, 0.f);
    }
    __device__ __forceinline__ bool bit(float v, float ea) const { return v + ea > 0.f; }
    __device__ __forceinline__ void store_bits(int r, int colbase, unsigned long long w, int) const {
        if (colbase >= half) bits[(long)r * wpr + ((colbase - half) >> 6)] = w;      // wave-uniform
    }
};

struct EpiBiasDual {
    static constexpr bool kColSum = false, kAddAux = true;   // C[r][c] = v + bias[c] and C2[r][c] = relu(v + bias[c]): pre-activation and activation in one pass
    float *C, *C2;
    int ldc;
    const float *bias;
    __device__ __forceinline__ int row_aux(int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int, int c, int) const { return bias[c]; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const {
        v += ea;
        C[(long)r * ldc + c] = v;
        C2[(long)r * ldc + c] = fmaxf(v, 0.f);
    }
};

// EpiBiasDual over a row list (GatherConv2ReluRows): row r -> output row rowlist[r]; the background row (-1) goes to bgz / bga
struct EpiBiasDualRows {
    static constexpr bool kColSum = false, kAddAux = true;
    float *C, *C2;
    int ldc;
    const float *bias;
    const int *rowlist, *rows_dev;
    float *bgz, *bga;
    __device__ __forceinline__ int row_aux(int r) const { return r < *rows_dev ? rowlist[r] : -2; }
    __device__ __forceinline__ float elem_aux(int, int c, int) const { return bias[c]; }
    __device__ __forceinline__ void store(int, int c, float v, int q, float ea) const {
        v += ea;
        if (q >= 0) {
            C[(long)q * ldc + c] = v;
            if (C2) C2[(long)q * ldc + c] = fmaxf(v, 0.f);      // conv3 stores the pre-activation only
        } else if (q == -1 && bgz) {
            bgz[c] = v;
            bga[c] = fmaxf(v, 0.f);
        }
    }
};

// conv2 corrections (CorrRows): column block j = c >> 6 is the agent's canonical conv2 output j; cslot[m*9+j] = its compact slot
// row (or -1).  On entry d2[slot] holds the env's shared PRE-activation of that pixel (conv2_prep_kernel copies it there); on exit
// a2_a - a2sh = relu(z + v) - relu(z), plus one word of 64 channel sign bits of relu(z + v) per slot row.  The slot index is
// resolved once per output row (row_aux_n: the workgroup's 64 columns are one canonical output), not per element.
struct EpiConv2Corr {
    static constexpr bool kColSum = false, kAddAux = true;
    static constexpr bool kRowBits = true;
    float *d2;
    unsigned long long *m2;
    const int *perm, *cslot;
    // second copy for conv3's gather form (SlotsToPatch): the output tile as it is, d2c[sorted row][canonical cell][64] -- cells the
    // agent does not have are not written (and never gathered).  nullptr: the slot-product form of conv3's corrections
    float *d2c;
    __device__ __forceinline__ int row_aux_n(int r, int colbase) const {
        const int m = perm[r];
        return m < 0 ? -1 : cslot[m * 9 + (colbase >> 6)];
    }
    __device__ __forceinline__ float elem_aux(int, int c, int slot) const { return slot >= 0 ? d2[(long)slot * 64 + (c & 63)] : 0.f; }
    __device__ __forceinline__ void store(int r, int c, float v, int slot, float z) const {
        if (slot < 0) return;
        const float d = fmaxf(v + z, 0.f) - fmaxf(z, 0.f);
        d2[(long)slot * 64 + (c & 63)] = d;
        if (d2c) d2c[(long)r * 576 + c] = d;
    }
    __device__ __forceinline__ bool bit(float v, float z) const { return v + z > 0.f; }
    __device__ __forceinline__ void store_bits(int, int, unsigned long long w, int slot) const {
        if (slot >= 0) m2[slot] = w;
    }
};

// SlotsToPatch's epilogue: v = the agent's correction of conv3's pre-activation at the row's patch pixel; z = the env's shared
// pre-activation there (z3sh, or the background row where the trunk's list form did not compute it: z offset < 0).  Writes
// d3 = relu(z + v) - relu(z) and, for pixels of the agent's own support, the 64 sign bits of relu(z + v) (m3) -- what
// expand_conv3_patch_kernel wrote.  Rows outside the own support have no live tap: v = 0, d3 = 0.
struct EpiPatchExpand {
    static constexpr bool kColSum = false, kAddAux = true;
    static constexpr bool kRowBits = true;
    // Everything the store pass needs of a row rides in ONE register (its descriptor's code word, fetched in the load pass): a load
    // behind the first store would wait for it (d3 may alias anything as far as the compiler knows), once per element.
    //   code = inside << 30 | affected << 29 | map pixel p << 23 | sample m << 5 | patch pixel q; -1: a dead row
    const int4 *desc;
    const float *z3sh, *z3bg;
    float *d3;
    unsigned long long *m3;
    __device__ __forceinline__ int row_aux_n(int r, int) const { return desc[r].x; }
    __device__ __forceinline__ float elem_aux(int, int c, int code) const {
        if (code < 0) return 0.f;
        const int m = (code >> 5) & 0x1FFFF, p = (code >> 23) & 63;
        return ((code >> 29) & 1) ? z3sh[(long)(m / 10) * 3136 + p * 64 + c] : z3bg[c];
    }
    __device__ __forceinline__ void store(int, int c, float v, int code, float z) const {
        if (code < 0) return;
        d3[(long)((code >> 5) & 0x1FFFF) * 1600 + (code & 31) * 64 + c] = fmaxf(v + z, 0.f) - fmaxf(z, 0.f);
    }
    __device__ __forceinline__ bool bit(float v, float z) const { return v + z > 0.f; }
    __device__ __forceinline__ void store_bits(int, int, unsigned long long w, int code) const {
        if (code >= 0 && ((code >> 30) & 1)) m3[(long)((code >> 5) & 0x1FFFF) * 25 + (code & 31)] = w;
    }
};

struct EpiPatchFwd {
    static constexpr bool kColSum = false, kAddAux = true;   // sorted row -> sample m: out[m][c] = relu(v + ysh[env(m)][c]) (ysh = per-env part incl. bias)
    float *out;
    const float *ysh;
    const int *perm;
    int ld;
    __device__ __forceinline__ int row_aux(int r) const { return perm[r]; }
    __device__ __forceinline__ float elem_aux(int, int c, int m) const { return ysh[(long)(max(m, 0) / 10) * ld + c]; }     // padding rows (m < 0) read env 0, unused
    __device__ __forceinline__ void store(int, int c, float v, int m, float ea) const {
        if (m >= 0) out[(long)m * ld + c] = fmaxf(v + ea, 0.f);
    }
};

struct EpiPermStore {
    static constexpr bool kColSum = false;   // sorted row -> sample m: out[m][c] = v
    float *out;
    const int *perm;
    int ld;
    __device__ __forceinline__ int row_aux(int r) const { return perm[r]; }
    __device__ __forceinline__ float elem_aux(int, int, int) const { return 0.f; }
    __device__ __forceinline__ void store(int, int c, float v, int m, float) const {
        if (m >= 0) out[(long)m * ld + c] = v;
    }
};

struct EpiGrad {
    static constexpr bool kColSum = false;   // dX[r][c] = v * (fwd[r][c] > 0): data gradient with the ReLU mask of the forward tensor fused
    float *dX;
    int ld;
    const float *fwd;   // forward activation of the same tensor (post-ReLU)
    __device__ __forceinline__ int row_aux(int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int r, int c, int) const { return fwd[(long)r * ld + c]; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const { dX[(long)r * ld + c] = ea > 0.f ? v : 0.f; }
};

// EpiGrad that also writes, per 64-row wave tile, the column sums of the masked gradient it stores: csum[wave tile][N].
// Summed over the wave tiles (slab_reduce_narrow_kernel) this is the bias gradient of the layer below -- dY never has
// to be read again for it.
struct EpiGradSum {
    static constexpr bool kColSum = true;
    float *dX;
    int ld;
    const float *fwd;
    float *csum;
    __device__ __forceinline__ int row_aux(int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int r, int c, int) const { return fwd[(long)r * ld + c]; }
    __device__ __forceinline__ float value(float v, float ea) const { return ea > 0.f ? v : 0.f; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const { dX[(long)r * ld + c] = value(v, ea); }
};

// EpiGradSum with the forward tensor's ReLU mask read from its sign bits (EpiBiasActBits): same values.  One 8-byte load per output
// row of the lane: a wave covers one 64-column block, the lane its columns l16 + 16 b -- their four bits ride in the row's aux word
// (a load per ELEMENT, as the fp32 mask needs, made this epilogue 50 % slower than reading the tensor)
struct EpiGradSumBits {
    static constexpr bool kColSum = true;
    static constexpr bool kRowAuxN = true;
    float *dX;
    int ld;
    const unsigned long long *bits;     // [rows][ld / 64]
    float *csum;
    __device__ __forceinline__ int row_aux_n(int r, int colbase) const {
        const unsigned long long w = bits[(long)r * (ld >> 6) + (colbase >> 6)] >> (threadIdx.x & 15);
        return (int)((w & 1ull) | ((w >> 15) & 2ull) | ((w >> 30) & 4ull) | ((w >> 45) & 8ull));
    }
    __device__ __forceinline__ float elem_aux(int, int c, int ra) const { return ((ra >> ((c >> 4) & 3)) & 1) ? 1.f : 0.f; }
    __device__ __forceinline__ float value(float v, float ea) const { return ea > 0.f ? v : 0.f; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float ea) const { dX[(long)r * ld + c] = value(v, ea); }
};

struct EpiStore {
    static constexpr bool kColSum = false;   // dX[r][c] = v
    float *dX;
    int ld;
    __device__ __forceinline__ int row_aux(int) const { return 0; }
    __device__ __forceinline__ float elem_aux(int, int, int) const { return 0.f; }
    __device__ __forceinline__ void store(int r, int c, float v, int, float) const { dX[(long)r * ld + c] = v; }
};

struct EpiGradPM {
    static constexpr bool kColSum = false;   // pixel-major rows (r = q*nsamp + n) -> dX[n][q][c], ReLU mask of the forward tensor fused
    float *dX;
    const float *fwd;
    int nsamp, pps, ld;
    __device__ __forceinline__ int row_aux(int r) const {      // element offset of the row (< 2^31: 40 960 x 81 x 64)
        int q = r / nsamp, n = r - q * nsamp;
        return (n * pps + q) * ld;
    }
    __device__ __forceinline__ float elem_aux(int, int c, int base) const { return fwd[(long)base + c]; }
    __device__ __forceinline__ void store(int, int c, float v, int base, float ea) const { dX[(long)base + c] = ea > 0.f ? v : 0.f; }
};

// data gradient of conv2: the four pixel-parity classes (py,px) read the SAME 2x2 source taps and differ
// only in their kernel, so they are the N = 4*32 columns of one GEMM: row (n, yh, xh), column cls*32 + c
// -> pixel (2yh+py, 2xh+px), channel c of the [n][20][20][32] tensor
// MASK: the ReLU mask of the forward tensor `fwd` (same layout; a pre-activation serves as well: relu'(s) = s > 0) is fused.
// SUM: the epilogue also writes the per-wave-tile column sums of what it stores (csum[wave tile][128]; folded over the four
// classes they are conv1's bias gradient).
template <bool MASK, bool SUM = false>
struct EpiGradStride2 {
    static constexpr bool kColSum = SUM;
    float *dX;
    const float *fwd;     // MASK = false: raw transposed convolution, fwd unused
    float *csum;
    __device__ __forceinline__ int row_aux(int r) const {      // offset of pixel (2yh, 2xh) (< 2^31: 40 960 x 12 800)
        int n = r / 100, q = r - n * 100;
        int yh = q / 10, xh = q - yh * 10;
        return ((n * 20 + 2 * yh) * 20 + 2 * xh) * 32;
    }
    __device__ __forceinline__ int col_off(int col) const {
        int cls = col >> 5, c = col & 31;
        return ((cls >> 1) * 20 + (cls & 1)) * 32 + c;
    }
    __device__ __forceinline__ float elem_aux(int, int c, int base) const { return MASK ? fwd[(long)base + col_off(c)] : 0.f; }
    __device__ __forceinline__ float value(float v, float ea) const { return (!MASK || ea > 0.f) ? v : 0.f; }
    __device__ __forceinline__ void store(int, int c, float v, int base, float ea) const { dX[(long)base + col_off(c)] = value(v, ea); }
};

// EpiGrad over a row list (GatherT3Rows): dX[q][c] = v * (fwd[q][c] > 0) at the listed rows q; the column sums are those of the
// UNMASKED v over the listed rows (per wave tile; trunk_closed3_kernel turns them into the unlisted rows' sum)
struct EpiGradRowsSum {
    static constexpr bool kColSum = true;
    float *dX;
    int ld;
    const float *fwd;
    float *csum;
    const int *rowlist, *rows_dev;
    __device__ __forceinline__ int row_aux(int r) const { return r < *rows_dev ? rowlist[r] : -1; }
    __device__ __forceinline__ float elem_aux(int, int c, int q) const { return fwd[(long)max(q, 0) * ld + c]; }
    __device__ __forceinline__ float value(float v, float) const { return v; }
    __device__ __forceinline__ void store(int, int c, float v, int q, float ea) const {
        if (q >= 0) dX[(long)q * ld + c] = ea > 0.f ? v : 0.f;
    }
};

// EpiGradStride2<true, true> over a block list (GatherT2Rows).  The column sums are taken RELATIVE to the background mask
// m_bg[c] = (b1[c] > 0): sum over the listed pixels of (m - m_bg) . v -- what the untouched pixels (pre-activation b1, mask m_bg)
// contribute to conv1's bias gradient is m_bg . (sum over ALL pixels of v), which has a closed form (trunk_closed_kernel).
// elem_aux codes the two masks as 2 m + m_bg.
struct EpiGradStride2Rows {
    static constexpr bool kColSum = true;
    float *dX;
    const float *fwd;
    float *csum;
    const float *b1;
    const int *blklist, *rows_dev;
    __device__ __forceinline__ int row_aux(int r) const {
        const int q = r < *rows_dev ? blklist[r] : -1;
        if (q < 0) return -1;
        const int n = q / 100, b = q - n * 100, yh = b / 10, xh = b - yh * 10;
        return ((n * 20 + 2 * yh) * 20 + 2 * xh) * 32;
    }
    __device__ __forceinline__ int col_off(int col) const {
        const int cls = col >> 5, c = col & 31;
        return ((cls >> 1) * 20 + (cls & 1)) * 32 + c;
    }
    __device__ __forceinline__ float elem_aux(int, int c, int base) const {
        const float s = fwd[(long)max(base, 0) + col_off(c)];
        return (s > 0.f ? 2.f : 0.f) + (b1[c & 31] > 0.f ? 1.f : 0.f);
    }
    __device__ __forceinline__ float value(float v, float ea) const { return ea == 2.f ? v : ea == 1.f ? -v : 0.f; }
    __device__ __forceinline__ void store(int, int c, float v, int base, float ea) const {
        if (base >= 0) dX[(long)base + col_off(c)] = ea >= 2.f ? v : 0.f;
    }
};

// c1b[c] += sum over wave tiles and the four classes of csum[tile][cls * 32 + c].  Two stages, both in a fixed order (bitwise
// reproducible): kFoldParts workgroups each fold a contiguous range of wave tiles (8 strided partial sums per channel, combined in
// order), then one wave adds the kFoldParts partial rows.  (A single workgroup walking all 6 400 tiles of a 4 096-env chunk took
// 268 us per chunk -- 4.7 % of the update's kernel time.)
constexpr int kFoldParts = 64;
// live_rows != nullptr: the GEMM ran over a row list and wrote the tiles of its live rows only (two wave tiles per 128 rows)
__global__ __launch_bounds__(256) void fold_class_sums_kernel(const float *__restrict__ csum, int parts, float *__restrict__ partial,
                                                              const int *__restrict__ live_rows = nullptr) {
    __shared__ float red[256];
    const int c = threadIdx.x & 31, k = threadIdx.x >> 5;      // 8 partial sums per channel
    if (live_rows) parts = min(parts, (*live_rows + 127) / 128 * 2);
    const int per = (parts + kFoldParts - 1) / kFoldParts, p0 = blockIdx.x * per, p1 = min(parts, p0 + per);
    float s = 0.f;
    for (int p = p0 + k; p < p1; p += 8) {
        const float *row = csum + (long)p * 128 + c;
        s += (row[0] + row[32]) + (row[64] + row[96]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j * 32 + threadIdx.x];
        partial[blockIdx.x * 32 + threadIdx.x] = t;
    }
}
// (fold_class_final_kernel: its body is a job of reduce_batch_kernel since round 4, net_reduce.inc)

// ---------------------------------------------------------------------------- fp32 operands on the fp16 matrix pipe
// (x, y) -> packed fp16 pairs h = fp16(.), l = fp16((. - h) * 2^11); 3 VALU instructions per element (v_cvt_pk_f16_f32,
// v_cvt_f32_f16, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32).  Element 0 of a pair is its low half.
constexpr float kLowScale = 2048.f;
__device__ __forceinline__ void split2(float x, float y, unsigned &h, unsigned &l) {
    const f32x2 xy = {x, y};
    const f16x2 hv = __builtin_convertvector(xy, f16x2);
    const f32x2 r = {(x - (float)hv[0]) * kLowScale, (y - (float)hv[1]) * kLowScale};
    const f16x2 lv = __builtin_convertvector(r, f16x2);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
// four consecutive-k values -> 4 fp16 per plane
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, uint2 &h, uint2 &l) {
    split2(x0, x1, h.x, l.x);
    split2(x2, x3, h.y, l.y);
}
// acc += h_a h_b, acl += l'_a h_b + h_a l'_b (acl carries the factor 2^11) on v_mfma_f32_16x16x32_f16: per FLOP the
// 16x16x32 form draws less power than 32x32x16, and on toggling operands the matrix pipe is power-limited
__device__ __forceinline__ void mfma_x3(f32x4 &acc, f32x4 &acl, const f16x8 (&a)[2], const f16x8 (&b)[2]) {
    acl = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acl, 0, 0, 0);
    acl = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acl, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
}
// LDS tiles: [row][32 fp16] = 64-byte rows, no padding.  The 16-byte chunk c (8 consecutive k) of row r is stored at chunk
// position c ^ swz(r), swz = {0,2,3,1}[(r >> 2) & 3].  The MFMA operand map (lane l: row l & 15, k = 8 (l >> 4) .. +7) then
// reads one ds_read_b128 per plane with the 16 lanes of every service group on 16 distinct 4-bank slots, and the staging
// stores of gemm_rowk (8 lanes per row, 8 bytes each) cover whole rows: both conflict free (checked by enumeration).
constexpr int kLdh = 32;
// Range guard of the fp16 operand form: every gemm_rowk output is a later GEMM's operand, so a tile whose accumulators leave the
// fp16 range (|v| > 65 504: the next split would turn it into inf, and ReLU's max would swallow the NaN that follows) raises this
// sticky flag; the host reads it at its next synchronisation point and fails the call with GRL_E_RANGE (net_conv.hip: range_check).
__device__ int *g_gemm_range_flag;      // -> one hipMalloc'd word per process, set when the first net is created
// The way back from the fp32 form (round 5): while a net computes on it, every gemm_rowk tile also records the largest |value| it
// hands on (float bits, compared as unsigned: one atomicMax per wave).  The host reads it at the end of an update; after a few
// updates that stayed well inside the fp16 range the net returns to the fast form (net_train.inc, train_apply).
__device__ unsigned *g_gemm_absmax;     // -> the word behind the flag
constexpr float kF16Max = 65504.f;
// wave layouts of the two large tile shapes (WGM x WGN waves; 4 x 2 and 8 x 1: eight waves with 32 x 64 wave tiles)
constexpr int kW128M = 4, kW128N = 2, kW256M = 8, kW256N = 1;
__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// ---- the fp32 form (template flag F32 of gemm_rowk / gemm_tn; net_conv.hip switches a net to it after a GRL_E_RANGE pass).
// The SAME LDS bytes hold the raw fp32 tile instead of the two fp16 planes: [row][32 floats], the eight 16-byte chunks of a row
// XOR-swizzled by row & 7 (staging stores cover whole rows; a fragment read -- lane l: row l & 15, k = 4 j + (l >> 4) -- puts the
// 64 lanes on 32 banks twice, the minimum for a 4-byte read).  One K = 32 step is eight v_mfma_f32_16x16x4_f32 per 16 x 16 tile
// into ONE accumulator set: no operand range beyond fp32's, 5.3x the matrix-pipe time of the three-product form (measured: 103 TFLOP/s over the update's GEMMs, 0.65 of the 157 TFLOP/s fp32-MFMA peak).
__device__ __forceinline__ int f32_off(int row, int k) { return row * 32 + ((((k >> 2) ^ (row & 7)) << 2) | (k & 3)); }
// The reduction order inside a K = 32 tile is free as long as both operands use the same one: lane (row l & 15, group kg = l >> 4)
// takes k = 8 kg .. 8 kg + 7 -- two 16-byte chunks of its row, i.e. two ds_read_b128 per fragment -- and MFMA j multiplies the j-th of
// them (round 4; until then k = 4 j + kg: eight ds_read_b32 per fragment, and the loop was bound by LDS issue at 0.65 of the fp32
// matrix-pipe peak).  The chunk swizzle keeps the reads conflict-free: a quarter wave's 16 rows hit chunk (2 kg [+ 1]) ^ (row & 7).
template <int TM, int TN>
__device__ __forceinline__ void mfma_f32_step(const float *__restrict__ Af, const float *__restrict__ Bf, int arow, int brow, int kg,
                                              f32x4 (&acc)[TM][TN]) {
    float4 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        a0[a] = *reinterpret_cast<const float4 *>(Af + f32_off(arow + a * 16, 8 * kg));
        a1[a] = *reinterpret_cast<const float4 *>(Af + f32_off(arow + a * 16, 8 * kg + 4));
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        b0[b] = *reinterpret_cast<const float4 *>(Bf + f32_off(brow + b * 16, 8 * kg));
        b1[b] = *reinterpret_cast<const float4 *>(Bf + f32_off(brow + b * 16, 8 * kg + 4));
    }
#define GRL_F32_STEP(AV, BV)                                                                                          \
    _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                    \
        _Pragma("unroll") for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, BV, acc[a][b], 0, 0, 0);
    GRL_F32_STEP(a0[a].x, b0[b].x) GRL_F32_STEP(a0[a].y, b0[b].y) GRL_F32_STEP(a0[a].z, b0[b].z) GRL_F32_STEP(a0[a].w, b0[b].w)
    GRL_F32_STEP(a1[a].x, b1[b].x) GRL_F32_STEP(a1[a].y, b1[b].y) GRL_F32_STEP(a1[a].z, b1[b].z) GRL_F32_STEP(a1[a].w, b1[b].w)
#undef GRL_F32_STEP
}

// optional hooks of a gather descriptor: whole output column tiles (gemm_rowk) / output row tiles (gemm_tn) known to be zero
// Per-workgroup context of a gather (wg_ctx): what tile_ok / bk need of the workgroup's tile, loaded ONCE before the K loop.  A
// load inside the loop is re-issued every iteration (the barriers are fences), and a byte or halfword one is a VECTOR load whose
// s_waitcnt vmcnt(0) also waits for the tile prefetch issued just before it: the loop then runs at one memory latency per K-tile
// (measured round 4: dense1's patch forward and the conv3 patch gather).
template <class T, class = void> struct ag_has_ctx : std::false_type {};
template <class T> struct ag_has_ctx<T, std::void_t<decltype(std::declval<const T &>().wg_ctx(0))>> : std::true_type {};
template <class AG> __device__ __forceinline__ auto ag_ctx(const AG &ag, int m0) {
    if constexpr (ag_has_ctx<AG>::value) return ag.wg_ctx(m0);
    else return m0;
}
template <class AG, class C> __device__ __forceinline__ bool ag_tile_ok(const AG &ag, const C &c, int m0, int k0) {
    if constexpr (ag_has_ctx<AG>::value) return ag.tile_ok_c(c, k0);
    else return ag.tile_ok(m0, k0);
}
template <class AG, class C> __device__ __forceinline__ int ag_bk(const AG &ag, const C &c, int m0, int k0) {
    if constexpr (ag_has_ctx<AG>::value) return ag.bk_c(c, k0);
    else return ag.bk(k0, m0);
}
template <class T, class = void> struct ag_has_n_ok : std::false_type {};
template <class T> struct ag_has_n_ok<T, std::void_t<decltype(std::declval<const T &>().n_ok(0, 0))>> : std::true_type {};
template <class T, class = void> struct ag_has_i_ok : std::false_type {};
template <class T> struct ag_has_i_ok<T, std::void_t<decltype(std::declval<const T &>().i_ok(0, 0))>> : std::true_type {};
template <class T, class = void> struct ag_pair_n : std::false_type {};
template <class T> struct ag_pair_n<T, std::void_t<decltype(T::kPairN)>> : std::bool_constant<T::kPairN> {};
template <class T, class = void> struct ag_pair_i : std::false_type {};
template <class T> struct ag_pair_i<T, std::void_t<decltype(T::kPairI)>> : std::bool_constant<T::kPairI> {};

// ---------------------------------------------------------------------------- C = A(rowk) * Bt^T
// Four or eight waves per workgroup (WGM x WGN).  The eight-wave forms keep a tile's bytes per FLOP and halve the wave tile
// (32 x 64: 64 accumulator registers instead of 128), so four waves fit a SIMD instead of two: with three MFMAs per product the
// loop waits on its loads and barriers more than on the matrix pipe, and more resident waves cover those waits (+4-18 % on the
// dense1 shapes, most on short K: tools/ubench/gemm_f16x3.hip).  The second launch bound is waves per SIMD (HIP-Clang): two
// workgroups per CU either way, i.e. a 256- or 128-register budget.  (Round 4: the 128 x 64 four-wave instances sit at 134-150 registers,
// three waves per SIMD; capped at 128 they spill 8-92 bytes per lane and the conv3 patch gather went 125 -> 193 us, conv2's class
// corrections 113 -> 151, the conv3 row-list data gradient 62 -> 80 -- not adopted.)
template <int BM, int BN, int WGM, int WGN, class AG, class Epi, int XCD_ORDER = 1, bool FENCE = true, bool F32 = false, int NBUF = 1, bool ACC1 = false>
__global__ __launch_bounds__(64 * WGM * WGN, ACC1 ? (WGM * WGN == 8 ? 6 : 4) : WGM * WGN / 2) void gemm_rowk(AG ag, const float *__restrict__ Bt, int ldb, int N, Epi epi) {
    constexpr int BK = 32, LDH = kLdh, NT = 64 * WGM * WGN, RPP = NT / 8;          // RPP: tile rows staged per pass (8 threads per row)
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;       // 16x16 MFMA tiles per wave
    constexpr int NA = BM / RPP;                      // float4 per thread for the A tile
    constexpr int NB = BN / RPP;                      // float4 per thread for the B tile
    static_assert((WGM * WGN == 4 || WGM * WGN == 8) && TM >= 1 && TN >= 1 && NA >= 1 && BN % RPP == 0, "4 or 8 waves");
    // NBUF = 2: tile k+1 is staged into the other LDS buffer while tile k is multiplied, one barrier per K-tile instead of two (staging
    // and the fragment reads of one iteration touch different buffers).  Per instance, measured in the product (round 4, 81 920-sample
    // chunk): the paired small-dense data gradient -11 %, dense1's per-env forward under the union mask -7 %, dense1's patch forward
    // -4 %; the 256 x 64 patch data gradient +9 %, the conv3 trunk list forward +7 %, every other instance within +-2.5 % -- so three
    // instances use it (tools/ubench/gemm_f16x3.hip had found -2.5 ... +5 % on the dense shapes).
    static_assert(NBUF == 1 || NBUF == 2, "one or two LDS buffers");
    __shared__ __attribute__((aligned(16))) unsigned short As[NBUF][2][BM * LDH];      // planes h, l'
    // ACC1: ONE accumulator set instead of two.  B is the weight operand of every gemm_rowk launch: a third plane holds h_b * 2^11 (exact
    // in fp16 while |b| < 32; beyond that it is inf, the tile's outputs are inf / NaN, the range guard fires and the net moves to the
    // fp32 form), so that h_a h_b 2^11 + l'_a h_b + h_a l'_b is one sum carrying 2^11: half the accumulator registers, a third
    // workgroup per CU.
    static_assert(!(ACC1 && F32), "the fp32 form has one accumulator set anyway");
    __shared__ __attribute__((aligned(16))) unsigned short Bs[NBUF][ACC1 ? 3 : 2][BN * LDH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    // Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest), each with its own L2: left alone, the
    // N tiles that share one A tile land on different XCDs and every one of them fetches it again.  Inside each run of
    // 8 M-tiles the order is therefore transposed: XCD x takes M-tile 8g + x with ALL its N tiles (inactive trailing
    // M tiles stay spread over the XCDs).  A last partial run keeps the launch order.  XCD_ORDER = false keeps the launch
    // order everywhere.  (Also measured for the wide-N dense1 patch data gradient: runs of four M tiles per XCD walked N tile by N
    // tile, so that a group's weight slice is fetched once per run -- half the bytes again, no faster: 40.9-41.6 against 38.5 ms.)
    // XCD_ORDER = 2 (one N tile, a live row count on the device: the sorted gathers): XCD x takes the CONTIGUOUS range of M tiles
    // [x tpx, (x + 1) tpx), tpx = ceil(live tiles / 8), in order -- rows that are neighbours in the sorted order re-read the same
    // operand lines (SlotsToPatch: a super-block's cells), which then stay in ONE 4 MB L2 instead of being fetched into all eight.
    int bx = blockIdx.x, by = blockIdx.y;
    if constexpr (XCD_ORDER == 2) {
        const int nt = (ag.live_rows() + BM - 1) / BM, tpx = (nt + 7) >> 3, w = blockIdx.y;      // gridDim.x == 1, gridDim.y >= nt + 7
        if ((w >> 3) >= tpx) return;
        by = (w & 7) * tpx + (w >> 3);
        if (by >= nt) return;
    }
    if (XCD_ORDER == 1) {
        const int nx = gridDim.x, L = by * nx + bx, g = L / (8 * nx);
        if ((g + 1) * 8 <= (int)gridDim.y) {
            const int l = L - g * 8 * nx;
            by = g * 8 + (l & 7);
            bx = l >> 3;
        }
    }
    const int m0 = by * BM, n0 = bx * BN;
    if (!ag.tile_active(m0)) return;     // block-uniform (padding tiles of the group-sorted layouts)
    constexpr bool PAIRN = ag_pair_n<AG>::value;      // the N tile is two independent 64-column runs (PatchRowsPairN)
    int2 pp = make_int2(0, 0);
    if constexpr (PAIRN) {
        static_assert(!PAIRN || (BM == 128 && BN == 128 && WGN == 2 && 64 * WGM * WGN == 512), "128 x 128 on eight waves: a wave column per pixel");
        pp = ag.npair(m0, bx);
        if (pp.x < 0) return;                    // fewer live pixels than this tile's first
    } else if constexpr (ag_has_n_ok<AG>::value) {      // a column tile that is zero for every row of the tile is neither computed nor stored:
        if (!ag.n_ok(m0, n0)) return;            // its consumer knows the same masks (agent_dz3_kernel)
    }
    const int M = ag.rows, K = ag.K();
    const auto wctx = ag_ctx(ag, m0);
    const int trow = tid >> 3, tk4 = (tid & 7) * 4;
    const int wo = (((tk4 >> 3) ^ swz(trow)) << 3) + (tk4 & 4);      // swizzled k offset of this thread's stores (rows trow + RPP i: same swizzle)

    // A rows owned by this thread: r = trow + RPP*i
    long aoff[NA];
    int ayx[NA];
    bool arow_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int r = m0 + trow + RPP * i;
        arow_ok[i] = r < M;
        int iy0, ix0;
        ag.row(arow_ok[i] ? r : 0, aoff[i], iy0, ix0);
        ayx[i] = (iy0 << 16) | (ix0 & 0xFFFF);
    }
    static_assert(NB == 1 || NB == 2 || NB == 4, "B tile of 1, 2 or 4 passes");
    const float *brow0 = Bt + (long)(ag.bn(PAIRN ? pp.x * 64 : n0, m0) + trow) * ldb + tk4;
    // second pass of B rows (RPP further down; PAIRN: the second pixel's rows -- the first one's again where the tile has only one)
    const float *brow1 = PAIRN ? Bt + (long)(ag.bn(max(pp.y, pp.x) * 64, m0) + trow) * ldb + tk4 : brow0 + (long)RPP * ldb;

    float4 ra[NA];
    float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0, rb2 = rb0, rb3 = rb0;
    unsigned vmask = 0;
#define GRL_LOAD_TILE(kt_)                                                                                 \
    {                                                                                                      \
        int toff, ty, tx;                                                                                  \
        ag.tap((kt_) * BK, toff, ty, tx);                                                                  \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                   \
            int iy0 = ayx[i] >> 16, ix0 = (int)(int16_t)(ayx[i] & 0xFFFF);                                 \
            bool v = arow_ok[i] && ag.ok(iy0, ix0, ty, tx);                                                \
            /* unconditional load from a clamped address + select: no branch, so the compiler keeps all */ \
            /* loads of the tile together at the top of the compute phase (prefetch distance = 1 tile) */  \
            ra[i] = *reinterpret_cast<const float4 *>(ag.base + (v ? aoff[i] + toff : 0L) + tk4);          \
            vmask = v ? (vmask | (1u << i)) : (vmask & ~(1u << i));   /* the zero-fill select happens at store time */ \
        }                                                                                                  \
        /* named scalars, not an array: an array here is "promoted" to LDS by the compiler */             \
        const int bko = ag_bk(ag, wctx, m0, (kt_) * BK);                                                             \
        rb0 = *reinterpret_cast<const float4 *>(brow0 + bko);                                              \
        if (NB > 1) rb1 = *reinterpret_cast<const float4 *>(brow1 + bko);                                  \
        if (NB > 2) rb2 = *reinterpret_cast<const float4 *>(brow0 + (long)2 * RPP * ldb + bko);            \
        if (NB > 2) rb3 = *reinterpret_cast<const float4 *>(brow0 + (long)3 * RPP * ldb + bko);            \
    }
#define GRL_STORE_PLANES(S_, row_, v4_)                                                                    \
    {                                                                                                      \
        if constexpr (F32) {                                                                               \
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(&S_[0][0]) + f32_off(row_, tk4)) = (v4_); \
        } else {                                                                                           \
            uint2 h_, l_;                                                                                  \
            split4((v4_).x, (v4_).y, (v4_).z, (v4_).w, h_, l_);                                            \
            *reinterpret_cast<uint2 *>(&S_[0][(row_) * LDH + wo]) = h_;                                    \
            *reinterpret_cast<uint2 *>(&S_[1][(row_) * LDH + wo]) = l_;                                    \
        }                                                                                                  \
    }
#define GRL_STORE_PLANES_B(S_, row_, v4_)                                                                  \
    {                                                                                                      \
        GRL_STORE_PLANES(S_, row_, v4_)                                                                    \
        if constexpr (ACC1) {                                                                              \
            uint2 h_, l_;                                                                                  \
            split4((v4_).x, (v4_).y, (v4_).z, (v4_).w, h_, l_);      /* (the compiler shares it with the one above) */ \
            const f16x2 k_ = {(_Float16)kLowScale, (_Float16)kLowScale};                                   \
            const f16x2 s0_ = __builtin_bit_cast(f16x2, h_.x) * k_, s1_ = __builtin_bit_cast(f16x2, h_.y) * k_; \
            *reinterpret_cast<uint2 *>(&S_[2][(row_) * LDH + wo]) = make_uint2(__builtin_bit_cast(unsigned, s0_), __builtin_bit_cast(unsigned, s1_)); \
        }                                                                                                  \
    }
#define GRL_STORE_TILE(buf_)                                                                               \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                   \
            const bool v = (vmask >> i) & 1u;                                                              \
            float4 t4 = ra[i];                                                                             \
            t4.x = v ? t4.x : 0.f; t4.y = v ? t4.y : 0.f; t4.z = v ? t4.z : 0.f; t4.w = v ? t4.w : 0.f;    \
            if (AG::kRelu) { t4.x = fmaxf(t4.x, 0.f); t4.y = fmaxf(t4.y, 0.f); t4.z = fmaxf(t4.z, 0.f); t4.w = fmaxf(t4.w, 0.f); } \
            GRL_STORE_PLANES(As[buf_], trow + RPP * i, t4)                                                 \
        }                                                                                                  \
        GRL_STORE_PLANES_B(Bs[buf_], trow, rb0)                                                            \
        if (NB > 1) GRL_STORE_PLANES_B(Bs[buf_], trow + RPP, rb1)                                          \
        if (NB > 2) GRL_STORE_PLANES_B(Bs[buf_], trow + 2 * RPP, rb2)                                      \
        if (NB > 2) GRL_STORE_PLANES_B(Bs[buf_], trow + 3 * RPP, rb3)                                      \
    }

    f32x4 acc[TM][TN], acl[ACC1 ? 1 : TM][ACC1 ? 1 : TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[a][b][r] = 0.f;
                if constexpr (!ACC1) acl[a][b][r] = 0.f;
            }

    const int nk = K / BK;
    // K-tiles whose tap is outside the image for the WHOLE workgroup are skipped (tile_ok is block-uniform)
    int kt = 0;
    while (kt < nk && !ag_tile_ok(ag, wctx, m0, kt * BK)) ++kt;
    GRL_LOAD_TILE(kt < nk ? kt : 0)
    // MFMA operand: lane (r = lane & 15, g = lane >> 4) holds k = 8g .. 8g+7 of row r -> one ds_read_b128 per plane
    const int l16 = lane & 15, kg = lane >> 4;
    const int ro = (kg ^ swz(l16)) << 3;
    const int aro = (wm * WM + l16) * LDH + ro;
    const int bro = (wn * WN + l16) * LDH + ro;
    // one K = 32 step on the tile in LDS buffer cb_
#define GRL_MFMA_TILE(cb_)                                                                                                             \
    {                                                                                                                                  \
        if constexpr (F32) {                                                                                                           \
            mfma_f32_step<TM, TN>(reinterpret_cast<const float *>(&As[cb_][0][0]), reinterpret_cast<const float *>(&Bs[cb_][0][0]),    \
                                  wm * WM + l16, wn * WN + l16, kg, acc);                                                              \
        } else if constexpr (ACC1) {                                                                                                   \
            f16x8 af[TM][2];                                                                                                           \
            _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                             \
                _Pragma("unroll") for (int p = 0; p < 2; ++p) af[a][p] = *reinterpret_cast<const f16x8 *>(&As[cb_][p][aro + a * 16 * LDH]); \
            _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                                           \
                f16x8 bf[3];                                                                                                           \
                _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(&Bs[cb_][p][bro + b * 16 * LDH]); \
                _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                                       \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][1], bf[0], acc[a][b], 0, 0, 0);                           \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[1], acc[a][b], 0, 0, 0);                           \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a][0], bf[2], acc[a][b], 0, 0, 0);                           \
                }                                                                                                                      \
            }                                                                                                                          \
        } else if constexpr (TM < TN) { /* keep the smaller operand's fragments live, stream the other one (fewer registers) */        \
            f16x8 af[TM][2];                                                                                                           \
            _Pragma("unroll") for (int a = 0; a < TM; ++a)                                                                             \
                _Pragma("unroll") for (int p = 0; p < 2; ++p) af[a][p] = *reinterpret_cast<const f16x8 *>(&As[cb_][p][aro + a * 16 * LDH]); \
            _Pragma("unroll") for (int b = 0; b < TN; ++b) {                                                                           \
                f16x8 bf[2];                                                                                                           \
                _Pragma("unroll") for (int p = 0; p < 2; ++p) bf[p] = *reinterpret_cast<const f16x8 *>(&Bs[cb_][p][bro + b * 16 * LDH]); \
                _Pragma("unroll") for (int a = 0; a < TM; ++a) mfma_x3(acc[a][b], acl[a][b], af[a], bf);                               \
            }                                                                                                                          \
        } else {                                                                                                                       \
            f16x8 bf[TN][2];                                                                                                           \
            _Pragma("unroll") for (int b = 0; b < TN; ++b)                                                                             \
                _Pragma("unroll") for (int p = 0; p < 2; ++p) bf[b][p] = *reinterpret_cast<const f16x8 *>(&Bs[cb_][p][bro + b * 16 * LDH]); \
            _Pragma("unroll") for (int a = 0; a < TM; ++a) {                                                                           \
                f16x8 af[2];                                                                                                           \
                _Pragma("unroll") for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const f16x8 *>(&As[cb_][p][aro + a * 16 * LDH]); \
                _Pragma("unroll") for (int b = 0; b < TN; ++b) mfma_x3(acc[a][b], acl[a][b], af, bf[b]);                               \
            }                                                                                                                          \
        }                                                                                                                              \
    }
    if constexpr (NBUF == 1) {
        while (kt < nk) {
            GRL_STORE_TILE(0)
            __syncthreads();
            int ktn = kt + 1;
            while (ktn < nk && !ag_tile_ok(ag, wctx, m0, ktn * BK)) ++ktn;
            {   // prefetch the next valid K-tile (the last iteration re-reads its own tile: branch-free)
                const int ktl = ktn < nk ? ktn : kt;
                GRL_LOAD_TILE(ktl)
            }
            if constexpr (FENCE) { GRL_SCHED_FENCE }
            GRL_MFMA_TILE(0)
            if constexpr (FENCE) { GRL_SCHED_FENCE }
            __syncthreads();
            kt = ktn;
        }
    } else {
        // tile kt is in buffer cb, tile ktn in the registers (loads in flight); an iteration stages ktn into the other buffer, starts
        // the loads of the tile after it and multiplies tile kt.  The barrier at its end covers both hazards: the staged tile is
        // complete before anyone reads it, and everyone has read buffer cb before the next iteration stages over it.
        int ktn = nk;
        if (kt < nk) {
            GRL_STORE_TILE(0)
            ktn = kt + 1;
            while (ktn < nk && !ag_tile_ok(ag, wctx, m0, ktn * BK)) ++ktn;
            GRL_LOAD_TILE(ktn < nk ? ktn : kt)
            __syncthreads();
        }
        int cb = 0;
        while (kt < nk) {
            int ktnn = ktn;
            if (ktn < nk) {      // block-uniform
                GRL_STORE_TILE(cb ^ 1)
                ktnn = ktn + 1;
                while (ktnn < nk && !ag_tile_ok(ag, wctx, m0, ktnn * BK)) ++ktnn;
                GRL_LOAD_TILE(ktnn < nk ? ktnn : ktn)
            }
            if constexpr (FENCE) { GRL_SCHED_FENCE }
            GRL_MFMA_TILE(cb)
            if constexpr (FENCE) { GRL_SCHED_FENCE }
            __syncthreads();
            kt = ktn;
            ktn = ktnn;
            cb ^= 1;
        }
    }
#undef GRL_MFMA_TILE
    // the cross terms carry 2^11
    if constexpr (!F32) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (ACC1) acc[a][b][r] *= 1.0f / kLowScale;      // ... and so does the main term here
                    else acc[a][b][r] = __builtin_fmaf(acl[a][b][r], 1.0f / kLowScale, acc[a][b][r]);
                }
    }
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg.  Two passes: all of the epilogue's loads, then
    // the stores (see the epilogue structs).
    // (PAIRN: wave column wn stores the tile's wn-th pixel; a missing second pixel puts the columns past N: nothing is stored)
    const int ecolbase = PAIRN ? ((wn == 0 ? pp.x : pp.y) < 0 ? N : (wn == 0 ? pp.x : pp.y) * 64) : n0 + wn * WN;
    const int erow = m0 + wm * WM + 4 * kg, ecol = ecolbase + l16;
    int rax[TM][4];
    float eax[TM][TN][4];
    bool out_of_range = false;
    float amax = 0.f;
    // the loads are unconditional on clamped coordinates: a per-element predicate would wrap every load in its own
    // exec-mask block with a wait behind it
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (epi_row_bits<Epi>::value || epi_row_auxn<Epi>::value) rax[a][r] = epi.row_aux_n(min(erow + a * 16 + r, M - 1), ecolbase);
            else rax[a][r] = epi.row_aux(min(erow + a * 16 + r, M - 1));
        }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                eax[a][b][r] = epi.elem_aux(min(erow + a * 16 + r, M - 1), min(ecol + b * 16, N - 1), rax[a][r]);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = erow + a * 16 + r, col = ecol + b * 16;
                if (row < M && col < N) epi.store(row, col, acc[a][b][r], rax[a][r], eax[a][b][r]);
                // range guard: what this tile hands to the next GEMM must be a finite fp16-range number (padding rows hold zeros)
                const float gv = epi_add_aux<Epi>::value ? acc[a][b][r] + eax[a][b][r] : acc[a][b][r];
                out_of_range |= !(fabsf(gv) <= kF16Max);
                if constexpr (F32) amax = fmaxf(amax, fabsf(gv));
            }
    if (!F32 && out_of_range) atomicOr(g_gemm_range_flag, 1);      // the fp32 form has no operand range to guard ...
    if constexpr (F32) {                                            // ... it records how far inside the fp16 range it is instead
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) atomicMax(g_gemm_absmax, __float_as_uint(amax));
    }
    if constexpr (Epi::kColSum) {      // column sums of the stored values over this wave's rows, in a fixed order
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            float sum = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (erow + a * 16 + r < M) sum += epi.value(acc[a][b][r], eax[a][b][r]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const int col = ecol + b * 16;
            if (kg == 0 && col < N) epi.csum[(long)(by * WGM + wm) * N + col] = sum;
        }
    }
    if constexpr (epi_row_bits<Epi>::value) {      // sign bits of the stored values: one word per row and 64-column block of this wave
        static_assert(WN == 64, "a wave covers exactly one 64-column block");
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned long long w = 0ull;
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    // lane (kg, l16) holds row 4*kg + r, column 16*b + l16: bits [16 kg, 16 kg + 16) of the ballot are this row's
                    const unsigned long long bal = __ballot(epi.bit(acc[a][b][r], eax[a][b][r]));
                    w |= ((bal >> (16 * kg)) & 0xFFFFull) << (16 * b);
                }
                const int row = erow + a * 16 + r;
                if (l16 == 0 && row < M) epi.store_bits(row, ecolbase, w, rax[a][r]);
            }
    }
#undef GRL_LOAD_TILE
#undef GRL_STORE_TILE
#undef GRL_STORE_PLANES
#undef GRL_STORE_PLANES_B
}

// ---------------------------------------------------------------------------- C[I,J] = A^T * B over rows m
// A element (m, i) = gathered patch element i of row m (same descriptors as above, i plays the role
// of k); B = dY[m][J] dense.  Block (bi, bj, chunk) reduces rows [chunk*mc, (chunk+1)*mc) and writes
// its partial tile to slab[chunk][I][J]; a follow-up kernel sums the slabs in a fixed order
// (deterministic, unlike float atomics).
// The reduction index m is the ROW of both operands in memory, the MFMA wants it contiguous per lane: a thread loads
// NA consecutive rows of the same four columns and writes each column's NA values (packed bf16) to the transposed LDS
// tile [column][m].  Column 4*c4 + j sits in LDS row j*(BM/4) + c4, which spreads a store's lanes over the banks; the
// epilogue undoes the permutation.
template <int BM, int BN, int WGM, int WGN, class AG, int XCD_ORDER = 1, bool F32 = false>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void gemm_tn(AG ag, const float *__restrict__ dY, int J, int mc, float *__restrict__ slab) {
    constexpr int BK = 32, LDH = kLdh, NT = 64 * WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int A4 = BM / 4, B4 = BN / 4;           // float4 per reduction row
    constexpr int NA = BK * A4 / NT, NB = BK * B4 / NT;
    static_assert((WGM * WGN == 4 || WGM * WGN == 8) && NA >= 1 && NB >= 1 && NA <= 4 && NB <= 4 && (NA != 3) && (NB != 3), "tile size");
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * LDH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave - wm * WGN;
    // XCD_ORDER 1: all tiles of one row range (grid z) on one XCD, so its L2 serves the re-reads of the same rows (see gemm_rowk).
    // XCD_ORDER 2: inside a row range the J tiles of one I tile share an XCD (runs of 8 I tiles are transposed), so the A columns
    // of an I tile are fetched into one L2 instead of gridDim.y of them -- for row ranges too large for one L2 (dense1 patch).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (XCD_ORDER == 2) {
        const int nx = gridDim.x, ny = gridDim.y, l = by * nx + bx, g = l / (8 * ny);
        if ((g + 1) * 8 <= nx) {
            const int r = l - g * 8 * ny;
            bx = g * 8 + (r & 7);
            by = r >> 3;
        } else {      // the last partial run of I tiles, compacted: l - 8 ny g indexes (bx - 8 g, by) with bx fastest
            const int r = l - g * 8 * ny, w = nx - g * 8;
            by = r / w;
            bx = g * 8 + (r - by * w);
        }
    }
    if (XCD_ORDER == 1) {
        const int nx = gridDim.x, T = nx * gridDim.y, L = bz * T + by * nx + bx, g = L / (8 * T);
        if ((g + 1) * 8 <= (int)gridDim.z) {
            const int l = L - g * 8 * T, t = l >> 3;
            bz = g * 8 + (l & 7);
            by = t / nx;
            bx = t - by * nx;
        }
    }
    constexpr bool PAIR = ag_pair_i<AG>::value;      // the I tile is two independent 64-column runs (PatchRowsPair)
    const int i0 = bx * BM, j0 = by * BN;
    const int I = ag.K();
    int mbeg, mend;
    ag.mrange(bz, mc, mbeg, mend);
    int2 pp = make_int2(0, 0);
    if constexpr (PAIR) {
        static_assert(!PAIR || BM == 128, "two 64-column halves");
        pp = ag.pair(bz, bx);
        if (pp.x < 0) return;                    // fewer live pixels than this tile's first: no slab tile
    } else if constexpr (ag_has_i_ok<AG>::value) {      // output rows that are zero for the whole row range: no slab tile; the reduction
        if (!ag.i_ok(bz, i0)) return;            // that follows knows the same masks (patch_dw_reduce_kernel)
    }

    // (PAIR: pp names two 64-column I tiles; a thread's four columns lie in one of them, so tap and validity are per thread)
    int toff, ty, tx;
    ag.tap(PAIR ? (((int)(threadIdx.x % (BM / 4)) < 16 ? pp.x : max(pp.y, 0)) * 64) : i0, toff, ty, tx);   // the BM-wide (64-wide) column run lies inside one tap row (checked on the host)

    // this thread's part of a tile: rows NA*(tid/A4) .. +NA-1 (A), NB*(tid/B4) .. +NB-1 (B), columns 4*ca .. 4*ca+3 / 4*cb ..
    // LDS has 32 banks of 4 bytes: a store instruction is served 128 bytes (32 x 4-byte or 16 x 8-byte lanes) per pass.  With the
    // plain map (row group = tid / A4) the lanes of one pass share a row group, i.e. one 8-byte half (NA = 4) or one dword pair
    // (NA = 2) of their 16-byte chunk, and columns c, c + 2 meet in the same bank: a 2-way conflict on every staging store
    // (SQ_LDS_BANK_CONFLICT 4-6 % of the wave cycles of every gemm_tn instance, round 2).  Flipping the row group's half / pair by
    // bit 1 of the column index spreads a pass over all 32 banks; which (row, column) elements a thread carries changes, where
    // they land in LDS does not.
    const int ca = tid % A4, cb = tid % B4;
    int ga = tid / A4, gb = tid / B4;
    if (NA == 4) ga ^= (ca >> 1) & 1;
    if (NA == 2) ga ^= ((ca >> 1) & 1) << 1;
    if (NB == 4) gb ^= (cb >> 1) & 1;
    if (NB == 2) gb ^= ((cb >> 1) & 1) << 1;
    if (NB == 1) gb ^= ((cb >> 1) & 1) << 2;      // one row per thread, stored as (m, m+1) pairs after a lane exchange: see GRL_STORE_TILE
    const int ma = NA * ga, mb = NB * gb;
    // column offset of this thread's four A columns inside the row, and whether its half of a pixel pair exists
    const int acol = PAIR ? toff + (ca & 15) * 4 : toff + ca * 4;
    const bool ahalf_ok = !PAIR || ca < 16 || pp.y >= 0;
    float4 ra[NA], rb[NB];
    unsigned vma = 0, vmb = 0;
    // physical rows of the tile about to be loaded (-1: past the range / padding).  They are fetched one tile ahead of
    // the data so that an indirection (PatchRows' sorted order) does not put two dependent global loads in one stage.
    int2 ia[NA];
    int ib[NB];
    unsigned iva = 0;      // which handles in ia name a row inside the range (and not a padding row)
#define GRL_LOAD_IDX(mt_)                                                                                          \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
            const int m = (mt_) + ma + i;                                                                          \
            ia[i] = ag.ahandle(m < mend ? m : mbeg);                                                               \
            iva = (m < mend && ag.hvalid(ia[i])) ? (iva | (1u << i)) : (iva & ~(1u << i));                         \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                           \
            const int m = (mt_) + mb + i;                                                                          \
            ib[i] = ag.bhandle(m < mend ? m : mbeg);                                                               \
            if (m >= mend) ib[i] = -1;                                                                             \
        }                                                                                                          \
    }
#define GRL_LOAD_TILE()                                                                                            \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
            const bool inr = (iva >> i) & 1u;                                                                      \
            long off; int iy0, ix0;                                                                                \
            ag.rowh(ia[i], off, iy0, ix0);                                                                         \
            const bool v = inr && ahalf_ok && ag.ok(iy0, ix0, ty, tx);                                             \
            ra[i] = *reinterpret_cast<const float4 *>(ag.base + (v ? off + acol : (long)(ca * 4)));                \
            vma = v ? (vma | (1u << i)) : (vma & ~(1u << i));                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                           \
            const bool v = ib[i] >= 0;                                                                             \
            rb[i] = *reinterpret_cast<const float4 *>(dY + (long)(v ? ib[i] : 0) * J + j0 + cb * 4);               \
            vmb = v ? (vmb | (1u << i)) : (vmb & ~(1u << i));                                                      \
        }                                                                                                          \
    }
    // tile row rho, reduction position m (0..31) -> swizzled LDS offset
#define GRL_TN_OFF(rho_, m_) ((rho_) * LDH + ((((m_) >> 3) ^ swz(rho_)) << 3) + ((m_) & 7))
    // one column's NV consecutive-m values -> NV packed fp16 per plane at S[p][GRL_TN_OFF(rho_, m_)] (F32: NV floats of the fp32 tile)
#define GRL_STORE_RUN(S_, NV_, rho_, m_, r_, comp_)                                                                \
    {                                                                                                              \
        if constexpr (F32) {                                                                                       \
            float *f_ = reinterpret_cast<float *>(&S_[0][0]) + f32_off(rho_, m_);                                  \
            if (NV_ == 4) *reinterpret_cast<float4 *>(f_) = make_float4(r_[0].comp_, r_[1 % NV_].comp_, r_[2 % NV_].comp_, r_[3 % NV_].comp_); \
            else if (NV_ == 2) *reinterpret_cast<float2 *>(f_) = make_float2(r_[0].comp_, r_[1 % NV_].comp_);      \
            else *f_ = r_[0].comp_;                                                                                \
        } else {                                                                                                   \
        const int o_ = GRL_TN_OFF(rho_, m_);                                                                       \
        if (NV_ == 4) {                                                                                            \
            uint2 h_, l_;                                                                                          \
            split4(r_[0].comp_, r_[1 % NV_].comp_, r_[2 % NV_].comp_, r_[3 % NV_].comp_, h_, l_);                  \
            *reinterpret_cast<uint2 *>(&S_[0][o_]) = h_;                                                           \
            *reinterpret_cast<uint2 *>(&S_[1][o_]) = l_;                                                           \
        } else {                                                                                                   \
            unsigned h_, l_;                                                                                       \
            split2(r_[0].comp_, r_[1 % NV_].comp_, h_, l_);                                                        \
            if (NV_ == 2) {                                                                                        \
                *reinterpret_cast<unsigned *>(&S_[0][o_]) = h_;                                                    \
                *reinterpret_cast<unsigned *>(&S_[1][o_]) = l_;                                                    \
            } else {                                                                                               \
                S_[0][o_] = (unsigned short)(h_ & 0xFFFFu);                                                        \
                S_[1][o_] = (unsigned short)(l_ & 0xFFFFu);                                                        \
            }                                                                                                      \
        }                                                                                                          \
        }                                                                                                          \
    }
#define GRL_STORE_TILE()                                                                                           \
    {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
            const bool v = (vma >> i) & 1u;                                                                        \
            float4 t4 = ra[i];                                                                                     \
            t4.x = v ? t4.x : 0.f; t4.y = v ? t4.y : 0.f; t4.z = v ? t4.z : 0.f; t4.w = v ? t4.w : 0.f;            \
            if (AG::kRelu) { t4.x = fmaxf(t4.x, 0.f); t4.y = fmaxf(t4.y, 0.f); t4.z = fmaxf(t4.z, 0.f); t4.w = fmaxf(t4.w, 0.f); } \
            ra[i] = t4;                                                                                            \
        }                                                                                                          \
        GRL_STORE_RUN(As, NA, 0 * A4 + ca, ma, ra, x)                                                  \
        GRL_STORE_RUN(As, NA, 1 * A4 + ca, ma, ra, y)                                                  \
        GRL_STORE_RUN(As, NA, 2 * A4 + ca, ma, ra, z)                                                  \
        GRL_STORE_RUN(As, NA, 3 * A4 + ca, ma, ra, w)                                                  \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                           \
            const bool v = (vmb >> i) & 1u;                                                                        \
            float4 t4 = rb[i];                                                                                     \
            t4.x = v ? t4.x : 0.f; t4.y = v ? t4.y : 0.f; t4.z = v ? t4.z : 0.f; t4.w = v ? t4.w : 0.f;            \
            rb[i] = t4;                                                                                            \
        }                                                                                                          \
        if (NB == 1 && !F32) {      /* one row per thread: trade two columns with the thread holding the neighbouring row (same columns) */ \
            const bool odd = gb & 1;                                                                               \
            const float4 t4 = rb[0];                                                                               \
            const float o0 = __shfl_xor(odd ? t4.x : t4.z, B4), o1 = __shfl_xor(odd ? t4.y : t4.w, B4);            \
            unsigned h0, l0, h1, l1;                                                                               \
            split2(odd ? o1 : t4.x, odd ? t4.w : o0, h0, l0);      /* even: column x rows (m, m+1); odd: column w rows (m-1, m) */ \
            split2(odd ? o0 : t4.y, odd ? t4.z : o1, h1, l1);      /* even: column y; odd: column z */            \
            const int q0 = GRL_TN_OFF((odd ? 3 : 0) * B4 + cb, mb & ~1), q1 = GRL_TN_OFF((odd ? 2 : 1) * B4 + cb, mb & ~1); \
            *reinterpret_cast<unsigned *>(&Bs[0][q0]) = h0;                                                        \
            *reinterpret_cast<unsigned *>(&Bs[1][q0]) = l0;                                                        \
            *reinterpret_cast<unsigned *>(&Bs[0][q1]) = h1;                                                        \
            *reinterpret_cast<unsigned *>(&Bs[1][q1]) = l1;                                                        \
        } else {                                                                                                   \
        GRL_STORE_RUN(Bs, NB, 0 * B4 + cb, mb, rb, x)                                                  \
        GRL_STORE_RUN(Bs, NB, 1 * B4 + cb, mb, rb, y)                                                  \
        GRL_STORE_RUN(Bs, NB, 2 * B4 + cb, mb, rb, z)                                                  \
        GRL_STORE_RUN(Bs, NB, 3 * B4 + cb, mb, rb, w)                                                  \
        }                                                                                                          \
    }
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
    const int bro = (wn * WN + l16) * LDH + ro;
    if (mbeg < mend) {
        GRL_LOAD_IDX(mbeg)
        GRL_LOAD_TILE()
        GRL_LOAD_IDX(mbeg + BK)
        for (int mt = mbeg; mt < mend; mt += BK) {
            GRL_STORE_TILE()
            __syncthreads();
            GRL_LOAD_TILE()                  // rows mt + BK .. (row 0, masked, past the end of the range)
            GRL_LOAD_IDX(mt + 2 * BK)
            {
                if constexpr (F32) {
                    mfma_f32_step<TM, TN>(reinterpret_cast<const float *>(&As[0][0]), reinterpret_cast<const float *>(&Bs[0][0]),
                                          wm * WM + l16, wn * WN + l16, kg, acc);
                } else if constexpr (TM < TN) {      // keep the smaller operand's fragments live, stream the other one (fewer registers)
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
                        for (int a = 0; a < TM; ++a) mfma_x3(acc[a][b], acl[a][b], af[a], bf);
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
                        for (int b = 0; b < TN; ++b) mfma_x3(acc[a][b], acl[a][b], af, bf[b]);
                    }
                }
            }
            __syncthreads();
        }
    }
    // the cross terms carry 2^11
    if constexpr (!F32) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = __builtin_fmaf(acl[a][b][r], 1.0f / kLowScale, acc[a][b][r]);
    }
    float *out = slab + (long)bz * I * J;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pa = wm * WM + a * 16 + 4 * kg + r;      // LDS rows -> tile columns
                const int pb = wn * WN + b * 16 + l16;
                int row = i0 + 4 * (pa % A4) + pa / A4;
                const int col = j0 + 4 * (pb % B4) + pb / B4;
                if constexpr (PAIR) {      // tile column c: pixel of its half, channel c & 63
                    const int c = row - i0, px = c < 64 ? pp.x : pp.y;
                    row = px < 0 ? I : px * 64 + (c & 63);
                }
                if (row < I && col < J) out[(long)row * J + col] = acc[a][b][r];
            }
#undef GRL_LOAD_TILE
#undef GRL_LOAD_IDX
#undef GRL_STORE_TILE
#undef GRL_STORE_RUN
#undef GRL_TN_OFF
}

// (slab_reduce_kernel: its body is a job of reduce_batch_kernel since round 4, net_reduce.inc)

// (slab_reduce_groups_kernel: its body is a job of reduce_batch_kernel since round 4, net_reduce.inc)

// Same for a short row (n up to a few thousand) and many chunks, where one lane per element would walk the chunks
// serially: 4 elements per workgroup, a wave per element -- lane p sums the chunks p, p + 64, ... in four interleaved partial sums,
// the 64 lanes are folded in a fixed shuffle tree.  (Until round 3: 16 elements per workgroup and 16 lanes per element -- 4 workgroups
// for a 64-wide bias gradient over 1 500 partial rows, 11 us of mostly waiting; ~20 such launches per chunk of the gradient step.)
constexpr int kNarrowCols = 4;
__global__ __launch_bounds__(256) void slab_reduce_narrow_kernel(const float *__restrict__ slab, int chunks, int n, float *__restrict__ dst,
                                                                 int accumulate) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * kNarrowCols + (threadIdx.x >> 6);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int c = lane;
        for (; c + 192 < chunks; c += 256) {
            s0 += slab[(long)c * n + i];
            s1 += slab[(long)(c + 64) * n + i];
            s2 += slab[(long)(c + 128) * n + i];
            s3 += slab[(long)(c + 192) * n + i];
        }
        for (; c < chunks; c += 64) s0 += slab[(long)c * n + i];
    }
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);      // every lane ends with the same value: a fixed tree
    if (lane == 0 && i < n) dst[i] = accumulate ? dst[i] + s : s;
}

// column sums of dY[M][J] over row chunks -> slab[chunk][J]   (bias gradients)
// 256 threads: min(J,256) column lanes x (256/J) row lanes, row lanes combined through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ dY, int M, int J, int mc, float *__restrict__ slab) {
    // J in {32, 64, 256, 512, 1024}: a row is J/4 float4 lanes, 256/(J/4) rows per pass, four passes in flight
    __shared__ float4 red[256];
    const int jv = J >> 2;
    const int rl = 256 / jv;
    const int jl = threadIdx.x % jv, rsub = threadIdx.x / jv;
    const int mbeg = blockIdx.y * mc, mend = min(M, mbeg + mc);
    const float4 *__restrict__ p = reinterpret_cast<const float4 *>(dY) + jl;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int m = mbeg + rsub;
    for (; m + 3 * rl < mend; m += 4 * rl) {
        const float4 v0 = p[(long)m * jv], v1 = p[(long)(m + rl) * jv], v2 = p[(long)(m + 2 * rl) * jv], v3 = p[(long)(m + 3 * rl) * jv];
        s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
        s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
        s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
        s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
    }
    for (; m < mend; m += rl) {
        const float4 v0 = p[(long)m * jv];
        s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
    }
    float4 s = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                           (s0.w + s1.w) + (s2.w + s3.w));
    red[threadIdx.x] = s;
    __syncthreads();
    if (rsub == 0) {
        for (int r = 1; r < rl; ++r) {
            const float4 v = red[r * jv + jl];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4 *>(slab + (long)blockIdx.y * J)[jl] = s;
    }
}

}  // namespace grl
