// gemm_bf16x3.hip -- C = epilogue(A * B^T) in fp32 accuracy on the bf16 matrix cores of gfx950 (MI355X), for the products of
// the 44446 image-token rows with FROZEN weights (reference FFN: models/GroundingDINO/transformer_for_adapter.py:877-886,
// fusion projections fuse_modules.py:99-248; every ZiRa task freezes them: groundingdino_dual_zero_rep_branch.py:722-745).
//
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 vector rate, 157 TF/s, and the library's fp32 GEMMs reach
// 107-143 TF/s of it on these shapes -- half of a training step.  v_mfma_f32_32x32x16_bf16 is 16 x faster per instruction.
// An fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significant bits, each rounded to nearest):
//     a = a1 + a2 + a3,   |a2| <= 2^-9 |a|,   |a3| <= 2^-18 |a|
// and a product of two bf16 numbers is exact in fp32.  With both operands split,
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1) + [a2 b3 + a3 b2 + a3 b3]
// and the bracket is below 2^-26 |a b|, a quarter of the rounding error of ONE fp32 multiplication: six bf16 MFMA terms per
// fragment pair give a product whose only error is the fp32 accumulation inside the matrix core -- which adds 16 products per
// instruction where an fp32 fma chain rounds after every one.  tests/test_gemm_bf16x3_gpu.py holds it against an fp64
// product beside the library's fp32 GEMM on the model's shapes.
// The frozen weight is split ONCE into three bf16 planes [3][N][K] (zira_split_bf16x3_f32; K contiguous, whatever the
// weight's own orientation); the activation is split on the fly between its global load and the LDS tile.
//
// Kernel: a block = BM x 128 tile of C (BM = 128, or 192 where that fills the chip's 512 block slots better), 256 threads =
// 2 x 2 waves of (BM / 2) x 64; K in steps of 32 through ONE LDS stage per operand and plane (rows padded to 80 bytes: the
// sixteen 16-byte fragment reads of a ds_read_b128 lane group fall on sixteen different bank quads), the next step's global
// loads in flight during the MFMAs, two blocks per CU so that one block's split / LDS-write phase runs beside the other's
// MFMAs.  The matrix core computes C^T tiles (its A operand is the weight fragment, its B operand the activation fragment):
// a lane then holds four consecutive columns of one row of C per four accumulator registers, and the epilogue reads its
// operands and writes C sixteen bytes per lane.  Epilogues: + bias, + bias and ReLU, mask by (aux > 0), + aux.
// Non-finite inputs give NaN (inf - inf in the split); the callers' activations are finite or the step is lost anyway.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "zira_msda.h"

#ifndef ZIRA_G3_CHUNK
#define ZIRA_G3_CHUNK 32   // K depth the matrix core sums from zero before the vector unit adds it to the running sum (32, 16; 0: never)
#endif
#ifndef ZIRA_G3_BM192_MARGIN
#define ZIRA_G3_BM192_MARGIN 0.05   // the 192-row tile is taken where it wastes this much less of the last round of block slots
#endif
#ifndef ZIRA_G3_DEV_NOLOAD
#define ZIRA_G3_DEV_NOLOAD 0
#endif
#ifndef ZIRA_G3_STAGGER
#define ZIRA_G3_STAGGER 0   // s_sleep argument (x 64 cycles) by which the block in a CU's second wave slot starts late
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBN = 128, kBK = 32, kThreads = 256;
constexpr int kRow = 80;   // bytes of an LDS row: 32 bf16 + 16 bytes of padding

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_ADD = 3 };

__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    f32x2 x = {a, b};
    bf16x2 h = __builtin_convertvector(x, bf16x2);   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }

// four fp32 numbers -> their three bf16 planes, four bf16 (8 bytes) each
__device__ __forceinline__ void split4(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3)
{
    p1.x = pk_bf16(v.x, v.y);
    p1.y = pk_bf16(v.z, v.w);
    const float rx = v.x - bf_lo(p1.x), ry = v.y - bf_hi(p1.x), rz = v.z - bf_lo(p1.y), rw = v.w - bf_hi(p1.y);   // exact
    p2.x = pk_bf16(rx, ry);
    p2.y = pk_bf16(rz, rw);
    p3.x = pk_bf16(rx - bf_lo(p2.x), ry - bf_hi(p2.x));   // (the differences are exact, and fit bf16 exactly)
    p3.y = pk_bf16(rz - bf_lo(p2.y), rw - bf_hi(p2.y));
}

// The epilogue: accumulator register 4 g + i of block (ni, mi) is C[m][n], m = row (lane & 31) of the block, n = 8 g + 4 (lane >> 5) + i
template <int MI, int NI, int EPI>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[NI][MI], const float *__restrict__ bias, const float *aux, float *C,
                                           int M, int N, int mbase, int nbase, int lane)
{
    const int lm = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mbase + mi * 32 + lm;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + ni * 32 + 8 * g + 4 * lh;
                const f32x16 &c = acc[ni][mi];
                float4 o = make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                const size_t at = (size_t)m * N + n;
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + n);
                    o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                    if (EPI == EPI_BIAS_RELU) {
                        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                    }
                } else {
                    const float4 h = *reinterpret_cast<const float4 *>(aux + at);
                    if (EPI == EPI_MASK) {
                        o.x = h.x > 0.f ? o.x : 0.f; o.y = h.y > 0.f ? o.y : 0.f;
                        o.z = h.z > 0.f ? o.z : 0.f; o.w = h.w > 0.f ? o.w : 0.f;
                    } else {
                        o.x += h.x; o.y += h.y; o.z += h.z; o.w += h.w;
                    }
                }
                *reinterpret_cast<float4 *>(C + at) = o;
            }
    }
}

template <int BM, int EPI>
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16x3_kernel(const float *__restrict__ A, const unsigned short *__restrict__ Bp,
                                                                 const float *__restrict__ bias, const float *aux, float *C, int M,
                                                                 int N, int K, int row_tiles, int col_tiles, int rt_per_xcd)
{
    constexpr int WM = BM / 2, MI = WM / 32, NI = 2;   // a wave: WM x 64 of C = MI x NI blocks of 32 x 32
    constexpr int AJ = BM / 32;                        // float4 loads of A per thread and K step
    constexpr int CH = (ZIRA_G3_CHUNK == 32 && BM > 128) ? 16 : ZIRA_G3_CHUNK;   // (the 192-row tile has no registers for both slices' fragments)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sA = smem;                          // [3][BM][kRow]
    unsigned char *sB = smem + 3 * BM * kRow;          // [3][kBN][kRow]

    // tile of this block: blocks b, b + 8, ... share an XCD (placement is for speed only); an XCD walks its own range of row
    // tiles, the column tiles of a row tile side by side (they share the rows of A in its L2)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int trow = xcd * rt_per_xcd + idx / col_tiles, tcol = idx % col_tiles;
    if (trow >= row_tiles || idx / col_tiles >= rt_per_xcd) return;
    const int m0 = trow * BM, n0 = tcol * kBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#if ZIRA_G3_STAGGER
    // The two blocks of a CU run the same program with the same period: started together they reach their barriers, their
    // split / LDS-write phase and their MFMA phase together, and the matrix pipe idles while both write.  The block whose
    // waves sit in the odd wave slots (HW_ID bits 3:0) starts half a K step late; blocks that replace finished ones inherit
    // the slot and with it the phase.
    if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u) __builtin_amdgcn_s_sleep(ZIRA_G3_STAGGER);
#endif

    // global -> register staging
    const int a_chunk = tid & 7, a_row = tid >> 3;     // row a_row + 32 j, floats 4 a_chunk .. + 3 of the K step
    const int b_chunk = tid & 3, b_row = tid >> 2;     // row b_row + 64 j, bf16 8 b_chunk .. + 7
    const float *ag[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int r = m0 + a_row + 32 * j;
        r = r < M ? r : M - 1;                         // (rows past the end: any finite data, their results are not stored)
        ag[j] = A + (size_t)r * K + a_chunk * 4;
    }
    const unsigned short *bg = Bp + (size_t)(n0 + b_row) * K + b_chunk * 8;
    const size_t bplane = (size_t)N * K, bj = (size_t)64 * K;

    // (staging registers as named scalars and fully unrolled code: arrays handed to lambdas by reference ended up in scratch)
    float4 ra[AJ];
    uint4 rb00, rb01, rb10, rb11, rb20, rb21;
#define ZIRA_GLOAD(k0_)                                                                             \
    do {                                                                                            \
        _Pragma("unroll") for (int j = 0; j < AJ; ++j) ra[j] = *reinterpret_cast<const float4 *>(ag[j] + (k0_)); \
        rb00 = *reinterpret_cast<const uint4 *>(bg + (k0_));                                        \
        rb01 = *reinterpret_cast<const uint4 *>(bg + bj + (k0_));                                   \
        rb10 = *reinterpret_cast<const uint4 *>(bg + bplane + (k0_));                               \
        rb11 = *reinterpret_cast<const uint4 *>(bg + bplane + bj + (k0_));                          \
        rb20 = *reinterpret_cast<const uint4 *>(bg + 2 * bplane + (k0_));                           \
        rb21 = *reinterpret_cast<const uint4 *>(bg + 2 * bplane + bj + (k0_));                      \
    } while (0)
    unsigned char *const wa = sA + a_row * kRow + a_chunk * 8;
    unsigned char *const wb = sB + b_row * kRow + b_chunk * 16;

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ni][mi][i] = 0.f;

    // fragment addresses: lane l holds k = 8 (l >> 5) .. + 7 of row (l & 31) of its 32-row block
    const unsigned char *fa = sA + (wm * WM + (lane & 31)) * kRow + (lane >> 5) * 16;
    const unsigned char *fb = sB + (wn * 64 + (lane & 31)) * kRow + (lane >> 5) * 16;

    ZIRA_GLOAD(0);
    for (int k0 = 0; k0 < K; k0 += kBK) {
        __syncthreads();   // the previous step's fragment reads are done
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            uint2 p1, p2, p3;
#if defined(ZIRA_G3_DEV_NOSPLIT)   // (developer ablation: what the split costs -- wrong results)
            p1.x = __float_as_uint(ra[j].x); p1.y = __float_as_uint(ra[j].y); p2.x = __float_as_uint(ra[j].z); p2.y = __float_as_uint(ra[j].w);
            p3 = p1;
#else
            split4(ra[j], p1, p2, p3);
#endif
            unsigned char *d = wa + 32 * j * kRow;
            *reinterpret_cast<uint2 *>(d) = p1;
            *reinterpret_cast<uint2 *>(d + BM * kRow) = p2;
            *reinterpret_cast<uint2 *>(d + 2 * BM * kRow) = p3;
        }
        *reinterpret_cast<uint4 *>(wb) = rb00;
        *reinterpret_cast<uint4 *>(wb + 64 * kRow) = rb01;
        *reinterpret_cast<uint4 *>(wb + kBN * kRow) = rb10;
        *reinterpret_cast<uint4 *>(wb + kBN * kRow + 64 * kRow) = rb11;
        *reinterpret_cast<uint4 *>(wb + 2 * kBN * kRow) = rb20;
        *reinterpret_cast<uint4 *>(wb + 2 * kBN * kRow + 64 * kRow) = rb21;
        __syncthreads();
#if !ZIRA_G3_DEV_NOLOAD   // (developer ablation: the K loop without its global loads -- wrong results, what the loads cost)
        if (k0 + kBK < K) ZIRA_GLOAD(k0 + kBK);
#endif
        if constexpr (CH == 32) {
        // Six terms per 16-deep slice, the small ones first; matrix-core A operand = weight fragment (rows n), B operand =
        // activation fragment (rows m): the accumulator block is C^T [n][m].  The twelve terms of a K step are summed inside
        // the matrix core FROM ZERO and the step's sum is added to the running sum by the vector unit (round to nearest):
        // whatever rounding the matrix core applies to its accumulator then acts on a short partial sum only, and the long
        // sum over K sees K / 32 correctly rounded additions where an fp32 fma chain has K (scripts/gemm_bf16x3_accuracy.py).
        {
            bf16x8 b[2][NI][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        b[ks][ni][p] = *reinterpret_cast<const bf16x8 *>(fb + p * kBN * kRow + ni * 32 * kRow + ks * 32);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                bf16x8 a[2][3];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        a[ks][p] = *reinterpret_cast<const bf16x8 *>(fa + p * BM * kRow + mi * 32 * kRow + ks * 32);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][ni][2], a[ks][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][ni][1], a[ks][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][ni][0], a[ks][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][ni][1], a[ks][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][ni][0], a[ks][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][ni][0], a[0][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1][ni][0], a[1][0], c, 0, 0, 0);
                    acc[ni][mi] += c;
                }
            }
        }
        } else {
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            // chunks of 16: as above with six terms per chunk (-DZIRA_G3_CHUNK=16), or the matrix core's own accumulation
            // throughout (-DZIRA_G3_CHUNK=0: fails the accuracy gate on post-ReLU operands at K = 2048)
            bf16x8 b[NI][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    b[ni][p] = *reinterpret_cast<const bf16x8 *>(fb + p * kBN * kRow + ni * 32 * kRow + ks * 32);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[p] = *reinterpret_cast<const bf16x8 *>(fa + p * BM * kRow + mi * 32 * kRow + ks * 32);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (CH == 0) c = acc[ni][mi];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][2], a[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][1], a[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][0], a[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][1], a[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][0], a[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][0], a[0], c, 0, 0, 0);
                    if (CH == 0) acc[ni][mi] = c;
                    else acc[ni][mi] += c;
                }
            }
        }
        }
    }

#undef ZIRA_GLOAD
    store_tile<MI, NI, EPI>(acc, bias, aux, C, M, N, m0 + wm * WM, n0 + wn * 64, lane);
}

// (Measured and dropped in round 5: a wave-specialised form -- eight waves per block, a producer and a consumer wave on every
// SIMD, two LDS stages, one barrier per K step, one block per CU.  Correct, and slower: 1821-1831 us for the model's eight
// products against 1567-1584 us for the kernel above, with the loads one or two K steps ahead alike; the tile's epilogue and
// prologue overlap with nothing when a CU holds one block, which the K = 256 products pay most.)

// W [rows][cols] fp32 -> planes [3][N][K] bf16 with B[n][k] = W[n][k] (transpose = 0: N = rows, K = cols) or W[k][n]
// (transpose = 1: N = cols, K = rows)
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ w, int rows, int cols, int transpose,
                                                           unsigned short *__restrict__ planes)
{
    const int N = transpose ? cols : rows, K = transpose ? rows : cols;
    const size_t total = (size_t)N * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e / K), k = (int)(e % K);
        const float v = transpose ? w[(size_t)k * cols + n] : w[e];
        const unsigned p1 = pk_bf16(v, 0.f);
        const float r1 = v - bf_lo(p1);
        const unsigned p2 = pk_bf16(r1, 0.f);
        const unsigned p3 = pk_bf16(r1 - bf_lo(p2), 0.f);
        planes[e] = (unsigned short)(p1 & 0xFFFFu);
        planes[total + e] = (unsigned short)(p2 & 0xFFFFu);
        planes[2 * total + e] = (unsigned short)(p3 & 0xFFFFu);
    }
}

template <int BM, int EPI>
int launch(const float *a, const unsigned short *bp, const float *bias, const float *aux, float *c, int M, int N, int K, hipStream_t st)
{
    const int rt = (M + BM - 1) / BM, ct = N / kBN, per = (rt + 7) / 8;
    static bool attr_set = false;
    const size_t lds = (size_t)3 * (BM + kBN) * kRow;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3_kernel<BM, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16x3_kernel<BM, EPI>), dim3(8 * per * ct), dim3(kThreads), lds, st, a, bp, bias, aux, c, M, N, K, rt, ct, per);
    return (int)hipGetLastError();
}

template <int BM>
int launch_epi(int epi, const float *a, const unsigned short *bp, const float *bias, const float *aux, float *c, int M, int N, int K,
               hipStream_t st)
{
    switch (epi) {
    case EPI_BIAS: return launch<BM, EPI_BIAS>(a, bp, bias, aux, c, M, N, K, st);
    case EPI_BIAS_RELU: return launch<BM, EPI_BIAS_RELU>(a, bp, bias, aux, c, M, N, K, st);
    case EPI_MASK: return launch<BM, EPI_MASK>(a, bp, bias, aux, c, M, N, K, st);
    case EPI_ADD: return launch<BM, EPI_ADD>(a, bp, bias, aux, c, M, N, K, st);
    }
    return -1;
}

}  // namespace

extern "C" int zira_split_bf16x3_f32(const float *w, int rows, int cols, int transpose, void *planes, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!w || !planes || rows <= 0 || cols <= 0) return -1;
    const size_t total = (size_t)rows * cols;
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, stream, w, rows, cols, transpose ? 1 : 0,
                       reinterpret_cast<unsigned short *>(planes));
    return (int)hipGetLastError();
}

extern "C" int zira_gemm_bf16x3_f32(const float *a, const void *b_planes, int M, int N, int K, int epilogue, const float *bias,
                                    const float *aux, float *c, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!a || !b_planes || !c || M <= 0 || N <= 0 || K <= 0 || N % kBN || K % kBK) return -1;
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) ? !bias : !aux) return -1;
    if (((uintptr_t)a | (uintptr_t)b_planes | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)aux) & 15) return -1;
    if ((unsigned long long)M * N >= (1ull << 40)) return -1;
    const unsigned short *bp = reinterpret_cast<const unsigned short *>(b_planes);
    // tile height: the one that wastes fewer of the chip's 512 block slots in its last round
    auto waste = [&](int bm) {
        const long long slots = 512;   // blocks the chip holds at a time
        const long long tiles = (long long)((M + bm - 1) / bm) * (N / kBN), rounds = (tiles + slots - 1) / slots;
        return (double)(rounds * slots - tiles) / (double)(rounds * slots);
    };
    static const int force_bm = [] { const char *e = getenv("ZIRA_G3_BM"); return e ? atoi(e) : 0; }();   // developer override
    if (force_bm == 192 || (force_bm != 128 && waste(192) + ZIRA_G3_BM192_MARGIN < waste(128)))
        return launch_epi<192>(epilogue, a, bp, bias, aux, c, M, N, K, stream);
    return launch_epi<128>(epilogue, a, bp, bias, aux, c, M, N, K, stream);
}
