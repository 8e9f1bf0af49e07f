// gemm_f16x2.hip -- C = epilogue(A * B^T) in fp32 accuracy on the f16 matrix cores of gfx950 (MI355X), for the products of
// the image-token rows with FROZEN weights that are not the feed-forward block (that one is csrc/ffn_f16x2.hip): the 256-wide
// projections of the deformable attention (reference models/GroundingDINO/ms_deform_attn.py:262-288, :338) and the backbone's
// linears, under the freeze of groundingdino_dual_zero_rep_branch.py:722-745.
//
// The arithmetic of csrc/ffn_f16x2.hip with the tiling of csrc/gemm_bf16x3.hip.  An fp32 number scaled by a power of two s so
// that its group's largest magnitude lies in [2^14, 2^15) is a1 + a2 + rest with a1 = f16(s a), a2 = f16(s a - a1), |rest| <=
// 2^-22 |s a|; THREE matrix-core terms a1 b1 + a1 b2 + a2 b1 per fragment pair instead of the six of the three-plane bfloat16
// split.  The group of an activation is (its row, the 32-deep K step): the eight lanes that load a row's 32 columns of a step
// find the maximum with three lane shuffles, and because the matrix core sums a 32-deep slice FROM ZERO and the vector unit adds
// the slice to the running sum (the lesson of gemm_bf16x3.hip), the slice's scale is undone exactly in that v_fma.  The frozen
// weight is split ONCE into two f16 planes [2][N][K] with one scale per row, undone in the epilogue.
// Against an fp64 product the result is closer than the library's fp32 GEMM (tests/test_gemm_f16x2_gpu.py).
//
// Kernel: a block = BM x 128 tile of C (BM = 128 or 192), 256 threads = 2 x 2 waves of (BM / 2) x 64; K in steps of 32 through
// ONE LDS stage per operand and plane (rows padded to 80 bytes), the next step's global loads in flight during the MFMAs, two
// blocks per CU.  The matrix core computes C^T tiles (its A operand is the weight fragment): a lane holds four consecutive
// columns of one row of C per four accumulator registers.  Epilogues: + bias, + bias and ReLU, mask by (aux > 0), + aux, + bias
// and exact GELU (the backbone's MLP, reference models/GroundingDINO/backbone/swin_transformer.py:40-62), and
// aux + row_scale[m / rows_per_scale] * (product + bias): a Swin block's residual with its stochastic-depth factor per image
// (swin_transformer.py:237-262).  N is a multiple of 32: the last column tile may be partly outside (its weight rows repeat
// the last one, nothing of theirs is stored).  K is a multiple of 4: the planes' rows are padded with zeros to whole 32-deep steps
// and the activation's missing columns are zeros (776 = 4 heads x 194 text tokens on the contraction side of the fusion block).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "zira_msda.h"

#ifndef ZIRA_G2_BM192_MARGIN
#define ZIRA_G2_BM192_MARGIN 0.05   // the 192-row tile is taken where it wastes this much less of the last round of block slots
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBN = 128, kBK = 32, kThreads = 256;
constexpr int kRow = 80;   // bytes of an LDS row: 32 f16 + 16 bytes of padding

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_ADD = 3, EPI_BIAS_GELU = 4, EPI_BIAS_RES = 5 };

__device__ __forceinline__ unsigned pk_f16(float a, float b)
{
    f32x2 x = {a, b};
    f16x2 h = __builtin_convertvector(x, f16x2);   // round to nearest even
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

// the power of two that brings amax into [2^14, 2^15), and its reciprocal (exact); amax = 0 or tiny: 2^100
__device__ __forceinline__ void pow2_scale(float amax, float &s, float &inv)
{
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    int se = 127 + 14 - (e - 127);
    se = se > 227 ? 227 : (se < 1 ? 1 : se);
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}

// four fp32 numbers (already scaled) -> their two f16 planes, four halves (8 bytes) each
__device__ __forceinline__ void split4(const float4 v, uint2 &p1, uint2 &p2)
{
    p1.x = pk_f16(v.x, v.y);
    p1.y = pk_f16(v.z, v.w);
    p2.x = pk_f16(v.x - f16_lo(p1.x), v.y - f16_hi(p1.x));   // (exact differences)
    p2.y = pk_f16(v.z - f16_lo(p1.y), v.w - f16_hi(p1.y));
}

// The epilogue: accumulator register 4 g + i of block (ni, mi) is C[m][n], m = row (lane & 31) of the block, n = 8 g + 4 (lane >> 5) + i
template <int MI, int NI, int EPI>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[NI][MI], const float *__restrict__ winv, const float *__restrict__ bias,
                                           const float *aux, const float *__restrict__ rscale, int rows_per_scale, float *C, int M, int N,
                                           int mbase, int nbase, int lane)
{
    const int lm = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = mbase + mi * 32 + lm;
        if (m >= M) continue;
        float rs = 1.f;
        if (EPI == EPI_BIAS_RES && rscale) rs = rscale[m / rows_per_scale];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            if (nbase + ni * 32 >= N) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + ni * 32 + 8 * g + 4 * lh;
                const f32x16 &c = acc[ni][mi];
                const float4 wi = *reinterpret_cast<const float4 *>(winv + n);
                float4 o = make_float4(c[4 * g] * wi.x, c[4 * g + 1] * wi.y, c[4 * g + 2] * wi.z, c[4 * g + 3] * wi.w);
                const size_t at = (size_t)m * N + n;
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + n);
                    o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                    if (EPI == EPI_BIAS_RELU) {
                        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                    }
                    if (EPI == EPI_BIAS_GELU) {   // 0.5 x (1 + erf(x / sqrt 2)), the arithmetic of ATen's GELU kernel
                        o.x = 0.5f * o.x * (1.f + erff(o.x * 0.70710678118654752440f));
                        o.y = 0.5f * o.y * (1.f + erff(o.y * 0.70710678118654752440f));
                        o.z = 0.5f * o.z * (1.f + erff(o.z * 0.70710678118654752440f));
                        o.w = 0.5f * o.w * (1.f + erff(o.w * 0.70710678118654752440f));
                    }
                    if (EPI == EPI_BIAS_RES) {
                        const float4 h = *reinterpret_cast<const float4 *>(aux + at);
                        o.x = fmaf(o.x, rs, h.x); o.y = fmaf(o.y, rs, h.y); o.z = fmaf(o.z, rs, h.z); o.w = fmaf(o.w, rs, h.w);
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
}

template <int BM, int EPI>
__global__ __launch_bounds__(kThreads, 2) void gemm_f16x2_kernel(const float *__restrict__ A, const unsigned short *__restrict__ Bp,
                                                                const float *__restrict__ winv, const float *__restrict__ bias,
                                                                const float *aux, const float *__restrict__ rscale, int rows_per_scale,
                                                                float *C, int M, int N, int K, int row_tiles, int col_tiles, int rt_per_xcd)
{
    constexpr int WM = BM / 2, MI = WM / 32, NI = 2;   // a wave: WM x 64 of C = MI x NI blocks of 32 x 32
    constexpr int AJ = BM / 32;                        // float4 loads of A per thread and K step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sA = smem;                          // [2][BM][kRow]
    unsigned char *sB = smem + 2 * BM * kRow;          // [2][kBN][kRow]
    float *sS = reinterpret_cast<float *>(smem + 2 * (BM + kBN) * kRow);   // [BM]: 1 / scale of each row's current slice

    // tile of this block: blocks b, b + 8, ... share an XCD (placement is for speed only); an XCD walks its own range of row
    // tiles, the column tiles of a row tile side by side (they share the rows of A in its L2)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int trow = xcd * rt_per_xcd + idx / col_tiles, tcol = idx % col_tiles;
    if (trow >= row_tiles || idx / col_tiles >= rt_per_xcd) return;
    const int m0 = trow * BM, n0 = tcol * kBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // global -> register staging
    const int a_chunk = tid & 7, a_row = tid >> 3;     // row a_row + 32 j, floats 4 a_chunk .. + 3 of the K step
    const int b_chunk = tid & 3, b_row = tid >> 2;     // row b_row + 64 j, halves 8 b_chunk .. + 7
    const float *ag[AJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int r = m0 + a_row + 32 * j;
        r = r < M ? r : M - 1;                         // (rows past the end: any finite data, their results are not stored)
        ag[j] = A + (size_t)r * K + a_chunk * 4;
    }
    // (weight rows past N -- the last column tile of an N that is not a multiple of 128 -- repeat row N - 1)
    const int Kp = (K + kBK - 1) / kBK * kBK;          // the planes' row length: K in whole steps, zeros behind K
    const unsigned short *bg = Bp + (size_t)(n0 + b_row < N ? n0 + b_row : N - 1) * Kp + b_chunk * 8;
    const size_t bplane = (size_t)N * Kp;
    const size_t bj = (size_t)((n0 + b_row + 64 < N ? n0 + b_row + 64 : N - 1) - (n0 + b_row < N ? n0 + b_row : N - 1)) * Kp;

    float4 ra[AJ];
    uint4 rb00, rb01, rb10, rb11;
#define ZIRA_GLOAD(k0_)                                                                             \
    do {                                                                                            \
        _Pragma("unroll") for (int j = 0; j < AJ; ++j)                                               \
            ra[j] = (a_chunk * 4 + (k0_) < K) ? *reinterpret_cast<const float4 *>(ag[j] + (k0_)) : make_float4(0.f, 0.f, 0.f, 0.f); \
        rb00 = *reinterpret_cast<const uint4 *>(bg + (k0_));                                        \
        rb01 = *reinterpret_cast<const uint4 *>(bg + bj + (k0_));                                   \
        rb10 = *reinterpret_cast<const uint4 *>(bg + bplane + (k0_));                               \
        rb11 = *reinterpret_cast<const uint4 *>(bg + bplane + bj + (k0_));                          \
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
    const float *fs = sS + wm * WM + (lane & 31);

    ZIRA_GLOAD(0);
    for (int k0 = 0; k0 < Kp; k0 += kBK) {
        __syncthreads();   // the previous step's fragment reads are done
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            // the row's 32 columns of this step sit in the 8 lanes tid & ~7 .. + 7: their largest magnitude, the scale, the planes
            float4 v = ra[j];
            float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            amax = fmaxf(amax, __shfl_xor(amax, 1));
            amax = fmaxf(amax, __shfl_xor(amax, 2));
            amax = fmaxf(amax, __shfl_xor(amax, 4));
            float s, inv;
            pow2_scale(amax, s, inv);
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            uint2 p1, p2;
            split4(v, p1, p2);
            unsigned char *d = wa + 32 * j * kRow;
            *reinterpret_cast<uint2 *>(d) = p1;
            *reinterpret_cast<uint2 *>(d + BM * kRow) = p2;
            if (a_chunk == 0) sS[a_row + 32 * j] = inv;
        }
        *reinterpret_cast<uint4 *>(wb) = rb00;
        *reinterpret_cast<uint4 *>(wb + 64 * kRow) = rb01;
        *reinterpret_cast<uint4 *>(wb + kBN * kRow) = rb10;
        *reinterpret_cast<uint4 *>(wb + kBN * kRow + 64 * kRow) = rb11;
        __syncthreads();
        if (k0 + kBK < Kp) ZIRA_GLOAD(k0 + kBK);
        // Three terms per 16-deep slice, the small ones first; matrix-core A operand = weight fragment (rows n), B operand =
        // activation fragment (rows m): the accumulator block is C^T [n][m].  The six terms of a K step are summed inside the
        // matrix core FROM ZERO and the step's sum, times 1 / (the row's scale in this step), is added to the running sum by
        // the vector unit (round to nearest).
        {
            f16x8 b[2][NI][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        b[ks][ni][p] = *reinterpret_cast<const f16x8 *>(fb + p * kBN * kRow + ni * 32 * kRow + ks * 32);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                f16x8 a[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        a[ks][p] = *reinterpret_cast<const f16x8 *>(fa + p * BM * kRow + mi * 32 * kRow + ks * 32);
                const float inv = fs[mi * 32];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks][ni][1], a[ks][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks][ni][0], a[ks][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ks][ni][0], a[ks][0], c, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[ni][mi][i] = fmaf(c[i], inv, acc[ni][mi][i]);
                }
            }
        }
    }

#undef ZIRA_GLOAD
    store_tile<MI, NI, EPI>(acc, winv, bias, aux, rscale, rows_per_scale, C, M, N, m0 + wm * WM, n0 + wn * 64, lane);
}

// W [rows][cols] fp32 -> planes [2][N][Kp] f16 (Kp = K rounded up to 32, zeros behind K) of W[n][k] * scale[n] and 1 / scale [N], with B[n][k] = W[n][k] (transpose = 0:
// N = rows, K = cols) or W[k][n] (transpose = 1: N = cols, K = rows); one block per row n
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float *__restrict__ w, int rows, int cols, int transpose,
                                                          unsigned short *__restrict__ planes, float *__restrict__ winv)
{
    __shared__ float red[256];
    const int N = transpose ? cols : rows, K = transpose ? rows : cols;
    const int n = blockIdx.x;
    auto at = [&](int k) { return transpose ? w[(size_t)k * cols + n] : w[(size_t)n * cols + k]; };
    float amax = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) amax = fmaxf(amax, fabsf(at(k)));
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    float s, inv;
    pow2_scale(red[0], s, inv);
    if (threadIdx.x == 0) winv[n] = inv;
    const int Kp = (K + kBK - 1) / kBK * kBK;              // rows padded with zeros to whole K steps
    const size_t total = (size_t)N * Kp;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        const float v = k < K ? at(k) * s : 0.f;
        const unsigned p1 = pk_f16(v, 0.f);
        const unsigned p2 = pk_f16(v - f16_lo(p1), 0.f);
        planes[(size_t)n * Kp + k] = (unsigned short)(p1 & 0xFFFFu);
        planes[total + (size_t)n * Kp + k] = (unsigned short)(p2 & 0xFFFFu);
    }
}

template <int BM, int EPI>
int launch(const float *a, const unsigned short *bp, const float *winv, const float *bias, const float *aux, const float *rscale, int rps,
           float *c, int M, int N, int K, hipStream_t st)
{
    const int rt = (M + BM - 1) / BM, ct = (N + kBN - 1) / kBN, per = (rt + 7) / 8;
    static bool attr_set = false;
    const size_t lds = (size_t)2 * (BM + kBN) * kRow + (size_t)BM * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16x2_kernel<BM, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16x2_kernel<BM, EPI>), dim3(8 * per * ct), dim3(kThreads), lds, st, a, bp, winv, bias, aux, rscale, rps, c, M, N, K, rt, ct,
                       per);
    return (int)hipGetLastError();
}

template <int BM>
int launch_epi(int epi, const float *a, const unsigned short *bp, const float *winv, const float *bias, const float *aux, const float *rscale,
               int rps, float *c, int M, int N, int K, hipStream_t st)
{
    switch (epi) {
    case EPI_BIAS: return launch<BM, EPI_BIAS>(a, bp, winv, bias, aux, rscale, rps, c, M, N, K, st);
    case EPI_BIAS_RELU: return launch<BM, EPI_BIAS_RELU>(a, bp, winv, bias, aux, rscale, rps, c, M, N, K, st);
    case EPI_MASK: return launch<BM, EPI_MASK>(a, bp, winv, bias, aux, rscale, rps, c, M, N, K, st);
    case EPI_ADD: return launch<BM, EPI_ADD>(a, bp, winv, bias, aux, rscale, rps, c, M, N, K, st);
    case EPI_BIAS_GELU: return launch<BM, EPI_BIAS_GELU>(a, bp, winv, bias, aux, rscale, rps, c, M, N, K, st);
    case EPI_BIAS_RES: return launch<BM, EPI_BIAS_RES>(a, bp, winv, bias, aux, rscale, rps, c, M, N, K, st);
    }
    return -1;
}

}  // namespace

// planes: 2 * N * K halves, then N floats (1 / scale of every row): 4 N K + 4 N bytes, 16-byte aligned
extern "C" int zira_split_f16x2_f32(const float *w, int rows, int cols, int transpose, void *planes, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!w || !planes || rows <= 0 || cols <= 0) return -1;
    const int N = transpose ? cols : rows, K = transpose ? rows : cols;
    unsigned short *p = reinterpret_cast<unsigned short *>(planes);
    const int Kp = (K + kBK - 1) / kBK * kBK;
    float *winv = reinterpret_cast<float *>(p + (size_t)2 * N * Kp);
    hipLaunchKernelGGL(split_f16x2_kernel, dim3(N), dim3(256), 0, stream, w, rows, cols, transpose ? 1 : 0, p, winv);
    return (int)hipGetLastError();
}

extern "C" int zira_gemm_f16x2_ex_f32(const float *a, const void *b_planes, int M, int N, int K, int epilogue, const float *bias,
                                      const float *aux, const float *row_scale, int rows_per_scale, float *c, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!a || !b_planes || !c || M <= 0 || N <= 0 || K <= 0 || N % 32 || K % 4) return -1;
    const bool needs_bias = epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU || epilogue == EPI_BIAS_GELU || epilogue == EPI_BIAS_RES;
    const bool needs_aux = epilogue == EPI_MASK || epilogue == EPI_ADD || epilogue == EPI_BIAS_RES;
    if ((needs_bias && !bias) || (needs_aux && !aux)) return -1;
    if (row_scale && (epilogue != EPI_BIAS_RES || rows_per_scale <= 0)) return -1;
    if (((uintptr_t)a | (uintptr_t)b_planes | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)aux) & 15) return -1;
    if ((unsigned long long)M * N >= (1ull << 40)) return -1;
    const unsigned short *bp = reinterpret_cast<const unsigned short *>(b_planes);
    const float *winv = reinterpret_cast<const float *>(bp + (size_t)2 * N * ((K + kBK - 1) / kBK * kBK));
    // tile height: the one that wastes fewer of the chip's 512 block slots in its last round
    auto waste = [&](int bm) {
        const long long slots = 512;
        const long long tiles = (long long)((M + bm - 1) / bm) * ((N + kBN - 1) / kBN), rounds = (tiles + slots - 1) / slots;
        return (double)(rounds * slots - tiles) / (double)(rounds * slots);
    };
    static const int force_bm = [] { const char *e = getenv("ZIRA_G2_BM"); return e ? atoi(e) : 0; }();   // developer override
    if (force_bm == 192 || (force_bm != 128 && waste(192) + ZIRA_G2_BM192_MARGIN < waste(128)))
        return launch_epi<192>(epilogue, a, bp, winv, bias, aux, row_scale, rows_per_scale, c, M, N, K, stream);
    return launch_epi<128>(epilogue, a, bp, winv, bias, aux, row_scale, rows_per_scale, c, M, N, K, stream);
}

extern "C" int zira_gemm_f16x2_f32(const float *a, const void *b_planes, int M, int N, int K, int epilogue, const float *bias,
                                   const float *aux, float *c, void *stream_)
{
    if (epilogue < EPI_BIAS || epilogue > EPI_ADD) return -1;
    return zira_gemm_f16x2_ex_f32(a, b_planes, M, N, K, epilogue, bias, aux, nullptr, 0, c, stream_);
}
