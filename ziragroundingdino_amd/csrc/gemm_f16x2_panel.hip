// gemm_f16x2_panel.hip -- C = epilogue((A [+ A2]) * B^T) in fp32 accuracy on the f16 matrix cores of gfx950 (MI355X) for the
// SKINNY frozen products of the image-token rows: K = 256 or 384, N a few hundred -- the value / query / output projections of
// the deformable attention and their input gradients (reference models/GroundingDINO/ms_deform_attn.py:262-288, :338 under the
// freeze of groundingdino_dual_zero_rep_branch.py:722-745; `A2` = the position code the reference adds to the query,
// transformer_for_adapter.py:893-900).  Same arithmetic as csrc/gemm_f16x2.hip (two f16 planes per operand, three exact terms,
// fp32 sums), another shape of work.
//
// These products read and write 45 MB tensors for a few GFLOP: they are bound by memory, and the tiled kernel
// (csrc/gemm_f16x2.hip: 128 x 128 tiles, a barrier pair and a dependent global load per 32-deep K step, 1.36 rounds of the chip's
// block slots) takes 43 us where the bytes take 18.  Here a block takes 32 ROWS AND ALL OF K: its 32 KB panel of A is requested
// at once, scaled per row (one scale per row over the whole K), split into the two planes and laid into LDS in fragment
// order ONCE; after one barrier every wave walks its share of the N / 32 column tiles with the weight fragments (packed
// fragment-major when the frozen weight was split: 1 KB per wave load, L2-resident) straight from global memory -- no K loop,
// no further barrier.  1389 small blocks, four or five per CU, keep the memory system busy.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRows = 32, kThreads = 256;

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_ADD = 3 };

__device__ __forceinline__ unsigned pk_f16(float a, float b)
{
    f32x2 x = {a, b};
    f16x2 h = __builtin_convertvector(x, f16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

__device__ __forceinline__ void pow2_scale(float amax, float &s, float &inv)
{
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    int se = 127 + 14 - (e - 127);
    se = se > 227 ? 227 : (se < 1 ? 1 : se);
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}

// KS = K / 16 matrix-core steps.  LDS: planes [2][KS][2 halves of a step][32 rows][8 halves] = 2 KS KB, then 32 floats (1 / scale).
template <int KS, int EPI, bool ADD2>
__global__ __launch_bounds__(kThreads, 4) void gemm_f16x2_panel_kernel(const float *__restrict__ A, const float *__restrict__ A2,
                                                                    const unsigned char *__restrict__ Wf, const float *__restrict__ winv,
                                                                    const float *__restrict__ bias, const float *aux, float *C, int M,
                                                                    int N)
{
    constexpr int K = 16 * KS, J = K / 32;                 // J float4 per thread: thread (row, c) takes columns 4 c + 32 j
    __shared__ __attribute__((aligned(16))) unsigned char planes[2 * KS * 1024];
    __shared__ float sinv[kRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * kRows;

    // ---- the panel: all of it requested at once ---------------------------------------------------------------------------------
    {
        const int row = tid >> 3, c = tid & 7;
        int m = m0 + row;
        m = m < M ? m : M - 1;                             // (rows past the end repeat the last row; nothing of theirs is stored)
        const float *ar = A + (size_t)m * K + 4 * c;
        float4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) v[j] = *reinterpret_cast<const float4 *>(ar + 32 * j);
        if (ADD2) {
            const float *br = A2 + (size_t)m * K + 4 * c;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const float4 b = *reinterpret_cast<const float4 *>(br + 32 * j);
                v[j].x += b.x; v[j].y += b.y; v[j].z += b.z; v[j].w += b.w;
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        float s, inv;
        pow2_scale(amax, s, inv);
        if (c == 0) sinv[row] = inv;
        // columns 4 c + 32 j .. + 3: step st = 2 j + (c >> 2), half hf = (c >> 1) & 1, halves 4 (c & 1) .. + 3 of the lane's eight
        unsigned char *dst = planes + ((c >> 2) * 2 + ((c >> 1) & 1)) * 512 + row * 16 + (c & 1) * 8;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float x = v[j].x * s, y = v[j].y * s, z = v[j].z * s, w = v[j].w * s;
            uint2 p1, p2;
            p1.x = pk_f16(x, y);
            p1.y = pk_f16(z, w);
            p2.x = pk_f16(x - f16_lo(p1.x), y - f16_hi(p1.x));
            p2.y = pk_f16(z - f16_lo(p1.y), w - f16_hi(p1.y));
            *reinterpret_cast<uint2 *>(dst + (2 * j) * 1024) = p1;
            *reinterpret_cast<uint2 *>(dst + (2 * j) * 1024 + KS * 1024) = p2;
        }
    }
    __syncthreads();

    // ---- column tiles: wave w takes tiles w, w + 4, ...; accumulator register 4 g + i is C[m][32 t + 8 g + 4 hf + i] -------------
    const int lm = lane & 31, hf = lane >> 5;
    const int m = m0 + lm;
    const float inv = sinv[lm];
    const unsigned char *pa = planes + hf * 512 + lm * 16;
    const int ntiles = N >> 5;
    for (int t = wave; t < ntiles; t += 4) {
        const unsigned char *wt = Wf + ((size_t)t * KS * 2) * 1024 + lane * 16;   // fragment (t, st, plane) at ((t KS + st) 2 + plane) KB
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        constexpr int AHEAD = 4, LA = 2;     // weight fragments (global, L2) four steps ahead, panel fragments (LDS) two
        f16x8 w1[AHEAD], w2[AHEAD], a1[LA], a2[LA];
#pragma unroll
        for (int st = 0; st < AHEAD; ++st) {
            w1[st] = *reinterpret_cast<const f16x8 *>(wt + (st * 2 + 0) * 1024);
            w2[st] = *reinterpret_cast<const f16x8 *>(wt + (st * 2 + 1) * 1024);
        }
#pragma unroll
        for (int st = 0; st < LA; ++st) {
            a1[st] = *reinterpret_cast<const f16x8 *>(pa + st * 1024);
            a2[st] = *reinterpret_cast<const f16x8 *>(pa + st * 1024 + KS * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const f16x8 u1 = w1[st % AHEAD], u2 = w2[st % AHEAD], b1 = a1[st % LA], b2 = a2[st % LA];
            if (st + AHEAD < KS) {
                w1[st % AHEAD] = *reinterpret_cast<const f16x8 *>(wt + ((st + AHEAD) * 2 + 0) * 1024);
                w2[st % AHEAD] = *reinterpret_cast<const f16x8 *>(wt + ((st + AHEAD) * 2 + 1) * 1024);
            }
            if (st + LA < KS) {
                a1[st % LA] = *reinterpret_cast<const f16x8 *>(pa + (st + LA) * 1024);
                a2[st % LA] = *reinterpret_cast<const f16x8 *>(pa + (st + LA) * 1024 + KS * 1024);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(u2, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(u1, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(u1, b1, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (m < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * t + 8 * g + 4 * hf;
                const float4 wi = *reinterpret_cast<const float4 *>(winv + n);
                float4 o = make_float4(acc[4 * g] * inv * wi.x, acc[4 * g + 1] * inv * wi.y, acc[4 * g + 2] * inv * wi.z, acc[4 * g + 3] * inv * wi.w);
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
}

// W [rows][cols] fp32 -> fragments [N / 32][K / 16][2 planes][64 lanes][8 halves] of W[n][k] * scale[n], then 1 / scale [N];
// B[n][k] = W[n][k] (transpose = 0) or W[k][n] (transpose = 1).  One block per row n.
__global__ __launch_bounds__(256) void split_f16x2_frag_kernel(const float *__restrict__ w, int rows, int cols, int transpose,
                                                               unsigned short *__restrict__ frags, float *__restrict__ winv)
{
    __shared__ float red[256];
    const int K = transpose ? rows : cols;
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
    const int KS = K / 16, t = n >> 5, lm = n & 31;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = at(k) * s;
        const unsigned p1 = pk_f16(v, 0.f);
        const unsigned p2 = pk_f16(v - f16_lo(p1), 0.f);
        const int st = k >> 4, hf = (k >> 3) & 1, e = k & 7;
        const size_t base = (((size_t)t * KS + st) * 2) * 512 + (size_t)(hf * 32 + lm) * 8 + e;   // in halves; plane 1 is 512 halves on
        frags[base] = (unsigned short)(p1 & 0xFFFFu);
        frags[base + 512] = (unsigned short)(p2 & 0xFFFFu);
    }
}

template <int KS, bool ADD2>
int launch_epi(int epi, const float *a, const float *a2, const unsigned char *wf, const float *winv, const float *bias, const float *aux,
               float *c, int M, int N, hipStream_t st)
{
    const dim3 grid((M + kRows - 1) / kRows), block(kThreads);
    switch (epi) {
    case EPI_BIAS: hipLaunchKernelGGL((gemm_f16x2_panel_kernel<KS, EPI_BIAS, ADD2>), grid, block, 0, st, a, a2, wf, winv, bias, aux, c, M, N); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_f16x2_panel_kernel<KS, EPI_BIAS_RELU, ADD2>), grid, block, 0, st, a, a2, wf, winv, bias, aux, c, M, N); break;
    case EPI_MASK: hipLaunchKernelGGL((gemm_f16x2_panel_kernel<KS, EPI_MASK, ADD2>), grid, block, 0, st, a, a2, wf, winv, bias, aux, c, M, N); break;
    case EPI_ADD: hipLaunchKernelGGL((gemm_f16x2_panel_kernel<KS, EPI_ADD, ADD2>), grid, block, 0, st, a, a2, wf, winv, bias, aux, c, M, N); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}

}  // namespace

// frags: 2 * N * K halves in fragment order, then N floats: 4 N K + 4 N bytes, 16-byte aligned; N % 32 == 0, K % 16 == 0
extern "C" int zira_split_f16x2_frag_f32(const float *w, int rows, int cols, int transpose, void *frags, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!w || !frags || rows <= 0 || cols <= 0) return -1;
    const int N = transpose ? cols : rows, K = transpose ? rows : cols;
    if (N % 32 || K % 16) return -1;
    unsigned short *p = reinterpret_cast<unsigned short *>(frags);
    float *winv = reinterpret_cast<float *>(p + (size_t)2 * N * K);
    hipLaunchKernelGGL(split_f16x2_frag_kernel, dim3(N), dim3(256), 0, stream, w, rows, cols, transpose ? 1 : 0, p, winv);
    return (int)hipGetLastError();
}

extern "C" int zira_gemm_f16x2_panel_f32(const float *a, const float *a2, const void *b_frags, int M, int N, int K, int epilogue,
                                         const float *bias, const float *aux, float *c, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!a || !b_frags || !c || M <= 0 || N <= 0 || N % 32 || (K != 256 && K != 384)) return -1;
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) ? !bias : !aux) return -1;
    if (((uintptr_t)a | (uintptr_t)a2 | (uintptr_t)b_frags | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)aux) & 15) return -1;
    const unsigned char *wf = reinterpret_cast<const unsigned char *>(b_frags);
    const float *winv = reinterpret_cast<const float *>(wf + (size_t)4 * N * K);
    if (K == 256)
        return a2 ? launch_epi<16, true>(epilogue, a, a2, wf, winv, bias, aux, c, M, N, stream)
                  : launch_epi<16, false>(epilogue, a, a2, wf, winv, bias, aux, c, M, N, stream);
    return a2 ? launch_epi<24, true>(epilogue, a, a2, wf, winv, bias, aux, c, M, N, stream)
              : launch_epi<24, false>(epilogue, a, a2, wf, winv, bias, aux, c, M, N, stream);
}
