// gemm_f16x2_rows.hip -- NOT BUILT INTO THE LIBRARY (round-6 experiment, measured and set aside; exports zira_gemm_f16x2_rows_f32
// with the arguments of zira_gemm_f16x2_panel_f32).  At M = 44446, K = 256: N = 256 35.9 us, N = 384 47.5 us against the panel kernel's
// 33.4 / 43.1.  Ablations (ZIRA_ROWS_ABLATE; N = 256 / 384): no matrix work 32.0 / 40.0, no stores 28.9 / 38.1, no loads of A 25.1 / 36.8,
// no copies of the weight 32.3 / 44.1, neither matrix work nor stores 24.0 / 27.2, neither loads of A nor stores 19.8 / 28.1 us: the
// weight copies + matrix work + barriers of eight tiles alone take 20 us because 348 blocks of 128 rows land 2 : 1 on the 256 CUs
// (a CU with two blocks shares its matrix pipes), and loads, matrix work and stores of the two resident blocks run in lockstep
// instead of overlapping.  The L2 read requests of the weight, which this layout cuts fourfold, were not what bounds the panel kernel.
//
// C = epilogue((A [+ A2]) * B^T) in fp32 accuracy on the f16 matrix cores of gfx950 (MI355X) for the skinny
// frozen products with K = 256 of the image-token rows: the value / query / output projections of the deformable attention and
// their input gradients (reference models/GroundingDINO/ms_deform_attn.py:262-288, :338 under the freeze of
// groundingdino_dual_zero_rep_branch.py:722-745).  Same arithmetic and the same fragment-major weights as
// csrc/gemm_f16x2_panel.hip (two f16 planes per operand, three exact terms per 16-deep step, fp32 sums), another division of
// the work.
//
// The panel kernel gives a block 32 rows and lets every wave fetch the weight fragments of its column tiles from L2: each 32
// rows read ALL of the weight (256 KB at N = 256), 2.8 M L2 read requests per launch next to the 0.7 M of the activations
// themselves, and the counters show the L2 request slots 45 % busy over the launch (profiles/r06_panel_pmc.txt).  Here a WAVE
// keeps its 32 rows in REGISTERS -- both planes of 32 x 256, 128 registers, loaded, scaled per row and split once -- and the four
// waves of a block (128 rows) walk the column tiles together: a tile's fragments (32 KB, contiguous in the packed weight) are
// copied global -> LDS once per block by the DMA path (no registers), two stages deep, and read by all four waves.  L2 reads of
// the weight drop fourfold; a tile costs one barrier.  Stores of tile t are issued after the barrier that ends it, so the wait
// for the next tile's copy (s_waitcnt vmcnt(0): loads and stores share the counter on gfx9) never waits for fresh stores.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

#ifndef ZIRA_ROWS_ABLATE
#define ZIRA_ROWS_ABLATE 0   // developer: 1 no matrix work, 2 no stores, 4 no loads of A, 8 no copies of the weight (results are wrong)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kThreads = 256, kKS = 16, kK = 256;
constexpr int kTileBytes = kKS * 2 * 1024;   // one column tile's fragments: [step][plane][64 lanes][8 halves]

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_ADD = 3 };

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

// 1 KB global -> LDS without registers (see csrc/ffn_f16x2.hip on why this is inline assembly)
__device__ __forceinline__ void dma_1k(const unsigned char *src_uniform, unsigned lane_off, unsigned dst_lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_off), "s"(dst_lds), "s"(src_uniform)
                 : "memory");
}

// a column tile's 32 KB: wave w copies kilobytes 8 w .. 8 w + 7
__device__ __forceinline__ void copy_tile(const unsigned char *src, unsigned dst, int wave, int lane)
{
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (!(ZIRA_ROWS_ABLATE & 8)) dma_1k(src + (wave * 8 + q) * 1024, voff, dst + (wave * 8 + q) * 1024);
}

template <int EPI, bool ADD2>
__global__ __launch_bounds__(kThreads, 2) void gemm_f16x2_rows_kernel(const float *__restrict__ A, const float *__restrict__ A2,
                                                                   const unsigned char *__restrict__ Wf, const float *__restrict__ winv,
                                                                   const float *__restrict__ bias, const float *aux, float *C, int M,
                                                                   int N)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two stages of kTileBytes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 31, hf = lane >> 5;
    const int m = blockIdx.x * kBM + wave * 32 + lm;
    const int mc = m < M ? m : M - 1;                      // (rows past the end repeat the last row; nothing of theirs is stored)
    const int ntiles = N >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    copy_tile(Wf, lds0, wave, lane);                       // tile 0 -> stage 0, under the panel's loads

    // ---- the wave's 32 rows: lane (lm, hf) holds columns 16 st + 8 hf .. + 7 of row lm for every step st -- the matrix core's
    // B-operand fragment as it stands.  Scale = the row's power of two (both lanes of a row agree through one exchange).
    f16x8 a1[kKS], a2[kKS];
    float inv;
    {
        const float *ar = A + (size_t)mc * kK + 8 * hf;
        float4 raw[2 * kKS];
#pragma unroll
        for (int st = 0; st < kKS; ++st) {
            if (ZIRA_ROWS_ABLATE & 4) {
                raw[2 * st] = raw[2 * st + 1] = make_float4(1.f + st + lm, 2.f, 3.f, 4.f + hf);
                continue;
            }
            raw[2 * st] = *reinterpret_cast<const float4 *>(ar + 16 * st);
            raw[2 * st + 1] = *reinterpret_cast<const float4 *>(ar + 16 * st + 4);
        }
        if (ADD2) {
            const float *br = A2 + (size_t)mc * kK + 8 * hf;
#pragma unroll
            for (int st = 0; st < 2 * kKS; ++st) {
                const float4 b = *reinterpret_cast<const float4 *>(br + 16 * (st >> 1) + 4 * (st & 1));
                raw[st].x += b.x; raw[st].y += b.y; raw[st].z += b.z; raw[st].w += b.w;
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * kKS; ++i)
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(raw[i].x), fabsf(raw[i].y))), fmaxf(fabsf(raw[i].z), fabsf(raw[i].w)));
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        float s;
        pow2_scale(amax, s, inv);
#pragma unroll
        for (int st = 0; st < kKS; ++st) {
            unsigned p1[4], p2[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 v = raw[2 * st + h];
                const float x = v.x * s, y = v.y * s, z = v.z * s, w = v.w * s;
                p1[2 * h] = pk_f16(x, y);
                p1[2 * h + 1] = pk_f16(z, w);
                p2[2 * h] = pk_f16(x - f16_lo(p1[2 * h]), y - f16_hi(p1[2 * h]));           // (exact differences)
                p2[2 * h + 1] = pk_f16(z - f16_lo(p1[2 * h + 1]), w - f16_hi(p1[2 * h + 1]));
            }
            const uint4 q1 = make_uint4(p1[0], p1[1], p1[2], p1[3]), q2 = make_uint4(p2[0], p2[1], p2[2], p2[3]);
            a1[st] = __builtin_bit_cast(f16x8, q1);
            a2[st] = __builtin_bit_cast(f16x8, q2);
        }
    }
    // the columns' 1 / scale and bias next to the stages (the epilogue reads them there: no global load in front of a tile's stores)
    float *sw = reinterpret_cast<float *>(smem + 2 * kTileBytes), *sb = sw + N;
    for (int n = threadIdx.x; n < N; n += kThreads) {
        sw[n] = winv[n];
        sb[n] = (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) ? bias[n] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile 0 has landed
    __syncthreads();

    // ---- column tiles: accumulator register 4 g + i of lane (lm, hf) is C[m][32 t + 8 g + 4 hf + i] ------------------------------
    // Order inside a tile: request the NEXT tile's copy; issue the stores of the PREVIOUS tile (its aux rows arrived during its
    // own matrix work); request this tile's aux rows; matrix work; wait for everything requested; barrier.
    constexpr bool kAux = EPI == EPI_MASK || EPI == EPI_ADD;
    f32x16 prev;                                           // the finished tile whose stores are still to be issued
    float4 hprev[4], hcur[4];                              // its rows of aux, and those of the tile in the matrix core
    auto fetch_aux = [&](int t, float4 (&h)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            h[g] = (kAux && m < M) ? *reinterpret_cast<const float4 *>(aux + (size_t)m * N + 32 * t + 8 * g + 4 * hf) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_tile = [&](const f32x16 &acc, const float4 (&h)[4], int t) {
        if (m >= M) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * t + 8 * g + 4 * hf;
            const float4 wi = *reinterpret_cast<const float4 *>(sw + n);
            float4 o = make_float4(acc[4 * g] * inv * wi.x, acc[4 * g + 1] * inv * wi.y, acc[4 * g + 2] * inv * wi.z, acc[4 * g + 3] * inv * wi.w);
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) {
                const float4 bv = *reinterpret_cast<const float4 *>(sb + n);
                o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                if (EPI == EPI_BIAS_RELU) {
                    o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
                }
            } else if (EPI == EPI_MASK) {
                o.x = h[g].x > 0.f ? o.x : 0.f; o.y = h[g].y > 0.f ? o.y : 0.f;
                o.z = h[g].z > 0.f ? o.z : 0.f; o.w = h[g].w > 0.f ? o.w : 0.f;
            } else {
                o.x += h[g].x; o.y += h[g].y; o.z += h[g].z; o.w += h[g].w;
            }
            if (!(ZIRA_ROWS_ABLATE & 2) || o.x == 12345.678f) *reinterpret_cast<float4 *>(C + (size_t)m * N + n) = o;
        }
    };
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char *stage = smem + (t & 1) * kTileBytes + lane * 16;
        if (t + 1 < ntiles) copy_tile(Wf + (size_t)(t + 1) * kTileBytes, lds0 + ((t + 1) & 1) * kTileBytes, wave, lane);
        if (t > 0) store_tile(prev, hprev, t - 1);
        if (kAux) fetch_aux(t, hcur);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f16x8 w1[2], w2[2];
        w1[0] = *reinterpret_cast<const f16x8 *>(stage);
        w2[0] = *reinterpret_cast<const f16x8 *>(stage + 1024);
#pragma unroll
        for (int st = 0; st < kKS; ++st) {
            if (st + 1 < kKS) {
                w1[(st + 1) & 1] = *reinterpret_cast<const f16x8 *>(stage + (st + 1) * 2048);
                w2[(st + 1) & 1] = *reinterpret_cast<const f16x8 *>(stage + (st + 1) * 2048 + 1024);
            }
            if (ZIRA_ROWS_ABLATE & 1) {
                if (st == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[0], a1[(t + wave) & 15], acc, 0, 0, 0);
                continue;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[st & 1], a1[st], acc, 0, 0, 0);   // the small terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[st & 1], a2[st], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[st & 1], a1[st], acc, 0, 0, 0);
        }
        prev = acc;
        if (kAux) {
#pragma unroll
            for (int g = 0; g < 4; ++g) hprev[g] = hcur[g];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's copy, this tile's aux rows (the previous tile's stores: long gone)
        __syncthreads();
    }
    store_tile(prev, hprev, ntiles - 1);
}

template <bool ADD2>
int launch_epi(int epi, const float *a, const float *a2, const unsigned char *wf, const float *winv, const float *bias, const float *aux,
               float *c, int M, int N, hipStream_t st)
{
    const dim3 grid((M + kBM - 1) / kBM), block(kThreads);
    const size_t lds = 2 * kTileBytes + (size_t)2 * N * sizeof(float);
#define ZIRA_ROWS_LAUNCH(E)                                                                                                      \
    do {                                                                                                                         \
        static bool attr = false;                                                                                                \
        if (!attr) {                                                                                                             \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_f16x2_rows_kernel<E, ADD2>),                    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kTileBytes + 2 * 1024 * 4);       \
            if (e != hipSuccess) return (int)e;                                                                                  \
            attr = true;                                                                                                         \
        }                                                                                                                        \
        hipLaunchKernelGGL((gemm_f16x2_rows_kernel<E, ADD2>), grid, block, lds, st, a, a2, wf, winv, bias, aux, c, M, N);          \
    } while (0)
    switch (epi) {
    case EPI_BIAS: ZIRA_ROWS_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_RELU: ZIRA_ROWS_LAUNCH(EPI_BIAS_RELU); break;
    case EPI_MASK: ZIRA_ROWS_LAUNCH(EPI_MASK); break;
    case EPI_ADD: ZIRA_ROWS_LAUNCH(EPI_ADD); break;
    default: return -1;
    }
#undef ZIRA_ROWS_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace

// b_frags as zira_split_f16x2_frag_f32 writes them; K == 256, N % 32 == 0.  Other arguments as zira_gemm_f16x2_panel_f32.
extern "C" int zira_gemm_f16x2_rows_f32(const float *a, const float *a2, const void *b_frags, int M, int N, int K, int epilogue,
                                        const float *bias, const float *aux, float *c, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!a || !b_frags || !c || M <= 0 || N <= 0 || N % 32 || N > 1024 || K != kK) return -1;
    if ((epilogue == EPI_BIAS || epilogue == EPI_BIAS_RELU) ? !bias : !aux) return -1;
    if (((uintptr_t)a | (uintptr_t)a2 | (uintptr_t)b_frags | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)aux) & 15) return -1;
    const unsigned char *wf = reinterpret_cast<const unsigned char *>(b_frags);
    const float *winv = reinterpret_cast<const float *>(wf + (size_t)4 * N * K);
    return a2 ? launch_epi<true>(epilogue, a, a2, wf, winv, bias, aux, c, M, N, stream)
              : launch_epi<false>(epilogue, a, a2, wf, winv, bias, aux, c, M, N, stream);
}
