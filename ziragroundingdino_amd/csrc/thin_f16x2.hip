// thin_f16x2.hip -- the THIN products of the image <-> text fusion block's image side on the f16 matrix cores of gfx950 (MI355X),
// in fp32 accuracy: per image, 22 k token rows against a matrix that has the text side's H * T (64 ... 128) on one of its sides,
//
//     C[b] [M, N] = A[b] [M, K] * W[b]  (+ A2[b] [M, K] * W2[b])  (+ bias[b] [N])  (+ res[b] [M, N])
//
// with (K, N) = (256, H T): the scores v (Wq^T k^T) and the two [M, 256] -> [M, H T] gradients of the backward, or
// (K, N) = (H T, 256): the image output [P_v]_h-cat (value_l Wo^T) + bias + residual and the two [M, H T] -> [M, 256] gradients,
// which the second source adds up in ONE pass over the output (g_v = e g_u + g_xm a^T: a contraction over the concatenated
// index).  Reference: the re-bracketed form of BiMultiHeadAttention.forward, models/GroundingDINO/fuse_modules.py:170-248
// (see transformer.py BiMultiHeadAttention.forward here for the algebra).  W changes every step (it is made from the text
// tokens), so it is split into fragment order by a small launch per call.
//
// These products move 57 MB for 1.5 GFLOP: memory decides.  The library's fp32 kernels take 37-41 us for each of them (1.4 TB/s);
// the arithmetic and block shape are those of csrc/gemm_f16x2_panel.hip: a block takes 32 rows and ALL of K, requests its panel
// at once, scales each row by a power of two (largest magnitude into [2^14, 2^15)), splits it into two f16 planes laid into LDS
// in matrix-core fragment order, and after one barrier every wave walks its share of the N / 32 column tiles with three exact
// matrix-core terms per 16-deep step (a2 w1 + a1 w2 + a1 w1), fp32 sums.  The two sources of a concatenated contraction keep
// their own row scales, column scales and accumulators (their magnitudes differ by orders: probabilities against gradients) and
// meet in the epilogue.  K and N need not be multiples of 32: the panel's missing columns are zeros and the last column tile
// stores only what exists (H T is a multiple of 4 = H).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRows = 32, kThreads = 256;

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

struct ThinArgs {
    const float *A, *A2;             // [B][M][K]; A2 = the second source of a concatenated contraction (CAT) or null
    const unsigned char *Wf, *Wf2;   // fragments of W[b] (frag_bytes apart), see thin_split_kernel
    const float *bias;               // [B][N] or null
    const float *res;                // [B][M][N] or null
    float *C;                        // [B][M][N]
    int M, N, K;
    size_t frag_bytes;
};

// One source's 32 x (16 KS) panel, requested: thread (row, c) takes columns 4 c + 32 j of row m0 + row; columns past K are zeros.
template <int KS>
__device__ __forceinline__ void panel_load(const float *__restrict__ A, int M, int K, int m0, int tid, float4 (&v)[KS / 2])
{
    const int row = tid >> 3, c = tid & 7;
    int m = m0 + row;
    m = m < M ? m : M - 1;                                  // (rows past the end repeat the last row; nothing of theirs is stored)
    const float *ar = A + (size_t)m * K + 4 * c;
#pragma unroll
    for (int j = 0; j < KS / 2; ++j)
        v[j] = (4 * c + 32 * j < K) ? *reinterpret_cast<const float4 *>(ar + 32 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ... scaled per row, split, laid down: planes [2][KS][2 halves of a step][32 rows][8 halves] in LDS, 1 / scale of each row in sinv
template <int KS>
__device__ __forceinline__ void panel_store(const float4 (&v)[KS / 2], int tid, unsigned char *planes, float *sinv)
{
    constexpr int J = KS / 2;
    const int row = tid >> 3, c = tid & 7;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    amax = fmaxf(amax, __shfl_xor(amax, 2));
    amax = fmaxf(amax, __shfl_xor(amax, 4));
    float s, inv;
    pow2_scale(amax, s, inv);
    if (c == 0) sinv[row] = inv;
    // columns 4 c + 32 j .. + 3: step 2 j + (c >> 2), half (c >> 1) & 1, halves 4 (c & 1) .. + 3 of the lane's eight
    unsigned char *dst = planes + ((c >> 2) * 2 + ((c >> 1) & 1)) * 512 + row * 16 + (c & 1) * 8;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const float x = v[j].x * s, y = v[j].y * s, z = v[j].z * s, w = v[j].w * s;
        uint2 p1, p2;
        p1.x = pk_f16(x, y);
        p1.y = pk_f16(z, w);
        p2.x = pk_f16(x - f16_lo(p1.x), y - f16_hi(p1.x));   // (exact differences)
        p2.y = pk_f16(z - f16_lo(p1.y), w - f16_hi(p1.y));
        *reinterpret_cast<uint2 *>(dst + (2 * j) * 1024) = p1;
        *reinterpret_cast<uint2 *>(dst + (2 * j) * 1024 + KS * 1024) = p2;
    }
}

// The weight fragments of a wave's column tiles are ONE stream (global memory, L2-resident), read AHEAD matrix-core steps ahead
// of their use across tile boundaries: the first steps are requested together with the panel, before the block's barrier.
template <int KS>
struct WeightRing {
    static constexpr int AHEAD = KS % 4 == 0 ? 4 : 2;   // (divides KS: the slots line up at tile boundaries)
    f16x8 w1[AHEAD], w2[AHEAD];
    __device__ __forceinline__ void request(const unsigned char *wt, int st, int slot)
    {
        w1[slot] = *reinterpret_cast<const f16x8 *>(wt + (st * 2 + 0) * 1024);
        w2[slot] = *reinterpret_cast<const f16x8 *>(wt + (st * 2 + 1) * 1024);
    }
    __device__ __forceinline__ void prime(const unsigned char *wt)
    {
#pragma unroll
        for (int st = 0; st < AHEAD; ++st) request(wt, st, st);
    }
};

// One source's contribution to a 32 x 32 tile of C^T; `next` = the fragments of the wave's next tile (or null).
template <int KS>
__device__ __forceinline__ f32x16 tile_product(WeightRing<KS> &ring, const unsigned char *wt, const unsigned char *next, const unsigned char *pa)
{
    constexpr int AHEAD = WeightRing<KS>::AHEAD, LA = 2;
    static_assert(KS % AHEAD == 0, "the ring's slots line up at tile boundaries");
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f16x8 a1[LA], a2[LA];
#pragma unroll
    for (int st = 0; st < LA; ++st) {
        a1[st] = *reinterpret_cast<const f16x8 *>(pa + st * 1024);
        a2[st] = *reinterpret_cast<const f16x8 *>(pa + st * 1024 + KS * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        const f16x8 u1 = ring.w1[st % AHEAD], u2 = ring.w2[st % AHEAD], b1 = a1[st % LA], b2 = a2[st % LA];
        if (st + AHEAD < KS)
            ring.request(wt, st + AHEAD, st % AHEAD);
        else if (next)
            ring.request(next, st + AHEAD - KS, st % AHEAD);
        if (st + LA < KS) {
            a1[st % LA] = *reinterpret_cast<const f16x8 *>(pa + (st + LA) * 1024);
            a2[st % LA] = *reinterpret_cast<const f16x8 *>(pa + (st + LA) * 1024 + KS * 1024);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(u2, b1, acc, 0, 0, 0);   // the small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(u1, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(u1, b1, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// KS = (K rounded up to 32) / 16 matrix-core steps per source.  grid (ceil(M / 32), B).
template <int KS, bool CAT>
__global__ __launch_bounds__(kThreads, CAT ? 3 : 4) void thin_f16x2_kernel(ThinArgs p)
{
    constexpr int S = CAT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) unsigned char planes[S * 2 * KS * 1024];
    __shared__ float sinv[S * kRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * kRows, b = blockIdx.y;
    const int M = p.M, N = p.N, K = p.K;
    const size_t abase = (size_t)b * M * K, cbase = (size_t)b * M * N;
    const int lm = lane & 31, hf = lane >> 5;
    const int m = m0 + lm;
    const int ntiles = (N + 31) >> 5;
    const unsigned char *wf = p.Wf + (size_t)b * p.frag_bytes + lane * 16, *wf2 = CAT ? p.Wf2 + (size_t)b * p.frag_bytes + lane * 16 : nullptr;
    auto frag = [&](const unsigned char *base, int t) { return base + ((size_t)t * KS * 2) * 1024; };   // fragment (t, st, plane) at ((t KS + st) 2 + plane) KB

    // everything the block waits for is requested up front: its panel(s), the first weight fragments of every wave's first tile
    // and that tile's rows of the residual
    float4 v[KS / 2], v2[CAT ? KS / 2 : 1];
    panel_load<KS>(p.A + abase, M, K, m0, tid, v);
    if (CAT) panel_load<KS>(p.A2 + abase, M, K, m0, tid, reinterpret_cast<float4 (&)[KS / 2]>(v2));
    WeightRing<KS> ring, ring2;
    if (wave < ntiles) {
        ring.prime(frag(wf, wave));
        if (CAT) ring2.prime(frag(wf2, wave));
    }
    auto residual = [&](int t, float4 (&h)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * t + 8 * g + 4 * hf;
            h[g] = (p.res && m < M && n < N) ? *reinterpret_cast<const float4 *>(p.res + cbase + (size_t)m * N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float4 h[4];
    if (wave < ntiles) residual(wave, h);
    panel_store<KS>(v, tid, planes, sinv);
    if (CAT) panel_store<KS>(reinterpret_cast<const float4 (&)[KS / 2]>(v2), tid, planes + 2 * KS * 1024, sinv + kRows);
    __syncthreads();

    // column tiles: wave w takes tiles w, w + 4, ...; the matrix core's first operand is the weight fragment, so accumulator
    // register 4 g + i of lane (lm, hf) is C[m0 + lm][32 t + 8 g + 4 hf + i]: 16-byte stores (one 4-byte store per register with
    // the operands swapped -- two full 128-byte lines per instruction -- was measured: 28 against 19 us)
    const float inv = sinv[lm], inv2 = CAT ? sinv[kRows + lm] : 0.f;
    const unsigned char *pa = planes + hf * 512 + lm * 16;
    const float *winv = reinterpret_cast<const float *>(p.Wf + (size_t)b * p.frag_bytes + (size_t)ntiles * KS * 2048);
    const float *winv2 = CAT ? reinterpret_cast<const float *>(p.Wf2 + (size_t)b * p.frag_bytes + (size_t)ntiles * KS * 2048) : nullptr;
    for (int t = wave; t < ntiles; t += 4) {
        const bool more = t + 4 < ntiles;
        const f32x16 acc = tile_product<KS>(ring, frag(wf, t), more ? frag(wf, t + 4) : nullptr, pa);
        f32x16 acc2;
        if (CAT) acc2 = tile_product<KS>(ring2, frag(wf2, t), more ? frag(wf2, t + 4) : nullptr, pa + 2 * KS * 1024);
        if (m < M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * t + 8 * g + 4 * hf;
                if (n >= N) continue;
                const float4 wi = *reinterpret_cast<const float4 *>(winv + n);
                float4 o = make_float4(acc[4 * g] * inv * wi.x, acc[4 * g + 1] * inv * wi.y, acc[4 * g + 2] * inv * wi.z,
                                       acc[4 * g + 3] * inv * wi.w);
                if (CAT) {
                    const float4 wj = *reinterpret_cast<const float4 *>(winv2 + n);
                    o.x = fmaf(acc2[4 * g] * inv2, wj.x, o.x);
                    o.y = fmaf(acc2[4 * g + 1] * inv2, wj.y, o.y);
                    o.z = fmaf(acc2[4 * g + 2] * inv2, wj.z, o.z);
                    o.w = fmaf(acc2[4 * g + 3] * inv2, wj.w, o.w);
                }
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(p.bias + (size_t)b * N + n);
                    o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                }
                o.x += h[g].x; o.y += h[g].y; o.z += h[g].z; o.w += h[g].w;
                *reinterpret_cast<float4 *>(p.C + cbase + (size_t)m * N + n) = o;
            }
        }
        if (more) residual(t + 4, h);
    }
}

// W[b] fp32 -> fragments [ceil(N / 32)][KS][2 planes][64 lanes][8 halves] of Wn[n][k] * scale[n] (rows n >= N and columns k >= K
// are zeros), then 1 / scale [32 ceil(N / 32)]; Wn[n][k] = W[b][n][k] (W is [N][K]) or W[b][k][n] (w_is_kn: W is [K][N]).
// grid (32 ceil(N / 32), B), one block per row n, one thread per k.
__global__ __launch_bounds__(256) void thin_split_kernel(const float *__restrict__ w, int N, int K, int KS, int w_is_kn,
                                                         unsigned char *__restrict__ frags, size_t frag_bytes)
{
    __shared__ float red[256];
    const int n = blockIdx.x, b = blockIdx.y, k = threadIdx.x;
    const int ntiles = gridDim.x >> 5;
    const float *wb = w + (size_t)b * N * K;
    float v = 0.f;
    if (n < N && k < K) v = w_is_kn ? wb[(size_t)k * N + n] : wb[(size_t)n * K + k];
    red[k] = fabsf(v);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (k < s) red[k] = fmaxf(red[k], red[k + s]);
        __syncthreads();
    }
    float s, inv;
    pow2_scale(red[0], s, inv);
    unsigned short *f = reinterpret_cast<unsigned short *>(frags + (size_t)b * frag_bytes);
    float *winv = reinterpret_cast<float *>(frags + (size_t)b * frag_bytes + (size_t)ntiles * KS * 2048);
    if (k == 0) winv[n] = inv;
    if (k < 16 * KS) {
        v *= s;
        const unsigned p1 = pk_f16(v, 0.f);
        const unsigned p2 = pk_f16(v - f16_lo(p1), 0.f);
        const int t = n >> 5, lm = n & 31, st = k >> 4, hf = (k >> 3) & 1, e = k & 7;
        const size_t base = (((size_t)t * KS + st) * 2) * 512 + (size_t)(hf * 32 + lm) * 8 + e;   // in halves; plane 2 is 512 halves on
        f[base] = (unsigned short)(p1 & 0xFFFFu);
        f[base + 512] = (unsigned short)(p2 & 0xFFFFu);
    }
}

inline int steps_of(int K)
{
    const int ks = ((K + 31) / 32) * 2;
    return (ks == 2 || ks == 4 || ks == 6 || ks == 8 || ks == 16) ? ks : 0;
}

inline size_t frag_bytes_of(int N, int ks)
{
    const size_t ntiles = (size_t)(N + 31) / 32;
    return ntiles * ks * 2048 + ntiles * 32 * sizeof(float);
}

template <int KS>
void launch(bool cat, const ThinArgs &p, int B, hipStream_t st)
{
    const dim3 grid((p.M + kRows - 1) / kRows, B), block(kThreads);
    if constexpr (KS <= 8) {
        if (cat) {
            hipLaunchKernelGGL((thin_f16x2_kernel<KS, true>), grid, block, 0, st, p);
            return;
        }
    }
    hipLaunchKernelGGL((thin_f16x2_kernel<KS, false>), grid, block, 0, st, p);
}

}  // namespace

extern "C" size_t zira_thin_f16x2_frag_bytes(int N, int K)
{
    const int ks = steps_of(K);
    return (ks && N > 0) ? frag_bytes_of(N, ks) : 0;
}

extern "C" int zira_thin_f16x2_split_f32(const float *w, int batch, int N, int K, int w_is_kn, void *frags, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int ks = steps_of(K);
    if (!w || !frags || batch <= 0 || N <= 0 || K <= 0 || !ks || ((uintptr_t)frags & 15)) return -1;
    const int ntiles = (N + 31) / 32;
    hipLaunchKernelGGL(thin_split_kernel, dim3(32 * ntiles, batch), dim3(256), 0, stream, w, N, K, ks, w_is_kn ? 1 : 0,
                       reinterpret_cast<unsigned char *>(frags), frag_bytes_of(N, ks));
    return (int)hipGetLastError();
}

extern "C" int zira_thin_f16x2_f32(const float *a, const void *frags, const float *a2, const void *frags2, int batch, int M, int N, int K,
                                   const float *bias, const float *res, float *c, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int ks = steps_of(K);
    if (!a || !frags || !c || batch <= 0 || batch > 65535 || M <= 0 || N <= 0 || K <= 0 || !ks || (N & 3) || (K & 3)) return -1;
    if ((a2 == nullptr) != (frags2 == nullptr) || (a2 && ks > 8)) return -1;
    if (((uintptr_t)a | (uintptr_t)a2 | (uintptr_t)frags | (uintptr_t)frags2 | (uintptr_t)c | (uintptr_t)bias | (uintptr_t)res) & 15) return -1;
    ThinArgs p;
    p.A = a; p.A2 = a2;
    p.Wf = reinterpret_cast<const unsigned char *>(frags); p.Wf2 = reinterpret_cast<const unsigned char *>(frags2);
    p.bias = bias; p.res = res; p.C = c;
    p.M = M; p.N = N; p.K = K;
    p.frag_bytes = frag_bytes_of(N, ks);
    const bool cat = a2 != nullptr;
    switch (ks) {
    case 2: launch<2>(cat, p, batch, stream); break;
    case 4: launch<4>(cat, p, batch, stream); break;
    case 6: launch<6>(cat, p, batch, stream); break;
    case 8: launch<8>(cat, p, batch, stream); break;
    case 16: launch<16>(cat, p, batch, stream); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
