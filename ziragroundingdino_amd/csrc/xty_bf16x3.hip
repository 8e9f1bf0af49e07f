// xty_bf16x3.hip -- the TALL reductions of the fusion block's image side on the bf16 matrix cores of gfx950 (MI355X) in fp32
// accuracy:   out[b] = P[b]^T Q[b],   P [N, n] (n = heads x text tokens: 64 ... 1024), Q [N, 256], N = the image tokens (22 k).
//
// These are the three reductions over the image tokens that the re-bracketed BiMultiHeadAttention leaves (reference
// models/GroundingDINO/fuse_modules.py:170-248; see csrc/xty.hip): (text probabilities)^T x tokens in the forward, tokens^T x
// (score gradients) and (probabilities)^T x (output gradients) in the backward.  csrc/xty.hip computes them on the fp32 matrix
// instruction (v_mfma_f32_32x32x2_f32, the fp32 vector rate): 37 us per call at n = 128 and 350 us at n = 776 (COCO-length
// captions), where the three of them were 6.4 ms of a 38.8 ms step.  Here each fp32 operand is split into THREE bfloat16 planes
// (x = x1 + x2 + x3 exactly to 2^-24: bfloat16 keeps the fp32 exponent, so no scaling and no bookkeeping of scales along the
// 22 k-long sums) and the six product terms of weight <= 2^-16 run on v_mfma_f32_32x32x16_bf16, 16 x the rate per
// instruction, with fp32 accumulation inside the matrix core -- the arithmetic of csrc/gemm_bf16x3.hip.
//
// The contraction index is the token, which is the SLOW index of both operands in memory.  A block stages a slice of 32 tokens
// -- its 64 columns of P and all 256 of Q -- row-major in LDS (coalesced global loads, 8-byte LDS stores, three planes) and the
// matrix-core operands are read with gfx950's transposing LDS read (ds_read_b64_tr_b16: 4 tokens x 16 columns per 16 lanes,
// delivered column-major), rows padded by 64 bytes so that the four token rows of a read fall into different banks.
// Grid: (token chunks x 64-column tiles of P, B); blocks that share a chunk of Q sit on one XCD next to each other.  Each block
// writes one partial [64, 256] tile; a second launch adds the chunks' partial tiles in a fixed order (deterministic, no
// atomics) and writes out[b] as [n, 256] or, transposed, [256, n].

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256, kTok = 32, kTP = 64, kQ = 256;
constexpr int kStrideP = kTP * 2 + 64, kStrideQ = kQ * 2 + 64;       // bytes per token row of a plane
constexpr int kPlaneP = kTok * kStrideP, kPlaneQ = kTok * kStrideQ;
constexpr int kLds = 3 * (kPlaneP + kPlaneQ);                          // 73728 bytes: two blocks per CU

__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    f32x2 x = {a, b};
    bf16x2 h = __builtin_convertvector(x, bf16x2);   // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xFFFF0000u); }

// four fp32 numbers -> their three bfloat16 planes, four halves (8 bytes) each
__device__ __forceinline__ void split4(const float4 v, uint2 &p1, uint2 &p2, uint2 &p3)
{
    p1.x = pk_bf16(v.x, v.y);
    p1.y = pk_bf16(v.z, v.w);
    const float rx = v.x - bf_lo(p1.x), ry = v.y - bf_hi(p1.x), rz = v.z - bf_lo(p1.y), rw = v.w - bf_hi(p1.y);   // exact
    p2.x = pk_bf16(rx, ry);
    p2.y = pk_bf16(rz, rw);
    p3.x = pk_bf16(rx - bf_lo(p2.x), ry - bf_hi(p2.x));   // (the differences are exact, and fit bfloat16 exactly)
    p3.y = pk_bf16(rz - bf_lo(p2.y), rw - bf_hi(p2.y));
}

// 8 tokens x 1 column per lane (the matrix core's operand: lane (i = l % 32, kg = l / 32) holds k = 8 kg .. + 7 of column i) from
// a row-major [token][column] plane: two transposing reads of 4 tokens x 16 columns per 16-lane group.  `lane_base` = the
// lane's address for k-step 0 of column block 0 (see the kernel); `off` = immediate byte offset of (plane, column block, k-step).
__device__ __forceinline__ bf16x8 frag_tr(const unsigned char *lane_base, int off, int stride)
{
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lane_base + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(lane_base + off + 4 * stride));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// part [B][chunks][ntiles * 64][256]; blockIdx.x = xcd + 8 * ((chunk / 8) * ntiles + tile) with chunk % 8 == xcd
__global__ __launch_bounds__(kThreads, 2) void xty_bf16x3_kernel(const float *__restrict__ P, const float *__restrict__ Q, int N, int n,
                                                               int chunk_rows, int chunks, int ntiles, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int chunk = (slot / ntiles) * 8 + xcd, tile = slot % ntiles, b = blockIdx.y;
    const int n_begin = chunk * chunk_rows;
    const int n_end = n_begin + chunk_rows < N ? n_begin + chunk_rows : N;
    const float *Pb = P + (size_t)b * N * n, *Qb = Q + (size_t)b * N * kQ;

    // staging: P slab 32 tokens x 64 columns = 2 float4 per thread, Q slab 32 x 256 = 8 float4 per thread
    const int p_tok = tid >> 4, p_c4 = tid & 15;          // + 16 tokens for the second piece
    const int q_tok = tid >> 6, q_c4 = tid & 63;          // + 4 tokens per piece
    const int p_col = tile * kTP + 4 * p_c4;
    const bool p_in = p_col < n;                          // (n % 4 == 0: a float4 is inside or outside)
    float4 rp[2], rq[8];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = t0 + p_tok + 16 * i;
            rp[i] = (p_in && t < n_end) ? *reinterpret_cast<const float4 *>(Pb + (size_t)t * n + p_col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + q_tok + 4 * i;
            rq[i] = t < n_end ? *reinterpret_cast<const float4 *>(Qb + (size_t)t * kQ + 4 * q_c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    unsigned char *const wp = smem + p_tok * kStrideP + p_c4 * 8;
    unsigned char *const wq = smem + 3 * kPlaneP + q_tok * kStrideQ + q_c4 * 8;
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint2 a, c, d;
            split4(rp[i], a, c, d);
            unsigned char *w = wp + 16 * i * kStrideP;
            *reinterpret_cast<uint2 *>(w) = a;
            *reinterpret_cast<uint2 *>(w + kPlaneP) = c;
            *reinterpret_cast<uint2 *>(w + 2 * kPlaneP) = d;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint2 a, c, d;
            split4(rq[i], a, c, d);
            unsigned char *w = wq + 4 * i * kStrideQ;
            *reinterpret_cast<uint2 *>(w) = a;
            *reinterpret_cast<uint2 *>(w + kPlaneQ) = c;
            *reinterpret_cast<uint2 *>(w + 2 * kPlaneQ) = d;
        }
    };

    // transposing reads: 16-lane group g = lane / 16 serves operand lanes i = 16 (g & 1) .. + 15 with kg = g >> 1; lane 4 q + p of
    // the group supplies the address of token row q, columns 4 p .. 4 p + 3 of the group's 16
    const int g = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
    const unsigned char *const ap = smem + (8 * (g >> 1) + gq) * kStrideP + (16 * (g & 1) + 4 * gp) * 2;
    const unsigned char *const aq = smem + 3 * kPlaneP + (8 * (g >> 1) + gq) * kStrideQ + (16 * (g & 1) + 4 * gp) * 2 + wave * 128;   // wave w: column blocks 2 w, 2 w + 1 of Q

    f32x16 acc[2][2];                                     // [column block of P][column block of Q]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(n_begin);
    for (int t0 = n_begin; t0 < n_end; t0 += kTok) {
        stage();
        __syncthreads();
        fetch(t0 + kTok);                                 // (past the chunk: zeros, no loads)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    fa[i][pl] = frag_tr(ap, pl * kPlaneP + i * 64 + ks * 16 * kStrideP, kStrideP);
                    fb[i][pl] = frag_tr(aq, pl * kPlaneQ + i * 64 + ks * 16 * kStrideQ, kStrideQ);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];                 // the six terms of weight <= 2^-16, the small ones first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        __syncthreads();                                  // the slice's reads are done before the next one is staged
    }

    // accumulator register r of lane (c = lane % 32, h = lane / 32) of block (i, j): row 32 i + 8 (r / 4) + 4 h + r % 4 of the tile,
    // column 64 wave + 32 j + c: 128 contiguous bytes per row and store instruction
    float *pt = part + (((size_t)b * chunks + chunk) * ntiles + tile) * (size_t)(kTP * kQ) + 64 * wave + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                pt[(size_t)row * kQ + 32 * j] = acc[i][j][r];
            }
}

// out[b][i][j] (transpose: out[b][j][i]) = sum over chunks of part[b][chunk][i][j], i < n, j < 256.  A block folds 16 float4 of
// the output: 16 chunk groups (thread / 16) each add every 16th partial tile, LDS joins the groups in a fixed order.
__global__ __launch_bounds__(kThreads) void xty_bf16x3_fold(const float *__restrict__ part, int chunks, int rows_padded, int n, int transpose,
                                                           float *__restrict__ out)
{
    __shared__ float4 red[16][16];
    const int grp = threadIdx.x >> 4, k = threadIdx.x & 15, b = blockIdx.y;
    const int idx = blockIdx.x * 16 + k;                  // float4 number idx of the [n][64] float4 of out[b]
    const int i = idx >> 6, j4 = idx & 63;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        const float *p = part + ((size_t)b * chunks * rows_padded + i) * kQ + 4 * j4;
        for (int c = grp; c < chunks; c += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)c * rows_padded * kQ);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[grp][k] = s;
    __syncthreads();
    if (grp != 0 || i >= n) return;
#pragma unroll
    for (int g = 1; g < 16; ++g) {
        const float4 v = red[g][k];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float *o = out + (size_t)b * n * kQ;
    if (!transpose) {
        *reinterpret_cast<float4 *>(o + (size_t)i * kQ + 4 * j4) = s;
    } else {
        o[(size_t)(4 * j4) * n + i] = s.x;
        o[(size_t)(4 * j4 + 1) * n + i] = s.y;
        o[(size_t)(4 * j4 + 2) * n + i] = s.z;
        o[(size_t)(4 * j4 + 3) * n + i] = s.w;
    }
}

inline int tiles_of(int n) { return (n + kTP - 1) / kTP; }

// token chunks: a multiple of 8 (one XCD per residue), about 400 blocks in all
inline int chunks_of(int B, int N, int n)
{
    int c = (400 + tiles_of(n) * B - 1) / (tiles_of(n) * B);
    c = (c + 7) & ~7;
    const int most = ((N + kTok - 1) / kTok + 7) & ~7;    // (no chunk shorter than a slice, where N allows)
    c = c > most ? most : c;
    return c < 8 ? 8 : (c > 256 ? 256 : c);
}

}  // namespace

extern "C" size_t zira_xty_bf16x3_workspace_floats(int B, int N, int n)
{
    if (B <= 0 || N <= 0 || n <= 0) return 0;
    return (size_t)B * chunks_of(B, N, n) * tiles_of(n) * kTP * kQ;
}

// P [B][N][n], Q [B][N][256] -> out [B][n][256] (transpose = 0) or [B][256][n] (transpose = 1); n % 4 == 0; 16-byte aligned
extern "C" int zira_xty_bf16x3_f32(const float *P, const float *Q, int B, int N, int n, int transpose, float *out, float *workspace,
                                   void *stream_)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    if (!P || !Q || !out || !workspace || B <= 0 || B > 65535 || N <= 0 || n <= 0 || (n & 3)) return -1;
    if (((uintptr_t)P | (uintptr_t)Q | (uintptr_t)out | (uintptr_t)workspace) & 15) return -1;
    const int ntiles = tiles_of(n), chunks = chunks_of(B, N, n);
    const int chunk_rows = (((N + chunks - 1) / chunks) + kTok - 1) / kTok * kTok;   // whole slices
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(xty_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(xty_bf16x3_kernel, dim3(chunks * ntiles, B), dim3(kThreads), kLds, st, P, Q, N, n, chunk_rows, chunks, ntiles, workspace);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(xty_bf16x3_fold, dim3((n * 64 + 15) / 16, B), dim3(kThreads), 0, st, workspace, chunks, ntiles * kTP, n,
                       transpose ? 1 : 0, out);
    return (int)hipGetLastError();
}
