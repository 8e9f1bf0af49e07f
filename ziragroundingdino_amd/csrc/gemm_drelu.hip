// gemm_drelu.hip -- C = (A * B) (.) (H > 0): the input gradient of a frozen  linear -> ReLU -> linear  block without the
// intermediate (C ABI: zira_gemm_drelu_f32).
//
// Reference: the FFN of the deformable encoder / decoder layers (transformer_for_adapter.py:823-826, :883-886,
// :1001-1006: linear2(dropout(relu(linear1(x))))).  With frozen weights autograd runs its backward as
//   gh = gy @ W2          [M, 2048]   (a GEMM writing 364 MB at the encoder shape)
//   g  = gh * (h > 0)     threshold_backward: reads gh and h, writes g: 1.1 GB, 198 us -- 1.2 ms per step
//   gx = g @ W1
// No GEMM library on this stack has a ReLU-gradient epilogue (hipBLASLt: dGELU only), so the first product and the mask are
// one kernel here: fp32 MFMA (v_mfma_f32_32x32x2_f32), 128 x 128 tiles of C per block, four waves of 64 x 64, K in steps of
// 16 through double-buffered LDS (A transposed to [k][m] on the way in and both tiles swizzled, so that every operand read
// is one conflict-free dword per lane), the next step's global loads in flight during the MFMAs; the epilogue reads h where it
// writes C.  Measured at M = 44446, N = 2048, K = 256: 480-490 us (97 TF/s; rocBLAS / hipBLASLt run the bare product in 350 us, at
// ~100 % of the MFMA cycles of the 2.05 GHz the chip holds under this load) against 350 + 190 us for mm + threshold_backward.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef ZIRA_GD_BK
#define ZIRA_GD_BK 16
#endif
#ifndef ZIRA_GD_OCC
#define ZIRA_GD_OCC 2
#endif
constexpr int kBM = 128, kBN = 128, kBK = ZIRA_GD_BK, kThreads = 256;
// LDS: rows of 128 floats, element (k, x) at column (x + 16 (k / 4) + 32 (k % 2)) % 128: the two k rows an MFMA operand
// read touches (lanes 0-31: row k, lanes 32-63: row k + 1) fall into opposite halves of the 64 banks, and the four k rows a
// wave writes at a time (16 columns each) into four different quarters -- no bank conflicts either way (a padded stride
// of 132 had every operand read two-way conflicted, with the LDS the busiest unit of the CU).
__device__ __forceinline__ int swz(int k, int x) { return k * 128 + ((x + 16 * (k >> 2) + 32 * (k & 1)) & 127); }

__device__ __forceinline__ unsigned rowmap(unsigned reg, unsigned hh) { return (reg & 3u) + 8u * (reg >> 2) + 4u * hh; }

// A [M, K] row-major (lda), B [K, N] row-major (ldb), H / C [M, N] row-major (ldc).  N % 128 == 0, K % 16 == 0.
__global__ __launch_bounds__(kThreads, ZIRA_GD_OCC) void gemm_nn_drelu(const float *__restrict__ A, const float *__restrict__ B,
                                                          const float *__restrict__ H, int M, int N, int K, int lda, int ldb,
                                                          int ldc, float *__restrict__ C)
{
    __shared__ __align__(16) float As[2][kBK * kBM];   // [k][m], swizzled
    __shared__ __align__(16) float Bs[2][kBK * kBN];   // [k][n], swizzled
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const unsigned wr = wave >> 1, wc = wave & 1;           // the wave's 64 x 64 quarter of the tile
    // consecutive blocks share the A rows (N / 128 column tiles of one row tile run together: A stays in L2)
    const int nbn = N / kBN;
    const int bm = (int)(blockIdx.x / nbn) * kBM, bn = (int)(blockIdx.x % nbn) * kBN;

    // global -> registers: A tile 128 x 16 = 512 float4 (thread t: rows t / 4 and t / 4 + 64, k-quad t % 4);
    //                      B tile 16 x 128 = 512 float4 (thread t: k rows t / 32 and t / 32 + 8, n-quad t % 32)
    const int am = (int)(tid >> 2), ak = (int)(tid & 3) * 4;
    const int bk = (int)(tid >> 5), bnq = (int)(tid & 31) * 4;
    const int am0 = bm + am < M ? bm + am : M - 1, am1 = bm + am + 64 < M ? bm + am + 64 : M - 1;   // (rows beyond M: clamped, never stored)
    const float *a0p = A + (size_t)am0 * lda + ak, *a1p = A + (size_t)am1 * lda + ak;
    const float *b0p = B + (size_t)bk * ldb + bn + bnq, *b1p = B + (size_t)(bk + 8) * ldb + bn + bnq;
    float4 ga0 = *reinterpret_cast<const float4 *>(a0p), ga1 = *reinterpret_cast<const float4 *>(a1p);
    float4 gb0 = *reinterpret_cast<const float4 *>(b0p), gb1 = *reinterpret_cast<const float4 *>(b1p);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto stage = [&](int buf) {
        As[buf][swz(ak + 0, am)] = ga0.x; As[buf][swz(ak + 1, am)] = ga0.y; As[buf][swz(ak + 2, am)] = ga0.z; As[buf][swz(ak + 3, am)] = ga0.w;
        As[buf][swz(ak + 0, am + 64)] = ga1.x; As[buf][swz(ak + 1, am + 64)] = ga1.y; As[buf][swz(ak + 2, am + 64)] = ga1.z; As[buf][swz(ak + 3, am + 64)] = ga1.w;
        *reinterpret_cast<float4 *>(&Bs[buf][swz(bk, bnq)]) = gb0;
        *reinterpret_cast<float4 *>(&Bs[buf][swz(bk + 8, bnq)]) = gb1;
    };
    stage(0);
    __syncthreads();

    // this lane's operand offsets inside a k-row pair (kp, kp + 1), per group of four k rows: swz(kp + hh, x) - kp * 128
    int oa0[kBK / 4], oa1[kBK / 4], ob0[kBK / 4], ob1[kBK / 4];
#pragma unroll
    for (int g4 = 0; g4 < kBK / 4; ++g4) {
        oa0[g4] = swz(4 * g4 + (int)hh, (int)(wr * 64 + r)) - 4 * g4 * 128;
        oa1[g4] = swz(4 * g4 + (int)hh, (int)(wr * 64 + 32 + r)) - 4 * g4 * 128;
        ob0[g4] = swz(4 * g4 + (int)hh, (int)(wc * 64 + r)) - 4 * g4 * 128;
        ob1[g4] = swz(4 * g4 + (int)hh, (int)(wc * 64 + 32 + r)) - 4 * g4 * 128;
    }
    const int nk = K / kBK;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {   // the next step's tiles: in flight during this step's MFMAs
            const int ko = (kt + 1) * kBK;
            ga0 = *reinterpret_cast<const float4 *>(a0p + ko);
            ga1 = *reinterpret_cast<const float4 *>(a1p + ko);
            gb0 = *reinterpret_cast<const float4 *>(b0p + (size_t)ko * ldb);
            gb1 = *reinterpret_cast<const float4 *>(b1p + (size_t)ko * ldb);
        }
#pragma unroll
        for (int kp = 0; kp < kBK; kp += 2) {
            const float a0 = As[buf][kp * 128 + oa0[kp >> 2]], a1 = As[buf][kp * 128 + oa1[kp >> 2]];
            const float b0 = Bs[buf][kp * 128 + ob0[kp >> 2]], b1 = Bs[buf][kp * 128 + ob1[kp >> 2]];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        // (measured and not kept: all operand reads of the step ahead of its MFMAs, 525 against 497 us; the hand-over
        // below in the middle of the MFMA sequence, 523-528 against 511-515 on the same box; three blocks per CU, 536; the
        // tiles of the last, partial round of blocks as 128 x 64 halves in a second launch, 458 + 69 against 500 us;
        // round 4: persistent blocks walking the tiles with the next tile's first operand loads issued before the epilogue --
        // 561 us with two blocks per CU (154 registers), 558 us with four (128 registers, 112 bytes of scratch) against
        // 460 us for this form on the same box: the prefetched operands live through the epilogue and cost the fourth block)
        if (kt + 1 < nk) {
            stage(buf ^ 1);      // (the other buffer: last read a step ago, behind the barrier at its end)
            __syncthreads();
        }
    }

    // epilogue: acc[i][j][reg] = (A B)[bm + wr * 64 + 32 i + rowmap(reg, hh)][bn + wc * 64 + 32 j + r].  The accumulator
    // layout has a lane per column; a 16 x 64 slab at a time goes through the wave's quarter of the (now idle) operand
    // buffers and comes back a row segment per 16 lanes, so that h is read and C written with 16-byte accesses (a dword
    // per lane and element took 128 memory instructions per lane and 63 us of the kernel; this form 32: 500 -> 480 us).
    __syncthreads();   // (every wave is done with the operand tiles)
    float *slab = &As[0][0] + wave * (16 * 64);   // As[2][2048] is 4 x 1024 floats: a quarter per wave
    static_assert(2 * kBK * kBM >= 4 * 16 * 64, "the epilogue's slabs live in the A operand buffers");
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // rows 32 i + 16 half .. + 15 of the wave's tile: accumulator registers 8 half .. 8 half + 7
#pragma unroll
            for (unsigned q = 0; q < 8; ++q) {
                const unsigned reg = 8 * half + q, lrow = rowmap(reg, hh) - 16 * half;
                slab[lrow * 64 + r] = acc[i][0][reg];
                slab[lrow * 64 + 32 + r] = acc[i][1][reg];
            }
            // lane l: column quad l % 16, rows l / 16 + 4 t
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = (int)(lane >> 4) + 4 * t, c4 = (int)(lane & 15) * 4;
                const int m = bm + (int)(wr * 64) + 32 * i + 16 * half + row;
                if (m < M) {
                    const float4 v = *reinterpret_cast<const float4 *>(slab + row * 64 + c4);
                    const size_t o = (size_t)m * ldc + bn + wc * 64 + c4;
                    const float4 hv = *reinterpret_cast<const float4 *>(H + o);
                    *reinterpret_cast<float4 *>(C + o) = make_float4(hv.x > 0.f ? v.x : 0.f, hv.y > 0.f ? v.y : 0.f,
                                                                    hv.z > 0.f ? v.z : 0.f, hv.w > 0.f ? v.w : 0.f);
                }
            }
        }
}

}  // namespace

extern "C" int zira_gemm_drelu_f32(const float *A, const float *B, const float *H, int M, int N, int K, float *C, void *stream)
{
    if (!A || !B || !H || !C || M <= 0 || N <= 0 || K <= 0 || N % kBN || K % kBK) return ZIRA_MSDA_EINVAL;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)H | (uintptr_t)C) & 15) return ZIRA_MSDA_EINVAL;
    const long long blocks = (long long)((M + kBM - 1) / kBM) * (N / kBN);
    if (blocks >= (1ll << 31)) return ZIRA_MSDA_EINVAL;
    hipLaunchKernelGGL(gemm_nn_drelu, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, A, B, H, M, N, K, K, N, N, C);
    return (int)hipGetLastError();
}
