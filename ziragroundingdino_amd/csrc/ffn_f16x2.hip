// ffn_f16x2.hip -- the FROZEN feed-forward block of the encoder layer as ONE launch per direction on the f16 matrix cores of
// gfx950 (MI355X), in fp32 accuracy:
//     forward    y  = relu(x W1^T + b1) W2^T + b2            (reference: models/GroundingDINO/transformer_for_adapter.py:877-886)
//     backward   gx = aux + ((gy W2) * [h > 0]) W1           (its autograd under the freeze of
//                                                              groundingdino_dual_zero_rep_branch.py:722-745: no weight gradients)
// Both are  out = epi( phi(A P^T) Q^T )  with A [M, 256], P [F, 256], Q [256, F]: forward P = W1, Q = W2, phi = + b1, ReLU (and
// the sign bits h > 0 are written, 1 bit per hidden unit: the [M, F] activation never exists in memory); backward P = W2^T,
// Q = W1^T, phi = those bits.  At the encoder shape (M = 44446, F = 2048) the library's four fp32 GEMMs move 1.8 GB and take
// 1.6 ms per layer; here a direction reads A, writes out, and streams 4 MB of packed weights per 128 rows from L2.
//
// Arithmetic.  v_mfma_f32_32x32x16_f16 is 16 x faster than the fp32 matrix instruction.  A row of fp32 numbers scaled by a
// power of two s so that its largest magnitude lies in [2^14, 2^15) is split as  s a = a1 + a2 + rest,  a1 = f16(s a),
// a2 = f16(s a - a1)  (both roundings to nearest, the difference exact): |rest| <= 2^-22 |s a| at worst, 2^-24 rms.  A product
// of two f16 numbers is exact in fp32, and of the four terms of (a1 + a2)(b1 + b2) the last is below 2^-22 |a b|: THREE
// matrix-core terms per fragment pair.  The matrix core sums a 32-deep slice (6 instructions) from zero and the vector unit adds
// the slice to the running fp32 sum (round to nearest; what the core does to its accumulator acts on a short sum only -- the
// lesson of gemm_bf16x3.hip).  Against an fp64 product the result is 2-4 x closer than the library's fp32 GEMM
// (tests/test_ffn_f16x2_gpu.py holds that gate on the model's shape; scripts/ffn_f16x2_accuracy.py is the arithmetic in numpy).
// Scales: one per row of A (a wave holds whole rows), one per row of P and of Q (made with the packed weights), one per
// (row, 32-wide slice) of the hidden tile -- all powers of two, applied exactly.
//
// Kernel.  A block = 128 rows = 4 PAIRS of waves, 32 rows per pair, both waves of a pair on one SIMD (8 waves, 256 registers
// each).  The FIRST wave of a pair keeps the two f16 planes of its 32 x 256 rows of A in registers as matrix-core B fragments
// (128 registers); per 32 hidden units ("step") it computes D = P_step A^T (48 instructions, one accumulator), applies phi and
// leaves the 32 x 32 hidden tile as fp32 in LDS (4 KB).  The SECOND wave keeps the pair's 32 x 256 tile of the output as 128
// fp32 registers; one step later it reads the hidden tile back lane for lane -- the accumulator layout of a 32 x 32 block is a
// valid B-fragment layout of the next product once the packed Q is permuted along its contraction index --, splits it into its
// two planes with the slice's own scale, and for each of the eight 32-column tiles of the output issues 6 instructions from
// zero and 16 v_fma into the running sum.  The two waves of a SIMD share its matrix pipe: one issues matrix instructions
// while the other is in its vector work (phi; split and the running sums), which the compiler's schedule of ONE wave
// holding both roles did not achieve (it also spilled: 128 + 128 registers of state plus fragments exceed the 512).
// The packed weights (P and Q fragments in exactly the order the waves read them, 1 KB per fragment = one ds_read_b128 per
// lane) stream L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, issued by the second waves) through two 64 KB stages, one step
// ahead, one barrier per step.  LDS: 2 x 64 KB + 2 x 16 KB of hidden tiles = all 160 KB of the CU.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kC = 256;                     // model width: contraction of the first product, columns of the second
constexpr int kRows = 128, kThreads = 512;  // rows per block: 4 pairs of waves, 32 rows per pair
constexpr int kFrag = 1024;                 // one fragment: 64 lanes x 8 halves
constexpr int kPBytes = 32 * kFrag;         // P fragments of one step: [p 0..7][t 0..1][plane 0..1]
constexpr int kQBytes = 32 * kFrag;         // Q fragments of one step: [tile 0..7][j 0..1][plane 0..1]
// The stream: P(0), then units s = 0 .. nsteps of [P(s + 1)][Q(s - 1)] (zeros where the index is outside 0 .. nsteps - 1):
// unit s is what the two roles read during step s, and one LDS stage.
constexpr int kUnit = kPBytes + kQBytes;
constexpr int kTBytes = 4096;               // a pair's 32 x 32 hidden tile in fp32
constexpr int kLdsT = 2 * kUnit;            // [buffer 0..1][pair 0..3][kTBytes] behind the two stages
constexpr int kLdsBytes = 2 * kUnit + 2 * 4 * kTBytes;   // 163840: all of the CU's LDS

enum { MODE_FWD = 0, MODE_BWD = 1 };

__device__ __forceinline__ unsigned pk_f16(float a, float b)
{
    f32x2 x = {a, b};
    f16x2 h = __builtin_convertvector(x, f16x2);   // round to nearest even
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float f16_lo(unsigned p)
{
    f16x2 h = __builtin_bit_cast(f16x2, p);
    return (float)h[0];
}
__device__ __forceinline__ float f16_hi(unsigned p)
{
    f16x2 h = __builtin_bit_cast(f16x2, p);
    return (float)h[1];
}

// the power of two that brings amax into [2^14, 2^15), and its reciprocal (exact); amax = 0 or tiny: 2^100; inf / NaN pass through
__device__ __forceinline__ void pow2_scale(float amax, float &s, float &inv)
{
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);   // amax in [2^(e-127), 2^(e-126))
    int se = 127 + 14 - (e - 127);                          // biased exponent of the scale
    se = se > 227 ? 227 : se;                               // <= 2^100: the reciprocal stays a normal number
    se = se < 1 ? 1 : se;
    s = __uint_as_float((unsigned)se << 23);
    inv = __uint_as_float((unsigned)(254 - se) << 23);
}

// max of the value in lanes l and l ^ 32, in both
__device__ __forceinline__ float pair_max(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// two fp32 -> one packed pair of each plane
__device__ __forceinline__ void split2(float x, float y, unsigned &p1, unsigned &p2)
{
    p1 = pk_f16(x, y);
    p2 = pk_f16(x - f16_lo(p1), y - f16_hi(p1));   // (exact differences)
}

// One wave copies 1 KB global -> LDS: lane l's 16 bytes from src + 16 l to LDS byte address dst + 16 l (src, dst wave-uniform).
// Inline assembly on purpose: issued through the builtin, the compiler orders every later ds_read behind the copy with
// s_waitcnt vmcnt(0) -- it cannot tell that the reads go to the OTHER stage -- and the stream would never run ahead of the
// matrix work.  The waves that issue these wait for them by hand (s_waitcnt vmcnt(0) before the step's barrier).
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

__device__ __forceinline__ f16x8 frag(const unsigned char *p) { return *reinterpret_cast<const f16x8 *>(p); }

// 4 NQ pieces of 1 KB from src to LDS byte address dst, piece q by the wave with part == q & 3
template <int NQ>
__device__ __forceinline__ void issue_pieces(const unsigned char *src, unsigned dst, int part, int lane)
{
    const unsigned voff = (unsigned)lane * 16u;
    src += part * kFrag;
    dst += part * kFrag;
#pragma unroll
    for (int q = 0; q < NQ; ++q) dma_1k(src + q * 4 * kFrag, voff, dst + q * 4 * kFrag);
}

template <int MODE>
__global__ __launch_bounds__(kThreads) void ffn_f16x2_kernel(const float *__restrict__ A, const unsigned char *__restrict__ stream,
                                                            int nsteps, const float *__restrict__ pbias,
                                                            const float *__restrict__ qinv,
                                                            const float *__restrict__ qbias, const float *aux, unsigned *mask,
                                                            float *out, int M, int nfull, int nsplit, float *partial,
                                                            unsigned *tickets)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave & 3;
    const int lm = lane & 31, hf = lane >> 5;
    // Blocks 0 .. nfull - 1 take 128 rows and all the steps.  The row blocks behind them -- the last, partly filled round of
    // the chip -- are cut into nsplit blocks each, every one with a share [s0, s1) of the steps; the shares' sums meet in
    // `partial` and the block that arrives last adds them in share order (host side: launch()).
    int rb = blockIdx.x, s0 = 0, s1 = nsteps, tail = -1, share = 0;
    if (rb >= nfull) {
        const int t = rb - nfull, per = nsteps / nsplit;
        tail = t / nsplit;
        share = t - tail * nsplit;
        rb = nfull + tail;
        s0 = share * per;
        s1 = s0 + per;
    }
    int m = rb * kRows + pair * 32 + lm;
    const bool live = m < M;
    m = live ? m : M - 1;            // (rows past the end compute on the last row; nothing of theirs is stored)
    const unsigned char *const lbase = smem + lane * 16;
    unsigned char *const tbase = smem + kLdsT + pair * kTBytes + lane * 16;   // + buffer * 4 * kTBytes + g * 1024

#if defined(ZIRA_FFN_PRIO) && ZIRA_FFN_PRIO == 1
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
#elif defined(ZIRA_FFN_PRIO) && ZIRA_FFN_PRIO == 2
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    if (wave < 4) {
        // ================= first wave of the pair: D = P_step A^T, phi, the hidden tile to LDS =================================
        // lane (lm, hf) holds columns 32 p + 16 hf .. + 15, p = 0 .. 7, of row m
        f16x8 a1[8][2], a2[8][2];
        float a_inv;
        {
            float4 v[8][4];
            const float *ar = A + (size_t)m * kC + 16 * hf;
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[p][q] = *reinterpret_cast<const float4 *>(ar + 32 * p + 4 * q);
            float amax = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[p][q].x), fabsf(v[p][q].y))), fmaxf(fabsf(v[p][q].z), fabsf(v[p][q].w)));
            amax = pair_max(amax);
            float a_s;
            pow2_scale(amax, a_s, a_inv);
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 lo = v[p][2 * t], hi = v[p][2 * t + 1];
                    unsigned x1[4], x2[4];
                    split2(lo.x * a_s, lo.y * a_s, x1[0], x2[0]);
                    split2(lo.z * a_s, lo.w * a_s, x1[1], x2[1]);
                    split2(hi.x * a_s, hi.y * a_s, x1[2], x2[2]);
                    split2(hi.z * a_s, hi.w * a_s, x1[3], x2[3]);
                    a1[p][t] = __builtin_bit_cast(f16x8, make_uint4(x1[0], x1[1], x1[2], x1[3]));
                    a2[p][t] = __builtin_bit_cast(f16x8, make_uint4(x2[0], x2[1], x2[2], x2[3]));
                }
        }
        // the sign bits: [m][hf][step] 16 bits each (F / 8 bytes per row), 128 bits per lane and 8 steps
        uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
        unsigned *const mrow = mask + (size_t)m * nsteps + hf * (nsteps >> 1);
        if (MODE == MODE_BWD) mbits = *reinterpret_cast<const uint4 *>(mrow + (s0 >> 3) * 4);
        // the bias of a step (times the scale of its row of P; forward only): register 4 g + e <-> hidden unit 8 g + 4 hf + e;
        // each quarter is reloaded for the next step right behind its last use
        float4 cb[4];
        const float *const brow = pbias + 4 * hf;
        if (MODE == MODE_FWD) {
#pragma unroll
            for (int g = 0; g < 4; ++g) cb[g] = *reinterpret_cast<const float4 *>(brow + 32 * s0 + 8 * g);
        }

        // The matrix core's A operand is the weight fragment (rows = hidden units), its B operand the rows of A.  Group
        // g = 2 p + t of a step: two fragments (plane 1, plane 2), three instructions; fragments are read two groups ahead.
#define ZIRA_FFN_PFRAG(PB_, G_, W1_, W2_)                                     \
    do {                                                                      \
        W1_ = frag((PB_) + ((G_) * 2 + 0) * kFrag);                           \
        W2_ = frag((PB_) + ((G_) * 2 + 1) * kFrag);                           \
    } while (0)
#define ZIRA_FFN_MFMA3(DST_, G_, W1_, W2_)                                                                          \
    do {                                                                                                            \
        DST_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2_, a1[(G_) >> 1][(G_) & 1], DST_, 0, 0, 0);                \
        DST_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1_, a2[(G_) >> 1][(G_) & 1], DST_, 0, 0, 0);                \
        DST_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1_, a1[(G_) >> 1][(G_) & 1], DST_, 0, 0, 0);                \
    } while (0)

        __syncthreads();                 // barrier 0: P(0) has landed in the second stage
        f32x16 d, dn;
        {
            const unsigned char *pb = lbase + kUnit;
            f16x8 w1[3], w2[3];
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            ZIRA_FFN_PFRAG(pb, 0, w1[0], w2[0]);
            ZIRA_FFN_PFRAG(pb, 1, w1[1], w2[1]);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g + 2 < 16) ZIRA_FFN_PFRAG(pb, g + 2, w1[(g + 2) % 3], w2[(g + 2) % 3]);
                ZIRA_FFN_MFMA3(d, g, w1[g % 3], w2[g % 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                 // barrier 1: everyone has read P(0)

        // step S: the first product of step S + 1 (P(S + 1) in stage ST) into DNEXT, and beside each of its 16 groups phi of
        // one register of DCUR (step S): scale, bias, ReLU and its sign bit / the saved bit
#define ZIRA_FFN_STEP1(S, ST, DCUR, DNEXT)                                                                                        \
    do {                                                                                                                          \
        const int s_ = (S);                                                                                                       \
        const int sn_ = s_ + 1 < nsteps ? s_ + 1 : s_;                                                                            \
        const unsigned char *pb = lbase + (ST) * kUnit;                                                                           \
        f16x8 w1[3], w2[3];                                                                                                       \
        float tv[4];                                                                                                              \
        unsigned word = MODE == MODE_BWD ? (mbits.x & 0xFFFFu) : 0u;                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) DNEXT[r] = 0.f;                                                            \
        ZIRA_FFN_PFRAG(pb, 0, w1[0], w2[0]);                                                                                      \
        ZIRA_FFN_PFRAG(pb, 1, w1[1], w2[1]);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                                          \
            if (g + 2 < 16) ZIRA_FFN_PFRAG(pb, g + 2, w1[(g + 2) % 3], w2[(g + 2) % 3]);                                          \
            ZIRA_FFN_MFMA3(DNEXT, g, w1[g % 3], w2[g % 3]);                                                                       \
            if (MODE == MODE_FWD) {                                                                                               \
                const float4 b4 = cb[g >> 2];                                                                                     \
                const float bg = (g & 3) == 0 ? b4.x : ((g & 3) == 1 ? b4.y : ((g & 3) == 2 ? b4.z : b4.w));                      \
                const float t = fmaxf(fmaf(DCUR[g], a_inv, bg), 0.f);                                                             \
                tv[g & 3] = t;                                                                                                    \
                word |= ((__float_as_uint(t) & 0x7FFFFFFFu) ? 1u : 0u) << g;                                                      \
            } else {                                                                                                              \
                tv[g & 3] = ((word >> g) & 1u) ? DCUR[g] * a_inv : 0.f;                                                           \
            }                                                                                                                     \
            if ((g & 3) == 3) {                                                                                                   \
                *reinterpret_cast<float4 *>(tbase + (s_ & 1) * 4 * kTBytes + (g >> 2) * 1024) = make_float4(tv[0], tv[1], tv[2], tv[3]); \
                if (MODE == MODE_FWD) cb[g >> 2] = *reinterpret_cast<const float4 *>(brow + 32 * sn_ + 8 * (g >> 2));            \
            }                                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
        }                                                                                                                         \
        /* the bits move on by one step: forward shifts the new word in at the top, backward shifts the used one out */          \
        mbits.x = __builtin_amdgcn_alignbit(mbits.y, mbits.x, 16);                                                                \
        mbits.y = __builtin_amdgcn_alignbit(mbits.z, mbits.y, 16);                                                                \
        mbits.z = __builtin_amdgcn_alignbit(mbits.w, mbits.z, 16);                                                                \
        mbits.w = __builtin_amdgcn_alignbit(MODE == MODE_FWD ? word : 0u, mbits.w, 16);                                           \
        if (MODE == MODE_FWD && (s_ & 7) == 7 && live) *reinterpret_cast<uint4 *>(mrow + (s_ >> 3) * 4) = mbits;                  \
        if (MODE == MODE_BWD && (s_ & 7) == 7 && s_ + 1 < s1) mbits = *reinterpret_cast<const uint4 *>(mrow + ((s_ + 1) >> 3) * 4); \
        __syncthreads();                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);   /* (nothing of a step moves into the next) */                                        \
    } while (0)

        for (int s = s0; s < s1; s += 2) {   // (s0 and s1 are multiples of 8)
            ZIRA_FFN_STEP1(s, 0, d, dn);
            ZIRA_FFN_STEP1(s + 1, 1, dn, d);
        }
        __syncthreads();                 // step s1: the second waves' last
#undef ZIRA_FFN_STEP1
#undef ZIRA_FFN_MFMA3
#undef ZIRA_FFN_PFRAG
        if (tail < 0) return;
        // (a block with a share of the steps: the first waves take part in the hand-over's two barriers)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            // every storing wave has drained its stores and passed the barrier: release them, then draw the ticket
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned old = __hip_atomic_fetch_add(tickets + tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(nsplit - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(tickets + tail, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            *reinterpret_cast<volatile int *>(smem) = last;
        }
        __syncthreads();
        return;
    }

    // ===================== second wave of the pair: the stream, the second product, the output tile ===============================
    const int part = wave - 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;   // LDS byte address of the first stage
    const unsigned char *const units = stream + kPBytes;                                    // unit s at units + s * kUnit
    issue_pieces<8>(s0 == 0 ? stream : units + (size_t)(s0 - 1) * kUnit, lds0 + kUnit, part, lane);   // P(s0) -> second stage
    issue_pieces<16>(units + (size_t)s0 * kUnit, lds0, part, lane);                        // unit s0 -> first stage
    float run[8][16];   // (plain floats: as 16-wide vectors each tile needs 16 CONTIGUOUS registers and the allocator spills whole tiles)
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
        for (int r = 0; r < 16; ++r) run[t8][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // barrier 0
    __builtin_amdgcn_s_barrier();        // barrier 1
    issue_pieces<16>(units + (size_t)(s0 + 1) * kUnit, lds0 + kUnit, part, lane);           // step s0: unit s0 + 1 -> second stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#define ZIRA_FFN_PIN16(X)                                                                                                         \
    asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]), "+v"(X[8]), "+v"(X[9]), \
                      "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]))
    // step S >= 1: the hidden tile of step S - 1 (buffer (S - 1) & 1) times Q(S - 1) (stage ST = S & 1)
#define ZIRA_FFN_STEP2(S, ST)                                                                                                     \
    do {                                                                                                                          \
        const int s_ = (S);                                                                                                       \
        if (s_ + 1 <= s1) issue_pieces<16>(units + (size_t)(s_ + 1) * kUnit, lds0 + ((ST) ^ 1) * kUnit, part, lane);               \
        float tv[16];                                                                                                             \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                           \
            const float4 x = *reinterpret_cast<const float4 *>(tbase + ((ST) ^ 1) * 4 * kTBytes + g * 1024);                      \
            tv[4 * g] = x.x; tv[4 * g + 1] = x.y; tv[4 * g + 2] = x.z; tv[4 * g + 3] = x.w;                                       \
        }                                                                                                                         \
        float tmax = 0.f;                                                                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, fabsf(tv[r]));                                          \
        tmax = pair_max(tmax);                                                                                                    \
        float t_s, t_inv;                                                                                                         \
        pow2_scale(tmax, t_s, t_inv);                                                                                             \
        f16x8 t1[2], t2[2];                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                           \
            unsigned x1[4], x2[4];                                                                                                \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) split2(tv[8 * j + 2 * q] * t_s, tv[8 * j + 2 * q + 1] * t_s, x1[q], x2[q]); \
            t1[j] = __builtin_bit_cast(f16x8, make_uint4(x1[0], x1[1], x1[2], x1[3]));                                            \
            t2[j] = __builtin_bit_cast(f16x8, make_uint4(x2[0], x2[1], x2[2], x2[3]));                                            \
        }                                                                                                                         \
        const unsigned char *qb = lbase + (ST) * kUnit + kPBytes;                                                                 \
        f16x8 q1[3], q2[3];   /* half tiles u = 2 t8 + j, read two ahead */                                                    \
        f32x16 c[2];          /* a tile's slice; the running sums of tile t8 are updated beside the first instructions of t8 + 1 */ \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                           \
            q1[u] = frag(qb + (u * 2 + 0) * kFrag);                                                                               \
            q2[u] = frag(qb + (u * 2 + 1) * kFrag);                                                                               \
        }                                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                                        \
        _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                                                          \
            const int t8 = u >> 1, j = u & 1;                                                                                     \
            if (u + 2 < 16) {                                                                                                     \
                q1[(u + 2) % 3] = frag(qb + ((u + 2) * 2 + 0) * kFrag);                                                           \
                q2[(u + 2) % 3] = frag(qb + ((u + 2) * 2 + 1) * kFrag);                                                           \
            }                                                                                                                     \
            if (j == 0) { _Pragma("unroll") for (int r = 0; r < 16; ++r) c[t8 & 1][r] = 0.f; }                                    \
            c[t8 & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q2[u % 3], t1[j], c[t8 & 1], 0, 0, 0);                             \
            c[t8 & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1[u % 3], t2[j], c[t8 & 1], 0, 0, 0);                             \
            c[t8 & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1[u % 3], t1[j], c[t8 & 1], 0, 0, 0);                             \
            if (j == 0 && t8 > 0) {                                                                                               \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) run[t8 - 1][r] = fmaf(c[(t8 - 1) & 1][r], t_inv, run[t8 - 1][r]);  \
                ZIRA_FFN_PIN16(run[t8 - 1]);   /* (or instruction selection orders the sums of all eight tiles behind the last   \
                                                  matrix instruction and every tile's slice stays live: 112 registers, spilled) */ \
            }                                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                                    \
        }                                                                                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) run[7][r] = fmaf(c[1][r], t_inv, run[7][r]);                               \
        ZIRA_FFN_PIN16(run[7]);                                                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                               \
        __builtin_amdgcn_s_barrier();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);   /* (nothing of a step moves into the next) */                                        \
    } while (0)

    for (int s = s0 + 1; s <= s1; s += 2) {
        ZIRA_FFN_STEP2(s, 1);
        ZIRA_FFN_STEP2(s + 1, 0);
    }
#undef ZIRA_FFN_STEP2
#undef ZIRA_FFN_PIN16

    if (tail >= 0) {
        // ---- a share of the steps: the sums to `partial`, lane for lane; the block that arrives last adds the shares in order ----
        float *const mine = partial + ((size_t)(tail * nsplit + share) * 4 + pair) * (32 * 64 * 4) + lane * 4;
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(mine + (t8 * 4 + g) * 256) = make_float4(run[t8][4 * g], run[t8][4 * g + 1], run[t8][4 * g + 2], run[t8][4 * g + 3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __syncthreads();                 // (thread 0 has drawn the ticket)
        if (*reinterpret_cast<volatile int *>(smem) == 0) return;
        const float *const first = partial + ((size_t)(tail * nsplit) * 4 + pair) * (32 * 64 * 4) + lane * 4;
#pragma unroll
        for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 acc = *reinterpret_cast<const float4 *>(first + (t8 * 4 + g) * 256);
                for (int sh = 1; sh < nsplit; ++sh) {
                    const float4 x = *reinterpret_cast<const float4 *>(first + (size_t)sh * 4 * (32 * 64 * 4) + (t8 * 4 + g) * 256);
                    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                }
                run[t8][4 * g] = acc.x; run[t8][4 * g + 1] = acc.y; run[t8][4 * g + 2] = acc.z; run[t8][4 * g + 3] = acc.w;
            }
    }

    // ---- out: register 4 g + e of tile t8 is column 32 t8 + 8 g + 4 hf + e of row m -----------------------------------------------
    if (!live) return;
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 32 * t8 + 8 * g + 4 * hf;
            const float4 qi = *reinterpret_cast<const float4 *>(qinv + n);
            float4 o = make_float4(run[t8][4 * g] * qi.x, run[t8][4 * g + 1] * qi.y, run[t8][4 * g + 2] * qi.z, run[t8][4 * g + 3] * qi.w);
            if (qbias) {
                const float4 b = *reinterpret_cast<const float4 *>(qbias + n);
                o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
            }
            const size_t at = (size_t)m * kC + n;
            if (aux) {
                const float4 h = *reinterpret_cast<const float4 *>(aux + at);
                o.x += h.x; o.y += h.y; o.z += h.z; o.w += h.w;
            }
            *reinterpret_cast<float4 *>(out + at) = o;
        }
}

// ---- packing the frozen weights (once per weight version) ---------------------------------------------------------------------------

// scale[r] = the power of two that brings max_c |w[r][c] colmul[c]| into [2^14, 2^15) (colmul may be null); inv[r] its reciprocal
__global__ __launch_bounds__(256) void row_scale_kernel(const float *__restrict__ w, long long row_stride, long long col_stride, int cols,
                                                        const float *__restrict__ colmul, float *__restrict__ scale, float *__restrict__ inv)
{
    __shared__ float red[256];
    const int r = blockIdx.x;
    float amax = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) amax = fmaxf(amax, fabsf(w[r * row_stride + c * col_stride] * (colmul ? colmul[c] : 1.f)));
    red[threadIdx.x] = amax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float s, i;
        pow2_scale(red[0], s, i);
        scale[r] = s;
        inv[r] = i;
    }
}

__global__ __launch_bounds__(256) void scaled_bias_kernel(const float *__restrict__ bias, const float *__restrict__ pscale, int F, float *__restrict__ out)
{
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h < F) out[h] = bias ? bias[h] * pscale[h] : 0.f;
}

__device__ __forceinline__ unsigned short plane_of(float x, int plane)
{
    const unsigned p1 = pk_f16(x, 0.f);
    if (plane == 0) return (unsigned short)(p1 & 0xFFFFu);
    return (unsigned short)(pk_f16(x - f16_lo(p1), 0.f) & 0xFFFFu);
}

// One thread per 16-byte lane piece of the stream.  P(h, k) = p[h * psh + k * psk], Q(n, h) = q[n * qsn + h * qsh].
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ p, long long psh, long long psk, const float *__restrict__ q,
                                                   long long qsn, long long qsh, int F, const float *__restrict__ pscale,
                                                   const float *__restrict__ pinv, const float *__restrict__ qscale,
                                                   unsigned char *__restrict__ stream)
{
    const int nsteps = F / 32;
    const long long pieces = ((long long)kPBytes + (long long)(nsteps + 1) * kUnit) / 16;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < pieces; id += (long long)gridDim.x * blockDim.x) {
        const long long off = id * 16;
        int c, within;
        bool is_p;
        if (off < kPBytes) {
            c = 0; within = (int)off; is_p = true;
        } else {
            const int u = (int)((off - kPBytes) / kUnit);
            within = (int)((off - kPBytes) % kUnit);
            is_p = within < kPBytes;
            c = is_p ? u + 1 : u - 1;
            if (!is_p) within -= kPBytes;
        }
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (c >= 0 && c < nsteps) {
            const int f = within / kFrag, lane = (within % kFrag) / 16, lm = lane & 31, hf = lane >> 5;
            unsigned short e8[8];
            if (is_p) {            // fragment ((pp * 2 + t) * 2 + plane), lane: P[32 c + lm][32 pp + 16 hf + 8 t + e]
                const int plane = f & 1, t = (f >> 1) & 1, pp = f >> 2;
                const int h = 32 * c + lm;
                for (int e = 0; e < 8; ++e) e8[e] = plane_of(p[h * psh + (32 * pp + 16 * hf + 8 * t + e) * psk] * pscale[h], plane);
            } else {               // fragment ((t8 * 2 + j) * 2 + plane), lane: Q[32 t8 + lm][32 c + 16 j + 8 (e / 4) + 4 hf + e % 4]
                const int plane = f & 1, j = (f >> 1) & 1, t8 = f >> 2;
                const int n = 32 * t8 + lm;
                for (int e = 0; e < 8; ++e) {       // (the hidden unit's 1 / scale of P rides on Q's column)
                    const int h = 32 * c + 16 * j + 8 * (e >> 2) + 4 * hf + (e & 3);
                    e8[e] = plane_of(q[n * qsn + h * qsh] * pinv[h] * qscale[n], plane);
                }
            }
            v = make_uint4(e8[0] | (unsigned)e8[1] << 16, e8[2] | (unsigned)e8[3] << 16, e8[4] | (unsigned)e8[5] << 16, e8[6] | (unsigned)e8[7] << 16);
        }
        *reinterpret_cast<uint4 *>(stream + (size_t)off) = v;
    }
}

__host__ __device__ inline size_t stream_bytes(int F) { return (size_t)kPBytes + (size_t)(F / 32 + 1) * kUnit; }

// How the row blocks are dealt: `nfull` whole blocks (complete rounds of the chip's CUs), and each row block of the last,
// partly filled round cut into `nsplit` shares of the steps so that the round fills (and takes 1 / nsplit of the time).
struct Deal { int blocks, nfull, nsplit, ntail; };
Deal deal(int M, int nsteps)
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int rbs = (M + kRows - 1) / kRows, rest = rbs % cus;
    int nsplit = 1;
    while (rest > 0 && nsplit < 4 && rest * nsplit * 2 <= cus && (nsteps / (nsplit * 2)) % 8 == 0) nsplit *= 2;
    Deal d;
    d.nsplit = nsplit;
    d.ntail = nsplit > 1 ? rest : 0;
    d.nfull = rbs - d.ntail;
    d.blocks = d.nfull + d.ntail * nsplit;
    return d;
}

size_t workspace_bytes(int M, int nsteps)
{
    const Deal d = deal(M, nsteps);
    return (size_t)d.ntail * 256 + (size_t)d.ntail * d.nsplit * kRows * kC * sizeof(float);   // tickets (zeroed once), then the shares' sums
}

template <int MODE>
int launch(const float *a, const unsigned char *stream, int nsteps, const float *pbias, const float *qinv, const float *qbias, const float *aux,
           unsigned *mask, float *out, int M, void *workspace, hipStream_t st)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ffn_f16x2_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    Deal d = deal(M, nsteps);
    if (!workspace && d.ntail) {   // no scratch: whole blocks only
        d.nfull += d.ntail; d.blocks = d.nfull; d.ntail = 0; d.nsplit = 1;
    }
    unsigned *tickets = reinterpret_cast<unsigned *>(workspace);
    float *partial = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(workspace) + (size_t)d.ntail * 256);
    hipLaunchKernelGGL((ffn_f16x2_kernel<MODE>), dim3(d.blocks), dim3(kThreads), kLdsBytes, st, a, stream, nsteps, pbias, qinv, qbias, aux, mask, out, M,
                       d.nfull, d.nsplit, partial, tickets);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t zira_ffn_f16x2_pack_bytes(int F)
{
    if (F <= 0 || F % 256) return 0;
    // the stream + [1 / scale, scale] of Q's 256 rows + [scale, 1 / scale, scaled bias] of P's F rows
    return stream_bytes(F) + (size_t)kC * 2 * sizeof(float) + (size_t)F * 3 * sizeof(float);
}

extern "C" int zira_ffn_f16x2_pack_f32(const float *p, long long p_row_stride, long long p_col_stride, const float *q, long long q_row_stride,
                                       long long q_col_stride, const float *p_bias, int F, void *packed, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!p || !q || !packed || F <= 0 || F % 256 || ((uintptr_t)packed & 15)) return -1;
    unsigned char *base = reinterpret_cast<unsigned char *>(packed);
    float *qinv = reinterpret_cast<float *>(base + stream_bytes(F)), *qscale = qinv + kC, *pscale = qscale + kC, *pinv = pscale + F, *pb = pinv + F;
    hipLaunchKernelGGL(row_scale_kernel, dim3(F), dim3(256), 0, stream, p, p_row_stride, p_col_stride, kC, (const float *)nullptr, pscale, pinv);
    hipLaunchKernelGGL(row_scale_kernel, dim3(kC), dim3(256), 0, stream, q, q_row_stride, q_col_stride, F, (const float *)pinv, qscale, qinv);
    hipLaunchKernelGGL(scaled_bias_kernel, dim3((F + 255) / 256), dim3(256), 0, stream, p_bias, (const float *)pscale, F, pb);
    const long long pieces = (long long)(stream_bytes(F) / 16);
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, p, p_row_stride, p_col_stride, q, q_row_stride,
                       q_col_stride, F, (const float *)pscale, (const float *)pinv, (const float *)qscale, base);
    return (int)hipGetLastError();
}

extern "C" size_t zira_ffn_f16x2_workspace_bytes(int M, int F)
{
    if (M <= 0 || F <= 0 || F % 256) return 0;
    return workspace_bytes(M, F / 32);
}

extern "C" int zira_ffn_f16x2_f32(const float *a, const void *packed, int M, int F, int backward, const float *q_bias, const float *aux,
                                  void *mask, float *out, void *workspace, void *stream_)
{
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!a || !packed || !mask || !out || M <= 0 || F <= 0 || F % 256) return -1;
    if (((uintptr_t)a | (uintptr_t)packed | (uintptr_t)mask | (uintptr_t)out | (uintptr_t)q_bias | (uintptr_t)aux | (uintptr_t)workspace) & 15) return -1;
    const unsigned char *base = reinterpret_cast<const unsigned char *>(packed);
    const float *qinv = reinterpret_cast<const float *>(base + stream_bytes(F)), *pb = qinv + 2 * kC + 2 * F;
    if (backward) return launch<MODE_BWD>(a, base, F / 32, pb, qinv, q_bias, aux, reinterpret_cast<unsigned *>(mask), out, M, workspace, stream);
    return launch<MODE_FWD>(a, base, F / 32, pb, qinv, q_bias, aux, reinterpret_cast<unsigned *>(mask), out, M, workspace, stream);
}
