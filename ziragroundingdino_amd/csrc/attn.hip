// attn.hip -- softmax(Q K^T * scale + key mask) V for the small fp32 attentions of the cross-modal decoder
// (self-attention over the 900 queries, cross-attention to the <= 256 text tokens; reference
// transformer_for_adapter.py:1043-1058 through nn.MultiheadAttention), forward and backward, head width 32.
//
// Why.  As two batched GEMMs and a softmax the decoder's self-attention core takes 82 us forward and 237 us with its
// backward per layer on MI355X (six score-sized tensors of 52 MB each go through HBM, the batched 900 x 32 x 900
// products run at 26 TF/s); the arithmetic is 1.7 + 4.1 GFLOP.  Here the scores never leave registers.
//
// Everything is built on `v_mfma_f32_32x32x2_f32` (exact fp32 products, fp32 accumulation: the numerics of an fmaf
// chain) with the operand ORIENTATION chosen so that no tile is ever transposed or moved between lanes:
//   * a 32 x 32 result has its column on the lane (lane & 31) and 16 of its rows in the lane's registers
//     (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)); at contraction step t the two lane halves supply the two
//     k-indices, and WHICH two is free as long as both operands agree -- so half h uses feature 16 h + t when the
//     contraction runs over the 32 features (a lane loads a contiguous half row), and row (t & 3) + 8 (t >> 2) + 4 h
//     of a previous result when it runs over that result's rows: register t of the previous accumulator IS the
//     operand, untouched.
//   * forward and dQ: S^T = K Q^T (keys x queries): a query's scores sit in ONE lane (two, with the other half), so
//     the softmax statistics are per-lane scalars; O^T += V^T P^T and dQ^T += K^T dS^T sum over S^T's rows.
//   * dK / dV: S = Q K^T (queries x keys): dV^T += dO^T P and dK^T += Q^T dS sum over S's rows.
// Forward: a wave owns 32 queries and walks the key tiles with the running max / sum of a flash attention; with
// many keys the four waves of a block split them and merge through LDS.  Backward: two kernels that recompute the
// probabilities from the saved log-sum-exp -- one block per query tile for dQ (it also forms delta = <dO, O>),
// one block per key tile for dK and dV; the four waves split the other dimension and add their partial tiles in
// LDS.  No atomics, no workspace beyond lse and delta ([B, H, L] floats each).
//
// Where the time goes (self-attention of 900 queries, 2 images x 8 heads: 464 blocks, two per CU on most CUs): the products
// alone are 13 / 20 / 26 us of MFMA issue (forward / dQ / dK+dV at 64 flop per cycle and SIMD) against 34.6 / 42.9 / 55.8 us
// measured -- 38-47 % of the fp32 MFMA rate.  Requesting the next tile's operands a round ahead (register double buffer) was
// measured and dropped: 33.5 / 47.0 / 74.0 us -- the dK+dV kernel then needs 288 registers, one wave per SIMD, and the CU's
// second block no longer runs beside the first, which is what had been hiding the loads.
//
// Layouts: q / k / v element (l, b, h, c) at ptr[(l * B + b) * ld + h * 32 + c] (ld = row stride in floats: the
// projections' outputs and slices of fused projections are used as they are); out, dq, dk, dv contiguous [L|S, B, H*32].
// `kpm`: optional additive key mask [B, S] (0 or -inf).  A query whose keys are all masked gets a zero row.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kD = 32;

__device__ __forceinline__ unsigned rowmap(unsigned reg, unsigned hh) { return (reg & 3u) + 8u * (reg >> 2) + 4u * hh; }
__device__ __forceinline__ float xhalf(float x) { return __shfl_xor(x, 32); }

struct AttnDims {
    int L, S, B, H;
    int ldq, ldk, ldv;
    float scale;
    int lddq, lddk, lddv;   // row strides of the gradients (the backward's outputs)
};

// 16 floats: features 16 hh .. 16 hh + 15 of row `row` of a [rows, B, ld] tensor, head h, batch b
__device__ __forceinline__ void load_half_row(const float *__restrict__ p, int row, int b, int B, int ld, int h, unsigned hh,
                                              float (&x)[16])
{
    const float4 *src = reinterpret_cast<const float4 *>(p + ((size_t)row * B + b) * ld + h * kD + 16 * hh);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = src[i];
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
}
// x[t] = element `col` of row base + rowmap(t, hh) (clamped to nrows - 1)
__device__ __forceinline__ void load_column(const float *__restrict__ p, int base, int nrows, int b, int B, int ld, int h,
                                            unsigned hh, unsigned col, float (&x)[16])
{
#pragma unroll
    for (unsigned t = 0; t < 16; ++t) {
        int row = base + (int)rowmap(t, hh);
        row = row < nrows ? row : nrows - 1;
        x[t] = p[((size_t)row * B + b) * ld + h * kD + col];
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int KSPLIT, bool MASK>
__global__ __launch_bounds__(256) void attn_fwd(const float *__restrict__ q, const float *__restrict__ k,
                                                const float *__restrict__ v, const float *__restrict__ kpm, AttnDims A,
                                                float *__restrict__ out, float *__restrict__ lse)
{
    __shared__ float sm[4][32], sl[4][32];
    __shared__ float so[32][33];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / A.H, h = bh - b * A.H;
    const int qt = KSPLIT == 4 ? (int)blockIdx.x : (int)blockIdx.x * 4 + (int)wave;
    const int q0 = qt * 32;
    if (KSPLIT == 1 && q0 >= A.L) return;   // wave-uniform
    const int nkt = (A.S + 31) / 32;

    float bq[16];
    {
        const int qr = q0 + (int)r < A.L ? q0 + (int)r : A.L - 1;
        load_half_row(q, qr, b, A.B, A.ldq, h, hh, bq);
#pragma unroll
        for (int t = 0; t < 16; ++t) bq[t] *= A.scale;
    }
    float m = -INFINITY, lsum = 0.f;
    f32x16 acc_o;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[i] = 0.f;

    for (int kt = KSPLIT == 4 ? (int)wave : 0; kt < nkt; kt += KSPLIT == 4 ? 4 : 1) {
        const int k0 = kt * 32;
        float ka[16], va[16];
        {
            const int kr = k0 + (int)r < A.S ? k0 + (int)r : A.S - 1;
            load_half_row(k, kr, b, A.B, A.ldk, h, hh, ka);
        }
        load_column(v, k0, A.S, b, A.B, A.ldv, h, hh, r, va);
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], bq[t], s, 0, 0, 0);
        // s[reg] = S^T[key k0 + rowmap(reg, hh)][query q0 + r]
        float mt = -INFINITY;
#pragma unroll
        for (unsigned reg = 0; reg < 16; ++reg) {
            const int kidx = k0 + (int)rowmap(reg, hh);
            float x = s[reg];
            if (kidx >= A.S) x = -INFINITY;
            else if (MASK) x += kpm[(size_t)b * A.S + kidx];
            s[reg] = x;
            mt = fmaxf(mt, x);
        }
        mt = fmaxf(mt, xhalf(mt));
        const float m_new = fmaxf(m, mt);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;   // (every key so far masked: all p = 0)
        const float alpha = __expf(m - m_use);                  // (m = -inf: 0)
        float ps = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float p = __expf(s[reg] - m_use);
            s[reg] = p;
            ps += p;
        }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[i] *= alpha;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(va[t], s[t], acc_o, 0, 0, 0);
        // acc_o[reg] = O^T[feature rowmap(reg, hh)][query q0 + r]
        m = m_new;
    }
    lsum += xhalf(lsum);

    if (KSPLIT == 1) {
        const int qi = q0 + (int)r;
        if (qi < A.L) {
            const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
            float *o = out + ((size_t)qi * A.B + b) * (A.H * kD) + h * kD;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(o + 8 * g + 4 * hh) =
                    make_float4(acc_o[4 * g] * inv, acc_o[4 * g + 1] * inv, acc_o[4 * g + 2] * inv, acc_o[4 * g + 3] * inv);
            if (hh == 0) lse[(size_t)bh * A.L + qi] = lsum > 0.f ? m + __logf(lsum) : -INFINITY;
        }
        return;
    }
    // merge the four key ranges
    if (hh == 0) {
        sm[wave][r] = m;
        sl[wave][r] = lsum;
    }
    for (unsigned w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (unsigned reg = 0; reg < 16; ++reg) {
                const unsigned dd = rowmap(reg, hh);
                // scale by exp(m_w - M) on the fly: M needs all four maxima, which are in LDS after the first barrier
                const float M = fmaxf(fmaxf(sm[0][r], sm[1][r]), fmaxf(sm[2][r], sm[3][r]));
                const float f = m == -INFINITY ? 0.f : __expf(m - M);
                if (w == 0) so[r][dd] = acc_o[reg] * f;
                else so[r][dd] += acc_o[reg] * f;
            }
        }
    }
    __syncthreads();
    {
        const unsigned qq = tid >> 3, d4 = tid & 7;
        const int qi = q0 + (int)qq;
        if (qi < A.L) {
            const float M = fmaxf(fmaxf(sm[0][qq], sm[1][qq]), fmaxf(sm[2][qq], sm[3][qq]));
            float Lsum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) Lsum += sm[w][qq] == -INFINITY ? 0.f : sl[w][qq] * __expf(sm[w][qq] - M);
            const float inv = Lsum > 0.f ? 1.f / Lsum : 0.f;
            const float *src = &so[qq][d4 * 4];
            *reinterpret_cast<float4 *>(out + ((size_t)qi * A.B + b) * (A.H * kD) + h * kD + d4 * 4) =
                make_float4(src[0] * inv, src[1] * inv, src[2] * inv, src[3] * inv);
            if (d4 == 0) lse[(size_t)bh * A.L + qi] = Lsum > 0.f ? M + __logf(Lsum) : -INFINITY;
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, dQ (and delta): a block per 32 queries, the waves split the key tiles
// ------------------------------------------------------------------------------------------
template <bool MASK>
__global__ __launch_bounds__(256) void attn_bwd_dq(const float *__restrict__ q, const float *__restrict__ k,
                                                   const float *__restrict__ v, const float *__restrict__ kpm,
                                                   const float *__restrict__ out, const float *__restrict__ dout,
                                                   const float *__restrict__ lse, AttnDims A, float *__restrict__ dq,
                                                   float *__restrict__ delta)
{
    __shared__ float so[32][33];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / A.H, h = bh - b * A.H;
    const int q0 = (int)blockIdx.x * 32;
    const int nkt = (A.S + 31) / 32;
    const int E = A.H * kD;
    const int qr = q0 + (int)r < A.L ? q0 + (int)r : A.L - 1;
    float bq[16], bdo[16];
    load_half_row(q, qr, b, A.B, A.ldq, h, hh, bq);
    load_half_row(dout, qr, b, A.B, E, h, hh, bdo);
    float dl;
    {
        float o[16];
        load_half_row(out, qr, b, A.B, E, h, hh, o);
        dl = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) dl = fmaf(bdo[t], o[t], dl);
        dl += xhalf(dl);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) bq[t] *= A.scale;
    const float lq = lse[(size_t)bh * A.L + qr];
    if (wave == 0 && hh == 0 && q0 + (int)r < A.L) delta[(size_t)bh * A.L + q0 + r] = dl;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int kt = (int)wave; kt < nkt; kt += 4) {
        const int k0 = kt * 32;
        const int kr = k0 + (int)r < A.S ? k0 + (int)r : A.S - 1;
        float ka[16], vr[16], kc[16];
        load_half_row(k, kr, b, A.B, A.ldk, h, hh, ka);
        load_half_row(v, kr, b, A.B, A.ldv, h, hh, vr);
        load_column(k, k0, A.S, b, A.B, A.ldk, h, hh, r, kc);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], bq[t], s, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[t], bdo[t], dp, 0, 0, 0);
        // s / dp [reg] = S^T / dP^T [key k0 + rowmap(reg, hh)][query q0 + r]
#pragma unroll
        for (unsigned reg = 0; reg < 16; ++reg) {
            const int kidx = k0 + (int)rowmap(reg, hh);
            float x = s[reg];
            if (MASK && kidx < A.S) x += kpm[(size_t)b * A.S + kidx];
            const float p = (kidx < A.S && lq != -INFINITY) ? __expf(x - lq) : 0.f;
            s[reg] = p * (dp[reg] - dl);     // dS^T
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[t], s[t], acc, 0, 0, 0);
        // acc[reg] = dQ^T[feature rowmap(reg, hh)][query q0 + r] (before the scale)
    }
    for (unsigned w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (unsigned reg = 0; reg < 16; ++reg) {
                const unsigned dd = rowmap(reg, hh);
                if (w == 0) so[r][dd] = acc[reg];
                else so[r][dd] += acc[reg];
            }
        }
    }
    __syncthreads();
    const unsigned qq = tid >> 3, d4 = tid & 7;
    if (q0 + (int)qq < A.L) {
        const float *src = &so[qq][d4 * 4];
        *reinterpret_cast<float4 *>(dq + ((size_t)(q0 + qq) * A.B + b) * A.lddq + h * kD + d4 * 4) =
            make_float4(src[0] * A.scale, src[1] * A.scale, src[2] * A.scale, src[3] * A.scale);
    }
}

// ------------------------------------------------------------------------------------------
// backward, dK and dV: a block per 32 keys and share z of the query tiles (tiles z, z + QS, ...), the waves split the
// share.  With QS > 1 the blocks write partial sums part[z][S][B][E] that attn_sum_parts adds up in a fixed order: the
// text side has 32 tokens -- one key block per (image, head), 16 blocks on 256 CUs, 40 us; 12 + 4 us in seven shares.
// ------------------------------------------------------------------------------------------
template <bool MASK>
__global__ __launch_bounds__(256) void attn_bwd_dkv(const float *__restrict__ q, const float *__restrict__ k,
                                                    const float *__restrict__ v, const float *__restrict__ kpm,
                                                    const float *__restrict__ dout, const float *__restrict__ lse,
                                                    const float *__restrict__ delta, AttnDims A, int QS, float *__restrict__ dk,
                                                    float *__restrict__ dv)
{
    __shared__ float sk[32][33], sv[32][33];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / A.H, h = bh - b * A.H;
    const int k0 = (int)blockIdx.x * 32;
    const int nqt = (A.L + 31) / 32;
    const int E = A.H * kD;
    const int kr = k0 + (int)r < A.S ? k0 + (int)r : A.S - 1;
    const bool kvalid = k0 + (int)r < A.S;
    float kb[16], vb[16];
    load_half_row(k, kr, b, A.B, A.ldk, h, hh, kb);
    load_half_row(v, kr, b, A.B, A.ldv, h, hh, vb);
    const float km = MASK ? kpm[(size_t)b * A.S + kr] : 0.f;
    f32x16 adk, adv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { adk[i] = 0.f; adv[i] = 0.f; }

    for (int qt = (int)blockIdx.z + QS * (int)wave; qt < nqt; qt += 4 * QS) {
        const int q0 = qt * 32;
        const int qr = q0 + (int)r < A.L ? q0 + (int)r : A.L - 1;
        float qa[16], da[16], qc[16], dc[16];
        load_half_row(q, qr, b, A.B, A.ldq, h, hh, qa);
        load_half_row(dout, qr, b, A.B, E, h, hh, da);
        load_column(q, q0, A.L, b, A.B, A.ldq, h, hh, r, qc);
        load_column(dout, q0, A.L, b, A.B, E, h, hh, r, dc);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[t] * A.scale, kb[t], s, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[t], vb[t], dp, 0, 0, 0);
        // s / dp [reg] = S / dP [query q0 + rowmap(reg, hh)][key k0 + r]
        // (lse / delta of the tile's 32 queries: one load per lane, handed round with ds_bpermute instead of 32 loads: 62 -> 55 us)
        const float lse_r = lse[(size_t)bh * A.L + qr], del_r = delta[(size_t)bh * A.L + qr];
#pragma unroll
        for (unsigned reg = 0; reg < 16; ++reg) {
            const int qi = q0 + (int)rowmap(reg, hh);
            const float lq = __shfl(lse_r, (int)rowmap(reg, hh)), dl = __shfl(del_r, (int)rowmap(reg, hh));
            const float p = (qi < A.L && kvalid && lq != -INFINITY) ? __expf(s[reg] + km - lq) : 0.f;
            s[reg] = p;                        // P
            dp[reg] = p * (dp[reg] - dl);      // dS
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) adv = __builtin_amdgcn_mfma_f32_32x32x2f32(dc[t], s[t], adv, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) adk = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[t], dp[t], adk, 0, 0, 0);
        // adv / adk [reg] = dV^T / dK^T [feature rowmap(reg, hh)][key k0 + r] (dK before the scale)
    }
    for (unsigned w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (unsigned reg = 0; reg < 16; ++reg) {
                const unsigned dd = rowmap(reg, hh);
                if (w == 0) { sk[r][dd] = adk[reg]; sv[r][dd] = adv[reg]; }
                else { sk[r][dd] += adk[reg]; sv[r][dd] += adv[reg]; }
            }
        }
    }
    __syncthreads();
    const unsigned kk = tid >> 3, d4 = tid & 7;
    if (k0 + (int)kk < A.S) {
        // QS > 1: share z of the partial sums, contiguous rows in scratch; otherwise the results with their own row strides
        const size_t rowi = ((size_t)blockIdx.z * A.S + (k0 + kk)) * A.B + b, f = h * kD + d4 * 4;
        const float *a = &sk[kk][d4 * 4], *c = &sv[kk][d4 * 4];
        *reinterpret_cast<float4 *>(dk + rowi * (QS > 1 ? E : A.lddk) + f) = make_float4(a[0] * A.scale, a[1] * A.scale, a[2] * A.scale, a[3] * A.scale);
        *reinterpret_cast<float4 *>(dv + rowi * (QS > 1 ? E : A.lddv) + f) = make_float4(c[0], c[1], c[2], c[3]);
    }
}

// out[i] = part[0][i] + part[1][i] + ... (n4 float4 elements per part)
__global__ __launch_bounds__(256) void attn_sum_parts(const float4 *__restrict__ pk, const float4 *__restrict__ pv, int parts, size_t n4,
                                                      float4 *__restrict__ dk, float4 *__restrict__ dv, int e4, int lddk4, int lddv4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t row = i / e4, col = i - row * e4;   // (rows of e4 float4s; the results have their own row strides)
    float4 a = pk[i], c = pv[i];
    for (int z = 1; z < parts; ++z) {
        const float4 x = pk[z * n4 + i], y = pv[z * n4 + i];
        a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        c.x += y.x; c.y += y.y; c.z += y.z; c.w += y.w;
    }
    dk[row * lddk4 + col] = a;
    dv[row * lddv4 + col] = c;
}

// shares of the query tiles in the dK / dV kernel when there are few key blocks: enough blocks for ~1000, at least one
// tile per wave
int dkv_query_shares(int L, int S, int B, int H)
{
    const int nqt = (L + 31) / 32, blocks = ((S + 31) / 32) * B * H;
    if (blocks >= 128) return 1;   // (measured at 900 keys: 62 us with one share, 70 with three)
    int qs = (1024 + blocks - 1) / blocks;
    qs = qs > 8 ? 8 : qs;
    qs = qs > nqt / 4 ? nqt / 4 : qs;
    return qs < 1 ? 1 : qs;
}

bool attn_args_ok(const void *q, const void *k, const void *v, int L, int S, int B, int H, int d, int ldq, int ldk, int ldv)
{
    if (!q || !k || !v || L <= 0 || S <= 0 || B <= 0 || H <= 0 || d != kD) return false;
    if (ldq < H * kD || ldk < H * kD || ldv < H * kD || (ldq & 3) || (ldk & 3) || (ldv & 3)) return false;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15)) return false;
    if ((long long)B * H > 65535) return false;
    return true;
}

}  // namespace

extern "C" {

int zira_attn_fwd_f32(const float *q, const float *k, const float *v, const float *key_mask, int L, int S, int B, int H,
                      int d, int ldq, int ldk, int ldv, float scale, float *out, float *lse, void *stream)
{
    if (!attn_args_ok(q, k, v, L, S, B, H, d, ldq, ldk, ldv) || !out || !lse) return ZIRA_MSDA_EINVAL;
    const AttnDims A = {L, S, B, H, ldq, ldk, ldv, scale, 0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    const int nqt = (L + 31) / 32;
    if (S >= 256) {
        if (key_mask) hipLaunchKernelGGL((attn_fwd<4, true>), dim3(nqt, B * H), dim3(256), 0, st, q, k, v, key_mask, A, out, lse);
        else hipLaunchKernelGGL((attn_fwd<4, false>), dim3(nqt, B * H), dim3(256), 0, st, q, k, v, key_mask, A, out, lse);
    } else {
        if (key_mask) hipLaunchKernelGGL((attn_fwd<1, true>), dim3((nqt + 3) / 4, B * H), dim3(256), 0, st, q, k, v, key_mask, A, out, lse);
        else hipLaunchKernelGGL((attn_fwd<1, false>), dim3((nqt + 3) / 4, B * H), dim3(256), 0, st, q, k, v, key_mask, A, out, lse);
    }
    return (int)hipGetLastError();
}

size_t zira_attn_bwd_scratch_floats(int L, int S, int B, int H)
{
    if (L <= 0 || S <= 0 || B <= 0 || H <= 0) return 0;
    const int qs = dkv_query_shares(L, S, B, H);
    return (size_t)B * H * L + (qs > 1 ? 2 * (size_t)qs * S * B * H * kD : 0);
}

int zira_attn_bwd_f32(const float *q, const float *k, const float *v, const float *key_mask, const float *out,
                      const float *dout, const float *lse, int L, int S, int B, int H, int d, int ldq, int ldk, int ldv,
                      float scale, float *dq, float *dk, float *dv, float *scratch, size_t scratch_floats, void *stream)
{
    return zira_attn_bwd_ld_f32(q, k, v, key_mask, out, dout, lse, L, S, B, H, d, ldq, ldk, ldv, scale, dq, dk, dv, H * kD, H * kD,
                                H * kD, scratch, scratch_floats, stream);
}

int zira_attn_bwd_ld_f32(const float *q, const float *k, const float *v, const float *key_mask, const float *out,
                         const float *dout, const float *lse, int L, int S, int B, int H, int d, int ldq, int ldk, int ldv,
                         float scale, float *dq, float *dk, float *dv, int lddq, int lddk, int lddv, float *scratch,
                         size_t scratch_floats, void *stream)
{
    if (!attn_args_ok(q, k, v, L, S, B, H, d, ldq, ldk, ldv) || !out || !dout || !lse || !dq || !dk || !dv || !scratch ||
        scratch_floats < (size_t)B * H * L || ((uintptr_t)scratch & 15))
        return ZIRA_MSDA_EINVAL;
    if (lddq < H * kD || lddk < H * kD || lddv < H * kD || ((lddq | lddk | lddv) & 3) ||
        (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15))
        return ZIRA_MSDA_EINVAL;
    const AttnDims A = {L, S, B, H, ldq, ldk, ldv, scale, lddq, lddk, lddv};
    hipStream_t st = (hipStream_t)stream;
    const int nqt = (L + 31) / 32, nkt = (S + 31) / 32;
    float *delta = scratch;                                   // [B, H, L]
    const size_t n = (size_t)S * B * H * kD, doff = (((size_t)B * H * L + 3) / 4) * 4;
    int qs = dkv_query_shares(L, S, B, H);
    if (scratch_floats < doff + 2 * (size_t)qs * n) qs = 1;   // (no room for partial sums: one share)
    float *pk = qs > 1 ? scratch + doff : dk, *pv = qs > 1 ? scratch + doff + (size_t)qs * n : dv;
    if (key_mask) {
        hipLaunchKernelGGL((attn_bwd_dq<true>), dim3(nqt, B * H), dim3(256), 0, st, q, k, v, key_mask, out, dout, lse, A, dq, delta);
        hipLaunchKernelGGL((attn_bwd_dkv<true>), dim3(nkt, B * H, qs), dim3(256), 0, st, q, k, v, key_mask, dout, lse, delta, A, qs, pk, pv);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq<false>), dim3(nqt, B * H), dim3(256), 0, st, q, k, v, key_mask, out, dout, lse, A, dq, delta);
        hipLaunchKernelGGL((attn_bwd_dkv<false>), dim3(nkt, B * H, qs), dim3(256), 0, st, q, k, v, key_mask, dout, lse, delta, A, qs, pk, pv);
    }
    if (qs > 1)
        hipLaunchKernelGGL(attn_sum_parts, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4 *>(pk),
                           reinterpret_cast<const float4 *>(pv), qs, n / 4, reinterpret_cast<float4 *>(dk), reinterpret_cast<float4 *>(dv),
                           H * kD / 4, lddk / 4, lddv / 4);
    return (int)hipGetLastError();
}

}  // extern "C"
