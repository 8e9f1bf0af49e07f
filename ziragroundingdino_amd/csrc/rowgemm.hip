// rowgemm.hip -- fp32 GEMMs of the cross-modal decoder's 900 queries per image (gfx950), with what stands around them
// in a post-LN layer folded into the same launch.
//
// A decoder layer (reference transformer_for_adapter.py:910-1073) applies a dozen nn.Linear to B x 900 rows of 256
// channels and, between them, adds the query position code, adds the residual connection, normalises, and in the
// backward undoes each of those as a kernel of its own: at M = 1800 rows every one of these is a launch of a few
// microseconds whatever it computes (the library GEMMs run at 24 TF/s there).  This kernel computes
//
//     C[M, N] = epilogue( prologue(A)[M, K] * op(W) )
//
// for a block of 16 rows per workgroup, with
//   prologue:  A,  A + pos (for the leading `pos_cols` output columns: q = k = tgt + query_pos, v = tgt in one launch),
//              or the LayerNorm input gradient of the rows (dy, x, gamma, mean, rstd -> dx, which is also written out:
//              it is the gradient of the residual connection in front of the LayerNorm);
//   op(W):     W^T for W [N, K] (nn.Linear forward) or W for W [K, N] (its input gradient with the weight as stored);
//   epilogue:  + bias, + res (a residual connection, or a gradient that arrives by another path: beta = 1), ReLU,
//              zero where mask <= 0 (the ReLU gradient), LayerNorm of the row (sum, mean and rstd written for the backward);
//   rows of A / C may live in batch-first order while the logical rows are (query, batch): the MSDA op's side.
//
// The 16 rows of A are staged once in LDS (row stride K + 4 floats: a wave's ds_read_b128 of 16 rows x 4 k-quads is
// conflict-free); every wave owns 16 * TW output columns and streams its slice of W from L2 / HBM straight into
// registers, kDepth chunks of 16 k ahead of the v_mfma_f32_16x16x4_f32 that consume them (exact fp32 products).  The
// contraction index of an MFMA may be permuted freely as long as both operands agree: lane (row or column l & 15,
// quad l >> 4) holds k = 16 c + 4 (l >> 4) + j for the j-th MFMA of chunk c, which makes both operand loads 16 bytes
// wide.  For W [K, N] a lane's vector spans TW consecutive columns, so accumulator i holds columns n0 + TW * (l & 15) + i
// and the result leaves the lane as one vector store per row.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "zira_msda.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kStage = 2;    // float4s of A a thread loads before it stores them to LDS
constexpr int kLnbK = 256;   // LayerNorm-backward prologue: the row length it is built for (d_model)

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}

// sum over the 16 lanes of a DPP row; every lane ends with the total
__device__ __forceinline__ float row16_sum(float x)
{
    x = dpp_add<0xB1>(x);   // quad_perm:[1,0,3,2]
    x = dpp_add<0x4E>(x);   // quad_perm:[2,3,0,1]
    x = dpp_add<0x141>(x);  // row_half_mirror
    x = dpp_add<0x140>(x);  // row_mirror
    return x;
}

template <int TW> struct WVec;
template <> struct WVec<1> { typedef float type; };
template <> struct WVec<2> { typedef f32x2 type; };
template <> struct WVec<4> { typedef f32x4 type; };
template <int TW> __device__ __forceinline__ float vget(const typename WVec<TW>::type &x, int i) { return x[i]; }
template <> __device__ __forceinline__ float vget<1>(const float &x, int) { return x; }
template <int TW> __device__ __forceinline__ void vset(typename WVec<TW>::type &x, int i, float v) { x[i] = v; }
template <> __device__ __forceinline__ void vset<1>(float &x, int, float v) { x = v; }

// NK: W is [N, K] (C = A W^T); otherwise W is [K, N] (C = A W).  BM: rows per workgroup (16 or 32: RT = BM / 16 row tiles
// share every W operand a wave loads -- with 16 rows the kernel asks the L2 for 8 bytes of W per 64 flops, which is what
// bounds it; 32 rows halve that).  TW: 16-column tiles per wave.  LNB: the operand is the LayerNorm input gradient of
// the rows of A (K == kLnbK).  kDepth: chunks of 16 k whose W operands are in flight per wave.
template <bool NK, int BM, int TW, bool LNB, int kDepth>
__global__ __launch_bounds__(512) void rowgemm_kernel(const zira_rowgemm_args p)
{
    constexpr int RT = BM / 16;
    extern __shared__ f32x4 smem4[];
    float *As = reinterpret_cast<float *>(smem4);
    const int K = p.k, lds_ld = K + 4;
    float *red = As + BM * lds_ld;   // [2][8 waves][16 rows] (LayerNorm epilogue: BM == 16)
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, nl = lane & 15;
    const int row0 = blockIdx.x * BM;
    const int n0w = (blockIdx.y * (nthreads >> 6) + wave) * (TW * 16);
    const int Q = p.batch > 0 ? p.m / p.batch : 0;
    auto mem_row = [&](int r, int batch_first) -> long {
        return batch_first ? (long)(r % p.batch) * Q + r / p.batch : (long)r;
    };
    const int nchunks = K >> 4;

    // ---- W operands: kDepth chunks of 16 k in flight per wave ----
    typedef typename WVec<TW>::type wvec;
    f32x4 wnk[kDepth][NK ? TW : 1];
    wvec wkn[kDepth][NK ? 1 : 4];
    // (per-lane part of the address once, the chunk's part is wave-uniform: scalar arithmetic, no address registers held)
    const float *wlane = NK ? p.w + (long)(n0w + nl) * p.ldw + 4 * kq : p.w + (long)(4 * kq) * p.ldw + n0w + TW * nl;
    auto load_w = [&](int d, int c) {
        if constexpr (NK) {
#pragma unroll
            for (int t = 0; t < TW; ++t)
                wnk[d][t] = *reinterpret_cast<const f32x4 *>(wlane + ((long)(16 * t) * p.ldw + 16 * c));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wkn[d][j] = *reinterpret_cast<const wvec *>(wlane + (long)(16 * c + j) * p.ldw);
        }
    };
    auto load_w_first = [&]() {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) load_w(d, d);
    };

    // ---- prologue: the BM rows of A into LDS.  Loads carry clamped addresses instead of branches so that a batch is
    // issued together; the first chunks of W go out right behind the first batch (loads return in order).
    if constexpr (!LNB) {
        const bool add_pos = p.pos != nullptr && (int)(blockIdx.y * (nthreads >> 6) * (TW * 16)) < p.pos_cols;
        // a unit = 32 float4 of a row, taken by half a wave
        const int ku = K >> 7, nunits = BM * ku, hw = tid >> 5, hl = tid & 31, nhw = nthreads >> 5;
        const float inv_ku = 1.0f / (float)ku;
        f32x4 va[kStage], vp[kStage];
        auto stage_load = [&](int u0) {
#pragma unroll
            for (int s1 = 0; s1 < kStage; ++s1) {
                const int u = min(u0 + s1 * nhw, nunits - 1);
                const int r = (int)(((float)u + 0.5f) * inv_ku), c4 = (u - r * ku) * 32 + hl, gr = min(row0 + r, p.m - 1);
                va[s1] = *reinterpret_cast<const f32x4 *>(p.a + mem_row(gr, p.a_batch_first) * p.lda + 4 * c4);
                vp[s1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (add_pos) vp[s1] = *reinterpret_cast<const f32x4 *>(p.pos + (long)gr * p.ldpos + 4 * c4);
            }
        };
        auto stage_store = [&](int u0) {
#pragma unroll
            for (int s1 = 0; s1 < kStage; ++s1) {
                const int u = u0 + s1 * nhw;
                if (u < nunits) {
                    const int r = (int)(((float)u + 0.5f) * inv_ku), c4 = (u - r * ku) * 32 + hl;
                    const f32x4 v = row0 + r < p.m ? va[s1] + vp[s1] : f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4 *>(As + r * lds_ld + 4 * c4) = v;
                }
            }
        };
        stage_load(hw);
        load_w_first();
        stage_store(hw);
        for (int u0 = hw + nhw * kStage; u0 < nunits; u0 += nhw * kStage) {
            stage_load(u0);
            stage_store(u0);
        }
    } else {
        // A := dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd
        // (the formula of zira_layernorm_bwd_f32); a row per group of 16 lanes, kLnbK / 64 float4s per lane
        constexpr int NV = kLnbK / 64;
        const int grp = tid >> 4, ngrp = nthreads >> 4;
        const float inv_k = 1.0f / (float)kLnbK;
        f32x4 gam[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v)
            gam[v] = p.lnb_gamma ? *reinterpret_cast<const f32x4 *>(p.lnb_gamma + 4 * (nl + 16 * v)) : f32x4{1.f, 1.f, 1.f, 1.f};
        f32x4 g[NV], xh[NV];
        float mu, rs;
        auto row_load = [&](int r) {
            const int gr = min(row0 + min(r, BM - 1), p.m - 1);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                g[v] = *reinterpret_cast<const f32x4 *>(p.a + mem_row(gr, p.a_batch_first) * p.lda + 4 * (nl + 16 * v));
                xh[v] = *reinterpret_cast<const f32x4 *>(p.lnb_x + (long)gr * kLnbK + 4 * (nl + 16 * v));
            }
            mu = p.lnb_mean[gr];
            rs = p.lnb_rstd[gr];
        };
        auto row_finish = [&](int r) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                g[v] = g[v] * gam[v];
                xh[v] = (xh[v] - mu) * rs;
                s1 += (g[v].x + g[v].y) + (g[v].z + g[v].w);
                const f32x4 gx = g[v] * xh[v];
                s2 += (gx.x + gx.y) + (gx.z + gx.w);
            }
            s1 = row16_sum(s1) * inv_k;
            s2 = row16_sum(s2) * inv_k;
            const bool has_row = r < BM, live = row0 + r < p.m;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c4 = nl + 16 * v;
                const f32x4 dx = live ? (g[v] - s1 - xh[v] * s2) * rs : f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_row) *reinterpret_cast<f32x4 *>(As + r * lds_ld + 4 * c4) = dx;
                if (has_row && live && p.lnb_dx != nullptr && blockIdx.y == 0)
                    *reinterpret_cast<f32x4 *>(p.lnb_dx + (long)(row0 + r) * kLnbK + 4 * c4) = dx;
            }
        };
        row_load(grp);
        load_w_first();
        row_finish(grp);
        for (int r = grp + ngrp; r < BM; r += ngrp) {
            row_load(r);
            row_finish(r);
        }
    }

    // ---- operands of the epilogue: requested now, used after the main loop ----
    // acc[rt][i][r] is row row0 + 16 rt + 4 kq + r, column col(i)
    auto col = [&](int i) { return NK ? n0w + 16 * i + nl : n0w + TW * nl + i; };
    const bool has_ln = p.ln_gamma != nullptr || p.ln_sum != nullptr;
    float e_bias[TW], e_gam[TW], e_bet[TW], e_res[RT][4][TW], e_mask[RT][4][TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        e_bias[i] = p.bias ? p.bias[col(i)] : 0.f;
        e_gam[i] = p.ln_gamma ? p.ln_gamma[col(i)] : 1.f;
        e_bet[i] = p.ln_beta ? p.ln_beta[col(i)] : 0.f;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = min(row0 + 16 * rt + 4 * kq + r, p.m - 1);
#pragma unroll
            for (int i = 0; i < TW; ++i) { e_res[rt][r][i] = 0.f; e_mask[rt][r][i] = 1.f; }
            if (p.res != nullptr) {
                const float *rr = p.res + (long)gr * p.ldres;
                if constexpr (NK) {
#pragma unroll
                    for (int i = 0; i < TW; ++i) e_res[rt][r][i] = rr[col(i)];
                } else {
                    const wvec x = *reinterpret_cast<const wvec *>(rr + col(0));
#pragma unroll
                    for (int i = 0; i < TW; ++i) e_res[rt][r][i] = vget<TW>(x, i);
                }
            }
            if (p.mask != nullptr) {
                const float *mr = p.mask + (long)gr * p.n;
                if constexpr (NK) {
#pragma unroll
                    for (int i = 0; i < TW; ++i) e_mask[rt][r][i] = mr[col(i)];
                } else {
                    const wvec x = *reinterpret_cast<const wvec *>(mr + col(0));
#pragma unroll
                    for (int i = 0; i < TW; ++i) e_mask[rt][r][i] = vget<TW>(x, i);
                }
            }
        }
    }
    __syncthreads();

    // ---- main loop ----
    f32x4 acc[RT][TW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[rt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *arow = As + nl * lds_ld + 4 * kq;
    f32x4 a_next[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a_next[rt] = *reinterpret_cast<const f32x4 *>(arow + 16 * rt * lds_ld);
    // one round = kDepth chunks; `refill`: the chunk kDepth ahead is requested into the slot just consumed (every round
    // but the last: unconditional inside a round, a branch there makes the compiler drain the ring)
    auto round = [&](int c0, auto refill) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int c = c0 + d;
            f32x4 a4[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {   // (read a chunk ahead: the scheduling fence below keeps the compiler from doing it)
                a4[rt] = a_next[rt];
                a_next[rt] = *reinterpret_cast<const f32x4 *>(arow + 16 * rt * lds_ld + 16 * min(c + 1, nchunks - 1));
            }
            if constexpr (NK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // (consecutive MFMAs on different accumulators)
#pragma unroll
                    for (int t = 0; t < TW; ++t)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[rt][j], wnk[d][t][j], acc[rt][t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < TW; ++i)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[rt][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[rt][j], vget<TW>(wkn[d][j], i), acc[rt][i], 0, 0, 0);
                }
            }
            if constexpr (decltype(refill)::value) load_w(d, c + kDepth);
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler must not gather the loads at the end of the round)
        }
    };
    int c0 = 0;
    for (; c0 + kDepth < nchunks; c0 += kDepth) round(c0, std::true_type{});
    round(c0, std::false_type{});

    // ---- epilogue ----
    float v[RT][TW][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[rt][i][r] + e_bias[i] + e_res[rt][r][i];
                x = e_mask[rt][r][i] > 0.f ? x : 0.f;
                v[rt][i][r] = p.relu ? fmaxf(x, 0.f) : x;
            }

    if constexpr (RT == 1) {
        if (has_ln) {
            // LayerNorm of the row: the block holds all N = 16 * TW * waves columns; two-pass variance
            const float inv_n = 1.0f / (float)p.n;
            const int nw = nthreads >> 6;
            float mu[4], rs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < TW; ++i) s += v[0][i][r];
                s = row16_sum(s);
                if (nl == 0) red[wave * 16 + 4 * kq + r] = s;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
                for (int w = 0; w < nw; ++w) s += red[w * 16 + 4 * kq + r];
                mu[r] = s * inv_n;
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < TW; ++i) {
                    const float dlt = v[0][i][r] - mu[r];
                    q += dlt * dlt;
                }
                q = row16_sum(q);
                if (nl == 0) red[128 + wave * 16 + 4 * kq + r] = q;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float q = 0.f;
                for (int w = 0; w < nw; ++w) q += red[128 + w * 16 + 4 * kq + r];
                rs[r] = rsqrtf(q * inv_n + p.ln_eps);
            }
            if (wave == 0 && nl == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gr = row0 + 4 * kq + r;
                    if (gr < p.m) {
                        if (p.ln_mean) p.ln_mean[gr] = mu[r];
                        if (p.ln_rstd) p.ln_rstd[gr] = rs[r];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + 4 * kq + r;
                if (p.ln_sum != nullptr && gr < p.m) {
                    float *sr = p.ln_sum + (long)gr * p.n;
                    if constexpr (NK) {
#pragma unroll
                        for (int i = 0; i < TW; ++i) sr[col(i)] = v[0][i][r];
                    } else {
                        wvec x;
#pragma unroll
                        for (int i = 0; i < TW; ++i) vset<TW>(x, i, v[0][i][r]);
                        *reinterpret_cast<wvec *>(sr + col(0)) = x;
                    }
                }
#pragma unroll
                for (int i = 0; i < TW; ++i) v[0][i][r] = (v[0][i][r] - mu[r]) * rs[r] * e_gam[i] + e_bet[i];
            }
        }
    }

#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + 16 * rt + 4 * kq + r;
            if (gr >= p.m) continue;
            float *cr = p.c + mem_row(gr, p.c_batch_first) * p.ldc;
            if constexpr (NK) {
#pragma unroll
                for (int i = 0; i < TW; ++i) cr[col(i)] = v[rt][i][r];
            } else {
                wvec x;
#pragma unroll
                for (int i = 0; i < TW; ++i) vset<TW>(x, i, v[rt][i][r]);
                *reinterpret_cast<wvec *>(cr + col(0)) = x;
            }
        }
    }
}

inline bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

}  // namespace

extern "C" int zira_rowgemm_f32(const zira_rowgemm_args *args, void *stream)
{
    if (args == nullptr) return -1;
    const zira_rowgemm_args &p = *args;
    if (p.a == nullptr || p.w == nullptr || p.c == nullptr) return -1;
    if (p.m < 0 || p.n <= 0 || p.k <= 0) return -2;
    if (p.m == 0) return 0;
    const bool ln = p.ln_gamma != nullptr || p.ln_sum != nullptr;
    // shapes: K in whole prefetch rounds; N in whole blocks of 128 columns (the LayerNorm epilogue needs the row in one
    // block: N == 256); vector accesses need 16-byte aligned rows
    if (p.k % 128 != 0 || p.k > 2048) return -3;
    if (ln ? p.n != 256 : p.n % 128 != 0) return -3;
    if ((p.lda | p.ldw | p.ldc) % 4 != 0 || !aligned16(p.a) || !aligned16(p.w) || !aligned16(p.c)) return -3;
    if (p.pos != nullptr && (p.ldpos % 4 != 0 || !aligned16(p.pos) || p.pos_cols % 128 != 0 || p.lnb_x != nullptr)) return -3;
    if (p.res != nullptr && (p.ldres % 4 != 0 || !aligned16(p.res))) return -3;
    if (p.mask != nullptr && !aligned16(p.mask)) return -3;
    if (p.lnb_x != nullptr) {
        if (p.k != kLnbK || p.lnb_mean == nullptr || p.lnb_rstd == nullptr || !aligned16(p.lnb_x)) return -3;
        if (p.lnb_gamma != nullptr && !aligned16(p.lnb_gamma)) return -3;
        if (p.lnb_dx != nullptr && !aligned16(p.lnb_dx)) return -3;
    }
    if ((p.a_batch_first || p.c_batch_first) && (p.batch <= 0 || p.m % p.batch != 0)) return -3;
    if (ln && p.lnb_x != nullptr) return -3;   // (no caller: a LayerNorm gradient in front and a LayerNorm behind)
    // Tiling.  LayerNorm epilogue: 16 rows x the whole row of 256 columns, 8 waves of 32 columns.  Otherwise 32 rows per
    // block (half the L2 traffic for W) x 128 columns, 4 waves of 32 columns -- or 64 columns, 4 waves of 16, while that
    // still leaves CUs idle (N = 256 at 1800 rows: 114 -> 228 blocks).
    const bool lnb = p.lnb_x != nullptr;
    const int bm = (ln || p.k > 1024) ? 16 : 32;   // (32 rows of more than 1024 floats do not fit the LDS)
    const int rows = (p.m + bm - 1) / bm;
    const bool narrow = bm == 32 && !p.w_is_nk && rows * (p.n / 128) < 160;
    const int threads = ln ? 512 : 256;
    const dim3 grid(rows, ln ? 1 : p.n / (narrow ? 64 : 128));
    const size_t lds = (size_t)(bm * (p.k + 4) + 256) * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kernel) -> int {
        if (lds > 48 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return -4;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, st, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    };
    const bool deep = p.k % 256 == 0;   // whole rounds of 16 chunks (the narrow form: its ring is half as wide)
    if (bm == 16) {   // (LayerNorm epilogue, or long rows; lnb has K == 256 and is not combined with ln)
        return p.w_is_nk ? launch(rowgemm_kernel<true, 16, 2, false, 8>) : launch(rowgemm_kernel<false, 16, 2, false, 8>);
    }
    if (p.w_is_nk) {
        if (lnb) return launch(rowgemm_kernel<true, 32, 2, true, 8>);
        return launch(rowgemm_kernel<true, 32, 2, false, 8>);
    }
    if (narrow) {
        if (lnb) return launch(rowgemm_kernel<false, 32, 1, true, 16>);
        return deep ? launch(rowgemm_kernel<false, 32, 1, false, 16>) : launch(rowgemm_kernel<false, 32, 1, false, 8>);
    }
    return lnb ? launch(rowgemm_kernel<false, 32, 2, true, 8>) : launch(rowgemm_kernel<false, 32, 2, false, 8>);
}
