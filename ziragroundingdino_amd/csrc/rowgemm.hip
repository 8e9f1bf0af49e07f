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

#include "zira_msda.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kBM = 16;      // rows per workgroup (the M of v_mfma_f32_16x16x4_f32)
constexpr int kDepth = 8;    // chunks of 16 k whose W operands are in flight per wave
constexpr int kStage = 4;    // float4s of A a thread loads before it stores them to LDS
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
template <> struct WVec<2> { typedef f32x2 type; };
template <> struct WVec<4> { typedef f32x4 type; };

// NK: W is [N, K] (C = A W^T); otherwise W is [K, N] (C = A W).  TW: 16-column tiles per wave.  LNB: the operand is the
// LayerNorm input gradient of the rows of A (K == kLnbK).
template <bool NK, int TW, bool LNB>
__global__ __launch_bounds__(256, 2) void rowgemm_kernel(const zira_rowgemm_args p)
{
    extern __shared__ f32x4 smem4[];
    float *As = reinterpret_cast<float *>(smem4);
    const int K = p.k, lds_ld = K + 4;
    float *red = As + kBM * lds_ld;   // [2][4 waves][16 rows]
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, nl = lane & 15;
    const int row0 = blockIdx.x * kBM;
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
    auto load_w = [&](int d, int c) {
        if constexpr (NK) {
#pragma unroll
            for (int t = 0; t < TW; ++t)
                wnk[d][t] = *reinterpret_cast<const f32x4 *>(p.w + (long)(n0w + 16 * t + nl) * p.ldw + 16 * c + 4 * kq);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wkn[d][j] = *reinterpret_cast<const wvec *>(p.w + (long)(16 * c + 4 * kq + j) * p.ldw + n0w + TW * nl);
        }
    };

    // ---- prologue: the 16 rows of A into LDS.  Loads carry clamped addresses instead of branches so that a batch is
    // issued together; the first chunks of W go out right behind the first batch (loads return in order).
    if constexpr (!LNB) {
        const bool add_pos = p.pos != nullptr && (int)(blockIdx.y * (nthreads >> 6) * (TW * 16)) < p.pos_cols;
        // a unit = 32 float4 of a row, taken by half a wave
        const int ku = K >> 7, nunits = kBM * ku, hw = tid >> 5, hl = tid & 31, nhw = nthreads >> 5;
        const float inv_ku = 1.0f / (float)ku;
        for (int u0 = hw; u0 < nunits; u0 += nhw * kStage) {
            f32x4 va[kStage], vp[kStage];
#pragma unroll
            for (int s1 = 0; s1 < kStage; ++s1) {
                const int u = min(u0 + s1 * nhw, nunits - 1);
                const int r = (int)(((float)u + 0.5f) * inv_ku), c4 = (u - r * ku) * 32 + hl, gr = min(row0 + r, p.m - 1);
                va[s1] = *reinterpret_cast<const f32x4 *>(p.a + mem_row(gr, p.a_batch_first) * p.lda + 4 * c4);
            }
            if (add_pos) {
#pragma unroll
                for (int s1 = 0; s1 < kStage; ++s1) {
                    const int u = min(u0 + s1 * nhw, nunits - 1);
                    const int r = (int)(((float)u + 0.5f) * inv_ku), c4 = (u - r * ku) * 32 + hl, gr = min(row0 + r, p.m - 1);
                    vp[s1] = *reinterpret_cast<const f32x4 *>(p.pos + (long)gr * p.ldpos + 4 * c4);
                }
            } else {
#pragma unroll
                for (int s1 = 0; s1 < kStage; ++s1) vp[s1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (u0 == hw) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) load_w(d, d);
            }
#pragma unroll
            for (int s1 = 0; s1 < kStage; ++s1) {
                const int u = u0 + s1 * nhw;
                if (u < nunits) {
                    const int r = (int)(((float)u + 0.5f) * inv_ku), c4 = (u - r * ku) * 32 + hl;
                    const f32x4 v = row0 + r < p.m ? va[s1] + vp[s1] : f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4 *>(As + r * lds_ld + 4 * c4) = v;
                }
            }
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
        for (int r = grp; r < kBM; r += ngrp) {
            const int gr = min(row0 + r, p.m - 1);
            f32x4 g[NV], xh[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                g[v] = *reinterpret_cast<const f32x4 *>(p.a + mem_row(gr, p.a_batch_first) * p.lda + 4 * (nl + 16 * v));
                xh[v] = *reinterpret_cast<const f32x4 *>(p.lnb_x + (long)gr * kLnbK + 4 * (nl + 16 * v));
            }
            const float mu = p.lnb_mean[gr], rs = p.lnb_rstd[gr];
            if (r == grp) {
#pragma unroll
                for (int d = 0; d < kDepth; ++d) load_w(d, d);
            }
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
            const bool live = row0 + r < p.m;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c4 = nl + 16 * v;
                const f32x4 dx = live ? (g[v] - s1 - xh[v] * s2) * rs : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(As + r * lds_ld + 4 * c4) = dx;
                if (live && p.lnb_dx != nullptr && blockIdx.y == 0)
                    *reinterpret_cast<f32x4 *>(p.lnb_dx + (long)gr * kLnbK + 4 * c4) = dx;
            }
        }
    }
    __syncthreads();

    // ---- main loop ----
    f32x4 acc[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *arow = As + nl * lds_ld + 4 * kq;
    f32x4 a_next = *reinterpret_cast<const f32x4 *>(arow);
    for (int c0 = 0; c0 < nchunks; c0 += kDepth) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int c = c0 + d;
            const f32x4 a4 = a_next;   // (read a chunk ahead: the scheduling fence below keeps the compiler from doing it)
            a_next = *reinterpret_cast<const f32x4 *>(arow + 16 * min(c + 1, nchunks - 1));
            if constexpr (NK) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // (consecutive MFMAs on different accumulators)
#pragma unroll
                    for (int t = 0; t < TW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], wnk[d][t][j], acc[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < TW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], wkn[d][j][i], acc[i], 0, 0, 0);
                }
            }
            load_w(d, min(c + kDepth, nchunks - 1));   // (unconditional: a branch here makes the compiler drain the ring)
            __builtin_amdgcn_sched_barrier(0);          // (and the scheduler must not gather the loads at the end of the round)
        }
    }

    // ---- epilogue ----
    // acc[i][r] is row row0 + 4 kq + r, column col(i)
    auto col = [&](int i) { return NK ? n0w + 16 * i + nl : n0w + TW * nl + i; };
    float v[TW][4];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const float b = p.bias ? p.bias[col(i)] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] = acc[i][r] + b;
    }
    if (p.res != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + 4 * kq + r;
            if (gr < p.m) {
                const float *rr = p.res + (long)gr * p.ldres;
                if constexpr (NK) {
#pragma unroll
                    for (int i = 0; i < TW; ++i) v[i][r] += rr[col(i)];
                } else {
                    const wvec x = *reinterpret_cast<const wvec *>(rr + col(0));
#pragma unroll
                    for (int i = 0; i < TW; ++i) v[i][r] += x[i];
                }
            }
        }
    }
    if (p.mask != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gr = row0 + 4 * kq + r;
            if (gr < p.m) {
                const float *mr = p.mask + (long)gr * p.n;
                if constexpr (NK) {
#pragma unroll
                    for (int i = 0; i < TW; ++i) v[i][r] = mr[col(i)] > 0.f ? v[i][r] : 0.f;
                } else {
                    const wvec x = *reinterpret_cast<const wvec *>(mr + col(0));
#pragma unroll
                    for (int i = 0; i < TW; ++i) v[i][r] = x[i] > 0.f ? v[i][r] : 0.f;
                }
            }
        }
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][r] = fmaxf(v[i][r], 0.f);
    }

    float y[TW][4];
    if (p.ln_gamma != nullptr || p.ln_sum != nullptr) {
        // LayerNorm of the row: the block holds all N = 16 * TW * waves columns; two-pass variance
        const float inv_n = 1.0f / (float)p.n;
        const int nw = nthreads >> 6;
        float mu[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < TW; ++i) s += v[i][r];
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
                const float dlt = v[i][r] - mu[r];
                q += dlt * dlt;
            }
            q = row16_sum(q);
            if (nl == 0) red[64 + wave * 16 + 4 * kq + r] = q;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float q = 0.f;
            for (int w = 0; w < nw; ++w) q += red[64 + w * 16 + 4 * kq + r];
            rs[r] = rsqrtf(q * inv_n + p.ln_eps);
        }
#pragma unroll
        for (int i = 0; i < TW; ++i) {
            const float gam = p.ln_gamma ? p.ln_gamma[col(i)] : 1.f;
            const float bet = p.ln_beta ? p.ln_beta[col(i)] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[i][r] = (v[i][r] - mu[r]) * rs[r] * gam + bet;
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
        if (p.ln_sum != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gr = row0 + 4 * kq + r;
                if (gr >= p.m) continue;
                float *sr = p.ln_sum + (long)gr * p.n;
                if constexpr (NK) {
#pragma unroll
                    for (int i = 0; i < TW; ++i) sr[col(i)] = v[i][r];
                } else {
                    wvec x;
#pragma unroll
                    for (int i = 0; i < TW; ++i) x[i] = v[i][r];
                    *reinterpret_cast<wvec *>(sr + col(0)) = x;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[i][r] = v[i][r];
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gr = row0 + 4 * kq + r;
        if (gr >= p.m) continue;
        float *cr = p.c + mem_row(gr, p.c_batch_first) * p.ldc;
        if constexpr (NK) {
#pragma unroll
            for (int i = 0; i < TW; ++i) cr[col(i)] = y[i][r];
        } else {
            wvec x;
#pragma unroll
            for (int i = 0; i < TW; ++i) x[i] = y[i][r];
            *reinterpret_cast<wvec *>(cr + col(0)) = x;
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
    if (p.k % (16 * kDepth) != 0 || p.k > 2048) return -3;
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
    const dim3 grid((p.m + kBM - 1) / kBM, ln ? 1 : p.n / 128);
    const size_t lds = (size_t)(kBM * (p.k + 4) + 128) * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto kernel) -> int {
        if (lds > 48 * 1024) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return -4;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    };
    const bool lnb = p.lnb_x != nullptr;
    if (ln) {
        if (lnb) return -3;   // (no caller: a LayerNorm gradient in front and a LayerNorm behind)
        return p.w_is_nk ? launch(rowgemm_kernel<true, 4, false>) : launch(rowgemm_kernel<false, 4, false>);
    }
    if (lnb) return p.w_is_nk ? launch(rowgemm_kernel<true, 2, true>) : launch(rowgemm_kernel<false, 2, true>);
    return p.w_is_nk ? launch(rowgemm_kernel<true, 2, false>) : launch(rowgemm_kernel<false, 2, false>);
}
