// textside.hip -- the TEXT side of an image <-> text fusion block (BiAttentionBlock with frozen, composed projections) as
// six launches instead of ~41 launch-bound ATen kernels per block (C ABI: zira_text_prep_fwd/bwd_f32, zira_text_out_fwd/bwd_f32).
//
// Reference: BiMultiHeadAttention / BiAttentionBlock (groundingdino/models/GroundingDINO/fuse_modules.py:99-305).  This
// package re-brackets the block's products around the B x T <= 512 text tokens (transformer.BiMultiHeadAttention.forward) and,
// while the six projections are frozen, composes the text side's double projections into constant matrices
// (_composed_text_side): what is left per block on the text side is
//     l_ln = LayerNorm(l);  [a | c | z] = l_ln [AC | Z] + bias   -> a [B, Dv, H T], c [B, H T], z [B, H T, Dv]   ("prep")
//     out_l = l_ln + scale (o0 + (u / colsum) O)                                                                  ("out")
// with u [B, H T, Dv] / colsum [B, H T] coming back from the image side -- a LayerNorm, three addmm of a few MFLOP, five layout
// copies, a division and the residual forward, twice that backward; each a launch of ~3 us on the critical path of the step.
// Here: tiles of 32 rows x 32 columns x 128 of K per block (both operands staged in LDS with one batch of independent loads),
// K split over blocks where it is long, the partial tiles summed by the small kernel
// that applies the epilogue -- seven launches per fusion block forward + backward.  Operands and results are read and written
// in the layouts above through index maps; the constant weights are kept in both orientations so that every weight load is
// coalesced.  Two earlier forms are on record in DESIGN.md: 8 x 8 output tiles through LDS chunks (10-61 us per launch: one
// dependent round trip per chunk, one block per CU) and four rows per block with a lane per column (10-41 us: each of the 16
// row blocks pulled the whole weight matrix through ONE CU's L2 port; these products need their bytes AND their 33-67 MFLOP
// spread over 64+ CUs -- a 64 x 64 x 128 tile is already 4 us of one CU's vector units --, which only a K split gives a 64-row
// problem).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int TR = 32, TC = 32, KC = 128, NTHR = 256;   // tile rows, tile columns, K per chunk
// A block of 4 waves computes 32 x 32 outputs over 128 of K: 131 K FMAs, ~1 us of one CU's vector units -- a 64-row product
// becomes 64 - 272 such blocks.  Thread: column lane % 32, rows 4 g .. 4 g + 3 with g = 2 wave + lane / 32.

struct Dims {
    int B, T, H, Dv, Dl;   // rows M = B T; text width Dl; image width Dv per head
};

struct P2 {
    const float *p0, *p1;
};

// One chunk: Xs [KC][TR], Ws [KC][TC] in LDS.  The operands come in three phases kept apart by scheduling barriers, so that
// the chunk costs ONE round trip to memory: XA(r, k) / WA(k, c) give the addresses (always valid: clamped indices), then all 48
// loads are issued, then XV(v0, v1, r, k) / WV(w, c) finish the values without touching memory (LDS and registers only; they
// return 0 outside the problem).  Left to itself the compiler issues each element's loads behind the previous element's
// arithmetic -- its integer divisions, the float division of u / colsum -- and a chunk becomes 16 dependent round trips
// (14 us of the 19 us launch, scripts/textside_target.py).  `row_fast` picks the staging order of X (which index is
// contiguous in memory).
constexpr int NX = KC * TR / NTHR, NW = KC * TC / NTHR;

struct Staged {
    float v0[NX], v1[NX], w[NW];
};

template <bool row_fast, class XA, class WA>
__device__ __forceinline__ void chunk_load(int k0, XA xa, WA wa, Staged &S)
{
    const int tid = threadIdx.x;
    P2 ax[NX];
    const float *aw[NW];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + j * NTHR;
        const int k = row_fast ? i / TR : i % KC, r = row_fast ? i % TR : i / KC;
        ax[j] = xa(r, k0 + k);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = tid + j * NTHR, k = i / TC, c = i - k * TC;
        aw[j] = wa(k0 + k, c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        S.v0[j] = *ax[j].p0;
        S.v1[j] = *ax[j].p1;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) S.w[j] = *aw[j];
    __builtin_amdgcn_sched_barrier(0);
}

template <bool row_fast, class XV, class WV>
__device__ __forceinline__ void chunk_finish(float *Xs, float *Ws, int k0, int kn, XV xv_, WV wv_, const Staged &S,
                                             float (&acc)[4])
{
    const int tid = threadIdx.x;
    __syncthreads();      // (the previous chunk's readers are done with the tiles)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + j * NTHR;
        const int k = row_fast ? i / TR : i % KC, r = row_fast ? i % TR : i / KC;
        Xs[k * TR + r] = k < kn ? xv_(S.v0[j], S.v1[j], r, k0 + k) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = tid + j * NTHR, k = i / TC, c = i - k * TC;
        Ws[i] = k < kn ? wv_(S.w[j], c) : 0.f;
    }
    __syncthreads();
    const int col = tid & 31, g = tid >> 5;
#pragma unroll 8
    for (int k = 0; k < KC; ++k) {
        const float w = Ws[k * TC + col];
        const float4 x = *reinterpret_cast<const float4 *>(Xs + k * TR + 4 * g);
        acc[0] = fmaf(x.x, w, acc[0]);
        acc[1] = fmaf(x.y, w, acc[1]);
        acc[2] = fmaf(x.z, w, acc[2]);
        acc[3] = fmaf(x.w, w, acc[3]);
    }
}

template <bool row_fast, class XA, class XV, class WA, class WV>
__device__ __forceinline__ void chunk_fma(float *Xs, float *Ws, int k0, int kn, XA xa, XV xv_, WA wa, WV wv_, float (&acc)[4])
{
    Staged S;
    chunk_load<row_fast>(k0, xa, wa, S);
    chunk_finish<row_fast>(Xs, Ws, k0, kn, xv_, wv_, S, acc);
}

// K <= 2 KC in one block: both chunks' loads are issued up front (together with whatever else the kernel needs from memory:
// one round trip), the second chunk's are in flight while the first is multiplied
template <class XA, class WA>
__device__ __forceinline__ void two_chunks_load(int K, XA xa, WA wa, Staged &S0, Staged &S1)
{
    chunk_load<false>(0, xa, wa, S0);
    if (K > KC) chunk_load<false>(KC, xa, wa, S1);
}

template <class XV, class WV, class AF>
__device__ __forceinline__ void two_chunks_finish(float *Xs, float *Ws, int K, XV xv_, WV wv_, const Staged &S0, const Staged &S1,
                                                  float (&acc)[4], AF after)
{
    chunk_finish<false>(Xs, Ws, 0, min(KC, K), xv_, wv_, S0, acc);
    after(0);
    if (K > KC) {
        chunk_finish<false>(Xs, Ws, KC, min(KC, K - KC), xv_, wv_, S1, acc);
        after(KC);
    }
}

// ---- prep forward: LayerNorm while the rows are staged, outputs scattered into a / c / z ----------------------------------------
// grid (column tiles of [AC | Z], row tiles); K = Dl in chunks inside the block
__global__ __launch_bounds__(NTHR) void text_prep_fwd_kernel(const float *__restrict__ l_in, const float *__restrict__ ln_w,
                                                             const float *__restrict__ ln_b, float eps, const float *__restrict__ W1,
                                                             const float *__restrict__ b1, Dims d, float *__restrict__ l_ln,
                                                             float *__restrict__ a, float *__restrict__ c, float *__restrict__ z,
                                                             float *__restrict__ stats)
{
    extern __shared__ float smem[];
    float *Xs = smem, *Ws = smem + KC * TR;
    __shared__ float st[TR][2];
    const int M = d.B * d.T, HD = d.H * d.Dv, N1 = 2 * HD + d.H;
    const int m0 = blockIdx.y * TR, n0 = blockIdx.x * TC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool first = blockIdx.x == 0;
    __shared__ float lw[256], lb[256];     // (Dl <= 256)
    auto XA = [&](int r, int k) {
        const float *p = l_in + (size_t)min(m0 + r, M - 1) * d.Dl + min(k, d.Dl - 1);
        return P2{p, p};
    };
    auto XV = [&](float v, float, int r, int k) { return (m0 + r < M && k < d.Dl) ? fmaf((v - st[r][0]) * st[r][1], lw[k], lb[k]) : 0.f; };
    auto WA = [&](int k, int cc) { return W1 + (size_t)min(k, d.Dl - 1) * N1 + min(n0 + cc, N1 - 1); };
    auto WV = [&](float w, int cc) { return n0 + cc < N1 ? w : 0.f; };
    // everything the block reads, in one round trip: the operands of both chunks, the LayerNorm parameters, the bias, and the
    // tile's rows once more for their statistics (a wave takes 8 rows; Dl <= 256: 4 values per lane)
    Staged S0, S1;
    two_chunks_load(d.Dl, XA, WA, S0, S1);
    const int kq = min((int)threadIdx.x, d.Dl - 1);
    const float lwv = ln_w[kq], lbv = ln_b[kq];
    const int n = n0 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    const float bias = b1[min(n, N1 - 1)];
    constexpr int RW = TR / (NTHR / 64);
    float x[RW][4];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int m = min(m0 + wave * RW + i, M - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[i][j] = l_in[(size_t)m * d.Dl + min(lane + 64 * j, d.Dl - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    lw[threadIdx.x] = lwv;
    lb[threadIdx.x] = lbv;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int r = wave * RW + i, m = m0 + r;
        float sx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sx[j] = (lane + 64 * j < d.Dl) ? x[i][j] : 0.f;
        float s = (sx[0] + sx[1]) + (sx[2] + sx[3]);
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w);
        const float mean = s / (float)d.Dl;
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = (lane + 64 * j < d.Dl) ? x[i][j] - mean : 0.f;
            v = fmaf(t, t, v);
        }
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w);
        const float rstd = rsqrtf(v / (float)d.Dl + eps);
        if (lane == 0) {
            st[r][0] = mean;
            st[r][1] = rstd;
            if (first && m < M) {
                stats[2 * m] = mean;
                stats[2 * m + 1] = rstd;
            }
        }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    two_chunks_finish(Xs, Ws, d.Dl, XV, WV, S0, S1, acc, [&](int k0) {     // (chunk_finish synchronises before it reads st / lw / lb)
        if (first) {     // the normalised rows themselves, from the staged tile
            for (int i = threadIdx.x; i < KC * TR; i += NTHR) {
                const int r = i / KC, k = i - r * KC;
                if (m0 + r < M && k0 + k < d.Dl) l_ln[(size_t)(m0 + r) * d.Dl + k0 + k] = Xs[k * TR + r];
            }
        }
    });
    if (n >= N1) return;

    int kind = 0, h = 0, dd = 0;
    if (n < HD) {
        h = n / d.Dv;
        dd = n - h * d.Dv;
    } else if (n < HD + d.H) {
        kind = 1;
        h = n - HD;
    } else {
        kind = 2;
        const int q = n - HD - d.H;
        h = q / d.Dv;
        dd = q - h * d.Dv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + 4 * g + j;
        if (m >= M) continue;
        const int b = m / d.T, t = m - b * d.T;
        const float s = acc[j] + bias;
        if (kind == 0) a[((size_t)b * d.Dv + dd) * (d.H * d.T) + h * d.T + t] = s;
        else if (kind == 1) c[(size_t)b * d.H * d.T + h * d.T + t] = s;
        else z[((size_t)b * d.H * d.T + h * d.T + t) * d.Dv + dd] = s;
    }
}

// ---- prep backward, partial products: part[ks][m][n] = sum over the block's K chunk of G(m, k) W1T[k][n] ----------------------
// grid (column tiles of Dl, K chunks, row tiles); G gathered from the gradients of a / c / z
__global__ __launch_bounds__(NTHR) void text_prep_bwd_kernel(const float *__restrict__ g_a, const float *__restrict__ g_c,
                                                             const float *__restrict__ g_z, const float *__restrict__ W1T, Dims d,
                                                             float *__restrict__ part)
{
    extern __shared__ float smem[];
    float *Xs = smem, *Ws = smem + KC * TR;
    const int M = d.B * d.T, HD = d.H * d.Dv, N1 = 2 * HD + d.H;
    const int m0 = blockIdx.z * TR, n0 = blockIdx.x * TC, k0 = blockIdx.y * KC;
    const int T = d.T, H = d.H, Dv = d.Dv, Dl = d.Dl, kn = min(KC, N1 - k0);
    auto WA = [=](int k, int cc) { return W1T + (size_t)min(k, N1 - 1) * Dl + min(n0 + cc, Dl - 1); };
    auto WV = [=](float w, int cc) { return n0 + cc < Dl ? w : 0.f; };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // three kinds of chunk (block-uniform): all of it in the columns of a (contiguous along the rows: consecutive t), all of it
    // in the columns of z (contiguous along K), or the one that holds the H columns of c
    if (k0 + KC <= HD) {
        if (g_a) {
            auto XA = [=](int r, int nn) {
                const int m = min(m0 + r, M - 1), b = m / T, t = m - b * T, h = nn / Dv, dd = nn - h * Dv;
                const float *p = g_a + ((size_t)b * Dv + dd) * (H * T) + h * T + t;
                return P2{p, p};
            };
            auto XV = [=](float v, float, int r, int) { return m0 + r < M ? v : 0.f; };
            chunk_fma<true>(Xs, Ws, k0, kn, XA, XV, WA, WV, acc);
        }
    } else if (k0 >= HD + H) {
        if (g_z) {
            auto XA = [=](int r, int nn) {
                const int m = min(m0 + r, M - 1), b = m / T, t = m - b * T;
                const int q = min(nn, N1 - 1) - HD - H, h = q / Dv, dd = q - h * Dv;
                const float *p = g_z + ((size_t)b * H * T + h * T + t) * Dv + dd;
                return P2{p, p};
            };
            auto XV = [=](float v, float, int r, int nn) { return (m0 + r < M && nn < N1) ? v : 0.f; };
            chunk_fma<false>(Xs, Ws, k0, kn, XA, XV, WA, WV, acc);
        }
    } else {
        const float *dummy = W1T;   // (a valid address for the loads of absent gradients)
        auto XA = [=](int r, int nn) {
            const int m = min(m0 + r, M - 1), n1 = min(nn, N1 - 1);
            const int b = m / T, t = m - b * T;
            const bool ra = n1 < HD, rc = !ra && n1 < HD + H;
            const int q = ra ? n1 : (rc ? 0 : n1 - HD - H), h = rc ? n1 - HD : q / Dv, dd = rc ? 0 : q - (q / Dv) * Dv;
            const float *base = ra ? g_a : (rc ? g_c : g_z);
            const size_t idx = ra ? ((size_t)b * Dv + dd) * (H * T) + h * T + t
                                  : (rc ? (size_t)b * H * T + h * T + t : ((size_t)b * H * T + h * T + t) * Dv + dd);
            const float *p = base ? base + idx : dummy;
            return P2{p, p};
        };
        auto XV = [=](float v, float, int r, int nn) {
            const float *base = nn < HD ? g_a : (nn < HD + H ? g_c : g_z);
            return (base && m0 + r < M && nn < N1) ? v : 0.f;
        };
        chunk_fma<false>(Xs, Ws, k0, kn, XA, XV, WA, WV, acc);
    }
    const int n = n0 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    if (n >= d.Dl) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + 4 * g + j;
        if (m < M) part[((size_t)blockIdx.y * M + m) * d.Dl + n] = acc[j];
    }
}

// ---- prep backward, finish: g_ln = sum of the partials + the direct gradient, LayerNorm backward on the saved statistics -------
__global__ __launch_bounds__(256) void text_prep_bwd_finish_kernel(const float *__restrict__ part, int nparts,
                                                                   const float *__restrict__ g_res, const float *__restrict__ l_in,
                                                                   const float *__restrict__ ln_w, const float *__restrict__ stats,
                                                                   int M, int Dl, float *__restrict__ g_in)
{
    __shared__ float ws[4][2];
    const int m = blockIdx.x, k = threadIdx.x, wave = k >> 6, lane = k & 63;   // a block per row, a channel per thread (Dl <= 256)
    const float mean = stats[2 * m], rstd = stats[2 * m + 1];
    float gw = 0.f, xh = 0.f;
    if (k < Dl) {
        float g0 = g_res ? g_res[(size_t)m * Dl + k] : 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        int p = 0;
        for (; p + 4 <= nparts; p += 4) {
            g0 += part[((size_t)p * M + m) * Dl + k];
            g1 += part[((size_t)(p + 1) * M + m) * Dl + k];
            g2 += part[((size_t)(p + 2) * M + m) * Dl + k];
            g3 += part[((size_t)(p + 3) * M + m) * Dl + k];
        }
        for (; p < nparts; ++p) g0 += part[((size_t)p * M + m) * Dl + k];
        gw = ((g0 + g1) + (g2 + g3)) * ln_w[k];
        xh = (l_in[(size_t)m * Dl + k] - mean) * rstd;
    }
    float s1 = gw, s2 = gw * xh;
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        s1 += __shfl_xor(s1, w);
        s2 += __shfl_xor(s2, w);
    }
    if (lane == 0) {
        ws[wave][0] = s1;
        ws[wave][1] = s2;
    }
    __syncthreads();
    s1 = (ws[0][0] + ws[1][0]) + (ws[2][0] + ws[3][0]);
    s2 = (ws[0][1] + ws[1][1]) + (ws[2][1] + ws[3][1]);
    const float inv = 1.f / (float)Dl;
    if (k < Dl) g_in[(size_t)m * Dl + k] = rstd * (gw - inv * s1 - xh * (inv * s2));
}

// scale of the text residual: gamma[n] (layer scale) times the per-sample stochastic-depth factor keep[b] when drawn
__device__ __forceinline__ float scale_of(const float *gamma, const float *keep, int b, int n)
{
    return keep ? gamma[n] * keep[b] : gamma[n];
}

// ---- out forward, partial products: part[ks][m][n] = sum over the K chunk of (u / colsum)(m, k) O[k][n] ------------------------
__global__ __launch_bounds__(NTHR) void text_out_fwd_kernel(const float *__restrict__ u, const float *__restrict__ colsum,
                                                            const float *__restrict__ O, Dims d, float *__restrict__ part)
{
    extern __shared__ float smem[];
    float *Xs = smem, *Ws = smem + KC * TR;
    const int M = d.B * d.T, HD = d.H * d.Dv;
    const int m0 = blockIdx.z * TR, n0 = blockIdx.x * TC, k0 = blockIdx.y * KC;
    auto XA = [&](int r, int k) {
        const int m = min(m0 + r, M - 1), kk = min(k, HD - 1);
        const int b = m / d.T, t = m - b * d.T, h = kk / d.Dv, dd = kk - h * d.Dv;
        const size_t row = (size_t)b * d.H * d.T + h * d.T + t;
        return P2{u + row * d.Dv + dd, colsum + row};
    };
    auto XV = [&](float v, float cs, int r, int k) { return (m0 + r < M && k < HD) ? v / cs : 0.f; };
    auto WA = [&](int k, int cc) { return O + (size_t)min(k, HD - 1) * d.Dl + min(n0 + cc, d.Dl - 1); };
    auto WV = [&](float w, int cc) { return n0 + cc < d.Dl ? w : 0.f; };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    chunk_fma<false>(Xs, Ws, k0, min(KC, HD - k0), XA, XV, WA, WV, acc);
    const int n = n0 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    if (n >= d.Dl) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + 4 * g + j;
        if (m < M) part[((size_t)blockIdx.y * M + m) * d.Dl + n] = acc[j];
    }
}

// ---- out forward, finish: out = l_ln + scale (o0 + sum of the partials) ----------------------------------------------------------
__global__ __launch_bounds__(256) void text_out_fwd_finish_kernel(const float *__restrict__ part, int nparts,
                                                                  const float *__restrict__ l_ln, const float *__restrict__ o0,
                                                                  const float *__restrict__ gamma, const float *__restrict__ keep, int M,
                                                                  int T, int Dl, float *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * Dl) return;
    const int m = (int)(i / Dl), n = (int)(i - (long long)m * Dl);
    float s = o0[n];
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * M * Dl + i];
    out[i] = fmaf(scale_of(gamma, keep, m / T, n), s, l_ln[i]);
}

// ---- out backward: g_u(m, k) = (sum_n scale g[m][n] O[k][n]) / colsum, written in u's layout; OT [Dl][H Dv] ---------------------
// grid (column tiles of H Dv, row tiles); K = Dl in chunks inside the block
__global__ __launch_bounds__(NTHR) void text_out_bwd_kernel(const float *__restrict__ g, const float *__restrict__ colsum,
                                                            const float *__restrict__ OT, const float *__restrict__ gamma,
                                                            const float *__restrict__ keep, Dims d, float *__restrict__ g_u)
{
    extern __shared__ float smem[];
    float *Xs = smem, *Ws = smem + KC * TR;
    const int M = d.B * d.T, HD = d.H * d.Dv;
    const int m0 = blockIdx.y * TR, n0 = blockIdx.x * TC;
    __shared__ float gam[256], rowk[TR];     // the layer scale (Dl <= 256) and the rows' stochastic-depth factors
    auto XA = [&](int r, int nn) {
        const float *p = g + (size_t)min(m0 + r, M - 1) * d.Dl + min(nn, d.Dl - 1);
        return P2{p, p};
    };
    auto XV = [&](float v, float, int r, int nn) { return (m0 + r < M && nn < d.Dl) ? v * (gam[nn] * rowk[r]) : 0.f; };
    auto WA = [&](int nn, int cc) { return OT + (size_t)min(nn, d.Dl - 1) * HD + min(n0 + cc, HD - 1); };
    auto WV = [&](float w, int cc) { return n0 + cc < HD ? w : 0.f; };
    Staged S0, S1;
    two_chunks_load(d.Dl, XA, WA, S0, S1);      // (with the scales below: one round trip)
    const float gv = gamma[min((int)threadIdx.x, d.Dl - 1)];
    const float kv = keep ? keep[min(m0 + (int)(threadIdx.x & (TR - 1)), M - 1) / d.T] : 1.f;
    __builtin_amdgcn_sched_barrier(0);
    gam[threadIdx.x] = gv;
    if (threadIdx.x < TR) rowk[threadIdx.x] = kv;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    two_chunks_finish(Xs, Ws, d.Dl, XV, WV, S0, S1, acc, [](int) {});
    const int k = n0 + (threadIdx.x & 31), rg = threadIdx.x >> 5;   // output column (h, dd)
    if (k >= HD) return;
    const int h = k / d.Dv, dd = k - h * d.Dv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + 4 * rg + j;
        if (m >= M) continue;
        const int b = m / d.T, t = m - b * d.T;
        const size_t row = (size_t)b * d.H * d.T + h * d.T + t;
        g_u[row * d.Dv + dd] = acc[j] / colsum[row];
    }
}

// g_colsum[row] = -(sum_d g_u[row][d] u[row][d]) / colsum[row]  (d (u / colsum) / d colsum); a wave per row of [B, H T]
__global__ __launch_bounds__(256) void text_colsum_bwd_kernel(const float *__restrict__ g_u, const float *__restrict__ u,
                                                              const float *__restrict__ colsum, int rows, int Dv,
                                                              float *__restrict__ g_colsum)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int k = lane; k < Dv; k += 64) s = fmaf(g_u[(size_t)r * Dv + k], u[(size_t)r * Dv + k], s);
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) s += __shfl_xor(s, w);
    if (lane == 0) g_colsum[r] = -s / colsum[r];
}

constexpr size_t kLds = (size_t)KC * (TR + TC) * sizeof(float);   // 32 KB

inline bool bad_dims(int B, int T, int H, int Dv, int Dl)
{
    return B <= 0 || T <= 0 || H <= 0 || Dv <= 0 || Dl <= 0 || Dl > 256 || (long long)B * T > (1 << 20) || (long long)H * Dv > (1 << 20);
}

}  // namespace

extern "C" size_t zira_text_side_scratch_floats(int B, int T, int H, int Dv, int Dl)
{
    if (bad_dims(B, T, H, Dv, Dl)) return 0;
    const size_t N1 = 2 * (size_t)H * Dv + H, parts = (N1 + KC - 1) / KC;
    return parts * (size_t)B * T * Dl;
}

extern "C" int zira_text_prep_fwd_f32(const float *l_in, const float *ln_w, const float *ln_b, float eps, const float *W1,
                                      const float *b1, int B, int T, int H, int Dv, int Dl, float *l_ln, float *a, float *c, float *z,
                                      float *stats, void *stream)
{
    if (!l_in || !ln_w || !ln_b || !W1 || !b1 || !l_ln || !a || !c || !z || !stats || bad_dims(B, T, H, Dv, Dl))
        return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T, N1 = 2 * H * Dv + H;
    hipLaunchKernelGGL(text_prep_fwd_kernel, dim3((N1 + TC - 1) / TC, (M + TR - 1) / TR), dim3(NTHR), kLds, (hipStream_t)stream, l_in, ln_w,
                       ln_b, eps, W1, b1, d, l_ln, a, c, z, stats);
    return (int)hipGetLastError();
}

extern "C" int zira_text_prep_bwd_f32(const float *g_a, const float *g_c, const float *g_z, const float *g_l_ln, const float *l_in,
                                      const float *ln_w, const float *stats, const float *W1T, int B, int T, int H, int Dv, int Dl,
                                      float *scratch, float *g_l_in, void *stream)
{
    if (!l_in || !ln_w || !stats || !W1T || !scratch || !g_l_in || bad_dims(B, T, H, Dv, Dl)) return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T, N1 = 2 * H * Dv + H, parts = (N1 + KC - 1) / KC;
    hipLaunchKernelGGL(text_prep_bwd_kernel, dim3((Dl + TC - 1) / TC, parts, (M + TR - 1) / TR), dim3(NTHR), kLds, (hipStream_t)stream, g_a,
                       g_c, g_z, W1T, d, scratch);
    hipLaunchKernelGGL(text_prep_bwd_finish_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, scratch, parts, g_l_ln, l_in, ln_w,
                       stats, M, Dl, g_l_in);
    return (int)hipGetLastError();
}

extern "C" int zira_text_out_fwd_f32(const float *u, const float *colsum, const float *l_ln, const float *O, const float *o0,
                                     const float *gamma, const float *keep, int B, int T, int H, int Dv, int Dl, float *scratch,
                                     float *out, void *stream)
{
    if (!u || !colsum || !l_ln || !O || !o0 || !gamma || !scratch || !out || bad_dims(B, T, H, Dv, Dl)) return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T, HD = H * Dv, parts = (HD + KC - 1) / KC;
    hipLaunchKernelGGL(text_out_fwd_kernel, dim3((Dl + TC - 1) / TC, parts, (M + TR - 1) / TR), dim3(NTHR), kLds, (hipStream_t)stream, u,
                       colsum, O, d, scratch);
    const long long n = (long long)M * Dl;
    hipLaunchKernelGGL(text_out_fwd_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scratch, parts, l_ln,
                       o0, gamma, keep, M, T, Dl, out);
    return (int)hipGetLastError();
}

extern "C" int zira_text_out_bwd_f32(const float *g, const float *u, const float *colsum, const float *OT, const float *gamma,
                                     const float *keep, int B, int T, int H, int Dv, int Dl, float *g_u, float *g_colsum, void *stream)
{
    if (!g || !u || !colsum || !OT || !gamma || !g_u || !g_colsum || bad_dims(B, T, H, Dv, Dl)) return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T, HD = H * Dv, rows = B * H * T;
    hipLaunchKernelGGL(text_out_bwd_kernel, dim3((HD + TC - 1) / TC, (M + TR - 1) / TR), dim3(NTHR), kLds, (hipStream_t)stream, g, colsum, OT,
                       gamma, keep, d, g_u);
    hipLaunchKernelGGL(text_colsum_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, g_u, u, colsum, rows, Dv, g_colsum);
    return (int)hipGetLastError();
}
