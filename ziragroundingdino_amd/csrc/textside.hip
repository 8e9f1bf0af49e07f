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
// Here: one small-tile fp32 GEMM kernel whose operand loads and result stores go through index maps (the layouts above are
// read and written in place), 8 x 8 outputs x 4 K-slices per block so that a product with M = 64 rows still fills the chip.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int TM = 8, TN = 8, KS = 4, KC = 128;   // block: 8 x 8 outputs, 4 K-slices of 32 per chunk of 128

struct Dims {
    int B, T, H, Dv, Dl;   // rows M = B T; text width Dl; image width Dv per head
};

// C[m][n] = sum_k X(m, k) W(k, n) for the block's 8 x 8 tile; X / W are callables (m, k) / (k, n) -> float that return 0
// outside the problem; the result goes to E(m, n, sum) for m < M, n < N.
template <class XF, class WF, class EF>
__device__ __forceinline__ void tile_gemm(int M, int N, int K, XF X, WF W, EF E)
{
    __shared__ float Xs[TM][KC + 1];
    __shared__ float Ws[KC][TN + 1];
    __shared__ float red[KS][TM * TN];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int ks = tid >> 6, o = tid & 63, om = o >> 3, on = o & 7;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
        for (int i = tid; i < TM * KC; i += 256) {
            const int m = i / KC, k = i - m * KC;
            Xs[m][k] = (m0 + m < M && k0 + k < K) ? X(m0 + m, k0 + k) : 0.f;
        }
        for (int i = tid; i < KC * TN; i += 256) {
            const int k = i / TN, n = i - k * TN;
            Ws[k][n] = (n0 + n < N && k0 + k < K) ? W(k0 + k, n0 + n) : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = ks * (KC / KS); k < (ks + 1) * (KC / KS); ++k) acc = fmaf(Xs[om][k], Ws[k][on], acc);
    }
    red[ks][o] = acc;
    __syncthreads();
    if (ks == 0) {
        const float s = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
        if (m0 + om < M && n0 + on < N) E(m0 + om, n0 + on, s);
    }
}

// ---- prep forward: LayerNorm in the operand load (row statistics first), outputs scattered into a / c / z ----------------------
__global__ __launch_bounds__(256) void text_prep_fwd_kernel(const float *__restrict__ l_in, const float *__restrict__ ln_w,
                                                            const float *__restrict__ ln_b, float eps, const float *__restrict__ W1,
                                                            const float *__restrict__ b1, Dims d, float *__restrict__ l_ln,
                                                            float *__restrict__ a, float *__restrict__ c, float *__restrict__ z,
                                                            float *__restrict__ stats)
{
    __shared__ float st[TM][2];
    const int M = d.B * d.T, HD = d.H * d.Dv, N1 = 2 * HD + d.H;
    const int m0 = blockIdx.y * TM;
    {   // mean / rstd of the block's 8 rows: 32 threads per row, two passes
        const int r = threadIdx.x >> 5, j = threadIdx.x & 31, m = m0 + r;
        float s = 0.f;
        if (m < M)
            for (int k = j; k < d.Dl; k += 32) s += l_in[(size_t)m * d.Dl + k];
#pragma unroll
        for (int w = 16; w > 0; w >>= 1) s += __shfl_xor(s, w);
        const float mean = s / (float)d.Dl;
        float v = 0.f;
        if (m < M)
            for (int k = j; k < d.Dl; k += 32) {
                const float t = l_in[(size_t)m * d.Dl + k] - mean;
                v = fmaf(t, t, v);
            }
#pragma unroll
        for (int w = 16; w > 0; w >>= 1) v += __shfl_xor(v, w);
        const float rstd = rsqrtf(v / (float)d.Dl + eps);
        if (j == 0) {
            st[r][0] = mean;
            st[r][1] = rstd;
            if (blockIdx.x == 0 && m < M) {
                stats[2 * m] = mean;
                stats[2 * m + 1] = rstd;
            }
        }
    }
    __syncthreads();
    if (blockIdx.x == 0) {   // the normalised rows themselves (the residual of the block's text output)
        for (int i = threadIdx.x; i < TM * d.Dl; i += 256) {
            const int r = i / d.Dl, k = i - r * d.Dl, m = m0 + r;
            if (m < M) l_ln[(size_t)m * d.Dl + k] = fmaf((l_in[(size_t)m * d.Dl + k] - st[r][0]) * st[r][1], ln_w[k], ln_b[k]);
        }
    }
    auto X = [&](int m, int k) { return fmaf((l_in[(size_t)m * d.Dl + k] - st[m - m0][0]) * st[m - m0][1], ln_w[k], ln_b[k]); };
    auto W = [&](int k, int n) { return W1[(size_t)k * N1 + n]; };
    auto E = [&](int m, int n, float s) {
        s += b1[n];
        const int b = m / d.T, t = m - b * d.T;
        if (n < HD) {
            const int h = n / d.Dv, dd = n - h * d.Dv;
            a[((size_t)b * d.Dv + dd) * (d.H * d.T) + h * d.T + t] = s;
        } else if (n < HD + d.H) {
            c[(size_t)b * d.H * d.T + (n - HD) * d.T + t] = s;
        } else {
            const int q = n - HD - d.H, h = q / d.Dv, dd = q - h * d.Dv;
            z[((size_t)b * d.H * d.T + h * d.T + t) * d.Dv + dd] = s;
        }
    };
    tile_gemm(M, N1, d.Dl, X, W, E);
}

// ---- prep backward, first half: g_ln[m][k] = sum_n G(m, n) W1[k][n], G gathered from the gradients of a / c / z ----------------
__global__ __launch_bounds__(256) void text_prep_bwd_kernel(const float *__restrict__ g_a, const float *__restrict__ g_c,
                                                            const float *__restrict__ g_z, const float *__restrict__ W1, Dims d,
                                                            float *__restrict__ g_ln)
{
    const int M = d.B * d.T, HD = d.H * d.Dv, N1 = 2 * HD + d.H;
    auto X = [&](int m, int n) {
        const int b = m / d.T, t = m - b * d.T;
        if (n < HD) {
            const int h = n / d.Dv, dd = n - h * d.Dv;
            return g_a ? g_a[((size_t)b * d.Dv + dd) * (d.H * d.T) + h * d.T + t] : 0.f;
        }
        if (n < HD + d.H) return g_c ? g_c[(size_t)b * d.H * d.T + (n - HD) * d.T + t] : 0.f;
        const int q = n - HD - d.H, h = q / d.Dv, dd = q - h * d.Dv;
        return g_z ? g_z[((size_t)b * d.H * d.T + h * d.T + t) * d.Dv + dd] : 0.f;
    };
    auto W = [&](int n, int k) { return W1[(size_t)k * N1 + n]; };
    auto E = [&](int m, int k, float s) { g_ln[(size_t)m * d.Dl + k] = s; };
    tile_gemm(M, d.Dl, N1, X, W, E);
}

// ---- prep backward, second half: LayerNorm backward of (g_ln + g_res) on the saved statistics; a wave per row ------------------
__global__ __launch_bounds__(256) void text_ln_bwd_kernel(const float *__restrict__ g_ln, const float *__restrict__ g_res,
                                                          const float *__restrict__ l_in, const float *__restrict__ ln_w,
                                                          const float *__restrict__ stats, int M, int Dl, float *__restrict__ g_in)
{
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= M) return;
    const float mean = stats[2 * m], rstd = stats[2 * m + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < Dl; k += 64) {
        const float g = (g_ln[(size_t)m * Dl + k] + (g_res ? g_res[(size_t)m * Dl + k] : 0.f)) * ln_w[k];
        const float xh = (l_in[(size_t)m * Dl + k] - mean) * rstd;
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        s1 += __shfl_xor(s1, w);
        s2 += __shfl_xor(s2, w);
    }
    const float inv = 1.f / (float)Dl;
    for (int k = lane; k < Dl; k += 64) {
        const float g = (g_ln[(size_t)m * Dl + k] + (g_res ? g_res[(size_t)m * Dl + k] : 0.f)) * ln_w[k];
        const float xh = (l_in[(size_t)m * Dl + k] - mean) * rstd;
        g_in[(size_t)m * Dl + k] = rstd * (g - inv * s1 - xh * (inv * s2));
    }
}

// scale of the text residual: gamma[n] (layer scale) times the per-sample stochastic-depth factor keep[b] when drawn
__device__ __forceinline__ float scale_of(const float *gamma, const float *keep, int b, int n)
{
    return keep ? gamma[n] * keep[b] : gamma[n];
}

// ---- out forward: out[m][n] = l_ln[m][n] + scale (o0[n] + sum_k (u / colsum)(m, k) O[k][n]) -------------------------------------
__global__ __launch_bounds__(256) void text_out_fwd_kernel(const float *__restrict__ u, const float *__restrict__ colsum,
                                                           const float *__restrict__ l_ln, const float *__restrict__ O,
                                                           const float *__restrict__ o0, const float *__restrict__ gamma,
                                                           const float *__restrict__ keep, Dims d, float *__restrict__ out)
{
    const int M = d.B * d.T, HD = d.H * d.Dv;
    auto X = [&](int m, int k) {
        const int b = m / d.T, t = m - b * d.T, h = k / d.Dv, dd = k - h * d.Dv;
        const size_t row = (size_t)b * d.H * d.T + h * d.T + t;
        return u[row * d.Dv + dd] / colsum[row];
    };
    auto W = [&](int k, int n) { return O[(size_t)k * d.Dl + n]; };
    auto E = [&](int m, int n, float s) {
        out[(size_t)m * d.Dl + n] = fmaf(scale_of(gamma, keep, m / d.T, n), s + o0[n], l_ln[(size_t)m * d.Dl + n]);
    };
    tile_gemm(M, d.Dl, HD, X, W, E);
}

// ---- out backward: g_u(m, k) = (sum_n scale g[m][n] O[k][n]) / colsum, written in u's layout ------------------------------------
__global__ __launch_bounds__(256) void text_out_bwd_kernel(const float *__restrict__ g, const float *__restrict__ colsum,
                                                           const float *__restrict__ O, const float *__restrict__ gamma,
                                                           const float *__restrict__ keep, Dims d, float *__restrict__ g_u)
{
    const int M = d.B * d.T, HD = d.H * d.Dv;
    auto X = [&](int m, int n) { return g[(size_t)m * d.Dl + n] * scale_of(gamma, keep, m / d.T, n); };
    auto W = [&](int n, int k) { return O[(size_t)k * d.Dl + n]; };
    auto E = [&](int m, int k, float s) {
        const int b = m / d.T, t = m - b * d.T, h = k / d.Dv, dd = k - h * d.Dv;
        const size_t row = (size_t)b * d.H * d.T + h * d.T + t;
        g_u[row * d.Dv + dd] = s / colsum[row];
    };
    tile_gemm(M, HD, d.Dl, X, W, E);
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

inline bool bad_dims(int B, int T, int H, int Dv, int Dl)
{
    return B <= 0 || T <= 0 || H <= 0 || Dv <= 0 || Dl <= 0 || (long long)B * T > (1 << 20) || (long long)H * Dv > (1 << 20);
}

}  // namespace

extern "C" int zira_text_prep_fwd_f32(const float *l_in, const float *ln_w, const float *ln_b, float eps, const float *W1,
                                      const float *b1, int B, int T, int H, int Dv, int Dl, float *l_ln, float *a, float *c, float *z,
                                      float *stats, void *stream)
{
    if (!l_in || !ln_w || !ln_b || !W1 || !b1 || !l_ln || !a || !c || !z || !stats || bad_dims(B, T, H, Dv, Dl))
        return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T, N1 = 2 * H * Dv + H;
    hipLaunchKernelGGL(text_prep_fwd_kernel, dim3((N1 + TN - 1) / TN, (M + TM - 1) / TM), dim3(256), 0, (hipStream_t)stream, l_in, ln_w,
                       ln_b, eps, W1, b1, d, l_ln, a, c, z, stats);
    return (int)hipGetLastError();
}

extern "C" int zira_text_prep_bwd_f32(const float *g_a, const float *g_c, const float *g_z, const float *g_l_ln, const float *l_in,
                                      const float *ln_w, const float *stats, const float *W1, int B, int T, int H, int Dv, int Dl,
                                      float *scratch, float *g_l_in, void *stream)
{
    if (!l_in || !ln_w || !stats || !W1 || !scratch || !g_l_in || bad_dims(B, T, H, Dv, Dl)) return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T;
    hipLaunchKernelGGL(text_prep_bwd_kernel, dim3((Dl + TN - 1) / TN, (M + TM - 1) / TM), dim3(256), 0, (hipStream_t)stream, g_a, g_c, g_z,
                       W1, d, scratch);
    hipLaunchKernelGGL(text_ln_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, scratch, g_l_ln, l_in, ln_w, stats, M, Dl,
                       g_l_in);
    return (int)hipGetLastError();
}

extern "C" int zira_text_out_fwd_f32(const float *u, const float *colsum, const float *l_ln, const float *O, const float *o0,
                                     const float *gamma, const float *keep, int B, int T, int H, int Dv, int Dl, float *out,
                                     void *stream)
{
    if (!u || !colsum || !l_ln || !O || !o0 || !gamma || !out || bad_dims(B, T, H, Dv, Dl)) return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T;
    hipLaunchKernelGGL(text_out_fwd_kernel, dim3((Dl + TN - 1) / TN, (M + TM - 1) / TM), dim3(256), 0, (hipStream_t)stream, u, colsum, l_ln,
                       O, o0, gamma, keep, d, out);
    return (int)hipGetLastError();
}

extern "C" int zira_text_out_bwd_f32(const float *g, const float *u, const float *colsum, const float *O, const float *gamma,
                                     const float *keep, int B, int T, int H, int Dv, int Dl, float *g_u, float *g_colsum, void *stream)
{
    if (!g || !u || !colsum || !O || !gamma || !g_u || !g_colsum || bad_dims(B, T, H, Dv, Dl)) return (int)hipErrorInvalidValue;
    const Dims d{B, T, H, Dv, Dl};
    const int M = B * T, HD = H * Dv, rows = B * H * T;
    hipLaunchKernelGGL(text_out_bwd_kernel, dim3((HD + TN - 1) / TN, (M + TM - 1) / TM), dim3(256), 0, (hipStream_t)stream, g, colsum, O,
                       gamma, keep, d, g_u);
    hipLaunchKernelGGL(text_colsum_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, g_u, u, colsum, rows, Dv, g_colsum);
    return (int)hipGetLastError();
}
