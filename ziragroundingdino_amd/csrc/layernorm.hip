// layernorm.hip -- row LayerNorm forward for gfx950 (fp32), for the wide activations of the step.
//
// The cross-modal encoder normalises [B*S, 256] activations (S = 22223 pixels) twice per layer
// (reference transformer.py DeformableTransformerEncoderLayer.forward, norm1 / norm2, and
// fuse_modules.py BiAttentionBlock layer_norm_v) and the frozen Swin-T normalises [B*H*W, 96..768]
// rows (swin_transformer.py SwinTransformerBlock norm1 / norm2, PatchMerging.norm).  ATen's
// vectorized_layer_norm_kernel spends a 256-thread block per row whatever the row length: 47 us for
// 44446 x 256 (1.9 TB/s) and 134 us for 133600 x 96 (0.76 TB/s) on MI355X.  The op is a pure
// stream (read x once, write y once): here a row is held in the registers of one lane group
// (32 or 64 lanes x float4), so it is read exactly once; mean and variance are two register passes
// (sum, then sum of squared deviations -- the same two-pass formula as the fp32 reference, not
// E[x^2] - mean^2), reduced with DPP / permlane swaps; no LDS, no barriers.
//
// Outputs y, mean[rows], rstd[rows] (what aten::native_layer_norm returns).  The input gradient is the
// same kind of stream (read dy and x, write dx) and has the same layout (ln_bwd_rows); the parameter
// gradients (column sums over all rows) stay with aten::native_layer_norm_backward -- the ZiRa
// fine-tune freezes every LayerNorm, so they are not computed at all on the path.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int kThreads = 256;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}

// sum over the G = 32 or 64 lanes of a group (aligned in the wave); every lane ends with the total
template <int G>
__device__ __forceinline__ float group_sum(float x)
{
    x = dpp_add<0xB1>(x);   // quad_perm:[1,0,3,2]
    x = dpp_add<0x4E>(x);   // quad_perm:[2,3,0,1]
    x = dpp_add<0x141>(x);  // row_half_mirror
    x = dpp_add<0x140>(x);  // row_mirror
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    if (G == 64) {
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    return x;
}

// One row per group of G lanes; lane j of the group holds the float4s j, j + G, ... (NV of them) of the row.
template <int G, int NV>
__global__ __launch_bounds__(kThreads) void ln_fwd_rows(
    const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
    long rows, int C, float eps, float *__restrict__ y, float *__restrict__ mean,
    float *__restrict__ rstd, const float *__restrict__ res, float *__restrict__ sum_out)
{
    // res != nullptr: the row that is normalised is x + res (the residual connection in front of a post-LN);
    // it is also written to sum_out, which the backward reads in place of x
    const int c4 = C >> 2;
    const int gl = threadIdx.x % G;
    const long groups_per_block = kThreads / G;
    const long g0 = (long)blockIdx.x * groups_per_block + threadIdx.x / G;
    const long gstride = (long)gridDim.x * groups_per_block;
    const float inv_c = 1.0f / (float)C;
    float4 w[NV], b[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = gl + k * G;
        w[k] = make_float4(1.f, 1.f, 1.f, 1.f);
        b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < c4) {
            if (gamma) w[k] = reinterpret_cast<const float4 *>(gamma)[i];
            if (beta) b[k] = reinterpret_cast<const float4 *>(beta)[i];
        }
    }
    for (long r = g0; r < rows; r += gstride) {
        const float4 *xr = reinterpret_cast<const float4 *>(x + r * C);
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = gl + k * G;
            v[k] = i < c4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (res && i < c4) {
                const float4 rv = reinterpret_cast<const float4 *>(res + r * C)[i];
                v[k] = make_float4(v[k].x + rv.x, v[k].y + rv.y, v[k].z + rv.z, v[k].w + rv.w);
                reinterpret_cast<float4 *>(sum_out + r * C)[i] = v[k];
            }
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        const float mu = group_sum<G>(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = gl + k * G;
            if (i < c4) {
                const float dx = v[k].x - mu, dy = v[k].y - mu, dz = v[k].z - mu, dw = v[k].w - mu;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float var = group_sum<G>(q) * inv_c;
        const float rs = rsqrtf(var + eps);
        float4 *yr = reinterpret_cast<float4 *>(y + r * C);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = gl + k * G;
            if (i < c4) {
                float4 o;
                o.x = (v[k].x - mu) * rs * w[k].x + b[k].x;
                o.y = (v[k].y - mu) * rs * w[k].y + b[k].y;
                o.z = (v[k].z - mu) * rs * w[k].z + b[k].z;
                o.w = (v[k].w - mu) * rs * w[k].w + b[k].w;
                yr[i] = o;
            }
        }
        if (gl == 0) {
            if (mean) mean[r] = mu;
            if (rstd) rstd[r] = rs;
        }
    }
}

// Input gradient of the same LayerNorm: with xhat = (x - mean) * rstd and g = dy * gamma,
// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)).  Same row-in-registers layout (x and dy).
template <int G, int NV>
__global__ __launch_bounds__(kThreads) void ln_bwd_rows(
    const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ gamma,
    const float *__restrict__ mean, const float *__restrict__ rstd, long rows, int C,
    float *__restrict__ dx)
{
    const int c4 = C >> 2;
    const int gl = threadIdx.x % G;
    const long groups_per_block = kThreads / G;
    const long g0 = (long)blockIdx.x * groups_per_block + threadIdx.x / G;
    const long gstride = (long)gridDim.x * groups_per_block;
    const float inv_c = 1.0f / (float)C;
    float4 w[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = gl + k * G;
        w[k] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (i < c4 && gamma) w[k] = reinterpret_cast<const float4 *>(gamma)[i];
    }
    for (long r = g0; r < rows; r += gstride) {
        const float4 *xr = reinterpret_cast<const float4 *>(x + r * C);
        const float4 *gr = reinterpret_cast<const float4 *>(dy + r * C);
        const float mu = mean[r], rs = rstd[r];
        float4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = gl + k * G;
            xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            g[k] = xh[k];
            if (i < c4) {
                const float4 xv = xr[i], gv = gr[i];
                xh[k] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[k] = make_float4(gv.x * w[k].x, gv.y * w[k].y, gv.z * w[k].z, gv.w * w[k].w);
            }
            s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
            s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
        }
        const float m1 = group_sum<G>(s1) * inv_c;
        const float m2 = group_sum<G>(s2) * inv_c;
        float4 *dr = reinterpret_cast<float4 *>(dx + r * C);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = gl + k * G;
            if (i < c4) {
                float4 o;
                o.x = rs * (g[k].x - m1 - xh[k].x * m2);
                o.y = rs * (g[k].y - m1 - xh[k].y * m2);
                o.z = rs * (g[k].z - m1 - xh[k].z * m2);
                o.w = rs * (g[k].w - m1 - xh[k].w * m2);
                dr[i] = o;
            }
        }
    }
}

inline unsigned ln_blocks(long rows, long groups_per_block)
{
    long blocks = (rows + groups_per_block - 1) / groups_per_block;
    const long cap = 256L * 8 * 4;  // 8 blocks per CU resident, a few rounds; grid-stride beyond
    return (unsigned)(blocks > cap ? cap : blocks);
}

template <int G, int NV>
int launch_ln_bwd(const float *dy, const float *x, const float *gamma, const float *mean,
                  const float *rstd, long rows, int C, float *dx, hipStream_t st)
{
    hipLaunchKernelGGL((ln_bwd_rows<G, NV>), dim3(ln_blocks(rows, kThreads / G)), dim3(kThreads), 0, st, dy,
                       x, gamma, mean, rstd, rows, C, dx);
    return (int)hipGetLastError();
}

template <int G, int NV>
int launch_ln(const float *x, const float *gamma, const float *beta, long rows, int C, float eps,
              float *y, float *mean, float *rstd, hipStream_t st, const float *res = nullptr,
              float *sum_out = nullptr)
{
    hipLaunchKernelGGL((ln_fwd_rows<G, NV>), dim3(ln_blocks(rows, kThreads / G)), dim3(kThreads), 0, st, x,
                       gamma, beta, rows, C, eps, y, mean, rstd, res, sum_out);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int zira_layernorm_fwd_f32(const float *x, const float *gamma, const float *beta, int64_t rows, int C,
                           float eps, float *y, float *mean, float *rstd, void *stream)
{
    if (!x || !y || rows < 0 || C <= 0 || (C & 3) || C > 1024) return ZIRA_MSDA_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return ZIRA_MSDA_EINVAL;
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int c4 = C >> 2;
    if (c4 <= 32) return launch_ln<32, 1>(x, gamma, beta, rows, C, eps, y, mean, rstd, st);
    if (c4 <= 64) return launch_ln<64, 1>(x, gamma, beta, rows, C, eps, y, mean, rstd, st);
    if (c4 <= 128) return launch_ln<64, 2>(x, gamma, beta, rows, C, eps, y, mean, rstd, st);
    if (c4 <= 192) return launch_ln<64, 3>(x, gamma, beta, rows, C, eps, y, mean, rstd, st);
    return launch_ln<64, 4>(x, gamma, beta, rows, C, eps, y, mean, rstd, st);
}

int zira_add_layernorm_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, int64_t rows,
                               int C, float eps, float *sum_out, float *y, float *mean, float *rstd, void *stream)
{
    if (!x || !res || !sum_out || !y || rows < 0 || C <= 0 || (C & 3) || C > 1024) return ZIRA_MSDA_EINVAL;
    if (((uintptr_t)x | (uintptr_t)res | (uintptr_t)sum_out | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15)
        return ZIRA_MSDA_EINVAL;
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int c4 = C >> 2;
    if (c4 <= 32) return launch_ln<32, 1>(x, gamma, beta, rows, C, eps, y, mean, rstd, st, res, sum_out);
    if (c4 <= 64) return launch_ln<64, 1>(x, gamma, beta, rows, C, eps, y, mean, rstd, st, res, sum_out);
    if (c4 <= 128) return launch_ln<64, 2>(x, gamma, beta, rows, C, eps, y, mean, rstd, st, res, sum_out);
    if (c4 <= 192) return launch_ln<64, 3>(x, gamma, beta, rows, C, eps, y, mean, rstd, st, res, sum_out);
    return launch_ln<64, 4>(x, gamma, beta, rows, C, eps, y, mean, rstd, st, res, sum_out);
}

int zira_layernorm_bwd_f32(const float *dy, const float *x, const float *gamma, const float *mean,
                           const float *rstd, int64_t rows, int C, float *dx, void *stream)
{
    if (!dy || !x || !mean || !rstd || !dx || rows < 0 || C <= 0 || (C & 3) || C > 1024) return ZIRA_MSDA_EINVAL;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma) & 15) return ZIRA_MSDA_EINVAL;
    if (rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int c4 = C >> 2;
    if (c4 <= 32) return launch_ln_bwd<32, 1>(dy, x, gamma, mean, rstd, rows, C, dx, st);
    if (c4 <= 64) return launch_ln_bwd<64, 1>(dy, x, gamma, mean, rstd, rows, C, dx, st);
    if (c4 <= 128) return launch_ln_bwd<64, 2>(dy, x, gamma, mean, rstd, rows, C, dx, st);
    if (c4 <= 192) return launch_ln_bwd<64, 3>(dy, x, gamma, mean, rstd, rows, C, dx, st);
    return launch_ln_bwd<64, 4>(dy, x, gamma, mean, rstd, rows, C, dx, st);
}

}  // extern "C"
