// rsb.hip -- epilogue of ZiRa's reparameterizable side branch (RSB) for gfx950.
//
// Reference semantics (groundingdino/models/GroundingDINO/groundingdino_dual_zero_rep_branch.py,
// RepZeroConv2d.forward :87-96 and RepZeroLinear.forward :119-128, training mode):
//     branch = scaling * F(x; W, b)            F = conv2d or linear
//     out    = branch + F(x; W_f, b_f)         (the slowly-learning "freeze" twin)
//     zl     = mean(smooth_l1(branch, 0)) + mean(smooth_l1(out, 0))      (beta = 1)
// The two F(.) are dense contractions and stay with the GEMM / convolution libraries (MFMA);
// everything after them is one pass over the activations here instead of ~10 elementwise /
// reduction launches: out and both loss sums in the forward, the three gradients in the
// backward.  HBM-bound: forward reads 2 and writes 1 float per element (12 B), backward
// reads 3 and writes 2 (20 B).
//
// Reductions are deterministic: per-block partial sums (wave DPP reduction, then LDS) are
// written to a scratch array and folded by a single block in a fixed order.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8; grid-stride beyond

__device__ __forceinline__ float smooth_l1_zero(float t)
{
    const float a = fabsf(t);
    return a < 1.f ? 0.5f * t * t : a - 0.5f;
}
__device__ __forceinline__ float smooth_l1_zero_grad(float t)
{
    return fabsf(t) < 1.f ? t : (t > 0.f ? 1.f : -1.f);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
// sum over the 64 lanes of a wave; every lane ends with the total
__device__ __forceinline__ float wave_sum(float x)
{
    x = dpp_add<0xB1>(x);   // quad_perm:[1,0,3,2]
    x = dpp_add<0x4E>(x);   // quad_perm:[2,3,0,1]
    x = dpp_add<0x141>(x);  // row_half_mirror
    x = dpp_add<0x140>(x);  // row_mirror
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// block-wide sum of up to 3 values -> thread 0 writes them to dst[0..NV)
template <int NV>
__device__ __forceinline__ void block_sum_store(float (&v)[NV], float *dst)
{
    __shared__ float red[NV][kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = wave_sum(v[i]);
        if (lane == 0) red[i][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) s += red[i][w];
            dst[i] = s;
        }
    }
}

__global__ __launch_bounds__(kThreads) void rsb_fwd_kernel(
    const float *__restrict__ yb, const float *__restrict__ yf, const float *__restrict__ scaling,
    size_t n, float *__restrict__ out, float *__restrict__ partial)
{
    const float s = scaling[0];
    float acc[2] = {0.f, 0.f};
    const size_t n4 = n >> 2;
    const float4 *yb4 = reinterpret_cast<const float4 *>(yb);
    const float4 *yf4 = reinterpret_cast<const float4 *>(yf);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
        const float4 b = yb4[i], f = yf4[i];
        float4 br = make_float4(s * b.x, s * b.y, s * b.z, s * b.w);
        float4 o = make_float4(br.x + f.x, br.y + f.y, br.z + f.z, br.w + f.w);
        out4[i] = o;
        acc[0] += smooth_l1_zero(br.x) + smooth_l1_zero(br.y) + smooth_l1_zero(br.z) + smooth_l1_zero(br.w);
        acc[1] += smooth_l1_zero(o.x) + smooth_l1_zero(o.y) + smooth_l1_zero(o.z) + smooth_l1_zero(o.w);
    }
    if (blockIdx.x == 0)  // scalar tail (n % 4 elements)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) {
            const float br = s * yb[i], o = br + yf[i];
            out[i] = o;
            acc[0] += smooth_l1_zero(br);
            acc[1] += smooth_l1_zero(o);
        }
    block_sum_store<2>(acc, partial + (size_t)blockIdx.x * 2);
}

// loss[0] = (sum of partial[.][0] + sum of partial[.][1]) / n
__global__ __launch_bounds__(kThreads) void rsb_fwd_finish(const float *__restrict__ partial,
                                                           int nblocks, float inv_n,
                                                           float *__restrict__ loss)
{
    float acc[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < nblocks; i += kThreads) {
        acc[0] += partial[i * 2];
        acc[1] += partial[i * 2 + 1];
    }
    float tot[2];
    __shared__ float sums[2];
    block_sum_store<2>(acc, sums);
    __syncthreads();
    if (threadIdx.x == 0) {
        tot[0] = sums[0]; tot[1] = sums[1];
        loss[0] = tot[0] * inv_n + tot[1] * inv_n;  // mean(sl1(branch)) + mean(sl1(out))
    }
}

__global__ __launch_bounds__(kThreads) void rsb_bwd_kernel(
    const float *__restrict__ yb, const float *__restrict__ yf, const float *__restrict__ scaling,
    const float *__restrict__ grad_out, const float *__restrict__ grad_loss, size_t n, float inv_n,
    float *__restrict__ g_yb, float *__restrict__ g_yf, float *__restrict__ partial)
{
    const float s = scaling[0];
    const float gl = grad_loss ? grad_loss[0] * inv_n : 0.f;
    float acc[1] = {0.f};
    const size_t n4 = n >> 2;
    const float4 *yb4 = reinterpret_cast<const float4 *>(yb);
    const float4 *yf4 = reinterpret_cast<const float4 *>(yf);
    const float4 *go4 = reinterpret_cast<const float4 *>(grad_out);
    float4 *gb4 = reinterpret_cast<float4 *>(g_yb);
    float4 *gf4 = reinterpret_cast<float4 *>(g_yf);
#define ZIRA_RSB_BWD_ELEM(B, F, GO, GB, GF)                         \
    {                                                               \
        const float br = s * (B), o = br + (F);                     \
        const float d_out = (GO) + gl * smooth_l1_zero_grad(o);     \
        const float d_br = d_out + gl * smooth_l1_zero_grad(br);    \
        (GF) = d_out;                                               \
        (GB) = s * d_br;                                            \
        acc[0] += (B) * d_br;                                       \
    }
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
        const float4 b = yb4[i], f = yf4[i];
        const float4 go = grad_out ? go4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gb, gf;
        ZIRA_RSB_BWD_ELEM(b.x, f.x, go.x, gb.x, gf.x)
        ZIRA_RSB_BWD_ELEM(b.y, f.y, go.y, gb.y, gf.y)
        ZIRA_RSB_BWD_ELEM(b.z, f.z, go.z, gb.z, gf.z)
        ZIRA_RSB_BWD_ELEM(b.w, f.w, go.w, gb.w, gf.w)
        gb4[i] = gb;
        gf4[i] = gf;
    }
    if (blockIdx.x == 0)
        for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += kThreads) {
            const float go = grad_out ? grad_out[i] : 0.f;
            ZIRA_RSB_BWD_ELEM(yb[i], yf[i], go, g_yb[i], g_yf[i])
        }
#undef ZIRA_RSB_BWD_ELEM
    block_sum_store<1>(acc, partial + blockIdx.x);
}

__global__ __launch_bounds__(kThreads) void rsb_bwd_finish(const float *__restrict__ partial,
                                                           int nblocks, float *__restrict__ g_scaling)
{
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < nblocks; i += kThreads) acc[0] += partial[i];
    block_sum_store<1>(acc, g_scaling);
}

inline int rsb_grid(size_t n)
{
    size_t blocks = ((n >> 2) + kThreads - 1) / kThreads;
    if (blocks < 1) blocks = 1;
    if (blocks > (size_t)kMaxBlocks) blocks = kMaxBlocks;
    return (int)blocks;
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

size_t zira_rsb_workspace_floats(size_t n)
{
    (void)n;
    return (size_t)kMaxBlocks * 2;
}

int zira_rsb_fwd_f32(const float *y_branch, const float *y_twin, const float *scaling, size_t n,
                     float *out, float *loss, float *workspace, void *stream)
{
    if (!y_branch || !y_twin || !scaling || !out || !loss || !workspace || n == 0 ||
        !aligned16(y_branch) || !aligned16(y_twin) || !aligned16(out))
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = rsb_grid(n);
    hipLaunchKernelGGL(rsb_fwd_kernel, dim3(grid), dim3(kThreads), 0, st, y_branch, y_twin, scaling,
                       n, out, workspace);
    hipLaunchKernelGGL(rsb_fwd_finish, dim3(1), dim3(kThreads), 0, st, workspace, grid,
                       1.0f / (float)n, loss);
    return (int)hipGetLastError();
}

int zira_rsb_bwd_f32(const float *y_branch, const float *y_twin, const float *scaling,
                     const float *grad_out, const float *grad_loss, size_t n, float *g_branch,
                     float *g_twin, float *g_scaling, float *workspace, void *stream)
{
    if (!y_branch || !y_twin || !scaling || !g_branch || !g_twin || !g_scaling || !workspace ||
        n == 0 || !aligned16(y_branch) || !aligned16(y_twin) || !aligned16(g_branch) ||
        !aligned16(g_twin) || (grad_out && !aligned16(grad_out)))
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = rsb_grid(n);
    hipLaunchKernelGGL(rsb_bwd_kernel, dim3(grid), dim3(kThreads), 0, st, y_branch, y_twin, scaling,
                       grad_out, grad_loss, n, 1.0f / (float)n, g_branch, g_twin, workspace);
    hipLaunchKernelGGL(rsb_bwd_finish, dim3(1), dim3(kThreads), 0, st, workspace, grid, g_scaling);
    return (int)hipGetLastError();
}

}  // extern "C"
