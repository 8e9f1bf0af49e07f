// groupnorm.hip -- GroupNorm of the four input projections (reference groundingdino_dual_zero_rep_branch.py:487-529: nn.GroupNorm(32, 256)
// behind each level's conv + side branch), forward and the input gradient, for gfx950 (MI355X), float32, NCHW.
//
// ATen's native_group_norm takes 106 us for the first level's 34 MB (2 x 256 x 100 x 167) and its backward 46 us: a (sample, group) is
// ONE contiguous run of (C / G) * H * W floats, reduced by a single workgroup row there.  Here a run is cut into S slices, one
// block each (about a thousand blocks per launch):
//   forward   k1: per slice (count, mean, M2) -- the mean first, then the squared distances to it (the slice is read twice, the
//                 second time from L2) -- to the workspace;  k2: every block joins the S partial triples (Chan's pairwise update,
//                 in double, slice order: the same bits in every block) and normalises its slice.  The optional second input is
//                 added on the way in (conv output + side branch, `main + branch` of _project_level) and the sum written out for the
//                 backward.
//   backward  k1: per slice s1 = sum(dy gamma), s2 = sum(dy gamma xhat);  k2: dx = rstd (dy gamma - s1 / n - xhat s2 / n).
//             gamma and beta are frozen under the ZiRa freeze (:722-745): no parameter gradients (the Python side falls back to
//             ATen when they are wanted).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // (red may still be read from the previous call)
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

struct Geom {
    int C, HW, G, S;
    long run;      // floats of a (sample, group): (C / G) * HW, a multiple of 4
    long slice;    // floats of a slice, a multiple of 4
};

__device__ __forceinline__ void slice_of(const Geom &g, long &base, long &j0, long &j1, int &bg, int &s)
{
    bg = blockIdx.x / g.S;
    s = blockIdx.x - bg * g.S;
    base = (long)bg * g.run;               // (b * C + grp * C / G) * HW: the runs of a tensor lie back to back
    j0 = (long)s * g.slice;
    j1 = j0 + g.slice < g.run ? j0 + g.slice : g.run;
}

// channels of the four elements j .. j + 3 of a run (HW >= 4: at most one channel border inside a float4)
__device__ __forceinline__ void channels_of(long j, int HW, int c0, int (&c)[4])
{
    const int jj = (int)j, q = jj / HW, r = jj - q * HW;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = c0 + q + (r + e >= HW ? 1 : 0);
}

__device__ __forceinline__ float4 load_sum(const float *x, const float *res, long at)
{
    float4 v = *reinterpret_cast<const float4 *>(x + at);
    if (res) {
        const float4 r = *reinterpret_cast<const float4 *>(res + at);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    return v;
}

__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const float *__restrict__ x, const float *__restrict__ res, Geom g,
                                                           float *__restrict__ ws)
{
    __shared__ float red[4];
    long base, j0, j1;
    int bg, s;
    slice_of(g, base, j0, j1, bg, s);
    float sum = 0.f;
    for (long j = j0 + 4 * threadIdx.x; j < j1; j += 4 * kThreads) {
        const float4 v = load_sum(x, res, base + j);
        sum += (v.x + v.y) + (v.z + v.w);
    }
    const float n = (float)(j1 > j0 ? j1 - j0 : 0);
    const float mean = n > 0.f ? block_sum(sum, red) / n : 0.f;
    float m2 = 0.f;
    for (long j = j0 + 4 * threadIdx.x; j < j1; j += 4 * kThreads) {
        const float4 v = load_sum(x, res, base + j);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        m2 += (a * a + b * b) + (c * c + d * d);
    }
    m2 = block_sum(m2, red);
    if (threadIdx.x == 0) {
        float *o = ws + (size_t)blockIdx.x * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

// the S partial triples of a run, joined in slice order (every block of the run computes the same bits)
__device__ __forceinline__ void join_stats(const float *ws, int bg, int S, float eps, float &mean, float &rstd)
{
    double n = 0.0, mu = 0.0, m2 = 0.0;
    for (int s = 0; s < S; ++s) {
        const float *p = ws + ((size_t)bg * S + s) * 3;
        const double nb = p[0], mb = p[1], qb = p[2];
        if (nb > 0.0) {
            const double d = mb - mu, nn = n + nb;
            mu += d * nb / nn;
            m2 += qb + d * d * n * nb / nn;
            n = nn;
        }
    }
    mean = (float)mu;
    rstd = rsqrtf((float)(m2 / n) + eps);
}

__global__ __launch_bounds__(kThreads) void gn_norm_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, Geom g, float eps,
                                                          const float *__restrict__ ws, float *__restrict__ sum_out, float *__restrict__ y,
                                                          float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    long base, j0, j1;
    int bg, s;
    slice_of(g, base, j0, j1, bg, s);
    float mean, rstd;
    join_stats(ws, bg, g.S, eps, mean, rstd);
    if (s == 0 && threadIdx.x == 0) {
        mean_out[bg] = mean;
        rstd_out[bg] = rstd;
    }
    const int c0 = (bg % g.G) * (g.C / g.G);
    for (long j = j0 + 4 * threadIdx.x; j < j1; j += 4 * kThreads) {
        const float4 v = load_sum(x, res, base + j);
        if (res) *reinterpret_cast<float4 *>(sum_out + base + j) = v;
        const float in[4] = {v.x, v.y, v.z, v.w};
        float o[4];
        int c[4];
        channels_of(j, g.HW, c0, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = gamma ? gamma[c[e]] : 1.f, b = beta ? beta[c[e]] : 0.f;
            o[e] = (in[e] - mean) * rstd * w + b;
        }
        *reinterpret_cast<float4 *>(y + base + j) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(kThreads) void gn_bwd_sums_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                              const float *__restrict__ gamma, const float *__restrict__ mean,
                                                              const float *__restrict__ rstd, Geom g, float *__restrict__ ws)
{
    __shared__ float red[4];
    long base, j0, j1;
    int bg, s;
    slice_of(g, base, j0, j1, bg, s);
    const float mu = mean[bg], rs = rstd[bg];
    const int c0 = (bg % g.G) * (g.C / g.G);
    float s1 = 0.f, s2 = 0.f;
    for (long j = j0 + 4 * threadIdx.x; j < j1; j += 4 * kThreads) {
        const float4 d = *reinterpret_cast<const float4 *>(dy + base + j), v = *reinterpret_cast<const float4 *>(x + base + j);
        const float dd[4] = {d.x, d.y, d.z, d.w}, vv[4] = {v.x, v.y, v.z, v.w};
        int c[4];
        channels_of(j, g.HW, c0, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = gamma ? gamma[c[e]] : 1.f;
            const float t = dd[e] * w;
            s1 += t;
            s2 += t * ((vv[e] - mu) * rs);
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        ws[(size_t)blockIdx.x * 2] = s1;
        ws[(size_t)blockIdx.x * 2 + 1] = s2;
    }
}

__global__ __launch_bounds__(kThreads) void gn_bwd_dx_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, Geom g, const float *__restrict__ ws,
                                                            float *__restrict__ dx)
{
    long base, j0, j1;
    int bg, s;
    slice_of(g, base, j0, j1, bg, s);
    double a1 = 0.0, a2 = 0.0;
    for (int t = 0; t < g.S; ++t) {
        a1 += ws[((size_t)bg * g.S + t) * 2];
        a2 += ws[((size_t)bg * g.S + t) * 2 + 1];
    }
    const float inv_n = 1.f / (float)g.run;
    const float m1 = (float)a1 * inv_n, m2 = (float)a2 * inv_n;
    const float mu = mean[bg], rs = rstd[bg];
    const int c0 = (bg % g.G) * (g.C / g.G);
    for (long j = j0 + 4 * threadIdx.x; j < j1; j += 4 * kThreads) {
        const float4 d = *reinterpret_cast<const float4 *>(dy + base + j), v = *reinterpret_cast<const float4 *>(x + base + j);
        const float dd[4] = {d.x, d.y, d.z, d.w}, vv[4] = {v.x, v.y, v.z, v.w};
        float o[4];
        int c[4];
        channels_of(j, g.HW, c0, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = gamma ? gamma[c[e]] : 1.f;
            const float xh = (vv[e] - mu) * rs;
            o[e] = rs * (dd[e] * w - m1 - xh * m2);
        }
        *reinterpret_cast<float4 *>(dx + base + j) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

inline bool make_geom(int B, int C, int HW, int G, Geom &g)
{
    if (B <= 0 || C <= 0 || HW < 4 || G <= 0 || C % G || (long)(C / G) * HW >= (1l << 30)) return false;
    g.C = C; g.HW = HW; g.G = G;
    g.run = (long)(C / G) * HW;
    if (g.run % 4) return false;
    int S = 1024 / (B * G);
    S = S < 1 ? 1 : (S > 64 ? 64 : S);
    while (S > 1 && g.run / S < 4 * kThreads) --S;       // (a slice keeps every thread of its block busy at least once)
    g.S = S;
    g.slice = ((g.run + S - 1) / S + 3) / 4 * 4;
    return true;
}

}  // namespace

extern "C" size_t zira_groupnorm_workspace_floats(int B, int C, int HW, int G)
{
    Geom g;
    if (!make_geom(B, C, HW, G, g)) return 0;
    return (size_t)B * G * g.S * 3;
}

// y = GroupNorm(x (+ res)) over [B, C, HW] with G groups; sum_out (needed with res) receives x + res; mean / rstd [B * G]
extern "C" int zira_groupnorm_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, int B, int C, int HW, int G,
                                      float eps, float *sum_out, float *y, float *mean, float *rstd, float *workspace, void *stream_)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    Geom g;
    if (!x || !y || !mean || !rstd || !workspace || (res && !sum_out) || !make_geom(B, C, HW, G, g)) return -1;
    if (((uintptr_t)x | (uintptr_t)res | (uintptr_t)y | (uintptr_t)sum_out) & 15) return -1;
    const dim3 grid((unsigned)(B * G * g.S));
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(kThreads), 0, st, x, res, g, workspace);
    hipLaunchKernelGGL(gn_norm_kernel, grid, dim3(kThreads), 0, st, x, res, gamma, beta, g, eps, workspace, sum_out, y, mean, rstd);
    return (int)hipGetLastError();
}

// dx of y = GroupNorm(x) for frozen gamma / beta; x = the normalised input (sum_out of the forward when it added two)
extern "C" int zira_groupnorm_bwd_f32(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd, int B, int C,
                                      int HW, int G, float *dx, float *workspace, void *stream_)
{
    hipStream_t st = reinterpret_cast<hipStream_t>(stream_);
    Geom g;
    if (!dy || !x || !mean || !rstd || !dx || !workspace || !make_geom(B, C, HW, G, g)) return -1;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) return -1;
    const dim3 grid((unsigned)(B * G * g.S));
    hipLaunchKernelGGL(gn_bwd_sums_kernel, grid, dim3(kThreads), 0, st, dy, x, gamma, mean, rstd, g, workspace);
    hipLaunchKernelGGL(gn_bwd_dx_kernel, grid, dim3(kThreads), 0, st, dy, x, gamma, mean, rstd, g, workspace, dx);
    return (int)hipGetLastError();
}
