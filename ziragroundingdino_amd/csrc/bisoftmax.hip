// Fused score post-processing of the bi-directional image <-> text attention (reference
// groundingdino/models/GroundingDINO/fuse_modules.py:165-200), forward and backward, for gfx950.
//
//   x[b,n,h,t] = xm[b,n,h,t] + c[b,h,t]                 scores of image token n, head h, text token t
//   x1 = clamp(x - max(x))                              (stable_softmax_2d: ONE global maximum; clamps +-50000)
//   p_v[b,n,h,:] = softmax_t(x1 masked to -inf on padded text tokens)
//   x2 = clamp(x1 - max_n x1)                           per (b,h,t) column
//   e[b,n,h,t] = exp(x2), 0 on padded image tokens;     colsum[b,h,t] = sum_n e      (p_l = e / colsum)
//
// PyTorch runs this as ~300 small kernels per layer and direction (transposes, two full
// reductions, clamps, masked fills, two softmaxes: 870 us per layer at N = 22223, H*T = 64).  Here
// the image tokens are walked twice in the forward (column maxima; then rows) and once in the
// backward, with per-block partial column reductions folded by a second tiny kernel -- fixed
// order, no atomics.  The tensors keep the [B, N, H*T] layout of the GEMMs on either side.
//
// Gradient: softmax backward for p_v per (n, h); e passes (g_e + g_colsum) * e; clamps pass the
// gradient where they did not clip.  The paths through the two maxima are omitted: they vanish
// identically for p_v (softmax is shift invariant) and for p_l = e / colsum (the caller divides),
// which is the only way e and colsum are used.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileFloats = 4096;  // LDS tile: rows * HT <= 4096 floats
constexpr float kClamp = 50000.f;

__device__ __forceinline__ float clampf(float x, int clamp_lo, int clamp_hi)
{
    if (clamp_lo) x = fmaxf(x, -kClamp);
    if (clamp_hi) x = fminf(x, kClamp);
    return x;
}

// partial column maxima of xm over a chunk of rows: part[b][chunk][HT]
__global__ __launch_bounds__(kThreads) void bis_colmax_partial(const float *__restrict__ xm, int N,
                                                               int HT, int chunk_rows,
                                                               float *__restrict__ part)
{
    __shared__ float red[kThreads];
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int n0 = chunk * chunk_rows, n1 = min(n0 + chunk_rows, N);
    const float *xb = xm + (size_t)b * N * HT;
    // thread -> column (threadIdx % cols_in_flight), row phase (threadIdx / cols); HT may exceed 256
    for (int j0 = 0; j0 < HT; j0 += kThreads) {
        const int cols = min(HT - j0, kThreads);
        const int phases = kThreads / cols > 0 ? kThreads / cols : 1;
        const int j = threadIdx.x % cols, ph = threadIdx.x / cols;
        float m = -INFINITY;
        if (ph < phases)
            for (int n = n0 + ph; n < n1; n += phases) m = fmaxf(m, xb[(size_t)n * HT + j0 + j]);
        red[threadIdx.x] = m;
        __syncthreads();
        if (threadIdx.x < cols) {
            for (int p = 1; p < phases; ++p) m = fmaxf(m, red[p * cols + threadIdx.x]);
            part[((size_t)b * chunks + chunk) * HT + j0 + threadIdx.x] = m;
        }
        __syncthreads();
    }
}

// fold partials over chunks: out[b][j] = op over chunk of part[b][chunk][j] (+ addend[b][j]);
// op 0 = max, 1 = sum.  One block per batch element; the threads beyond the HT columns take every
// phases-th chunk each and LDS joins them in a fixed order.
__global__ __launch_bounds__(kThreads) void bis_fold(const float *__restrict__ part, int chunks, int HT,
                                                     int op, const float *__restrict__ addend,
                                                     float *__restrict__ out)
{
    __shared__ float red[kThreads];
    const int b = blockIdx.x;
    for (int j0 = 0; j0 < HT; j0 += kThreads) {
        const int cols = min(HT - j0, kThreads);
        const int phases = kThreads / cols;
        const int j = threadIdx.x % cols, ph = threadIdx.x / cols;
        float acc = op ? 0.f : -INFINITY;
        if (ph < phases) {
            const float *p = part + (size_t)b * chunks * HT + j0 + j;
            for (int c = ph; c < chunks; c += phases) {
                const float v = p[(size_t)c * HT];
                acc = op ? acc + v : fmaxf(acc, v);
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < cols) {
            for (int q = 1; q < phases; ++q) {
                const float v = red[q * cols + threadIdx.x];
                acc = op ? acc + v : fmaxf(acc, v);
            }
            const int i = b * HT + j0 + threadIdx.x;
            if (addend) acc += addend[i];
            out[i] = acc;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void bis_global_max(const float *__restrict__ colmax, int total,
                                                           float *__restrict__ gmax)
{
    __shared__ float red[kThreads];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < total; i += kThreads) m = fmaxf(m, colmax[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) gmax[0] = red[0];
}

// rows: p_v, e and the partial column sums of e
__global__ __launch_bounds__(kThreads) void bis_rows_fwd(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const uint8_t *__restrict__ mask_l /* [B,T] or null */,
    const uint8_t *__restrict__ mask_v /* [B,N] or null */, int N, int H, int T, int rows_per_block,
    int stable, int clamp_lo, int clamp_hi, float *__restrict__ pv, float *__restrict__ e,
    float *__restrict__ part_sum)
{
    extern __shared__ float tile[];  // [rows][HT]: x1, then e;  then colacc[HT]
    const int HT = H * T;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    float *colacc = tile + (size_t)rows_per_block * HT;
    for (int j = threadIdx.x; j < HT; j += kThreads) colacc[j] = 0.f;
    const float g = stable ? gmax[0] : 0.f;
    const float *cb = c + (size_t)b * HT;
    const int ntiles = (N + rows_per_block - 1) / rows_per_block;
    for (int tl = chunk; tl < ntiles; tl += chunks) {
    const int n0 = tl * rows_per_block, rows = min(rows_per_block, N - n0);
    const float *xb = xm + ((size_t)b * N + n0) * HT;
    __syncthreads();
    for (int i = threadIdx.x; i < rows * HT; i += kThreads)
        tile[i] = clampf(xb[i] + cb[i % HT] - g, clamp_lo, clamp_hi);  // x1
    __syncthreads();
    // one (row, head) per thread: softmax over the T text tokens
    float *pvb = pv + ((size_t)b * N + n0) * HT;
    for (int rh = threadIdx.x; rh < rows * H; rh += kThreads) {
        const int r = rh / H, h = rh - r * H;
        const float *x1 = tile + r * HT + h * T;
        float m = -INFINITY;
        for (int t = 0; t < T; ++t)
            if (!mask_l || !mask_l[b * T + t]) m = fmaxf(m, x1[t]);
        float s = 0.f;
        for (int t = 0; t < T; ++t)
            if (!mask_l || !mask_l[b * T + t]) s += expf(x1[t] - m);
        const float inv = 1.f / s;  // all text tokens masked: 0 * inf = nan, as torch.softmax of all -inf
        float *o = pvb + (size_t)r * HT + h * T;
        for (int t = 0; t < T; ++t) o[t] = (!mask_l || !mask_l[b * T + t]) ? expf(x1[t] - m) * inv : 0.f;
    }
    __syncthreads();
    // e = exp(clamp(x1 - colmax1)), 0 on padded image tokens
    const float *cm = colmax + (size_t)b * HT;
    float *eb = e + ((size_t)b * N + n0) * HT;
    for (int i = threadIdx.x; i < rows * HT; i += kThreads) {
        const int r = i / HT, j = i - r * HT;
        const float cm1 = clampf(cm[j] - g, clamp_lo, clamp_hi);  // max_n x1 (clamp is monotone)
        float v = expf(clampf(tile[i] - cm1, clamp_lo, clamp_hi));
        if (mask_v && mask_v[(size_t)b * N + n0 + r]) v = 0.f;
        tile[i] = v;
        eb[i] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += tile[r * HT + j];
        colacc[j] += s;
    }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) part_sum[((size_t)b * chunks + chunk) * HT + j] = colacc[j];
}

__global__ __launch_bounds__(kThreads) void bis_rows_bwd(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const uint8_t *__restrict__ mask_l, const float *__restrict__ pv,
    const float *__restrict__ e, const float *__restrict__ g_pv, const float *__restrict__ g_e,
    const float *__restrict__ g_colsum, int N, int H, int T, int rows_per_block, int stable,
    int clamp_lo, int clamp_hi, float *__restrict__ g_xm, float *__restrict__ part_gc)
{
    extern __shared__ float tile[];  // [rows][HT]: gradient w.r.t. x1 from the p_v branch, then g_x; colacc[HT]
    const int HT = H * T;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    float *colacc = tile + (size_t)rows_per_block * HT;
    for (int j = threadIdx.x; j < HT; j += kThreads) colacc[j] = 0.f;
    const float g = stable ? gmax[0] : 0.f;
    const int ntiles = (N + rows_per_block - 1) / rows_per_block;
    for (int tl = chunk; tl < ntiles; tl += chunks) {
    const int n0 = tl * rows_per_block, rows = min(rows_per_block, N - n0);
    const size_t base = ((size_t)b * N + n0) * HT;
    __syncthreads();
    // softmax backward per (row, head): p * (g - <g, p>)
    for (int rh = threadIdx.x; rh < rows * H; rh += kThreads) {
        const int r = rh / H, h = rh - r * H;
        const float *p = pv + base + (size_t)r * HT + h * T;
        const float *gp = g_pv + base + (size_t)r * HT + h * T;
        float dot = 0.f;
        for (int t = 0; t < T; ++t) dot = fmaf(gp[t], p[t], dot);
        float *o = tile + r * HT + h * T;
        for (int t = 0; t < T; ++t) o[t] = p[t] * (gp[t] - dot);  // 0 on masked text tokens (p = 0)
    }
    __syncthreads();
    const float *cb = c + (size_t)b * HT;
    const float *cm = colmax + (size_t)b * HT;
    const float *gcs = g_colsum + (size_t)b * HT;
    for (int i = threadIdx.x; i < rows * HT; i += kThreads) {
        const int j = i % HT;
        const float xs = xm[base + i] + cb[j] - g;                 // before clamp 1
        const float x1 = clampf(xs, clamp_lo, clamp_hi);
        const float cm1 = clampf(cm[j] - g, clamp_lo, clamp_hi);
        const float d2 = x1 - cm1;                                 // before clamp 2
        const bool pass2 = !((clamp_lo && d2 < -kClamp) || (clamp_hi && d2 > kClamp));
        const bool pass1 = !((clamp_lo && xs < -kClamp) || (clamp_hi && xs > kClamp));
        float gl = (g_e[base + i] + gcs[j]) * e[base + i];         // through exp (e = 0 on padded image tokens)
        if (!pass2) gl = 0.f;
        float gx = tile[i] + gl;
        if (!pass1) gx = 0.f;
        tile[i] = gx;
        g_xm[base + i] = gx;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += tile[r * HT + j];
        colacc[j] += s;
    }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) part_gc[((size_t)b * chunks + chunk) * HT + j] = colacc[j];
}

inline int rows_per_block(int HT)
{
    int r = kTileFloats / HT;
    return r < 1 ? 1 : (r > 64 ? 64 : r);
}
inline int colmax_chunks(int N) { int c = (N + 255) / 256; return c > 128 ? 128 : (c < 1 ? 1 : c); }
// row kernels: at most kMaxRowBlocks blocks per batch element, each striding over the row tiles
constexpr int kMaxRowBlocks = 512;
inline int row_blocks(int N, int HT)
{
    const int tiles = (N + rows_per_block(HT) - 1) / rows_per_block(HT);
    return tiles < kMaxRowBlocks ? tiles : kMaxRowBlocks;
}

}  // namespace

extern "C" {

size_t zira_bisoftmax_workspace_floats(int B, int N, int H, int T)
{
    if (B <= 0 || N <= 0 || H <= 0 || T <= 0) return 0;
    const int HT = H * T;
    const size_t row_chunks = (size_t)row_blocks(N, HT);
    const size_t chunks = row_chunks > (size_t)colmax_chunks(N) ? row_chunks : (size_t)colmax_chunks(N);
    return (size_t)B * chunks * HT + (size_t)B * HT + 8;  // partials, column maxima, global maximum
}

int zira_bisoftmax_fwd_f32(const float *xm, const float *c, const uint8_t *mask_l, const uint8_t *mask_v,
                           int B, int N, int H, int T, int stable, int clamp_lo, int clamp_hi, float *pv,
                           float *e, float *colsum, float *colmax, float *gmax, float *workspace,
                           void *stream)
{
    if (!xm || !c || !pv || !e || !colsum || !colmax || !gmax || !workspace || B <= 0 || N <= 0 || H <= 0 ||
        T <= 0 || (size_t)H * T > kTileFloats)
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int HT = H * T, total = B * HT;
    (void)total;
    const int cchunks = colmax_chunks(N), crow = (N + cchunks - 1) / cchunks;
    hipLaunchKernelGGL(bis_colmax_partial, dim3(cchunks, B), dim3(kThreads), 0, st, xm, N, HT, crow, workspace);
    hipLaunchKernelGGL(bis_fold, dim3(B), dim3(kThreads), 0, st, workspace, cchunks, HT, 0, c, colmax);
    hipLaunchKernelGGL(bis_global_max, dim3(1), dim3(kThreads), 0, st, colmax, total, gmax);
    const int R = rows_per_block(HT), rchunks = row_blocks(N, HT);
    hipLaunchKernelGGL(bis_rows_fwd, dim3(rchunks, B), dim3(kThreads), ((size_t)R + 1) * HT * sizeof(float), st,
                       xm, c, colmax, gmax, mask_l, mask_v, N, H, T, R, stable, clamp_lo, clamp_hi, pv, e, workspace);
    hipLaunchKernelGGL(bis_fold, dim3(B), dim3(kThreads), 0, st, workspace, rchunks, HT, 1,
                       (const float *)nullptr, colsum);
    return (int)hipGetLastError();
}

int zira_bisoftmax_bwd_f32(const float *xm, const float *c, const uint8_t *mask_l, int B, int N, int H, int T,
                           int stable, int clamp_lo, int clamp_hi, const float *pv, const float *e,
                           const float *colmax, const float *gmax, const float *g_pv, const float *g_e,
                           const float *g_colsum, float *g_xm, float *g_c, float *workspace, void *stream)
{
    if (!xm || !c || !pv || !e || !colmax || !gmax || !g_pv || !g_e || !g_colsum || !g_xm || !g_c ||
        !workspace || B <= 0 || N <= 0 || H <= 0 || T <= 0 || (size_t)H * T > kTileFloats)
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int HT = H * T;
    const int R = rows_per_block(HT), rchunks = row_blocks(N, HT);
    hipLaunchKernelGGL(bis_rows_bwd, dim3(rchunks, B), dim3(kThreads), ((size_t)R + 1) * HT * sizeof(float), st,
                       xm, c, colmax, gmax, mask_l, pv, e, g_pv, g_e, g_colsum, N, H, T, R, stable, clamp_lo,
                       clamp_hi, g_xm, workspace);
    hipLaunchKernelGGL(bis_fold, dim3(B), dim3(kThreads), 0, st, workspace, rchunks, HT, 1,
                       (const float *)nullptr, g_c);
    return (int)hipGetLastError();
}

}  // extern "C"
