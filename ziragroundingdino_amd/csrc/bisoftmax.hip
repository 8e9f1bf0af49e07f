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
constexpr int kTileFloats = 2048;  // LDS tile: rows * HT <= 2048 floats (two tiles per block)
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
// op 0 = max, 1 = sum.  One block per batch element and slab of kFoldCols columns: thread (column, phase) takes every
// kFoldPhases-th chunk and LDS joins the phases in a fixed order.  (One block per batch element with a thread per column
// read 512 partials in sequence per thread at H * T >= 128: 47 us, 16 calls per step; 32 columns per block: 14 us.)
constexpr int kFoldCols = 8, kFoldPhases = kThreads / kFoldCols;
__global__ __launch_bounds__(kThreads) void bis_fold(const float *__restrict__ part, int chunks, int HT,
                                                     int op, const float *__restrict__ addend,
                                                     float *__restrict__ out)
{
    __shared__ float red[kThreads];
    const int b = blockIdx.x, j = blockIdx.y * kFoldCols + threadIdx.x % kFoldCols, ph = threadIdx.x / kFoldCols;
    float acc = op ? 0.f : -INFINITY;
    if (j < HT) {
        const float *p = part + (size_t)b * chunks * HT + j;
        for (int c = ph; c < chunks; c += kFoldPhases) {
            const float v = p[(size_t)c * HT];
            acc = op ? acc + v : fmaxf(acc, v);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ph == 0 && j < HT) {
        for (int q = 1; q < kFoldPhases; ++q) {
            const float v = red[q * kFoldCols + threadIdx.x];
            acc = op ? acc + v : fmaxf(acc, v);
        }
        const int i = b * HT + j;
        if (addend) acc += addend[i];
        out[i] = acc;
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

// rows: p_v, e and the partial column sums of e.  All global traffic is whole-tile and coalesced; the softmax over the T
// text tokens of a (row, head) is one thread's walk through LDS, each thread starting at another token (rh % T) so that
// the lanes of a wave hit different banks whatever T is (with T = 32 and every thread starting at token 0, the 64 lanes
// of a wave shared two banks: 85 us for a 136 MB pass).
__global__ __launch_bounds__(kThreads) void bis_rows_fwd(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const uint8_t *__restrict__ mask_l /* [B,T] or null */,
    const uint8_t *__restrict__ mask_v /* [B,N] or null */, int N, int H, int T, int rows_per_block,
    int stable, int clamp_lo, int clamp_hi, int vec, float *__restrict__ pv, float *__restrict__ e,
    float *__restrict__ part_sum)
{
    extern __shared__ __align__(16) float tile[];  // [rows][HT]: x1;  [rows][HT]: p_v, then e;  colacc[HT];  text mask [T] as floats
    const int HT = H * T;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    float *tile2 = tile + (size_t)rows_per_block * HT;
    float *colacc = tile2 + (size_t)rows_per_block * HT;
    float *live = colacc + HT;   // 1 = text token takes part, 0 = padded
    for (int j = threadIdx.x; j < HT; j += kThreads) colacc[j] = 0.f;
    for (int t = threadIdx.x; t < T; t += kThreads) live[t] = (mask_l && mask_l[b * T + t]) ? 0.f : 1.f;
    const float g = stable ? gmax[0] : 0.f;
    const float *cb = c + (size_t)b * HT;
    const float *cm = colmax + (size_t)b * HT;
    const int j_first = threadIdx.x % HT, r_first = threadIdx.x / HT, j_step = kThreads % HT, r_step = kThreads / HT;
    const int j4_first = (4 * threadIdx.x) % HT, r4_first = (4 * threadIdx.x) / HT, j4_step = (4 * kThreads) % HT, r4_step = (4 * kThreads) / HT;
    int G = 0;   // lanes per (row, head) group: the power of two >= T, or 0 when T > 64
    if (T <= 64) for (G = 1; G < T; G <<= 1) {}
    const int ntiles = (N + rows_per_block - 1) / rows_per_block;
    for (int tl = chunk; tl < ntiles; tl += chunks) {
        const int n0 = tl * rows_per_block, rows = min(rows_per_block, N - n0);
        const size_t base = ((size_t)b * N + n0) * HT;
        __syncthreads();
        // (column / row of element i without a division per element: i advances by kThreads; 16-byte accesses when H * T is
        // a multiple of 4 and the tensors are 16-byte aligned)
        if (vec) {
            for (int i = 4 * threadIdx.x, j = j4_first; i < rows * HT; i += 4 * kThreads, j = j + j4_step >= HT ? j + j4_step - HT : j + j4_step) {
                const float4 x = *reinterpret_cast<const float4 *>(xm + base + i), cc = *reinterpret_cast<const float4 *>(cb + j);
                *reinterpret_cast<float4 *>(tile + i) =
                    make_float4(clampf(x.x + cc.x - g, clamp_lo, clamp_hi), clampf(x.y + cc.y - g, clamp_lo, clamp_hi),
                                clampf(x.z + cc.z - g, clamp_lo, clamp_hi), clampf(x.w + cc.w - g, clamp_lo, clamp_hi));
            }
        } else {
            for (int i = threadIdx.x, j = j_first; i < rows * HT; i += kThreads, j = j + j_step >= HT ? j + j_step - HT : j + j_step)
                tile[i] = clampf(xm[base + i] + cb[j] - g, clamp_lo, clamp_hi);  // x1
        }
        __syncthreads();
        // softmax over the T text tokens of a (row, head): a group of G >= T lanes with wave reductions for T <= 64 ...
        if (G) {
            const int lane_t = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = kThreads / G;
            const bool on = lane_t < T && live[lane_t < T ? lane_t : 0] != 0.f;
            for (int rh0 = 0; rh0 < rows * H; rh0 += ngrp) {   // (whole waves stay in the loop: the reductions are wave-wide)
                const int rh = rh0 + grp;
                const bool in = rh < rows * H && lane_t < T;
                const int idx = in ? (rh / H) * HT + (rh % H) * T + lane_t : 0;
                const float x = (in && on) ? tile[idx] : -INFINITY;
                float m = x;
                for (int d = 1; d < G; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
                const float ex = (in && on) ? expf(x - m) : 0.f;
                float sum = ex;
                for (int d = 1; d < G; d <<= 1) sum += __shfl_xor(sum, d);
                if (in) tile2[idx] = on ? ex * (1.f / sum) : 0.f;   // all text tokens masked: 0 * inf = nan, as torch.softmax of all -inf
            }
        } else
        // ... or one thread's walk for longer texts
        for (int rh = threadIdx.x; rh < rows * H; rh += kThreads) {
            const int r = rh / H, h = rh - r * H, t0 = rh % T;
            const float *x1 = tile + r * HT + h * T;
            float m = -INFINITY;
            for (int k = 0, t = t0; k < T; ++k, t = t + 1 == T ? 0 : t + 1)
                if (live[t] != 0.f) m = fmaxf(m, x1[t]);
            float s = 0.f;
            for (int k = 0, t = t0; k < T; ++k, t = t + 1 == T ? 0 : t + 1)
                if (live[t] != 0.f) s += expf(x1[t] - m);
            const float inv = 1.f / s;  // all text tokens masked: 0 * inf = nan, as torch.softmax of all -inf
            float *o = tile2 + r * HT + h * T;
            for (int k = 0, t = t0; k < T; ++k, t = t + 1 == T ? 0 : t + 1)
                o[t] = live[t] != 0.f ? expf(x1[t] - m) * inv : 0.f;
        }
        __syncthreads();
        // p_v out; e = exp(clamp(x1 - colmax1)), 0 on padded image tokens
        if (vec) {
            for (int i = 4 * threadIdx.x, j = j4_first, r = r4_first; i < rows * HT; i += 4 * kThreads) {
                *reinterpret_cast<float4 *>(pv + base + i) = *reinterpret_cast<const float4 *>(tile2 + i);
                const float4 x = *reinterpret_cast<const float4 *>(tile + i), cc = *reinterpret_cast<const float4 *>(cm + j);
                const bool dead = mask_v && mask_v[(size_t)b * N + n0 + r];
                float4 v;
                v.x = dead ? 0.f : expf(clampf(x.x - clampf(cc.x - g, clamp_lo, clamp_hi), clamp_lo, clamp_hi));
                v.y = dead ? 0.f : expf(clampf(x.y - clampf(cc.y - g, clamp_lo, clamp_hi), clamp_lo, clamp_hi));
                v.z = dead ? 0.f : expf(clampf(x.z - clampf(cc.z - g, clamp_lo, clamp_hi), clamp_lo, clamp_hi));
                v.w = dead ? 0.f : expf(clampf(x.w - clampf(cc.w - g, clamp_lo, clamp_hi), clamp_lo, clamp_hi));
                *reinterpret_cast<float4 *>(tile2 + i) = v;
                *reinterpret_cast<float4 *>(e + base + i) = v;
                r += r4_step + (j + j4_step >= HT ? 1 : 0);
                j = j + j4_step >= HT ? j + j4_step - HT : j + j4_step;
            }
        } else
        for (int i = threadIdx.x, j = j_first, r = r_first; i < rows * HT; i += kThreads) {
            pv[base + i] = tile2[i];
            const float cm1 = clampf(cm[j] - g, clamp_lo, clamp_hi);  // max_n x1 (clamp is monotone)
            float v = expf(clampf(tile[i] - cm1, clamp_lo, clamp_hi));
            if (mask_v && mask_v[(size_t)b * N + n0 + r]) v = 0.f;
            tile2[i] = v;
            e[base + i] = v;
            r += r_step + (j + j_step >= HT ? 1 : 0);
            j = j + j_step >= HT ? j + j_step - HT : j + j_step;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < HT; j += kThreads) {
            float s = 0.f;
            for (int r = 0; r < rows; ++r) s += tile2[r * HT + j];
            colacc[j] += s;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) part_sum[((size_t)b * chunks + chunk) * HT + j] = colacc[j];
}

// The same for T = 32 text tokens (the ODinW-like captions of the bench; H * T a multiple of 64): a (row, head) is half a
// wave, so everything stays in registers -- a lane owns column 64 hf + lane of its wave's rows, the softmax reductions are
// five xor-shuffles inside 32 lanes (in the order of the G-lane groups above: the same p_v bit for bit), the column sums of
// e are per-lane registers joined through LDS once at the end.  No tile, no barrier inside the loop, four rows of loads
// in flight per wave: 42 us -> HBM time for the 68 MB.
template <int HALVES>
__global__ __launch_bounds__(kThreads) void bis_rows_fwd_t32(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const uint8_t *__restrict__ mask_l, const uint8_t *__restrict__ mask_v, int N,
    int stable, int clamp_lo, int clamp_hi, float *__restrict__ pv, float *__restrict__ e, float *__restrict__ part_sum)
{
    constexpr int HT = 64 * HALVES, T = 32, U = 4;
    __shared__ float red[kThreads / 64][HT];
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = stable ? gmax[0] : 0.f;
    float cj[HALVES], cm1[HALVES], acc[HALVES];
    bool live[HALVES];
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf) {
        const int j = 64 * hf + lane;
        cj[hf] = c[(size_t)b * HT + j];
        cm1[hf] = clampf(colmax[(size_t)b * HT + j] - g, clamp_lo, clamp_hi);   // max_n x1 (clamp is monotone)
        live[hf] = !(mask_l && mask_l[b * T + (j & (T - 1))]);
        acc[hf] = 0.f;
    }
    const int nw = (kThreads / 64) * chunks;   // waves per batch element: wave w takes rows w, w + nw, ...
    const float *xb = xm + (size_t)b * N * HT + lane;
    for (int r0 = chunk * (kThreads / 64) + wave; r0 < N; r0 += nw * U) {
        float x[U][HALVES];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = r0 + u * nw < N ? r0 + u * nw : N - 1;   // (clamped: no branch round the loads)
#pragma unroll
            for (int hf = 0; hf < HALVES; ++hf) x[u][hf] = xb[(size_t)row * HT + 64 * hf];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = r0 + u * nw;
            if (row >= N) break;   // wave-uniform
            const bool dead = mask_v && mask_v[(size_t)b * N + row];
            const size_t o = ((size_t)b * N + row) * HT + lane;
#pragma unroll
            for (int hf = 0; hf < HALVES; ++hf) {
                const float x1 = clampf(x[u][hf] + cj[hf] - g, clamp_lo, clamp_hi);
                float m = live[hf] ? x1 : -INFINITY;
#pragma unroll
                for (int d = 1; d < T; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
                const float ex = live[hf] ? expf(x1 - m) : 0.f;
                float sum = ex;
#pragma unroll
                for (int d = 1; d < T; d <<= 1) sum += __shfl_xor(sum, d);
                pv[o + 64 * hf] = live[hf] ? ex * (1.f / sum) : 0.f;
                const float ev = dead ? 0.f : expf(clampf(x1 - cm1[hf], clamp_lo, clamp_hi));
                e[o + 64 * hf] = ev;
                acc[hf] += ev;
            }
        }
    }
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf) red[wave][64 * hf + lane] = acc[hf];
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) {
        float t = red[0][j];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t += red[w][j];
        part_sum[((size_t)b * chunks + chunk) * HT + j] = t;
    }
}

// Longer texts (T > 64: COCO-like captions pad to ~195 tokens, reference utils.py:234-269): a WAVE per image token.  Lane l
// holds columns l, l + 64, ... (KPL of them) of its row in registers; a head's T columns are spread over the lanes, so its
// softmax maximum and sum are two wave reductions per head.  No LDS tile, no barrier in the loop, whole rows of coalesced
// loads in flight.  (The tile kernel above walks a (row, head) with ONE thread for T > 64: 2.3 ms per call at T = 194 against
// the 0.1 ms its 414 MB take.)
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

template <int KPL>
__global__ __launch_bounds__(kThreads) void bis_rows_fwd_wave(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const uint8_t *__restrict__ mask_l, const uint8_t *__restrict__ mask_v, int N, int H, int T,
    int stable, int clamp_lo, int clamp_hi, float *__restrict__ pv, float *__restrict__ e, float *__restrict__ part_sum)
{
    __shared__ float red[kThreads / 64][64 * KPL];
    const int HT = H * T;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = stable ? gmax[0] : 0.f;
    float cj[KPL], cm1[KPL], acc[KPL];
    int hk[KPL];          // the head of column lane + 64 k (H: no such column)
    bool live[KPL];       // ... and whether its text token takes part
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int j = lane + 64 * k;
        const bool valid = j < HT;
        hk[k] = valid ? j / T : H;
        live[k] = valid && !(mask_l && mask_l[b * T + (j - hk[k] * T)]);
        cj[k] = valid ? c[(size_t)b * HT + j] : 0.f;
        cm1[k] = valid ? clampf(colmax[(size_t)b * HT + j] - g, clamp_lo, clamp_hi) : 0.f;   // max_n x1 (clamp is monotone)
        acc[k] = 0.f;
    }
    const int nw = (kThreads / 64) * chunks;   // waves per batch element: wave w takes rows w, w + nw, ...
    for (int row = chunk * (kThreads / 64) + wave; row < N; row += nw) {
        const size_t o = ((size_t)b * N + row) * HT + lane;
        float x1[KPL], mk[KPL], ex[KPL], sk[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) x1[k] = hk[k] < H ? xm[o + 64 * k] : 0.f;
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            x1[k] = clampf(x1[k] + cj[k] - g, clamp_lo, clamp_hi);
            mk[k] = -INFINITY;
            sk[k] = 1.f;
        }
        for (int h = 0; h < H; ++h) {
            float loc = -INFINITY;
#pragma unroll
            for (int k = 0; k < KPL; ++k) loc = (hk[k] == h && live[k]) ? fmaxf(loc, x1[k]) : loc;
            loc = wave_max(loc);
#pragma unroll
            for (int k = 0; k < KPL; ++k) mk[k] = hk[k] == h ? loc : mk[k];
        }
#pragma unroll
        for (int k = 0; k < KPL; ++k) ex[k] = live[k] ? expf(x1[k] - mk[k]) : 0.f;
        for (int h = 0; h < H; ++h) {
            float loc = 0.f;
#pragma unroll
            for (int k = 0; k < KPL; ++k) loc += hk[k] == h ? ex[k] : 0.f;
            loc = wave_sum(loc);
#pragma unroll
            for (int k = 0; k < KPL; ++k) sk[k] = hk[k] == h ? loc : sk[k];
        }
        const bool dead = mask_v && mask_v[(size_t)b * N + row];
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            if (hk[k] < H) {
                pv[o + 64 * k] = live[k] ? ex[k] * (1.f / sk[k]) : 0.f;
                const float ev = dead ? 0.f : expf(clampf(x1[k] - cm1[k], clamp_lo, clamp_hi));
                e[o + 64 * k] = ev;
                acc[k] += ev;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) red[wave][64 * k + lane] = acc[k];
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) {
        float t = red[0][j];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t += red[w][j];
        part_sum[((size_t)b * chunks + chunk) * HT + j] = t;
    }
}

template <int KPL>
__global__ __launch_bounds__(kThreads) void bis_rows_bwd_wave(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const float *__restrict__ pv, const float *__restrict__ e, const float *__restrict__ g_pv,
    const float *__restrict__ g_e, const float *__restrict__ g_colsum, int N, int H, int T, int stable, int clamp_lo, int clamp_hi,
    float *__restrict__ g_xm, float *__restrict__ part_gc)
{
    __shared__ float red[kThreads / 64][64 * KPL];
    const int HT = H * T;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = stable ? gmax[0] : 0.f;
    float cj[KPL], cm1[KPL], gcs[KPL], acc[KPL];
    int hk[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int j = lane + 64 * k;
        const bool valid = j < HT;
        hk[k] = valid ? j / T : H;
        cj[k] = valid ? c[(size_t)b * HT + j] : 0.f;
        cm1[k] = valid ? clampf(colmax[(size_t)b * HT + j] - g, clamp_lo, clamp_hi) : 0.f;
        gcs[k] = valid ? g_colsum[(size_t)b * HT + j] : 0.f;
        acc[k] = 0.f;
    }
    const int nw = (kThreads / 64) * chunks;
    for (int row = chunk * (kThreads / 64) + wave; row < N; row += nw) {
        const size_t o = ((size_t)b * N + row) * HT + lane;
        float pp[KPL], gp[KPL], xv[KPL], ge[KPL], ev[KPL], dk[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            const bool valid = hk[k] < H;
            pp[k] = valid ? pv[o + 64 * k] : 0.f;
            gp[k] = valid ? g_pv[o + 64 * k] : 0.f;
            xv[k] = valid ? xm[o + 64 * k] : 0.f;
            ge[k] = valid ? g_e[o + 64 * k] : 0.f;
            ev[k] = valid ? e[o + 64 * k] : 0.f;
            dk[k] = 0.f;
        }
        // softmax backward per (row, head): p * (g - <g, p>)   (0 on masked text tokens: p = 0)
        for (int h = 0; h < H; ++h) {
            float loc = 0.f;
#pragma unroll
            for (int k = 0; k < KPL; ++k) loc = hk[k] == h ? fmaf(gp[k], pp[k], loc) : loc;
            loc = wave_sum(loc);
#pragma unroll
            for (int k = 0; k < KPL; ++k) dk[k] = hk[k] == h ? loc : dk[k];
        }
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            if (hk[k] < H) {
                const float gpv = pp[k] * (gp[k] - dk[k]);
                const float xs = xv[k] + cj[k] - g;                        // before clamp 1
                const float x1 = clampf(xs, clamp_lo, clamp_hi);
                const float d2 = x1 - cm1[k];                              // before clamp 2
                const bool pass2 = !((clamp_lo && d2 < -kClamp) || (clamp_hi && d2 > kClamp));
                const bool pass1 = !((clamp_lo && xs < -kClamp) || (clamp_hi && xs > kClamp));
                float gl = (ge[k] + gcs[k]) * ev[k];                       // through exp (e = 0 on padded image tokens)
                if (!pass2) gl = 0.f;
                float gx = gpv + gl;
                if (!pass1) gx = 0.f;
                g_xm[o + 64 * k] = gx;
                acc[k] += gx;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) red[wave][64 * k + lane] = acc[k];
    __syncthreads();
    for (int j = threadIdx.x; j < HT; j += kThreads) {
        float t = red[0][j];
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) t += red[w][j];
        part_gc[((size_t)b * chunks + chunk) * HT + j] = t;
    }
}

__global__ __launch_bounds__(kThreads) void bis_rows_bwd(
    const float *__restrict__ xm, const float *__restrict__ c, const float *__restrict__ colmax,
    const float *__restrict__ gmax, const uint8_t *__restrict__ mask_l, const float *__restrict__ pv,
    const float *__restrict__ e, const float *__restrict__ g_pv, const float *__restrict__ g_e,
    const float *__restrict__ g_colsum, int N, int H, int T, int rows_per_block, int stable,
    int clamp_lo, int clamp_hi, int vec, float *__restrict__ g_xm, float *__restrict__ part_gc)
{
    extern __shared__ __align__(16) float tile[];  // [rows][HT]: p_v, then the gradient w.r.t. x1 from the p_v branch, then g_x;  [rows][HT]: g_pv;  colacc[HT]
    const int HT = H * T;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    float *tile2 = tile + (size_t)rows_per_block * HT;
    float *colacc = tile2 + (size_t)rows_per_block * HT;
    for (int j = threadIdx.x; j < HT; j += kThreads) colacc[j] = 0.f;
    const float g = stable ? gmax[0] : 0.f;
    const float *cb = c + (size_t)b * HT;
    const float *cm = colmax + (size_t)b * HT;
    const float *gcs = g_colsum + (size_t)b * HT;
    const int j_first = threadIdx.x % HT, j_step = kThreads % HT;
    const int j4_first = (4 * threadIdx.x) % HT, j4_step = (4 * kThreads) % HT;
    int G = 0;   // lanes per (row, head) group: the power of two >= T, or 0 when T > 64
    if (T <= 64) for (G = 1; G < T; G <<= 1) {}
    const int ntiles = (N + rows_per_block - 1) / rows_per_block;
    for (int tl = chunk; tl < ntiles; tl += chunks) {
        const int n0 = tl * rows_per_block, rows = min(rows_per_block, N - n0);
        const size_t base = ((size_t)b * N + n0) * HT;
        __syncthreads();
        if (vec) {
            for (int i = 4 * threadIdx.x; i < rows * HT; i += 4 * kThreads) {
                *reinterpret_cast<float4 *>(tile + i) = *reinterpret_cast<const float4 *>(pv + base + i);
                *reinterpret_cast<float4 *>(tile2 + i) = *reinterpret_cast<const float4 *>(g_pv + base + i);
            }
        } else {
            for (int i = threadIdx.x; i < rows * HT; i += kThreads) {
                tile[i] = pv[base + i];
                tile2[i] = g_pv[base + i];
            }
        }
        __syncthreads();
        // softmax backward per (row, head): p * (g - <g, p>): a lane group per (row, head) for T <= 64 ...
        if (G) {
            const int lane_t = threadIdx.x & (G - 1), grp = threadIdx.x / G, ngrp = kThreads / G;
            for (int rh0 = 0; rh0 < rows * H; rh0 += ngrp) {
                const int rh = rh0 + grp;
                const bool in = rh < rows * H && lane_t < T;
                const int idx = in ? (rh / H) * HT + (rh % H) * T + lane_t : 0;
                const float pp = in ? tile[idx] : 0.f, gg = in ? tile2[idx] : 0.f;
                float dot = pp * gg;
                for (int d = 1; d < G; d <<= 1) dot += __shfl_xor(dot, d);
                if (in) tile[idx] = pp * (gg - dot);   // 0 on masked text tokens (p = 0)
            }
        } else
        // ... or one thread's walk, the threads starting at different tokens (see bis_rows_fwd)
        for (int rh = threadIdx.x; rh < rows * H; rh += kThreads) {
            const int r = rh / H, h = rh - r * H, t0 = rh % T;
            float *p = tile + r * HT + h * T;
            const float *gp = tile2 + r * HT + h * T;
            float dot = 0.f;
            for (int k = 0, t = t0; k < T; ++k, t = t + 1 == T ? 0 : t + 1) dot = fmaf(gp[t], p[t], dot);
            for (int k = 0, t = t0; k < T; ++k, t = t + 1 == T ? 0 : t + 1) p[t] = p[t] * (gp[t] - dot);  // 0 on masked text tokens (p = 0)
        }
        __syncthreads();
        auto grad_x = [&](float xmv, float cbv, float cmv, float gev, float gcv, float ev, float gpv) {
            const float xs = xmv + cbv - g;                            // before clamp 1
            const float x1 = clampf(xs, clamp_lo, clamp_hi);
            const float cm1 = clampf(cmv - g, clamp_lo, clamp_hi);
            const float d2 = x1 - cm1;                                 // before clamp 2
            const bool pass2 = !((clamp_lo && d2 < -kClamp) || (clamp_hi && d2 > kClamp));
            const bool pass1 = !((clamp_lo && xs < -kClamp) || (clamp_hi && xs > kClamp));
            float gl = (gev + gcv) * ev;                               // through exp (e = 0 on padded image tokens)
            if (!pass2) gl = 0.f;
            float gx = gpv + gl;
            if (!pass1) gx = 0.f;
            return gx;
        };
        if (vec) {
            for (int i = 4 * threadIdx.x, j = j4_first; i < rows * HT; i += 4 * kThreads, j = j + j4_step >= HT ? j + j4_step - HT : j + j4_step) {
                const float4 x = *reinterpret_cast<const float4 *>(xm + base + i), ge = *reinterpret_cast<const float4 *>(g_e + base + i);
                const float4 ev = *reinterpret_cast<const float4 *>(e + base + i), gp = *reinterpret_cast<const float4 *>(tile + i);
                const float4 c4 = *reinterpret_cast<const float4 *>(cb + j), m4 = *reinterpret_cast<const float4 *>(cm + j);
                const float4 s4 = *reinterpret_cast<const float4 *>(gcs + j);
                const float4 o = make_float4(grad_x(x.x, c4.x, m4.x, ge.x, s4.x, ev.x, gp.x), grad_x(x.y, c4.y, m4.y, ge.y, s4.y, ev.y, gp.y),
                                             grad_x(x.z, c4.z, m4.z, ge.z, s4.z, ev.z, gp.z), grad_x(x.w, c4.w, m4.w, ge.w, s4.w, ev.w, gp.w));
                *reinterpret_cast<float4 *>(tile + i) = o;
                *reinterpret_cast<float4 *>(g_xm + base + i) = o;
            }
        } else
        for (int i = threadIdx.x, j = j_first; i < rows * HT; i += kThreads, j = j + j_step >= HT ? j + j_step - HT : j + j_step) {
            const float gx = grad_x(xm[base + i], cb[j], cm[j], g_e[base + i], gcs[j], e[base + i], tile[i]);
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
    (void)mask_l;
}

inline int rows_per_block(int HT)
{
    int r = kTileFloats / HT;
    return r < 1 ? 1 : (r > 64 ? 64 : r);
}
inline int colmax_chunks(int N) { int c = (N + 63) / 64; return c > 512 ? 512 : (c < 1 ? 1 : c); }   // (87 chunks of 256 rows: 174 blocks, 24 us for 23 MB)
// row kernels: at most kMaxRowBlocks blocks per batch element, each striding over the row tiles
constexpr int kMaxRowBlocks = 512;
inline int row_blocks(int N, int HT)
{
    const int tiles = (N + rows_per_block(HT) - 1) / rows_per_block(HT);
    return tiles < kMaxRowBlocks ? tiles : kMaxRowBlocks;
}

// the wave-per-row kernels: T > 64 (the tile kernels keep the lane-group forms for short texts), H * T <= 1024
inline bool use_wave_rows(int H, int T) { return T > 64 && H * T <= 1024; }
inline int wave_row_blocks(int N)
{
    const int blocks = (N + kThreads / 64 - 1) / (kThreads / 64);
    return blocks < kMaxRowBlocks ? blocks : kMaxRowBlocks;
}
#define ZIRA_BIS_KPL(HT_, CALL_)                                   \
    do {                                                           \
        if ((HT_) <= 256) { constexpr int KPL = 4; CALL_; }        \
        else if ((HT_) <= 512) { constexpr int KPL = 8; CALL_; }   \
        else if ((HT_) <= 768) { constexpr int KPL = 12; CALL_; }  \
        else if ((HT_) <= 832) { constexpr int KPL = 13; CALL_; }  \
        else { constexpr int KPL = 16; CALL_; }                    \
    } while (0)

}  // namespace

extern "C" {

size_t zira_bisoftmax_workspace_floats(int B, int N, int H, int T)
{
    if (B <= 0 || N <= 0 || H <= 0 || T <= 0) return 0;
    const int HT = H * T;
    const size_t row_chunks = use_wave_rows(H, T) ? (size_t)wave_row_blocks(N) : (size_t)row_blocks(N, HT);
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
    hipLaunchKernelGGL(bis_fold, dim3(B, (HT + kFoldCols - 1) / kFoldCols), dim3(kThreads), 0, st, workspace, cchunks, HT, 0, c, colmax);
    hipLaunchKernelGGL(bis_global_max, dim3(1), dim3(kThreads), 0, st, colmax, total, gmax);
    const int R = rows_per_block(HT);
    int rchunks = row_blocks(N, HT);
    const int vec = (HT % 4 == 0) && !(((uintptr_t)xm | (uintptr_t)c | (uintptr_t)colmax | (uintptr_t)pv | (uintptr_t)e) & 15);
    if (use_wave_rows(H, T)) {   // a wave per image token
        rchunks = wave_row_blocks(N);
        ZIRA_BIS_KPL(HT, hipLaunchKernelGGL(bis_rows_fwd_wave<KPL>, dim3(rchunks, B), dim3(kThreads), 0, st, xm, c, colmax, gmax, mask_l, mask_v, N,
                                            H, T, stable, clamp_lo, clamp_hi, pv, e, workspace));
    } else if (T == 32 && (HT == 64 || HT == 128 || HT == 256)) {   // a (row, head) = half a wave: the register form
        const dim3 grid(rchunks, B);
        if (HT == 64) hipLaunchKernelGGL(bis_rows_fwd_t32<1>, grid, dim3(kThreads), 0, st, xm, c, colmax, gmax, mask_l, mask_v, N, stable, clamp_lo, clamp_hi, pv, e, workspace);
        else if (HT == 128) hipLaunchKernelGGL(bis_rows_fwd_t32<2>, grid, dim3(kThreads), 0, st, xm, c, colmax, gmax, mask_l, mask_v, N, stable, clamp_lo, clamp_hi, pv, e, workspace);
        else hipLaunchKernelGGL(bis_rows_fwd_t32<4>, grid, dim3(kThreads), 0, st, xm, c, colmax, gmax, mask_l, mask_v, N, stable, clamp_lo, clamp_hi, pv, e, workspace);
    } else
    hipLaunchKernelGGL(bis_rows_fwd, dim3(rchunks, B), dim3(kThreads), (((size_t)2 * R + 1) * HT + T) * sizeof(float), st,
                       xm, c, colmax, gmax, mask_l, mask_v, N, H, T, R, stable, clamp_lo, clamp_hi, vec, pv, e, workspace);
    hipLaunchKernelGGL(bis_fold, dim3(B, (HT + kFoldCols - 1) / kFoldCols), dim3(kThreads), 0, st, workspace, rchunks, HT, 1,
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
    const int R = rows_per_block(HT);
    int rchunks = row_blocks(N, HT);
    const int vec = (HT % 4 == 0) && !(((uintptr_t)xm | (uintptr_t)c | (uintptr_t)colmax | (uintptr_t)pv | (uintptr_t)e | (uintptr_t)g_pv |
                                        (uintptr_t)g_e | (uintptr_t)g_colsum | (uintptr_t)g_xm) & 15);
    if (use_wave_rows(H, T)) {
        rchunks = wave_row_blocks(N);
        ZIRA_BIS_KPL(HT, hipLaunchKernelGGL(bis_rows_bwd_wave<KPL>, dim3(rchunks, B), dim3(kThreads), 0, st, xm, c, colmax, gmax, pv, e, g_pv, g_e,
                                            g_colsum, N, H, T, stable, clamp_lo, clamp_hi, g_xm, workspace));
    } else
    hipLaunchKernelGGL(bis_rows_bwd, dim3(rchunks, B), dim3(kThreads), ((size_t)2 * R + 1) * HT * sizeof(float), st,
                       xm, c, colmax, gmax, mask_l, pv, e, g_pv, g_e, g_colsum, N, H, T, R, stable, clamp_lo,
                       clamp_hi, vec, g_xm, workspace);
    hipLaunchKernelGGL(bis_fold, dim3(B, (HT + kFoldCols - 1) / kFoldCols), dim3(kThreads), 0, st, workspace, rchunks, HT, 1,
                       (const float *)nullptr, g_c);
    return (int)hipGetLastError();
}

}  // extern "C"
