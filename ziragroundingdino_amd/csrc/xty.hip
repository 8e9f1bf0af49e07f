// Tall-reduction product  out[z] = X[z]^T * Y[z]  for gfx950 (MI355X), float32.
//
//   X[z]: N x a (row-major) or, with x_transposed, a x N;   Y[z]: N x b;   out[z]: a x b
//   N is the number of image tokens (22 k), a and b are 64..256.
//
// This is the shape the re-bracketed BiAttention (transformer.py BiMultiHeadAttention, reference
// fuse_modules.py:99-248) leaves on the image side: (text probabilities)^T x tokens in the
// forward, tokens^T x (score gradients) and (probabilities)^T x (output gradients) in the
// backward.  A 64 x 256 output with a 22 k-long reduction is 0.7 GFLOP over 28 MB; a GEMM library
// launches one or two tiles for it (109 / 85 us measured with rocBLAS).  Here the reduction is
// split over `chunks` workgroups per output tile (a 64 x 256 or 256 x 64 tile, 8 x 8 outputs per
// thread, operands staged through LDS 32 rows at a time, the next rows in flight meanwhile), partial tiles go to a workspace and a
// second kernel folds them in a fixed order (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "zira_msda.h"

namespace {

constexpr int kThreads = 256;
constexpr int kKT = 32;        // reduction rows staged per step
constexpr int kMaxChunks = 128;

template <int TA, int TB>
__global__ __launch_bounds__(kThreads) void xty_partial(const float *__restrict__ X,
                                                        const float *__restrict__ Y, int N, int a,
                                                        int b, int x_transposed, int chunk_rows,
                                                        int b_tiles, float *__restrict__ part)
{
    static_assert(TA * TB == kThreads * 64, "8 x 8 outputs per thread");
    constexpr int XV = kKT * TA / 4 / kThreads;  // float4 (or 4 scalars) of X per thread and step
    constexpr int YV = kKT * TB / 4 / kThreads;
    __shared__ float xs[kKT][TA];
    __shared__ float ys[kKT][TB];
    const int chunk = blockIdx.x, chunks = gridDim.x, z = blockIdx.z;
    const int a0 = (blockIdx.y / b_tiles) * TA, b0 = (blockIdx.y % b_tiles) * TB;
    const int tid = threadIdx.x;
    // thread (ti, tj) owns rows {4ti..4ti+3, TA/2+4ti..} x columns {4tj..4tj+3, TB/2+4tj..}:
    // consecutive threads read consecutive float4 from LDS
    const int tj = tid % (TB / 8), ti = tid / (TB / 8);
    const float *Xz = X + (size_t)z * N * a;
    const float *Yz = Y + (size_t)z * N * b;
    const int n_begin = chunk * chunk_rows;
    const int n_end = (n_begin + chunk_rows < N) ? n_begin + chunk_rows : N;

    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;

    // global -> registers for the step starting at row n0 (zeros outside the chunk / the matrix)
    float4 xr[XV], yr[YV];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int e = tid + u * kThreads;
            xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x_transposed) {  // X is a x N: this thread takes 4 consecutive n of one column i
                const int i = e / (kKT / 4), k = (e % (kKT / 4)) * 4, n = n0 + k;
                if (a0 + i < a) {
                    const float *src = Xz + (size_t)(a0 + i) * N + n;
                    if (n < n_end) xr[u].x = src[0];
                    if (n + 1 < n_end) xr[u].y = src[1];
                    if (n + 2 < n_end) xr[u].z = src[2];
                    if (n + 3 < n_end) xr[u].w = src[3];
                }
            } else {
                const int k = e / (TA / 4), i = (e % (TA / 4)) * 4, n = n0 + k;
                if (n < n_end && a0 + i < a) xr[u] = *reinterpret_cast<const float4 *>(Xz + (size_t)n * a + a0 + i);
            }
        }
#pragma unroll
        for (int u = 0; u < YV; ++u) {
            const int e = tid + u * kThreads;
            const int k = e / (TB / 4), j = (e % (TB / 4)) * 4, n = n0 + k;
            yr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < n_end && b0 + j < b) yr[u] = *reinterpret_cast<const float4 *>(Yz + (size_t)n * b + b0 + j);
        }
    };
    auto stage = [&]() {  // registers -> LDS
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int e = tid + u * kThreads;
            if (x_transposed) {
                const int i = e / (kKT / 4), k = (e % (kKT / 4)) * 4;
                xs[k][i] = xr[u].x; xs[k + 1][i] = xr[u].y; xs[k + 2][i] = xr[u].z; xs[k + 3][i] = xr[u].w;
            } else {
                const int k = e / (TA / 4), i = (e % (TA / 4)) * 4;
                *reinterpret_cast<float4 *>(&xs[k][i]) = xr[u];
            }
        }
#pragma unroll
        for (int u = 0; u < YV; ++u) {
            const int e = tid + u * kThreads;
            const int k = e / (TB / 4), j = (e % (TB / 4)) * 4;
            *reinterpret_cast<float4 *>(&ys[k][j]) = yr[u];
        }
    };

    fetch(n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += kKT) {
        stage();
        __syncthreads();
        if (n0 + kKT < n_end) fetch(n0 + kKT);  // next step's rows travel while this one is multiplied
#pragma unroll 8
        for (int k = 0; k < kKT; ++k) {
            const float4 xa = *reinterpret_cast<const float4 *>(&xs[k][4 * ti]);
            const float4 xb = *reinterpret_cast<const float4 *>(&xs[k][TA / 2 + 4 * ti]);
            const float4 ya = *reinterpret_cast<const float4 *>(&ys[k][4 * tj]);
            const float4 yb = *reinterpret_cast<const float4 *>(&ys[k][TB / 2 + 4 * tj]);
            const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float yv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(xv[i], yv[j], acc[i][j]);
        }
        __syncthreads();
    }
    // partial tile -> workspace [z][chunk][a][b]
    float *pz = part + ((size_t)z * chunks + chunk) * a * b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = a0 + (i < 4 ? 4 * ti + i : TA / 2 + 4 * ti + i - 4);
        if (row >= a) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = b0 + (h ? TB / 2 : 0) + 4 * tj;
            if (col < b)
                *reinterpret_cast<float4 *>(pz + (size_t)row * b + col) =
                    make_float4(acc[i][4 * h], acc[i][4 * h + 1], acc[i][4 * h + 2], acc[i][4 * h + 3]);
        }
    }
}

// out = sum over chunks of the partial tiles.  A block folds 16 float4 of the output: 16 chunk
// groups (thread / 16) each add every 16th partial, LDS joins them in a fixed order.
__global__ __launch_bounds__(kThreads) void xty_fold(const float *__restrict__ part, int chunks,
                                                     size_t tile /* a*b */, size_t total4,
                                                     float *__restrict__ out)
{
    __shared__ float4 red[16][16];
    const int o = threadIdx.x % 16, cg = threadIdx.x / 16;
    const size_t t = (size_t)blockIdx.x * 16 + o;  // one float4 of out
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < total4) {
        const size_t e = t * 4, z = e / tile, r = e - z * tile;
        const float *p = part + z * chunks * tile + r;
        for (int c = cg; c < chunks; c += 16) {
            const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)c * tile);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[cg][o] = s;
    __syncthreads();
    if (cg == 0 && t < total4) {
#pragma unroll
        for (int g = 1; g < 16; ++g) {
            const float4 v = red[g][o];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + t * 4) = s;
    }
}

inline int xty_chunks(int N)
{
    int c = (N + 127) / 128;  // >= 128 reduction rows per workgroup
    if (c > kMaxChunks) c = kMaxChunks;
    return c < 1 ? 1 : c;
}

}  // namespace

extern "C" {

size_t zira_xty_workspace_floats(int B, int N, int a, int b)
{
    if (B <= 0 || N <= 0 || a <= 0 || b <= 0) return 0;
    return (size_t)B * xty_chunks(N) * a * b;
}

int zira_xty_f32(const float *X, const float *Y, int B, int N, int a, int b, int x_transposed,
                 float *out, float *workspace, void *stream)
{
    if (!X || !Y || !out || !workspace || B <= 0 || N <= 0 || a <= 0 || b <= 0 || (a & 3) || (b & 3))
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = xty_chunks(N);
    const int chunk_rows = (N + chunks - 1) / chunks;
    if (a <= b) {  // wide tile along b
        const int a_tiles = (a + 63) / 64, b_tiles = (b + 255) / 256;
        hipLaunchKernelGGL((xty_partial<64, 256>), dim3(chunks, a_tiles * b_tiles, B), dim3(kThreads), 0,
                           st, X, Y, N, a, b, x_transposed, chunk_rows, b_tiles, workspace);
    } else {
        const int a_tiles = (a + 255) / 256, b_tiles = (b + 63) / 64;
        hipLaunchKernelGGL((xty_partial<256, 64>), dim3(chunks, a_tiles * b_tiles, B), dim3(kThreads), 0,
                           st, X, Y, N, a, b, x_transposed, chunk_rows, b_tiles, workspace);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const size_t tile = (size_t)a * b, total4 = (size_t)B * tile / 4;
    hipLaunchKernelGGL(xty_fold, dim3((unsigned)((total4 + 15) / 16)), dim3(kThreads), 0, st,
                       workspace, chunks, tile, total4, out);
    return (int)hipGetLastError();
}

}  // extern "C"
