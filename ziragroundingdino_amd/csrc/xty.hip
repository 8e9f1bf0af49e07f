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
// split over `chunks` workgroups per output tile (a 64 x 256 or 256 x 64 tile on fp32 MFMA,
// operands straight from global memory), partial tiles go to a workspace and a second kernel
// folds them in a fixed order (deterministic, no atomics).
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "zira_msda.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxChunks = 128;

typedef float v16f __attribute__((ext_vector_type(16)));
// 16-byte load from a 4-byte aligned address (rows of odd length): one global_load_dwordx4 on gfx950
__device__ __forceinline__ float4 load4_unaligned(const float *p)
{
    float4 v;
    __builtin_memcpy(&v, p, sizeof(v));
    return v;
}

// One workgroup = 4 waves = one TA x TB tile of the output (64 x 256 or 256 x 64) over one chunk of
// the reduction; a wave owns a 64 x 64 sub-tile as 2 x 2 MFMA blocks (v_mfma_f32_32x32x2_f32: two
// reduction rows per instruction).  The operands of that instruction are exactly one float per
// lane -- lane l holds X[n + l/32][i0 + l%32] and Y[n + l/32][j0 + l%32] -- so they are loaded
// straight from global memory (two coalesced 128-byte rows per load), no LDS staging.
// kUnroll row pairs are fetched before the first of their MFMAs is issued.
template <int TA, int TB, bool XT>
__global__ __launch_bounds__(kThreads) void xty_partial(const float *__restrict__ X,
                                                        const float *__restrict__ Y, int N, int a,
                                                        int b, int chunk_rows, int b_tiles,
                                                        float *__restrict__ part)
{
    static_assert(TA * TB == 64 * 256, "four 64 x 64 wave tiles");
    constexpr int kUnroll = 8;  // MFMA steps (row pairs) per batch of operand loads
    const int chunk = blockIdx.x, chunks = gridDim.x, z = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int a0 = (blockIdx.y / b_tiles) * TA + (TA == 256 ? 64 * wave : 0);
    const int b0 = (blockIdx.y % b_tiles) * TB + (TB == 256 ? 64 * wave : 0);
    const int k = lane >> 5, c = lane & 31;
    const float *Xz = X + (size_t)z * N * a;
    const float *Yz = Y + (size_t)z * N * b;
    const int n_begin = chunk * chunk_rows;
    const int n_end = (n_begin + chunk_rows < N) ? n_begin + chunk_rows : N;
    const bool ia[2] = {a0 + c < a, a0 + 32 + c < a};
    const bool jb[2] = {b0 + c < b, b0 + 32 + c < b};

    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A batch = kUnroll MFMA steps' worth of operands in registers; two batches alternate so that
    // one is in flight while the other one is multiplied (the kernel is otherwise bound by the
    // latency of its own loads: one workgroup per CU).
    struct Batch {
        float xa[XT ? 1 : kUnroll][2];
        float4 xq[XT ? kUnroll / 2 : 1][2];
        float yb[kUnroll][2];
    };
    constexpr int kRows = 2 * kUnroll;  // reduction rows per batch
    // Loads are unconditional (indices clamped into the matrix) and values outside the chunk / the
    // matrix are cleared with a bit mask: written with predicates the compiler emitted a branch and
    // a full wait per load, and the MFMA pipe idled.
    const int ca[2] = {ia[0] ? a0 + c : a - 1, ia[1] ? a0 + 32 + c : a - 1};
    const int cb[2] = {jb[0] ? b0 + c : b - 1, jb[1] ? b0 + 32 + c : b - 1};
    const unsigned ma[2] = {ia[0] ? 0xFFFFFFFFu : 0u, ia[1] ? 0xFFFFFFFFu : 0u};
    const unsigned mb[2] = {jb[0] ? 0xFFFFFFFFu : 0u, jb[1] ? 0xFFFFFFFFu : 0u};
    auto keep = [](float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); };
    const bool cols_full = a0 + 64 <= a && b0 + 64 <= b;  // wave-uniform
    auto fetch = [&](Batch &t, int n0) {
        if (cols_full && n0 + kRows <= n_end) {  // interior batch: no clamps, no masks
            const float *yp = Yz + (size_t)(n0 + k) * b + b0 + c;
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                t.yb[u][0] = yp[(size_t)2 * u * b];
                t.yb[u][1] = yp[(size_t)2 * u * b + 32];
            }
            if (!XT) {
                const float *xp = Xz + (size_t)(n0 + k) * a + a0 + c;
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    t.xa[u][0] = xp[(size_t)2 * u * a];
                    t.xa[u][1] = xp[(size_t)2 * u * a + 32];
                }
            } else {
                const float *xp = Xz + (size_t)(a0 + c) * N + n0;
#pragma unroll
                for (int q = 0; q < kUnroll / 2; ++q) {
                    t.xq[q][0] = load4_unaligned(xp + 4 * q);
                    t.xq[q][1] = load4_unaligned(xp + (size_t)32 * N + 4 * q);
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int n = n0 + 2 * u + k;
            const unsigned mrow = n < n_end ? 0xFFFFFFFFu : 0u;
            const int nc = n < N ? n : N - 1;
            const float *yr = Yz + (size_t)nc * b;
#pragma unroll
            for (int h = 0; h < 2; ++h) t.yb[u][h] = keep(yr[cb[h]], mrow & mb[h]);
            if (!XT) {
                const float *xr = Xz + (size_t)nc * a;
#pragma unroll
                for (int h = 0; h < 2; ++h) t.xa[u][h] = keep(xr[ca[h]], mrow & ma[h]);
            }
        }
        if (XT) {  // X stored a x N: a lane reads 4 consecutive n of its column = two MFMA steps
#pragma unroll
            for (int q = 0; q < kUnroll / 2; ++q) {
                const int n = n0 + 4 * q;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // one 16-byte read (rows of X have odd length: 4-byte alignment only); the last
                    // rows of the matrix are read element by element
                    const float *src = Xz + (size_t)ca[h] * N + n;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n + 3 < N) {
                        v = load4_unaligned(src);
                    } else {
                        if (n < N) v.x = src[0];
                        if (n + 1 < N) v.y = src[1];
                        if (n + 2 < N) v.z = src[2];
                    }
                    v.x = keep(v.x, (n < n_end ? 0xFFFFFFFFu : 0u) & ma[h]);
                    v.y = keep(v.y, (n + 1 < n_end ? 0xFFFFFFFFu : 0u) & ma[h]);
                    v.z = keep(v.z, (n + 2 < n_end ? 0xFFFFFFFFu : 0u) & ma[h]);
                    v.w = keep(v.w, (n + 3 < n_end ? 0xFFFFFFFFu : 0u) & ma[h]);
                    t.xq[q][h] = v;
                }
            }
        }
    };
    auto multiply = [&](const Batch &t) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float xv;
                if (XT) {
                    const float4 q4 = t.xq[u / 2][i];
                    xv = (u & 1) ? (k ? q4.w : q4.z) : (k ? q4.y : q4.x);
                } else {
                    xv = t.xa[u][i];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, t.yb[u][j], acc[i][j], 0, 0, 0);
            }
    };
    Batch ping, pong;
    fetch(ping, n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += 2 * kRows) {
        fetch(pong, n0 + kRows);
        multiply(ping);
        fetch(ping, n0 + 2 * kRows);
        multiply(pong);
    }
    // partial tile -> workspace [z][chunk][a][b]; accumulator register r of lane l is the element
    // (row 8*(r/4) + 4*(l/32) + r%4, column l%32) of its 32 x 32 block
    float *pz = part + ((size_t)z * chunks + chunk) * a * b;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = b0 + 32 * j + c;
            if (col >= b) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = a0 + 32 * i + 8 * (r / 4) + 4 * k + (r % 4);
                if (row < a) pz[(size_t)row * b + col] = acc[i][j][r];
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
        const dim3 grid(chunks, a_tiles * b_tiles, B);
        if (x_transposed)
            hipLaunchKernelGGL((xty_partial<64, 256, true>), grid, dim3(kThreads), 0, st, X, Y, N, a, b,
                               chunk_rows, b_tiles, workspace);
        else
            hipLaunchKernelGGL((xty_partial<64, 256, false>), grid, dim3(kThreads), 0, st, X, Y, N, a, b,
                               chunk_rows, b_tiles, workspace);
    } else {
        const int a_tiles = (a + 255) / 256, b_tiles = (b + 63) / 64;
        const dim3 grid(chunks, a_tiles * b_tiles, B);
        if (x_transposed)
            hipLaunchKernelGGL((xty_partial<256, 64, true>), grid, dim3(kThreads), 0, st, X, Y, N, a, b,
                               chunk_rows, b_tiles, workspace);
        else
            hipLaunchKernelGGL((xty_partial<256, 64, false>), grid, dim3(kThreads), 0, st, X, Y, N, a, b,
                               chunk_rows, b_tiles, workspace);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const size_t tile = (size_t)a * b, total4 = (size_t)B * tile / 4;
    hipLaunchKernelGGL(xty_fold, dim3((unsigned)((total4 + 15) / 16)), dim3(kThreads), 0, st,
                       workspace, chunks, tile, total4, out);
    return (int)hipGetLastError();
}

}  // extern "C"
