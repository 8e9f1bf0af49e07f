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
#include <stdint.h>

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

// The fusion block's own three products at 4 heads x 32 text tokens -- 128 x 256 (twice) and 256 x 128 outputs -- in a form
// whose loads are 16 / 8 bytes per lane.  xty_partial issues one 4-byte load per lane and MFMA operand (256 bytes per
// instruction); the CU's texture addresser takes 16 cycles per wave instruction whatever its width, so that for these two
// block-per-CU shapes the addresser was as busy as the MFMA pipe (512 cycles each per row pair) and the kernel ran at
// 40 % of the fp32 MFMA rate (54 us; this form: 36 us).  Here the 128-wide operand is read as one float4 per lane -- lane c holds columns
// 4 c .. 4 c + 3, which feed four MFMA blocks whose row (column) number c stands for column 4 c + q -- and a 64-column
// slice of the 256-wide operand as one float2 per lane: 2 load instructions per 8 MFMAs.  A block owns a 128 x 64
// (64 x 128 with SWAP) slice of the output for 1 / 32 of the rows; its four waves take every fourth row pair and add
// their tiles through LDS in a fixed order, so a block writes one partial tile (8 MB of partials instead of 34).
template <bool SWAP>
__global__ __launch_bounds__(kThreads) void xty_rows128(const float *__restrict__ X, const float *__restrict__ Y, int N,
                                                        int chunk_rows, float *__restrict__ part)
{
    constexpr int a = SWAP ? 256 : 128, b = SWAP ? 128 : 256;
    constexpr int IA = SWAP ? 2 : 4, JB = SWAP ? 4 : 2;     // MFMA blocks per wave along a / b = floats per lane and load
    constexpr int RT = SWAP ? 64 : 128, CT = SWAP ? 128 : 64;
    constexpr int kU = 8;                                     // row pairs per batch of loads
    extern __shared__ float tile[];   // [4 waves][RT][CT]: 128 KB
    const int chunk = blockIdx.x, chunks = gridDim.x, cb = blockIdx.y, z = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = lane >> 5, c = lane & 31;
    const float *Xz = X + (size_t)z * N * a + (SWAP ? cb * 64 : 0) + IA * c;
    const float *Yz = Y + (size_t)z * N * b + (SWAP ? 0 : cb * 64) + JB * c;
    const int n_begin = chunk * chunk_rows;
    const int n_end = (n_begin + chunk_rows < N) ? n_begin + chunk_rows : N;
    const int nsteps = n_end > n_begin ? (n_end - n_begin + 7) / 8 : 0;   // a step = 8 rows: a pair per wave

    v16f acc[IA][JB];
#pragma unroll
    for (int i = 0; i < IA; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Batch {
        float xa[kU][IA];
        float yb[kU][JB];
    };
    auto load_x = [&](const float *p, float (&x)[IA]) {
        if (IA == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            x[0] = v.x; x[1] = v.y; x[IA - 2] = v.z; x[IA - 1] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(p);
            x[0] = v.x; x[1] = v.y;
        }
    };
    auto load_y = [&](const float *p, float (&y)[JB]) {
        if (JB == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            y[0] = v.x; y[1] = v.y; y[JB - 2] = v.z; y[JB - 1] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(p);
            y[0] = v.x; y[1] = v.y;
        }
    };
    // Loads never depend on where the chunk ends (row numbers clamped into the matrix, no branch): the rows past the end are
    // cleared when they are multiplied.  (Cleared right after the load, every tail batch waited for ALL loads in flight.)
    auto fetch = [&](Batch &t, int s0) {   // steps s0 .. s0 + kU - 1
        const int n0 = n_begin + 8 * s0 + 2 * wave + k;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int n = n0 + 8 * u;
            const size_t nc = (size_t)(n < N ? n : N - 1);
            load_x(Xz + nc * a, t.xa[u]);
            load_y(Yz + nc * b, t.yb[u]);
        }
    };
    auto multiply = [&](const Batch &t, int s0) {   // (one form for every batch: a branch here and the compiler re-rolls the ring)
        const int n0 = n_begin + 8 * s0 + 2 * wave + k;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            float xa[IA], yb[JB];
            const unsigned m = n0 + 8 * u < n_end ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int i = 0; i < IA; ++i) xa[i] = __uint_as_float(__float_as_uint(t.xa[u][i]) & m);
#pragma unroll
            for (int j = 0; j < JB; ++j) yb[j] = __uint_as_float(__float_as_uint(t.yb[u][j]) & m);
#pragma unroll
            for (int i = 0; i < IA; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i], yb[j], acc[i][j], 0, 0, 0);
        }
    };
    // two batches alternate: one is on its way while the other one is multiplied (one wave per SIMD: nothing else hides the
    // loads).  A ring of three measured the same (38.2 / 36.5 us against 37.2 / 35.3): the loads are hidden, what is left above
    // the 24 us of MFMA issue (12 batches x 64 x 64 cycles) is the launch, the first batch, the LDS exchange and the tile's store.
    const int nb = (nsteps + kU - 1) / kU;
    Batch ping, pong;
    fetch(ping, 0);
    for (int bi = 0; bi < nb; bi += 2) {
        fetch(pong, (bi + 1) * kU);
        __builtin_amdgcn_sched_barrier(0);
        multiply(ping, bi * kU);
        __builtin_amdgcn_sched_barrier(0);
        fetch(ping, (bi + 2) * kU);
        __builtin_amdgcn_sched_barrier(0);
        multiply(pong, (bi + 1) * kU);   // (unconditional: under `if` the compiler sinks pong's loads into the branch, next to their uses)
        __builtin_amdgcn_sched_barrier(0);
    }
    // the four waves' tiles side by side in LDS, added in wave order on the way out; register r of lane l is the element
    // (8 (r / 4) + 4 k + r % 4, c) of its 32 x 32 block, and number n of block q along a stands for column IA n + q (JB n + q along b)
    {
        float *mine = tile + wave * (RT * CT);
#pragma unroll
        for (int i = 0; i < IA; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = IA * (8 * (r / 4) + 4 * k + (r % 4)) + i;
                float *dst = mine + row * CT + JB * c;
                if (JB == 4) *reinterpret_cast<float4 *>(dst) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][JB - 2][r], acc[i][JB - 1][r]);
                else *reinterpret_cast<float2 *>(dst) = make_float2(acc[i][0][r], acc[i][1][r]);
            }
    }
    __syncthreads();
    float *pz = part + ((size_t)z * chunks + chunk) * a * b + (SWAP ? (size_t)cb * 64 * b : (size_t)cb * 64);
#pragma unroll
    for (int t = 0; t < RT * CT / 4 / kThreads; ++t) {
        const int idx = threadIdx.x + t * kThreads, row = idx / (CT / 4), c4 = idx % (CT / 4);
        const float *src = tile + row * CT + c4 * 4;
        float4 o = *reinterpret_cast<const float4 *>(src);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = *reinterpret_cast<const float4 *>(src + w * (RT * CT));
            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        }
        *reinterpret_cast<float4 *>(pz + (size_t)row * b + c4 * 4) = o;
    }
}
constexpr size_t kRowsLds = 4 * 128 * 64 * sizeof(float);
constexpr int kRowsChunks = 32;   // 32 chunks x 4 slices x 2 images = one block per CU at the bench shape

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
    const size_t tile = (size_t)a * b, total4 = (size_t)B * tile / 4;
    if (!x_transposed && N >= 128 * kRowsChunks && ((a == 128 && b == 256) || (a == 256 && b == 128)) &&
        !(((uintptr_t)X | (uintptr_t)Y | (uintptr_t)workspace) & 15)) {
        // (xty_chunks(N) = 128 here: the workspace the caller sized covers the 32 chunks)
        const int chunk_rows = (((N + kRowsChunks - 1) / kRowsChunks) + 7) & ~7;
        const dim3 grid(kRowsChunks, 4, B);
        const void *fn = a == 128 ? reinterpret_cast<const void *>(&xty_rows128<false>) : reinterpret_cast<const void *>(&xty_rows128<true>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowsLds) != hipSuccess) return ZIRA_MSDA_EINVAL;
        if (a == 128) hipLaunchKernelGGL((xty_rows128<false>), grid, dim3(kThreads), kRowsLds, st, X, Y, N, chunk_rows, workspace);
        else hipLaunchKernelGGL((xty_rows128<true>), grid, dim3(kThreads), kRowsLds, st, X, Y, N, chunk_rows, workspace);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(xty_fold, dim3((unsigned)((total4 + 15) / 16)), dim3(kThreads), 0, st, workspace, kRowsChunks, tile,
                           total4, out);
        return (int)hipGetLastError();
    }
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
    hipLaunchKernelGGL(xty_fold, dim3((unsigned)((total4 + 15) / 16)), dim3(kThreads), 0, st,
                       workspace, chunks, tile, total4, out);
    return (int)hipGetLastError();
}

}  // extern "C"
