// refpoints.hip -- the sine embedding of the decoder's reference boxes in one launch (C ABI: zira_sine_embed_f32).
//
// Reference: gen_sineembed_for_position (groundingdino/models/GroundingDINO/utils.py:204-231) -- per coordinate
// x * 2 pi / 10000^(2 (i // 2) / 128), sin on even and cos on odd channels, parts ordered (y, x, w, h).  It runs once
// per decoder layer on [900, B, 4] boxes without gradients (the boxes are detached between layers); as PyTorch ops it
// is 6 launch-bound kernels per layer.  The arithmetic is the PyTorch chain's, operation by operation (separately
// rounded multiply and divide, the same libm sin / cos), so the result is bit-identical to it.  (inverse_sigmoid was
// tried the same way: ATen's log differs from libm's logf in the last bit of a third of the values -- left to PyTorch.)
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

// out[row, part * T + i] with part p reading coordinate order[p]; a thread per output element
__global__ __launch_bounds__(256) void sine_embed_kernel(const float *__restrict__ pos, const float *__restrict__ dim_t,
                                                         long long rows, int C, int T, float scale,
                                                         float *__restrict__ out)
{
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C * T) return;
    const int i = (int)(idx % T);
    const long long rp = idx / T;
    const int part = (int)(rp % C);
    const long long row = rp / C;
    const int coord = part == 0 ? 1 : (part == 1 ? 0 : part);   // (y, x, w, h) <- (x, y, w, h)
    const float arg = __fdiv_rn(__fmul_rn(pos[row * C + coord], scale), dim_t[i]);
    out[idx] = (i & 1) ? cosf(arg) : sinf(arg);
}

// ---- what a decoder layer needs of the current boxes, one launch (reference transformer_for_adapter.py:760-770) ----
//   ref_in[q, b, l, c] = ref[q, b, c] * ratio[b, l, c & 1]      (reference_points[:, :, None] * cat([valid_ratios, valid_ratios], -1))
//   ref_bf[b, q, l, c] = the same, batch-first (the layout the MSDA sampling kernel reads)
//   sine[q, b, :]      = the sine embedding of ref_in[q, b, 0, :]  (gen_sineembed_for_position: the kernel above)
// A thread per sine element; the threads with i < L also write the two box tensors.  Same multiplies and divides as the op chain.
__global__ __launch_bounds__(256) void decoder_prep_kernel(const float *__restrict__ ref, const float *__restrict__ ratio,
                                                           const float *__restrict__ dim_t, int Q, int B, int L, int T, float scale,
                                                           float *__restrict__ ref_in, float *__restrict__ ref_bf,
                                                           float *__restrict__ sine)
{
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Q * B * 4 * T) return;
    const int i = (int)(idx % T);
    const long long rp = idx / T;
    const int part = (int)(rp & 3);
    const long long row = rp >> 2;           // q * B + b
    const int b = (int)(row % B);
    const long long q = row / B;
    if (i < L) {
        const float v = __fmul_rn(ref[row * 4 + part], ratio[((long long)b * L + i) * 2 + (part & 1)]);
        ref_in[(row * L + i) * 4 + part] = v;
        ref_bf[(((long long)b * Q + q) * L + i) * 4 + part] = v;
    }
    const int coord = part == 0 ? 1 : (part == 1 ? 0 : part);   // (y, x, w, h) <- (x, y, w, h)
    const float x = __fmul_rn(ref[row * 4 + coord], ratio[(long long)b * L * 2 + (coord & 1)]);
    const float arg = __fdiv_rn(__fmul_rn(x, scale), dim_t[i]);
    sine[idx] = (i & 1) ? cosf(arg) : sinf(arg);
}

// ---- iterative box refinement: the last layer of the box MLP with the inverse-sigmoid / sigmoid around it ----
// new_ref[row, j] = sigmoid(<h[row, :], w[j, :]> + b[j] + log(max(x, eps) / max(1 - x, eps))),  x = clamp(ref[row, j], 0, 1)
// (reference transformer_for_adapter.py:790-797: delta_unsig = bbox_embed(output); (delta_unsig + inverse_sigmoid(ref)).sigmoid();
// util/misc.py:704-708).  A wave per row: a lane takes float4s of the row, four dot products, wave reduction.
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

__global__ __launch_bounds__(256) void box_refine_fwd_kernel(const float *__restrict__ h, const float *__restrict__ w,
                                                             const float *__restrict__ b, const float *__restrict__ ref,
                                                             long long rows, int K, float eps, float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 4 * lane; k < K; k += 256) {
        const float4 x = *reinterpret_cast<const float4 *>(h + row * K + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 y = *reinterpret_cast<const float4 *>(w + (long long)j * K + k);
            acc[j] += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = wave_sum(acc[j]);
    if (lane < 4) {
        const float d = (lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3]) + b[lane];
        const float x = fminf(fmaxf(ref[row * 4 + lane], 0.f), 1.f);
        const float inv = logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
        out[row * 4 + lane] = 1.f / (1.f + expf(-(d + inv)));
    }
}

// g_h[row, k] = (sum_j g_new[row, j] * s (1 - s) * w[j, k]) where h[row, k] > 0, s = new_ref[row, j]: the gradient in front
// of the ReLU that feeds the last layer of the box MLP
__global__ __launch_bounds__(256) void box_refine_bwd_kernel(const float *__restrict__ g_new, const float *__restrict__ new_ref,
                                                             const float *__restrict__ w, const float *__restrict__ h,
                                                             long long rows, int K, float *__restrict__ g_h)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float gd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float s = new_ref[row * 4 + j];
        gd[j] = g_new[row * 4 + j] * (s * (1.f - s));
    }
    for (int k = 4 * lane; k < K; k += 256) {
        const float4 m = *reinterpret_cast<const float4 *>(h + row * K + k);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 y = *reinterpret_cast<const float4 *>(w + (long long)j * K + k);
            r.x += gd[j] * y.x; r.y += gd[j] * y.y; r.z += gd[j] * y.z; r.w += gd[j] * y.w;
        }
        r.x = m.x > 0.f ? r.x : 0.f; r.y = m.y > 0.f ? r.y : 0.f; r.z = m.z > 0.f ? r.z : 0.f; r.w = m.w > 0.f ? r.w : 0.f;
        *reinterpret_cast<float4 *>(g_h + row * K + k) = r;
    }
}

}  // namespace

extern "C" int zira_box_refine_fwd_f32(const float *h, const float *w, const float *b, const float *ref, long long rows, int K,
                                       float eps, float *new_ref, void *stream)
{
    if (!h || !w || !b || !ref || !new_ref || rows < 0 || K <= 0 || (K & 3)) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)h | (uintptr_t)w) & 15) != 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(box_refine_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, h, w, b, ref,
                       rows, K, eps, new_ref);
    return (int)hipGetLastError();
}

extern "C" int zira_box_refine_bwd_f32(const float *g_new, const float *new_ref, const float *w, const float *h, long long rows,
                                       int K, float *g_h, void *stream)
{
    if (!g_new || !new_ref || !w || !h || !g_h || rows < 0 || K <= 0 || (K & 3)) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)h | (uintptr_t)w | (uintptr_t)g_h) & 15) != 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(box_refine_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, g_new, new_ref,
                       w, h, rows, K, g_h);
    return (int)hipGetLastError();
}

extern "C" int zira_decoder_prep_f32(const float *ref, const float *ratio, const float *dim_t, int Q, int B, int L, int T,
                                    float scale, float *ref_in, float *ref_bf, float *sine, void *stream)
{
    if (!ref || !ratio || !dim_t || !ref_in || !ref_bf || !sine || Q < 0 || B <= 0 || L <= 0 || T < L) return (int)hipErrorInvalidValue;
    const long long n = (long long)Q * B * 4 * T;
    if (n == 0) return 0;
    hipLaunchKernelGGL(decoder_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref, ratio, dim_t,
                       Q, B, L, T, scale, ref_in, ref_bf, sine);
    return (int)hipGetLastError();
}

extern "C" int zira_sine_embed_f32(const float *pos, const float *dim_t, long long rows, int C, int T, float scale,
                                   float *out, void *stream)
{
    if (!pos || !dim_t || !out || rows < 0 || (C != 2 && C != 4) || T <= 0) return (int)hipErrorInvalidValue;
    const long long n = rows * C * T;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sine_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, dim_t,
                       rows, C, T, scale, out);
    return (int)hipGetLastError();
}
