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

// ---- level geometry from the padding mask: valid ratios, the encoder's reference points, the two-stage proposals ----
// Three op chains of the reference that depend on nothing but the padding mask and the level table and that PyTorch runs as
// ~150 launch-bound kernels per step (3 us each on the critical path):
//   valid ratios      transformer_for_adapter.py:226-233   [B, L, 2] = (valid columns / W, valid rows / H) of every level
//   reference points  transformer_for_adapter.py:482-497   [B, S, L, 2] = pixel centre / (ratio * extent) of its level, * ratio of level j
//   proposals         utils.py:56-116                      [B, S, 4] = (pixel centre / valid extent, 0.05 * 2^level), as p / (1 - p)
//                                                          with +inf where padded or outside (0.01, 0.99) -- the caller's ATen
//                                                          log finishes it (ATen's log and libm's logf differ in the last bit)
// Same fp32 operations in the same order as the op chains (separately rounded: contraction is off), so the results are
// bit-identical: linspace(0.5, H - 0.5, H)[i] is exactly i + 0.5 (its step is exactly 1).
__device__ __forceinline__ int level_of(const int64_t *__restrict__ start, int L, long long s)
{
    int l = 0;
    for (int i = 1; i < L; ++i) l = s >= start[i] ? i : l;
    return l;
}

// counts[b, l] = (valid columns, valid rows) as floats; a block per (b, l)
__global__ __launch_bounds__(256) void level_counts_kernel(const unsigned char *__restrict__ mask, const int64_t *__restrict__ shapes,
                                                           const int64_t *__restrict__ start, long long S, int L,
                                                           float *__restrict__ counts, float *__restrict__ ratios)
{
#pragma clang fp contract(off)
    __shared__ unsigned cnt[2];
    const int b = blockIdx.x / L, l = blockIdx.x % L;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const unsigned char *m = mask + (long long)b * S + start[l];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned cw = 0, ch = 0;
    for (int x = threadIdx.x; x < W; x += 256) cw += m[x] ? 0u : 1u;
    for (int y = threadIdx.x; y < H; y += 256) ch += m[(long long)y * W] ? 0u : 1u;
    if (cw) atomicAdd(&cnt[0], cw);
    if (ch) atomicAdd(&cnt[1], ch);
    __syncthreads();
    if (threadIdx.x < 2) {
        const float c = (float)cnt[threadIdx.x];
        if (counts) counts[((long long)b * L + l) * 2 + threadIdx.x] = c;
        // (ATen divides a tensor by a host scalar as a multiplication by the scalar's fp32 reciprocal)
        if (ratios) ratios[((long long)b * L + l) * 2 + threadIdx.x] = __fmul_rn(c, __fdiv_rn(1.f, (float)(threadIdx.x ? H : W)));
    }
}

__global__ __launch_bounds__(256) void encoder_ref_points_kernel(const float *__restrict__ ratios, const int64_t *__restrict__ shapes,
                                                                 const int64_t *__restrict__ start, int B, long long S, int L,
                                                                 float *__restrict__ out)
{
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * S) return;
    const long long s = idx % S;
    const int b = (int)(idx / S);
    const int l = level_of(start, L, s);
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long long r = s - start[l];
    const int y = (int)(r / W), x = (int)(r - (long long)y * W);
    const float *vr = ratios + (long long)b * L * 2;
    const float rx = __fdiv_rn((float)x + 0.5f, __fmul_rn(vr[2 * l], (float)W));
    const float ry = __fdiv_rn((float)y + 0.5f, __fmul_rn(vr[2 * l + 1], (float)H));
    float2 *o = reinterpret_cast<float2 *>(out) + idx * L;
    for (int j = 0; j < L; ++j) o[j] = make_float2(__fmul_rn(rx, vr[2 * j]), __fmul_rn(ry, vr[2 * j + 1]));
}

__global__ __launch_bounds__(256) void encoder_proposals_kernel(const unsigned char *__restrict__ mask, const float *__restrict__ counts,
                                                                const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                                                int B, long long S, int L, float *__restrict__ odds,
                                                                unsigned char *__restrict__ drop)
{
#pragma clang fp contract(off)
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * S) return;
    const long long s = idx % S;
    const int b = (int)(idx / S);
    const int l = level_of(start, L, s);
    const int W = (int)shapes[2 * l + 1];
    const long long r = s - start[l];
    const int y = (int)(r / W), x = (int)(r - (long long)y * W);
    const float *c = counts + ((long long)b * L + l) * 2;
    const float wh = __fmul_rn(0.05f, (float)(1u << l));
    const float p[4] = {__fdiv_rn((float)x + 0.5f, c[0]), __fdiv_rn((float)y + 0.5f, c[1]), wh, wh};
    bool bad = mask[idx] != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) bad = bad || !(p[k] > 0.01f && p[k] < 0.99f);
    const float inf = __uint_as_float(0x7f800000u);
    float4 o;
    o.x = bad ? inf : __fdiv_rn(p[0], 1.f - p[0]);
    o.y = bad ? inf : __fdiv_rn(p[1], 1.f - p[1]);
    o.z = bad ? inf : __fdiv_rn(p[2], 1.f - p[2]);
    o.w = bad ? inf : __fdiv_rn(p[3], 1.f - p[3]);
    reinterpret_cast<float4 *>(odds)[idx] = o;
    drop[idx] = bad ? 1 : 0;
}

// ---- the box head's last step, elementwise: out = sigmoid(delta + inverse_sigmoid(ref)) and its gradients ----
// (groundingdino_dual_zero_rep_branch.py:563-569 with inverse_sigmoid util/misc.py:704-708: as ATen ops 8 launches forward and
//  ~20 backward on 43 200 elements; autograd's conventions at the clamps: the gradient passes where min <= x <= max)
__global__ __launch_bounds__(256) void box_head_fwd_kernel(const float *__restrict__ delta, const float *__restrict__ ref, long long n,
                                                           float eps, float *__restrict__ out)
{
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = fminf(fmaxf(ref[i], 0.f), 1.f);
    const float inv = logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
    out[i] = 1.f / (1.f + expf(-(delta[i] + inv)));
}

__global__ __launch_bounds__(256) void box_head_bwd_kernel(const float *__restrict__ g_out, const float *__restrict__ out,
                                                           const float *__restrict__ ref, long long n, float eps,
                                                           float *__restrict__ g_delta, float *__restrict__ g_ref)
{
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = out[i];
    const float gu = g_out[i] * (s * (1.f - s));
    if (g_delta) g_delta[i] = gu;
    if (g_ref) {
        const float r = ref[i];
        float d = 0.f;
        if (r >= 0.f && r <= 1.f) {
            const float om = 1.f - r;
            d = (r >= eps ? 1.f / r : 0.f) + (om >= eps ? 1.f / om : 0.f);
        }
        g_ref[i] = gu * d;
    }
}

// ---- sine position encoding of a feature level from its padding mask (PositionEmbeddingSineHW, position_encoding.py:78-134) ----
// out[b, y, x, :] = (pos_y | pos_x), pos_*[i] = sin / cos (even / odd i) of embed / dim_t[i]; embed_y = number of unpadded pixels of
// column x in rows 0 .. y (cumsum), embed_x along the row; normalised: embed / (last + eps) * scale.  As ATen ops: two cumsums, the
// normalisation, two pow / div, four sin / cos, two stacks, a cat -- ~15 launches per level, most of them passes over the level's
// [B, H, W, 256] output.  Here: a block per (b, y) image row; same separately rounded fp32 operations, bit-identical.
// grid (B * H, ceil(W / 64)): a block writes 64 pixels of one image row.  (Until round 6 a block took a whole row and every thread
// walked its column's H mask bytes and its row's x + 1 alone: 64 us for the 100 x 167 level, all of it latency of one block per CU.)
__global__ __launch_bounds__(256) void sine_pos_hw_kernel(const unsigned char *__restrict__ mask, int H, int W, int F, int normalize,
                                                          float scale, float eps, const float *__restrict__ dim_t_y,
                                                          const float *__restrict__ dim_t_x, float *__restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ int part_upto[4][64], part_total[4][64], wave_count[4];
    __shared__ float ey[64], ex[64];
    __shared__ float x_last;
    const int b = blockIdx.x / H, y = blockIdx.x - b * H;
    const int x0 = blockIdx.y * 64, x1 = x0 + 64 < W ? x0 + 64 : W;
    const unsigned char *m = mask + (size_t)b * H * W;
    const int tx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // column counts of the block's 64 columns: wave w sums rows w, w + 4, ...
    {
        const int x = x0 + tx;
        int upto = 0, total = 0;
        if (x < x1)
            for (int yy = wv; yy < H; yy += 4) {
                const int v = m[(size_t)yy * W + x] ? 0 : 1;
                total += v;
                upto += yy <= y ? v : 0;
            }
        part_upto[wv][tx] = upto;
        part_total[wv][tx] = total;
    }
    // the row's running count of unpadded pixels up to the block's columns, and the row's total (its last element): 256 pixels per
    // pass, a wave's 64 by ballot, the waves joined through LDS
    int row_total = 0;
    for (int xb = 0; xb < W; xb += 256) {
        const int x = xb + threadIdx.x;
        const bool v = x < W && !m[(size_t)y * W + x];
        const unsigned long long bal = __ballot(v);
        if (tx == 0) wave_count[wv] = __popcll(bal);
        __syncthreads();
        int carry = 0;
        for (int w = 0; w < wv; ++w) carry += wave_count[w];
        const int incl = row_total + carry + __popcll(bal & ((2ull << tx) - 1ull));   // unpadded pixels in [0, x]
        if (x >= x0 && x < x1) ex[x - x0] = (float)incl;
        row_total += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
    if (threadIdx.x < 64 && x0 + (int)threadIdx.x < x1) {
        const int upto = part_upto[0][tx] + part_upto[1][tx] + part_upto[2][tx] + part_upto[3][tx];
        const int total = part_total[0][tx] + part_total[1][tx] + part_total[2][tx] + part_total[3][tx];
        float fy = (float)upto;
        if (normalize) fy = __fmul_rn(__fdiv_rn(fy, __fadd_rn((float)total, eps)), scale);
        ey[tx] = fy;
    }
    if (threadIdx.x == 0) x_last = (float)row_total;     // (the running count at x = W - 1)
    __syncthreads();
    const float xl = x_last;
    const int C = 2 * F;
    float *o = out + (((size_t)b * H + y) * W + x0) * C;
    // channels 2 k and 2 k + 1 share dim_t (temperature^(2 (i // 2) / F)) and so the angle: its sine and cosine in one evaluation
    // (the same bits as sinf / cosf on their own: tests/test_geometry_gpu.py)
    for (int i = threadIdx.x; i < (x1 - x0) * F; i += 256) {
        const int x = i / F, c = 2 * (i - x * F);
        float e, d, d1;
        if (c < F) {
            e = ey[x];
            d = dim_t_y[c];
            d1 = dim_t_y[c + 1];
        } else {
            e = ex[x];
            if (normalize) e = __fmul_rn(__fdiv_rn(e, __fadd_rn(xl, eps)), scale);
            d = dim_t_x[c - F];
            d1 = dim_t_x[c - F + 1];
        }
        const float a = __fdiv_rn(e, d);
        float sn, cs;
        sincosf(a, &sn, &cs);
        if (d1 != d) cs = cosf(__fdiv_rn(e, d1));        // (a caller's own dim_t table)
        *reinterpret_cast<float2 *>(o + (size_t)x * C + c) = make_float2(sn, cs);
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

extern "C" int zira_level_valid_ratios_f32(const void *mask, const int64_t *shapes, const int64_t *start, int B, long long S, int L,
                                           float *counts, float *ratios, void *stream)
{
    if (!mask || !shapes || !start || B <= 0 || S <= 0 || L <= 0 || L > 30 || (!counts && !ratios)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(level_counts_kernel, dim3((unsigned)(B * L)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned char *>(mask), shapes, start, S, L, counts, ratios);
    return (int)hipGetLastError();
}

extern "C" int zira_encoder_ref_points_f32(const float *ratios, const int64_t *shapes, const int64_t *start, int B, long long S, int L,
                                           float *ref_points, void *stream)
{
    if (!ratios || !shapes || !start || !ref_points || B <= 0 || S <= 0 || L <= 0 || L > 30) return (int)hipErrorInvalidValue;
    if ((uintptr_t)ref_points & 7) return (int)hipErrorInvalidValue;
    const long long n = (long long)B * S;
    hipLaunchKernelGGL(encoder_ref_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ratios, shapes,
                       start, B, S, L, ref_points);
    return (int)hipGetLastError();
}

extern "C" int zira_encoder_proposals_f32(const void *mask, const int64_t *shapes, const int64_t *start, int B, long long S, int L,
                                          float *counts_scratch, float *odds, void *drop, void *stream)
{
    if (!mask || !shapes || !start || !counts_scratch || !odds || !drop || B <= 0 || S <= 0 || L <= 0 || L > 30)
        return (int)hipErrorInvalidValue;
    if ((uintptr_t)odds & 15) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(level_counts_kernel, dim3((unsigned)(B * L)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned char *>(mask), shapes, start, S, L, counts_scratch, (float *)nullptr);
    const long long n = (long long)B * S;
    hipLaunchKernelGGL(encoder_proposals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned char *>(mask), counts_scratch, shapes, start, B, S, L, odds,
                       reinterpret_cast<unsigned char *>(drop));
    return (int)hipGetLastError();
}

extern "C" int zira_box_head_fwd_f32(const float *delta, const float *ref, long long n, float eps, float *out, void *stream)
{
    if (!delta || !ref || !out || n < 0) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(box_head_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, delta, ref, n, eps, out);
    return (int)hipGetLastError();
}

extern "C" int zira_box_head_bwd_f32(const float *g_out, const float *out, const float *ref, long long n, float eps, float *g_delta,
                                     float *g_ref, void *stream)
{
    if (!g_out || !out || !ref || (!g_delta && !g_ref) || n < 0) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    hipLaunchKernelGGL(box_head_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_out, out, ref, n,
                       eps, g_delta, g_ref);
    return (int)hipGetLastError();
}

extern "C" int zira_sine_pos_hw_f32(const void *mask, int B, int H, int W, int F, int normalize, float scale, float eps,
                                    const float *dim_t_y, const float *dim_t_x, float *out, void *stream)
{
    if (!mask || !dim_t_y || !dim_t_x || !out || B <= 0 || H <= 0 || W <= 0 || F <= 0 || (F & 1) || W > 8192) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(sine_pos_hw_kernel, dim3((unsigned)(B * H), (unsigned)((W + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned char *>(mask), H, W, F, normalize, scale, eps, dim_t_y, dim_t_x, out);
    return (int)hipGetLastError();
}
