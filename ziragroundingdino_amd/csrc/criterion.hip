// criterion.hip -- the set-prediction losses of all prediction sets of a step (final layer, auxiliary decoder layers, two-stage
// encoder output) in four launches forward and two backward (C ABI: zira_stacked_losses_fwd_f32 / _bwd_f32).
//
// Reference: SetCriterion.loss_labels / loss_boxes (groundingdino/models/GroundingDINO/criterion/criterion.py:104-181) with
// sigmoid_focal_loss (:31-59) and generalized_box_iou (util/box_ops.py:39-66), once per prediction set; this package's
// criterion already stacks the sets ([S, B, Q, C] logits, [S, B, Q, 4] boxes, matched on the device by csrc/lsap.hip) and ran
// the losses as ~60 ATen kernels forward and ~75 backward on 70-element tensors -- launch-bound, 3 us each on the critical
// path of the replayed step.  Same formulas, fp32, in the op chain's order where that is cheap (the focal sum is reduced in
// double instead of ATen's fp32 tree: 1e-6 relative); the backward is the closed form of what autograd derives, with
// autograd's conventions at the kinks (ties of min / max split the gradient in halves, clamp passes it at >= 0, sign(0) = 0).
//
// Matches: q_idx / t_idx [S, M] int64 -- set s matches query q_idx[s, k] of image image_of[k] with target t_idx[s, k] of the
// concatenated targets (matcher.forward_stacked_device); every (s, image, query) occurs at most once.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int kRowsPerBlock = 4;   // a wave per (set, image, query) row

__device__ __forceinline__ float wave_sum_f(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// index of the pair of row (s, b, q) in the set's M matches, or -1 (wave-uniform)
__device__ __forceinline__ int find_pair(const int64_t *__restrict__ q_idx, const int64_t *__restrict__ image_of, int M, int s, int b,
                                         int q, int lane)
{
    for (int k0 = 0; k0 < M; k0 += 64) {
        const int k = k0 + lane;
        const bool hit = k < M && image_of[k] == b && q_idx[(long long)s * M + k] == q;
        const unsigned long long m = __ballot(hit);
        if (m) return k0 + (int)__builtin_ctzll(m);
    }
    return -1;
}

struct Focal {
    float loss, dx;
};

template <bool GRAD>
__device__ __forceinline__ Focal focal(float x, float t, float alpha, float gamma)
{
#pragma clang fp contract(off)
    const float p = 1.f / (1.f + expf(-x));
    const float lsig = fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
    const float ce = (1.f - t) * x - lsig;
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float om = 1.f - pt;
    const float mod = gamma == 2.f ? om * om : powf(om, gamma);
    const float at = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
    Focal r;
    r.loss = at * (ce * mod);
    r.dx = 0.f;
    if (GRAD) {
        const float dmod = gamma == 2.f ? 2.f * om : gamma * powf(om, gamma - 1.f);
        const float dpt = (2.f * t - 1.f) * (p * (1.f - p));
        r.dx = at * ((p - t) * mod - ce * dmod * dpt);
    }
    return r;
}

__global__ __launch_bounds__(64 * kRowsPerBlock) void focal_fwd_kernel(const float *__restrict__ logits, const int64_t *__restrict__ q_idx,
                                                                     const int64_t *__restrict__ t_idx, const int64_t *__restrict__ image_of,
                                                                     const int64_t *__restrict__ labels, int B, int Q, int C, int M,
                                                                     float alpha, float gamma, double *__restrict__ partial)
{
    __shared__ double acc[kRowsPerBlock];
    const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * kRowsPerBlock + wave;   // b * Q + q
    float sum = 0.f;
    if (r < (long long)B * Q) {
        const int b = (int)(r / Q), q = (int)(r - (long long)b * Q);
        const int k = find_pair(q_idx, image_of, M, s, b, q, lane);
        const int label = k >= 0 ? (int)labels[t_idx[(long long)s * M + k]] : -1;
        const float *x = logits + ((long long)s * B * Q + r) * C;
        for (int c = lane; c < C; c += 64) sum += focal<false>(x[c], c == label ? 1.f : 0.f, alpha, gamma).loss;
        sum = wave_sum_f(sum);
    }
    if (lane == 0) acc[wave] = (double)sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < kRowsPerBlock; ++i) t += acc[i];
        partial[(long long)s * gridDim.x + blockIdx.x] = t;
    }
}

// GIoU of one matched pair and (GRAD) its gradient with respect to the predicted box (cx, cy, w, h)
struct PairLoss {
    float l1, giou_loss;
    float g[4];
};

template <bool GRAD>
__device__ __forceinline__ PairLoss pair_loss(const float4 sb, const float4 tb, float g_l1, float g_giou)
{
#pragma clang fp contract(off)
    const float eps = 1e-6f;
    PairLoss r;
    const float d[4] = {sb.x - tb.x, sb.y - tb.y, sb.z - tb.z, sb.w - tb.w};
    r.l1 = ((fabsf(d[0]) + fabsf(d[1])) + fabsf(d[2])) + fabsf(d[3]);
    const float x0 = sb.x - 0.5f * sb.z, y0 = sb.y - 0.5f * sb.w, x1 = sb.x + 0.5f * sb.z, y1 = sb.y + 0.5f * sb.w;
    const float X0 = tb.x - 0.5f * tb.z, Y0 = tb.y - 0.5f * tb.w, X1 = tb.x + 0.5f * tb.z, Y1 = tb.y + 0.5f * tb.w;
    const float area1 = (x1 - x0) * (y1 - y0), area2 = (X1 - X0) * (Y1 - Y0);
    const float iw_raw = fminf(x1, X1) - fmaxf(x0, X0), ih_raw = fminf(y1, Y1) - fmaxf(y0, Y0);
    const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
    const float inter = iw * ih;
    const float uni = area1 + area2 - inter;
    const float iou = inter / (uni + eps);
    const float ew_raw = fmaxf(x1, X1) - fminf(x0, X0), eh_raw = fmaxf(y1, Y1) - fminf(y0, Y0);
    const float ew = fmaxf(ew_raw, 0.f), eh = fmaxf(eh_raw, 0.f);
    const float area = ew * eh;
    const float giou = iou - (area - uni) / (area + eps);
    r.giou_loss = 1.f - giou;
    if (GRAD) {
        // d giou / d (inter, union, enclosing area); union = area1 + area2 - inter
        const float ue = uni + eps, ae = area + eps;
        const float d_inter = 1.f / ue, d_uni = -inter / (ue * ue) + 1.f / ae, d_area = -((ae - (area - uni)) / (ae * ae));
        const float G = -g_giou;                         // the loss is 1 - giou
        const float G_inter = G * (d_inter - d_uni), G_area1 = G * d_uni, G_area = G * d_area;
        const float G_iw = iw_raw >= 0.f ? G_inter * ih : 0.f, G_ih = ih_raw >= 0.f ? G_inter * iw : 0.f;
        const float G_ew = ew_raw >= 0.f ? G_area * eh : 0.f, G_eh = eh_raw >= 0.f ? G_area * ew : 0.f;
        auto lt = [](float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); };   // share of a in min(a, b)
        auto gt = [](float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); };   // share of a in max(a, b)
        const float gx1 = G_iw * lt(x1, X1) + G_ew * gt(x1, X1) + G_area1 * (y1 - y0);
        const float gx0 = -G_iw * gt(x0, X0) - G_ew * lt(x0, X0) - G_area1 * (y1 - y0);
        const float gy1 = G_ih * lt(y1, Y1) + G_eh * gt(y1, Y1) + G_area1 * (x1 - x0);
        const float gy0 = -G_ih * gt(y0, Y0) - G_eh * lt(y0, Y0) - G_area1 * (x1 - x0);
        auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
        r.g[0] = (gx0 + gx1) + g_l1 * sgn(d[0]);
        r.g[1] = (gy0 + gy1) + g_l1 * sgn(d[1]);
        r.g[2] = 0.5f * (gx1 - gx0) + g_l1 * sgn(d[2]);
        r.g[3] = 0.5f * (gy1 - gy0) + g_l1 * sgn(d[3]);
    }
    return r;
}

// one block: the pairs' L1 / GIoU losses, the focal partials' sums, the three per-set results
__global__ __launch_bounds__(256) void losses_finish_kernel(const float *__restrict__ boxes, const int64_t *__restrict__ q_idx,
                                                            const int64_t *__restrict__ t_idx, const int64_t *__restrict__ image_of,
                                                            const float *__restrict__ boxes_all, const float *__restrict__ num_boxes,
                                                            const double *__restrict__ partial, int nbx, int S, int B, int Q, int M,
                                                            float *__restrict__ pair_l1, float *__restrict__ pair_giou,
                                                            float *__restrict__ out)
{
#pragma clang fp contract(off)
    const float nb = *num_boxes;
    for (int i = threadIdx.x; i < S * M; i += 256) {
        const int s = i / M, k = i - s * M;
        const long long row = ((long long)s * B + image_of[k]) * Q + q_idx[i];
        const float4 sb = reinterpret_cast<const float4 *>(boxes)[row];
        const float4 tb = reinterpret_cast<const float4 *>(boxes_all)[t_idx[i]];
        const PairLoss r = pair_loss<false>(sb, tb, 0.f, 0.f);
        pair_l1[i] = r.l1;
        pair_giou[i] = r.giou_loss;
    }
    __syncthreads();
    // a wave per set and quantity, fixed order: lanes stride, then the wave tree
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int job = wave; job < 3 * S; job += 4) {
        const int s = job / 3, what = job - 3 * s;
        if (what == 0) {
            double t = 0.0;
            for (int i = lane; i < nbx; i += 64) t += partial[(long long)s * nbx + i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) out[s] = ((float)(t / (double)Q) / nb) * (float)Q;    // loss.mean(2).sum((1, 2)) / num_boxes * Q
        } else {
            const float *src = what == 1 ? pair_l1 : pair_giou;
            float t = 0.f;
            for (int i = lane; i < M; i += 64) t += src[(long long)s * M + i];
            t = wave_sum_f(t);
            if (lane == 0) out[what * S + s] = t / nb;
        }
    }
}

__global__ __launch_bounds__(64 * kRowsPerBlock) void losses_bwd_kernel(const float *__restrict__ logits, const float *__restrict__ boxes,
                                                                      const int64_t *__restrict__ q_idx, const int64_t *__restrict__ t_idx,
                                                                      const int64_t *__restrict__ image_of, const int64_t *__restrict__ labels,
                                                                      const float *__restrict__ boxes_all, const float *__restrict__ num_boxes,
                                                                      const float *__restrict__ g_out, int S, int B, int Q, int C, int M,
                                                                      float alpha, float gamma, float *__restrict__ g_logits,
                                                                      float *__restrict__ g_boxes)
{
#pragma clang fp contract(off)
    const int s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * kRowsPerBlock + wave;
    if (r >= (long long)B * Q) return;
    const int b = (int)(r / Q), q = (int)(r - (long long)b * Q);
    const float nb = *num_boxes;
    const int k = find_pair(q_idx, image_of, M, s, b, q, lane);
    const long long row = (long long)s * B * Q + r;
    if (g_logits) {
        const int label = k >= 0 ? (int)labels[t_idx[(long long)s * M + k]] : -1;
        const float gs = g_out[s] / nb;
        const float *x = logits + row * C;
        for (int c = lane; c < C; c += 64) g_logits[row * C + c] = gs * focal<true>(x[c], c == label ? 1.f : 0.f, alpha, gamma).dx;
    }
    if (g_boxes && lane < 4) {
        float g = 0.f;
        if (k >= 0) {
            const float4 sb = reinterpret_cast<const float4 *>(boxes)[row];
            const float4 tb = reinterpret_cast<const float4 *>(boxes_all)[t_idx[(long long)s * M + k]];
            const PairLoss p = pair_loss<true>(sb, tb, g_out[S + s] / nb, g_out[2 * S + s] / nb);
            g = lane == 0 ? p.g[0] : lane == 1 ? p.g[1] : lane == 2 ? p.g[2] : p.g[3];
        }
        g_boxes[row * 4 + lane] = g;
    }
}

}  // namespace

extern "C" size_t zira_stacked_losses_scratch_bytes(int S, int B, int Q, int M)
{
    if (S <= 0 || B <= 0 || Q <= 0 || M < 0) return 0;
    const size_t nbx = ((size_t)B * Q + kRowsPerBlock - 1) / kRowsPerBlock;
    return (size_t)S * nbx * sizeof(double) + 2 * (size_t)S * (M > 0 ? M : 1) * sizeof(float);
}

extern "C" int zira_stacked_losses_fwd_f32(const float *logits, const float *boxes, const int64_t *q_idx, const int64_t *t_idx,
                                           const int64_t *image_of, const int64_t *labels, const float *boxes_all,
                                           const float *num_boxes, int S, int B, int Q, int C, int M, float alpha, float gamma,
                                           void *scratch, float *out, void *stream)
{
    if (!logits || !boxes || !q_idx || !t_idx || !image_of || !labels || !boxes_all || !num_boxes || !scratch || !out)
        return (int)hipErrorInvalidValue;
    if (S <= 0 || B <= 0 || Q <= 0 || C <= 0 || M <= 0 || S > 65535) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)boxes | (uintptr_t)boxes_all) & 15) || ((uintptr_t)scratch & 7)) return (int)hipErrorInvalidValue;
    const unsigned nbx = (unsigned)(((long long)B * Q + kRowsPerBlock - 1) / kRowsPerBlock);
    double *partial = reinterpret_cast<double *>(scratch);
    float *pair_l1 = reinterpret_cast<float *>(partial + (size_t)S * nbx), *pair_giou = pair_l1 + (size_t)S * M;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(nbx, (unsigned)S), dim3(64 * kRowsPerBlock), 0, st, logits, q_idx, t_idx, image_of, labels, B,
                       Q, C, M, alpha, gamma, partial);
    hipLaunchKernelGGL(losses_finish_kernel, dim3(1), dim3(256), 0, st, boxes, q_idx, t_idx, image_of, boxes_all, num_boxes, partial,
                       (int)nbx, S, B, Q, M, pair_l1, pair_giou, out);
    return (int)hipGetLastError();
}

extern "C" int zira_stacked_losses_bwd_f32(const float *logits, const float *boxes, const int64_t *q_idx, const int64_t *t_idx,
                                           const int64_t *image_of, const int64_t *labels, const float *boxes_all,
                                           const float *num_boxes, const float *g_out, int S, int B, int Q, int C, int M, float alpha,
                                           float gamma, float *g_logits, float *g_boxes, void *stream)
{
    if (!logits || !boxes || !q_idx || !t_idx || !image_of || !labels || !boxes_all || !num_boxes || !g_out || (!g_logits && !g_boxes))
        return (int)hipErrorInvalidValue;
    if (S <= 0 || B <= 0 || Q <= 0 || C <= 0 || M <= 0 || S > 65535) return (int)hipErrorInvalidValue;
    if (((uintptr_t)boxes | (uintptr_t)boxes_all) & 15) return (int)hipErrorInvalidValue;
    const unsigned nbx = (unsigned)(((long long)B * Q + kRowsPerBlock - 1) / kRowsPerBlock);
    hipLaunchKernelGGL(losses_bwd_kernel, dim3(nbx, (unsigned)S), dim3(64 * kRowsPerBlock), 0, (hipStream_t)stream, logits, boxes, q_idx,
                       t_idx, image_of, labels, boxes_all, num_boxes, g_out, S, B, Q, C, M, alpha, gamma, g_logits, g_boxes);
    return (int)hipGetLastError();
}
