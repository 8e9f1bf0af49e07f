// sampling.hip -- what the MSDA module does between its query projection and the native op, in one launch each way
// (C ABI: zira_msda_sampling_fwd_f32 / zira_msda_sampling_bwd_f32).
//
// Reference: MultiScaleDeformableAttention.forward, groundingdino/models/GroundingDINO/ms_deform_attn.py:290-325:
//   sampling_offsets(query).view(B, Q, M, L, P, 2); attention_weights(query).view(B, Q, M, L * P).softmax(-1);
//   reference_points [.., 2]:  loc = ref[:, :, None, :, None, :] + offsets / (W_l, H_l)
//   reference_points [.., 4]:  loc = ref_xy + offsets / P * ref_wh * 0.5
// As PyTorch ops that is a softmax, a divide and an add over [B, Q, M, L, P(, 2)] (45 MB at the encoder shape) forward, and a
// softmax backward, a multiply and a concatenation of the two halves of the projection's gradient backward: ~110 us
// per encoder layer, ~10 launch-bound kernels per decoder layer.  Here a lane per (query, head, level, point) reads its
// logit and its offset pair from the row of the ONE projection GEMM (offsets first, then logits: the layout
// ms_deform_attn.py's fused projection produces), the L * P lanes of a (query, head) find the softmax with DPP / permute
// reductions, and the backward writes the whole gradient row (both halves) itself.
//
// The location arithmetic is the reference's, operation by operation (separately rounded divide, multiply, add), so the
// pixel a sample falls into is the one the reference picks; the softmax is exp(x - max) / sum in fp32 (the sum order is not
// PyTorch's: last-bit differences).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

// reductions over the LP consecutive lanes of a (query, head) group (LP a power of two <= 64)
__device__ __forceinline__ float group_max(float x, int LP)
{
    for (int d = 1; d < LP; d <<= 1) x = fmaxf(x, __shfl_xor(x, d));
    return x;
}
__device__ __forceinline__ float group_sum(float x, int LP)
{
    for (int d = 1; d < LP; d <<= 1) x += __shfl_xor(x, d);
    return x;
}

struct SamplingDims {
    unsigned items;          // N * M * LP (< 2^31)
    unsigned M, L, P, LP, R, ld;
    unsigned lpsh, psh;      // log2(LP), log2(P): L * P is a power of two, so both are
    unsigned mmul, mshift;   // x / M == (x * mmul) >> mshift for x < 2^31
};
struct SamplingItem {
    unsigned n, m, l, lp;
    bool ok;
    unsigned i;
};
__device__ __forceinline__ SamplingItem sampling_item(const SamplingDims &D)
{
    SamplingItem t;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    t.ok = i < D.items;
    t.i = t.ok ? i : 0u;
    t.lp = t.i & (D.LP - 1);
    const unsigned nm = t.i >> D.lpsh;
    t.n = (unsigned)(((unsigned long long)nm * D.mmul) >> D.mshift);
    t.m = nm - t.n * D.M;
    t.l = t.lp >> D.psh;
    return t;
}

// item i = ((n * M + m) * LP + lp): logit at proj[n * ld + M * LP * 2 + m * LP + lp], offsets at proj[n * ld + (m * LP + lp) * 2]
__global__ __launch_bounds__(256) void sampling_fwd(const float *__restrict__ proj, const float *__restrict__ ref,
                                                    const int64_t *__restrict__ shapes, SamplingDims D,
                                                    float *__restrict__ loc, float *__restrict__ attn)
{
#pragma clang fp contract(off)
    const SamplingItem t = sampling_item(D);
    const bool ok = t.ok;
    const unsigned i = t.i, lp = t.lp, m = t.m, l = t.l;
    const size_t n = t.n;
    const float *row = proj + n * D.ld;
    const float x = row[D.M * D.LP * 2 + m * D.LP + lp];
    const float2 off = *reinterpret_cast<const float2 *>(row + (size_t)(m * D.LP + lp) * 2);
    const float *rp = ref + (n * D.L + l) * D.R;
    float2 o;
    if (D.R == 2) {
        const float Wl = (float)shapes[2 * l + 1], Hl = (float)shapes[2 * l];
        o.x = __fadd_rn(rp[0], __fdiv_rn(off.x, Wl));
        o.y = __fadd_rn(rp[1], __fdiv_rn(off.y, Hl));
    } else {
        const float Pf = (float)D.P;
        o.x = __fadd_rn(rp[0], __fmul_rn(__fmul_rn(__fdiv_rn(off.x, Pf), rp[2]), 0.5f));
        o.y = __fadd_rn(rp[1], __fmul_rn(__fmul_rn(__fdiv_rn(off.y, Pf), rp[3]), 0.5f));
    }
    const float mx = group_max(x, (int)D.LP);
    const float e = expf(x - mx);
    const float a = e / group_sum(e, (int)D.LP);
    if (ok) {
        *reinterpret_cast<float2 *>(loc + (size_t)i * 2) = o;
        attn[i] = a;
    }
}

__global__ __launch_bounds__(256) void sampling_bwd(const float *__restrict__ grad_loc, const float *__restrict__ grad_attn,
                                                    const float *__restrict__ attn, const float *__restrict__ ref,
                                                    const int64_t *__restrict__ shapes, SamplingDims D,
                                                    float *__restrict__ grad_proj)
{
#pragma clang fp contract(off)
    const SamplingItem t = sampling_item(D);
    const bool ok = t.ok;
    const unsigned i = t.i, lp = t.lp, m = t.m, l = t.l;
    const size_t n = t.n;
    const float a = attn[i], ga = grad_attn[i];
    const float2 gl = *reinterpret_cast<const float2 *>(grad_loc + (size_t)i * 2);
    const float dot = group_sum(a * ga, (int)D.LP);           // softmax backward: a * (g - <a, g>)
    const float gx = a * (ga - dot);
    float2 go;
    if (D.R == 2) {
        const float Wl = (float)shapes[2 * l + 1], Hl = (float)shapes[2 * l];
        go.x = __fdiv_rn(gl.x, Wl);
        go.y = __fdiv_rn(gl.y, Hl);
    } else {
        const float *rp = ref + (n * D.L + l) * D.R;
        const float Pf = (float)D.P;
        go.x = __fdiv_rn(__fmul_rn(__fmul_rn(gl.x, 0.5f), rp[2]), Pf);
        go.y = __fdiv_rn(__fmul_rn(__fmul_rn(gl.y, 0.5f), rp[3]), Pf);
    }
    if (ok) {
        float *row = grad_proj + n * D.ld;
        row[D.M * D.LP * 2 + m * D.LP + lp] = gx;
        *reinterpret_cast<float2 *>(row + (size_t)(m * D.LP + lp) * 2) = go;
    }
}

bool sampling_dims(long long N, int M, int L, int P, int R, int ld, SamplingDims &D)
{
    const int LP = L * P;
    if (N < 0 || M <= 0 || L <= 0 || P <= 0 || (R != 2 && R != 4) || LP > 64 || (LP & (LP - 1)) || ld < M * LP * 3 || (ld & 1)) return false;
    if (N * M * LP >= (1ll << 31)) return false;
    D.items = (unsigned)(N * M * LP);
    D.M = (unsigned)M; D.L = (unsigned)L; D.P = (unsigned)P; D.LP = (unsigned)LP; D.R = (unsigned)R; D.ld = (unsigned)ld;
    D.lpsh = 0;
    while ((1 << D.lpsh) < LP) ++D.lpsh;
    D.psh = 0;
    while ((1 << D.psh) < P) ++D.psh;
    unsigned sh = 0;
    while ((1ull << sh) < (unsigned)M) ++sh;
    D.mshift = 31 + sh;
    D.mmul = (unsigned)(((1ull << (31 + sh)) / (unsigned)M) + 1);
    return true;
}

}  // namespace

extern "C" {

int zira_msda_sampling_fwd_f32(const float *proj, int ld, const float *ref, int R, const int64_t *shapes, long long N, int M,
                               int L, int P, float *loc, float *attn, void *stream)
{
    SamplingDims D;
    if (!proj || !ref || !shapes || !loc || !attn || !sampling_dims(N, M, L, P, R, ld, D) || ((uintptr_t)proj & 7) ||
        ((uintptr_t)loc & 7))
        return ZIRA_MSDA_EINVAL;
    if (D.items == 0) return 0;
    hipLaunchKernelGGL(sampling_fwd, dim3((unsigned)((D.items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, proj, ref, shapes, D,
                       loc, attn);
    return (int)hipGetLastError();
}

int zira_msda_sampling_bwd_f32(const float *grad_loc, const float *grad_attn, const float *attn, const float *ref, int R,
                               const int64_t *shapes, long long N, int M, int L, int P, float *grad_proj, int ld, void *stream)
{
    SamplingDims D;
    if (!grad_loc || !grad_attn || !attn || !ref || !shapes || !grad_proj || !sampling_dims(N, M, L, P, R, ld, D) ||
        ((uintptr_t)grad_proj & 7) || ((uintptr_t)grad_loc & 7))
        return ZIRA_MSDA_EINVAL;
    if (D.items == 0) return 0;
    hipLaunchKernelGGL(sampling_bwd, dim3((unsigned)((D.items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_loc, grad_attn,
                       attn, ref, shapes, D, grad_proj);
    return (int)hipGetLastError();
}

}  // extern "C"
