// catlogits.hip -- token logits -> category logits (C ABI: zira_cat_logits_{fwd,bwd}_f32).
//
// Reference: recover_to_cls_logits (groundingdino/models/GroundingDINO/utils.py:312-320): for every image b and
// category c,   new[.., b, q, c] = max over the tokens t of category c of logits[.., b, q, t],
// `for_fill` in every other column.  The reference loops over images and categories with a boolean-mask gather
// per category; in PyTorch ops that is a masked_fill over a [rows, n_cat, n_tok] broadcast, a max, two more
// fills and a slice assignment per image (and their autograd nodes).  Here: one pass forward (a thread per
// output element, the category's tokens scanned from the row), one pass backward (the gradient of a max goes
// to its first arg-max token).  A category without tokens, or whose tokens are all -inf, reads `for_fill`.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

// rows = R * B * Q (R leading "set" dims flattened), image of a row = (row / Q) % B
__global__ __launch_bounds__(256) void cat_logits_fwd(const float *__restrict__ logits, const uint8_t *__restrict__ mask,
                                                      const int *__restrict__ n_cat, const int *__restrict__ n_tok,
                                                      long long rows, int B, int Q, int T, int Cmax, int Tmax,
                                                      float for_fill, float *__restrict__ out, int *__restrict__ arg)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * T) return;
    const long long row = idx / T;
    const int c = (int)(idx - row * T);
    const int b = (int)((row / Q) % B);
    float best = -INFINITY;
    int at = -1;
    if (c < n_cat[b]) {
        const float *lr = logits + row * T;
        const uint8_t *m = mask + ((size_t)b * Cmax + c) * Tmax;
        const int nt = n_tok[b];
        for (int t = 0; t < nt; ++t) {
            const float v = lr[t];
            if (m[t] && v > best) { best = v; at = t; }
        }
    }
    out[idx] = at >= 0 ? best : for_fill;
    if (c < Cmax) arg[row * Cmax + c] = at;
}

__global__ __launch_bounds__(256) void cat_logits_bwd(const float *__restrict__ grad_out, const int *__restrict__ arg,
                                                      const int *__restrict__ n_cat, long long rows, int B, int Q,
                                                      int T, int Cmax, float *__restrict__ grad_logits)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * T) return;
    const long long row = idx / T;
    const int t = (int)(idx - row * T);
    const int b = (int)((row / Q) % B);
    const int nc = n_cat[b];
    const int *a = arg + row * Cmax;
    const float *go = grad_out + row * T;
    float g = 0.f;
    for (int c = 0; c < nc; ++c)
        if (a[c] == t) g += go[c];
    grad_logits[idx] = g;
}

}  // namespace

extern "C" {

int zira_cat_logits_fwd_f32(const float *logits, const uint8_t *cat_token_mask, const int32_t *n_cat,
                            const int32_t *n_tok, long long rows, int B, int Q, int T, int Cmax, int Tmax,
                            float for_fill, float *out, int32_t *argmax, void *stream)
{
    if (rows < 0 || B <= 0 || Q <= 0 || T <= 0 || Cmax < 0 || Tmax < 0 || Cmax > T || Tmax > T ||
        rows % ((long long)B * Q) != 0 || rows * T >= (1ll << 40))
        return ZIRA_MSDA_EINVAL;
    if (rows == 0) return 0;
    if (!logits || !n_cat || !n_tok || !out || (Cmax && (!cat_token_mask || !argmax))) return ZIRA_MSDA_EINVAL;
    const long long total = rows * T;
    hipLaunchKernelGGL(cat_logits_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits,
                       cat_token_mask, n_cat, n_tok, rows, B, Q, T, Cmax, Tmax, for_fill, out, argmax);
    return (int)hipGetLastError();
}

int zira_cat_logits_bwd_f32(const float *grad_out, const int32_t *argmax, const int32_t *n_cat, long long rows, int B,
                            int Q, int T, int Cmax, float *grad_logits, void *stream)
{
    if (rows < 0 || B <= 0 || Q <= 0 || T <= 0 || Cmax < 0 || Cmax > T || rows % ((long long)B * Q) != 0)
        return ZIRA_MSDA_EINVAL;
    if (rows == 0) return 0;
    if (!grad_out || !n_cat || !grad_logits || (Cmax && !argmax)) return ZIRA_MSDA_EINVAL;
    const long long total = rows * T;
    hipLaunchKernelGGL(cat_logits_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_out,
                       argmax, n_cat, rows, B, Q, T, Cmax, grad_logits);
    return (int)hipGetLastError();
}

}  // extern "C"
