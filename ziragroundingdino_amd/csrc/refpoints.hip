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

}  // namespace

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
