// winattn.hip -- (shifted-)window attention of the frozen Swin backbone, forward only (C ABI: zira_window_attn_f32).
//
// Reference: WindowAttention.forward and SwinTransformerBlock.forward
// (groundingdino/models/GroundingDINO/backbone/swin_transformer.py:128-160, :222-270): pad the token map to a
// multiple of the window, cyclic shift, cut into windows, per window and head softmax(q k^T * scale + relative
// position bias + shift mask) v, merge the windows, shift back, crop.  In PyTorch ops that is a pad, two rolls, two
// window permutes, the q/k/v and output transposes, a gathered-and-repeated bias tensor ([windows, heads, N, N]:
// 80 MB at the first stage of an 800x1333 image) and the attention itself -- a dozen kernels and copies per block.
// Here one kernel reads q, k, v straight from the qkv projection of the UN-partitioned token map and writes the
// attention output back in token order: window, shift, padding and crop are index arithmetic (a padded token's
// q/k/v are the projection's bias, since the reference pads the normalised map with zeros), the bias comes from a
// small [heads, N, N] table, the shift mask from the token's region id.
//
// One wavefront per (image, window, head): k and v of the window's N tokens are staged in LDS, lane r < N owns query
// row r and runs an online softmax over the N keys (every lane reads the same k_j / v_j: LDS broadcast), 32-wide
// accumulator in registers (head_dim = 32 in every Swin variant).  fp32 throughout.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

constexpr int kHD = 32;        // head dim
constexpr int kRowPad = 36;    // LDS row stride in floats (16-B aligned, 8 distinct bank groups across lanes)

// region of a coordinate of the rolled, padded map (swin_transformer.py:373-385): 0 | 1 | 2
__device__ __forceinline__ int region(int x, int Xp, int ws, int shift)
{
    return x < Xp - ws ? 0 : (x < Xp - shift ? 1 : 2);
}

template <int kChunk>   // 7: 7 x 7 windows, a window row of keys per softmax step; 0: any window up to 16 x 16, a key per step
__global__ __launch_bounds__(256) void window_attn_kernel(const float *__restrict__ qkv, const float *__restrict__ qkv_bias,
                                                          const float *__restrict__ bias_t, int B, int H, int W, int heads,
                                                          int ws, int shift, float scale, int nitems, int lds_per_wave,
                                                          float *__restrict__ out)
{
    extern __shared__ float winattn_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * (blockDim.x >> 6) + wave;
    if (item >= nitems) return;                       // wave-uniform
    const int N = ws * ws, C = heads * kHD;
    const int nwx = (W + ws - 1) / ws, nwy = (H + ws - 1) / ws, Hp = nwy * ws, Wp = nwx * ws;
    const int h = item % heads, win = item / heads;
    const int wx = win % nwx, wy = (win / nwx) % nwy, b = win / (nwx * nwy);
    float *ks = winattn_lds + (size_t)wave * lds_per_wave;   // [N][kRowPad]
    float *vs = ks + (size_t)N * kRowPad;                      // [N][kRowPad]
    int *reg = reinterpret_cast<int *>(vs + (size_t)N * kRowPad);   // [N] region id (shifted blocks)

    // token t of the window -> source token of the un-shifted, un-padded map (or -1: padding)
    auto source = [&](int t, int &rid) {
        const int hp = wy * ws + t / ws, wp = wx * ws + t % ws;     // rolled, padded coordinates
        rid = shift ? region(hp, Hp, ws, shift) * 3 + region(wp, Wp, ws, shift) : 0;
        int ho = hp + shift, wo = wp + shift;                       // x_rolled[hp] = x[(hp + shift) mod Hp]
        if (ho >= Hp) ho -= Hp;
        if (wo >= Wp) wo -= Wp;
        return (ho < H && wo < W) ? (b * H + ho) * W + wo : -1;
    };
    for (int t = lane; t < N; t += 64) {
        int rid;
        const int tok = source(t, rid);
        const float *kp = tok >= 0 ? qkv + (size_t)tok * 3 * C + C + h * kHD : qkv_bias + C + h * kHD;
        const float *vp = tok >= 0 ? qkv + (size_t)tok * 3 * C + 2 * C + h * kHD : qkv_bias + 2 * C + h * kHD;
#pragma unroll
        for (int c = 0; c < kHD; c += 4) {
            *reinterpret_cast<float4 *>(ks + t * kRowPad + c) = *reinterpret_cast<const float4 *>(kp + c);
            *reinterpret_cast<float4 *>(vs + t * kRowPad + c) = *reinterpret_cast<const float4 *>(vp + c);
        }
        reg[t] = rid;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float *bt = bias_t + (size_t)h * N * N;                   // [j][r]: coalesced over the lanes' rows
    for (int r = lane; r < N; r += 64) {
        int rid;
        const int tok = source(r, rid);
        const float *qp = tok >= 0 ? qkv + (size_t)tok * 3 * C + h * kHD : qkv_bias + h * kHD;
        float q[kHD], o[kHD];
#pragma unroll
        for (int c = 0; c < kHD; c += 4) {
            const float4 t4 = *reinterpret_cast<const float4 *>(qp + c);
            q[c] = t4.x * scale; q[c + 1] = t4.y * scale; q[c + 2] = t4.z * scale; q[c + 3] = t4.w * scale;
            o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        if (kChunk > 0) {
            // Online softmax a window row (kChunk keys) at a time: the accumulator is rescaled once per chunk instead of once
            // per key (o * corr + p * v is two instructions per output pair), and a dot product runs on two packed
            // accumulators instead of one 32-long dependent chain.
            for (int j0 = 0; j0 < N; j0 += kChunk) {
                float sc[kChunk > 0 ? kChunk : 1];
                float mc = m;
#pragma unroll
                for (int jj = 0; jj < kChunk; ++jj) {
                    const int j = j0 + jj;
                    const float *kj = ks + j * kRowPad;
                    float2 a0 = make_float2(0.f, 0.f), a1 = make_float2(0.f, 0.f);
#pragma unroll
                    for (int c = 0; c < kHD; c += 4) {
                        const float4 k4 = *reinterpret_cast<const float4 *>(kj + c);
                        a0.x = fmaf(q[c], k4.x, a0.x); a0.y = fmaf(q[c + 1], k4.y, a0.y);
                        a1.x = fmaf(q[c + 2], k4.z, a1.x); a1.y = fmaf(q[c + 3], k4.w, a1.y);
                    }
                    float sj = (a0.x + a0.y) + (a1.x + a1.y) + bt[(size_t)j * N + r];
                    if (shift && reg[j] != rid) sj -= 100.0f;
                    sc[jj] = sj;
                    mc = fmaxf(mc, sj);
                }
                const float corr = __expf(m - mc);
                l *= corr;
#pragma unroll
                for (int c = 0; c < kHD; ++c) o[c] *= corr;
#pragma unroll
                for (int jj = 0; jj < kChunk; ++jj) {
                    const float p = __expf(sc[jj] - mc);
                    l += p;
                    const float *vj = vs + (j0 + jj) * kRowPad;
#pragma unroll
                    for (int c = 0; c < kHD; c += 4) {
                        const float4 v4 = *reinterpret_cast<const float4 *>(vj + c);
                        o[c] = fmaf(p, v4.x, o[c]); o[c + 1] = fmaf(p, v4.y, o[c + 1]);
                        o[c + 2] = fmaf(p, v4.z, o[c + 2]); o[c + 3] = fmaf(p, v4.w, o[c + 3]);
                    }
                }
                m = mc;
            }
        } else
        for (int j = 0; j < N; ++j) {
            const float *kj = ks + j * kRowPad;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < kHD; c += 4) {
                const float4 k4 = *reinterpret_cast<const float4 *>(kj + c);
                s = fmaf(q[c], k4.x, s); s = fmaf(q[c + 1], k4.y, s); s = fmaf(q[c + 2], k4.z, s); s = fmaf(q[c + 3], k4.w, s);
            }
            s += bt[(size_t)j * N + r];
            if (shift && reg[j] != rid) s -= 100.0f;
            const float m_new = fmaxf(m, s);
            const float corr = __expf(m - m_new), p = __expf(s - m_new);
            l = fmaf(l, corr, p);
            const float *vj = vs + j * kRowPad;
#pragma unroll
            for (int c = 0; c < kHD; c += 4) {
                const float4 v4 = *reinterpret_cast<const float4 *>(vj + c);
                o[c] = fmaf(o[c], corr, p * v4.x); o[c + 1] = fmaf(o[c + 1], corr, p * v4.y);
                o[c + 2] = fmaf(o[c + 2], corr, p * v4.z); o[c + 3] = fmaf(o[c + 3], corr, p * v4.w);
            }
            m = m_new;
        }
        if (tok >= 0) {     // the reference crops the padding away again
            const float inv = 1.0f / l;
            float *op = out + (size_t)tok * C + h * kHD;
#pragma unroll
            for (int c = 0; c < kHD; c += 4)
                *reinterpret_cast<float4 *>(op + c) = make_float4(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
        }
    }
}

}  // namespace

extern "C" {

int zira_window_attn_f32(const float *qkv, const float *qkv_bias, const float *bias_t, int B, int H, int W, int heads,
                         int head_dim, int window, int shift, float scale, float *out, void *stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || head_dim != kHD || window <= 0 || window > 16 || shift < 0 ||
        shift >= window)
        return ZIRA_MSDA_EINVAL;
    if (!qkv || !qkv_bias || !bias_t || !out) return ZIRA_MSDA_EINVAL;
    const int N = window * window;
    const int nwx = (W + window - 1) / window, nwy = (H + window - 1) / window;
    const long long nitems = (long long)B * nwx * nwy * heads;
    if (nitems >= (1ll << 31) || (long long)B * H * W * 3 * heads * kHD >= (1ll << 40)) return ZIRA_MSDA_EINVAL;
    const int lds_per_wave = 2 * N * kRowPad + ((N + 3) & ~3);                     // floats
    int waves = (int)((60 * 1024) / (lds_per_wave * 4));
    if (waves > 4) waves = 4;
    if (waves < 1) return ZIRA_MSDA_EINVAL;
    const unsigned blocks = (unsigned)((nitems + waves - 1) / waves);
    if (window == 7)
        hipLaunchKernelGGL(window_attn_kernel<7>, dim3(blocks), dim3(waves * 64), (size_t)waves * lds_per_wave * 4,
                           (hipStream_t)stream, qkv, qkv_bias, bias_t, B, H, W, heads, window, shift, scale, (int)nitems,
                           lds_per_wave, out);
    else
        hipLaunchKernelGGL(window_attn_kernel<0>, dim3(blocks), dim3(waves * 64), (size_t)waves * lds_per_wave * 4,
                           (hipStream_t)stream, qkv, qkv_bias, bias_t, B, H, W, heads, window, shift, scale, (int)nitems,
                           lds_per_wave, out);
    return (int)hipGetLastError();
}

}  // extern "C"
