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


// ------------------------------------------------------------------------------------------
// 12 x 12 windows (the 384-pixel Swin-B / L variants: 144 tokens per window) on the matrix cores
// ------------------------------------------------------------------------------------------
// One block per (image, window, head), one wave per tile of 32 queries (five for 144 tokens).  The block stages the
// window's K and V rows (padding tokens: the projection's bias) and the tokens' shift-mask regions in LDS once; every
// wave then walks the five key tiles in the MFMA form of csrc/attn.hip: S^T = K Q^T on `v_mfma_f32_32x32x2_f32` (exact
// fp32 products) with a query's scores in ONE lane's accumulator registers -- bias, mask, running maximum and sum are
// per-lane scalars -- and O^T += V^T P^T takes those registers directly as its operand.  Input / output either fp32 or
// bf16 (configs[3] runs the qkv projection under bf16 autocast; the arithmetic here is fp32 either way).
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned rowmap32(unsigned reg, unsigned hh) { return (reg & 3u) + 8u * (reg >> 2) + 4u * hh; }

__device__ __forceinline__ float bf16_to_f32(unsigned short x) { return __uint_as_float((unsigned)x << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float x)   // round to nearest even (torch's conversion)
{
    unsigned u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// 16 consecutive elements as floats
__device__ __forceinline__ void load16(const float *p, float (&x)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = reinterpret_cast<const float4 *>(p)[i];
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
}
__device__ __forceinline__ void load16(const unsigned short *p, float (&x)[16])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[8 * i + 2 * k] = __uint_as_float(w[k] << 16);
            x[8 * i + 2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
        }
    }
}
// the projection's bias as a padding token's q / k / v: rounded to the tensor's precision, as the projection would
template <typename T>
__device__ __forceinline__ void load16_bias(const float *p, float (&x)[16])
{
    load16(p, x);
    if (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = bf16_to_f32(f32_to_bf16(x[i]));
    }
}

template <int WS, typename T>
__global__ __launch_bounds__(64 * ((WS * WS + 31) / 32)) void window_attn_mfma(
    const T *__restrict__ qkv, const float *__restrict__ qkv_bias, const float *__restrict__ bias_t, int B, int H, int W,
    int heads, int shift, float scale, T *__restrict__ out)
{
    constexpr int N = WS * WS, NT = (N + 31) / 32, NTHR = 64 * NT, KP = 36;   // KP: LDS row stride in floats
    __shared__ float ks[N][KP], vs[N][KP];
    __shared__ int toks[NT * 32], rids[NT * 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned r = lane & 31, hh = lane >> 5;
    const int C = heads * kHD;
    const int nwx = (W + WS - 1) / WS, nwy = (H + WS - 1) / WS, Hp = nwy * WS, Wp = nwx * WS;
    const int item = blockIdx.x, h = item % heads, win = item / heads;
    const int wx = win % nwx, wy = (win / nwx) % nwy, b = win / (nwx * nwy);

    // stage: source token (or -1: padding) and region of every token of the window, its K and V rows
    for (int t = tid; t < NT * 32; t += NTHR) {
        const int tc = t < N ? t : N - 1;
        const int hp = wy * WS + tc / WS, wp = wx * WS + tc % WS;     // rolled, padded coordinates
        int ho = hp + shift, wo = wp + shift;                         // x_rolled[hp] = x[(hp + shift) mod Hp]
        if (ho >= Hp) ho -= Hp;
        if (wo >= Wp) wo -= Wp;
        toks[t] = (ho < H && wo < W) ? (b * H + ho) * W + wo : -1;
        rids[t] = shift ? region(hp, Hp, WS, shift) * 3 + region(wp, Wp, WS, shift) : 0;
    }
    __syncthreads();
    for (int i = tid; i < N * 4; i += NTHR) {       // a token's K or V row half: 16 floats
        const int t = i >> 2, part = i & 3, which = part >> 1, half = part & 1;   // which: 0 K, 1 V
        const int tok = toks[t];
        float x[16];
        const size_t off = (size_t)(1 + which) * C + h * kHD + 16 * half;
        if (tok >= 0) load16(qkv + (size_t)tok * 3 * C + off, x);
        else load16_bias<T>(qkv_bias + off, x);
        float *dst = (which ? vs[t] : ks[t]) + 16 * half;
#pragma unroll
        for (int c = 0; c < 16; c += 4) *reinterpret_cast<float4 *>(dst + c) = make_float4(x[c], x[c + 1], x[c + 2], x[c + 3]);
    }
    // this wave's 32 queries
    const int tq = wave * 32 + (int)r, tqc = tq < N ? tq : N - 1;
    const int tokq = toks[tqc], ridq = rids[tqc];
    float bq[16];
    {
        const size_t off = (size_t)h * kHD + 16 * hh;
        if (tokq >= 0) load16(qkv + (size_t)tokq * 3 * C + off, bq);
        else load16_bias<T>(qkv_bias + off, bq);
#pragma unroll
        for (int t = 0; t < 16; ++t) bq[t] *= scale;
    }
    __syncthreads();

    const float *bt = bias_t + (size_t)h * N * N;   // [key j][query i]
    float m = -INFINITY, lsum = 0.f;
    f32x16 acc_o;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[i] = 0.f;
    for (int kt = 0; kt < NT; ++kt) {
        const int k0 = kt * 32;
        float ka[16], va[16];
        {
            const int kr = k0 + (int)r < N ? k0 + (int)r : N - 1;
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                const float4 v4 = *reinterpret_cast<const float4 *>(&ks[kr][16 * hh + c]);
                ka[c] = v4.x; ka[c + 1] = v4.y; ka[c + 2] = v4.z; ka[c + 3] = v4.w;
            }
#pragma unroll
            for (unsigned t = 0; t < 16; ++t) {
                const int vr = k0 + (int)rowmap32(t, hh);
                va[t] = vs[vr < N ? vr : N - 1][r];
            }
        }
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], bq[t], s, 0, 0, 0);
        // s[reg] = S^T[key k0 + rowmap(reg, hh)][query tq]
        float mt = -INFINITY;
#pragma unroll
        for (unsigned reg = 0; reg < 16; ++reg) {
            const int kidx = k0 + (int)rowmap32(reg, hh);
            float x = s[reg];
            if (kidx >= N) {
                x = -INFINITY;
            } else {
                x += bt[(size_t)kidx * N + tqc];
                if (shift && rids[kidx] != ridq) x -= 100.0f;
            }
            s[reg] = x;
            mt = fmaxf(mt, x);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m, mt);
        const float alpha = __expf(m - m_new);   // (m = -inf: 0; every tile holds real keys, so m_new is finite)
        float ps = 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float p = __expf(s[reg] - m_new);
            s[reg] = p;
            ps += p;
        }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[i] *= alpha;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc_o = __builtin_amdgcn_mfma_f32_32x32x2f32(va[t], s[t], acc_o, 0, 0, 0);
        // acc_o[reg] = O^T[feature rowmap(reg, hh)][query tq]
        m = m_new;
    }
    lsum += __shfl_xor(lsum, 32);
    if (tq < N && tokq >= 0) {     // (the reference crops the padding away again)
        const float inv = 1.0f / lsum;
        T *o = out + (size_t)tokq * C + h * kHD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v4 = make_float4(acc_o[4 * g] * inv, acc_o[4 * g + 1] * inv, acc_o[4 * g + 2] * inv, acc_o[4 * g + 3] * inv);
            if (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(o) + 8 * g + 4 * hh) = v4;
            } else {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16(v4.x) | ((unsigned)f32_to_bf16(v4.y) << 16);
                pk.y = (unsigned)f32_to_bf16(v4.z) | ((unsigned)f32_to_bf16(v4.w) << 16);
                *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(o) + 8 * g + 4 * hh) = pk;
            }
        }
    }
}

template <typename T>
int launch_window_attn(const T *qkv, const float *qkv_bias, const float *bias_t, int B, int H, int W, int heads, int window,
                       int shift, float scale, T *out, hipStream_t st)
{
    const int nwx = (W + window - 1) / window, nwy = (H + window - 1) / window;
    const long long nitems = (long long)B * nwx * nwy * heads;
    if (nitems >= (1ll << 31) || (long long)B * H * W * 3 * heads * kHD >= (1ll << 40)) return ZIRA_MSDA_EINVAL;
    if (window == 7)          // 49-token windows on the matrix cores: two waves of 32 queries per (image, window, head)
        hipLaunchKernelGGL((window_attn_mfma<7, T>), dim3((unsigned)nitems), dim3(64 * 2), 0, st, qkv, qkv_bias, bias_t, B, H, W, heads,
                           shift, scale, out);
    else if (window == 12)
        hipLaunchKernelGGL((window_attn_mfma<12, T>), dim3((unsigned)nitems), dim3(64 * 5), 0, st, qkv, qkv_bias, bias_t, B, H, W, heads,
                           shift, scale, out);
    else
        return ZIRA_MSDA_EINVAL;
    return (int)hipGetLastError();
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
#ifndef ZIRA_WINATTN_MFMA7
#define ZIRA_WINATTN_MFMA7 1   // 7 x 7 windows on the matrix-core kernel as well (0: the one-wave-per-window vector kernel)
#endif
    if (window == 12 || (window == 7 && ZIRA_WINATTN_MFMA7))   // the MFMA kernel
        return launch_window_attn<float>(qkv, qkv_bias, bias_t, B, H, W, heads, window, shift, scale, out, (hipStream_t)stream);
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

int zira_window_attn_bf16(const void *qkv, const float *qkv_bias, const float *bias_t, int B, int H, int W, int heads,
                          int head_dim, int window, int shift, float scale, void *out, void *stream)
{
    if (B <= 0 || H <= 0 || W <= 0 || heads <= 0 || head_dim != kHD || window != 12 || shift < 0 || shift >= window)
        return ZIRA_MSDA_EINVAL;
    if (!qkv || !qkv_bias || !bias_t || !out) return ZIRA_MSDA_EINVAL;
    return launch_window_attn<unsigned short>(reinterpret_cast<const unsigned short *>(qkv), qkv_bias, bias_t, B, H, W, heads,
                                              window, shift, scale, reinterpret_cast<unsigned short *>(out), (hipStream_t)stream);
}

}  // extern "C"
