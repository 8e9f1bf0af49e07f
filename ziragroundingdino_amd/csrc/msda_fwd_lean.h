// msda_fwd_lean.h -- building blocks of the "lean" forward of multi-scale deformable attention (D = 16 / 32 / 64) for
// gfx950, shared by csrc/msda.hip (the forward kernel, the atomic backward) and csrc/msda_tiles.hip (the forward fused
// with the sparse backward's plan).  Arithmetic to match: reference ms_deform_im2col_cuda.cuh:237-299 + :33-84.
// Layout notes are in csrc/msda.hip ("lean" path).  Everything here is internal linkage (one copy per TU).
#ifndef ZIRA_MSDA_FWD_LEAN_H_
#define ZIRA_MSDA_FWD_LEAN_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef ZIRA_K1_GATHERS
#define ZIRA_K1_GATHERS 8   // row gathers a wave keeps in flight in chunk_dots
#endif

namespace {

constexpr unsigned kLeanWavesPerBlock = 4;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}

struct ItemId {
    unsigned item;  // flat (b, q, m)
    unsigned b, m;
    bool ok;
};

// n / d for n < 2^31 with a host-prepared multiplier (no hardware integer divide on the GPU:
// a plain `/` costs ~30 scalar instructions per wave, and the scalar unit is shared by the CU)
struct FastDiv {
    unsigned mul, shift, d;
};
__device__ __forceinline__ unsigned fast_div(unsigned n, FastDiv f)
{
    return (unsigned)(((unsigned long long)n * f.mul) >> f.shift);
}

// head-major placement (see head_major_item), wave-uniform / scalar, 32-bit
__device__ __forceinline__ ItemId lean_item(unsigned nitems, unsigned per, FastDiv Q, FastDiv M,
                                            unsigned bid = blockIdx.x, unsigned waves_per_block = kLeanWavesPerBlock, unsigned wave_offset = 0)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + wave_offset;
    const unsigned xcd = bid & 7, idx = bid >> 3;
    const unsigned t0 = xcd * per;
    const unsigned t = t0 + idx * waves_per_block + wave;
    const unsigned t1 = (t0 + per < nitems) ? t0 + per : nitems;
    ItemId r;
    r.ok = t < t1;
    const unsigned g = fast_div(t, Q), q = t - g * Q.d;
    r.b = fast_div(g, M);
    r.m = g - r.b * M.d;
    r.item = (r.b * Q.d + q) * M.d + r.m;
    return r;
}

struct Entry {
    float w;        // bilinear weight x attention weight, 0 for a corner that contributes nothing
    unsigned offb;  // BYTE offset of the corner's value row inside the batch element (0 if unused)
    // backward only
    float wb, cx, cy, a, Wf, Hf;
    unsigned lvl, pix, hw;  // level, pixel index inside the level, pixels in the level
    bool inb;
};

template <bool kNeedGrad>
__device__ __forceinline__ Entry entry_setup(const int64_t *__restrict__ shapes,
                                             const int64_t *__restrict__ start,
                                             const float *__restrict__ loc_i,
                                             const float *__restrict__ att_i, unsigned s,
                                             unsigned c, unsigned LP, float invP, unsigned M,
                                             unsigned D, unsigned m)
{
#pragma clang fp contract(off)
    Entry k;
    const bool act = s < LP;
    const unsigned sc = act ? s : 0u;
    const unsigned l = (unsigned)(((float)sc + 0.5f) * invP);  // == sc / P
    // low dwords of the int64 level table (sizes are < 2^31)
    const int2 hw = make_int2(reinterpret_cast<const int *>(shapes)[4 * l],
                              reinterpret_cast<const int *>(shapes)[4 * l + 2]);
    const int st = reinterpret_cast<const int *>(start)[2 * l];
    const float2 xy = *reinterpret_cast<const float2 *>(loc_i + 2 * sc);
    const float a = att_i[sc];
    const float Hf = (float)hw.x, Wf = (float)hw.y;
    const float h_im = xy.y * Hf - 0.5f;
    const float w_im = xy.x * Wf - 0.5f;
    const bool valid = act && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const float lh = h_im - hf, lw = w_im - wf;
    const int dy = (int)(c >> 1), dx = (int)(c & 1);
    const int y = (int)hf + dy, x = (int)wf + dx;
    const float wy = dy ? lh : 1.f - lh;
    const float wx = dx ? lw : 1.f - lw;
    k.inb = valid && y >= 0 && y < hw.x && x >= 0 && x < hw.y;
    k.wb = k.inb ? wy * wx : 0.f;
    k.w = k.wb * a;
    k.offb = k.inb ? ((unsigned)(st + y * hw.y + x) * M + m) * (D * 4u) : 0u;
    if (kNeedGrad) {
        k.cx = k.inb ? (dx ? wy : -wy) : 0.f;
        k.cy = k.inb ? (dy ? wx : -wx) : 0.f;
        k.a = valid ? a : 0.f;
        k.Wf = Wf; k.Hf = Hf;
        k.lvl = l;
        k.pix = (unsigned)(y * hw.y + x);
        k.hw = (unsigned)(hw.x * hw.y);
    }
    return k;
}

__device__ __forceinline__ float4 load_row16(const float *__restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + byte_off);
}

inline FastDiv make_fast_div(unsigned d)
{
    FastDiv f;
    f.d = d;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = 31 + s;
    f.mul = (unsigned)(((1ull << (31 + s)) / d) + 1);
    return f;
}

// The forward of one (b, q, m) item by one wave (see the layout notes above); `id.ok` is wave-uniform.
template <int CQR>
__device__ __forceinline__ void fwd_lean_item(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                              const int64_t *__restrict__ start, const float *__restrict__ loc,
                                              const float *__restrict__ attn, unsigned S, unsigned M, unsigned LP,
                                              float invP, const ItemId id, float *__restrict__ out)
{
    constexpr unsigned D = 16 * CQR;
    constexpr unsigned SLOTS = 16 / CQR;  // value rows per gather instruction
    constexpr unsigned NI = 64 / SLOTS;   // gather instructions per 64-entry chunk
    const unsigned lane = threadIdx.x & 63;
    const float *vb = value + (size_t)id.b * S * M * D;  // uniform
    const float *loc_i = loc + (size_t)id.item * LP * 2;
    const float *att_i = attn + (size_t)id.item * LP;

    const unsigned R = lane >> 4;
    const unsigned slot = (lane & 15) / CQR;
    const unsigned cq = R * CQR + (lane & (CQR - 1));
    const int bp = (int)(slot * 4);  // ds_bpermute byte address of source lane `slot`
    const unsigned lane_off = cq * 16;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned s0 = 0; s0 < LP; s0 += 16) {
        const Entry k = entry_setup<false>(shapes, start, loc_i, att_i, s0 + (lane >> 2),
                                           lane & 3, LP, invP, M, D, id.m);
        const int offb_i = (int)k.offb, w_i = __float_as_int(k.w);
#pragma unroll
        for (unsigned j = 0; j < NI; ++j) {
            const int a = bp + (int)(j * SLOTS * 4);
            const unsigned oj = (unsigned)__builtin_amdgcn_ds_bpermute(a, offb_i);
            const float wj = __int_as_float(__builtin_amdgcn_ds_bpermute(a, w_i));
            const float4 v = load_row16(vb, oj + lane_off);
            acc.x = fmaf(wj, v.x, acc.x);
            acc.y = fmaf(wj, v.y, acc.y);
            acc.z = fmaf(wj, v.z, acc.z);
            acc.w = fmaf(wj, v.w, acc.w);
        }
    }
    // reduce over the SLOTS lanes (same DPP row) that hold the same channel quad
    if (CQR <= 1) { acc.x = dpp_add<0x121>(acc.x); acc.y = dpp_add<0x121>(acc.y);
                    acc.z = dpp_add<0x121>(acc.z); acc.w = dpp_add<0x121>(acc.w); }
    if (CQR <= 2) { acc.x = dpp_add<0x122>(acc.x); acc.y = dpp_add<0x122>(acc.y);
                    acc.z = dpp_add<0x122>(acc.z); acc.w = dpp_add<0x122>(acc.w); }
    acc.x = dpp_add<0x124>(acc.x); acc.y = dpp_add<0x124>(acc.y);
    acc.z = dpp_add<0x124>(acc.z); acc.w = dpp_add<0x124>(acc.w);
    acc.x = dpp_add<0x128>(acc.x); acc.y = dpp_add<0x128>(acc.y);
    acc.z = dpp_add<0x128>(acc.z); acc.w = dpp_add<0x128>(acc.w);
    if (slot == 0) *reinterpret_cast<float4 *>(out + (size_t)id.item * D + cq * 4) = acc;
}


// ---- the gather half of the backward: grad_sampling_loc and grad_attn_weight of one (b, q, m) item by one wave ----
// (reference ms_deform_im2col_cuda.cuh:123-158; the four corner rows of a sample are gathered as in the forward and dotted
//  with the query's grad_out row)
template <int CQ>
__device__ __forceinline__ float sum_over_row_lanes(float x)
{
    x = dpp_add<0xB1>(x);                 // quad_perm:[1,0,3,2]   (xor 1)
    x = dpp_add<0x4E>(x);                 // quad_perm:[2,3,0,1]   (xor 2)
    if (CQ >= 8) x = dpp_add<0x141>(x);   // row_half_mirror       (xor 4 on quad sums)
    if (CQ >= 16) x = dpp_add<0x140>(x);  // row_mirror            (xor 8 on octet sums)
    return x;
}

// <grad_out row, value row> for the 64 entries of a chunk; entry e's result lands in lane e.
template <int CQ>
__device__ __forceinline__ float chunk_dots(const float *__restrict__ vb, const Entry &k,
                                            float4 g4, unsigned lane)
{
    constexpr unsigned SLOTS = 64 / CQ, NI = CQ;
    const unsigned slot = lane / CQ, cq = lane % CQ;
    const int bp = (int)(slot * 4);
    const int back = (int)((lane % SLOTS) * CQ * 4);
    const int offb_i = (int)k.offb;
    float mine = 0.f;
    // All NI row gathers are issued before the first dot product.  Written as two loops with a
    // scheduling barrier in between: left to itself the compiler put `s_waitcnt vmcnt(0)` behind
    // every load (NI dependent round trips per chunk, K1 at twice the forward's time).
    constexpr unsigned G = NI < ZIRA_K1_GATHERS ? NI : ZIRA_K1_GATHERS;  // gathers in flight
#pragma unroll
    for (unsigned j0 = 0; j0 < NI; j0 += G) {
        float4 v[G];
#pragma unroll
        for (unsigned j = 0; j < G; ++j) {
            const unsigned oj =
                (unsigned)__builtin_amdgcn_ds_bpermute(bp + (int)((j0 + j) * SLOTS * 4), offb_i);
            v[j] = load_row16(vb, oj + cq * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (unsigned j = 0; j < G; ++j) {
            float d = v[j].x * g4.x;
            d = fmaf(v[j].y, g4.y, d);
            d = fmaf(v[j].z, g4.z, d);
            d = fmaf(v[j].w, g4.w, d);
            d = sum_over_row_lanes<CQ>(d);
            const float t = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(d)));
            if (lane / SLOTS == j0 + j) mine = t;
        }
    }
    return k.inb ? mine : 0.f;
}

// grad_sampling_loc / grad_attn_weight of the chunk's 16 samples from the per-entry dots
__device__ __forceinline__ void store_sample_grads(const Entry &k, float d, unsigned lane,
                                                   unsigned s, unsigned LP,
                                                   float *__restrict__ gl_i,
                                                   float *__restrict__ ga_i)
{
    float ga = k.wb * d, gx = k.cx * d, gy = k.cy * d;
    ga = dpp_add<0xB1>(ga); gx = dpp_add<0xB1>(gx); gy = dpp_add<0xB1>(gy);
    ga = dpp_add<0x4E>(ga); gx = dpp_add<0x4E>(gx); gy = dpp_add<0x4E>(gy);
    if ((lane & 3) == 0 && s < LP) {
        ga_i[s] = ga;
        float2 gl;
        gl.x = k.Wf * k.a * gx;
        gl.y = k.Hf * k.a * gy;
        *reinterpret_cast<float2 *>(gl_i + 2 * s) = gl;
    }
}


template <int CQR>
__device__ __forceinline__ void bwd_home_item(const float *__restrict__ grad_out, const float *__restrict__ value,
                                              const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                              const float *__restrict__ loc, const float *__restrict__ attn, unsigned S, unsigned M,
                                              unsigned LP, float invP, const ItemId id, float *__restrict__ grad_loc,
                                              float *__restrict__ grad_attn)
{
    constexpr unsigned D = 16 * CQR, CQ = 4 * CQR;
    const unsigned lane = threadIdx.x & 63;
    const float *vb = value + (size_t)id.b * S * M * D;
    const float *loc_i = loc + (size_t)id.item * LP * 2;
    const float *att_i = attn + (size_t)id.item * LP;
    float *gl_i = grad_loc + (size_t)id.item * LP * 2;
    float *ga_i = grad_attn + (size_t)id.item * LP;
    const float4 g4 = *reinterpret_cast<const float4 *>(grad_out + (size_t)id.item * D + (lane % CQ) * 4);
    for (unsigned s0 = 0; s0 < LP; s0 += 16) {
        const unsigned s = s0 + (lane >> 2);
        const Entry k = entry_setup<true>(shapes, start, loc_i, att_i, s, lane & 3, LP, invP, M, D, id.m);
        const float d = chunk_dots<CQ>(vb, k, g4, lane);
        store_sample_grads(k, d, lane, s, LP, gl_i, ga_i);
    }
}

// Two items by one wave, their loads interleaved: the item is a chain of two dependent round trips (locations / weights, then
// the corner rows), and inside msda_bwd_tile_accum the gather waves run at four waves per SIMD beside the accumulate blocks --
// too few to hide it (scripts/tile_timeline.py: a gather block took 3.6 us with cold operands, 2.2 warm).  Both items' sixteen
// row gathers are in flight together (128 bytes per lane).  id0.ok; id1.ok may be false (wave-uniform): then item 0 alone.
template <int CQR>
__device__ __forceinline__ void bwd_home_item2(const float *__restrict__ grad_out, const float *__restrict__ value,
                                               const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                               const float *__restrict__ loc, const float *__restrict__ attn, unsigned S, unsigned M,
                                               unsigned LP, float invP, const ItemId id0, const ItemId id1,
                                               float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr unsigned D = 16 * CQR, CQ = 4 * CQR, SLOTS = 64 / CQ, NI = CQ;
    if (!id1.ok) {
        bwd_home_item<CQR>(grad_out, value, shapes, start, loc, attn, S, M, LP, invP, id0, grad_loc, grad_attn);
        return;
    }
    const unsigned lane = threadIdx.x & 63;
    const float *vb0 = value + (size_t)id0.b * S * M * D, *vb1 = value + (size_t)id1.b * S * M * D;
    const float4 g0 = *reinterpret_cast<const float4 *>(grad_out + (size_t)id0.item * D + (lane % CQ) * 4);
    const float4 g1 = *reinterpret_cast<const float4 *>(grad_out + (size_t)id1.item * D + (lane % CQ) * 4);
    const unsigned slot = lane / CQ, cq = lane % CQ;
    const int bp = (int)(slot * 4), back = (int)((lane % SLOTS) * CQ * 4);
    for (unsigned s0 = 0; s0 < LP; s0 += 16) {
        const unsigned s = s0 + (lane >> 2);
        const Entry k0 = entry_setup<true>(shapes, start, loc + (size_t)id0.item * LP * 2, attn + (size_t)id0.item * LP, s,
                                           lane & 3, LP, invP, M, D, id0.m);
        const Entry k1 = entry_setup<true>(shapes, start, loc + (size_t)id1.item * LP * 2, attn + (size_t)id1.item * LP, s,
                                           lane & 3, LP, invP, M, D, id1.m);
        float4 v0[NI], v1[NI];
#pragma unroll
        for (unsigned j = 0; j < NI; ++j) {
            const unsigned o0 = (unsigned)__builtin_amdgcn_ds_bpermute(bp + (int)(j * SLOTS * 4), (int)k0.offb);
            const unsigned o1 = (unsigned)__builtin_amdgcn_ds_bpermute(bp + (int)(j * SLOTS * 4), (int)k1.offb);
            v0[j] = load_row16(vb0, o0 + cq * 16);
            v1[j] = load_row16(vb1, o1 + cq * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (unsigned j = 0; j < NI; ++j) {
            float d0 = v0[j].x * g0.x, d1 = v1[j].x * g1.x;
            d0 = fmaf(v0[j].y, g0.y, d0); d1 = fmaf(v1[j].y, g1.y, d1);
            d0 = fmaf(v0[j].z, g0.z, d0); d1 = fmaf(v1[j].z, g1.z, d1);
            d0 = fmaf(v0[j].w, g0.w, d0); d1 = fmaf(v1[j].w, g1.w, d1);
            d0 = sum_over_row_lanes<CQ>(d0);
            d1 = sum_over_row_lanes<CQ>(d1);
            const float t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(d0)));
            const float t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(back, __float_as_int(d1)));
            if (lane / SLOTS == j) { m0 = t0; m1 = t1; }
        }
        store_sample_grads(k0, k0.inb ? m0 : 0.f, lane, s, LP, grad_loc + (size_t)id0.item * LP * 2, grad_attn + (size_t)id0.item * LP);
        store_sample_grads(k1, k1.inb ? m1 : 0.f, lane, s, LP, grad_loc + (size_t)id1.item * LP * 2, grad_attn + (size_t)id1.item * LP);
    }
}

}  // namespace

#endif
