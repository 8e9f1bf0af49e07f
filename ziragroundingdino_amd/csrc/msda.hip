// msda.hip -- multi-scale deformable attention sampling + aggregation for gfx950 (MI355X).
//
// Hand-written for CDNA4 (64-wide wavefronts); not derived from the reference's CUDA
// kernels.  The arithmetic that has to match is that of the reference op
// (groundingdino/models/GroundingDINO/csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:
//  forward :237-299 + bilinear :33-84, backward :87-159 inside :301-403); the C ABI that
// wraps these kernels is declared in include/zira_msda.h.
//
// Work decomposition ("rows" path, fp32, D = 4*LPR channels per head):
//   * one wavefront owns one (b, q, m) item: LP = L*P samples, 4 bilinear corners each;
//   * phase 1: lane i owns corner (i & 3) of sample (i >> 2) of the current 16-sample chunk:
//     it reads that sample's (x, y, attn) -- 192 contiguous bytes per chunk for the wave --
//     and derives the corner's value-row offset and its weight, once (the reference
//     recomputes this per channel, 32x);
//   * phase 2: the 64 corner rows of the chunk are gathered RPI = 64/LPR rows at a time:
//     LPR consecutive lanes read one whole D*4-byte row with one 16-byte load each (a
//     128-B line for D = 32), the row's (offset, weight) arriving by ds_bpermute;
//   * the RPI partial sums are folded with xor-shuffles and LPR lanes store the D outputs.
//   The backward reuses phases 1-2 to form <grad_out, value_row> per corner row, hands each
//   dot product back to the lane that owns the corner (which holds the bilinear
//   coefficients), and reduces the 4 corners of a sample with two xor-shuffles -- no LDS
//   barrier, no serial reduction.  grad_value is accumulated with hardware fp32 atomics
//   shaped as whole D*4-byte rows (two 128-B segments per wave instruction for D = 32).
//
// Any other D, and float64, run the "generic" element-per-thread kernels further down.

#include <hip/hip_runtime.h>
#include <stdint.h>


#include "msda_internal.h"
#include "zira_msda.h"

#ifndef ZIRA_K1_GATHERS
#define ZIRA_K1_GATHERS 8
#endif

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;

// Head-major placement.  All items of one (b, m) "group" read the same value slice
// value[b, :, m, :] (S*D*4 bytes: 2.8 MB at S=22223, D=32), which fits one XCD's 4 MiB L2.
// The item list is therefore walked group-major (b, m, q) and cut into 8 equal contiguous
// chunks, one per XCD (blocks bid, bid+8, ... share an XCD under round-robin dispatch): every
// XCD then works through ~B*M/8 groups one after the other, fetches each value row from
// HBM / Infinity Cache about once and serves the 4-corner re-reads from its own L2.
// Placement only affects speed; any block->XCD assignment gives the same results.
// Returns the flat (b, q, m) item index for (block, wave), or -1 when the wave has no item.
__device__ __forceinline__ long head_major_item(int bid, int wave, long nitems, int Q, int M,
                                                int &b, int &m)
{
    const int xcd = bid & 7, idx = bid >> 3;
    const long per = (nitems + 7) >> 3;
    const long t0 = xcd * per;
    const long t = t0 + (long)idx * kWavesPerBlock + wave;
    const long t1 = (t0 + per < nitems) ? t0 + per : nitems;
    if (t >= t1) return -1;
    const long g = t / Q;
    const int q = (int)(t - g * Q);
    b = (int)(g / M);
    m = (int)(g - (long)b * M);
    return ((long)b * Q + q) * M + m;
}

inline int head_major_grid(long nitems)
{
    const long per = (nitems + 7) >> 3;
    return (int)(8 * ((per + kWavesPerBlock - 1) / kWavesPerBlock));
}

struct Corner {
    float w;     // bilinear weight * attention weight (0 when the corner contributes nothing)
    int off;     // element offset of the corner's value row inside this batch element
    float wb;    // bilinear weight alone
    float cx;    // d(sample)/d(w_im) coefficient of this corner's value
    float cy;    // d(sample)/d(h_im) coefficient of this corner's value
    float a;     // attention weight of the sample
    float Wf, Hf;
    bool inb;    // corner inside the map and sample inside the (-1,H)x(-1,W) window
};

// Phase 1 for one lane: sample s of item, corner c = (dy, dx).
// Pixel coordinates are formed with separately rounded mul and sub (no fma contraction) so
// that floor() lands on the same pixel as the CPU restatement in oracle/msda_oracle.c.
template <bool kNeedGrad>
__device__ __forceinline__ Corner corner_setup(const int64_t *__restrict__ shapes,
                                               const int64_t *__restrict__ start,
                                               const float *__restrict__ loc_i,
                                               const float *__restrict__ att_i, int s, int c,
                                               int LP, int P, int M, int D, int m)
{
#pragma clang fp contract(off)
    Corner k;
    k.w = 0.f; k.off = 0; k.wb = 0.f; k.cx = 0.f; k.cy = 0.f; k.a = 0.f;
    k.Wf = 0.f; k.Hf = 0.f; k.inb = false;
    if (s < LP) {
        const int l = s / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const int st = (int)start[l];
        const float2 xy = *reinterpret_cast<const float2 *>(loc_i + 2 * s);
        const float a = att_i[s];
        const float h_im = xy.y * (float)H - 0.5f;
        const float w_im = xy.x * (float)W - 0.5f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
            const float hf = floorf(h_im), wf = floorf(w_im);
            const float lh = h_im - hf, lw = w_im - wf;
            const int dy = c >> 1, dx = c & 1;
            const int y = (int)hf + dy, x = (int)wf + dx;
            const float wy = dy ? lh : 1.f - lh;
            const float wx = dx ? lw : 1.f - lw;
            if (y >= 0 && y <= H - 1 && x >= 0 && x <= W - 1) {
                k.inb = true;
                k.wb = wy * wx;
                k.w = k.wb * a;
                k.off = ((st + y * W + x) * M + m) * D;
                if (kNeedGrad) {
                    k.cx = dx ? wy : -wy;
                    k.cy = dy ? wx : -wx;
                }
            }
            if (kNeedGrad) { k.a = a; k.Wf = (float)W; k.Hf = (float)H; }
        }
    }
    return k;
}

__device__ __forceinline__ float4 shfl_xor4(float4 v, int mask)
{
    float4 r;
    r.x = __shfl_xor(v.x, mask);
    r.y = __shfl_xor(v.y, mask);
    r.z = __shfl_xor(v.z, mask);
    r.w = __shfl_xor(v.w, mask);
    return r;
}

// ------------------------------------------------------------------------------------------
// forward, rows path
// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(kBlock) void msda_fwd_rows(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ start, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int M, int L, int Q, int P, long nitems,
    float *__restrict__ out)
{
    constexpr int D = 4 * LPR;
    constexpr int RPI = kWave / LPR;  // value rows gathered per wave instruction
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int b, m;
    const long item = head_major_item(blockIdx.x, wave, nitems, Q, M, b, m);
    if (item < 0) return;  // wave-uniform
    const int LP = L * P;
    const float *vb = value + (size_t)b * S * M * D;
    const float *loc_i = loc + item * LP * 2;
    const float *att_i = attn + item * LP;
    const int r = lane / LPR, cq = lane % LPR;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < LP; s0 += 16) {
        const Corner k = corner_setup<false>(shapes, start, loc_i, att_i, s0 + (lane >> 2),
                                             lane & 3, LP, P, M, D, m);
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const int src = j * RPI + r;
            const float wj = __shfl(k.w, src);
            const int oj = __shfl(k.off, src);
            const float4 v = *reinterpret_cast<const float4 *>(vb + oj + cq * 4);
            acc.x = fmaf(wj, v.x, acc.x);
            acc.y = fmaf(wj, v.y, acc.y);
            acc.z = fmaf(wj, v.z, acc.z);
            acc.w = fmaf(wj, v.w, acc.w);
        }
    }
#pragma unroll
    for (int d = LPR; d < kWave; d <<= 1) {
        const float4 o = shfl_xor4(acc, d);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    if (lane < LPR) *reinterpret_cast<float4 *>(out + item * D + cq * 4) = acc;
}

// ------------------------------------------------------------------------------------------
// backward, rows path (grad_value by fp32 atomics)
// ------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(kBlock) void msda_bwd_rows_atomic(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ attn, int S, int M, int L, int Q,
    int P, long nitems, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn)
{
    constexpr int D = 4 * LPR;
    constexpr int RPI = kWave / LPR;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int b, m;
    const long item = head_major_item(blockIdx.x, wave, nitems, Q, M, b, m);
    if (item < 0) return;
    const int LP = L * P;
    const size_t boff = (size_t)b * S * M * D;
    const float *vb = value + boff;
    float *gvb = grad_value + boff;
    const float *loc_i = loc + item * LP * 2;
    const float *att_i = attn + item * LP;
    const float *g_i = grad_out + item * D;
    const int r = lane / LPR, cq = lane % LPR;
    const float4 g4 = *reinterpret_cast<const float4 *>(g_i + cq * 4);

    for (int s0 = 0; s0 < LP; s0 += 16) {
        const int s = s0 + (lane >> 2);
        const Corner k =
            corner_setup<true>(shapes, start, loc_i, att_i, s, lane & 3, LP, P, M, D, m);

        // <grad_out, value_row> for the 64 corner rows; row i's result ends in lane i.
        float dot_mine = 0.f;
#pragma unroll
        for (int j = 0; j < LPR; ++j) {
            const int src = j * RPI + r;
            const int oj = __shfl(k.off, src);
            const float4 v = *reinterpret_cast<const float4 *>(vb + oj + cq * 4);
            float d = v.x * g4.x + v.y * g4.y + v.z * g4.z + v.w * g4.w;
#pragma unroll
            for (int x = 1; x < LPR; x <<= 1) d += __shfl_xor(d, x);
            const float t = __shfl(d, (lane % RPI) * LPR);
            if (lane / RPI == j) dot_mine = t;
        }
        const float d = k.inb ? dot_mine : 0.f;
        float ga = k.wb * d, gx = k.cx * d, gy = k.cy * d;
        ga += __shfl_xor(ga, 1); gx += __shfl_xor(gx, 1); gy += __shfl_xor(gy, 1);
        ga += __shfl_xor(ga, 2); gx += __shfl_xor(gx, 2); gy += __shfl_xor(gy, 2);
        if ((lane & 3) == 0 && s < LP) {
            grad_attn[item * LP + s] = ga;
            float2 gl;
            gl.x = k.Wf * k.a * gx;
            gl.y = k.Hf * k.a * gy;
            *reinterpret_cast<float2 *>(grad_loc + (item * LP + s) * 2) = gl;
        }

        // grad_value rows: every wave instruction adds whole rows (64/D rows of D floats).
        if constexpr (D <= kWave) {
            constexpr int RPA = kWave / D;
            const int ch = lane % D, rr = lane / D;
            const float gch = g_i[ch];
#pragma unroll 4
            for (int it = 0; it < kWave / RPA; ++it) {
                const int src = it * RPA + rr;
                const float wj = __shfl(k.w, src);
                const int oj = __shfl(k.off, src);
                if (wj != 0.f) unsafeAtomicAdd(gvb + oj + ch, wj * gch);
            }
        } else {
            for (int row = 0; row < kWave; ++row) {
                const float wj = __shfl(k.w, row);
                const int oj = __shfl(k.off, row);
                if (wj != 0.f)
                    for (int ch = lane; ch < D; ch += kWave)
                        unsafeAtomicAdd(gvb + oj + ch, wj * g_i[ch]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// "lean" path: D = 16*CQR in {16, 32, 64}.
//
// The op turned out to be VALU-issue bound on MI355X, not bandwidth bound (removing every
// gather load from an earlier version only took it from 11.8 to 7.4 us at the north-star
// shape), so this version is organised around instruction count per (b, q, m) item:
//   * all index arithmetic is wave-uniform and kept on the scalar unit (32-bit);
//   * phase 1: lane e owns entry e = (sample e>>2, corner e&3) of the 16-sample chunk -- 64
//     distinct (offset, weight) pairs, nothing computed twice;
//   * gather: a 16-lane DPP row R serves channel quads [R*CQR, (R+1)*CQR); inside the row the
//     lanes are (slot, cq_local), slot = one of SLOTS = 16/CQR value rows per instruction.
//     Entry -> slot hand-off is one ds_bpermute per operand whose source lane (j*SLOTS+slot)
//     folds into the instruction's immediate offset;  value rows are addressed as
//     scalar base + 32-bit byte offset (no 64-bit VALU adds);  4 channels x weight is two
//     v_pk_fma_f32;
//   * because all SLOTS partial sums of a channel live in one DPP row, the final reduction
//     is log2(SLOTS) v_add_f32_dpp row rotations -- no cross-row traffic, no LDS.
// ------------------------------------------------------------------------------------------
}  // namespace
#include "msda_fwd_lean.h"
namespace {

template <int CQR>
__global__ __launch_bounds__(kBlock, 8) void msda_fwd_lean(
    const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ start, const float *__restrict__ loc,
    const float *__restrict__ attn, unsigned S, FastDiv Mdiv, unsigned LP, FastDiv Qdiv,
    float invP, unsigned nitems, unsigned per_xcd, float *__restrict__ out)
{
    const ItemId id = lean_item(nitems, per_xcd, Qdiv, Mdiv);
    if (!id.ok) return;  // wave-uniform
    fwd_lean_item<CQR>(value, shapes, start, loc, attn, S, Mdiv.d, LP, invP, id, out);
}

// Backward, lean path, grad_value by fp32 atomics (used when no workspace is supplied).
//
// Gather layout: CQ = D/4 consecutive lanes read one value row, so <grad_out, row> folds with
// DPP quad permutes / row mirrors inside a 16-lane row.  Each dot product travels back to the
// lane that owns the (sample, corner) entry with one ds_bpermute; the four corners of a
// sample are then combined with two more quad permutes.
// (sum_over_row_lanes, chunk_dots, store_sample_grads and bwd_home_item: csrc/msda_fwd_lean.h, shared with the fused
// home + accumulate launch of csrc/msda_tiles.hip)

// kScatter = false (msda_bwd_home): grad_sampling_loc and grad_attn_weight only -- the gather half of the backward, a wave
// per (b, q, m) like the forward; grad_value then comes from the tile accumulate kernel (csrc/msda_tiles.hip).
template <int CQR, bool kScatter>
__global__ __launch_bounds__(kBlock, 8) void msda_bwd_lean_atomic(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ attn, unsigned S, FastDiv Mdiv,
    unsigned LP, FastDiv Qdiv, float invP, unsigned nitems, unsigned per_xcd,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr unsigned D = 16 * CQR, CQ = 4 * CQR;
    const unsigned M = Mdiv.d;
    const ItemId id = lean_item(nitems, per_xcd, Qdiv, Mdiv);
    if (!id.ok) return;
    const unsigned lane = threadIdx.x & 63;
    const size_t boff = (size_t)id.b * S * M * D;
    const float *vb = value + boff;
    float *gvb = grad_value + boff;
    const float *loc_i = loc + (size_t)id.item * LP * 2;
    const float *att_i = attn + (size_t)id.item * LP;
    const float *g_i = grad_out + (size_t)id.item * D;
    float *gl_i = grad_loc + (size_t)id.item * LP * 2;
    float *ga_i = grad_attn + (size_t)id.item * LP;
    const float4 g4 = *reinterpret_cast<const float4 *>(g_i + (lane % CQ) * 4);

    for (unsigned s0 = 0; s0 < LP; s0 += 16) {
        const unsigned s = s0 + (lane >> 2);
        const Entry k = entry_setup<true>(shapes, start, loc_i, att_i, s, lane & 3, LP, invP, M,
                                          D, id.m);
        const float d = chunk_dots<CQ>(vb, k, g4, lane);
        store_sample_grads(k, d, lane, s, LP, gl_i, ga_i);

        // grad_value: whole rows per wave instruction (64/D rows of D floats)
        constexpr unsigned RPA = 64 / D >= 1 ? 64 / D : 1;
        const unsigned ch = lane % D, rr = lane / D;
        const int w_i = __float_as_int(k.w), offb_i = (int)k.offb;
        if (kScatter && D <= 64) {
            const float gch = g_i[ch];
#pragma unroll 4
            for (unsigned it = 0; it < 64 / RPA; ++it) {
                const int a = (int)((it * RPA + rr) * 4);
                const float wj = __int_as_float(__builtin_amdgcn_ds_bpermute(a, w_i));
                const unsigned oj = (unsigned)__builtin_amdgcn_ds_bpermute(a, offb_i);
                if (wj != 0.f)
                    unsafeAtomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(gvb) + oj) + ch,
                                    wj * gch);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward without global atomics ("tiled" path): two kernels and a caller-provided workspace.
//
// Global fp32 atomics execute at the memory side on MI355X (~1.3 TB/s of added bytes chip
// wide): the 118 MB of corner rows the north-star shape scatters into grad_value cost ~90 us
// that way, 5x the rest of the backward.  Instead:
//
//   K1 (msda_bwd_items): one 1024-thread block per IPB = 48 consecutive queries of one
//      (b, m) head.  Every wave handles 3 items: gather + dot products -> grad_sampling_loc /
//      grad_attn_weight (as in the atomic kernel), and turns each contributing (sample,
//      corner) into an 8-byte entry {item_in_block:16 | row_in_tile:16, weight}.  grad_value
//      of head (b, m) is cut into NT = L*T tiles (level l, t-th of T equal pixel ranges);
//      the block counting-sorts its entries by tile in LDS and writes them as one contiguous
//      run per tile into its private slice of the workspace, plus a {offset, count}
//      descriptor per (tile, block).
//   K2 (msda_bwd_tiles): one block per tile.  The tile's grad_value rows live in LDS
//      (<= 60 KB); the block walks the runs addressed to it, multiplies grad_out rows by the
//      entry weights and accumulates with LDS atomics (ds_add_f32), then stores every row
//      of the tile exactly once with plain 16-byte stores.
//
// grad_value is therefore written once, never zero-filled and never touched by a global
// atomic; there is no capacity limit or overflow path (a block's slice holds all of its
// IPB*LP*4 possible entries).  Precondition (as in the reference module,
// ms_deform_attn.py:284): the levels tile [0, S) exactly.
// ------------------------------------------------------------------------------------------
constexpr unsigned kItemsPerWave = 3;  // K1: items per wave; a block has 4 (sparse calls) or 8 (dense) waves
constexpr unsigned kK1DenseWaves = 8;
#ifndef ZIRA_K2_THREADS
#define ZIRA_K2_THREADS 256
#endif
#ifndef ZIRA_K2_MINWAVES
#define ZIRA_K2_MINWAVES 4
#endif
#ifndef ZIRA_K2_U
#define ZIRA_K2_U 4   // grad_out rows in flight per lane
#endif
#ifndef ZIRA_K2_EPT
#define ZIRA_K2_EPT 8
#endif
#ifndef ZIRA_K2_FETCH_GROUP
#define ZIRA_K2_FETCH_GROUP 8
#endif
constexpr unsigned kK2Threads = ZIRA_K2_THREADS;
constexpr unsigned kMaxTileRows = 4095;
#ifndef ZIRA_TILE_ENTRIES
#define ZIRA_TILE_ENTRIES 1024
#endif
constexpr unsigned kTargetTileEntries = ZIRA_TILE_ENTRIES;
constexpr unsigned kInvalidEntry = 0xFFFFFFFFu;

struct TilePlan {
    unsigned T;        // tiles per level
    unsigned NT;       // tiles per head = L * T
    unsigned ipb;      // items (queries) per K1 block = waves * kItemsPerWave
    unsigned nblk;     // K1 blocks per head = ceil(Q / ipb)
    unsigned chunks;   // 16-sample chunks per item = ceil(LP / 16)
    unsigned eblk;     // entry slots per K1 block = ipb * chunks * 64
    unsigned rows;     // LDS rows per tile (upper bound: ceil(S / T))
    unsigned wave_k2;  // 1: msda_bwd_tiles_wave (a wave per tile), 0: msda_bwd_tiles (a block per tile)
    unsigned qwords;   // words of K2's slice queue (wave_k2 only, else 0)
    unsigned runlist;  // 1: K1 publishes a compact run list per tile (dense calls), 0: the tile x block matrix
    unsigned run_base; // runlist: word offset of the records behind the counters
};

// block-granular head-major placement: virtual block id for (XCD = bid & 7, index = bid >> 3)
__device__ __forceinline__ bool xcd_chunk_block(unsigned nvirt, unsigned per, unsigned &vb)
{
    const unsigned xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    vb = xcd * per + idx;
    return idx < per && vb < nvirt;
}

// pixels per tile of a level with hw pixels: ceil(hw / T)
__device__ __forceinline__ unsigned tile_span(unsigned hw, FastDiv T)
{
    return fast_div(hw + T.d - 1, T);
}

// Occupancy: both variants run at <= 80 VGPRs (no spills) with all eight gathers of a chunk in
// flight.  Dense calls use 8-wave blocks (24 queries, 36 KB of LDS: 3 blocks per CU); 16-wave
// blocks needed 64 VGPRs (spills) to fit twice and were 5 % slower once the runs came as lists.
template <int CQR, unsigned kK1Waves>
__global__ __launch_bounds__(kK1Waves * 64, 6) void msda_bwd_items(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ attn, unsigned S, FastDiv Mdiv,
    unsigned LP, float invP, unsigned Q, FastDiv nblkdiv, unsigned nvirt, unsigned per_xcd,
    FastDiv Tdiv, TilePlan plan, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    unsigned *__restrict__ desc, uint2 *__restrict__ region, unsigned *__restrict__ queue)
{
    constexpr unsigned D = 16 * CQR, CQ = 4 * CQR;
    constexpr unsigned kK1Threads = kK1Waves * 64, kIPB = kK1Waves * kItemsPerWave;
    extern __shared__ unsigned lds_k1[];
    unsigned *hist = lds_k1;                    // [NT]   counts, later exclusive offsets
    unsigned *stag = lds_k1 + plan.NT;          // [eblk][3] key, weight, (tile << 16 | rank)
    unsigned *sorted = stag + plan.eblk * 3;    // [eblk][2] key, weight in tile order

    unsigned vblk;
    if (!xcd_chunk_block(nvirt, per_xcd, vblk)) return;  // block-uniform
    const unsigned M = Mdiv.d;
    const unsigned g = fast_div(vblk, nblkdiv), blk = vblk - g * nblkdiv.d;
    const unsigned b = fast_div(g, Mdiv), m = g - b * M;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const float *vb = value + (size_t)b * S * M * D;

    // K2's slice queue (wave-per-tile variant) starts empty: header and slots zeroed here
    for (unsigned i = vblk * kK1Threads + threadIdx.x; i < plan.qwords; i += nvirt * kK1Threads) queue[i] = 0;
    for (unsigned i = threadIdx.x; i < plan.NT; i += kK1Threads) hist[i] = 0;
    __syncthreads();

    for (unsigned it = 0; it < kItemsPerWave; ++it) {
        const unsigned item_local = it * kK1Waves + wave;
        const unsigned q = blk * kIPB + item_local;
        unsigned *st_i = stag + (size_t)(item_local * plan.chunks) * 64 * 3;
        if (q >= Q) {  // wave-uniform: no such query, mark the slots empty
            for (unsigned ch = 0; ch < plan.chunks; ++ch) st_i[(ch * 64 + lane) * 3 + 2] = kInvalidEntry;
            continue;
        }
        const unsigned item = (b * Q + q) * M + m;
        const float *loc_i = loc + (size_t)item * LP * 2;
        const float *att_i = attn + (size_t)item * LP;
        const float *g_i = grad_out + (size_t)item * D;
        float *gl_i = grad_loc + (size_t)item * LP * 2;
        float *ga_i = grad_attn + (size_t)item * LP;
        const float4 g4 = *reinterpret_cast<const float4 *>(g_i + (lane % CQ) * 4);
        for (unsigned ch = 0; ch < plan.chunks; ++ch) {
            const unsigned s = ch * 16 + (lane >> 2);
            const Entry k = entry_setup<true>(shapes, start, loc_i, att_i, s, lane & 3, LP, invP,
                                              M, D, m);
            const float d = chunk_dots<CQ>(vb, k, g4, lane);
            store_sample_grads(k, d, lane, s, LP, gl_i, ga_i);

            unsigned tr = kInvalidEntry, key = 0;
            if (k.inb && k.w != 0.f) {
                const unsigned span = tile_span(k.hw, Tdiv);
                // t = pix / span: float estimate, then exact fix-up
                unsigned t = (unsigned)(((float)k.pix + 0.5f) * __builtin_amdgcn_rcpf((float)span));
                if (t * span > k.pix) --t;
                else if ((t + 1) * span <= k.pix) ++t;
                const unsigned tile = k.lvl * plan.T + t;
                const unsigned rank = atomicAdd(&hist[tile], 1u);
                key = (item_local << 16) | (k.pix - t * span);
                tr = (tile << 16) | rank;
            }
            unsigned *e = st_i + (ch * 64 + lane) * 3;
            e[0] = key;
            e[1] = __float_as_uint(k.w);
            e[2] = tr;
        }
    }
    __syncthreads();

    // exclusive scan of the tile histogram in chunks of kK1Threads tiles (wave shuffles, wave
    // totals through LDS); hist[] is overwritten with the offsets, desc gets {offset, count}
    __shared__ unsigned wave_tot[kK1Waves];
    unsigned total = 0;
    for (unsigned c0 = 0; c0 < plan.NT; c0 += kK1Threads) {
        const unsigned ti = c0 + threadIdx.x;
        const unsigned n_mine = ti < plan.NT ? hist[ti] : 0u;
        unsigned incl = n_mine;
#pragma unroll
        for (unsigned dlt = 1; dlt < 64; dlt <<= 1) {
            const unsigned o = __shfl_up(incl, dlt);
            if (lane >= dlt) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned wbase = 0, ctot = 0;
#pragma unroll
        for (unsigned w = 0; w < kK1Waves; ++w) {
            const unsigned tot = wave_tot[w];
            if (w < wave) wbase += tot;
            ctot += tot;
        }
        const unsigned excl = total + wbase + incl - n_mine;
        if (ti < plan.NT) {
            hist[ti] = excl;
            if (plan.runlist) {
                // dense calls: a block of consecutive pixel-grid queries touches a few dozen of the
                // ~1400 tiles of its head; only those runs are published, appended to the tile's list
                // (desc = [heads * NT] run counters, zeroed by the host, then [heads * NT][nblk] records)
                if (n_mine) {
                    const size_t tg = (size_t)g * plan.NT + ti;
                    const unsigned at = atomicAdd(&desc[tg], 1u);
                    uint2 *runs = reinterpret_cast<uint2 *>(desc + plan.run_base) + tg * plan.nblk;
                    runs[at] = make_uint2((blk << 16) | n_mine, excl);
                }
            } else {
                desc[((size_t)g * plan.NT + ti) * plan.nblk + blk] = (excl << 16) | n_mine;
            }
        }
        total += ctot;
        __syncthreads();
    }

    for (unsigned i = threadIdx.x; i < plan.eblk; i += kK1Threads) {
        const unsigned tr = stag[i * 3 + 2];
        if (tr != kInvalidEntry) {
            const unsigned dst = hist[tr >> 16] + (tr & 0xffffu);
            sorted[dst * 2] = stag[i * 3];
            sorted[dst * 2 + 1] = stag[i * 3 + 1];
        }
    }
    __syncthreads();
    uint2 *out = region + (size_t)vblk * plan.eblk;
    const uint2 *src = reinterpret_cast<const uint2 *>(sorted);
    for (unsigned i = threadIdx.x; i < total; i += kK1Threads) out[i] = src[i];
}

// ---- K2 helpers -------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x)
{
    return __builtin_amdgcn_update_dpp(0u, x, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float4 dpp_f4(float4 v)
{
    float4 r;
    r.x = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v.x)));
    r.y = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v.y)));
    r.z = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v.z)));
    r.w = __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v.w)));
    return r;
}
__device__ __forceinline__ void add4(float4 &a, const float4 &b)
{
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
}

// exclusive prefix sum of one value per thread over the block (kK2Threads threads);
// returns the block total through `total`.  `scratch` holds kK2Threads/64 words.
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned *scratch,
                                                         unsigned &total)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = v;
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();  // scratch may still be read from a previous scan
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    total = 0;
#pragma unroll
    for (unsigned w = 0; w < kK2Threads / 64; ++w) {
        const unsigned t = scratch[w];
        if (w < wave) base += t;
        total += t;
    }
    return base + incl - v;
}

// K2: one block per grad_value tile (head (b, m), level l, pixel range).
//
// Two earlier versions of this kernel accumulated the tile in LDS: with ds_add_f32 (LDS fp32
// atomics run at ~2 cycles per LANE on gfx950: 110 of 160 us) and with plain LDS
// read-modify-writes under a row-ownership scheme (fine per entry, but the 44 KB tile forces
// 4096 blocks = 5 rounds of a ~12 us dependent-latency chain).  This version keeps no tile at
// all: the tile's entries (runs written by K1, one per K1 block) are counting-sorted by tile
// row in LDS, every wave owns a contiguous range of rows, and a row's sum is formed in
// registers: one wave instruction fetches the grad_out rows of NSLOT = 256/D entries (D*4
// contiguous bytes each, 16 B per lane), U of them are in flight, a segmented DPP scan folds
// neighbouring entries of the same row, and each finished row is stored once with plain
// 16-byte stores.  Rows without entries are stored as zeros.  The dependent memory chain per
// block is descriptor -> entries -> grad_out rows -> store, and the tile count is chosen so
// that all blocks are resident at once (one round).
// When a tile has more than `cap` entries (the LDS batch), later batches read-modify-write
// the rows they touch; a row is always handled by the same wave, so program order suffices.
// How a finished row sum reaches grad_value:
//   kRowStore  the row belongs to this batch alone: plain 16-byte store
//   kRowRmw    a later batch of the same tile (block-per-tile K2, tile larger than the LDS batch):
//              read-modify-write; the earlier value may come from another wave of the block (same
//              CU, same L1: the block barrier between batches orders it)
//   kRowAtomic the tile is shared between waves anywhere on the chip (wave-per-tile K2, slices of
//              a heavy tile): device-scope fp32 atomics onto rows the tile's owner zeroed with
//              device-scope stores before it published the slices.  The eight XCDs have private
//              L2s: plain stores would need a full L2 write-back (__threadfence) to be seen by
//              another XCD, which costs milliseconds when hundreds of owners do it.
enum : int { kRowStore = 0, kRowRmw = 1, kRowAtomic = 2 };

__device__ __forceinline__ void flush_row(float *p, float4 acc, int mode)
{
    if (mode == kRowAtomic) {
        unsafeAtomicAdd(p + 0, acc.x);
        unsafeAtomicAdd(p + 1, acc.y);
        unsafeAtomicAdd(p + 2, acc.z);
        unsafeAtomicAdd(p + 3, acc.w);
        return;
    }
    if (mode == kRowRmw) add4(acc, *reinterpret_cast<const float4 *>(p));
    *reinterpret_cast<float4 *>(p) = acc;
}

struct RowCarry {
    unsigned row;
    float4 val;
};

template <unsigned NSLOT>
__device__ __forceinline__ void rowsum_step(unsigned row, float4 val, bool valid, unsigned last,
                                            unsigned slot, unsigned cq, int mode,
                                            float *__restrict__ gv_t, size_t row_stride,
                                            RowCarry &carry)
{
    constexpr unsigned kInvalidRow = 0xFFFFFFFFu;
    if (carry.row != kInvalidRow) {  // wave-uniform
        const unsigned row_first = __builtin_amdgcn_readfirstlane(row);
        if (row_first == carry.row) {
            if (slot == 0) add4(val, carry.val);
        } else if (slot == 0) {
            flush_row(gv_t + carry.row * row_stride, carry.val, mode);
        }
    }
    // segmented inclusive scan over the NSLOT adjacent lanes (entries are row-sorted)
    {
        const unsigned nr = dpp_u32<0x111>(row);  // row_shr:1
        const float4 nv = dpp_f4<0x111>(val);
        if (slot >= 1 && nr == row) add4(val, nv);
    }
    if (NSLOT > 2) {
        const unsigned nr = dpp_u32<0x112>(row);
        const float4 nv = dpp_f4<0x112>(val);
        if (slot >= 2 && nr == row) add4(val, nv);
    }
    if (NSLOT > 4) {
        const unsigned nr = dpp_u32<0x114>(row);
        const float4 nv = dpp_f4<0x114>(val);
        if (slot >= 4 && nr == row) add4(val, nv);
    }
    if (NSLOT > 8) {
        const unsigned nr = dpp_u32<0x118>(row);
        const float4 nv = dpp_f4<0x118>(val);
        if (slot >= 8 && nr == row) add4(val, nv);
    }
    const unsigned next_row = dpp_u32<0x101>(row);  // row_shl:1
    const bool tail = valid && slot != last && (slot == NSLOT - 1 || next_row != row);
    if (tail) flush_row(gv_t + row * row_stride, val, mode);
    // The last entry's running sum travels on to the next NSLOT entries.  Only slot 0 ever
    // consumes it, and lane (cq, 0) sits NSLOT-1 lanes below lane (cq, NSLOT-1) in the same
    // DPP row, so a full step hands it over with row_shl:(NSLOT-1).  A partial step is the
    // last one of the wave's range: its tail is stored right away by the lanes that hold it.
    if (last == NSLOT - 1) {
        carry.row = __builtin_amdgcn_readlane(row, NSLOT - 1);
        carry.val = dpp_f4<0x100 + (NSLOT - 1)>(val);
    } else {
        carry.row = kInvalidRow;
        if (slot == last) flush_row(gv_t + row * row_stride, val, mode);
    }
}

// entry e of the tile -> K1 block and position in the K1 region, through the run prefix in LDS
__device__ __forceinline__ void locate_tile_entry(const unsigned *pre, const unsigned *runoff,
                                                  unsigned nblk, unsigned eblk, unsigned e,
                                                  unsigned &blk, unsigned &pos)
{
    unsigned lo = 0, hi = nblk;  // largest blk with pre[blk] <= e
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (pre[mid] <= e) lo = mid; else hi = mid;
    }
    blk = lo;
    pos = lo * eblk + runoff[lo] + (e - pre[lo]);
}

// entry e of the tile -> index of its run (largest r with pre[r] <= e) among `nruns` runs
__device__ __forceinline__ unsigned locate_run(const unsigned *pre, unsigned nruns, unsigned e)
{
    unsigned lo = 0, hi = nruns;
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (pre[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// entry e of the tile -> (K1 block, entry) through the run prefix kept in LDS
__device__ __forceinline__ uint2 fetch_tile_entry(const uint2 *__restrict__ reg_g,
                                                  const unsigned *pre, const unsigned *runoff,
                                                  unsigned nblk, unsigned eblk, unsigned e,
                                                  unsigned &blk)
{
    unsigned lo = 0, hi = nblk;  // largest blk with pre[blk] <= e
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) >> 1;
        if (pre[mid] <= e) lo = mid; else hi = mid;
    }
    blk = lo;
    return reg_g[(size_t)lo * eblk + runoff[lo] + (e - pre[lo])];
}

// Row sums of the row-sorted entries sorted[0, n) (LDS), formed by NW waves.
//
// A wave is NSLOT groups of D/4 lanes (lane = channel quad `cq` of group `slot`).  The range is
// cut into NW * NSLOT equal slices; group `slot` of wave `wave` walks slice wave * NSLOT + slot IN
// ORDER, one entry per step (U grad_out rows in flight), keeping the running sum of the current
// row in registers: a row change stores the finished row with plain 16-byte stores.  That is ~4
// VALU instructions per entry where a segmented scan across the groups costs ~12 (K2 is
// VALU-bound), and the slices are balanced by entries, not by rows.  Only a slice's first and
// last row can be shared with a neighbouring slice: those 2 * NW * NSLOT partial sums go through
// `part` (LDS) and are folded by wave 0 with 2 * NW steps of the segmented scan.  Slices without a
// second row (or without entries) contribute zero records on a neighbouring row, which keeps the
// record list row-sorted.  With NW == 1 `part` may alias `sorted` (it is written after the last
// read of the range); with NW > 1 the caller's barriers separate the two.
constexpr unsigned kRowsumPartWords = 512 + 32;  // per wave: 2 * NSLOT records of 64 / NSLOT float4 + their rows

template <unsigned NSLOT, unsigned U, unsigned NW, bool kPrefetch>
__device__ __forceinline__ void rowsum_slices(const uint2 *sorted, unsigned n,
                                              const float *__restrict__ g_bm,
                                              float *__restrict__ gv_t, size_t row_stride,
                                              int mode, unsigned wave, unsigned slot,
                                              unsigned cq, unsigned *part)
{
    constexpr unsigned kInvalidRow = 0xFFFFFFFFu;
    constexpr unsigned CQN = 64 / NSLOT;
    // n > 0 (block-uniform, checked by the caller)
    const unsigned per = (n + NW * NSLOT - 1) / (NW * NSLOT);
    const unsigned sid = wave * NSLOT + slot;
    const unsigned a = sid * per;
    const unsigned b = (a + per < n) ? a + per : n;
    const unsigned lastrow = sorted[n - 1].x & 0xfffu;
    const unsigned hrow = a < b ? (sorted[a].x & 0xfffu) : lastrow;
    unsigned cur = hrow;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), hval = acc;
    // two batches of U entries in flight: while one is folded the other one's grad_out rows load
    struct Batch {
        unsigned row[U];
        float w[U];
        float4 g[U];
    };
    // Block-per-tile K2 (kPrefetch false) issues branch-free and behind a scheduling barrier: with
    // the loads under `if (e < b)` the compiler has been seen to wait for each one before issuing
    // the next (536 -> 625 us on the encoder shape); lanes past their slice read the last entry of
    // the range and ignore it.  The wave-per-tile K2 schedules the predicated form well and saves
    // the wasted loads (48.8 vs 50.3 us).  (Its fold does wait with vmcnt(0), i.e. for the batch
    // issued just before it as well; a branch-free ping/pong with exact vmcnt(U..) waits was
    // measured again later: 46.4 -> 45.9 us with uniform locations, but 78 -> 81.5 us on the
    // model's clustered ones.  Dropping the helper launch's walk-time atomics altogether (wrong results)
    // only takes the model-shaped backward from 83 to 73 us, and parking finished rows in LDS to issue
    // their atomics four rows at a time made it slower (82.7 -> 88.5 us).)
    auto issue = [&](Batch &t, unsigned i) {
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            const unsigned e = a + i + u;
            if (kPrefetch) {
                t.row[u] = kInvalidRow;
                t.w[u] = 0.f;
                t.g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < b) {
                    const uint2 en = sorted[e];
                    t.row[u] = en.x & 0xfffu;
                    t.w[u] = __uint_as_float(en.y);
                    t.g[u] = *reinterpret_cast<const float4 *>(g_bm + (size_t)(en.x >> 12) * row_stride);
                }
            } else {
                const uint2 en = sorted[e < n ? e : n - 1];
                t.row[u] = e < b ? (en.x & 0xfffu) : kInvalidRow;
                t.w[u] = __uint_as_float(en.y);
                t.g[u] = *reinterpret_cast<const float4 *>(g_bm + (size_t)(en.x >> 12) * row_stride);
            }
        }
    };
    auto fold = [&](const Batch &t) {
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            if (t.row[u] != kInvalidRow) {
                if (t.row[u] != cur) {
                    if (cur == hrow) hval = acc;
                    else flush_row(gv_t + cur * row_stride, acc, mode);
                    cur = t.row[u];
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                acc.x = fmaf(t.w[u], t.g[u].x, acc.x);
                acc.y = fmaf(t.w[u], t.g[u].y, acc.y);
                acc.z = fmaf(t.w[u], t.g[u].z, acc.z);
                acc.w = fmaf(t.w[u], t.g[u].w, acc.w);
            }
        }
    };
    if (kPrefetch) {  // wave-per-tile K2: one tile per wave, the chain of round trips is what costs
        Batch ping, pong;
        issue(ping, 0);
        for (unsigned i = 0; i < per; i += 2 * U) {
            issue(pong, i + U);
            fold(ping);
            issue(ping, i + 2 * U);
            fold(pong);
        }
    } else {          // block-per-tile K2: the registers buy more than the overlap (measured)
        for (unsigned i = 0; i < per; i += U) {
            Batch t;
            issue(t, i);
            __builtin_amdgcn_sched_barrier(0);  // keep the U loads together, ahead of the first use
            fold(t);
        }
    }
    float4 tval = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cur == hrow) hval = acc; else tval = acc;
    if (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();  // every lane is done with sorted[]
    }
    float4 *pv = reinterpret_cast<float4 *>(part);      // [2 * NW * NSLOT][CQN]
    unsigned *pr = part + 4 * 2 * NW * NSLOT * CQN;     // [2 * NW * NSLOT]
    pv[(2 * sid) * CQN + cq] = hval;
    pv[(2 * sid + 1) * CQN + cq] = tval;
    if (cq == 0) { pr[2 * sid] = hrow; pr[2 * sid + 1] = cur; }
}

template <unsigned NSLOT, unsigned NW>
__device__ __forceinline__ void rowsum_fold(const unsigned *part, float *__restrict__ gv_t,
                                            size_t row_stride, int mode, unsigned slot,
                                            unsigned cq)
{
    constexpr unsigned kInvalidRow = 0xFFFFFFFFu;
    constexpr unsigned CQN = 64 / NSLOT;
    const float4 *pv = reinterpret_cast<const float4 *>(part);
    const unsigned *pr = part + 4 * 2 * NW * NSLOT * CQN;
    RowCarry carry;
    carry.row = kInvalidRow;
    carry.val = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (unsigned h = 0; h < 2 * NW; ++h) {
        const unsigned k = h * NSLOT + slot;
        rowsum_step<NSLOT>(pr[k], pv[k * CQN + cq], true, NSLOT - 1, slot, cq, mode, gv_t, row_stride,
                           carry);
    }
    if (carry.row != kInvalidRow && slot == 0) flush_row(gv_t + carry.row * row_stride, carry.val, mode);
}

// K2, wave-per-tile variant for sparse calls (decoder cross-attention: a few hundred queries, a
// few hundred entries per tile).  Same algorithm as msda_bwd_tiles, but every WAVE owns a tile
// of <= kWaveTileRows rows and runs it start to finish on its own -- no block barriers, 20
// independent tiles in flight per CU instead of 4 -- so that the cost of a tile is one dependent
// chain (descriptors -> entries -> grad_out rows -> store).
//
// Queries cluster on objects, so some tiles hold 10-20x the mean number of entries (measured in
// the model; uniform synthetic inputs do not show it: ~190 +- 14 entries per tile).  A tile with
// more than kHeavyTile entries is cut into slices of kSliceEntries: its owner handles slice 0
// like any other tile and publishes slices 1.. in a small queue in the workspace (one fetch-add
// per heavy tile).  A second launch of the same kernel (kHelpers = true, a fixed grid that strides
// over the queue; it ends at once when the queue is empty) adds the remaining slices onto the
// stored rows with fp32 atomics -- the only place the tiled path uses them.  Shorter slices or a
// lower threshold were measured and lose (every slice pays the descriptor prefix again and adds
// its own atomics: 128-entry slices 86 us, threshold 256 94 us, 512 / 512 83 us in the model).
// The kernel boundary is what makes the owners' plain stores visible to the atomics: the eight
// XCDs have private L2s and fp32 atomics execute at the memory side.  (A first version let
// finishing owner waves pop slices inside the same launch: 5000 waves contending for one queue
// head with device-scope compare-and-swap took 14 ms.)
constexpr unsigned kWaveTileRows = 512;   // rows per tile (upper bound)
#ifndef ZIRA_K2W_CAP
#define ZIRA_K2W_CAP 512
#endif
#ifndef ZIRA_K2W_U
#define ZIRA_K2W_U 4   // grad_out rows per lane and batch; two batches are in flight (helpers: half)
#endif
#ifndef ZIRA_K2W_UH
#define ZIRA_K2W_UH (ZIRA_K2W_U / 2)   // ... of the slice launch
#endif
#ifndef ZIRA_K2W_MINWAVES
#define ZIRA_K2W_MINWAVES 5  // waves per SIMD: 20 per CU (<= 96 VGPRs, ~7 KB of LDS per wave)
#endif
#ifndef ZIRA_K2W_HEAVY
#define ZIRA_K2W_HEAVY 512
#endif
#ifndef ZIRA_K2W_SLICE
#define ZIRA_K2W_SLICE 320
#endif
constexpr unsigned kWaveTileCap = ZIRA_K2W_CAP;    // LDS sort capacity of a wave (entries)
constexpr unsigned kHeavyTile = ZIRA_K2W_HEAVY;    // tiles above this many entries are sliced
constexpr unsigned kSliceEntries = ZIRA_K2W_SLICE; // entries per slice of a heavy tile
static_assert(kHeavyTile <= kWaveTileCap && kSliceEntries <= kWaveTileCap, "a slice / light tile must fit the LDS sort");
constexpr unsigned kWaveK2Waves = 4;
// LDS words of a wave's sorted entries; the partial records of rowsum_slices alias them
constexpr unsigned kWaveSortWords = 2 * kWaveTileCap > kRowsumPartWords ? 2 * kWaveTileCap : kRowsumPartWords;
constexpr unsigned kWaveHelperBlocks = 512;  // helper launch: 2048 waves stride over the queue (an empty launch costs ~2 us whatever the grid)
// LDS words of one wave of msda_bwd_tiles_wave: rowcnt[R], rowbase[R+1], pre[nblk+1], runoff[nblk],
// runblk[nblk] (rounded up to an even count), then the sorted entries
__host__ __device__ inline unsigned wave_meta_words(unsigned R, unsigned nblk)
{
    return (2 * R + 2 + 3 * nblk + 1 + 1) & ~1u;
}
constexpr unsigned kQueueHeader = 4;       // words: [0] tail, [1] head, [2..3] unused
constexpr unsigned kQueueSliceBits = 13;   // item = ((virtual tile << 13) | slice) + 1

__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v, unsigned lane)
{
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

template <int D, bool kHelpers, bool kRunList>
__global__ __launch_bounds__(kWaveK2Waves * 64, kHelpers ? 4 : ZIRA_K2W_MINWAVES) void msda_bwd_tiles_wave(
    const float *__restrict__ grad_out, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ start, unsigned S, FastDiv Mdiv, unsigned Q, unsigned nvirt,
    unsigned per_xcd, FastDiv Tdiv, FastDiv NTdiv, TilePlan plan,
    const unsigned *__restrict__ desc, const uint2 *__restrict__ region,
    unsigned *__restrict__ queue, float *__restrict__ grad_value)
{
    constexpr unsigned NSLOT = 256 / D;
    constexpr unsigned U = kHelpers ? ZIRA_K2W_UH : ZIRA_K2W_U;
    constexpr unsigned EPL = kWaveTileCap / 64;  // entries per lane and slice
    constexpr unsigned kInvalidRow = 0xFFFFFFFFu;
    extern __shared__ unsigned lds_k2w[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned R = plan.rows;
    const unsigned per_wave = wave_meta_words(R, plan.nblk) + kWaveSortWords;  // words (even)
    unsigned *base = lds_k2w + (size_t)wave * per_wave;
    unsigned *rowcnt = base;                      // [R]
    unsigned *rowbase = rowcnt + R;               // [R + 1]
    unsigned *pre = rowbase + R + 1;              // [nblk + 1]
    unsigned *runoff = pre + plan.nblk + 1;       // [nblk] position of the run in the head's region
    unsigned *runblk = runoff + plan.nblk;        // [nblk] K1 block of the run
    uint2 *sorted = reinterpret_cast<uint2 *>(base + wave_meta_words(R, plan.nblk));
    const unsigned slot = lane % NSLOT, cq = lane / NSLOT;
    const unsigned M = Mdiv.d;
    const size_t row_stride = (size_t)M * D;

    unsigned vb2 = 0, slice = 0, qi = 0, qn = 0;
    if (kHelpers) {  // stride over the published slices
        qi = blockIdx.x * kWaveK2Waves + wave;
        qn = __builtin_amdgcn_readfirstlane(queue[0]);
    } else {         // wave-granular head-major placement of the tiles
        const unsigned xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        vb2 = xcd * per_xcd + idx * kWaveK2Waves + wave;
        if (idx * kWaveK2Waves + wave >= per_xcd || vb2 >= nvirt) return;  // wave-uniform
    }
    const unsigned stamp_id = kHelpers ? 6000u + blockIdx.x * kWaveK2Waves + wave : vb2;  // (developer stamps)
    (void)stamp_id;
    for (;; qi += gridDim.x * kWaveK2Waves) {
        if (kHelpers) {
            if (qi >= qn) break;
            const unsigned item = __builtin_amdgcn_readfirstlane(queue[kQueueHeader + qi]) - 1;
            vb2 = item >> kQueueSliceBits;
            slice = item & ((1u << kQueueSliceBits) - 1);
        }
        bool more = false;  // dense calls: the owner walks the slices of an overfull tile itself
        const unsigned g = fast_div(vb2, NTdiv), tile = vb2 - g * plan.NT;
        const unsigned l = fast_div(tile, Tdiv), t = tile - l * plan.T;
        const unsigned b = fast_div(g, Mdiv), m = g - b * M;
        const unsigned hw = (unsigned)shapes[2 * l] * (unsigned)shapes[2 * l + 1];
        const unsigned st = (unsigned)start[l];
        const unsigned span = tile_span(hw, Tdiv);
        const unsigned p0 = t * span;
        if (p0 < hw) {  // else: tile past the end of a small level (owners only)
            const unsigned rows = (hw - p0 < span) ? hw - p0 : span;
            const float *g_bm = grad_out + ((size_t)b * Q * M + m) * D + cq * 4;
            float *gv_t = grad_value + (((size_t)b * S + st + p0) * M + m) * D + cq * 4;
            const size_t tg = (size_t)g * plan.NT + tile;
            const unsigned nruns =
                kRunList ? __builtin_amdgcn_readfirstlane(desc[tg]) : plan.nblk;  // see msda_bwd_tiles
            const unsigned *dsc = desc + tg * plan.nblk;
            const uint2 *runs = reinterpret_cast<const uint2 *>(desc + plan.run_base) + tg * plan.nblk;
            const uint2 *reg_g = region + (size_t)g * plan.nblk * plan.eblk;

            // run-length prefix over the runs of this tile
            unsigned N = 0;
            for (unsigned c0 = 0; c0 < nruns; c0 += 64) {
                const unsigned i = c0 + lane;
                unsigned n = 0, off = 0, rb = i;
                if (i < nruns) {
                    if (kRunList) {
                        const uint2 rec = runs[i];
                        n = rec.x & 0xffffu; rb = rec.x >> 16; off = rec.y;
                    } else {
                        const unsigned dd = dsc[i];
                        n = dd & 0xffffu; off = dd >> 16;
                    }
                }
                const unsigned incl = wave_inclusive_scan(n, lane);
                if (i < nruns) {
                    pre[i] = N + incl - n;
                    if (kRunList) { runoff[i] = rb * plan.eblk + off; runblk[i] = rb; }
                    else runoff[i] = off;  // run i is K1 block i
                }
                N += __shfl(incl, 63);
            }
            if (lane == 0) pre[nruns] = N;

            const bool heavy = N > kHeavyTile;  // wave-uniform
            if (!kHelpers && heavy && !kRunList && lane == 0) {  // publish slices 1 .. extra
                const unsigned extra = (N - 1) / kSliceEntries;
                const unsigned at = __hip_atomic_fetch_add(&queue[0], extra, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned k = 1; k <= extra; ++k)
                    queue[kQueueHeader + at + k - 1] = ((vb2 << kQueueSliceBits) | k) + 1;
            }
            // Dense calls (run lists) have a dozen rounds of tiles per wave slot, so an overfull tile
            // is no tail: its owner takes the slices one after the other (read-modify-write of its
            // own rows) instead of publishing them for the atomic helper launch.
            const int mode = kHelpers ? kRowAtomic : (slice == 0 ? kRowStore : kRowRmw);
            const unsigned span_e = heavy ? kSliceEntries : kWaveTileCap;
            more = !kHelpers && kRunList && heavy && (slice + 1) * span_e < N;
            const unsigned e_lo = slice * span_e;
            const unsigned nb = (N - e_lo < span_e) ? N - e_lo : span_e;

            for (unsigned i = lane; i < rows; i += 64) rowcnt[i] = 0;
            __builtin_amdgcn_wave_barrier();
            // three separate sweeps so that the binary searches (LDS), the entry loads (global) and
            // the rank atomics (LDS) of the EPL entries of a lane overlap instead of chaining
            unsigned keyr[EPL], wr[EPL], rankr[EPL], posr[EPL];
#pragma unroll
            for (unsigned u = 0; u < EPL; ++u) {
                const unsigned i = lane + u * 64;
                keyr[u] = kInvalidRow;
                posr[u] = 0;
                if (i < nb) {
                    if (kRunList) {
                        const unsigned r = locate_run(pre, nruns, e_lo + i);
                        posr[u] = runoff[r] + (e_lo + i - pre[r]);
                        keyr[u] = runblk[r] * plan.ipb;
                    } else {
                        unsigned blk;
                        locate_tile_entry(pre, runoff, plan.nblk, plan.eblk, e_lo + i, blk, posr[u]);
                        keyr[u] = blk * plan.ipb;
                    }
                }
            }
            uint2 enr[EPL];
#pragma unroll
            for (unsigned u = 0; u < EPL; ++u)
                enr[u] = keyr[u] != kInvalidRow ? reg_g[posr[u]] : make_uint2(0u, 0u);
#pragma unroll
            for (unsigned u = 0; u < EPL; ++u) {
                if (keyr[u] != kInvalidRow) {
                    const unsigned row = enr[u].x & 0xffffu;
                    keyr[u] = ((keyr[u] + (enr[u].x >> 16)) << 12) | row;
                    wr[u] = enr[u].y;
                    rankr[u] = atomicAdd(&rowcnt[row], 1u);
                }
            }
            __builtin_amdgcn_wave_barrier();
            unsigned run = 0;
            for (unsigned c0 = 0; c0 < rows; c0 += 64) {  // exclusive prefix over the rows
                const unsigned r = c0 + lane;
                const unsigned n = r < rows ? rowcnt[r] : 0u;
                const unsigned incl = wave_inclusive_scan(n, lane);
                if (r < rows) rowbase[r] = run + incl - n;
                run += __shfl(incl, 63);
            }
            if (lane == 0) rowbase[rows] = run;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (unsigned u = 0; u < EPL; ++u)
                if (keyr[u] != kInvalidRow)
                    sorted[rowbase[keyr[u] & 0xfffu] + rankr[u]] = make_uint2(keyr[u], wr[u]);
            __builtin_amdgcn_wave_barrier();

            if (!kHelpers && slice == 0) {  // rows nobody contributes to (in slice 0) are stored as zeros
                for (unsigned r = slot; r < rows; r += NSLOT)
                    if (rowbase[r + 1] == rowbase[r])
                        *reinterpret_cast<float4 *>(gv_t + r * row_stride) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (nb) {
                unsigned *part = reinterpret_cast<unsigned *>(sorted);
                rowsum_slices<NSLOT, U, 1, true>(sorted, nb, g_bm, gv_t, row_stride, mode, 0, slot, cq, part);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                rowsum_fold<NSLOT, 1>(part, gv_t, row_stride, mode, slot, cq);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (!kHelpers) {
            if (!more) break;
            ++slice;
            qi -= gridDim.x * kWaveK2Waves;  // (undo the loop increment: same wave, same tile)
        }
    }
}

// ------------------------------------------------------------------------------------------
// generic path: any D, float or double; one thread per (b, q, m, c)
// ------------------------------------------------------------------------------------------
template <typename T>
struct Sample {
    T w1, w2, w3, w4, lh, lw;
    int o1, o2, o3, o4;  // element offsets inside the batch element, -1 = outside
    bool valid;
};

template <typename T>
__device__ __forceinline__ T mul_sub_half(T a, T b)
{
#pragma clang fp contract(off)
    const T prod = a * b;
    return prod - (T)0.5;
}

template <typename T>
__device__ __forceinline__ Sample<T> sample_setup(T lx, T ly, int H, int W, int st, int M, int D,
                                                  int m)
{
    Sample<T> s;
    const T h_im = mul_sub_half<T>(ly, (T)H), w_im = mul_sub_half<T>(lx, (T)W);
    s.valid = h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W;
    s.o1 = s.o2 = s.o3 = s.o4 = -1;
    s.w1 = s.w2 = s.w3 = s.w4 = s.lh = s.lw = 0;
    if (s.valid) {
        const T hf = floor(h_im), wf = floor(w_im);
        const int hl = (int)hf, wl = (int)wf, hh = hl + 1, wh = wl + 1;
        s.lh = h_im - hf; s.lw = w_im - wf;
        const T hhw = 1 - s.lh, hww = 1 - s.lw;
        s.w1 = hhw * hww; s.w2 = hhw * s.lw; s.w3 = s.lh * hww; s.w4 = s.lh * s.lw;
        if (hl >= 0 && wl >= 0) s.o1 = ((st + hl * W + wl) * M + m) * D;
        if (hl >= 0 && wh <= W - 1) s.o2 = ((st + hl * W + wh) * M + m) * D;
        if (hh <= H - 1 && wl >= 0) s.o3 = ((st + hh * W + wl) * M + m) * D;
        if (hh <= H - 1 && wh <= W - 1) s.o4 = ((st + hh * W + wh) * M + m) * D;
    }
    return s;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void msda_fwd_generic(
    const T *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ start, const T *__restrict__ loc, const T *__restrict__ attn,
    int S, int M, int D, int L, int Q, int P, long n, T *__restrict__ out)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long item = idx / D;
        const int m = (int)(item % M);
        const int b = (int)(item / ((long)M * Q));
        const T *vb = value + (size_t)b * S * M * D;
        T col = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)start[l];
            for (int p = 0; p < P; ++p) {
                const long si = item * L * P + l * P + p;
                const Sample<T> s = sample_setup<T>(loc[2 * si], loc[2 * si + 1], H, W, st, M, D, m);
                if (!s.valid) continue;
                const T v1 = s.o1 >= 0 ? vb[s.o1 + c] : (T)0;
                const T v2 = s.o2 >= 0 ? vb[s.o2 + c] : (T)0;
                const T v3 = s.o3 >= 0 ? vb[s.o3 + c] : (T)0;
                const T v4 = s.o4 >= 0 ? vb[s.o4 + c] : (T)0;
                col += (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * attn[si];
            }
        }
        out[idx] = col;
    }
}

// grad_loc / grad_attn must be zero on entry (they are reduced over the D channel threads
// with atomics); the C entry point zero-fills all three outputs.
template <typename T>
__global__ __launch_bounds__(kBlock) void msda_bwd_generic(
    const T *__restrict__ grad_out, const T *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
    const T *__restrict__ loc, const T *__restrict__ attn, int S, int M, int D, int L, int Q,
    int P, long n, T *__restrict__ grad_value, T *__restrict__ grad_loc,
    T *__restrict__ grad_attn)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % D);
        const long item = idx / D;
        const int m = (int)(item % M);
        const int b = (int)(item / ((long)M * Q));
        const size_t boff = (size_t)b * S * M * D;
        const T *vb = value + boff;
        T *gvb = grad_value + boff;
        const T top = grad_out[idx];
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)start[l];
            for (int p = 0; p < P; ++p) {
                const long si = item * L * P + l * P + p;
                const Sample<T> s = sample_setup<T>(loc[2 * si], loc[2 * si + 1], H, W, st, M, D, m);
                if (!s.valid) continue;
                const T a = attn[si], tgv = top * a;
                const T hhw = 1 - s.lh, hww = 1 - s.lw;
                T v1 = 0, v2 = 0, v3 = 0, v4 = 0, gh = 0, gw = 0;
                if (s.o1 >= 0) { v1 = vb[s.o1 + c]; gh -= hww * v1; gw -= hhw * v1;
                                 unsafeAtomicAdd(gvb + s.o1 + c, s.w1 * tgv); }
                if (s.o2 >= 0) { v2 = vb[s.o2 + c]; gh -= s.lw * v2; gw += hhw * v2;
                                 unsafeAtomicAdd(gvb + s.o2 + c, s.w2 * tgv); }
                if (s.o3 >= 0) { v3 = vb[s.o3 + c]; gh += hww * v3; gw -= s.lh * v3;
                                 unsafeAtomicAdd(gvb + s.o3 + c, s.w3 * tgv); }
                if (s.o4 >= 0) { v4 = vb[s.o4 + c]; gh += s.lw * v4; gw += s.lh * v4;
                                 unsafeAtomicAdd(gvb + s.o4 + c, s.w4 * tgv); }
                const T val = s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
                unsafeAtomicAdd(grad_attn + si, top * val);
                unsafeAtomicAdd(grad_loc + 2 * si, (T)W * gw * tgv);
                unsafeAtomicAdd(grad_loc + 2 * si + 1, (T)H * gh * tgv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline int lpr_for(int D)
{
    switch (D) {
        case 4: return 1;
        case 8: return 2;
        case 16: return 4;
        case 32: return 8;
        case 64: return 16;
        case 128: return 32;
        case 256: return 64;
        default: return 0;
    }
}

inline bool args_ok(const void *value, const void *shapes, const void *start, const void *loc,
                    const void *attn, int B, int S, int M, int D, int L, int Q, int P)
{
    if (!value || !shapes || !start || !loc || !attn) return false;
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0) return false;
    const long long per_value = (long long)S * M * D;
    const long long per_loc = (long long)Q * M * L * P * 2;
    if (per_value >= (1LL << 31) || per_loc >= (1LL << 31)) return false;
    return true;
}

// the lean kernels use 32-bit byte offsets inside a batch element and 32-bit item counts
inline bool lean_ok(int B, int S, int M, int D, int L, int Q, int P)
{
    return (long long)S * M * D * 4 < (1LL << 32) && (long long)B * Q * M < (1LL << 31) &&
           (long long)L * P < (1 << 20);
}

// mul, shift with (n * mul) >> shift == n / d for every n < 2^31:
// s = ceil(log2 d), mul = floor(2^(31+s) / d) + 1 (< 2^32), shift = 31 + s.

inline int generic_grid(long n)
{
    long blocks = (n + kBlock - 1) / kBlock;
    const long cap = 256L * 8;  // 256 CUs x 8 blocks, grid-stride beyond
    if (blocks > cap) blocks = cap;
    return (int)blocks;
}

template <int LPR>
int launch_fwd_rows(const float *value, const int64_t *shapes, const int64_t *start,
                    const float *loc, const float *attn, int B, int S, int M, int L, int Q, int P,
                    float *out, hipStream_t st)
{
    const long nitems = (long)B * Q * M;
    const int grid = head_major_grid(nitems);
    hipLaunchKernelGGL(msda_fwd_rows<LPR>, dim3(grid), dim3(kBlock), 0, st, value, shapes, start,
                       loc, attn, S, M, L, Q, P, nitems, out);
    return (int)hipGetLastError();
}

template <int CQR>
int launch_fwd_lean(const float *value, const int64_t *shapes, const int64_t *start,
                    const float *loc, const float *attn, int B, int S, int M, int L, int Q, int P,
                    float *out, hipStream_t st)
{
    const unsigned nitems = (unsigned)B * Q * M;
    hipLaunchKernelGGL(msda_fwd_lean<CQR>, dim3(head_major_grid(nitems)), dim3(kBlock), 0, st,
                       value, shapes, start, loc, attn, (unsigned)S, make_fast_div((unsigned)M),
                       (unsigned)(L * P), make_fast_div((unsigned)Q), 1.0f / (float)P, nitems,
                       (nitems + 7) >> 3, out);
    return (int)hipGetLastError();
}

template <int CQR>
int launch_bwd_lean_atomic(const float *grad_out, const float *value, const int64_t *shapes,
                           const int64_t *start, const float *loc, const float *attn, int B, int S,
                           int M, int L, int Q, int P, float *gv, float *gl, float *ga,
                           hipStream_t st)
{
    const unsigned nitems = (unsigned)B * Q * M;
    hipLaunchKernelGGL((msda_bwd_lean_atomic<CQR, true>), dim3(head_major_grid(nitems)), dim3(kBlock), 0,
                       st, grad_out, value, shapes, start, loc, attn, (unsigned)S,
                       make_fast_div((unsigned)M), (unsigned)(L * P), make_fast_div((unsigned)Q),
                       1.0f / (float)P, nitems, (nitems + 7) >> 3, gv, gl, ga);
    return (int)hipGetLastError();
}

// grad_sampling_loc and grad_attn_weight alone (the planned sparse backward's gather half)
template <int CQR>
int launch_bwd_home(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start,
                    const float *loc, const float *attn, int B, int S, int M, int L, int Q, int P, float *gl, float *ga,
                    hipStream_t st)
{
    const unsigned nitems = (unsigned)B * Q * M;
    hipLaunchKernelGGL((msda_bwd_lean_atomic<CQR, false>), dim3(head_major_grid(nitems)), dim3(kBlock), 0,
                       st, grad_out, value, shapes, start, loc, attn, (unsigned)S,
                       make_fast_div((unsigned)M), (unsigned)(L * P), make_fast_div((unsigned)Q),
                       1.0f / (float)P, nitems, (nitems + 7) >> 3, static_cast<float *>(nullptr), gl, ga);
    return (int)hipGetLastError();
}

template <int LPR>
int launch_bwd_rows(const float *grad_out, const float *value, const int64_t *shapes,
                    const int64_t *start, const float *loc, const float *attn, int B, int S, int M,
                    int L, int Q, int P, float *gv, float *gl, float *ga, hipStream_t st)
{
    const long nitems = (long)B * Q * M;
    const int grid = head_major_grid(nitems);
    hipLaunchKernelGGL(msda_bwd_rows_atomic<LPR>, dim3(grid), dim3(kBlock), 0, st, grad_out, value,
                       shapes, start, loc, attn, S, M, L, Q, P, nitems, gv, gl, ga);
    return (int)hipGetLastError();
}

// ---- tiled backward: plan, workspace layout, launch ---------------------------------------

inline unsigned device_cu_count()
{
    static unsigned cus = 0;  // one device per process (one process per GPU)
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;  // MI355X
        cus = (unsigned)n;
    }
    return cus;
}

inline bool make_tile_plan(int B, int S, int M, int D, int L, int Q, int P, TilePlan &p)
{
    if (!(D == 16 || D == 32 || D == 64) || !lean_ok(B, S, M, D, L, Q, P)) return false;
    const unsigned LP = (unsigned)L * P;
    p.chunks = (LP + 15) / 16;
    const unsigned heads = (unsigned)B * M;
    if ((unsigned long long)heads * Q >= 16 * 4096) return false;   // dense calls: csrc/msda_cells.hip (or the atomic kernels)
    p.ipb = 4u * kItemsPerWave;    // small K1 blocks: balance across the chip for a few hundred queries
    p.eblk = p.ipb * p.chunks * 64;
    if (p.eblk >= 65536) return false;
    // K2: a wave per tile of <= kWaveTileRows rows, as many tiles as wave slots (ZIRA_K2W_MINWAVES per SIMD) so that the
    // grid is one round (the per-tile critical path is a chain of dependent memory round trips: rounds cost)
    p.wave_k2 = 1;
    p.runlist = 0;
    p.run_base = 0;
    const unsigned t_min = ((unsigned)S + kWaveTileRows - 1) / kWaveTileRows;
    const unsigned t_want = (device_cu_count() * 4 * ZIRA_K2W_MINWAVES) / (heads * (unsigned)L);
    p.T = t_want > t_min ? t_want : t_min;
    if (p.T > (unsigned)S) p.T = (unsigned)S;
    p.NT = (unsigned)L * p.T;
    p.nblk = ((unsigned)Q + p.ipb - 1) / p.ipb;
    p.rows = ((unsigned)S + p.T - 1) / p.T;
    const size_t per_wave = (size_t)wave_meta_words(p.rows, p.nblk) + kWaveSortWords;
    // a heavy tile publishes ceil(N / kSliceEntries) - 1 slices: at most (all entries) / kSliceEntries in total
    const unsigned long long all_entries = (unsigned long long)heads * Q * L * P * 4;
    p.qwords = kQueueHeader + (unsigned)(all_entries / kSliceEntries) + 1;
    return p.NT <= 4096 && per_wave * 4 * kWaveK2Waves <= 64 * 1024 && Q < (1 << 20) &&
           (unsigned long long)heads * p.nblk * p.eblk < (1ull << 32) &&
           (unsigned long long)heads * p.NT < (1ull << (32 - kQueueSliceBits)) &&
           ((size_t)p.NT + (size_t)p.eblk * 5) * 4 <= 150 * 1024;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline size_t tile_desc_bytes(const TilePlan &p, int B, int M)
{
    if (p.runlist)  // run counters, then up to nblk 8-byte records per tile (sparsely touched)
        return (size_t)p.run_base * sizeof(unsigned) + align256((size_t)B * M * p.NT * p.nblk * sizeof(uint2));
    return align256((size_t)B * M * p.NT * p.nblk * sizeof(unsigned));
}

inline size_t tile_region_bytes(const TilePlan &p, int B, int M)
{
    return align256((size_t)B * M * p.nblk * p.eblk * sizeof(uint2));
}

inline size_t tile_workspace_bytes(const TilePlan &p, int B, int M)
{
    return tile_desc_bytes(p, B, M) + tile_region_bytes(p, B, M) + (size_t)p.qwords * sizeof(unsigned);
}

template <int CQR>
int launch_bwd_tiled(const TilePlan &p, const float *grad_out, const float *value,
                     const int64_t *shapes, const int64_t *start, const float *loc,
                     const float *attn, int B, int S, int M, int L, int Q, int P, float *gv,
                     float *gl, float *ga, void *ws, hipStream_t st)
{
    constexpr int D = 16 * CQR;
    unsigned *desc = reinterpret_cast<unsigned *>(ws);
    uint2 *region = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(ws) + tile_desc_bytes(p, B, M));
    unsigned *queue = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(region) + tile_region_bytes(p, B, M));
    const unsigned heads = (unsigned)B * M;
    const FastDiv Mdiv = make_fast_div((unsigned)M), Tdiv = make_fast_div(p.T);

    const unsigned nv1 = heads * p.nblk, per1 = (nv1 + 7) >> 3;
    const size_t lds1 = ((size_t)p.NT + (size_t)p.eblk * 5) * 4;
    if (lds1 > 64 * 1024) {  // opt in to more than 64 KB of dynamic LDS (LP > 16 only)
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_items<CQR, 4>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        if (ea != hipSuccess) return (int)ea;
    }
    hipLaunchKernelGGL((msda_bwd_items<CQR, 4>), dim3(per1 * 8), dim3(4 * 64), lds1, st,
                       grad_out, value, shapes, start, loc, attn, (unsigned)S, Mdiv,
                       (unsigned)(L * P), 1.0f / (float)P, (unsigned)Q, make_fast_div(p.nblk),
                       nv1, per1, Tdiv, p, gl, ga, desc, region, queue);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;

    const unsigned nvw = heads * p.NT, perw = (nvw + 7) >> 3;
    const unsigned blocks_per_xcd = (perw + kWaveK2Waves - 1) / kWaveK2Waves;
    const size_t per_wave = (size_t)wave_meta_words(p.rows, p.nblk) + kWaveSortWords;
    hipLaunchKernelGGL((msda_bwd_tiles_wave<D, false, false>), dim3(blocks_per_xcd * 8), dim3(kWaveK2Waves * 64),
                       per_wave * 4 * kWaveK2Waves, st, grad_out, shapes, start, (unsigned)S, Mdiv,
                       (unsigned)Q, nvw, perw, Tdiv, make_fast_div(p.NT), p, desc, region, queue, gv);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // slices 1.. of heavy tiles (ends at once when there are none)
    hipLaunchKernelGGL((msda_bwd_tiles_wave<D, true, false>), dim3(kWaveHelperBlocks), dim3(kWaveK2Waves * 64),
                       per_wave * 4 * kWaveK2Waves, st, grad_out, shapes, start, (unsigned)S, Mdiv,
                       (unsigned)Q, nvw, perw, Tdiv, make_fast_div(p.NT), p, desc, region, queue, gv);
    return (int)hipGetLastError();
}

template <typename T>
int fwd_generic(const T *value, const int64_t *shapes, const int64_t *start, const T *loc,
                const T *attn, int B, int S, int M, int D, int L, int Q, int P, T *out,
                hipStream_t st)
{
    const long n = (long)B * Q * M * D;
    hipLaunchKernelGGL(msda_fwd_generic<T>, dim3(generic_grid(n)), dim3(kBlock), 0, st, value,
                       shapes, start, loc, attn, S, M, D, L, Q, P, n, out);
    return (int)hipGetLastError();
}

// Zero-fill on the caller's stream by a KERNEL: hipMemsetAsync becomes a memset node when the stream is being captured,
// and ROCm 7.2 replays such a node out of order with the kernels around it (found in round 4, scripts/repro_memset_graph.py).
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4 *__restrict__ p, size_t n16, unsigned char *__restrict__ tail, unsigned ntail)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
inline hipError_t zero_fill_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    unsigned char *b = reinterpret_cast<unsigned char *>(ptr);
    const size_t head = ((uintptr_t)b & 15) ? 16 - ((uintptr_t)b & 15) : 0;
    if (head >= bytes || bytes < 64) {   // tiny: bytes by one block
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (uint4 *)nullptr, (size_t)0, b, (unsigned)(bytes < 256 ? bytes : 0));
        if (bytes >= 256) return hipErrorInvalidValue;   // (unreachable: < 64 here)
        return hipGetLastError();
    }
    if (head) {   // unaligned start (never the case for torch / hipMalloc storage): the first bytes on their own
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, st, (uint4 *)nullptr, (size_t)0, b, (unsigned)head);
        b += head;
        bytes -= head;
    }
    const size_t n16 = bytes / 16;
    const unsigned ntail = (unsigned)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint4 *>(b), n16, b + n16 * 16, ntail);
    return hipGetLastError();
}

template <typename T>
int bwd_generic(const T *grad_out, const T *value, const int64_t *shapes, const int64_t *start,
                const T *loc, const T *attn, int B, int S, int M, int D, int L, int Q, int P,
                T *gv, T *gl, T *ga, hipStream_t st)
{
    const long n = (long)B * Q * M * D;
    const size_t nsamp = (size_t)B * Q * M * L * P;
    hipError_t e = zero_fill_async(gv, sizeof(T) * (size_t)B * S * M * D, st);
    if (e != hipSuccess) return (int)e;
    e = zero_fill_async(gl, sizeof(T) * nsamp * 2, st);
    if (e != hipSuccess) return (int)e;
    e = zero_fill_async(ga, sizeof(T) * nsamp, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(msda_bwd_generic<T>, dim3(generic_grid(n)), dim3(kBlock), 0, st, grad_out,
                       value, shapes, start, loc, attn, S, M, D, L, Q, P, n, gv, gl, ga);
    return (int)hipGetLastError();
}

}  // namespace

namespace {
// The planned sparse backward (D = 32): grad_sampling_loc / grad_attn_weight from a gather pass like the forward's (a wave per
// (b, q, m); every sample, also those outside the window, whose gradients are zero), grad_value from the plan's tiles -- one launch.
int planned_backward(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start, const float *loc,
                     const float *attn, int B, int S, int M, int D, int L, int Q, int P, float *gv, float *gl, float *ga,
                     void *plan, size_t plan_bytes, hipStream_t st)
{
    if (D != 32 || !lean_ok(B, S, M, D, L, Q, P)) return -1;
    return zira::tiles_backward_planned_f32(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P, gv, gl, ga, plan,
                                            plan_bytes, st);
}
}  // namespace

extern "C" {

#ifndef ZIRA_FWD_PATCH
#define ZIRA_FWD_PATCH 0   // 1 (developer builds: scripts/build_variant.sh with EXTRA_SRC=dev/msda_patch.hip): dense calls with Q = S and D = 32
                           // take the LDS-patch forward of csrc/dev/msda_patch.hip -- correct, and slower than the lean kernel
#endif

int zira_msda_fwd_f32(const float *value, const int64_t *shapes, const int64_t *start,
                      const float *loc, const float *attn, int B, int S, int M, int D, int L,
                      int Q, int P, float *out, void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !out)
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#if ZIRA_FWD_PATCH
    if (D == 32 && Q == S && (unsigned long long)B * M * Q >= 16 * 4096) {   // every pixel is a query (the encoder): LDS patches
        const int rc = zira::patch_forward_f32(value, shapes, start, loc, attn, B, S, M, D, L, Q, P, out, st);
        if (rc != -1) return rc;
    }
#endif
    if (lean_ok(B, S, M, D, L, Q, P)) {
        if (D == 16) return launch_fwd_lean<1>(value, shapes, start, loc, attn, B, S, M, L, Q, P, out, st);
        if (D == 32) return launch_fwd_lean<2>(value, shapes, start, loc, attn, B, S, M, L, Q, P, out, st);
        if (D == 64) return launch_fwd_lean<4>(value, shapes, start, loc, attn, B, S, M, L, Q, P, out, st);
    }
#define ZIRA_FWD_CASE(LPR_)                                                                      \
    case LPR_:                                                                                   \
        return launch_fwd_rows<LPR_>(value, shapes, start, loc, attn, B, S, M, L, Q, P, out, st);
    switch (lpr_for(D)) {
        ZIRA_FWD_CASE(1) ZIRA_FWD_CASE(2) ZIRA_FWD_CASE(4) ZIRA_FWD_CASE(8) ZIRA_FWD_CASE(16)
        ZIRA_FWD_CASE(32) ZIRA_FWD_CASE(64)
        default: break;
    }
#undef ZIRA_FWD_CASE
    return fwd_generic<float>(value, shapes, start, loc, attn, B, S, M, D, L, Q, P, out, st);
}

int zira_msda_bwd_f32(const float *grad_out, const float *value, const int64_t *shapes,
                      const int64_t *start, const float *loc, const float *attn, int B, int S,
                      int M, int D, int L, int Q, int P, float *gv, float *gl, float *ga,
                      void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !grad_out || !gv ||
        !gl || !ga)
        return ZIRA_MSDA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int lpr = lpr_for(D);
    if (lpr == 0)
        return bwd_generic<float>(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P,
                                  gv, gl, ga, st);
    hipError_t e = zero_fill_async(gv, sizeof(float) * (size_t)B * S * M * D, st);
    if (e != hipSuccess) return (int)e;
    if (lean_ok(B, S, M, D, L, Q, P)) {
        if (D == 16) return launch_bwd_lean_atomic<1>(grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, gv, gl, ga, st);
        if (D == 32) return launch_bwd_lean_atomic<2>(grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, gv, gl, ga, st);
        if (D == 64) return launch_bwd_lean_atomic<4>(grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, gv, gl, ga, st);
    }
#define ZIRA_BWD_CASE(LPR_)                                                                      \
    case LPR_:                                                                                   \
        return launch_bwd_rows<LPR_>(grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, \
                                     gv, gl, ga, st);
    switch (lpr) {
        ZIRA_BWD_CASE(1) ZIRA_BWD_CASE(2) ZIRA_BWD_CASE(4) ZIRA_BWD_CASE(8) ZIRA_BWD_CASE(16)
        ZIRA_BWD_CASE(32) ZIRA_BWD_CASE(64)
        default: break;
    }
#undef ZIRA_BWD_CASE
    return ZIRA_MSDA_EINVAL;
}

// Which workspace backward serves a call: the cell kernels (csrc/msda_cells.hip: bin + LDS accumulate for D = 32,
// bin + walk for D = 16 / 64) for dense calls (encoder self-attention: every pixel is a query), plan + tile accumulate
// (csrc/msda_tiles.hip) or the entry sort below for sparse ones (decoder cross-attention).
static bool use_cells_path(int B, int M, int Q)
{
    return (unsigned long long)B * M * Q >= 16 * 4096;
}

#ifndef ZIRA_SPARSE_TILES
#define ZIRA_SPARSE_TILES 1   // sparse calls with D = 32: plan + tile accumulate (csrc/msda_tiles.hip); 0: the round-2 entry sort
#endif

size_t zira_msda_bwd_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P)
{
    TilePlan p;
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0) return 0;
    if (use_cells_path(B, M, Q)) {
        const size_t n = zira::cells_workspace_bytes(B, S, M, D, L, Q, P);
        if (n) return n;
    } else if (ZIRA_SPARSE_TILES) {
        const size_t n = zira::tiles_plan_bytes(B, S, M, D, L, Q, P);
        if (n) return n;
    }
    if (!make_tile_plan(B, S, M, D, L, Q, P, p)) return 0;
    return tile_workspace_bytes(p, B, M);
}

int zira_msda_bwd_f32_ws(const float *grad_out, const float *value, const int64_t *shapes,
                         const int64_t *start, const float *loc, const float *attn, int B, int S,
                         int M, int D, int L, int Q, int P, float *gv, float *gl, float *ga,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !grad_out || !gv ||
        !gl || !ga)
        return ZIRA_MSDA_EINVAL;
    if (workspace && use_cells_path(B, M, Q) && !((uintptr_t)workspace & 15)) {
        const size_t need = zira::cells_workspace_bytes(B, S, M, D, L, Q, P);
        if (need && workspace_bytes >= need)
            return zira::cells_backward_f32(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P,
                                            gv, gl, ga, workspace, workspace_bytes, (hipStream_t)stream);
    }
    if (ZIRA_SPARSE_TILES && workspace && !use_cells_path(B, M, Q) && !((uintptr_t)workspace & 15)) {
        // no plan from the forward pass: plan here, in front of the accumulate kernel (the workspace is the plan buffer)
        const size_t need = zira::tiles_plan_bytes(B, S, M, D, L, Q, P);
        if (need && workspace_bytes >= need) {
            int rc = zira::tiles_plan_f32(shapes, start, loc, attn, B, S, M, D, L, Q, P, workspace, workspace_bytes, (hipStream_t)stream);
            if (rc == 0) rc = planned_backward(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P, gv, gl, ga, workspace,
                                               workspace_bytes, (hipStream_t)stream);
            if (rc != -1) return rc;
        }
    }
    TilePlan p;
    if (!workspace || !make_tile_plan(B, S, M, D, L, Q, P, p) ||
        workspace_bytes < tile_workspace_bytes(p, B, M) || ((uintptr_t)workspace & 15))
        return zira_msda_bwd_f32(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P,
                                 gv, gl, ga, stream);
    hipStream_t st = (hipStream_t)stream;
    if (D == 16) return launch_bwd_tiled<1>(p, grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, gv, gl, ga, workspace, st);
    if (D == 32) return launch_bwd_tiled<2>(p, grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, gv, gl, ga, workspace, st);
    return launch_bwd_tiled<4>(p, grad_out, value, shapes, start, loc, attn, B, S, M, L, Q, P, gv, gl, ga, workspace, st);
}

size_t zira_msda_plan_bytes(int B, int S, int M, int D, int L, int Q, int P)
{
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0) return 0;
    if (!ZIRA_SPARSE_TILES || use_cells_path(B, M, Q)) return 0;
    return zira::tiles_plan_bytes(B, S, M, D, L, Q, P);
}

int zira_msda_plan_f32(const int64_t *shapes, const int64_t *start, const float *loc, const float *attn, int B, int S, int M,
                       int D, int L, int Q, int P, void *plan, size_t plan_bytes, void *stream)
{
    if (!shapes || !start || !loc || !attn || !plan || B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0)
        return ZIRA_MSDA_EINVAL;
    const size_t need = zira_msda_plan_bytes(B, S, M, D, L, Q, P);
    if (!need || plan_bytes < need || ((uintptr_t)plan & 15)) return ZIRA_MSDA_EINVAL;
    const int rc = zira::tiles_plan_f32(shapes, start, loc, attn, B, S, M, D, L, Q, P, plan, plan_bytes, (hipStream_t)stream);
    return rc == -1 ? ZIRA_MSDA_EINVAL : rc;
}

int zira_msda_fwd_plan_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc,
                           const float *attn, int B, int S, int M, int D, int L, int Q, int P, float *out, void *plan,
                           size_t plan_bytes, void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !out || !plan) return ZIRA_MSDA_EINVAL;
    const size_t need = zira_msda_plan_bytes(B, S, M, D, L, Q, P);
    if (!need || plan_bytes < need || ((uintptr_t)plan & 15)) return ZIRA_MSDA_EINVAL;
    if (lean_ok(B, S, M, D, L, Q, P)) {   // one launch: the plan's blocks beside the gather's
        const int rc = zira::tiles_fwd_plan_f32(value, shapes, start, loc, attn, B, S, M, D, L, Q, P, out, plan, plan_bytes,
                                                (hipStream_t)stream);
        if (rc != -1) return rc;
    }
    const int rc = zira_msda_fwd_f32(value, shapes, start, loc, attn, B, S, M, D, L, Q, P, out, stream);
    if (rc != 0) return rc;
    return zira_msda_plan_f32(shapes, start, loc, attn, B, S, M, D, L, Q, P, plan, plan_bytes, stream);
}

int zira_msda_bwd_planned_f32(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start,
                              const float *loc, const float *attn, int B, int S, int M, int D, int L, int Q, int P,
                              float *gv, float *gl, float *ga, void *plan, size_t plan_bytes, void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !grad_out || !gv || !gl || !ga || !plan)
        return ZIRA_MSDA_EINVAL;
    const size_t need = zira_msda_plan_bytes(B, S, M, D, L, Q, P);
    if (!need || plan_bytes < need || ((uintptr_t)plan & 15)) return ZIRA_MSDA_EINVAL;
    const int rc = planned_backward(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P, gv, gl, ga, plan, plan_bytes,
                                    (hipStream_t)stream);
    return rc == -1 ? ZIRA_MSDA_EINVAL : rc;
}

int zira_msda_fwd_f64(const double *value, const int64_t *shapes, const int64_t *start,
                      const double *loc, const double *attn, int B, int S, int M, int D, int L,
                      int Q, int P, double *out, void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !out)
        return ZIRA_MSDA_EINVAL;
    return fwd_generic<double>(value, shapes, start, loc, attn, B, S, M, D, L, Q, P, out,
                               (hipStream_t)stream);
}

int zira_msda_bwd_f64(const double *grad_out, const double *value, const int64_t *shapes,
                      const int64_t *start, const double *loc, const double *attn, int B, int S,
                      int M, int D, int L, int Q, int P, double *gv, double *gl, double *ga,
                      void *stream)
{
    if (!args_ok(value, shapes, start, loc, attn, B, S, M, D, L, Q, P) || !grad_out || !gv ||
        !gl || !ga)
        return ZIRA_MSDA_EINVAL;
    return bwd_generic<double>(grad_out, value, shapes, start, loc, attn, B, S, M, D, L, Q, P, gv,
                               gl, ga, (hipStream_t)stream);
}

const char *zira_msda_version(void) { return "zira_msda 0.1 gfx950"; }

const char *zira_msda_variant_f32(int D)
{
    // D = 16 / 32 / 64 take the lean kernels (and, with a workspace, the tiled backward) whenever
    // the call passes lean_ok(); the other specialised widths use the row-per-group kernels
    if (D == 16 || D == 32 || D == 64)
        return D == 32 ? "fwd msda_fwd_lean (sparse calls that need gradients: msda_fwd_plan = forward + the backward's plan in one "
                         "launch); bwd with workspace: msda_bwd_bin + msda_bwd_accum + msda_bwd_fold (dense calls) / "
                         "[msda_plan unless planned by the forward] + msda_bwd_tile_accum + msda_bwd_fold (sparse calls); "
                         "without: msda_bwd_lean_atomic"
                       : "fwd msda_fwd_lean; bwd with workspace: msda_bwd_bin + msda_bwd_walk + msda_bwd_fold (dense calls) / "
                         "msda_bwd_items + msda_bwd_tiles_wave (sparse calls); without: msda_bwd_lean_atomic";
    switch (lpr_for(D)) {
        case 1: return "rows<1>";
        case 2: return "rows<2>";
        case 4: return "rows<4>";
        case 8: return "rows<8>";
        case 16: return "rows<16>";
        case 32: return "rows<32>";
        case 64: return "rows<64>";
        default: return "generic";
    }
}

}  // extern "C"
