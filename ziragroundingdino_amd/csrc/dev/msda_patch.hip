// dev/msda_patch.hip -- NOT in the shipped library (developer builds only: EXTRA_SRC=dev/msda_patch.hip scripts/build_variant.sh
// patch -DZIRA_FWD_PATCH=1).  Forward of multi-scale deformable attention for gfx950 when EVERY PIXEL IS A QUERY (the encoder's
// self-attention: Q = S, D = 32): value patches staged in LDS, gathers served from LDS.  Round 5's answer to the review's
// "build the row-sharing forward and report the measurement": correct (the 64 oracle / shape cases of tests/test_msda_gpu.py
// pass with it) and SLOWER than the lean kernel -- 227-249 us against 172 us at the encoder shape; by ablation 35 us for the
// sample table, 11 us for its box atomics, 82 us for staging (one exposed round trip per item: a CU holds ONE block of 140 KB)
// and 125 us for the gather (its index arithmetic is done by all eight lanes of a query; the LDS floor of the 2.9 GB is 42 us).
//
// Arithmetic to match: reference csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:237-299 + :33-84 (bilinear sample of the four
// corner rows, weighted by the attention weight, summed over levels and points).  The decomposition is not the reference's.
//
// Why.  The lean forward (csrc/msda.hip, msda_fwd_lean: a wave per (b, q, m), 64 row gathers of 128 bytes each) is bound by
// the rate at which a CU's L1 takes rows from L2: 5.7 M samples x 4 corners x 128 B = 2.9 GB per encoder call, 172-186 us.  In
// the encoder the queries ARE the pixels, and a query samples around its own position on every level: the 128 queries of an
// 8 x 16 pixel tile touch a few hundred distinct pixels per level (the tile plus the reach of the offsets, scaled to the
// level), ~6 value rows per query instead of 64.  So the unit of work is a TILE OF QUERIES of one level and one head:
//   1. a thread per (query, sample) derives the sample's top-left pixel and its four corner weights (times the attention
//      weight; zero for corners outside the level) into an LDS table and extends the level's bounding box of touched pixels
//      (LDS atomic min / max) -- the locations are DATA: nothing is assumed about them;
//   2. every level whose box fits what is left of the LDS budget is staged: its value rows (this head's 128 bytes per pixel)
//      are copied into LDS once, all loads in flight together; a level that does not fit (queries of a coarse level looking
//      at a fine one, or locations that are not local at all) stays in global memory;
//   3. eight lanes (four channels each) own a query and walk its samples: the table entry by LDS broadcast, the four corner
//      rows from the patch (or, for an unstaged level, from global memory as the lean kernel does), 16 FMAs.  A lane
//      accumulates its own query's channels: no cross-lane reduction.  The level of a step is wave-uniform (all slots of a
//      wave are at the same sample index), so staged / unstaged is a uniform branch.
// Blocks are persistent (one per CU: 140 KB of LDS) and walk the (head, tile) items of their XCD's heads, so that a head's
// value slice stays in one L2.  The level table is read on the device (the C ABI has device pointers only).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../msda_internal.h"

#ifndef ZIRA_PATCH_DEV_SKIP
#define ZIRA_PATCH_DEV_SKIP 0   // developer ablation (wrong results): 1 no box atomics, 2 no staging, 4 no gather
#endif
#ifndef ZIRA_PATCH_ROWS
#define ZIRA_PATCH_ROWS 800   // value rows (128 bytes each) the patches of a block may hold
#endif

namespace {

constexpr int TQH = 8, TQW = 16, NQ = TQH * TQW;   // queries per tile
constexpr int NTHR = 512, MAXLP = 16, MAXL = 8;
constexpr int kD = 32;

struct Desc {
    int ymin, xmin, h, w, base, staged, H, W;   // bounding box of the touched pixels, patch base (rows), level size
};

struct LevelInfo {
    int H, W, start, tiles_x, tile_base;
};

__device__ __forceinline__ float4 fma4(float w, const float4 v, float4 a)
{
    a.x = fmaf(w, v.x, a.x);
    a.y = fmaf(w, v.y, a.y);
    a.z = fmaf(w, v.z, a.z);
    a.w = fmaf(w, v.w, a.w);
    return a;
}

__global__ __launch_bounds__(NTHR) void msda_fwd_patch(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                                                        const int64_t *__restrict__ start, const float *__restrict__ loc,
                                                        const float *__restrict__ attn, int B, int S, int M, int L, int P,
                                                        float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tw = reinterpret_cast<float4 *>(smem);                      // [NQ * MAXLP] corner weights (w00, w01, w10, w11)
    int *tc = reinterpret_cast<int *>(smem + NQ * MAXLP * 16);          // [NQ * MAXLP] top-left pixel: y0 << 16 | (x0 & 0xffff)
    float *patch = reinterpret_cast<float *>(smem + NQ * MAXLP * 20);   // [ZIRA_PATCH_ROWS][32]
    __shared__ LevelInfo lv[MAXL];
    __shared__ Desc desc[MAXL];
    __shared__ int box[MAXL][4];     // ymin, ymax, xmin, xmax
    __shared__ int ntiles;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LP = L * P, Q = S;
    if (tid == 0) {
        int t = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            lv[l].H = H;
            lv[l].W = W;
            lv[l].start = (int)start[l];
            lv[l].tiles_x = (W + TQW - 1) / TQW;
            lv[l].tile_base = t;
            t += ((H + TQH - 1) / TQH) * lv[l].tiles_x;
        }
        ntiles = t;
    }
    __syncthreads();
    const int T = ntiles;
    // this block's items: the heads of its XCD (blocks b, b + 8, ... share an XCD), every tile of them
    const int heads = B * M, hpx = (heads + 7) / 8, xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int h_lo = xcd * hpx, h_hi = min(heads, h_lo + hpx);
    const int nitems = max(0, h_hi - h_lo) * T;

    for (int item = j0; item < nitems; item += nj) {
        const int head = h_lo + item / T, tile = item % T;
        const int b = head / M, m = head - b * M;
        int lq = 0;
        for (int l = 1; l < L; ++l) lq = tile >= lv[l].tile_base ? l : lq;
        const int tl = tile - lv[lq].tile_base, ty0 = (tl / lv[lq].tiles_x) * TQH, tx0 = (tl % lv[lq].tiles_x) * TQW;
        const int Hq = lv[lq].H, Wq = lv[lq].W, stq = lv[lq].start;
        __syncthreads();      // (the previous item's table and patches are no longer read)
        if (tid < L) {
            box[tid][0] = 1 << 30;
            box[tid][1] = -1;
            box[tid][2] = 1 << 30;
            box[tid][3] = -1;
        }
        __syncthreads();
        // ---- 1. the sample table and the levels' bounding boxes -----------------------------------------------------------
        {
#pragma clang fp contract(off)
            // (all of a thread's locations and weights are requested before any is used: clamped addresses, no branches)
            constexpr int NS = NQ * MAXLP / NTHR;
            float2 xy[NS];
            float aw[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int i = min(tid + j * NTHR, NQ * LP - 1);
                const int qi = i / LP, s = i - qi * LP;
                const int qy = min(ty0 + qi / TQW, Hq - 1), qx = min(tx0 + qi % TQW, Wq - 1);
                const size_t at = ((size_t)((size_t)b * Q + stq + qy * Wq + qx) * M + m) * LP + s;
                xy[j] = *reinterpret_cast<const float2 *>(loc + 2 * at);
                aw[j] = attn[at];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int i = tid + j * NTHR;
                if (i >= NQ * LP) break;
                const int qi = i / LP, s = i - qi * LP;
                const int qy = ty0 + qi / TQW, qx = tx0 + qi % TQW;
                float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                int yx = 0;
                if (qy < Hq && qx < Wq) {
                    const int l = s / P;
                    const float a = aw[j];
                    const int H = lv[l].H, W = lv[l].W;
                    const float Hf = (float)H, Wf = (float)W;
                    const float h_im = xy[j].y * Hf - 0.5f, w_im = xy[j].x * Wf - 0.5f;
                    if (h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf) {
                        const float hf = floorf(h_im), wf = floorf(w_im);
                        const float lh = h_im - hf, lw = w_im - wf;
                        const int y0 = (int)hf, x0 = (int)wf;
                        const bool y0ok = y0 >= 0, y1ok = y0 + 1 < H, x0ok = x0 >= 0, x1ok = x0 + 1 < W;
                        w4.x = (y0ok && x0ok) ? ((1.f - lh) * (1.f - lw)) * a : 0.f;
                        w4.y = (y0ok && x1ok) ? ((1.f - lh) * lw) * a : 0.f;
                        w4.z = (y1ok && x0ok) ? (lh * (1.f - lw)) * a : 0.f;
                        w4.w = (y1ok && x1ok) ? (lh * lw) * a : 0.f;
                        yx = (int)(((unsigned)y0 << 16) | ((unsigned)x0 & 0xffffu));
#if ZIRA_PATCH_DEV_SKIP & 1
                        if (i == 0) { box[l][0] = 0; box[l][1] = 9; box[l][2] = 0; box[l][3] = 9; }
#else
                        atomicMin(&box[l][0], y0ok ? y0 : 0);
                        atomicMax(&box[l][1], y1ok ? y0 + 1 : H - 1);
                        atomicMin(&box[l][2], x0ok ? x0 : 0);
                        atomicMax(&box[l][3], x1ok ? x0 + 1 : W - 1);
#endif
                    }
                }
                tw[i] = w4;
                tc[i] = yx;
            }
        }
        __syncthreads();
        // ---- 2. which levels are staged -----------------------------------------------------------------------------------
        if (tid == 0) {
            int used = 0;
            for (int l = 0; l < L; ++l) {
                Desc d;
                d.H = lv[l].H;
                d.W = lv[l].W;
                d.ymin = box[l][0];
                d.xmin = box[l][2];
                d.h = box[l][1] - box[l][0] + 1;
                d.w = box[l][3] - box[l][2] + 1;
                d.staged = 0;
                d.base = used;
                if (box[l][1] < 0) {          // no sample of the tile lies on this level
                    d.h = d.w = 0;
                    d.ymin = d.xmin = 0;
                } else if (d.h * d.w <= ZIRA_PATCH_ROWS - used) {
                    d.staged = 1;
                    used += d.h * d.w;
                }
                desc[l] = d;
            }
        }
        __syncthreads();
        // ---- 3. staging: 16-byte pieces, all in flight ----------------------------------------------------------------------
        const float *vb = value + ((size_t)b * S * M + m) * kD;     // + pixel * M * kD
#if !(ZIRA_PATCH_DEV_SKIP & 2)
        {
            // the staged patches are one list of 16-byte pieces (level by level); a thread takes pieces tid, tid + 512, ...:
            // addresses first, then every load, then the LDS writes
            constexpr int NP = (ZIRA_PATCH_ROWS * 8 + NTHR - 1) / NTHR;
            int cum[MAXL + 1];
            cum[0] = 0;
#pragma unroll
            for (int l = 0; l < MAXL; ++l) cum[l + 1] = cum[l] + ((l < L && desc[l].staged) ? desc[l].h * desc[l].w * 8 : 0);
            const int total = cum[MAXL];
            float4 v[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int c = min(tid + j * NTHR, max(total - 1, 0));
                int l = 0;
#pragma unroll
                for (int k = 1; k < MAXL; ++k) l = c >= cum[k] ? k : l;
                const Desc d = desc[l];
                const int cc = c - cum[l], pix = cc >> 3, piece = cc & 7, py = pix / max(d.w, 1), px = pix - py * d.w;
                const int yy = min(d.ymin + py, d.H - 1), xx = min(d.xmin + px, d.W - 1);
                v[j] = *reinterpret_cast<const float4 *>(vb + ((size_t)lv[l].start + (size_t)yy * d.W + xx) * M * kD + piece * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int c = tid + j * NTHR;
                if (c < total) {
                    int l = 0;
#pragma unroll
                    for (int k = 1; k < MAXL; ++k) l = c >= cum[k] ? k : l;
                    const int cc = c - cum[l];
                    *reinterpret_cast<float4 *>(patch + ((size_t)desc[l].base * 8 + cc) * 4) = v[j];
                }
            }
        }
#endif
        __syncthreads();
        // ---- 4. gather: eight lanes per query ---------------------------------------------------------------------------------
        const int slot = lane >> 3, cq = lane & 7;
        for (int qi = wave * 8 + slot; qi < ((ZIRA_PATCH_DEV_SKIP & 4) ? 0 : NQ); qi += (NTHR / 64) * 8) {      // (two rounds; the trip count is wave-uniform)
            const int qy = ty0 + qi / TQW, qx = tx0 + qi % TQW;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 0; l < L; ++l) {
                const Desc d = desc[l];
                if (d.h == 0) continue;
                for (int p = 0; p < P; ++p) {
                    const int i = qi * LP + l * P + p;
                    const float4 w4 = tw[i];
                    const int yx = tc[i];
                    const int y0 = yx >> 16, x0 = (int)(short)(yx & 0xffff);
                    if (d.staged) {
                        const int ry0 = min(max(y0 - d.ymin, 0), d.h - 1), ry1 = min(max(y0 + 1 - d.ymin, 0), d.h - 1);
                        const int rx0 = min(max(x0 - d.xmin, 0), d.w - 1), rx1 = min(max(x0 + 1 - d.xmin, 0), d.w - 1);
                        const float *pb = patch + (size_t)d.base * kD + cq * 4;
                        const float4 v00 = *reinterpret_cast<const float4 *>(pb + (ry0 * d.w + rx0) * kD);
                        const float4 v01 = *reinterpret_cast<const float4 *>(pb + (ry0 * d.w + rx1) * kD);
                        const float4 v10 = *reinterpret_cast<const float4 *>(pb + (ry1 * d.w + rx0) * kD);
                        const float4 v11 = *reinterpret_cast<const float4 *>(pb + (ry1 * d.w + rx1) * kD);
                        acc = fma4(w4.x, v00, acc);
                        acc = fma4(w4.y, v01, acc);
                        acc = fma4(w4.z, v10, acc);
                        acc = fma4(w4.w, v11, acc);
                    } else {
                        const int gy0 = min(max(y0, 0), d.H - 1), gy1 = min(max(y0 + 1, 0), d.H - 1);
                        const int gx0 = min(max(x0, 0), d.W - 1), gx1 = min(max(x0 + 1, 0), d.W - 1);
                        const float *gb = vb + (size_t)lv[l].start * M * kD + cq * 4;
                        const float4 v00 = *reinterpret_cast<const float4 *>(gb + (size_t)(gy0 * d.W + gx0) * M * kD);
                        const float4 v01 = *reinterpret_cast<const float4 *>(gb + (size_t)(gy0 * d.W + gx1) * M * kD);
                        const float4 v10 = *reinterpret_cast<const float4 *>(gb + (size_t)(gy1 * d.W + gx0) * M * kD);
                        const float4 v11 = *reinterpret_cast<const float4 *>(gb + (size_t)(gy1 * d.W + gx1) * M * kD);
                        acc = fma4(w4.x, v00, acc);
                        acc = fma4(w4.y, v01, acc);
                        acc = fma4(w4.z, v10, acc);
                        acc = fma4(w4.w, v11, acc);
                    }
                }
            }
            if (qy < Hq && qx < Wq) {
                const int q = stq + qy * Wq + qx;
                *reinterpret_cast<float4 *>(out + ((size_t)((size_t)b * Q + q) * M + m) * kD + cq * 4) = acc;
            }
        }
    }
}

constexpr size_t kLds = (size_t)NQ * MAXLP * 20 + (size_t)ZIRA_PATCH_ROWS * kD * 4;

}  // namespace

namespace zira {

// -1: not applicable (the caller takes the lean kernel); else a hipError_t
int patch_forward_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc, const float *attn, int B,
                      int S, int M, int D, int L, int Q, int P, float *out, hipStream_t st)
{
    if (D != kD || Q != S || L < 1 || L > MAXL || L * P > MAXLP || P < 1 || (long long)B * M > (1 << 20)) return -1;
    if ((long long)S * M * kD >= (1ll << 31)) return -1;     // (32-bit row arithmetic inside a batch element)
    if ((uintptr_t)value & 15 || (uintptr_t)out & 15 || (uintptr_t)loc & 7) return -1;
    static int attr = [] {      // one device per process
        return (int)hipFuncSetAttribute(reinterpret_cast<const void *>(msda_fwd_patch), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kLds);
    }();
    if (attr != 0) return -1;
    hipLaunchKernelGGL(msda_fwd_patch, dim3(256), dim3(NTHR), kLds, st, value, shapes, start, loc, attn, B, S, M, L, P, out);
    return (int)hipGetLastError();
}

}  // namespace zira
