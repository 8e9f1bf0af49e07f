// msda_box.hip -- forward of multi-scale deformable attention for gfx950 (MI355X), dense calls (encoder
// self-attention: every pixel of the value maps is a query, Q == S): value boxes staged in LDS.
//
// Arithmetic to match: reference csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:237-299 (bilinear :33-84).
//
// Why.  msda_fwd_lean (csrc/msda.hip) gathers 64 value rows of 128 bytes per (b, q, m) item through the vector
// L1: at the encoder shape that is 2.9 GB of row traffic per call and the kernel runs at the rate at which a CU's
// L1 accepts rows (177 us, 11 % of the HBM roofline).  But neighbouring queries of the encoder sample the SAME
// pixels: query q is a pixel of the maps, its reference point is that pixel's centre and its offsets are a few
// pixels (reference transformer_for_adapter.py:482-497, :893-900).  So a block takes a TILE of queries -- 8 x 16
// pixels of level 0 (or the 4 x 4 pixels of level 1 above them) and one head --, stages, level by level, the box of
// value rows around the tile's footprint (+- 4 pixels) in LDS once, and its items read their corner rows from
// there (`ds_read_b128`: 256 B/clk per CU against 64 for the L1).  A corner outside the box -- an offset beyond the
// halo, or a caller whose queries are not the pixels of the maps at all -- is simply read from global memory like
// before: the result is the same for ANY sampling locations, the tile order and the boxes only decide the speed.
// Queries of the two coarse levels (6 % of them; their footprints on level 0 are too large to stage) and every other
// kind of call keep msda_fwd_lean.
//
// Work decomposition: 512 threads = 8 waves; a wave owns 8 of the tile's 64 queries as 2 groups of 4.  Per level:
// stage the box (all threads, 16-byte loads), barrier, then per group one 64-lane chunk -- lane e = (item e >> 4,
// point (e >> 2) & 3, corner e & 3) computes its corner's address and weight once -- and 8 gather instructions of
// 8 rows x 8 channel quads (the entry -> row-slot hand-off by ds_bpermute as in msda_fwd_lean), accumulated per item;
// after the last level the 8 row slots of an item are folded with DPP row rotations and stored.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msda_internal.h"

#ifndef ZIRA_BOX_HALO
#define ZIRA_BOX_HALO 4
#endif

namespace zira {
namespace {

constexpr unsigned kBoxThreads = 512, kBoxWaves = kBoxThreads / 64;
#ifndef ZIRA_BOX_GROUPS
#define ZIRA_BOX_GROUPS 2                   // chunks of 4 queries per wave: the tile has 8 waves x 4 x this many queries
#endif
constexpr unsigned kGroups = ZIRA_BOX_GROUPS;
constexpr int kTQH = 8, kTQW = 4 * (int)kGroups;   // query tile on level 0; (kTQH / 2) x (kTQW / 2) on level 1
constexpr int kHalo = ZIRA_BOX_HALO;
constexpr unsigned kBoxMaxPix = (kTQH + 2 * kHalo + 2) * (kTQW + 2 * kHalo + 2);   // largest box (level 0 under a level-0 tile)
constexpr unsigned kBoxMaxLevels = 8;

template <int CTRL>
__device__ __forceinline__ float dpp_addb(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}

struct BLevel {
    int H, W;
    unsigned st;
};

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned uniu(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }

// ceil(a / b) and floor(a / b) for small non-negative ints
__device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

__global__ __launch_bounds__(kBoxThreads, kGroups <= 2 ? 6 : 4) void msda_fwd_box(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ attn, unsigned S, unsigned M, unsigned L, unsigned Q,
    unsigned heads, float *__restrict__ out)
{
    constexpr unsigned D = 32, P = 4;
    __shared__ __attribute__((aligned(16))) float box[kBoxMaxPix * D];
    __shared__ BLevel lv[kBoxMaxLevels];
    __shared__ unsigned tot[4];

    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < L) {
        lv[tid].H = (int)shapes[2 * tid];
        lv[tid].W = (int)shapes[2 * tid + 1];
        lv[tid].st = (unsigned)start[tid];
    }
    __syncthreads();
    // tiles per head: level-0 tiles of kTQH x kTQW queries, then level-1 tiles of half that (the same footprint)
    const int H0 = uni(lv[0].H), W0 = uni(lv[0].W);
    const unsigned nty0 = (unsigned)cdiv(H0, kTQH), ntx0 = (unsigned)cdiv(W0, kTQW);
    unsigned nty1 = 0, ntx1 = 0;
    if (L > 1) {
        nty1 = (unsigned)cdiv(uni(lv[1].H), kTQH / 2);
        ntx1 = (unsigned)cdiv(uni(lv[1].W), kTQW / 2);
    }
    // Queries are taken to be the pixels of the maps (q = start[l] + y W + x) for PLACEMENT only.  Levels 0 and 1 go by
    // 2-D tiles when they sit where that reading puts them; every other query index goes by flat runs of 128 with no
    // box (all corners gathered directly), so that every q in [0, Q) is computed whatever the tables say.
    const unsigned st1 = L > 1 ? uniu(lv[1].st) : 0u, hw1 = L > 1 ? (unsigned)(uni(lv[1].H) * uni(lv[1].W)) : 0u;
    const bool regular = uniu(lv[0].st) == 0 && L > 1 && st1 == (unsigned)(H0 * W0) &&
                         (unsigned long long)st1 + hw1 <= Q;
    const unsigned nt0 = regular ? nty0 * ntx0 : 0u, nt1 = regular ? nty1 * ntx1 : 0u;
    const unsigned q_first = regular ? st1 + hw1 : 0u;   // first query of the flat runs
    const unsigned ntf = (Q - q_first + kTQH * kTQW - 1) / (kTQH * kTQW);
    const unsigned ntile = nt0 + nt1 + ntf;
    const unsigned LP = L * P;

    // lane roles inside a chunk (as in msda_fwd_lean, CQR = 2): DPP row R = 2 channel quads, 8 row slots
    const unsigned Rr = lane >> 4, slot = (lane & 15) >> 1, cq = Rr * 2 + (lane & 1);
    const int bp = (int)(slot * 4);
    const unsigned e_item = lane >> 4, e_pt = (lane >> 2) & 3, e_c = lane & 3;

    // head-major over the XCDs (blocks b, b + 8, ... share an XCD: its L2 keeps the head's value slice)
    const unsigned xcd = blockIdx.x & 7, nbx = gridDim.x >> 3;
    const unsigned hp = (heads + 7) >> 3;
    const unsigned h_lo = xcd * hp, h_hi = h_lo + hp < heads ? h_lo + hp : heads;
    const unsigned nwork = h_lo < h_hi ? (h_hi - h_lo) * ntile : 0u;
    for (unsigned wi = blockIdx.x >> 3; wi < nwork; wi += nbx) {
        const unsigned h = uniu(h_lo + wi / ntile), t = uniu(wi % ntile);
        const unsigned b = h / M, m = h - b * M;
        // the tile: query level, origin, extent -- or a flat run of query indices (no boxes)
        unsigned lq = 0, qbase = 0, nq;
        int qy0 = 0, qx0 = 0, qh = kTQH, qw = kTQW;
        bool flat = false;
        if (t < nt0) {
            qy0 = (int)(t / ntx0) * kTQH;
            qx0 = (int)(t % ntx0) * kTQW;
        } else if (t < nt0 + nt1) {
            const unsigned t1 = t - nt0;
            lq = 1;
            qy0 = (int)(t1 / ntx1) * (kTQH / 2);
            qx0 = (int)(t1 % ntx1) * (kTQW / 2);
            qh = kTQH / 2;
            qw = kTQW / 2;
        } else {
            flat = true;
            qbase = q_first + (t - nt0 - nt1) * (unsigned)(kTQH * kTQW);
        }
        const int Hq = uni(lv[lq].H), Wq = uni(lv[lq].W);
        const unsigned stq = uniu(lv[lq].st);
        const int qh_e = Hq - qy0 < qh ? Hq - qy0 : qh, qw_e = Wq - qx0 < qw ? Wq - qx0 : qw;   // inside the map
        if (flat) nq = Q - qbase < (unsigned)(kTQH * kTQW) ? Q - qbase : (unsigned)(kTQH * kTQW);
        else nq = (unsigned)(qh_e * qw_e);
        const float *vb = value + (size_t)b * S * M * D;
        // query index of item `it` of the tile
        const float rqw = 1.f / (float)qw_e;
        auto item_q = [&](unsigned it) {
            if (flat) return qbase + it;
            const unsigned iy = (unsigned)(((float)it + 0.5f) * rqw), ix = it - iy * (unsigned)qw_e;   // it / qw_e (exact for it < 128)
            return stq + (unsigned)(qy0 + (int)iy) * (unsigned)Wq + (unsigned)(qx0 + (int)ix);
        };
        unsigned q_of[kGroups];   // the query of this lane's entry in each of the wave's four chunks
#pragma unroll
        for (unsigned g = 0; g < kGroups; ++g) {
            const unsigned it = 4 * (wave + kBoxWaves * g) + e_item;
            q_of[g] = it < nq ? item_q(it) : 0u;
        }

        float4 acc[kGroups][4];   // [group of this wave][item of the group]
#pragma unroll
        for (unsigned g = 0; g < kGroups; ++g)
#pragma unroll
            for (unsigned i = 0; i < 4; ++i) acc[g][i] = make_float4(0.f, 0.f, 0.f, 0.f);

        for (unsigned l = 0; l < L; ++l) {
            const int H = uni(lv[l].H), W = uni(lv[l].W);
            const unsigned st = uniu(lv[l].st);
            // footprint of the tile on level l (pixel centres scaled by the level sizes), + halo, clipped
            int by0 = (int)floorf((float)qy0 * (float)H / (float)Hq - 0.5f) - kHalo;
            int by1 = (int)ceilf((float)(qy0 + qh_e) * (float)H / (float)Hq - 0.5f) + kHalo + 1;
            int bx0 = (int)floorf((float)qx0 * (float)W / (float)Wq - 0.5f) - kHalo;
            int bx1 = (int)ceilf((float)(qx0 + qw_e) * (float)W / (float)Wq - 0.5f) + kHalo + 1;
            by0 = by0 < 0 ? 0 : by0;
            bx0 = bx0 < 0 ? 0 : bx0;
            by1 = by1 > H ? H : by1;
            bx1 = bx1 > W ? W : bx1;
            int bh = by1 - by0, bw = bx1 - bx0;
            if (bh < 0 || flat) bh = 0;
            if (bw < 0 || flat) bw = 0;
            while ((unsigned)(bh * bw) > kBoxMaxPix && bh > 1) --bh;   // (odd level ratios: trim, the rest is gathered directly)
            if ((unsigned)(bh * bw) > kBoxMaxPix) bw = (int)(kBoxMaxPix / (unsigned)bh);
            by0 = uni(by0); bx0 = uni(bx0); bh = uni(bh); bw = uni(bw);
            const unsigned npix = (unsigned)(bh * bw);

            __syncthreads();   // (the previous level's box is no longer read)
            {
                const float rbw = bw > 0 ? 1.f / (float)bw : 0.f;
                for (unsigned x = tid; x < npix * (D / 4); x += kBoxThreads) {
                    const unsigned c4 = x & 7, pix = x >> 3;
                    unsigned r = (unsigned)(((float)pix + 0.5f) * rbw);
                    r = r * (unsigned)bw > pix ? r - 1 : ((r + 1) * (unsigned)bw <= pix ? r + 1 : r);
                    const unsigned c = pix - r * (unsigned)bw;
                    const size_t gp = (size_t)st + (size_t)(by0 + (int)r) * W + (unsigned)(bx0 + (int)c);
                    *reinterpret_cast<float4 *>(box + pix * D + c4 * 4) =
                        *reinterpret_cast<const float4 *>(vb + (gp * M + m) * D + c4 * 4);
                }
            }
            __syncthreads();

#pragma unroll
            for (unsigned g = 0; g < kGroups; ++g) {
                // chunk: items 4 * (wave + 8 g) .. + 3 of the tile, point e_pt, corner e_c
                const unsigned it = 4 * (wave + kBoxWaves * g) + e_item;
                if (4 * (wave + kBoxWaves * g) >= nq) continue;   // wave-uniform: no item of this group exists
                __builtin_amdgcn_sched_barrier(0);                 // (one group at a time: the four together spill)
                const bool act = it < nq;
                const unsigned q = q_of[g];
                const size_t sidx = ((size_t)(b * Q + q) * M + m) * LP + l * P + e_pt;
                float w = 0.f;
                unsigned off = 0;      // byte offset: inside the box (bit 31 clear) or inside the batch element (bit 31 set)
                if (act) {
#pragma clang fp contract(off)
                    const float2 xy = *reinterpret_cast<const float2 *>(loc + sidx * 2);
                    const float a = attn[sidx];
                    const float Hf = (float)H, Wf = (float)W;
                    const float h_im = xy.y * Hf - 0.5f;
                    const float w_im = xy.x * Wf - 0.5f;
                    const bool valid = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const float lh = h_im - hf, lw = w_im - wf;
                    const int dy = (int)(e_c >> 1), dx = (int)(e_c & 1);
                    const int y = (int)hf + dy, x = (int)wf + dx;
                    const float wy = dy ? lh : 1.f - lh;
                    const float wx = dx ? lw : 1.f - lw;
                    const bool inb = valid && y >= 0 && y < H && x >= 0 && x < W;
                    w = inb ? (wy * wx) * a : 0.f;
                    if (inb) {
                        const unsigned ry = (unsigned)(y - by0), rx = (unsigned)(x - bx0);
                        if (ry < (unsigned)bh && rx < (unsigned)bw) off = (ry * (unsigned)bw + rx) * (D * 4u);
                        else off = 0x80000000u | (((st + (unsigned)y * (unsigned)W + (unsigned)x) * M + m) * (D * 4u));
                    }
                }
                const int off_i = (int)off, w_i = __float_as_int(w);
#pragma unroll
                for (unsigned j0 = 0; j0 < 8; j0 += 4) {   // four rows in flight (eight cost more registers than the occupancy allows)
                    float4 v[4];
                    float wj[4];
#pragma unroll
                    for (unsigned jj = 0; jj < 4; ++jj) {
                        const int a = bp + (int)((j0 + jj) * 8 * 4);
                        const unsigned oj = (unsigned)__builtin_amdgcn_ds_bpermute(a, off_i);
                        wj[jj] = __int_as_float(__builtin_amdgcn_ds_bpermute(a, w_i));
                        if (oj & 0x80000000u)
                            v[jj] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(vb) + (oj & 0x7FFFFFFFu) + cq * 16);
                        else
                            v[jj] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(box) + oj + cq * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (unsigned jj = 0; jj < 4; ++jj) {
                        float4 &ac = acc[g][(j0 + jj) >> 1];
                        ac.x = fmaf(wj[jj], v[jj].x, ac.x);
                        ac.y = fmaf(wj[jj], v[jj].y, ac.y);
                        ac.z = fmaf(wj[jj], v[jj].z, ac.z);
                        ac.w = fmaf(wj[jj], v[jj].w, ac.w);
                    }
                }
            }
        }

        // fold the 8 row slots of every item (lanes of one DPP row that hold the same channel quad) and store
#pragma unroll
        for (unsigned g = 0; g < kGroups; ++g) {
#pragma unroll
            for (unsigned i = 0; i < 4; ++i) {
                const unsigned it = 4 * (wave + kBoxWaves * g) + i;
                if (it >= nq) continue;   // wave-uniform
                float4 a4 = acc[g][i];
                a4.x = dpp_addb<0x122>(a4.x); a4.y = dpp_addb<0x122>(a4.y); a4.z = dpp_addb<0x122>(a4.z); a4.w = dpp_addb<0x122>(a4.w);
                a4.x = dpp_addb<0x124>(a4.x); a4.y = dpp_addb<0x124>(a4.y); a4.z = dpp_addb<0x124>(a4.z); a4.w = dpp_addb<0x124>(a4.w);
                a4.x = dpp_addb<0x128>(a4.x); a4.y = dpp_addb<0x128>(a4.y); a4.z = dpp_addb<0x128>(a4.z); a4.w = dpp_addb<0x128>(a4.w);
                const unsigned q = item_q(it);
                if (slot == 0) *reinterpret_cast<float4 *>(out + ((size_t)(b * Q + q) * M + m) * D + cq * 4) = a4;
            }
        }
    }
}

inline unsigned box_cu_count()
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

}  // namespace

bool box_forward_applies(int B, int S, int M, int D, int L, int Q, int P)
{
    if (D != 32 || P != 4 || Q != S || L < 2 || L > (int)kBoxMaxLevels) return false;
    if ((unsigned long long)B * M * Q < 16 * 4096) return false;                    // dense calls only
    if ((unsigned long long)S * M * D * 4 >= (1ull << 31)) return false;           // 31-bit byte offsets inside a batch element
    if ((unsigned long long)B * Q * M * L * P * 2 >= (1ull << 31)) return false;
    return true;
}

// Launches the box kernel for the queries of levels 0 and 1 (as pixels of the value maps: q = start[l] + y W + x);
// the caller runs the plain gather kernel for the queries from `*q_rest` on (device-side tables: the host passes S and
// gets the split point through the level table it cannot read -- so the plain kernel takes its range from the table too).
int box_forward_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc,
                    const float *attn, int B, int S, int M, int L, int Q, float *out, hipStream_t st)
{
    (void)B;
    const unsigned grid = box_cu_count() * (kGroups <= 2 ? 3 : 2);
    hipLaunchKernelGGL(msda_fwd_box, dim3(grid), dim3(kBoxThreads), 0, st, value, shapes, start, loc, attn, (unsigned)S,
                       (unsigned)M, (unsigned)L, (unsigned)Q, (unsigned)(B * M), out);
    return (int)hipGetLastError();
}

}  // namespace zira
