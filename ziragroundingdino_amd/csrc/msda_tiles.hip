// msda_tiles.hip -- backward of multi-scale deformable attention for gfx950 (MI355X), sparse calls
// (decoder cross-attention: a few hundred queries per image): "plan + tile accumulate".
//
// Arithmetic to match: reference csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:87-159 inside :301-403
// (grad_value += w_corner * (attn * grad_out), grad_attn = <grad_out, sample>, grad_loc from the corner
// differences).  The decomposition is not the reference's and not round 2's entry sort either.
//
// Why this shape.  Round 2 turned every (sample, corner) into an 8-byte entry, sorted the entries by
// destination tile (K1) and walked each tile's entries in row order (K2): 46 us on uniform sampling
// locations, 75-80 us when the decoder's queries pile up on a few objects (a tile with thousands of
// entries is one wave's serial chain; its slices went to a third launch with global atomics).  Measured on
// MI355X (scripts/lds_atomic_rates2.hip): `ds_add_f64` takes 8.6 cycles per wave instruction whatever the
// addresses are -- 64 lanes on one row cost the same as 64 lanes on 8 rows -- while `ds_add_f32` takes
// 193.  So the destination can simply be SUMMED IN LDS IN DOUBLE, in any order, and a pile-up on a few
// pixels costs nothing extra.  What is left to balance is the number of samples per work item, and every
// (head, level) unit holds exactly Q * P of them however they are spread: a cheap pass over the sampling
// locations counts them per tile, sorts them by tile and cuts busy tiles by record range.
//
//   plan   (msda_bwd_plan)  one 1024-thread block per (head, level) unit; no value / grad_out traffic.  Pass 1: a
//          thread per query computes the pixel coordinates of its samples and counts each sample into every tile
//          (16 x 8 pixels) its 2 x 2 corner block touches (LDS histogram); samples outside the window get zero
//          gradients.  A block scan turns the histogram into record offsets.  Pass 2 writes one 16-byte RECORD per
//          (sample, touched tile): {sample id : 18 | home position : 7 | flags : 7, the four corner positions
//          inside the tile as bytes (255 = corner not in this tile), lw, lh}, sorted by tile.  Then per tile a
//          16-byte header {first record, count (0 when split), origin, level | split}; a tile with more than `cap`
//          records is SPLIT: all its K = ceil(n / cap) shares become extra work items {first record, count,
//          origin, level}, and its pixels are zeroed here (the shares meet through global fp32 atomics).
//   accum  (msda_bwd_tile_accum)  persistent blocks (256 threads, 4 per CU) dealing out the work items of one XCD's
//          heads in turn: split shares first, then tiles from the coarsest level down.  Per item: 32 records per
//          block step, 8 lanes x 4 channels per record: the grad_out row (one 128-byte gather), the corner terms
//          w * (attn * g) added to the tile's accumulators in LDS (`ds_add_f64`; the two records of a 16-lane
//          group use different bank halves), and -- in the tile that owns the sample -- the four value rows,
//          their dot products with the grad_out row and grad_sampling_loc / grad_attn_weight.  The loop is a
//          two-stage software pipeline (loads of step i + 1 in flight while step i is added), all loads and stores
//          unconditional (clamped record index, dump line) so that the compiler's in-order vmcnt waits stay exact;
//          the next item's header, records and first loads are issued before the current tile is flushed.
//          Flush: plain 16-byte stores of every pixel for unsplit tiles (no zero-fill of grad_value anywhere), fp32
//          atomics of the non-zero pixels for shares of split tiles.
//
// Measured (MI355X, decoder shape B=2 S=22223 M=8 L=4 Q=900 P=4): plan 12 us + accumulate 38-40 us; uniform
// 51-53, in-model locations 50-52, all queries on 5 % of the map 46, on one pixel block 42 us (round 2:
// 46 / 75-80 / - / -).  Two other decompositions were measured in round 3 and lost: handing
// the items out through a counter per XCD (same-address returning atomics from 128 blocks: 74 us), and sorting
// the corner terms by destination pixel so that a wave gathers a packet of rows (49-60 us: latency-bound
// chains of dependent loads per wave).
//
// The sums are formed in double from exact products of fp32 factors (w and attn * g rounded to fp32 as in
// the reference, cuh:117-147) and rounded to fp32 once: at least as close to the reference as an fp32
// accumulation in any order; inf / NaN propagate as they do there.  Geometry is derived on the device from
// the int64 level table (the C ABI has device pointers only); the host sizes the workspace from S.
// Precondition, as in the reference module (ms_deform_attn.py:284): the levels tile [0, S).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "msda_internal.h"

#ifndef ZIRA_TILE_CAP
#define ZIRA_TILE_CAP 320      // samples a work item should hold (plan: K = ceil(samples / cap))
#endif
#ifndef ZIRA_TILE_BLOCKS_PER_CU
#define ZIRA_TILE_BLOCKS_PER_CU 4
#endif
#ifndef ZIRA_TILE_STAMPS
#define ZIRA_TILE_STAMPS 0     // 1: developer build with per-block phase times (scripts/tile_stamps.py); 0 in shipped builds
#endif
#ifndef ZIRA_TILE_ABL
#define ZIRA_TILE_ABL 0        // developer timing builds (wrong results): 1 no accumulator adds, 2 no home work, 4 no flush,
                               // 8 all value rows = pixel 0, 16 all grad_out rows = query 0
#endif

namespace zira {
namespace {

constexpr unsigned kTMaxLevels = 16;
constexpr unsigned kInvalidCell = 0xFFFFFFFFu;
constexpr unsigned kInvalidItem = 0xFFFFFFFFu;
constexpr unsigned kKmax = 64;          // query shares per tile at most (6-bit field of an item word)
constexpr unsigned kTH = 16, kTW = 8;   // tile: 16 x 8 pixels
constexpr unsigned kAccThreads = 256;
constexpr unsigned kPlanThreads = 1024;
#ifndef ZIRA_PLAN_SPLIT
#define ZIRA_PLAN_SPLIT 1
#endif
constexpr unsigned kPlanSplit = ZIRA_PLAN_SPLIT;      // plan blocks per (head, level) unit: each takes a quarter of the unit's tiles

struct FastDivT {
    unsigned mul, shift, d;
};
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDivT f)
{
    return (unsigned)(((unsigned long long)n * f.mul) >> f.shift);
}
inline FastDivT make_fdiv(unsigned d)
{
    FastDivT f;
    f.d = d;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.shift = 31 + s;
    f.mul = (unsigned)(((1ull << (31 + s)) / d) + 1);
    return f;
}

struct TileGeom {
    unsigned B, S, M, L, Q, P, LP, heads;
    unsigned ntmax;   // tiles per head: upper bound from S (stride of the K table)
    unsigned cap;     // samples a work item should hold
    unsigned ecap;    // extra-item slots per (head, level) unit
    unsigned hp;      // heads per XCD (0: fewer than 8 heads, one global item range)
    FastDivT Mdiv, Pdiv, LSdiv;   // LSdiv: L * kPlanSplit
};

struct TLevel {
    int H, W;
    unsigned st, nty, ntx, tbase;
};
constexpr unsigned kTLevelWords = sizeof(TLevel) / 4;

// Per-level tile grid from the device-side int64 tables into LDS; the same in both kernels.  Called by every
// thread of the block; ends with a barrier.  Returns the tiles per head.
__device__ __forceinline__ unsigned tile_levels(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                                unsigned L, TLevel *lv, unsigned *tot)
{
    if (threadIdx.x < L) {
        const unsigned l = threadIdx.x;
        TLevel v;
        v.H = (int)shapes[2 * l];
        v.W = (int)shapes[2 * l + 1];
        v.st = (unsigned)start[l];
        v.nty = ((unsigned)v.H + kTH - 1) / kTH;
        v.ntx = ((unsigned)v.W + kTW - 1) / kTW;
        v.tbase = 0;
        lv[l] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned nt = 0;
        for (unsigned l = 0; l < L; ++l) {
            lv[l].tbase = nt;
            nt += lv[l].nty * lv[l].ntx;
        }
        *tot = nt;
    }
    __syncthreads();
    return *tot;
}

struct Cell {
    bool valid;
    int cy, cx;      // top-left pixel + 1: cy in [0, H], cx in [0, W]
    float lw, lh;
};

// Pixel coordinates exactly as the oracle forms them (mul, then sub, no fma contraction), so that floor()
// picks the same pixel (reference cuh:285-288, :38-45).
__device__ __forceinline__ Cell cell_of(float x, float y, int H, int W)
{
#pragma clang fp contract(off)
    Cell c;
    const float Hf = (float)H, Wf = (float)W;
    const float h_im = y * Hf - 0.5f;
    const float w_im = x * Wf - 0.5f;
    c.valid = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
    const float hf = floorf(h_im), wf = floorf(w_im);
    c.lh = h_im - hf;
    c.lw = w_im - wf;
    c.cy = c.valid ? (int)hf + 1 : 0;
    c.cx = c.valid ? (int)wf + 1 : 0;
    return c;
}

#if ZIRA_TILE_STAMPS   // developer build: per-block phase times (100 MHz counter), scripts/tile_stamps.py
__device__ unsigned long long zira_tile_stamps[16 * 2048];
__device__ unsigned long long zira_plan_stamps[16 * 2048];
#define TSTAMP_DECL unsigned long long ts_t = wall_clock64(), ts_acc[16] = {0}
#define TSTAMP(i)                                       \
    do {                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long ts_n = wall_clock64(); \
        ts_acc[i] += ts_n - ts_t;                       \
        ts_t = ts_n;                                    \
    } while (0)
#define TSTAMP_COUNT(i) ts_acc[i] += 1
#define PSTAMP_FLUSH                                                                                 \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 2048)                                                   \
            for (int ts_i = 0; ts_i < 16; ++ts_i) zira_plan_stamps[blockIdx.x * 16 + ts_i] = ts_acc[ts_i]; \
    } while (0)
#define TSTAMP_FLUSH                                                                                 \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 2048)                                                   \
            for (int ts_i = 0; ts_i < 16; ++ts_i) zira_tile_stamps[blockIdx.x * 16 + ts_i] = ts_acc[ts_i]; \
    } while (0)
#else
#define TSTAMP_DECL
#define TSTAMP(i)
#define TSTAMP_COUNT(i)
#define TSTAMP_FLUSH
#define PSTAMP_FLUSH
#endif

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
// exclusive prefix sum over a[0, n) in LDS, in place, by the whole block (kPlanThreads threads); returns the total.
// `scr` holds kPlanThreads / 64 + 1 words.
__device__ __forceinline__ unsigned block_scan_inplace(unsigned *a, unsigned n, unsigned *scr)
{
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned per = (n + kPlanThreads - 1) / kPlanThreads;
    const unsigned i0 = tid * per, i1 = i0 + per < n ? i0 + per : n;
    unsigned sum = 0;
    for (unsigned i = i0; i < i1; ++i) sum += a[i];
    unsigned incl = sum;
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) scr[wave] = incl;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (unsigned w = 0; w < kPlanThreads / 64; ++w) {
        const unsigned t = scr[w];
        base += w < wave ? t : 0u;
        total += t;
    }
    unsigned run = base + incl - sum;
    for (unsigned i = i0; i < i1; ++i) {
        const unsigned c = a[i];
        a[i] = run;
        run += c;
    }
    __syncthreads();
    return total;
}

// The tiles a sample's in-map corners fall into (1, 2 or 4) and which of them owns the sample (the tile of the
// bottom-right corner clamped into the map): f(tile index inside the level, tile row, tile column, home)
template <typename F>
__device__ __forceinline__ void for_each_touched_tile(const Cell &c, const TLevel &Lv, F f)
{
    const unsigned cy = (unsigned)c.cy, cx = (unsigned)c.cx;
    const unsigned hy = cy < (unsigned)Lv.H ? cy : (unsigned)Lv.H - 1, hx = cx < (unsigned)Lv.W ? cx : (unsigned)Lv.W - 1;
    const unsigned yA = cy >= 1 ? cy - 1 : 0u, xA = cx >= 1 ? cx - 1 : 0u;     // first in-map corner row / column
    const unsigned tyA = yA / kTH, tyB = hy / kTH, txA = xA / kTW, txB = hx / kTW;
    f(tyB * Lv.ntx + txB, tyB, txB, true);
    if (txA != txB) f(tyB * Lv.ntx + txA, tyB, txA, false);
    if (tyA != tyB) {
        f(tyA * Lv.ntx + txB, tyA, txB, false);
        if (txA != txB) f(tyA * Lv.ntx + txA, tyA, txA, false);
    }
}

// The 16-byte record of a sample in one of the tiles it touches: everything the accumulate kernel would otherwise
// derive per lane from the cell.  word 0 = sample (q * P + p):18 | position of the clamped bottom-right corner in the
// tile:7 | flags:7 (home, x step, y step, the four corner-in-map masks); word 1 = accumulator row of each corner
// (a byte each, 255: not in this tile); words 2, 3 = the bilinear fractions lw, lh.
constexpr unsigned kNoRow = 255;
__device__ __forceinline__ uint4 make_record(unsigned sid, unsigned cy, unsigned cx, unsigned lw, unsigned lh,
                                             const TLevel &Lv, unsigned ty, unsigned tx, bool home)
{
    const unsigned H = (unsigned)Lv.H, W = (unsigned)Lv.W, ty0 = ty * kTH, tx0 = tx * kTW;
    unsigned rows = 0;
#pragma unroll
    for (unsigned cc = 0; cc < 4; ++cc) {
        const unsigned y = cy - 1 + (cc >> 1), x = cx - 1 + (cc & 1);     // (unsigned: -1 wraps and fails the tests)
        const bool in = y < H && x < W && (y - ty0) < kTH && (x - tx0) < kTW;
        rows |= (in ? (y - ty0) * kTW + (x - tx0) : kNoRow) << (8 * cc);
    }
    const unsigned hy = cy < H ? cy : H - 1, hx = cx < W ? cx : W - 1;
    const unsigned yA = cy >= 1 ? cy - 1 : 0u, xA = cx >= 1 ? cx - 1 : 0u;
    const unsigned hpos = home ? (hy - ty0) * kTW + (hx - tx0) : 0u;
    unsigned fl = home ? 1u : 0u;
    fl |= (hx != xA) ? 2u : 0u;                                // the right column is another pixel
    fl |= (hy != yA) ? 4u : 0u;                                // the bottom row is another pixel
    const bool y0in = cy >= 1, y1in = cy < H, x0in = cx >= 1, x1in = cx < W;
    fl |= (y0in && x0in ? 8u : 0u) | (y0in && x1in ? 16u : 0u) | (y1in && x0in ? 32u : 0u) | (y1in && x1in ? 64u : 0u);
    return make_uint4(sid | (hpos << 18) | (fl << 25), rows, lw, lh);
}

// One query per thread and pass (kOnePass: Q <= kPlanThreads, P <= 4): the cells and the ranks inside their tiles stay
// in registers between the count and the copy-out, so the sampling locations are read once.
template <bool kOnePass>
__global__ __launch_bounds__(kPlanThreads) void msda_bwd_plan(
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start, const float *__restrict__ loc,
    TileGeom G, uint4 *__restrict__ tiletab, unsigned *__restrict__ ecount, uint4 *__restrict__ extras,
    uint4 *__restrict__ recs, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn)
{
    constexpr unsigned D = 32;
    extern __shared__ unsigned lds_plan[];
    TLevel *lv = reinterpret_cast<TLevel *>(lds_plan);          // [kTMaxLevels]
    unsigned *misc = lds_plan + kTLevelWords * kTMaxLevels;     // [8]
    unsigned *scr = misc + 8;                                   // [kPlanThreads / 64 + 1]
    unsigned *split = scr + kPlanThreads / 64 + 1;              // [ecap] tiles of this level that are split
    unsigned *hist = split + G.ecap;                            // [tiles of this level] counts, then offsets / cursors

    // Every block of a unit walks all of the unit's samples but only keeps those in its own range of tiles: the
    // blocks need nothing from each other (a range has its own record region and extra-item slots).
    TSTAMP_DECL;
    const unsigned tid = threadIdx.x;
    const unsigned sub = blockIdx.x, unit = sub / kPlanSplit, part = sub - unit * kPlanSplit;
    const unsigned h = unit / G.L, l = unit - h * G.L;
    const unsigned b = fdiv(h, G.Mdiv), m = h - b * G.M;
    const unsigned NT = tile_levels(shapes, start, G.L, lv, misc);
    if (NT > G.ntmax) return;   // (cannot happen: ntmax bounds the tile count of any level table that tiles [0, S))
    const TLevel Lv = lv[l];
    const unsigned ntl_all = Lv.nty * Lv.ntx;
    const unsigned t_lo = (unsigned)(((unsigned long long)ntl_all * part) / kPlanSplit);
    const unsigned ntl = (unsigned)(((unsigned long long)ntl_all * (part + 1)) / kPlanSplit) - t_lo;   // tiles t_lo .. t_lo + ntl of the level
    for (unsigned i = tid; i < ntl; i += kPlanThreads) hist[i] = 0;
    if (tid < 2) misc[2 + tid] = 0;
    __syncthreads();
    TSTAMP(0);

    // pass 1: samples per tile (a sample counts in every tile one of its corners falls into); zero gradients for
    // samples outside (-1, H) x (-1, W) (cuh:365-367)
    unsigned cellv[4], lwv[4], lhv[4], trk[4][4];   // kOnePass: cell word, fractions, rank inside each touched tile
    if (kOnePass) {
#pragma unroll
        for (unsigned p = 0; p < 4; ++p) {
            cellv[p] = kInvalidCell;
#pragma unroll
            for (unsigned i = 0; i < 4; ++i) trk[p][i] = kInvalidItem;
        }
    }
    for (unsigned q = tid; q < G.Q; q += kPlanThreads) {
        const size_t base = ((size_t)(b * G.Q + q) * G.M + m) * G.LP + (size_t)l * G.P;
        if (kOnePass) {
            float2 xy[4];
            if (G.P == 4) {   // 32 contiguous, 32-byte aligned bytes: two 16-byte loads
                const float4 u0 = *reinterpret_cast<const float4 *>(loc + base * 2), u1 = *reinterpret_cast<const float4 *>(loc + base * 2 + 4);
                xy[0] = make_float2(u0.x, u0.y); xy[1] = make_float2(u0.z, u0.w);
                xy[2] = make_float2(u1.x, u1.y); xy[3] = make_float2(u1.z, u1.w);
            } else {
#pragma unroll
                for (unsigned p = 0; p < 4; ++p)
                    xy[p] = p < G.P ? *reinterpret_cast<const float2 *>(loc + (base + p) * 2) : make_float2(-9.f, -9.f);
            }
#pragma unroll
            for (unsigned p = 0; p < 4; ++p) {
                const Cell c = cell_of(xy[p].x, xy[p].y, Lv.H, Lv.W);
                if (c.valid) {
                    cellv[p] = ((unsigned)c.cy << 16) | (unsigned)c.cx;
                    lwv[p] = __float_as_uint(c.lw);
                    lhv[p] = __float_as_uint(c.lh);
                    unsigned i = 0;
                    for_each_touched_tile(c, Lv, [&](unsigned t, unsigned, unsigned, bool) {
                        if (t - t_lo < ntl) {
                            const unsigned w = atomicAdd(&hist[t - t_lo], 1u);   // rank inside the tile
                            // (the first call is the home tile; i is a compile-time constant after inlining)
                            if (i == 0) trk[p][0] = w; else if (i == 1) trk[p][1] = w; else if (i == 2) trk[p][2] = w; else trk[p][3] = w;
                        }
                        ++i;
                    });
                } else if (p < G.P && part == 0) {
                    grad_attn[base + p] = 0.f;
                    *reinterpret_cast<float2 *>(grad_loc + (base + p) * 2) = make_float2(0.f, 0.f);
                }
            }
        } else {
            for (unsigned p = 0; p < G.P; ++p) {
                const float2 xy = *reinterpret_cast<const float2 *>(loc + (base + p) * 2);
                const Cell c = cell_of(xy.x, xy.y, Lv.H, Lv.W);
                if (c.valid) {
                    for_each_touched_tile(c, Lv, [&](unsigned t, unsigned, unsigned, bool) { if (t - t_lo < ntl) atomicAdd(&hist[t - t_lo], 1u); });
                } else if (part == 0) {
                    grad_attn[base + p] = 0.f;
                    *reinterpret_cast<float2 *>(grad_loc + (base + p) * 2) = make_float2(0.f, 0.f);
                }
            }
        }
    }
    __syncthreads();
    TSTAMP(1);

    // per tile: {offset, count} of its records, query shares, extra work items for shares 1..K-1
    uint4 *tt = tiletab + (size_t)h * G.ntmax + Lv.tbase + t_lo;
    uint4 *ex = extras + (size_t)sub * G.ecap;
    const unsigned rbase = part * G.Q * G.P * 4;   // this range's record region inside the unit's
    unsigned mine[4];   // the counts of up to 4 tiles per thread survive the scan (ntl <= 4 * kPlanThreads)
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) mine[r] = tid + r * kPlanThreads < ntl ? hist[tid + r * kPlanThreads] : 0u;
    __syncthreads();
    block_scan_inplace(hist, ntl, scr);
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned t = tid + r * kPlanThreads;
        if (t >= ntl) break;
        const unsigned off = rbase + hist[t], n = mine[r];
        unsigned K = (n + G.cap - 1) / G.cap;
        K = K < 1 ? 1u : (K > kKmax ? kKmax : K);
        // an item header: {first record, records, tile origin y | x << 16, level | split << 8}: nothing left to divide
        const unsigned tl = t_lo + t, tyy = tl / Lv.ntx, txx = tl - tyy * Lv.ntx;
        const unsigned org = (tyy * kTH) | ((txx * kTW) << 16);
        tt[t] = make_uint4(off, K > 1 ? 0u : n, org, l | (K > 1 ? 256u : 0u));   // (a split tile's own item idles)
        if (K > 1) {
            const unsigned pos = atomicAdd(&misc[2], K);   // all K shares of a split tile are extra items
            for (unsigned k = 0; k < K; ++k) {
                const unsigned e0 = (unsigned)(((unsigned long long)n * k) / K), e1 = (unsigned)(((unsigned long long)n * (k + 1)) / K);
                if (pos + k < G.ecap) ex[pos + k] = make_uint4(off + e0, e1 - e0, org, l);
            }
            const unsigned sp = atomicAdd(&misc[3], 1u);
            if (sp < G.ecap) split[sp] = t;
        }
    }
    __syncthreads();
    TSTAMP(2);

    // pass 2: the records, tile by tile
    uint4 *rc = recs + (size_t)sub * G.Q * G.P * 4;
    if (kOnePass) {
        if (tid < G.Q) {
#pragma unroll
            for (unsigned p = 0; p < 4; ++p) {
                if (cellv[p] == kInvalidCell) continue;
                const unsigned sid = tid * G.P + p;
                Cell c;
                c.valid = true;
                c.cy = (int)(cellv[p] >> 16);
                c.cx = (int)(cellv[p] & 0xFFFFu);
                unsigned i = 0;
                for_each_touched_tile(c, Lv, [&](unsigned t, unsigned ty, unsigned tx, bool home) {   // (the same order as in pass 1)
                    const unsigned rank = i == 0 ? trk[p][0] : (i == 1 ? trk[p][1] : (i == 2 ? trk[p][2] : trk[p][3]));
                    if (t - t_lo < ntl)
                        rc[hist[t - t_lo] + rank] = make_record(sid, (unsigned)c.cy, (unsigned)c.cx, lwv[p], lhv[p], Lv, ty, tx, home);
                    ++i;
                });
            }
        }
    } else {
        for (unsigned q = tid; q < G.Q; q += kPlanThreads) {
            const size_t base = ((size_t)(b * G.Q + q) * G.M + m) * G.LP + (size_t)l * G.P;
            for (unsigned p = 0; p < G.P; ++p) {
                const float2 xy = *reinterpret_cast<const float2 *>(loc + (base + p) * 2);
                const Cell c = cell_of(xy.x, xy.y, Lv.H, Lv.W);
                if (c.valid) {
                    const unsigned sid = q * G.P + p;
                    for_each_touched_tile(c, Lv, [&](unsigned t, unsigned ty, unsigned tx, bool home) {
                        if (t - t_lo >= ntl) return;
                        const unsigned pos = atomicAdd(&hist[t - t_lo], 1u);
                        rc[pos] = make_record(sid, (unsigned)c.cy, (unsigned)c.cx, __float_as_uint(c.lw), __float_as_uint(c.lh), Lv, ty, tx, home);
                    });
                }
            }
        }
    }
    TSTAMP(3);
    const unsigned ne = misc[2] < G.ecap ? misc[2] : G.ecap, ns = misc[3] < G.ecap ? misc[3] : G.ecap;
    if (tid == 0) ecount[sub] = ne;
    // the shares of a split tile meet through fp32 atomics: its pixels start at zero
    float *gvl = grad_value + (((size_t)b * G.S + Lv.st) * G.M + m) * D;
    for (unsigned s = 0; s < ns; ++s) {
        const unsigned t = t_lo + split[s];
        const unsigned ty = t / Lv.ntx, tx = t - ty * Lv.ntx;
        for (unsigned i = tid; i < kTH * kTW * (D / 4); i += kPlanThreads) {
            const unsigned c4 = i % (D / 4), pix = i / (D / 4);
            const unsigned y = ty * kTH + pix / kTW, x = tx * kTW + pix % kTW;
            if (y < (unsigned)Lv.H && x < (unsigned)Lv.W)
                *reinterpret_cast<float4 *>(gvl + ((size_t)y * Lv.W + x) * G.M * D + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    TSTAMP(4);
    TSTAMP_COUNT(8);
    PSTAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------
// accumulate
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_addf(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
// sum over the 8 consecutive lanes of a sample group; every lane gets the total
__device__ __forceinline__ float group_sum8(float x)
{
    x = dpp_addf<0xB1>(x);    // quad_perm:[1,0,3,2]
    x = dpp_addf<0x4E>(x);    // quad_perm:[2,3,0,1]
    x = dpp_addf<0x141>(x);   // row_half_mirror
    return x;
}
__device__ __forceinline__ float dot4f(float4 a, float4 b)
{
    float acc = a.x * b.x;
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}
__device__ __forceinline__ unsigned uni(unsigned x) { return __builtin_amdgcn_readfirstlane(x); }


// what a record says about its sample in this tile (the same in the 8 lanes of the sample's group)
struct Dec {
    float lw, lh;
    unsigned rows;        // accumulator row of each corner (a byte each; kNoRow when not in this tile)
    unsigned pixb;        // byte offset of the clamped top-left pixel's value row in the level (home samples), else 0
    unsigned fl;          // bit 0 home, bit 1 x step, bit 2 y step, bits 3-6 corner-in-map masks, bit 7 live
    unsigned oi;          // index of the sample in grad_attn (x 2 in grad_loc), relative to query 0 of the head
    unsigned gob;         // byte offset of the query's grad_out row, relative to query 0 of the head
};
struct Ld {
    float a;
    float4 g, v00, v01, v10, v11;
};
// a work item: (head, level, tile, share k of K); everything block-uniform (scalar registers)
struct Item {
    unsigned n, ty0, tx0, the, twe, H, W, lP;
    unsigned live;        // 0: the own item of a split tile (its shares are extra items): nothing to sum, nothing to write
    unsigned hq;          // item index of query 0 of the head: b * Q * M + m
    size_t voff;          // float offset of the level's pixel 0, this head, in value / grad_value
    size_t roff;          // first record
};
template <typename T>
__device__ __forceinline__ T ldg(const void *base, unsigned byte_off)   // scalar base + 32-bit vector offset
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

__global__ __launch_bounds__(kAccThreads, ZIRA_TILE_BLOCKS_PER_CU) void msda_bwd_tile_accum(
    const float *__restrict__ grad_out, const float *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ start, const float *__restrict__ attn, TileGeom G,
    const uint4 *__restrict__ tiletab, const unsigned *__restrict__ ecount, const uint4 *__restrict__ extras,
    const uint4 *__restrict__ recs, float *__restrict__ dump, float *__restrict__ grad_value,
    float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr unsigned D = 32, LPS = 8, NTHR = kAccThreads, NW = NTHR / 64, NPIX = kTH * kTW, NG = 8;
    constexpr unsigned SPB = NW * NG;     // samples per block step
    static_assert(NPIX < kNoRow, "a corner's accumulator row is a byte");
    extern __shared__ double lds_acc[];
    double *acc = lds_acc;                                                   // [(NPIX + NG) * D]: tile, then a trash row per group
    unsigned *words = reinterpret_cast<unsigned *>(acc + (NPIX + NG) * D);
    TLevel *lv = reinterpret_cast<TLevel *>(words);                          // [kTMaxLevels]
    unsigned *misc = words + kTLevelWords * kTMaxLevels;                     // [8]
    unsigned *epre = misc + 8;                                               // [units of this block's heads + 1]: extras before unit u

    TSTAMP_DECL;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned grp = lane / LPS, j = lane % LPS;
    const unsigned odd = grp & 1;
    // item ranges: XCD x works through heads [x hp, (x + 1) hp) (the value slice of a head fits its L2);
    // placement is for speed only
    unsigned v, vstep, hbase, nh;
    if (G.hp) {
        const unsigned xcd = blockIdx.x & 7;
        hbase = xcd * G.hp;
        nh = hbase >= G.heads ? 0u : (G.heads - hbase < G.hp ? G.heads - hbase : G.hp);
        v = blockIdx.x >> 3;
        vstep = gridDim.x >> 3;
    } else {
        hbase = 0;
        nh = G.heads;
        v = blockIdx.x;
        vstep = gridDim.x;
    }
    const unsigned nunits = nh * G.L * kPlanSplit;   // (head, level, tile range): what a plan block wrote
    for (unsigned u = tid; u < nunits; u += NTHR) epre[u + 1] = ecount[(size_t)hbase * G.L * kPlanSplit + u];
    const unsigned NT = tile_levels(shapes, start, G.L, lv, misc);
    if (NT > G.ntmax) return;
    if (tid == 0) {
        unsigned run = 0;
        epre[0] = 0;
        for (unsigned u = 0; u < nunits; ++u) {
            run += epre[u + 1];
            epre[u + 1] = run;
        }
    }
    __syncthreads();
    // (scalar integer division is a long software loop on the CU's one scalar unit: the items below divide by
    // multiplication with constants formed once per block)
    const float rNT = 1.f / (float)NT;
    const unsigned E = epre[nunits];              // extra items (the shares of split tiles) of this block's heads: they come first
    const unsigned rs = G.M * D;                  // floats between pixels
    const unsigned trashb = (NPIX + grp) * D * 8; // byte offset of this group's trash row
    const unsigned gsel = wave * NG + grp;        // this group's sample inside a block step
    const unsigned mlp = G.M * G.LP;
    TSTAMP(0);

    // header of an item: {first record, records, tile origin, level | idle << 8} and its head (loads issued here, used later)
    struct Hdr {
        uint4 w;
        unsigned h;
    };
    auto issue_hdr = [&](auto ex, unsigned idx) {
        Hdr hd;
        if (decltype(ex)::value) {   // idx-th extra item of this block's heads
            unsigned u = 0;
            for (unsigned uu = 1; uu < nunits; ++uu) u = idx >= epre[uu] ? uu : u;
            hd.h = hbase + fdiv(u, G.LSdiv);
            hd.w = extras[((size_t)hbase * G.L * kPlanSplit + u) * G.ecap + (idx - epre[u])];
        } else {                     // idx-th tile of this block's heads
            unsigned hl = (unsigned)((float)idx * rNT);      // idx / NT: estimate and fix-up (idx < 2^24)
            hl = hl * NT > idx ? hl - 1 : ((hl + 1) * NT <= idx ? hl + 1 : hl);
            const unsigned t = NT - 1 - (idx - hl * NT);    // coarse levels (most samples per tile) first
            hd.h = hbase + hl;
            hd.w = tiletab[(size_t)hd.h * G.ntmax + t];
        }
        return hd;
    };
    auto make_item = [&](const Hdr &hd) {
        Item it;
        const unsigned off = uni(hd.w.x), cnt = uni(hd.w.y), org = uni(hd.w.z), lw = uni(hd.w.w), h = uni(hd.h);
        const unsigned l = lw & 255u;
        it.live = (lw >> 8) ? 0u : 1u;
        it.n = cnt;
        it.ty0 = org & 0xFFFFu;
        it.tx0 = org >> 16;
        it.H = uni((unsigned)lv[l].H);
        it.W = uni((unsigned)lv[l].W);
        it.the = it.H - it.ty0 < kTH ? it.H - it.ty0 : kTH;   // rows / columns of the tile inside the map
        it.twe = it.W - it.tx0 < kTW ? it.W - it.tx0 : kTW;
        const unsigned b = fdiv(h, G.Mdiv), m = h - b * G.M;
        it.roff = ((size_t)h * G.L + l) * kPlanSplit * G.Q * G.P * 4 + off;
        it.voff = (((size_t)b * G.S + uni(lv[l].st)) * G.M + m) * D;
        it.hq = b * G.Q * G.M + m;
        it.lP = l * G.P;
        return it;
    };
    // Every load and store of the item pipeline is issued unconditionally (clamped or redirected addresses): loads
    // and stores share one in-order counter on gfx950, and the compiler can only let a wave wait for exactly the
    // load it needs when it knows how many memory operations were issued after it.
    auto fetch = [&](const Item &it, unsigned s) {   // the record of this group at block step s
        const unsigned e = s * SPB + gsel;
        return ldg<uint4>(recs + it.roff, (e < it.n ? e : 0u) * 16u);   // (record 0 is readable for every item: the workspace ends with a pad)
    };
    auto decode = [&](const Item &it, const uint4 &r, unsigned s) {
        Dec d;
        const bool ok = s * SPB + gsel < it.n;
        const unsigned sid = r.x & 0x3FFFFu, hpos = (r.x >> 18) & 127u;
        const unsigned fl = ok ? ((r.x >> 25) & ((ZIRA_TILE_ABL & 2) ? 0x7Eu : 0x7Fu)) | 0x80u : 0u;
        const unsigned q = fdiv(sid, G.Pdiv), p = sid - q * G.P;
        // the clamped top-left pixel = the clamped bottom-right one minus the steps
        const unsigned py = it.ty0 + hpos / kTW - ((fl >> 2) & 1u), px = it.tx0 + hpos % kTW - ((fl >> 1) & 1u);
        d.lw = __uint_as_float(r.z);
        d.lh = __uint_as_float(r.w);
        d.rows = (ok && !(ZIRA_TILE_ABL & 1)) ? r.y : 0xFFFFFFFFu;
        d.pixb = ((fl & 1u) && !(ZIRA_TILE_ABL & 8)) ? (py * it.W + px) * (rs * 4u) : 0u;
        d.fl = fl;
        d.oi = ok ? q * mlp + it.lP + p : 0u;
        d.gob = (ok && !(ZIRA_TILE_ABL & 16)) ? q * (rs * 4u) : 0u;
        return d;
    };
    auto issue = [&](const Item &it, const Dec &d) {
        Ld x;
        x.a = ldg<float>(attn + (size_t)it.hq * G.LP, d.oi * 4u);
        x.g = ldg<float4>(grad_out + (size_t)it.hq * D, d.gob + j * 16u);
        // (samples in a neighbour's halo read pixel 0's rows and ignore them)
        const float *vb = value + it.voff;
        const unsigned o = d.pixb + j * 16u;
        const unsigned dx = (d.fl & 2u) ? rs * 4u : 0u, dy = (d.fl & 4u) ? it.W * rs * 4u : 0u;
        x.v00 = ldg<float4>(vb, o);
        x.v01 = ldg<float4>(vb, o + dx);
        x.v10 = ldg<float4>(vb, o + dy);
        x.v11 = ldg<float4>(vb, o + dy + dx);
        return x;
    };
    auto compute = [&](const Item &it, const Dec &d, const Ld &x) {
        const float lw = d.lw, lh = d.lh;
        const float a = (d.fl & 0x80u) ? x.a : 0.f;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float w00 = __fmul_rn(hh, hw), w01 = __fmul_rn(hh, lw), w10 = __fmul_rn(lh, hw), w11 = __fmul_rn(lh, lw);
        const float4 g4 = x.g;
        {   // home tile of the sample: grad_attn_weight, grad_sampling_loc (cuh:123-158); the others store to a dump word
            const float p00 = (d.fl & 8u) ? dot4f(g4, x.v00) : 0.f, p01 = (d.fl & 16u) ? dot4f(g4, x.v01) : 0.f;
            const float p10 = (d.fl & 32u) ? dot4f(g4, x.v10) : 0.f, p11 = (d.fl & 64u) ? dot4f(g4, x.v11) : 0.f;
            float ga = __fmul_rn(w00, p00);
            ga = fmaf(w01, p01, ga);
            ga = fmaf(w10, p10, ga);
            ga = fmaf(w11, p11, ga);
            float gx = fmaf(hh, __fsub_rn(p01, p00), __fmul_rn(lh, __fsub_rn(p11, p10)));
            float gy = fmaf(hw, __fsub_rn(p10, p00), __fmul_rn(lw, __fsub_rn(p11, p01)));
            ga = group_sum8(ga);
            gx = group_sum8(gx);
            gy = group_sum8(gy);
            const bool st = (d.fl & 1u) && j == 0;
            float *ga_h = grad_attn + (size_t)it.hq * G.LP, *gl_h = grad_loc + (size_t)it.hq * G.LP * 2;
            *(st ? ga_h + d.oi : dump + lane) = ga;
            *reinterpret_cast<float2 *>(st ? gl_h + 2 * (size_t)d.oi : dump + 64 + 2 * lane) =
                make_float2(__fmul_rn(__fmul_rn((float)it.W, a), gx), __fmul_rn(__fmul_rn((float)it.H, a), gy));
        }
        // corner rows: term = w * (a * g), both factors rounded to fp32 as the reference forms them, the product
        // and the sum in double.  Accumulator word kk * 8 + j holds channel 4 j + kk; the odd group of a 16-lane
        // row swaps kk 0 <-> 1 and 2 <-> 3 so that its eight 8-byte words fall into the other half of the banks.
        const float gs0 = odd ? g4.y : g4.x, gs1 = odd ? g4.x : g4.y, gs2 = odd ? g4.w : g4.z, gs3 = odd ? g4.z : g4.w;
        const double tt[4] = {(double)__fmul_rn(gs0, a), (double)__fmul_rn(gs1, a), (double)__fmul_rn(gs2, a),
                              (double)__fmul_rn(gs3, a)};
        const double wc[4] = {(double)w00, (double)w01, (double)w10, (double)w11};
        const unsigned o0 = (j + (odd ? 8u : 0u)) * 8u, o1 = (j + (odd ? 0u : 8u)) * 8u;
#pragma unroll
        for (unsigned cc = 0; cc < 4; ++cc) {
            const unsigned row = (d.rows >> (8 * cc)) & 255u;
            char *ap = reinterpret_cast<char *>(acc) + (row == kNoRow ? trashb : row * (D * 8u));
            atomicAdd(reinterpret_cast<double *>(ap + o0), wc[cc] * tt[0]);
            atomicAdd(reinterpret_cast<double *>(ap + o1), wc[cc] * tt[1]);
            atomicAdd(reinterpret_cast<double *>(ap + o0 + 128), wc[cc] * tt[2]);
            atomicAdd(reinterpret_cast<double *>(ap + o1 + 128), wc[cc] * tt[3]);
        }
    };

    // Items are pipelined: the header of item i + 1 is requested when item i starts; its first records are fetched and
    // the loads of its first step issued before item i is written out, so that an item begins with its operands in flight.
    // Two passes over this block's share of the items: the shares of split tiles (atomic write-out), then the tiles
    // (exactly four stores per thread, so that the wait for the next item's operands need not cover them).
    auto run = [&](auto ex, unsigned v, const unsigned vend) {
        constexpr bool kEx = decltype(ex)::value;
        if (v >= vend) return;
        Item it = make_item(issue_hdr(ex, v));
        uint4 r1 = fetch(it, 0), r2 = fetch(it, 1);
        Dec d0 = decode(it, r1, 0);
        Ld x0 = issue(it, d0);
        r1 = r2;
        r2 = fetch(it, 2);
        TSTAMP(1);
        for (;;) {
            const unsigned vn = v + vstep;
            const bool more = vn < vend;
            Hdr hn = issue_hdr(ex, more ? vn : v);
            __syncthreads();   // (the previous item's LDS is no longer read)
            for (unsigned x = tid; x < NPIX * D / 2; x += NTHR) reinterpret_cast<uint4 *>(acc)[x] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            TSTAMP(2);
            // (wave w holds records 8 w .. 8 w + 7 of every 32: a wave without records at a step skips it)
            const unsigned nsteps = it.n > wave * NG ? (it.n - wave * NG + SPB - 1) / SPB : 0u;
            for (unsigned s = 0; s < nsteps; s += 2) {   // two steps in flight: the loads of step s + 1 are issued before step s is
                const Dec d1 = decode(it, r1, s + 1);      // summed (two copies of the body: no register moves between steps)
                const Ld x1 = issue(it, d1);
                r1 = fetch(it, s + 3);
                compute(it, d0, x0);
                if (s + 1 >= nsteps) break;
                d0 = decode(it, r2, s + 2);
                x0 = issue(it, d0);
                r2 = fetch(it, s + 4);
                compute(it, d1, x1);
            }
            TSTAMP(3);
            const Item nx = make_item(hn);
            r1 = fetch(nx, 0);
            r2 = fetch(nx, 1);
            __syncthreads();
            d0 = decode(nx, r1, 0);
            x0 = issue(nx, d0);
            r1 = r2;
            r2 = fetch(nx, 2);
            TSTAMP(4);

            // ---- write-out -------------------------------------------------------------------------
            if (!(ZIRA_TILE_ABL & 4)) {
                if (!kEx) {   // every pixel of the tile once, plain stores (grad_value is never zero-filled)
#pragma unroll
                    for (unsigned x = tid; x < NPIX * LPS; x += NTHR) {
                        const unsigned c4 = x % LPS, pix = x / LPS, r = pix / kTW, c = pix - r * kTW;
                        const double *ap = acc + pix * D + c4;
                        const float4 o = make_float4((float)ap[0], (float)ap[LPS], (float)ap[2 * LPS], (float)ap[3 * LPS]);
                        float *dst = (r < it.the && c < it.twe && it.live)
                                         ? grad_value + it.voff + ((size_t)(it.ty0 + r) * it.W + (it.tx0 + c)) * rs + c4 * 4
                                         : dump + 192 + 4 * lane;   // (pixels of an edge tile outside the map; idle items)
                        *reinterpret_cast<float4 *>(dst) = o;
                    }
                } else {      // a share of a split tile: its non-zero pixels are added to rows the plan kernel zeroed
                    for (unsigned x = tid; x < NPIX * D; x += NTHR) {
                        const unsigned ch = x % D, pix = x / D, r = pix / kTW, c = pix - r * kTW;
                        if (r >= it.the || c >= it.twe) continue;
                        const float o = (float)acc[pix * D + (ch & 3u) * LPS + (ch >> 2)];
                        if (o != 0.f) unsafeAtomicAdd(grad_value + it.voff + ((size_t)(it.ty0 + r) * it.W + (it.tx0 + c)) * rs + ch, o);
                    }
                }
            }
            TSTAMP(5);
            TSTAMP_COUNT(8);
            if (!more) break;
            it = nx;
            v = vn;
        }
    };
    run(std::true_type{}, v, E);
    run(std::false_type{}, (v + vstep - E % vstep) % vstep, nh * NT);   // (the deal goes on where the extra items left it: item counts differ by one at most)
    TSTAMP_FLUSH;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct TilesLayout {
    TileGeom G;
    size_t off_tiletab, off_ecount, off_extras, off_dump, off_recs, total;
    bool one_pass;
    size_t lds_plan, lds_acc;
};

inline bool make_tiles_layout(int B, int S, int M, int D, int L, int Q, int P, TilesLayout &T)
{
    if (D != 32 || L > (int)kTMaxLevels) return false;
    const unsigned long long heads = (unsigned long long)B * M;
    if ((unsigned long long)S * M * D >= (1ull << 31) || (unsigned long long)Q * M * L * P * 2 >= (1ull << 31)) return false;
    if (heads * L >= (1ull << 20) || (unsigned long long)Q * P >= (1ull << 18)) return false;   // 18-bit sample field of a record
    if ((unsigned long long)S * M * D * 4 >= (1ull << 32) || (unsigned long long)Q * M * D * 4 >= (1ull << 32)) return false;   // 32-bit byte offsets
    TileGeom &G = T.G;
    G.B = (unsigned)B; G.S = (unsigned)S; G.M = (unsigned)M; G.L = (unsigned)L; G.Q = (unsigned)Q; G.P = (unsigned)P;
    G.LP = (unsigned)(L * P);
    G.heads = (unsigned)heads;
    // tiles of a level with n pixels: ceil(H / TH) * ceil(W / TW) <= n / min(TH, TW) + 1 for every H * W = n
    const unsigned tmin = kTH < kTW ? kTH : kTW;
    G.ntmax = (unsigned)S / tmin + (unsigned)L;
    if (G.ntmax >= (1u << 20)) return false;           // 20-bit tile field of an item word
    G.cap = ZIRA_TILE_CAP;
    G.ecap = (unsigned)(((unsigned long long)Q * P * 8) / G.cap) + 1;   // sum of K over split tiles <= 2 records / cap; a sample has <= 4 records
    G.hp = heads >= 8 ? (unsigned)((heads + 7) / 8) : 0u;
    G.Mdiv = make_fdiv((unsigned)M);
    G.Pdiv = make_fdiv((unsigned)P);
    G.LSdiv = make_fdiv((unsigned)L * kPlanSplit);
    if (heads * G.ntmax >= (1ull << 24)) return false;   // (tile items are indexed through a float estimate)
    if (G.ntmax > 4 * kPlanThreads) return false;      // (the plan kernel keeps a level's tile counts in registers across its scan)
    T.lds_plan = (kTLevelWords * kTMaxLevels + 8 + kPlanThreads / 64 + 1 + (size_t)G.ecap + G.ntmax) * 4;
    if (T.lds_plan > 64 * 1024) return false;
    const size_t units_per_block = (size_t)(G.hp ? G.hp : G.heads) * L * kPlanSplit;
    T.lds_acc = (size_t)(kTH * kTW + 8) * 32 * 8 + (kTLevelWords * kTMaxLevels + 8 + units_per_block + 1) * 4;
    if (T.lds_acc > 40 * 1024) return false;           // (four blocks per CU)
    T.one_pass = (unsigned)Q <= kPlanThreads && P <= 4;
    size_t o = 0;
    T.off_tiletab = o; o += align256(heads * G.ntmax * 16);
    T.off_ecount = o;  o += align256(heads * L * kPlanSplit * 4);
    T.off_extras = o;  o += align256(heads * L * kPlanSplit * G.ecap * 16);
    T.off_dump = o;    o += 2048;                                           // where redirected stores go (never read)
    // a sample has a record in every tile it touches (<= 4), and any tile range of a unit may receive all of them; pad
    T.off_recs = o;    o += align256(heads * L * kPlanSplit * (size_t)Q * P * 4 * 16) + 256;
    T.total = o;
    return true;
}

inline unsigned tiles_cu_count()
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

size_t tiles_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P)
{
    TilesLayout T;
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0) return 0;
    return make_tiles_layout(B, S, M, D, L, Q, P, T) ? T.total : 0;
}

int tiles_backward_f32(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start,
                       const float *loc, const float *attn, int B, int S, int M, int D, int L, int Q, int P,
                       float *gv, float *gl, float *ga, void *ws, size_t ws_bytes, hipStream_t st)
{
    TilesLayout T;
    if (!make_tiles_layout(B, S, M, D, L, Q, P, T) || !ws || ws_bytes < T.total || ((uintptr_t)ws & 15)) return -1;
    char *w = reinterpret_cast<char *>(ws);
    uint4 *tiletab = reinterpret_cast<uint4 *>(w + T.off_tiletab);
    unsigned *ecount = reinterpret_cast<unsigned *>(w + T.off_ecount);
    uint4 *extras = reinterpret_cast<uint4 *>(w + T.off_extras);
    uint4 *recs = reinterpret_cast<uint4 *>(w + T.off_recs);
    if (T.one_pass)
        hipLaunchKernelGGL(msda_bwd_plan<true>, dim3(T.G.heads * T.G.L * kPlanSplit), dim3(kPlanThreads), T.lds_plan, st, shapes, start,
                           loc, T.G, tiletab, ecount, extras, recs, gv, gl, ga);
    else
        hipLaunchKernelGGL(msda_bwd_plan<false>, dim3(T.G.heads * T.G.L * kPlanSplit), dim3(kPlanThreads), T.lds_plan, st, shapes, start,
                           loc, T.G, tiletab, ecount, extras, recs, gv, gl, ga);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const unsigned grid = tiles_cu_count() * ZIRA_TILE_BLOCKS_PER_CU;
    hipLaunchKernelGGL(msda_bwd_tile_accum, dim3(grid), dim3(kAccThreads), T.lds_acc, st, grad_out, value, shapes,
                       start, attn, T.G, tiletab, ecount, extras, recs, reinterpret_cast<float *>(w + T.off_dump), gv, gl, ga);
    return (int)hipGetLastError();
}

}  // namespace zira

#if ZIRA_TILE_STAMPS
extern "C" int zira_dev_read_tile_stamps(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(zira::zira_tile_stamps), sizeof(unsigned long long) * n);
}
extern "C" int zira_dev_read_plan_stamps(unsigned long long *host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(zira::zira_plan_stamps), sizeof(unsigned long long) * n);
}
#endif
