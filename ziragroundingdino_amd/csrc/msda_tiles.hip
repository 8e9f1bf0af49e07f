// msda_tiles.hip -- backward of multi-scale deformable attention for gfx950 (MI355X), sparse calls
// (decoder cross-attention: a few hundred queries per image): "plan (at forward time) + tile accumulate".
//
// Arithmetic to match: reference csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:87-159 inside :301-403
// (grad_value += w_corner * (attn * grad_out), grad_attn = <grad_out, sample>, grad_loc from the corner
// differences).  The decomposition is not the reference's.
//
// Why this shape.  `ds_add_f64` takes 8.6 cycles per wave instruction on MI355X whatever the addresses are
// (scripts/lds_atomic_rates2.hip; `ds_add_f32` takes 193), so a 16 x 8-pixel tile of grad_value can simply be SUMMED
// IN LDS IN DOUBLE, in any order, and a pile-up of the decoder's queries on a few pixels costs nothing extra.  What
// has to be organised is which samples go to which tile and how the tiles are dealt to the CUs -- and all of that
// depends on the sampling locations (and, for the records' weights, the attention weights) only, which are known in the
// FORWARD pass.  Round 3 planned inside the backward
// call (15 us on 64 CUs in front of the accumulate kernel); since round 4 the plan is made in the forward pass -- by
// the first 64 blocks of the forward's own grid (msda_fwd_plan / zira_msda_fwd_plan_f32; msda_plan is the same code as a
// kernel of its own) -- and the backward (zira_msda_bwd_planned_f32) starts with everything resolved:
//
//   plan   (plan_unit)  one 1024-thread block per (head, level) unit; touches only the sampling locations and attention weights.  Pass 1: a
//          thread per query computes the pixel cells of its samples and counts each sample into every tile its
//          2 x 2 corner block touches (LDS histogram; the rank inside the tile comes back from the atomic).  A block
//          scan turns the histogram into record offsets.  Per tile a 32-byte WORK ITEM {first record, records, tile
//          origin, level size, value row of the level's pixel 0, query 0 of the head, ...}; a tile with more than
//          `cap` records is SPLIT into K = ceil(n / cap) items whose sums meet in a small fold launch behind the
//          accumulate kernel.  Pass 2 writes one 16-byte RECORD per
//          (sample, touched tile) with everything the accumulate kernel needs already resolved: where the query's
//          grad_out row is, the attention weight, the LDS byte offsets of its four corner rows (a trash row for corners
//          outside this tile), lw, lh.  A sample outside (-1, H) x (-1, W) has no record.
//          The items of a unit are stored by size class (eight classes by record count, i.e. by 32-record block steps) and the
//          unit's counts published; nothing else crosses between plan blocks -- no atomics on global memory, no fence,
//          nothing to initialise.  The deal is made by the accumulate blocks themselves: a prefix over the class counts of
//          the units of a group of heads (= one XCD's share) lays all items of the group on one virtual ring, heavy
//          ones first, and block k takes position k of the even rounds of nbg positions and nbg - 1 - k of the odd ones:
//          longest-processing-time-first in eight steps, the same for every run.  (Measured on the way: a sort by the group's last plan block costs 6 us for the
//          device-scope release / acquire + 3 us for the sort; ring cursors in global memory need a clearing launch --
//          hipMemsetAsync is replayed out of order inside a hipGraph on ROCm 7.2 -- that rocprofv3 shows at 4.6 us.)
//   accum  (msda_bwd_tile_accum)  ONE launch with two kinds of blocks (256 threads): persistent accumulate blocks, 3 per CU,
//          and behind them gather blocks that pass through the fourth slot of every CU -- a wave per (b, q, m) forms
//          grad_sampling_loc / grad_attn_weight exactly as the forward forms its output (csrc/msda_fwd_lean.h:
//          bwd_home_item).  Accumulate, per item: 32 records per block step, 8 lanes x 4 channels per record: the grad_out
//          row (one 128-byte gather, the step's only dependent load) and the corner terms w * (attn * g) added to the
//          tile's accumulators in LDS (`ds_add_f64`).  Records are requested four steps ahead, rows two; all loads
//          unconditional (clamped record index) so that the compiler's in-order vmcnt waits stay exact; the next item's
//          first records are requested before the current item's steps.  (Until the middle of round 4 the accumulate
//          blocks also did the gather half for the tile that owns a sample -- attention weight, four value rows, dot
//          products -- from loads that depended on the record: two dependent round trips per step, 1.3 us per step,
//          35 us for the kernel; with the weight in the record and the gather half in its own waves a step is bound by
//          the LDS adds -- 13 us of steps per block for 9.3 us of `ds_add_f64` issue per CU -- and the launch takes 27.8 us
//          for both halves.)  Flush: every pixel of an unsplit tile once with 16-byte
//          stores (no zero-fill of grad_value anywhere), each thread clearing the accumulator words it read; empty
//          tiles are written as zeros without touching LDS; shares of split tiles store their sums to partial tiles (write-through stores) and count themselves in; the share that arrives last adds the partial tiles up in a fixed order and writes the tile (no second launch since round 5).
//
// The sums are formed in double from exact products of fp32 factors (w and attn * g rounded to fp32 as in
// the reference, cuh:117-147) and rounded to fp32 once: at least as close to the reference as an fp32
// accumulation in any order; inf / NaN propagate as they do there.  Geometry is derived on the device from
// the int64 level table (the C ABI has device pointers only); the host sizes the plan buffer from S.
// Precondition, as in the reference module (ms_deform_attn.py:284): the levels tile [0, S).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "dev/stamps.h"
#include "msda_fwd_lean.h"
#include "msda_internal.h"

#ifndef ZIRA_TILE_CAP
#define ZIRA_TILE_CAP 640      // records a work item holds at most (plan: K = ceil(records / cap); 320: +3 us, 224: +10 us)
#endif
#ifndef ZIRA_TILE_BLOCKS_PER_CU
#define ZIRA_TILE_BLOCKS_PER_CU 3   // accumulate blocks per CU ...
#endif
#ifndef ZIRA_TILE_SLOTS_PER_CU
#define ZIRA_TILE_SLOTS_PER_CU 4    // ... of the four that fit (LDS, registers): the fourth is where the gather blocks pass through
#endif
#ifndef ZIRA_HOME_ITEMS_PER_WAVE
#define ZIRA_HOME_ITEMS_PER_WAVE 1   // (b, q, m) items a gather wave of msda_bwd_tile_accum takes (1; 2: their loads interleaved)
#endif
#ifndef ZIRA_TILE_MIX_ORDER
#define ZIRA_TILE_MIX_ORDER 1        // every other accumulate block walks its items light-first (see msda_bwd_tile_accum)
#endif
#ifndef ZIRA_TILE_ROWS
#define ZIRA_TILE_ROWS 16      // pixel rows of a tile (x 8 columns)
#endif
#ifndef ZIRA_TILE_THREADS
#define ZIRA_TILE_THREADS 256  // threads of an accumulate block
#endif

namespace zira {
namespace {

constexpr unsigned kTMaxLevels = 16;
constexpr unsigned kKmax = 64;          // shares per tile at most
constexpr unsigned kTH = ZIRA_TILE_ROWS, kTW = 8;   // tile: 16 x 8 pixels
constexpr unsigned kNPix = kTH * kTW;
constexpr unsigned kRowBytes = 32 * 8;  // one accumulator row: 32 channels in double
constexpr unsigned kTrash = kNPix * kRowBytes;   // LDS byte offset of the trash row (corners owned by another tile)
static_assert(kTrash < 65536, "a corner's LDS offset is 16 bits of a record");
constexpr unsigned kAccThreads = ZIRA_TILE_THREADS;
constexpr unsigned kPlanThreads = 1024;
constexpr unsigned kClasses = 8;        // size classes of the deal, by block steps (32 records): 17+, 13-16, 9-12, 5-8, 3-4, 2, 1, none (empty tiles)
__host__ __device__ constexpr unsigned size_class(unsigned n)
{
    return n > 512 ? 0u : (n > 384 ? 1u : (n > 256 ? 2u : (n > 128 ? 3u : (n > 64 ? 4u : (n > 32 ? 5u : (n > 0 ? 6u : 7u))))));
}
constexpr unsigned kUcnt = 16;          // words a unit publishes: kClasses counts, its first item slot, spare
constexpr unsigned kNoCell = 0xFFFFFFFFu, kNoRank = 0xFFFFFFFFu;
constexpr unsigned kItemShare = 1u << 16;    // item flag: a share of a split tile (its sums go to a partial tile of the workspace);
                                             // a share's word b.z also holds (shares of the tile - 1) << 17 and its own number << 24
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

struct PlanGeom {
    unsigned B, S, M, L, Q, P, LP, heads;
    unsigned ntmax;     // tiles per head: upper bound from S
    unsigned cap;       // records a work item should hold
    unsigned ecap;      // share / split-tile slots per (head, level) unit
    unsigned rcap;      // record slots per unit: Q * P * 4
    unsigned iph;       // item slots per head: ntmax + L * ecap
    unsigned ng, hp;    // groups of heads (8: one per XCD, or 1), heads per group
    unsigned nbg;       // accumulate blocks per group
    unsigned ccap;      // item slots per size class: heads * iph
    FastDivT Mdiv, NBGdiv;
};

// The plan buffer (device memory, caller-owned):
struct PlanPtrs {
    unsigned *ucnt;     // [units][kUcnt] items of a unit per size class, and the unit's first item slot
    unsigned *tcnt;     // [units * ecap]       shares of a split tile that have stored their partial tile (indexed by the tile's first
                        //                      partial tile; zero between launches: the plan clears them, the last arriver resets its own)
    float *partial;     // [units * ecap][kNPix * 32]  sums of the shares of split tiles (written by the accumulate kernel)
    uint4 *citems;      // [kClasses][ccap][2]  items by size class, a unit's at its own slots
    uint4 *recs;        // [units * rcap]       16-byte records, sorted by tile inside a unit
};

struct TLevel {
    int H, W;
    unsigned st, nty, ntx, tbase;
};
constexpr unsigned kTLevelWords = sizeof(TLevel) / 4;

// Per-level tile grid from the device-side int64 tables into LDS.  Called by every thread of the block; ends with a
// barrier.  Returns the tiles per head.
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

// A 16-byte record = one sample in one tile its 2 x 2 corner block touches:
//   x = query (22 bits) | tile row of the top-left corner + 1 (5 bits) << 22 | its tile column + 1 (4 bits) << 27
//       (corners are at rows r - 1, r and columns c - 1, c of the sample's cell; a corner outside the tile or the map goes
//        to the tile's trash row: the accumulate block tests it against the tile's extent inside the map)
//   y = attention weight,  z = lw,  w = lh   (the bilinear fractions)
__device__ __forceinline__ uint4 make_record(unsigned q, unsigned abits, unsigned cy, unsigned cx, unsigned lw, unsigned lh,
                                             unsigned ty, unsigned tx)
{
    // tile-relative position of corner 00 is (cy - 1 - ty0, cx - 1 - tx0) in [-1, kTH) x [-1, kTW); stored + 1
    const unsigned pr = cy - ty * kTH, pc = cx - tx * kTW;   // = (cy - 1 - ty0) + 1 etc., in [0, kTH] x [0, kTW]
    return make_uint4(q | (pr << 22) | (pc << 27), abits, lw, lh);
}
static_assert(kTH <= 30 && kTW <= 14, "a corner's tile position is 5 + 4 bits of a record");

struct ItemCtx {
    unsigned HW, vrow, hq, lP, unit;
};
__device__ __forceinline__ void store_item(uint4 *dst, unsigned off, unsigned n, unsigned org, const ItemCtx &C, unsigned flags)
{
    dst[0] = make_uint4(off, n, org, C.HW);
    dst[1] = make_uint4(C.vrow, C.hq, C.lP | flags, C.unit);
}

// One query per thread and pass (kOnePass: Q <= kPlanThreads, P <= 4): the cells and the ranks inside their tiles stay
// in registers between the count and the copy-out, so the sampling locations are read once.
template <bool kOnePass>
__device__ __forceinline__ void plan_unit(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                          const float *__restrict__ loc, const float *__restrict__ attn, const PlanGeom &G,
                                          const PlanPtrs &W, const unsigned unit)
{
    constexpr unsigned D = 32;
    extern __shared__ unsigned lds_plan[];
    TLevel *lv = reinterpret_cast<TLevel *>(lds_plan);          // [kTMaxLevels]
    unsigned *misc = lds_plan + kTLevelWords * kTMaxLevels;     // [16]
    unsigned *scr = misc + 16;                                   // [kPlanThreads / 64 + 1]
    unsigned *hist = scr + kPlanThreads / 64 + 1;               // [tiles of this level] counts, then offsets; later the schedule's words

    TSTAMP_DECL;
    const unsigned tid = threadIdx.x;
    const unsigned h = unit / G.L, l = unit - h * G.L;
    const unsigned b = fdiv(h, G.Mdiv), m = h - b * G.M;
    const unsigned NT = tile_levels(shapes, start, G.L, lv, misc);
    if (NT > G.ntmax) {   // (a level table that does not tile [0, S): no items, nothing is accumulated)
        if (tid < kUcnt) W.ucnt[unit * kUcnt + tid] = 0;
        return;
    }
    const TLevel Lv = lv[l];
    const unsigned ntl = Lv.nty * Lv.ntx;
    for (unsigned i = tid; i < ntl; i += kPlanThreads) hist[i] = 0;
    if (tid < 16) misc[tid] = 0;
    __syncthreads();
    TSTAMP(0);

    // pass 1: records per tile (a sample counts in every tile one of its corners falls into; a sample outside the
    // window has no record: its gradients are zero, and msda_bwd_home writes them)
    unsigned cellv[4], lwv[4], lhv[4];   // kOnePass: cell word, fractions
    // ... and the rank inside each touched tile: 16 words per thread in LDS, word-major (no bank conflicts), so that the
    // kernel fits 64 registers and the gather blocks of msda_fwd_plan get two blocks per CU
    unsigned *trk = hist + G.ntmax + tid;   // trk[(p * 4 + i) * kPlanThreads]
    if (kOnePass) {
#pragma unroll
        for (unsigned p = 0; p < 4; ++p) {
            cellv[p] = kNoCell;
#pragma unroll
            for (unsigned i = 0; i < 4; ++i) trk[(p * 4 + i) * kPlanThreads] = kNoRank;
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
                    xy[p] = p < G.P ? *reinterpret_cast<const float2 *>(loc + (base + p) * 2) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (unsigned p = 0; p < 4; ++p) {
                if (p >= G.P) continue;
                const Cell c = cell_of(xy[p].x, xy[p].y, Lv.H, Lv.W);
                if (c.valid) {
                    cellv[p] = ((unsigned)c.cy << 16) | (unsigned)c.cx;
                    lwv[p] = __float_as_uint(c.lw);
                    lhv[p] = __float_as_uint(c.lh);
                    unsigned i = 0;
                    for_each_touched_tile(c, Lv, [&](unsigned t, unsigned, unsigned, bool) {
                        const unsigned w = atomicAdd(&hist[t], 1u);   // rank inside the tile
                        // (the first call is the home tile; i is a compile-time constant after inlining)
                        trk[(p * 4 + i) * kPlanThreads] = w;
                        ++i;
                    });
                }
            }
        } else {
            for (unsigned p = 0; p < G.P; ++p) {
                const float2 xy = *reinterpret_cast<const float2 *>(loc + (base + p) * 2);
                const Cell c = cell_of(xy.x, xy.y, Lv.H, Lv.W);
                if (c.valid)
                    for_each_touched_tile(c, Lv, [&](unsigned t, unsigned, unsigned, bool) { atomicAdd(&hist[t], 1u); });
            }
        }
    }
    __syncthreads();
    TSTAMP(1);

    // per tile: its work item(s), by size class
    ItemCtx IC;
    IC.HW = (unsigned)Lv.H | ((unsigned)Lv.W << 16);
    IC.vrow = (b * G.S + Lv.st) * G.M + m;
    IC.hq = b * G.Q * G.M + m;
    IC.lP = l * G.P;
    IC.unit = unit;
    for (unsigned i = tid; i < G.ecap; i += kPlanThreads) W.tcnt[(size_t)unit * G.ecap + i] = 0;
    const unsigned rbase = unit * G.rcap;   // this unit's record region
    unsigned mine[4];   // the counts of up to 4 tiles per thread survive the scan (ntl <= 4 * kPlanThreads)
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) mine[r] = tid + r * kPlanThreads < ntl ? hist[tid + r * kPlanThreads] : 0u;
    __syncthreads();
    block_scan_inplace(hist, ntl, scr);
    unsigned ioff[4], ipos[4], islot[4];   // first record; position inside the unit's items of its class | class << 28 | K << 20; first partial tile
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned t = tid + r * kPlanThreads;
        ioff[r] = ipos[r] = islot[r] = 0;
        if (t >= ntl) continue;
        const unsigned n = mine[r];
        unsigned K = (n + G.cap - 1) / G.cap;
        K = K < 1 ? 1u : (K > kKmax ? kKmax : K);
        const unsigned cls = size_class((n + K - 1) / K);   // (a share's size for split tiles)
        ioff[r] = rbase + hist[t];
        ipos[r] = atomicAdd(&misc[5 + cls], K) | (cls << 28) | (K << 20);
        if (K > 1) {   // split: the shares' sums go to K partial tiles of the workspace; the last share to finish adds them up
            const unsigned sl = atomicAdd(&misc[4], K);
            islot[r] = unit * G.ecap + sl;
        }
    }
    __syncthreads();
    TSTAMP(2);

    // pass 2: the records, tile by tile
    uint4 *rc = W.recs + (size_t)rbase;
    if (kOnePass) {
        if (tid < G.Q) {
            const size_t abase = ((size_t)(b * G.Q + tid) * G.M + m) * G.LP + (size_t)l * G.P;
            unsigned abv[4];
            if (G.P == 4) {   // 16 contiguous, aligned bytes
                const uint4 u = *reinterpret_cast<const uint4 *>(attn + abase);
                abv[0] = u.x; abv[1] = u.y; abv[2] = u.z; abv[3] = u.w;
            } else {
#pragma unroll
                for (unsigned p = 0; p < 4; ++p) abv[p] = p < G.P ? __float_as_uint(attn[abase + p]) : 0u;
            }
#pragma unroll
            for (unsigned p = 0; p < 4; ++p) {
                if (cellv[p] == kNoCell) continue;
                const unsigned ab = abv[p];
                Cell c;
                c.valid = true;
                c.cy = (int)(cellv[p] >> 16);
                c.cx = (int)(cellv[p] & 0xFFFFu);
                unsigned i = 0;
                for_each_touched_tile(c, Lv, [&](unsigned t, unsigned ty, unsigned tx, bool) {   // (the same order as in pass 1)
                    const unsigned rank = trk[(p * 4 + i) * kPlanThreads];
                    rc[hist[t] + rank] = make_record(tid, ab, (unsigned)c.cy, (unsigned)c.cx, lwv[p], lhv[p], ty, tx);
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
                    const unsigned ab = __float_as_uint(attn[base + p]);
                    for_each_touched_tile(c, Lv, [&](unsigned t, unsigned ty, unsigned tx, bool) {
                        const size_t o = (size_t)atomicAdd(&hist[t], 1u);   // (the offsets become cursors)
                        rc[o] = make_record(q, ab, (unsigned)c.cy, (unsigned)c.cx, __float_as_uint(c.lw), __float_as_uint(c.lh), ty, tx);
                    });
                }
            }
        }
    }
    TSTAMP(3);
    const unsigned ubase = h * G.iph + Lv.tbase + l * G.ecap;   // this unit's item slots (in each class array)
    if (tid < kClasses) W.ucnt[unit * kUcnt + tid] = misc[5 + tid];
    if (tid == kClasses) W.ucnt[unit * kUcnt + kClasses] = ubase;
    TSTAMP(4);
    // the items, by class, in the unit's own slots
#pragma unroll
    for (unsigned r = 0; r < 4; ++r) {
        const unsigned t = tid + r * kPlanThreads;
        if (t >= ntl) break;
        const unsigned cls = ipos[r] >> 28, K = (ipos[r] >> 20) & 255u, n = mine[r];
        const unsigned tyy = t / Lv.ntx, txx = t - tyy * Lv.ntx;
        const unsigned org = (tyy * kTH) | ((txx * kTW) << 16);
        for (unsigned k = 0; k < K; ++k) {
            const unsigned rr = ubase + (ipos[r] & 0xFFFFFu) + k;
            if (rr >= G.ccap) continue;   // (cannot happen: a unit has ntl + ecap slots)
            uint4 *dst = W.citems + ((size_t)cls * G.ccap + rr) * 2;
            const unsigned e0 = (unsigned)(((unsigned long long)n * k) / K), e1 = (unsigned)(((unsigned long long)n * (k + 1)) / K);
            IC.unit = K > 1 ? islot[r] + k : 0u;   // (word b.w of a share: its partial tile)
            store_item(dst, ioff[r] + e0, e1 - e0, org, IC, K > 1 ? (kItemShare | ((K - 1u) << 17) | (k << 24)) : 0u);
        }
    }
    TSTAMP(5);
    TSTAMP_COUNT(8);
    PSTAMP_FLUSH;
}

template <bool kOnePass>
__global__ __launch_bounds__(kPlanThreads) void msda_plan(const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
                                                          const float *__restrict__ loc, const float *__restrict__ attn,
                                                          PlanGeom G, PlanPtrs W)
{
    plan_unit<kOnePass>(shapes, start, loc, attn, G, W, blockIdx.x);
}

// Forward gather + plan in ONE launch (zira_msda_fwd_plan_f32): the first `units8` blocks plan a (head, level) unit each
// (they are the long pole and start first), the others run the forward of 16 (b, q, m) items each, one per wave, exactly
// as msda_fwd_lean does (csrc/msda_fwd_lean.h).  Two kernels on two streams do not overlap here -- neither eagerly (the
// forward's waves fill every register file before a 1024-thread plan block fits) nor as branches of a replayed hipGraph
// (measured: 36.5 us = the sum) -- while one grid does: the plan runs on 64 CUs beside the gather.  Keeping the kernel within
// the forward's 64 registers (two-pass plan, 84 bytes of scratch per lane) was measured at 24.4 us for the call; with the
// plan's 100 registers and one forward block per CU 19.0 us.
#ifndef ZIRA_FUSED_WAVES
#define ZIRA_FUSED_WAVES 8   // (64 registers: two 1024-thread blocks per CU, i.e. the gather at the plain forward's occupancy)
#endif
#ifndef ZIRA_FUSED_ONEPASS
#define ZIRA_FUSED_ONEPASS 1
#endif
__global__ __launch_bounds__(kPlanThreads, ZIRA_FUSED_WAVES) void msda_fwd_plan(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ start,
    const float *__restrict__ loc, const float *__restrict__ attn, unsigned S, FastDiv Mdiv, unsigned LP, FastDiv Qdiv,
    float invP, unsigned nitems, unsigned per_xcd, float *__restrict__ out, PlanGeom G, PlanPtrs W, unsigned units,
    unsigned units8, unsigned onepass)
{
    if (blockIdx.x < units8) {
        if (blockIdx.x < units) {
            if (onepass) plan_unit<true>(shapes, start, loc, attn, G, W, blockIdx.x);
            else plan_unit<false>(shapes, start, loc, attn, G, W, blockIdx.x);
        }
        return;
    }
    // (the plan's registers -- 100 against the forward's 64 -- leave room for ONE such block per CU; two items per wave with
    // interleaved gathers were measured and lost: 20.6 against 19.0 us for the call)
    const ItemId id = lean_item(nitems, per_xcd, Qdiv, Mdiv, blockIdx.x - units8, kPlanThreads / 64);
    if (!id.ok) return;  // wave-uniform
    fwd_lean_item<2>(value, shapes, start, loc, attn, S, Mdiv.d, LP, invP, id, out);
}

// ------------------------------------------------------------------------------------------
// accumulate
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned uni(unsigned x) { return __builtin_amdgcn_readfirstlane(x); }

// a work item: a tile or a share of a split tile; everything block-uniform (scalar registers)
struct Item {
    unsigned n, ty0, tx0, the, twe, H, W;
    unsigned share, slot; // a share of a split tile: its sums go to partial tile `slot` of the workspace
    unsigned nshare, kshare;   // ... of `nshare` shares, this one number `kshare` (the tile's first partial tile is slot - kshare)
    unsigned hq;          // item index of query 0 of the head: b * Q * M + m
    size_t voff;          // float offset of the level's pixel 0, this head, in grad_value
    size_t roff;          // first record
};
struct Hdr {
    uint4 a, b;
};
template <typename T>
__device__ __forceinline__ T ldg(const void *base, unsigned byte_off)   // scalar base + 32-bit vector offset
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// The step loop's only dependent load is the record's grad_out row (requested two steps ahead, the records four): the
// attention weight rides in the record, the value rows are the gather blocks' business.  (Measured and dropped: a "resident"
// form -- a block serves ONE head and keeps that head's 900 grad_out rows, 115 KB, in LDS beside the tile, so that no load of
// the loop depends on another -- leaves room for one block per CU, and an item's header and first records are then a serial
// latency chain nothing else on the CU hides: 39.4 us at 512 threads, 45.4 at 256, 40.8 at 1024, against 24.9 us for this form.)
// The gather half of the backward (grad_sampling_loc, grad_attn_weight: a wave per (b, q, m), csrc/msda_fwd_lean.h) rides
// in the same launch: blocks [0, nacc) accumulate -- three per CU --, the others take NTHR / 64 items each and pass through
// the fourth slot of every CU while the accumulate blocks wait on their loads (two kernels one after the other: 9.2 + 24.9 us).
struct HomeArgs {
    const float *value, *loc, *attn;
    const int64_t *shapes, *start;
    float *grad_loc, *grad_attn;
    unsigned S, LP, nitems, per_xcd, nacc;
    float invP;
    FastDiv Mdiv, Qdiv;
};
template <unsigned NTHR>
__global__ __launch_bounds__(NTHR, ZIRA_TILE_SLOTS_PER_CU) void msda_bwd_tile_accum(
    const float *__restrict__ grad_out, PlanGeom G, const unsigned *__restrict__ ucnt, const uint4 *__restrict__ citems,
    const uint4 *__restrict__ recs, float *__restrict__ dump, float *partial, unsigned *tcnt, float *__restrict__ grad_value,
    const HomeArgs HA)
{
    BTIME_DECL;
    if (blockIdx.x >= HA.nacc) {
        // (ZIRA_HOME_ITEMS_PER_WAVE = 2: a wave takes two neighbouring queries of a head and interleaves their loads,
        //  csrc/msda_fwd_lean.h bwd_home_item2 -- measured equal to one item per wave)
        constexpr unsigned IPW = ZIRA_HOME_ITEMS_PER_WAVE, WPB = NTHR / 64;
        const unsigned w0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (IPW - 1);   // wave w -> slots IPW w, IPW w + 1
        const ItemId id = lean_item(HA.nitems, HA.per_xcd, HA.Qdiv, HA.Mdiv, blockIdx.x - HA.nacc, WPB * IPW, w0);
        if (!id.ok) return;  // wave-uniform
        if (IPW == 2) {
            const ItemId id1 = lean_item(HA.nitems, HA.per_xcd, HA.Qdiv, HA.Mdiv, blockIdx.x - HA.nacc, WPB * IPW, w0 + 1);
            bwd_home_item2<2>(grad_out, HA.value, HA.shapes, HA.start, HA.loc, HA.attn, HA.S, HA.Mdiv.d, HA.LP, HA.invP, id, id1,
                              HA.grad_loc, HA.grad_attn);
        } else {
            bwd_home_item<2>(grad_out, HA.value, HA.shapes, HA.start, HA.loc, HA.attn, HA.S, HA.Mdiv.d, HA.LP, HA.invP, id,
                             HA.grad_loc, HA.grad_attn);
        }
        BTIME_FLUSH(2);
        return;
    }
    constexpr unsigned D = 32, LPS = 8, NW = NTHR / 64, NG = 8;
    constexpr unsigned SPB = NW * NG;     // records per block step
    extern __shared__ double lds_acc[];   // [(kNPix + 1) * D]: the tile, then the trash row; the deal's table
    double *acc = lds_acc;

    TSTAMP_DECL;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned grp = lane / LPS, j = lane % LPS;
    // this block's list of items: group g = the heads it works for (those whose value slices an XCD keeps in its L2;
    // placement is for speed only), k = its number inside the group
    const unsigned g = blockIdx.x % G.ng, k = blockIdx.x / G.ng;
    // The units of the group (nug of them) publish their class counts; laid end to end -- class-major, so heavy items
    // first -- they form one ring, and block k takes positions k, k + nbg, ...
    unsigned *tab = reinterpret_cast<unsigned *>(acc + (kNPix + 1) * D);   // [kClasses nug] prefix, then [nug] first item slot of the unit, [1] items, [1] flag
    const unsigned h0 = g * G.hp;
    if (h0 >= G.heads) return;
    const unsigned nug = (G.heads - h0 < G.hp ? G.heads - h0 : G.hp) * G.L, ne = kClasses * nug;
    for (unsigned e = tid; e < ne; e += NTHR) {
        const unsigned c = e / nug, u = e - c * nug;
        tab[e] = ucnt[(h0 * G.L + u) * kUcnt + c];
    }
    for (unsigned u = tid; u < nug; u += NTHR) tab[ne + u] = ucnt[(h0 * G.L + u) * kUcnt + kClasses];
    __syncthreads();
    if (wave == 0) {   // exclusive prefix over the ne counts, 64 at a time
        unsigned run = 0;
        for (unsigned e0 = 0; e0 < ne; e0 += 64) {
            const unsigned v = e0 + lane < ne ? tab[e0 + lane] : 0u;
            unsigned incl = v;
#pragma unroll
            for (unsigned d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (e0 + lane < ne) tab[e0 + lane] = run + incl - v;
            run += __shfl(incl, 63);
        }
        if (lane == 0) tab[ne + nug] = run;
    }
    __syncthreads();
    // The ring is dealt in rounds of nbg positions, every other round backwards (block k: position k of the even rounds,
    // nbg - 1 - k of the odd ones): the block that drew the heaviest item of a round draws the lightest of the next.
    const unsigned total = uni(tab[ne + nug]);
    unsigned *flag = tab + ne + nug + 1;   // what the share counter returned, for the whole block
    const unsigned full = fdiv(total, G.NBGdiv), rem = total - full * G.nbg;
    const unsigned cnt = full + (((full & 1u) ? G.nbg - 1u - k : k) < rem ? 1u : 0u);
    if (cnt == 0) return;
    const unsigned pre_l = lane < ne ? tab[lane] : 0xFFFFFFFFu;   // (the common case: ne <= 64, the whole prefix in one wave's lanes)
    __syncthreads();   // (tab is read; below it is only touched again through load_hdr's reads)
    for (unsigned x = tid; x < (kNPix + 1) * D / 2; x += NTHR) reinterpret_cast<uint4 *>(acc)[x] = make_uint4(0u, 0u, 0u, 0u);

    const unsigned rs = G.M * D;                  // floats between pixels
    const unsigned gsel = wave * NG + grp;        // this group's record inside a block step
    // Channel 4 j + kk of an accumulator row lives in word kk * 8 + j: a group's 8 lanes add to 8 consecutive words, the
    // fast pattern for ds_add_f64 (measured against it, whole kernel: word (kk >> 1) * 16 + 2 j + (kk & 1), a lane's
    // channel pairs adjacent, 37.4 against 33.7 us; word 4 j + kk, a lane's four channels contiguous, 59 us).  The
    // write-out gathers a lane's four channels with four 8-byte reads 64 bytes apart.
    const unsigned odd = grp & 1u;
    constexpr unsigned kAccWord[4] = {0u, 8u, 16u, 24u}, kSwap = 1;   // odd groups: kk 0 <-> 1, 2 <-> 3 (the other 64 bytes)
    const unsigned lb = j * 8u;
    unsigned lbk[4];   // byte offset inside a row of the word this lane's add number kk goes to
#pragma unroll
    for (unsigned kk = 0; kk < 4; ++kk) lbk[kk] = lb + (odd ? kAccWord[kk ^ kSwap] : kAccWord[kk]) * 8u;

    // Every block's list runs from heavy items (many block steps, LDS-bound) to light ones (a step or none: 16 KB of stores
    // each, bound by the chip's write bandwidth -- scripts/tile_timeline.py: ~3 us per item whatever its size, 0.5 us per
    // step); with all blocks walking it the same way the stores of a launch pile up in its second half.  Every other block
    // of a group walks its list backwards.
    const bool backwards = ZIRA_TILE_MIX_ORDER && (k & 1u);
    auto load_hdr = [&](unsigned ii) {   // the ii-th item this block takes: ring position k + i nbg -> (class, unit, position in the unit's class items)
        const unsigned i = backwards ? cnt - 1u - ii : ii;
        const unsigned r = i * G.nbg + ((i & 1u) ? G.nbg - 1u - k : k);
        unsigned e;
        if (ne <= 64) {
            e = (unsigned)__popcll(__ballot(pre_l <= r)) - 1u;
        } else {
            e = 0;
            for (unsigned e0 = 0; e0 < ne; e0 += 64)
                e += (unsigned)__popcll(__ballot(e0 + lane < ne && tab[e0 + lane] <= r));
            e -= 1u;
        }
        e = uni(e);
        const unsigned c = e / nug, u = e - c * nug;
        const unsigned idx = uni(tab[ne + u]) + (r - uni(tab[e]));
        const uint4 *p = citems + ((size_t)c * G.ccap + idx) * 2;
        Hdr h;
        h.a = p[0];
        h.b = p[1];
        return h;
    };
    auto make_item = [&](const Hdr &hd) {
        Item it;
        const unsigned off = uni(hd.a.x), org = uni(hd.a.z), hw = uni(hd.a.w), vrow = uni(hd.b.x);
        it.n = uni(hd.a.y);
        it.hq = uni(hd.b.y);
        it.share = uni(hd.b.z) & kItemShare;
        it.nshare = ((uni(hd.b.z) >> 17) & 127u) + 1u;
        it.kshare = uni(hd.b.z) >> 24;
        it.slot = uni(hd.b.w);
        it.ty0 = org & 0xFFFFu;
        it.tx0 = org >> 16;
        it.H = hw & 0xFFFFu;
        it.W = hw >> 16;
        it.the = it.H - it.ty0 < kTH ? it.H - it.ty0 : kTH;   // rows / columns of the tile inside the map
        it.twe = it.W - it.tx0 < kTW ? it.W - it.tx0 : kTW;
        it.roff = (size_t)off;
        it.voff = (size_t)vrow * D;
        return it;
    };
    // Every load of the item pipeline is issued unconditionally (clamped addresses): loads and stores share one in-order
    // counter on gfx950, and the compiler can only let a wave wait for exactly the load it needs when it knows how many
    // memory operations were issued after it.
    typedef uint4 Raw;
    auto fetch = [&](const Item &it, unsigned s) {   // the record of this group at block step s
        const unsigned e = s * SPB + gsel;
        const unsigned o = (e < it.n ? e : 0u) * 16u;   // (record 0 is readable for every item: the record region ends with a pad)
        return ldg<uint4>(recs + it.roff, o);
    };
    auto row_of = [&](const Item &it, const Raw &r) {   // this lane's four channels of the record's grad_out row
        return ldg<float4>(grad_out + (size_t)it.hq * D, (r.x & 0x3FFFFFu) * (G.M * 128u) + j * 16u);
    };
    auto compute = [&](const Item &it, const Raw &r, const float4 g4, unsigned s) {
        const bool ok = s * SPB + gsel < it.n;
        const float lw = __uint_as_float(r.z), lh = __uint_as_float(r.w);
        const float a = ok ? __uint_as_float(r.y) : 0.f;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float w00 = __fmul_rn(hh, hw), w01 = __fmul_rn(hh, lw), w10 = __fmul_rn(lh, hw), w11 = __fmul_rn(lh, lw);
        // corner rows: term = w * (a * g), both factors rounded to fp32 as the reference forms them (cuh:117-147), the
        // product and the sum in double.  The two groups of a 16-lane LDS row issue their four adds in different orders
        // (kk ^ swap), so that one instruction finds them in different halves of the banks.
        const float gs[4] = {odd ? g4.y : g4.x, odd ? g4.x : g4.y, odd ? g4.w : g4.z, odd ? g4.z : g4.w};
        const double tt[4] = {(double)__fmul_rn(gs[0], a), (double)__fmul_rn(gs[1], a), (double)__fmul_rn(gs[2], a),
                              (double)__fmul_rn(gs[3], a)};
        const double wc[4] = {(double)w00, (double)w01, (double)w10, (double)w11};
        // corner rows of the tile's accumulators (unsigned: position -1 wraps and fails the tests; outside -> the trash row)
        const unsigned pr = ((r.x >> 22) & 31u) - 1u, pc = (r.x >> 27) - 1u;
        unsigned oc[4];
#pragma unroll
        for (unsigned cc = 0; cc < 4; ++cc) {
            const unsigned y = pr + (cc >> 1), x = pc + (cc & 1u);
            oc[cc] = (ok && y < it.the && x < it.twe) ? (y * kTW + x) * kRowBytes : kTrash;
        }
#pragma unroll
        for (unsigned cc = 0; cc < 4; ++cc) {
            char *ap = reinterpret_cast<char *>(acc) + oc[cc];
#pragma unroll
            for (unsigned kk = 0; kk < 4; ++kk) {   // (tt[kk] belongs to channel 4 j + (kk ^ swap) in the odd groups)
#ifdef ZIRA_DEV_HALF_ADDS   // developer ablation (results wrong): the LDS adds of a packed two-channel form, at best
                if (kk & 1u) continue;
#endif
                atomicAdd(reinterpret_cast<double *>(ap + lbk[kk]), wc[cc] * tt[kk]);
            }
        }
    };

    // Items are pipelined: the header of item i + 2 is requested when item i starts, the first records of item i + 1 before
    // item i's steps.  Inside an item the records of steps s + 4, s + 5 and the grad_out rows of steps s + 2, s + 3 are on
    // their way while steps s, s + 1 are summed.
    Hdr h1 = load_hdr(cnt > 1 ? 1u : 0u);
    Item it = make_item(load_hdr(0));
    Raw ra = fetch(it, 0), rb = fetch(it, 1), rc = fetch(it, 2), rd = fetch(it, 3);
    Item nx = make_item(h1);
    __syncthreads();   // (the accumulators are clear)
    BTIME_MARK;
    TSTAMP(0);
    for (unsigned i = 0;; ++i) {
        const bool more = i + 1 < cnt;
        const Hdr h2 = load_hdr(i + 2 < cnt ? i + 2 : cnt - 1);
        const Raw na = fetch(nx, 0), nb = fetch(nx, 1), nc = fetch(nx, 2), nd = fetch(nx, 3);
        // (wave w holds records 8 w .. 8 w + 7 of every SPB: a wave without records at a step skips it)
        const unsigned nsteps = it.n > wave * NG ? (it.n - wave * NG + SPB - 1) / SPB : 0u;
        BTIME_NOTE(1u | ((it.n ? 0ull : 1ull) << 32), (it.n + SPB - 1) / SPB | ((unsigned long long)(it.share ? 1u : 0u) << 32));
        float4 g0 = row_of(it, ra), g1 = row_of(it, rb);
        for (unsigned s = 0; s < nsteps; s += 2) {   // (two copies of the body per round)
            const float4 g2 = row_of(it, rc);
            const Raw re = fetch(it, s + 4);
            compute(it, ra, g0, s);
            const float4 g3 = row_of(it, rd);
            const Raw rf = fetch(it, s + 5);
            if (s + 1 < nsteps) compute(it, rb, g1, s + 1);
            ra = rc; rb = rd; rc = re; rd = rf;
            g0 = g2; g1 = g3;
        }
        TSTAMP(1);
        if (it.n) __syncthreads();   // every add of the item has landed
        ra = na; rb = nb; rc = nc; rd = nd;
        TSTAMP(2);

        // ---- write-out (each thread clears the accumulator words it reads) -------------------------------
        // Thread t gets channels 4 (t & 7) .. + 3 of pixel PPR i + (t >> 3) in round i and stores them with one 16-byte
        // store -- a wave writes eight whole 128-byte rows.
        {
            constexpr unsigned PPR = NTHR / 8, NR = kNPix / PPR;          // pixels per round, rounds
            static_assert(PPR % kTW == 0 && kNPix % PPR == 0, "a round covers whole pixel rows");
            const unsigned c4 = tid & 7u, p0 = tid >> 3;                  // channel quad, pixel of round 0 (then + PPR per round)
            const unsigned pc = p0 % kTW, pr0 = p0 / kTW;
            const bool colin = pc < it.twe;
            float *gbase = grad_value + it.voff + ((size_t)(it.ty0 + pr0) * it.W + (it.tx0 + pc)) * rs + c4 * 4;
            const size_t rstep = (size_t)(PPR / kTW) * it.W * rs;
            char *lp = reinterpret_cast<char *>(acc) + (tid >> 3) * kRowBytes + (tid & 7u) * 8u;
            constexpr unsigned kRound = PPR * kRowBytes;
            auto take4 = [&](unsigned ii) {   // read this thread's four channels of round ii and clear them
                char *q = lp + ii * kRound;
                float4 o;
                double *w = reinterpret_cast<double *>(q);
                o = make_float4((float)w[0], (float)w[8], (float)w[16], (float)w[24]);
                w[0] = 0.0; w[8] = 0.0; w[16] = 0.0; w[24] = 0.0;
                return o;
            };
            if (!it.share) {   // every pixel of the tile once, plain stores (grad_value is never zero-filled)
#pragma unroll
                for (unsigned ii = 0; ii < NR; ++ii) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (it.n) o = take4(ii);
                    float *dst = (colin && (PPR / kTW) * ii + pr0 < it.the) ? gbase + ii * rstep : dump + 192 + 4 * lane;   // (pixels of an edge tile outside the map)
                    *reinterpret_cast<float4 *>(dst) = o;
                }
            } else {
                // A share of a split tile: the whole tile, as it is, to its partial tile -- WRITE-THROUGH (agent-scope 8-byte
                // stores: they leave this XCD's L2 for memory), every wave drains its stores, the block's one lane counts the
                // share in (agent-scope atomic), and the share that finds all the others counted adds the partial tiles up
                // in their fixed order 0 .. K - 1 (read with agent-scope loads: past this CU's L1) and writes the tile's
                // pixels: the same sum in every run, whoever arrives last.  cdna_hip_programming.md section 5 ("in-launch
                // split-K reduction", write-through form) / Guideline 16 R1.  No other block waits: nothing can deadlock.
                // NOTE: this is that guide's write-through recipe, not a C++ release / acquire pair: the ordering rests on gfx950
                // behaviour the guide measured -- agent-scope (sc1) stores have left the XCD's L2 when the storing wave's
                // s_waitcnt vmcnt(0) returns, the counter add comes behind every storing wave's wait and a workgroup barrier,
                // and EVERY load of the handed-off bytes is an agent-scope (sc1) load, past this CU's L1.  Pinned to the target:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "msda_bwd_tile_accum's in-launch hand-off is validated for gfx950 only (write-through stores + sc1 loads)"
#endif
                typedef unsigned long long u64;
                const size_t toff = (size_t)p0 * D + c4 * 4;          // this thread's float4 of a tile, round 0
                u64 *pb = reinterpret_cast<u64 *>(partial + (size_t)it.slot * kNPix * D + toff);
#pragma unroll
                for (unsigned ii = 0; ii < NR; ++ii) {
                    const float4 o = take4(ii);
                    u64 *q = pb + (size_t)ii * PPR * D / 2;
                    __hip_atomic_store(q, ((u64)__float_as_uint(o.y) << 32) | __float_as_uint(o.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(q + 1, ((u64)__float_as_uint(o.w) << 32) | __float_as_uint(o.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its partial tile has left
                __syncthreads();
                unsigned *tc = tcnt + (it.slot - it.kshare);
                if (tid == 0) *flag = __hip_atomic_fetch_add(tc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (uni(*flag) == it.nshare - 1u) {                   // every share of the tile is in memory
                    const u64 *p0k = reinterpret_cast<const u64 *>(partial + (size_t)(it.slot - it.kshare) * kNPix * D + toff);
#pragma unroll
                    for (unsigned ii = 0; ii < NR; ++ii) {
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (unsigned kk = 0; kk < it.nshare; ++kk) {   // (fixed order: the same sum in every run)
                            const u64 *q = p0k + (size_t)kk * kNPix * D / 2 + (size_t)ii * PPR * D / 2;
                            const u64 lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const u64 hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const float4 b = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                                                         __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
                            if (kk == 0) a = b;
                            else { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
                        }
                        float *dst = (colin && (PPR / kTW) * ii + pr0 < it.the) ? gbase + ii * rstep : dump + 192 + 4 * lane;
                        *reinterpret_cast<float4 *>(dst) = a;
                    }
                    if (tid == 0) __hip_atomic_store(tc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the plan may serve another backward)
                }
            }
        }
        if (it.n) __syncthreads();   // the tile is clear again before the next item adds to it
        TSTAMP(3);
        TSTAMP_COUNT(8);
        if (!more) break;
        it = nx;
        nx = make_item(h2);
    }
    TSTAMP_FLUSH;
    BTIME_FLUSH(1);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct TilesLayout {
    PlanGeom G;
    size_t ctl_bytes, off_tcnt, off_citems, off_dump, off_recs, off_partial, total;
    bool one_pass;
    size_t lds_plan, lds_acc;
    unsigned units, grid;
};

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

inline bool make_tiles_layout(int B, int S, int M, int D, int L, int Q, int P, TilesLayout &T)
{
    if (D != 32 || L > (int)kTMaxLevels) return false;
    const unsigned long long heads = (unsigned long long)B * M;
    if ((unsigned long long)S * M * D >= (1ull << 31) || (unsigned long long)Q * M * L * P * 2 >= (1ull << 31)) return false;
    if (heads * L >= (1ull << 20)) return false;
    if ((unsigned long long)S * M * D * 4 >= (1ull << 32) || (unsigned long long)Q * M * D * 4 >= (1ull << 32)) return false;   // 32-bit byte offsets
    if ((unsigned long long)B * S * M >= (1ull << 32) || (unsigned long long)B * Q * M >= (1ull << 32)) return false;          // 32-bit row indices
    PlanGeom &G = T.G;
    G.B = (unsigned)B; G.S = (unsigned)S; G.M = (unsigned)M; G.L = (unsigned)L; G.Q = (unsigned)Q; G.P = (unsigned)P;
    G.LP = (unsigned)(L * P);
    G.heads = (unsigned)heads;
    T.units = G.heads * G.L;
    // tiles of a level with n pixels: ceil(H / TH) * ceil(W / TW) <= n / min(TH, TW) + 1 for every H * W = n
    const unsigned tmin = kTH < kTW ? kTH : kTW;
    G.ntmax = (unsigned)S / tmin + (unsigned)L;
    if (G.ntmax > 4 * kPlanThreads) return false;      // (the plan kernel keeps a level's tile counts in registers across its scan)
    G.cap = ZIRA_TILE_CAP;
    if ((unsigned long long)Q * P * 4 >= (1ull << 24)) return false;
    if ((unsigned)Q >= (1u << 22)) return false;       // (22 bits of a record)
    G.rcap = (unsigned)Q * P * 4;                       // a sample has a record in every tile it touches (<= 4)
    G.ecap = (unsigned)(((unsigned long long)G.rcap * 2) / G.cap) + 1;   // sum of K over split tiles <= 2 records / cap
    G.iph = G.ntmax + G.L * G.ecap;
    if ((unsigned long long)T.units * G.rcap >= (1ull << 31)) return false;   // 32-bit record indices
    // accumulate blocks: a group = the heads whose value slices one XCD keeps in its L2
    G.ng = heads >= 8 ? 8u : 1u;
    G.hp = G.ng > 1 ? (unsigned)((heads + 7) / 8) : G.heads;
    T.grid = tiles_cu_count() * ZIRA_TILE_BLOCKS_PER_CU;
    T.grid &= ~7u;
    if (T.grid < 8 || T.grid < G.ng) return false;
    G.nbg = T.grid / G.ng;
    if (heads * G.iph >= (1ull << 28)) return false;
    G.ccap = (unsigned)(heads * G.iph);
    if ((unsigned long long)G.hp * L * (kClasses + 1) + 1 > 1024) return false;   // (the deal's prefix table lives in LDS beside the tile)
    G.Mdiv = make_fdiv((unsigned)M);
    G.NBGdiv = make_fdiv(G.nbg);
    T.lds_plan = (kTLevelWords * kTMaxLevels + 16 + kPlanThreads / 64 + 1 + G.ntmax + 16 * kPlanThreads) * 4;
    if (T.lds_plan > 78 * 1024) return false;   // (two blocks per CU)
    T.lds_acc = (size_t)(kNPix + 1) * 32 * 8 + (((size_t)G.hp * L * (kClasses + 1) + 2 + 3) & ~(size_t)3) * 4;
    T.one_pass = (unsigned)Q <= kPlanThreads && P <= 4;
    size_t o = 0;
    T.ctl_bytes = align256((size_t)T.units * kUcnt * 4);   // ucnt[units][kUcnt]
    o += T.ctl_bytes;
    T.off_tcnt = o; o += align256((size_t)T.units * G.ecap * 4);
    T.off_citems = o; o += align256((size_t)kClasses * G.ccap * 32);
    T.off_dump = o;   o += 2048;                                            // where redirected stores go (never read)
    T.off_recs = o;   o += align256((size_t)T.units * G.rcap * 16) + 256;   // (+ pad: record 0 of an empty tail item)
    if ((unsigned long long)T.units * G.ecap >= (1ull << 24)) return false;   // 24-bit partial-tile index
    T.off_partial = o; o += align256((size_t)T.units * G.ecap * kNPix * 32 * 4);   // (worst case; what is touched is one 16 KB tile per share)
    T.total = o;
    return true;
}

// Dynamic LDS above 48 KB is an opt-in on runtimes that enforce the default limit (the plan kernels take up to 78 KB, 77 KB
// at the decoder shape): declared once per kernel; a refusal turns the planned path off (-1: the callers fall back).
template <typename K>
inline bool allow_lds(K kernel, size_t bytes, bool &done)
{
    if (done || bytes <= 48 * 1024) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done = true;
    return true;
}

inline PlanPtrs plan_ptrs(const TilesLayout &T, void *plan)
{
    char *w = reinterpret_cast<char *>(plan);
    PlanPtrs W;
    W.ucnt = reinterpret_cast<unsigned *>(w);
    W.tcnt = reinterpret_cast<unsigned *>(w + T.off_tcnt);
    W.partial = reinterpret_cast<float *>(w + T.off_partial);
    W.citems = reinterpret_cast<uint4 *>(w + T.off_citems);
    W.recs = reinterpret_cast<uint4 *>(w + T.off_recs);
    return W;
}

}  // namespace

size_t tiles_plan_bytes(int B, int S, int M, int D, int L, int Q, int P)
{
    TilesLayout T;
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0) return 0;
    return make_tiles_layout(B, S, M, D, L, Q, P, T) ? T.total : 0;
}

int tiles_plan_f32(const int64_t *shapes, const int64_t *start, const float *loc, const float *attn, int B, int S, int M, int D,
                   int L, int Q, int P, void *plan, size_t plan_bytes, hipStream_t st)
{
    TilesLayout T;
    if (!make_tiles_layout(B, S, M, D, L, Q, P, T) || !plan || plan_bytes < T.total || ((uintptr_t)plan & 15)) return -1;
    const PlanPtrs W = plan_ptrs(T, plan);
    static bool lds_one = false, lds_two = false;   // (one device per process)
    if (!(T.one_pass ? allow_lds(msda_plan<true>, 78 * 1024, lds_one) : allow_lds(msda_plan<false>, 78 * 1024, lds_two))) return -1;
    if (T.one_pass)
        hipLaunchKernelGGL(msda_plan<true>, dim3(T.units), dim3(kPlanThreads), T.lds_plan, st, shapes, start, loc, attn, T.G, W);
    else
        hipLaunchKernelGGL(msda_plan<false>, dim3(T.units), dim3(kPlanThreads), T.lds_plan, st, shapes, start, loc, attn, T.G, W);
    return (int)hipGetLastError();
}

int tiles_fwd_plan_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc, const float *attn,
                       int B, int S, int M, int D, int L, int Q, int P, float *out, void *plan, size_t plan_bytes, hipStream_t st)
{
    TilesLayout T;
    if (!make_tiles_layout(B, S, M, D, L, Q, P, T) || !plan || plan_bytes < T.total || ((uintptr_t)plan & 15)) return -1;
    if ((unsigned long long)B * Q * M >= (1ull << 31) || L * P > 64) return -1;
    const PlanPtrs W = plan_ptrs(T, plan);
    static bool lds_ok = false;
    if (!allow_lds(msda_fwd_plan, 78 * 1024, lds_ok)) return -1;
    const unsigned nitems = (unsigned)B * Q * M, per = (nitems + 7) >> 3, wpb = kPlanThreads / 64;
    const unsigned units8 = (T.units + 7) & ~7u;
    const unsigned grid = units8 + 8 * ((per + wpb - 1) / wpb);
    hipLaunchKernelGGL(msda_fwd_plan, dim3(grid), dim3(kPlanThreads), T.lds_plan, st, value, shapes, start, loc, attn, (unsigned)S,
                       make_fast_div((unsigned)M), (unsigned)(L * P), make_fast_div((unsigned)Q), 1.0f / (float)P, nitems, per,
                       out, T.G, W, T.units, units8, (T.one_pass && ZIRA_FUSED_ONEPASS) ? 1u : 0u);
    return (int)hipGetLastError();
}

int tiles_backward_planned_f32(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start,
                               const float *loc, const float *attn, int B, int S, int M, int D, int L, int Q, int P, float *gv,
                               float *gl, float *ga, void *plan, size_t plan_bytes, hipStream_t st)
{
    TilesLayout T;
    if (!make_tiles_layout(B, S, M, D, L, Q, P, T) || !plan || plan_bytes < T.total || ((uintptr_t)plan & 15)) return -1;
    const PlanPtrs W = plan_ptrs(T, plan);
    float *dump = reinterpret_cast<float *>(reinterpret_cast<char *>(plan) + T.off_dump);
    if ((unsigned long long)B * Q * M >= (1ull << 31) || L * P > 64) return -1;
    HomeArgs HA;
    HA.value = value; HA.loc = loc; HA.attn = attn; HA.shapes = shapes; HA.start = start; HA.grad_loc = gl; HA.grad_attn = ga;
    HA.S = (unsigned)S; HA.LP = (unsigned)(L * P); HA.nitems = (unsigned)B * Q * M; HA.per_xcd = (HA.nitems + 7) >> 3;
    HA.nacc = T.grid; HA.invP = 1.0f / (float)P; HA.Mdiv = make_fast_div((unsigned)M); HA.Qdiv = make_fast_div((unsigned)Q);
    if (T.grid & 7u) return -1;   // (the gather blocks' XCD interleave starts at a multiple of 8)
    const unsigned wpb = kAccThreads / 64 * ZIRA_HOME_ITEMS_PER_WAVE;   // items a gather block takes
    hipLaunchKernelGGL((msda_bwd_tile_accum<kAccThreads>), dim3(T.grid + 8 * ((HA.per_xcd + wpb - 1) / wpb)), dim3(kAccThreads),
                       T.lds_acc, st, grad_out, T.G, W.ucnt, W.citems, W.recs, dump, W.partial, W.tcnt, gv, HA);
    return (int)hipGetLastError();
}

}  // namespace zira

ZIRA_DEV_STAMP_READERS
ZIRA_DEV_BTIME_READER
