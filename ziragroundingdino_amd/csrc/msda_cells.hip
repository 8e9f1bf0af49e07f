// msda_cells.hip -- backward of multi-scale deformable attention for gfx950 (MI355X), dense calls:
// "cell walk" and LDS accumulate.
//
// Arithmetic to match: reference csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:87-159 (bilinear
// col2im: grad_value += w_corner * attn * grad_out, grad_attn = <grad_out, sample>, grad_loc from
// the corner differences) inside :301-403.  The decomposition is not the reference's.
//
// Why this shape.  A sample touches the 2 x 2 pixels around its location.  The reference scatters
// 4 corner rows per sample with global fp32 atomics (memory-side on MI355X: ~1.3 TB/s); round 1
// sorted 8-byte per-corner entries by tile and re-gathered a grad_out row per entry -- two passes
// bounded by the rate at which a CU's L1 takes 128-byte rows from L2, plus the entry round trip.
// Here the unit of work is the DESTINATION, and every sample is visited once:
//
//   * a sample with top-left pixel (y0, x0) lives in cell (u, v) = (y0 + 1, x0 + 1) of its level,
//     u in [0, H], v in [0, W]; its corners are the pixels (u-1 | u, v-1 | v).
//   * a "walker" (a group of LPG lanes, D / LPG channels per lane) owns cell row u of a segment of
//     tw pixels and visits the row's samples in the order of v ("step" c = v - x0).  During step
//     c it holds four accumulators in registers: pixels (u, c-1), (u, c) [corners dy = 1, its own
//     pixel row] and (u-1, c-1), (u-1, c) [dy = 0, the row above].  The NG = 64 / LPG walkers of a
//     wave take NG consecutive cell rows and run in lockstep per step, so when a step ends the
//     wave hands every walker's finished "row above" pixel to the walker above with one lane
//     shuffle, and each grad_value pixel is stored exactly once with plain 16-byte stores: no
//     atomics, no zero-fill, no LDS accumulators.  A tile is therefore NG - 1 pixel rows: its last
//     walker runs the cell row below the tile, only for what that row gives the tile's last pixel row.
//   * grad_sampling_loc / grad_attn_weight of a sample need <grad_out row, value rows of its 4
//     corners>: the walker keeps the value rows (u-1 | u) x (c-1 | c) in registers (two new rows per
//     step, requested one step ahead), so value is read about twice in total instead of four
//     gathered rows per sample.
//   * per sample the walk needs its 16-byte record and one grad_out row gather (8 rows in round 1).
//
// Since round 2 the walk (K2 below) serves D = 16 / 64; dense D = 32 calls (the model's) sum the tiles in LDS
// instead -- msda_bwd_accum, further down: the same bin kernel and records, tiles of 15 x 8 pixels, the
// corner rows added into 64-bit fixed-point accumulators with ds_add_u64 (exact sums, no ordering needed),
// the walk kept behind it as the float path for non-finite gradients.
//
// Kernels (caller-provided workspace):
//   K1 msda_bwd_bin    one thread per sample: pixel coordinates, bilinear fractions; valid samples
//                      become 16-byte records {q | p | cell-in-tile, lw, lh, attn}; a block (QB queries
//                      of one head) counting-sorts its records by tile and writes one contiguous run per
//                      tile plus a row of {offset, count} descriptors.  A cell on the top row / left
//                      column of its tile is also needed by the neighbouring tile, so its record is copied
//                      there (<= 4 copies; (1 + 1/15)(1 + 1/tw) on average).  Samples outside (-1, H) x
//                      (-1, W) get their zero gradients here.  No value / grad_out traffic.
//   K2 msda_bwd_walk   one wave per work item (waves stride over the items of their XCD's heads): reads
//                      the tile's runs, counting-sorts the records by (walker, step) in LDS -- a padded
//                      stream vis[k][walker] of 4-byte words in which all walkers are in the same step at
//                      the same k -- then walks it with the records 6 and the grad_out rows 3 elements ahead.
//   K3 msda_bwd_fold   dense calls only: sums the partial rows of split tiles (below).
//
// Levels differ in samples per pixel by orders of magnitude (every level receives Q*P samples per
// head; at the encoder shape that is 5 per pixel on the 100 x 167 level and 326 on the 13 x 21
// one), and a walker is a serial chain.  The geometry is therefore per level, chosen ON THE
// DEVICE from the int64 level table (the host never reads it: the C ABI only has device
// pointers; it only knows bounds that follow from S): the segment width shrinks until a tile
// holds about `vstar` records, and where that is not enough (dense calls) a tile becomes K work
// items, each taking 1/K of the tile's records and writing partial rows that K3 adds up.
// Precondition, as in the reference module (ms_deform_attn.py:284): the levels tile [0, S).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "msda_internal.h"

#ifndef ZIRA_WALK_DF
#define ZIRA_WALK_DF 4       // walk: the record of stream element k + DF is requested while element k is processed
#endif
#ifndef ZIRA_WALK_DR
#define ZIRA_WALK_DR 2       // ... and the grad_out row of element k + DR
#endif
#ifndef ZIRA_WALK_MINWAVES
#define ZIRA_WALK_MINWAVES 2 // walk: waves per SIMD the register allocation must allow
#endif
#ifndef ZIRA_WALK_LPG32
#define ZIRA_WALK_LPG32 4    // lanes per walker for D = 32 (4: 16 walkers x 8 channels per lane; 8: 8 x 4)
#endif
#ifndef ZIRA_WALK_VSTAR_DENSE
#define ZIRA_WALK_VSTAR_DENSE 1536   // records a work item should hold
#endif
#ifndef ZIRA_WALK_VSTAR_SPARSE
#define ZIRA_WALK_VSTAR_SPARSE 384
#endif
#ifndef ZIRA_WALK_GRID_DENSE
#define ZIRA_WALK_GRID_DENSE 2048
#endif
#ifndef ZIRA_WALK_GRID_SPARSE
#define ZIRA_WALK_GRID_SPARSE 4096
#endif

#ifndef ZIRA_DENSE_ACCUM
#define ZIRA_DENSE_ACCUM 1   // dense calls with D = 32: msda_bwd_accum (LDS fixed-point accumulators) instead of the walk
#endif
#ifndef ZIRA_ACC_TWL
#define ZIRA_ACC_TWL 3       // accumulate: log2 of the tile width (tiles of 15 x 8 pixels: two blocks per CU)
#endif
#ifndef ZIRA_ACC_VSTAR
#define ZIRA_ACC_VSTAR 8192  // accumulate: records a work item should hold
#endif
#ifndef ZIRA_ACC_THREADS
#define ZIRA_ACC_THREADS 512
#endif
#ifndef ZIRA_ACC_MINW
#define ZIRA_ACC_MINW 4    // accumulate: waves per SIMD the register allocation must allow
#endif
#ifndef ZIRA_ACC_SKIP_TRASH
#define ZIRA_ACC_SKIP_TRASH 0   // accumulate: 1 = the lanes of a corner that belongs to a neighbouring tile sit the adds out (exec mask):
                                // measured in round 5, SLOWER (386 -> 393 us: a masked ds_add_u64 costs what a full one does)
#endif
#ifndef ZIRA_ACC_DR
#define ZIRA_ACC_DR 3      // accumulate: grad_out rows requested this many records ahead (DR + 1 divides 8)
#endif

namespace {

constexpr unsigned kBinThreads = 256;
constexpr unsigned kMaxLevels = 16;
constexpr unsigned kInvalidVisit = 0xFFFFFFFFu;
constexpr unsigned kMaxSplit = 64;

struct FastDiv {
    unsigned mul, shift, d;
};
__host__ __device__ __forceinline__ unsigned fast_div(unsigned n, FastDiv f)
{
    return (unsigned)(((unsigned long long)n * f.mul) >> f.shift);
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

struct CellGeom {
    unsigned S, M, L, P, LP, Q, heads, D;
    unsigned ng, thp;          // walkers per wave (64 / LPG) and pixel rows per tile (ng - 1)
    unsigned twl_max, twl_min; // log2 of the segment width: upper / lower limit of the per-level choice
    unsigned vstar;            // records a work item should hold
    unsigned split;            // 1: tiles above vstar are split into K work items (dense calls; needs K3)
    unsigned prows_max;        // partial rows per head the workspace has room for
    unsigned QB, nblk, slice;  // queries per bin block, bin blocks per head, record slots per block
    unsigned ntmax;            // capacity of the tile histogram (>= tiles per head for any level shapes)
    unsigned cap;              // words of a walk wave's stream (LDS)
    FastDiv LPdiv, Pdiv, Mdiv, nblkdiv, thpdiv;
};

// record word 0: q (19 bits) | p (4) | ul (4: walker inside the tile, 0 .. thp) | vl (5: step, 0 .. tw)
constexpr unsigned kQBits = 19, kPBits = 4, kUlBits = 4, kCellShift = kQBits + kPBits;
constexpr unsigned kRefBits = 24;

struct Level {
    int H, W;
    unsigned nty;     // tile rows (thp pixel rows each)
    unsigned ntx;     // segments per pixel row
    unsigned tbase;   // first tile of the level (tiles: what K1 sorts by)
    unsigned twl;     // log2 segment width
    unsigned K;       // work items per tile
    unsigned wbase;   // first work item of the level
    unsigned pbase;   // first partial row of the level (K > 1)
};
constexpr unsigned kLevelWords = sizeof(Level) / 4;

// Per-level geometry from the device-side int64 table into LDS; identical in every kernel of a
// call (same code, same inputs).  tot[0] = tiles per head, tot[1] = work items per head, tot[2] =
// partial rows per head.  Called by every thread of the block; ends with a barrier.
__device__ __forceinline__ void load_levels(const int64_t *__restrict__ shapes, const CellGeom &G,
                                            Level *lv, unsigned *tot)
{
    if (threadIdx.x < G.L) {  // thread l: segment width and split factor of level l
        const unsigned l = threadIdx.x;
        Level v;
        v.H = (int)shapes[2 * l];
        v.W = (int)shapes[2 * l + 1];
        const float hw = (float)v.H * (float)v.W;
        const unsigned rows = (unsigned)v.H + 1 < G.ng ? (unsigned)v.H + 1 : G.ng;
        const float qp = (float)rows * (float)G.Q * (float)G.P;
        unsigned twl = G.twl_max;
        // records of a tile ~ cell rows * (tw + 1) * (samples per pixel)
        while (twl > G.twl_min && qp * (float)((1u << twl) + 1) > (float)G.vstar * hw) --twl;
        unsigned K = 1;
        if (G.split) {
            const float vt = qp * (float)((1u << twl) + 1) / hw;
            K = (unsigned)ceilf(vt / (float)G.vstar);
            if (K < 1) K = 1;
            if (K > kMaxSplit) K = kMaxSplit;
        }
        v.twl = twl;
        v.K = K;
        v.ntx = ((unsigned)v.W + (1u << twl) - 1) >> twl;
        v.nty = fast_div((unsigned)v.H + G.thp - 1, G.thpdiv);
        v.tbase = v.wbase = v.pbase = 0;
        lv[l] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int pass = 0; pass < 2; ++pass) {  // pass 1: without splitting, if the partial rows do not fit
            unsigned nt = 0, nw = 0, np = 0;
            for (unsigned l = 0; l < G.L; ++l) {
                const unsigned ntile = lv[l].nty * lv[l].ntx;
                if (pass) lv[l].K = 1;
                lv[l].tbase = nt;
                lv[l].wbase = nw;
                lv[l].pbase = np;
                nt += ntile;
                nw += ntile * lv[l].K;
                if (lv[l].K > 1) np += ntile * lv[l].K * (G.thp << lv[l].twl);
            }
            tot[0] = nt; tot[1] = nw; tot[2] = np;
            if (np <= G.prows_max) break;
        }
    }
    __syncthreads();
}

struct SampleGeo {
    bool valid;
    int u, v;       // cell
    float lw, lh, a;
};

// Pixel coordinates exactly as the oracle forms them (mul, then sub, no fma contraction), so that
// floor() picks the same pixel.
__device__ __forceinline__ SampleGeo sample_geo(float x, float y, float a, int H, int W)
{
#pragma clang fp contract(off)
    SampleGeo g;
    const float Hf = (float)H, Wf = (float)W;
    const float h_im = y * Hf - 0.5f;
    const float w_im = x * Wf - 0.5f;
    g.valid = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
    const float hf = floorf(h_im), wf = floorf(w_im);
    g.lh = h_im - hf;
    g.lw = w_im - wf;
    g.u = g.valid ? (int)hf + 1 : 0;
    g.v = g.valid ? (int)wf + 1 : 0;
    g.a = a;
    return g;
}

// The tiles that need cell (u, v): its own tile row (walker u mod thp), the tile row above when u is
// the first cell row of a tile row (that tile's last walker), its own segment (step v - x0) and the
// segment to the left when v is the first column of a segment (that segment's step tw).
struct TileSet {
    unsigned nr, nc;
    unsigned ty0, ul0, ty1, ul1, sg0, vl0, sg1, vl1;
};
__device__ __forceinline__ TileSet tiles_of_cell(int u, int v, int W, unsigned nty, unsigned thp, FastDiv thpdiv,
                                                 unsigned twl)
{
    TileSet t;
    const unsigned tw = 1u << twl;
    const unsigned ty = fast_div((unsigned)u, thpdiv), ul = (unsigned)u - ty * thp;
    const bool own = ty < nty;          // (u == H with H a multiple of thp has no tile row of its own)
    const bool above = ul == 0 && u >= 1;
    t.nr = (own ? 1u : 0u) + (above ? 1u : 0u);
    t.ty0 = own ? ty : ty - 1;
    t.ul0 = own ? ul : thp;
    t.ty1 = ty - 1;
    t.ul1 = thp;
    const unsigned vh = (unsigned)(v < W - 1 ? v : W - 1);
    t.sg0 = vh >> twl;
    t.vl0 = (unsigned)v - (t.sg0 << twl);
    const bool colL = v >= 1 && v <= W - 1 && ((unsigned)v & (tw - 1)) == 0;
    t.nc = colL ? 2u : 1u;
    t.sg1 = colL ? ((unsigned)v >> twl) - 1 : 0u;
    t.vl1 = tw;
    return t;
}

#define ZIRA_WAVE_SYNC()                                          \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
    } while (0)

// ------------------------------------------------------------------------------------------
// K1: bin
// ------------------------------------------------------------------------------------------
// tile histogram with two 16-bit counters per LDS word (a block emits < 65536 records)
__device__ __forceinline__ unsigned hist_add(unsigned *hist, unsigned t)
{
    const unsigned sh = (t & 1) * 16;
    return (atomicAdd(&hist[t >> 1], 1u << sh) >> sh) & 0xffffu;
}
__device__ __forceinline__ unsigned hist_get(const unsigned *hist, unsigned t)
{
    return (hist[t >> 1] >> ((t & 1) * 16)) & 0xffffu;
}

__global__ __launch_bounds__(kBinThreads) void msda_bwd_bin(
    const float *__restrict__ loc, const float *__restrict__ attn,
    const int64_t *__restrict__ shapes, CellGeom G, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, unsigned *__restrict__ desc, uint4 *__restrict__ region,
    unsigned *__restrict__ tickets, const float *__restrict__ grad_out, unsigned *__restrict__ amax_blk,
    float *__restrict__ grad_value)
{
    extern __shared__ unsigned lds_bin[];
    Level *lv = reinterpret_cast<Level *>(lds_bin);                 // [kMaxLevels]
    unsigned *misc = lds_bin + kLevelWords * kMaxLevels;            // [16]: totals, wave totals
    unsigned *hist = misc + 16;                                     // [(ntmax + 1) / 2] packed counts, later offsets
    unsigned *ranks = hist + (G.ntmax + 1) / 2;                     // [QB * LP][2]: 4 x u16

    const unsigned g = fast_div(blockIdx.x, G.nblkdiv), blk = blockIdx.x - g * G.nblk;
    const unsigned b = fast_div(g, G.Mdiv), m = g - b * G.M;
    if (blockIdx.x == 0 && threadIdx.x < 8) tickets[threadIdx.x * 16] = 0;  // the walk's work counters, one line per XCD
    load_levels(shapes, G, lv, misc);
    const unsigned NT = misc[0];
    if (NT > G.ntmax) {
        // The level tables do not tile [0, S) (the caller's precondition, include/zira_msda.h): refuse rather than overrun
        // LDS -- every kernel of the call returns here -- but leave defined outputs behind: all three gradients zero.
        const size_t nsamp = (size_t)G.heads / G.M * G.Q * G.M * G.LP, nval = (size_t)G.heads * G.S * G.D;
        for (size_t i = (size_t)blockIdx.x * kBinThreads + threadIdx.x; i < nval; i += (size_t)gridDim.x * kBinThreads) grad_value[i] = 0.f;
        for (size_t i = (size_t)blockIdx.x * kBinThreads + threadIdx.x; i < nsamp; i += (size_t)gridDim.x * kBinThreads) {
            grad_attn[i] = 0.f;
            grad_loc[2 * i] = grad_loc[2 * i + 1] = 0.f;
        }
        return;
    }
    const unsigned NTW = (NT + 1) / 2;
    for (unsigned i = threadIdx.x; i < NTW; i += kBinThreads) hist[i] = 0;
    __syncthreads();

    // (accumulate path) largest |grad_out| of the block's rows and largest |attention weight| of its samples, as bit
    // patterns (NaN > inf > finite): they set the fixed-point scale of the accumulators
    unsigned mx_g = 0, mx_a = 0;
    constexpr unsigned kGoLoads = 4;   // (rows of the block = QB * D / 4 float4 pieces; 256 threads take up to 4 each up front,
    uint4 gob[kGoLoads];               //  so that their latency runs behind the sample pass)
    const unsigned d4 = G.D / 4, ngo = G.QB * d4;
    if (amax_blk) {
#pragma unroll
        for (unsigned r = 0; r < kGoLoads; ++r) {
            const unsigned i = threadIdx.x + r * kBinThreads;
            const unsigned ql = i / d4, c = i - ql * d4, q = blk * G.QB + ql;
            gob[r] = make_uint4(0u, 0u, 0u, 0u);
            if (i < ngo && q < G.Q)
                gob[r] = *reinterpret_cast<const uint4 *>(grad_out + ((size_t)(b * G.Q + q) * G.M + m) * G.D + c * 4);
        }
    }
    const unsigned nsamp = G.QB * G.LP;
    for (unsigned idx = threadIdx.x; idx < nsamp; idx += kBinThreads) {
        const unsigned ql = fast_div(idx, G.LPdiv), s = idx - ql * G.LP, q = blk * G.QB + ql;
        if (q >= G.Q) continue;
        const size_t si = ((size_t)(b * G.Q + q) * G.M + m) * G.LP + s;
        const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * si);
        const unsigned l = fast_div(s, G.Pdiv);
        const Level L = lv[l];
        const SampleGeo geo = sample_geo(xy.x, xy.y, attn[si], L.H, L.W);
        mx_a = max(mx_a, __float_as_uint(geo.a) & 0x7fffffffu);
        if (!geo.valid) {  // contributes nothing anywhere (cuh:288): its gradients are zero
            *reinterpret_cast<float2 *>(grad_loc + 2 * si) = make_float2(0.f, 0.f);
            grad_attn[si] = 0.f;
            continue;
        }
        const TileSet t = tiles_of_cell(geo.u, geo.v, L.W, L.nty, G.thp, G.thpdiv, L.twl);
        const unsigned t00 = L.tbase + t.ty0 * L.ntx, t10 = L.tbase + t.ty1 * L.ntx;
        unsigned r00 = hist_add(hist, t00 + t.sg0), r01 = 0, r10 = 0, r11 = 0;
        if (t.nc > 1) r01 = hist_add(hist, t00 + t.sg1);
        if (t.nr > 1) {
            r10 = hist_add(hist, t10 + t.sg0);
            if (t.nc > 1) r11 = hist_add(hist, t10 + t.sg1);
        }
        ranks[idx * 2] = r00 | (r01 << 16);
        ranks[idx * 2 + 1] = r10 | (r11 << 16);
    }
    if (amax_blk) {
        {
#pragma unroll
            for (unsigned r = 0; r < kGoLoads; ++r) {
                const unsigned m01 = max(gob[r].x & 0x7fffffffu, gob[r].y & 0x7fffffffu);
                const unsigned m23 = max(gob[r].z & 0x7fffffffu, gob[r].w & 0x7fffffffu);
                mx_g = max(mx_g, max(m01, m23));
            }
            for (unsigned i = threadIdx.x + kGoLoads * kBinThreads; i < ngo; i += kBinThreads) {   // (D = 64 with 64 queries)
                const unsigned ql = i / d4, c = i - ql * d4, q = blk * G.QB + ql;
                if (q >= G.Q) continue;
                const uint4 gb = *reinterpret_cast<const uint4 *>(grad_out + ((size_t)(b * G.Q + q) * G.M + m) * G.D + c * 4);
                mx_g = max(mx_g, max(max(gb.x & 0x7fffffffu, gb.y & 0x7fffffffu), max(gb.z & 0x7fffffffu, gb.w & 0x7fffffffu)));
            }
        }
        if (threadIdx.x < 2) misc[12 + threadIdx.x] = 0;
        __syncthreads();
        atomicMax(&misc[12], mx_g);
        atomicMax(&misc[13], mx_a);
    }
    __syncthreads();
    if (amax_blk && threadIdx.x < 2) amax_blk[2 * blockIdx.x + threadIdx.x] = misc[12 + threadIdx.x];

    // exclusive scan of the histogram, two tiles per thread; hist[] becomes the offsets and this
    // block's descriptor row gets {offset << 16 | count} for every tile of the head
    unsigned *wave_tot = misc + 4;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *dcol = desc + (size_t)g * NT * G.nblk + blk;  // [tile][block]: a walk wave reads one row
    unsigned total = 0;
    for (unsigned c0 = 0; c0 < NTW; c0 += kBinThreads) {
        const unsigned wi = c0 + threadIdx.x;
        const unsigned packed = wi < NTW ? hist[wi] : 0u;
        const unsigned n0 = packed & 0xffffu, n1 = packed >> 16, n_mine = n0 + n1;
        unsigned incl = n_mine;
#pragma unroll
        for (unsigned d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned wbase = 0, ctot = 0;
#pragma unroll
        for (unsigned w = 0; w < kBinThreads / 64; ++w) {
            const unsigned tot = wave_tot[w];
            if (w < wave) wbase += tot;
            ctot += tot;
        }
        const unsigned o0 = total + wbase + incl - n_mine, o1 = o0 + n0;
        if (wi < NTW) {
            hist[wi] = (o0 & 0xffffu) | (o1 << 16);
            dcol[(size_t)(2 * wi) * G.nblk] = (o0 << 16) | n0;
            if (2 * wi + 1 < NT) dcol[(size_t)(2 * wi + 1) * G.nblk] = (o1 << 16) | n1;
        }
        total += ctot;
        __syncthreads();
    }

    uint4 *out = region + ((size_t)g * G.nblk + blk) * G.slice;
    const unsigned ulb = kUlBits;
    for (unsigned idx = threadIdx.x; idx < nsamp; idx += kBinThreads) {
        const unsigned ql = fast_div(idx, G.LPdiv), s = idx - ql * G.LP, q = blk * G.QB + ql;
        if (q >= G.Q) continue;
        const size_t si = ((size_t)(b * G.Q + q) * G.M + m) * G.LP + s;
        const float2 xy = *reinterpret_cast<const float2 *>(loc + 2 * si);
        const unsigned l = fast_div(s, G.Pdiv);
        const Level L = lv[l];
        const SampleGeo geo = sample_geo(xy.x, xy.y, attn[si], L.H, L.W);
        if (!geo.valid) continue;
        const TileSet t = tiles_of_cell(geo.u, geo.v, L.W, L.nty, G.thp, G.thpdiv, L.twl);
        const unsigned r0 = ranks[idx * 2], r1 = ranks[idx * 2 + 1];
        const unsigned t00 = L.tbase + t.ty0 * L.ntx, t10 = L.tbase + t.ty1 * L.ntx;
        uint4 rec;
        rec.y = __float_as_uint(geo.lw);
        rec.z = __float_as_uint(geo.lh);
        rec.w = __float_as_uint(geo.a);
        const unsigned w0 = q | ((s - l * G.P) << kQBits);
        const unsigned c0w = t.vl0 << (kCellShift + ulb), c1w = t.vl1 << (kCellShift + ulb);
        rec.x = w0 | (t.ul0 << kCellShift) | c0w;
        out[hist_get(hist, t00 + t.sg0) + (r0 & 0xffffu)] = rec;
        if (t.nc > 1) {
            rec.x = w0 | (t.ul0 << kCellShift) | c1w;
            out[hist_get(hist, t00 + t.sg1) + (r0 >> 16)] = rec;
        }
        if (t.nr > 1) {
            rec.x = w0 | (t.ul1 << kCellShift) | c0w;
            out[hist_get(hist, t10 + t.sg0) + (r1 & 0xffffu)] = rec;
            if (t.nc > 1) {
                rec.x = w0 | (t.ul1 << kCellShift) | c1w;
                out[hist_get(hist, t10 + t.sg1) + (r1 >> 16)] = rec;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: walk
// ------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x)
{
    return x + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}

// sum over the LPG (4, 8 or 16) consecutive lanes of a walker; every lane gets the total
template <int LPG>
__device__ __forceinline__ float group_sum(float x)
{
    x = dpp_add<0xB1>(x);                  // quad_perm:[1,0,3,2]
    x = dpp_add<0x4E>(x);                  // quad_perm:[2,3,0,1]
    if (LPG >= 8) x = dpp_add<0x141>(x);   // row_half_mirror
    if (LPG >= 16) x = dpp_add<0x140>(x);  // row_mirror
    return x;
}

// lane (8 g + j) <- lane (8 g + I): `row_newbcast` broadcasts one lane of every 16-lane row; the two halves of a
// row take different source lanes, selected with the bank mask (banks = 4 lanes)
template <unsigned I>
__device__ __forceinline__ unsigned bcast8(unsigned x)
{
    unsigned r = __builtin_amdgcn_update_dpp(0u, x, 0x150 + I, 0xf, 0xf, true);
    return __builtin_amdgcn_update_dpp(r, x, 0x150 + 8 + I, 0xf, 0xc, false);
}

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, unsigned lane)
{
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ float dot4(float4 a, float4 b, float acc)
{
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}
__device__ __forceinline__ void axpy4(float4 &acc, float w, float4 g)
{
    acc.x = fmaf(w, g.x, acc.x);
    acc.y = fmaf(w, g.y, acc.y);
    acc.z = fmaf(w, g.z, acc.z);
    acc.w = fmaf(w, g.w, acc.w);
}

// work item -> (level, tile row, segment, k)
struct Item {
    unsigned l, ty, seg, k, tile;
};
__device__ __forceinline__ Item decode_item(const Level *lv, unsigned L, unsigned item)
{
    Item it;
    unsigned l = 0;
    while (l + 1 < L && lv[l + 1].wbase <= item) ++l;
    const unsigned rem = item - lv[l].wbase;
    const unsigned bt = rem / lv[l].K;
    it.l = l;
    it.k = rem - bt * lv[l].K;
    it.ty = bt / lv[l].ntx;
    it.seg = bt - it.ty * lv[l].ntx;
    it.tile = lv[l].tbase + bt;
    return it;
}

// The stream of a work item is a padded 2-D array in LDS, vis[k][walker] (4 bytes per element): the
// records of all NG walkers are laid out so that at position k every walker is in the same step c,
// shorter lists padded with idle words.  What happens between steps -- the finished pixel is stored,
// the accumulators and the value columns move on, the next value column is requested -- is then
// done once per wave and step, with no divergence.
//   stream word: bits 0-23 record index in the head's region, bit 29 "home" (this visit also does the
//   sample's grad_sampling_loc / grad_attn_weight); idle = 0xFFFFFFFF
constexpr unsigned kFlagBit = 1u << 29;
constexpr unsigned kIdleWord = 0xFFFFFFFFu;

template <int D, int LPG>
__global__ __launch_bounds__(64, ZIRA_WALK_MINWAVES) void msda_bwd_walk(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start, CellGeom G,
    const unsigned *__restrict__ desc, const uint4 *__restrict__ region, float *__restrict__ partial,
    unsigned *__restrict__ tickets, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_attn, const unsigned *__restrict__ only_if)
{
    if (only_if && *only_if == 0) return;   // (behind msda_bwd_accum: only when that kernel declined)
    constexpr unsigned NG = 64 / LPG;       // walkers per wave
    constexpr unsigned THP = NG - 1;        // pixel rows per tile
    constexpr unsigned NV = D / (4 * LPG);  // float4 pieces of a row per lane
    static_assert(NV >= 1, "a lane holds at least 4 channels");
    extern __shared__ unsigned lds_walk[];
    Level *lv = reinterpret_cast<Level *>(lds_walk);         // [kMaxLevels]
    unsigned *misc = lds_walk + kLevelWords * kMaxLevels;    // [16]
    unsigned *runpre = misc + 16;                            // [nblk + 1] record prefix of the non-empty runs
    unsigned *runoff = runpre + G.nblk + 1;                  // [nblk]     position of the run in the head's region
    const unsigned TW1max = (1u << G.twl_max) + 1;
    unsigned *cnt = runoff + G.nblk;                         // [NG * TW1max] records per (walker, step); later ranks
    unsigned *segend = cnt + NG * TW1max;                    // [TW1max] padded end of every step
    unsigned *vis = segend + TW1max;                         // [cap]

    const unsigned lane = threadIdx.x;
    const unsigned grp = lane / LPG, j = lane % LPG;
    load_levels(shapes, G, lv, misc);
    const unsigned NT = misc[0], NW = misc[1];
    if (NT > G.ntmax) return;
    const unsigned nvirt = G.heads * NW, per = (nvirt + 7) >> 3;
    const unsigned xcd = blockIdx.x & 7;
    constexpr unsigned ulmask = (1u << kUlBits) - 1, vlmask = (1u << (9 - kUlBits)) - 1;

    // The work items of an XCD's share (a contiguous range: whole heads, so that a head's grad_out slice
    // stays in that XCD's L2) are handed out by a counter: items differ in cost, a fixed assignment
    // left a third of the waves idle at the end.
    for (;;) {
        unsigned tix = 0;
        if (lane == 0) tix = atomicAdd(&tickets[xcd * 16], 1u);
        tix = __builtin_amdgcn_readfirstlane(tix);
        const unsigned vt = xcd * per + tix;
        if (tix >= per || vt >= nvirt) break;
        const unsigned head = vt / NW;
        // last items first: the coarse levels hold the heaviest items (most records per pixel)
        const Item it = decode_item(lv, G.L, NW - 1 - (vt - head * NW));
        const unsigned b = fast_div(head, G.Mdiv), m = head - b * G.M;
        const Level Lv = lv[it.l];
        const unsigned tw = 1u << Lv.twl, TW1 = tw + 1, NB = NG * TW1;
        const int H = Lv.H, W = Lv.W;
        const unsigned st = (unsigned)start[it.l];
        const uint4 *reg_h = region + (size_t)head * G.nblk * G.slice;

        // ---- the tile's runs: one (possibly empty) per bin block of the head -------------------
        unsigned n = 0, nruns = 0;
        const unsigned *drow = desc + ((size_t)head * NT + it.tile) * G.nblk;
        for (unsigned c0 = 0; c0 < G.nblk; c0 += 64 * 8) {
            unsigned d[8];
#pragma unroll
            for (unsigned k = 0; k < 8; ++k) {  // (all descriptor loads of the chunk in flight together)
                const unsigned i = c0 + 64 * k + lane;
                d[k] = i < G.nblk ? drow[i] : 0u;
            }
#pragma unroll
            for (unsigned k = 0; k < 8; ++k) {
                if (c0 + 64 * k < G.nblk) {  // wave-uniform
                    const unsigned i = c0 + 64 * k + lane;
                    const unsigned cn = d[k] & 0xffffu;
                    const unsigned incl = wave_incl_scan(cn, lane);
                    const unsigned long long mask = __ballot(cn != 0);
                    const unsigned slot = nruns + __popcll(mask & ((1ull << lane) - 1));
                    if (cn) {
                        runpre[slot] = n + incl - cn;
                        runoff[slot] = i * G.slice + (d[k] >> 16);
                    }
                    n += __shfl(incl, 63);
                    nruns += __popcll(mask);
                }
            }
        }
        if (lane == 0) runpre[nruns] = n;
        ZIRA_WAVE_SYNC();
        // This work item's share of the tile's records: every K-th one, and when they do not fit one pass,
        // every (K * NP)-th one per pass.  Interleaved on purpose: consecutive records come from consecutive
        // queries, i.e. from one spatial band, and would all land in one or two walkers of the tile.
        const unsigned n_mine = n > it.k ? (n - it.k + Lv.K - 1) / Lv.K : 0u;
        const unsigned RC = G.cap / 3;  // (the records' words wait behind the stream buffer, see below)
        unsigned NP = (n_mine + RC - 1) / RC;
        if (NP < 1) NP = 1;
        const unsigned stride = Lv.K * NP;

        // walker state that does not depend on the pass.  Addresses are a wave-uniform base plus a 32-bit
        // element offset (one head's slice of value / grad_out is far below 2^32 bytes)
        const int y = (int)(it.ty * THP + grp);           // the walker's cell row = the pixel row it accumulates
        const bool owner = grp < THP && y < H;            // ... and stores (the last walker serves the tile's last row)
        const bool has_bot = y < H, has_top = y >= 1 && y <= H;
        const int x0 = (int)(it.seg * tw);                // first pixel of the segment
        const unsigned rs = G.M * D;                      // floats between consecutive pixels / queries of a head
        const float *vbase = value + (((size_t)b * G.S + st) * G.M + m) * D;       // pixel 0 of the level, this head
        float *gvbase = grad_value + (((size_t)b * G.S + st) * G.M + m) * D;
        const unsigned yb = (unsigned)(y < H ? y : H - 1), ytp = (unsigned)(y >= 1 ? (y - 1 < H ? y - 1 : H - 1) : 0);
        const unsigned rowoff_b = yb * (unsigned)W * rs + j * 4;     // value row y     (clamped; masked below)
        const unsigned rowoff_t = ytp * (unsigned)W * rs + j * 4;    // value row y - 1
        const unsigned okb_mask = has_bot ? 0xFFFFFFFFu : 0u, okt_mask = has_top ? 0xFFFFFFFFu : 0u;
        // where pixel x of this walker goes: grad_value, or the work item's partial rows
        float *obase;
        unsigned ooff, ostride;
        if (Lv.K > 1) {
            const size_t prow = (size_t)head * G.prows_max + Lv.pbase +
                                (size_t)((it.tile - Lv.tbase) * Lv.K + it.k) * (THP * tw);
            obase = partial + prow * D;
            ooff = (grp < THP ? grp : 0u) * tw * D + j * 4;
            ostride = D;
        } else {
            obase = gvbase;
            ooff = rowoff_b + (unsigned)x0 * rs;
            ostride = rs;
        }
        const float *gbase = grad_out + ((size_t)b * G.Q * G.M + m) * D;  // query 0, this head
        float *ga_h = grad_attn + ((size_t)b * G.Q * G.M + m) * G.LP + (size_t)it.l * G.P;  // + q * M * LP + p
        float *gl_h = grad_loc + 2 * (((size_t)b * G.Q * G.M + m) * G.LP + (size_t)it.l * G.P);
        const unsigned mlp = G.M * G.LP;

        // Passes.  The records' (run, index, walker, step) words wait at the end of the stream buffer while
        // the stream is laid out at its start; a pass whose padded stream does not fit takes its records in
        // several chunks (nb halved until it fits).
        bool rmw = false;
        for (unsigned pass = 0; pass < NP; ++pass) {
          const unsigned first = it.k + Lv.K * pass;                     // record index of the pass's first record
          const unsigned npass = n > first ? (n - first + stride - 1) / stride : 0u;
          for (unsigned r_lo = 0; r_lo == 0 || r_lo < npass;) {
            if (r_lo == 0 && pass > 0 && npass == 0) break;              // (nothing left for this pass)
            unsigned nb = npass - r_lo < RC ? npass - r_lo : RC;
            unsigned Ltot;
            for (;;) {
                for (unsigned i = lane; i < NB; i += 64) cnt[i] = 0;
                ZIRA_WAVE_SYNC();
                // ---- sweep 1 (global): count the records per (walker, step), keep their words in LDS ----
                unsigned rp = 0;
                for (unsigned i0 = lane; i0 < nb; i0 += 64 * 8) {
                    unsigned w0[8], tg[8];
#pragma unroll
                    for (unsigned k = 0; k < 8; ++k) {
                        const unsigned i = i0 + 64 * k;
                        w0[k] = 0; tg[k] = 0;
                        if (i < nb) {
                            const unsigned e = first + (r_lo + i) * stride;
                            while (e >= runpre[rp + 1]) ++rp;
                            const unsigned idx = e - runpre[rp];
                            tg[k] = rp | (idx << 10);
                            w0[k] = reinterpret_cast<const unsigned *>(reg_h + runoff[rp] + idx)[0];
                        }
                    }
#pragma unroll
                    for (unsigned k = 0; k < 8; ++k) {
                        const unsigned i = i0 + 64 * k;
                        if (i < nb) {
                            const unsigned ul = (w0[k] >> kCellShift) & ulmask;
                            const unsigned vl = (w0[k] >> (kCellShift + kUlBits)) & vlmask;
                            atomicAdd(&cnt[ul * TW1 + vl], 1u);
                            vis[G.cap - 1 - i] = tg[k] | (ul << 22) | (vl << 26);
                        }
                    }
                }
                ZIRA_WAVE_SYNC();
                // padded length of every step = the longest list of any walker; lane = step
                unsigned mx = 0;
                if (lane < TW1)
                    for (unsigned r = 0; r < NG; ++r) {
                        const unsigned c = cnt[r * TW1 + lane];
                        mx = c > mx ? c : mx;
                    }
                const unsigned incl = wave_incl_scan(mx, lane);
                if (lane < TW1) segend[lane] = incl;
                Ltot = __shfl(incl, 63);
                ZIRA_WAVE_SYNC();
                if (Ltot * NG + nb <= G.cap || nb <= 1) break;
                nb >>= 1;  // (unbalanced tile) take fewer records in this chunk
            }
            for (unsigned i = lane; i < Ltot * NG; i += 64) vis[i] = kIdleWord;
            for (unsigned i = lane; i < NB; i += 64) cnt[i] = 0;  // now the rank counters
            ZIRA_WAVE_SYNC();
            // ---- sweep 2 (LDS): place the stream words -------------------------------------------
            for (unsigned i = lane; i < nb; i += 64) {
                const unsigned t = vis[G.cap - 1 - i];
                const unsigned pos = runoff[t & 1023u] + ((t >> 10) & 4095u);
                const unsigned ul = (t >> 22) & 15u, vl = t >> 26;
                const int u = (int)(it.ty * THP + ul), v = x0 + (int)vl;
                // the sample's gradients are formed where its cell is at home: in its own segment (not as the
                // left neighbour's step tw) and by its own walker (the tile's last walker only for the cell
                // row below the map, which has no tile of its own)
                const bool home = (vl < tw || v == W) && (ul < THP || u == H);
                const unsigned rk = atomicAdd(&cnt[ul * TW1 + vl], 1u);
                const unsigned s_lo = vl ? segend[vl - 1] : 0u;
                vis[(s_lo + rk) * NG + ul] = pos | (home ? kFlagBit : 0u);
            }
            ZIRA_WAVE_SYNC();

            // ---- walk ----------------------------------------------------------------------------
            float4 aBp[NV], aBc[NV], aTp[NV], aTc[NV];   // pixels (y, c-1), (y, c), (y-1, c-1), (y-1, c)
            float4 vtl[NV], vtr[NV], vtn[NV], vbl[NV], vbr[NV], vbn[NV];  // value rows (y-1, y) x (left, right, next)
#pragma unroll
            for (unsigned k = 0; k < NV; ++k) {
                aBp[k] = aBc[k] = aTp[k] = aTc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                vtl[k] = vtr[k] = vbl[k] = vbr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            auto masked = [](float4 v, unsigned mk) {  // exact zeros where mk == 0, whatever was loaded
                return make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                                   __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
            };
            auto load_col = [&](int x, float4 *t, float4 *bt) {  // value rows (y-1, x) and (y, x), zeros outside
                const unsigned okx = (x >= 0 && x < W) ? 0xFFFFFFFFu : 0u;
                const unsigned xc = (unsigned)(x < 0 ? 0 : (x > W - 1 ? W - 1 : x));
                float4 tb[NV], tt[NV];
#pragma unroll
                for (unsigned k = 0; k < NV; ++k) {  // unconditional loads from clamped addresses
                    tb[k] = *reinterpret_cast<const float4 *>(vbase + (rowoff_b + xc * rs + k * LPG * 4));
                    tt[k] = *reinterpret_cast<const float4 *>(vbase + (rowoff_t + xc * rs + k * LPG * 4));
                }
#pragma unroll
                for (unsigned k = 0; k < NV; ++k) {
                    bt[k] = masked(tb[k], okx & okb_mask);
                    t[k] = masked(tt[k], okx & okt_mask);
                }
            };
            // what happens between step c - 1 and step c (c = 0 .. tw + 1), once per wave
            auto transition = [&](unsigned c) {
                const int xl = (int)c - 2;  // the pixel that step c - 1 completed
                if (xl >= 0) {              // (wave-uniform)
#pragma unroll
                    for (unsigned k = 0; k < NV; ++k) {
                        // the row-above sums of the walker below belong to this walker's pixel row
                        float4 a = aBp[k];
                        a.x += __shfl_down(aTp[k].x, LPG);
                        a.y += __shfl_down(aTp[k].y, LPG);
                        a.z += __shfl_down(aTp[k].z, LPG);
                        a.w += __shfl_down(aTp[k].w, LPG);
                        if (owner && x0 + xl < W) {
                            float4 *p = reinterpret_cast<float4 *>(obase + (ooff + (unsigned)xl * ostride + k * LPG * 4));
                            if (rmw) {
                                const float4 o = *p;
                                a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
                            }
                            *p = a;
                        }
                    }
                }
#pragma unroll
                for (unsigned k = 0; k < NV; ++k) {
                    aBp[k] = aBc[k]; aBc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    aTp[k] = aTc[k]; aTc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    vtl[k] = vtr[k]; vtr[k] = vtn[k];
                    vbl[k] = vbr[k]; vbr[k] = vbn[k];
                }
                if (c < tw) load_col(x0 + (int)c + 1, vtn, vbn);
            };
            load_col(x0 - 1, vtr, vbr);
            load_col(x0, vtn, vbn);

            // Ring of NR stream elements per walker: the record of element k + DF and the grad_out row of
            // element k + DR are requested while element k is processed, so no wait is ever for the
            // youngest load in flight (memory latency ~ 1 us >> the instructions of an element).
            constexpr unsigned DF = ZIRA_WALK_DF, DR = ZIRA_WALK_DR, NR = DF + 1;
            static_assert(DR >= 1 && DF > DR, "records are fetched ahead of the rows");
            struct Elem {
                unsigned wd;
                uint4 rc;
                float4 row[NV];
            };
            Elem ring[NR];
            auto fetch = [&](Elem &e, unsigned k) {  // stream word (LDS), then the record load
                const bool ok = k < Ltot;
                const unsigned w = vis[(ok ? k : 0u) * NG + grp];
                e.wd = ok ? w : kIdleWord;
                e.rc = reg_h[e.wd != kIdleWord ? (e.wd & ((1u << kRefBits) - 1)) : 0u];
            };
            auto issue_row = [&](Elem &e) {
                const unsigned q = e.wd != kIdleWord ? (e.rc.x & ((1u << kQBits) - 1)) : 0u;
                const unsigned o = q * rs + j * 4;
#pragma unroll
                for (unsigned k = 0; k < NV; ++k)
                    e.row[k] = *reinterpret_cast<const float4 *>(gbase + (o + k * LPG * 4));
            };
            auto visit = [&](const Elem &e) {
                if (e.wd == kIdleWord) return;
                const float lw = __uint_as_float(e.rc.y), lh = __uint_as_float(e.rc.z), a = __uint_as_float(e.rc.w);
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float s0 = hh * a, s1 = lh * a;  // corners dy = 0 (row above) / dy = 1 (this row)
                const float wTl = s0 * hw, wTr = s0 * lw, wBl = s1 * hw, wBr = s1 * lw;
#pragma unroll
                for (unsigned k = 0; k < NV; ++k) {
                    axpy4(aTp[k], wTl, e.row[k]);
                    axpy4(aTc[k], wTr, e.row[k]);
                    axpy4(aBp[k], wBl, e.row[k]);
                    axpy4(aBc[k], wBr, e.row[k]);
                }
                if (e.wd & kFlagBit) {
                    // dots with the value rows of the four corners: top-left, top-right, bottom-left, bottom-right
                    float p00 = 0.f, p01 = 0.f, p10 = 0.f, p11 = 0.f;
#pragma unroll
                    for (unsigned k = 0; k < NV; ++k) {
                        p00 = dot4(e.row[k], vtl[k], p00);
                        p01 = dot4(e.row[k], vtr[k], p01);
                        p10 = dot4(e.row[k], vbl[k], p10);
                        p11 = dot4(e.row[k], vbr[k], p11);
                    }
                    // (explicitly rounded operations: every unrolled copy of this code must give the same bits)
                    float ga = __fmul_rn(__fmul_rn(hh, hw), p00);
                    ga = fmaf(__fmul_rn(hh, lw), p01, ga);
                    ga = fmaf(__fmul_rn(lh, hw), p10, ga);
                    ga = fmaf(__fmul_rn(lh, lw), p11, ga);
                    float gx = fmaf(hh, __fsub_rn(p01, p00), __fmul_rn(lh, __fsub_rn(p11, p10)));
                    float gy = fmaf(hw, __fsub_rn(p10, p00), __fmul_rn(lw, __fsub_rn(p11, p01)));
                    ga = group_sum<LPG>(ga);
                    gx = group_sum<LPG>(gx);
                    gy = group_sum<LPG>(gy);
                    if (j == 0) {
                        const unsigned q = e.rc.x & ((1u << kQBits) - 1);
                        const unsigned pp = (e.rc.x >> kQBits) & ((1u << kPBits) - 1);
                        const unsigned oi = q * mlp + pp;
                        ga_h[oi] = ga;
                        *reinterpret_cast<float2 *>(gl_h + 2 * oi) =
                            make_float2(__fmul_rn(__fmul_rn((float)W, a), gx), __fmul_rn(__fmul_rn((float)H, a), gy));
                    }
                }
            };
            // wave-uniform position in the list of steps
            unsigned cur = 0;
            unsigned next_end = __builtin_amdgcn_readfirstlane(segend[0]);
            transition(0);
            auto process = [&](const Elem &e, unsigned k) {
                while (k == next_end && cur < tw) {  // (empty steps: several at once)
                    ++cur;
                    next_end = __builtin_amdgcn_readfirstlane(segend[cur]);
                    transition(cur);
                }
                visit(e);
            };

#pragma unroll
            for (unsigned r = 0; r < DF; ++r) fetch(ring[r], r);
#pragma unroll
            for (unsigned r = 0; r < DR; ++r) issue_row(ring[r]);
            for (unsigned k0 = 0; k0 < Ltot; k0 += NR) {
#pragma unroll
                for (unsigned r = 0; r < NR; ++r) {
                    issue_row(ring[(r + DR) % NR]);
                    fetch(ring[(r + DF) % NR], k0 + r + DF);
                    __builtin_amdgcn_sched_barrier(0);
                    process(ring[r], k0 + r);
                }
            }
            // the steps that are left (they are empty or done), then the last pixel
            for (unsigned c = cur + 1; c <= tw + 1; ++c) transition(c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a later chunk re-reads what this one stored
            ZIRA_WAVE_SYNC();
            rmw = true;
            r_lo += nb > 0 ? nb : 1;
            if (nb == 0) break;
          }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K3: grad_value rows of split tiles = sum of their K partial rows (fixed order)
// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void msda_bwd_fold(const int64_t *__restrict__ shapes,
                                                     const int64_t *__restrict__ start, CellGeom G,
                                                     const float *__restrict__ partial,
                                                     float *__restrict__ grad_value)
{
    constexpr unsigned LPR = D / 4;       // lanes per row
    constexpr unsigned RPB = 256 / LPR;   // rows per block iteration
    __shared__ unsigned lds_fold[kLevelWords * kMaxLevels + 16];
    Level *lv = reinterpret_cast<Level *>(lds_fold);
    unsigned *misc = lds_fold + kLevelWords * kMaxLevels;
    load_levels(shapes, G, lv, misc);
    if (misc[0] > G.ntmax || misc[2] == 0) return;
    const unsigned r = threadIdx.x / LPR, c4 = threadIdx.x % LPR;
    for (unsigned l = 0; l < G.L; ++l) {
        const Level Lv = lv[l];
        if (Lv.K <= 1) continue;
        const unsigned tw = 1u << Lv.twl, tpix = G.thp * tw;
        const unsigned ntile = Lv.nty * Lv.ntx;
        const unsigned st = (unsigned)start[l];
        // units = (head, tile, pixel in tile), grid-strided
        const unsigned long long units = (unsigned long long)G.heads * ntile * tpix;
        for (unsigned long long i = (unsigned long long)blockIdx.x * RPB + r; i < units;
             i += (unsigned long long)gridDim.x * RPB) {
            const unsigned pix = (unsigned)(i % tpix);
            const unsigned long long ht = i / tpix;
            const unsigned bt = (unsigned)(ht % ntile), head = (unsigned)(ht / ntile);
            const unsigned ty = bt / Lv.ntx, seg = bt - ty * Lv.ntx;
            const unsigned y = ty * G.thp + pix / tw, x = seg * tw + pix % tw;
            if (y >= (unsigned)Lv.H || x >= (unsigned)Lv.W) continue;
            const float *p = partial + ((size_t)head * G.prows_max + Lv.pbase + (size_t)bt * Lv.K * tpix + pix) * D + c4 * 4;
            float4 acc = *reinterpret_cast<const float4 *>(p);
            for (unsigned k = 1; k < Lv.K; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)k * tpix * D);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const unsigned b = fast_div(head, G.Mdiv), m = head - b * G.M;
            *reinterpret_cast<float4 *>(grad_value + (((size_t)b * G.S + st + (size_t)y * Lv.W + x) * G.M + m) * D + c4 * 4) = acc;
        }
    }
}


// ------------------------------------------------------------------------------------------
// K2': accumulate (dense calls, D = 32)
// ------------------------------------------------------------------------------------------
// The walk is bound by the instructions that keep 16 walkers in lockstep (sorting a tile's records by
// (walker, step), padding, the transitions).  This kernel does not order anything: a block takes a
// work item, puts the tile's value rows (plus the one-pixel frame the home cells need) and a zeroed
// accumulator tile into LDS and streams the item's records in the order the bin kernel left them.
// Eight lanes (four channels each) handle one record: the four <grad_out, value row> dots from the LDS
// value tile give grad_attn / grad_loc for home records, and the four corner rows are ADDED TO THE LDS
// ACCUMULATORS -- not as floats: `ds_add_f32` executes at ~3 cycles per LANE on gfx950 (193 cycles per
// wave instruction measured), `ds_add_u64` at 7 cycles per wave instruction.  Every term w * (a * g)
// (w = the corner's bilinear weight and a * g rounded to fp32 as in the reference, cuh:117-147; their
// product taken exactly in double) is scaled by a power of two chosen from max|grad_out| * max|attn|
// (found by the bin kernel) so that it lies below 2^38, rounded to an integer and added as a 64-bit
// integer.  The sum of a pixel is therefore EXACT (no rounding between
// terms), independent of the order of the records, and rounded to fp32 once
// when the tile is flushed: grad_value becomes run-to-run identical and at least as close to the
// reference as an fp32 accumulation in any order.  Non-finite inputs (the bin kernel's maxima catch
// them) cannot be represented: the kernel then leaves everything to the walk, which is launched
// behind it and returns at once in the normal case.
constexpr double kMagic = 6755399441055744.0;   // 1.5 * 2^52: (double)x + kMagic holds rint(x) in its low mantissa bits
constexpr int kAccBits = 38;

template <int D, int NTHR>
__global__ __launch_bounds__(NTHR, ZIRA_ACC_MINW) void msda_bwd_accum(
    const float *__restrict__ grad_out, const float *__restrict__ value,
    const int64_t *__restrict__ shapes, const int64_t *__restrict__ start, CellGeom G,
    const unsigned *__restrict__ desc, const uint4 *__restrict__ region, float *__restrict__ partial,
    unsigned *__restrict__ tickets, const unsigned *__restrict__ amax_blk, unsigned *__restrict__ flag,
    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr unsigned LPS = D / 4;        // lanes per record
    constexpr unsigned RPW = 64 / LPS;     // records per wave step
    constexpr unsigned NWV = NTHR / 64;
    static_assert(LPS == 8, "D = 32");
    extern __shared__ unsigned lds_acc[];
    Level *lv = reinterpret_cast<Level *>(lds_acc);            // [kMaxLevels]
    unsigned *misc = lds_acc + kLevelWords * kMaxLevels;       // [16]
    unsigned *runpre = misc + 16;                              // [nblk + 1]
    unsigned *runoff = runpre + G.nblk + 1;                    // [nblk]
    const unsigned twm = 1u << G.twl_max, VC = twm + 2, VR = G.thp + 2;
    unsigned *hmax = runoff + G.nblk;                          // [2 * heads]: max |grad_out|, max |attn| of every head (bit patterns)
    unsigned hdr = kLevelWords * kMaxLevels + 16 + 2 * G.nblk + 1 + 2 * G.heads;
    hdr = (hdr + 3) & ~3u;
    float *val = reinterpret_cast<float *>(lds_acc + hdr);     // [VR][VC][D]  pixels (ty*thp - 1 + r, x0 - 1 + c)
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(val + VR * VC * D);  // [thp][twm][D]

    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned grp = lane / LPS, j = lane % LPS;
    const unsigned oddg = grp & 1u;
    load_levels(shapes, G, lv, misc);
    const unsigned NT = misc[0], NW = misc[1];
    if (NT > G.ntmax) {   // (the bin kernel zero-filled the outputs; the flag the walk launch reads must not be garbage)
        if (blockIdx.x == 0 && tid == 0) *flag = 0u;
        return;
    }

    // The bin kernel's maxima, PER HEAD: the fixed-point scale of an item follows its own head's largest possible
    // term, so one head (or image) with outsized gradients does not cost the others their low bits.
    {
        if (tid < 2) misc[8 + tid] = 0;
        __syncthreads();
        unsigned ag = 0, aa = 0;
        for (unsigned hd = wave; hd < G.heads; hd += NWV) {   // a wave per head: lanes stride over the head's bin blocks
            unsigned ug = 0, ua = 0;
            for (unsigned i = lane; i < G.nblk; i += 64) {
                const uint2 w = reinterpret_cast<const uint2 *>(amax_blk)[(size_t)hd * G.nblk + i];
                ug = max(ug, w.x);
                ua = max(ua, w.y);
            }
#pragma unroll
            for (unsigned d = 32; d >= 1; d >>= 1) {
                ug = max(ug, (unsigned)__shfl_xor((int)ug, (int)d));
                ua = max(ua, (unsigned)__shfl_xor((int)ua, (int)d));
            }
            if (lane == 0) {
                hmax[2 * hd] = ug;
                hmax[2 * hd + 1] = ua;
            }
            ag = max(ag, ug);
            aa = max(aa, ua);
        }
        if (lane == 0) {
            atomicMax(&misc[8], ag);
            atomicMax(&misc[9], aa);
        }
        __syncthreads();
    }
    const bool bad = misc[8] >= 0x7f800000u || misc[9] >= 0x7f800000u;
    if (blockIdx.x == 0 && tid == 0) *flag = bad ? 1u : 0u;
    if (bad) return;

    const unsigned nvirt = G.heads * NW, per = (nvirt + 7) >> 3;
    const unsigned xcd = blockIdx.x & 7;
    constexpr unsigned ulmask = (1u << kUlBits) - 1, vlmask = (1u << (9 - kUlBits)) - 1;
    const unsigned THP = G.thp;
    const unsigned rs = G.M * D;
    const unsigned mlp = G.M * G.LP;

    for (;;) {
        __syncthreads();  // (the previous item's LDS is no longer read)
        if (tid == 0) misc[10] = atomicAdd(&tickets[xcd * 16], 1u);   // (taking it an item ahead loses: 435 against 401 us)
        __syncthreads();
        const unsigned tix = misc[10];
        const unsigned vt = xcd * per + tix;
        if (tix >= per || vt >= nvirt) break;
        const unsigned head = vt / NW;
        const Item it = decode_item(lv, G.L, NW - 1 - (vt - head * NW));
        const unsigned b = fast_div(head, G.Mdiv), m = head - b * G.M;
        // |g| < 2^(eg + 1), |a| < 2^(ea + 1) in this head: every term w * (a * g) (w <= 1) is below 2^(eg + ea + 2).  The
        // power of two that takes it below 2^kAccBits is applied in two exact steps so that no intermediate leaves the
        // normal range whatever the magnitudes: sA goes into the attention weight (a * g then peaks near 1), sC into
        // the corner weight.
        float scaleA;
        int sA, sBase;   // sBase + bits = the exponent of scaleC for an item whose terms may use `bits` bits
        {
            const unsigned ug = hmax[2 * head], ua = hmax[2 * head + 1];
            const int eg = (int)(ug >> 23) - 127, ea = (int)(ua >> 23) - 127;
            sA = -(eg + ea + 2);
            sA = sA > 120 - ea ? 120 - ea : (sA < -120 - ea ? -120 - ea : sA);
            sA = sA > 126 ? 126 : (sA < -126 ? -126 : sA);
            sBase = -(eg + ea + 2) - sA;
            scaleA = __uint_as_float((unsigned)(sA + 127) << 23);
        }
        const Level Lv = lv[it.l];
        const unsigned tw = 1u << Lv.twl;
        const int H = Lv.H, W = Lv.W;
        const unsigned st = (unsigned)start[it.l];
        const uint4 *reg_h = region + (size_t)head * G.nblk * G.slice;
        const int y0 = (int)(it.ty * THP), x0 = (int)(it.seg * tw);
        const float *vbase = value + (((size_t)b * G.S + st) * G.M + m) * D;
        float *gvbase = grad_value + (((size_t)b * G.S + st) * G.M + m) * D;

        // ---- wave 0: the tile's runs; the others: value tile, zeroed accumulators ---------------
        if (wave == 0) {
            unsigned n = 0, nruns = 0;
            const unsigned *drow = desc + ((size_t)head * NT + it.tile) * G.nblk;
            for (unsigned c0 = 0; c0 < G.nblk; c0 += 64 * 8) {
                unsigned d[8];
#pragma unroll
                for (unsigned k = 0; k < 8; ++k) {
                    const unsigned i = c0 + 64 * k + lane;
                    d[k] = i < G.nblk ? drow[i] : 0u;
                }
#pragma unroll
                for (unsigned k = 0; k < 8; ++k) {
                    if (c0 + 64 * k < G.nblk) {
                        const unsigned i = c0 + 64 * k + lane;
                        const unsigned cn = d[k] & 0xffffu;
                        const unsigned incl = wave_incl_scan(cn, lane);
                        const unsigned long long mask = __ballot(cn != 0);
                        const unsigned slot = nruns + __popcll(mask & ((1ull << lane) - 1));
                        if (cn) {
                            runpre[slot] = n + incl - cn;
                            runoff[slot] = i * G.slice + (d[k] >> 16);
                        }
                        n += __shfl(incl, 63);
                        nruns += __popcll(mask);
                    }
                }
            }
            if (lane == 0) {
                runpre[nruns] = n;
                misc[11] = n;
                misc[12] = nruns;
            }
            ZIRA_WAVE_SYNC();
            // This work item's share of the records: whole runs (the ORDER of the records inside a run is not
            // reproducible -- the bin kernel ranks them with an LDS atomic -- but the set is, and so is every sum
            // over whole runs): share k starts at the first run boundary at or behind n k / K.
            if (lane < 2) {
                const unsigned t = (unsigned)(((unsigned long long)n * (it.k + lane)) / Lv.K);
                unsigned lo = 0, hi = nruns;      // smallest i with runpre[i] >= t
                while (lo < hi) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (runpre[mid] >= t) hi = mid; else lo = mid + 1;
                }
                misc[13 + lane] = runpre[lo];
            }
        }
        for (unsigned i = tid; i < VR * (tw + 2) * LPS; i += NTHR) {
            const unsigned c4 = i % LPS, pix = i / LPS, r = pix / (tw + 2), c = pix - r * (tw + 2);
            const int y = y0 - 1 + (int)r, x = x0 - 1 + (int)c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y >= 0 && y < H && x >= 0 && x < W)
                v = *reinterpret_cast<const float4 *>(vbase + ((size_t)y * W + x) * rs + c4 * 4);
            *reinterpret_cast<float4 *>(val + (r * VC + c) * D + c4 * 4) = v;
        }
        for (unsigned i = tid; i < THP * twm * D / 2; i += NTHR)
            reinterpret_cast<uint4 *>(acc)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        const unsigned n = misc[11], nruns = misc[12];
        const unsigned e0 = misc[13], e1 = misc[14];
        // The integer added for a term is the bit pattern of (term + 1.5 * 2^52) as it comes out of the fma: its low
        // 51 bits are the term in two's complement, the bits above (exponent, the 2^51 of the magic) are the same for
        // every term and only pile up above bit 50.  The sum is therefore read back from the low 51 bits, and the terms
        // get as many bits as leave room for this item's (e1 - e0) records below 2^50: kAccBits for up to 4096.
        int bits = 50 - (32 - (int)__builtin_clz((e1 - e0) | 1u));
        bits = bits > kAccBits ? kAccBits : bits;
        int sC = sBase + bits;
        sC = sC > 120 ? 120 : (sC < -100 ? -100 : sC);
        const float scaleC = __uint_as_float((unsigned)(sC + 127) << 23);
        const double unscale = __longlong_as_double((long long)(1023 - (sA + sC)) << 52);

        const float *gbase = grad_out + ((size_t)b * G.Q * G.M + m) * D;
        float *ga_h = grad_attn + ((size_t)b * G.Q * G.M + m) * G.LP + (size_t)it.l * G.P;
        float *gl_h = grad_loc + 2 * (((size_t)b * G.Q * G.M + m) * G.LP + (size_t)it.l * G.P);
        const float Wa = (float)W, Ha = (float)H;

        // ---- records: 64 per wave and round; group g takes records 8 g .. 8 g + 7 of the round, one per step ------
        // A lane loads ONE record (lane L: record c0 + L, a coalesced kilobyte per wave) a round ahead and derives
        // what does not depend on the channel (accumulator rows of the four corners, value-tile position, output
        // index) ONCE; at step i the group's lanes read those words out of lane 8 g + i with DPP row broadcasts.
        // The grad_out row of step i + DR is requested while step i is computed (ring of NR rows).  Corners that
        // belong to a neighbouring tile (which holds a copy of the record) go to a per-group trash row instead of
        // being branched around.
        constexpr unsigned DR = ZIRA_ACC_DR, NR = DR + 1;
        static_assert(RPW % NR == 0, "ring slots are static across rounds");
        struct Prep {
            unsigned lw, lh, a, a01, a23, vh, oi, qo;
        };
        auto locate = [&](unsigned c0) {
            uint4 rec = make_uint4(kInvalidVisit, 0u, 0u, 0u);
            const unsigned e = c0 + lane;
            if (c0 < e1 && e < e1) {
                unsigned lo = 0, hi = nruns;      // runpre[lo] <= e < runpre[hi]
                while (hi - lo > 1) {
                    const unsigned mid = (lo + hi) >> 1;
                    if (runpre[mid] <= e) lo = mid; else hi = mid;
                }
                rec = reg_h[runoff[lo] + (e - runpre[lo])];
            }
            return rec;
        };
        const unsigned trash = (THP * twm + grp) * D;   // (accumulator rows behind the tile, one per group)
        auto prepare = [&](const uint4 &rec) {
            Prep P;
            const unsigned w0 = rec.x;
            const bool okr = w0 != kInvalidVisit;
            const unsigned ul = (w0 >> kCellShift) & ulmask, vl = (w0 >> (kCellShift + kUlBits)) & vlmask;
            const unsigned q = w0 & ((1u << kQBits) - 1), pp = (w0 >> kQBits) & ((1u << kPBits) - 1);
            unsigned ad[4];
#pragma unroll
            for (unsigned c = 0; c < 4; ++c) {
                const unsigned pr = ul - 1 + (c >> 1), pc = vl - 1 + (c & 1);   // (unsigned: -1 wraps and fails the test)
                ad[c] = (okr && pr < THP && pc < tw) ? (pr * twm + pc) * D : trash;
            }
            const int u = y0 + (int)ul, v = x0 + (int)vl;
            const bool home = okr && (vl < tw || v == W) && (ul < THP || u == H);
            P.lw = rec.y;
            P.lh = rec.z;
            P.a = okr ? rec.w : 0u;
            P.a01 = ad[0] | (ad[1] << 16);
            P.a23 = ad[2] | (ad[3] << 16);
            P.vh = (home ? ((ul * VC + vl) * D) : 0u) | (home ? 0x10000u : 0u);
            P.oi = q * mlp + pp;
            P.qo = okr ? q * rs : 0u;
            return P;
        };
        float4 ring[NR];
        auto issue_row = [&](float4 &dst, unsigned qo) {
            dst = *reinterpret_cast<const float4 *>(gbase + (qo + j * 4));
        };
        Prep cur = prepare(locate(e0 + wave * 64)), nprep;
        uint4 nxt;
        auto step = [&](auto ic) {
            constexpr unsigned i = decltype(ic)::value;
            {   // the row of step i + DR (of the next round when it is past this one's end)
                constexpr unsigned t = i + DR;
                if (t == RPW) nprep = prepare(nxt);
                issue_row(ring[t % NR], t < RPW ? bcast8<t % RPW>(cur.qo) : bcast8<t % RPW>(nprep.qo));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float lw = __uint_as_float(bcast8<i>(cur.lw)), lh = __uint_as_float(bcast8<i>(cur.lh));
            const float a = __uint_as_float(bcast8<i>(cur.a));
            const unsigned a01 = bcast8<i>(cur.a01), a23 = bcast8<i>(cur.a23), vh = bcast8<i>(cur.vh);
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float w00 = __fmul_rn(hh, hw), w01 = __fmul_rn(hh, lw), w10 = __fmul_rn(lh, hw), w11 = __fmul_rn(lh, lw);
            const float4 g4 = ring[i % NR];
            if (vh >> 16) {
                const float *vp = val + ((vh & 0xffffu) + j * 4);
                const float4 v00 = *reinterpret_cast<const float4 *>(vp);
                const float4 v01 = *reinterpret_cast<const float4 *>(vp + D);
                const float4 v10 = *reinterpret_cast<const float4 *>(vp + VC * D);
                const float4 v11 = *reinterpret_cast<const float4 *>(vp + VC * D + D);
                const float p00 = dot4(g4, v00, 0.f), p01 = dot4(g4, v01, 0.f);
                const float p10 = dot4(g4, v10, 0.f), p11 = dot4(g4, v11, 0.f);
                float ga = __fmul_rn(w00, p00);
                ga = fmaf(w01, p01, ga);
                ga = fmaf(w10, p10, ga);
                ga = fmaf(w11, p11, ga);
                float gx = fmaf(hh, __fsub_rn(p01, p00), __fmul_rn(lh, __fsub_rn(p11, p10)));
                float gy = fmaf(hw, __fsub_rn(p10, p00), __fmul_rn(lw, __fsub_rn(p11, p01)));
                ga = group_sum<LPS>(ga);
                gx = group_sum<LPS>(gx);
                gy = group_sum<LPS>(gy);
                const unsigned oi = bcast8<i>(cur.oi);
                if (j == 0) {
                    ga_h[oi] = ga;
                    *reinterpret_cast<float2 *>(gl_h + 2 * oi) =
                        make_float2(__fmul_rn(__fmul_rn(Wa, a), gx), __fmul_rn(__fmul_rn(Ha, a), gy));
                }
            }
            // corner rows: term = w * (a * g), as the reference forms it, times 2^(sA + sC) (exact: folded into a and w)
            const float as = __fmul_rn(a, scaleA);
            // (the two records of a 16-lane LDS group add their words k and k ^ 1 in opposite order: the eight 8-byte
            // words of one then fall into the other half of the banks -- 7.1 instead of 8.6 cycles per ds_add_u64,
            // scripts/lds_atomic_rates2.hip)
            const float gs0 = oddg ? g4.y : g4.x, gs1 = oddg ? g4.x : g4.y, gs2 = oddg ? g4.w : g4.z, gs3 = oddg ? g4.z : g4.w;
            const double tt[4] = {(double)__fmul_rn(gs0, as), (double)__fmul_rn(gs1, as), (double)__fmul_rn(gs2, as),
                                  (double)__fmul_rn(gs3, as)};
            const double wc[4] = {(double)__fmul_rn(w00, scaleC), (double)__fmul_rn(w01, scaleC),
                                  (double)__fmul_rn(w10, scaleC), (double)__fmul_rn(w11, scaleC)};
            const unsigned ad[4] = {a01 & 0xffffu, a01 >> 16, a23 & 0xffffu, a23 >> 16};
#pragma unroll
            for (unsigned c = 0; c < 4; ++c) {
                // accumulator row layout: slot k * LPS + j holds channel 4 j + k, so that the lanes of a record
                // add to consecutive 8-byte words (32-byte lane strides run at half the rate)
                unsigned long long *ap = acc + (ad[c] + j);
#if ZIRA_ACC_SKIP_TRASH
                if (ad[c] != trash)
#endif
#pragma unroll
                for (unsigned k = 0; k < 4; ++k) {
#ifdef ZIRA_DEV_HALF_ADDS   // developer ablation (results wrong): the LDS adds of a packed two-channel form, at best
                    if (k & 1u) continue;
#endif
                    const double dd = fma(wc[c], tt[k], kMagic);   // (the product of two floats is exact in double)
                    atomicAdd(ap + (k ^ oddg) * LPS, (unsigned long long)__double_as_longlong(dd));
                }
            }
        };
        if (e0 + wave * 64 < e1) {
            issue_row(ring[0], bcast8<0>(cur.qo));
            if (DR > 1) issue_row(ring[1 % NR], bcast8<1>(cur.qo));
            if (DR > 2) issue_row(ring[2 % NR], bcast8<2>(cur.qo));
        }
        for (unsigned c0 = e0 + wave * 64; c0 < e1; c0 += NWV * 64) {
            nxt = locate(c0 + NWV * 64);
            step(std::integral_constant<unsigned, 0>{});
            step(std::integral_constant<unsigned, 1>{});
            step(std::integral_constant<unsigned, 2>{});
            step(std::integral_constant<unsigned, 3>{});
            step(std::integral_constant<unsigned, 4>{});
            step(std::integral_constant<unsigned, 5>{});
            step(std::integral_constant<unsigned, 6>{});
            step(std::integral_constant<unsigned, 7>{});
            cur = nprep;
        }
        __syncthreads();

        // ---- flush: every pixel of the tile once ------------------------------------------------
        float *obase;
        size_t orow;      // floats between pixel rows
        unsigned ostride; // floats between pixels
        if (Lv.K > 1) {
            const size_t prow = (size_t)head * G.prows_max + Lv.pbase + (size_t)((it.tile - Lv.tbase) * Lv.K + it.k) * (THP * tw);
            obase = partial + prow * D;
            orow = (size_t)tw * D;
            ostride = D;
        } else {
            obase = gvbase + ((size_t)y0 * W + x0) * rs;
            orow = (size_t)W * rs;
            ostride = rs;
        }
        for (unsigned i = tid; i < THP * tw * LPS; i += NTHR) {
            const unsigned c4 = i % LPS, pix = i / LPS, r = pix / tw, c = pix - r * tw;
            if (y0 + (int)r >= H || x0 + (int)c >= W) continue;
            const long long *ap = reinterpret_cast<const long long *>(acc + ((r * twm + c) * D + c4));
            float4 o;
            o.x = (float)((double)((ap[0] << 13) >> 13) * unscale);        // (sign extension from bit 50)
            o.y = (float)((double)((ap[LPS] << 13) >> 13) * unscale);
            o.z = (float)((double)((ap[2 * LPS] << 13) >> 13) * unscale);
            o.w = (float)((double)((ap[3 * LPS] << 13) >> 13) * unscale);
            *reinterpret_cast<float4 *>(obase + r * orow + (size_t)c * ostride + c4 * 4) = o;
        }
    }
}

// ---- host side ---------------------------------------------------------------------------

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int lpg_for(int D)
{
    if (D == 16) return 4;
    if (D == 32) return ZIRA_WALK_LPG32;
    if (D == 64) return 8;
    return 0;
}

inline size_t bin_lds_bytes(const CellGeom &G)
{
    return (kLevelWords * kMaxLevels + 16 + ((size_t)G.ntmax + 1) / 2 + 2 * (size_t)G.QB * G.LP) * 4;
}

inline bool use_accum(int B, int M, int D, int Q)
{
    return ZIRA_DENSE_ACCUM && D == 32 && (unsigned long long)B * M * Q >= 16 * 4096;
}

inline bool make_geom(int B, int S, int M, int D, int L, int Q, int P, CellGeom &G)
{
    G.D = D;
    const int lpg = lpg_for(D);
    if (!lpg) return false;
    if (L > (int)kMaxLevels || P > (1 << kPBits) || Q >= (1 << kQBits)) return false;
    if ((long long)B * S * M * D >= (1LL << 31) || (long long)B * Q * M * L * P * 2 >= (1LL << 31)) return false;
    G.S = S; G.M = M; G.L = L; G.P = P; G.LP = L * P; G.Q = Q; G.heads = B * M;
    G.ng = 64 / lpg;
    G.thp = G.ng - 1;
    const bool dense = (unsigned long long)G.heads * Q >= 16 * 4096;
    G.twl_max = 3;
    G.twl_min = dense ? 2 : 1;
    G.vstar = dense ? ZIRA_WALK_VSTAR_DENSE : ZIRA_WALK_VSTAR_SPARSE;
    if (use_accum(B, M, D, Q)) {  // one tile shape for all levels; busy levels only get more work items per tile
        G.twl_max = G.twl_min = ZIRA_ACC_TWL;
        G.vstar = ZIRA_ACC_VSTAR;
    }
    G.split = dense ? 1u : 0u;
    G.QB = 64;
    while ((unsigned long long)G.QB * G.LP * 4 > 4096 && G.QB > 1) G.QB >>= 1;  // (12-bit index inside a run)
    if ((unsigned long long)G.QB * G.LP * 4 > 4096) return false;
    G.nblk = (Q + G.QB - 1) / G.QB;
    if (G.nblk > 1024) return false;                                           // (10-bit run number)
    G.slice = G.QB * G.LP * 4;
    if ((unsigned long long)G.nblk * G.slice >= (1ull << kRefBits)) return false;
    // tiles per head for any shapes with sum(H * W) = S and the narrowest segments:
    // ceil(H/thp) * ceil(W/tw) <= H*W*(thp+tw-1)/(thp*tw) + 1 per level
    const unsigned twm = 1u << G.twl_min;
    G.ntmax = (unsigned)(((unsigned long long)S * (G.thp + twm - 1)) / (G.thp * twm)) + L;
    // partial rows per head: a split level has about (its records / vstar) work items of thp * tw rows;
    // all levels together hold ~1.3 records per sample.  The device falls back to K = 1 beyond this.
    G.prows_max = 0;
    if (G.split) {
        const unsigned long long items = (3ull * Q * G.LP / 2) / G.vstar + 4ull * L;
        G.prows_max = (unsigned)(items * G.thp * (1u << G.twl_max) * 2);
    }
    G.cap = (G.vstar * 5 / 2 + 511) & ~511u;  // padded stream: records x (longest list / mean list)
    if (G.cap > 10240) G.cap = 10240;         // (the walk takes larger shares in several passes)
    G.LPdiv = make_fast_div(G.LP);
    G.Pdiv = make_fast_div(P);
    G.Mdiv = make_fast_div(M);
    G.nblkdiv = make_fast_div(G.nblk);
    G.thpdiv = make_fast_div(G.thp);
    if (bin_lds_bytes(G) > 64 * 1024) return false;
    return true;
}

inline size_t desc_bytes(const CellGeom &G) { return align256((size_t)G.heads * G.nblk * G.ntmax * 4) + 512; }  // + 8 ticket lines
inline size_t region_bytes(const CellGeom &G) { return align256((size_t)G.heads * G.nblk * G.slice * 16); }
inline size_t partial_bytes(const CellGeom &G, int D) { return align256((size_t)G.heads * G.prows_max * D * 4); }
inline size_t amax_bytes(const CellGeom &G) { return align256((size_t)G.heads * G.nblk * 8) + 256; }  // + the flag word
inline size_t accum_lds_bytes(const CellGeom &G, int D)
{
    size_t hdr = kLevelWords * kMaxLevels + 16 + 2 * (size_t)G.nblk + 1 + 2 * (size_t)G.heads;
    hdr = (hdr + 3) & ~(size_t)3;
    const size_t twm = (size_t)1 << G.twl_max;
    return (hdr + (G.thp + 2) * (twm + 2) * D) * 4 + (G.thp * twm + 8) * D * 8;   // (+ 8 trash rows)
}

template <int D, int LPG>
int launch_walk(const CellGeom &G, const float *grad_out, const float *value, const int64_t *shapes,
                const int64_t *start, const unsigned *desc, const uint4 *region, float *partial,
                unsigned *tickets, float *gv, float *gl, float *ga, hipStream_t st, const unsigned *only_if = nullptr)
{
    const unsigned NG = 64 / LPG, TW1 = (1u << G.twl_max) + 1;
    const size_t lds2 = (kLevelWords * kMaxLevels + 16 + 2 * (size_t)G.nblk + 1 + (size_t)NG * TW1 + TW1 + G.cap) * 4;
    if (lds2 > 64 * 1024) return (int)hipErrorInvalidValue;
    // one wave per block; the waves of an XCD stride over the work items of its heads
    // (behind the accumulate kernel the launch is normally a no-op: a small grid keeps it cheap, the ticket counters
    // make any grid size correct)
    const unsigned grid = only_if ? 512u : (G.split ? ZIRA_WALK_GRID_DENSE : ZIRA_WALK_GRID_SPARSE);
    hipLaunchKernelGGL((msda_bwd_walk<D, LPG>), dim3(grid), dim3(64), lds2, st, grad_out, value, shapes, start,
                       G, desc, region, partial, tickets, gv, gl, ga, only_if);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !G.split) return (int)e;
    hipLaunchKernelGGL(msda_bwd_fold<D>, dim3(1024), dim3(256), 0, st, shapes, start, G, partial, gv);
    return (int)hipGetLastError();
}

}  // namespace

namespace zira {

size_t cells_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P)
{
    CellGeom G;
    if (!make_geom(B, S, M, D, L, Q, P, G)) return 0;
    return desc_bytes(G) + region_bytes(G) + partial_bytes(G, D) + amax_bytes(G);
}

int cells_backward_f32(const float *grad_out, const float *value, const int64_t *shapes,
                       const int64_t *start, const float *loc, const float *attn, int B, int S,
                       int M, int D, int L, int Q, int P, float *gv, float *gl, float *ga,
                       void *ws, size_t ws_bytes, hipStream_t st)
{
    CellGeom G;
    if (!make_geom(B, S, M, D, L, Q, P, G) ||
        ws_bytes < desc_bytes(G) + region_bytes(G) + partial_bytes(G, D) + amax_bytes(G))
        return (int)hipErrorInvalidValue;
    unsigned *desc = reinterpret_cast<unsigned *>(ws);
    uint4 *region = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(ws) + desc_bytes(G));
    float *partial = reinterpret_cast<float *>(reinterpret_cast<char *>(region) + region_bytes(G));
    unsigned *tickets = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(ws) + desc_bytes(G) - 512);
    const bool accum = use_accum(B, M, D, Q);
    unsigned *amax = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(partial) + partial_bytes(G, D));
    unsigned *flag = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(amax) + amax_bytes(G) - 256);
    hipLaunchKernelGGL(msda_bwd_bin, dim3(G.heads * G.nblk), dim3(kBinThreads), bin_lds_bytes(G), st, loc, attn, shapes,
                       G, gl, ga, desc, region, tickets, grad_out, accum ? amax : (unsigned *)nullptr, gv);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (accum) {
        const size_t lds = accum_lds_bytes(G, D);
        if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
        static bool attr_set = false;
        if (!attr_set) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_bwd_accum<32, ZIRA_ACC_THREADS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        const unsigned per_cu = (unsigned)((160 * 1024) / lds), by_threads = 2048 / ZIRA_ACC_THREADS;
        const unsigned bpc = per_cu < by_threads ? per_cu : by_threads;
        hipLaunchKernelGGL((msda_bwd_accum<32, ZIRA_ACC_THREADS>), dim3(256 * (bpc ? bpc : 1)), dim3(ZIRA_ACC_THREADS), lds, st,
                           grad_out, value, shapes, start, G, desc, region, partial, tickets, amax, flag, gv, gl, ga);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        // non-finite grad_out / attention weights: the accumulate kernel raised the flag and did nothing
        return launch_walk<32, ZIRA_WALK_LPG32>(G, grad_out, value, shapes, start, desc, region, partial, tickets, gv, gl, ga, st, flag);
    }
    if (D == 16) return launch_walk<16, 4>(G, grad_out, value, shapes, start, desc, region, partial, tickets, gv, gl, ga, st);
    if (D == 32) return launch_walk<32, ZIRA_WALK_LPG32>(G, grad_out, value, shapes, start, desc, region, partial, tickets, gv, gl, ga, st);
    return launch_walk<64, 8>(G, grad_out, value, shapes, start, desc, region, partial, tickets, gv, gl, ga, st);
}


}  // namespace zira
