// lsap.hip -- batched rectangular linear sum assignment on the device (C ABI: zira_lsap_f32).
//
// The reference matches predictions to ground truth with scipy.optimize.linear_sum_assignment, once
// per image and prediction set, on a device->host copy of the cost matrix
// (groundingdino/models/GroundingDINO/matcher/matcher.py:105-151): 14 host round trips per step, each
// a full GPU drain.  Here every (set, image) problem is solved by one wavefront on the device, so the
// indices never leave HBM and the step has no host synchronisation left in the criterion.
//
// The algorithm is the one scipy uses (rectangular shortest-augmenting-path Jonker-Volgenant, Crouse
// 2016, scipy/optimize/rectangular_lsap), restated so that the RESULT IS IDENTICAL, ties included:
// float32 costs are widened to float64 as numpy does, tall matrices are transposed, the dual updates
// and reduced costs are formed in the same order, and the choice among equal shortest-path costs
// follows scipy's scan -- "the first column of the scan wins, unless a later one is still unassigned,
// then the last unassigned one wins" -- over the same `remaining` array (filled in reverse, entries
// removed by moving the last one into the hole).  The scan over the remaining columns is the only
// O(columns) part and runs on the 64 lanes (lane l takes positions l, l + 64, ...); its winner is
// combined across lanes with exactly that rule.  Everything else (path bookkeeping) is scalar.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "zira_msda.h"

namespace {

struct Pick {
    double val;   // lowest shortest-path cost seen
    int first;    // first scan position with that cost (INT_MAX: none)
    int lastu;    // last scan position with that cost whose column is unassigned (-1: none)
};

__device__ __forceinline__ Pick merge(const Pick &a, const Pick &b)
{
    if (a.val < b.val) return a;
    if (b.val < a.val) return b;
    Pick r;
    r.val = a.val;
    r.first = a.first < b.first ? a.first : b.first;
    r.lastu = a.lastu > b.lastu ? a.lastu : b.lastu;
    return r;
}

__device__ __forceinline__ Pick wave_merge(Pick p)
{
#pragma unroll
    for (int d = 32; d; d >>= 1) {
        Pick o;
        o.val = __shfl_xor(p.val, d);
        o.first = __shfl_xor(p.first, d);
        o.lastu = __shfl_xor(p.lastu, d);
        p = merge(p, o);
    }
    return p;
}

constexpr int kInfeasible = 1;

// One 64-thread block per problem (set s, image b).  meta = [toff_0 .. toff_B | moff_0 .. moff_B]:
// target-column offsets and match offsets (moff_{b+1} - moff_b = min(Q, n_b)).
__global__ __launch_bounds__(64) void lsap_kernel(const float *__restrict__ cost, int B, int Q, int Ttot,
                                                  const int *__restrict__ meta, int64_t *__restrict__ q_out,
                                                  int64_t *__restrict__ t_out, int Mtot, int t_global,
                                                  char *__restrict__ ws, size_t ws_stride, int use_lds,
                                                  int *__restrict__ status)
{
    extern __shared__ double lsap_lds[];
    const int prob = blockIdx.x, s = prob / B, b = prob - s * B;
    const int lane = threadIdx.x;
    const int toff = meta[b], n_b = meta[b + 1] - toff, moff = meta[B + 1 + b];
    const bool tr = n_b < Q;                       // scipy: a tall matrix (more rows than columns) is transposed
    const int nr = tr ? n_b : Q, nc = tr ? Q : n_b;
    if (nr == 0) return;
    const float *base = cost + ((size_t)(s * B + b) * Q) * Ttot + toff;
    const size_t stride_i = tr ? 1 : (size_t)Ttot, stride_j = tr ? (size_t)Ttot : 1;

    char *mem = use_lds ? reinterpret_cast<char *>(lsap_lds) : ws + (size_t)prob * ws_stride;
    double *sp = reinterpret_cast<double *>(mem);           // [nc] shortest path costs
    double *v = sp + nc;                                     // [nc]
    double *u = v + nc;                                      // [nr]
    int *path = reinterpret_cast<int *>(u + nr);             // [nc]
    int *row4col = path + nc;                                // [nc]
    int *remaining = row4col + nc;                           // [nc]
    int *SC = remaining + nc;                                // [nc]
    int *col4row = SC + nc;                                  // [nr]
    int *SR = col4row + nr;                                  // [nr]

    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
    __syncthreads();

    for (int cur = 0; cur < nr; ++cur) {
        // ---- shortest augmenting path from row `cur`
        for (int it = lane; it < nc; it += 64) { remaining[it] = nc - it - 1; SC[it] = 0; sp[it] = INFINITY; }
        for (int i = lane; i < nr; i += 64) SR[i] = 0;
        __syncthreads();
        double min_val = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            const float *ci = base + (size_t)i * stride_i;
            Pick p;
            p.val = INFINITY; p.first = 0x7fffffff; p.lastu = -1;
            for (int it = lane; it < num_remaining; it += 64) {
                const int j = remaining[it];
                const double r = ((min_val + (double)ci[(size_t)j * stride_j]) - ui) - v[j];
                double spj = sp[j];
                if (r < spj) { path[j] = i; sp[j] = r; spj = r; }
                const bool unassigned = row4col[j] == -1;
                if (spj < p.val) { p.val = spj; p.first = it; p.lastu = unassigned ? it : -1; }
                else if (spj == p.val && unassigned) p.lastu = it;
            }
            p = wave_merge(p);
            min_val = p.val;
            if (min_val == INFINITY) {  // infeasible cost matrix (inf / NaN entries): scipy raises
                // The flag tells the host (matcher.check()); the pairs still get VALID indices -- target k with
                // query k -- so that the gathers the criterion issues from them stay inside their tensors (the
                // outputs are torch.empty memory, and an out-of-range index is a device-side assert, i.e. an abort).
                if (lane == 0 && status) atomicOr(status, kInfeasible);
                int64_t *qf = q_out + (size_t)s * Mtot + moff, *tf = t_out + (size_t)s * Mtot + moff;
                for (int k = lane; k < nr; k += 64) { qf[k] = k; tf[k] = k + (t_global ? toff : 0); }
                return;
            }
            const int index = p.lastu >= 0 ? p.lastu : p.first;
            __syncthreads();           // every lane has read `remaining` / `sp` before they change
            const int j = remaining[index];
            const int r4c = row4col[j];
            if (r4c == -1) sink = j; else i = r4c;
            --num_remaining;
            const int moved = remaining[num_remaining];
            __syncthreads();
            if (lane == 0) { SC[j] = 1; remaining[index] = moved; }
            __syncthreads();
        }
        // ---- dual variables
        if (lane == 0) u[cur] += min_val;
        for (int r = lane; r < nr; r += 64)
            if (SR[r] && r != cur) u[r] += min_val - sp[col4row[r]];
        for (int j = lane; j < nc; j += 64)
            if (SC[j]) v[j] -= min_val - sp[j];
        __syncthreads();
        // ---- augment the previous solution along the path
        if (lane == 0) {
            int j = sink;
            for (;;) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r];
                col4row[r] = j;
                j = t;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }

    // ---- (row_ind, col_ind) as scipy returns them
    int64_t *qo = q_out + (size_t)s * Mtot + moff, *to = t_out + (size_t)s * Mtot + moff;
    const int tadd = t_global ? toff : 0;
    if (tr) {  // rows of the transposed problem are targets: pairs sorted by query index (argsort of col4row)
        for (int r = lane; r < nr; r += 64) {
            const int q = col4row[r];
            int rank = 0;
            for (int k = 0; k < nr; ++k) rank += col4row[k] < q;
            qo[rank] = q;
            to[rank] = r + tadd;
        }
    } else {
        for (int r = lane; r < nr; r += 64) { qo[r] = r; to[r] = col4row[r] + tadd; }
    }
}

// ---- matching cost ------------------------------------------------------------------------------
// cost[n, t] = w_bbox * L1(box_n, tbox_t) + w_class * (pos - neg)(sigmoid(logit[n, id_t])) + w_giou * (-GIoU)
// (reference matcher.py:105-141, util/box_ops.py:9-66), one thread per (prediction, target).  Every
// product and sum is rounded on its own, in the order the reference's chain of PyTorch kernels rounds
// them (no fma contraction), so that the assignments computed from it are the reference's.
__device__ __forceinline__ float mulr(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float addr(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float subr(float a, float b) { return __fsub_rn(a, b); }

__device__ __forceinline__ float divr(float a, float b) { return a / b; }   // IEEE divide (rcp-based forms are further from ATen's)

constexpr int kBadBoxes = 2;

__global__ __launch_bounds__(256) void match_cost_kernel(const float *__restrict__ logits, const float *__restrict__ boxes,
                                                         const int64_t *__restrict__ tgt_ids,
                                                         const float *__restrict__ tgt_boxes, int N, int C, int T,
                                                         float w_class, float w_bbox, float w_giou, float alpha,
                                                         float gamma, float *__restrict__ cost, int *__restrict__ status)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * T) return;
    const int n = idx / T, t = idx - n * T;
    // classification: focal-style cost on the target's class probability
    const float x = logits[(size_t)n * C + tgt_ids[t]];
    const float p = divr(1.0f, addr(1.0f, expf(-x)));
    const float q = subr(1.0f, p);
    const float pg = gamma == 2.0f ? mulr(p, p) : powf(p, gamma);
    const float qg = gamma == 2.0f ? mulr(q, q) : powf(q, gamma);
    const float neg = mulr(mulr(subr(1.0f, alpha), pg), -logf(addr(q, 1e-8f)));
    const float pos = mulr(mulr(alpha, qg), -logf(addr(p, 1e-8f)));
    const float cost_class = subr(pos, neg);
    // boxes: L1 on (cx, cy, w, h); GIoU on the corners
    const float4 a = reinterpret_cast<const float4 *>(boxes)[n], b = reinterpret_cast<const float4 *>(tgt_boxes)[t];
    const float d0 = fabsf(subr(a.x, b.x)), d1 = fabsf(subr(a.y, b.y)), d2 = fabsf(subr(a.z, b.z)), d3 = fabsf(subr(a.w, b.w));
    const float cost_bbox = addr(addr(d0, d2), addr(d1, d3));    // the pairing of torch.cdist's shuffle reduction
    const float ax0 = subr(a.x, mulr(0.5f, a.z)), ay0 = subr(a.y, mulr(0.5f, a.w));
    const float ax1 = addr(a.x, mulr(0.5f, a.z)), ay1 = addr(a.y, mulr(0.5f, a.w));
    const float bx0 = subr(b.x, mulr(0.5f, b.z)), by0 = subr(b.y, mulr(0.5f, b.w));
    const float bx1 = addr(b.x, mulr(0.5f, b.z)), by1 = addr(b.y, mulr(0.5f, b.w));
    if (status && (!(ax1 >= ax0) || !(ay1 >= ay0) || !(bx1 >= bx0) || !(by1 >= by0))) atomicOr(status, kBadBoxes);
    const float area_a = mulr(subr(ax1, ax0), subr(ay1, ay0)), area_b = mulr(subr(bx1, bx0), subr(by1, by0));
    const float iw = fmaxf(subr(fminf(ax1, bx1), fmaxf(ax0, bx0)), 0.0f);
    const float ih = fmaxf(subr(fminf(ay1, by1), fmaxf(ay0, by0)), 0.0f);
    const float inter = mulr(iw, ih);
    const float uni = subr(addr(area_a, area_b), inter);
    const float iou = divr(inter, addr(uni, 1e-6f));
    const float ew = fmaxf(subr(fmaxf(ax1, bx1), fminf(ax0, bx0)), 0.0f);
    const float eh = fmaxf(subr(fmaxf(ay1, by1), fminf(ay0, by0)), 0.0f);
    const float earea = mulr(ew, eh);
    const float giou = subr(iou, divr(subr(earea, uni), addr(earea, 1e-6f)));
    cost[idx] = addr(addr(mulr(w_bbox, cost_bbox), mulr(w_class, cost_class)), mulr(w_giou, -giou));
}

inline size_t problem_bytes(int Q, int Tmax)
{
    const size_t nc = (size_t)(Q > Tmax ? Q : Tmax), nr = (size_t)(Q > Tmax ? Tmax : Q);
    // nr = min(Q, n_b) <= min(Q, Tmax) and nc = max(Q, n_b) <= max(Q, Tmax) for every problem of the call
    return ((nc * (8 + 8 + 4 * 4) + nr * (8 + 4 + 4)) + 15) & ~(size_t)15;
}

}  // namespace

extern "C" {

size_t zira_lsap_workspace_bytes(int nsets, int B, int Q, int Tmax)
{
    if (nsets <= 0 || B <= 0 || Q <= 0 || Tmax < 0) return 0;
    const size_t per = problem_bytes(Q, Tmax);
    return per > 64 * 1024 ? per * (size_t)nsets * B : 0;   // the per-problem state lives in LDS when it fits
}

int zira_lsap_f32(const float *cost, int nsets, int B, int Q, int Ttot, int Tmax, const int32_t *meta,
                  int64_t *q_idx, int64_t *t_idx, int Mtot, int t_global, int32_t *status, void *workspace,
                  size_t workspace_bytes, void *stream)
{
    if (nsets <= 0 || B <= 0 || Q <= 0 || Ttot < 0 || Tmax < 0 || Tmax > Ttot || Mtot < 0) return ZIRA_MSDA_EINVAL;
    if (Ttot == 0 || Mtot == 0) return 0;  // nothing to match
    if (!cost || !meta || !q_idx || !t_idx) return ZIRA_MSDA_EINVAL;
    const size_t need = zira_lsap_workspace_bytes(nsets, B, Q, Tmax);
    if (need && (!workspace || workspace_bytes < need)) return ZIRA_MSDA_EINVAL;
    const size_t per = problem_bytes(Q, Tmax);
    const int use_lds = per <= 64 * 1024;
    hipLaunchKernelGGL(lsap_kernel, dim3(nsets * B), dim3(64), use_lds ? per : 0, (hipStream_t)stream, cost, B, Q,
                       Ttot, meta, q_idx, t_idx, Mtot, t_global, reinterpret_cast<char *>(workspace), per, use_lds,
                       status);
    return (int)hipGetLastError();
}

int zira_match_cost_f32(const float *logits, const float *boxes, const int64_t *tgt_ids, const float *tgt_boxes,
                        int N, int C, int T, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                        float *cost, int32_t *status, void *stream)
{
    if (N < 0 || C <= 0 || T < 0 || (long long)N * T >= (1ll << 31)) return ZIRA_MSDA_EINVAL;
    if (N == 0 || T == 0) return 0;
    if (!logits || !boxes || !tgt_ids || !tgt_boxes || !cost) return ZIRA_MSDA_EINVAL;
    const int total = N * T;
    hipLaunchKernelGGL(match_cost_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, boxes,
                       tgt_ids, tgt_boxes, N, C, T, w_class, w_bbox, w_giou, alpha, gamma, cost, status);
    return (int)hipGetLastError();
}

}  // extern "C"
