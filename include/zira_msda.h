/*
 * zira_msda.h -- C ABI of the MI355X-native multi-scale deformable attention (MSDA) op.
 *
 * This is the drop-in boundary for the reference's native extension `groundingdino._C`
 * (pybind module, reference csrc/vision.cpp:53-56), i.e. for
 *
 *   at::Tensor  ms_deform_attn_forward (value, spatial_shapes, level_start_index,
 *                                       sampling_loc, attn_weight, im2col_step)
 *   vector<Tensor> ms_deform_attn_backward(value, spatial_shapes, level_start_index,
 *                                       sampling_loc, attn_weight, grad_output, im2col_step)
 *
 * declared at reference csrc/MsDeformAttn/ms_deform_attn.h:21-61 and implemented (CUDA
 * only) at csrc/MsDeformAttn/ms_deform_attn_cuda.cu:21-81 / :84-154 on top of the kernels in
 * csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh.  The reference binds those two functions
 * through libtorch; this library exposes the same two operations with plain pointers and
 * sizes so that any host (ctypes, pybind, cgo ...) can bind it -- see INTEGRATION.md.
 *
 * Contract (all entry points):
 *   - every pointer is a DEVICE pointer (hipMalloc / torch ROCm tensor storage), including
 *     the two int64 arrays -- exactly as the reference requires (ms_deform_attn_cuda.cu:35-39);
 *   - tensors are dense row-major ("contiguous"):
 *       value[B,S,M,D]  spatial_shapes[L,2]=(H,W)  level_start_index[L]
 *       sampling_loc[B,Q,M,L,P,2] (x, y normalised to [0,1])   attn_weight[B,Q,M,L,P]
 *       out / grad_out [B,Q,M*D]
 *   - buffers are caller-owned; outputs need NOT be zeroed by the caller: the forward
 *     overwrites `out`, the backward zero-fills `grad_value` itself (asynchronously, on
 *     `stream`) and overwrites `grad_sampling_loc` / `grad_attn_weight`;
 *   - `stream` is a hipStream_t (NULL = the default stream); work is enqueued, never
 *     synchronised: no host sync, no allocation, safe to capture in a hipGraph;
 *   - re-entrant and stateless (thread-safe for distinct output buffers);
 *   - return value: 0 (hipSuccess) or a hipError_t cast to int; never throws, never prints
 *     (the reference only printf()s launch errors, cuh:948-952 -- we return them);
 *     ZIRA_MSDA_EINVAL marks argument errors detected before any launch;
 *   - the reference's `im2col_step` only chunks the batch into several launches
 *     (ms_deform_attn_cuda.cu:51-76); it does not change results.  This ABI takes the whole
 *     batch in one launch; the Python binding keeps the argument and its divisibility
 *     check for signature compatibility.
 *   - sizes: per-batch-element element counts (S*M*D and Q*M*L*P*2) must be < 2^31
 *     (the reference uses 32-bit indices throughout, cuh:255-269).
 */
#ifndef ZIRA_MSDA_H_
#define ZIRA_MSDA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZIRA_MSDA_EINVAL 1 /* == hipErrorInvalidValue */

/* Replaces ms_deform_attn_forward for float32 (ms_deform_attn.h:21-40). */
int zira_msda_fwd_f32(const float *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const float *sampling_loc,
                      const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                      float *out, void *stream);

/* Replaces ms_deform_attn_backward for float32 (ms_deform_attn.h:42-61). */
int zira_msda_bwd_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const float *sampling_loc,
                      const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                      float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                      void *stream);

/* Workspace float32 backward: same results up to summation order, 2-6x faster than the plain entry point (no fp32
 * atomics on the common path).  Dense calls (B*M*Q >= 65536: encoder self-attention, every pixel a query) take the cell
 * kernels -- samples binned by the 2x2 pixel block they touch into tiles of 15 x 8 pixels; for D = 32 a tile's four
 * corner rows per sample are summed in LDS in 64-bit fixed point (scale per (image, head) from max|grad_out| *
 * max|attn| of that head: the sums are exact and grad_value is identical from run to run; non-finite gradients fall
 * back to a float walk of the same tiles), for D = 16 / 64 the tiles are walked with the window's accumulators in
 * registers.  Sparse calls (decoder cross-attention) with D = 32 take "plan + tile accumulate" (csrc/msda_tiles.hip,
 * see the planned entry points below): without a plan from the forward pass this call plans first, the workspace being
 * the plan buffer.  Other sparse calls keep the round-2 entry sort (per-block counting sort of corner contributions,
 * per-tile row sums).
 * `zira_msda_bwd_workspace_bytes` returns the scratch size it needs for these dimensions (75 MB at
 * B=2,S=22223,M=8,D=32,L=4,Q=900,P=4 -- sized for the worst case, about a fifth is touched; 496 MB at Q=S), or 0 when
 * no workspace path applies (the plain entry point is then the only one).  The workspace is caller-owned DEVICE
 * memory, 16-byte aligned, needs no initialisation and may be reused by later calls on the same stream; with
 * workspace == NULL or too small the call degrades to zira_msda_bwd_f32.  Every element of grad_value,
 * grad_sampling_loc and grad_attn_weight is written by the call, none needs pre-zeroing.
 * Extra precondition: the levels tile [0, S) exactly (level_start_index[l] + H_l*W_l == level_start_index[l+1], last
 * one == S), which the reference module asserts (ms_deform_attn.py:284).  The tables live on the device, so the call
 * cannot check them up front: when the dense kernels find more tiles than S allows they refuse -- and zero-fill all
 * three gradients instead of leaving them uninitialised. */
size_t zira_msda_bwd_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P);

int zira_msda_bwd_f32_ws(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                         const int64_t *level_start_index, const float *sampling_loc,
                         const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                         float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Planned backward for sparse calls (decoder cross-attention: B*M*Q < 65536, D = 32).  How the backward's work is cut
 * -- which samples add to which 16 x 8-pixel tile of grad_value, which tiles are split, which CU takes which tile --
 * depends on spatial_shapes / level_start_index / sampling_loc only, all of which exist in the FORWARD pass
 * (reference ms_deform_attn.py:50: the autograd Function saves them there), as does the attention weight that the plan
 * writes into every sample's record.  So the planning is a call of its own:
 *
 *   zira_msda_plan_bytes      size of the plan buffer for these dimensions on the current device (0: no planned path;
 *                             use zira_msda_bwd_f32_ws)
 *   zira_msda_fwd_plan_f32    the forward AND the plan in ONE launch: the first 64 workgroups plan a (head, level) unit
 *                             each, the others run the gather (14.7 us for both at the north-star shape against 9.9 +
 *                             9.8 us apart: two kernels on two streams do not overlap on this stack, one grid does)
 *   zira_msda_plan_f32        the plan alone: reads the level tables, sampling_loc and attn_weight
 *   zira_msda_bwd_planned_f32 the backward from a plan: ONE launch whose persistent workgroups sum grad_value tile by
 *                             tile in LDS (the only load that depends on a record is the query's grad_out row) while
 *                             forward-style waves -- one per (b, q, m) -- gather the value rows again and form
 *                             grad_sampling_loc / grad_attn_weight.  The shares of a split tile meet INSIDE that launch
 *                             (since round 5; there is no fold launch): each share writes its partial tile through to
 *                             memory and draws a ticket from the tile's agent-scope counter, and the share that arrives
 *                             last adds all of them in share order.  `plan` must come from zira_msda_*plan_f32 for the
 *                             same dimensions, level tables, sampling_loc AND attn_weight.  Same outputs / contract as
 *                             zira_msda_bwd_f32_ws; deterministic apart from the order of the LDS sums in double (no fp32
 *                             atomics anywhere).
 * The plan buffer needs no initialisation and 16-byte alignment; the backward uses regions of it as scratch (the partial
 * tiles and their tickets: hence `plan` is not const there).  A plan is NOT REENTRANT: it may serve several backward calls one
 * after the other on a stream, never two at the same time (they would race on the tickets and the partial tiles).  Limits of the planned path (zira_msda_plan_bytes returns 0 beyond them and the callers take
 * zira_msda_bwd_f32_ws): D = 32, L <= 16, and the plan kernel's LDS tables -- S / 8 + L tile counters beside 64 KB of
 * per-thread ranks -- within 78 KB, i.e. S up to about 27 000 value rows per image (the decoder shape has 22 223).  Where the
 * runtime refuses the kernels' dynamic-LDS opt-in, zira_msda_fwd_plan_f32 falls back to zira_msda_fwd_f32 +
 * zira_msda_plan_f32 and those return ZIRA_MSDA_EINVAL rather than a launch error. */
size_t zira_msda_plan_bytes(int B, int S, int M, int D, int L, int Q, int P);

int zira_msda_plan_f32(const int64_t *spatial_shapes, const int64_t *level_start_index, const float *sampling_loc,
                       const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P, void *plan,
                       size_t plan_bytes, void *stream);

int zira_msda_fwd_plan_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start_index,
                           const float *sampling_loc, const float *attn_weight, int B, int S, int M, int D, int L,
                           int Q, int P, float *out, void *plan, size_t plan_bytes, void *stream);

int zira_msda_bwd_planned_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                              const int64_t *level_start_index, const float *sampling_loc,
                              const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                              float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                              void *plan, size_t plan_bytes, void *stream);

/* float64 twins: the reference dispatches AT_DISPATCH_FLOATING_TYPES = {float, double}
 * (ms_deform_attn_cuda.cu:65, :135). */
int zira_msda_fwd_f64(const double *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const double *sampling_loc,
                      const double *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                      double *out, void *stream);

int zira_msda_bwd_f64(const double *grad_out, const double *value, const int64_t *spatial_shapes,
                      const int64_t *level_start_index, const double *sampling_loc,
                      const double *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                      double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                      void *stream);

/* Host-memory twins (float32; no stream, returns when done): where the reference's CPU entry points raise
 * "Not implemented on the CPU" (csrc/MsDeformAttn/ms_deform_attn_cpu.cpp:17-41) a binding can offer the op on host
 * tensors instead.  All pointers are HOST pointers; same layouts and semantics as the device entry points; a few
 * std::threads over the (batch, head) pairs.  Product code (csrc/msda_cpu.cpp), independent of the test oracle. */
int zira_msda_fwd_cpu_f32(const float *value, const int64_t *spatial_shapes,
                          const int64_t *level_start_index, const float *sampling_loc,
                          const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                          float *out);

int zira_msda_bwd_cpu_f32(const float *grad_out, const float *value, const int64_t *spatial_shapes,
                          const int64_t *level_start_index, const float *sampling_loc,
                          const float *attn_weight, int B, int S, int M, int D, int L, int Q, int P,
                          float *grad_value, float *grad_sampling_loc, float *grad_attn_weight);

/* ---- small fp32 attention of the cross-modal decoder ------------------------------------------
 * out = softmax(q k^T * scale + key_mask) v per (batch, head), head width d = 32: what nn.MultiheadAttention computes
 * between its projections for the decoder's self-attention over the queries and its cross-attention to the text tokens
 * (reference transformer_for_adapter.py:1029-1054).  q [L, B, H*32], k / v [S, B, H*32] with row strides ldq / ldk / ldv
 * (floats between consecutive (l, b) rows: slices of fused projections are used in place; 16-byte aligned, strides
 * multiples of 4); key_mask: optional additive [B, S] (0 or -inf); out [L, B, H*32] contiguous; lse [B, H, L] (the
 * row-wise log-sum-exp the backward recomputes the probabilities from).  One launch, scores never leave registers
 * (v_mfma_f32_32x32x2_f32: exact fp32 products).  A query whose keys are all masked gets a zero row (torch gives NaN). */
int zira_attn_fwd_f32(const float *q, const float *k, const float *v, const float *key_mask, int L, int S, int B, int H,
                      int d, int ldq, int ldk, int ldv, float scale, float *out, float *lse, void *stream);

/* Gradients of the above: dq [L, B, H*32], dk / dv [S, B, H*32] contiguous, every element written.  `scratch`
 * (16-byte aligned, zira_attn_bwd_scratch_floats() floats; B*H*L is the minimum) holds <dout, out> per query and, when
 * there are few key blocks, the partial dk / dv sums of the shares of the query range, which a third launch adds up in
 * a fixed order (no atomics: the result does not depend on the run). */
size_t zira_attn_bwd_scratch_floats(int L, int S, int B, int H);
int zira_attn_bwd_f32(const float *q, const float *k, const float *v, const float *key_mask, const float *out,
                      const float *dout, const float *lse, int L, int S, int B, int H, int d, int ldq, int ldk, int ldv,
                      float scale, float *dq, float *dk, float *dv, float *scratch, size_t scratch_floats, void *stream);

/* The same with row strides lddq / lddk / lddv (floats between consecutive (l, b) rows; multiples of 4, >= H*32) for the
 * gradients: dq, dk and dv may be column slices of one [rows, B, 3*H*32] buffer, which the input gradient of a packed
 * in-projection then reads as ONE operand (the decoder's self-attention: q, k, v come from one GEMM and go back through one). */
int zira_attn_bwd_ld_f32(const float *q, const float *k, const float *v, const float *key_mask, const float *out,
                         const float *dout, const float *lse, int L, int S, int B, int H, int d, int ldq, int ldk, int ldv,
                         float scale, float *dq, float *dk, float *dv, int lddq, int lddk, int lddv, float *scratch,
                         size_t scratch_floats, void *stream);

/* ---- ZiRa reparameterizable side branch (RSB): fused epilogue ---------------------------
 * Replaces the elementwise / reduction tail of RepZeroConv2d.forward and
 * RepZeroLinear.forward in training mode (reference
 * groundingdino_dual_zero_rep_branch.py:87-96 and :119-128):
 *     out  = scaling * y_branch + y_twin
 *     loss = mean(smooth_l1(scaling * y_branch, 0)) + mean(smooth_l1(out, 0))     (beta = 1)
 * y_branch = F(x; W, b) and y_twin = F(x; W_f, b_f) are produced by the caller's GEMM /
 * convolution library.  All pointers are device pointers to n contiguous floats (16-byte
 * aligned), `scaling` / `loss` / `g_scaling` / `grad_loss` point to one float on the device,
 * `workspace` to zira_rsb_workspace_floats(n) floats of scratch (no initialisation needed).
 * The backward returns d/dy_branch, d/dy_twin and d/dscaling given grad_out (d/dout, may be
 * NULL = zeros) and grad_loss (d/dloss, may be NULL = 0).  Deterministic reductions. */
size_t zira_rsb_workspace_floats(size_t n);

int zira_rsb_fwd_f32(const float *y_branch, const float *y_twin, const float *scaling, size_t n,
                     float *out, float *loss, float *workspace, void *stream);

int zira_rsb_bwd_f32(const float *y_branch, const float *y_twin, const float *scaling,
                     const float *grad_out, const float *grad_loss, size_t n, float *g_branch,
                     float *g_twin, float *g_scaling, float *workspace, void *stream);

/* ---- tall-reduction product for the cross-modal fusion layers -------------------------------
 * out[z] = X[z]^T * Y[z],  z < B;  X[z] is N x a (row-major; with x_transposed != 0 it is
 * stored a x N), Y[z] is N x b, out[z] is a x b;  a, b multiples of 4.  N is the number of image
 * tokens (tens of thousands), a and b are a few hundred at most: the three products of this
 * shape in BiMultiHeadAttention (reference fuse_modules.py:188-222 after re-bracketing around the
 * text side) -- a GEMM library runs them on one or two tiles.
 * `workspace`: zira_xty_workspace_floats(B, N, a, b) floats of device scratch, no initialisation
 * needed.  Deterministic (partial tiles folded in a fixed order). */
size_t zira_xty_workspace_floats(int B, int N, int a, int b);

int zira_xty_f32(const float *X, const float *Y, int B, int N, int a, int b, int x_transposed,
                 float *out, float *workspace, void *stream);

/* ---- fused score post-processing of the bi-directional attention ---------------------------
 * (reference fuse_modules.py:165-200: global-max shift, clamps, softmax over the text tokens,
 * column-max shift, clamps, softmax over the image tokens -- ~300 PyTorch kernels per layer and
 * direction).  Layout [B, N, H*T] throughout (N image tokens, H heads, T text tokens).
 *   x = xm + c;  x1 = clamp(x - max(x));  pv = softmax_t(x1 | mask_l);
 *   e = exp(clamp(x1 - max_n x1)) (0 where mask_v);  colsum = sum_n e      (p_l = e / colsum)
 * mask_l [B,T] / mask_v [B,N]: bytes, non-zero = padded, may be NULL.  colmax [B,H*T] and gmax [1]
 * are outputs of the forward that the backward needs again.  The backward takes the gradients
 * w.r.t. pv, e and colsum and returns those w.r.t. xm and c; it assumes e and colsum are only
 * used as e / colsum (then the paths through the two maxima vanish).  H*T <= 4096.
 * workspace: zira_bisoftmax_workspace_floats(B, N, H, T) floats, no initialisation needed. */
size_t zira_bisoftmax_workspace_floats(int B, int N, int H, int T);

int zira_bisoftmax_fwd_f32(const float *xm, const float *c, const uint8_t *mask_l, const uint8_t *mask_v,
                           int B, int N, int H, int T, int stable, int clamp_lo, int clamp_hi, float *pv,
                           float *e, float *colsum, float *colmax, float *gmax, float *workspace,
                           void *stream);

int zira_bisoftmax_bwd_f32(const float *xm, const float *c, const uint8_t *mask_l, int B, int N, int H, int T,
                           int stable, int clamp_lo, int clamp_hi, const float *pv, const float *e,
                           const float *colmax, const float *gmax, const float *g_pv, const float *g_e,
                           const float *g_colsum, float *g_xm, float *g_c, float *workspace, void *stream);

/* ---- Row LayerNorm forward (csrc/layernorm.hip) --------------------------------------------
 * y[r, :] = (x[r, :] - mean_r) * rstd_r * gamma + beta over the last dimension C, the nn.LayerNorm
 * of the encoder layers (reference transformer.py:810-811 norm1 / norm2, fuse_modules.py:262
 * layer_norm_v) and of the Swin-T blocks (swin_transformer.py:215, 258, 304), for row-major fp32
 * x [rows, C] with C % 4 == 0, C <= 1024 and 16-byte aligned pointers.  gamma / beta may be NULL
 * (1 / 0); mean / rstd [rows] are the statistics aten::native_layer_norm returns (either may be
 * NULL), so the backward can stay with aten::native_layer_norm_backward.  Two-pass variance. */
int zira_layernorm_fwd_f32(const float *x, const float *gamma, const float *beta, int64_t rows, int C,
                           float eps, float *y, float *mean, float *rstd, void *stream);

/* Input gradient of the above: dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)) with g = dy * gamma,
 * xhat = (x - mean) * rstd; mean / rstd as returned by the forward.  (Parameter gradients: ATen.) */
int zira_layernorm_bwd_f32(const float *dy, const float *x, const float *gamma, const float *mean,
                           const float *rstd, int64_t rows, int C, float *dx, void *stream);

/* y = LayerNorm(x + res), the residual connection in front of a post-LN layer in one pass: the sum (same rounding as a
 * separate add) is also written to sum_out [rows, C], which zira_layernorm_bwd_f32 takes as its x; the gradient it
 * returns belongs to x and res alike.  Other arguments as zira_layernorm_fwd_f32. */
int zira_add_layernorm_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, int64_t rows,
                               int C, float eps, float *sum_out, float *y, float *mean, float *rstd, void *stream);

/* ---- Batched linear sum assignment (Hungarian matching) -------------------------------------
 * Replaces the per-image scipy.optimize.linear_sum_assignment calls of
 * groundingdino/models/GroundingDINO/matcher/matcher.py:105-151 (HungarianMatcher.forward: `C.cpu()`
 * followed by one scipy call per image), for `nsets` prediction sets at once.
 *   cost   [nsets, B, Q, Ttot] float32, device: image b's targets are columns meta[b] .. meta[b+1]-1
 *   meta   [2 * (B + 1)] int32, device: target-column offsets, then match offsets
 *          (moff[b+1] - moff[b] = min(Q, n_b)); Tmax = largest n_b, Mtot = moff[B]
 *   q_idx, t_idx [nsets, Mtot] int64, device: scipy's (row_ind, col_ind) of image b at
 *          [moff[b], moff[b+1]); t_global != 0 adds meta[b] to col_ind (index into the concatenated targets)
 *   status optional int32 on the device, OR-ed with 1 when a cost matrix is infeasible (inf / NaN:
 *          scipy raises ValueError there); never cleared by the call
 * Same assignment as scipy for the same float32 costs, ties included.  Nothing is copied to the host.
 * zira_lsap_workspace_bytes() is 0 while the per-problem state fits in LDS (max(Q, Tmax) <~ 2000). */
size_t zira_lsap_workspace_bytes(int nsets, int B, int Q, int Tmax);

/* The matching cost in front of it (matcher.py:105-141 with cost_class_type "focal_loss_cost", box_ops.py:39-66):
 *   cost[n, t] = w_bbox * |box_n - tbox_t|_1 + w_class * (pos - neg) + w_giou * (-GIoU(box_n, tbox_t))
 * logits [N, C], boxes [N, 4] and tgt_boxes [T, 4] (cx, cy, w, h), tgt_ids [T] int64 -> cost [N, T] float32, all on
 * the device; every product and sum rounded separately, in the order of the reference's chain of PyTorch kernels.
 * status (optional, device int32) is OR-ed with 2 if a box has x1 < x0 or y1 < y0 (the reference asserts there). */
int zira_match_cost_f32(const float *logits, const float *boxes, const int64_t *tgt_ids, const float *tgt_boxes,
                        int N, int C, int T, float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                        float *cost, int32_t *status, void *stream);

int zira_lsap_f32(const float *cost, int nsets, int B, int Q, int Ttot, int Tmax, const int32_t *meta,
                  int64_t *q_idx, int64_t *t_idx, int Mtot, int t_global, int32_t *status, void *workspace,
                  size_t workspace_bytes, void *stream);


/* ---- Token logits -> category logits ---------------------------------------------------------
 * Replaces recover_to_cls_logits (groundingdino/models/GroundingDINO/utils.py:312-320):
 *   out[r, c] = max over the tokens t < n_tok[b] with cat_token_mask[b, c, t] of logits[r, t]   for c < n_cat[b],
 *   for_fill elsewhere (also for a category without tokens or with -inf logits only);  b = (r / Q) % B.
 * logits / out / grad_* [rows, T] float32 with rows a multiple of B * Q (leading dims = stacked prediction
 * sets); cat_token_mask [B, Cmax, Tmax] uint8 (the per-image category x token masks of
 * bertwarper.generate_masks_with_special_tokens_and_transfer_map, zero-padded); n_cat, n_tok [B] int32;
 * argmax [rows, Cmax] int32 is written by the forward (first arg-max token, -1 = none) and read by the backward,
 * which sends each category's gradient to that token.  Everything on the device. */
int zira_cat_logits_fwd_f32(const float *logits, const uint8_t *cat_token_mask, const int32_t *n_cat,
                            const int32_t *n_tok, long long rows, int B, int Q, int T, int Cmax, int Tmax,
                            float for_fill, float *out, int32_t *argmax, void *stream);

int zira_cat_logits_bwd_f32(const float *grad_out, const int32_t *argmax, const int32_t *n_cat, long long rows, int B,
                            int Q, int T, int Cmax, float *grad_logits, void *stream);


/* ---- (Shifted-)window attention of the frozen Swin backbone, forward only ---------------------
 * Replaces WindowAttention.forward together with the pad / roll / window_partition / window_reverse / crop around
 * it in SwinTransformerBlock.forward (backbone/swin_transformer.py:128-160, :222-270):
 *   qkv      [B, H, W, 3, heads, 32]  output of the block's qkv projection on the un-partitioned token map
 *   qkv_bias [3 * heads * 32]         that projection's bias (= q, k, v of a padding token)
 *   bias_t   [heads, N, N]            relative position bias, transposed: bias_t[h, j, i] = table[index[i, j], h]
 *   out      [B, H, W, heads * 32]    attention output in token order (input of the block's proj)
 * window = window size (N = window^2 tokens), shift = 0 or window / 2, scale = 32^-0.5.  Device pointers. */
int zira_window_attn_f32(const float *qkv, const float *qkv_bias, const float *bias_t, int B, int H, int W, int heads,
                         int head_dim, int window, int shift, float scale, float *out, void *stream);

/* The same for bfloat16 qkv / out (qkv_bias and bias_t stay float32), 12 x 12 windows only: the 384-pixel Swin-B / L
 * variants of BASELINE configs[3], whose qkv projection runs under bf16 autocast (swin_transformer.py:775-788).  A
 * padding token's q / k / v are the bias rounded to bfloat16, as the projection would leave them; scores, softmax and
 * the weighted sum are formed in float32 on the matrix cores (v_mfma_f32_32x32x2_f32), one block per (window, head). */
int zira_window_attn_bf16(const void *qkv, const float *qkv_bias, const float *bias_t, int B, int H, int W, int heads,
                          int head_dim, int window, int shift, float scale, void *out, void *stream);


/* ---- MSDA module: attention softmax + sampling locations, forward and backward ---------------------
 * zira_msda_sampling_{fwd,bwd}_f32 replace what MultiScaleDeformableAttention.forward does between its query projections
 * and the native op (groundingdino/models/GroundingDINO/ms_deform_attn.py:290-325): softmax of the attention logits
 * over the L*P samples of a (query, head), and sampling_locations = reference point + offset / (W_l, H_l)  (R = 2) or
 * reference box centre + offset / P * box size * 0.5  (R = 4), each operation rounded as in the reference.
 * proj [N, ld] holds, per query row, M*L*P*2 offsets followed by M*L*P logits (ld >= 3*M*L*P, even); ref [N, L, R];
 * shapes int64 [L, 2] on the device; loc [N, M, L, P, 2], attn [N, M, L, P] contiguous.  L*P must be a power of two
 * <= 64.  The backward writes both parts of every row of grad_proj (reference points get no gradient). */
int zira_msda_sampling_fwd_f32(const float *proj, int ld, const float *ref, int R, const int64_t *shapes, long long N, int M,
                               int L, int P, float *loc, float *attn, void *stream);
int zira_msda_sampling_bwd_f32(const float *grad_loc, const float *grad_attn, const float *attn, const float *ref, int R,
                               const int64_t *shapes, long long N, int M, int L, int P, float *grad_proj, int ld, void *stream);

/* ---- Row-block GEMM of the decoder's queries with its prologues and epilogues (csrc/rowgemm.hip) ----------------
 * C[M, N] = epilogue( prologue(A)[M, K] * op(W) ) in float32 on v_mfma_f32_16x16x4_f32 (exact fp32 products), 16 rows per
 * workgroup: the nn.Linear calls of a decoder layer on its B x 900 query rows (reference transformer_for_adapter.py:
 * 1001-1006 FFN, :1029-1071 attention in / out projections; ms_deform_attn.py:262-288, :338 the MSDA module's) together
 * with the position-code add in front of them and the residual add + LayerNorm behind them, and the same for their input
 * gradients.  All pointers are device pointers; every field not used stays 0 / NULL.
 *   a, lda            A [M, K], row stride lda floats (rows in batch-first order when a_batch_first, see below)
 *   pos, ldpos        optional [M, K]: the block's A is a + pos for output columns < pos_cols (a multiple of 128), a beyond
 *   w, ldw, w_is_nk   w_is_nk = 1: W [N, K] and C = A W^T (forward of nn.Linear with its weight as stored);
 *                     w_is_nk = 0: W [K, N] and C = A W   (the input gradient, again with the weight as stored)
 *   bias              optional [N]
 *   res, ldres        optional [M, N] added to the product (residual connection; or beta = 1 accumulation of a gradient)
 *   mask              optional [M, N] contiguous: C = 0 where mask <= 0 (ReLU gradient, mask = the ReLU's output)
 *   relu              C = max(C, 0)
 *   ln_gamma, ln_beta, ln_eps, ln_sum, ln_mean, ln_rstd
 *                     LayerNorm epilogue, taken when ln_gamma or ln_sum is set; needs N == 256.  c receives the normalised
 *                     rows; ln_sum [M, N] (optional) the rows before normalisation, ln_mean / ln_rstd [M] (optional) the
 *                     statistics: what the backward needs.  Two-pass variance.
 *   lnb_x, lnb_gamma, lnb_mean, lnb_rstd, lnb_dx
 *                     LayerNorm-backward prologue, taken when lnb_x is set (K == 256): a is dy [M, K], the operand of the
 *                     product is dx = rstd (g - mean_c g - xhat mean_c(g xhat)), g = dy gamma, and dx is also written to
 *                     lnb_dx [M, K] (optional): the gradient of the residual connection in front of the LayerNorm.
 *   batch, a_batch_first, c_batch_first
 *                     logical row r = q * batch + b ([queries, batch, C] tensors, as the decoder holds them); an operand marked
 *                     batch-first lives at row b * (M / batch) + q ([batch, queries, C]: the MSDA op's side).  pos, res, mask
 *                     and the LayerNorm arrays are always in logical row order.
 * Shapes: K a multiple of 128 and <= 2048; N a multiple of 128; pointers 16-byte aligned, strides multiples of 4.
 * Returns 0, -1 (NULL pointer), -2 (bad sizes), -3 (unsupported shape / alignment), -4 (launch failure). */
typedef struct zira_rowgemm_args {
    const float *a; int lda;
    const float *pos; int ldpos; int pos_cols;
    const float *w; int ldw; int w_is_nk;
    const float *bias;
    const float *res; int ldres;
    const float *mask;
    int relu;
    const float *ln_gamma; const float *ln_beta; float ln_eps; float *ln_sum; float *ln_mean; float *ln_rstd;
    const float *lnb_x; const float *lnb_gamma; const float *lnb_mean; const float *lnb_rstd; float *lnb_dx;
    float *c; int ldc;
    int m, n, k;
    int batch, a_batch_first, c_batch_first;
} zira_rowgemm_args;

int zira_rowgemm_f32(const zira_rowgemm_args *args, void *stream);

/* ---- Frozen FFN backward: GEMM with the ReLU gradient in its epilogue --------------------------
 * C[M, N] = (A[M, K] * B[K, N]) where H[M, N] > 0, else 0 (all row-major, contiguous; N % 128 == 0, K % 16 == 0; A and B
 * 16-byte aligned).  For linear2(relu(linear1(x))) with frozen weights (transformer_for_adapter.py:883-886, :1001-1006):
 * A = grad of the block's output, B = linear2.weight [d_model, d_ffn], H = relu(linear1(x)) -> C = the gradient in front of
 * the ReLU; replaces autograd's  mm + threshold_backward. */
int zira_gemm_drelu_f32(const float *A, const float *B, const float *H, int M, int N, int K, float *C, void *stream);

/* ---- fp32-accurate GEMM on the bf16 matrix cores for products with FROZEN weights -------------------
 * C[M, N] = epilogue(A[M, K] * B^T) for the image-token rows times a frozen weight (reference FFN
 * transformer_for_adapter.py:877-886, its backward under the freeze of groundingdino_dual_zero_rep_branch.py:722-745;
 * stands for F.linear / autograd's mm there).  Every fp32 number is exactly a1 + a2 + a3 with a_i bfloat16; the six
 * product terms a_i b_j with i + j <= 4 reproduce a b to 2^-26 |a b|, each product exact and the sums in fp32 inside
 * the matrix core (csrc/gemm_bf16x3.hip).
 *   zira_split_bf16x3_f32: w [rows, cols] fp32 row-major -> planes [3][N][K] bfloat16 (3 * rows * cols * 2 bytes, 16-byte
 *     aligned) with B[n][k] = w[n][k] (transpose = 0: N = rows, K = cols) or w[k][n] (transpose = 1: N = cols, K = rows).
 *     Done once per weight version.
 *   zira_gemm_bf16x3_f32: A [M, K] fp32 row-major, b_planes from the call above, C [M, N] row-major; N % 128 == 0,
 *     K % 32 == 0, all pointers 16-byte aligned.  epilogue 0: + bias[N];  1: + bias[N], ReLU;  2: where aux[M, N] > 0, else 0
 *     (the ReLU gradient of the FFN backward, aux = the saved activations);  3: + aux[M, N] (aux may be C itself: the
 *     gradient that reaches the input through the residual connection).  Non-finite A gives NaN.
 * Return 0, a hipError_t, or -1 for unsupported arguments.  Device pointers; enqueue only. */
int zira_split_bf16x3_f32(const float *w, int rows, int cols, int transpose, void *planes, void *stream);
int zira_gemm_bf16x3_f32(const float *A, const void *b_planes, int M, int N, int K, int epilogue, const float *bias,
                         const float *aux, float *C, void *stream);

/* ---- The same products in two-plane f16 arithmetic (csrc/gemm_f16x2.hip): each fp32 operand, scaled by a power of two per
 * (row, 32-deep K step) of A and per row of the weight, is a1 + a2 to 2^-22 with a_i f16; three exact product terms, fp32
 * sums.  Same shapes, epilogues and return codes as zira_gemm_bf16x3_f32.
 *   zira_split_f16x2_f32: planes = 2 * N * Kp halves (Kp = K rounded up to 32: rows padded with zeros) followed by N floats
 *   (1 / scale of every row): 4 N Kp + 4 N bytes.  zira_gemm_f16x2_f32 / _ex_f32 take any K % 4 == 0 with such planes. */
int zira_split_f16x2_f32(const float *w, int rows, int cols, int transpose, void *planes, void *stream);
int zira_gemm_f16x2_f32(const float *A, const void *b_planes, int M, int N, int K, int epilogue, const float *bias,
                        const float *aux, float *C, void *stream);
/* ... with the epilogues of the frozen backbone's blocks (reference models/GroundingDINO/backbone/swin_transformer.py:40-62,
 * :237-262) and N % 32 == 0 (N = 96, 192, 288, 576 of Swin-T's first stages):
 *   4: C = gelu(A B^T + bias), the exact (erf) form, the arithmetic of ATen's kernel;
 *   5: C = aux + row_scale[m / rows_per_scale] * (A B^T + bias): the block's residual with its per-image stochastic-depth
 *      factor (row_scale NULL: 1).  Epilogues 0-3 as above (row_scale must be NULL). */
int zira_gemm_f16x2_ex_f32(const float *A, const void *b_planes, int M, int N, int K, int epilogue, const float *bias,
                           const float *aux, const float *row_scale, int rows_per_scale, float *C, void *stream);

/* ---- The skinny ones among them (K = 256 or 384, N % 32 == 0: the 256-wide projections of the deformable attention and their
 * input gradients), bound by memory: a block takes 32 rows and ALL of K -- one burst of loads, one barrier, no K loop -- and the
 * weight fragments come packed in the order the matrix core reads them (csrc/gemm_f16x2_panel.hip).  Same arithmetic,
 * epilogues and return codes; A2 (or NULL) is added to A element-wise on the way in (the position code of the query,
 * transformer_for_adapter.py:893-900).
 *   zira_split_f16x2_frag_f32: frags = 2 * N * K halves in fragment order, then N floats: 4 N K + 4 N bytes. */
int zira_split_f16x2_frag_f32(const float *w, int rows, int cols, int transpose, void *frags, void *stream);
int zira_gemm_f16x2_panel_f32(const float *A, const float *A2, const void *b_frags, int M, int N, int K, int epilogue,
                              const float *bias, const float *aux, float *C, void *stream);

/* ---- The frozen feed-forward block as ONE launch per direction, on the f16 matrix cores in fp32 accuracy ----
 * forward   y  = relu(x W1^T + b1) W2^T + b2        (reference FFN: transformer_for_adapter.py:877-886)
 * backward  gx = aux + ((gy W2) * [h > 0]) W1       (its autograd under the freeze of
 *                                                    groundingdino_dual_zero_rep_branch.py:722-745: no weight gradients)
 * Both are out = epi(phi(A P^T) Q^T), A [M, 256] fp32 row-major, P [F, 256], Q [256, F] (csrc/ffn_f16x2.hip): each fp32
 * operand, scaled by a power of two, is split into two f16 planes, three exact product terms per pair; sums in fp32.  The
 * hidden activation [M, F] never exists in memory: the forward writes its sign bits (F / 8 bytes per row), the backward reads
 * them.  d_model is 256; F % 256 == 0; pointers 16-byte aligned.
 *   zira_ffn_f16x2_pack_bytes(F): size of one packed direction (0: unsupported F).
 *   zira_ffn_f16x2_pack_f32: P(h, k) = p[h * p_row_stride + k * p_col_stride], Q(n, h) = q[n * q_row_stride + h * q_col_stride]
 *     (strides in floats), p_bias [F] or NULL -> packed.  Forward: p = W1 (F x 256 row-major: strides 256, 1), q = W2
 *     (256 x F: strides F, 1), p_bias = b1.  Backward: p = W2 read transposed (strides 1, F), q = W1 read transposed
 *     (strides 1, 256), p_bias NULL.  Once per weight version.
 *   zira_ffn_f16x2_f32: backward = 0: out = relu(A P^T + p_bias) Q^T (+ q_bias[256]) (+ aux[M, 256]), mask [M, F / 32] words
 *     WRITTEN; backward = 1: out = ((A P^T) * mask bits) Q^T (+ q_bias) (+ aux), mask READ.  aux may be out.
 *     workspace: zira_ffn_f16x2_workspace_bytes(M, F) bytes, ZEROED ONCE by the caller (the launch leaves it zeroed where it
 *     must be), not shared by launches that may run at the same time; or NULL.  With it the row blocks of the chip's last,
 *     partly filled round are cut into shares of the hidden units whose sums meet there and are added in a fixed order by the
 *     block that arrives last (results repeat bit for bit from launch to launch); without it every block takes 128 whole rows.
 * Return 0, a hipError_t, or -1 for unsupported arguments.  Device pointers; enqueue only. */
size_t zira_ffn_f16x2_pack_bytes(int F);
size_t zira_ffn_f16x2_workspace_bytes(int M, int F);
int zira_ffn_f16x2_pack_f32(const float *p, long long p_row_stride, long long p_col_stride, const float *q, long long q_row_stride,
                            long long q_col_stride, const float *p_bias, int F, void *packed, void *stream);
int zira_ffn_f16x2_f32(const float *A, const void *packed, int M, int F, int backward, const float *q_bias, const float *aux,
                       void *mask, float *out, void *workspace, void *stream);

/* ---- The tall reductions again, on the bf16 matrix cores in fp32 accuracy (csrc/xty_bf16x3.hip): out[b] = P[b]^T Q[b] with
 * P [B][N][n], Q [B][N][256] token-major, n % 4 == 0; each operand split into three bfloat16 planes, six product terms, fp32
 * sums; partial tiles per token chunk meet in a second launch in a fixed order (results repeat bit for bit).
 *   out [B][n][256], or with transpose != 0 [B][256][n] (= Q^T P).  workspace: zira_xty_bf16x3_workspace_floats(B, N, n) floats,
 *   not shared by launches that may run at the same time.  Return 0, a hipError_t, or -1 for unsupported arguments. */
size_t zira_xty_bf16x3_workspace_floats(int B, int N, int n);
int zira_xty_bf16x3_f32(const float *P, const float *Q, int B, int N, int n, int transpose, float *out, float *workspace, void *stream);

/* ---- The THIN products of the fusion block's image side, batched over the images, in fp32 accuracy on the f16 matrix cores
 * (csrc/thin_f16x2.hip; the re-bracketed BiMultiHeadAttention.forward of models/GroundingDINO/fuse_modules.py:170-248 and
 * its autograd: one side of every product is the text side's H * T):
 *     C[b] [M, N] = A[b] [M, K] Wn[b]^T  (+ A2[b] [M, K] Wn2[b]^T)  (+ bias[b] [N])  (+ res[b] [M, N])
 * K <= 128 or K == 256, N <= 2048, K % 4 == 0, N % 4 == 0; all tensors dense, 16-byte aligned.  The small operand changes with
 * every step; it is split per call:
 *   zira_thin_f16x2_frag_bytes(N, K): bytes of ONE batch element's fragments (0: unsupported K).
 *   zira_thin_f16x2_split_f32: w [batch][N][K] (w_is_kn = 0) or [batch][K][N] (w_is_kn = 1) -> frags [batch][frag_bytes].
 *   zira_thin_f16x2_f32: a2 / frags2 (both or neither; K <= 128): a second source whose product is added -- a contraction over
 *     the concatenated index with each source scaled on its own; bias [batch][N] or NULL; res [batch][M][N] or NULL (may be c).
 * Return 0, a hipError_t, or -1 for unsupported arguments.  Device pointers; enqueue only. */
size_t zira_thin_f16x2_frag_bytes(int N, int K);
int zira_thin_f16x2_split_f32(const float *w, int batch, int N, int K, int w_is_kn, void *frags, void *stream);
int zira_thin_f16x2_f32(const float *a, const void *frags, const float *a2, const void *frags2, int batch, int M, int N, int K,
                        const float *bias, const float *res, float *c, void *stream);

/* ---- GroupNorm of the input projections (reference groundingdino_dual_zero_rep_branch.py:487-529: nn.GroupNorm(32, 256) behind each
 * level's conv + side branch), NCHW fp32, forward and input gradient (csrc/groupnorm.hip).  x, res, y, sum_out, dy, dx: [B, C, HW];
 * gamma / beta [C] or NULL; mean / rstd [B * G].  res (or NULL) is added to x on the way in and the sum written to sum_out (the
 * backward's x).  The backward is for FROZEN gamma / beta: it returns dx only.  (C / G) * HW % 4 == 0, HW >= 4.
 * workspace: zira_groupnorm_workspace_floats(B, C, HW, G) floats (0: unsupported shape).  Return 0, a hipError_t, or -1. */
size_t zira_groupnorm_workspace_floats(int B, int C, int HW, int G);
int zira_groupnorm_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, int B, int C, int HW, int G, float eps,
                           float *sum_out, float *y, float *mean, float *rstd, float *workspace, void *stream);
int zira_groupnorm_bwd_f32(const float *dy, const float *x, const float *gamma, const float *mean, const float *rstd, int B, int C, int HW,
                           int G, float *dx, float *workspace, void *stream);

/* ---- Decoder reference boxes: sine embedding, forward only -------------------------------------
 * zira_sine_embed_f32 replaces gen_sineembed_for_position (groundingdino/models/GroundingDINO/utils.py:204-231):
 *   pos [rows, C] (x, y[, w, h]), C = 2 or 4;  dim_t [T] = temperature^(2 (i // 2) / T);  scale = 2 pi;
 *   out [rows, C * T], parts ordered (y, x[, w, h]), sin on even / cos on odd channels of pos * scale / dim_t.
 * Bit-identical to the PyTorch op chain it stands for.  Device pointers. */
int zira_sine_embed_f32(const float *pos, const float *dim_t, long long rows, int C, int T, float scale, float *out,
                        void *stream);

/* What a decoder layer needs of the current boxes, in one launch (transformer_for_adapter.py:760-770): for ref [Q, B, 4] (sigmoid
 * space, no gradient) and valid ratios ratio [B, L, 2]
 *   ref_in [Q, B, L, 4] = ref[:, :, None] * cat([ratio, ratio], -1)     ref_bf [B, Q, L, 4] = the same, batch-first
 *   sine   [Q, B, 4 T]  = zira_sine_embed_f32 of ref_in[:, :, 0, :]     (T >= L)
 * with the multiplies and divides of the PyTorch op chain (bit-identical to it). */
int zira_decoder_prep_f32(const float *ref, const float *ratio, const float *dim_t, int Q, int B, int L, int T, float scale,
                          float *ref_in, float *ref_bf, float *sine, void *stream);

/* Iterative box refinement of the decoder (transformer_for_adapter.py:790-797, inverse_sigmoid util/misc.py:704-708): the
 * last layer of the box MLP (4 outputs) with the inverse-sigmoid of the current boxes added and the sigmoid taken,
 *   new_ref[r, j] = sigmoid(<h[r, :], w[j, :]> + b[j] + log(max(x, eps) / max(1 - x, eps))),  x = clamp(ref[r, j], 0, 1),
 * for h [rows, K] (the MLP's second hidden layer after its ReLU), w [4, K], b [4], ref / new_ref [rows, 4]; and the gradient
 * it sends back in front of that ReLU: g_h[r, k] = (sum_j g_new[r, j] s (1 - s) w[j, k]) where h[r, k] > 0, s = new_ref[r, j].
 * K a multiple of 4; h, w, g_h 16-byte aligned.  Device pointers. */
int zira_box_refine_fwd_f32(const float *h, const float *w, const float *b, const float *ref, long long rows, int K, float eps,
                            float *new_ref, void *stream);
int zira_box_refine_bwd_f32(const float *g_new, const float *new_ref, const float *w, const float *h, long long rows, int K,
                            float *g_h, void *stream);


/* Level geometry from the padding mask (three op chains of the reference, ~150 launch-bound ATen kernels per training step):
 *   mask [B, S] bytes (non-zero = padded), shapes [L, 2] / start [L] int64 on the device (the levels tile [0, S)).
 * zira_level_valid_ratios_f32 (get_valid_ratio, transformer_for_adapter.py:226-233, per level and stacked at :260):
 *   counts [B, L, 2] = (unpadded columns of the level's first row, unpadded rows of its first column) as floats,
 *   ratios [B, L, 2] = counts * (1 / (W, H)) (how ATen divides by a host scalar); either output may be null.
 * zira_encoder_ref_points_f32 (get_reference_points, transformer_for_adapter.py:482-497): ratios [B, L, 2] ->
 *   ref_points [B, S, L, 2] = ((x + 0.5) / (ratio_x W), (y + 0.5) / (ratio_y H)) of the pixel's own level, times the ratios of level j.
 * zira_encoder_proposals_f32 (gen_encoder_output_proposals, utils.py:56-116, learnedwh = None): odds [B, S, 4] = p / (1 - p)
 *   for p = ((x + 0.5) / valid columns, (y + 0.5) / valid rows, 0.05 2^l, 0.05 2^l), +inf where the pixel is padded or any p is
 *   outside (0.01, 0.99); drop [B, S] bytes = 1 there (the rows of `memory` the reference zeroes).  The caller takes the log
 *   (ATen's log and libm's logf differ in the last bit).  counts_scratch: B * L * 2 floats.
 * The same separately rounded fp32 operations as the op chains: bit-identical.  Return 0 or a hipError_t; enqueue only. */
int zira_level_valid_ratios_f32(const void *mask, const int64_t *shapes, const int64_t *start, int B, long long S, int L,
                                float *counts, float *ratios, void *stream);
int zira_encoder_ref_points_f32(const float *ratios, const int64_t *shapes, const int64_t *start, int B, long long S, int L,
                                float *ref_points, void *stream);
int zira_encoder_proposals_f32(const void *mask, const int64_t *shapes, const int64_t *start, int B, long long S, int L,
                               float *counts_scratch, float *odds, void *drop, void *stream);


/* The set-prediction losses of all S prediction sets of a step at once (SetCriterion.loss_labels / loss_boxes,
 * criterion/criterion.py:104-181, with sigmoid_focal_loss :31-59 and generalized_box_iou util/box_ops.py:39-66):
 *   logits [S, B, Q, C], boxes [S, B, Q, 4] (cx, cy, w, h);  matches q_idx / t_idx [S, M] int64 with image_of [M] int64: set s
 *   matches query q_idx[s, k] of image image_of[k] with target t_idx[s, k] of the concatenated targets (labels [T] int64,
 *   boxes_all [T, 4]); num_boxes: one float on the device.
 *   out [3, S] = (focal loss summed over the set / num_boxes, sum of the pairs' L1 / num_boxes, sum of (1 - GIoU) / num_boxes).
 * _bwd: g_out [3, S] -> g_logits [S, B, Q, C] and g_boxes [S, B, Q, 4] (either may be null), every element written.
 * scratch: zira_stacked_losses_scratch_bytes(S, B, Q, M).  fp32 in the op chain's order, the focal sum reduced in double;
 * gradients follow autograd's conventions at the kinks.  Return 0 or a hipError_t; enqueue only. */
size_t zira_stacked_losses_scratch_bytes(int S, int B, int Q, int M);
int zira_stacked_losses_fwd_f32(const float *logits, const float *boxes, const int64_t *q_idx, const int64_t *t_idx,
                                const int64_t *image_of, const int64_t *labels, const float *boxes_all, const float *num_boxes,
                                int S, int B, int Q, int C, int M, float alpha, float gamma, void *scratch, float *out, void *stream);
int zira_stacked_losses_bwd_f32(const float *logits, const float *boxes, const int64_t *q_idx, const int64_t *t_idx,
                                const int64_t *image_of, const int64_t *labels, const float *boxes_all, const float *num_boxes,
                                const float *g_out, int S, int B, int Q, int C, int M, float alpha, float gamma, float *g_logits,
                                float *g_boxes, void *stream);


/* The box head's last step, elementwise (groundingdino_dual_zero_rep_branch.py:563-569, inverse_sigmoid util/misc.py:704-708):
 *   out[i] = sigmoid(delta[i] + log(max(x, eps) / max(1 - x, eps))),  x = clamp(ref[i], 0, 1);
 *   _bwd: g_delta[i] = g_out[i] out (1 - out);  g_ref[i] = g_delta[i] * ([x >= eps] / x + [1 - x >= eps] / (1 - x)) where
 *   0 <= ref[i] <= 1, else 0 (autograd's clamp convention); either gradient may be null.  n elements, device pointers. */
int zira_box_head_fwd_f32(const float *delta, const float *ref, long long n, float eps, float *out, void *stream);
int zira_box_head_bwd_f32(const float *g_out, const float *out, const float *ref, long long n, float eps, float *g_delta,
                          float *g_ref, void *stream);


/* The text side of an image <-> text fusion block whose six projections are frozen and composed (BiMultiHeadAttention /
 * BiAttentionBlock, fuse_modules.py:99-305; this package's re-bracketing around the B x T text tokens): M = B T rows,
 * H heads, image width Dv, text width Dl.  All fp32, contiguous, on the device; enqueue only; 0 or a hipError_t.
 * zira_text_prep_fwd_f32: l_ln [B, T, Dl] = LayerNorm(l_in; ln_w, ln_b, eps), stats [M, 2] = (mean, rstd), and
 *     [A | C | Z] = l_ln W1 + b1 (W1 [Dl, 2 H Dv + H] = [AC | Z] of _composed_text_side) scattered as
 *     a [B, Dv, H T] (a[b, d, h T + t] = A[b t, h Dv + d]),  c [B, H T],  z [B, H T, Dv].
 * zira_text_prep_bwd_f32: g_l_in [B, T, Dl] from g_a / g_c / g_z (same layouts; any may be null) and g_l_ln (the gradient that
 *     reaches l_ln directly; may be null); W1T [2 H Dv + H, Dl] = W1 transposed; scratch: zira_text_side_scratch_floats floats
 *     (the partial products of the K split; also enough for zira_text_out_fwd_f32).
 * zira_text_out_fwd_f32: out [B, T, Dl] = l_ln + scale (o0 + U O),  U[b t, h Dv + d] = u[b, h T + t, d] / colsum[b, h T + t],
 *     O [H Dv, Dl], scale[b, n] = gamma[n] * keep[b] (keep: the per-sample stochastic-depth factor, may be null).
 * zira_text_out_bwd_f32: g [B, T, Dl] -> g_u [B, H T, Dv], g_colsum [B, H T]  (the gradient of l_ln is g itself); OT [Dl, H Dv] =
 *     O transposed.  Limit: Dl <= 256 (otherwise hipErrorInvalidValue). */
size_t zira_text_side_scratch_floats(int B, int T, int H, int Dv, int Dl);
int zira_text_prep_fwd_f32(const float *l_in, const float *ln_w, const float *ln_b, float eps, const float *W1, const float *b1,
                           int B, int T, int H, int Dv, int Dl, float *l_ln, float *a, float *c, float *z, float *stats, void *stream);
int zira_text_prep_bwd_f32(const float *g_a, const float *g_c, const float *g_z, const float *g_l_ln, const float *l_in,
                           const float *ln_w, const float *stats, const float *W1T, int B, int T, int H, int Dv, int Dl,
                           float *scratch, float *g_l_in, void *stream);
int zira_text_out_fwd_f32(const float *u, const float *colsum, const float *l_ln, const float *O, const float *o0,
                          const float *gamma, const float *keep, int B, int T, int H, int Dv, int Dl, float *scratch, float *out,
                          void *stream);
int zira_text_out_bwd_f32(const float *g, const float *u, const float *colsum, const float *OT, const float *gamma,
                          const float *keep, int B, int T, int H, int Dv, int Dl, float *g_u, float *g_colsum, void *stream);


/* Sine position encoding of one feature level from its padding mask (PositionEmbeddingSineHW, position_encoding.py:78-134):
 *   mask [B, H, W] bytes (non-zero = padded);  dim_t_y / dim_t_x [F] = temperature^(2 (i // 2) / F) (the caller forms them with the
 *   reference's ops once);  out [B, H, W, 2 F] = (pos_y | pos_x) -- the reference's [B, 2 F, H, W] result is a permuted view of it.
 *   normalize: embed / (last + eps) * scale.  The same separately rounded fp32 operations and libm's sinf / cosf: bit-identical to
 *   the op chain (~15 ATen launches per level).  F even.  Return 0 or a hipError_t; enqueue only. */
int zira_sine_pos_hw_f32(const void *mask, int B, int H, int W, int F, int normalize, float scale, float eps, const float *dim_t_y,
                         const float *dim_t_x, float *out, void *stream);


/* Human-readable build tag, e.g. "zira_msda 0.1 gfx950". Static storage. */
const char *zira_msda_version(void);

/* Names of the kernels the f32 entry points pick for channel width D (the lean / tiled
 * kernels for D = 16, 32, 64, "rows<N>" for 4, 8, 128, 256, "generic" otherwise).
 * Host-only helper for tests/bench labelling. Static storage. */
const char *zira_msda_variant_f32(int D);

#ifdef __cplusplus
}
#endif
#endif /* ZIRA_MSDA_H_ */
