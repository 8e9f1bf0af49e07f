// msda_internal.h -- links the translation units of libzira_msda.so (not part of the C ABI).
#ifndef ZIRA_MSDA_INTERNAL_H_
#define ZIRA_MSDA_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace zira {

// "Cell walk" backward (csrc/msda_cells.hip): bin kernel + walk kernel, caller-provided workspace.
// cells_workspace_bytes() returns 0 when the path does not apply to these dimensions.
size_t cells_workspace_bytes(int B, int S, int M, int D, int L, int Q, int P);

int cells_backward_f32(const float *grad_out, const float *value, const int64_t *shapes,
                       const int64_t *start, const float *loc, const float *attn, int B, int S,
                       int M, int D, int L, int Q, int P, float *grad_value, float *grad_loc,
                       float *grad_attn, void *workspace, size_t workspace_bytes, hipStream_t st);

// "Plan + tile accumulate" backward (csrc/msda_tiles.hip) for sparse calls with D = 32.  The plan depends on the
// sampling locations and attention weights (zira_msda_plan_f32: at forward time, beside the gather); the backward takes it as a handle.
// tiles_plan_bytes() returns 0 when the path does not apply to these dimensions; the launchers return -1 when they
// cannot serve the call (the caller then takes another path), else a hipError_t.
size_t tiles_plan_bytes(int B, int S, int M, int D, int L, int Q, int P);

int tiles_plan_f32(const int64_t *shapes, const int64_t *start, const float *loc, const float *attn, int B, int S, int M, int D,
                   int L, int Q, int P, void *plan, size_t plan_bytes, hipStream_t st);

// forward + plan in one launch (D = 32); -1: not applicable
int tiles_fwd_plan_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc, const float *attn,
                       int B, int S, int M, int D, int L, int Q, int P, float *out, void *plan, size_t plan_bytes, hipStream_t st);

// all three gradients: the gather half (grad_sampling_loc / grad_attn_weight) rides in the accumulate launch
int tiles_backward_planned_f32(const float *grad_out, const float *value, const int64_t *shapes, const int64_t *start,
                               const float *loc, const float *attn, int B, int S, int M, int D, int L, int Q, int P,
                               float *grad_value, float *grad_loc, float *grad_attn, void *plan, size_t plan_bytes,
                               hipStream_t st);

// Forward for calls in which every pixel is a query (Q = S, D = 32: the encoder): value patches staged in LDS
// (csrc/msda_patch.hip).  -1: not applicable (the caller takes the lean kernel), else a hipError_t.
int patch_forward_f32(const float *value, const int64_t *shapes, const int64_t *start, const float *loc, const float *attn, int B,
                      int S, int M, int D, int L, int Q, int P, float *out, hipStream_t st);

}  // namespace zira

#endif
