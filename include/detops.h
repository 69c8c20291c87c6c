/*
 * detops.h — C ABI of libdetops_gfx950.so, the MI355X (gfx950 / CDNA4) detection-head
 * operator library.
 *
 * This is the drop-in boundary for the hot path of facebookresearch/maskrcnn-benchmark:
 * every entry point below replaces one function of the reference's pybind module
 * `maskrcnn_benchmark._C` (reference: maskrcnn_benchmark/csrc/vision.cpp:9-25), which the
 * reference's `maskrcnn_benchmark.layers` package binds.  The host side that re-exports the
 * exact 14-name `_C` surface on top of this ABI is
 * `maskrcnn-benchmark_amd/maskrcnn_benchmark/_C.py` (ctypes).
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless the name ends in `_host`.
 *   - Tensors are dense, contiguous, NCHW, fp32 unless the function name says otherwise.
 *   - `stream` is a `hipStream_t` passed as `void*` (0 = the null stream).  Every launch goes
 *     to that stream; no entry point synchronises the host, allocates or frees device memory
 *     (scratch comes from caller-provided workspaces sized by the *_workspace_bytes queries),
 *     so every call is hipGraph-capturable and re-entrant per stream.
 *   - Return value: 0 on success, a positive hipError_t if a HIP call failed, or one of the
 *     negative DETOPS_E* codes for argument errors.  Nothing is written on error.
 *   - ROI rows are (batch_index, x1, y1, x2, y2) in image coordinates, batch index stored as
 *     float (reference: maskrcnn_benchmark/modeling/poolers.py:78-89).
 */
#ifndef DETOPS_H_
#define DETOPS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DETOPS_ABI_VERSION 1

#define DETOPS_EINVAL   (-1) /* bad shape / null pointer / negative size            */
#define DETOPS_EWORKSPACE (-2) /* workspace too small (see *_workspace_bytes)       */
#define DETOPS_EUNSUPPORTED (-3) /* configuration outside what the kernels implement */

typedef void* detops_stream_t; /* hipStream_t */

/* Library identification: returns DETOPS_ABI_VERSION; `arch` (may be NULL) receives a static
 * string naming the ISA the device code was built for ("gfx950"). */
int detops_version(const char** arch);

/* Tuning / test switches (all default 0 = automatic).  Read once at library load from the environment variable
 * DETOPS_TUNING="key=value,key=value"; this call changes one at run time (tests, A/B measurements — process-wide,
 * not thread-safe against concurrent launches).  The launch paths never read the environment.  Keys:
 *   roi_bwd_impl      1 ring (needs a workspace) | 2 scan | 3 atomic scatter
 *   roi_bwd_seg       ring: hits per hit-list segment (default 32, >= 8)
 *   roi_bwd_groups    scan: ROI-list split        roi_bwd_scan_ct   scan: channels per workgroup (4 | 16)
 *   roi_bwd_debug     ablation bits (1 = skip the walk)
 *   roi_fwd_impl      1 generic gather kernel    roi_fwd_order     1 never rank | 2 rank even tiny maps
 *   roi_fwd_order_mink  smallest K that gets the ranking pre-pass
 *   dcn_col2im        1 gather | 2 scatter | 3 ell    dcn_fused  1 force | 2 off    dcn_gather_xcd  1 plain block order
 *   dcn_nhwc          2 = never use the channels-last pipeline (the layers then run the reference-layout kernels)
 * Returns 0, or DETOPS_EINVAL for an unknown key. */
int detops_tuning_set(const char* key, int value);
int detops_tuning_get(const char* key, int* value);

/* ------------------------------------------------------------------------------------------
 * ROIAlign  — replaces _C.roi_align_forward / _C.roi_align_backward
 *   reference: csrc/ROIAlign.h:11-25 (forward), :27-45 (backward);
 *              csrc/cpu/ROIAlign_cpu.cpp:113-257, csrc/cuda/ROIAlign_cuda.cu:64-122,177-346.
 *   input  [N,C,H,W]   rois [K,5]   output / grad_out [K,C,PH,PW]   grad_in [N,C,H,W]
 *   sampling_ratio <= 0 selects the adaptive grid ceil(roi_h/PH) x ceil(roi_w/PW).
 *   backward: `zero_grad_in` != 0 makes the call zero-fill grad_in first (the reference
 *   allocates it with at::zeros, ROIAlign_cuda.cu:316); 0 accumulates into the caller's buffer
 *   (used by the multi-level pooler to fuse several ROI sets into one gradient map).
 * ---------------------------------------------------------------------------------------- */
int detops_roi_align_forward_f32(const float* input, const float* rois, float* output,
                                 int N, int C, int H, int W, int K, int PH, int PW,
                                 float spatial_scale, int sampling_ratio,
                                 detops_stream_t stream);

int detops_roi_align_backward_f32(const float* grad_out, const float* rois, float* grad_in,
                                  int N, int C, int H, int W, int K, int PH, int PW,
                                  float spatial_scale, int sampling_ratio, int zero_grad_in,
                                  detops_stream_t stream);

/* Forward with a caller-provided workspace (`detops_roi_align_forward_workspace_bytes(K, PH, PW, sampling_ratio)`
 * bytes; 0 = not used for this shape): ONE pre-pass launch (a) builds per-ROI sample records — footprint bounds and,
 * per output bin, the patch offsets and bilinear weights of its samples by the reference arithmetic — which every
 * channel-chunk workgroup of the main launch loads instead of re-deriving, and (b) for K >= 384 ranks the ROIs by
 * (level, image, position) so that the main launch visits overlapping footprints while they are still in the XCD's
 * L2.  The output is identical, bit for bit, to the workspace-free call (what detops_roi_align_forward_f32 /
 * detops_roi_align_fpn_forward_f32 do); workspace == NULL or too small selects that path. */
size_t detops_roi_align_forward_workspace_bytes(int K, int PH, int PW, int sampling_ratio);

int detops_roi_align_forward_ws_f32(const float* input, const float* rois, float* output,
                                    int N, int C, int H, int W, int K, int PH, int PW,
                                    float spatial_scale, int sampling_ratio, void* workspace,
                                    size_t workspace_bytes, detops_stream_t stream);

/* Backward with a caller-provided workspace: selects the binned pixel-owner kernel (a pre-pass launch
 * builds per-ROI adjoint rows and per-tile hit lists in the workspace, the main launch only gathers).
 * `detops_roi_align_backward_workspace_bytes` returns the size for a set of maps (H_host/W_host:
 * num_levels entries; 1 for the single-map call), or 0 when the shape is served by the
 * workspace-free kernels.  workspace == NULL or too small: same result through the workspace-free
 * path (what detops_roi_align_backward_f32 / detops_roi_align_fpn_backward_f32 do). */
size_t detops_roi_align_backward_workspace_bytes(const int* H_host, const int* W_host,
                                                 int num_levels, int N, int C, int K, int PH, int PW);

int detops_roi_align_backward_ws_f32(const float* grad_out, const float* rois, float* grad_in,
                                     int N, int C, int H, int W, int K, int PH, int PW,
                                     float spatial_scale, int sampling_ratio, int zero_grad_in,
                                     void* workspace, size_t workspace_bytes,
                                     detops_stream_t stream);

/* Multi-level (FPN) ROIAlign in ONE launch — the sync-free form of
 * modeling/poolers.py:91-121 (LevelMapper :11-42 + per-level ROIAlign :116-119).
 *   inputs[l] : device pointer to level l's feature map [N,C,H[l],W[l]], scale[l] its stride^-1
 *   rois [K,5]; the FPN level of each ROI is computed on device with the reference's formula
 *   lvl = clamp(floor(canonical_level + log2(sqrt(area)/canonical_scale + eps)), k_min, k_max)
 *   with area = (x2-x1+1)*(y2-y1+1)  (structures/bounding_box.py:212-216, TO_REMOVE = 1).
 *   `levels_out` (may be NULL) receives the 0-based level index per ROI as int32.
 *   num_levels <= DETOPS_MAX_LEVELS.  The pointer/shape arrays are HOST arrays (copied into
 *   the kernel argument block).
 */
#define DETOPS_MAX_LEVELS 8
int detops_roi_align_fpn_forward_f32(const float* const* inputs_host, const int* H_host,
                                     const int* W_host, const float* scale_host,
                                     int num_levels, const float* rois, float* output,
                                     int32_t* levels_out, int N, int C, int K, int PH, int PW,
                                     int sampling_ratio, int k_min, int k_max,
                                     float canonical_scale, float canonical_level, float eps,
                                     detops_stream_t stream);

int detops_roi_align_fpn_forward_ws_f32(const float* const* inputs_host, const int* H_host,
                                        const int* W_host, const float* scale_host,
                                        int num_levels, const float* rois, float* output,
                                        int32_t* levels_out, int N, int C, int K, int PH, int PW,
                                        int sampling_ratio, int k_min, int k_max,
                                        float canonical_scale, float canonical_level, float eps,
                                        void* workspace, size_t workspace_bytes,
                                        detops_stream_t stream);

int detops_roi_align_fpn_backward_ws_f32(const float* grad_out, const float* rois,
                                         const int32_t* levels, float* const* grad_inputs_host,
                                         const int* H_host, const int* W_host,
                                         const float* scale_host, int num_levels, int N, int C,
                                         int K, int PH, int PW, int sampling_ratio,
                                         int zero_grad_in, void* workspace, size_t workspace_bytes,
                                         detops_stream_t stream);

/* The binned backward in two calls.  Its pre-pass (per-ROI adjoint rows + per-tile hit lists) depends on the ROIs and the
 * map shapes only: `..._prepare_f32` may be issued at FORWARD time, on any stream, into a workspace of
 * detops_roi_align_backward_workspace_bytes(...) bytes that the caller keeps until the backward pass; `..._prepared_f32`
 * (same shapes, same workspace, stream-ordered after the prepare call) then launches the main kernel alone.  `prepare`
 * returns DETOPS_EUNSUPPORTED when the shape is served by the other kernels: use detops_roi_align_fpn_backward_ws_f32. */
int detops_roi_align_fpn_backward_prepare_f32(const float* rois, const int32_t* levels, const int* H_host,
                                              const int* W_host, const float* scale_host, int num_levels, int N,
                                              int C, int K, int PH, int PW, int sampling_ratio, void* workspace,
                                              size_t workspace_bytes, detops_stream_t stream);

int detops_roi_align_fpn_backward_prepared_f32(const float* grad_out, float* const* grad_inputs_host,
                                               const int* H_host, const int* W_host, const float* scale_host,
                                               int num_levels, int N, int C, int K, int PH, int PW,
                                               int zero_grad_in, void* workspace, size_t workspace_bytes,
                                               detops_stream_t stream);

int detops_roi_align_fpn_backward_f32(const float* grad_out, const float* rois,
                                      const int32_t* levels, float* const* grad_inputs_host,
                                      const int* H_host, const int* W_host,
                                      const float* scale_host, int num_levels, int N, int C,
                                      int K, int PH, int PW, int sampling_ratio,
                                      int zero_grad_in, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ROIAlign over a CHANNELS-LAST pyramid (csrc/roi_align_nhwc.hip) — the same operator as above (reference
 * csrc/cuda/ROIAlign_cuda.cu:64-122 forward, :125-254 backward, modeling/poolers.py:91-121) for feature maps stored
 * [N, H[l], W[l], C] (what MIOpen's implicit-GEMM convolutions read and write natively: a detector whose backbone runs
 * channels-last needs no layout transposes around its pooler).  num_levels == 1 is the single-map operator.
 *   forward : inputs[l] NHWC; output [K, C, PH, PW] (output_nhwc = 0) or [K, PH, PW, C] (output_nhwc = 1).  Reference
 *             operation order with FP contraction off: bit-identical to the reference CPU kernel (and to the NCHW
 *             kernels above) for finite inputs.  workspace (detops_roi_align_fpn_forward_nhwc_workspace_bytes(K) bytes,
 *             may be NULL): the ROI visiting order (L2 locality only; the output does not depend on it).
 *   backward: grad_out [K, C, PH, PW] (grad_out_nhwc = 0) or [K, PH, PW, C] (1); grad_inputs[l] NHWC, every element
 *             written exactly once (zero_grad_in = 1) or added to (0); atomic-free, bit-reproducible run to run;
 *             `levels` = the forward's levels_out (required when num_levels > 1); workspace REQUIRED
 *             (detops_roi_align_fpn_backward_nhwc_workspace_bytes: per-ROI adjoint rows + per-tile hit lists).
 * ---------------------------------------------------------------------------------------- */
size_t detops_roi_align_fpn_forward_nhwc_workspace_bytes(int K);
int detops_roi_align_fpn_forward_nhwc_f32(const float* const* inputs_host, const int* H_host, const int* W_host,
                                          const float* scale_host, int num_levels, const float* rois, float* output,
                                          int output_nhwc, int32_t* levels_out, int N, int C, int K, int PH, int PW,
                                          int sampling_ratio, int k_min, int k_max, float canonical_scale,
                                          float canonical_level, float eps, void* workspace, size_t workspace_bytes,
                                          detops_stream_t stream);
/* The pixel-owner ring backward (csrc/roi_align_bwd.hip) with channels-last gradient maps: same kernels, the store epilogue
 * writes a thread's channel sums as consecutive floats.  grad_out [K, C, PH, PW]; workspace of
 * detops_roi_align_backward_workspace_bytes(...) bytes.  DETOPS_EUNSUPPORTED when the ring plan does not serve the shape
 * (small / under-filled maps, other bin counts): use detops_roi_align_fpn_backward_nhwc_f32 then. */
int detops_roi_align_fpn_backward_ring_nhwc_f32(const float* grad_out, const float* rois, const int32_t* levels,
                                                float* const* grad_inputs_host, const int* H_host, const int* W_host,
                                                const float* scale_host, int num_levels, int N, int C, int K, int PH,
                                                int PW, int sampling_ratio, int zero_grad_in, void* workspace,
                                                size_t workspace_bytes, detops_stream_t stream);
size_t detops_roi_align_fpn_backward_nhwc_workspace_bytes(const int* H_host, const int* W_host, int num_levels, int N,
                                                          int C, int K, int PH, int PW);
int detops_roi_align_fpn_backward_nhwc_f32(const float* grad_out, int grad_out_nhwc, const float* rois,
                                           const int32_t* levels, float* const* grad_inputs_host, const int* H_host,
                                           const int* W_host, const float* scale_host, int num_levels, int N, int C,
                                           int K, int PH, int PW, int sampling_ratio, int zero_grad_in, void* workspace,
                                           size_t workspace_bytes, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ROIPool — replaces _C.roi_pool_forward / _C.roi_pool_backward
 *   reference: csrc/ROIPool.h:11-45, csrc/cuda/ROIPool_cuda.cu:16-108 (CUDA-only there).
 *   argmax [K,C,PH,PW] int32: index h*W+w inside the channel plane, -1 for an empty bin.
 * ---------------------------------------------------------------------------------------- */
int detops_roi_pool_forward_f32(const float* input, const float* rois, float* output,
                                int32_t* argmax, int N, int C, int H, int W, int K, int PH,
                                int PW, float spatial_scale, detops_stream_t stream);

int detops_roi_pool_backward_f32(const float* grad_out, const float* rois,
                                 const int32_t* argmax, float* grad_in, int N, int C, int H,
                                 int W, int K, int PH, int PW, int zero_grad_in,
                                 detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NMS — replaces _C.nms
 *   reference: csrc/nms.h:10-28, csrc/cpu/nms_cpu.cpp:5-75 (the CPU semantics are the
 *   contract: suppress j iff IoU(i,j) >= threshold, +1 pixel convention, result = kept
 *   ORIGINAL indices in ascending order, int64).
 *   boxes [n,4] xyxy fp32, scores [n] fp32.  Sort order: score descending, ties broken by
 *   ascending original index (stable).
 *   keep      [n]  int64, first *num_keep entries valid (ascending original indices)
 *   num_keep  [1]  int32 on device (the caller decides when/if to read it back); -1 = FAILED: a workgroup of the
 *                  single-launch kernel waited for another one of the same launch beyond its polling budget (seconds:
 *                  the compute units were held by other work).  The repair pass described at
 *                  detops_nms_batched_status_f32 redoes such a segment before the call's work ends, so a caller only ever
 *                  SEES -1 (and an all-zero keep mask) with the repair switched off (tuning "nms_no_repair" = 1: tests)
 *   Entirely device-side: no host round trip (the reference's CUDA path copies the n x n/64
 *   bitmask to the host and scans it there, csrc/cuda/nms.cu:100-123).
 * ---------------------------------------------------------------------------------------- */
size_t detops_nms_workspace_bytes(int n);

int detops_nms_f32(const float* boxes, const float* scores, int n, float threshold,
                   int64_t* keep, int32_t* num_keep, void* workspace, size_t workspace_bytes,
                   detops_stream_t stream);

/* Batched / segmented NMS: S independent problems in one launch sequence (the RPN's
 * per-(image, level) loop, modeling/rpn/inference.py:111-121, and the per-class loop of
 * modeling/roi_heads/box_head/inference.py:122-135).
 *   seg_offsets [S+1] int32 on device: segment s owns rows seg_offsets[s] .. seg_offsets[s+1]-1
 *   of boxes/scores; max_n >= the longest segment (host-known upper bound).
 *   keep [total] int64: segment s writes its kept indices (LOCAL to the segment, ascending)
 *   at keep[seg_offsets[s] ...]; num_keep [S] int32.
 */
size_t detops_nms_batched_workspace_bytes(int num_segments, int max_n);

int detops_nms_batched_f32(const float* boxes, const float* scores,
                           const int32_t* seg_offsets, int num_segments, int max_n,
                           float threshold, int64_t* keep, int32_t* num_keep, void* workspace,
                           size_t workspace_bytes, detops_stream_t stream);

/* Same computation, dense result: keep_mask [total] uint8, keep_mask[seg_offsets[s] + i] = 1 iff
 * box i of segment s survives.  Fixed-shape output for callers that never read a count back
 * (the padded RPN proposal path; replaces the `keep` index list + nonzero of
 * modeling/rpn/inference.py:111-121). */
int detops_nms_batched_mask_f32(const float* boxes, const float* scores,
                                const int32_t* seg_offsets, int num_segments, int max_n,
                                float threshold, uint8_t* keep_mask, int32_t* num_keep,
                                void* workspace, size_t workspace_bytes, detops_stream_t stream);

/* Both result forms optional (keep and / or keep_mask, at least one) + a sticky STATUS word.  Every entry point of this
 * section launches, right behind the single-launch kernel and on the same stream, a repair pass: a segment the single
 * launch marked as failed (num_keep = -1, see above) is redone by ONE workgroup that waits for nobody, so that — like the
 * reference (csrc/cuda/nms.cu:70-131) — no segment is ever dropped; num_keep[s] then holds the real count.  *status
 * (int32 on device, may be NULL; never cleared by the library) is incremented once per redone segment: a trainer reads it
 * with its next loss read-back and reports how often the fast path gave up — no synchronisation of its own. */
int detops_nms_batched_status_f32(const float* boxes, const float* scores, const int32_t* seg_offsets,
                                  int num_segments, int max_n, float threshold, int64_t* keep, uint8_t* keep_mask,
                                  int32_t* num_keep, int32_t* status, void* workspace, size_t workspace_bytes,
                                  detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SigmoidFocalLoss — replaces _C.sigmoid_focalloss_forward / _backward
 *   reference: csrc/SigmoidFocalLoss.h:10-41, csrc/cuda/SigmoidFocalLoss_cuda.cu:20-101.
 *   logits [R,C] fp32, targets [R] int32 in {-1 ignore, 0 background, 1..C}, losses [R,C].
 *   `_sum` additionally (or only, if losses == NULL) accumulates sum(losses) into *loss_sum
 *   (fp32, device; must be zeroed by the caller) — what layers/sigmoid_focal_loss.py:66-67
 *   does with a separate .sum().
 * ---------------------------------------------------------------------------------------- */
int detops_sigmoid_focal_loss_forward_f32(const float* logits, const int32_t* targets,
                                          float* losses, int R, int C, float gamma,
                                          float alpha, detops_stream_t stream);

int detops_sigmoid_focal_loss_backward_f32(const float* logits, const int32_t* targets,
                                           const float* d_losses, float* d_logits, int R,
                                           int C, float gamma, float alpha,
                                           detops_stream_t stream);

/* backward of the summed loss: every element shares ONE upstream gradient, read from the device
 * scalar *d_loss_scalar (no [R,C] broadcast buffer). */
int detops_sigmoid_focal_loss_backward_scalar_f32(const float* logits, const int32_t* targets,
                                                  const float* d_loss_scalar, float* d_logits,
                                                  int R, int C, float gamma, float alpha,
                                                  detops_stream_t stream);

int detops_sigmoid_focal_loss_forward_sum_f32(const float* logits, const int32_t* targets,
                                              float* losses /* nullable */, float* loss_sum,
                                              int R, int C, float gamma, float alpha,
                                              detops_stream_t stream);

/* As _forward_sum, but workgroup partial sums are spread over `num_slots` words
 * (partial_sums[num_slots], zeroed by the caller; sum(losses) = sum of the slots): thousands of
 * workgroups retiring together on one word serialise in L2. */
int detops_sigmoid_focal_loss_forward_partial_sums_f32(const float* logits, const int32_t* targets,
                                                       float* losses /* nullable */,
                                                       float* partial_sums, int num_slots, int R,
                                                       int C, float gamma, float alpha,
                                                       detops_stream_t stream);

/* Two-stage, atomic-free form of _forward_sum: stage 1 leaves one sum per workgroup in `workspace`
 * (`detops_sigmoid_focal_loss_sum_workspace_bytes()` bytes, need not be initialised), stage 2 adds them in a fixed
 * order and OVERWRITES loss_sum[0] — no zero-fill launch, no host-side reduction, bit-reproducible run to run. */
size_t detops_sigmoid_focal_loss_sum_workspace_bytes(void);
int detops_sigmoid_focal_loss_forward_sum_ws_f32(const float* logits, const int32_t* targets,
                                                 float* losses, float* loss_sum, int R, int C,
                                                 float gamma, float alpha, void* workspace,
                                                 size_t workspace_bytes, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Deformable convolution building blocks — the kernels behind _C.deform_conv_forward,
 * _backward_input, _backward_parameters, _C.modulated_deform_conv_forward/_backward
 *   reference: csrc/deform_conv.h:11-191, csrc/cuda/deform_conv_kernel_cuda.cu:197-874,
 *              host orchestration csrc/cuda/deform_conv_cuda.cu:158-691.
 *   dtype: 0 = fp32, 1 = fp16, 2 = bf16 (all buffers share it; interpolation runs in fp32).
 *   im       [B, C, H, W]
 *   offset   [B, dg*2*kh*kw, Ho, Wo]   channel 2*(i*kw+j) = dh, +1 = dw
 *   mask     [B, dg*kh*kw,   Ho, Wo]   NULL selects the v1 (un-modulated) operator
 *   col      [C*kh*kw, B*Ho*Wo]        row (c*kh+i)*kw+j, column (b*Ho+ho)*Wo+wo
 *   col2im accumulates into grad_im (caller zero-fills, as deform_conv_func.py:88 does);
 *   col2im_coord overwrites grad_offset (and grad_mask when mask != NULL).
 * ---------------------------------------------------------------------------------------- */
#define DETOPS_F32 0
#define DETOPS_F16 1
#define DETOPS_BF16 2

int detops_deformable_im2col(const void* im, const void* offset, const void* mask, void* col,
                             int dtype, int B, int C, int H, int W, int kh, int kw, int pad_h,
                             int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                             int deformable_group, detops_stream_t stream);

int detops_deformable_col2im(const void* col, const void* offset, const void* mask,
                             void* grad_im, int dtype, int B, int C, int H, int W, int kh,
                             int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                             int dil_h, int dil_w, int deformable_group,
                             detops_stream_t stream);

/* col2im without atomics on the data: the scatter pattern (shared by all channels of a deformable
 * group) is inverted once per call into per-pixel gather lists held in a caller workspace
 * (>= detops_deformable_col2im_workspace_bytes(...) bytes, device memory, any contents); each
 * gradient pixel is then summed in registers in a fixed order (deterministic).  A NULL / too
 * small workspace, or a shape beyond the 32-bit index plan (workspace_bytes() == 0), runs the
 * atomic scatter kernels of detops_deformable_col2im instead. */
size_t detops_deformable_col2im_workspace_bytes(int B, int C, int H, int W, int kh, int kw,
                                                int pad_h, int pad_w, int stride_h, int stride_w,
                                                int dil_h, int dil_w, int deformable_group);

int detops_deformable_col2im_ws(const void* col, const void* offset, const void* mask,
                                void* grad_im, int dtype, int B, int C, int H, int W, int kh,
                                int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                int dil_h, int dil_w, int deformable_group, void* workspace,
                                size_t workspace_bytes, detops_stream_t stream);

int detops_deformable_col2im_coord(const void* col, const void* im, const void* offset,
                                   const void* mask, void* grad_offset, void* grad_mask,
                                   int dtype, int B, int C, int H, int W, int kh, int kw,
                                   int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                   int dil_w, int deformable_group, detops_stream_t stream);

/* Fused deformable-convolution FORWARD: one implicit GEMM on the matrix cores
 * (v_mfma_f32_32x32x16_{f16,bf16}); the deformed operand tile is built in LDS, `columns` is never written.
 * Replaces the im2col + GEMM sequence of csrc/cuda/deform_conv_cuda.cu:228-245 / :520-560 for 16-bit storage,
 * convolution groups == 1 and (C / deformable_group) % 32 == 0; other shapes return DETOPS_EUNSUPPORTED
 * (workspace_bytes() == 0) and the caller runs detops_deformable_im2col + its GEMM.
 *   im [B,C,H,W], weight [Cout,C,kh,kw], offset / mask as above (mask NULL = v1), bias [Cout] or NULL,
 *   out [B,Cout,Ho,Wo]; all of `dtype`.  workspace: NHWC copy of im + tap-major copy of weight. */
size_t detops_deform_conv_forward_fused_workspace_bytes(int dtype, int B, int C, int H, int W, int Cout,
                                                        int kh, int kw, int pad_h, int pad_w, int stride_h,
                                                        int stride_w, int dil_h, int dil_w,
                                                        int deformable_group);

int detops_deform_conv_forward_fused(const void* im, const void* weight, const void* offset,
                                     const void* mask, const void* bias, void* out, int dtype, int B, int C,
                                     int H, int W, int Cout, int kh, int kw, int pad_h, int pad_w,
                                     int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group,
                                     void* workspace, size_t workspace_bytes, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Deformable position-sensitive ROI pooling — replaces _C.deform_psroi_pooling_forward /
 * _C.deform_psroi_pooling_backward
 *   reference: csrc/deform_pool.h:11-70, csrc/cuda/deform_pool_cuda.cu:38-87,
 *              csrc/cuda/deform_pool_kernel_cuda.cu:31-364 (CUDA-only there).
 *   data [N,C,H,W] with C >= output_dim * group_size^2, rois [K,5],
 *   trans [K, channels_trans, part_size, part_size] (NULL when no_trans; channels_trans = 2 *
 *   num_classes, output_dim must be a multiple of num_classes),
 *   out / top_count / out_grad [K, output_dim, pooled_size, pooled_size] (top_count holds the
 *   number of in-bounds samples of each bin as float, like the reference).
 *   backward: `zero_grads` != 0 zero-fills data_grad and trans_grad first (the reference's caller
 *   passes torch.zeros_like buffers, layers/dcn/deform_pool_func.py:72-73); the kernel accumulates.
 * ---------------------------------------------------------------------------------------- */
int detops_deform_psroi_pool_forward_f32(const float* data, const float* rois, const float* trans,
                                         float* out, float* top_count, int N, int C, int H, int W,
                                         int K, int channels_trans, int no_trans,
                                         float spatial_scale, int output_dim, int group_size,
                                         int pooled_size, int part_size, int sample_per_part,
                                         float trans_std, detops_stream_t stream);

int detops_deform_psroi_pool_backward_f32(const float* out_grad, const float* data,
                                          const float* rois, const float* trans,
                                          const float* top_count, float* data_grad,
                                          float* trans_grad, int N, int C, int H, int W, int K,
                                          int channels_trans, int no_trans, float spatial_scale,
                                          int output_dim, int group_size, int pooled_size,
                                          int part_size, int sample_per_part, float trans_std,
                                          int zero_grads, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training-target assignment on the device (extensions; the reference does these with ATen compositions
 * and host round trips: SURVEY.md section 8 rows f2 / f3).
 *
 * detops_match_boxes_f32 — IoU + Matcher fused (structures/boxlist_ops.py:53-89, modeling/matcher.py:42-112):
 *   gt_boxes [N, M, 4] xyxy fp32 (padded rows flagged 0 in gt_valid [N, M] uint8), boxes [K, 4] shared by
 *   the N images (boxes_batched = 0) or [N, K, 4]; matched_idxs [N, K] int64 = arg-max ground truth (first
 *   index among ties), -1 below low_threshold, -2 between the thresholds; with allow_low_quality_matches every
 *   box attaining some valid ground truth's maximum IoU keeps its arg-max.  The [M, K] matrix is never stored.
 *   workspace (only with the low-quality rule): >= detops_match_boxes_workspace_bytes(N, M).
 *
 * detops_sample_labels — BalancedPositiveNegativeSampler (modeling/balanced_positive_negative_sampler.py:19-68):
 *   labels [N, n] (label_dtype: fp32 or int64; >= 1 positive, 0 negative, < 0 ignored); per row a uniformly
 *   random subset of min(#pos, max_positives) positives and min(#neg, B - that) negatives, as 0/1 masks
 *   [N, n] and (optional, both or neither) a fixed-length list sampled_idx / sampled_valid [N, B], positives
 *   first.  Deterministic in (labels, seed); B <= 512.
 *
 * detops_mask_targets — mask-head targets (roi_heads/mask_head/loss.py:11-42 via
 *   structures/segmentation_mask.py:118-158): for ROI p crop instance mask_index[p] of masks [G, H, W] to the
 *   rounded, clamped box and resize bilinearly (align_corners = False) to M x M; integer masks truncate the
 *   interpolated value, bool masks map non-zero to 1.  out [P, M, M] fp32.
 * ---------------------------------------------------------------------------------------- */
#define DETOPS_LABEL_F32 0
#define DETOPS_LABEL_I64 1
#define DETOPS_MASK_U8 0
#define DETOPS_MASK_F32 1
#define DETOPS_MASK_BOOL 2

size_t detops_match_boxes_workspace_bytes(int N, int M);

int detops_match_boxes_f32(const float* gt_boxes, const uint8_t* gt_valid, const float* boxes,
                           int boxes_batched, int N, int M, int K, float high_threshold,
                           float low_threshold, int allow_low_quality_matches, int64_t* matched_idxs,
                           void* workspace, size_t workspace_bytes, detops_stream_t stream);

size_t detops_sample_labels_workspace_bytes(int N, int batch_size_per_image);

int detops_sample_labels(const void* labels, int label_dtype, int N, int n, int batch_size_per_image,
                         int max_positives, uint64_t seed, uint8_t* pos_mask, uint8_t* neg_mask,
                         int64_t* sampled_idx, uint8_t* sampled_valid, void* workspace,
                         size_t workspace_bytes, detops_stream_t stream);

/* the same with a device word mixed into the seed when the kernels run (NULL: exactly detops_sample_labels): for launches captured
 * into a HIP graph, whose arguments are frozen at capture */
int detops_sample_labels_dseed(const void* labels, int label_dtype, int N, int n, int batch_size_per_image,
                               int max_positives, uint64_t seed, const uint64_t* seed_dev, uint8_t* pos_mask,
                               uint8_t* neg_mask, int64_t* sampled_idx, uint8_t* sampled_valid, void* workspace,
                               size_t workspace_bytes, detops_stream_t stream);

int detops_mask_targets(const void* masks, int mask_dtype, const int64_t* mask_index, const float* boxes,
                        int G, int H, int W, int P, int M, float* out, detops_stream_t stream);

/* detops_rpn_decode_f32 — the box path of RPN proposal selection for one feature level
 *   (modeling/rpn/inference.py:75-110: gather of the pre-NMS top-k anchors' deltas, BoxCoder.decode —
 *   modeling/box_coder.py:61-95 —, clip_to_image(remove_empty=False) — structures/bounding_box.py:202-213 —,
 *   remove_small_boxes — structures/boxlist_ops.py:34-50 — as a mask): ~35 ATen launches per level in the reference.
 *   box_regression [N, 4A, H, W] fp32 (the head's own layout), topk_idx / topk_scores [N, k] (positions in the
 *   (y, x, a) order of permute_and_flatten, modeling/rpn/utils.py:9-13), anchors [A*H*W, 4] in that order,
 *   image_hw [N, 2] = (height, width) fp32.  Writes boxes [N][.][4] and scores [N][.] at the caller's column offset
 *   (row strides in elements), and this level's slice of the NMS input: nms_boxes [N*k, 4], nms_scores [N*k]
 *   (boxes smaller than min_size are moved far away with score -1), ok [N*k] uint8. */
int detops_rpn_decode_f32(const float* box_regression, const int64_t* topk_idx, const float* topk_scores,
                          const float* anchors, const float* image_hw, int N, int A, int H, int W, int k,
                          float wx, float wy, float ww, float wh, float bbox_xform_clip, float min_size,
                          float* boxes, int64_t boxes_row_stride, float* scores, int64_t scores_row_stride,
                          float* nms_boxes, float* nms_scores, uint8_t* ok, detops_stream_t stream);

/* detops_match_labels — Matcher output -> training labels (modeling/rpn/loss.py:70-88 and
 *   roi_heads/box_head/loss.py:56-72 of the reference: the clamp / gather / three masked assignments per image):
 *   matched [N, K] int64; gt_labels [N, M] int64 or NULL (NULL: every match labels 1, the RPN rule);
 *   valid [N, K] uint8 or NULL (anchor visibility / proposal validity: false -> -1).
 *   out [N, K]: float32 (out_dtype 0) or int64 (out_dtype 1);  >= 0 -> class, -1 -> 0, -2 -> -1. */
int detops_match_labels(const int64_t* matched, const int64_t* gt_labels, const uint8_t* valid, int N, int K, int M,
                        int out_dtype, void* out, detops_stream_t stream);

/* detops_roi_head_targets_f32 — the box head's sampled slots (roi_heads/box_head/loss.py:56-110: labels, BoxCoder.encode of
 *   the matched ground truth — modeling/box_coder.py:27-51 —, the per-image indexing by the sampled positions):
 *   boxes [N, K, 4], matched [N, K], gt_boxes [N, M, 4], gt_labels [N, M], valid [N, K] or NULL, idx / slot_valid [N, B]
 *   (detops_sample_labels' slot list), objectness [N, K] or NULL -> per slot: box [N, B, 4], label [N, B] int64 (-1 for an
 *   unfilled slot), regression target [N, B, 4], matched index [N, B], objectness [N, B] (out_objectness may be NULL). */
int detops_roi_head_targets_f32(const float* boxes, const int64_t* matched, const float* gt_boxes, const int64_t* gt_labels,
                                const uint8_t* valid, const int64_t* idx, const uint8_t* slot_valid, const float* objectness,
                                int N, int K, int M, int B, float wx, float wy, float ww, float wh, float* out_boxes,
                                int64_t* out_labels, float* out_regression_targets, int64_t* out_matched,
                                float* out_objectness, detops_stream_t stream);

/* ---- ROI-head losses: value and gradient in one pass (csrc/head_loss.hip) ------------------------------------------------
 * detops_fastrcnn_loss_f32 — FastRCNNLossComputation.__call__ (roi_heads/box_head/loss.py:140-193 of the reference):
 *   class_logits [R, C], box_regression [R, D] (D = 4C, or >= 8 with cls_agnostic: columns 4..7), labels [R] int64
 *   (-1 = not sampled), regression_targets [R, 4].  losses2 = {cross_entropy(sum over labels >= 0), smooth_l1(beta) summed
 *   over labels > 0 on the class's four columns} / max(#(labels >= 0), 1).  grad_logits [R, C] and grad_box [R, D] receive
 *   d losses2[0] / d class_logits and d losses2[1] / d box_regression (every element written).
 * detops_mask_loss_f32 — MaskRCNNLossComputation.__call__ (roi_heads/mask_head/loss.py:113-143): mask_logits [P, C, M, M],
 *   labels [P] int64 (> 0 = positive), mask_targets [P, M, M] -> loss1 = mean BCE-with-logits between each positive ROI's
 *   class plane and its target (0 without positives); grad_logits [P, C, M, M] fully written.
 * detops_head_loss_backward_f32 — grad_a *= upstream_a[0], grad_b *= upstream_b[0] (device scalars), in place, one launch. */
size_t detops_fastrcnn_loss_workspace_bytes(int R);
int detops_fastrcnn_loss_f32(const float* class_logits, const float* box_regression, const int64_t* labels,
                             const float* regression_targets, int R, int C, int D, int cls_agnostic, float beta,
                             float* grad_logits, float* grad_box, float* losses2, void* workspace, size_t workspace_bytes,
                             detops_stream_t stream);
size_t detops_mask_loss_workspace_bytes(int P);
int detops_mask_loss_f32(const float* mask_logits, const int64_t* labels, const float* mask_targets, int P, int C, int M,
                         float* grad_logits, float* loss1, void* workspace, size_t workspace_bytes, detops_stream_t stream);
int detops_head_loss_backward_f32(float* grad_a, int64_t count_a, const float* upstream_a, float* grad_b, int64_t count_b,
                                  const float* upstream_b, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused FrozenBatchNorm2d affine (+ residual) (+ ReLU) — the elementwise tail of every backbone
 * convolution: layers/batch_norm.py:19-31 (`x * scale + bias`), then `F.relu_`, and in the
 * bottleneck tail `out += identity; relu` (modeling/backbone/resnet.py:343-366).
 *   x, residual (nullable), y: [N, C, HW] contiguous (NCHW), dtype code as above (fp32 arithmetic)
 *   scale, bias: fp32 [C]   (scale = weight * rsqrt(running_var), bias = bias - running_mean * scale)
 *   forward : y = [relu]( x * scale[c] + bias[c] [+ residual] )
 *   backward: g = relu ? (y > 0 ? grad_y : 0) : grad_y;  grad_x = g * scale[c];
 *             grad_residual (nullable) = g
 * ---------------------------------------------------------------------------------------- */
int detops_frozen_bn_act_forward(const void* x, const float* scale, const float* bias,
                                 const void* residual, void* y, int dtype, int N, int C, int HW,
                                 int relu, detops_stream_t stream);

int detops_frozen_bn_act_backward(const void* grad_y, const void* y, const float* scale,
                                  void* grad_x, void* grad_residual, int dtype, int N, int C,
                                  int HW, int relu, detops_stream_t stream);

/* The same pair for channels-last activations ([rows = N*H*W, C], the channel the fastest index — what MIOpen's NHWC
 * convolution kernels read and write without a layout transpose): same arithmetic, same rounding. */
int detops_frozen_bn_act_forward_nhwc(const void* x, const float* scale, const float* bias,
                                      const void* residual, void* y, int dtype, int64_t rows, int C,
                                      int relu, detops_stream_t stream);

int detops_frozen_bn_act_backward_nhwc(const void* grad_y, const void* y, const float* scale,
                                       void* grad_x, void* grad_residual, int dtype, int64_t rows, int C,
                                       int relu, detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bias (+ ReLU) behind a channels-last convolution (csrc/bias_act.hip) — `conv + bias [+ relu]` of the detector's biased
 * convolutions (reference modeling/backbone/fpn.py:30-40, modeling/rpn/rpn.py:61-76, roi_heads/mask_head): the forward is
 * detops_frozen_bn_act_forward_nhwc with scale 1; the backward is one pass over the gradient:
 *   g = relu ? (y > 0 ? grad_y : 0) : grad_y ;  grad_x = g (may alias grad_y when relu = 0) ;  grad_bias[c] = sum_rows g[., c]
 * x, y, grad_* are [rows = N*H*W, C] fp32 (16-byte aligned); C must satisfy detops_bias_act_supported (a multiple of 4
 * dividing 1024); deterministic (per-workgroup partial sums added in index order: `workspace` of
 * detops_bias_act_backward_workspace_bytes(rows, C) bytes), no atomics.
 * ---------------------------------------------------------------------------------------- */
int detops_bias_act_supported(int C);
size_t detops_bias_act_backward_workspace_bytes(int64_t rows, int C);
int detops_bias_act_backward_nhwc_f32(const float* grad_y, const float* y, float* grad_x, float* grad_bias, int64_t rows,
                                      int C, int relu, void* workspace, size_t workspace_bytes, detops_stream_t stream);
/* the same with fp32 / fp16 / bf16 storage (dtype code; grad_bias and the partial sums stay fp32): the autocast configurations */
int detops_bias_act_backward_nhwc(const void* grad_y, const void* y, void* grad_x, float* grad_bias, int dtype, int64_t rows,
                                  int C, int relu, void* workspace, size_t workspace_bytes, detops_stream_t stream);
/* out[c] = sum_r x[r, c] of a row-major [rows, C] fp32 matrix, any C <= 256 (bias gradients of the channel counts the fused
 * pass does not serve: the RPN's 3 / 12, the mask logits' 81); deterministic; workspace of detops_column_sum_workspace_bytes. */
size_t detops_column_sum_workspace_bytes(int64_t rows, int C);
int detops_column_sum_f32(const float* x, float* out, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                          detops_stream_t stream);
int detops_column_sum(const void* x, float* out, int dtype, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                      detops_stream_t stream);

/* RPN loss in one pass over the head outputs (reference modeling/rpn/loss.py:92-127 with
 * modeling/box_coder.py:27-51 and modeling/rpn/utils.py:9-45): objectness = BCE-with-logits over the sampled
 * anchors, box = smooth-L1(beta) against BoxCoder.encode(matched gt, anchor) over the sampled positives, both
 * divided by max(#sampled, 1).
 *   objectness[l] [N, A, H_l, W_l], box_regression[l] [N, 4A, H_l, W_l]: the per-level head outputs as written by
 *   the heads (HOST arrays of device pointers); anchors [T, 4], T = A * sum_l H_l W_l, ordered level, y, x, a;
 *   matched_idxs [N, T] (Matcher output), pos_mask / neg_mask [N, T] (sampler output), gt_boxes [N, M, 4].
 *   grad_*[l]: same shapes as the inputs, OVERWRITTEN with d(sum of the per-anchor losses) / d(input);
 *   losses3 (device, 3 floats) = {objectness loss, box loss, 1 / max(#sampled, 1)}.
 * detops_rpn_loss_backward_f32 scales the stored gradients in place by upstream * losses3[2] (device scalars). */
size_t detops_rpn_loss_workspace_bytes(void);
int detops_rpn_loss_f32(const float* const* objectness_host, const float* const* box_regression_host,
                        const int* H_host, const int* W_host, int num_levels, int anchors_per_location,
                        const float* anchors, const int64_t* matched_idxs, const uint8_t* pos_mask,
                        const uint8_t* neg_mask, const float* gt_boxes, int N, int M, int T, float beta,
                        const float* weights4_host, float* const* grad_objectness_host,
                        float* const* grad_box_regression_host, float* losses3, void* workspace,
                        size_t workspace_bytes, detops_stream_t stream);
int detops_rpn_loss_backward_f32(float* const* grad_objectness_host, float* const* grad_box_regression_host,
                                 const int* H_host, const int* W_host, int num_levels, int anchors_per_location,
                                 int N, int T, const float* upstream_objectness, const float* upstream_box,
                                 const float* inv_count, detops_stream_t stream);

/* ----------------------------------------------------------------------------------------
 * CPU branch — HOST pointers, no stream.  The reference `_C` serves exactly two operators for CPU tensors
 * (csrc/nms.h:19-27 -> csrc/cpu/nms_cpu.cpp, csrc/ROIAlign.h:19-24 -> csrc/cpu/ROIAlign_cpu.cpp) and raises
 * "Not implemented on the CPU" for the rest; so does this library.  These are NOT a fallback of the device entry
 * points (which never call them and fail when there is no GPU); they exist so that a caller holding CPU tensors
 * gets the reference's behaviour from the drop-in.
 *   detops_nms_cpu_f32: keep[n] receives the kept ORIGINAL indices in ascending order, *num_keep their count.
 * ---------------------------------------------------------------------------------------- */
int detops_nms_cpu_f32(const float* boxes, const float* scores, int n, float iou_threshold,
                       int64_t* keep, int32_t* num_keep);
int detops_roi_align_forward_cpu_f32(const float* input, const float* rois, float* output, int N, int C,
                                     int H, int W, int K, int PH, int PW, float spatial_scale,
                                     int sampling_ratio);

/* ------------------------------------------------------------------------------------------
 * Deformable convolution, channels-last pipeline (extension; what the layers use for the model's shapes).
 *   reference: the same operators as above (csrc/cuda/deform_conv_cuda.cu:158-691), restructured so that every
 *   per-sampling-point operand is channel-fastest and the GEMMs are plain library GEMMs on those layouts:
 *     xT [B, H*W, C] = NHWC copy of the input                       detops_nchw_to_nhwc
 *     colT [B*Ho*Wo, kh*kw, C] deformed (and modulated) columns     detops_deformable_im2col_nhwc
 *     offset / mask gradients from colsG_T [B*Ho*Wo, kh*kw, C]      detops_deformable_coord_nhwc
 *     S_T [B*H*W, kh*kw, Cout] = transposed sampling of gT [B, Ho*Wo, Cout] (the NHWC output gradient):
 *     grad_input[b] = W2T [C, kh*kw*Cout] x S_T[b]^T                 detops_deformable_transposed_sample
 *   detops_deformable_nhwc_supported: 1 when these kernels serve the shape (deformable_group == 1, C and Cout
 *   16-byte-vector counts that are powers of two in [16, 256]); otherwise use the reference-layout entry points.
 *   dtype: DETOPS_F32 / F16 / BF16; offset [B, 2*kh*kw, Ho, Wo] and mask [B, kh*kw, Ho, Wo] (NULL = v1) as above.
 * ---------------------------------------------------------------------------------------- */
int detops_nchw_to_nhwc(const void* in, void* out, int dtype, int B, int C, int HW, detops_stream_t stream);
int detops_deformable_nhwc_supported(int dtype, int C, int Cout, int deformable_group);
int detops_deformable_im2col_nhwc(const void* xT, const void* offset, const void* mask, void* colT, int dtype,
                                  int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
                                  int stride_w, int dil_h, int dil_w, int deformable_group, detops_stream_t stream);
int detops_deformable_coord_nhwc(const void* colsG_T, const void* xT, const void* offset, const void* mask,
                                 void* grad_offset, void* grad_mask, int dtype, int B, int C, int H, int W, int kh,
                                 int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                 int deformable_group, detops_stream_t stream);
size_t detops_deformable_transposed_sample_workspace_bytes(int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                                 int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group);
int detops_deformable_transposed_sample(const void* gT, const void* offset, const void* mask, void* S_T, int dtype, int B, int C,
                              int H, int W, int Cout, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                              int dil_h, int dil_w, int deformable_group, void* workspace, size_t workspace_bytes,
                              detops_stream_t stream);
/* grad_in_T [B, H*W, C] = col2im of the channel-fastest column gradient colsG_T [B*Ho*Wo, kh*kw, C] as a gather over the same
 * inverted index (reference col2im: csrc/cuda/deform_conv_kernel_cuda.cu:353-413, an atomic scatter): no S_T, no second GEMM.
 * workspace: detops_deformable_transposed_sample_workspace_bytes(...). */
int detops_deformable_col2im_nhwc(const void* colsG_T, const void* offset, const void* mask, void* grad_in_T, int dtype, int B,
                                  int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                  int dil_h, int dil_w, int deformable_group, void* workspace, size_t workspace_bytes,
                                  detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The data-parallel step's bucket kernels (csrc/optim.hip) — replace, per gradient bucket of engine/ddp_step.py, the
 * multi-tensor copy into the bucket and torch.optim.SGD's update (reference: tools/train_net.py:45-54 wraps the model in
 * DistributedDataParallel, engine/trainer.py:93-99 `optimizer.step()`, solver/build.py:7-20 the two hyper-parameter
 * groups).  A bucket is three flat fp32 arrays of one layout: parameters, gradients, momentum.
 *   detops_pack_f32: dst[dst_offsets[i] .. + counts[i]) = srcs[i][0 .. counts[i]) for i < n <= detops_pack_max_tensors();
 *     srcs / counts / dst_offsets are HOST arrays (the pointers travel in the kernel arguments), srcs[i] device pointers.
 *   detops_sgd_momentum_flat_f32: for every element  g' = g + wd * p;  m = momentum * m + g';  p = p - lr * m
 *     (torch.optim.SGD with dampening 0, no Nesterov; the momentum array starts as zeros, which makes the first step
 *     m = g' like torch's), with (lr, wd) = the weights' pair on [0, split) and the biases' pair on [split, n);
 *     split a multiple of 4, all three arrays 16-byte aligned.
 * ---------------------------------------------------------------------------------------- */
int detops_pack_max_tensors(void);
int detops_pack_f32(const void* const* srcs, const int64_t* counts, const int64_t* dst_offsets, int n, float* dst,
                    detops_stream_t stream);
int detops_sgd_momentum_flat_f32(float* params, const float* grads, float* momentum_buf, int64_t n, int64_t split,
                                 float lr_weights, float wd_weights, float lr_biases, float wd_biases, float momentum,
                                 detops_stream_t stream);
/* Measurement tool (not on the training path): `workgroups` workgroups of 1024 threads hold their CUs' wave slots for
 * `microseconds` on `stream` — a one-GPU stand-in for the CUs a ring all-reduce kernel occupies beside the compute stream at
 * N > 1 (engine/ddp_step.py: DETOPS_DDP_STANDIN, profiles/r06_ddp_contention.txt). */
int detops_debug_occupy(int workgroups, int microseconds, detops_stream_t stream);
/* diagnosis: wall-clock timeline (10 ns ticks) of segment 0 in the last single-launch NMS that ran with tuning nms_debug & 4
 * (csrc/nms.hip g_nms_timeline); host_out: n <= 6464 words.  Synchronises the device. */
int detops_debug_nms_timeline(int64_t* host_out, int n);

/* ------------------------------------------------------------------------------------------
 * FPN top-down step (csrc/fpn_topdown.hip) — replaces the pair `F.interpolate(last_inner, mode="nearest")` +
 * `inner_lateral + inner_top_down` of reference modeling/backbone/fpn.py:59-64 (and its backward: identity towards the
 * lateral, block sum towards the coarser map).  NCHW planes (= N * C), lateral / out [planes, H, W], top [planes, h, w];
 * dtype DETOPS_F32 / F16 / BF16 for all tensors of a call, fp32 arithmetic; nearest index = ATen's
 * min(int(floorf(dst * (float)in / out)), in - 1).  backward: grad_top [planes, h, w] is OVERWRITTEN.
 * ---------------------------------------------------------------------------------------- */
int detops_fpn_topdown_forward(const void* lateral, const void* top, void* out, int dtype, int planes, int H, int W, int h, int w,
                               detops_stream_t stream);
int detops_fpn_topdown_backward(const void* grad_out, void* grad_top, int dtype, int planes, int H, int W, int h, int w,
                                detops_stream_t stream);
/* channels-last form: lateral / out / grad_out [N, H, W, C], top / grad_top [N, h, w, C] */
int detops_fpn_topdown_forward_nhwc(const void* lateral, const void* top, void* out, int dtype, int N, int C, int H, int W,
                                    int h, int w, detops_stream_t stream);
int detops_fpn_topdown_backward_nhwc(const void* grad_out, void* grad_top, int dtype, int N, int C, int H, int W, int h, int w,
                                     detops_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Double precision (csrc/f64_ops.hip, csrc/cpu_branch.hip) — the reference dispatches these operators over
 * AT_DISPATCH_FLOATING_TYPES, i.e. float AND double (csrc/cuda/ROIAlign_cuda.cu:283,329, ROIPool_cuda.cu:137,185,
 * SigmoidFocalLoss_cuda.cu:129,173, csrc/cpu/ROIAlign_cpu.cpp:242, csrc/cpu/nms_cpu.cpp:71).  Same argument meaning as the
 * _f32 entry points; plain one-thread-per-element kernels with the reference's arithmetic in double (API completeness: no
 * configuration of the training path computes in double).  detops_nms_sorted_f64 takes boxes already sorted by descending
 * score and returns a byte per sorted box (1 = kept); the two *_cpu_f64 entry points are host code for CPU tensors.
 * ---------------------------------------------------------------------------------------- */
int detops_roi_align_forward_f64(const double* input, const double* rois, double* output, int N, int C, int H, int W, int K,
                                 int PH, int PW, float spatial_scale, int sampling_ratio, detops_stream_t stream);
int detops_roi_align_backward_f64(const double* grad_out, const double* rois, double* grad_in, int N, int C, int H, int W,
                                  int K, int PH, int PW, float spatial_scale, int sampling_ratio, int zero_grad_in,
                                  detops_stream_t stream);
int detops_roi_pool_forward_f64(const double* input, const double* rois, double* output, int32_t* argmax, int N, int C, int H,
                                int W, int K, int PH, int PW, float spatial_scale, detops_stream_t stream);
int detops_roi_pool_backward_f64(const double* grad_out, const double* rois, const int32_t* argmax, double* grad_in, int N,
                                 int C, int H, int W, int K, int PH, int PW, int zero_grad_in, detops_stream_t stream);
int detops_sigmoid_focal_loss_forward_f64(const double* logits, const int32_t* targets, double* losses, int num_rows,
                                          int num_classes, float gamma, float alpha, detops_stream_t stream);
int detops_sigmoid_focal_loss_backward_f64(const double* logits, const int32_t* targets, const double* d_losses,
                                           double* d_logits, int num_rows, int num_classes, float gamma, float alpha,
                                           detops_stream_t stream);
size_t detops_nms_sorted_f64_workspace_bytes(int n);
int detops_nms_sorted_f64(const double* sorted_boxes, int n, float iou_threshold, unsigned char* keep_sorted, void* workspace,
                          size_t workspace_bytes, detops_stream_t stream);
int detops_roi_align_forward_cpu_f64(const double* input, const double* rois, double* output, int N, int C, int H, int W,
                                     int K, int PH, int PW, float spatial_scale, int sampling_ratio);
int detops_nms_cpu_f64(const double* boxes, const double* scores, int n, float iou_threshold, int64_t* keep,
                       int32_t* num_keep);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* DETOPS_H_ */
