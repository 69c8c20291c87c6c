/*
 * detops_oracle.c — CPU restatement of the reference algorithms for the detection-head hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (maskrcnn-benchmark_amd/) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker.
 *
 * Every function restates one reference routine in plain C (single thread, fp32 arithmetic in the
 * reference's own evaluation order) and cites the reference file:line it follows (paths relative
 * to /root/reference/maskrcnn_benchmark/csrc/).  Compile with -ffp-contract=off so that no
 * multiply-add is fused where the reference (x86-64 build, default flags, no FMA) did not fuse.
 *
 * Parity pinning:
 *   - oracle_nms_f32 and oracle_roi_align_forward_f32 are pinned bit-for-bit against the
 *     reference's own CPU kernels compiled from /root/reference (oracle/_ref, built by
 *     oracle/build_ref.py) and against the reference's golden vectors tests/test_nms.py
 *     (tests/golden/nms_reference_tests.npz).
 *   - ROIAlign backward, ROIPool, SigmoidFocalLoss and the deformable-conv routines exist only as
 *     CUDA in the reference and no reference test pins their values: PARITY UNPINNED for those —
 *     they are source-faithful restatements of the .cu files, cross-checked against independent
 *     PyTorch autograd formulations in tests/.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* std::max / std::min exactly as <algorithm> defines them (used by cpu/nms_cpu.cpp:49-55 and
 * cpu/ROIAlign_cpu.cpp:156-157): max(a,b) = (a < b) ? b : a;  min(a,b) = (b < a) ? b : a. */
static inline float std_maxf(float a, float b) { return (a < b) ? b : a; }
static inline float std_minf(float a, float b) { return (b < a) ? b : a; }

/* ============================================================================================
 * ROIAlign forward — cpu/ROIAlign_cpu.cpp:17-111 (pre_calc_for_bilinear_interpolate) and
 * :113-219 (ROIAlignForward_cpu_kernel), T = float.
 * ========================================================================================== */
typedef struct {
  int pos1, pos2, pos3, pos4;
  float w1, w2, w3, w4;
} precalc_t;

static void pre_calc_bilinear(int height, int width, int pooled_height, int pooled_width,
                              int iy_upper, int ix_upper, float roi_start_h, float roi_start_w,
                              float bin_size_h, float bin_size_w, int roi_bin_grid_h,
                              int roi_bin_grid_w, precalc_t* pre_calc) {
  int idx = 0;
  for (int ph = 0; ph < pooled_height; ph++) {
    for (int pw = 0; pw < pooled_width; pw++) {
      for (int iy = 0; iy < iy_upper; iy++) {
        /* ROIAlign_cpu.cpp:36-38 */
        const float yy = roi_start_h + ph * bin_size_h +
                         (float)(iy + .5f) * bin_size_h / (float)(roi_bin_grid_h);
        for (int ix = 0; ix < ix_upper; ix++) {
          const float xx = roi_start_w + pw * bin_size_w +
                           (float)(ix + .5f) * bin_size_w / (float)(roi_bin_grid_w);
          float x = xx, y = yy;
          precalc_t pc;
          /* :47-61 out of the map: contributes zero */
          if (y < -1.0 || y > height || x < -1.0 || x > width) {
            memset(&pc, 0, sizeof(pc));
            pre_calc[idx++] = pc;
            continue;
          }
          if (y <= 0) y = 0; /* :63-68 */
          if (x <= 0) x = 0;
          int y_low = (int)y, x_low = (int)x, y_high, x_high;
          if (y_low >= height - 1) { /* :75-87 */
            y_high = y_low = height - 1;
            y = (float)y_low;
          } else {
            y_high = y_low + 1;
          }
          if (x_low >= width - 1) {
            x_high = x_low = width - 1;
            x = (float)x_low;
          } else {
            x_high = x_low + 1;
          }
          float ly = y - y_low, lx = x - x_low;
          /* :91 `T hy = 1. - ly` — double literal: the subtraction is done in double */
          float hy = (float)(1. - ly), hx = (float)(1. - lx);
          pc.w1 = hy * hx; pc.w2 = hy * lx; pc.w3 = ly * hx; pc.w4 = ly * lx; /* :92 */
          pc.pos1 = y_low * width + x_low;  /* :95-99 */
          pc.pos2 = y_low * width + x_high;
          pc.pos3 = y_high * width + x_low;
          pc.pos4 = y_high * width + x_high;
          pre_calc[idx++] = pc;
        }
      }
    }
  }
}

ORACLE_API void oracle_roi_align_forward_f32(const float* bottom_data, const float* bottom_rois,
                                             float* top_data, int N, int channels, int height,
                                             int width, int n_rois, int pooled_height,
                                             int pooled_width, float spatial_scale,
                                             int sampling_ratio) {
  (void)N;
  for (int n = 0; n < n_rois; n++) { /* :134 */
    int index_n = n * channels * pooled_width * pooled_height;
    const float* r = bottom_rois + n * 5; /* roi_cols = 5, :128 */
    int roi_batch_ind = (int)r[0];
    r++;
    float roi_start_w = r[0] * spatial_scale; /* :146-149, no rounding */
    float roi_start_h = r[1] * spatial_scale;
    float roi_end_w = r[2] * spatial_scale;
    float roi_end_h = r[3] * spatial_scale;
    float roi_width = std_maxf(roi_end_w - roi_start_w, 1.f); /* :156-157 */
    float roi_height = std_maxf(roi_end_h - roi_start_h, 1.f);
    float bin_size_h = roi_height / (float)pooled_height; /* :158-159 */
    float bin_size_w = roi_width / (float)pooled_width;
    int roi_bin_grid_h = (sampling_ratio > 0) ? sampling_ratio
                                              : (int)ceilf(roi_height / pooled_height); /* :162-166 */
    int roi_bin_grid_w =
        (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_width);
    const float count = (float)(roi_bin_grid_h * roi_bin_grid_w); /* :169 */

    size_t npc = (size_t)roi_bin_grid_h * roi_bin_grid_w * pooled_width * pooled_height;
    precalc_t* pre_calc = (precalc_t*)malloc(sizeof(precalc_t) * (npc ? npc : 1));
    pre_calc_bilinear(height, width, pooled_height, pooled_width, roi_bin_grid_h,
                      roi_bin_grid_w, roi_start_h, roi_start_w, bin_size_h, bin_size_w,
                      roi_bin_grid_h, roi_bin_grid_w, pre_calc);

    for (int c = 0; c < channels; c++) { /* :190-217 */
      int index_n_c = index_n + c * pooled_width * pooled_height;
      const float* offset_bottom_data =
          bottom_data + (size_t)(roi_batch_ind * channels + c) * height * width;
      int pci = 0;
      for (int ph = 0; ph < pooled_height; ph++) {
        for (int pw = 0; pw < pooled_width; pw++) {
          float output_val = 0.f;
          for (int iy = 0; iy < roi_bin_grid_h; iy++) {
            for (int ix = 0; ix < roi_bin_grid_w; ix++) {
              precalc_t pc = pre_calc[pci];
              output_val += pc.w1 * offset_bottom_data[pc.pos1] +
                            pc.w2 * offset_bottom_data[pc.pos2] +
                            pc.w3 * offset_bottom_data[pc.pos3] +
                            pc.w4 * offset_bottom_data[pc.pos4];
              pci++;
            }
          }
          output_val /= count;
          top_data[index_n_c + ph * pooled_width + pw] = output_val;
        }
      }
    }
    free(pre_calc);
  }
}

/* ============================================================================================
 * ROIAlign backward — cuda/ROIAlign_cuda.cu:125-175 (bilinear_interpolate_gradient) and
 * :177-254 (RoIAlignBackwardFeature); grad_input zero-initialised as at :316.
 * The CUDA kernel's atomicAdd order is unspecified; this restatement adds in ascending flat
 * index order (n, c, ph, pw, iy, ix, tap).  `acc64` != 0 accumulates in double and rounds once
 * at the end — the order-independent reference used to bound re-association noise.
 * ========================================================================================== */
static void bilinear_interpolate_gradient(int height, int width, float y, float x, float* w1,
                                          float* w2, float* w3, float* w4, int* x_low,
                                          int* x_high, int* y_low, int* y_high) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) { /* :133-138 */
    *w1 = *w2 = *w3 = *w4 = 0.f;
    *x_low = *x_high = *y_low = *y_high = -1;
    return;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  *y_low = (int)y;
  *x_low = (int)x;
  if (*y_low >= height - 1) {
    *y_high = *y_low = height - 1;
    y = (float)*y_low;
  } else {
    *y_high = *y_low + 1;
  }
  if (*x_low >= width - 1) {
    *x_high = *x_low = width - 1;
    x = (float)*x_low;
  } else {
    *x_high = *x_low + 1;
  }
  float ly = y - *y_low, lx = x - *x_low;
  float hy = (float)(1. - ly), hx = (float)(1. - lx);
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx; /* :172 */
}

ORACLE_API void oracle_roi_align_backward_f32(const float* top_diff, const float* bottom_rois,
                                              float* bottom_diff, int N, int channels,
                                              int height, int width, int num_rois,
                                              int pooled_height, int pooled_width,
                                              float spatial_scale, int sampling_ratio,
                                              int acc64) {
  size_t total_in = (size_t)N * channels * height * width;
  double* acc = NULL;
  if (acc64) acc = (double*)calloc(total_in ? total_in : 1, sizeof(double));
  memset(bottom_diff, 0, total_in * sizeof(float)); /* :316 at::zeros */
  long nthreads = (long)num_rois * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; index++) {
    int pw = index % pooled_width; /* :186-189 */
    int ph = (index / pooled_width) % pooled_height;
    int c = (index / pooled_width / pooled_height) % channels;
    int n = index / pooled_width / pooled_height / channels;
    const float* r = bottom_rois + n * 5;
    int roi_batch_ind = (int)r[0];
    float roi_start_w = r[1] * spatial_scale;
    float roi_start_h = r[2] * spatial_scale;
    float roi_end_w = r[3] * spatial_scale;
    float roi_end_h = r[4] * spatial_scale;
    float roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
    float roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
    float bin_size_h = roi_height / (float)pooled_height;
    float bin_size_w = roi_width / (float)pooled_width;
    size_t plane = (size_t)(roi_batch_ind * channels + c) * height * width;
    const float top_diff_this_bin = top_diff[index]; /* :211-213 */
    int roi_bin_grid_h =
        (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_height);
    int roi_bin_grid_w =
        (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_width);
    const float count = (float)(roi_bin_grid_h * roi_bin_grid_w);
    for (int iy = 0; iy < roi_bin_grid_h; iy++) {
      const float y = roi_start_h + ph * bin_size_h +
                      (float)(iy + .5f) * bin_size_h / (float)(roi_bin_grid_h); /* :224 */
      for (int ix = 0; ix < roi_bin_grid_w; ix++) {
        const float x = roi_start_w + pw * bin_size_w +
                        (float)(ix + .5f) * bin_size_w / (float)(roi_bin_grid_w);
        float w1, w2, w3, w4;
        int x_low, x_high, y_low, y_high;
        bilinear_interpolate_gradient(height, width, y, x, &w1, &w2, &w3, &w4, &x_low,
                                      &x_high, &y_low, &y_high);
        float g1 = top_diff_this_bin * w1 / count; /* :239-242 */
        float g2 = top_diff_this_bin * w2 / count;
        float g3 = top_diff_this_bin * w3 / count;
        float g4 = top_diff_this_bin * w4 / count;
        if (x_low >= 0 && x_high >= 0 && y_low >= 0 && y_high >= 0) { /* :244-250 */
          if (acc64) {
            acc[plane + y_low * width + x_low] += g1;
            acc[plane + y_low * width + x_high] += g2;
            acc[plane + y_high * width + x_low] += g3;
            acc[plane + y_high * width + x_high] += g4;
          } else {
            bottom_diff[plane + y_low * width + x_low] += g1;
            bottom_diff[plane + y_low * width + x_high] += g2;
            bottom_diff[plane + y_high * width + x_low] += g3;
            bottom_diff[plane + y_high * width + x_high] += g4;
          }
        }
      }
    }
  }
  if (acc64) {
    for (size_t i = 0; i < total_in; i++) bottom_diff[i] = (float)acc[i];
    free(acc);
  }
}

/* ============================================================================================
 * ROIPool forward / backward — cuda/ROIPool_cuda.cu:16-77, :79-108.
 * ========================================================================================== */
ORACLE_API void oracle_roi_pool_forward_f32(const float* bottom_data, const float* bottom_rois,
                                            float* top_data, int32_t* argmax_data, int N,
                                            int channels, int height, int width, int num_rois,
                                            int pooled_height, int pooled_width,
                                            float spatial_scale) {
  (void)N;
  long nthreads = (long)num_rois * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; index++) {
    int pw = index % pooled_width;
    int ph = (index / pooled_width) % pooled_height;
    int c = (index / pooled_width / pooled_height) % channels;
    int n = index / pooled_width / pooled_height / channels;
    const float* r = bottom_rois + n * 5;
    int roi_batch_ind = (int)r[0];
    /* :30-33 `round` of a float expression, then implicit float->int conversion */
    int roi_start_w = (int)roundf(r[1] * spatial_scale);
    int roi_start_h = (int)roundf(r[2] * spatial_scale);
    int roi_end_w = (int)roundf(r[3] * spatial_scale);
    int roi_end_h = (int)roundf(r[4] * spatial_scale);
    int roi_width = roi_end_w - roi_start_w + 1; /* :36-37 */
    if (roi_width < 1) roi_width = 1;
    int roi_height = roi_end_h - roi_start_h + 1;
    if (roi_height < 1) roi_height = 1;
    float bin_size_h = (float)roi_height / (float)pooled_height; /* :38-41 */
    float bin_size_w = (float)roi_width / (float)pooled_width;
    int hstart = (int)floorf((float)ph * bin_size_h); /* :43-50 */
    int wstart = (int)floorf((float)pw * bin_size_w);
    int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
    int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
#define CLIP(v, lo, hi) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
    hstart = CLIP(hstart + roi_start_h, 0, height); /* :53-56 */
    hend = CLIP(hend + roi_start_h, 0, height);
    wstart = CLIP(wstart + roi_start_w, 0, width);
    wend = CLIP(wend + roi_start_w, 0, width);
#undef CLIP
    int is_empty = (hend <= hstart) || (wend <= wstart);
    float maxval = is_empty ? 0 : -FLT_MAX; /* :60 */
    int maxidx = -1;
    const float* offset_bottom_data =
        bottom_data + (size_t)(roi_batch_ind * channels + c) * height * width;
    for (int h = hstart; h < hend; ++h) {
      for (int w = wstart; w < wend; ++w) {
        int bottom_index = h * width + w;
        if (offset_bottom_data[bottom_index] > maxval) { /* :68 strict > */
          maxval = offset_bottom_data[bottom_index];
          maxidx = bottom_index;
        }
      }
    }
    top_data[index] = maxval;
    argmax_data[index] = maxidx;
  }
}

ORACLE_API void oracle_roi_pool_backward_f32(const float* top_diff, const float* bottom_rois,
                                             const int32_t* argmax_data, float* bottom_diff,
                                             int N, int channels, int height, int width,
                                             int num_rois, int pooled_height,
                                             int pooled_width) {
  memset(bottom_diff, 0, (size_t)N * channels * height * width * sizeof(float));
  long nthreads = (long)num_rois * channels * pooled_height * pooled_width;
  for (long index = 0; index < nthreads; index++) {
    int c = (index / pooled_width / pooled_height) % channels;
    int n = index / pooled_width / pooled_height / channels;
    int roi_batch_ind = (int)bottom_rois[n * 5];
    size_t bottom_offset = (size_t)(roi_batch_ind * channels + c) * height * width; /* :94 */
    int argmax = argmax_data[index];
    if (argmax != -1) bottom_diff[bottom_offset + argmax] += top_diff[index]; /* :100-105 */
  }
}

/* ============================================================================================
 * NMS — cpu/nms_cpu.cpp:5-65 (nms_cpu_kernel<float>).
 * Sort: at::sort(descending) there; here a stable merge sort on (score desc) so equal scores
 * keep ascending index order (tie order is not pinned by the reference; tests use distinct
 * scores when comparing against the compiled reference).
 * Returns the number of kept boxes; keep[] holds ascending original indices (at::nonzero, :64).
 * ========================================================================================== */
static void merge_sort_desc(const float* s, int64_t* idx, int64_t* tmp, int lo, int hi) {
  if (hi - lo < 2) return;
  int mid = (lo + hi) / 2;
  merge_sort_desc(s, idx, tmp, lo, mid);
  merge_sort_desc(s, idx, tmp, mid, hi);
  int i = lo, j = mid, k = lo;
  while (i < mid && j < hi) {
    /* take right only if strictly greater -> stable */
    if (s[idx[j]] > s[idx[i]]) tmp[k++] = idx[j++];
    else tmp[k++] = idx[i++];
  }
  while (i < mid) tmp[k++] = idx[i++];
  while (j < hi) tmp[k++] = idx[j++];
  memcpy(idx + lo, tmp + lo, sizeof(int64_t) * (hi - lo));
}

ORACLE_API int oracle_nms_f32(const float* dets, const float* scores, int ndets,
                              float threshold, int64_t* keep) {
  if (ndets <= 0) return 0; /* :13-15 */
  float* areas = (float*)malloc(sizeof(float) * ndets);
  int64_t* order = (int64_t*)malloc(sizeof(int64_t) * ndets);
  int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * ndets);
  uint8_t* suppressed = (uint8_t*)calloc(ndets, 1);
  for (int i = 0; i < ndets; i++) {
    const float* b = dets + 4 * i;
    areas[i] = (b[2] - b[0] + 1) * (b[3] - b[1] + 1); /* :22 */
    order[i] = i;
  }
  merge_sort_desc(scores, order, tmp, 0, ndets); /* :24 */
  for (int64_t _i = 0; _i < ndets; _i++) { /* :38-63 */
    int64_t i = order[_i];
    if (suppressed[i] == 1) continue;
    float ix1 = dets[4 * i], iy1 = dets[4 * i + 1], ix2 = dets[4 * i + 2],
          iy2 = dets[4 * i + 3];
    float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < ndets; _j++) {
      int64_t j = order[_j];
      if (suppressed[j] == 1) continue;
      float xx1 = std_maxf(ix1, dets[4 * j]);
      float yy1 = std_maxf(iy1, dets[4 * j + 1]);
      float xx2 = std_minf(ix2, dets[4 * j + 2]);
      float yy2 = std_minf(iy2, dets[4 * j + 3]);
      float w = std_maxf(0.f, xx2 - xx1 + 1);
      float h = std_maxf(0.f, yy2 - yy1 + 1);
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr >= threshold) suppressed[j] = 1; /* :60 */
    }
  }
  int k = 0;
  for (int i = 0; i < ndets; i++)
    if (!suppressed[i]) keep[k++] = i; /* :64 */
  free(areas); free(order); free(tmp); free(suppressed);
  return k;
}

/* ============================================================================================
 * SigmoidFocalLoss forward / backward — cuda/SigmoidFocalLoss_cuda.cu:20-58, :61-101, T = float.
 * The CUDA source mixes double literals into float expressions; the promotions below follow the
 * C++ usual arithmetic conversions of that source exactly.
 * ========================================================================================== */
ORACLE_API void oracle_sigmoid_focal_loss_forward_f32(const float* logits,
                                                      const int32_t* targets, float* losses,
                                                      int num, int num_classes, float gamma,
                                                      float alpha) {
  long nthreads = (long)num * num_classes;
  for (long i = 0; i < nthreads; i++) {
    int n = i / num_classes;
    int d = i % num_classes;
    int t = targets[n];
    float c1 = (float)(t == (d + 1));           /* :34 */
    float c2 = (float)((t >= 0) & (t != (d + 1))); /* :35 */
    float zn = (float)(1.0 - alpha);            /* :37 */
    float zp = alpha;
    float x = logits[i];
    float p = (float)(1. / (1. + expf(-x)));    /* :41 */
    float term1 = powf((float)(1. - p), gamma) * logf(fmaxf(p, FLT_MIN)); /* :44 */
    float term2 = (float)(powf(p, gamma) *
                          (-1. * x * (x >= 0) -
                           logf((float)(1. + expf((float)(x - 2. * x * (x >= 0))))))); /* :47-49 */
    float l = 0.0f;
    l += -c1 * term1 * zp; /* :52 */
    l += -c2 * term2 * zn; /* :53 */
    losses[i] = l;
  }
}

ORACLE_API void oracle_sigmoid_focal_loss_backward_f32(const float* logits,
                                                       const int32_t* targets,
                                                       const float* d_losses, float* d_logits,
                                                       int num, int num_classes, float gamma,
                                                       float alpha) {
  long nthreads = (long)num * num_classes;
  for (long i = 0; i < nthreads; i++) {
    int n = i / num_classes;
    int d = i % num_classes;
    int t = targets[n];
    float c1 = (float)(t == (d + 1));
    float c2 = (float)((t >= 0) & (t != (d + 1)));
    float zn = (float)(1.0 - alpha);
    float zp = alpha;
    float x = logits[i];
    float p = (float)(1. / (1. + expf(-x))); /* :82 */
    /* :85-86  (1-p)**g * (1 - p - g*p*log(p)) */
    float term1 = (float)(powf((float)(1. - p), gamma) *
                          (1. - p - (p * gamma * logf(fmaxf(p, FLT_MIN)))));
    /* :89-92  (p**g) * (g*(1-p)*log(1-p) - p) */
    float term2 = (float)(powf(p, gamma) *
                          ((-1. * x * (x >= 0) -
                            logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))) *
                               (1. - p) * gamma -
                           p));
    float g = 0.0f;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    g = g * d_losses[i]; /* :96 */
    d_logits[i] = g;
  }
}

/* ============================================================================================
 * Deformable convolution — cuda/deform_conv_kernel_cuda.cu.
 *   bilinear sample with zero padding      :91-122  (v1)  == :474-505 (modulated twin)
 *   get_gradient_weight                    :124-150
 *   get_coordinate_weight                  :152-195
 *   deformable_im2col_gpu_kernel           :197-250 ; modulated :577-640 (x mask at :634)
 *   deformable_col2im_gpu_kernel           :286-342 ; modulated :642-700
 *   deformable_col2im_coord_gpu_kernel     :380-443 ; modulated :702-774
 * mask == NULL selects v1.  All fp32.
 * ========================================================================================== */
static float dcn_im2col_bilinear(const float* bottom_data, int data_width, int height, int width,
                                 float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = bottom_data[h_low * data_width + w_low];
  if (h_low >= 0 && w_high <= width - 1) v2 = bottom_data[h_low * data_width + w_high];
  if (h_high <= height - 1 && w_low >= 0) v3 = bottom_data[h_high * data_width + w_low];
  if (h_high <= height - 1 && w_high <= width - 1)
    v4 = bottom_data[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

static float dcn_get_gradient_weight(float argmax_h, float argmax_w, int h, int w, int height,
                                     int width) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int argmax_h_low = (int)floorf(argmax_h), argmax_w_low = (int)floorf(argmax_w);
  int argmax_h_high = argmax_h_low + 1, argmax_w_high = argmax_w_low + 1;
  float weight = 0;
  if (h == argmax_h_low && w == argmax_w_low) weight = (h + 1 - argmax_h) * (w + 1 - argmax_w);
  if (h == argmax_h_low && w == argmax_w_high) weight = (h + 1 - argmax_h) * (argmax_w + 1 - w);
  if (h == argmax_h_high && w == argmax_w_low) weight = (argmax_h + 1 - h) * (w + 1 - argmax_w);
  if (h == argmax_h_high && w == argmax_w_high)
    weight = (argmax_h + 1 - h) * (argmax_w + 1 - w);
  return weight;
}

static float dcn_get_coordinate_weight(float argmax_h, float argmax_w, int height, int width,
                                       const float* im_data, int data_width, int bp_dir) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int argmax_h_low = (int)floorf(argmax_h), argmax_w_low = (int)floorf(argmax_w);
  int argmax_h_high = argmax_h_low + 1, argmax_w_high = argmax_w_low + 1;
  float weight = 0;
  if (bp_dir == 0) {
    if (argmax_h_low >= 0 && argmax_w_low >= 0)
      weight += -1 * (argmax_w_low + 1 - argmax_w) * im_data[argmax_h_low * data_width + argmax_w_low];
    if (argmax_h_low >= 0 && argmax_w_high <= width - 1)
      weight += -1 * (argmax_w - argmax_w_low) * im_data[argmax_h_low * data_width + argmax_w_high];
    if (argmax_h_high <= height - 1 && argmax_w_low >= 0)
      weight += (argmax_w_low + 1 - argmax_w) * im_data[argmax_h_high * data_width + argmax_w_low];
    if (argmax_h_high <= height - 1 && argmax_w_high <= width - 1)
      weight += (argmax_w - argmax_w_low) * im_data[argmax_h_high * data_width + argmax_w_high];
  } else if (bp_dir == 1) {
    if (argmax_h_low >= 0 && argmax_w_low >= 0)
      weight += -1 * (argmax_h_low + 1 - argmax_h) * im_data[argmax_h_low * data_width + argmax_w_low];
    if (argmax_h_low >= 0 && argmax_w_high <= width - 1)
      weight += (argmax_h_low + 1 - argmax_h) * im_data[argmax_h_low * data_width + argmax_w_high];
    if (argmax_h_high <= height - 1 && argmax_w_low >= 0)
      weight += -1 * (argmax_h - argmax_h_low) * im_data[argmax_h_high * data_width + argmax_w_low];
    if (argmax_h_high <= height - 1 && argmax_w_high <= width - 1)
      weight += (argmax_h - argmax_h_low) * im_data[argmax_h_high * data_width + argmax_w_high];
  }
  return weight;
}

/* data_im [B,C,H,W], data_offset [B,dg*2*kh*kw,Ho,Wo], data_mask [B,dg*kh*kw,Ho,Wo] or NULL,
 * data_col [C*kh*kw, B*Ho*Wo].  (batch_size == parallel_imgs in the reference's per-step call.) */
ORACLE_API void oracle_deformable_im2col_f32(const float* data_im, const float* data_offset,
                                             const float* data_mask, float* data_col,
                                             int batch_size, int num_channels, int height,
                                             int width, int kernel_h, int kernel_w, int pad_h,
                                             int pad_w, int stride_h, int stride_w,
                                             int dilation_h, int dilation_w,
                                             int deformable_group) {
  int height_col = (height + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
  int width_col = (width + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  int channel_per_deformable_group = num_channels / deformable_group;
  long n = (long)num_channels * height_col * width_col * batch_size; /* :264 */
  for (long index = 0; index < n; index++) {
    const int w_col = index % width_col; /* :210-214 */
    const int h_col = (index / width_col) % height_col;
    const int b_col = (index / width_col / height_col) % batch_size;
    const int c_im = (index / width_col / height_col) / batch_size;
    const int c_col = c_im * kernel_h * kernel_w;
    const int deformable_group_index = c_im / channel_per_deformable_group;
    const int h_in = h_col * stride_h - pad_h;
    const int w_in = w_col * stride_w - pad_w;
    float* data_col_ptr =
        data_col + (((size_t)c_col * batch_size + b_col) * height_col + h_col) * width_col + w_col;
    const float* data_im_ptr = data_im + ((size_t)b_col * num_channels + c_im) * height * width;
    const float* data_offset_ptr =
        data_offset + ((size_t)b_col * deformable_group + deformable_group_index) * 2 *
                          kernel_h * kernel_w * height_col * width_col;
    const float* data_mask_ptr =
        data_mask ? data_mask + ((size_t)b_col * deformable_group + deformable_group_index) *
                                    kernel_h * kernel_w * height_col * width_col
                  : NULL;
    for (int i = 0; i < kernel_h; ++i) {
      for (int j = 0; j < kernel_w; ++j) {
        const int data_offset_h_ptr = ((2 * (i * kernel_w + j)) * height_col + h_col) * width_col + w_col;
        const int data_offset_w_ptr = ((2 * (i * kernel_w + j) + 1) * height_col + h_col) * width_col + w_col;
        const float offset_h = data_offset_ptr[data_offset_h_ptr];
        const float offset_w = data_offset_ptr[data_offset_w_ptr];
        float val = 0.f;
        const float h_im = h_in + i * dilation_h + offset_h;
        const float w_im = w_in + j * dilation_w + offset_w;
        if (h_im > -1 && w_im > -1 && h_im < height && w_im < width) /* :236 */
          val = dcn_im2col_bilinear(data_im_ptr, width, height, width, h_im, w_im);
        if (data_mask_ptr) {
          const int data_mask_hw_ptr = ((i * kernel_w + j) * height_col + h_col) * width_col + w_col;
          val = val * data_mask_ptr[data_mask_hw_ptr]; /* :634 */
        }
        *data_col_ptr = val;
        data_col_ptr += (size_t)batch_size * height_col * width_col;
      }
    }
  }
}

/* grad_im [B,C,H,W] is ACCUMULATED into (caller zero-fills).  Flat-index order of additions. */
ORACLE_API void oracle_deformable_col2im_f32(const float* data_col, const float* data_offset,
                                             const float* data_mask, float* grad_im,
                                             int batch_size, int channels, int height, int width,
                                             int kernel_h, int kernel_w, int pad_h, int pad_w,
                                             int stride_h, int stride_w, int dilation_h,
                                             int dilation_w, int deformable_group) {
  int height_col = (height + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
  int width_col = (width + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  int channel_per_deformable_group = channels / deformable_group;
  long n = (long)channels * kernel_h * kernel_w * height_col * width_col * batch_size; /* :357 */
  for (long index = 0; index < n; index++) {
    const int j = (index / width_col / height_col / batch_size) % kernel_w; /* :300-302 */
    const int i = (index / width_col / height_col / batch_size / kernel_w) % kernel_h;
    const int c = index / width_col / height_col / batch_size / kernel_w / kernel_h;
    const int deformable_group_index = c / channel_per_deformable_group;
    int w_out = index % width_col;
    int h_out = (index / width_col) % height_col;
    int b = (index / width_col / height_col) % batch_size;
    int w_in = w_out * stride_w - pad_w;
    int h_in = h_out * stride_h - pad_h;
    const float* data_offset_ptr =
        data_offset + ((size_t)b * deformable_group + deformable_group_index) * 2 * kernel_h *
                          kernel_w * height_col * width_col;
    const int data_offset_h_ptr = ((2 * (i * kernel_w + j)) * height_col + h_out) * width_col + w_out;
    const int data_offset_w_ptr = ((2 * (i * kernel_w + j) + 1) * height_col + h_out) * width_col + w_out;
    const float offset_h = data_offset_ptr[data_offset_h_ptr];
    const float offset_w = data_offset_ptr[data_offset_w_ptr];
    const float cur_inv_h_data = h_in + i * dilation_h + offset_h;
    const float cur_inv_w_data = w_in + j * dilation_w + offset_w;
    float cur_top_grad = data_col[index];
    if (data_mask) {
      const float* data_mask_ptr =
          data_mask + ((size_t)b * deformable_group + deformable_group_index) * kernel_h *
                          kernel_w * height_col * width_col;
      const int data_mask_hw_ptr = ((i * kernel_w + j) * height_col + h_out) * width_col + w_out;
      cur_top_grad = data_col[index] * data_mask_ptr[data_mask_hw_ptr]; /* :681 */
    }
    const int cur_h = (int)cur_inv_h_data; /* :323-324 truncation toward zero */
    const int cur_w = (int)cur_inv_w_data;
    for (int dy = -2; dy <= 2; dy++) {
      for (int dx = -2; dx <= 2; dx++) {
        if (cur_h + dy >= 0 && cur_h + dy < height && cur_w + dx >= 0 && cur_w + dx < width &&
            fabsf(cur_inv_h_data - (cur_h + dy)) < 1 && fabsf(cur_inv_w_data - (cur_w + dx)) < 1) {
          size_t cur_bottom_grad_pos =
              (((size_t)b * channels + c) * height + cur_h + dy) * width + cur_w + dx;
          float weight = dcn_get_gradient_weight(cur_inv_h_data, cur_inv_w_data, cur_h + dy,
                                                 cur_w + dx, height, width);
          grad_im[cur_bottom_grad_pos] += weight * cur_top_grad; /* :337 atomicAdd */
        }
      }
    }
  }
}

/* grad_offset [B, dg*2*kh*kw, Ho, Wo] overwritten; grad_mask [B, dg*kh*kw, Ho, Wo] overwritten
 * when data_mask != NULL. */
ORACLE_API void oracle_deformable_col2im_coord_f32(
    const float* data_col, const float* data_im, const float* data_offset,
    const float* data_mask, float* grad_offset, float* grad_mask, int batch_size, int channels,
    int height, int width, int kernel_h, int kernel_w, int pad_h, int pad_w, int stride_h,
    int stride_w, int dilation_h, int dilation_w, int deformable_group) {
  int height_col = (height + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
  int width_col = (width + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  int offset_channels = 2 * kernel_h * kernel_w * deformable_group;
  int channel_per_deformable_group = channels * kernel_h * kernel_w / deformable_group; /* :456 */
  long n = (long)height_col * width_col * offset_channels * batch_size; /* :455 */
  for (long index = 0; index < n; index++) {
    float val = 0, mval = 0;
    int w = index % width_col;
    int h = (index / width_col) % height_col;
    int c = (index / width_col / height_col) % offset_channels;
    int b = (index / width_col / height_col) / offset_channels;
    const int deformable_group_index = c / (2 * kernel_h * kernel_w);
    const int col_step = kernel_h * kernel_w;
    int cnt = 0;
    const float* data_col_ptr = data_col + (size_t)deformable_group_index *
                                               channel_per_deformable_group * batch_size *
                                               width_col * height_col;
    const float* data_im_ptr =
        data_im + ((size_t)b * deformable_group + deformable_group_index) *
                      channel_per_deformable_group / kernel_h / kernel_w * height * width;
    const float* data_offset_ptr =
        data_offset + ((size_t)b * deformable_group + deformable_group_index) * 2 * kernel_h *
                          kernel_w * height_col * width_col;
    const float* data_mask_ptr =
        data_mask ? data_mask + ((size_t)b * deformable_group + deformable_group_index) *
                                    kernel_h * kernel_w * height_col * width_col
                  : NULL;
    const int offset_c = c - deformable_group_index * 2 * kernel_h * kernel_w;
    for (int col_c = (offset_c / 2); col_c < channel_per_deformable_group; col_c += col_step) {
      const long col_pos = (((long)(col_c * batch_size + b) * height_col) + h) * width_col + w;
      const int bp_dir = offset_c % 2;
      int j = (col_pos / width_col / height_col / batch_size) % kernel_w;
      int i = (col_pos / width_col / height_col / batch_size / kernel_w) % kernel_h;
      int w_out = col_pos % width_col;
      int h_out = (col_pos / width_col) % height_col;
      int w_in = w_out * stride_w - pad_w;
      int h_in = h_out * stride_h - pad_h;
      const int data_offset_h_ptr = (((2 * (i * kernel_w + j)) * height_col + h_out) * width_col + w_out);
      const int data_offset_w_ptr = (((2 * (i * kernel_w + j) + 1) * height_col + h_out) * width_col + w_out);
      const float offset_h = data_offset_ptr[data_offset_h_ptr];
      const float offset_w = data_offset_ptr[data_offset_w_ptr];
      float inv_h = h_in + i * dilation_h + offset_h;
      float inv_w = w_in + j * dilation_w + offset_w;
      float mask = 1.f;
      if (data_mask_ptr)
        mask = data_mask_ptr[(((i * kernel_w + j) * height_col + h_out) * width_col + w_out)];
      if (inv_h <= -1 || inv_w <= -1 || inv_h >= height || inv_w >= width) {
        inv_h = inv_w = -2; /* :431-434 */
      } else if (data_mask_ptr) {
        mval += data_col_ptr[col_pos] *
                dcn_im2col_bilinear(data_im_ptr + (size_t)cnt * height * width, width, height,
                                    width, inv_h, inv_w); /* :760 */
      }
      const float weight = dcn_get_coordinate_weight(
          inv_h, inv_w, height, width, data_im_ptr + (size_t)cnt * height * width, width, bp_dir);
      if (data_mask_ptr)
        val += weight * data_col_ptr[col_pos] * mask; /* :765 */
      else
        val += weight * data_col_ptr[col_pos]; /* :438 */
      cnt += 1;
    }
    grad_offset[index] = val;
    if (data_mask_ptr && offset_c % 2 == 0) /* :770-772 */
      grad_mask[((((size_t)b * deformable_group + deformable_group_index) * kernel_h * kernel_w +
                  offset_c / 2) * height_col + h) * width_col + w] = mval;
  }
}

/* Plain fp32 GEMM helpers used by the end-to-end deformable-conv restatements below:
 * C[M,N] (+)= A[M,K] * B[K,N], k-ascending accumulation in fp32 — the arithmetic the reference
 * delegates to cuBLAS via at::addmm_ (deform_conv_cuda.cu:237-242); order/rounding of a BLAS
 * GEMM is unpinned, parity for it is tolerance-based. */
static void gemm_nn_acc(float* C, const float* A, const float* B, int M, int N, int K) {
  for (int m = 0; m < M; m++)
    for (int k = 0; k < K; k++) {
      float a = A[(size_t)m * K + k];
      const float* brow = B + (size_t)k * N;
      float* crow = C + (size_t)m * N;
      for (int nn = 0; nn < N; nn++) crow[nn] += a * brow[nn];
    }
}

/* Deformable conv forward, end to end: deform_conv_cuda.cu:158-266 (v1, mask == NULL, no bias)
 * and :496-575 (modulated, bias optional).  group >= 1.  The reference's im2col_step batching
 * only changes how many images share one GEMM; results per output element are the same sums.
 * input [B,C,H,W], weight [Cout, C/group, kh, kw], out [B,Cout,Ho,Wo]. */
ORACLE_API void oracle_deform_conv_forward_f32(const float* input, const float* offset,
                                               const float* mask, const float* weight,
                                               const float* bias, float* out, int B, int C,
                                               int H, int W, int Cout, int kh, int kw, int pad_h,
                                               int pad_w, int stride_h, int stride_w, int dil_h,
                                               int dil_w, int group, int deformable_group) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  size_t colrows = (size_t)C * kh * kw, colcols = (size_t)Ho * Wo;
  float* col = (float*)malloc(sizeof(float) * colrows * colcols);
  for (int b = 0; b < B; b++) {
    oracle_deformable_im2col_f32(input + (size_t)b * C * H * W,
                                 offset + (size_t)b * deformable_group * 2 * kh * kw * Ho * Wo,
                                 mask ? mask + (size_t)b * deformable_group * kh * kw * Ho * Wo : NULL,
                                 col, 1, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h,
                                 dil_w, deformable_group);
    float* ob = out + (size_t)b * Cout * Ho * Wo;
    memset(ob, 0, sizeof(float) * Cout * colcols);
    int Mg = Cout / group, Kg = (C / group) * kh * kw;
    for (int g = 0; g < group; g++)
      gemm_nn_acc(ob + (size_t)g * Mg * colcols, weight + (size_t)g * Mg * Kg,
                  col + (size_t)g * Kg * colcols, Mg, (int)colcols, Kg);
    if (bias)
      for (int co = 0; co < Cout; co++)
        for (size_t p = 0; p < colcols; p++) ob[(size_t)co * colcols + p] += bias[co];
  }
  free(col);
}

/* Deformable conv backward, end to end: deform_conv_cuda.cu:268-380 (input+offset grads),
 * :382-494 (weight grad, scale = 1) and :577-691 (modulated: + mask grad, + bias grad).
 * All grad buffers are overwritten. grad_mask / grad_bias may be NULL (v1 / no bias). */
ORACLE_API void oracle_deform_conv_backward_f32(
    const float* input, const float* offset, const float* mask, const float* weight,
    const float* grad_out, float* grad_input, float* grad_offset, float* grad_mask,
    float* grad_weight, float* grad_bias, int B, int C, int H, int W, int Cout, int kh, int kw,
    int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int group,
    int deformable_group) {
  int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  size_t colrows = (size_t)C * kh * kw, colcols = (size_t)Ho * Wo;
  float* col = (float*)malloc(sizeof(float) * colrows * colcols);
  int Mg = Cout / group, Kg = (C / group) * kh * kw;
  memset(grad_input, 0, sizeof(float) * (size_t)B * C * H * W);
  memset(grad_weight, 0, sizeof(float) * (size_t)Cout * Kg);
  if (grad_bias) memset(grad_bias, 0, sizeof(float) * Cout);
  for (int b = 0; b < B; b++) {
    const float* in_b = input + (size_t)b * C * H * W;
    const float* off_b = offset + (size_t)b * deformable_group * 2 * kh * kw * Ho * Wo;
    const float* mask_b = mask ? mask + (size_t)b * deformable_group * kh * kw * Ho * Wo : NULL;
    const float* go_b = grad_out + (size_t)b * Cout * colcols;
    /* columns = W^T * gradOut  (:338-341 / :628-631) */
    memset(col, 0, sizeof(float) * colrows * colcols);
    for (int g = 0; g < group; g++) {
      const float* Wg = weight + (size_t)g * Mg * Kg;
      const float* Gg = go_b + (size_t)g * Mg * colcols;
      float* Cg = col + (size_t)g * Kg * colcols;
      for (int m = 0; m < Mg; m++)
        for (int k = 0; k < Kg; k++) {
          float a = Wg[(size_t)m * Kg + k];
          const float* grow = Gg + (size_t)m * colcols;
          float* crow = Cg + (size_t)k * colcols;
          for (size_t p = 0; p < colcols; p++) crow[p] += a * grow[p];
        }
    }
    oracle_deformable_col2im_coord_f32(
        col, in_b, off_b, mask_b,
        grad_offset + (size_t)b * deformable_group * 2 * kh * kw * Ho * Wo,
        grad_mask ? grad_mask + (size_t)b * deformable_group * kh * kw * Ho * Wo : NULL, 1, C, H,
        W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group);
    oracle_deformable_col2im_f32(col, off_b, mask_b, grad_input + (size_t)b * C * H * W, 1, C, H,
                                 W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                                 deformable_group);
    /* gradW += gradOut * cols^T (:466-472 / :666-670) */
    oracle_deformable_im2col_f32(in_b, off_b, mask_b, col, 1, C, H, W, kh, kw, pad_h, pad_w,
                                 stride_h, stride_w, dil_h, dil_w, deformable_group);
    for (int g = 0; g < group; g++) {
      const float* Gg = go_b + (size_t)g * Mg * colcols;
      const float* Cg = col + (size_t)g * Kg * colcols;
      float* GWg = grad_weight + (size_t)g * Mg * Kg;
      for (int m = 0; m < Mg; m++)
        for (int k = 0; k < Kg; k++) {
          float acc = 0.f;
          const float* grow = Gg + (size_t)m * colcols;
          const float* crow = Cg + (size_t)k * colcols;
          for (size_t p = 0; p < colcols; p++) acc += grow[p] * crow[p];
          GWg[(size_t)m * Kg + k] += acc;
        }
    }
    if (grad_bias) /* :671-676 gradOut * ones */
      for (int co = 0; co < Cout; co++) {
        float acc = 0.f;
        for (size_t p = 0; p < colcols; p++) acc += go_b[(size_t)co * colcols + p];
        grad_bias[co] += acc;
      }
  }
  free(col);
}

/* ============================================================================================
 * LevelMapper — modeling/poolers.py:33-42 (FPN paper eq. 1) with BoxList.area()
 * (structures/bounding_box.py:212-216, xyxy mode, TO_REMOVE = 1).  torch computes this in fp32:
 *   s = sqrt(area); target = floor(lvl0 + log2(s / s0 + eps)); clamp(k_min, k_max) - k_min.
 * ========================================================================================== */
ORACLE_API void oracle_fpn_level_f32(const float* rois /* [K,5] */, int K, int k_min, int k_max,
                                     float canonical_scale, float canonical_level, float eps,
                                     int32_t* levels) {
  for (int i = 0; i < K; i++) {
    const float* r = rois + 5 * i;
    float area = (r[3] - r[1] + 1) * (r[4] - r[2] + 1);
    float s = sqrtf(area);
    float t = floorf(canonical_level + log2f(s / canonical_scale + eps));
    if (t < k_min) t = (float)k_min;
    if (t > k_max) t = (float)k_max;
    levels[i] = (int32_t)t - k_min;
  }
}

/* ============================================================================================
 * Deformable PS-ROI pooling — cuda/deform_pool_kernel_cuda.cu:31-52 (bilinear_interp), :53-147
 * (DeformablePSROIPoolForwardKernel), :149-264 (DeformablePSROIPoolBackwardAccKernel),
 * scalar_t = float; host shape logic :266-305, :307-364.  CUDA-only in the reference and no
 * reference test pins its values: PARITY UNPINNED (source-faithful restatement).
 *
 * The float/double mixing of the .cu file is kept: literals such as 0.5, 0.1, 0., 1. are double
 * there, so `x * scale - 0.5`, the `max(.., 0.1)`, the border test and the clamp run in double
 * and are rounded back to float on assignment.
 *
 * data [N,C,H,W], rois [K,5], trans [K, 2*num_classes, part, part] (ignored when no_trans),
 * out / top_count [K, output_dim, P, P].  Backward accumulates into zeroed data_diff / trans_diff;
 * acc64 != 0 accumulates in double (bounds the reorder noise of the device's atomics).
 * ========================================================================================== */
typedef struct {
  int roi_batch_ind, part_h, part_w, class_id, gw, gh;
  float roi_width, roi_height, sub_bin_size_h, sub_bin_size_w, wstart, hstart;
} psroi_geom_t;

static psroi_geom_t psroi_geometry(const float* bottom_rois, const float* bottom_trans, int n,
                                   int ctop, int ph, int pw, float spatial_scale,
                                   int pooled_height, int pooled_width, int no_trans,
                                   float trans_std, int sample_per_part, int group_size,
                                   int part_size, int num_classes, int channels_each_class) {
  psroi_geom_t g;
  const float* r = bottom_rois + (size_t)n * 5;
  g.roi_batch_ind = (int)r[0];
  float roi_start_w = (float)((double)(roundf(r[1]) * spatial_scale) - 0.5); /* :83-86 */
  float roi_start_h = (float)((double)(roundf(r[2]) * spatial_scale) - 0.5);
  float roi_end_w = (float)((double)((float)((double)roundf(r[3]) + 1.) * spatial_scale) - 0.5);
  float roi_end_h = (float)((double)((float)((double)roundf(r[4]) + 1.) * spatial_scale) - 0.5);
  double rw = (double)(roi_end_w - roi_start_w); /* :89-90 max(float, double 0.1) */
  double rh = (double)(roi_end_h - roi_start_h);
  g.roi_width = (float)(rw < 0.1 ? 0.1 : rw);
  g.roi_height = (float)(rh < 0.1 ? 0.1 : rh);
  float bin_size_h = g.roi_height / (float)pooled_height; /* :93-94 */
  float bin_size_w = g.roi_width / (float)pooled_width;
  g.sub_bin_size_h = bin_size_h / (float)sample_per_part; /* :96-97 */
  g.sub_bin_size_w = bin_size_w / (float)sample_per_part;
  g.part_h = (int)floorf((float)ph / pooled_height * part_size); /* :99-100 */
  g.part_w = (int)floorf((float)pw / pooled_width * part_size);
  g.class_id = ctop / channels_each_class;
  float trans_x = 0.f, trans_y = 0.f; /* :102-103 */
  if (!no_trans) {
    trans_x = bottom_trans[(((size_t)(n * num_classes + g.class_id) * 2) * part_size + g.part_h) *
                               part_size + g.part_w] * trans_std;
    trans_y = bottom_trans[(((size_t)(n * num_classes + g.class_id) * 2 + 1) * part_size + g.part_h) *
                               part_size + g.part_w] * trans_std;
  }
  float wstart = (float)pw * bin_size_w + roi_start_w; /* :105-108 */
  wstart += trans_x * g.roi_width;
  float hstart = (float)ph * bin_size_h + roi_start_h;
  hstart += trans_y * g.roi_height;
  g.wstart = wstart;
  g.hstart = hstart;
  int gw = (int)floorf((float)pw * group_size / pooled_width); /* :112-115 */
  int gh = (int)floorf((float)ph * group_size / pooled_height);
  g.gw = gw < 0 ? 0 : (gw > group_size - 1 ? group_size - 1 : gw);
  g.gh = gh < 0 ? 0 : (gh > group_size - 1 ? group_size - 1 : gh);
  return g;
}

/* :126-133: returns 0 if the sample is skipped, else clamps w,h in place */
static int psroi_sample(float* w, float* h, int width, int height) {
  if ((double)*w < -0.5 || (double)*w > width - 0.5 || (double)*h < -0.5 || (double)*h > height - 0.5)
    return 0;
  double wd = (double)*w < 0. ? 0. : (double)*w;
  double hd = (double)*h < 0. ? 0. : (double)*h;
  wd = wd > width - 1. ? width - 1. : wd;
  hd = hd > height - 1. ? height - 1. : hd;
  *w = (float)wd;
  *h = (float)hd;
  return 1;
}

ORACLE_API void oracle_deform_psroi_pool_forward_f32(
    const float* bottom_data, const float* bottom_rois, const float* bottom_trans, float* top_data,
    float* top_count, int N, int channels, int height, int width, int num_rois, int channels_trans,
    int no_trans, float spatial_scale, int output_dim, int group_size, int pooled_size,
    int part_size, int sample_per_part, float trans_std) {
  (void)N;
  const int pooled_height = pooled_size, pooled_width = pooled_size;
  const int num_classes = no_trans ? 1 : channels_trans / 2; /* :289-290 */
  const int channels_each_class = no_trans ? output_dim : output_dim / num_classes;
  long count_all = (long)num_rois * output_dim * pooled_height * pooled_width;
  for (long index = 0; index < count_all; index++) {
    int pw = index % pooled_width; /* :77-80 */
    int ph = (index / pooled_width) % pooled_height;
    int ctop = (index / pooled_width / pooled_height) % output_dim;
    int n = index / pooled_width / pooled_height / output_dim;
    psroi_geom_t g = psroi_geometry(bottom_rois, bottom_trans, n, ctop, ph, pw, spatial_scale,
                                    pooled_height, pooled_width, no_trans, trans_std,
                                    sample_per_part, group_size, part_size, num_classes,
                                    channels_each_class);
    float sum = 0;
    int count = 0;
    const float* offset_bottom_data = bottom_data + (size_t)(g.roi_batch_ind * channels) * height * width;
    for (int ih = 0; ih < sample_per_part; ih++) {
      for (int iw = 0; iw < sample_per_part; iw++) {
        float w = g.wstart + iw * g.sub_bin_size_w; /* :123-124 */
        float h = g.hstart + ih * g.sub_bin_size_h;
        if (!psroi_sample(&w, &h, width, height)) continue;
        int c = (ctop * group_size + g.gh) * group_size + g.gw;
        const float* data = offset_bottom_data + (size_t)c * height * width;
        /* bilinear_interp :31-52 (value12 is the (y2, x1) tap) */
        int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        float dist_x = w - x1, dist_y = h - y1;
        float value11 = data[y1 * width + x1], value12 = data[y2 * width + x1];
        float value21 = data[y1 * width + x2], value22 = data[y2 * width + x2];
        float val = (1 - dist_x) * (1 - dist_y) * value11 + (1 - dist_x) * dist_y * value12 +
                    dist_x * (1 - dist_y) * value21 + dist_x * dist_y * value22;
        sum += val;
        count++;
      }
    }
    top_data[index] = count == 0 ? 0.f : sum / count; /* :142-143 */
    top_count[index] = (float)count;
  }
}

ORACLE_API void oracle_deform_psroi_pool_backward_f32(
    const float* top_diff, const float* bottom_data, const float* bottom_rois,
    const float* bottom_trans, const float* top_count, float* bottom_data_diff,
    float* bottom_trans_diff, int N, int channels, int height, int width, int num_rois,
    int channels_trans, int no_trans, float spatial_scale, int output_dim, int group_size,
    int pooled_size, int part_size, int sample_per_part, float trans_std, int acc64) {
  const int pooled_height = pooled_size, pooled_width = pooled_size;
  const int num_classes = no_trans ? 1 : channels_trans / 2;
  const int channels_each_class = no_trans ? output_dim : output_dim / num_classes;
  const size_t n_data = (size_t)N * channels * height * width;
  const size_t n_trans = no_trans ? 0 : (size_t)num_rois * channels_trans * part_size * part_size;
  double* dacc = NULL;
  double* tacc = NULL;
  memset(bottom_data_diff, 0, n_data * sizeof(float));
  if (n_trans) memset(bottom_trans_diff, 0, n_trans * sizeof(float));
  if (acc64) {
    dacc = (double*)calloc(n_data ? n_data : 1, sizeof(double));
    tacc = (double*)calloc(n_trans ? n_trans : 1, sizeof(double));
  }
#define PS_ADD_D(i, v) do { if (acc64) dacc[i] += (double)(v); else bottom_data_diff[i] += (v); } while (0)
#define PS_ADD_T(i, v) do { if (acc64) tacc[i] += (double)(v); else bottom_trans_diff[i] += (v); } while (0)
  long count_all = (long)num_rois * output_dim * pooled_height * pooled_width;
  for (long index = 0; index < count_all; index++) {
    int pw = index % pooled_width;
    int ph = (index / pooled_width) % pooled_height;
    int ctop = (index / pooled_width / pooled_height) % output_dim;
    int n = index / pooled_width / pooled_height / output_dim;
    psroi_geom_t g = psroi_geometry(bottom_rois, bottom_trans, n, ctop, ph, pw, spatial_scale,
                                    pooled_height, pooled_width, no_trans, trans_std,
                                    sample_per_part, group_size, part_size, num_classes,
                                    channels_each_class);
    if (top_count[index] <= 0) continue; /* :206-209 */
    float diff_val = top_diff[index] / top_count[index];
    size_t img = (size_t)g.roi_batch_ind * channels * height * width;
    for (int ih = 0; ih < sample_per_part; ih++) {
      for (int iw = 0; iw < sample_per_part; iw++) {
        float w = g.wstart + iw * g.sub_bin_size_w;
        float h = g.hstart + ih * g.sub_bin_size_h;
        if (!psroi_sample(&w, &h, width, height)) continue;
        int c = (ctop * group_size + g.gh) * group_size + g.gw;
        int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h); /* :233-236 */
        float dist_x = w - x0, dist_y = h - y0;
        float q00 = (1 - dist_x) * (1 - dist_y);
        float q01 = (1 - dist_x) * dist_y;
        float q10 = dist_x * (1 - dist_y);
        float q11 = dist_x * dist_y;
        size_t base = img + (size_t)c * height * width;
        PS_ADD_D(base + y0 * width + x0, q00 * diff_val); /* :243-246 */
        PS_ADD_D(base + y1 * width + x0, q01 * diff_val);
        PS_ADD_D(base + y0 * width + x1, q10 * diff_val);
        PS_ADD_D(base + y1 * width + x1, q11 * diff_val);
        if (no_trans) continue;
        float U00 = bottom_data[base + y0 * width + x0]; /* :252-255 */
        float U01 = bottom_data[base + y1 * width + x0];
        float U10 = bottom_data[base + y0 * width + x1];
        float U11 = bottom_data[base + y1 * width + x1];
        float diff_x = (U11 * dist_y + U10 * (1 - dist_y) - U01 * dist_y - U00 * (1 - dist_y)) *
                       trans_std * diff_val; /* :256-259 */
        diff_x *= g.roi_width;
        float diff_y = (U11 * dist_x + U01 * (1 - dist_x) - U10 * dist_x - U00 * (1 - dist_x)) *
                       trans_std * diff_val;
        diff_y *= g.roi_height;
        size_t tb = (((size_t)(n * num_classes + g.class_id) * 2) * part_size + g.part_h) * part_size + g.part_w;
        PS_ADD_T(tb, diff_x); /* :261-262 */
        PS_ADD_T(tb + (size_t)part_size * part_size, diff_y);
      }
    }
  }
#undef PS_ADD_D
#undef PS_ADD_T
  if (acc64) {
    for (size_t i = 0; i < n_data; i++) bottom_data_diff[i] = (float)dacc[i];
    for (size_t i = 0; i < n_trans; i++) bottom_trans_diff[i] = (float)tacc[i];
    free(dacc);
    free(tacc);
  }
}
