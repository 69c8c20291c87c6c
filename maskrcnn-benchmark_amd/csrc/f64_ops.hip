// f64_ops.hip — the double-precision forms of the reference's floating-point-dispatched operators for gfx950 (MI355X).
//
// The reference dispatches ROIAlign / ROIPool / SigmoidFocalLoss / nms over AT_DISPATCH_FLOATING_TYPES (float AND double:
// csrc/cuda/ROIAlign_cuda.cu:283,329, ROIPool_cuda.cu:137,185, SigmoidFocalLoss_cuda.cu:129,173, csrc/cpu/nms_cpu.cpp:71,
// csrc/cpu/ROIAlign_cpu.cpp:242).  No configuration of the training path uses double, so these are plain API-completeness
// kernels — one thread per output element, the reference's arithmetic in T = double operation by operation, hardware fp64
// atomics (global_atomic_add_f64) for the two scatters — not tuned: the fp32 kernels of this library are the product.
#include <algorithm>

#include "detops_common.h"

namespace {

constexpr int kF64Block = 256;

struct TapD { int lo, hi; double l, h; };

// one sampling coordinate (csrc/cuda/ROIAlign_cuda.cu:15-49, T = double)
__device__ __forceinline__ TapD axis_d(double start, double bin, int p, int i, int grid, int size) {
#pragma clang fp contract(off)
  TapD t{0, 0, 0., 0.};
  double c = start + p * bin + static_cast<double>(i + .5f) * bin / static_cast<double>(grid);
  if (c < -1.0 || c > static_cast<double>(size)) return t;
  if (c <= 0.) c = 0.;
  int lo = static_cast<int>(c), hi;
  if (lo >= size - 1) { hi = lo = size - 1; c = static_cast<double>(lo); } else { hi = lo + 1; }
  t.lo = lo; t.hi = hi;
  t.l = c - static_cast<double>(lo);
  t.h = 1. - t.l;
  return t;
}

struct GeoD { int b; double sw, sh, bw, bh; int gh, gw; double count; };
__device__ __forceinline__ GeoD geo_d(const double* roi, double scale, int PH, int PW, int sr) {
#pragma clang fp contract(off)
  GeoD g;
  g.b = static_cast<int>(roi[0]);
  g.sw = roi[1] * scale; g.sh = roi[2] * scale;
  const double ew = roi[3] * scale, eh = roi[4] * scale;
  const double rw = fmax(ew - g.sw, 1.), rh = fmax(eh - g.sh, 1.);
  g.bh = rh / static_cast<double>(PH); g.bw = rw / static_cast<double>(PW);
  g.gh = sr > 0 ? sr : static_cast<int>(ceil(rh / PH));
  g.gw = sr > 0 ? sr : static_cast<int>(ceil(rw / PW));
  g.count = static_cast<double>(g.gh * g.gw);
  return g;
}

__global__ void __launch_bounds__(kF64Block)
roi_align_fwd_f64_kernel(const double* __restrict__ in, const double* __restrict__ rois, double* __restrict__ out, int C, int H,
                         int W, int PH, int PW, double scale, int sr, int64_t total) {
#pragma clang fp contract(off)
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; o < total; o += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int pw = static_cast<int>(o % PW), ph = static_cast<int>((o / PW) % PH);
    const int c = static_cast<int>((o / (static_cast<int64_t>(PW) * PH)) % C);
    const int64_t k = o / (static_cast<int64_t>(PW) * PH * C);
    const GeoD g = geo_d(rois + k * 5, scale, PH, PW, sr);
    const double* d = in + (static_cast<int64_t>(g.b) * C + c) * H * W;
    double acc = 0.;
    for (int iy = 0; iy < g.gh; ++iy) {
      const TapD y = axis_d(g.sh, g.bh, ph, iy, g.gh, H);
      for (int ix = 0; ix < g.gw; ++ix) {
        const TapD x = axis_d(g.sw, g.bw, pw, ix, g.gw, W);
        const double w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
        acc += w1 * d[y.lo * W + x.lo] + w2 * d[y.lo * W + x.hi] + w3 * d[y.hi * W + x.lo] + w4 * d[y.hi * W + x.hi];
      }
    }
    out[o] = acc / g.count;
  }
}

// csrc/cuda/ROIAlign_cuda.cu:176-254 (T = double): one thread per pooled gradient element, 4 atomic adds per sample
__global__ void __launch_bounds__(kF64Block)
roi_align_bwd_f64_kernel(const double* __restrict__ gout, const double* __restrict__ rois, double* __restrict__ gin, int C, int H,
                         int W, int PH, int PW, double scale, int sr, int64_t total) {
#pragma clang fp contract(off)
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; o < total; o += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int pw = static_cast<int>(o % PW), ph = static_cast<int>((o / PW) % PH);
    const int c = static_cast<int>((o / (static_cast<int64_t>(PW) * PH)) % C);
    const int64_t k = o / (static_cast<int64_t>(PW) * PH * C);
    const GeoD g = geo_d(rois + k * 5, scale, PH, PW, sr);
    double* d = gin + (static_cast<int64_t>(g.b) * C + c) * H * W;
    const double top = gout[o];
    for (int iy = 0; iy < g.gh; ++iy) {
      const TapD y = axis_d(g.sh, g.bh, ph, iy, g.gh, H);
      if (y.l == 0. && y.h == 0.) continue;
      for (int ix = 0; ix < g.gw; ++ix) {
        const TapD x = axis_d(g.sw, g.bw, pw, ix, g.gw, W);
        if (x.l == 0. && x.h == 0.) continue;
        const double w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
        atomicAdd(d + y.lo * W + x.lo, top * w1 / g.count);
        atomicAdd(d + y.lo * W + x.hi, top * w2 / g.count);
        atomicAdd(d + y.hi * W + x.lo, top * w3 / g.count);
        atomicAdd(d + y.hi * W + x.hi, top * w4 / g.count);
      }
    }
  }
}

// csrc/cuda/ROIPool_cuda.cu:16-72 (T = double)
__global__ void __launch_bounds__(kF64Block)
roi_pool_fwd_f64_kernel(const double* __restrict__ in, const double* __restrict__ rois, double* __restrict__ out,
                        int32_t* __restrict__ argmax, int C, int H, int W, int PH, int PW, double scale, int64_t total) {
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; o < total; o += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int pw = static_cast<int>(o % PW), ph = static_cast<int>((o / PW) % PH);
    const int c = static_cast<int>((o / (static_cast<int64_t>(PW) * PH)) % C);
    const int64_t k = o / (static_cast<int64_t>(PW) * PH * C);
    const double* roi = rois + k * 5;
    const int b = static_cast<int>(roi[0]);
    const int rsw = static_cast<int>(round(roi[1] * scale)), rsh = static_cast<int>(round(roi[2] * scale));
    const int rew = static_cast<int>(round(roi[3] * scale)), reh = static_cast<int>(round(roi[4] * scale));
    const int rw = max(rew - rsw + 1, 1), rh = max(reh - rsh + 1, 1);
    const double bh = static_cast<double>(rh) / static_cast<double>(PH), bw = static_cast<double>(rw) / static_cast<double>(PW);
    int hs = static_cast<int>(floor(static_cast<double>(ph) * bh)), ws = static_cast<int>(floor(static_cast<double>(pw) * bw));
    int he = static_cast<int>(ceil(static_cast<double>(ph + 1) * bh)), we = static_cast<int>(ceil(static_cast<double>(pw + 1) * bw));
    hs = min(max(hs + rsh, 0), H); he = min(max(he + rsh, 0), H);
    ws = min(max(ws + rsw, 0), W); we = min(max(we + rsw, 0), W);
    const bool empty = (he <= hs) || (we <= ws);
    double mv = empty ? 0. : -1.7976931348623157e308;      // the reference starts at -FLT_MAX; any finite input beats either
    int mi = -1;
    const double* d = in + (static_cast<int64_t>(b) * C + c) * H * W;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w)
        if (d[h * W + w] > mv) { mv = d[h * W + w]; mi = h * W + w; }
    out[o] = mv;
    argmax[o] = mi;
  }
}

__global__ void __launch_bounds__(kF64Block)
roi_pool_bwd_f64_kernel(const double* __restrict__ gout, const double* __restrict__ rois, const int32_t* __restrict__ argmax,
                        double* __restrict__ gin, int C, int H, int W, int PH, int PW, int64_t total) {
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; o < total; o += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int c = static_cast<int>((o / (static_cast<int64_t>(PW) * PH)) % C);
    const int64_t k = o / (static_cast<int64_t>(PW) * PH * C);
    const int b = static_cast<int>(rois[k * 5]);
    const int a = argmax[o];
    if (a != -1) atomicAdd(gin + (static_cast<int64_t>(b) * C + c) * H * W + a, gout[o]);
  }
}

// csrc/cuda/SigmoidFocalLoss_cuda.cu:20-99 (T = double; the transcendental calls are the reference's float ones: expf / logf / powf
// on the value converted to float, as `expf(-logits[i])` does with a double argument)
__global__ void __launch_bounds__(kF64Block)
focal_fwd_f64_kernel(const double* __restrict__ logits, const int32_t* __restrict__ targets, double* __restrict__ losses, int C,
                     float gamma, float alpha, int64_t total) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int64_t n = i / C;
    const int d = static_cast<int>(i % C);
    const int t = targets[n];
    const double c1 = (t == (d + 1)), c2 = (t >= 0 & t != (d + 1));
    const double zn = (1.0 - alpha), zp = alpha;
    const double x = logits[i];
    const double p = 1. / (1. + expf(static_cast<float>(-x)));
    const double term1 = powf(static_cast<float>(1. - p), gamma) * logf(static_cast<float>(fmax(p, 1.17549435e-38)));
    const double term2 = powf(static_cast<float>(p), gamma) * (-1. * x * (x >= 0) - logf(static_cast<float>(1. + expf(static_cast<float>(x - 2. * x * (x >= 0))))));
    double l = 0.0;
    l += -c1 * term1 * zp;
    l += -c2 * term2 * zn;
    losses[i] = l;
  }
}

__global__ void __launch_bounds__(kF64Block)
focal_bwd_f64_kernel(const double* __restrict__ logits, const int32_t* __restrict__ targets, const double* __restrict__ dl,
                     double* __restrict__ dx, int C, float gamma, float alpha, int64_t total) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int64_t n = i / C;
    const int d = static_cast<int>(i % C);
    const int t = targets[n];
    const double c1 = (t == (d + 1)), c2 = (t >= 0 & t != (d + 1));
    const double zn = (1.0 - alpha), zp = alpha;
    const double x = logits[i];
    const double p = 1. / (1. + expf(static_cast<float>(-x)));
    const double term1 = powf(static_cast<float>(1. - p), gamma) * (1. - p - (p * gamma * logf(static_cast<float>(fmax(p, 1.17549435e-38)))));
    const double term2 = powf(static_cast<float>(p), gamma) *
                         ((-1. * x * (x >= 0) - logf(static_cast<float>(1. + expf(static_cast<float>(x - 2. * x * (x >= 0)))))) * (1. - p) * gamma - p);
    double g = 0.0;
    g += -c1 * term1 * zp;
    g += -c2 * term2 * zn;
    dx[i] = g * dl[i];
  }
}

// NMS on boxes already sorted by descending score (csrc/cpu/nms_cpu.cpp:37-63 semantics, T = double): pass 1 the upper-triangle
// suppression bits, pass 2 one thread walks the chain.
__global__ void __launch_bounds__(kF64Block)
nms_mask_f64_kernel(const double* __restrict__ boxes, int n, float thr, unsigned long long* __restrict__ mask, int words) {
#pragma clang fp contract(off)
  const int64_t total = static_cast<int64_t>(n) * words;
  for (int64_t o = static_cast<int64_t>(blockIdx.x) * kF64Block + threadIdx.x; o < total; o += static_cast<int64_t>(gridDim.x) * kF64Block) {
    const int i = static_cast<int>(o / words), wd = static_cast<int>(o % words);
    const double* bi = boxes + static_cast<int64_t>(i) * 4;
    const double ai = (bi[2] - bi[0] + 1) * (bi[3] - bi[1] + 1);
    unsigned long long bits = 0ull;
    for (int q = 0; q < 64; ++q) {
      const int j = wd * 64 + q;
      if (j <= i || j >= n) continue;
      const double* bj = boxes + static_cast<int64_t>(j) * 4;
      const double aj = (bj[2] - bj[0] + 1) * (bj[3] - bj[1] + 1);
      const double xx1 = fmax(bi[0], bj[0]), yy1 = fmax(bi[1], bj[1]), xx2 = fmin(bi[2], bj[2]), yy2 = fmin(bi[3], bj[3]);
      const double w = fmax(0., xx2 - xx1 + 1), h = fmax(0., yy2 - yy1 + 1);
      const double inter = w * h;
      const double ovr = inter / (ai + aj - inter);
      if (ovr >= thr) bits |= 1ull << q;
    }
    mask[o] = bits;
  }
}

__global__ void nms_scan_f64_kernel(const unsigned long long* __restrict__ mask, int n, int words, unsigned long long* __restrict__ removed,
                                    unsigned char* __restrict__ keep_sorted) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int w = 0; w < words; ++w) removed[w] = 0ull;
  for (int i = 0; i < n; ++i) {
    const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;
    keep_sorted[i] = dead ? 0 : 1;
    if (!dead)
      for (int w = i >> 6; w < words; ++w) removed[w] |= mask[static_cast<int64_t>(i) * words + w];
  }
}

inline unsigned f64_grid(int64_t total) { return static_cast<unsigned>(std::min<int64_t>(std::max<int64_t>(1, ceil_div64(total, kF64Block)), 1 << 16)); }

}  // namespace

DETOPS_API int detops_roi_align_forward_f64(const double* input, const double* rois, double* output, int N, int C, int H, int W,
                                            int K, int PH, int PW, float spatial_scale, int sampling_ratio, detops_stream_t stream) {
  if (N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0 || H < 0 || W < 0) return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(K) * C * PH * PW;
  if (total == 0) return 0;
  if (!input || !rois || !output || H == 0 || W == 0 || N == 0) return DETOPS_EINVAL;
  hipLaunchKernelGGL(roi_align_fwd_f64_kernel, dim3(f64_grid(total)), dim3(kF64Block), 0, as_stream(stream), input, rois, output, C, H, W, PH,
                     PW, static_cast<double>(spatial_scale), sampling_ratio, total);
  return launch_status();
}

DETOPS_API int detops_roi_align_backward_f64(const double* grad_out, const double* rois, double* grad_in, int N, int C, int H, int W,
                                             int K, int PH, int PW, float spatial_scale, int sampling_ratio, int zero_grad_in,
                                             detops_stream_t stream) {
  if (N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0 || H < 0 || W < 0) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t gin_n = static_cast<int64_t>(N) * C * H * W;
  if (gin_n == 0) return 0;
  if (!grad_in) return DETOPS_EINVAL;
  if (zero_grad_in) DETOPS_HIP_TRY(hipMemsetAsync(grad_in, 0, sizeof(double) * gin_n, st));
  const int64_t total = static_cast<int64_t>(K) * C * PH * PW;
  if (total == 0) return 0;
  if (!grad_out || !rois) return DETOPS_EINVAL;
  hipLaunchKernelGGL(roi_align_bwd_f64_kernel, dim3(f64_grid(total)), dim3(kF64Block), 0, st, grad_out, rois, grad_in, C, H, W, PH, PW,
                     static_cast<double>(spatial_scale), sampling_ratio, total);
  return launch_status();
}

DETOPS_API int detops_roi_pool_forward_f64(const double* input, const double* rois, double* output, int32_t* argmax, int N, int C, int H,
                                           int W, int K, int PH, int PW, float spatial_scale, detops_stream_t stream) {
  if (N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0 || H < 0 || W < 0) return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(K) * C * PH * PW;
  if (total == 0) return 0;
  if (!input || !rois || !output || !argmax) return DETOPS_EINVAL;
  hipLaunchKernelGGL(roi_pool_fwd_f64_kernel, dim3(f64_grid(total)), dim3(kF64Block), 0, as_stream(stream), input, rois, output, argmax, C, H,
                     W, PH, PW, static_cast<double>(spatial_scale), total);
  return launch_status();
}

DETOPS_API int detops_roi_pool_backward_f64(const double* grad_out, const double* rois, const int32_t* argmax, double* grad_in, int N, int C,
                                            int H, int W, int K, int PH, int PW, int zero_grad_in, detops_stream_t stream) {
  if (N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0 || H < 0 || W < 0) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t gin_n = static_cast<int64_t>(N) * C * H * W;
  if (gin_n == 0) return 0;
  if (!grad_in) return DETOPS_EINVAL;
  if (zero_grad_in) DETOPS_HIP_TRY(hipMemsetAsync(grad_in, 0, sizeof(double) * gin_n, st));
  const int64_t total = static_cast<int64_t>(K) * C * PH * PW;
  if (total == 0) return 0;
  if (!grad_out || !rois || !argmax) return DETOPS_EINVAL;
  hipLaunchKernelGGL(roi_pool_bwd_f64_kernel, dim3(f64_grid(total)), dim3(kF64Block), 0, st, grad_out, rois, argmax, grad_in, C, H, W, PH, PW,
                     total);
  return launch_status();
}

DETOPS_API int detops_sigmoid_focal_loss_forward_f64(const double* logits, const int32_t* targets, double* losses, int num_rows,
                                                     int num_classes, float gamma, float alpha, detops_stream_t stream) {
  if (num_rows < 0 || num_classes < 0) return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(num_rows) * num_classes;
  if (total == 0) return 0;
  if (!logits || !targets || !losses) return DETOPS_EINVAL;
  hipLaunchKernelGGL(focal_fwd_f64_kernel, dim3(f64_grid(total)), dim3(kF64Block), 0, as_stream(stream), logits, targets, losses, num_classes,
                     gamma, alpha, total);
  return launch_status();
}

DETOPS_API int detops_sigmoid_focal_loss_backward_f64(const double* logits, const int32_t* targets, const double* d_losses, double* d_logits,
                                                      int num_rows, int num_classes, float gamma, float alpha, detops_stream_t stream) {
  if (num_rows < 0 || num_classes < 0) return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(num_rows) * num_classes;
  if (total == 0) return 0;
  if (!logits || !targets || !d_losses || !d_logits) return DETOPS_EINVAL;
  hipLaunchKernelGGL(focal_bwd_f64_kernel, dim3(f64_grid(total)), dim3(kF64Block), 0, as_stream(stream), logits, targets, d_losses, d_logits,
                     num_classes, gamma, alpha, total);
  return launch_status();
}

// boxes [n, 4] SORTED by descending score (ties: ascending original index); keep_sorted [n] bytes (1 = kept);
// workspace: n * ceil(n / 64) + ceil(n / 64) 8-byte words (detops_nms_sorted_f64_workspace_bytes)
DETOPS_API size_t detops_nms_sorted_f64_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t words = static_cast<size_t>((n + 63) / 64);
  return 8 * (static_cast<size_t>(n) * words + words);
}

DETOPS_API int detops_nms_sorted_f64(const double* sorted_boxes, int n, float iou_threshold, unsigned char* keep_sorted, void* workspace,
                                     size_t workspace_bytes, detops_stream_t stream) {
  if (n < 0) return DETOPS_EINVAL;
  if (n == 0) return 0;
  if (!sorted_boxes || !keep_sorted || !workspace || workspace_bytes < detops_nms_sorted_f64_workspace_bytes(n)) return DETOPS_EINVAL;
  const int words = (n + 63) / 64;
  unsigned long long* mask = static_cast<unsigned long long*>(workspace);
  unsigned long long* removed = mask + static_cast<size_t>(n) * words;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(nms_mask_f64_kernel, dim3(f64_grid(static_cast<int64_t>(n) * words)), dim3(kF64Block), 0, st, sorted_boxes, n,
                     iou_threshold, mask, words);
  int e = launch_status();
  if (e) return e;
  hipLaunchKernelGGL(nms_scan_f64_kernel, dim3(1), dim3(64), 0, st, static_cast<const unsigned long long*>(mask), n, words, removed, keep_sorted);
  return launch_status();
}
