// cpu_branch.hip — HOST code only: the two operators the reference also serves for CPU tensors.
//
// The reference `_C` dispatches on the tensor's device: `nms` (csrc/nms.h:10-28 -> csrc/cpu/nms_cpu.cpp:5-75)
// and `ROIAlign_forward` (csrc/ROIAlign.h:11-25 -> csrc/cpu/ROIAlign_cpu.cpp:113-257) have CPU
// implementations, every other operator raises "Not implemented on the CPU".  These are the same two, for the
// same case — a caller that hands CPU tensors to the drop-in (data-loader side box filtering, unit tests of model
// code on a machine without a GPU).  They are NOT a fallback of the device path: device pointers never reach
// this file, and the device entry points do not call it.  Written against the reference's semantics, not its
// source: one stable score sort, greedy suppression with `>=`, separable tap tables for the pooling.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <thread>
#include <vector>

#include "detops.h"

#define DETOPS_API extern "C" __attribute__((visibility("default")))

namespace {

// T = float | double: the reference dispatches both (AT_DISPATCH_FLOATING_TYPES, ROIAlign_cpu.cpp:242, nms_cpu.cpp:71)
template <typename T>
struct AxisTap {
  int lo, hi;
  T l, h;   // weight of hi / of lo; both 0 for a sample outside the map
};

// one sampling coordinate -> taps, in the reference's order of operations (ROIAlign_cpu.cpp:36-92)
template <typename T>
inline AxisTap<T> axis_tap(T start, T bin, int p, int i, int grid, int size) {
#pragma clang fp contract(off)
  AxisTap<T> t{0, 0, T(0), T(0)};
  T c = start + p * bin + static_cast<T>(i + .5f) * bin / static_cast<T>(grid);
  if (c < T(-1.0) || c > static_cast<T>(size)) return t;
  if (c <= T(0)) c = T(0);
  int lo = static_cast<int>(c), hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    c = static_cast<T>(lo);
  } else {
    hi = lo + 1;
  }
  t.lo = lo;
  t.hi = hi;
  t.l = c - static_cast<T>(lo);
  t.h = static_cast<T>(1. - static_cast<double>(t.l));
  return t;
}

template <typename T>
void pool_rois(const T* input, const T* rois, T* output, int C, int H, int W, int PH, int PW,
               T scale, int sr, int k_begin, int k_end) {
#pragma clang fp contract(off)
  std::vector<AxisTap<T>> ty, tx;
  const size_t plane = static_cast<size_t>(H) * W;
  for (int k = k_begin; k < k_end; ++k) {
    const T* roi = rois + static_cast<size_t>(k) * 5;
    const int b = static_cast<int>(roi[0]);
    const T start_w = roi[1] * scale, start_h = roi[2] * scale;
    const T end_w = roi[3] * scale, end_h = roi[4] * scale;
    const T rw = std::max(end_w - start_w, T(1)), rh = std::max(end_h - start_h, T(1));
    const T bin_h = rh / static_cast<T>(PH), bin_w = rw / static_cast<T>(PW);
    const int gh = sr > 0 ? sr : static_cast<int>(std::ceil(rh / PH));
    const int gw = sr > 0 ? sr : static_cast<int>(std::ceil(rw / PW));
    const T count = static_cast<T>(gh * gw);
    ty.resize(static_cast<size_t>(PH) * gh);
    tx.resize(static_cast<size_t>(PW) * gw);
    for (int p = 0; p < PH; ++p)
      for (int i = 0; i < gh; ++i) ty[static_cast<size_t>(p) * gh + i] = axis_tap<T>(start_h, bin_h, p, i, gh, H);
    for (int p = 0; p < PW; ++p)
      for (int i = 0; i < gw; ++i) tx[static_cast<size_t>(p) * gw + i] = axis_tap<T>(start_w, bin_w, p, i, gw, W);
    for (int c = 0; c < C; ++c) {
      const T* d = input + (static_cast<size_t>(b) * C + c) * plane;
      T* o = output + (static_cast<size_t>(k) * C + c) * PH * PW;
      for (int ph = 0; ph < PH; ++ph) {
        for (int pw = 0; pw < PW; ++pw) {
          T acc = T(0);
          for (int iy = 0; iy < gh; ++iy) {
            const AxisTap<T>& y = ty[static_cast<size_t>(ph) * gh + iy];
            const T* r0 = d + static_cast<size_t>(y.lo) * W;
            const T* r1 = d + static_cast<size_t>(y.hi) * W;
            for (int ix = 0; ix < gw; ++ix) {
              const AxisTap<T>& x = tx[static_cast<size_t>(pw) * gw + ix];
              const T w1 = y.h * x.h, w2 = y.h * x.l, w3 = y.l * x.h, w4 = y.l * x.l;
              acc += w1 * r0[x.lo] + w2 * r0[x.hi] + w3 * r1[x.lo] + w4 * r1[x.hi];
            }
          }
          o[ph * PW + pw] = acc / count;
        }
      }
    }
  }
}

template <typename T>
int roi_align_forward_cpu(const T* input, const T* rois, T* output, int N, int C, int H, int W, int K, int PH, int PW,
                          float spatial_scale, int sampling_ratio) {
  if (N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0 || H < 0 || W < 0) return DETOPS_EINVAL;
  if (K == 0 || C == 0) return 0;
  if (!input || !rois || !output || H == 0 || W == 0 || N == 0) return DETOPS_EINVAL;
  for (int k = 0; k < K; ++k) {
    const T b = rois[static_cast<size_t>(k) * 5];
    if (!(b >= T(0) && b < static_cast<T>(N))) return DETOPS_EINVAL;   // a host caller gets an error, not a wild read
  }
  const T scale = static_cast<T>(spatial_scale);      // `const T& spatial_scale` of the reference kernel (a float argument upstream)
  const int workers = std::max(1, std::min<int>(static_cast<int>(std::thread::hardware_concurrency()), K / 8));
  if (workers == 1) {
    pool_rois<T>(input, rois, output, C, H, W, PH, PW, scale, sampling_ratio, 0, K);
    return 0;
  }
  std::vector<std::thread> pool;
  for (int t = 0; t < workers; ++t) {
    const int k0 = static_cast<int>(static_cast<int64_t>(K) * t / workers);
    const int k1 = static_cast<int>(static_cast<int64_t>(K) * (t + 1) / workers);
    pool.emplace_back(pool_rois<T>, input, rois, output, C, H, W, PH, PW, scale, sampling_ratio, k0, k1);
  }
  for (auto& th : pool) th.join();
  return 0;
}

template <typename T>
int nms_cpu(const T* boxes, const T* scores, int n, float iou_threshold, int64_t* keep, int32_t* num_keep) {
#pragma clang fp contract(off)
  if (n < 0 || !num_keep) return DETOPS_EINVAL;
  *num_keep = 0;
  if (n == 0) return 0;
  if (!boxes || !scores || !keep) return DETOPS_EINVAL;
  std::vector<int> order(static_cast<size_t>(n));
  std::iota(order.begin(), order.end(), 0);
  // descending score, ties in ascending index (what the device sort does; NaN scores order last)
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const T sa = scores[a], sb = scores[b];
    if (sa != sa) return false;
    if (sb != sb) return true;
    return sa > sb;
  });
  std::vector<T> area(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    const T* b = boxes + static_cast<size_t>(i) * 4;
    area[i] = (b[2] - b[0] + T(1)) * (b[3] - b[1] + T(1));
  }
  std::vector<uint8_t> dead(static_cast<size_t>(n), 0);
  for (int a = 0; a < n; ++a) {
    const int i = order[a];
    if (dead[i]) continue;
    const T* bi = boxes + static_cast<size_t>(i) * 4;
    for (int c = a + 1; c < n; ++c) {
      const int j = order[c];
      if (dead[j]) continue;
      const T* bj = boxes + static_cast<size_t>(j) * 4;
      const T xx1 = std::fmax(bi[0], bj[0]), yy1 = std::fmax(bi[1], bj[1]);
      const T xx2 = std::fmin(bi[2], bj[2]), yy2 = std::fmin(bi[3], bj[3]);
      const T w = std::fmax(T(0), xx2 - xx1 + T(1)), h = std::fmax(T(0), yy2 - yy1 + T(1));
      const T inter = w * h;
      const T ovr = inter / (area[i] + area[j] - inter);
      if (ovr >= iou_threshold) dead[j] = 1;       // nms_cpu.cpp:59-60: `ovr >= threshold` (a float threshold, promoted)
    }
  }
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (!dead[i]) keep[m++] = i;   // ascending original index, like nonzero(suppressed == 0)
  *num_keep = m;
  return 0;
}

}  // namespace

DETOPS_API int detops_roi_align_forward_cpu_f32(const float* input, const float* rois, float* output, int N,
                                                int C, int H, int W, int K, int PH, int PW,
                                                float spatial_scale, int sampling_ratio) {
  return roi_align_forward_cpu<float>(input, rois, output, N, C, H, W, K, PH, PW, spatial_scale, sampling_ratio);
}

DETOPS_API int detops_roi_align_forward_cpu_f64(const double* input, const double* rois, double* output, int N,
                                                int C, int H, int W, int K, int PH, int PW,
                                                float spatial_scale, int sampling_ratio) {
  return roi_align_forward_cpu<double>(input, rois, output, N, C, H, W, K, PH, PW, spatial_scale, sampling_ratio);
}

DETOPS_API int detops_nms_cpu_f32(const float* boxes, const float* scores, int n, float iou_threshold,
                                  int64_t* keep, int32_t* num_keep) {
  return nms_cpu<float>(boxes, scores, n, iou_threshold, keep, num_keep);
}

DETOPS_API int detops_nms_cpu_f64(const double* boxes, const double* scores, int n, float iou_threshold,
                                  int64_t* keep, int32_t* num_keep) {
  return nms_cpu<double>(boxes, scores, n, iou_threshold, keep, num_keep);
}
