// roi_align_common.h — shared pieces of the ROIAlign forward / backward kernels (gfx950, fp32 NCHW).
// Reference arithmetic: maskrcnn_benchmark/csrc/cpu/ROIAlign_cpu.cpp:36-92, csrc/cuda/ROIAlign_cuda.cu:15-49.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "detops_common.h"

namespace {

constexpr int kBlock = 256;

struct __align__(16) Tap {
  int lo, hi;   // y axis: pre-multiplied by W
  float l, h;   // frac, 1-frac (both 0 for a sample outside the map)
};

struct Level {
  const float* in;  // forward: feature map; backward: unused
  float* gin;       // backward: gradient map
  int H, W;
  float scale;
};

struct Levels {
  Level lv[DETOPS_MAX_LEVELS];
  int num;
  // LevelMapper parameters (only read when num > 1 and levels == nullptr)
  int k_min, k_max;
  float s0, lvl0, eps;
};

// One axis sample, reference order of operations (ROIAlign_cpu.cpp:36-92 / ROIAlign_cuda.cu:15-49).
__device__ __forceinline__ Tap axis_entry(float start, float bin, int p, int i, int grid, int size,
                                          int premul) {
#pragma clang fp contract(off)
  Tap t;
  float c = start + p * bin + static_cast<float>(i + .5f) * bin / static_cast<float>(grid);
  if (c < -1.0f || c > static_cast<float>(size)) {
    t.lo = 0; t.hi = 0; t.l = 0.f; t.h = 0.f;
    return t;
  }
  if (c <= 0.f) c = 0.f;
  int lo = static_cast<int>(c), hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    c = static_cast<float>(lo);
  } else {
    hi = lo + 1;
  }
  const float l = c - static_cast<float>(lo);
  t.lo = lo * premul;
  t.hi = hi * premul;
  t.l = l;
  t.h = static_cast<float>(1. - static_cast<double>(l));  // `T hy = 1. - ly` (double literal)
  return t;
}

struct RoiGeom {
  int b;
  float start_w, start_h, bin_w, bin_h;
  int gh, gw;
  float count;
};

__device__ __forceinline__ RoiGeom roi_geometry(const float* __restrict__ roi, float scale, int PH,
                                                int PW, int sr) {
#pragma clang fp contract(off)
  RoiGeom g;
  g.b = static_cast<int>(roi[0]);
  g.start_w = roi[1] * scale;
  g.start_h = roi[2] * scale;
  const float end_w = roi[3] * scale;
  const float end_h = roi[4] * scale;
  const float rw = fmaxf(end_w - g.start_w, 1.f);
  const float rh = fmaxf(end_h - g.start_h, 1.f);
  g.bin_h = rh / static_cast<float>(PH);
  g.bin_w = rw / static_cast<float>(PW);
  g.gh = (sr > 0) ? sr : static_cast<int>(ceilf(rh / PH));
  g.gw = (sr > 0) ? sr : static_cast<int>(ceilf(rw / PW));
  g.count = static_cast<float>(g.gh * g.gw);
  return g;
}

// LevelMapper (reference modeling/poolers.py:33-42) on device, fp32 like torch.
__device__ __forceinline__ int fpn_level(const float* __restrict__ roi, const Levels& L) {
#pragma clang fp contract(off)
  const float area = (roi[3] - roi[1] + 1.f) * (roi[4] - roi[2] + 1.f);
  const float s = sqrtf(area);
  float t = floorf(L.lvl0 + log2f(s / L.s0 + L.eps));
  t = fminf(fmaxf(t, static_cast<float>(L.k_min)), static_cast<float>(L.k_max));
  return static_cast<int>(t) - L.k_min;
}

inline bool bad_dims(int N, int C, int K, int PH, int PW) {
  return N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0;
}

}  // namespace
