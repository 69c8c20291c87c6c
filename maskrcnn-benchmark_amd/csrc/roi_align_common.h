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

// ---- ROI visiting order (forward kernels): rank sort of (level, image, 16-row band, column) keys in LDS
constexpr int kOrderMaxK = 4096;
constexpr int kOrderMinK = 384;
constexpr int kOrderLanes = 16;     // lanes that share one ROI's count
constexpr int kOrderBlock = 1024;   // 16 waves: enough to hide the LDS read latency of the count loop

__device__ __forceinline__ unsigned long long roi_order_key(const Levels& L, const float* __restrict__ rois,
                                                            const int32_t* __restrict__ levels_in, int i) {
  const float* roi = rois + static_cast<size_t>(i) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[i] : fpn_level(roi, L);
  float scale = L.lv[0].scale;
#pragma unroll
  for (int l = 1; l < DETOPS_MAX_LEVELS; ++l)
    if (l == lvl) scale = L.lv[l].scale;
  const int b = static_cast<int>(roi[0]);
  const int xc = static_cast<int>((roi[1] + roi[3]) * 0.5f * scale);
  const int yc = static_cast<int>((roi[2] + roi[4]) * 0.0625f * scale);
  const unsigned key = (static_cast<unsigned>(lvl & 7) << 29) | (static_cast<unsigned>(min(max(b, 0), 127)) << 22) |
                       (static_cast<unsigned>(min(max(yc, 0), 1023)) << 12) | static_cast<unsigned>(min(max(xc, 0), 4095));
  return (static_cast<unsigned long long>(key) << 32) | static_cast<unsigned>(i);
}

// Ranking role of the pre-pass (one 1024-thread workgroup per 64 ROIs): see roi_fwd_prep_kernel.
__device__ __forceinline__ void roi_order_role(const Levels& L, const float* __restrict__ rois,
                                               const int32_t* __restrict__ levels_in, int K,
                                               int32_t* __restrict__ order, unsigned long long* keys, int block) {
  const int tid = threadIdx.x;
  const int Kp = (K + 2 * kOrderLanes - 1) / (2 * kOrderLanes) * (2 * kOrderLanes);   // padded with +inf keys
  for (int i = tid; i < Kp; i += kOrderBlock)
    keys[i] = i < K ? roi_order_key(L, rois, levels_in, i) : ~0ull;
  __syncthreads();
  const int r = (block * kOrderBlock + tid) / kOrderLanes;
  const int sub = tid & (kOrderLanes - 1);
  int cnt = 0;
  if (r < K) {
    const unsigned long long mine = keys[r];
#pragma unroll 8
    for (int j = 2 * sub; j < Kp; j += 2 * kOrderLanes) {   // one 16-byte LDS read = two keys; same address across ROIs: broadcast
      const unsigned long long k0 = keys[j], k1 = keys[j + 1];
      cnt += (k0 < mine ? 1 : 0) + (k1 < mine ? 1 : 0);
    }
  }
  cnt += __shfl_down(cnt, 8);
  cnt += __shfl_down(cnt, 4);
  cnt += __shfl_down(cnt, 2);
  cnt += __shfl_down(cnt, 1);
  if (r < K && sub == 0) { order[cnt] = r; DETOPS_STAT("fwd.ranked_rois", 1); }
}

// ---- the pixels a ROI's taps can reach, and its compact adjoint rows (backward kernels)
struct RoiExtent {
  int b;
  int fy0, ny, fx0, nx;   // rows / columns any tap of the ROI can reach, clipped to the map (n <= 0: none)
};

// The rows / columns any tap of the ROI can reach: taps of a sample at coordinate c are floor(c), floor(c) + 1 (clamped to
// the map), the sample coordinates grow with the sample index, so the reach is [floor(first sample), floor(last sample) + 1]
// — evaluated with the SAME fp32 expression as axis_entry (roi_align_common.h), clamped like it.  (Round 3 used the scaled
// rectangle + 2: two rows / columns more per ROI, ~10 % more (tile, ROI) hits for nothing.)
__device__ __forceinline__ void axis_reach(float start, float bin, int P, int grid, int size, int& lo, int& n) {
#pragma clang fp contract(off)
  const float c0 = start + 0 * bin + static_cast<float>(0 + .5f) * bin / static_cast<float>(grid);
  const float c1 = start + (P - 1) * bin + static_cast<float>((grid - 1) + .5f) * bin / static_cast<float>(grid);
  lo = 0; n = 0;
  if (!(c1 >= -1.0f) || !(c0 <= static_cast<float>(size))) return;      // every sample outside the map (or NaN geometry)
  const int first = (c0 <= 0.f) ? 0 : min(static_cast<int>(c0), size - 1);
  const int last = (c1 >= static_cast<float>(size - 1)) ? size - 1 : static_cast<int>(fmaxf(c1, 0.f)) + 1;
  lo = first;
  n = last - first + 1;
}

__device__ __forceinline__ RoiExtent roi_extent(const float* __restrict__ roi, float scale, int H, int W, int PH, int PW, int sr) {
  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  RoiExtent e;
  e.b = g.b;
  axis_reach(g.start_h, g.bin_h, PH, g.gh, H, e.fy0, e.ny);
  axis_reach(g.start_w, g.bin_w, PW, g.gw, W, e.fx0, e.nx);
  return e;
}

// One wave builds one ROI's compact adjoint rows (a lane per (axis, footprint pixel)): the pixel's row of the adjoint
// matrix as {first contributing bin | other axis' longest range << 8 | count << 16, w[0], w[1], ...} (zero beyond),
// candidate samples from the inverse of the sample-coordinate map, decided by the exact reference arithmetic.  Rows
// are footprint-relative (row 0 = the ROI's first reachable pixel): y rows at slot[i * PPH], x rows at
// slot[ax_off + i * PPW].  `slot` may be global memory (the ring's pre-pass) or LDS (the acc kernel).
__device__ __forceinline__ void build_adjoint_rows(const float* __restrict__ roi, float scale, int H, int W, int PH, int PW,
                                                   int sr, float* slot, size_t ax_off, int PPH, int PPW, int lane) {
  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  const RoiExtent e = roi_extent(roi, scale, H, W, PH, PW, sr);
  const int ny = max(e.ny, 0), nx = max(e.nx, 0);
  int cmax_y = 0, cmax_x = 0;   // longest bin range among this lane's rows, per axis
  for (int p = lane; p < ny + nx; p += kWave) {
    const bool isy = p < ny;
    const int pi = isy ? p : p - ny;
    const int pix = (isy ? e.fy0 : e.fx0) + pi;
    const int PB = isy ? PH : PW, PP = isy ? PPH : PPW;
    const int grid = isy ? g.gh : g.gw, size = isy ? H : W;
    const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
    float* row = slot + (isy ? 0 : ax_off) + static_cast<size_t>(pi) * PP;
    const float inv = 1.f / static_cast<float>(grid);
    for (int q = 0; q < PP; ++q) row[q] = 0.f;
    // only samples whose coordinate lies within one pixel of `pix` can have a tap on it (border pixels also
    // collect the clamped samples: c in [-1, 0] -> pixel 0, c in [size-1, size] -> pixel size-1).  Candidate
    // sample range from the inverse of c(s) = start + (s + .5) * bin / grid, widened by one sample on each
    // side; the exact reference arithmetic then decides.
    const float step = bin * inv;
    const float clo = (pix == 0) ? -1.f : static_cast<float>(pix - 1);
    const float chi = (pix == size - 1) ? static_cast<float>(size) : static_cast<float>(pix + 1);
    const int ns = PB * grid;
    int s0 = static_cast<int>(fminf(fmaxf(floorf((clo - start) / step - 0.5f) - 1.f, 0.f), static_cast<float>(ns)));
    int s1 = static_cast<int>(fminf(fmaxf(ceilf((chi - start) / step - 0.5f) + 1.f, -1.f), static_cast<float>(ns - 1)));
    if (!(step > 0.f)) { s0 = 0; s1 = ns - 1; }   // degenerate geometry (NaN / inf): look at everything
    int lo = -1, hi = -1;
    int q = s0 / grid, i = s0 - q * grid;
    float w = 0.f;
    for (int sidx = s0; sidx <= s1; ++sidx) {   // per bin: samples in ascending order, like the scan kernel
      const Tap tp = axis_entry(start, bin, q, i, grid, size, 1);
      if (tp.lo == pix) w += tp.h * inv;
      if (tp.hi == pix) w += tp.l * inv;
      if (++i == grid || sidx == s1) {
        if (w != 0.f) {
          if (lo < 0) lo = q;
          hi = q;
          row[1 + q - lo] = w;
        }
        w = 0.f; i = 0; ++q;
      }
    }
    row[0] = __int_as_float((lo >= 0) ? (lo | ((hi - lo + 1) << 16)) : 0);
    if (lo >= 0) { if (isy) cmax_y = max(cmax_y, hi - lo + 1); else cmax_x = max(cmax_x, hi - lo + 1); }
  }
  // bits 8..15 of every row head: the ROI's longest bin range on the OTHER axis — the ring walk learns both of its
  // trip counts from the two AY heads of a wave's rows (two readlanes) instead of wave-wide ballots over the pixels
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    cmax_y = max(cmax_y, __shfl(cmax_y, lane ^ off));
    cmax_x = max(cmax_x, __shfl(cmax_x, lane ^ off));
  }
  for (int p = lane; p < ny + nx; p += kWave) {
    const bool isy = p < ny;
    float* row = slot + (isy ? static_cast<size_t>(p) * PPH : ax_off + static_cast<size_t>(p - ny) * PPW);
    row[0] = __int_as_float(__float_as_int(row[0]) | ((isy ? cmax_x : cmax_y) << 8));
  }
}

inline bool bad_dims(int N, int C, int K, int PH, int PW) {
  return N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0;
}

}  // namespace
