// roi_align.hip — ROIAlign forward / backward for gfx950 (MI355X), fp32 NCHW.
//
// Replaces the reference's RoIAlignForward / RoIAlignBackwardFeature
// (maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:64-122, :177-254) and its CPU forward
// (csrc/cpu/ROIAlign_cpu.cpp:113-219) behind detops_roi_align_{forward,backward}_f32 and the
// multi-level detops_roi_align_fpn_* entry points (include/detops.h).
//
// Design (not a translation of the reference's one-thread-per-output kernels):
//   * one workgroup per (ROI, channel chunk).  The sampling geometry of a ROI is SEPARABLE:
//     the y taps depend on (ph, iy) only and the x taps on (pw, ix) only, so wave 0.. build two
//     small axis tables in LDS — PH*gh + PW*gw entries of {low, high, frac, 1-frac} — once per
//     workgroup instead of re-deriving 4 indices + 4 weights per output element per channel
//     (the reference CUDA kernel redoes that C times; the CPU kernel keeps a PH*PW*gh*gw table).
//   * the per-sample arithmetic keeps the reference's evaluation order with FP contraction off
//     (w = hy*hx ...; val = w1*v1 + w2*v2 + w3*v3 + w4*v4; acc += val; acc /= count), so the
//     forward is bit-identical to the reference CPU kernel for finite inputs.
//   * lanes run over (channel, bin) with bin fastest: stores are fully coalesced (the output of a
//     (ROI, chunk) is one contiguous run), gathers hit a <= ~30x30 px patch per channel that
//     stays L1/L2 resident.
//   * backward: the ROI's footprint patch [channels x py x px] is accumulated in LDS with
//     ds_add_f32 (no global atomics inside a ROI), then flushed with row-contiguous
//     global_atomic_add_f32 — one atomic per touched input pixel per ROI instead of 4 per sample.
#include <type_traits>

#include "detops_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTabBig = 512;    // axis-table entries per axis kept in LDS (adaptive grids);
constexpr int kTabSmall = 32;   // fixed sampling_ratio: PH*sr, PW*sr <= 32 covers 7x7..14x14 @ sr 2
constexpr int kPatchFloats = 8192;  // backward LDS patch budget (32 KiB)

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

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int PH_, int PW_, int kTabCap>
__global__ void __launch_bounds__(kBlock)
roi_align_fwd_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                     int32_t* __restrict__ levels_out, float* __restrict__ out, int C, int K,
                     int PHr, int PWr, int sr, int CT, int chunks) {
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  __shared__ Tap tabY[kTabCap];
  __shared__ Tap tabX[kTabCap];

  const int k = blockIdx.x / chunks;
  const int chunk = blockIdx.x - k * chunks;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  if (levels_out && chunk == 0 && threadIdx.x == 0) levels_out[k] = lvl;
  // wave-uniform select (num <= 8): avoids dynamic indexing of the kernarg struct
  const float* in = L.lv[0].in; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { in = L.lv[i].in; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  const int ny = PH * g.gh, nx = PW * g.gw;
  const bool use_tab = (ny <= kTabCap) && (nx <= kTabCap);
  if (use_tab) {
    for (int t = threadIdx.x; t < ny + nx; t += kBlock) {
      if (t < ny) {
        tabY[t] = axis_entry(g.start_h, g.bin_h, t / g.gh, t % g.gh, g.gh, H, W);
      } else {
        const int u = t - ny;
        tabX[u] = axis_entry(g.start_w, g.bin_w, u / g.gw, u % g.gw, g.gw, W, 1);
      }
    }
    __syncthreads();
  }

  const int c0 = chunk * CT;
  const int cend = min(C, c0 + CT);
  const int total = (cend - c0) * bins;
  const size_t plane = static_cast<size_t>(H) * W;
  const float* base = in + (static_cast<size_t>(g.b) * C + c0) * plane;
  float* obase = out + (static_cast<size_t>(k) * C + c0) * bins;

  for (int o = threadIdx.x; o < total; o += kBlock) {
#pragma clang fp contract(off)
    const int cl = o / bins;
    const int bin = o - cl * bins;
    const int ph = bin / PW;
    const int pw = bin - ph * PW;
    const float* d = base + static_cast<size_t>(cl) * plane;
    float acc = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      const Tap ty = use_tab ? tabY[ph * g.gh + iy]
                             : axis_entry(g.start_h, g.bin_h, ph, iy, g.gh, H, W);
      const float* r0 = d + ty.lo;
      const float* r1 = d + ty.hi;
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap tx = use_tab ? tabX[pw * g.gw + ix]
                               : axis_entry(g.start_w, g.bin_w, pw, ix, g.gw, W, 1);
        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
        const float v1 = r0[tx.lo], v2 = r0[tx.hi], v3 = r1[tx.lo], v4 = r1[tx.hi];
        acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
      }
    }
    obase[o] = acc / g.count;
  }
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
template <int PH_, int PW_, int kTabCap>
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                     const float* __restrict__ gout, int C, int K, int PHr, int PWr, int sr, int CT,
                     int chunks) {
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  __shared__ Tap tabY[kTabCap];
  __shared__ Tap tabX[kTabCap];
  __shared__ float patch[kPatchFloats];
  __shared__ int s_bounds[4];  // ymin, ymax, xmin, xmax over valid taps (tabY holds RAW rows here)

  const int k = blockIdx.x / chunks;
  const int chunk = blockIdx.x - k * chunks;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  const int ny = PH * g.gh, nx = PW * g.gw;
  const bool use_tab = (ny <= kTabCap) && (nx <= kTabCap);
  const int c0 = chunk * CT;
  const int cend = min(C, c0 + CT);
  const size_t plane = static_cast<size_t>(H) * W;
  float* gbase = gin + (static_cast<size_t>(g.b) * C + c0) * plane;
  const float* gobase = gout + (static_cast<size_t>(k) * C + c0) * bins;
  // (g*w)/count in the reference; for power-of-two counts the reciprocal multiply is exact
  const int icount = g.gh * g.gw;
  const bool pow2 = (icount & (icount - 1)) == 0;
  const float inv_count = 1.f / g.count;

  int py = 0, px = 0, ymin = 0, xmin = 0;
  bool lds_path = false;
  if (use_tab) {
    if (threadIdx.x == 0) {
      s_bounds[0] = 0x7fffffff; s_bounds[1] = -1; s_bounds[2] = 0x7fffffff; s_bounds[3] = -1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ny + nx; t += kBlock) {
      if (t < ny) {
        const Tap e = axis_entry(g.start_h, g.bin_h, t / g.gh, t % g.gh, g.gh, H, 1);
        tabY[t] = e;
        if (e.l != 0.f || e.h != 0.f) { atomicMin(&s_bounds[0], e.lo); atomicMax(&s_bounds[1], e.hi); }
      } else {
        const int u = t - ny;
        const Tap e = axis_entry(g.start_w, g.bin_w, u / g.gw, u % g.gw, g.gw, W, 1);
        tabX[u] = e;
        if (e.l != 0.f || e.h != 0.f) { atomicMin(&s_bounds[2], e.lo); atomicMax(&s_bounds[3], e.hi); }
      }
    }
    __syncthreads();
    if (s_bounds[1] < 0 || s_bounds[3] < 0) return;  // ROI entirely outside the map (uniform)
    ymin = s_bounds[0];
    py = s_bounds[1] - ymin + 1;
    xmin = s_bounds[2];
    px = s_bounds[3] - xmin + 1;
    lds_path = (py * px) <= kPatchFloats;
  }

  if (lds_path) {
    const int area = py * px;
    const int ctb = min(cend - c0, kPatchFloats / area);
    for (int cs = c0; cs < cend; cs += ctb) {
      const int cn = min(ctb, cend - cs);
      for (int e = threadIdx.x; e < cn * area; e += kBlock) patch[e] = 0.f;
      __syncthreads();
      const int total = cn * bins;
      const float* go = gobase + static_cast<size_t>(cs - c0) * bins;
      for (int o = threadIdx.x; o < total; o += kBlock) {
#pragma clang fp contract(off)
        const int cl = o / bins;
        const int bin = o - cl * bins;
        const int ph = bin / PW;
        const int pw = bin - ph * PW;
        const float gval = go[o];
        float* pp = patch + cl * area;
        for (int iy = 0; iy < g.gh; ++iy) {
          const Tap ty = tabY[ph * g.gh + iy];
          if (ty.l == 0.f && ty.h == 0.f) continue;
          const int r0 = (ty.lo - ymin) * px - xmin;
          const int r1 = (ty.hi - ymin) * px - xmin;
          for (int ix = 0; ix < g.gw; ++ix) {
            const Tap tx = tabX[pw * g.gw + ix];
            if (tx.l == 0.f && tx.h == 0.f) continue;
            const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
            float g1 = gval * w1, g2 = gval * w2, g3 = gval * w3, g4 = gval * w4;
            if (pow2) { g1 *= inv_count; g2 *= inv_count; g3 *= inv_count; g4 *= inv_count; }
            else { g1 /= g.count; g2 /= g.count; g3 /= g.count; g4 /= g.count; }
            atomicAdd(pp + r0 + tx.lo, g1);
            atomicAdd(pp + r0 + tx.hi, g2);
            atomicAdd(pp + r1 + tx.lo, g3);
            atomicAdd(pp + r1 + tx.hi, g4);
          }
        }
      }
      __syncthreads();
      float* gb = gbase + static_cast<size_t>(cs - c0) * plane + static_cast<size_t>(ymin) * W + xmin;
      for (int e = threadIdx.x; e < cn * area; e += kBlock) {
        const float v = patch[e];
        if (v != 0.f) {
          const int cl = e / area;
          const int r = e - cl * area;
          const int y = r / px;
          const int x = r - y * px;
          atomicAdd(gb + static_cast<size_t>(cl) * plane + y * W + x, v);
        }
      }
      __syncthreads();
    }
    return;
  }

  // Fallback (huge adaptive grids or patches): direct global atomics, reference-style scatter.
  const int total = (cend - c0) * bins;
  for (int o = threadIdx.x; o < total; o += kBlock) {
#pragma clang fp contract(off)
    const int cl = o / bins;
    const int bin = o - cl * bins;
    const int ph = bin / PW;
    const int pw = bin - ph * PW;
    const float gval = gobase[o];
    float* d = gbase + static_cast<size_t>(cl) * plane;
    for (int iy = 0; iy < g.gh; ++iy) {
      const Tap ty = use_tab ? tabY[ph * g.gh + iy]
                             : axis_entry(g.start_h, g.bin_h, ph, iy, g.gh, H, 1);
      if (ty.l == 0.f && ty.h == 0.f) continue;
      float* d0 = d + static_cast<size_t>(ty.lo) * W;
      float* d1 = d + static_cast<size_t>(ty.hi) * W;
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap tx = use_tab ? tabX[pw * g.gw + ix]
                               : axis_entry(g.start_w, g.bin_w, pw, ix, g.gw, W, 1);
        if (tx.l == 0.f && tx.h == 0.f) continue;
        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
        atomicAdd(d0 + tx.lo, gval * w1 / g.count);
        atomicAdd(d0 + tx.hi, gval * w2 / g.count);
        atomicAdd(d1 + tx.lo, gval * w3 / g.count);
        atomicAdd(d1 + tx.hi, gval * w4 / g.count);
      }
    }
  }
}

// Channel chunk per workgroup: enough workgroups to fill 256 CUs several times over while
// keeping the per-workgroup table build amortised over >= 16 channels.
inline int pick_chunk(int C, int K) {
  int CT = 64;
  while (CT > 16 && static_cast<int64_t>(K) * ceil_div64(C, CT) < 4 * kNumCU) CT >>= 1;
  if (CT > C) CT = C;
  return CT;
}

template <int V> using IC = std::integral_constant<int, V>;

template <typename F>
inline void dispatch_shape(int PH, int PW, int sr, F&& f) {
  const bool small = sr > 0 && PH * sr <= kTabSmall && PW * sr <= kTabSmall;
  if (PH == 7 && PW == 7 && small) f(IC<7>{}, IC<7>{}, IC<kTabSmall>{});
  else if (PH == 14 && PW == 14 && small) f(IC<14>{}, IC<14>{}, IC<kTabSmall>{});
  else if (PH == 7 && PW == 7) f(IC<7>{}, IC<7>{}, IC<kTabBig>{});
  else if (PH == 14 && PW == 14) f(IC<14>{}, IC<14>{}, IC<kTabBig>{});
  else if (small) f(IC<0>{}, IC<0>{}, IC<kTabSmall>{});
  else f(IC<0>{}, IC<0>{}, IC<kTabBig>{});
}

int run_forward(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                float* out, int C, int K, int PH, int PW, int sr, hipStream_t st) {
  if (K == 0 || C == 0) return 0;
  const int CT = pick_chunk(C, K);
  const int chunks = static_cast<int>(ceil_div64(C, CT));
  const dim3 grid(static_cast<unsigned>(K) * chunks);
  dispatch_shape(PH, PW, sr, [&](auto ph, auto pw, auto tab) {
    hipLaunchKernelGGL((roi_align_fwd_kernel<decltype(ph)::value, decltype(pw)::value, decltype(tab)::value>), grid,
                       dim3(kBlock), 0, st, L, rois, levels_in, levels_out, out, C, K, PH, PW, sr,
                       CT, chunks);
  });
  return launch_status();
}

int run_backward(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                 int C, int K, int PH, int PW, int sr, hipStream_t st) {
  if (K == 0 || C == 0) return 0;
  const int CT = pick_chunk(C, K);
  const int chunks = static_cast<int>(ceil_div64(C, CT));
  const dim3 grid(static_cast<unsigned>(K) * chunks);
  dispatch_shape(PH, PW, sr, [&](auto ph, auto pw, auto tab) {
    hipLaunchKernelGGL((roi_align_bwd_kernel<decltype(ph)::value, decltype(pw)::value, decltype(tab)::value>), grid,
                       dim3(kBlock), 0, st, L, rois, levels_in, gout, C, K, PH, PW, sr, CT, chunks);
  });
  return launch_status();
}

inline bool bad_dims(int N, int C, int K, int PH, int PW) {
  return N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0;
}

}  // namespace

DETOPS_API int detops_roi_align_forward_f32(const float* input, const float* rois, float* output,
                                            int N, int C, int H, int W, int K, int PH, int PW,
                                            float spatial_scale, int sampling_ratio,
                                            detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || H < 0 || W < 0) return DETOPS_EINVAL;
  if (K == 0 || C == 0) return 0;
  if (!input || !rois || !output || H == 0 || W == 0 || N == 0) return DETOPS_EINVAL;
  Levels L{};
  L.num = 1;
  L.lv[0] = Level{input, nullptr, H, W, spatial_scale};
  return run_forward(L, rois, nullptr, nullptr, output, C, K, PH, PW, sampling_ratio,
                     as_stream(stream));
}

DETOPS_API int detops_roi_align_backward_f32(const float* grad_out, const float* rois,
                                             float* grad_in, int N, int C, int H, int W, int K,
                                             int PH, int PW, float spatial_scale,
                                             int sampling_ratio, int zero_grad_in,
                                             detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || H < 0 || W < 0) return DETOPS_EINVAL;
  const size_t bytes = sizeof(float) * static_cast<size_t>(N) * C * H * W;
  if (bytes && !grad_in) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (zero_grad_in && bytes) DETOPS_HIP_TRY(hipMemsetAsync(grad_in, 0, bytes, st));
  if (K == 0 || C == 0 || bytes == 0) return 0;
  if (!grad_out || !rois) return DETOPS_EINVAL;
  Levels L{};
  L.num = 1;
  L.lv[0] = Level{nullptr, grad_in, H, W, spatial_scale};
  return run_backward(L, rois, nullptr, grad_out, C, K, PH, PW, sampling_ratio, st);
}

DETOPS_API int detops_roi_align_fpn_forward_f32(
    const float* const* inputs_host, const int* H_host, const int* W_host, const float* scale_host,
    int num_levels, const float* rois, float* output, int32_t* levels_out, int N, int C, int K,
    int PH, int PW, int sampling_ratio, int k_min, int k_max, float canonical_scale,
    float canonical_level, float eps, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS ||
      !inputs_host || !H_host || !W_host || !scale_host)
    return DETOPS_EINVAL;
  if (k_max - k_min + 1 != num_levels) return DETOPS_EINVAL;
  if (K == 0 || C == 0) return 0;
  if (!rois || !output) return DETOPS_EINVAL;
  Levels L{};
  L.num = num_levels;
  L.k_min = k_min; L.k_max = k_max; L.s0 = canonical_scale; L.lvl0 = canonical_level; L.eps = eps;
  for (int i = 0; i < num_levels; ++i) {
    if (!inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{inputs_host[i], nullptr, H_host[i], W_host[i], scale_host[i]};
  }
  hipStream_t st = as_stream(stream);
  if (num_levels == 1 && levels_out) DETOPS_HIP_TRY(hipMemsetAsync(levels_out, 0, sizeof(int32_t) * K, st));
  return run_forward(L, rois, nullptr, levels_out, output, C, K, PH, PW, sampling_ratio, st);
}

DETOPS_API int detops_roi_align_fpn_backward_f32(
    const float* grad_out, const float* rois, const int32_t* levels, float* const* grad_inputs_host,
    const int* H_host, const int* W_host, const float* scale_host, int num_levels, int N, int C,
    int K, int PH, int PW, int sampling_ratio, int zero_grad_in, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS ||
      !grad_inputs_host || !H_host || !W_host || !scale_host)
    return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (!grad_inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{nullptr, grad_inputs_host[i], H_host[i], W_host[i], scale_host[i]};
    if (zero_grad_in) {
      const size_t bytes = sizeof(float) * static_cast<size_t>(N) * C * H_host[i] * W_host[i];
      if (bytes) DETOPS_HIP_TRY(hipMemsetAsync(grad_inputs_host[i], 0, bytes, st));
    }
  }
  if (K == 0 || C == 0 || N == 0) return 0;
  if (!grad_out || !rois || (num_levels > 1 && !levels)) return DETOPS_EINVAL;
  return run_backward(L, rois, levels, grad_out, C, K, PH, PW, sampling_ratio, st);
}
