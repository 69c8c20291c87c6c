// roi_pool.hip — ROIPool (max) forward / backward for gfx950 (MI355X), fp32 NCHW.
//
// Replaces RoIPoolFForward / RoIPoolFBackward (reference csrc/cuda/ROIPool_cuda.cu:16-77, :79-108)
// behind detops_roi_pool_{forward,backward}_f32.  No reference model config uses ROIPool (the
// Pooler hard-codes ROIAlign, modeling/poolers.py:66); it is part of the exported operator API, so
// it gets the same workgroup-per-(ROI, channel chunk) layout as ROIAlign — the integer bin windows
// (hstart/hend, wstart/wend) are computed once per workgroup into LDS instead of once per output
// element per channel.  Backward (round 6): a gradient plane that fits the LDS is OWNED by one workgroup
// (roi_pool_bwd_owner_kernel: the plane of one (image, channel) is summed in LDS and written once, coalesced — no
// zero-fill pass, no global atomics); larger planes keep the reference's scatter with global atomics.
#include <float.h>

#include "detops_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBinsAxis = 64;  // PH, PW <= 64 use the LDS window table

struct RoiWin {
  int b, start_w, start_h;
  float bin_h, bin_w;
};

__device__ __forceinline__ RoiWin roi_window(const float* __restrict__ roi, float scale, int PH, int PW) {
#pragma clang fp contract(off)
  RoiWin r;
  r.b = static_cast<int>(roi[0]);
  r.start_w = static_cast<int>(roundf(roi[1] * scale));  // ROIPool_cuda.cu:30-33
  r.start_h = static_cast<int>(roundf(roi[2] * scale));
  const int end_w = static_cast<int>(roundf(roi[3] * scale));
  const int end_h = static_cast<int>(roundf(roi[4] * scale));
  const int rw = max(end_w - r.start_w + 1, 1);  // :36-37
  const int rh = max(end_h - r.start_h + 1, 1);
  r.bin_h = static_cast<float>(rh) / static_cast<float>(PH);
  r.bin_w = static_cast<float>(rw) / static_cast<float>(PW);
  return r;
}

__device__ __forceinline__ int2 window(int p, float bin, int start, int size) {
#pragma clang fp contract(off)
  int lo = static_cast<int>(floorf(static_cast<float>(p) * bin));       // :43-50
  int hi = static_cast<int>(ceilf(static_cast<float>(p + 1) * bin));
  lo = min(max(lo + start, 0), size);                                  // :53-56
  hi = min(max(hi + start, 0), size);
  return make_int2(lo, hi);
}

__global__ void __launch_bounds__(kBlock)
roi_pool_fwd_kernel(const float* __restrict__ in, const float* __restrict__ rois,
                    float* __restrict__ out, int32_t* __restrict__ argmax, int C, int H, int W,
                    int PH, int PW, float scale, int CT, int chunks) {
  __shared__ int2 winY[kMaxBinsAxis];
  __shared__ int2 winX[kMaxBinsAxis];
  const int k = blockIdx.x / chunks;
  const int chunk = blockIdx.x - k * chunks;
  const RoiWin r = roi_window(rois + static_cast<size_t>(k) * 5, scale, PH, PW);
  const bool tab = PH <= kMaxBinsAxis && PW <= kMaxBinsAxis;
  if (tab) {
    for (int t = threadIdx.x; t < PH + PW; t += kBlock) {
      if (t < PH) winY[t] = window(t, r.bin_h, r.start_h, H);
      else winX[t - PH] = window(t - PH, r.bin_w, r.start_w, W);
    }
    __syncthreads();
  }
  const int bins = PH * PW;
  const int c0 = chunk * CT, cend = min(C, c0 + CT);
  const int total = (cend - c0) * bins;
  const size_t plane = static_cast<size_t>(H) * W;
  const float* base = in + (static_cast<size_t>(r.b) * C + c0) * plane;
  const size_t obase = (static_cast<size_t>(k) * C + c0) * bins;
  for (int o = threadIdx.x; o < total; o += kBlock) {
    const int cl = o / bins;
    const int bin = o - cl * bins;
    const int ph = bin / PW, pw = bin - ph * PW;
    const int2 wy = tab ? winY[ph] : window(ph, r.bin_h, r.start_h, H);
    const int2 wx = tab ? winX[pw] : window(pw, r.bin_w, r.start_w, W);
    const bool empty = (wy.y <= wy.x) || (wx.y <= wx.x);
    float maxval = empty ? 0.f : -FLT_MAX;  // :60
    int maxidx = -1;
    const float* d = base + static_cast<size_t>(cl) * plane;
    for (int h = wy.x; h < wy.y; ++h)
      for (int w = wx.x; w < wx.y; ++w) {
        const float v = d[h * W + w];
        if (v > maxval) { maxval = v; maxidx = h * W + w; }  // strict >, first max wins (:68)
      }
    out[obase + o] = maxval;
    argmax[obase + o] = maxidx;
  }
}

__global__ void __launch_bounds__(kBlock)
roi_pool_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                    const int32_t* __restrict__ argmax, float* __restrict__ gin, int C, int H, int W,
                    int bins, int64_t total) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
  const size_t plane = static_cast<size_t>(H) * W;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += stride) {
    const int a = argmax[i];
    if (a < 0) continue;  // :100
    const int64_t kc = i / bins;
    const int k = static_cast<int>(kc / C);
    const int c = static_cast<int>(kc - static_cast<int64_t>(k) * C);
    const int b = static_cast<int>(rois[static_cast<size_t>(k) * 5]);
    atomicAdd(gin + (static_cast<size_t>(b) * C + c) * plane + a, gout[i]);
  }
}

// One workgroup per (channel, image): the H x W gradient plane lives in LDS.  The ROIs of the image are found in rounds of
// kBlock (their batch index is column 0 of the ROI row), then the lanes walk (ROI of the round, bin) pairs — consecutive lanes
// read consecutive bins of one (ROI, channel) block — and add into the plane with LDS atomics (several bins, and several ROIs,
// may share an argmax pixel; the order of those additions is the hardware's, as in the reference's atomicAdd).  The plane is
// written once (or added to what grad_in holds when the caller accumulates).
__global__ void __launch_bounds__(kBlock)
roi_pool_bwd_owner_kernel(const float* __restrict__ gout, const float* __restrict__ rois,
                          const int32_t* __restrict__ argmax, float* __restrict__ gin, int C, int plane, int K,
                          int bins, int accumulate) {
  DETOPS_DYNAMIC_LDS(float, map);            // [plane] floats
  __shared__ int s_list[kBlock];
  __shared__ int s_cnt;
  const int c = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x;
  for (int i = tid; i < plane; i += kBlock) map[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += kBlock) {
    if (tid == 0) s_cnt = 0;
    __syncthreads();                         // (also: the zero-fill / the previous round's additions are done)
    const int k = k0 + tid;
    if (k < K && static_cast<int>(rois[static_cast<size_t>(k) * 5]) == b) s_list[atomicAdd(&s_cnt, 1)] = k;
    __syncthreads();
    const int total = s_cnt * bins;
    constexpr int U = 4;                     // independent (argmax, gradient) loads in flight per lane
    for (int e0 = tid; e0 < total; e0 += U * kBlock) {
      int a[U];
      float g[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = min(e0 + u * kBlock, total - 1);
        const int j = e / bins, bin = e - j * bins;
        const size_t idx = (static_cast<size_t>(s_list[j]) * C + c) * bins + bin;
        a[u] = (e0 + u * kBlock < total) ? argmax[idx] : -1;
        g[u] = gout[idx];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (a[u] >= 0 && a[u] < plane) atomicAdd(&map[a[u]], g[u]);   // a < 0: empty bin (ROIPool_cuda.cu:100)
    }
    __syncthreads();                         // every lane has read the round's count and list before the next round resets them
  }
  float* dst = gin + (static_cast<size_t>(b) * C + c) * plane;
  for (int i = tid; i < plane; i += kBlock) dst[i] = accumulate ? dst[i] + map[i] : map[i];
}

constexpr size_t kOwnerMaxLds = 144 * 1024;  // one plane per workgroup; 160 KB per CU

}  // namespace

DETOPS_API int detops_roi_pool_forward_f32(const float* input, const float* rois, float* output,
                                           int32_t* argmax, int N, int C, int H, int W, int K,
                                           int PH, int PW, float spatial_scale,
                                           detops_stream_t stream) {
  if (N < 0 || C < 0 || H < 0 || W < 0 || K < 0 || PH <= 0 || PW <= 0) return DETOPS_EINVAL;
  if (K == 0 || C == 0) return 0;
  if (!input || !rois || !output || !argmax || N == 0 || H == 0 || W == 0) return DETOPS_EINVAL;
  int CT = 64;
  while (CT > 16 && static_cast<int64_t>(K) * ceil_div64(C, CT) < 4 * kNumCU) CT >>= 1;
  if (CT > C) CT = C;
  const int chunks = static_cast<int>(ceil_div64(C, CT));
  hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3(static_cast<unsigned>(K) * chunks), dim3(kBlock), 0,
                     as_stream(stream), input, rois, output, argmax, C, H, W, PH, PW, spatial_scale,
                     CT, chunks);
  return launch_status();
}

DETOPS_API int detops_roi_pool_backward_f32(const float* grad_out, const float* rois,
                                            const int32_t* argmax, float* grad_in, int N, int C,
                                            int H, int W, int K, int PH, int PW, int zero_grad_in,
                                            detops_stream_t stream) {
  if (N < 0 || C < 0 || H < 0 || W < 0 || K < 0 || PH <= 0 || PW <= 0) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const size_t bytes = sizeof(float) * static_cast<size_t>(N) * C * H * W;
  if (bytes && !grad_in) return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(K) * C * PH * PW;
  const size_t plane_bytes = sizeof(float) * static_cast<size_t>(H) * W;
  // the owner form: the plane fits the LDS and there are enough (image, channel) planes to fill the chip
  // (tuning roi_bwd_impl — tests, A/B: 3 forces the atomic scatter, 1 the owner form wherever the plane fits)
  const int impl = detops_tuning().roi_bwd_impl;
  const bool owner = total > 0 && bytes > 0 && plane_bytes <= kOwnerMaxLds && N <= 65535 && impl != 3 &&
                     static_cast<int64_t>(PH) * PW <= (1 << 20) && (static_cast<int64_t>(N) * C >= kNumCU || impl == 1);
  if (owner) {
    if (!grad_out || !rois || !argmax) return DETOPS_EINVAL;
    hipLaunchKernelGGL(roi_pool_bwd_owner_kernel, dim3(static_cast<unsigned>(C), static_cast<unsigned>(N)), dim3(kBlock),
                       plane_bytes, st, grad_out, rois, argmax, grad_in, C, H * W, K, PH * PW, zero_grad_in ? 0 : 1);
    return launch_status();
  }
  if (zero_grad_in && bytes) DETOPS_HIP_TRY(hipMemsetAsync(grad_in, 0, bytes, st));
  if (total == 0 || bytes == 0) return 0;
  if (!grad_out || !rois || !argmax) return DETOPS_EINVAL;
  const int blocks = static_cast<int>(std::min<int64_t>(ceil_div64(total, kBlock), kNumCU * 8));
  hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(blocks), dim3(kBlock), 0, st, grad_out, rois, argmax,
                     grad_in, C, H, W, PH * PW, total);
  return launch_status();
}
