// bias_act.hip — per-channel bias (+ ReLU) behind a channels-last convolution, forward and backward, for gfx950 (MI355X), fp32 / fp16 / bf16 storage (fp32 sums).
//
// The detector's convolutions WITH a bias (FPN lateral / output convolutions, the RPN head, the mask head: reference
// modeling/backbone/fpn.py:30-40, modeling/rpn/rpn.py:61-76, roi_heads/mask_head/roi_mask_feature_extractors.py:41-61,
// roi_mask_predictors.py:12-35) are `conv + bias [+ relu]`.  On a channels-last activation PyTorch runs that as up to three
// elementwise passes, and in the backward pass the bias gradient `grad.sum((0, 2, 3))` becomes a strided column reduction that
// its generic reduce kernel serves at ~0.1 TB/s (332 us per call on average, 2.3 ms of the fp32 step: profiles/r06d_*).
// Here: the convolution is called without its bias; the forward is the fused FrozenBN stream with scale 1 (frozen_bn.hip:
// y = [relu](x + b), one pass); the backward below is ONE pass that applies the ReLU mask, writes grad_x and accumulates
// the per-channel column sums in registers (a thread's channel window is loop-invariant), reduced per workgroup through LDS
// into `partials`; a second small launch adds the workgroups' partials in index order: deterministic, no atomics.
#include <algorithm>

#include "detops_common.h"

namespace {

constexpr int kBaThreads = 256;
constexpr int kBaV = 4;                               // floats per thread and vector
constexpr int kBaSpan = kBaThreads * kBaV;            // elements a workgroup covers per pass: C must divide it
constexpr int kBaMaxBlocks = 512;

template <typename T> struct BaIo;
template <> struct BaIo<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct BaIo<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct BaIo<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};
template <typename T> struct alignas(sizeof(T) * kBaV) BaVec { T v[kBaV]; };

// g = relu ? (y > 0 ? gy : 0) : gy ;  grad_x = g ;  partial[block][c] = sum over the block's rows of g[., c]  (fp32 sums;
// T = float | __half | __hip_bfloat16 storage: the autocast configurations)
template <typename T, bool kRelu>
__global__ void __launch_bounds__(kBaThreads)
bias_act_bwd_nhwc_kernel(const T* __restrict__ gy, const T* __restrict__ y, T* __restrict__ gx,
                         float* __restrict__ partials, int C, int64_t nvec) {
  __shared__ float red[kBaSpan];
  const int tid = threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kBaThreads;      // in vectors; stride * 4 is a multiple of C
  float acc[kBaV] = {0.f, 0.f, 0.f, 0.f};
  using VT = BaVec<T>;
  const VT* gv = reinterpret_cast<const VT*>(gy);
  const VT* yv = reinterpret_cast<const VT*>(y);
  VT* xo = reinterpret_cast<VT*>(gx);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBaThreads + tid; i < nvec; i += stride) {
    VT g = gv[i];
    if (kRelu) {
      const VT m = yv[i];
#pragma unroll
      for (int j = 0; j < kBaV; ++j)
        if (!(BaIo<T>::ld(m.v[j]) > 0.f)) g.v[j] = BaIo<T>::st(0.f);
      xo[i] = g;
    } else if (gx != gy) {
      xo[i] = g;
    }
#pragma unroll
    for (int j = 0; j < kBaV; ++j) acc[j] += BaIo<T>::ld(g.v[j]);
  }
  // threads t, t + C/4, t + 2C/4, ... hold sums of the same channels: add them in thread order
#pragma unroll
  for (int j = 0; j < kBaV; ++j) red[tid * kBaV + j] = acc[j];
  __syncthreads();
  if (tid * kBaV < C) {
    float s[kBaV] = {0.f, 0.f, 0.f, 0.f};
    for (int o = tid * kBaV; o < kBaSpan; o += C)
#pragma unroll
      for (int j = 0; j < kBaV; ++j) s[j] += red[o + j];
    *reinterpret_cast<float4*>(partials + static_cast<size_t>(blockIdx.x) * C + tid * kBaV) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

// grad_bias[c] = sum_b partials[b][c]: a workgroup per 32 channels, 32 lanes of partial rows per channel (lane r adds rows
// r, r + 32, ... in order), then the 32 lane sums in lane order through LDS — a fixed order, and 32x shorter dependent
// chains than one thread per channel (which took 156 us for 1024 partial rows).
constexpr int kBfCh = 32, kBfRows = 32;
__global__ void __launch_bounds__(kBfCh * kBfRows)
bias_grad_finish_kernel(const float* __restrict__ partials, float* __restrict__ gb, int C, int blocks) {
  __shared__ float red[kBfRows][kBfCh + 1];
  const int cl = threadIdx.x % kBfCh, r = threadIdx.x / kBfCh;
  const int c = static_cast<int>(blockIdx.x) * kBfCh + cl;
  float s = 0.f;
  if (c < C)
    for (int b = r; b < blocks; b += kBfRows) s += partials[static_cast<size_t>(b) * C + c];
  red[r][cl] = s;
  __syncthreads();
  if (r == 0 && c < C) {
    float t = red[0][cl];
    for (int q = 1; q < kBfRows; ++q) t += red[q][cl];
    gb[c] = t;
  }
}

// Column sums of a [rows, C] matrix for ANY C <= 256 (the RPN's 3 / 12-channel outputs, the mask logits' 81): a thread is
// (row lane, channel) with the channel count padded to a power of two, consecutive threads read consecutive addresses;
// per-workgroup partial sums in row-lane order through LDS, then bias_grad_finish_kernel.
template <typename T>
__global__ void __launch_bounds__(kBaThreads)
column_sum_kernel(const T* __restrict__ x, float* __restrict__ partials, int C, int Cp, int64_t rows) {
  __shared__ float red[kBaThreads];
  const int tid = threadIdx.x;
  const int c = tid % Cp, rl = tid / Cp, rpb = kBaThreads / Cp;      // rows per workgroup pass
  float s = 0.f;
  if (c < C)
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * rpb + rl; r < rows; r += static_cast<int64_t>(gridDim.x) * rpb)
      s += BaIo<T>::ld(x[r * C + c]);
  red[tid] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = red[c];
    for (int q = 1; q < rpb; ++q) t += red[q * Cp + c];
    partials[static_cast<size_t>(blockIdx.x) * C + c] = t;
  }
}

int ba_blocks(int64_t nvec, int C) {
  int64_t blocks = ceil_div64(nvec, static_cast<int64_t>(kBaThreads) * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > kBaMaxBlocks) blocks = kBaMaxBlocks;
  return static_cast<int>(blocks);
}

}  // namespace

// C must divide 1024 (every width of the detector's biased convolutions: 256) and be a multiple of 4
DETOPS_API int detops_bias_act_supported(int C) { return (C >= 4 && C <= kBaSpan && kBaSpan % C == 0 && C % 4 == 0) ? 1 : 0; }

DETOPS_API size_t detops_bias_act_backward_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || !detops_bias_act_supported(C)) return 0;
  return sizeof(float) * static_cast<size_t>(ba_blocks(rows * C / kBaV, C)) * C;
}

namespace {
template <typename T>
int bias_act_backward(const void* grad_y, const void* y, void* grad_x, float* grad_bias, int64_t rows, int C, int relu, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  const int64_t nvec = rows * C / kBaV;
  const int blocks = ba_blocks(nvec, C);
  if (workspace_bytes < sizeof(float) * static_cast<size_t>(blocks) * C) return DETOPS_EINVAL;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x) | reinterpret_cast<uintptr_t>(y);
  if (bits & (sizeof(T) * kBaV - 1)) return DETOPS_EINVAL;
  float* partials = static_cast<float*>(workspace);
  const T* gp = static_cast<const T*>(grad_y);
  const T* yp = static_cast<const T*>(y);
  T* xp = static_cast<T*>(grad_x);
  if (relu)
    hipLaunchKernelGGL((bias_act_bwd_nhwc_kernel<T, true>), dim3(blocks), dim3(kBaThreads), 0, st, gp, yp, xp, partials, C, nvec);
  else
    hipLaunchKernelGGL((bias_act_bwd_nhwc_kernel<T, false>), dim3(blocks), dim3(kBaThreads), 0, st, gp, yp, xp, partials, C, nvec);
  int e = launch_status();
  if (e) return e;
  hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(static_cast<unsigned>(ceil_div64(C, kBfCh))), dim3(kBfCh * kBfRows), 0, st, partials,
                     grad_bias, C, blocks);
  return launch_status();
}

}  // namespace

// grad_y / y / grad_x: dtype code DETOPS_F32 | F16 | BF16 (one per call); grad_bias and the partial sums are fp32
DETOPS_API int detops_bias_act_backward_nhwc(const void* grad_y, const void* y, void* grad_x, float* grad_bias, int dtype,
                                             int64_t rows, int C, int relu, void* workspace, size_t workspace_bytes,
                                             detops_stream_t stream) {
  if (rows < 0 || C < 0) return DETOPS_EINVAL;
  if (!detops_bias_act_supported(C)) return DETOPS_EUNSUPPORTED;
  if (!grad_y || !grad_x || !grad_bias || (relu && !y) || !workspace) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (rows == 0) { DETOPS_HIP_TRY(hipMemsetAsync(grad_bias, 0, sizeof(float) * C, st)); return 0; }
  switch (dtype) {
    case DETOPS_F32: return bias_act_backward<float>(grad_y, y, grad_x, grad_bias, rows, C, relu, workspace, workspace_bytes, st);
    case DETOPS_F16: return bias_act_backward<__half>(grad_y, y, grad_x, grad_bias, rows, C, relu, workspace, workspace_bytes, st);
    case DETOPS_BF16: return bias_act_backward<__hip_bfloat16>(grad_y, y, grad_x, grad_bias, rows, C, relu, workspace, workspace_bytes, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_bias_act_backward_nhwc_f32(const float* grad_y, const float* y, float* grad_x, float* grad_bias,
                                                 int64_t rows, int C, int relu, void* workspace, size_t workspace_bytes,
                                                 detops_stream_t stream) {
  return detops_bias_act_backward_nhwc(grad_y, y, grad_x, grad_bias, DETOPS_F32, rows, C, relu, workspace, workspace_bytes, stream);
}

namespace {
template <typename T>
int column_sum_run(const void* x, float* out, int64_t rows, int C, void* workspace, size_t workspace_bytes, hipStream_t st) {
  int Cp = 1;
  while (Cp < C) Cp <<= 1;
  const int rpb = kBaThreads / Cp;
  int64_t blocks = ceil_div64(rows, static_cast<int64_t>(rpb) * 16);
  blocks = std::max<int64_t>(1, std::min<int64_t>(blocks, kBaMaxBlocks));
  if (workspace_bytes < sizeof(float) * static_cast<size_t>(blocks) * C) return DETOPS_EINVAL;
  float* partials = static_cast<float*>(workspace);
  hipLaunchKernelGGL(column_sum_kernel<T>, dim3(static_cast<unsigned>(blocks)), dim3(kBaThreads), 0, st, static_cast<const T*>(x), partials, C,
                     Cp, rows);
  int e = launch_status();
  if (e) return e;
  hipLaunchKernelGGL(bias_grad_finish_kernel, dim3(static_cast<unsigned>(ceil_div64(C, kBfCh))), dim3(kBfCh * kBfRows), 0, st, partials, out,
                     C, static_cast<int>(blocks));
  return launch_status();
}
}  // namespace

// out[c] = sum_r x[r, c] for a row-major [rows, C] fp32 matrix, C <= 256 (any value); deterministic.  workspace:
// detops_column_sum_workspace_bytes(rows, C) bytes.
DETOPS_API size_t detops_column_sum_workspace_bytes(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C > kBaThreads) return 0;
  return sizeof(float) * static_cast<size_t>(kBaMaxBlocks) * C;
}

// x: dtype code DETOPS_F32 | F16 | BF16; out fp32
DETOPS_API int detops_column_sum(const void* x, float* out, int dtype, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                                 detops_stream_t stream) {
  if (rows < 0 || C <= 0) return DETOPS_EINVAL;
  if (C > kBaThreads) return DETOPS_EUNSUPPORTED;
  if (!out || (rows > 0 && (!x || !workspace))) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (rows == 0) { DETOPS_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * C, st)); return 0; }
  switch (dtype) {
    case DETOPS_F32: return column_sum_run<float>(x, out, rows, C, workspace, workspace_bytes, st);
    case DETOPS_F16: return column_sum_run<__half>(x, out, rows, C, workspace, workspace_bytes, st);
    case DETOPS_BF16: return column_sum_run<__hip_bfloat16>(x, out, rows, C, workspace, workspace_bytes, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_column_sum_f32(const float* x, float* out, int64_t rows, int C, void* workspace, size_t workspace_bytes,
                                     detops_stream_t stream) {
  return detops_column_sum(x, out, DETOPS_F32, rows, C, workspace, workspace_bytes, stream);
}
