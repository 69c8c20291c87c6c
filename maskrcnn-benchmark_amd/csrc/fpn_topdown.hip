// fpn_topdown.hip — the feature pyramid's top-down step for gfx950 (MI355X): out = lateral + nearest_upsample(top).
//
// Replaces `F.interpolate(last_inner, scale_factor=2, mode="nearest")` + `inner_lateral + inner_top_down` of the reference's
// FPN (modeling/backbone/fpn.py:59-64): two launches and a full-size temporary per pyramid level (137 MB written and read
// again at P2 for a 2-image batch) become ONE streaming pass; the backward of the pair (an identity towards the lateral, a
// block sum towards the coarser map) is one pass over the gradient.  NCHW and channels-last (NHWC) forms, fp32 / fp16 / bf16 storage, fp32 arithmetic.
// Source index = ATen's nearest rule, min(int(floorf(dst * (float)in / out)), in - 1) — any size ratio, not only 2x.
#include "detops_common.h"

namespace {

constexpr int kTd = 256;

template <typename T> struct Io;
template <> struct Io<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct Io<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct Io<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};

__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  const int s = static_cast<int>(floorf(static_cast<float>(dst) * scale));
  return s < in_size - 1 ? s : in_size - 1;
}

// grid (planes, row groups); a thread owns V consecutive columns of a row
template <typename T, int V>
__global__ void __launch_bounds__(kTd)
topdown_fwd_kernel(const T* __restrict__ lat, const T* __restrict__ top, T* __restrict__ out, int H, int W, int h, int w,
                   float sh, float sw) {
  const size_t plane = blockIdx.x;
  const T* l = lat + plane * H * W;
  const T* t = top + plane * h * w;
  T* o = out + plane * H * W;
  const int wv = W / V;
  for (int i = blockIdx.y * kTd + threadIdx.x; i < H * wv; i += gridDim.y * kTd) {
    const int y = i / wv, x0 = (i - y * wv) * V;
    const T* trow = t + static_cast<size_t>(nearest_src(y, sh, h)) * w;
    struct alignas(sizeof(T) * V) Vec { T v[V]; };
    const Vec a = *reinterpret_cast<const Vec*>(l + static_cast<size_t>(y) * W + x0);
    Vec r;
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = Io<T>::st(Io<T>::ld(a.v[j]) + Io<T>::ld(trow[nearest_src(x0 + j, sw, w)]));
    *reinterpret_cast<Vec*>(o + static_cast<size_t>(y) * W + x0) = r;
  }
}

// grad_top[r][c] = sum of grad[y][x] over the fine pixels whose nearest source is (r, c); a thread per coarse pixel.  The
// candidates are the few rows / columns around r * H / h; the FORWARD's index rule decides, so the two passes agree for
// every size ratio.  Fixed summation order (row-major): deterministic.
template <typename T>
__global__ void __launch_bounds__(kTd)
topdown_bwd_kernel(const T* __restrict__ g, T* __restrict__ gtop, int H, int W, int h, int w, float sh, float sw) {
  const size_t plane = blockIdx.x;
  const T* gp = g + plane * H * W;
  T* o = gtop + plane * h * w;
  for (int i = blockIdx.y * kTd + threadIdx.x; i < h * w; i += gridDim.y * kTd) {
    const int r = i / w, c = i - r * w;
    const int y_lo = max(0, static_cast<int>((static_cast<long long>(r) * H) / h) - 1);
    const int y_hi = min(H - 1, static_cast<int>((static_cast<long long>(r + 1) * H) / h) + 1);
    const int x_lo = max(0, static_cast<int>((static_cast<long long>(c) * W) / w) - 1);
    const int x_hi = min(W - 1, static_cast<int>((static_cast<long long>(c + 1) * W) / w) + 1);
    float acc = 0.f;
    for (int y = y_lo; y <= y_hi; ++y) {
      if (nearest_src(y, sh, h) != r) continue;
      for (int x = x_lo; x <= x_hi; ++x)
        if (nearest_src(x, sw, w) == c) acc += Io<T>::ld(gp[static_cast<size_t>(y) * W + x]);
    }
    o[i] = Io<T>::st(acc);
  }
}

template <typename T>
int run_fwd(const void* lat, const void* top, void* out, int planes, int H, int W, int h, int w, hipStream_t st) {
  const float sh = static_cast<float>(h) / static_cast<float>(H), sw = static_cast<float>(w) / static_cast<float>(W);
  const uintptr_t bits = reinterpret_cast<uintptr_t>(lat) | reinterpret_cast<uintptr_t>(out);
  const int vmax = 16 / static_cast<int>(sizeof(T));
  int V = 1;
  for (int v = vmax; v > 1; v >>= 1)
    if (W % v == 0 && bits % (v * sizeof(T)) == 0) { V = v; break; }
  const int64_t work = static_cast<int64_t>(H) * (W / V);
  const int gy = static_cast<int>(std::min<int64_t>(64, std::max<int64_t>(1, ceil_div64(work, kTd * 4))));
  const dim3 grid(planes, gy);
  const T* l = static_cast<const T*>(lat);
  const T* t = static_cast<const T*>(top);
  T* o = static_cast<T*>(out);
  switch (V) {
    case 8: hipLaunchKernelGGL((topdown_fwd_kernel<T, (sizeof(T) == 2 ? 8 : 1)>), grid, dim3(kTd), 0, st, l, t, o, H, W, h, w, sh, sw); break;
    case 4: hipLaunchKernelGGL((topdown_fwd_kernel<T, 4>), grid, dim3(kTd), 0, st, l, t, o, H, W, h, w, sh, sw); break;
    case 2: hipLaunchKernelGGL((topdown_fwd_kernel<T, 2>), grid, dim3(kTd), 0, st, l, t, o, H, W, h, w, sh, sw); break;
    default: hipLaunchKernelGGL((topdown_fwd_kernel<T, 1>), grid, dim3(kTd), 0, st, l, t, o, H, W, h, w, sh, sw); break;
  }
  return launch_status();
}

template <typename T>
int run_bwd(const void* g, void* gtop, int planes, int H, int W, int h, int w, hipStream_t st) {
  const float sh = static_cast<float>(h) / static_cast<float>(H), sw = static_cast<float>(w) / static_cast<float>(W);
  const int gy = static_cast<int>(std::min<int64_t>(64, std::max<int64_t>(1, ceil_div64(static_cast<int64_t>(h) * w, kTd * 2))));
  hipLaunchKernelGGL((topdown_bwd_kernel<T>), dim3(planes, gy), dim3(kTd), 0, st, static_cast<const T*>(g), static_cast<T*>(gtop),
                     H, W, h, w, sh, sw);
  return launch_status();
}

// ---- channels-last (NHWC) forms: lateral / out [N, H, W, C], top [N, h, w, C]; a thread owns V consecutive channels of one
// fine pixel (forward) or of one coarse pixel (backward), so every access is a 16-byte vector and the nearest-source index is
// computed once per vector instead of once per element.
template <typename T, int V>
__global__ void __launch_bounds__(kTd)
topdown_fwd_nhwc_kernel(const T* __restrict__ lat, const T* __restrict__ top, T* __restrict__ out, int64_t total_vec, int C, int H,
                        int W, int h, int w, float sh, float sw) {
  struct alignas(sizeof(T) * V) Vec { T v[V]; };
  const int cv = C / V;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kTd + threadIdx.x; i < total_vec; i += static_cast<int64_t>(gridDim.x) * kTd) {
    const int64_t pix = i / cv;
    const int c = static_cast<int>(i - pix * cv) * V;
    const int x = static_cast<int>(pix % W);
    const int64_t ny = pix / W;
    const int y = static_cast<int>(ny % H);
    const int64_t n = ny / H;
    const int64_t src = ((n * h + nearest_src(y, sh, h)) * w + nearest_src(x, sw, w)) * C + c;
    const Vec a = *reinterpret_cast<const Vec*>(lat + i * V);
    const Vec t = *reinterpret_cast<const Vec*>(top + src);
    Vec r;
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = Io<T>::st(Io<T>::ld(a.v[j]) + Io<T>::ld(t.v[j]));
    *reinterpret_cast<Vec*>(out + i * V) = r;
  }
}

template <typename T, int V>
__global__ void __launch_bounds__(kTd)
topdown_bwd_nhwc_kernel(const T* __restrict__ g, T* __restrict__ gtop, int64_t total_vec, int C, int H, int W, int h, int w,
                        float sh, float sw) {
  struct alignas(sizeof(T) * V) Vec { T v[V]; };
  const int cv = C / V;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kTd + threadIdx.x; i < total_vec; i += static_cast<int64_t>(gridDim.x) * kTd) {
    const int64_t pix = i / cv;
    const int cc = static_cast<int>(i - pix * cv) * V;
    const int c = static_cast<int>(pix % w);
    const int64_t nr = pix / w;
    const int r = static_cast<int>(nr % h);
    const int64_t n = nr / h;
    const int y_lo = max(0, static_cast<int>((static_cast<long long>(r) * H) / h) - 1);
    const int y_hi = min(H - 1, static_cast<int>((static_cast<long long>(r + 1) * H) / h) + 1);
    const int x_lo = max(0, static_cast<int>((static_cast<long long>(c) * W) / w) - 1);
    const int x_hi = min(W - 1, static_cast<int>((static_cast<long long>(c + 1) * W) / w) + 1);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int y = y_lo; y <= y_hi; ++y) {
      if (nearest_src(y, sh, h) != r) continue;
      for (int x = x_lo; x <= x_hi; ++x) {
        if (nearest_src(x, sw, w) != c) continue;
        const Vec v = *reinterpret_cast<const Vec*>(g + ((n * H + y) * W + x) * C + cc);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += Io<T>::ld(v.v[j]);
      }
    }
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = Io<T>::st(acc[j]);
    *reinterpret_cast<Vec*>(gtop + i * V) = o;
  }
}

template <typename T>
static int nhwc_vec(int C, const void* a, const void* b, const void* c) {
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
  for (int v = 16 / static_cast<int>(sizeof(T)); v > 1; v >>= 1)
    if (C % v == 0 && bits % (v * sizeof(T)) == 0) return v;
  return 1;
}

static inline unsigned nhwc_blocks(int64_t total_vec) {
  return static_cast<unsigned>(std::min<int64_t>(1 << 16, std::max<int64_t>(1, ceil_div64(total_vec, kTd * 2))));
}

template <typename T>
int run_fwd_nhwc(const void* lat, const void* top, void* out, int N, int C, int H, int W, int h, int w, hipStream_t st) {
  const float sh = static_cast<float>(h) / static_cast<float>(H), sw = static_cast<float>(w) / static_cast<float>(W);
  const T* l = static_cast<const T*>(lat);
  const T* t = static_cast<const T*>(top);
  T* o = static_cast<T*>(out);
  const int V = nhwc_vec<T>(C, lat, top, out);
  const int64_t total_vec = static_cast<int64_t>(N) * H * W * (C / V);
  const dim3 grid(nhwc_blocks(total_vec));
#define TD_LAUNCH(VV) hipLaunchKernelGGL((topdown_fwd_nhwc_kernel<T, VV>), grid, dim3(kTd), 0, st, l, t, o, total_vec, C, H, W, h, w, sh, sw)
  switch (V) {
    case 8: TD_LAUNCH((sizeof(T) == 2 ? 8 : 1)); break;
    case 4: TD_LAUNCH(4); break;
    case 2: TD_LAUNCH(2); break;
    default: TD_LAUNCH(1); break;
  }
#undef TD_LAUNCH
  return launch_status();
}

template <typename T>
int run_bwd_nhwc(const void* g, void* gtop, int N, int C, int H, int W, int h, int w, hipStream_t st) {
  const float sh = static_cast<float>(h) / static_cast<float>(H), sw = static_cast<float>(w) / static_cast<float>(W);
  const T* gp = static_cast<const T*>(g);
  T* o = static_cast<T*>(gtop);
  const int V = nhwc_vec<T>(C, g, gtop, nullptr);
  const int64_t total_vec = static_cast<int64_t>(N) * h * w * (C / V);
  const dim3 grid(nhwc_blocks(total_vec));
#define TD_LAUNCH(VV) hipLaunchKernelGGL((topdown_bwd_nhwc_kernel<T, VV>), grid, dim3(kTd), 0, st, gp, o, total_vec, C, H, W, h, w, sh, sw)
  switch (V) {
    case 8: TD_LAUNCH((sizeof(T) == 2 ? 8 : 1)); break;
    case 4: TD_LAUNCH(4); break;
    case 2: TD_LAUNCH(2); break;
    default: TD_LAUNCH(1); break;
  }
#undef TD_LAUNCH
  return launch_status();
}

bool bad_shape(int planes, int H, int W, int h, int w) {
  return planes < 0 || H < 0 || W < 0 || h <= 0 || w <= 0 || static_cast<int64_t>(H) * W > 0x7fffffff ||
         static_cast<int64_t>(h) * w > 0x7fffffff;
}

}  // namespace

DETOPS_API int detops_fpn_topdown_forward(const void* lateral, const void* top, void* out, int dtype, int planes, int H, int W,
                                          int h, int w, detops_stream_t stream) {
  if (bad_shape(planes, H, W, h, w)) return DETOPS_EINVAL;
  if (planes == 0 || H == 0 || W == 0) return 0;
  if (!lateral || !top || !out) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case DETOPS_F32: return run_fwd<float>(lateral, top, out, planes, H, W, h, w, st);
    case DETOPS_F16: return run_fwd<__half>(lateral, top, out, planes, H, W, h, w, st);
    case DETOPS_BF16: return run_fwd<__hip_bfloat16>(lateral, top, out, planes, H, W, h, w, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_fpn_topdown_backward(const void* grad_out, void* grad_top, int dtype, int planes, int H, int W, int h, int w,
                                           detops_stream_t stream) {
  if (bad_shape(planes, H, W, h, w)) return DETOPS_EINVAL;
  if (planes == 0) return 0;
  if (!grad_out || !grad_top) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case DETOPS_F32: return run_bwd<float>(grad_out, grad_top, planes, H, W, h, w, st);
    case DETOPS_F16: return run_bwd<__half>(grad_out, grad_top, planes, H, W, h, w, st);
    case DETOPS_BF16: return run_bwd<__hip_bfloat16>(grad_out, grad_top, planes, H, W, h, w, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_fpn_topdown_forward_nhwc(const void* lateral, const void* top, void* out, int dtype, int N, int C, int H, int W,
                                               int h, int w, detops_stream_t stream) {
  if (N < 0 || C < 0 || bad_shape(1, H, W, h, w)) return DETOPS_EINVAL;
  if (N == 0 || C == 0 || H == 0 || W == 0) return 0;
  if (!lateral || !top || !out) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case DETOPS_F32: return run_fwd_nhwc<float>(lateral, top, out, N, C, H, W, h, w, st);
    case DETOPS_F16: return run_fwd_nhwc<__half>(lateral, top, out, N, C, H, W, h, w, st);
    case DETOPS_BF16: return run_fwd_nhwc<__hip_bfloat16>(lateral, top, out, N, C, H, W, h, w, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_fpn_topdown_backward_nhwc(const void* grad_out, void* grad_top, int dtype, int N, int C, int H, int W, int h,
                                                int w, detops_stream_t stream) {
  if (N < 0 || C < 0 || bad_shape(1, H, W, h, w)) return DETOPS_EINVAL;
  if (N == 0 || C == 0) return 0;
  if (!grad_out || !grad_top) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case DETOPS_F32: return run_bwd_nhwc<float>(grad_out, grad_top, N, C, H, W, h, w, st);
    case DETOPS_F16: return run_bwd_nhwc<__half>(grad_out, grad_top, N, C, H, W, h, w, st);
    case DETOPS_BF16: return run_bwd_nhwc<__hip_bfloat16>(grad_out, grad_top, N, C, H, W, h, w, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}
