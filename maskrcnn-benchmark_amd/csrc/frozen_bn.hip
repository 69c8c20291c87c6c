// frozen_bn.hip — fused FrozenBatchNorm2d affine (+ residual add) (+ ReLU), forward and backward,
// for gfx950 (MI355X).  NCHW and channels-last (NHWC) forms, fp32 / fp16 / bf16 storage, fp32 arithmetic.
//
// Replaces the elementwise chain the reference backbone runs after every convolution
// (layers/batch_norm.py:19-31  `x * scale + bias`, then `F.relu_`, and in the bottleneck's tail
// `out += identity; F.relu_(out)`, modeling/backbone/resnet.py:343-366): three to four full passes
// over the activation in PyTorch become ONE streaming pass (read x [+ residual], write y), and the
// backward's ReLU-mask + scale becomes one pass as well.  Pure HBM streaming: one workgroup row
// per (n, c) plane so scale/bias are wave-uniform scalars, 16-byte vector accesses when the plane
// size allows it.
#include "detops_common.h"

namespace {

constexpr int kThreads = 256;

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float load(const float v) { return v; }
  static __device__ __forceinline__ float store(float v) { return v; }
};
template <> struct Cvt<__half> {
  static __device__ __forceinline__ float load(const __half v) { return __half2float(v); }
  static __device__ __forceinline__ __half store(float v) { return __float2half(v); }
};
template <> struct Cvt<__hip_bfloat16> {
  static __device__ __forceinline__ float load(const __hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 store(float v) { return __float2bfloat16(v); }
};

template <typename T, int V> struct alignas(sizeof(T) * V) Vec { T v[V]; };

// y = [relu]( x * scale[c] + bias[c] [+ residual] )
template <typename T, int V, bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
frozen_bn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ bias,
                     const T* __restrict__ res, T* __restrict__ y, int C, int HW) {
#pragma clang fp contract(off)
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float s = scale[c], b = bias[c];
  const size_t base = static_cast<size_t>(plane) * HW;
  const int nvec = HW / V;
  using VT = Vec<T, V>;
  const VT* xv = reinterpret_cast<const VT*>(x + base);
  const VT* rv = kRes ? reinterpret_cast<const VT*>(res + base) : nullptr;
  VT* yv = reinterpret_cast<VT*>(y + base);
  for (int i = blockIdx.y * kThreads + threadIdx.x; i < nvec; i += gridDim.y * kThreads) {
    const VT a = xv[i];
    VT r;
    if (kRes) r = rv[i];
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = Cvt<T>::load(a.v[j]) * s + b;       // torch: (x * scale) + bias, two roundings in fp32
      if (kRes) t = t + Cvt<T>::load(r.v[j]);
      if (kRelu) t = t > 0.f ? t : 0.f;
      o.v[j] = Cvt<T>::store(t);
    }
    yv[i] = o;
  }
}

// g = relu ? (y > 0 ? gy : 0) : gy ;  grad_x = g * scale[c] ;  grad_res = g (optional)
template <typename T, int V, bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
frozen_bn_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, const float* __restrict__ scale,
                     T* __restrict__ gx, T* __restrict__ gres, int C, int HW) {
  const int plane = blockIdx.x;
  const float s = scale[plane % C];
  const size_t base = static_cast<size_t>(plane) * HW;
  const int nvec = HW / V;
  using VT = Vec<T, V>;
  const VT* gv = reinterpret_cast<const VT*>(gy + base);
  const VT* yv = kRelu ? reinterpret_cast<const VT*>(y + base) : nullptr;
  VT* xo = reinterpret_cast<VT*>(gx + base);
  VT* ro = kRes ? reinterpret_cast<VT*>(gres + base) : nullptr;
  for (int i = blockIdx.y * kThreads + threadIdx.x; i < nvec; i += gridDim.y * kThreads) {
    const VT g = gv[i];
    VT m;
    if (kRelu) m = yv[i];
    VT ox, orr;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = Cvt<T>::load(g.v[j]);
      if (kRelu && !(Cvt<T>::load(m.v[j]) > 0.f)) t = 0.f;
      ox.v[j] = Cvt<T>::store(t * s);
      if (kRes) orr.v[j] = Cvt<T>::store(t);
    }
    xo[i] = ox;
    if (kRes) ro[i] = orr;
  }
}

// ---- channels-last (NHWC) forms: the activation is [rows = N*H*W, C] with the channel as the fastest index, so a thread's
// V consecutive elements are V consecutive CHANNELS (C % V == 0) and scale / bias arrive as vectors.  The grid stride is a
// multiple of C whenever the host can arrange it (every backbone width is a power of two): the thread's channel window and
// its scale / bias registers are then loop-invariant; otherwise they are re-derived per vector.
template <typename T, int V, bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
frozen_bn_fwd_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ bias,
                          const T* __restrict__ res, T* __restrict__ y, int C, int64_t nvec) {
#pragma clang fp contract(off)
  using VT = Vec<T, V>;
  const VT* xv = reinterpret_cast<const VT*>(x);
  const VT* rv = kRes ? reinterpret_cast<const VT*>(res) : nullptr;
  VT* yv = reinterpret_cast<VT*>(y);
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  const bool invariant = (stride * V) % C == 0;
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  float s[V], b[V];
  int c0 = static_cast<int>((i * V) % C);
#pragma unroll
  for (int j = 0; j < V; ++j) { s[j] = scale[c0 + j]; b[j] = bias[c0 + j]; }
  for (; i < nvec; i += stride) {
    if (!invariant) {
      c0 = static_cast<int>((i * V) % C);
#pragma unroll
      for (int j = 0; j < V; ++j) { s[j] = scale[c0 + j]; b[j] = bias[c0 + j]; }
    }
    const VT a = xv[i];
    VT r;
    if (kRes) r = rv[i];
    VT o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = Cvt<T>::load(a.v[j]) * s[j] + b[j];
      if (kRes) t = t + Cvt<T>::load(r.v[j]);
      if (kRelu) t = t > 0.f ? t : 0.f;
      o.v[j] = Cvt<T>::store(t);
    }
    yv[i] = o;
  }
}

template <typename T, int V, bool kRelu, bool kRes>
__global__ void __launch_bounds__(kThreads)
frozen_bn_bwd_nhwc_kernel(const T* __restrict__ gy, const T* __restrict__ y, const float* __restrict__ scale,
                          T* __restrict__ gx, T* __restrict__ gres, int C, int64_t nvec) {
  using VT = Vec<T, V>;
  const VT* gv = reinterpret_cast<const VT*>(gy);
  const VT* yv = kRelu ? reinterpret_cast<const VT*>(y) : nullptr;
  VT* xo = reinterpret_cast<VT*>(gx);
  VT* ro = kRes ? reinterpret_cast<VT*>(gres) : nullptr;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreads;
  const bool invariant = (stride * V) % C == 0;
  int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
  float s[V];
  int c0 = static_cast<int>((i * V) % C);
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = scale[c0 + j];
  for (; i < nvec; i += stride) {
    if (!invariant) {
      c0 = static_cast<int>((i * V) % C);
#pragma unroll
      for (int j = 0; j < V; ++j) s[j] = scale[c0 + j];
    }
    const VT g = gv[i];
    VT m;
    if (kRelu) m = yv[i];
    VT ox, orr;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = Cvt<T>::load(g.v[j]);
      if (kRelu && !(Cvt<T>::load(m.v[j]) > 0.f)) t = 0.f;
      ox.v[j] = Cvt<T>::store(t * s[j]);
      if (kRes) orr.v[j] = Cvt<T>::store(t);
    }
    xo[i] = ox;
    if (kRes) ro[i] = orr;
  }
}

// grid of the channels-last kernels: ~8 vectors per thread, a whole number of channel periods per grid stride when C allows
static inline unsigned nhwc_grid(int64_t nvec, int C, int V) {
  int64_t blocks = ceil_div64(nvec, static_cast<int64_t>(kThreads) * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 16384) blocks = 16384;
  const int64_t per_block = static_cast<int64_t>(kThreads) * V;      // elements one block covers per pass
  if (C % per_block != 0 && per_block % C != 0) return static_cast<unsigned>(blocks);   // no period to respect
  const int64_t unit = C > per_block ? C / per_block : 1;            // blocks per channel period
  blocks = ceil_div64(blocks, unit) * unit;
  return static_cast<unsigned>(blocks);
}

template <typename T, int V>
int launch_fwd_nhwc(const void* x, const float* scale, const float* bias, const void* res, void* y, int64_t n, int C,
                    int relu, hipStream_t st) {
  const int64_t nvec = n / V;
  const dim3 grid(nhwc_grid(nvec, C, V));
  const T* xp = static_cast<const T*>(x);
  const T* rp = static_cast<const T*>(res);
  T* yp = static_cast<T*>(y);
#define FB_LAUNCH(R, S) hipLaunchKernelGGL((frozen_bn_fwd_nhwc_kernel<T, V, R, S>), grid, dim3(kThreads), 0, st, xp, scale, bias, rp, yp, C, nvec)
  if (relu) { if (res) FB_LAUNCH(true, true); else FB_LAUNCH(true, false); }
  else      { if (res) FB_LAUNCH(false, true); else FB_LAUNCH(false, false); }
#undef FB_LAUNCH
  return launch_status();
}

template <typename T, int V>
int launch_bwd_nhwc(const void* gy, const void* y, const float* scale, void* gx, void* gres, int64_t n, int C, int relu,
                    hipStream_t st) {
  const int64_t nvec = n / V;
  const dim3 grid(nhwc_grid(nvec, C, V));
  const T* gp = static_cast<const T*>(gy);
  const T* yp = static_cast<const T*>(y);
  T* xp = static_cast<T*>(gx);
  T* rp = static_cast<T*>(gres);
#define FB_LAUNCH(R, S) hipLaunchKernelGGL((frozen_bn_bwd_nhwc_kernel<T, V, R, S>), grid, dim3(kThreads), 0, st, gp, yp, scale, xp, rp, C, nvec)
  if (relu) { if (gres) FB_LAUNCH(true, true); else FB_LAUNCH(true, false); }
  else      { if (gres) FB_LAUNCH(false, true); else FB_LAUNCH(false, false); }
#undef FB_LAUNCH
  return launch_status();
}

template <typename T, int V>
int launch_fwd(const void* x, const float* scale, const float* bias, const void* res, void* y, int N, int C,
               int HW, int relu, hipStream_t st) {
  const int nvec = HW / V;
  const int64_t chunks = ceil_div64(nvec, kThreads * 4);  // >= 4 vectors in flight per thread
  const dim3 grid(static_cast<unsigned>(N) * C, static_cast<unsigned>(chunks < 64 ? chunks : 64));
  const T* xp = static_cast<const T*>(x);
  const T* rp = static_cast<const T*>(res);
  T* yp = static_cast<T*>(y);
#define FB_LAUNCH(R, S) hipLaunchKernelGGL((frozen_bn_fwd_kernel<T, V, R, S>), grid, dim3(kThreads), 0, st, xp, scale, bias, rp, yp, C, HW)
  if (relu) { if (res) FB_LAUNCH(true, true); else FB_LAUNCH(true, false); }
  else      { if (res) FB_LAUNCH(false, true); else FB_LAUNCH(false, false); }
#undef FB_LAUNCH
  return launch_status();
}

template <typename T, int V>
int launch_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gres, int N, int C, int HW,
               int relu, hipStream_t st) {
  const int nvec = HW / V;
  const int64_t chunks = ceil_div64(nvec, kThreads * 4);  // >= 4 vectors in flight per thread
  const dim3 grid(static_cast<unsigned>(N) * C, static_cast<unsigned>(chunks < 64 ? chunks : 64));
  const T* gp = static_cast<const T*>(gy);
  const T* yp = static_cast<const T*>(y);
  T* xp = static_cast<T*>(gx);
  T* rp = static_cast<T*>(gres);
#define FB_LAUNCH(R, S) hipLaunchKernelGGL((frozen_bn_bwd_kernel<T, V, R, S>), grid, dim3(kThreads), 0, st, gp, yp, scale, xp, rp, C, HW)
  if (relu) { if (gres) FB_LAUNCH(true, true); else FB_LAUNCH(true, false); }
  else      { if (gres) FB_LAUNCH(false, true); else FB_LAUNCH(false, false); }
#undef FB_LAUNCH
  return launch_status();
}

// widest vector (in elements) such that every plane start stays aligned: V | HW and V*sizeof(T) <= 16
template <typename T> int pick_vec(int HW, const void* a, const void* b, const void* c, const void* d) {
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                         reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(d);
  for (int v = 16 / static_cast<int>(sizeof(T)); v > 1; v >>= 1)
    if (HW % v == 0 && (bits % (v * sizeof(T))) == 0) return v;
  return 1;
}

template <typename T>
int dispatch_fwd(const void* x, const float* scale, const float* bias, const void* res, void* y, int N, int C,
                 int HW, int relu, hipStream_t st) {
  switch (pick_vec<T>(HW, x, res, y, nullptr)) {
    case 8: return launch_fwd<T, 8>(x, scale, bias, res, y, N, C, HW, relu, st);
    case 4: return launch_fwd<T, 4>(x, scale, bias, res, y, N, C, HW, relu, st);
    case 2: return launch_fwd<T, 2>(x, scale, bias, res, y, N, C, HW, relu, st);
    default: return launch_fwd<T, 1>(x, scale, bias, res, y, N, C, HW, relu, st);
  }
}

template <typename T>
int dispatch_bwd(const void* gy, const void* y, const float* scale, void* gx, void* gres, int N, int C, int HW,
                 int relu, hipStream_t st) {
  switch (pick_vec<T>(HW, gy, y, gx, gres)) {
    case 8: return launch_bwd<T, 8>(gy, y, scale, gx, gres, N, C, HW, relu, st);
    case 4: return launch_bwd<T, 4>(gy, y, scale, gx, gres, N, C, HW, relu, st);
    case 2: return launch_bwd<T, 2>(gy, y, scale, gx, gres, N, C, HW, relu, st);
    default: return launch_bwd<T, 1>(gy, y, scale, gx, gres, N, C, HW, relu, st);
  }
}

// channels-last: V | C (so a vector never straddles a pixel) and every base pointer V-aligned
template <typename T> int pick_vec_nhwc(int C, const void* a, const void* b, const void* c, const void* d) {
  return pick_vec<T>(C, a, b, c, d);
}

template <typename T>
int dispatch_fwd_nhwc(const void* x, const float* scale, const float* bias, const void* res, void* y, int64_t n, int C,
                      int relu, hipStream_t st) {
  switch (pick_vec_nhwc<T>(C, x, res, y, nullptr)) {
    case 8: return launch_fwd_nhwc<T, 8>(x, scale, bias, res, y, n, C, relu, st);
    case 4: return launch_fwd_nhwc<T, 4>(x, scale, bias, res, y, n, C, relu, st);
    case 2: return launch_fwd_nhwc<T, 2>(x, scale, bias, res, y, n, C, relu, st);
    default: return launch_fwd_nhwc<T, 1>(x, scale, bias, res, y, n, C, relu, st);
  }
}

template <typename T>
int dispatch_bwd_nhwc(const void* gy, const void* y, const float* scale, void* gx, void* gres, int64_t n, int C, int relu,
                      hipStream_t st) {
  switch (pick_vec_nhwc<T>(C, gy, y, gx, gres)) {
    case 8: return launch_bwd_nhwc<T, 8>(gy, y, scale, gx, gres, n, C, relu, st);
    case 4: return launch_bwd_nhwc<T, 4>(gy, y, scale, gx, gres, n, C, relu, st);
    case 2: return launch_bwd_nhwc<T, 2>(gy, y, scale, gx, gres, n, C, relu, st);
    default: return launch_bwd_nhwc<T, 1>(gy, y, scale, gx, gres, n, C, relu, st);
  }
}

}  // namespace

DETOPS_API int detops_frozen_bn_act_forward(const void* x, const float* scale, const float* bias,
                                            const void* residual, void* y, int dtype, int N, int C, int HW,
                                            int relu, detops_stream_t stream) {
  if (N < 0 || C < 0 || HW < 0) return DETOPS_EINVAL;
  if (N == 0 || C == 0 || HW == 0) return 0;
  if (!x || !scale || !bias || !y) return DETOPS_EINVAL;
  if (static_cast<int64_t>(N) * C > 0x7fffffff / 2) return DETOPS_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case DETOPS_F32: return dispatch_fwd<float>(x, scale, bias, residual, y, N, C, HW, relu, st);
    case DETOPS_F16: return dispatch_fwd<__half>(x, scale, bias, residual, y, N, C, HW, relu, st);
    case DETOPS_BF16: return dispatch_fwd<__hip_bfloat16>(x, scale, bias, residual, y, N, C, HW, relu, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_frozen_bn_act_backward(const void* grad_y, const void* y, const float* scale, void* grad_x,
                                             void* grad_residual, int dtype, int N, int C, int HW, int relu,
                                             detops_stream_t stream) {
  if (N < 0 || C < 0 || HW < 0) return DETOPS_EINVAL;
  if (N == 0 || C == 0 || HW == 0) return 0;
  if (!grad_y || !scale || !grad_x || (relu && !y)) return DETOPS_EINVAL;
  if (static_cast<int64_t>(N) * C > 0x7fffffff / 2) return DETOPS_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  switch (dtype) {
    case DETOPS_F32: return dispatch_bwd<float>(grad_y, y, scale, grad_x, grad_residual, N, C, HW, relu, st);
    case DETOPS_F16: return dispatch_bwd<__half>(grad_y, y, scale, grad_x, grad_residual, N, C, HW, relu, st);
    case DETOPS_BF16: return dispatch_bwd<__hip_bfloat16>(grad_y, y, scale, grad_x, grad_residual, N, C, HW, relu, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_frozen_bn_act_forward_nhwc(const void* x, const float* scale, const float* bias,
                                                 const void* residual, void* y, int dtype, int64_t rows, int C,
                                                 int relu, detops_stream_t stream) {
  if (rows < 0 || C < 0) return DETOPS_EINVAL;
  if (rows == 0 || C == 0) return 0;
  if (!x || !scale || !bias || !y) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t n = rows * C;
  switch (dtype) {
    case DETOPS_F32: return dispatch_fwd_nhwc<float>(x, scale, bias, residual, y, n, C, relu, st);
    case DETOPS_F16: return dispatch_fwd_nhwc<__half>(x, scale, bias, residual, y, n, C, relu, st);
    case DETOPS_BF16: return dispatch_fwd_nhwc<__hip_bfloat16>(x, scale, bias, residual, y, n, C, relu, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_frozen_bn_act_backward_nhwc(const void* grad_y, const void* y, const float* scale, void* grad_x,
                                                  void* grad_residual, int dtype, int64_t rows, int C, int relu,
                                                  detops_stream_t stream) {
  if (rows < 0 || C < 0) return DETOPS_EINVAL;
  if (rows == 0 || C == 0) return 0;
  if (!grad_y || !scale || !grad_x || (relu && !y)) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const int64_t n = rows * C;
  switch (dtype) {
    case DETOPS_F32: return dispatch_bwd_nhwc<float>(grad_y, y, scale, grad_x, grad_residual, n, C, relu, st);
    case DETOPS_F16: return dispatch_bwd_nhwc<__half>(grad_y, y, scale, grad_x, grad_residual, n, C, relu, st);
    case DETOPS_BF16: return dispatch_bwd_nhwc<__hip_bfloat16>(grad_y, y, scale, grad_x, grad_residual, n, C, relu, st);
    default: return DETOPS_EUNSUPPORTED;
  }
}
