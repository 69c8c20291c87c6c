// optim.hip — the data-parallel step's two streaming kernels for gfx950 (MI355X), fp32.
//
// The reference hands its gradients to torch's DistributedDataParallel and steps torch.optim.SGD (tools/train_net.py:45-54,
// engine/trainer.py:93-99, solver/build.py:7-20).  Here a bucket of the wrapper (engine/ddp_step.py) is THREE flat fp32
// arrays of the same layout — parameters, gradients, momentum — and a bucket costs two launches of this file:
//
//   detops_pack_f32                the gradients autograd produced (separate tensors) -> the bucket's gradient array, one
//                                  launch for up to 48 tensors (pointer table in the kernel arguments: nothing to upload);
//   detops_sgd_momentum_flat_f32   g <- g + wd * p ; m <- momentum * m + g ; p <- p - lr * m over the whole bucket
//                                  (torch.optim.SGD, dampening 0, no Nesterov): one pass, 3 reads + 2 writes per element.
//                                  The reference's two hyper-parameter groups (weights | biases: BIAS_LR_FACTOR,
//                                  WEIGHT_DECAY_BIAS) are two element ranges of the bucket: [0, split) and [split, n).
//
// Both are pure HBM streams (16-byte accesses, grid-stride); what they buy is on the HOST: torch's multi-tensor copy +
// fused SGD cost ~140 us of launch-path time per bucket inside the backward pass, these two ~15 us.
#include "detops_common.h"

namespace {

constexpr int kBlockT = 256;
constexpr int kPackMax = 48;          // tensors per launch
constexpr int kPackChunk = 8192;      // elements per workgroup

struct PackArgs {
  const float* src[kPackMax];
  long long dst_off[kPackMax];        // element offset in the bucket
  long long count[kPackMax];
  int chunk_begin[kPackMax + 1];      // first workgroup of tensor t
  int n;
};

__global__ void __launch_bounds__(kBlockT)
pack_kernel(PackArgs a, float* __restrict__ dst) {
  const int b = blockIdx.x;
  // which tensor?  compile-time indices only (a dynamically indexed by-value argument struct is copied to scratch):
  // 48 wave-uniform compare / select groups on scalars; chunk_begin beyond the last tensor = the grid size
  const float* __restrict__ s = a.src[0];
  long long off = a.dst_off[0], cnt = a.count[0];
  int first = a.chunk_begin[0];
#pragma unroll
  for (int i = 1; i < kPackMax; ++i)
    if (b >= a.chunk_begin[i]) { s = a.src[i]; off = a.dst_off[i]; cnt = a.count[i]; first = a.chunk_begin[i]; }
  const long long e0 = static_cast<long long>(b - first) * kPackChunk;
  const long long e1 = e0 + kPackChunk < cnt ? e0 + kPackChunk : cnt;
  float* d = dst + off;
  const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
  if (vec) {
    const long long v0 = e0 / 4, v1 = e1 / 4;
    const float4* sv = reinterpret_cast<const float4*>(s);
    float4* dv = reinterpret_cast<float4*>(d);
    for (long long i = v0 + threadIdx.x; i < v1; i += kBlockT) dv[i] = sv[i];
    for (long long i = v1 * 4 + threadIdx.x; i < e1; i += kBlockT) d[i] = s[i];
  } else {
    for (long long i = e0 + threadIdx.x; i < e1; i += kBlockT) d[i] = s[i];
  }
}

__global__ void __launch_bounds__(kBlockT)
sgd_momentum_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long long n,
                         long long split, float lr_w, float wd_w, float lr_b, float wd_b, float mom) {
  const long long nv = n / 4;
  float4* pv = reinterpret_cast<float4*>(p);
  const float4* gv = reinterpret_cast<const float4*>(g);
  float4* mv = reinterpret_cast<float4*>(m);
  for (long long i = static_cast<long long>(blockIdx.x) * kBlockT + threadIdx.x; i < nv;
       i += static_cast<long long>(gridDim.x) * kBlockT) {
    const bool w = 4 * i < split;                      // split is a multiple of 4: a vector never straddles it
    const float lr = w ? lr_w : lr_b, wd = w ? wd_w : wd_b;
    float4 pp = pv[i], mm = mv[i];
    const float4 gg = gv[i];
    float ge[4] = {gg.x, gg.y, gg.z, gg.w}, pe[4] = {pp.x, pp.y, pp.z, pp.w}, me[4] = {mm.x, mm.y, mm.z, mm.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = ge[k] + wd * pe[k];              // torch.optim.SGD: grad.add(param, alpha=weight_decay)
      me[k] = mom * me[k] + d;                         //                  buf.mul_(momentum).add_(grad)
      pe[k] = pe[k] - lr * me[k];                      //                  param.add_(buf, alpha=-lr)
    }
    pv[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    mv[i] = make_float4(me[0], me[1], me[2], me[3]);
  }
  if (blockIdx.x == 0) {                               // n is not a multiple of 4: the tail, element-wise
    for (long long i = nv * 4 + threadIdx.x; i < n; i += kBlockT) {
      const bool w = i < split;
      const float lr = w ? lr_w : lr_b, wd = w ? wd_w : wd_b;
      const float d = g[i] + wd * p[i];
      const float mm = mom * m[i] + d;
      m[i] = mm;
      p[i] = p[i] - lr * mm;
    }
  }
}

}  // namespace

DETOPS_API int detops_pack_max_tensors(void) { return kPackMax; }

DETOPS_API int detops_pack_f32(const void* const* srcs, const int64_t* counts, const int64_t* dst_offsets, int n,
                               float* dst, detops_stream_t stream) {
  if (n < 0 || n > kPackMax || (n > 0 && (!srcs || !counts || !dst_offsets || !dst))) return DETOPS_EINVAL;
  PackArgs a{};
  int chunks = 0;
  for (int i = 0; i < n; ++i) {
    if (counts[i] < 0 || dst_offsets[i] < 0 || (counts[i] > 0 && !srcs[i])) return DETOPS_EINVAL;
    a.src[i] = static_cast<const float*>(srcs[i]);
    a.count[i] = counts[i];
    a.dst_off[i] = dst_offsets[i];
    a.chunk_begin[i] = chunks;
    const int64_t c = (counts[i] + kPackChunk - 1) / kPackChunk;
    if (c > (1 << 30) - chunks) return DETOPS_EUNSUPPORTED;
    chunks += static_cast<int>(c);
  }
  for (int i = n; i <= kPackMax; ++i) a.chunk_begin[i] = chunks;
  a.n = n;
  if (chunks == 0) return 0;
  hipLaunchKernelGGL(pack_kernel, dim3(chunks), dim3(kBlockT), 0, as_stream(stream), a, dst);
  return launch_status();
}

DETOPS_API int detops_sgd_momentum_flat_f32(float* params, const float* grads, float* momentum_buf, int64_t n, int64_t split,
                                            float lr_weights, float wd_weights, float lr_biases, float wd_biases,
                                            float momentum, detops_stream_t stream) {
  if (n < 0 || split < 0 || split > n || (split & 3) != 0) return DETOPS_EINVAL;
  if (n == 0) return 0;
  if (!params || !grads || !momentum_buf) return DETOPS_EINVAL;
  if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(momentum_buf)) & 15)
    return DETOPS_EINVAL;
  const int64_t want = (n / 4 + kBlockT - 1) / kBlockT;
  const int blocks = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(want, 16 * 1024)));
  hipLaunchKernelGGL(sgd_momentum_flat_kernel, dim3(blocks), dim3(kBlockT), 0, as_stream(stream), params, grads, momentum_buf,
                     static_cast<long long>(n), static_cast<long long>(split), lr_weights, wd_weights, lr_biases, wd_biases,
                     momentum);
  return launch_status();
}

// ---- contention stand-in (measurement tool, not part of the training path): `workgroups` workgroups of 1024 threads that
// each hold a CU's wave slots for `microseconds` (100 MHz wall clock) — what a ring all-reduce kernel of RCCL does to the
// compute stream's kernels at N > 1 (a fixed number of channels = workgroups resident for the duration of a 25 MB bucket),
// reproducible on ONE GPU where the 1-rank collective is a no-op.  engine/ddp_step.py launches it on the wrapper's side
// stream when DETOPS_DDP_STANDIN="workgroups:microseconds" is set (tools/gpu/r06_ddp_contention.sh).
namespace {
__global__ void __launch_bounds__(1024)
occupy_kernel(long long ticks, int* __restrict__ sink) {
  const long long t0 = detops_wall_clock();
  int spins = 0;
  while (detops_wall_clock() - t0 < ticks) {
#ifndef DETOPS_CPU_EMU
    __builtin_amdgcn_s_sleep(32);
#endif
    if (++spins > (1 << 26)) break;     // bounded whatever the clock does
#ifdef DETOPS_CPU_EMU
    break;
#endif
  }
  if (sink && spins < 0) *sink = spins;  // never true: keeps the loop
}
}  // namespace

DETOPS_API int detops_debug_occupy(int workgroups, int microseconds, detops_stream_t stream) {
  if (workgroups < 0 || microseconds < 0 || workgroups > 4096 || microseconds > 100000) return DETOPS_EINVAL;
  if (workgroups == 0 || microseconds == 0) return 0;
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(1024), 0, as_stream(stream), static_cast<long long>(microseconds) * 100,
                     static_cast<int*>(nullptr));
  return launch_status();
}
