// head_loss.hip — the ROI heads' losses for gfx950 (MI355X), fp32: value AND gradient in one pass.
//
//   detops_fastrcnn_loss_f32   FastRCNNLossComputation.__call__ (reference roi_heads/box_head/loss.py:140-193):
//                              softmax cross-entropy over the sampled ROIs + smooth-L1 (beta 1) on the positives'
//                              class-specific deltas, both divided by the number of sampled ROIs
//   detops_mask_loss_f32       MaskRCNNLossComputation.__call__ (roi_heads/mask_head/loss.py:113-143):
//                              BCE-with-logits between each positive ROI's class plane and its mask target (mean)
//   detops_head_loss_backward_f32   the stored gradients times the upstream scalar, in place
//
// As ATen compositions these are ~35 (box head) and ~30 (mask head) launches per step counting the autograd mirror of each
// op — small tensors ([1024, 81], [1024, 324], [256, 81, 28, 28]), so the time is the launches, not the bytes.  Each
// loss here is two launches forward (the pass + a fixed-order sum of its per-row partials) and one backward, the same
// shape as the RPN loss of targets.hip.  Every workgroup recounts the normaliser (#sampled / #positive) from the label
// vector itself — it is a few KB and spares a counting launch.  Sums are formed in a fixed order: bit-reproducible.
#include <cmath>

#include "detops_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
  return __shfl(v, 0);
}

__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off));
  return __shfl(v, 0);
}

// #labels with lo <= label < hi — the SAME predicate the loss rows use, so that a malformed label (>= #classes) drops out
// of the sum and of its normaliser alike; counted by the whole workgroup (every thread returns the count)
__device__ __forceinline__ float block_count_in(const int64_t* __restrict__ labels, int n, int64_t lo, int64_t hi,
                                                float* s_red) {
  float c = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) c += (labels[i] >= lo && labels[i] < hi) ? 1.f : 0.f;
  c = wave_sum(c);
  if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x / kWave] = c;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < kWavesPerBlock; ++j) t += s_red[j];
  __syncthreads();
  return t;
}

// One wave per sampled ROI.  partial[r] = {cross-entropy of row r (0 when ignored), smooth-L1 of row r (0 unless positive)};
// grad_logits [R, C] = (softmax - onehot) / #sampled, grad_box [R, D] = d smooth-L1 / #sampled on the four class columns.
__global__ void __launch_bounds__(kBlock)
fastrcnn_loss_kernel(const float* __restrict__ logits, const float* __restrict__ box, const int64_t* __restrict__ labels,
                     const float* __restrict__ targets, int R, int C, int D, int agnostic, float beta,
                     float* __restrict__ grad_logits, float* __restrict__ grad_box, float* __restrict__ partial) {
  __shared__ float s_red[kWavesPerBlock];
  const float inv = 1.f / fmaxf(block_count_in(labels, R, 0, C, s_red), 1.f);
  const int lane = threadIdx.x & (kWave - 1);
  const int r = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (r >= R) return;
  const int64_t label = labels[r];
  const float* x = logits + static_cast<size_t>(r) * C;
  float* gl = grad_logits + static_cast<size_t>(r) * C;
  float* gb = grad_box + static_cast<size_t>(r) * D;
  float ce = 0.f;
  if (label >= 0 && label < C) {
    float m = -INFINITY;
    for (int c = lane; c < C; c += kWave) m = fmaxf(m, x[c]);
    m = wave_max_all(m);
    float s = 0.f;
    for (int c = lane; c < C; c += kWave) s += expf(x[c] - m);
    s = wave_sum(s);
    const float lse = logf(s);
    ce = (m + lse) - x[label];                      // -log_softmax(x)[label]
    for (int c = lane; c < C; c += kWave) gl[c] = (expf(x[c] - m - lse) - (c == label ? 1.f : 0.f)) * inv;
  } else {
    for (int c = lane; c < C; c += kWave) gl[c] = 0.f;
  }
  const int col0 = (label > 0) ? (agnostic ? 4 : 4 * static_cast<int>(label)) : -1;
  float l1 = 0.f;
  for (int c = lane; c < D; c += kWave) {
    float g = 0.f;
    if (col0 >= 0 && c >= col0 && c < col0 + 4) {
      const float d = box[static_cast<size_t>(r) * D + c] - targets[static_cast<size_t>(r) * 4 + (c - col0)];
      const float ad = fabsf(d);
      if (ad < beta) { l1 += 0.5f * ad * ad / beta; g = d / beta; }
      else { l1 += ad - 0.5f * beta; g = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f); }
    }
    gb[c] = g * inv;
  }
  l1 = wave_sum(l1);
  if (lane == 0) { partial[2 * r] = ce; partial[2 * r + 1] = l1; }
}

// fixed-order sum of the per-row partials -> out = {sum0 * inv, sum1 * inv}; inv = 1 / max(count(lo <= labels < hi) * per, 1)
__global__ void __launch_bounds__(kBlock)
head_loss_finish_kernel(const float* __restrict__ partial, int rows, int width, const int64_t* __restrict__ labels, int n,
                        int64_t lo, int64_t hi, float per, float* __restrict__ out) {
  __shared__ float s_red[kWavesPerBlock];
  __shared__ float s_sum[2][kWavesPerBlock];
  const float inv = 1.f / fmaxf(block_count_in(labels, n, lo, hi, s_red) * per, 1.f);
  float v[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < rows; i += kBlock)
    for (int k = 0; k < width; ++k) v[k] += partial[i * width + k];
  for (int k = 0; k < 2; ++k) {
    float w = v[k];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) w += __shfl_down(w, off);
    if ((threadIdx.x & (kWave - 1)) == 0) s_sum[k][threadIdx.x / kWave] = w;
  }
  __syncthreads();
  if (threadIdx.x < width) {
    float t = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) t += s_sum[threadIdx.x][j];
    out[threadIdx.x] = t * inv;
  }
}

// grid (P, kMaskSplit): workgroup (p, y) writes the gradient of the class planes c = y, y + kMaskSplit, ... of ROI p —
// zeros, except on the plane of p's own class (positive ROIs), where it also forms the ROI's BCE sum -> partial[p]
constexpr int kMaskSplit = 8;

__global__ void __launch_bounds__(kBlock)
mask_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ targets,
                 int P, int C, int S, float* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float s_red[kWavesPerBlock];
  const float inv = 1.f / fmaxf(block_count_in(labels, P, 1, C, s_red) * static_cast<float>(S), 1.f);
  const int p = blockIdx.x;
  const int64_t label = labels[p];
  const bool pos = label > 0 && label < C;
  const int own = pos ? static_cast<int>(label) : 0;
  float loss = 0.f;
  for (int c = blockIdx.y; c < C; c += kMaskSplit) {
    float* g = grad + (static_cast<size_t>(p) * C + c) * S;
    if (pos && c == own) {
      const float* x = logits + (static_cast<size_t>(p) * C + c) * S;
      const float* t = targets + static_cast<size_t>(p) * S;
      for (int i = threadIdx.x; i < S; i += kBlock) {
        const float xi = x[i], ti = t[i];
        const float e = expf(-fabsf(xi));
        loss += fmaxf(xi, 0.f) - xi * ti + log1pf(e);
        const float sig = (xi >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
        g[i] = (sig - ti) * inv;
      }
    } else {
      for (int i = threadIdx.x; i < S; i += kBlock) g[i] = 0.f;
    }
  }
  if (static_cast<int>(blockIdx.y) != own % kMaskSplit) return;      // one workgroup per ROI reports (0 for a non-positive)
  loss = wave_sum(loss);
  if ((threadIdx.x & (kWave - 1)) == 0) s_red[threadIdx.x / kWave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int j = 0; j < kWavesPerBlock; ++j) t += s_red[j];
    partial[p] = t;
  }
}

__global__ void __launch_bounds__(kBlock)
head_loss_scale_kernel(float* __restrict__ a, int64_t na, const float* __restrict__ up_a, float* __restrict__ b,
                       int64_t nb, const float* __restrict__ up_b) {
  const float sa = na ? up_a[0] : 0.f, sb = nb ? up_b[0] : 0.f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < na + nb;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    if (i < na) a[i] *= sa; else b[i - na] *= sb;
  }
}

}  // namespace

DETOPS_API size_t detops_fastrcnn_loss_workspace_bytes(int R) { return static_cast<size_t>(R < 0 ? 0 : R) * 2 * sizeof(float); }

DETOPS_API int detops_fastrcnn_loss_f32(const float* class_logits, const float* box_regression, const int64_t* labels,
                                        const float* regression_targets, int R, int C, int D, int cls_agnostic, float beta,
                                        float* grad_logits, float* grad_box, float* losses2, void* workspace,
                                        size_t workspace_bytes, detops_stream_t stream) {
  if (R <= 0 || C <= 0 || D <= 0 || !(beta > 0.f) || (cls_agnostic ? D < 8 : D != 4 * C)) return DETOPS_EINVAL;
  if (!class_logits || !box_regression || !labels || !regression_targets || !grad_logits || !grad_box || !losses2 || !workspace ||
      workspace_bytes < detops_fastrcnn_loss_workspace_bytes(R))
    return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(fastrcnn_loss_kernel, dim3((R + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, st, class_logits,
                     box_regression, labels, regression_targets, R, C, D, cls_agnostic, beta, grad_logits, grad_box, partial);
  hipLaunchKernelGGL(head_loss_finish_kernel, dim3(1), dim3(kBlock), 0, st, partial, R, 2, labels, R,
                     static_cast<int64_t>(0), static_cast<int64_t>(C), 1.f, losses2);
  return launch_status();
}

DETOPS_API size_t detops_mask_loss_workspace_bytes(int P) { return static_cast<size_t>(P < 0 ? 0 : P) * sizeof(float); }

DETOPS_API int detops_mask_loss_f32(const float* mask_logits, const int64_t* labels, const float* mask_targets, int P, int C,
                                    int M, float* grad_logits, float* loss1, void* workspace, size_t workspace_bytes,
                                    detops_stream_t stream) {
  if (P <= 0 || C <= 0 || M <= 0) return DETOPS_EINVAL;
  if (!mask_logits || !labels || !mask_targets || !grad_logits || !loss1 || !workspace ||
      workspace_bytes < detops_mask_loss_workspace_bytes(P))
    return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(mask_loss_kernel, dim3(P, kMaskSplit), dim3(kBlock), 0, st, mask_logits, labels, mask_targets, P, C, M * M,
                     grad_logits, partial);
  hipLaunchKernelGGL(head_loss_finish_kernel, dim3(1), dim3(kBlock), 0, st, partial, P, 1, labels, P, static_cast<int64_t>(1),
                     static_cast<int64_t>(C), static_cast<float>(M * M), loss1);
  return launch_status();
}

DETOPS_API int detops_head_loss_backward_f32(float* grad_a, int64_t count_a, const float* upstream_a, float* grad_b,
                                             int64_t count_b, const float* upstream_b, detops_stream_t stream) {
  if (count_a < 0 || count_b < 0 || (count_a && (!grad_a || !upstream_a)) || (count_b && (!grad_b || !upstream_b)))
    return DETOPS_EINVAL;
  const int64_t total = count_a + count_b;
  if (total == 0) return 0;
  const int blocks = static_cast<int>(std::min<int64_t>(ceil_div64(total, kBlock * 4), 2048));
  hipLaunchKernelGGL(head_loss_scale_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), grad_a, count_a, upstream_a,
                     count_b ? grad_b : nullptr, count_b, upstream_b);
  return launch_status();
}
