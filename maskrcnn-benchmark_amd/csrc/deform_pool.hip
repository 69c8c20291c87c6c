// deform_pool.hip — deformable position-sensitive ROI pooling, forward / backward, gfx950, fp32 NCHW.
//
// Replaces DeformablePSROIPoolForwardKernel / DeformablePSROIPoolBackwardAccKernel
// (reference csrc/cuda/deform_pool_kernel_cuda.cu:53-147, :149-264) and their host wrappers
// (csrc/cuda/deform_pool_cuda.cu:38-87, csrc/deform_pool.h:11-70) behind
// detops_deform_psroi_pool_{forward,backward}_f32.  No reference model config instantiates
// DeformRoIPooling (it is exported operator API only), so the layout is the
// straightforward one for a gather op on wave64 hardware:
//   * one workgroup per (ROI, class) — the learned offset (trans_x, trans_y) and hence the whole
//     sampling lattice of a bin are shared by every output channel of a class, so the
//     P*P*S*S sample table {tap offset, x/y fractions, valid} is built ONCE per workgroup in LDS
//     (the reference re-derives it per output element: channels_each_class times);
//   * lanes then run over (channel-in-class, bin) with bin fastest: output stores are one
//     contiguous run per workgroup, the position-sensitive input plane index depends on
//     (channel, gh, gw) only;
//   * backward: the data gradient is a scatter with cross-ROI collisions -> fp32 global atomics
//     (hardware float atomics, -munsafe-fp-atomics), the offset gradient is reduced over the
//     class's channels inside the workgroup (LDS atomics) and leaves with ONE global atomic per
//     (part cell, axis) instead of one per sample per channel.
// Arithmetic follows the .cu file's float/double mixing with FP contraction off (see
// oracle/detops_oracle.c: oracle_deform_psroi_pool_*).
#include "detops_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxSamples = 2048;  // P*P*S*S entries of the LDS sample table (7*7*4*4 = 784), 32 KiB
constexpr int kMaxPart = 1024;     // part_size^2 * 2 offset-gradient cells reduced in LDS
constexpr int kWin = 4;            // backward: side of a thread's private LDS window of input-gradient pixels

struct Sample {
  int off;        // y0 * W + x0, or -1 for a skipped sample
  int dxy;        // (x1 - x0) | ((y1 - y0) << 1) | (x0 << 2): ceil == floor on integer coordinates
  float dist_x, dist_y;
};

// Per-(ROI, class, bin) sampling geometry, reference :83-115.
struct Lattice {
  float wstart, hstart, sub_w, sub_h, roi_width, roi_height;
  int b, part_h, part_w, gw, gh;
};

__device__ __forceinline__ Lattice make_lattice(const float* __restrict__ roi,
                                                const float* __restrict__ trans, int n, int class_id,
                                                int ph, int pw, float spatial_scale, int P,
                                                int no_trans, float trans_std, int S, int group_size,
                                                int part_size, int num_classes) {
#pragma clang fp contract(off)
  Lattice g;
  g.b = static_cast<int>(roi[0]);
  const float roi_start_w = static_cast<float>(static_cast<double>(roundf(roi[1]) * spatial_scale) - 0.5);
  const float roi_start_h = static_cast<float>(static_cast<double>(roundf(roi[2]) * spatial_scale) - 0.5);
  const float roi_end_w = static_cast<float>(
      static_cast<double>(static_cast<float>(static_cast<double>(roundf(roi[3])) + 1.) * spatial_scale) - 0.5);
  const float roi_end_h = static_cast<float>(
      static_cast<double>(static_cast<float>(static_cast<double>(roundf(roi[4])) + 1.) * spatial_scale) - 0.5);
  const double rw = static_cast<double>(roi_end_w - roi_start_w);
  const double rh = static_cast<double>(roi_end_h - roi_start_h);
  g.roi_width = static_cast<float>(rw < 0.1 ? 0.1 : rw);
  g.roi_height = static_cast<float>(rh < 0.1 ? 0.1 : rh);
  const float bin_h = g.roi_height / static_cast<float>(P);
  const float bin_w = g.roi_width / static_cast<float>(P);
  g.sub_h = bin_h / static_cast<float>(S);
  g.sub_w = bin_w / static_cast<float>(S);
  g.part_h = static_cast<int>(floorf(static_cast<float>(ph) / P * part_size));
  g.part_w = static_cast<int>(floorf(static_cast<float>(pw) / P * part_size));
  float trans_x = 0.f, trans_y = 0.f;
  if (!no_trans) {
    const size_t tb = ((static_cast<size_t>(n * num_classes + class_id) * 2) * part_size + g.part_h) * part_size + g.part_w;
    trans_x = trans[tb] * trans_std;
    trans_y = trans[tb + static_cast<size_t>(part_size) * part_size] * trans_std;
  }
  float wstart = static_cast<float>(pw) * bin_w + roi_start_w;
  wstart += trans_x * g.roi_width;
  float hstart = static_cast<float>(ph) * bin_h + roi_start_h;
  hstart += trans_y * g.roi_height;
  g.wstart = wstart;
  g.hstart = hstart;
  const int gw = static_cast<int>(floorf(static_cast<float>(pw) * group_size / P));
  const int gh = static_cast<int>(floorf(static_cast<float>(ph) * group_size / P));
  g.gw = min(max(gw, 0), group_size - 1);
  g.gh = min(max(gh, 0), group_size - 1);
  return g;
}

__device__ __forceinline__ Sample make_sample(const Lattice& g, int ih, int iw, int H, int W) {
#pragma clang fp contract(off)
  Sample s;
  float w = g.wstart + iw * g.sub_w;
  float h = g.hstart + ih * g.sub_h;
  const double wd = static_cast<double>(w), hd = static_cast<double>(h);
  if (wd < -0.5 || wd > W - 0.5 || hd < -0.5 || hd > H - 0.5) {
    s.off = -1; s.dxy = 0; s.dist_x = 0.f; s.dist_y = 0.f;
    return s;
  }
  w = static_cast<float>(fmin(fmax(wd, 0.), W - 1.));
  h = static_cast<float>(fmin(fmax(hd, 0.), H - 1.));
  const int x0 = static_cast<int>(floorf(w)), x1 = static_cast<int>(ceilf(w));
  const int y0 = static_cast<int>(floorf(h)), y1 = static_cast<int>(ceilf(h));
  s.off = y0 * W + x0;
  s.dxy = (x1 - x0) | ((y1 - y0) << 1) | (x0 << 2);
  s.dist_x = w - x0;
  s.dist_y = h - y0;
  return s;
}

// grid: K * num_classes workgroups.
template <bool kTable>
__global__ void __launch_bounds__(kBlock)
psroi_fwd_kernel(const float* __restrict__ data, const float* __restrict__ rois,
                 const float* __restrict__ trans, float* __restrict__ out, float* __restrict__ top_count,
                 int C, int H, int W, int P, int no_trans, float spatial_scale, int output_dim,
                 int group_size, int part_size, int S, float trans_std, int num_classes, int cec) {
  __shared__ Sample s_tab[kTable ? kMaxSamples : 1];
  __shared__ int s_plane[kTable ? 1024 : 1];  // per bin: gh * group_size + gw
  const int n = blockIdx.x / num_classes;
  const int class_id = blockIdx.x - n * num_classes;
  const float* roi = rois + static_cast<size_t>(n) * 5;
  const int bins = P * P, SS = S * S;
  if (kTable) {
    for (int t = threadIdx.x; t < bins * SS; t += kBlock) {
      const int bin = t / SS, s = t - bin * SS;
      const int ph = bin / P, pw = bin - ph * P;
      const Lattice g = make_lattice(roi, trans, n, class_id, ph, pw, spatial_scale, P, no_trans,
                                     trans_std, S, group_size, part_size, num_classes);
      s_tab[t] = make_sample(g, s / S, s - (s / S) * S, H, W);
      if (s == 0) s_plane[bin] = g.gh * group_size + g.gw;
    }
    __syncthreads();
  }
  const int b = static_cast<int>(roi[0]);
  const size_t plane = static_cast<size_t>(H) * W;
  const float* img = data + static_cast<size_t>(b) * C * plane;
  const int c_first = class_id * cec;
  const int total = cec * bins;
  const size_t obase = (static_cast<size_t>(n) * output_dim + c_first) * bins;
  for (int o = threadIdx.x; o < total; o += kBlock) {
#pragma clang fp contract(off)
    const int cl = o / bins, bin = o - cl * bins;
    const int ctop = c_first + cl;
    Lattice g{};
    int pl;
    if (kTable) {
      pl = s_plane[bin];
    } else {
      const int ph = bin / P, pw = bin - ph * P;
      g = make_lattice(roi, trans, n, class_id, ph, pw, spatial_scale, P, no_trans, trans_std, S,
                       group_size, part_size, num_classes);
      pl = g.gh * group_size + g.gw;
    }
    const float* d = img + static_cast<size_t>(ctop * group_size * group_size + pl) * plane;
    float sum = 0.f;
    int count = 0;
    for (int s = 0; s < SS; ++s) {
      const Sample sp = kTable ? s_tab[bin * SS + s] : make_sample(g, s / S, s - (s / S) * S, H, W);
      if (sp.off < 0) continue;
      const int dx = sp.dxy & 1, dyW = ((sp.dxy >> 1) & 1) * W;
      const float v11 = d[sp.off], v12 = d[sp.off + dyW], v21 = d[sp.off + dx], v22 = d[sp.off + dyW + dx];
      const float ax = 1 - sp.dist_x, ay = 1 - sp.dist_y;
      const float val = ax * ay * v11 + ax * sp.dist_y * v12 + sp.dist_x * ay * v21 + sp.dist_x * sp.dist_y * v22;
      sum += val;
      ++count;
    }
    out[obase + o] = count == 0 ? 0.f : sum / count;
    top_count[obase + o] = static_cast<float>(count);
  }
}

template <bool kTable>
__global__ void __launch_bounds__(kBlock)
psroi_bwd_kernel(const float* __restrict__ top_diff, const float* __restrict__ data,
                 const float* __restrict__ rois, const float* __restrict__ trans,
                 const float* __restrict__ top_count, float* __restrict__ data_diff,
                 float* __restrict__ trans_diff, int C, int H, int W, int P, int no_trans,
                 float spatial_scale, int output_dim, int group_size, int part_size, int S,
                 float trans_std, int num_classes, int cec) {
  __shared__ Sample s_tab[kTable ? kMaxSamples : 1];
  __shared__ int s_plane[kTable ? 1024 : 1];
  __shared__ int s_part[kTable ? 1024 : 1];
  __shared__ float s_tgrad[kMaxPart];
  __shared__ float s_roi_wh[2];
  // A bin's S x S samples land on a few pixels of ONE channel plane (window = bin size + 1): the reference adds 4 S^2 values
  // with global atomics (deform_psroi_pooling_cuda.cu:236-243).  Here a thread first collects them in its private kWin x kWin
  // LDS window anchored at its first sample's pixel and adds each touched pixel once (round 6; samples outside the window —
  // bins larger than 3 pixels — keep the direct adds).
  __shared__ float s_win[kWin * kWin][kBlock];
  const int n = blockIdx.x / num_classes;
  const int class_id = blockIdx.x - n * num_classes;
  const float* roi = rois + static_cast<size_t>(n) * 5;
  const int bins = P * P, SS = S * S;
  const int pcells = part_size * part_size;
  const bool lds_tgrad = !no_trans && 2 * pcells <= kMaxPart;
  if (lds_tgrad)
    for (int t = threadIdx.x; t < 2 * pcells; t += kBlock) s_tgrad[t] = 0.f;
  if (kTable) {
    for (int t = threadIdx.x; t < bins * SS; t += kBlock) {
      const int bin = t / SS, s = t - bin * SS;
      const int ph = bin / P, pw = bin - ph * P;
      const Lattice g = make_lattice(roi, trans, n, class_id, ph, pw, spatial_scale, P, no_trans,
                                     trans_std, S, group_size, part_size, num_classes);
      s_tab[t] = make_sample(g, s / S, s - (s / S) * S, H, W);
      if (s == 0) {
        s_plane[bin] = g.gh * group_size + g.gw;
        s_part[bin] = g.part_h * part_size + g.part_w;
      }
      if (t == 0) { s_roi_wh[0] = g.roi_width; s_roi_wh[1] = g.roi_height; }
    }
  }
  __syncthreads();
  const int b = static_cast<int>(roi[0]);
  const size_t plane = static_cast<size_t>(H) * W;
  const size_t img = static_cast<size_t>(b) * C * plane;
  const int c_first = class_id * cec;
  const int total = cec * bins;
  const size_t obase = (static_cast<size_t>(n) * output_dim + c_first) * bins;
  const size_t tbase = (static_cast<size_t>(n * num_classes + class_id) * 2) * pcells;
  for (int o = threadIdx.x; o < total; o += kBlock) {
#pragma clang fp contract(off)
    const float cnt = top_count[obase + o];
    if (cnt <= 0.f) continue;
    const float diff_val = top_diff[obase + o] / cnt;
    const int cl = o / bins, bin = o - cl * bins;
    const int ctop = c_first + cl;
    Lattice g{};
    int pl, part;
    float roi_w, roi_h;
    if (kTable) {
      pl = s_plane[bin]; part = s_part[bin]; roi_w = s_roi_wh[0]; roi_h = s_roi_wh[1];
    } else {
      const int ph = bin / P, pw = bin - ph * P;
      g = make_lattice(roi, trans, n, class_id, ph, pw, spatial_scale, P, no_trans, trans_std, S,
                       group_size, part_size, num_classes);
      pl = g.gh * group_size + g.gw; part = g.part_h * part_size + g.part_w;
      roi_w = g.roi_width; roi_h = g.roi_height;
    }
    const size_t base = img + static_cast<size_t>(ctop * group_size * group_size + pl) * plane;
    float tx = 0.f, ty = 0.f;
#pragma unroll
    for (int cell = 0; cell < kWin * kWin; ++cell) s_win[cell][threadIdx.x] = 0.f;
    int win_x = -1, win_row = 0;                      // anchor: column and row offset (y * W) of the first valid sample
    for (int s = 0; s < SS; ++s) {
      const Sample sp = kTable ? s_tab[bin * SS + s] : make_sample(g, s / S, s - (s / S) * S, H, W);
      if (sp.off < 0) continue;
      const int dx = sp.dxy & 1, dy = (sp.dxy >> 1) & 1, dyW = dy * W;
      const int x0 = sp.dxy >> 2;
      const float ax = 1 - sp.dist_x, ay = 1 - sp.dist_y;
      const float q00 = ax * ay, q01 = ax * sp.dist_y, q10 = sp.dist_x * ay, q11 = sp.dist_x * sp.dist_y;
      if (win_x < 0) { win_x = x0; win_row = sp.off - x0; }
      const int rx = x0 - win_x, rr = sp.off - x0 - win_row;          // rr = (y0 - anchor row) * W
      const int ry = (rr >= W) + (rr >= 2 * W) + (rr >= 3 * W);
      if (rx >= 0 && rx + dx < kWin && rr >= 0 && rr < 4 * W && ry + dy < kWin) {
        s_win[ry * kWin + rx][threadIdx.x] += q00 * diff_val;
        s_win[(ry + dy) * kWin + rx][threadIdx.x] += q01 * diff_val;
        s_win[ry * kWin + rx + dx][threadIdx.x] += q10 * diff_val;
        s_win[(ry + dy) * kWin + rx + dx][threadIdx.x] += q11 * diff_val;
      } else {
        float* dd = data_diff + base + sp.off;
        atomicAdd(dd, q00 * diff_val);
        atomicAdd(dd + dyW, q01 * diff_val);
        atomicAdd(dd + dx, q10 * diff_val);
        atomicAdd(dd + dyW + dx, q11 * diff_val);
      }
      if (no_trans) continue;
      const float* u = data + base + sp.off;
      const float U00 = u[0], U01 = u[dyW], U10 = u[dx], U11 = u[dyW + dx];
      float diff_x = (U11 * sp.dist_y + U10 * ay - U01 * sp.dist_y - U00 * ay) * trans_std * diff_val;
      diff_x *= roi_w;
      float diff_y = (U11 * sp.dist_x + U01 * ax - U10 * sp.dist_x - U00 * ax) * trans_std * diff_val;
      diff_y *= roi_h;
      tx += diff_x;
      ty += diff_y;
    }
    if (win_x >= 0) {
      float* wd = data_diff + base + win_row + win_x;
#pragma unroll
      for (int cell = 0; cell < kWin * kWin; ++cell) {
        const float v = s_win[cell][threadIdx.x];
        if (v != 0.f) atomicAdd(wd + (cell / kWin) * W + (cell % kWin), v);
      }
    }
    if (!no_trans) {
      if (lds_tgrad) {
        atomicAdd(&s_tgrad[part], tx);
        atomicAdd(&s_tgrad[pcells + part], ty);
      } else {
        atomicAdd(trans_diff + tbase + part, tx);
        atomicAdd(trans_diff + tbase + pcells + part, ty);
      }
    }
  }
  if (lds_tgrad) {
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * pcells; t += kBlock) {
      const float v = s_tgrad[t];
      if (v != 0.f) atomicAdd(trans_diff + tbase + t, v);  // several classes never share a cell; ROIs never do
    }
  }
}

struct PsArgs {
  int num_classes, cec;
};

inline int ps_validate(int N, int C, int H, int W, int K, int channels_trans, int no_trans,
                       int output_dim, int group_size, int pooled_size, int part_size,
                       int sample_per_part, PsArgs* a) {
  if (N < 0 || C < 0 || H < 0 || W < 0 || K < 0 || output_dim <= 0 || group_size <= 0 ||
      pooled_size <= 0 || part_size <= 0 || sample_per_part <= 0)
    return DETOPS_EINVAL;
  if (!no_trans && (channels_trans < 2 || (channels_trans & 1))) return DETOPS_EINVAL;
  a->num_classes = no_trans ? 1 : channels_trans / 2;          // deform_pool_kernel_cuda.cu:289-290
  a->cec = no_trans ? output_dim : output_dim / a->num_classes;
  if (a->cec <= 0) return DETOPS_EINVAL;
  // every position-sensitive plane index (ctop*g + gh)*g + gw must exist in the input
  if (static_cast<int64_t>(output_dim) * group_size * group_size > C) return DETOPS_EINVAL;
  return 0;
}

}  // namespace

DETOPS_API int detops_deform_psroi_pool_forward_f32(
    const float* data, const float* rois, const float* trans, float* out, float* top_count, int N,
    int C, int H, int W, int K, int channels_trans, int no_trans, float spatial_scale, int output_dim,
    int group_size, int pooled_size, int part_size, int sample_per_part, float trans_std,
    detops_stream_t stream) {
  PsArgs a{};
  const int rc = ps_validate(N, C, H, W, K, channels_trans, no_trans, output_dim, group_size,
                             pooled_size, part_size, sample_per_part, &a);
  if (rc) return rc;
  if (K == 0) return 0;
  if (!data || !rois || !out || !top_count || (!no_trans && !trans) || N == 0 || H == 0 || W == 0)
    return DETOPS_EINVAL;
  const int P = pooled_size, S = sample_per_part;
  // channels of a class beyond num_classes*cec (output_dim not divisible) belong to no workgroup in
  // the reference either: class_id = ctop / cec can reach num_classes there and reads past `trans`;
  // refuse that configuration instead of reproducing the out-of-bounds read.
  if (a.cec * a.num_classes != output_dim) return DETOPS_EUNSUPPORTED;
  const bool table = P * P * S * S <= kMaxSamples && P * P <= 1024;
  const dim3 grid(static_cast<unsigned>(K) * a.num_classes);
  if (table)
    hipLaunchKernelGGL(psroi_fwd_kernel<true>, grid, dim3(kBlock), 0, as_stream(stream), data, rois, trans,
                       out, top_count, C, H, W, P, no_trans, spatial_scale, output_dim, group_size,
                       part_size, S, trans_std, a.num_classes, a.cec);
  else
    hipLaunchKernelGGL(psroi_fwd_kernel<false>, grid, dim3(kBlock), 0, as_stream(stream), data, rois, trans,
                       out, top_count, C, H, W, P, no_trans, spatial_scale, output_dim, group_size,
                       part_size, S, trans_std, a.num_classes, a.cec);
  return launch_status();
}

DETOPS_API int detops_deform_psroi_pool_backward_f32(
    const float* out_grad, const float* data, const float* rois, const float* trans,
    const float* top_count, float* data_grad, float* trans_grad, int N, int C, int H, int W, int K,
    int channels_trans, int no_trans, float spatial_scale, int output_dim, int group_size,
    int pooled_size, int part_size, int sample_per_part, float trans_std, int zero_grads,
    detops_stream_t stream) {
  PsArgs a{};
  const int rc = ps_validate(N, C, H, W, K, channels_trans, no_trans, output_dim, group_size,
                             pooled_size, part_size, sample_per_part, &a);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const size_t dbytes = sizeof(float) * static_cast<size_t>(N) * C * H * W;
  const size_t tbytes = no_trans ? 0 : sizeof(float) * static_cast<size_t>(K) * channels_trans * part_size * part_size;
  if ((dbytes && !data_grad) || (tbytes && !trans_grad)) return DETOPS_EINVAL;
  if (zero_grads) {
    if (dbytes) DETOPS_HIP_TRY(hipMemsetAsync(data_grad, 0, dbytes, st));
    if (tbytes) DETOPS_HIP_TRY(hipMemsetAsync(trans_grad, 0, tbytes, st));
  }
  if (K == 0 || dbytes == 0) return 0;
  if (!out_grad || !data || !rois || !top_count || (!no_trans && !trans)) return DETOPS_EINVAL;
  if (a.cec * a.num_classes != output_dim) return DETOPS_EUNSUPPORTED;
  const int P = pooled_size, S = sample_per_part;
  const bool table = P * P * S * S <= kMaxSamples && P * P <= 1024;
  const dim3 grid(static_cast<unsigned>(K) * a.num_classes);
  if (table)
    hipLaunchKernelGGL(psroi_bwd_kernel<true>, grid, dim3(kBlock), 0, st, out_grad, data, rois, trans,
                       top_count, data_grad, trans_grad, C, H, W, P, no_trans, spatial_scale, output_dim,
                       group_size, part_size, S, trans_std, a.num_classes, a.cec);
  else
    hipLaunchKernelGGL(psroi_bwd_kernel<false>, grid, dim3(kBlock), 0, st, out_grad, data, rois, trans,
                       top_count, data_grad, trans_grad, C, H, W, P, no_trans, spatial_scale, output_dim,
                       group_size, part_size, S, trans_std, a.num_classes, a.cec);
  return launch_status();
}
