// deform_conv.hip — deformable convolution (v1 / modulated v2) building blocks for gfx950.
//
// Replaces deformable_im2col / col2im / col2im_coord and their modulated twins
// (reference csrc/cuda/deform_conv_kernel_cuda.cu:197-250, :286-342, :380-443, :577-774) behind
// detops_deformable_{im2col,col2im,col2im_coord}.  fp32 / fp16 / bf16 storage, fp32 arithmetic.
//
// Thread mapping differs from the reference on purpose.  There, every (channel, pixel) thread
// re-reads the 2*kh*kw offsets and re-derives the bilinear weights, i.e. C times per sampling
// point.  Here one thread owns a SAMPLING POINT (b, ho, wo, tap): it reads its offset pair (and
// mask) once, computes the 4 tap indices + weights once, and then walks a chunk of channels.
// Lanes run along wo, so offset loads, column stores/loads and gradient stores are coalesced and
// the 4 gathers of neighbouring lanes fall into neighbouring addresses.
#include <cstdlib>

#include "detops_devlib.h"

namespace {

constexpr int kBlock = 256;

template <typename T> __device__ __forceinline__ float ld(const T* p) { return static_cast<float>(*p); }
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v) { *p = static_cast<T>(v); }
template <> __device__ __forceinline__ void st<__half>(__half* p, float v) { *p = __float2half(v); }
template <> __device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ void atomic_add_t(float* p, float v) { atomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_t(__half* p, float v) {
  // 16-bit atomics: CAS on the containing dword
  unsigned* base = reinterpret_cast<unsigned*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(3));
  const bool hi = reinterpret_cast<uintptr_t>(p) & 2;
  unsigned old = *base, assumed;
  do {
    assumed = old;
    const unsigned short cur = hi ? (assumed >> 16) : (assumed & 0xffffu);
    const __half nh = __float2half(__half2float(__ushort_as_half(cur)) + v);
    const unsigned short nb = __half_as_ushort(nh);
    const unsigned repl = hi ? ((assumed & 0xffffu) | (static_cast<unsigned>(nb) << 16))
                             : ((assumed & 0xffff0000u) | nb);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}
__device__ __forceinline__ void atomic_add_t(__hip_bfloat16* p, float v) {
  unsigned* base = reinterpret_cast<unsigned*>(reinterpret_cast<uintptr_t>(p) & ~uintptr_t(3));
  const bool hi = reinterpret_cast<uintptr_t>(p) & 2;
  unsigned old = *base, assumed;
  do {
    assumed = old;
    const unsigned short cur = hi ? (assumed >> 16) : (assumed & 0xffffu);
    const float f = __uint_as_float(static_cast<unsigned>(cur) << 16) + v;
    const __hip_bfloat16 nbf = __float2bfloat16(f);
    const unsigned short nb = *reinterpret_cast<const unsigned short*>(&nbf);
    const unsigned repl = hi ? ((assumed & 0xffffu) | (static_cast<unsigned>(nb) << 16))
                             : ((assumed & 0xffff0000u) | nb);
    old = atomicCAS(base, assumed, repl);
  } while (old != assumed);
}

// Two adjacent elements (p 4-byte aligned for the 16-bit types) raised by (v0, v1): ONE hardware atomic for half / bf16
// (detops_common.h: detops_atomic_add2).
__device__ __forceinline__ void atomic_add2_t(float* p, float v0, float v1) { atomicAdd(p, v0); atomicAdd(p + 1, v1); }
__device__ __forceinline__ void atomic_add2_t(__half* p, float v0, float v1) { detops_atomic_add2(p, v0, v1); }
__device__ __forceinline__ void atomic_add2_t(__hip_bfloat16* p, float v0, float v1) { detops_atomic_add2(p, v0, v1); }

struct Geom {
  int B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, Ho, Wo;
};

// The 4 bilinear taps of one sampling point (zero padding outside the map):
// deformable_im2col_bilinear, deform_conv_kernel_cuda.cu:91-122.
struct Sample {
  int i1, i2, i3, i4;      // flat indices h*W+w, -1 when that tap is outside
  float w1, w2, w3, w4;    // hh*hw, hh*lw, lh*hw, lh*lw
  float lh, lw;
  int hl, wl;              // floor(h_im), floor(w_im): the low corner (may lie outside the map)
  bool inside;             // h_im > -1 && w_im > -1 && h_im < H && w_im < W   (:236)
};

__device__ __forceinline__ Sample make_sample(float h_im, float w_im, int H, int W) {
  Sample s;
  s.inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < static_cast<float>(H)) &&
             (w_im < static_cast<float>(W));
  const int h_low = static_cast<int>(floorf(h_im)), w_low = static_cast<int>(floorf(w_im));
  const int h_high = h_low + 1, w_high = w_low + 1;
  s.hl = h_low; s.wl = w_low;
  s.lh = h_im - static_cast<float>(h_low);
  s.lw = w_im - static_cast<float>(w_low);
  const float hh = 1.f - s.lh, hw = 1.f - s.lw;
  s.w1 = hh * hw; s.w2 = hh * s.lw; s.w3 = s.lh * hw; s.w4 = s.lh * s.lw;
  const bool t = h_low >= 0, b = h_high <= H - 1, l = w_low >= 0, r = w_high <= W - 1;
  s.i1 = (s.inside && t && l) ? h_low * W + w_low : -1;
  s.i2 = (s.inside && t && r) ? h_low * W + w_high : -1;
  s.i3 = (s.inside && b && l) ? h_high * W + w_low : -1;
  s.i4 = (s.inside && b && r) ? h_high * W + w_high : -1;
  return s;
}

// Decode a sampling-point id: p = ((b*K + tap)*Ho + ho)*Wo + wo   (wo fastest -> coalescing)
struct Point {
  int b, tap, ho, wo, pix;  // pix = ho*Wo + wo
};
__device__ __forceinline__ Point decode(int64_t p, const Geom& g) {
  Point q;
  const int K = g.kh * g.kw;
  q.wo = static_cast<int>(p % g.Wo);
  int64_t r = p / g.Wo;
  q.ho = static_cast<int>(r % g.Ho);
  r /= g.Ho;
  q.tap = static_cast<int>(r % K);
  q.b = static_cast<int>(r / K);
  q.pix = q.ho * g.Wo + q.wo;
  return q;
}

template <typename T>
__device__ __forceinline__ Sample point_sample(const Point& q, const Geom& g, int dgi,
                                               const T* __restrict__ offset) {
  const int K = g.kh * g.kw;
  const int i = q.tap / g.kw, j = q.tap - i * g.kw;
  const size_t HWo = static_cast<size_t>(g.Ho) * g.Wo;
  const T* op = offset + (static_cast<size_t>(q.b) * g.dg + dgi) * 2 * K * HWo;
  const float off_h = ld(op + (2 * q.tap) * HWo + q.pix);
  const float off_w = ld(op + (2 * q.tap + 1) * HWo + q.pix);
  const float h_im = static_cast<float>(q.ho * g.stride_h - g.pad_h + i * g.dil_h) + off_h;
  const float w_im = static_cast<float>(q.wo * g.stride_w - g.pad_w + j * g.dil_w) + off_w;
  return make_sample(h_im, w_im, g.H, g.W);
}

// ------------------------------------------------------------------------------------ im2col
// grid.x over sampling points, grid.y over (deformable group, channel chunk)
template <typename T>
__global__ void __launch_bounds__(kBlock)
im2col_kernel(const T* __restrict__ im, const T* __restrict__ offset, const T* __restrict__ mask,
              T* __restrict__ col, Geom g, int cchunk, int64_t npoints) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= npoints) return;
  const int cpg = g.C / g.dg;
  const int chunks_per_g = (cpg + cchunk - 1) / cchunk;
  const int dgi = blockIdx.y / chunks_per_g;
  const int c0 = dgi * cpg + (blockIdx.y - dgi * chunks_per_g) * cchunk;
  const int c1 = min(c0 + cchunk, (dgi + 1) * cpg);
  const Point q = decode(p, g);
  const int K = g.kh * g.kw;
  const Sample s = point_sample(q, g, dgi, offset);
  float m = 1.f;
  if (mask)
    m = ld(mask + ((static_cast<size_t>(q.b) * g.dg + dgi) * K + q.tap) * g.Ho * g.Wo + q.pix);
  const size_t plane = static_cast<size_t>(g.H) * g.W;
  const size_t ncol = static_cast<size_t>(g.B) * g.Ho * g.Wo;
  const T* ip = im + (static_cast<size_t>(q.b) * g.C + c0) * plane;
  T* cp = col + (static_cast<size_t>(c0) * K + q.tap) * ncol + static_cast<size_t>(q.b) * g.Ho * g.Wo + q.pix;
  // Taps outside the map read element 0 and are replaced by 0 afterwards: the loads become
  // unconditional, so 4 channels x 4 taps are in flight before the first use (the loop is
  // latency-bound otherwise: one L2 round trip per channel).
  const bool t1 = s.i1 >= 0, t2 = s.i2 >= 0, t3 = s.i3 >= 0, t4 = s.i4 >= 0;
  const int j1 = t1 ? s.i1 : 0, j2 = t2 ? s.i2 : 0, j3 = t3 ? s.i3 : 0, j4 = t4 ? s.i4 : 0;
  const size_t cstep = static_cast<size_t>(K) * ncol;
  constexpr int U = 4;
  for (int c = c0; c < c1; c += U) {
    float v1[U], v2[U], v3[U], v4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const T* pu = ip + static_cast<size_t>(min(u, c1 - 1 - c)) * plane;  // tail: re-read the last channel
      v1[u] = ld(pu + j1); v2[u] = ld(pu + j2); v3[u] = ld(pu + j3); v4[u] = ld(pu + j4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c + u < c1) {
        const float a1 = t1 ? v1[u] : 0.f, a2 = t2 ? v2[u] : 0.f, a3 = t3 ? v3[u] : 0.f, a4 = t4 ? v4[u] : 0.f;
        const float v = s.inside ? (s.w1 * a1 + s.w2 * a2 + s.w3 * a3 + s.w4 * a4) : 0.f;
        st(cp + static_cast<size_t>(u) * cstep, mask ? v * m : v);
      }
    }
    ip += static_cast<size_t>(U) * plane;
    cp += static_cast<size_t>(U) * cstep;
  }
}

// ------------------------------------------------------------------------------------ col2im
template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_kernel(const T* __restrict__ col, const T* __restrict__ offset, const T* __restrict__ mask,
              T* __restrict__ grad_im, Geom g, int cchunk, int64_t npoints) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= npoints) return;
  const int cpg = g.C / g.dg;
  const int chunks_per_g = (cpg + cchunk - 1) / cchunk;
  const int dgi = blockIdx.y / chunks_per_g;
  const int c0 = dgi * cpg + (blockIdx.y - dgi * chunks_per_g) * cchunk;
  const int c1 = min(c0 + cchunk, (dgi + 1) * cpg);
  const Point q = decode(p, g);
  const int K = g.kh * g.kw;
  const Sample s = point_sample(q, g, dgi, offset);
  if (!s.inside) return;  // get_gradient_weight returns 0 outside (:127-131)
  float m = 1.f;
  if (mask)
    m = ld(mask + ((static_cast<size_t>(q.b) * g.dg + dgi) * K + q.tap) * g.Ho * g.Wo + q.pix);
  const size_t plane = static_cast<size_t>(g.H) * g.W;
  const size_t ncol = static_cast<size_t>(g.B) * g.Ho * g.Wo;
  T* gp = grad_im + (static_cast<size_t>(q.b) * g.C + c0) * plane;
  const T* cp = col + (static_cast<size_t>(c0) * K + q.tap) * ncol + static_cast<size_t>(q.b) * g.Ho * g.Wo + q.pix;
  for (int c = c0; c < c1; ++c) {
    const float gv = ld(cp) * m;  // cur_top_grad (:322 / :681)
    if (s.i1 >= 0) atomic_add_t(gp + s.i1, s.w1 * gv);
    if (s.i2 >= 0) atomic_add_t(gp + s.i2, s.w2 * gv);
    if (s.i3 >= 0) atomic_add_t(gp + s.i3, s.w3 * gv);
    if (s.i4 >= 0) atomic_add_t(gp + s.i4, s.w4 * gv);
    gp += plane;
    cp += static_cast<size_t>(K) * ncol;
  }
}

// ------------------------------------------------------------------------------------ col2im, tiled
// The kernel above issues 4 global atomics per column element (36 per image pixel for a 3x3 kernel)
// and measures ~40 GB/s.  Here a workgroup owns a TY x TX tile of OUTPUT positions of one image for
// a chunk of channels: a thread owns one output position, derives each tap's bilinear footprint
// once, and accumulates CC channels at a time into an LDS window that covers the tile's input
// footprint plus a halo for the learned offsets (ds_add_f32).  Taps that land outside the window
// (|offset| > halo) go straight to the global map.  The window is then flushed with one
// row-contiguous global atomic per non-zero element: ~(window/tile) = 3 atomics per image pixel
// instead of 36, and every column element is read exactly once, coalesced along wo.
constexpr int kColTY = 8, kColTX = 32, kColHalo = 4;
constexpr int kColLdsFloats = 8192 - 32;
constexpr int kColCC = 8;  // channels accumulated per LDS pass

template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_tile_kernel(const T* __restrict__ col, const T* __restrict__ offset, const T* __restrict__ mask,
                   T* __restrict__ grad_im, Geom g, int cchunk, int tiles_x, int tiles_y, int RH, int RW,
                   int CC) {
  DETOPS_DYNAMIC_LDS(float, win);  // [CC][RH*RW]
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tix = t % tiles_x; t /= tiles_x;
  const int tiy = t % tiles_y;
  const int b = t / tiles_y;
  const int cpg = g.C / g.dg;
  const int chunks_per_g = (cpg + cchunk - 1) / cchunk;
  const int dgi = blockIdx.y / chunks_per_g;
  const int c0 = dgi * cpg + (blockIdx.y - dgi * chunks_per_g) * cchunk;
  const int c1 = min(c0 + cchunk, (dgi + 1) * cpg);
  const int ho = tiy * kColTY + tid / kColTX;
  const int wo = tix * kColTX + (tid & (kColTX - 1));
  const bool live = ho < g.Ho && wo < g.Wo;
  const int pix = ho * g.Wo + wo;
  const int K = g.kh * g.kw;
  const size_t HWo = static_cast<size_t>(g.Ho) * g.Wo;
  const size_t plane = static_cast<size_t>(g.H) * g.W;
  const size_t ncol = static_cast<size_t>(g.B) * HWo;
  // window origin in input coordinates
  const int ry0 = tiy * kColTY * g.stride_h - g.pad_h - kColHalo;
  const int rx0 = tix * kColTX * g.stride_w - g.pad_w - kColHalo;
  const int rarea = RH * RW;
  const T* op = offset + (static_cast<size_t>(b) * g.dg + dgi) * 2 * K * HWo;
  const T* mp = mask ? mask + (static_cast<size_t>(b) * g.dg + dgi) * K * HWo : nullptr;

  for (int cs = c0; cs < c1; cs += CC) {
    const int cn = min(CC, c1 - cs);
    for (int e = tid; e < cn * rarea; e += kBlock) win[e] = 0.f;
    __syncthreads();
    if (live) {
      T* gp0 = grad_im + (static_cast<size_t>(b) * g.C + cs) * plane;
      for (int tap = 0; tap < K; ++tap) {
        const int i = tap / g.kw, j = tap - i * g.kw;
        const float h_im = static_cast<float>(ho * g.stride_h - g.pad_h + i * g.dil_h) + ld(op + (2 * tap) * HWo + pix);
        const float w_im = static_cast<float>(wo * g.stride_w - g.pad_w + j * g.dil_w) + ld(op + (2 * tap + 1) * HWo + pix);
        const Sample s = make_sample(h_im, w_im, g.H, g.W);
        if (!s.inside) continue;  // get_gradient_weight returns 0 outside (:127-131)
        const float m = mp ? ld(mp + tap * HWo + pix) : 1.f;
        const int h_low = static_cast<int>(floorf(h_im)), w_low = static_cast<int>(floorf(w_im));
        const int ly = h_low - ry0, lx = w_low - rx0;
        const bool in_win = ly >= 0 && ly + 1 < RH && lx >= 0 && lx + 1 < RW;
        const int wbase = ly * RW + lx;
        const T* cp = col + (static_cast<size_t>(cs) * K + tap) * ncol + static_cast<size_t>(b) * HWo + pix;
        // all channel values of this tap first (kColCC independent loads in flight), then the scatter:
        // with the load inside the scatter loop every iteration waits one HBM round trip
        float gvs[kColCC];
#pragma unroll
        for (int c = 0; c < kColCC; ++c)
          gvs[c] = c < cn ? ld(cp + static_cast<size_t>(c) * K * ncol) * m : 0.f;  // cur_top_grad (:322 / :681)
#pragma unroll
        for (int c = 0; c < kColCC; ++c) {
          if (c >= cn) break;
          const float gv = gvs[c];
          if (in_win) {
            float* w = win + c * rarea + wbase;
            if (s.i1 >= 0) atomicAdd(w, s.w1 * gv);
            if (s.i2 >= 0) atomicAdd(w + 1, s.w2 * gv);
            if (s.i3 >= 0) atomicAdd(w + RW, s.w3 * gv);
            if (s.i4 >= 0) atomicAdd(w + RW + 1, s.w4 * gv);
          } else {
            T* gp = gp0 + static_cast<size_t>(c) * plane;
            if (s.i1 >= 0) atomic_add_t(gp + s.i1, s.w1 * gv);
            if (s.i2 >= 0) atomic_add_t(gp + s.i2, s.w2 * gv);
            if (s.i3 >= 0) atomic_add_t(gp + s.i3, s.w3 * gv);
            if (s.i4 >= 0) atomic_add_t(gp + s.i4, s.w4 * gv);
          }
        }
      }
    }
    __syncthreads();
    // flush: lanes run along the window row -> row-contiguous atomics
    for (int e = tid; e < cn * rarea; e += kBlock) {
      const float v = win[e];
      if (v == 0.f) continue;
      const int c = e / rarea;
      const int r = e - c * rarea;
      const int y = r / RW;
      const int x = r - y * RW;
      const int iy = ry0 + y, ix = rx0 + x;
      if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
        atomic_add_t(grad_im + (static_cast<size_t>(b) * g.C + cs + c) * plane + static_cast<size_t>(iy) * g.W + ix, v);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ col2im, gather
// The scatter kernels above are bound by atomic throughput: 4 atomics per (sampling point, channel)
// — 155 M for one layer2-sized call — and both the global and the LDS (ds_add_f32) form retire
// roughly one lane every four clocks per CU (measured: 959 us for 155 M LDS atomics).  But the
// sampling geometry is shared by all channels of a deformable group, so the scatter pattern can be
// INVERTED once per call and reused by every channel:
//
//   1. count   one thread per sampling point (b, dg, tap, ho, wo): +1 (int atomic) on the
//              (target pixel, tap) counter of each of its <= 4 bilinear targets   [B*dg*H*W*K ints]
//   2. scan    exclusive prefix sum of the counters (hipCUB) -> sub-list start per (pixel, tap);
//              a pixel's K sub-lists are adjacent, so its whole list is one contiguous range
//   3. fill    the same threads again: claim a slot in each target's sub-list and store
//              (column index of the sampling point, bilinear weight * mask)
//   4. sort    one thread per (pixel, tap) orders its handful of entries by column index: the
//              summation order becomes deterministic, and neighbouring pixels walk their lists
//              tap by tap in step, which makes the gathers below coalesce
//   5. gather  one thread per (pixel, channel chunk): grad_im[b,c,y,x] += sum_e w_e * col[c, e]
//              — plain loads, register accumulation, coalesced stores; no atomics on data.
//
// Offsets of any magnitude are handled uniformly (no window / halo assumption).  Needs a caller
// workspace (detops_deformable_col2im_workspace_bytes); without one the scatter path runs.
struct ColEntry {
  int32_t colidx;  // tap * ncol + b * Ho*Wo + ho*Wo + wo : offset inside one channel's K rows of `col`
  float w;
};

template <typename T, bool FILL>
__global__ void __launch_bounds__(kBlock)
col2im_index_kernel(const T* __restrict__ offset, const T* __restrict__ mask, Geom g, int64_t npoints,
                    int32_t* __restrict__ counter, const int32_t* __restrict__ start,
                    ColEntry* __restrict__ entries) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= npoints) return;
  const int dgi = blockIdx.y;
  const Point q = decode(p, g);
  const int K = g.kh * g.kw;
  const Sample s = point_sample(q, g, dgi, offset);
  if (!s.inside) return;
  const int HWo = g.Ho * g.Wo;
  const size_t base = static_cast<size_t>(q.b * g.dg + dgi) * g.H * g.W;
  const int tgt[4] = {s.i1, s.i2, s.i3, s.i4};
  const float wgt[4] = {s.w1, s.w2, s.w3, s.w4};
  float m = 1.f;
  if (FILL && mask) m = ld(mask + ((static_cast<size_t>(q.b) * g.dg + dgi) * K + q.tap) * HWo + q.pix);
  const int32_t colidx = q.tap * (g.B * HWo) + q.b * HWo + q.pix;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (tgt[t] < 0) continue;
    const size_t slot = (base + tgt[t]) * K + q.tap;
    const int pos = atomicAdd(counter + slot, 1);  // count pass: the tally; fill pass: the claimed slot
    if (FILL) entries[start[slot] + pos] = ColEntry{colidx, wgt[t] * m};
  }
}

// order each (pixel, tap) sub-list by column index (a handful of entries: insertion sort)
__global__ void __launch_bounds__(kBlock)
col2im_sort_kernel(const int32_t* __restrict__ start, int64_t nslots, ColEntry* __restrict__ entries) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= nslots) return;
  const int s0 = start[i], n = start[i + 1] - s0;
  ColEntry* e = entries + s0;
  for (int a = 1; a < n; ++a) {
    const ColEntry x = e[a];
    int b = a - 1;
    while (b >= 0 && e[b].colidx > x.colidx) { e[b + 1] = e[b]; --b; }
    e[b + 1] = x;
  }
}

constexpr int kGatherCC = 16;  // channels per thread pass (independent loads in flight)

template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_gather_kernel(const T* __restrict__ col, const int32_t* __restrict__ start,
                     const ColEntry* __restrict__ entries, T* __restrict__ grad_im, Geom g, int cchunk,
                     int xcd_remap) {
  const int HW = g.H * g.W;
  // (pixel block, channel chunk, image) from the linear workgroup id; with xcd_remap every XCD owns
  // a contiguous run of pixel blocks of one chunk, so the overlapping column windows of
  // neighbouring pixel blocks are fetched into ONE L2 instead of all eight
  int64_t lin = (static_cast<int64_t>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (xcd_remap) lin = xcd_contiguous(lin, static_cast<int64_t>(gridDim.x) * gridDim.y * gridDim.z);
  const int bx = static_cast<int>(lin % gridDim.x);
  const int by = static_cast<int>((lin / gridDim.x) % gridDim.y);
  const int b = static_cast<int>(lin / (static_cast<int64_t>(gridDim.x) * gridDim.y));
  const int pix = bx * kBlock + threadIdx.x;   // pixel within the image plane
  if (pix >= HW) return;
  const int cpg = g.C / g.dg;
  const int chunks_per_g = (cpg + cchunk - 1) / cchunk;
  const int dgi = by / chunks_per_g;
  const int c0 = dgi * cpg + (by - dgi * chunks_per_g) * cchunk;
  const int c1 = min(c0 + cchunk, (dgi + 1) * cpg);
  const int K = g.kh * g.kw;
  const size_t chan_stride = static_cast<size_t>(K) * g.B * g.Ho * g.Wo;   // one channel's K rows of col
  const size_t li = (static_cast<size_t>(b) * g.dg + dgi) * HW + pix;
  const int s0 = start[li * K], s1 = start[li * K + K];
  for (int cs = c0; cs < c1; cs += kGatherCC) {
    float acc[kGatherCC];
#pragma unroll
    for (int c = 0; c < kGatherCC; ++c) acc[c] = 0.f;
    const T* cbase = col + static_cast<size_t>(cs) * chan_stride;
    for (int j = s0; j < s1; ++j) {
      const ColEntry en = entries[j];
      const T* cp = cbase + en.colidx;
#pragma unroll
      for (int c = 0; c < kGatherCC; ++c)
        if (cs + c < c1) acc[c] = fmaf(en.w, ld(cp + static_cast<size_t>(c) * chan_stride), acc[c]);
    }
    T* gp = grad_im + (static_cast<size_t>(b) * g.C + cs) * HW + pix;
#pragma unroll
    for (int c = 0; c < kGatherCC; ++c)
      if (cs + c < c1) st(gp + static_cast<size_t>(c) * HW, ld(gp + static_cast<size_t>(c) * HW) + acc[c]);
  }
}

// ------------------------------------------------------------------------------------ col2im_coord
// One thread per sampling point produces BOTH offset gradients (d/dh, d/dw) and the mask gradient,
// looping over the channels of its deformable group in ascending order (the reference's
// accumulation order, :413-439 / :738-766).
template <typename T, int S>
__global__ void __launch_bounds__(kBlock)
col2im_coord_kernel(const T* __restrict__ col, const T* __restrict__ im, const T* __restrict__ offset,
                    const T* __restrict__ mask, T* __restrict__ grad_offset, T* __restrict__ grad_mask,
                    Geom g, int64_t npoints_per_dg) {
  // S channel slices per sampling point: lanes run along the points (coalesced), the slices of a
  // point sit in different waves / wave quarters and are combined through LDS in slice order.
  constexpr int PTS = kBlock / S;
  __shared__ float s_part[3][kBlock];
  const int pt = threadIdx.x % PTS, slice = threadIdx.x / PTS;
  const int64_t p = static_cast<int64_t>(blockIdx.x) * PTS + pt;
  const bool live = p < npoints_per_dg;
  const int dgi = blockIdx.y;
  const int cpg = g.C / g.dg;
  const Point q = decode(live ? p : 0, g);
  const int K = g.kh * g.kw;
  const Sample s = point_sample(q, g, dgi, offset);
  const size_t HWo = static_cast<size_t>(g.Ho) * g.Wo;
  float m = 1.f;
  if (mask) m = ld(mask + ((static_cast<size_t>(q.b) * g.dg + dgi) * K + q.tap) * HWo + q.pix);
  float gh = 0.f, gw = 0.f, gm = 0.f;
  if (live && s.inside) {
    const size_t plane = static_cast<size_t>(g.H) * g.W;
    const size_t cstep = static_cast<size_t>(K) * g.B * HWo;
    const int per = (cpg + S - 1) / S;
    const int cs = slice * per, ce = min(cpg, cs + per);
    const int c0 = dgi * cpg + cs;
    const T* ip = im + (static_cast<size_t>(q.b) * g.C + c0) * plane;
    const T* cp = col + (static_cast<size_t>(c0) * K + q.tap) * (static_cast<size_t>(g.B) * HWo) +
                  static_cast<size_t>(q.b) * HWo + q.pix;
    const float hw = 1.f - s.lw, hh = 1.f - s.lh;
    const bool t1 = s.i1 >= 0, t2 = s.i2 >= 0, t3 = s.i3 >= 0, t4 = s.i4 >= 0;
    const int j1 = t1 ? s.i1 : 0, j2 = t2 ? s.i2 : 0, j3 = t3 ? s.i3 : 0, j4 = t4 ? s.i4 : 0;
    constexpr int U = 4;  // channels in flight (5 independent loads each)
    for (int c = cs; c < ce; c += U) {
      float gv[U], v1[U], v2[U], v3[U], v4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int uu = min(u, ce - 1 - c);  // tail: re-read the last channel, discarded below
        const T* pu = ip + static_cast<size_t>(uu) * plane;
        gv[u] = ld(cp + static_cast<size_t>(uu) * cstep);
        v1[u] = ld(pu + j1); v2[u] = ld(pu + j2); v3[u] = ld(pu + j3); v4[u] = ld(pu + j4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c + u < ce) {
          const float a1 = t1 ? v1[u] : 0.f, a2 = t2 ? v2[u] : 0.f, a3 = t3 ? v3[u] : 0.f, a4 = t4 ? v4[u] : 0.f;
          // get_coordinate_weight, bp_dir 0 (:170-180) and 1 (:181-191)
          const float wh = -hw * a1 - s.lw * a2 + hw * a3 + s.lw * a4;
          const float ww = -hh * a1 + hh * a2 - s.lh * a3 + s.lh * a4;
          gh += wh * gv[u] * m;
          gw += ww * gv[u] * m;
          gm += gv[u] * (s.w1 * a1 + s.w2 * a2 + s.w3 * a3 + s.w4 * a4);  // :760
        }
      }
      ip += static_cast<size_t>(U) * plane;
      cp += static_cast<size_t>(U) * cstep;
    }
  }
  if (S > 1) {
    s_part[0][threadIdx.x] = gh; s_part[1][threadIdx.x] = gw; s_part[2][threadIdx.x] = gm;
    __syncthreads();
    if (slice == 0) {
      for (int j = 1; j < S; ++j) {
        gh += s_part[0][j * PTS + pt]; gw += s_part[1][j * PTS + pt]; gm += s_part[2][j * PTS + pt];
      }
    }
  }
  if (!live || slice != 0) return;
  T* gop = grad_offset + (static_cast<size_t>(q.b) * g.dg + dgi) * 2 * K * HWo;
  st(gop + (2 * q.tap) * HWo + q.pix, gh);
  st(gop + (2 * q.tap + 1) * HWo + q.pix, gw);
  if (mask && grad_mask)
    st(grad_mask + ((static_cast<size_t>(q.b) * g.dg + dgi) * K + q.tap) * HWo + q.pix, gm);
}

inline int make_geom(Geom& g, int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                     int stride_h, int stride_w, int dil_h, int dil_w, int dg) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 || stride_w <= 0 ||
      dil_h <= 0 || dil_w <= 0 || dg <= 0 || C % dg != 0 || pad_h < 0 || pad_w < 0)
    return DETOPS_EINVAL;
  g = Geom{B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, 0, 0};
  g.Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  g.Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  if (g.Ho < 1 || g.Wo < 1) return DETOPS_EINVAL;
  return 0;
}

inline int pick_cchunk(int cpg, int64_t npoints) {
  // enough workgroups to fill the chip, but keep >= 16 channels per thread to amortise the setup
  int cc = cpg;
  while (cc > 16 && ceil_div64(npoints, kBlock) * ceil_div64(cpg, cc) < 8 * kNumCU) cc = (cc + 1) / 2;
  return cc;
}

template <typename T>
int im2col_t(const void* im, const void* offset, const void* mask, void* col, const Geom& g,
             hipStream_t st_) {
  const int64_t np = static_cast<int64_t>(g.B) * g.kh * g.kw * g.Ho * g.Wo;
  if (np == 0) return 0;
  const int cpg = g.C / g.dg;
  const int cc = pick_cchunk(cpg, np);
  const dim3 grid(static_cast<unsigned>(ceil_div64(np, kBlock)),
                  static_cast<unsigned>(g.dg * ceil_div64(cpg, cc)));
  hipLaunchKernelGGL(im2col_kernel<T>, grid, dim3(kBlock), 0, st_, static_cast<const T*>(im),
                     static_cast<const T*>(offset), static_cast<const T*>(mask), static_cast<T*>(col),
                     g, cc, np);
  return launch_status();
}

template <typename T>
int col2im_t(const void* col, const void* offset, const void* mask, void* grad_im, const Geom& g,
             hipStream_t st_) {
  const int64_t np = static_cast<int64_t>(g.B) * g.kh * g.kw * g.Ho * g.Wo;
  if (np == 0) return 0;
  const int cpg = g.C / g.dg;
  // tiled path: LDS window = tile footprint + halo
  const int RH = (kColTY - 1) * g.stride_h + (g.kh - 1) * g.dil_h + 2 + 2 * kColHalo;
  const int RW = ((kColTX - 1) * g.stride_w + (g.kw - 1) * g.dil_w + 2 + 2 * kColHalo) | 1;  // odd: bank spread
  const int CC = min(kColCC, kColLdsFloats / (RH * RW));
  if (CC >= 1) {
    const int tiles_x = static_cast<int>(ceil_div64(g.Wo, kColTX)), tiles_y = static_cast<int>(ceil_div64(g.Ho, kColTY));
    const int64_t tiles = static_cast<int64_t>(g.B) * tiles_x * tiles_y;
    int cc = cpg;  // channels per workgroup: multiples of CC, enough workgroups to fill the chip
    while (cc > CC && tiles * ceil_div64(cpg, cc) < 4 * kNumCU) cc = max(CC, (cc + 1) / 2);
    cc = static_cast<int>(ceil_div64(cc, CC)) * CC;
    const dim3 grid(static_cast<unsigned>(tiles), static_cast<unsigned>(g.dg * ceil_div64(cpg, cc)));
    hipLaunchKernelGGL(col2im_tile_kernel<T>, grid, dim3(kBlock), sizeof(float) * CC * RH * RW, st_,
                       static_cast<const T*>(col), static_cast<const T*>(offset), static_cast<const T*>(mask),
                       static_cast<T*>(grad_im), g, cc, tiles_x, tiles_y, RH, RW, CC);
    return launch_status();
  }
  const int cc = pick_cchunk(cpg, np);
  const dim3 grid(static_cast<unsigned>(ceil_div64(np, kBlock)),
                  static_cast<unsigned>(g.dg * ceil_div64(cpg, cc)));
  hipLaunchKernelGGL(col2im_kernel<T>, grid, dim3(kBlock), 0, st_, static_cast<const T*>(col),
                     static_cast<const T*>(offset), static_cast<const T*>(mask),
                     static_cast<T*>(grad_im), g, cc, np);
  return launch_status();
}

inline size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

// ------------------------------------------------------------------------------------ col2im, ELL gather
// Default for fp32 and for large maps (measured r02a: 1.3-1.75x over the CSR pipeline at layer2/3).
// rocprofv3 of the CSR pipeline above (profiles/r01e_opbench_kernel_stats.csv) shows that at layer2 size
// the index build is a third of the time (count 47 + fill 90 + sort 150 us of 817) and that the gather
// reads its per-pixel entry lists with one cache line per lane.  Here the inverted index is a fixed-width
// table instead: every (pixel, tap) owns kEllCap slots, one 64-byte record ([b*dg][tap][pixel][slot]),
// so ONE pass fills it (slot = atomicAdd on the (pixel, tap) counter), a coalesced pass sorts each
// slot column by column index (deterministic summation order), and the gather reads counters and
// entries coalesced across the wave.  The rare (pixel, tap) with more than kEllCap contributions
// spills to an overflow list that a small atomic kernel adds afterwards.
constexpr int kEllCap = 8;

struct __attribute__((aligned(8))) EllEntry {   // one 8-byte record per slot: a scattered fill writes ONE line per entry
  int32_t idx;      // column index: tap * B * Ho * Wo + b * Ho * Wo + output pixel
  float w;          // bilinear weight (x mask)
};

struct EllOverflow {
  int32_t li;       // (b*dg + dgi) * H*W + pixel
  int32_t colidx;
  float w;
};

__device__ __forceinline__ int slot_fetch_add(int32_t* p) { return detops_fetch_add_relaxed(p, 1); }

template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_ell_fill_kernel(const T* __restrict__ offset, const T* __restrict__ mask, Geom g, int64_t npoints,
                       int32_t* __restrict__ counter, EllEntry* __restrict__ ent,
                       int32_t* __restrict__ ovf_count, EllOverflow* __restrict__ ovf, int ovf_cap) {
  const int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= npoints) return;
  const int dgi = blockIdx.y;
  const Point q = decode(p, g);
  const int K = g.kh * g.kw;
  const Sample s = point_sample(q, g, dgi, offset);
  if (!s.inside) return;
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo;
  const int tgt[4] = {s.i1, s.i2, s.i3, s.i4};
  const float wgt[4] = {s.w1, s.w2, s.w3, s.w4};
  float m = 1.f;
  if (mask) m = ld(mask + ((static_cast<size_t>(q.b) * g.dg + dgi) * K + q.tap) * HWo + q.pix);
  const int32_t colidx = q.tap * (g.B * HWo) + q.b * HWo + q.pix;
  const size_t img = static_cast<size_t>(q.b) * g.dg + dgi;
  // the four slot requests first (independent relaxed atomics: four round trips in flight instead of one after the
  // other), then the four record stores
  int pos[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    pos[t] = -1;
    if (tgt[t] >= 0) pos[t] = slot_fetch_add(counter + (img * K + q.tap) * HW + tgt[t]);   // counter: [img][tap][pixel]
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (tgt[t] < 0) continue;
    if (pos[t] < kEllCap) {
      const size_t e = ((img * K + q.tap) * HW + tgt[t]) * kEllCap + pos[t];   // [img][tap][pixel][slot]
      ent[e] = EllEntry{colidx, wgt[t] * m};
    } else {
      const int o = atomicAdd(ovf_count, 1);
      if (o < ovf_cap) ovf[o] = EllOverflow{static_cast<int32_t>(img * HW + tgt[t]), colidx, wgt[t] * m};
    }
  }
}

// sort the (<= kEllCap) entries of each (pixel, tap) by column index: coalesced loads / stores, in registers
__global__ void __launch_bounds__(kBlock)
col2im_ell_sort_kernel(const int32_t* __restrict__ counter, int64_t ncols, int HW, EllEntry* __restrict__ ent) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;   // (img, tap, pixel) linear
  if (i >= ncols) return;
  const int n = min(counter[i], kEllCap);
  if (n < 2) return;
  const int64_t it = i / HW;             // img * K + tap
  const int pix = static_cast<int>(i - it * HW);
  EllEntry* ee = ent + (static_cast<size_t>(it) * HW + pix) * kEllCap;
  int32_t k[kEllCap];
  float w[kEllCap];
#pragma unroll
  for (int j = 0; j < kEllCap; ++j) {
    const EllEntry r = j < n ? ee[j] : EllEntry{0x7fffffff, 0.f};
    k[j] = r.idx;
    w[j] = r.w;
  }
  // odd-even transposition sort on 8 registers (static indices)
#pragma unroll
  for (int r = 0; r < kEllCap; ++r) {
#pragma unroll
    for (int j = (r & 1); j + 1 < kEllCap; j += 2) {
      const bool sw = k[j] > k[j + 1];
      const int32_t ka = sw ? k[j + 1] : k[j], kb = sw ? k[j] : k[j + 1];
      const float wa = sw ? w[j + 1] : w[j], wb = sw ? w[j] : w[j + 1];
      k[j] = ka; k[j + 1] = kb; w[j] = wa; w[j + 1] = wb;
    }
  }
#pragma unroll
  for (int j = 0; j < kEllCap; ++j)
    if (j < n) ee[j] = EllEntry{k[j], w[j]};
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_ell_gather_kernel(const T* __restrict__ col, const int32_t* __restrict__ counter,
                         const EllEntry* __restrict__ ent, T* __restrict__ grad_im, Geom g, int cchunk, int xcd_remap) {
  const int HW = g.H * g.W;
  int64_t lin = (static_cast<int64_t>(blockIdx.z) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  if (xcd_remap) lin = xcd_contiguous(lin, static_cast<int64_t>(gridDim.x) * gridDim.y * gridDim.z);
  const int bx = static_cast<int>(lin % gridDim.x);
  const int by = static_cast<int>((lin / gridDim.x) % gridDim.y);
  const int b = static_cast<int>(lin / (static_cast<int64_t>(gridDim.x) * gridDim.y));
  const int pix = bx * kBlock + threadIdx.x;
  const bool live = pix < HW;            // dead lanes stay for the wave-wide ballots
  const int cpg = g.C / g.dg;
  const int chunks_per_g = (cpg + cchunk - 1) / cchunk;
  const int dgi = by / chunks_per_g;
  const int c0 = dgi * cpg + (by - dgi * chunks_per_g) * cchunk;
  const int c1 = min(c0 + cchunk, (dgi + 1) * cpg);
  const int K = g.kh * g.kw;
  const size_t chan_stride = static_cast<size_t>(K) * g.B * g.Ho * g.Wo;
  const size_t img = static_cast<size_t>(b) * g.dg + dgi;
  const int p = live ? pix : 0;
  for (int cs = c0; cs < c1; cs += kGatherCC) {
    float acc[kGatherCC];
#pragma unroll
    for (int c = 0; c < kGatherCC; ++c) acc[c] = 0.f;
    const T* cbase = col + static_cast<size_t>(cs) * chan_stride;
    for (int tap = 0; tap < K; ++tap) {
      const int n = live ? min(counter[(img * K + tap) * HW + p], kEllCap) : 0;
      const EllEntry* ee = ent + ((img * K + tap) * HW + p) * kEllCap;
      for (int j = 0; j < kEllCap; ++j) {
        if (__ballot(j < n) == 0ull) break;        // wave-uniform trip count
        if (j < n) {
          const EllEntry r = ee[j];
          const int32_t ci = r.idx;
          const float w = r.w;
          const T* cp = cbase + ci;
#pragma unroll
          for (int c = 0; c < kGatherCC; ++c)
            if (cs + c < c1) acc[c] = fmaf(w, ld(cp + static_cast<size_t>(c) * chan_stride), acc[c]);
        }
      }
    }
    if (live) {
      T* gp = grad_im + (static_cast<size_t>(b) * g.C + cs) * HW + pix;
#pragma unroll
      for (int c = 0; c < kGatherCC; ++c)
        if (cs + c < c1) st(gp + static_cast<size_t>(c) * HW, ld(gp + static_cast<size_t>(c) * HW) + acc[c]);
    }
  }
}

// contributions beyond kEllCap per (pixel, tap): added with atomics after the gather (rare)
template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_ell_overflow_kernel(const T* __restrict__ col, const int32_t* __restrict__ ovf_count,
                           const EllOverflow* __restrict__ ovf, int ovf_cap, T* __restrict__ grad_im, Geom g) {
  const int n = min(*ovf_count, ovf_cap);
  const int HW = g.H * g.W;
  const int cpg = g.C / g.dg;
  const size_t chan_stride = static_cast<size_t>(g.kh) * g.kw * g.B * g.Ho * g.Wo;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < static_cast<int64_t>(n) * cpg;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int e = static_cast<int>(i / cpg);
    const int cl = static_cast<int>(i - static_cast<int64_t>(e) * cpg);
    const EllOverflow o = ovf[e];
    const int imgi = o.li / HW, pix = o.li - imgi * HW;
    const int b = imgi / g.dg, dgi = imgi - b * g.dg;
    const int c = dgi * cpg + cl;
    const float v = o.w * ld(col + static_cast<size_t>(c) * chan_stride + o.colidx);
    atomic_add_t(grad_im + (static_cast<size_t>(b) * g.C + c) * HW + pix, v);
  }
}

struct EllPlan {
  int64_t ncols, npoints_per_dg;     // ncols = B*dg*K*H*W  (pixel, tap) columns
  int ovf_cap;
  size_t off_count, off_ovf_count, off_ent, off_ovf, total;
};

inline bool ell_plan(const Geom& g, EllPlan& P) {
  const int64_t K = static_cast<int64_t>(g.kh) * g.kw;
  const int64_t HWo = static_cast<int64_t>(g.Ho) * g.Wo;
  P.ncols = static_cast<int64_t>(g.B) * g.dg * K * g.H * g.W;
  P.npoints_per_dg = static_cast<int64_t>(g.B) * K * HWo;
  const int64_t max_entries = 4 * P.npoints_per_dg * g.dg;
  if (P.ncols * kEllCap > 0x7fffffff || max_entries > 0x7fffffff || K * g.B * HWo > 0x7fffffff ||
      static_cast<int64_t>(g.B) * g.dg * g.H * g.W > 0x7fffffff)
    return false;
  P.ovf_cap = static_cast<int>(max_entries);
  size_t o = 0;
  P.off_count = o;     o = align256(o + sizeof(int32_t) * P.ncols);
  P.off_ovf_count = o; o = align256(o + sizeof(int32_t));
  P.off_ent = o;       o = align256(o + sizeof(EllEntry) * P.ncols * kEllCap);
  P.off_ovf = o;       o = align256(o + sizeof(EllOverflow) * static_cast<size_t>(P.ovf_cap));
  P.total = o;
  return true;
}

// ------------------------------------------------------------------------------------ ELL index, tile-owner build
// The scatter build above asks a global counter for every tap of every sampling point (1.2 M returning atomics on
// [tap][pixel] counters at the layer2 size), then a second launch sorts the slots.  Here a workgroup OWNS one
// (image, kernel tap, 16 x 32 input tile): it scans the sampling points whose undisplaced position lies within
// kEllM pixels of the tile (learned offsets are small: the window catches all but the far tail), keeps the bilinear
// corners that land inside its tile in LDS lists, sorts every pixel's <= 8 entries and writes the counter and the
// pixel's whole 64-byte record ONCE — no global atomics with return, no clear, no sort launch, every sector of the
// table written in full.  Every corner is handled exactly once: by the workgroup of the tile it lands in when that
// workgroup's window covers the point (`ell_covers`, the same predicate on both sides), otherwise by the point's HOME
// workgroup (the tile of its undisplaced position, clamped into the map), which appends it to the overflow list the
// atomic kernel adds later.
// Measured (layer2 size, fp16, index build + gather + overflow): 146 -> 129 us.  The window scan costs 8 us and the
// record write 10 us; what is left is the slot requests — LDS atomics WITH RETURN are slow on this hardware (~15 cycles
// per requesting lane and CU; branch-free issue of a point's four requests changes nothing), so a wave compacts the
// corners that are its tile's own through ballots first and requests slots 64 lanes dense (141 -> 129 us).
constexpr int kEllTH = 16, kEllTW = 32;   // input tile (pixels); small maps use 8 rows
constexpr int kEllM = 8;                  // displacement margin scanned around a tile
constexpr int kEllThreads = 256;          // few, fat waves: the slot requests are per wave instruction
constexpr int kEllQueue = 128;            // records a wave queues before it requests slots for 64 of them

__device__ __forceinline__ bool ell_covers(int zy, int zx, int y0, int x0, int th) {
  return zy >= y0 - kEllM && zy < y0 + th + kEllM && zx >= x0 - kEllM && zx < x0 + kEllTW + kEllM;
}
__device__ __forceinline__ int ceil_div_signed(int a, int s) { return a >= 0 ? (a + s - 1) / s : -((-a) / s); }

template <typename T>
__global__ void __launch_bounds__(kEllThreads)
ell_build_kernel(const T* __restrict__ offset, const T* __restrict__ mask, Geom g, int th, int tiles_x,
                 int32_t* __restrict__ counter, EllEntry* __restrict__ ent, int32_t* __restrict__ ovf_count,
                 EllOverflow* __restrict__ ovf, int ovf_cap) {
  __shared__ int s_cnt[kEllTH * kEllTW];
  __shared__ int s_queue[(kEllThreads / 64) * 3 * kEllQueue];
  __shared__ EllEntry s_ent[kEllCap][kEllTH * kEllTW];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int th_log2 = th == 8 ? 3 : 4;
  const int ty = static_cast<int>(blockIdx.x) / tiles_x, tx = static_cast<int>(blockIdx.x) - ty * tiles_x;
  const int tap = blockIdx.y;
  const int img = blockIdx.z, b = img / g.dg, dgi = img - b * g.dg;
  const int y0 = ty * th, x0 = tx * kEllTW;
  const int npx = th * kEllTW;
  const int K = g.kh * g.kw, HW = g.H * g.W, HWo = g.Ho * g.Wo;
  for (int l = tid; l < npx; l += nthr) s_cnt[l] = 0;
  __syncthreads();
  const int ky = tap / g.kw, kx = tap - ky * g.kw;
  // output pixels whose undisplaced (clamped) position can fall inside the window: a superset, each is tested exactly
  const int by_ = g.pad_h - ky * g.dil_h, bx_ = g.pad_w - kx * g.dil_w;
  int oy_lo = (y0 - kEllM <= 0) ? 0 : max(0, ceil_div_signed(y0 - kEllM + by_, g.stride_h));
  int oy_hi = (y0 + th + kEllM >= g.H) ? g.Ho : min(g.Ho, ceil_div_signed(y0 + th + kEllM + by_, g.stride_h));
  int ox_lo = (x0 - kEllM <= 0) ? 0 : max(0, ceil_div_signed(x0 - kEllM + bx_, g.stride_w));
  int ox_hi = (x0 + kEllTW + kEllM >= g.W) ? g.Wo : min(g.Wo, ceil_div_signed(x0 + kEllTW + kEllM + bx_, g.stride_w));
  const int ncx = max(0, ox_hi - ox_lo), ncy = max(0, oy_hi - oy_lo);
  const size_t lbase = static_cast<size_t>(img) * HW;
  auto spill = [&](int tgt, int32_t colidx, float w) {
    const int o = atomicAdd(ovf_count, 1);
    if (o < ovf_cap) ovf[o] = EllOverflow{static_cast<int32_t>(lbase + tgt), colidx, w};
  };
  // Slot requests are LDS atomics with return: ~500 cycles per wave instruction on this hardware whatever the number of
  // active lanes.  So a wave first COMPACTS the corners that are this tile's own into its private queue (ballot + lane
  // prefix, no atomics) and requests slots 64 lanes dense: a third of the atomic instructions.
  const int lane = tid & 63;
  int* qbase = s_queue + (tid >> 6) * (3 * kEllQueue);           // this wave's queue: [l | column index | weight bits] x kEllQueue
  int qn = 0;                                                     // records queued (wave-uniform)
  auto drain = [&](int take) {                                    // the oldest `take` (<= 64) records request their slots
    DETOPS_WAVE_SYNC();
    if (lane < take) {
      const int l = qbase[lane], ci = qbase[kEllQueue + lane];
      const float w = __int_as_float(qbase[2 * kEllQueue + lane]);
      const int pos = atomicAdd(&s_cnt[l], 1);
      if (pos < kEllCap) s_ent[pos][l] = EllEntry{ci, w};
      else {
        const int py = y0 + l / kEllTW, px = x0 + (l & (kEllTW - 1));
        spill(py * g.W + px, ci, w);
      }
    }
    DETOPS_WAVE_SYNC();
    const int rest = qn - take;                                   // <= 64: move it to the front
    int m0 = 0, m1 = 0, m2 = 0;
    if (lane < rest) { m0 = qbase[take + lane]; m1 = qbase[kEllQueue + take + lane]; m2 = qbase[2 * kEllQueue + take + lane]; }
    DETOPS_WAVE_SYNC();
    if (lane < rest) { qbase[lane] = m0; qbase[kEllQueue + lane] = m1; qbase[2 * kEllQueue + lane] = m2; }
    qn = rest;
  };
  const int ncand = ncy * ncx;
  for (int i0 = 0; i0 < ncand; i0 += nthr) {                      // wave-uniform trip count (ballots inside)
    const int i = i0 + tid;
    const int r = i / max(ncx, 1);
    const int oy = oy_lo + r, ox = ox_lo + (i - r * ncx);
    const int zy = min(max(oy * g.stride_h - by_, 0), g.H - 1), zx = min(max(ox * g.stride_w - bx_, 0), g.W - 1);
    bool live = i < ncand && ell_covers(zy, zx, y0, x0, th);
    const bool home = zy >= y0 && zy < y0 + th && zx >= x0 && zx < x0 + kEllTW;
    Point q;
    q.b = b; q.tap = tap; q.ho = live ? oy : 0; q.wo = live ? ox : 0; q.pix = q.ho * g.Wo + q.wo;
    const Sample s = point_sample(q, g, dgi, offset);
    live = live && s.inside;
    float m = 1.f;
    if (mask) m = ld(mask + ((static_cast<size_t>(b) * g.dg + dgi) * K + tap) * HWo + q.pix);
    const int32_t colidx = tap * (g.B * HWo) + b * HWo + q.pix;
    const int tgt[4] = {s.i1, s.i2, s.i3, s.i4};
    const float wgt[4] = {s.w1 * m, s.w2 * m, s.w3 * m, s.w4 * m};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int cy = s.hl + (t >> 1), cx = s.wl + (t & 1);        // corner coordinates from the sample's floor
      const bool valid = live && tgt[t] >= 0;
      const bool mine = valid && cy >= y0 && cy < y0 + th && cx >= x0 && cx < x0 + kEllTW;
      const unsigned long long mk = __ballot(mine);
      if (mine) {
        const int at = qn + __popcll(mk & ((1ull << lane) - 1ull));
        qbase[at] = (cy - y0) * kEllTW + (cx - x0);
        qbase[kEllQueue + at] = colidx;
        qbase[2 * kEllQueue + at] = __float_as_int(wgt[t]);
      }
      qn += __popcll(mk);
      if (qn >= 64) drain(64);
      if (valid && !mine && home && !ell_covers(zy, zx, (cy >> th_log2) << th_log2, (cx / kEllTW) * kEllTW, th))
        spill(tgt[t], colidx, wgt[t]);                            // displaced beyond the window of the tile it lands in
    }
  }
  if (qn > 0) drain(qn);
  __syncthreads();
  // every pixel's entries by column index (same summation order every run), counters and slots out in full rows
  const size_t col0 = (static_cast<size_t>(img) * K + tap) * HW;
  EllEntry* ebase = ent + (static_cast<size_t>(img) * K + tap) * HW * kEllCap;   // [pixel][slot]: one 64-byte record per pixel
  for (int l = tid; l < npx; l += nthr) {
    const int py = y0 + l / kEllTW, px = x0 + (l & (kEllTW - 1));
    if (py >= g.H || px >= g.W) continue;
    const int cnt = s_cnt[l];
    const int n = min(cnt, kEllCap);
    int32_t k[kEllCap];
    float w[kEllCap];
#pragma unroll
    for (int j = 0; j < kEllCap; ++j) {
      const EllEntry e = j < n ? s_ent[j][l] : EllEntry{0x7fffffff, 0.f};
      k[j] = e.idx; w[j] = e.w;
    }
    if (n >= 2) {
#pragma unroll
      for (int r = 0; r < kEllCap; ++r) {
#pragma unroll
        for (int j = (r & 1); j + 1 < kEllCap; j += 2) {
          const bool sw = k[j] > k[j + 1];
          const int32_t ka = sw ? k[j + 1] : k[j], kb = sw ? k[j] : k[j + 1];
          const float wa = sw ? w[j + 1] : w[j], wb = sw ? w[j] : w[j + 1];
          k[j] = ka; k[j + 1] = kb; w[j] = wa; w[j + 1] = wb;
        }
      }
    }
    const int pix = py * g.W + px;
    counter[col0 + pix] = cnt;
    // the whole record, used slots or not: four 16-byte stores, every 64-byte sector of the table written in full
    float4* rec = reinterpret_cast<float4*>(ebase + static_cast<size_t>(pix) * kEllCap);
#pragma unroll
    for (int j = 0; j < kEllCap; j += 2)
      rec[j / 2] = make_float4(__int_as_float(k[j]), w[j], __int_as_float(k[j + 1]), w[j + 1]);
  }
}

// Builds counters + slots (+ the overflow list) of the inverted index for one call.  `dcn_ell_build` = 1 keeps the
// scatter build (A/B).
template <typename T>
int ell_build(const void* offset, const void* mask, const Geom& g, const EllPlan& P, int32_t* count, int32_t* ovf_count,
              EllEntry* ent, EllOverflow* ovf, size_t clear_bytes, hipStream_t st_) {
  if (detops_tuning().dcn_ell_build == 1) {
    DETOPS_HIP_TRY(hipMemsetAsync(count, 0, clear_bytes, st_));   // counters + overflow counter
    const dim3 pgrid(static_cast<unsigned>(ceil_div64(P.npoints_per_dg, kBlock)), static_cast<unsigned>(g.dg));
    hipLaunchKernelGGL(col2im_ell_fill_kernel<T>, pgrid, dim3(kBlock), 0, st_, static_cast<const T*>(offset),
                       static_cast<const T*>(mask), g, P.npoints_per_dg, count, ent, ovf_count, ovf, P.ovf_cap);
    hipLaunchKernelGGL(col2im_ell_sort_kernel, dim3(static_cast<unsigned>(ceil_div64(P.ncols, kBlock))), dim3(kBlock),
                       0, st_, static_cast<const int32_t*>(count), P.ncols, g.H * g.W, ent);
    return launch_status();
  }
  DETOPS_HIP_TRY(hipMemsetAsync(ovf_count, 0, sizeof(int32_t), st_));
  const int K = g.kh * g.kw;
  const int tiles_x = static_cast<int>(ceil_div64(g.W, kEllTW));
  int th = kEllTH;
  if (ceil_div64(g.H, th) * tiles_x * K * g.B * g.dg < 2 * kNumCU) th = 8;   // small maps: more, smaller workgroups
  const int nthr = kEllThreads;
  const int tiles_y = static_cast<int>(ceil_div64(g.H, th));
  if (K > 65535 || static_cast<int64_t>(g.B) * g.dg > 65535) return DETOPS_EUNSUPPORTED;
  hipLaunchKernelGGL(ell_build_kernel<T>, dim3(static_cast<unsigned>(tiles_x * tiles_y), static_cast<unsigned>(K),
                                               static_cast<unsigned>(g.B * g.dg)),
                     dim3(nthr), 0, st_, static_cast<const T*>(offset), static_cast<const T*>(mask), g, th, tiles_x,
                     count, ent, ovf_count, ovf, P.ovf_cap);
  return launch_status();
}

template <typename T>
int col2im_ell_t(const void* col, const void* offset, const void* mask, void* grad_im, const Geom& g,
                 const EllPlan& P, void* ws, hipStream_t st_) {
  if (P.npoints_per_dg == 0) return 0;
  char* w = static_cast<char*>(ws);
  int32_t* count = reinterpret_cast<int32_t*>(w + P.off_count);
  int32_t* ovf_count = reinterpret_cast<int32_t*>(w + P.off_ovf_count);
  EllEntry* ent = reinterpret_cast<EllEntry*>(w + P.off_ent);
  EllOverflow* ovf = reinterpret_cast<EllOverflow*>(w + P.off_ovf);
  { const int rc = ell_build<T>(offset, mask, g, P, count, ovf_count, ent, ovf, P.off_ent - P.off_count, st_); if (rc) return rc; }
  const int cpg = g.C / g.dg;
  const int64_t pix_blocks = ceil_div64(static_cast<int64_t>(g.H) * g.W, kBlock) * g.B;
  int cc = cpg;
  while (cc > kGatherCC && pix_blocks * g.dg * ceil_div64(cpg, cc) < 4 * kNumCU) cc = max(kGatherCC, cc / 2);
  cc = static_cast<int>(ceil_div64(cc, kGatherCC)) * kGatherCC;
  const int xcd_remap = detops_tuning().dcn_gather_xcd != 1;
  const dim3 ggrid(static_cast<unsigned>(ceil_div64(static_cast<int64_t>(g.H) * g.W, kBlock)),
                   static_cast<unsigned>(g.dg * ceil_div64(cpg, cc)), static_cast<unsigned>(g.B));
  hipLaunchKernelGGL(col2im_ell_gather_kernel<T>, ggrid, dim3(kBlock), 0, st_, static_cast<const T*>(col),
                     static_cast<const int32_t*>(count), static_cast<const EllEntry*>(ent), static_cast<T*>(grad_im), g, cc,
                     xcd_remap);
  hipLaunchKernelGGL(col2im_ell_overflow_kernel<T>, dim3(kNumCU), dim3(kBlock), 0, st_, static_cast<const T*>(col),
                     static_cast<const int32_t*>(ovf_count), static_cast<const EllOverflow*>(ovf), P.ovf_cap,
                     static_cast<T*>(grad_im), g);
  return launch_status();
}

// Workspace carve of the gather path (all offsets 256-byte aligned).
struct GatherPlan {
  int64_t nslots, npoints_per_dg, max_entries;
  size_t off_count, off_cursor, off_start, off_entries, off_scan, scan_bytes, total;
};

// false: the shape does not fit the 32-bit index plan (the scatter kernels handle it)
inline bool gather_plan(const Geom& g, GatherPlan& P) {
  const int64_t K = static_cast<int64_t>(g.kh) * g.kw;
  const int64_t HWo = static_cast<int64_t>(g.Ho) * g.Wo;
  P.nslots = static_cast<int64_t>(g.B) * g.dg * g.H * g.W * K;
  P.npoints_per_dg = static_cast<int64_t>(g.B) * K * HWo;
  P.max_entries = 4 * P.npoints_per_dg * g.dg;
  if (P.nslots + 1 > 0x7fffffff || P.max_entries > 0x7fffffff || K * g.B * HWo > 0x7fffffff) return false;
  P.scan_bytes = 0;
  if (!detops_exclusive_sum_i32_bytes(static_cast<int>(P.nslots + 1), &P.scan_bytes)) return false;
  size_t o = 0;
  P.off_count = o;   o = align256(o + sizeof(int32_t) * (P.nslots + 1));
  P.off_cursor = o;  o = align256(o + sizeof(int32_t) * P.nslots);
  P.off_start = o;   o = align256(o + sizeof(int32_t) * (P.nslots + 1));
  P.off_entries = o; o = align256(o + sizeof(ColEntry) * P.max_entries);
  P.off_scan = o;    o = align256(o + P.scan_bytes);
  P.total = o;
  return true;
}

template <typename T>
int col2im_gather_t(const void* col, const void* offset, const void* mask, void* grad_im, const Geom& g,
                    const GatherPlan& P, void* ws, hipStream_t st_) {
  if (P.npoints_per_dg == 0) return 0;
  char* w = static_cast<char*>(ws);
  int32_t* count = reinterpret_cast<int32_t*>(w + P.off_count);
  int32_t* cursor = reinterpret_cast<int32_t*>(w + P.off_cursor);
  int32_t* start = reinterpret_cast<int32_t*>(w + P.off_start);
  ColEntry* entries = reinterpret_cast<ColEntry*>(w + P.off_entries);
  // count and cursor are adjacent: one clear
  DETOPS_HIP_TRY(hipMemsetAsync(count, 0, P.off_start - P.off_count, st_));
  const dim3 pgrid(static_cast<unsigned>(ceil_div64(P.npoints_per_dg, kBlock)), static_cast<unsigned>(g.dg));
  hipLaunchKernelGGL((col2im_index_kernel<T, false>), pgrid, dim3(kBlock), 0, st_, static_cast<const T*>(offset),
                     static_cast<const T*>(mask), g, P.npoints_per_dg, count, static_cast<const int32_t*>(nullptr),
                     static_cast<ColEntry*>(nullptr));
  {
    const int rc = detops_exclusive_sum_i32(w + P.off_scan, P.scan_bytes, count, start, static_cast<int>(P.nslots + 1), st_);
    if (rc) return rc;
  }
  hipLaunchKernelGGL((col2im_index_kernel<T, true>), pgrid, dim3(kBlock), 0, st_, static_cast<const T*>(offset),
                     static_cast<const T*>(mask), g, P.npoints_per_dg, cursor, static_cast<const int32_t*>(start),
                     entries);
  hipLaunchKernelGGL(col2im_sort_kernel, dim3(static_cast<unsigned>(ceil_div64(P.nslots, kBlock))), dim3(kBlock), 0,
                     st_, static_cast<const int32_t*>(start), P.nslots, entries);
  const int cpg = g.C / g.dg;
  const int64_t pix_blocks = ceil_div64(static_cast<int64_t>(g.H) * g.W, kBlock) * g.B;
  int cc = cpg;  // channels per workgroup: whole passes of kGatherCC, enough workgroups to fill the chip
  while (cc > kGatherCC && pix_blocks * g.dg * ceil_div64(cpg, cc) < 4 * kNumCU) cc = max(kGatherCC, cc / 2);
  cc = static_cast<int>(ceil_div64(cc, kGatherCC)) * kGatherCC;
  const int xcd_remap = detops_tuning().dcn_gather_xcd != 1;   // 1: plain block order (A/B measurements)
  const dim3 ggrid(static_cast<unsigned>(ceil_div64(static_cast<int64_t>(g.H) * g.W, kBlock)),
                   static_cast<unsigned>(g.dg * ceil_div64(cpg, cc)), static_cast<unsigned>(g.B));
  hipLaunchKernelGGL(col2im_gather_kernel<T>, ggrid, dim3(kBlock), 0, st_, static_cast<const T*>(col),
                     static_cast<const int32_t*>(start), static_cast<const ColEntry*>(entries),
                     static_cast<T*>(grad_im), g, cc, xcd_remap);
  return launch_status();
}

template <typename T>
int coord_t(const void* col, const void* im, const void* offset, const void* mask, void* goff,
            void* gmask, const Geom& g, hipStream_t st_) {
  const int64_t np = static_cast<int64_t>(g.B) * g.kh * g.kw * g.Ho * g.Wo;
  if (np == 0) return 0;
  // channel slices per point: enough workgroups for the chip on small maps (C5: 2 x 9 x 25 x 42 points)
  const int cpg = g.C / g.dg;
  int S = 1;
  while (S < 16 && cpg / (S * 4) >= 8 && ceil_div64(np, kBlock / S) * g.dg < 4 * kNumCU) S *= 4;
#define COORD_LAUNCH(S_)                                                                                     \
  hipLaunchKernelGGL((col2im_coord_kernel<T, S_>),                                                           \
                     dim3(static_cast<unsigned>(ceil_div64(np, kBlock / S_)), static_cast<unsigned>(g.dg)),  \
                     dim3(kBlock), 0, st_, static_cast<const T*>(col), static_cast<const T*>(im),            \
                     static_cast<const T*>(offset), static_cast<const T*>(mask), static_cast<T*>(goff),      \
                     static_cast<T*>(gmask), g, np)
  if (S == 1) COORD_LAUNCH(1);
  else if (S == 4) COORD_LAUNCH(4);
  else COORD_LAUNCH(16);
#undef COORD_LAUNCH
  return launch_status();
}

// ------------------------------------------------------------------------------------ fused forward (MFMA)
// Deformable convolution forward as ONE implicit GEMM on the matrix cores (fp16 / bf16):
//     out[co, n] = sum_{tap, c} W[co, c, tap] * column(c, tap, n),      n = (b, ho, wo)
// The reference (and the unfused path above) materialises `columns` [C*kh*kw, B*Ho*Wo] in HBM — 77 MB at the
// layer2 shape of cfg-5 — and hands it to a library GEMM (csrc/cuda/deform_conv_cuda.cu:228-245).  Here the
// deformed B-operand tile is built in LDS and consumed by v_mfma_f32_32x32x16_{f16,bf16} directly:
//   * pre-pass (one launch, small): the input goes NCHW -> NHWC so that the 4 bilinear taps of a sampling
//     point read 32 CONTIGUOUS bytes per 16 channels, and the weights go [Cout, C, taps] -> [taps, Cout, C] so
//     that an A tile of one tap is contiguous in its K dimension (channels);
//   * main kernel: workgroup tile 128 (Cout) x 64 (pixels), K-step = up to 128 channels of one tap.  The
//     channels of a sampling point are contiguous in the NHWC copy: 16 adjacent lanes read 256 bytes of one
//     corner, so a wave-wide gather touches a few full cache lines.  Each thread interpolates 8 channels of its
//     pixels in fp32, applies the modulation mask, rounds to the storage type (what the im2col kernel would have
//     stored) and writes 16 bytes of the B tile.  Four waves of 1 x 2 MFMA tiles (32 x 64 outputs each, fp32
//     accumulators) consume A and B from LDS; rows are 16 bytes longer than the K-step, which makes the
//     ds_read_b128 fragment reads conflict-free.  The next K-step's gathers and weight tile are issued before
//     the MFMAs of the current one (register prefetch; LDS-only barriers keep them in flight).
// Requirements (else the caller uses the unfused path): 16-bit storage, conv groups == 1, (C / dg) % 32 == 0.
typedef _Float16 dcn_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 dcn_b8 __attribute__((ext_vector_type(8)));
typedef float dcn_f16v __attribute__((ext_vector_type(16)));
typedef unsigned short dcn_u16;

constexpr int kFM = 128, kFN = 64;   // workgroup tile: Cout x pixels
constexpr int kFK = 32;              // channel granularity of the fused plan ((C / dg) % 32 == 0)

#define DCN_MFMA_F16(a, b, c) DETOPS_MFMA_32x32x16_F16(a, b, c)
#define DCN_MFMA_BF16(a, b, c) DETOPS_MFMA_32x32x16_BF16(a, b, c)

// pre-pass: blocks [0, nb_im): input NCHW -> NHWC (32 channels x 64 pixels per block through LDS);
//           the rest: weights [Cout, C, T] -> [T, Cout, C]
__global__ void __launch_bounds__(kBlock)
dcn_fused_prep_kernel(const dcn_u16* __restrict__ im, const dcn_u16* __restrict__ w, dcn_u16* __restrict__ imT,
                      dcn_u16* __restrict__ wt, int B, int C, int HW, int Cout, int T, int nb_im) {
  __shared__ dcn_u16 tile[32][66];
  const int tid = threadIdx.x;
  if (static_cast<int>(blockIdx.x) < nb_im) {
    const int pblocks = (HW + 63) / 64, cblocks = (C + 31) / 32;
    int r = blockIdx.x;
    const int pb = r % pblocks; r /= pblocks;
    const int cb = r % cblocks;
    const int b = r / cblocks;
    const int p0 = pb * 64, c0 = cb * 32;
    for (int e = tid; e < 32 * 64; e += kBlock) {   // read: pixels fastest
      const int c = e >> 6, pp = e & 63;
      tile[c][pp] = (c0 + c < C && p0 + pp < HW) ? im[(static_cast<size_t>(b) * C + c0 + c) * HW + p0 + pp] : dcn_u16(0);
    }
    __syncthreads();
    for (int e = tid; e < 32 * 64; e += kBlock) {   // write: channels fastest
      const int pp = e >> 5, c = e & 31;
      if (c0 + c < C && p0 + pp < HW) imT[(static_cast<size_t>(b) * HW + p0 + pp) * C + c0 + c] = tile[c][pp];
    }
    return;
  }
  const int64_t n = static_cast<int64_t>(Cout) * C * T;
  for (int64_t e = static_cast<int64_t>(blockIdx.x - nb_im) * kBlock + tid; e < n;
       e += static_cast<int64_t>(gridDim.x - nb_im) * kBlock) {
    const int c = static_cast<int>(e % C);      // destination order [t][co][c]: coalesced writes
    const int64_t r = e / C;
    const int co = static_cast<int>(r % Cout);
    const int t = static_cast<int>(r / Cout);
    wt[e] = w[(static_cast<size_t>(co) * C + c) * T + t];
  }
}

template <bool BF16> struct DcnConv;
template <> struct DcnConv<false> {
  static __device__ __forceinline__ float to_f(dcn_u16 v) { return __half2float(__ushort_as_half(v)); }
  static __device__ __forceinline__ dcn_u16 from_f(float f) { return __half_as_ushort(__float2half(f)); }
};
template <> struct DcnConv<true> {
  static __device__ __forceinline__ float to_f(dcn_u16 v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }
  static __device__ __forceinline__ dcn_u16 from_f(float f) {
    const __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<const dcn_u16*>(&h);
  }
};

// 8 channels of one tap (16 bytes) as a register vector (a struct of 16-bit fields is not promoted to registers:
// the first version kept these staging arrays in scratch memory with a vmcnt(0) after every load)
typedef unsigned int DcnRaw8 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dcn_u16 dcn_elem(const DcnRaw8& r, int e) { return static_cast<dcn_u16>((r[e >> 1] >> ((e & 1) * 16)) & 0xffffu); }

// BKC = channels of one tap per K-step (32 / 64 / 128).  Gather mapping: the BKC channels of a sampling
// point are 2 * BKC CONTIGUOUS bytes of the NHWC copy, read by LPP = BKC / 8 adjacent lanes (16 bytes each) —
// a wave-wide load touches 64 / LPP pixels x full cache lines instead of 64 different lines (the
// one-pixel-per-lane mapping of the first version kept the texture addresser busy for 64 line lookups per
// instruction: 174 us at the layer2 shape).  256 / LPP pixels per pass, kFN / (256 / LPP) passes per step.
template <bool BF16, typename T, int BKC>
__global__ void __launch_bounds__(kBlock)
dcn_fused_fwd_kernel(const dcn_u16* __restrict__ imT, const dcn_u16* __restrict__ wt, const T* __restrict__ offset,
                     const T* __restrict__ mask, const T* __restrict__ bias, T* __restrict__ out, Geom g, int Cout) {
  using CV = DcnConv<BF16>;
  constexpr int LPP = BKC / 8;                 // lanes per pixel
  constexpr int PPP = kBlock / LPP;            // pixels per pass
  constexpr int NP = kFN / PPP;                // passes per step (1, 2 or 4)
  constexpr int LD = BKC + 8;                  // LDS row stride in 16-bit elements: + 16 bytes -> conflict-free b128 reads
  constexpr int APT = kFM * (BKC / 8) / kBlock;   // 16-byte weight pieces per thread and step
  DETOPS_DYNAMIC_LDS(DcnRaw8, lds8);
  dcn_u16* sA = reinterpret_cast<dcn_u16*>(lds8);                 // [kFM][LD]
  dcn_u16* sB = sA + kFM * LD;                                    // [kFN][LD]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int co0 = blockIdx.y * kFM;
  const int64_t npix = static_cast<int64_t>(g.B) * g.Ho * g.Wo;
  const int64_t pix0 = static_cast<int64_t>(blockIdx.x) * kFN;
  const int K = g.kh * g.kw, cpg = g.C / g.dg;
  const size_t HWo = static_cast<size_t>(g.Ho) * g.Wo;

  // ---- B-tile builder role: channel octet cq of pixel (pass * PPP + pg)
  const int cq = tid % LPP, pg = tid / LPP;
  int pb[NP], pho[NP], pwo[NP];
  bool pok[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int64_t n = pix0 + ps * PPP + pg;
    pok[ps] = n < npix;
    pb[ps] = 0; pho[ps] = 0; pwo[ps] = 0;
    if (pok[ps]) {
      pb[ps] = static_cast<int>(n / static_cast<int64_t>(HWo));
      const int r = static_cast<int>(n - static_cast<int64_t>(pb[ps]) * HWo);
      pho[ps] = r / g.Wo; pwo[ps] = r - pho[ps] * g.Wo;
    }
  }

  dcn_f16v acc[2];   // wave w: output rows 32 w .. 32 w + 31, two 32-column tiles
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int steps_per_tap = cpg / BKC;
  const int nsteps = g.dg * K * steps_per_tap;
  // staged data of the NEXT step
  DcnRaw8 raw[NP][4];
  DcnRaw8 araw[APT];
  float wq[NP][4], mq[NP];   // bilinear weights (0 for a tap outside the map) and modulation of the staged step

  auto issue = [&](int step) {   // global -> registers for K-step `step`
    const int dgi = step / (K * steps_per_tap);
    const int rem = step - dgi * K * steps_per_tap;
    const int tap = rem / steps_per_tap;
    const int c0 = dgi * cpg + (rem - tap * steps_per_tap) * BKC;
    const int ti = tap / g.kw, tj = tap - ti * g.kw;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      float off_h = 0.f, off_w = 0.f;
      mq[ps] = 1.f;
      if (pok[ps]) {
        const T* op = offset + (static_cast<size_t>(pb[ps]) * g.dg + dgi) * 2 * K * HWo + static_cast<size_t>(pho[ps]) * g.Wo + pwo[ps];
        off_h = ld(op + (2 * tap) * HWo);
        off_w = ld(op + (2 * tap + 1) * HWo);
        if (mask) mq[ps] = ld(mask + ((static_cast<size_t>(pb[ps]) * g.dg + dgi) * K + tap) * HWo + static_cast<size_t>(pho[ps]) * g.Wo + pwo[ps]);
      }
      const float h_im = static_cast<float>(pho[ps] * g.stride_h - g.pad_h + ti * g.dil_h) + off_h;
      const float w_im = static_cast<float>(pwo[ps] * g.stride_w - g.pad_w + tj * g.dil_w) + off_w;
      const Sample sm = make_sample(h_im, w_im, g.H, g.W);
      const int idx[4] = {sm.i1, sm.i2, sm.i3, sm.i4};
      const float ws[4] = {sm.w1, sm.w2, sm.w3, sm.w4};
      const dcn_u16* imb = imT + static_cast<size_t>(pb[ps]) * g.H * g.W * g.C + c0 + 8 * cq;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = pok[ps] && sm.inside && idx[k] >= 0;
        wq[ps][k] = ok ? ws[k] : 0.f;
        // a tap outside the map reads element 0 of the image (always valid) with weight 0
        raw[ps][k] = *reinterpret_cast<const DcnRaw8*>(imb + static_cast<size_t>(ok ? idx[k] : 0) * g.C);
      }
    }
    const dcn_u16* wbase = wt + (static_cast<size_t>(tap) * Cout) * g.C + c0;
#pragma unroll
    for (int q = 0; q < APT; ++q) {
      const int piece = tid + q * kBlock;
      const int row = piece / (BKC / 8), seg = piece - row * (BKC / 8);
      if (co0 + row < Cout) araw[q] = *reinterpret_cast<const DcnRaw8*>(wbase + static_cast<size_t>(co0 + row) * g.C + 8 * seg);
      else araw[q] = DcnRaw8{0u, 0u, 0u, 0u};
    }
  };
  auto commit = [&]() {   // registers -> LDS: interpolate the B tile, copy the A tile
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      DcnRaw8 o = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // same expression as the im2col kernel: w1*a1 + w2*a2 + w3*a3 + w4*a4 (taps outside contribute 0)
        const float v = wq[ps][0] * CV::to_f(dcn_elem(raw[ps][0], e)) + wq[ps][1] * CV::to_f(dcn_elem(raw[ps][1], e)) +
                        wq[ps][2] * CV::to_f(dcn_elem(raw[ps][2], e)) + wq[ps][3] * CV::to_f(dcn_elem(raw[ps][3], e));
        o[e >> 1] |= static_cast<unsigned>(CV::from_f(mask ? v * mq[ps] : v)) << ((e & 1) * 16);
      }
      *reinterpret_cast<DcnRaw8*>(sB + (ps * PPP + pg) * LD + 8 * cq) = o;
    }
#pragma unroll
    for (int q = 0; q < APT; ++q) {
      const int piece = tid + q * kBlock;
      const int row = piece / (BKC / 8), seg = piece - row * (BKC / 8);
      *reinterpret_cast<DcnRaw8*>(sA + row * LD + 8 * seg) = araw[q];
    }
  };

  issue(0);
  for (int step = 0; step < nsteps; ++step) {
    commit();
    DETOPS_LDS_BARRIER();   // LDS-only: the prefetch below stays in flight across the barriers
    if (step + 1 < nsteps) issue(step + 1);
#pragma unroll
    for (int ks = 0; ks < BKC / 16; ++ks) {
      const int koff = ks * 16 + (lane >> 5) * 8;
      if constexpr (BF16) {
        const dcn_b8 a = *reinterpret_cast<const dcn_b8*>(sA + (wave * 32 + (lane & 31)) * LD + koff);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const dcn_b8 b = *reinterpret_cast<const dcn_b8*>(sB + (j * 32 + (lane & 31)) * LD + koff);
          acc[j] = DCN_MFMA_BF16(a, b, acc[j]);
        }
      } else {
        const dcn_h8 a = *reinterpret_cast<const dcn_h8*>(sA + (wave * 32 + (lane & 31)) * LD + koff);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const dcn_h8 b = *reinterpret_cast<const dcn_h8*>(sB + (j * 32 + (lane & 31)) * LD + koff);
          acc[j] = DCN_MFMA_F16(a, b, acc[j]);
        }
      }
    }
    DETOPS_LDS_BARRIER();   // the next commit rewrites both tiles
  }

  // ---- epilogue: C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t n = pix0 + j * 32 + (lane & 31);
    if (n >= npix) continue;
    const int b = static_cast<int>(n / static_cast<int64_t>(HWo));
    const size_t pix = static_cast<size_t>(n - static_cast<int64_t>(b) * HWo);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (co < Cout) {
        float v = acc[j][r];
        if (bias) v += ld(bias + co);
        st(out + (static_cast<size_t>(b) * Cout + co) * HWo + pix, v);
      }
    }
  }
}

inline size_t dcn_fused_ws_bytes(const Geom& g, int Cout) {
  const size_t im = (static_cast<size_t>(g.B) * g.C * g.H * g.W * 2 + 255) & ~static_cast<size_t>(255);
  const size_t w = (static_cast<size_t>(Cout) * g.C * g.kh * g.kw * 2 + 255) & ~static_cast<size_t>(255);
  return im + w;
}

inline bool dcn_fused_ok(const Geom& g, int dtype, int Cout) {
  return (dtype == DETOPS_F16 || dtype == DETOPS_BF16) && Cout > 0 && (g.C / g.dg) % kFK == 0 &&
         static_cast<int64_t>(g.B) * g.Ho * g.Wo < (1ll << 31) && static_cast<int64_t>(g.H) * g.W < (1ll << 30);
}

// Is the fused kernel the faster path?  Its grid is (pixels / 64) x (Cout / 128) workgroups and every Cout tile
// rebuilds the deformed operand, so it wins where pixels are many and Cout tiles few (measured,
// profiles/r02f_opbench_dcn_fused.log, fused vs im2col + library GEMM: 105 vs 179 us at [2,128,100,168] = 94 TF/s;
// 108 vs 103 us at [2,256,50,84]; 164 vs 72 us at [2,512,25,42], where 132 workgroups leave half the chip idle).
// Tuning dcn_fused = 1 overrides (tests, measurements).
inline bool dcn_fused_preferred(const Geom& g, int Cout) {
  if (detops_tuning().dcn_fused == 1) return true;
  const int64_t wgs = ceil_div64(static_cast<int64_t>(g.B) * g.Ho * g.Wo, kFN) * ceil_div64(Cout, kFM);
  return wgs >= 2 * kNumCU && Cout <= 2 * kFM;
}

template <bool BF16, typename T>
int dcn_fused_forward(const void* im, const void* weight, const void* offset, const void* mask, const void* bias,
                      void* out, const Geom& g, int Cout, void* ws, hipStream_t st_) {
  unsigned char* base = static_cast<unsigned char*>(ws);
  const size_t im_bytes = (static_cast<size_t>(g.B) * g.C * g.H * g.W * 2 + 255) & ~static_cast<size_t>(255);
  dcn_u16* imT = reinterpret_cast<dcn_u16*>(base);
  dcn_u16* wt = reinterpret_cast<dcn_u16*>(base + im_bytes);
  const int HW = g.H * g.W, T_ = g.kh * g.kw;
  const int nb_im = g.B * ((g.C + 31) / 32) * ((HW + 63) / 64);
  const int nb_w = static_cast<int>(std::min<int64_t>(ceil_div64(static_cast<int64_t>(Cout) * g.C * T_, kBlock), 4 * kNumCU));
  hipLaunchKernelGGL(dcn_fused_prep_kernel, dim3(static_cast<unsigned>(nb_im + nb_w)), dim3(kBlock), 0, st_,
                     static_cast<const dcn_u16*>(im), static_cast<const dcn_u16*>(weight), imT, wt, g.B, g.C, HW, Cout, T_,
                     nb_im);
  const int64_t npix = static_cast<int64_t>(g.B) * g.Ho * g.Wo;
  const dim3 grid(static_cast<unsigned>(ceil_div64(npix, kFN)), static_cast<unsigned>(ceil_div64(Cout, kFM)));
  const int cpg = g.C / g.dg;
#define DCN_FUSED_LAUNCH(BKC_)                                                                                        \
  hipLaunchKernelGGL((dcn_fused_fwd_kernel<BF16, T, BKC_>), grid, dim3(kBlock),                                       \
                     static_cast<size_t>(kFM + kFN) * (BKC_ + 8) * sizeof(dcn_u16), st_, imT, wt,                     \
                     static_cast<const T*>(offset), static_cast<const T*>(mask), static_cast<const T*>(bias),         \
                     static_cast<T*>(out), g, Cout)
  if (cpg % 128 == 0) DCN_FUSED_LAUNCH(128);
  else if (cpg % 64 == 0) DCN_FUSED_LAUNCH(64);
  else DCN_FUSED_LAUNCH(32);
#undef DCN_FUSED_LAUNCH
  return launch_status();
}


// ------------------------------------------------------------------------------------ channels-last pipeline
// The reference's column matrix is [C * kh * kw, B * Ho * Wo] with the PIXEL index fastest, so one sampling point's
// channels are B * Ho * Wo elements apart: the im2col store, the col2im gather and the coordinate-gradient reduction
// all move 2-byte (fp16) elements with a lane stride of a whole row — 1.8 .. 8 % of the HBM peak at the cfg-5 shapes
// (profiles/r02h_opbench.log).  The kernels below keep every per-sampling-point operand CHANNEL-fastest instead:
//     xT    [B, H * W, C]                 input, NHWC                       (nchw_to_nhwc_kernel)
//     colT  [B * Ho * Wo, kh * kw, C]     deformed columns                   (im2col_nhwc_kernel)
//     gT    [B, Ho * Wo, Cout]            output gradient, NHWC
//     S_T   [B * H * W, kh * kw, Cout]    "transposed sampling" of gT        (sampleT_gather_kernel)
// A group of SUB = min(C / V, 64) adjacent lanes owns one pixel and moves its channels as 16-byte vectors (V = 4 fp32 or
// 8 half elements), so the four bilinear corners of a sampling point are four contiguous runs.  The GEMMs around them
// are plain library GEMMs on these layouts (the host side: maskrcnn_benchmark/_C.py):
//     forward        out[b]        = W2 [Cout, K C]  x colT[b]^T
//     weight grad    dW2           = gT^T [Cout, B HW] x colT [B HW, K C]
//     column grad    colsG_T       = gT [B HW, Cout] x W2 [Cout, K C]           -> coord_nhwc_kernel (offset / mask gradients)
//     input grad     grad_in[b]    = W2T [C, K Cout] x S_T[b]^T
// The input gradient uses the TRANSPOSED sampling operator instead of the reference's scatter (col2im): pixel p of
// the gradient map gathers, per tap, the output-gradient vectors of the sampling points that touch it (the same
// fixed-width inverted index as col2im_ell above), and the channel mixing W^T is applied afterwards by the GEMM —
// the gather moves contiguous Cout-vectors instead of single column elements.  Requires deformable_group == 1, conv
// groups == 1 and C / V, Cout / V powers of two in [16, 256] (every model shape); anything else stays on the
// reference-layout kernels above.
template <typename T> struct VecT;
template <> struct VecT<float> { static constexpr int N = 4; };
template <> struct VecT<__half> { static constexpr int N = 8; };
template <> struct VecT<__hip_bfloat16> { static constexpr int N = 8; };

typedef unsigned int RawV4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ void vec_load(const T* p, float* v);
template <> __device__ __forceinline__ void vec_load<float>(const float* p, float* v) {
  const float4 r = *reinterpret_cast<const float4*>(p);
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
template <> __device__ __forceinline__ void vec_load<__half>(const __half* p, float* v) {
  const RawV4 r = *reinterpret_cast<const RawV4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __half2float(__ushort_as_half(static_cast<unsigned short>(r[i] & 0xffffu)));
    v[2 * i + 1] = __half2float(__ushort_as_half(static_cast<unsigned short>(r[i] >> 16)));
  }
}
template <> __device__ __forceinline__ void vec_load<__hip_bfloat16>(const __hip_bfloat16* p, float* v) {
  const RawV4 r = *reinterpret_cast<const RawV4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r[i] << 16);
    v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ void vec_store(T* p, const float* v);
template <> __device__ __forceinline__ void vec_store<float>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void vec_store<__half>(__half* p, const float* v) {
  RawV4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = static_cast<unsigned>(__half_as_ushort(__float2half(v[2 * i]))) |
           (static_cast<unsigned>(__half_as_ushort(__float2half(v[2 * i + 1]))) << 16);
  *reinterpret_cast<RawV4*>(p) = r;
}
template <> __device__ __forceinline__ void vec_store<__hip_bfloat16>(__hip_bfloat16* p, const float* v) {
  RawV4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __hip_bfloat16 a = __float2bfloat16(v[2 * i]), b = __float2bfloat16(v[2 * i + 1]);
    r[i] = static_cast<unsigned>(*reinterpret_cast<const unsigned short*>(&a)) |
           (static_cast<unsigned>(*reinterpret_cast<const unsigned short*>(&b)) << 16);
  }
  *reinterpret_cast<RawV4*>(p) = r;
}

// [B, C, HW] -> [B, HW, C] through a 32-channel x 64-pixel LDS tile (element size 2 or 4 bytes)
template <typename U>
__global__ void __launch_bounds__(kBlock)
nchw_to_nhwc_kernel(const U* __restrict__ in, U* __restrict__ out, int C, int HW) {
  __shared__ U tile[32][64 + 4 / sizeof(U)];
  const int tid = threadIdx.x;
  const int pblocks = (HW + 63) / 64, cblocks = (C + 31) / 32;
  int r = blockIdx.x;
  const int pb = r % pblocks; r /= pblocks;
  const int cb = r % cblocks;
  const int b = r / cblocks;
  const int p0 = pb * 64, c0 = cb * 32;
  for (int e = tid; e < 32 * 64; e += kBlock) {   // read: pixels fastest
    const int c = e >> 6, pp = e & 63;
    tile[c][pp] = (c0 + c < C && p0 + pp < HW) ? in[(static_cast<size_t>(b) * C + c0 + c) * HW + p0 + pp] : U(0);
  }
  __syncthreads();
  for (int e = tid; e < 32 * 64; e += kBlock) {   // write: channels fastest
    const int pp = e >> 5, c = e & 31;
    if (c0 + c < C && p0 + pp < HW) out[(static_cast<size_t>(b) * HW + p0 + pp) * C + c0 + c] = tile[c][pp];
  }
}

// lanes per pixel / vectors per lane of a channel count (CV = channels / V, a power of two in [16, 256])
struct NhwcMap { int sub, nv, ppb; };   // ppb = pixels per 256-thread block
inline bool nhwc_map(int channels, int V, NhwcMap& m) {
  if (channels % V) return false;
  const int cv = channels / V;
  if (cv < 16 || cv > 256 || (cv & (cv - 1))) return false;
  m.sub = cv < 64 ? cv : 64;
  m.nv = cv / m.sub;
  m.ppb = kBlock / m.sub;
  return true;
}

// the sampling point (b, tap, ho, wo) of deformable group 0
template <typename T>
__device__ __forceinline__ Sample tap_sample(const Geom& g, int b, int tap, int ho, int wo, int pix,
                                             const T* __restrict__ offset, const T* __restrict__ mask, float& m) {
  const int K = g.kh * g.kw;
  const int i = tap / g.kw, j = tap - i * g.kw;
  const size_t HWo = static_cast<size_t>(g.Ho) * g.Wo;
  const T* op = offset + static_cast<size_t>(b) * 2 * K * HWo;
  const float off_h = ld(op + (2 * tap) * HWo + pix), off_w = ld(op + (2 * tap + 1) * HWo + pix);
  m = mask ? ld(mask + (static_cast<size_t>(b) * K + tap) * HWo + pix) : 1.f;
  return make_sample(static_cast<float>(ho * g.stride_h - g.pad_h + i * g.dil_h) + off_h,
                     static_cast<float>(wo * g.stride_w - g.pad_w + j * g.dil_w) + off_w, g.H, g.W);
}

// colT[q, tap, c] = mask * bilinear(xT[b, :, c]; sampling point (q, tap)), q = b * Ho * Wo + pix
template <typename T>
__global__ void __launch_bounds__(kBlock)
im2col_nhwc_kernel(const T* __restrict__ xT, const T* __restrict__ offset, const T* __restrict__ mask,
                   T* __restrict__ colT, Geom g, int sub, int nv, int64_t npix) {
  constexpr int V = VecT<T>::N;
  const int ls = threadIdx.x % sub;
  const int64_t q = static_cast<int64_t>(blockIdx.x) * (kBlock / sub) + threadIdx.x / sub;
  if (q >= npix) return;
  const int HWo = g.Ho * g.Wo, K = g.kh * g.kw;
  const int b = static_cast<int>(q / HWo), pix = static_cast<int>(q - static_cast<int64_t>(b) * HWo);
  const int ho = pix / g.Wo, wo = pix - ho * g.Wo;
  const T* xb = xT + static_cast<size_t>(b) * g.H * g.W * g.C;
  for (int tap = 0; tap < K; ++tap) {
    float m;
    const Sample s = tap_sample(g, b, tap, ho, wo, pix, offset, mask, m);
    const int idx[4] = {s.i1, s.i2, s.i3, s.i4};
    const float wgt[4] = {s.w1 * m, s.w2 * m, s.w3 * m, s.w4 * m};
    T* dst = colT + (static_cast<size_t>(q) * K + tap) * g.C;
    for (int k = 0; k < nv; ++k) {
      const int c = (k * sub + ls) * V;
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (idx[t] >= 0) {     // wave-uniform per pixel group only: exec-masked
          float v[V];
          vec_load(xb + static_cast<size_t>(idx[t]) * g.C + c, v);
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] = fmaf(wgt[t], v[e], acc[e]);
        }
      }
      vec_store(dst + c, acc);
    }
  }
}

// offset / mask gradients from the column gradient colsG_T [q, tap, c] (channel-fastest) and the NHWC input:
// get_coordinate_weight / mask gradient of deform_conv_kernel_cuda.cu:152-195, :738-772, summed over the channels
// by the lanes of a pixel group (butterfly over `sub` adjacent lanes) — the same sums in a different order
template <typename T>
__global__ void __launch_bounds__(kBlock)
coord_nhwc_kernel(const T* __restrict__ colsG, const T* __restrict__ xT, const T* __restrict__ offset,
                  const T* __restrict__ mask, T* __restrict__ grad_offset, T* __restrict__ grad_mask, Geom g,
                  int sub, int nv, int64_t npix) {
  constexpr int V = VecT<T>::N;
  const int ls = threadIdx.x % sub;
  const int64_t q0 = static_cast<int64_t>(blockIdx.x) * (kBlock / sub) + threadIdx.x / sub;
  const bool live = q0 < npix;                 // dead groups stay for the wave-wide shuffles
  const int64_t q = live ? q0 : npix - 1;
  const int HWo = g.Ho * g.Wo, K = g.kh * g.kw;
  const int b = static_cast<int>(q / HWo), pix = static_cast<int>(q - static_cast<int64_t>(b) * HWo);
  const int ho = pix / g.Wo, wo = pix - ho * g.Wo;
  const T* xb = xT + static_cast<size_t>(b) * g.H * g.W * g.C;
  const int lane = threadIdx.x & (kWave - 1);
  for (int tap = 0; tap < K; ++tap) {
    float m;
    const Sample s = tap_sample(g, b, tap, ho, wo, pix, offset, mask, m);
    const int idx[4] = {s.i1, s.i2, s.i3, s.i4};
    const float hw = 1.f - s.lw, hh = 1.f - s.lh;
    const float ch[4] = {-hw, -s.lw, hw, s.lw};          // d/dh weights of the four corners
    const float cw[4] = {-hh, hh, -s.lh, s.lh};          // d/dw
    const float cm[4] = {s.w1, s.w2, s.w3, s.w4};
    float gh = 0.f, gw = 0.f, gm = 0.f;
    if (s.inside) {
      const T* gp = colsG + (static_cast<size_t>(q) * K + tap) * g.C;
      for (int k = 0; k < nv; ++k) {
        const int c = (k * sub + ls) * V;
        float gv[V];
        vec_load(gp + c, gv);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (idx[t] >= 0) {
            float v[V];
            vec_load(xb + static_cast<size_t>(idx[t]) * g.C + c, v);
            float dot = 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) dot = fmaf(gv[e], v[e], dot);
            gh = fmaf(ch[t], dot, gh); gw = fmaf(cw[t], dot, gw); gm = fmaf(cm[t], dot, gm);
          }
        }
      }
    }
    for (int o = sub >> 1; o >= 1; o >>= 1) {
      gh += __shfl(gh, lane ^ o); gw += __shfl(gw, lane ^ o); gm += __shfl(gm, lane ^ o);
    }
    if (live && ls == 0) {
      const size_t HWo_ = static_cast<size_t>(HWo);
      T* gop = grad_offset + static_cast<size_t>(b) * 2 * K * HWo_;
      st(gop + (2 * tap) * HWo_ + pix, gh * m);
      st(gop + (2 * tap + 1) * HWo_ + pix, gw * m);
      if (mask && grad_mask) st(grad_mask + (static_cast<size_t>(b) * K + tap) * HWo_ + pix, gm);
    }
  }
}

// S_T[p, tap, co] = sum over the sampling points (q, tap) whose bilinear footprint contains gradient-map pixel p of
// weight * gT[q, co]  (weight = bilinear corner weight x modulation mask, from the fixed-width inverted index)
template <typename T>
__global__ void __launch_bounds__(kBlock)
sampleT_gather_kernel(const T* __restrict__ gT, const int32_t* __restrict__ counter, const EllEntry* __restrict__ ent,
                      T* __restrict__ S_T, Geom g, int Cout, int sub, int nv, int64_t npix) {
  constexpr int V = VecT<T>::N;
  const int ls = threadIdx.x % sub;
  const int64_t gp = static_cast<int64_t>(blockIdx.x) * (kBlock / sub) + threadIdx.x / sub;   // b * H * W + pixel
  if (gp >= npix) return;
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo, K = g.kh * g.kw;
  const int b = static_cast<int>(gp / HW), p = static_cast<int>(gp - static_cast<int64_t>(b) * HW);
  for (int tap = 0; tap < K; ++tap) {
    const size_t col = (static_cast<size_t>(b) * K + tap) * HW + p;             // counter: [img][tap][pixel]
    const int n = min(counter[col], kEllCap);
    const EllEntry* ee = ent + ((static_cast<size_t>(b) * K + tap) * HW + p) * kEllCap;
    int32_t qi[kEllCap];
    float wi[kEllCap];
    // the pixel's whole 64-byte record in four 16-byte loads (same addresses across the pixel group); slots >= n may
    // hold anything (the scatter build leaves them unwritten): they are replaced before use
    const float4* rec = reinterpret_cast<const float4*>(ee);
#pragma unroll
    for (int j = 0; j < kEllCap; j += 2) {
      const float4 r = rec[j / 2];
      qi[j] = (j < n ? __float_as_int(r.x) : tap * (g.B * HWo)) - tap * (g.B * HWo);   // column index -> b * Ho * Wo + pix
      wi[j] = j < n ? r.y : 0.f;
      qi[j + 1] = (j + 1 < n ? __float_as_int(r.z) : tap * (g.B * HWo)) - tap * (g.B * HWo);
      wi[j + 1] = j + 1 < n ? r.w : 0.f;
    }
    T* dst = S_T + (static_cast<size_t>(gp) * K + tap) * Cout;
    for (int k = 0; k < nv; ++k) {
      const int co = (k * sub + ls) * V;
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
      for (int j = 0; j < kEllCap; ++j) {
        if (j < n) {
          float v[V];
          vec_load(gT + static_cast<size_t>(qi[j]) * Cout + co, v);
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] = fmaf(wi[j], v[e], acc[e]);
        }
      }
      vec_store(dst + co, acc);
    }
  }
}

// contributions beyond kEllCap per (pixel, tap): added with atomics after the gather (rare)
template <typename T>
__global__ void __launch_bounds__(kBlock)
sampleT_overflow_kernel(const T* __restrict__ gT, const int32_t* __restrict__ ovf_count, const EllOverflow* __restrict__ ovf,
                        int ovf_cap, T* __restrict__ S_T, Geom g, int Cout) {
  const int n = min(*ovf_count, ovf_cap);
  const int HWo = g.Ho * g.Wo, K = g.kh * g.kw;
  const int half_c = Cout / 2;                    // channel pairs: one packed atomic each (Cout is even in the plan)
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < static_cast<int64_t>(n) * half_c;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int e = static_cast<int>(i / half_c);
    const int co = 2 * static_cast<int>(i - static_cast<int64_t>(e) * half_c);
    const EllOverflow o = ovf[e];                 // li = b * H * W + pixel (dg == 1), colidx = tap * B * HWo + q
    const int tap = o.colidx / (g.B * HWo), q = o.colidx - tap * (g.B * HWo);
    const T* src = gT + static_cast<size_t>(q) * Cout + co;
    atomic_add2_t(S_T + (static_cast<size_t>(o.li) * K + tap) * Cout + co, o.w * ld(src), o.w * ld(src + 1));
  }
}

// grad_in_T[p, c] = sum over taps and over the sampling points (q, tap) whose bilinear footprint contains pixel p of
// weight * colsG_T[q, tap, c]: the reference's col2im (csrc/cuda/deform_conv_kernel_cuda.cu:353-413) as a GATHER over the same
// inverted index, on the channel-fastest column gradient that the offset gradients need anyway.  Against the transposed
// sampling above it reads the same number of channel vectors per pixel, writes C instead of kh*kw*Cout values per pixel and
// needs no second conv-sized GEMM behind it (round 6).
template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_nhwc_gather_kernel(const T* __restrict__ colsG, const int32_t* __restrict__ counter, const EllEntry* __restrict__ ent,
                          T* __restrict__ ginT, Geom g, int sub, int nv, int64_t npix) {
  constexpr int V = VecT<T>::N;
  constexpr int kMaxNv = 4;                       // C / V <= 256 lanes-vectors, sub = min(C / V, 64)
  const int ls = threadIdx.x % sub;
  const int64_t gp = static_cast<int64_t>(blockIdx.x) * (kBlock / sub) + threadIdx.x / sub;   // b * H * W + pixel
  if (gp >= npix) return;
  const int HW = g.H * g.W, HWo = g.Ho * g.Wo, K = g.kh * g.kw, C = g.C;
  const int b = static_cast<int>(gp / HW), p = static_cast<int>(gp - static_cast<int64_t>(b) * HW);
  float acc[kMaxNv][V];
#pragma unroll
  for (int k = 0; k < kMaxNv; ++k)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[k][e] = 0.f;
  for (int tap = 0; tap < K; ++tap) {
    const size_t col = (static_cast<size_t>(b) * K + tap) * HW + p;             // counter: [img][tap][pixel]
    const int n = min(counter[col], kEllCap);
    const float4* rec = reinterpret_cast<const float4*>(ent + col * kEllCap);
    int32_t qi[kEllCap];
    float wi[kEllCap];
#pragma unroll
    for (int j = 0; j < kEllCap; j += 2) {
      const float4 r = rec[j / 2];
      qi[j] = (j < n ? __float_as_int(r.x) : tap * (g.B * HWo)) - tap * (g.B * HWo);   // column index -> b * Ho * Wo + pix
      wi[j] = j < n ? r.y : 0.f;
      qi[j + 1] = (j + 1 < n ? __float_as_int(r.z) : tap * (g.B * HWo)) - tap * (g.B * HWo);
      wi[j + 1] = j + 1 < n ? r.w : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kEllCap; ++j) {
      if (j < n) {                                 // uniform over the pixel's lane group
        const T* src = colsG + (static_cast<size_t>(qi[j]) * K + tap) * C;
#pragma unroll
        for (int k = 0; k < kMaxNv; ++k) {
          if (k < nv) {
            float v[V];
            vec_load(src + (k * sub + ls) * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[k][e] = fmaf(wi[j], v[e], acc[k][e]);
          }
        }
      }
    }
  }
  T* dst = ginT + static_cast<size_t>(gp) * C;
#pragma unroll
  for (int k = 0; k < kMaxNv; ++k)
    if (k < nv) vec_store(dst + (k * sub + ls) * V, acc[k]);
}

// contributions beyond kEllCap per (pixel, tap): added with packed atomics after the gather (rare)
template <typename T>
__global__ void __launch_bounds__(kBlock)
col2im_nhwc_overflow_kernel(const T* __restrict__ colsG, const int32_t* __restrict__ ovf_count, const EllOverflow* __restrict__ ovf,
                            int ovf_cap, T* __restrict__ ginT, Geom g) {
  const int n = min(*ovf_count, ovf_cap);
  const int HWo = g.Ho * g.Wo, K = g.kh * g.kw, C = g.C;
  const int half_c = C / 2;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < static_cast<int64_t>(n) * half_c;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int e = static_cast<int>(i / half_c);
    const int c = 2 * static_cast<int>(i - static_cast<int64_t>(e) * half_c);
    const EllOverflow o = ovf[e];                 // li = b * H * W + pixel (dg == 1), colidx = tap * B * HWo + q
    const int tap = o.colidx / (g.B * HWo), q = o.colidx - tap * (g.B * HWo);
    const T* src = colsG + (static_cast<size_t>(q) * K + tap) * C + c;
    atomic_add2_t(ginT + static_cast<size_t>(o.li) * C + c, o.w * ld(src), o.w * ld(src + 1));
  }
}

template <typename T>
int nchw_to_nhwc_t(const void* in, void* out, int B, int C, int HW, hipStream_t st_) {
  if (static_cast<int64_t>(B) * ((HW + 63) / 64) * ((C + 31) / 32) > 0x7fffffff) return DETOPS_EUNSUPPORTED;
  const dim3 grid(static_cast<unsigned>(B * ((HW + 63) / 64) * ((C + 31) / 32)));
  if (sizeof(T) == 4)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<unsigned int>, grid, dim3(kBlock), 0, st_, static_cast<const unsigned int*>(in),
                       static_cast<unsigned int*>(out), C, HW);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<unsigned short>, grid, dim3(kBlock), 0, st_, static_cast<const unsigned short*>(in),
                       static_cast<unsigned short*>(out), C, HW);
  return launch_status();
}

template <typename T>
int im2col_nhwc_t(const void* xT, const void* offset, const void* mask, void* colT, const Geom& g, hipStream_t st_) {
  NhwcMap m;
  if (g.dg != 1 || !nhwc_map(g.C, VecT<T>::N, m)) return DETOPS_EUNSUPPORTED;
  const int64_t npix = static_cast<int64_t>(g.B) * g.Ho * g.Wo;
  if (npix == 0) return 0;
  hipLaunchKernelGGL(im2col_nhwc_kernel<T>, dim3(static_cast<unsigned>(ceil_div64(npix, m.ppb))), dim3(kBlock), 0, st_,
                     static_cast<const T*>(xT), static_cast<const T*>(offset), static_cast<const T*>(mask),
                     static_cast<T*>(colT), g, m.sub, m.nv, npix);
  return launch_status();
}

template <typename T>
int coord_nhwc_t(const void* colsG, const void* xT, const void* offset, const void* mask, void* goff, void* gmask,
                 const Geom& g, hipStream_t st_) {
  NhwcMap m;
  if (g.dg != 1 || !nhwc_map(g.C, VecT<T>::N, m)) return DETOPS_EUNSUPPORTED;
  const int64_t npix = static_cast<int64_t>(g.B) * g.Ho * g.Wo;
  if (npix == 0) return 0;
  hipLaunchKernelGGL(coord_nhwc_kernel<T>, dim3(static_cast<unsigned>(ceil_div64(npix, m.ppb))), dim3(kBlock), 0, st_,
                     static_cast<const T*>(colsG), static_cast<const T*>(xT), static_cast<const T*>(offset),
                     static_cast<const T*>(mask), static_cast<T*>(goff), static_cast<T*>(gmask), g, m.sub, m.nv, npix);
  return launch_status();
}

template <typename T>
int sampleT_t(const void* gT, const void* offset, const void* mask, void* S_T, const Geom& g, int Cout, const EllPlan& P,
              void* ws, hipStream_t st_) {
  NhwcMap m;
  if (g.dg != 1 || !nhwc_map(Cout, VecT<T>::N, m)) return DETOPS_EUNSUPPORTED;
  const int64_t npix = static_cast<int64_t>(g.B) * g.H * g.W;
  if (npix == 0 || P.npoints_per_dg == 0) return 0;
  char* w = static_cast<char*>(ws);
  int32_t* count = reinterpret_cast<int32_t*>(w + P.off_count);
  int32_t* ovf_count = reinterpret_cast<int32_t*>(w + P.off_ovf_count);
  EllEntry* ent = reinterpret_cast<EllEntry*>(w + P.off_ent);
  EllOverflow* ovf = reinterpret_cast<EllOverflow*>(w + P.off_ovf);
  { const int rc = ell_build<T>(offset, mask, g, P, count, ovf_count, ent, ovf, P.off_ent - P.off_count, st_); if (rc) return rc; }
  hipLaunchKernelGGL(sampleT_gather_kernel<T>, dim3(static_cast<unsigned>(ceil_div64(npix, m.ppb))), dim3(kBlock), 0, st_,
                     static_cast<const T*>(gT), static_cast<const int32_t*>(count), static_cast<const EllEntry*>(ent),
                     static_cast<T*>(S_T), g, Cout, m.sub, m.nv, npix);
  hipLaunchKernelGGL(sampleT_overflow_kernel<T>, dim3(kNumCU), dim3(kBlock), 0, st_, static_cast<const T*>(gT),
                     static_cast<const int32_t*>(ovf_count), static_cast<const EllOverflow*>(ovf), P.ovf_cap,
                     static_cast<T*>(S_T), g, Cout);
  return launch_status();
}

template <typename T>
int col2im_nhwc_t(const void* colsG, const void* offset, const void* mask, void* ginT, const Geom& g, const EllPlan& P, void* ws,
                  hipStream_t st_) {
  NhwcMap m;
  if (g.dg != 1 || !nhwc_map(g.C, VecT<T>::N, m) || m.nv > 4) return DETOPS_EUNSUPPORTED;
  const int64_t npix = static_cast<int64_t>(g.B) * g.H * g.W;
  if (npix == 0) return 0;
  if (P.npoints_per_dg == 0) {
    DETOPS_HIP_TRY(hipMemsetAsync(ginT, 0, sizeof(T) * static_cast<size_t>(npix) * g.C, st_));
    return 0;
  }
  char* w = static_cast<char*>(ws);
  int32_t* count = reinterpret_cast<int32_t*>(w + P.off_count);
  int32_t* ovf_count = reinterpret_cast<int32_t*>(w + P.off_ovf_count);
  EllEntry* ent = reinterpret_cast<EllEntry*>(w + P.off_ent);
  EllOverflow* ovf = reinterpret_cast<EllOverflow*>(w + P.off_ovf);
  { const int rc = ell_build<T>(offset, mask, g, P, count, ovf_count, ent, ovf, P.off_ent - P.off_count, st_); if (rc) return rc; }
  hipLaunchKernelGGL(col2im_nhwc_gather_kernel<T>, dim3(static_cast<unsigned>(ceil_div64(npix, m.ppb))), dim3(kBlock), 0, st_,
                     static_cast<const T*>(colsG), static_cast<const int32_t*>(count), static_cast<const EllEntry*>(ent),
                     static_cast<T*>(ginT), g, m.sub, m.nv, npix);
  hipLaunchKernelGGL(col2im_nhwc_overflow_kernel<T>, dim3(kNumCU), dim3(kBlock), 0, st_, static_cast<const T*>(colsG),
                     static_cast<const int32_t*>(ovf_count), static_cast<const EllOverflow*>(ovf), P.ovf_cap,
                     static_cast<T*>(ginT), g);
  return launch_status();
}

}  // namespace

#define DETOPS_DTYPE_SWITCH(dtype, CALL)                              \
  switch (dtype) {                                                    \
    case DETOPS_F32: return CALL(float);                              \
    case DETOPS_F16: return CALL(__half);                             \
    case DETOPS_BF16: return CALL(__hip_bfloat16);                    \
    default: return DETOPS_EUNSUPPORTED;                              \
  }

DETOPS_API int detops_deformable_im2col(const void* im, const void* offset, const void* mask,
                                        void* col, int dtype, int B, int C, int H, int W, int kh,
                                        int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                        int dil_h, int dil_w, int deformable_group,
                                        detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                         deformable_group))
    return rc;
  if (B == 0) return 0;
  if (!im || !offset || !col) return DETOPS_EINVAL;
#define CALL(T) im2col_t<T>(im, offset, mask, col, g, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

DETOPS_API int detops_deformable_col2im(const void* col, const void* offset, const void* mask,
                                        void* grad_im, int dtype, int B, int C, int H, int W,
                                        int kh, int kw, int pad_h, int pad_w, int stride_h,
                                        int stride_w, int dil_h, int dil_w, int deformable_group,
                                        detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                         deformable_group))
    return rc;
  if (B == 0) return 0;
  if (!col || !offset || !grad_im) return DETOPS_EINVAL;
#define CALL(T) col2im_t<T>(col, offset, mask, grad_im, g, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

DETOPS_API size_t detops_deformable_col2im_workspace_bytes(int B, int C, int H, int W, int kh, int kw,
                                                            int pad_h, int pad_w, int stride_h, int stride_w,
                                                            int dil_h, int dil_w, int deformable_group) {
  Geom g;
  if (make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return 0;
  GatherPlan P;
  if (B == 0 || !gather_plan(g, P)) return 0;
  EllPlan E;   // the ELL and CSR paths share the workspace: size for the larger plan
  const size_t ell = ell_plan(g, E) ? E.total : 0;
  return P.total > ell ? P.total : ell;
}

DETOPS_API int detops_deformable_col2im_ws(const void* col, const void* offset, const void* mask,
                                           void* grad_im, int dtype, int B, int C, int H, int W,
                                           int kh, int kw, int pad_h, int pad_w, int stride_h,
                                           int stride_w, int dil_h, int dil_w, int deformable_group,
                                           void* workspace, size_t workspace_bytes,
                                           detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                         deformable_group))
    return rc;
  if (B == 0) return 0;
  if (!col || !offset || !grad_im) return DETOPS_EINVAL;
  GatherPlan P;
  // Default: the fixed-width (ELL) inverted index for fp32 and for large maps, the CSR index + gather
  // otherwise (profiles/r02a_opbench_experimental_ab.log, ELL vs CSR: fp32 448 vs 785 / 281 vs 355 /
  // 231 vs 235 us at layer2/3/4; fp16 401 vs 592 / 264 vs 264 / 212 vs 152 us).  Both are free of data
  // atomics except ELL's overflow list.  tuning dcn_col2im = 1 gather | 2 scatter | 3 ell forces one
  // (A/B measurements, tests).
  const int e = detops_tuning().dcn_col2im;
  const bool want_gather = e != 2;
  const bool big = static_cast<int64_t>(g.B) * g.H * g.W >= 16384;
  const bool want_ell = e ? e == 3 : (dtype == DETOPS_F32 || big);
  if (want_ell && workspace) {
    EllPlan E;
    if (ell_plan(g, E) && workspace_bytes >= E.total) {
#define CALL(T) col2im_ell_t<T>(col, offset, mask, grad_im, g, E, workspace, as_stream(stream))
      DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
    }
  }
  const bool scatter = !want_gather || !workspace || !gather_plan(g, P) || workspace_bytes < P.total;
  if (scatter) {
#define CALL(T) col2im_t<T>(col, offset, mask, grad_im, g, as_stream(stream))
    DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
  }
#define CALL(T) col2im_gather_t<T>(col, offset, mask, grad_im, g, P, workspace, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

DETOPS_API int detops_deformable_col2im_coord(const void* col, const void* im, const void* offset,
                                              const void* mask, void* grad_offset, void* grad_mask,
                                              int dtype, int B, int C, int H, int W, int kh, int kw,
                                              int pad_h, int pad_w, int stride_h, int stride_w,
                                              int dil_h, int dil_w, int deformable_group,
                                              detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                         deformable_group))
    return rc;
  if (B == 0) return 0;
  if (!col || !im || !offset || !grad_offset || (mask && !grad_mask)) return DETOPS_EINVAL;
#define CALL(T) coord_t<T>(col, im, offset, mask, grad_offset, grad_mask, g, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}


/* Fused deformable-convolution forward (implicit GEMM on MFMA, fp16 / bf16, conv groups == 1): see the kernel
 * comment above.  Returns DETOPS_EUNSUPPORTED when the shape is outside the fused plan (the caller then runs
 * im2col + GEMM). */
DETOPS_API size_t detops_deform_conv_forward_fused_workspace_bytes(int dtype, int B, int C, int H, int W, int Cout,
                                                                   int kh, int kw, int pad_h, int pad_w, int stride_h,
                                                                   int stride_w, int dil_h, int dil_w,
                                                                   int deformable_group) {
  Geom g;
  if (make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return 0;
  if (B == 0 || !dcn_fused_ok(g, dtype, Cout) || !dcn_fused_preferred(g, Cout)) return 0;
  return dcn_fused_ws_bytes(g, Cout);
}

DETOPS_API int detops_deform_conv_forward_fused(const void* im, const void* weight, const void* offset,
                                                const void* mask, const void* bias, void* out, int dtype, int B,
                                                int C, int H, int W, int Cout, int kh, int kw, int pad_h, int pad_w,
                                                int stride_h, int stride_w, int dil_h, int dil_w,
                                                int deformable_group, void* workspace, size_t workspace_bytes,
                                                detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group))
    return rc;
  if (B == 0) return 0;
  if (!im || !weight || !offset || !out) return DETOPS_EINVAL;
  if (!dcn_fused_ok(g, dtype, Cout)) return DETOPS_EUNSUPPORTED;
  if (!workspace || workspace_bytes < dcn_fused_ws_bytes(g, Cout)) return DETOPS_EWORKSPACE;
  if (dtype == DETOPS_F16)
    return dcn_fused_forward<false, __half>(im, weight, offset, mask, bias, out, g, Cout, workspace, as_stream(stream));
  return dcn_fused_forward<true, __hip_bfloat16>(im, weight, offset, mask, bias, out, g, Cout, workspace, as_stream(stream));
}

// ------------------------------------------------------------------------------------ channels-last pipeline: C ABI
DETOPS_API int detops_nchw_to_nhwc(const void* in, void* out, int dtype, int B, int C, int HW, detops_stream_t stream) {
  if (B < 0 || C < 0 || HW < 0) return DETOPS_EINVAL;
  if (B == 0 || C == 0 || HW == 0) return 0;
  if (!in || !out) return DETOPS_EINVAL;
#define CALL(T) nchw_to_nhwc_t<T>(in, out, B, C, HW, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

// 1 when the channels-last kernels serve this shape (deformable_group == 1, channels / (16 bytes of elements) a power of
// two in [16, 256] for both channel counts), else 0: the caller then uses the reference-layout entry points
DETOPS_API int detops_deformable_nhwc_supported(int dtype, int C, int Cout, int deformable_group) {
  const int V = dtype == DETOPS_F32 ? 4 : 8;
  NhwcMap m;
  return (dtype == DETOPS_F32 || dtype == DETOPS_F16 || dtype == DETOPS_BF16) && deformable_group == 1 &&
         nhwc_map(C, V, m) && nhwc_map(Cout, V, m);
}

DETOPS_API int detops_deformable_im2col_nhwc(const void* xT, const void* offset, const void* mask, void* colT, int dtype,
                                             int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                             int stride_h, int stride_w, int dil_h, int dil_w, int deformable_group,
                                             detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return rc;
  if (B == 0) return 0;
  if (!xT || !offset || !colT) return DETOPS_EINVAL;
#define CALL(T) im2col_nhwc_t<T>(xT, offset, mask, colT, g, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

DETOPS_API int detops_deformable_coord_nhwc(const void* colsG_T, const void* xT, const void* offset, const void* mask,
                                            void* grad_offset, void* grad_mask, int dtype, int B, int C, int H, int W,
                                            int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                            int dil_w, int deformable_group, detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return rc;
  if (B == 0) return 0;
  if (!colsG_T || !xT || !offset || !grad_offset) return DETOPS_EINVAL;
#define CALL(T) coord_nhwc_t<T>(colsG_T, xT, offset, mask, grad_offset, grad_mask, g, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

DETOPS_API size_t detops_deformable_transposed_sample_workspace_bytes(int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                                             int stride_h, int stride_w, int dil_h, int dil_w,
                                                             int deformable_group) {
  Geom g;
  if (make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return 0;
  EllPlan P;
  if (B == 0 || g.dg != 1 || !ell_plan(g, P)) return 0;
  return P.total;
}

DETOPS_API int detops_deformable_transposed_sample(const void* gT, const void* offset, const void* mask, void* S_T, int dtype, int B,
                                         int C, int H, int W, int Cout, int kh, int kw, int pad_h, int pad_w, int stride_h,
                                         int stride_w, int dil_h, int dil_w, int deformable_group, void* workspace,
                                         size_t workspace_bytes, detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return rc;
  if (B == 0) return 0;
  if (!gT || !offset || !S_T || Cout <= 0) return DETOPS_EINVAL;
  EllPlan P;
  if (g.dg != 1 || !ell_plan(g, P)) return DETOPS_EUNSUPPORTED;
  if (!workspace || workspace_bytes < P.total) return DETOPS_EWORKSPACE;
#define CALL(T) sampleT_t<T>(gT, offset, mask, S_T, g, Cout, P, workspace, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}

// grad_in_T [B, H*W, C] (the channels-last input gradient) from the channel-fastest column gradient colsG_T [B*Ho*Wo, kh*kw, C]:
// the reference's col2im as a gather (see col2im_nhwc_gather_kernel).  workspace: detops_deformable_transposed_sample_workspace_bytes.
DETOPS_API int detops_deformable_col2im_nhwc(const void* colsG_T, const void* offset, const void* mask, void* grad_in_T, int dtype,
                                             int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
                                             int stride_w, int dil_h, int dil_w, int deformable_group, void* workspace,
                                             size_t workspace_bytes, detops_stream_t stream) {
  Geom g;
  if (int rc = make_geom(g, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, deformable_group)) return rc;
  if (B == 0) return 0;
  if (!colsG_T || !offset || !grad_in_T) return DETOPS_EINVAL;
  EllPlan P;
  if (g.dg != 1 || !ell_plan(g, P)) return DETOPS_EUNSUPPORTED;
  if (!workspace || workspace_bytes < P.total) return DETOPS_EWORKSPACE;
#define CALL(T) col2im_nhwc_t<T>(colsG_T, offset, mask, grad_in_T, g, P, workspace, as_stream(stream))
  DETOPS_DTYPE_SWITCH(dtype, CALL)
#undef CALL
}
