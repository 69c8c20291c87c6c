// roi_align_bwd.hip — ROIAlign backward for gfx950 (MI355X), fp32 NCHW.
//
// Replaces the reference's RoIAlignBackwardFeature (maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:125-254, zero-init
// :316) behind detops_roi_align_backward_* and detops_roi_align_fpn_backward_* (include/detops.h).
//
// Formulation (not the reference's atomic scatter): ROIAlign's sampling is separable, so its adjoint factors per
// ROI r and channel c into two tiny matrices,
//     grad_in[b_r, c, y, x] += sum_{ph, pw} AY_r[y, ph] * AX_r[x, pw] * grad_out[r, c, ph, pw],
// with AY_r[y, ph] = (1/gh) * sum_iy ([y == lo] * hy + [y == hi] * ly) built from exactly the reference's sample
// taps.  A workgroup OWNS an 8 x 32 pixel tile of one gradient map and a chunk of channels; every thread owns one
// pixel and keeps its channel sums in registers: no atomics on the data, no zero-fill pass (each gradient element
// is written exactly once), bit-reproducible run to run.
//
// Three kernels, one per regime:
//   ring   (default for the model's shapes, needs a caller workspace)  roi_bwd_prep_kernel + roi_align_bwd_ring_kernel
//          pre-pass: per-ROI compact adjoint rows + per-tile hit lists; tiles with long hit lists are SPLIT into
//          segments handled by extra workgroups (launched first) whose partial sums the last arriver combines in
//          segment order.  Main kernel: a ring of LDS slots filled by LDS-DMA, several hits in flight across the
//          per-hit barrier (counted vmcnt waits), separable two-pass walk.
//   scan   (small maps, no workspace)  roi_align_bwd_scan_kernel: every workgroup scans the ROI list itself; the
//          ROI list is split over blockIdx.y for under-filled launches (partial sums by atomics, tiny maps only).
//   atomic (any shape: bin counts beyond both LDS plans)  roi_align_bwd_atomic_kernel: one thread per pooled
//          gradient element, hardware fp32 atomics into the zero-filled map.
#include "roi_align_common.h"

namespace {

constexpr int kGTH = 8, kGTW = 32;     // tile: 8 rows x 32 columns = 256 threads
constexpr int kGBins = 256;            // scan kernel: (ROIs per batch) * bins <= kGBins when bins <= kGBins
constexpr int kGRowPad = kGTW + 1;     // scan kernel: transposition buffer row stride (bank spread)

// ------------------------------------------------------------------------------------------
// ring backward: plan, workspace, pre-pass
// ------------------------------------------------------------------------------------------
constexpr int kSegDefault = 32;   // a tile's hit list is split once it holds >= 2 * seg hits, into floor(hits / seg) segments
constexpr int kMaxSeg = 8;        // segments per tile (= workgroups per (tile, channel chunk))
constexpr int kEntRound = 64;     // hit-list entries parked in LDS per round (one round unless a split was refused)
constexpr int kCtrlBytes = 256;   // control block: [0] extra segments, [1] partial slots

struct RingPlan {
  int first_tile[DETOPS_MAX_LEVELS];   // tile-id offset per level (coarsest level first: its tiles see the most ROIs)
  int n_tiles[DETOPS_MAX_LEVELS];
  int tiles_x[DETOPS_MAX_LEVELS], tiles_y[DETOPS_MAX_LEVELS];
  int num_tiles, cap;                  // cap = hit-list capacity per tile (= K)
  int chunks, accumulate;
  int PPH, PPW;                        // adjoint-row strides: {head, weights ...}, multiple of 4 floats
  int Hmax, Wmax;                      // per-ROI table slot = Hmax * PPH + Wmax * PPW floats
  int tab_blocks;                      // pre-pass: blocks [0, tab_blocks) = role A, the rest role B
  int seg, extra_cap, slot_cap;        // split policy and the capacities of the extras table / partial-sum slots
  int split_ceil, max_seg;
  int debug;                           // ablation bits (tuning roi_bwd_debug): 1 = skip the walk
  int nhwc;                            // the gradient maps are channels-last ([N, H, W, C]): only the store epilogue differs
};

struct RingWs {
  int* ctrl;         // [64] zeroed per call: [0] extra segments appended, [1] partial slots handed out; [16 + 4 l ...] level l's {grad map pointer, H, W}
  int* arrive;       // [num_tiles * chunks] zeroed per call: contributors of a split (tile, chunk) that have published
  int4* heads;       // [num_tiles] {hit count, segments | hits per segment << 8, first partial slot | image << 16, level | tile row << 4 | tile column << 16}
  int2* extras;      // [extra_cap] {tile, segment >= 1} (tile < 0: refused)
  int4* lists;       // [num_tiles][cap] {roi * C * bins, roi * table bytes, fy0 | ny << 16, fx0 | nx << 16}, ascending ROI index
  float* tabs;       // [K][Hmax * PPH + Wmax * PPW], then 16 bytes of zeros
  float* partials;   // [slot_cap][chunks][256 pixels][CT]
  long long* timeline;   // diagnosis only (roi_bwd_debug & 64): [grid.y][chunks] {start, end, hits, unit}
};

struct RingLayout { size_t off_arrive, off_heads, off_extras, off_lists, off_tabs, off_partials, off_timeline, zero_bytes, total; };

// tile id -> (level, image, first row, first column); wave-uniform
struct TileGeom { int lvl, b, y0, x0; };
__device__ __forceinline__ TileGeom tile_geom(const Levels& L, const RingPlan& P, int tile) {
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i < L.num && tile >= P.first_tile[i] && tile < P.first_tile[i] + P.n_tiles[i]) lvl = i;
  int ntx = P.tiles_x[0], nty = P.tiles_y[0], first = P.first_tile[0];
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { ntx = P.tiles_x[i]; nty = P.tiles_y[i]; first = P.first_tile[i]; }
  int rem = tile - first;
  const int tix = rem % ntx; rem /= ntx;
  TileGeom g;
  g.lvl = lvl; g.b = rem / nty; g.y0 = (rem % nty) * kGTH; g.x0 = tix * kGTW;
  return g;
}

constexpr int kPrepTiles = 4;      // tiles per role-B workgroup (1 per wave)
constexpr int kPrepRois = 1024;    // ROI footprints parked in LDS per pass

// Pre-pass, one launch, two roles.
//   role A — one wave per ROI, a lane per (axis, footprint pixel): the pixel's row of the adjoint matrix in COMPACT
//       form {first contributing bin | other axis' longest range << 8 | count << 16, w[0], w[1], ...} (zero beyond), candidate samples from the
//       inverse of the sample-coordinate map, decided by the exact reference arithmetic.  Rows are stored
//       footprint-relative (row 0 = the ROI's first reachable pixel) in a per-ROI slot of the workspace.
//   role B — a wave per gradient-map tile (4 tiles per workgroup sharing one LDS copy of the ROI footprints):
//       ordered (ascending ROI index) ballot compaction of the ROIs that reach the tile -> hit list + head.  A tile
//       whose list is long asks for extra segments: partial-sum slots and extras-table entries are handed out by two
//       atomics on the (zeroed) control block; a refused request leaves the tile unsplit (correct, just a long chain).
__global__ void __launch_bounds__(kBlock)
roi_bwd_prep_kernel(Levels L, RingPlan P, RingWs ws, const float* __restrict__ rois,
                    const int32_t* __restrict__ levels_in, int K, int C, int PH, int PW, int sr) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (static_cast<int>(blockIdx.x) < P.tab_blocks) {
    const int r = static_cast<int>(blockIdx.x) * (kBlock / kWave) + wave;
    if (blockIdx.x == 0 && tid < 4)   // the zero piece behind the last ROI's rows (rows of pixels outside a footprint)
      ws.tabs[static_cast<size_t>(K) * (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW) + tid] = 0.f;
    if (r >= K) return;
    const float* roi = rois + static_cast<size_t>(r) * 5;
    const int lvl = (L.num > 1) ? levels_in[r] : 0;
    int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
    for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
      if (i == lvl) { H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }
    if (lvl < 0 || lvl >= L.num) return;
    float* slot = ws.tabs + static_cast<size_t>(r) * (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW);
    build_adjoint_rows(roi, scale, H, W, PH, PW, sr, slot, static_cast<size_t>(P.Hmax) * P.PPH, P.PPH, P.PPW, lane);
    return;
  }
  // ---- role B: hit lists.  Per pass of <= 1024 ROIs: (1) all threads park the ROIs' footprints {roi | level |
  //      image, fy0, fx0, ny | nx} in LDS — every ROI is read from memory once per WORKGROUP, not once per tile —,
  //      (2) each wave takes a tile and compacts the ROIs reaching it in ascending index (ballot + popcount).
  __shared__ int4 s_ext[kPrepRois];
  const int tile = (static_cast<int>(blockIdx.x) - P.tab_blocks) * kPrepTiles + wave;
  const bool have_tile = tile < P.num_tiles;
  const TileGeom tg = tile_geom(L, P, have_tile ? tile : 0);
  int Ht = L.lv[0].H, Wt = L.lv[0].W;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == tg.lvl) { Ht = L.lv[i].H; Wt = L.lv[i].W; }
  const int y1 = min(Ht, tg.y0 + kGTH) - 1, x1 = min(Wt, tg.x0 + kGTW) - 1;
  const int key = (tg.lvl << 16) | (tg.b << 19);
  int4* list = ws.lists + static_cast<size_t>(have_tile ? tile : 0) * P.cap;
  const unsigned gstride = static_cast<unsigned>(C) * PH * PW;      // floats of grad_out per ROI
  const unsigned tab_stride_b = static_cast<unsigned>((static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW) * 4);
  int c = 0;
  for (int r0 = 0; r0 < K; r0 += kPrepRois) {
    const int nr = min(kPrepRois, K - r0);
    // all four ROIs of a thread are fetched before any is used (one memory round trip per pass, not four)
    constexpr int kPer = kPrepRois / kBlock;
    float rv[kPer][5];
    int rlv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      const int r = r0 + min(i, nr - 1);
      const float* roi = rois + static_cast<size_t>(r) * 5;
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) rv[k][cc] = roi[cc];
      rlv[k] = (L.num > 1) ? levels_in[r] : 0;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      if (i >= nr) continue;
      const int r = r0 + i;
      const int rl = rlv[k];
      int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
      for (int q = 1; q < DETOPS_MAX_LEVELS; ++q)
        if (q == rl) { H = L.lv[q].H; W = L.lv[q].W; scale = L.lv[q].scale; }
      int4 ent = make_int4(-1, 0, 0, 0);   // never matches a tile
      if (rl >= 0 && rl < L.num) {
        const RoiExtent e = roi_extent(rv[k], scale, H, W, PH, PW, sr);
        if (e.ny > 0 && e.nx > 0 && e.b >= 0 && e.b < 4096)
          ent = make_int4(r | (rl << 16) | (e.b << 19), e.fy0, e.fx0, (e.ny << 16) | e.nx);
      }
      s_ext[i] = ent;
    }
    __syncthreads();
    if (have_tile) {
      for (int i0 = 0; i0 < nr; i0 += 4 * kWave) {   // four footprints per lane in flight per trip
        int4 en[4];
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = s_ext[min(i0 + u * kWave + lane, nr - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          hit[u] = i0 + u * kWave + lane < nr && (en[u].x & ~0xffff) == key && en[u].x >= 0 && en[u].y <= y1 &&
                   en[u].y + (en[u].w >> 16) - 1 >= tg.y0 && en[u].z <= x1 && en[u].z + (en[u].w & 0xffff) - 1 >= tg.x0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned long long m = __ballot(hit[u]);
          if (hit[u]) {   // digest: what the main kernel's address arithmetic needs, nothing to derive per unit
            const unsigned r = static_cast<unsigned>(en[u].x & 0xffff);
            list[c + __popcll(m & ((1ull << lane) - 1ull))] =
                make_int4(static_cast<int>(r * gstride), static_cast<int>(r * tab_stride_b), en[u].y | (en[u].w & ~0xffff), en[u].z | (en[u].w << 16));
          }
          c += __popcll(m);
        }
      }
    }
    __syncthreads();   // s_ext is rewritten by the next pass
  }
  if (!have_tile || lane != 0) return;
  int nseg = 1, sbase = 0;
  // split policy (plan): 0 = floor(c / seg) segments of seg .. 2 seg - 1 hits once c >= 2 seg; 1 = ceil(c / seg)
  // segments of at most seg hits once c > seg + seg / 4
  const bool split = P.split_ceil ? (c > P.seg + P.seg / 4) : (c >= 2 * P.seg);
  if (split) {
    const int want = min(P.max_seg, P.split_ceil ? (c + P.seg - 1) / P.seg : c / P.seg);
    const int base = atomicAdd(&ws.ctrl[1], want);
    if (base + want <= P.slot_cap) {
      const int e0 = atomicAdd(&ws.ctrl[0], want - 1);
      const bool ok = e0 + want - 1 <= P.extra_cap;
      for (int s = 1; s < want; ++s)
        if (e0 + s - 1 < P.extra_cap) ws.extras[e0 + s - 1] = ok ? make_int2(tile, s) : make_int2(-1, 0);
      if (ok) { nseg = want; sbase = base; }
    }
  }
  // {hits, segments | hits per segment << 8, first partial slot | image << 16, level | tile row << 4 | tile column << 16}
  ws.heads[tile] = make_int4(c, nseg | (((c + nseg - 1) / nseg) << 8), sbase | (tg.b << 16),
                             tg.lvl | ((tg.y0 / kGTH) << 4) | ((tg.x0 / kGTW) << 16));
  if (tile == 0) DETOPS_STAT("bwdr.prep", 1);
}

// ------------------------------------------------------------------------------------------
// ring backward: main kernel
// ------------------------------------------------------------------------------------------
// Compile-time geometry of one staged hit.  A hit's data is a set of 16-byte PIECES, one LDS-DMA lane each:
//   pooled gradients  piece (c, r, q) = grad_out[roi, c0 + c, r, off(q) .. off(q) + 3], off(q) = min(4 q, PW - 4):
//                     NP pieces cover a bin row (they overlap where PW is not a multiple of 4 and never leave the row,
//                     so no piece reads beyond the tensor) — CHANNEL-major in LDS exactly as in memory, which is what
//                     LDS-DMA can do (lane-linear destination) without a transposing pass through registers;
//   adjoint rows      the tile's 32 AX rows and 8 AY rows {first bin | count << 16, w0, w1, ...} (PPW / 4, PPH / 4
//                     pieces each), or a 16-byte block of zeros for tile pixels outside the ROI's footprint.
// Every wave issues the same number of LDS-DMA instructions per hit — NIG over its quarter of the gradient pieces,
// one over its quarter of the row pieces — each with at least one active lane, so a wave's vmcnt arithmetic is a
// compile-time constant.  An instruction is homogeneous (gradients OR rows): its lanes share one scalar base
// (saddr + per-lane 32-bit offset form), so a gradient piece costs no vector arithmetic per hit at all.  (One wave
// staging a whole hit — the address arithmetic once instead of four times — was measured: the same for 7x7 bins,
// 15 % slower for 14x14, whose 17 instructions per hit then sit in one wave's issue queue.)
template <int PH, int PW, int CT>
struct RingGeom {
  static_assert(PW >= 4 && PW <= 16 && PH <= 16, "bin rows are staged as 1, 2 or 4 pieces of 4 bins");
  static constexpr int BINS = PH * PW;
  static constexpr int NP = PW <= 4 ? 1 : (PW <= 8 ? 2 : 4);
  static constexpr int PPH = (PH + 4) & ~3, PPW = (PW + 4) & ~3;
  static constexpr int RPY = PPH / 4, RPX = PPW / 4;
  static constexpr int GP = CT * PH * NP;                  // gradient pieces
  static constexpr int AXP = kGTW * RPX, AYP = kGTH * RPY; // adjoint-row pieces
  static constexpr int TP = AXP + AYP;
  static_assert(GP % 4 == 0 && TP % 4 == 0 && TP / 4 <= 64, "pieces are dealt evenly to the four waves");
  static constexpr int GPW = GP / 4, TPW = TP / 4;         // per wave
  static constexpr int NIG = (GPW + 63) / 64;              // gradient LDS-DMA instructions per hit and wave
  static constexpr int NI = NIG + 1;                       // + one for the rows
  static constexpr int SLOT_FLOATS = (GP + TP) * 4;
  static constexpr int PWP = PW <= 8 ? 8 : 16;             // strip columns
  static constexpr int TS = CT + 4;                        // strip row stride: bin columns land on distinct bank quads
  static constexpr int STRIP_FLOATS = (kBlock / kWave) * 2 * PWP * TS;
  static constexpr int CPR = 32 / NP;                      // channels per pass-1 round (a half-wave = one tile row)
  static constexpr int ROUNDS = (CT + CPR - 1) / CPR;
  __host__ __device__ static constexpr int off(int q) { return 4 * q < PW - 4 ? 4 * q : PW - 4; }
  static constexpr size_t lds_bytes(int nring) {
    return sizeof(float) * (static_cast<size_t>(nring) * SLOT_FLOATS + STRIP_FLOATS) + sizeof(int4) * (kEntRound + 1);
  }
};

// gradient piece p of a staged ROI -> its LDS piece index when every wave's share (GPW pieces) is padded to GPWp
__host__ __device__ constexpr int wave_region(int p, int GPW, int GPWp) { return (p / GPW) * GPWp + (p % GPW); }

typedef float f2v __attribute__((ext_vector_type(2)));   // packed fp32 pairs: v_pk_fma_f32 (one instruction, two FMAs per lane)
__device__ __forceinline__ f2v pk_fma(float w, float a, float b, f2v acc) {
  return __builtin_elementwise_fma(f2v{w, w}, f2v{a, b}, acc);
}

// One workgroup per unit (tile, segment, CT-channel chunk).  lanes = pixels: wave w owns rows 2w, 2w+1 of the
// 8 x 32 tile (lane = (row & 1) * 32 + column), so a wave's store of one channel is two full 128-byte rows straight
// from the accumulator registers.
//   ring   NR LDS slots of one hit each.  Per hit: wait until this wave's pieces of the hit have landed (counted
//          vmcnt: the NEXT NR - 2 hits stay in flight) | LDS-only barrier | issue the hit NR - 1 ahead into the slot
//          the previous walk has just released | walk.  One barrier per hit, no drain of the DMA queue, no staging
//          registers: the walk of hit j overlaps the memory round trips of hits j + 1 .. j + NR - 1.
//   walk   separable, per wave: pass 1 — lane = (row of the wave, channel, piece) builds T[row][bin column][channel] =
//          sum_a AY[row][a] * g[channel][a][bin column] from one 16-byte LDS read per bin row and parks it in a
//          per-wave LDS strip; pass 2 — lane = pixel sums AX[x][b] * T[row][b][:] over its <= 3 (typically) bin
//          columns.  Weights beyond a pixel's own range are stored as zeros, bin indices are clamped into the staged
//          block (a zero weight times a staged value: exact for finite gradients).  The kernel is bound by VALU
//          issue (PMC: 35 M wave-instructions per box-head launch before this form), hence: packed FMAs, scalar
//          address arithmetic, straight-line code for the common <= 3 x <= 3 bin ranges.
//   split tiles  a unit that is one of several segments of its tile publishes its partial sums (write-through
//          stores, counted wait, barrier, relaxed agent-scope ticket); the unit that draws the last ticket re-reads
//          ALL segments' partials in segment order and stores the tile — same sum order every run.
template <int PH, int PW, int CT, int NR>
__global__ void __launch_bounds__(kBlock, (CT <= 16 ? 6 : 4))
roi_align_bwd_ring_kernel(Levels L, RingPlan P, RingWs ws, const float* __restrict__ gout, int C) {
  using G = RingGeom<PH, PW, CT>;
  static_assert(NR >= 2 && NR <= 4 && CT % 4 == 0, "");
  constexpr int CG = CT / 4;
  DETOPS_DYNAMIC_LDS(float, lds);                                    // ONE LDS object (a second one makes hipcc drain vmcnt)
  float* ring = lds;                                                 // [NR][SLOT_FLOATS]
  float* strip = ring + NR * G::SLOT_FLOATS;                         // [waves][2][PWP][TS]
  int4* s_ent = reinterpret_cast<int4*>(strip + G::STRIP_FLOATS);    // [kEntRound] hit entries of the round

  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int yl = 2 * wave + (lane >> 5), xl = lane & 31;             // this thread's pixel within the tile
  const unsigned tab_stride_b = static_cast<unsigned>((static_cast<size_t>(P.Hmax) * G::PPH + static_cast<size_t>(P.Wmax) * G::PPW) * 4);
  const unsigned ax_off_b = static_cast<unsigned>(static_cast<size_t>(P.Hmax) * G::PPH * 4);

  // ---- this lane's pieces of a hit (fixed for the whole launch)
  // gradients: channel of the piece and its byte offset inside a channel's [PH][PW] block
  int gch[G::NIG];
  unsigned grest[G::NIG];
  bool gact[G::NIG];
#pragma unroll
  for (int i = 0; i < G::NIG; ++i) {
    const int pw_ = 64 * i + lane;
    gact[i] = pw_ < G::GPW;
    const int p = wave * G::GPW + min(pw_, G::GPW - 1);
    const int c = p / (PH * G::NP), rem = p - c * (PH * G::NP);
    const int r = rem / G::NP, q = rem - r * G::NP;
    gch[i] = c;
    grest[i] = static_cast<unsigned>((r * PW + G::off(q)) * 4);
  }
  // rows: piece t of the wave's share -> AX row (column xx, piece k4) or AY row (row yy, piece k4)
  const int t_ = wave * G::TPW + min(lane, G::TPW - 1);
  const bool tact = lane < G::TPW;
  const bool isx = t_ < G::AXP;
  const int tu = isx ? t_ : t_ - G::AXP;
  const int trel = isx ? tu / G::RPX : tu / G::RPY;                   // tile column / row of the piece's pixel
  const unsigned tk_b = static_cast<unsigned>((isx ? tu % G::RPX : tu % G::RPY) * 16) + (isx ? ax_off_b : 0u);
  const unsigned tstride_b = isx ? G::PPW * 4 : G::PPH * 4;
  const float* tabs_base = ws.tabs;
  const unsigned zero_rel0 = static_cast<unsigned>(static_cast<size_t>(P.cap) * tab_stride_b);   // the zero piece sits behind the last ROI's rows

  // ---- unit state (wave-uniform), rewritten per unit of the persistent loop
  int tile = 0, seg = 0, chunk = 0, hits = 0, nseg = 1, slot_base = 0, n = 0, y0 = 0, x0 = 0, c0 = 0, wy0 = 0;
  int img = 0, lvl = 0;
  const int4* list = ws.lists;
  unsigned goff_u[G::NIG];   // per unit: a ragged last chunk clamps its channels to C - 1 (their sums are never stored)

  auto issue = [&](int j, int slot) {                                // hit j of the round -> ring slot
    if (P.debug & 2) return;                                         // ablation: walk whatever the LDS holds
    const int4 en = s_ent[j];                                        // {g index, table offset, fy0 | ny << 16, fx0 | nx << 16}
    const unsigned gidx = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(en.x));
    const unsigned tab_off = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(en.y));
    const int yp = __builtin_amdgcn_readfirstlane(en.z), xp = __builtin_amdgcn_readfirstlane(en.w);
    const int fy0 = yp & 0xffff, ny = yp >> 16, fx0 = xp & 0xffff, nx = xp >> 16;
    const float* gb = gout + (static_cast<size_t>(gidx) + static_cast<size_t>(c0) * G::BINS);
    const float* tb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(tabs_base) + tab_off);
    float* dst = ring + slot * G::SLOT_FLOATS;
#pragma unroll
    for (int i = 0; i < G::NIG; ++i)
      glds16_async_so(gact[i], gb, goff_u[i], dst + (wave * G::GPW + 64 * i) * 4);
    const int d = trel + (isx ? x0 - fx0 : y0 - fy0);
    const bool valid = static_cast<unsigned>(d) < static_cast<unsigned>(isx ? nx : ny);
    const unsigned voff = valid ? static_cast<unsigned>(d) * tstride_b + tk_b : zero_rel0 - tab_off;
    glds16_async_so(tact, tb, voff, dst + (G::GP + wave * G::TPW) * 4);
  };

  // ---- which unit?  the extra segments of split tiles come first in the grid (they are the long chains)
  int* s_tick = reinterpret_cast<int*>(s_ent + kEntRound);          // arrival ticket of a split tile
  const long long t_start = (P.debug & 64) ? detops_wall_clock() : 0ll;
  f2v acc[CT / 2];
  {
    // grid = (chunks, extras + tiles): linear workgroup id = unit * chunks + chunk, i.e. workgroup b still runs chunk
    // b % chunks on XCD b % 8 — an XCD's L2 keeps seeing the same channel slices of grad_out
    chunk = static_cast<int>(blockIdx.x);
    const int u = static_cast<int>(blockIdx.y);
    if (u < P.extra_cap) {
      if (u >= min(ws.ctrl[0], P.extra_cap)) return;
      const int2 x = ws.extras[u];
      if (x.x < 0) return;
      tile = __builtin_amdgcn_readfirstlane(x.x); seg = __builtin_amdgcn_readfirstlane(x.y);
    } else {
      tile = u - P.extra_cap; seg = 0;
    }
    const int4 hd = ws.heads[tile];
    hits = __builtin_amdgcn_readfirstlane(hd.x);
    {
      const int hy_ = __builtin_amdgcn_readfirstlane(hd.y), hz_ = __builtin_amdgcn_readfirstlane(hd.z), hw_ = __builtin_amdgcn_readfirstlane(hd.w);
      nseg = hy_ & 0xff;
      const int per = hy_ >> 8, first = seg * per;
      n = max(0, min(per, hits - first));                            // this unit's hits: [first, first + n)
      slot_base = hz_ & 0xffff; img = hz_ >> 16;
      lvl = hw_ & 0xf; y0 = ((hw_ >> 4) & 0xfff) * kGTH; x0 = (hw_ >> 16) * kGTW;
      list = ws.lists + static_cast<size_t>(tile) * P.cap + first;
    }
    c0 = chunk * CT;
#pragma unroll
    for (int i = 0; i < G::NIG; ++i) goff_u[i] = static_cast<unsigned>(min(gch[i], C - 1 - c0) * (G::BINS * 4)) + grest[i];
    wy0 = y0 + 2 * wave;                                             // the wave's two rows: wy0, wy0 + 1
    if (tid == 0) { DETOPS_STAT("bwdr.units", 1); DETOPS_STAT("bwdr.hits", n); }
    struct TimelineExit {   // written when the unit leaves, whichever path it takes
      long long* slot; long long t0; int n, tile, seg; bool on;
      __device__ ~TimelineExit() { if (on) { slot[0] = t0; slot[1] = detops_wall_clock(); slot[2] = n; slot[3] = tile * 16 + seg; } }
    } timeline_exit{ws.timeline + 4 * (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x), t_start, n, tile, seg, (P.debug & 64) && tid == 0};
#pragma unroll
    for (int c = 0; c < CT / 2; ++c) acc[c] = f2v{0.f, 0.f};

    for (int r0 = 0; r0 < n; r0 += kEntRound) {
      const int nr = min(kEntRound, n - r0);
      if (r0 > 0) { DETOPS_LDS_BARRIER(); if (tid == 0) DETOPS_STAT("bwdr.rounds", 1); }   // the previous round's last walk has read s_ent
      if (tid < nr) s_ent[tid] = list[r0 + tid];
      __syncthreads();
  #pragma unroll
      for (int j = 0; j < NR - 1; ++j)
        if (j < nr) issue(j, j);
      int slot = 0, islot = NR - 1;                                  // ring slots of hit j / of the hit issued at j
      for (int j = 0; j < nr; ++j) {
        // this wave's pieces of hit j have landed once at most NI * (hits issued after j) instructions are outstanding
        const int later = min(NR - 2, nr - 1 - j);
        if (later >= 2) DETOPS_VMCNT_WAIT(2 * G::NI);
        else if (later == 1) DETOPS_VMCNT_WAIT(G::NI);
        else DETOPS_VMCNT_WAIT(0);
        DETOPS_LDS_BARRIER();                                        // every wave's pieces of hit j; walk j - 1 is over everywhere
        if (j + NR - 1 < nr) issue(j + NR - 1, islot);
        const float* sb = ring + slot * G::SLOT_FLOATS;
        slot = (slot + 1 == NR) ? 0 : slot + 1;
        islot = (islot + 1 == NR) ? 0 : islot + 1;
        if (P.debug & 1) continue;
        // ---- walk hit j.  The entry word and this pixel's two row heads are requested TOGETHER, before the first
        // test looks at any of them (one LDS round trip in front of the walk instead of three dependent ones; the
        // kernel is bound by these chains, not by issue slots: SQ_WAIT_ANY 0.5 at 16 waves per CU)
        const float* ayp = sb + (G::GP + G::AXP) * 4 + yl * G::PPH;    // this pixel's AY row / AX row (compact)
        const float* axp = sb + G::GP * 4 + xl * G::PPW;
        int eyp_v = s_ent[j].z;                                        // fy0 | ny << 16
        float4 hy = *reinterpret_cast<const float4*>(ayp);
        float4 hx = *reinterpret_cast<const float4*>(axp);
        DETOPS_KEEP_TOGETHER3(eyp_v, hy.x, hx.x);
        const int eyp = __builtin_amdgcn_readfirstlane(eyp_v);
        const int efy0 = eyp & 0xffff, eny = eyp >> 16;
        if (efy0 + eny - 1 < wy0 || efy0 > wy0 + 1) continue;          // this ROI misses the wave's two rows
        if (lane == 0) DETOPS_STAT("bwdr.wave_hits", 1);
        const int hyb = __float_as_int(hy.x), hxb = __float_as_int(hx.x);   // first bin | other axis' longest range << 8 | count << 16
      const int ny = hyb >> 16, xlo = hxb & 0xff, nx = hxb >> 16;
      const int r1 = lane >> 5, l5 = lane & 31;
      const int q = l5 % G::NP, cl = l5 / G::NP;
      const float* ay1 = ayp;                                          // (2 * wave + r1 == yl: the same row)
      const float4 hy1 = hy;
      const int ylo1 = __float_as_int(hy1.x) & 0xff;
      float* tw = strip + wave * (2 * G::PWP * G::TS);
      const float* tr = tw + (lane >> 5) * G::PWP * G::TS;
      // trip counts from the two AY heads of the wave's rows (lanes 0 and 32): exact for the rows, the ROI-wide bound for
      // the columns (a zero weight beyond a pixel's own range)
      const int h0 = __builtin_amdgcn_readlane(hyb, 0), h1 = __builtin_amdgcn_readlane(hyb, 32);
      const int na_ = max(h0 >> 16, h1 >> 16), nb_ = max((h0 >> 8) & 0xff, (h1 >> 8) & 0xff);
      if (na_ == 0 || nb_ == 0) continue;                            // nothing of this ROI reaches the wave's pixels
      if (na_ <= 3 && nb_ <= 3) {
        // ---- common case: <= 3 bins per axis — straight-line, weights from the staged heads
        const int na = na_, nb = nb_;
        DETOPS_WAVE_SYNC();                                          // the previous hit's pass-2 reads of the strip are done
          // pass 1, branch-free: all three bin rows of every round are requested before the first product is formed
          // (weights beyond the row's own range are stored as zeros, indices clamped into the staged block): one LDS
          // round trip for the pass instead of one per (round, row) behind scalar branches
          (void)na;
          float4 g4[G::ROUNDS][3];
  #pragma unroll
          for (int k = 0; k < G::ROUNDS; ++k) {
            const int c = min(k * G::CPR + cl, CT - 1);
            const float* gc = sb + (c * PH * G::NP + q) * 4;
  #pragma unroll
            for (int a = 0; a < 3; ++a) g4[k][a] = *reinterpret_cast<const float4*>(gc + min(ylo1 + a, PH - 1) * (G::NP * 4));
          }
          if (G::ROUNDS == 2) DETOPS_PIN6(g4[0][0].x, g4[0][1].x, g4[0][2].x, g4[G::ROUNDS - 1][0].x, g4[G::ROUNDS - 1][1].x, g4[G::ROUNDS - 1][2].x);
  #pragma unroll
          for (int k = 0; k < G::ROUNDS; ++k) {
            const int c = k * G::CPR + cl;
            f2v t0 = pk_fma(hy1.y, g4[k][0].x, g4[k][0].y, f2v{0.f, 0.f}), t1 = pk_fma(hy1.y, g4[k][0].z, g4[k][0].w, f2v{0.f, 0.f});
            t0 = pk_fma(hy1.z, g4[k][1].x, g4[k][1].y, t0); t1 = pk_fma(hy1.z, g4[k][1].z, g4[k][1].w, t1);
            t0 = pk_fma(hy1.w, g4[k][2].x, g4[k][2].y, t0); t1 = pk_fma(hy1.w, g4[k][2].z, g4[k][2].w, t1);
            if (c < CT) {
              float* td = tw + (r1 * G::PWP + min(4 * q, PW - 4)) * G::TS + c;   // bin columns off(q) .. off(q) + 3
              td[0] = t0.x; td[G::TS] = t0.y; td[2 * G::TS] = t1.x; td[3 * G::TS] = t1.y;
            }
          }
          DETOPS_WAVE_SYNC();
          // (exec-masking these reads for lanes the ROI does not reach was measured: no gain, 1.45 -> 1.5 us per hit)
          {
            const float* tp = tr + min(xlo, PW - 1) * G::TS;
  #pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const float4 t4 = *reinterpret_cast<const float4*>(tp + 4 * cg);
              acc[2 * cg] = pk_fma(hx.y, t4.x, t4.y, acc[2 * cg]); acc[2 * cg + 1] = pk_fma(hx.y, t4.z, t4.w, acc[2 * cg + 1]);
            }
          }
          if (nb >= 2) {
            const float* tp = tr + min(xlo + 1, PW - 1) * G::TS;
  #pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const float4 t4 = *reinterpret_cast<const float4*>(tp + 4 * cg);
              acc[2 * cg] = pk_fma(hx.z, t4.x, t4.y, acc[2 * cg]); acc[2 * cg + 1] = pk_fma(hx.z, t4.z, t4.w, acc[2 * cg + 1]);
            }
          }
          if (nb >= 3) {
            const float* tp = tr + min(xlo + 2, PW - 1) * G::TS;
  #pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const float4 t4 = *reinterpret_cast<const float4*>(tp + 4 * cg);
              acc[2 * cg] = pk_fma(hx.w, t4.x, t4.y, acc[2 * cg]); acc[2 * cg + 1] = pk_fma(hx.w, t4.z, t4.w, acc[2 * cg + 1]);
            }
          }
        } else {
          // ---- long ranges (ROIs smaller than their bin grid, slivers): loops over the full compact rows
          int na = 0, nb = 0;
          while (__ballot(na < ny) != 0ull) ++na;
          while (__ballot(nb < nx) != 0ull) ++nb;
          if (lane == 0) DETOPS_STAT("bwdr.long_hits", 1);
          DETOPS_WAVE_SYNC();
  #pragma unroll
          for (int k = 0; k < G::ROUNDS; ++k) {
            const int c = k * G::CPR + cl;
            if (c < CT) {
              const float* gc = sb + (c * PH * G::NP + q) * 4;
              f2v t0 = f2v{0.f, 0.f}, t1 = f2v{0.f, 0.f};
              for (int a = 0; a < na; ++a) {                           // wave-uniform trip count; zero weights beyond the row's range
                const float wya = ay1[1 + a];
                const float4 g4 = *reinterpret_cast<const float4*>(gc + min(ylo1 + a, PH - 1) * (G::NP * 4));
                t0 = pk_fma(wya, g4.x, g4.y, t0); t1 = pk_fma(wya, g4.z, g4.w, t1);
              }
              float* td = tw + (r1 * G::PWP + min(4 * q, PW - 4)) * G::TS + c;
              td[0] = t0.x; td[G::TS] = t0.y; td[2 * G::TS] = t1.x; td[3 * G::TS] = t1.y;
            }
          }
          DETOPS_WAVE_SYNC();
          for (int b2 = 0; b2 < nb; ++b2) {                            // wave-uniform trip count
            const float wxb = axp[1 + b2];
            const float* tp = tr + min(xlo + b2, PW - 1) * G::TS;
  #pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const float4 t4 = *reinterpret_cast<const float4*>(tp + 4 * cg);
              acc[2 * cg] = pk_fma(wxb, t4.x, t4.y, acc[2 * cg]); acc[2 * cg + 1] = pk_fma(wxb, t4.z, t4.w, acc[2 * cg + 1]);
            }
          }
        }
      }
    }

    // ---- epilogue
    bool store = !(hits == 0 && P.accumulate);                       // nothing to add
    if (nseg > 1) {
      // one of several segments of this (tile, chunk): publish, and combine if last
      const size_t unit_floats = static_cast<size_t>(kGTH * kGTW) * CT;
      float* mine = ws.partials + (static_cast<size_t>(slot_base + seg) * P.chunks + chunk) * unit_floats + (yl * kGTW + xl) * CT;
#pragma unroll
      for (int cg = 0; cg < CG; ++cg)
        store_f4_wt(mine + 4 * cg, make_float4(acc[2 * cg].x, acc[2 * cg].y, acc[2 * cg + 1].x, acc[2 * cg + 1].y));
      DETOPS_VMCNT_WAIT(0);
      __syncthreads();
      if (tid == 0) s_tick[0] = atomicAdd(&ws.arrive[tile * P.chunks + chunk], 1);
      __syncthreads();
      store = __builtin_amdgcn_readfirstlane(s_tick[0]) == nseg - 1;
      if (store) {
        if (tid == 0) { DETOPS_ACQUIRE_AGENT(); DETOPS_STAT("bwdr.combines", 1); }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CT / 2; ++c) acc[c] = f2v{0.f, 0.f};
        for (int s = 0; s < nseg; ++s) {                             // segment order: the same sum every run
          const float* part = ws.partials + (static_cast<size_t>(slot_base + s) * P.chunks + chunk) * unit_floats + (yl * kGTW + xl) * CT;
#pragma unroll
          for (int cg = 0; cg < CG; ++cg) {
            const float4 v = *reinterpret_cast<const float4*>(part + 4 * cg);
            acc[2 * cg] += f2v{v.x, v.y}; acc[2 * cg + 1] += f2v{v.z, v.w};
          }
        }
      }
    }
    if (store) {
      // two full 128-byte rows per wave and channel, straight from the accumulators; every in-map element of the
      // tile is written exactly once (zeros where no ROI reaches)
      // the level's gradient map from THIS launch's arguments (wave-uniform select, once per unit): the pre-pass may
      // have run at forward time, before the gradient maps existed
      float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W;
#pragma unroll
      for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
        if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; }
      if (P.nhwc && c0 + CT <= C && !P.accumulate && (C & 3) == 0 && !(P.debug & 256)) {
        // channels-last maps: the thread's CT channel sums are CT consecutive floats of its pixel's channel vector.  Stored
        // straight from the accumulators, one instruction would touch 64 lines with 16 bytes each; instead the wave's
        // two pixel rows go through its LDS patch one after the other and leave as whole CT * 4-byte pieces: CG
        // consecutive lanes per pixel, 64 / CG pixels per store instruction (CT = 32: eight full 128-byte lines).
        constexpr int PITCH = CT + 4;                                  // floats per pixel row of the patch: 16-byte aligned, bank quads rotate
        static_assert(4 * 32 * PITCH <= NR * G::SLOT_FLOATS + G::STRIP_FLOATS, "the transposition patch lives in the ring");
        float* patch = lds + wave * (32 * PITCH);
        const int q = lane % CG, pp = lane / CG;
        constexpr int PPI = kWave / CG;                               // pixels per store instruction
        for (int h = 0; h < 2; ++h) {
          __syncthreads();                                             // the last walk / the previous row's reads are over everywhere
          if ((lane >> 5) == h) {
#pragma unroll
            for (int c = 0; c < CG; ++c)
              *reinterpret_cast<float4*>(patch + xl * PITCH + 4 * c) = make_float4(acc[2 * c].x, acc[2 * c].y, acc[2 * c + 1].x, acc[2 * c + 1].y);
          }
          __syncthreads();
          const int yy = y0 + 2 * wave + h;
          float* row = gin + ((static_cast<size_t>(img) * H + yy) * W + x0) * C + c0 + 4 * q;
#pragma unroll
          for (int i = 0; i < 32 / PPI; ++i) {
            const int px = i * PPI + pp;
            const float4 v = *reinterpret_cast<const float4*>(patch + px * PITCH + 4 * q);
            if (yy < H && x0 + px < W) *reinterpret_cast<float4*>(row + static_cast<size_t>(px) * C) = v;
          }
        }
      } else if (P.nhwc) {
        if (y0 + yl < H && x0 + xl < W) {
          float* dst = gin + ((static_cast<size_t>(img) * H + (y0 + yl)) * W + (x0 + xl)) * C + c0;
          if (c0 + CT <= C && !P.accumulate && (C & 3) == 0) {
#pragma unroll
            for (int c = 0; c < CT / 4; ++c)
              *reinterpret_cast<float4*>(dst + 4 * c) = make_float4(acc[2 * c].x, acc[2 * c].y, acc[2 * c + 1].x, acc[2 * c + 1].y);
          } else {
#pragma unroll
            for (int c = 0; c < CT; ++c) {
              if (c0 + c < C) {
                float v = (c & 1) ? acc[c / 2].y : acc[c / 2].x;
                if (P.accumulate) v += dst[c];
                dst[c] = v;
              }
            }
          }
        }
      } else if (y0 + yl < H && x0 + xl < W) {
        const size_t plane = static_cast<size_t>(H) * W;
        float* dst = gin + (static_cast<size_t>(img) * C + c0) * plane + static_cast<size_t>(y0 + yl) * W + (x0 + xl);
        if (c0 + CT <= C && !P.accumulate) {
#pragma unroll
          for (int c = 0; c < CT / 2; ++c) { dst[(2 * c) * plane] = acc[c].x; dst[(2 * c + 1) * plane] = acc[c].y; }
        } else {
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            if (c0 + c < C) {
              float v = (c & 1) ? acc[c / 2].y : acc[c / 2].x;
              if (P.accumulate) v += dst[c * plane];
              dst[c * plane] = v;
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// scan backward (no workspace; small maps).  Same pixel-owner formulation; every workgroup finds the ROIs that
// reach its tile itself: per round of 256 ROIs an ordered ballot compaction, then per batch of hits (1) the dense
// per-axis coefficient rows of the tile in LDS (reference tap arithmetic), (2) grad_out[r, c0:c0+CT] staged as
// float4 channel groups, (3) every lane walks its own contiguous range of contributing bins.  Waves are 8 x 8 pixel
// blocks.  Under-filled launches (cfg-1: one 14 x 14 map = 2 tiles) split the ROI list over blockIdx.y; the groups'
// partial sums are combined with atomics into the pre-zeroed (tiny) map — the only configuration with atomics.
// ------------------------------------------------------------------------------------------

struct GPlan {
  int first_item[DETOPS_MAX_LEVELS];   // workgroup-id offset per level (coarsest level first)
  int n_items[DETOPS_MAX_LEVELS];
  int tiles_x[DETOPS_MAX_LEVELS], tiles_y[DETOPS_MAX_LEVELS];
  int chunks, accumulate, batch;       // batch = ROIs staged per round
  int groups;                          // > 1: the ROI list is split over blockIdx.y (small maps), see below
};

struct __align__(16) GHit {   // 32 bytes
  float start_w, start_h, bin_w, bin_h;
  int k, gh, gw;
  int xspan;   // (ix0 << 16) | ix1: columns the ROI's taps can reach (conservative), map width < 32768
};

template <int PH_, int PW_, int CT>
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_scan_kernel(Levels L, GPlan P, const float* __restrict__ rois,
                            const int32_t* __restrict__ levels_in, const float* __restrict__ gout,
                            int C, int K, int PHr, int PWr, int sr) {
  static_assert(CT % 4 == 0, "channels are staged as float4 groups");
  constexpr int CG = CT / 4;
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  const int PPH = (PH + 3) & ~3, PPW = (PW + 3) & ~3;   // coefficient rows padded to float4
  const int slots = max(bins, kGBins);                   // (ROI, bin) slots per float4 channel group

  DETOPS_DYNAMIC_LDS(float, g_lds);
  float4* gs4 = reinterpret_cast<float4*>(g_lds);        // [CG][slots] float4   (also the store buffer)
  const int region = max(slots * CT, CT * kGTH * kGRowPad);
  float* ayt = g_lds + region;                           // [batch][kGTH][PPH]
  float* axt = ayt + P.batch * kGTH * PPH;               // [batch][kGTW][PPW]
  __shared__ GHit s_hit[kBlock];
  __shared__ int s_wcount[kBlock / kWave];

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // ---- decode the work item (levels are laid out coarsest first: their tiles see the most ROIs)
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i < L.num && static_cast<int>(blockIdx.x) >= P.first_item[i] &&
        static_cast<int>(blockIdx.x) < P.first_item[i] + P.n_items[i]) lvl = i;
  float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
  int ntx = P.tiles_x[0], nty = P.tiles_y[0], first = P.first_item[0];
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale;
                    ntx = P.tiles_x[i]; nty = P.tiles_y[i]; first = P.first_item[i]; }
  int rem = static_cast<int>(blockIdx.x) - first;
  const int chunk = rem % P.chunks; rem /= P.chunks;
  const int tix = rem % ntx; rem /= ntx;
  const int tiy = rem % nty;
  const int b = rem / nty;
  const int y0 = tiy * kGTH, x0 = tix * kGTW;
  const int y1 = min(H, y0 + kGTH) - 1, x1 = min(W, x0 + kGTW) - 1;  // inclusive
  const int c0 = chunk * CT;
  // this thread's pixel: waves are 8x8 blocks side by side
  const int yl = lane >> 3, xl = wave * 8 + (lane & 7);
  const int wx0 = x0 + wave * 8, wx1 = wx0 + 7;

  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.f;

  // Small maps (e.g. the 14 x 14 cfg-1 map: 2 tiles) leave the chip idle and make one workgroup walk
  // hundreds of ROIs.  There the ROI list is split over blockIdx.y; each group adds its partial
  // sums with global atomics into the pre-zeroed (tiny) map — the only configuration with atomics.
  const int r_begin = static_cast<int>(static_cast<int64_t>(K) * blockIdx.y / P.groups);
  const int r_end = static_cast<int>(static_cast<int64_t>(K) * (blockIdx.y + 1) / P.groups);
  for (int kb = r_begin; kb < r_end; kb += kBlock) {
    // ---- which of ROIs [kb, kb+256) touch this tile?  ordered compaction into s_hit
    const int r = kb + tid;
    bool hit = false;
    GHit h{};
    if (r < r_end) {
      const float* roi = rois + static_cast<size_t>(r) * 5;
      const int rl = (L.num > 1) ? levels_in[r] : 0;
      if (rl == lvl && static_cast<int>(roi[0]) == b) {
#pragma clang fp contract(off)
        const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
        const float rh = g.bin_h * static_cast<float>(PH), rw = g.bin_w * static_cast<float>(PW);
        // rows/cols any tap of this ROI can touch (conservative): floor(first) .. floor(last)+2
        const float fy0 = floorf(fmaxf(g.start_h, 0.f)), fy1 = floorf(g.start_h + rh) + 2.f;
        const float fx0 = floorf(fmaxf(g.start_w, 0.f)), fx1 = floorf(g.start_w + rw) + 2.f;
        hit = fy0 <= static_cast<float>(y1) && fy1 >= static_cast<float>(y0) &&
              fx0 <= static_cast<float>(x1) && fx1 >= static_cast<float>(x0);
        h.start_w = g.start_w; h.start_h = g.start_h; h.bin_w = g.bin_w; h.bin_h = g.bin_h;
        h.k = r; h.gh = g.gh; h.gw = g.gw;
        const int ix0 = static_cast<int>(fminf(fx0, static_cast<float>(W)));
        const int ix1 = static_cast<int>(fminf(fmaxf(fx1, 0.f), static_cast<float>(W)));
        h.xspan = (ix0 << 16) | ix1;
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcount[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int j = 0; j < kBlock / kWave; ++j) {
      const int v = s_wcount[j];
      if (j < wave) before += v;
      total += v;
    }
    if (hit) s_hit[before + __popcll(m & ((1ull << lane) - 1ull))] = h;
    __syncthreads();
    if (tid == 0) { DETOPS_STAT("bwd.scan_rounds", 1); DETOPS_STAT("bwd.hits", total); }

    for (int h0 = 0; h0 < total; h0 += P.batch) {
      const int nb = min(P.batch, total - h0);
      if (tid == 0) DETOPS_STAT("bwd.batches", 1);
      // ---- (1) per-axis coefficient rows of the batch, restricted to this tile
      for (int t = tid; t < nb * (PH + PW); t += kBlock) {
        const int j = t / (PH + PW);
        const int q = t - j * (PH + PW);
        const GHit hj = s_hit[h0 + j];
        if (q < PH) {
          float* row = ayt + j * kGTH * PPH + q;
          for (int i = 0; i < kGTH; ++i) row[i * PPH] = 0.f;
          const float inv = 1.f / static_cast<float>(hj.gh);
          for (int i = 0; i < hj.gh; ++i) {
            const Tap e = axis_entry(hj.start_h, hj.bin_h, q, i, hj.gh, H, 1);
            const int a0 = e.lo - y0, a1 = e.hi - y0;
            if (a0 >= 0 && a0 < kGTH) row[a0 * PPH] += e.h * inv;
            if (a1 >= 0 && a1 < kGTH) row[a1 * PPH] += e.l * inv;
          }
        } else {
          const int qq = q - PH;
          float* row = axt + j * kGTW * PPW + qq;
          for (int i = 0; i < kGTW; ++i) row[i * PPW] = 0.f;
          const float inv = 1.f / static_cast<float>(hj.gw);
          for (int i = 0; i < hj.gw; ++i) {
            const Tap e = axis_entry(hj.start_w, hj.bin_w, qq, i, hj.gw, W, 1);
            const int a0 = e.lo - x0, a1 = e.hi - x0;
            if (a0 >= 0 && a0 < kGTW) row[a0 * PPW] += e.h * inv;
            if (a1 >= 0 && a1 < kGTW) row[a1 * PPW] += e.l * inv;
          }
        }
      }
      // ---- (2) stage grad_out[r, c0:c0+CT, :, :] as float4 channel groups: [cg][j*bins + bin]
      //      (global reads run along a channel's contiguous bins; LDS writes are 16-byte, lane-contiguous)
      for (int u = tid; u < nb * bins * CG; u += kBlock) {
        const int cg = u / (nb * bins);
        const int jb = u - cg * (nb * bins);
        const int j = jb / bins;
        const int bin = jb - j * bins;
        const int cbase = c0 + cg * 4;
        const float* src = gout + (static_cast<size_t>(s_hit[h0 + j].k) * C + cbase) * bins + bin;
        float4 v;
        v.x = (cbase + 0 < C) ? src[0] : 0.f;
        v.y = (cbase + 1 < C) ? src[bins] : 0.f;
        v.z = (cbase + 2 < C) ? src[2 * bins] : 0.f;
        v.w = (cbase + 3 < C) ? src[3 * bins] : 0.f;
        gs4[cg * slots + jb] = v;
      }
      __syncthreads();
      // ---- (3) every pixel gathers from the bins that reach it
      for (int j = 0; j < nb; ++j) {
        const int xspan = s_hit[h0 + j].xspan;
        const int jx0 = xspan >> 16, jx1 = xspan & 0xffff;
        if (jx1 < wx0 || jx0 > wx1) continue;   // this ROI misses the wave's 8 columns
        if (lane == 0) DETOPS_STAT("bwd.wave_roi_tasks", 1);
        const float* ayr = ayt + (j * kGTH + yl) * PPH;
        const float* axr = axt + (j * kGTW + xl) * PPW;
        const float4* gj = gs4 + j * bins;
        // every lane walks its OWN contiguous range of contributing bins (the bins with a sample
        // within one pixel of it: 2-4 per axis for model-sized ROIs); trip counts are the wave maxima,
        // gradient reads are per-lane ds_read_b128.  Measured against the union-of-the-wave walk it
        // replaced (profiles/r02a_opbench_experimental_ab.log): box head 217 -> 189 us, mask head
        // 211 -> 147 us, bit-identical sums.
        int ylo = PH, yhi = -1, xlo = PW, xhi = -1;
        for (int ph = 0; ph < PH; ++ph) if (ayr[ph] != 0.f) { ylo = min(ylo, ph); yhi = ph; }
        for (int pw = 0; pw < PW; ++pw) if (axr[pw] != 0.f) { xlo = min(xlo, pw); xhi = pw; }
        const int ny = yhi - ylo + 1, nx = xhi - xlo + 1;   // <= 0: nothing reaches this pixel
        int na = 0, nb_ = 0;
        while (__ballot(na < ny) != 0ull) ++na;
        while (__ballot(nb_ < nx) != 0ull) ++nb_;
        for (int a = 0; a < na; ++a) {
          const int ph = min(ylo + a, PH - 1);
          const float wy = (a < ny) ? ayr[ph] : 0.f;
          for (int b2 = 0; b2 < nb_; ++b2) {
            if (lane == 0) DETOPS_STAT("bwd.bodies_lane_walk", 1);
            const int pw = min(xlo + b2, PW - 1);
            const float w = (b2 < nx) ? wy * axr[pw] : 0.f;
            if (w != 0.f) {
              DETOPS_STAT("bwd.active_lane_bodies", 1);
              const float4* gp = gj + ph * PW + pw;
#pragma unroll
              for (int cg = 0; cg < CG; ++cg) {
                const float4 g4 = gp[cg * slots];
                acc[4 * cg + 0] = fmaf(w, g4.x, acc[4 * cg + 0]);
                acc[4 * cg + 1] = fmaf(w, g4.y, acc[4 * cg + 1]);
                acc[4 * cg + 2] = fmaf(w, g4.z, acc[4 * cg + 2]);
                acc[4 * cg + 3] = fmaf(w, g4.w, acc[4 * cg + 3]);
              }
            }
          }
        }
      }
      __syncthreads();  // the next batch (or scan round, or the store) rewrites the staging region
    }
  }

  if (tid == 0) DETOPS_STAT("bwd.workgroups", 1);
  // ---- store: registers -> LDS [c][8][33] -> full 128-byte rows; every in-map element of the
  //      tile is written exactly once (zeros where no ROI reaches)
  float* tb = g_lds;
#pragma unroll
  for (int c = 0; c < CT; ++c) tb[(c * kGTH + yl) * kGRowPad + xl] = acc[c];
  __syncthreads();
  const size_t plane = static_cast<size_t>(H) * W;
  const int cn = min(CT, C - c0);
  float* gb = gin + (static_cast<size_t>(b) * C + c0) * plane;
  for (int e = tid; e < cn * kGTH * kGTW; e += kBlock) {
    const int c = e / (kGTH * kGTW);
    const int pix = e - c * (kGTH * kGTW);
    const int yy = pix / kGTW, xx = pix - yy * kGTW;
    if (y0 + yy <= y1 && x0 + xx <= x1) {
      float* dst = gb + static_cast<size_t>(c) * plane + static_cast<size_t>(y0 + yy) * W + (x0 + xx);
      float v = tb[(c * kGTH + yy) * kGRowPad + xx];
      if (P.groups > 1) {
        if (v != 0.f) atomicAdd(dst, v);
      } else {
        if (P.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}
// ------------------------------------------------------------------------------------------
// acc backward (small maps with a workspace — BASELINE configs[0]: 512 ROIs on ONE 14 x 14 map).  The ring kernel's
// pixel-owner tiles leave such a launch with two tiles; the scan kernel splits the ROI list and pays with data atomics
// (119 us, 4.6x its own forward).  Here the gradient map of one image x a CT-channel chunk LIVES IN LDS (14 x 14 x 16
// floats = 12.5 KB): a workgroup walks its share of the ROI list in index order and adds every ROI's footprint into
// the LDS map — lanes enumerate the ROI's OWN pixels (no idle lanes over a fixed tile), the adds are LDS adds to
// addresses no other lane touches in the same pass, passes are separated by barriers: no global atomics, the same
// sum order every run.  Per round of <= 8 ROIs: four waves build the ROIs' compact adjoint rows straight into LDS
// (build_adjoint_rows: no pre-pass launch) while the first pooled-gradient blocks are already in flight; per ROI
// (one LDS-only barrier each):
//   staging  pooled gradients [c][bin] by LDS-DMA into a ring of slots, ahead of the walk, completion counted with
//            s_waitcnt vmcnt(N);
//   pass 1   task = (footprint row, channel, bin-row piece): T[row][bin column][channel] = sum_a AY[row][a] g[c][a][.];
//   pass 2   task = (footprint row, footprint column, 4-channel group): sum_b AX[col][b] T[row][b][cg] -> ds_add into
//            the LDS map at the pixel.
// The ROI list is split over `groups` workgroups per (image, chunk); each stores its LDS map as a partial and a second
// launch (roi_align_bwd_acc_combine_kernel) adds the partials in GROUP ORDER into the gradient map: no counters to
// clear, no tickets, two launches in all (the ticket form's last arriver summed 32 partials with one memory round
// trip per partial: 33 us of an 87 us call).
// ------------------------------------------------------------------------------------------
constexpr int kAccMaxDim = 32;      // map height / width served
constexpr int kAccRound = 8;        // candidate ROIs per round (= adjoint-row slots in LDS)

struct AccPlan {
  int H, W, HWp;                    // HWp: H * W rounded up to 4
  int chunks, groups, accumulate;
  int debug;                        // ablation bits (tuning roi_bwd_debug): 1 skip the passes, 2 skip the staging
};

template <int PH, int PW, int CT>
struct AccGeom {
  using G = RingGeom<PH, PW, CT>;
  static constexpr int GPWp = ((G::GPW + 63) / 64) * 64;    // per-wave gradient region padded to whole instructions
  static constexpr int SLOTF = 4 * GPWp * 4;                // floats per ring slot
  __host__ __device__ static constexpr int rows_floats(int H, int W) { return H * G::PPH + W * G::PPW; }
  __host__ __device__ static constexpr size_t lds_bytes(int nring, int H, int W, int HWp) {
    return sizeof(float) * (static_cast<size_t>(nring) * SLOTF + 2 * static_cast<size_t>(H) * G::PWP * G::TS +
                            static_cast<size_t>(CT) * HWp + static_cast<size_t>(kAccRound) * rows_floats(H, W)) +
           sizeof(int4) * (kAccRound + 1);
  }
};

template <int PH, int PW, int CT, int NR>
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_acc_kernel(AccPlan P, const float* __restrict__ rois, const float* __restrict__ gout,
                         float* __restrict__ gin, float* __restrict__ partials, float scale, int C, int K, int sr) {
  using G = RingGeom<PH, PW, CT>;
  using A = AccGeom<PH, PW, CT>;
  static_assert(NR >= 3 && NR <= 4 && CT % 4 == 0, "");
  constexpr int CG = CT / 4;
  constexpr int ROWT = CT * G::NP;                                   // pass-1 tasks per footprint row
  constexpr int SLOTF = A::SLOTF;
  DETOPS_DYNAMIC_LDS(float, lds);
  const int ROWF = A::rows_floats(P.H, P.W);
  float* ring = lds;                                                 // [NR][SLOTF]
  float* tbuf = ring + NR * SLOTF;                                   // [2][H][PWP][TS]
  float* acc = tbuf + 2 * P.H * G::PWP * G::TS;                      // [CT][HWp]
  float* rtab = acc + CT * P.HWp;                                    // [kAccRound][H * PPH + W * PPW] compact adjoint rows
  int4* s_ent = reinterpret_cast<int4*>(rtab + kAccRound * ROWF);    // [kAccRound] {roi, fy0 | ny << 16, fx0 | nx << 16, candidate}
  int* s_cnt = reinterpret_cast<int*>(s_ent + kAccRound);

  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int chunk = static_cast<int>(blockIdx.x);
  const int b = static_cast<int>(blockIdx.y) / P.groups, grp = static_cast<int>(blockIdx.y) % P.groups;
  const int c0 = chunk * CT;

  // ---- this lane's gradient pieces of a staged ROI (fixed for the launch)
  bool gact[G::NIG];
  unsigned goff_u[G::NIG];
#pragma unroll
  for (int i = 0; i < G::NIG; ++i) {
    const int pw_ = 64 * i + lane;
    gact[i] = pw_ < G::GPW;
    const int p = wave * G::GPW + min(pw_, G::GPW - 1);
    const int c = p / (PH * G::NP), rem = p - c * (PH * G::NP);
    const int r = rem / G::NP, q = rem - r * G::NP;
    goff_u[i] = static_cast<unsigned>(min(c, C - 1 - c0) * (G::BINS * 4)) + static_cast<unsigned>((r * PW + G::off(q)) * 4);
  }
  for (int e = tid; e < CT * P.HWp; e += kBlock) acc[e] = 0.f;

  auto issue = [&](int j, int slot) {
    if (P.debug & 2) return;
    const unsigned r = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(s_ent[j].x));
    const float* gb = gout + (static_cast<size_t>(r) * C + c0) * G::BINS;
    float* dst = ring + slot * SLOTF;
#pragma unroll
    for (int i = 0; i < G::NIG; ++i)
      glds16_async_so(gact[i], gb, goff_u[i], dst + (wave * A::GPWp + 64 * i) * 4);
  };
  constexpr int NI = G::NIG;                                         // LDS-DMA instructions per ROI and wave

  const int r_begin = static_cast<int>(static_cast<int64_t>(K) * grp / P.groups);
  const int r_end = static_cast<int>(static_cast<int64_t>(K) * (grp + 1) / P.groups);
  int tpar = 0;                                                      // T buffer parity, continued across rounds
  for (int r0 = r_begin; r0 < r_end; r0 += kAccRound) {
    // ---- the round's ROIs that belong to this image and reach the map, in index order (wave 0: ballot compaction)
    __syncthreads();                                                 // the previous round's walks are over (s_ent, rtab, ring, T)
    if (tid < kWave) {
      const int r = r0 + tid;
      bool ok = false;
      int4 ent = make_int4(0, 0, 0, 0);
      if (tid < kAccRound && r < r_end) {
        const RoiExtent e = roi_extent(rois + static_cast<size_t>(r) * 5, scale, P.H, P.W, PH, PW, sr);
        ok = e.b == b && e.ny > 0 && e.nx > 0;
        ent = make_int4(r, e.fy0 | (e.ny << 16), e.fx0 | (e.nx << 16), 0);
      }
      const unsigned long long m = __ballot(ok);
      if (ok) s_ent[__popcll(m & ((1ull << lane) - 1ull))] = ent;
      if (lane == 0) s_cnt[0] = __popcll(m);
    }
    __syncthreads();
    const int nr = __builtin_amdgcn_readfirstlane(s_cnt[0]);
    if (nr == 0) continue;
    if (tid == 0) { DETOPS_STAT("bwda.rounds", 1); DETOPS_STAT("bwda.rois", nr); }
#pragma unroll
    for (int j = 0; j < NR - 1; ++j)
      if (j < nr) issue(j, j);
    // ---- the round's adjoint rows while the first gradient blocks are in flight.  DENSE rows {head, w[bin 0 .. PB-1]},
    //      built SAMPLE-major: a lane per (ROI, axis, bin) runs the bin's samples in order through the reference tap
    //      arithmetic and adds each tap's weight into the (pixel, bin) cell — a cell is only ever touched by its bin's
    //      lane, so the sums are ordered and race-free; then a lane per (ROI, pixel) finds the row's non-zero range.
    //      (The pixel-major search of the ring's pre-pass — build_adjoint_rows — cost 4 us per ROI and wave here.)
    if (!(P.debug & 32)) {
      for (int e = tid; e < nr * ROWF; e += kBlock) rtab[e] = 0.f;
      DETOPS_LDS_BARRIER();
      for (int t = tid; t < nr * (PH + PW); t += kBlock) {
        const int i = t / (PH + PW), qa = t - i * (PH + PW);
        const bool isy = qa < PH;
        const int q = isy ? qa : qa - PH;
        const int4 en = s_ent[i];
        const RoiGeom g = roi_geometry(rois + static_cast<size_t>(en.x) * 5, scale, PH, PW, sr);
        const int f0 = (isy ? en.y : en.z) & 0xffff, n = (isy ? en.y : en.z) >> 16;
        const int grid = isy ? g.gh : g.gw, size = isy ? P.H : P.W, PP = isy ? G::PPH : G::PPW;
        const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
        const float inv = 1.f / static_cast<float>(grid);
        float* rows = rtab + i * ROWF + (isy ? 0 : P.H * G::PPH) + 1 + q;
        for (int sidx = 0; sidx < grid; ++sidx) {
          const Tap tp = axis_entry(start, bin, q, sidx, grid, size, 1);
          const int a0 = tp.lo - f0, a1 = tp.hi - f0;
          if (tp.h != 0.f && static_cast<unsigned>(a0) < static_cast<unsigned>(n)) rows[a0 * PP] += tp.h * inv;
          if (tp.l != 0.f && static_cast<unsigned>(a1) < static_cast<unsigned>(n)) rows[a1 * PP] += tp.l * inv;
        }
      }
      DETOPS_LDS_BARRIER();
      for (int t = tid; t < nr * 2 * kAccMaxDim; t += kBlock) {
        const int i = t / (2 * kAccMaxDim), p = t - i * (2 * kAccMaxDim);
        const int4 en = s_ent[i];
        const int ny = en.y >> 16, nx = en.z >> 16;
        if (p < ny + nx) {
          const bool isy = p < ny;
          float* row = rtab + i * ROWF + (isy ? p * G::PPH : P.H * G::PPH + (p - ny) * G::PPW);
          const int PB = isy ? PH : PW;
          int lo = -1, hi = -1;
          for (int q = 0; q < PB; ++q)
            if (row[1 + q] != 0.f) { if (lo < 0) lo = q; hi = q; }
          row[0] = __int_as_float((lo >= 0) ? (lo | ((hi - lo + 1) << 16)) : 0);
        }
      }
    }
    {
      const int later = min(NR - 2, nr - 1);                         // ROIs issued after ROI 0
      if (later >= 2) DETOPS_VMCNT_WAIT(2 * NI); else if (later == 1) DETOPS_VMCNT_WAIT(NI); else DETOPS_VMCNT_WAIT(0);
    }
    DETOPS_LDS_BARRIER();                                            // ROI 0 has landed everywhere, every row table is written
    int slot = 0, islot = NR - 1;
    for (int j = 0; j < nr; ++j) {
      const float* sb = ring + slot * SLOTF;
      const int4 en = s_ent[j];
      const int yp = __builtin_amdgcn_readfirstlane(en.y), xp = __builtin_amdgcn_readfirstlane(en.z);
      const int fy0 = yp & 0xffff, ny = yp >> 16, fx0 = xp & 0xffff, nx = xp >> 16;
      float* T = tbuf + tpar * (P.H * G::PWP * G::TS);
      // ---- pass 1: T[row][bin column][channel]
      const float* ayr = rtab + j * ROWF;
      for (int tb = 0; tb < ((P.debug & 1) ? 0 : ny * ROWT); tb += kBlock) {   // wave-uniform trips (the ballots below)
        const bool act = tb + tid < ny * ROWT;
        const int t = min(tb + tid, ny * ROWT - 1);
        const int iy = t / ROWT, l5 = t - iy * ROWT;
        const int q = l5 % G::NP, cl = l5 / G::NP;
        const float* ay = ayr + iy * G::PPH;
        const float4 hy = *reinterpret_cast<const float4*>(ay);
        const int hb = __float_as_int(hy.x);
        const int ylo = hb & 0xff, na = hb >> 16;
        // every load of the task is independent of the others (weights beyond the row's range are stored as zeros,
        // bin rows are clamped into the staged block): all issued back to back, ONE LDS round trip per task — small
        // maps mean ROIs smaller than their bin grid, i.e. long ranges (a dependent loop cost ~130 cycles per step)
        f2v t0 = f2v{0.f, 0.f}, t1 = f2v{0.f, 0.f};
        const int pbase = cl * PH * G::NP + q;
        if (__ballot(act && na > 3) == 0ull) {
          float4 g4[3];
#pragma unroll
          for (int a = 0; a < 3; ++a)
            g4[a] = *reinterpret_cast<const float4*>(sb + wave_region(pbase + min(ylo + a, PH - 1) * G::NP, G::GPW, A::GPWp) * 4);
          float wv[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) wv[a] = ay[1 + min(ylo + a, PH - 1)];   // dense row; zero beyond the range
#pragma unroll
          for (int a = 0; a < 3; ++a) { t0 = pk_fma((a < na) ? wv[a] : 0.f, g4[a].x, g4[a].y, t0); t1 = pk_fma((a < na) ? wv[a] : 0.f, g4[a].z, g4[a].w, t1); }
        } else {
          constexpr int NW = (PH + 3) / 4;                             // float4 reads covering the row's PH weights
          float wv[4 * NW + 4];
#pragma unroll
          for (int k4 = 0; k4 <= NW; ++k4) {
            if (k4 * 4 < G::PPH) {
              const float4 w4 = *reinterpret_cast<const float4*>(ay + 4 * k4);
              wv[4 * k4] = w4.x; wv[4 * k4 + 1] = w4.y; wv[4 * k4 + 2] = w4.z; wv[4 * k4 + 3] = w4.w;
            }
          }
          float4 g4[PH];                                               // dense row: bin a's weight is wv[1 + a], zero outside the range
#pragma unroll
          for (int a = 0; a < PH; ++a)
            g4[a] = *reinterpret_cast<const float4*>(sb + wave_region(pbase + a * G::NP, G::GPW, A::GPWp) * 4);
#pragma unroll
          for (int a = 0; a < PH; ++a) { t0 = pk_fma(wv[1 + a], g4[a].x, g4[a].y, t0); t1 = pk_fma(wv[1 + a], g4[a].z, g4[a].w, t1); }
        }
        float* td = T + (iy * G::PWP + min(4 * q, PW - 4)) * G::TS + cl;
        if (act) { td[0] = t0.x; td[G::TS] = t0.y; td[2 * G::TS] = t1.x; td[3 * G::TS] = t1.y; }
      }
      // ---- ROI j + 1 has landed (this wave's pieces), T of ROI j is complete, pass 2 of ROI j - 1 is over everywhere
      {
        const int later = min(NR - 3, nr - 2 - j);                   // ROIs issued after ROI j + 1 so far
        if (later >= 1) DETOPS_VMCNT_WAIT(NI); else DETOPS_VMCNT_WAIT(0);
      }
      DETOPS_LDS_BARRIER();
      if (j + NR - 1 < nr) issue(j + NR - 1, islot);                 // into the slot ROI j - 1 has just released
      // ---- pass 2: the ROI's own pixels
      const float* axr = ayr + P.H * G::PPH;
      const float inv_nx = 1.f / static_cast<float>(nx);
      for (int tb = 0; tb < ((P.debug & (1 | 8)) ? 0 : ny * nx * CG); tb += kBlock) {
        const bool act = tb + tid < ny * nx * CG;
        const int t = min(tb + tid, ny * nx * CG - 1);
        const int rowcg = static_cast<int>((static_cast<float>(t) + 0.5f) * inv_nx);
        const int xi = t - rowcg * nx;
        const int iy = rowcg / CG, cg = rowcg - iy * CG;
        const float* ax = axr + xi * G::PPW;
        const float4 hx = *reinterpret_cast<const float4*>(ax);
        const int hb = __float_as_int(hx.x);
        const int xlo = hb & 0xff, nb = hb >> 16;
        const float* tr = T + iy * G::PWP * G::TS + 4 * cg;
        f2v v0 = f2v{0.f, 0.f}, v1 = f2v{0.f, 0.f};
        if (__ballot(act && nb > 3) == 0ull) {
          float4 t4[3];
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) t4[b2] = *reinterpret_cast<const float4*>(tr + min(xlo + b2, PW - 1) * G::TS);
          float wv[3];
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) wv[b2] = ax[1 + min(xlo + b2, PW - 1)];
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) { v0 = pk_fma((b2 < nb) ? wv[b2] : 0.f, t4[b2].x, t4[b2].y, v0); v1 = pk_fma((b2 < nb) ? wv[b2] : 0.f, t4[b2].z, t4[b2].w, v1); }
        } else {
          constexpr int NW = (PW + 3) / 4;
          float wv[4 * NW + 4];
#pragma unroll
          for (int k4 = 0; k4 <= NW; ++k4) {
            if (k4 * 4 < G::PPW) {
              const float4 w4 = *reinterpret_cast<const float4*>(ax + 4 * k4);
              wv[4 * k4] = w4.x; wv[4 * k4 + 1] = w4.y; wv[4 * k4 + 2] = w4.z; wv[4 * k4 + 3] = w4.w;
            }
          }
          float4 t4[PW];
#pragma unroll
          for (int b2 = 0; b2 < PW; ++b2) t4[b2] = *reinterpret_cast<const float4*>(tr + b2 * G::TS);
#pragma unroll
          for (int b2 = 0; b2 < PW; ++b2) { v0 = pk_fma(wv[1 + b2], t4[b2].x, t4[b2].y, v0); v1 = pk_fma(wv[1 + b2], t4[b2].z, t4[b2].w, v1); }
        }
        float* ap = acc + (4 * cg) * P.HWp + (fy0 + iy) * P.W + (fx0 + xi);
        // plain read-add-write: no other lane touches these four cells in this pass, passes are barrier-separated
        // (ds_add_f32 cost 18 us of a 63 us call: ~670 cycles per wave-wide LDS float atomic)
        if (act && nb > 0 && !(P.debug & 16)) {
          const float a0 = ap[0], a1 = ap[P.HWp], a2 = ap[2 * P.HWp], a3 = ap[3 * P.HWp];
          ap[0] = a0 + v0.x; ap[P.HWp] = a1 + v0.y; ap[2 * P.HWp] = a2 + v1.x; ap[3 * P.HWp] = a3 + v1.y;
        }
      }
      tpar ^= 1;
      slot = (slot + 1 == NR) ? 0 : slot + 1;
      islot = (islot + 1 == NR) ? 0 : islot + 1;
    }
  }
  __syncthreads();                                                   // every pass-2 add has landed in the LDS map

  // ---- epilogue: the gradient map itself (one group) or this group's partial map
  if (tid == 0) DETOPS_STAT("bwda.units", 1);
  const int HW = P.H * P.W;
  if (P.groups > 1) {
    float* mine = partials + ((static_cast<size_t>(b) * P.chunks + chunk) * P.groups + grp) * (static_cast<size_t>(CT) * P.HWp);
    for (int e = tid * 4; e < CT * P.HWp; e += kBlock * 4)
      *reinterpret_cast<float4*>(mine + e) = *reinterpret_cast<const float4*>(acc + e);
    return;
  }
  float* gb = gin + (static_cast<size_t>(b) * C + c0) * HW;
  const int cn = min(CT, C - c0);
  for (int e = tid; e < cn * HW; e += kBlock) {
    const int c = e / HW, pix = e - c * HW;
    float v = acc[c * P.HWp + pix];
    if (P.accumulate) v += gb[e];
    gb[e] = v;
  }
}

// grad_in[b, c, pix] (+)= sum over groups, in group order, of the groups' partial maps.  One thread per element, the
// loads of a thread independent of each other (16 in flight per trip).
template <int CT>
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_acc_combine_kernel(AccPlan P, const float* __restrict__ partials, float* __restrict__ gin, int N, int C) {
  const int HW = P.H * P.W;
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (idx >= static_cast<int64_t>(N) * C * HW) return;
  const int pix = static_cast<int>(idx % HW);
  const int c = static_cast<int>((idx / HW) % C), b = static_cast<int>(idx / (static_cast<int64_t>(HW) * C));
  const int chunk = c / CT, cl = c - chunk * CT;
  const size_t unit = static_cast<size_t>(CT) * P.HWp;
  const float* p = partials + (static_cast<size_t>(b) * P.chunks + chunk) * P.groups * unit + static_cast<size_t>(cl) * P.HWp + pix;
  float sum = P.accumulate ? gin[idx] : 0.f;
  int g2 = 0;
  for (; g2 + 16 <= P.groups; g2 += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[static_cast<size_t>(g2 + u) * unit];
#pragma unroll
    for (int u = 0; u < 16; ++u) sum += v[u];
  }
  for (; g2 < P.groups; ++g2) sum += p[static_cast<size_t>(g2) * unit];
  gin[idx] = sum;
  if (idx == 0) DETOPS_STAT("bwda.combines", 1);
}

// ------------------------------------------------------------------------------------------
// atomic backward (universal fallback: any bin count, any map).  One thread per pooled-gradient element, the
// reference's sample loop (ROIAlign_cuda.cu:224-252) over the axis taps, hardware fp32 atomics into the map
// (zero-filled by the caller of this kernel unless accumulating).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_atomic_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                            const float* __restrict__ gout, int C, int64_t total, int PH, int PW, int sr) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (idx >= total) return;
  const int pw = static_cast<int>(idx % PW), ph = static_cast<int>((idx / PW) % PH);
  const int c = static_cast<int>((idx / (static_cast<int64_t>(PW) * PH)) % C);
  const int k = static_cast<int>(idx / (static_cast<int64_t>(PW) * PH * C));
  const float* roi = rois + static_cast<size_t>(k) * 5;
  const int lvl = (L.num > 1) ? levels_in[k] : 0;
  if (lvl < 0 || lvl >= L.num) return;
  float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }
  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  const float gv = gout[idx];
  float* plane = gin + (static_cast<size_t>(g.b) * C + c) * (static_cast<size_t>(H) * W);
  for (int iy = 0; iy < g.gh; ++iy) {
    const Tap ty = axis_entry(g.start_h, g.bin_h, ph, iy, g.gh, H, W);
    if (ty.l == 0.f && ty.h == 0.f) continue;       // sample outside the map
    for (int ix = 0; ix < g.gw; ++ix) {
#pragma clang fp contract(off)
      const Tap tx = axis_entry(g.start_w, g.bin_w, pw, ix, g.gw, W, 1);
      if (tx.l == 0.f && tx.h == 0.f) continue;
      atomicAdd(plane + ty.lo + tx.lo, gv * (ty.h * tx.h) / g.count);
      atomicAdd(plane + ty.lo + tx.hi, gv * (ty.h * tx.l) / g.count);
      atomicAdd(plane + ty.hi + tx.lo, gv * (ty.l * tx.h) / g.count);
      atomicAdd(plane + ty.hi + tx.hi, gv * (ty.l * tx.l) / g.count);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host: plans, launches, dispatch
// ------------------------------------------------------------------------------------------
constexpr int kRingCT = 16;      // channels per unit (tuning roi_bwd_ct = 32: twice the channels per unit, half the units)
constexpr int kRingSlots = 3;    // LDS ring depth (hits in flight per workgroup)

inline int ring_ct(int PH) {
  // 7x7 bins: 32-channel units (half the units, half the per-hit scalar work: box head 93.9 -> 82.8 us on the log-uniform
  // ROI set, 107 -> 105 us on the model's; tuning roi_bwd_ct = 16 restores the 16-channel units).  14x14 bins: two
  // 32-channel slots would leave one workgroup per CU.
  const int t = detops_tuning().roi_bwd_ct;
  return (PH == 7 && t != 16) ? 32 : kRingCT;
}

inline bool ring_shape(int PH, int PW) { return (PH == 7 && PW == 7) || (PH == 14 && PW == 14); }

// Plan + workspace carve of the ring backward.  false: shape outside the plan.
bool ring_plan(const Levels& L, int N, int C, int K, int PH, int PW, RingPlan& P, RingLayout& lay) {
  if (!ring_shape(PH, PW) || K <= 0 || K > 65535 || N > 4096 || C <= 0) return false;
  if (static_cast<int64_t>(K) * C * PH * PW > 0xfffffff0ll) return false;   // 32-bit float index into grad_out
  P = RingPlan{};
  P.PPH = (PH + 4) & ~3;
  P.PPW = (PW + 4) & ~3;
  P.cap = K;
  const int CT = ring_ct(PH);
  P.chunks = static_cast<int>(ceil_div64(C, CT));
  int64_t tiles = 0;
  for (int i = L.num - 1; i >= 0; --i) {  // coarsest level first (its tiles see the most ROIs)
    if (L.lv[i].W > 32767 || L.lv[i].H > 32767) return false;
    P.Hmax = max(P.Hmax, L.lv[i].H);
    P.Wmax = max(P.Wmax, L.lv[i].W);
    P.tiles_x[i] = static_cast<int>(ceil_div64(L.lv[i].W, kGTW));
    P.tiles_y[i] = static_cast<int>(ceil_div64(L.lv[i].H, kGTH));
    const int64_t n = static_cast<int64_t>(N) * P.tiles_x[i] * P.tiles_y[i];
    P.first_tile[i] = static_cast<int>(tiles);
    P.n_tiles[i] = static_cast<int>(n);
    tiles += n;
  }
  if (tiles <= 0 || tiles > 65535 - 128 || tiles * P.chunks > 0x3fffffff) return false;   // grid.y = extras + tiles
  // per-ROI row tables are addressed with 32-bit byte offsets from the table base
  if ((static_cast<size_t>(K) + 1) * (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW) * 4 > 0xfffffff0ull) return false;
  P.num_tiles = static_cast<int>(tiles);
  P.tab_blocks = static_cast<int>(ceil_div64(K, kBlock / kWave));   // role A: one wave per ROI
  const int seg = detops_tuning().roi_bwd_seg;
  P.seg = seg > 0 ? max(8, seg) : kSegDefault;
  // extra segments / partial-sum slots: sized by the map set, not by the data (a refused split is only slower)
  const int ecap = detops_tuning().roi_bwd_extras > 0 ? detops_tuning().roi_bwd_extras : 128;
  P.extra_cap = static_cast<int>(std::min<int64_t>(ecap, std::max<int64_t>(8, tiles / 4)));
  P.slot_cap = 2 * P.extra_cap;
  P.split_ceil = detops_tuning().roi_bwd_split == 1;
  P.max_seg = detops_tuning().roi_bwd_maxseg > 0 ? std::min(detops_tuning().roi_bwd_maxseg, 32) : kMaxSeg;
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t o = kCtrlBytes;
  lay.off_arrive = o;   o = up(o + sizeof(int) * static_cast<size_t>(P.num_tiles) * P.chunks);
  lay.zero_bytes = o;   // [0, zero_bytes): control block + arrival counters, zeroed per call
  lay.off_heads = o;    o = up(o + sizeof(int4) * P.num_tiles);
  lay.off_extras = o;   o = up(o + sizeof(int2) * P.extra_cap);
  lay.off_lists = o;    o = up(o + sizeof(int4) * static_cast<size_t>(P.num_tiles) * P.cap);
  lay.off_tabs = o;     o = up(o + sizeof(float) * static_cast<size_t>(K) *
                                   (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW) + 16);   // + the zero piece
  lay.off_partials = o; o = up(o + sizeof(float) * static_cast<size_t>(P.slot_cap) * P.chunks * (kGTH * kGTW) * CT);
  lay.off_timeline = o;
  if (detops_tuning().roi_bwd_debug & 64)     // diagnosis: {start, end, hits, unit} per workgroup (tools/gpu/ring_timeline.py)
    o = up(o + 32 * static_cast<size_t>(P.extra_cap + P.num_tiles) * P.chunks);
  lay.total = o;
  return true;
}

// -1: not applicable (no / too small workspace, shape outside the plan, underfilled launch) -> scan kernel
// phase: 0 = pre-pass + main kernel (one call), 1 = pre-pass only (may run at forward time, on another stream: it
// needs the ROIs and the map shapes, not the gradient), 2 = main kernel only (the workspace holds a phase-1 result)
int run_backward_ring(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                      int N, int C, int K, int PH, int PW, int sr, int accumulate, void* workspace,
                      size_t workspace_bytes, bool forced, hipStream_t st, int phase = 0, int nhwc = 0) {
  if (C == 0 || N == 0) return 0;
  if (!workspace || K == 0) return -1;
  RingPlan P; RingLayout lay;
  if (!ring_plan(L, N, C, K, PH, PW, P, lay) || workspace_bytes < lay.total) return -1;
  P.nhwc = nhwc;
  // underfilled launches (a handful of tiles): the scan kernel's ROI-list split serves them better
  if (!forced && static_cast<int64_t>(P.num_tiles) * P.chunks < 2 * kNumCU) return -1;
  P.accumulate = accumulate;
  P.debug = detops_tuning().roi_bwd_debug;
  unsigned char* base = static_cast<unsigned char*>(workspace);
  RingWs ws{reinterpret_cast<int*>(base), reinterpret_cast<int*>(base + lay.off_arrive),
            reinterpret_cast<int4*>(base + lay.off_heads), reinterpret_cast<int2*>(base + lay.off_extras),
            reinterpret_cast<int4*>(base + lay.off_lists), reinterpret_cast<float*>(base + lay.off_tabs),
            reinterpret_cast<float*>(base + lay.off_partials), reinterpret_cast<long long*>(base + lay.off_timeline)};
  if (phase != 2) {
    DETOPS_HIP_TRY(hipMemsetAsync(base, 0, lay.zero_bytes, st));
    hipLaunchKernelGGL(roi_bwd_prep_kernel, dim3(static_cast<unsigned>(P.tab_blocks + ceil_div64(P.num_tiles, kPrepTiles))), dim3(kBlock), 0, st,
                       L, P, ws, rois, levels_in, K, C, PH, PW, sr);
    if (phase == 1) return launch_status();
  }
  const dim3 grid(static_cast<unsigned>(P.chunks), static_cast<unsigned>(P.extra_cap + P.num_tiles));
  if (PH == 7) {
    const int nr = detops_tuning().roi_bwd_ring ? detops_tuning().roi_bwd_ring : kRingSlots;
#define RING_LAUNCH(CT_, NR_) hipLaunchKernelGGL((roi_align_bwd_ring_kernel<7, 7, CT_, NR_>), grid, dim3(kBlock), (RingGeom<7, 7, CT_>::lds_bytes(NR_)), st, L, P, ws, gout, C)
    if (ring_ct(PH) == 32) { if (nr == 2) RING_LAUNCH(32, 2); else RING_LAUNCH(32, 3); }
    else if (nr == 2) RING_LAUNCH(16, 2); else if (nr == 4) RING_LAUNCH(16, 4); else RING_LAUNCH(16, 3);
#undef RING_LAUNCH
  } else {
    using G = RingGeom<14, 14, kRingCT>;
    hipLaunchKernelGGL((roi_align_bwd_ring_kernel<14, 14, kRingCT, 2>), grid, dim3(kBlock), G::lds_bytes(2), st,
                       L, P, ws, gout, C);
  }
  return launch_status();
}

// ---- acc backward: plan, workspace (the groups' partial maps), launch
constexpr int kAccMaxGroups = 64;

inline bool acc_shape(int H, int W, int PH, int PW) {
  return ring_shape(PH, PW) && H <= kAccMaxDim && W <= kAccMaxDim && H > 0 && W > 0;
}

bool acc_plan(int N, int C, int H, int W, int K, int PH, int PW, AccPlan& P, size_t& bytes) {
  if (!acc_shape(H, W, PH, PW) || K <= 0 || K > 65535 || N <= 0 || N > 4096 || C <= 0) return false;
  if (static_cast<int64_t>(K) * C * PH * PW > 0xfffffff0ll) return false;
  P = AccPlan{};
  P.H = H; P.W = W; P.HWp = (H * W + 3) & ~3;
  // ring + two T buffers + the map + the round's row tables must fit the CU's LDS (14x14 bins on a 31 x 15 map: 182 KB)
  const size_t lds = PH == 7 ? AccGeom<7, 7, kRingCT>::lds_bytes(4, H, W, P.HWp) : AccGeom<14, 14, kRingCT>::lds_bytes(3, H, W, P.HWp);
  if (lds > 150 * 1024) return false;
  P.chunks = static_cast<int>(ceil_div64(C, kRingCT));
  // ROI-list split: ~2 workgroups per CU (measured on cfg-1: 16 / 32 / 64 groups -> 50 / 33 / 37 us), one round
  // (8 ROIs) or more per workgroup
  int64_t groups = ceil_div64(2 * kNumCU, static_cast<int64_t>(N) * P.chunks);
  groups = std::min<int64_t>(groups, std::max<int64_t>(1, K / kAccRound));
  groups = std::max<int64_t>(1, std::min<int64_t>(groups, kAccMaxGroups));
  if (detops_tuning().roi_bwd_groups) groups = max(1, min(kAccMaxGroups, detops_tuning().roi_bwd_groups));
  P.groups = static_cast<int>(groups);
  bytes = sizeof(float) * static_cast<size_t>(N) * P.chunks * kAccMaxGroups * kRingCT * P.HWp;
  return true;
}

// -1: not applicable (no / too small workspace, shape outside the plan) -> the other kernels
int run_backward_acc(const Levels& L, const float* rois, const float* gout, int N, int C, int K, int PH, int PW, int sr,
                     int accumulate, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (C == 0 || N == 0) return 0;
  if (!workspace || K == 0 || L.num != 1) return -1;
  AccPlan P; size_t need = 0;
  const int H = L.lv[0].H, W = L.lv[0].W;
  if (!acc_plan(N, C, H, W, K, PH, PW, P, need) || workspace_bytes < need) return -1;
  P.accumulate = accumulate;
  P.debug = detops_tuning().roi_bwd_debug;
  float* partials = static_cast<float*>(workspace);
  const dim3 grid(static_cast<unsigned>(P.chunks), static_cast<unsigned>(N * P.groups));
  if (PH == 7) {
    using A = AccGeom<7, 7, kRingCT>;
    hipLaunchKernelGGL((roi_align_bwd_acc_kernel<7, 7, kRingCT, 4>), grid, dim3(kBlock), A::lds_bytes(4, H, W, P.HWp), st,
                       P, rois, gout, L.lv[0].gin, partials, L.lv[0].scale, C, K, sr);
  } else {
    using A = AccGeom<14, 14, kRingCT>;
    hipLaunchKernelGGL((roi_align_bwd_acc_kernel<14, 14, kRingCT, 3>), grid, dim3(kBlock), A::lds_bytes(3, H, W, P.HWp), st,
                       P, rois, gout, L.lv[0].gin, partials, L.lv[0].scale, C, K, sr);
  }
  if (P.groups > 1) {
    const int64_t total = static_cast<int64_t>(N) * C * H * W;
    hipLaunchKernelGGL((roi_align_bwd_acc_combine_kernel<kRingCT>), dim3(static_cast<unsigned>(ceil_div64(total, kBlock))),
                       dim3(kBlock), 0, st, P, partials, L.lv[0].gin, N, C);
  }
  return launch_status();
}

// Scan pixel-owner backward launch.  Returns -1 when the shape does not fit its LDS plan (huge bin counts): the
// caller then uses the atomic kernel.
int run_backward_scan(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                        int N, int C, int K, int PH, int PW, int sr, int accumulate, hipStream_t st) {
  if (C == 0 || N == 0) return 0;
  const int bins = PH * PW;
  const int PPH = (PH + 3) & ~3, PPW = (PW + 3) & ~3;
  auto count_items = [&](int ct) {
    int64_t items = 0;
    for (int i = 0; i < L.num; ++i)
      items += static_cast<int64_t>(N) * ceil_div64(L.lv[i].H, kGTH) * ceil_div64(L.lv[i].W, kGTW) * ceil_div64(C, ct);
    return items;
  };
  // 16 channels per workgroup unless that leaves the chip underfilled (small maps) or the staged
  // gradient block would not fit in LDS (bins > 256)
  int CT = 16;
  if (count_items(16) < 2 * kNumCU) CT = 4;
  if (detops_tuning().roi_bwd_scan_ct) CT = (detops_tuning().roi_bwd_scan_ct == 16) ? 16 : 4;   // tests / A-B
  if (bins > kGBins) CT = 4;
  const int slots = max(bins, kGBins);
  const int batch = max(1, kGBins / bins);
  const size_t region = static_cast<size_t>(max(slots * CT, CT * kGTH * kGRowPad));
  const size_t lds = sizeof(float) * (region + static_cast<size_t>(batch) * (kGTH * PPH + kGTW * PPW));
  if (lds > 56 * 1024) return -1;
  for (int i = 0; i < L.num; ++i)
    if (L.lv[i].W > 32767) return -1;  // GHit::xspan packs two 15-bit column indices
  GPlan P{};
  P.chunks = static_cast<int>(ceil_div64(C, CT));
  P.accumulate = accumulate;
  P.batch = batch;
  // ROI-list split for underfilled launches: ~16 ROIs per workgroup, at most 32 groups
  // (cfg-1, 512 ROIs on one 14 x 14 map: 1037 -> 111 us at 7x7 bins, 4640 -> 374 us at 14x14)
  P.groups = 1;
  if (count_items(CT) < 2 * kNumCU && K > 64) P.groups = static_cast<int>(std::min<int64_t>(32, ceil_div64(K, 16)));
  if (detops_tuning().roi_bwd_groups) P.groups = max(1, min(64, detops_tuning().roi_bwd_groups));     // tests / A-B
  int64_t items = 0;
  for (int i = L.num - 1; i >= 0; --i) {  // coarsest level first
    P.tiles_x[i] = static_cast<int>(ceil_div64(L.lv[i].W, kGTW));
    P.tiles_y[i] = static_cast<int>(ceil_div64(L.lv[i].H, kGTH));
    const int64_t n = static_cast<int64_t>(N) * P.tiles_x[i] * P.tiles_y[i] * P.chunks;
    if (items + n > 0x7fffffff) return DETOPS_EUNSUPPORTED;
    P.first_item[i] = static_cast<int>(items);
    P.n_items[i] = static_cast<int>(n);
    items += n;
  }
  if (items == 0) return 0;
  if (P.groups > 1 && !accumulate)
    for (int i = 0; i < L.num; ++i)
      DETOPS_HIP_TRY(hipMemsetAsync(L.lv[i].gin, 0, sizeof(float) * static_cast<size_t>(N) * C * L.lv[i].H * L.lv[i].W, st));
  const dim3 grid(static_cast<unsigned>(items), static_cast<unsigned>(P.groups));
#define SCAN_LAUNCH(PH_, PW_, CT_)                                                                          \
  hipLaunchKernelGGL((roi_align_bwd_scan_kernel<PH_, PW_, CT_>), grid, dim3(kBlock), lds, st, L, P, rois, \
                     levels_in, gout, C, K, PH, PW, sr)
  if (PH == 7 && PW == 7) { if (CT == 16) SCAN_LAUNCH(7, 7, 16); else SCAN_LAUNCH(7, 7, 4); }
  else if (PH == 14 && PW == 14) { if (CT == 16) SCAN_LAUNCH(14, 14, 16); else SCAN_LAUNCH(14, 14, 4); }
  else { if (CT == 16) SCAN_LAUNCH(0, 0, 16); else SCAN_LAUNCH(0, 0, 4); }
#undef SCAN_LAUNCH
  return launch_status();
}
int run_backward_atomic(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                        int N, int C, int K, int PH, int PW, int sr, int accumulate, hipStream_t st) {
  if (C == 0 || N == 0) return 0;
  if (!accumulate)
    for (int i = 0; i < L.num; ++i)
      DETOPS_HIP_TRY(hipMemsetAsync(L.lv[i].gin, 0, sizeof(float) * static_cast<size_t>(N) * C * L.lv[i].H * L.lv[i].W, st));
  const int64_t total = static_cast<int64_t>(K) * C * PH * PW;
  if (total == 0) return 0;
  if (ceil_div64(total, kBlock) > 0x7fffffff) return DETOPS_EUNSUPPORTED;
  hipLaunchKernelGGL(roi_align_bwd_atomic_kernel, dim3(static_cast<unsigned>(ceil_div64(total, kBlock))), dim3(kBlock), 0, st,
                     L, rois, levels_in, gout, C, total, PH, PW, sr);
  return launch_status();
}

// Dispatch: the acc kernel for one small map with a workspace (the map lives in LDS); the ring kernel when the caller
// supplies a workspace, the shape is in its plan and the launch fills the chip; otherwise the scan kernel (small maps: ROI-list split); the atomic kernel for bin counts beyond both LDS
// plans.  Tuning `roi_bwd_impl` = 1 (ring wherever its plan applies, under-filled launches included) | 2 | 3 forces
// one (tests, A/B measurements).
int run_backward(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                 int N, int C, int K, int PH, int PW, int sr, int accumulate, hipStream_t st,
                 void* workspace = nullptr, size_t workspace_bytes = 0) {
  const int impl = detops_tuning().roi_bwd_impl;
  if (impl == 0 || impl == 4) {             // small single maps with a workspace: the whole map accumulates in LDS
    const int rc = run_backward_acc(L, rois, gout, N, C, K, PH, PW, sr, accumulate, workspace, workspace_bytes, st);
    if (rc != -1) return rc;
  }
  if (impl == 0 || impl == 1) {
    const bool forced = impl == 1 && K > 0;
    const int rc = run_backward_ring(L, rois, levels_in, gout, N, C, K, PH, PW, sr, accumulate, workspace,
                                     workspace_bytes, forced, st);
    if (rc != -1) return rc;                // -1: no workspace / shape outside the plan -> the workspace-free kernels
  }
  if (impl != 3) {
    const int rc = run_backward_scan(L, rois, levels_in, gout, N, C, K, PH, PW, sr, accumulate, st);
    if (rc != -1) return rc;
  }
  return run_backward_atomic(L, rois, levels_in, gout, N, C, K, PH, PW, sr, accumulate, st);
}

}  // namespace

DETOPS_API int detops_roi_align_backward_ws_f32(const float* grad_out, const float* rois,
                                                float* grad_in, int N, int C, int H, int W, int K,
                                                int PH, int PW, float spatial_scale,
                                                int sampling_ratio, int zero_grad_in, void* workspace,
                                                size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || H < 0 || W < 0) return DETOPS_EINVAL;
  const size_t bytes = sizeof(float) * static_cast<size_t>(N) * C * H * W;
  if (bytes == 0) return 0;
  if (!grad_in) return DETOPS_EINVAL;
  if (K > 0 && (!grad_out || !rois)) return DETOPS_EINVAL;
  if (K == 0 && !zero_grad_in) return 0;
  Levels L{};
  L.num = 1;
  L.lv[0] = Level{nullptr, grad_in, H, W, spatial_scale};
  return run_backward(L, rois, nullptr, grad_out, N, C, K, PH, PW, sampling_ratio,
                      zero_grad_in ? 0 : 1, as_stream(stream), workspace, workspace_bytes);
}

DETOPS_API int detops_roi_align_backward_f32(const float* grad_out, const float* rois,
                                             float* grad_in, int N, int C, int H, int W, int K,
                                             int PH, int PW, float spatial_scale,
                                             int sampling_ratio, int zero_grad_in,
                                             detops_stream_t stream) {
  return detops_roi_align_backward_ws_f32(grad_out, rois, grad_in, N, C, H, W, K, PH, PW, spatial_scale,
                                          sampling_ratio, zero_grad_in, nullptr, 0, stream);
}

DETOPS_API size_t detops_roi_align_backward_workspace_bytes(const int* H_host, const int* W_host,
                                                            int num_levels, int N, int C, int K, int PH,
                                                            int PW) {
  if (!H_host || !W_host || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || bad_dims(N, C, K, PH, PW) ||
      K == 0 || C == 0 || N == 0)
    return 0;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (H_host[i] <= 0 || W_host[i] <= 0) return 0;
    L.lv[i] = Level{nullptr, nullptr, H_host[i], W_host[i], 1.f};
  }
  size_t need = 0;
  RingPlan P; RingLayout lay;
  if (ring_plan(L, N, C, K, PH, PW, P, lay)) need = lay.total;
  AccPlan AP; size_t abytes = 0;
  if (num_levels == 1 && acc_plan(N, C, H_host[0], W_host[0], K, PH, PW, AP, abytes)) need = std::max(need, abytes);
  return need;
}

DETOPS_API int detops_roi_align_fpn_backward_ws_f32(
    const float* grad_out, const float* rois, const int32_t* levels, float* const* grad_inputs_host,
    const int* H_host, const int* W_host, const float* scale_host, int num_levels, int N, int C,
    int K, int PH, int PW, int sampling_ratio, int zero_grad_in, void* workspace, size_t workspace_bytes,
    detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS ||
      !grad_inputs_host || !H_host || !W_host || !scale_host)
    return DETOPS_EINVAL;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (!grad_inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{nullptr, grad_inputs_host[i], H_host[i], W_host[i], scale_host[i]};
  }
  if (C == 0 || N == 0) return 0;
  if (K > 0 && (!grad_out || !rois || (num_levels > 1 && !levels))) return DETOPS_EINVAL;
  if (K == 0 && !zero_grad_in) return 0;
  return run_backward(L, rois, levels, grad_out, N, C, K, PH, PW, sampling_ratio,
                      zero_grad_in ? 0 : 1, as_stream(stream), workspace, workspace_bytes);
}

// ---- the ring backward in two calls: the pre-pass (hit lists + adjoint rows: a function of the ROIs and the map shapes
// only) can be issued at FORWARD time, on any stream, into a workspace the caller keeps; the backward pass then launches
// the main kernel alone.  Returns DETOPS_EUNSUPPORTED when the ring plan does not apply (the caller uses the one-call entry).
DETOPS_API int detops_roi_align_fpn_backward_prepare_f32(
    const float* rois, const int32_t* levels, const int* H_host, const int* W_host, const float* scale_host,
    int num_levels, int N, int C, int K, int PH, int PW, int sampling_ratio, void* workspace, size_t workspace_bytes,
    detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || !H_host || !W_host || !scale_host)
    return DETOPS_EINVAL;
  if (K <= 0 || C == 0 || N == 0 || !rois || (num_levels > 1 && !levels) || !workspace) return DETOPS_EUNSUPPORTED;
  const int impl = detops_tuning().roi_bwd_impl;
  if (impl != 0 && impl != 1) return DETOPS_EUNSUPPORTED;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{nullptr, nullptr, H_host[i], W_host[i], scale_host[i]};
  }
  const int rc = run_backward_ring(L, rois, levels, nullptr, N, C, K, PH, PW, sampling_ratio, 0, workspace, workspace_bytes,
                                   impl == 1, as_stream(stream), 1);
  return rc == -1 ? DETOPS_EUNSUPPORTED : rc;
}

DETOPS_API int detops_roi_align_fpn_backward_prepared_f32(
    const float* grad_out, float* const* grad_inputs_host, const int* H_host, const int* W_host, const float* scale_host,
    int num_levels, int N, int C, int K, int PH, int PW, int zero_grad_in, void* workspace, size_t workspace_bytes,
    detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || !grad_inputs_host || !H_host ||
      !W_host || !scale_host || !grad_out || !workspace || K <= 0)
    return DETOPS_EINVAL;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (!grad_inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{nullptr, grad_inputs_host[i], H_host[i], W_host[i], scale_host[i]};
  }
  const int rc = run_backward_ring(L, nullptr, nullptr, grad_out, N, C, K, PH, PW, 0, zero_grad_in ? 0 : 1, workspace,
                                   workspace_bytes, true, as_stream(stream), 2);
  return rc == -1 ? DETOPS_EUNSUPPORTED : rc;
}

DETOPS_API int detops_roi_align_fpn_backward_f32(
    const float* grad_out, const float* rois, const int32_t* levels, float* const* grad_inputs_host,
    const int* H_host, const int* W_host, const float* scale_host, int num_levels, int N, int C,
    int K, int PH, int PW, int sampling_ratio, int zero_grad_in, detops_stream_t stream) {
  return detops_roi_align_fpn_backward_ws_f32(grad_out, rois, levels, grad_inputs_host, H_host, W_host,
                                              scale_host, num_levels, N, C, K, PH, PW, sampling_ratio,
                                              zero_grad_in, nullptr, 0, stream);
}

// The ring backward with CHANNELS-LAST gradient maps (grad_inputs[l] stored [N, H, W, C]): the same pre-pass, walk and
// deterministic combine; only the store epilogue differs (a thread's channel sums are consecutive floats of its pixel's
// channel vector).  grad_out stays [K, C, PH, PW].  Returns DETOPS_EUNSUPPORTED when the ring plan does not serve the shape
// (the caller then uses detops_roi_align_fpn_backward_nhwc_f32, csrc/roi_align_nhwc.hip).
DETOPS_API int detops_roi_align_fpn_backward_ring_nhwc_f32(
    const float* grad_out, const float* rois, const int32_t* levels, float* const* grad_inputs_host, const int* H_host,
    const int* W_host, const float* scale_host, int num_levels, int N, int C, int K, int PH, int PW, int sampling_ratio,
    int zero_grad_in, void* workspace, size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || !grad_inputs_host || !H_host || !W_host ||
      !scale_host)
    return DETOPS_EINVAL;
  if (K <= 0 || C == 0 || N == 0 || !grad_out || !rois || (num_levels > 1 && !levels) || !workspace) return DETOPS_EUNSUPPORTED;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (!grad_inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{nullptr, grad_inputs_host[i], H_host[i], W_host[i], scale_host[i]};
  }
  const int rc = run_backward_ring(L, rois, levels, grad_out, N, C, K, PH, PW, sampling_ratio, zero_grad_in ? 0 : 1, workspace,
                                   workspace_bytes, false, as_stream(stream), 0, 1);
  return rc == -1 ? DETOPS_EUNSUPPORTED : rc;
}
