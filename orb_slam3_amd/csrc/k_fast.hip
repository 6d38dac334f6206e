// FAST-9/16: threshold-independent score, per-cell NMS + threshold fallback + compaction, fused in one kernel.
// gfx950 only.
//
// Reference stage: ComputeKeyPointsOctTree cell loop, src/features/orb_feature_extractor.cpp:227-276
// (cv::FAST :259-266).  Closed form used:
//   d_k = v - ring_k,  M_dark = max over the 16 arcs of min over 9 contiguous d,
//   M_bright = max over arcs of min(-d),  score = max(M_dark, M_bright) - 1,  corner at t  <=>  score >= t.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "orbgpu_internal.h"
#include "tile_stage.hpp"

namespace {

// three-input min / max, pinned: left to itself the compiler shares two-input partial results between neighbouring
// windows and ends up with twice as many instructions
__device__ __forceinline__ int vmin3(int a, int b, int c) {
  int r;
  asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int vmax3(int a, int b, int c) {
  int r;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// the same score for a pixel that already passed the opposite-pair test (no early-out): 16 sub + 64 min3/max3 for
// the 9-windows of both polarities + 16 for the two reductions
__device__ __forceinline__ int fast_score16_full(int v, const int (&r)[16], int min_thr) {
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = v - r[k];
  int mn3[16], mx3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    mn3[k] = vmin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    mx3[k] = vmax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
  int mn9[16], mx9[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    mn9[k] = vmin3(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
    mx9[k] = vmax3(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
  }
  int m_dark = vmax3(mn9[0], mn9[1], mn9[2]), m_brightneg = vmin3(mx9[0], mx9[1], mx9[2]);
#pragma unroll
  for (int k = 3; k < 15; k += 2) {
    m_dark = vmax3(m_dark, mn9[k], mn9[k + 1]);
    m_brightneg = vmin3(m_brightneg, mx9[k], mx9[k + 1]);
  }
  m_dark = max(m_dark, mn9[15]);
  m_brightneg = min(m_brightneg, mx9[15]);
  int s = max(m_dark, -m_brightneg) - 1;
  return s >= min_thr ? s : 0;
}

typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;

__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

// bytes b and b+1 of the 12-byte window {w0,w1,w2} zero-extended into the two u16 halves
template <int B>
__device__ __forceinline__ u16x2 pair_bytes(const unsigned (&w)[3]) {
  constexpr int lo = B >> 2;
  constexpr int k0 = B & 3, k1 = k0 + 1;  // byte positions inside the 8-byte pool {w[lo+1], w[lo]}
  constexpr unsigned sel = (unsigned)k0 | (0x0cu << 8) | ((unsigned)k1 << 16) | (0x0cu << 24);
  const unsigned hi = (lo + 1 < 3) ? w[lo + 1 < 3 ? lo + 1 : 2] : 0u;
  return as_u16x2(__builtin_amdgcn_perm(hi, w[lo], sel));
}

// ------------------------------------------------------------------------------------------------
// Fused FAST score + per-cell NMS + threshold fallback + compaction on CELL-ALIGNED tiles.
//
// cv::FAST runs per cell in the reference (orb_feature_extractor.cpp:244-266), so its 3x3 non-maximum suppression is
// blind across cell seams and a workgroup that owns whole cells needs nothing from its neighbours.
// A workgroup takes one "group": up to 4 horizontally adjacent cells of one cell row (<= 128 pixels wide, <= 64 rows),
// stages the pixels of their detection rectangles plus the 3-pixel ring apron, and keeps every intermediate in LDS:
//   phase 1  quick test on pixel PAIRS in packed u16 (4 opposite ring pairs), 4 pixels x 1 row per lane and pass, the
//            group's quads (4-pixel columns x rows) dealt out flat so that no lane idles whatever the cell geometry;
//            candidate flags come from saturating packed arithmetic, no compares;
//   push     survivors -> LDS queue, one LDS atomic per wave;
//   phase 2  full 9-arc score per survivor -> score tile in LDS (zero apron);
//   phase 3  3x3 NMS per scored pixel, neighbours outside the pixel's own cell masked; survivors set a bit in their
//            cell's row mask, one mask for "kept" and one for "kept and >= iniThFAST";
//   phase 4  one wave per cell: threshold fallback decided after NMS, survivors written in raster order to the cell's
//            fixed slot of the candidate pool, CellEntry written.
// No score map, no bit plane and no second kernel: the only global traffic is the tile read and the candidates.
// DUMP = true additionally scatters the non-zero scores into a global score map (orbgpu_debug_dump's stage test).
struct FastGroup { short level, row, j0, ncell; };  // cells j0 .. j0 + ncell - 1 of cell row `row`

constexpr int FC_TP = 36;               // dwords per staged / score row (<= 33 quads + 2 apron dwords)
constexpr int FC_TPB = FC_TP * 4;
constexpr int FC_MAXK = 4;              // cells per group
constexpr int FC_QPITCH = 128;          // queue entries per row: only pixels of the cells are pushed, <= 128 per row (with 136 a 33-row launch took 20.8 KB, 7 workgroups per CU)
constexpr int FC_MAXROWS = 64;          // rows of a cell (hcell <= 64 is checked at create time)

__host__ __device__ constexpr size_t fast_cells_lds_bytes(int rows) {
  return (size_t)(rows + 6) * FC_TPB       // pixel tile
         + (size_t)(rows + 2) * FC_TPB     // score tile
         + (size_t)rows * FC_QPITCH * 2    // survivor queue (u16)
         + (size_t)rows * FC_MAXK * 8      // row mask per (row, cell): kept (u64); 8 workgroups per CU need <= 20 KB
         + FC_TP * 4                       // column validity mask per quad
         + 16;
}

__device__ __forceinline__ int div_small(int q, int d, float inv_d) {
  // q / d for 0 <= q < 2^20, 1 <= d: estimate + two-sided fix-up
  int r = (int)((float)q * inv_d);
  const int rem = q - r * d;
  if (rem < 0) r--;
  if (rem >= d) r++;
  return r;
}

// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps inside each 16-lane row, then the row totals
// travel with row_bcast:15 / row_bcast:31 (lanes without a source add 0)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// PART only names the instantiation: 0 = the launch the roofline is quoted on (levels >= 1, or all levels), 1 = the
// launch for level 0, which runs on the second stream beside the resize chain (profilers keep the two apart by name)
template <bool DUMP, int PART = 0>
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ pyr, DeviceGeom g,
                                                    const FastGroup *__restrict__ groups, int groups_per_frame,
                                                    CellEntry *__restrict__ cells, unsigned *__restrict__ cands,
                                                    int n_frames, Level0Src lvl0, int lds_rows,
                                                    uint8_t *__restrict__ score_dump) {
  extern __shared__ __attribute__((aligned(16))) unsigned fc_lds[];
  unsigned *tile = fc_lds;
  unsigned *stile = tile + (lds_rows + 6) * FC_TP;
  unsigned short *queue = reinterpret_cast<unsigned short *>(stile + (lds_rows + 2) * FC_TP);
  // [row][cell] u64: pixels kept by NMS -- at iniThFAST first; a cell that kept nothing then (its masks are still zero)
  // is the only kind the minThFAST pass writes to, so one mask serves both passes
  unsigned *rmask = reinterpret_cast<unsigned *>(queue + lds_rows * FC_QPITCH);
  unsigned *colmask = rmask + lds_rows * FC_MAXK * 2;
  int *q_count = reinterpret_cast<int *>(colmask + FC_TP);
  int item, frame;
  if (!xcd_item_frame(groups_per_frame, n_frames, item, frame)) return;
  const FastGroup G = groups[item];
  const int l = G.level;
  const LevelGeom L = g.lv[l];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // detection rectangle of the group: rows of cell row G.row, columns of cells j0 .. j0 + ncell - 1
  const int Y0 = ORB_EDGE + G.row * L.hcell, Y1 = min(ORB_EDGE - 1 + (G.row + 1) * L.hcell, L.h - ORB_EDGE - 1);
  const int X0 = ORB_EDGE + G.j0 * L.wcell, X1 = min(ORB_EDGE - 1 + (G.j0 + G.ncell) * L.wcell, L.w - ORB_EDGE - 1);
  const int nrow = Y1 - Y0 + 1, ncolpx = X1 - X0 + 1;
  const int cell_slot = ((L.wcell + 1) >> 1) * ((L.hcell + 1) >> 1);
  CellEntry *ce = cells + (size_t)frame * g.cells_per_frame + L.cell_off + G.row * L.ncols + G.j0;
  if (nrow <= 0 || ncolpx <= 0) {  // nothing to detect in any cell of the group
    if (tid < G.ncell) ce[tid] = CellEntry{0, 0};
    return;
  }
  const int xs0 = X0 & ~3;                         // first quad column; tile byte column c <-> level x = xs0 - 4 + c
  const int ncg = ((X1 - xs0) >> 2) + 1;           // quads per row, <= 33
  const int ndw = ncg + 2;                         // staged dwords per row
  const int nq4 = ncg * nrow;
  const uint8_t *src = level_src(pyr, g, lvl0, frame, l, L.off) + (size_t)(Y0 - 3) * L.pitch + (xs0 - 4);

  // ---- stage (nrow + 6) x ndw dwords: wave = row (w, w + 4, ...), lane = dword; every address is inside the image
  //      (Y0 >= 19, Y1 <= h - 20, X1 <= w - 20).  All loads of a wave are issued before its first LDS write ----
  {
    constexpr int NR = (FC_MAXROWS + 6 + 3) / 4;
    const int nst = nrow + 6;
    unsigned v[NR];
    // lanes beyond the row's dwords repeat its last dword (same address, same LDS slot): no per-lane predicate
    const int ln = min(lane, ndw - 1);
    const uint8_t *lsrc = src + 4 * ln;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int r = wave + 4 * k;  // wave-uniform
      v[k] = 0;
      if (r < nst) v[k] = *reinterpret_cast<const unsigned *>(lsrc + (size_t)r * L.pitch);
    }
    // meanwhile: clear the score tile and the row masks (16 bytes per lane and store), per-quad column masks
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < (nrow + 2) * (FC_TP / 4); i += 256) reinterpret_cast<uint4 *>(stile)[i] = z4;
    if (tid < (nrow * FC_MAXK + 1) / 2) reinterpret_cast<uint4 *>(rmask)[tid] = z4;
    if (tid < 2) q_count[tid] = 0;  // [0] packed queue counts, [1] bit j: cell j falls back to minThFAST
    if (tid < FC_TP) {
      unsigned m = 0;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int x = xs0 + 4 * tid + p;
        if (x >= X0 && x <= X1) m |= 1u << (8 * p);
      }
      colmask[tid] = m;
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int r = wave + 4 * k;
      if (r < nst) tile[r * FC_TP + ln] = v[k];
    }
  }
  __syncthreads();

  const int t_lo = g.fast_min, t_hi = g.fast_init;
  const float inv_ncg = 1.0f / (float)ncg;
  // candidate at threshold t <=> M = max(v - max_k min(pair_k), min_k max(pair_k) - v) > t: with saturating u16
  // arithmetic M + (255 - t) has bit 8 set exactly then.  Two thresholds, one M.
  const unsigned short bias_lo = (unsigned short)(255 - t_lo), bias_hi = (unsigned short)(255 - t_hi);
  const u16x2 bias_lo2 = {bias_lo, bias_lo}, bias_hi2 = {bias_hi, bias_hi};
  // the lane's quad of pass k is q = tid + 256 k = (row ry, quad column cq), advanced by (d_r, d_c) per pass
  int ry = div_small(tid, ncg, inv_ncg), cq = tid - ry * ncg;
  const int d_r = __builtin_amdgcn_readfirstlane(div_small(256, ncg, inv_ncg)), d_c = 256 - d_r * ncg;
  const int qcap = lds_rows * FC_QPITCH;
  // ---- phase 1 + push, 8 passes (2048 quads) at a time ------------------------------------------------------
  // Survivors of the quick test at iniThFAST go to the front of the queue.  Those that only pass at minThFAST stay in
  // the lane's registers (one bit per pixel, keep_lo): a pixel scoring below iniThFAST can neither be kept nor suppress
  // anything in a cell that keeps a corner at iniThFAST (NMS is a strict > against a value >= iniThFAST), so they are
  // only pushed (to the back of the queue) and scored when a cell of the group falls back to minThFAST.
  unsigned keep_lo[2] = {0, 0};  // nq4 <= 33 x 64 quads: two blocks of 2048
  // one LDS atomic per wave for a block's survivors; the order inside the queue is free.  Entries are (quad << 2 | pixel)
  auto push_block = [&](unsigned m, int qbase, bool back) {
    const int cnt = __popc(m);
    const int incl = wave_incl_scan(cnt);
    const int tot = __builtin_amdgcn_readlane(incl, 63);
    if (tot == 0) return;
    int basep = 0;
    if (lane == 0) basep = atomicAdd(q_count, back ? tot << 16 : tot);
    basep = __builtin_amdgcn_readfirstlane(basep);
    int pos = back ? qcap - 1 - ((basep >> 16) + incl - cnt) : (basep & 0xffff) + incl - cnt;
    const int dir = back ? -1 : 1;
    const unsigned qt = (unsigned)(qbase + tid) << 2;
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      queue[pos] = (unsigned short)(qt + ((unsigned)(bit & 7) << 10) + ((unsigned)bit >> 3));
      pos += dir;
    }
  };
  for (int qbase = 0; qbase < nq4; qbase += 2048) {
    unsigned acc_lo = 0, acc_hi = 0;  // bit 8 * p + pass: pixel p of this lane's quad in pass `pass` may be a corner
    const int npass = min(8, (nq4 - qbase + 255) >> 8);
#pragma unroll 1
    for (int pass = 0; pass < npass; pass++) {
      if (qbase + 256 * pass + tid < nq4) {
        const unsigned *tw = tile + ry * FC_TP + cq;
        // 12-byte windows (columns x-4 .. x+7) of the 7 rows y-3 .. y+3; x = xs0 + 4 cq
        unsigned w[7][3];
#pragma unroll
        for (int j = 0; j < 7; j++)
#pragma unroll
          for (int c = 0; c < 3; c++) w[j][c] = tw[j * FC_TP + c];
        unsigned lo_bytes[2], hi_bytes[2];
#pragma unroll
        for (int pr = 0; pr < 2; pr++) {
          u16x2 r[16];
          // ring (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3);
          // only the even positions are used by the quick test
          if (pr == 0) {
            r[0] = pair_bytes<4>(w[6]);  r[2] = pair_bytes<6>(w[5]);  r[4] = pair_bytes<7>(w[3]);  r[6] = pair_bytes<6>(w[1]);
            r[8] = pair_bytes<4>(w[0]);  r[10] = pair_bytes<2>(w[1]); r[12] = pair_bytes<1>(w[3]); r[14] = pair_bytes<2>(w[5]);
          } else {
            r[0] = pair_bytes<6>(w[6]);  r[2] = pair_bytes<8>(w[5]);  r[4] = pair_bytes<9>(w[3]);  r[6] = pair_bytes<8>(w[1]);
            r[8] = pair_bytes<6>(w[0]);  r[10] = pair_bytes<4>(w[1]); r[12] = pair_bytes<3>(w[3]); r[14] = pair_bytes<4>(w[5]);
          }
          const u16x2 v = pr == 0 ? pair_bytes<4>(w[3]) : pair_bytes<6>(w[3]);
          u16x2 mx_of_min = __builtin_elementwise_min(r[0], r[8]);
          u16x2 mn_of_max = __builtin_elementwise_max(r[0], r[8]);
#pragma unroll
          for (int k = 2; k < 8; k += 2) {
            mx_of_min = __builtin_elementwise_max(mx_of_min, __builtin_elementwise_min(r[k], r[k + 8]));
            mn_of_max = __builtin_elementwise_min(mn_of_max, __builtin_elementwise_max(r[k], r[k + 8]));
          }
          const u16x2 dk = __builtin_elementwise_sub_sat(v, mx_of_min);   // how far below the centre a dark arc can sit
          const u16x2 br = __builtin_elementwise_sub_sat(mn_of_max, v);   // how far above a bright one
          const u16x2 m = __builtin_elementwise_max(dk, br);
          lo_bytes[pr] = as_u32(m + bias_lo2);
          hi_bytes[pr] = as_u32(m + bias_hi2);
        }
        // bytes 1 and 3 of both results: 0 or 1 per pixel, pixel p in byte p
        const unsigned cm = colmask[cq];
        acc_lo |= (__builtin_amdgcn_perm(lo_bytes[1], lo_bytes[0], 0x07050301u) & cm) << pass;
        acc_hi |= (__builtin_amdgcn_perm(hi_bytes[1], hi_bytes[0], 0x07050301u) & cm) << pass;
      }
      cq += d_c;
      ry += d_r;
      if (cq >= ncg) {
        cq -= ncg;
        ry++;
      }
    }
    if (qbase == 0) keep_lo[0] = acc_lo & ~acc_hi;  // only at minThFAST
    else keep_lo[1] = acc_lo & ~acc_hi;
    push_block(acc_hi, qbase, false);
  }
  __syncthreads();
  const int n_hi = *q_count & 0xffff;
  const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
  uint8_t *sb = reinterpret_cast<uint8_t *>(stile);
  const int bnd0 = X0 - xs0;  // tile column - 4 of the first pixel of cell j0
  // full 9-arc score of queue entry q -> score tile; the entry becomes (row << 8 | column).  need_mask: cells whose
  // entries are scored (bit j), all of them for the front of the queue
  auto score_entry = [&](int q, unsigned need_mask) {
    const int e = queue[q];
    const int ey = div_small(e >> 2, ncg, inv_ncg), cx = 4 * ((e >> 2) - ey * ncg) + (e & 3);
    queue[q] = (unsigned short)((ey << 8) | cx);
    if (need_mask != 0xfu) {
      const int rel = cx - bnd0;
      const int j = (rel >= L.wcell) + (rel >= 2 * L.wcell) + (rel >= 3 * L.wcell);
      if (!((need_mask >> j) & 1u)) return;
    }
    const uint8_t *c = tb + (ey + 3) * FC_TPB + cx + 4;
    constexpr int RP = FC_TPB;
    int r[16];
    r[0] = c[3 * RP];       r[1] = c[3 * RP + 1];   r[2] = c[2 * RP + 2];   r[3] = c[RP + 3];
    r[4] = c[3];            r[5] = c[-RP + 3];      r[6] = c[-2 * RP + 2];  r[7] = c[-3 * RP + 1];
    r[8] = c[-3 * RP];      r[9] = c[-3 * RP - 1];  r[10] = c[-2 * RP - 2]; r[11] = c[-RP - 3];
    r[12] = c[-3];          r[13] = c[RP - 3];      r[14] = c[2 * RP - 2];  r[15] = c[3 * RP - 1];
    const int sc = fast_score16_full((int)c[0], r, t_lo);
    if (sc) {
      sb[(ey + 1) * FC_TPB + cx + 4] = (uint8_t)sc;
      if (DUMP) score_dump[(size_t)frame * g.frame_bytes + L.off + (size_t)(Y0 + ey) * L.pitch + (xs0 + cx)] = (uint8_t)sc;
    }
  };
  // 3x3 NMS of (decoded) queue entry q inside the pixel's own cell; survivors with score >= t_keep in the cells of
  // need_mask set their bit in the cell's row mask
  auto nms_entry = [&](int q, int t_keep, unsigned need_mask) {
    const int e = queue[q];
    const int ey = e >> 8, cx = e & 255;
    const uint8_t *c = sb + (ey + 1) * FC_TPB + cx + 4;
    const int v = c[0];
    if (v < t_keep) return;  // the score tile holds 0 below minThFAST
    const int rel = cx - bnd0;
    const int j = (rel >= L.wcell) + (rel >= 2 * L.wcell) + (rel >= 3 * L.wcell);
    if (!((need_mask >> j) & 1u)) return;
    const int xx = rel - j * L.wcell;  // column inside the cell
    int nl = vmax3(c[-FC_TPB - 1], c[-1], c[FC_TPB - 1]);
    int nr = vmax3(c[-FC_TPB + 1], c[1], c[FC_TPB + 1]);
    nl = xx == 0 ? 0 : nl;                  // "NMS does not see across the seam"
    nr = xx == L.wcell - 1 ? 0 : nr;        // (beyond X1 nothing was scored)
    const int nm = vmax3(c[-FC_TPB], c[FC_TPB], max(nl, nr));
    if (v > nm) atomicOr(rmask + ((ey * FC_MAXK + j) << 1) + (xx >> 5), 1u << (xx & 31));
  };
  // wave = cell, lane = row: survivors of the cell's row masks in raster order -> the cell's fixed slot of the
  // candidate pool; returns the number written (wave-uniform)
  auto emit_cell = [&](int j) -> int {
    const int cx0 = X0 + j * L.wcell;
    uint2 mk = make_uint2(0, 0);
    if (lane < nrow) mk = *reinterpret_cast<const uint2 *>(rmask + ((lane * FC_MAXK + j) << 1));
    const int cnt = __popc(mk.x) + __popc(mk.y);
    const int incl = wave_incl_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return 0;
    // every cell owns a fixed slot of ceil(wCell/2) * ceil(hCell/2) entries in its level's sub-pool: NMS survivors
    // are never adjacent, so the slot cannot overflow and no atomic is needed
    const int basei = (G.row * L.ncols + G.j0 + j) * cell_slot;
    if (lane == 0) ce[j] = CellEntry{basei, total};
    unsigned long long bits = ((unsigned long long)mk.y << 32) | mk.x;
    int pos = incl - cnt;
    unsigned *out = cands + (size_t)frame * g.cands_per_frame + L.cand_off + basei;
    const uint8_t *srow = sb + (lane + 1) * FC_TPB + 4 + (cx0 - xs0);
    const unsigned ybits = (unsigned)(Y0 + lane - ORB_MIN_BORDER) << 12;
    while (bits) {
      const int xx = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      out[pos++] = (unsigned)(cx0 + xx - ORB_MIN_BORDER) | ybits | ((unsigned)srow[xx] << 24);
    }
    return total;
  };

  // ---- iniThFAST: score the front of the queue, NMS, emit ---------------------------------------------------------
  for (int q = tid; q < n_hi; q += 256) score_entry(q, 0xfu);
  __syncthreads();
  for (int q = tid; q < n_hi; q += 256) nms_entry(q, t_hi, 0xfu);
  __syncthreads();
  if (wave < G.ncell) {
    // a cell without columns (beyond the last detection column) holds nothing at either threshold
    const bool has_cols = X0 + wave * L.wcell <= X1;
    const int total = has_cols ? emit_cell(wave) : 0;
    if (total == 0) {
      if (has_cols) {
        if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(q_count + 1), 1u << wave);  // falls back to minThFAST
      } else if (lane == 0) {
        ce[wave] = CellEntry{0, 0};
      }
    }
  }
  __syncthreads();
  // ---- minThFAST for the cells that kept nothing (orb_feature_extractor.cpp:263-266: decided per cell AFTER NMS) ----
  const unsigned fallback = (unsigned)q_count[1];
  const unsigned score_mask = DUMP ? 0xfu : fallback;
  if (score_mask == 0) return;
  push_block(keep_lo[0], 0, true);
  if (nq4 > 2048) push_block(keep_lo[1], 2048, true);
  __syncthreads();
  const int n_lo = *q_count >> 16;
  for (int q = qcap - 1 - tid; q >= qcap - n_lo; q -= 256) score_entry(q, score_mask);
  if (fallback == 0) return;
  __syncthreads();
  for (int q = tid; q < n_hi; q += 256) nms_entry(q, 1, fallback);
  for (int q = qcap - 1 - tid; q >= qcap - n_lo; q -= 256) nms_entry(q, 1, fallback);
  __syncthreads();
  if (wave < G.ncell && ((fallback >> wave) & 1u)) {
    if (emit_cell(wave) == 0 && lane == 0) ce[wave] = CellEntry{(G.row * L.ncols + G.j0 + wave) * cell_slot, 0};
  }
}

}  // namespace

size_t fast_cells_lds(int rows) { return fast_cells_lds_bytes(rows); }

// groups of up to FC_MAXK cells, <= 128 pixels wide, balanced over a cell row; host side of k_fast_cells
int build_fast_groups(const DeviceGeom &g, std::vector<short> &table) {
  table.clear();
  for (int l = 0; l < g.levels; l++) {
    const LevelGeom &L = g.lv[l];
    const int kmax = std::max(1, std::min(FC_MAXK, 128 / L.wcell));
    const int gcols = (L.ncols + kmax - 1) / kmax;
    for (int i = 0; i < L.nrows; i++)
      for (int q = 0; q < gcols; q++) {
        const int j0 = q * L.ncols / gcols, j1 = (q + 1) * L.ncols / gcols;
        table.push_back((short)l);
        table.push_back((short)i);
        table.push_back((short)j0);
        table.push_back((short)(j1 - j0));
      }
  }
  return (int)(table.size() / 4);
}

// part: -1 = every group in one launch, 0 = the groups of levels >= 1, 1 = the groups of level 0 (the table is level-major)
void launch_fast_cells(const orbgpu_ctx *c, FrameSlice fs, uint8_t *score_dump, int part) {
  const DeviceGeom &g = c->geom;
  const int first = part == 0 ? c->n_fast_groups_l0 : 0;
  const int count = part == 1 ? c->n_fast_groups_l0 : c->n_fast_groups - first;
  if (count <= 0) return;
  dim3 grid(count * xcd_frames(fs.n));
  const FastGroup *groups = reinterpret_cast<const FastGroup *>(c->d_fast_groups) + first;
  // dynamic LDS by the tallest cell of the launch's own levels: 32 rows (level 0 of a VGA frame) fit 8 workgroups per CU,
  // 33 rows and more only 7 -- and the kernel's time follows its occupancy
  int rows = 0;
  for (int l = (part == 0 ? 1 : 0); l < (part == 1 ? 1 : g.levels); l++) rows = std::max(rows, g.lv[l].hcell);
  const size_t lds = fast_cells_lds_bytes(rows);
  const uint8_t *pyr = c->d_pyr + (size_t)fs.f0 * g.frame_bytes;
  CellEntry *cells = c->d_cells + (size_t)fs.f0 * g.cells_per_frame;
  unsigned *cands = c->d_cands + (size_t)fs.f0 * g.cands_per_frame;
  if (score_dump)
    hipLaunchKernelGGL((k_fast_cells<true, 0>), grid, dim3(256), lds, fs.stream, pyr, g, groups, count, cells, cands,
                       fs.n, slice_lvl0(c, fs), rows, score_dump);
  else if (part == 1)
    hipLaunchKernelGGL((k_fast_cells<false, 1>), grid, dim3(256), lds, fs.stream, pyr, g, groups, count, cells, cands,
                       fs.n, slice_lvl0(c, fs), rows, (uint8_t *)nullptr);
  else
    hipLaunchKernelGGL((k_fast_cells<false, 0>), grid, dim3(256), lds, fs.stream, pyr, g, groups, count, cells, cands,
                       fs.n, slice_lvl0(c, fs), rows, (uint8_t *)nullptr);
}


