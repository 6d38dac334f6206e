// FAST-9/16 kernels: threshold-independent score map, per-cell NMS + threshold fallback + compaction.
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

constexpr int TILE_W = 128;
constexpr int TILE_DW = STAGE_DW;

__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }

// score of one pixel from its centre value and the 16 ring values (ring order of cv::FAST, App. A.2)
__device__ __forceinline__ int fast_score16(int v, const int (&r)[16], int min_thr) {
  int d[16];
#pragma unroll
  for (int k = 0; k < 16; k++) d[k] = v - r[k];
  // necessary condition for a 9-arc: every opposite pair (k, k+8) has one member inside the arc
  int lo = 255, hi = -255;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    lo = min(lo, max(d[k], d[k + 8]));   // dark arc needs all of these > t
    hi = max(hi, min(d[k], d[k + 8]));   // bright arc needs all of these < -t
  }
  if (lo <= min_thr && hi >= -min_thr) return 0;
  // sliding 9-window min / max over the circular ring: m3 then 3 x m3
  int mn3[16], mx3[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    mn3[k] = imin3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    mx3[k] = imax3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
  }
  int m_dark = -255, m_brightneg = 255;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    m_dark = max(m_dark, imin3(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
    m_brightneg = min(m_brightneg, imax3(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
  }
  int s = max(m_dark, -m_brightneg) - 1;
  return s >= min_thr ? s : 0;
}

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

// One launch covers all levels of all frames: blockIdx.x = tile over (level, ty, tx), blockIdx.y = frame.
// Tile 128 x 16 output pixels, staged with a 3-row / 4-column apron as dwords.
//   phase 1: every lane runs the cheap opposite-pair test on 2 rows x 2 pixel PAIRS in packed u16 math
//            (v_perm_b32 picks the two pixels' ring bytes, v_pk_min/max_u16 reduce them); ~13 % of the pixels of a
//            textured frame survive and are pushed on an LDS queue;
//   phase 2: the queue is drained densely: one lane per surviving pixel computes the full 9-arc score from the LDS tile;
//   phase 3: the score tile is written back with coalesced dword stores.
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

constexpr int FT_H = 32;                 // FAST tile: 128 x 32 pixels, 256 threads, 4 px x 4 rows per thread
constexpr int FT_ROWS = FT_H + 6;

__global__ __launch_bounds__(256) void k_fast_score(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ score,
                                                    unsigned *__restrict__ nzmask, DeviceGeom g, int n_frames, Level0Src lvl0) {
  __shared__ unsigned tile[FT_ROWS * TILE_DW];
  __shared__ unsigned out_tile[FT_H * TILE_W / 4];
  __shared__ unsigned short queue[FT_H * TILE_W];
  __shared__ unsigned rowmask[FT_H * 4];  // bit (cx & 31) of rowmask[ry*4 + cx/32]: score(cx, ry) != 0
  __shared__ int q_count;
  int t, frame, l = 0;
  if (!xcd_item_frame(g.ftiles_per_frame, n_frames, t, frame)) return;
#pragma unroll 1
  while (l + 1 < g.levels && t >= g.lv[l + 1].ftile_off) l++;
  const LevelGeom L = g.lv[l];
  t -= L.ftile_off;
  const int tiles_x = (L.w + TILE_W - 1) / TILE_W;
  const int x0 = (t % tiles_x) * TILE_W, y0 = (t / tiles_x) * FT_H;
  const size_t base = (size_t)frame * g.frame_bytes + L.off;
  const uint8_t *src = level_src(pyr, g, lvl0, frame, l, L.off);

  if (threadIdx.x == 0) q_count = 0;
  if (threadIdx.x < FT_H * 4) rowmask[threadIdx.x] = 0;
#pragma unroll
  for (int k = 0; k < FT_H * TILE_W / 4 / 256; k++) out_tile[threadIdx.x + 256 * k] = 0;
  // 38 x 34 dwords; pixels outside the image are never part of a scored ring
  stage_tile<FT_ROWS, false>(tile, src, L.w, L.h, L.pitch, x0, y0);
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63;
  const int min_thr = g.fast_min;
  const u16x2 thr2 = {(unsigned short)min_thr, (unsigned short)min_thr};
  // ---- phase 1: packed quick test -------------------------------------------------------------
  unsigned mask16 = 0;
#pragma unroll 1
  for (int part = 0; part < FT_H / 8; part++) {
    const int ry = ty + 8 * part;
    const int y = y0 + ry, x = x0 + 4 * tx;
    unsigned mask = 0;  // bit p: pixel x+p may be a corner
    if (y >= ORB_EDGE && y <= L.h - ORB_EDGE - 1) {  // only the cells' detection rectangles are ever read
      // 12-byte windows (columns x-4 .. x+7) of the 7 rows y-3 .. y+3
      unsigned w[7][3];
#pragma unroll
      for (int j = 0; j < 7; j++)
#pragma unroll
        for (int q = 0; q < 3; q++) w[j][q] = tile[(ry + j) * TILE_DW + tx + q];
#pragma unroll
      for (int pr = 0; pr < 2; pr++) {
        u16x2 r[16];
        // pair = pixels x+2pr, x+2pr+1
        // ring (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
        if (pr == 0) {
          r[0] = pair_bytes<4>(w[6]);  r[1] = pair_bytes<5>(w[6]);  r[2] = pair_bytes<6>(w[5]);  r[3] = pair_bytes<7>(w[4]);
          r[4] = pair_bytes<7>(w[3]);  r[5] = pair_bytes<7>(w[2]);  r[6] = pair_bytes<6>(w[1]);  r[7] = pair_bytes<5>(w[0]);
          r[8] = pair_bytes<4>(w[0]);  r[9] = pair_bytes<3>(w[0]);  r[10] = pair_bytes<2>(w[1]); r[11] = pair_bytes<1>(w[2]);
          r[12] = pair_bytes<1>(w[3]); r[13] = pair_bytes<1>(w[4]); r[14] = pair_bytes<2>(w[5]); r[15] = pair_bytes<3>(w[6]);
        } else {
          r[0] = pair_bytes<6>(w[6]);  r[1] = pair_bytes<7>(w[6]);  r[2] = pair_bytes<8>(w[5]);  r[3] = pair_bytes<9>(w[4]);
          r[4] = pair_bytes<9>(w[3]);  r[5] = pair_bytes<9>(w[2]);  r[6] = pair_bytes<8>(w[1]);  r[7] = pair_bytes<7>(w[0]);
          r[8] = pair_bytes<6>(w[0]);  r[9] = pair_bytes<5>(w[0]);  r[10] = pair_bytes<4>(w[1]); r[11] = pair_bytes<3>(w[2]);
          r[12] = pair_bytes<3>(w[3]); r[13] = pair_bytes<3>(w[4]); r[14] = pair_bytes<4>(w[5]); r[15] = pair_bytes<5>(w[6]);
        }
        const u16x2 v = pr == 0 ? pair_bytes<4>(w[3]) : pair_bytes<6>(w[3]);
        // a dark 9-arc needs min(ring_k, ring_k+8) < v - t for every opposite pair, a bright one max(...) > v + t.
        // Only the 4 even pairs are tested: measured on the bench frames, 8 pairs reject a little more (13 % survive)
        // but cost more than the extra full scores do (0.45 ms vs 0.39 ms per batch); 2 or 3 pairs are worse again
        u16x2 mx_of_min = __builtin_elementwise_min(r[0], r[8]);
        u16x2 mn_of_max = __builtin_elementwise_max(r[0], r[8]);
#pragma unroll
        for (int k = 2; k < 8; k += 2) {
          mx_of_min = __builtin_elementwise_max(mx_of_min, __builtin_elementwise_min(r[k], r[k + 8]));
          mn_of_max = __builtin_elementwise_min(mn_of_max, __builtin_elementwise_max(r[k], r[k + 8]));
        }
        // all values are 0..255 (+t): plain u16 arithmetic cannot wrap
        const u16x2 dark_lim = mx_of_min + thr2;   // candidate iff dark_lim < v
        const u16x2 bright_lim = v + thr2;         // candidate iff bright_lim < mn_of_max
        const unsigned c0 = (dark_lim.x < v.x) | (bright_lim.x < mn_of_max.x);
        const unsigned c1 = (dark_lim.y < v.y) | (bright_lim.y < mn_of_max.y);
        mask |= (c0 | (c1 << 1)) << (2 * pr);
      }
      // the cells' detection rectangles tile [19, w-20] x [19, h-20] (k_cell_detect); nothing outside is read
#pragma unroll
      for (int p = 0; p < 4; p++)
        if (x + p < ORB_EDGE || x + p > L.w - ORB_EDGE - 1) mask &= ~(1u << p);
    }
    mask16 |= mask << (4 * part);
  }
  // ---- push the survivors: one LDS atomic per wave; the order inside the queue does not matter -----
  {
    const int cnt = __popc(mask16);  // 0..16
    int before = 0;                  // survivors of lower lanes: sum over the bits of cnt of popc(ballot & below)
    int tot = 0;
#pragma unroll
    for (int b = 0; b < 5; b++) {
      const unsigned long long bal = __ballot((cnt >> b) & 1);
      before += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u)) << b;
      tot += __popcll(bal) << b;
    }
    if (tot) {
      int basep = 0;
      if (lane == 0) basep = atomicAdd(&q_count, tot);
      int pos = __shfl(basep, 0) + before;
      const int idx0 = ty * TILE_W + 4 * tx;  // bit 4*part + p  ->  pixel (ty + 8*part, 4*tx + p)
      unsigned m = mask16;
      while (m) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        queue[pos++] = (unsigned short)(idx0 + (bit >> 2) * (8 * TILE_W) + (bit & 3));
      }
    }
  }
  __syncthreads();
  // ---- phase 2: full score of the survivors ------------------------------------------------------
  const int nq = q_count;
  const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
  uint8_t *ob = reinterpret_cast<uint8_t *>(out_tile);
  for (int q = threadIdx.x; q < nq; q += 256) {
    const int idx = queue[q];
    const int ry = idx >> 7, cx = idx & 127;
    const uint8_t *c = tb + (ry + 3) * (TILE_DW * 4) + cx + 4;
    constexpr int RP = TILE_DW * 4;
    int r[16];
    r[0] = c[3 * RP];       r[1] = c[3 * RP + 1];   r[2] = c[2 * RP + 2];   r[3] = c[RP + 3];
    r[4] = c[3];            r[5] = c[-RP + 3];      r[6] = c[-2 * RP + 2];  r[7] = c[-3 * RP + 1];
    r[8] = c[-3 * RP];      r[9] = c[-3 * RP - 1];  r[10] = c[-2 * RP - 2]; r[11] = c[-RP - 3];
    r[12] = c[-3];          r[13] = c[RP - 3];      r[14] = c[2 * RP - 2];  r[15] = c[3 * RP - 1];
    const int sc = fast_score16_full((int)c[0], r, min_thr);
    ob[idx] = (uint8_t)sc;
    if (sc) atomicOr(&rowmask[ry * 4 + (cx >> 5)], 1u << (cx & 31));
  }
  __syncthreads();
  // ---- phase 3: coalesced write-back ---------------------------------------------------------------
#pragma unroll
  for (int part = 0; part < FT_H / 8; part++) {
    const int ry = ty + 8 * part;
    const int y = y0 + ry, x = x0 + 4 * tx;
    if (y >= L.h || x >= L.pitch) continue;
    *reinterpret_cast<unsigned *>(score + base + (size_t)y * L.pitch + x) = out_tile[ry * (TILE_W / 4) + tx];
  }
  // the non-zero bit plane: 4 dwords per tile row (k_cell_detect finds its queue entries from it)
  if (threadIdx.x < FT_H * 4 && y0 + (int)(threadIdx.x >> 2) < L.h)
    nzmask[(size_t)frame * g.mask_dwords_per_frame + L.mask_off + (size_t)(y0 + (threadIdx.x >> 2)) * L.mask_pitch + (x0 >> 5) +
           (threadIdx.x & 3)] = rowmask[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// One wave per FAST cell, four independent waves per workgroup.  The cell's detection rectangle (the pixels
// cv::FAST scans inside the reference's ROI: x in [19 + j*wCell, min(18 + (j+1)*wCell, w-20)], likewise y; SURVEY
// App. B.2) is staged in LDS with aligned dword loads (16 dwords x 4 rows per wave step, no integer division);
// bytes outside the rectangle are zeroed and neighbours outside it read as 0: "NMS does not see across the seam".
// The score map is ~92 % zeros, so the non-zero pixels are first compacted, in raster order, into an LDS queue and
// only those are run through the 3x3 NMS -- once, for both thresholds.  The threshold fallback is decided per cell
// AFTER NMS (orb_feature_extractor.cpp:263-266); survivors leave in raster order.
constexpr int CELL_MAX = 64;       // rectangle rows; columns must fit 16 aligned dwords (wCell <= 60)
constexpr int CELL_DW = 16;
constexpr int CELL_PITCH = CELL_DW * 4;
constexpr int CQ_CAP = 512;        // queue entries; denser cells take the division-based dense path
constexpr int CELLS_PER_WG = 4;
// per wave: (cell_rows + 3) staged rows (two zero rows above, one below), two ballot arrays, the queue
__host__ __device__ constexpr size_t cell_lds_bytes(int cell_rows) { return (size_t)(cell_rows + 3) * CELL_PITCH + (size_t)cell_rows * 16 + CQ_CAP * 2; }

__global__ __launch_bounds__(64 * CELLS_PER_WG) void k_cell_detect(const uint8_t *__restrict__ score,
                                                                   const unsigned *__restrict__ nzmask, DeviceGeom g,
                                                                   CellEntry *__restrict__ cells,
                                                                   unsigned *__restrict__ cands, int n_frames,
                                                                   int cell_rows) {
  // dynamic LDS, per wave: score rectangle [cell_rows + 3][16 dwords] | NMS ballots hi, lo [cell_rows] | queue [CQ_CAP]
  extern __shared__ __attribute__((aligned(16))) unsigned char cd_lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  unsigned char *wbase = cd_lds + (size_t)wave * cell_lds_bytes(cell_rows);
  unsigned *s32 = reinterpret_cast<unsigned *>(wbase);
  const uint8_t *sc = reinterpret_cast<const uint8_t *>(s32) + 2 * CELL_PITCH;  // rectangle row 0, staged column 0
  unsigned long long *m_hi = reinterpret_cast<unsigned long long *>(wbase + (size_t)(cell_rows + 3) * CELL_PITCH);
  unsigned long long *m_lo = m_hi + cell_rows;
  unsigned short *queue = reinterpret_cast<unsigned short *>(m_lo + cell_rows);
  int f, c, l = 0;
  if (!xcd_item_frame((g.cells_per_frame + CELLS_PER_WG - 1) / CELLS_PER_WG, n_frames, c, f)) return;
  c = c * CELLS_PER_WG + wave;  // wave-uniform: the cell geometry below runs on the scalar ALU
  if (c >= g.cells_per_frame) return;
#pragma unroll 1
  while (l + 1 < g.levels && c >= g.lv[l + 1].cell_off) l++;
  const LevelGeom L = g.lv[l];
  const int ci = c - L.cell_off;
  int i = (int)(((float)ci + 0.5f) * (1.0f / (float)L.ncols));  // ci / ncols, exact for ci < 2^20
  i = __builtin_amdgcn_readfirstlane(i);
  const int j = ci - i * L.ncols;
  const int x0 = ORB_EDGE + j * L.wcell, x1 = min(ORB_EDGE - 1 + (j + 1) * L.wcell, L.w - ORB_EDGE - 1);
  const int y0 = ORB_EDGE + i * L.hcell, y1 = min(ORB_EDGE - 1 + (i + 1) * L.hcell, L.h - ORB_EDGE - 1);
  const int rw = x1 - x0 + 1, rh = y1 - y0 + 1;
  CellEntry *ce = cells + (size_t)f * g.cells_per_frame + c;
  if (rw <= 0 || rh <= 0) {
    if (lane == 0) *ce = CellEntry{0, 0};
    return;
  }
  const uint8_t *src = score + (size_t)f * g.frame_bytes + L.off;
  const int xa = x0 & ~3;          // staged byte column cx of level x is x - xa
  const int cxa = x0 - xa;         // first rectangle column, 0..3
  const int ndwr = (cxa + rw + 3) >> 2;
  const int dd = lane & 15, rsub = lane >> 4;
  const unsigned long long below = (1ull << lane) - 1ull;
  // byte mask of this lane's dword column
  unsigned bmask = 0;
#pragma unroll
  for (int b = 0; b < 4; b++)
    if (4 * dd + b >= cxa && 4 * dd + b < cxa + rw) bmask |= 0xffu << (8 * b);

  // ---- stage the rectangle (4 rows per step, all loads in flight) ------------------------------------------
  // no per-lane predication: rows / dwords beyond the rectangle load a clamped (in-bounds) address and are masked to 0
  unsigned w[16];  // rh <= cell_rows <= 64 = 16 steps of 4 rows
  const uint8_t *lsrc = src + (size_t)y0 * L.pitch + xa + 4 * min(dd, ndwr - 1);
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int r = 4 * u + rsub;
    w[u] = 0;
    if (4 * u < cell_rows)  // kernel-uniform
      w[u] = *reinterpret_cast<const unsigned *>(lsrc + (size_t)min(r, rh - 1) * L.pitch) & (r < rh ? bmask : 0u);
  }
  // ---- queue of the non-zero pixels in raster order, from the bit plane k_fast_score wrote: lane = row ---------
  unsigned long long rowbits = 0;
  if (lane < rh) {
    const unsigned *mrow = nzmask + (size_t)f * g.mask_dwords_per_frame + L.mask_off + (size_t)(y0 + lane) * L.mask_pitch;
    const int d0 = x0 >> 5, sft = x0 & 31;
    const unsigned m0 = mrow[d0];
    const unsigned m1 = d0 + 1 < L.mask_pitch ? mrow[d0 + 1] : 0u;
    const unsigned m2 = d0 + 2 < L.mask_pitch ? mrow[d0 + 2] : 0u;
    const unsigned long long lo = ((unsigned long long)m1 << 32) | m0;
    rowbits = sft ? (lo >> sft) | ((unsigned long long)m2 << (64 - sft)) : lo;
    rowbits &= (rw >= 64) ? ~0ull : ((1ull << rw) - 1ull);
  }
  const int cnt = __popcll(rowbits);
  int incl = cnt;  // inclusive scan over the rows (lanes)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  const int nq = __shfl(incl, 63);
#pragma unroll
  for (int u = 0; u < 16; u++) {
    const int r = 4 * u + rsub;
    if (4 * u < cell_rows) s32[(r + 2) * CELL_DW + dd] = w[u];  // rows rh .. cell_rows-1 receive zeros
  }
  if (lane < 2 * CELL_DW) s32[lane] = 0;                             // the two zero rows above the rectangle
  else if (lane < 3 * CELL_DW) s32[(rh + 2) * CELL_DW + (lane - 2 * CELL_DW)] = 0;  // and the one below
  if (nq <= CQ_CAP) {
    int pos = incl - cnt;
    unsigned long long bits = rowbits;
    while (bits) {
      const int bpos = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      queue[pos++] = (unsigned short)((lane << 6) | (cxa + bpos));
    }
  }
  __builtin_amdgcn_wave_barrier();  // LDS traffic of one wave is processed in order
  const bool dense = nq > CQ_CAP;   // queue overflow: walk every pixel instead (k -> row, column by division)
  const int n_ent = dense ? rw * rh : nq;
  const int nchunks = (n_ent + 63) / 64;
  const int t_hi = g.fast_init, t_lo = g.fast_min;

  auto entry = [&](int k, int &yy, int &cx) {
    if (dense) {
      yy = k / rw;
      cx = cxa + (k - yy * rw);
    } else {
      const int e = queue[k];
      yy = e >> 6;
      cx = e & 63;
    }
  };
  // the rectangle sits between zero rows (two above, one below) and every staged row ends in zero bytes (columns
  // >= cxa + rw <= 63), so the 8 neighbours are read without bounds checks: outside the rectangle they are 0
  int tot_hi = 0, tot_lo = 0;
  for (int ch = 0; ch < nchunks; ch++) {
    const int k = ch * 64 + lane;
    bool kh = false, kl = false;
    if (k < n_ent) {
      int yy, cx;
      entry(k, yy, cx);
      const uint8_t *c = sc + yy * CELL_PITCH + cx;
      const int v = c[0];
      // 3x3 non-maximum suppression.  One sweep serves both thresholds: for v >= t_hi, "greater than every
      // neighbour that is itself >= t_hi" and "greater than every neighbour" are the same condition
      const int n0 = vmax3(c[-CELL_PITCH - 1], c[-CELL_PITCH], c[-CELL_PITCH + 1]);
      const int n1 = vmax3(c[-1], c[1], c[CELL_PITCH - 1]);
      const int n2 = vmax3(c[CELL_PITCH], c[CELL_PITCH + 1], n0);
      kl = v >= t_lo && v > max(n1, n2);  // the score map holds 0 below fast_min
      kh = kl && v >= t_hi;
    }
    const unsigned long long bh = __ballot(kh), bl = __ballot(kl);
    tot_hi += __popcll(bh);
    tot_lo += __popcll(bl);
    if (lane == 0) {
      m_hi[ch] = bh;
      m_lo[ch] = bl;
    }
  }
  const bool use_hi = tot_hi > 0;
  const int total = use_hi ? tot_hi : tot_lo;
  // every cell owns a fixed slot of ceil(wCell/2) * ceil(hCell/2) entries in its level's sub-pool: NMS survivors
  // are never adjacent, so the slot cannot overflow and no atomic is needed
  const int basei = ci * (((L.wcell + 1) >> 1) * ((L.hcell + 1) >> 1));
  if (lane == 0) *ce = CellEntry{basei, total};
  if (total == 0) return;
  __builtin_amdgcn_wave_barrier();
  unsigned *out = cands + (size_t)f * g.cands_per_frame + L.cand_off + basei;
  const unsigned long long *mm = use_hi ? m_hi : m_lo;
  int run = 0;
  for (int ch = 0; ch < nchunks; ch++) {
    const unsigned long long m = mm[ch];
    if (m == 0) continue;
    if ((m >> lane) & 1ull) {
      int yy, cx;
      entry(ch * 64 + lane, yy, cx);
      out[run + __popcll(m & below)] =
          pack_cand(x0 + (cx - cxa) - ORB_MIN_BORDER, y0 + yy - ORB_MIN_BORDER, sc[yy * CELL_PITCH + cx]);
    }
    run += __popcll(m);
  }
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
constexpr int FC_QPITCH = 136;          // queue entries per row (<= 33 quads x 4 pixels)
constexpr int FC_MAXROWS = 64;          // rows of a cell (hcell <= 64 is checked at create time)

__host__ __device__ constexpr size_t fast_cells_lds_bytes(int rows) {
  return (size_t)(rows + 6) * FC_TPB       // pixel tile
         + (size_t)(rows + 2) * FC_TPB     // score tile
         + (size_t)rows * FC_QPITCH * 2    // survivor queue (u16)
         + (size_t)rows * FC_MAXK * 16     // row masks per (row, cell): kept (u64), kept-hi (u64)
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

template <bool DUMP>
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ pyr, DeviceGeom g,
                                                    const FastGroup *__restrict__ groups, int groups_per_frame,
                                                    CellEntry *__restrict__ cells, unsigned *__restrict__ cands,
                                                    int n_frames, Level0Src lvl0, int lds_rows,
                                                    uint8_t *__restrict__ score_dump) {
  extern __shared__ __attribute__((aligned(16))) unsigned fc_lds[];
  unsigned *tile = fc_lds;
  unsigned *stile = tile + (lds_rows + 6) * FC_TP;
  unsigned short *queue = reinterpret_cast<unsigned short *>(stile + (lds_rows + 2) * FC_TP);
  unsigned *rmask = reinterpret_cast<unsigned *>(queue + lds_rows * FC_QPITCH);  // [row][cell]{lo.x, lo.y, hi.x, hi.y}
  unsigned *colmask = rmask + lds_rows * FC_MAXK * 4;
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
    const uint8_t *lsrc = src + 4 * lane;
    const bool on = lane < ndw;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const int r = wave + 4 * k;  // wave-uniform
      v[k] = 0;
      if (r < nst && on) v[k] = *reinterpret_cast<const unsigned *>(lsrc + (size_t)r * L.pitch);
    }
    // meanwhile: clear the score tile and the row masks (16 bytes per lane and store), per-quad column masks
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < (nrow + 2) * (FC_TP / 4); i += 256) reinterpret_cast<uint4 *>(stile)[i] = z4;
    if (tid < nrow * FC_MAXK) reinterpret_cast<uint4 *>(rmask)[tid] = z4;
    if (tid == 0) *q_count = 0;
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
      if (r < nst && on) tile[r * FC_TP + lane] = v[k];
    }
  }
  __syncthreads();

  const int t_lo = g.fast_min, t_hi = g.fast_init;
  const float inv_ncg = 1.0f / (float)ncg;
  // candidate <=> max(v - max_k min(pair_k), min_k max(pair_k) - v) > t: with saturating u16 arithmetic the sum
  // max(...) + (255 - t) has bit 8 set exactly then
  const unsigned short bias = (unsigned short)(255 - t_lo);
  const u16x2 bias2 = {bias, bias};
  // the lane's quad of pass k is q = tid + 256 k = (row ry, quad column cq), advanced by (d_r, d_c) per pass
  int ry = div_small(tid, ncg, inv_ncg), cq = tid - ry * ncg;
  const int d_r = __builtin_amdgcn_readfirstlane(div_small(256, ncg, inv_ncg)), d_c = 256 - d_r * ncg;
  // ---- phase 1 + push, 8 passes (2048 quads) at a time ------------------------------------------------------
  for (int qbase = 0; qbase < nq4; qbase += 2048) {
    unsigned acc = 0;  // bit 8 * p + pass: pixel p of this lane's quad in pass `pass` may be a corner
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
        unsigned hi_bytes[2];
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
          hi_bytes[pr] = as_u32(__builtin_elementwise_max(dk, br) + bias2);
        }
        // bytes 1 and 3 of both results: 0 or 1 per pixel, pixel p in byte p
        const unsigned flags = __builtin_amdgcn_perm(hi_bytes[1], hi_bytes[0], 0x07050301u) & colmask[cq];
        acc |= flags << pass;
      }
      cq += d_c;
      ry += d_r;
      if (cq >= ncg) {
        cq -= ncg;
        ry++;
      }
    }
    // ---- push the survivors as (quad << 2 | pixel): one LDS atomic per wave; the order inside the queue is free ----
    {
      const int cnt = __popc(acc);  // 0..32
      const int incl = wave_incl_scan(cnt);
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      if (tot) {
        int basep = 0;
        if (lane == 0) basep = atomicAdd(q_count, tot);
        int pos = __builtin_amdgcn_readfirstlane(basep) + incl - cnt;
        const unsigned qt = (unsigned)(qbase + tid) << 2;
        unsigned m = acc;
        while (m) {
          const int bit = __ffs(m) - 1;
          m &= m - 1;
          queue[pos++] = (unsigned short)(qt + ((unsigned)(bit & 7) << 10) + ((unsigned)bit >> 3));
        }
      }
    }
  }
  __syncthreads();
  // ---- phase 2: full score of the survivors -> score tile; the queue entry becomes (row << 8 | column) ------------
  const int nq = *q_count;
  const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
  uint8_t *sb = reinterpret_cast<uint8_t *>(stile);
  for (int q = tid; q < nq; q += 256) {
    const int e = queue[q];
    const int ey = div_small(e >> 2, ncg, inv_ncg), cx = 4 * ((e >> 2) - ey * ncg) + (e & 3);
    const uint8_t *c = tb + (ey + 3) * FC_TPB + cx + 4;
    constexpr int RP = FC_TPB;
    int r[16];
    r[0] = c[3 * RP];       r[1] = c[3 * RP + 1];   r[2] = c[2 * RP + 2];   r[3] = c[RP + 3];
    r[4] = c[3];            r[5] = c[-RP + 3];      r[6] = c[-2 * RP + 2];  r[7] = c[-3 * RP + 1];
    r[8] = c[-3 * RP];      r[9] = c[-3 * RP - 1];  r[10] = c[-2 * RP - 2]; r[11] = c[-RP - 3];
    r[12] = c[-3];          r[13] = c[RP - 3];      r[14] = c[2 * RP - 2];  r[15] = c[3 * RP - 1];
    const int sc = fast_score16_full((int)c[0], r, t_lo);
    queue[q] = (unsigned short)((ey << 8) | cx);
    if (sc) {
      sb[(ey + 1) * FC_TPB + cx + 4] = (uint8_t)sc;
      if (DUMP) score_dump[(size_t)frame * g.frame_bytes + L.off + (size_t)(Y0 + ey) * L.pitch + (xs0 + cx)] = (uint8_t)sc;
    }
  }
  __syncthreads();
  // ---- phase 3: 3x3 NMS inside the pixel's own cell; one sweep serves both thresholds ------------------------------
  const int bnd0 = X0 - xs0;  // tile column - 4 of the first pixel of cell j0
  for (int q = tid; q < nq; q += 256) {
    const int e = queue[q];
    const int ey = e >> 8, cx = e & 255;
    const uint8_t *c = sb + (ey + 1) * FC_TPB + cx + 4;
    const int v = c[0];
    if (v == 0) continue;
    const int rel = cx - bnd0;
    const int j = (rel >= L.wcell) + (rel >= 2 * L.wcell) + (rel >= 3 * L.wcell);
    const int xx = rel - j * L.wcell;  // column inside the cell
    int nl = vmax3(c[-FC_TPB - 1], c[-1], c[FC_TPB - 1]);
    int nr = vmax3(c[-FC_TPB + 1], c[1], c[FC_TPB + 1]);
    nl = xx == 0 ? 0 : nl;                  // "NMS does not see across the seam"
    nr = xx == L.wcell - 1 ? 0 : nr;        // (beyond X1 nothing was scored)
    const int nm = vmax3(c[-FC_TPB], c[FC_TPB], max(nl, nr));
    if (v > nm) {                           // the score tile holds 0 below fast_min
      unsigned *mrow = rmask + ((ey * FC_MAXK + j) << 2) + (xx >> 5);
      atomicOr(mrow, 1u << (xx & 31));
      if (v >= t_hi) atomicOr(mrow + 2, 1u << (xx & 31));
    }
  }
  __syncthreads();
  // ---- phase 4: wave = cell, lane = row; fallback decided per cell AFTER NMS (orb_feature_extractor.cpp:263-266) ----
  if (wave >= G.ncell) return;
  {
    const int j = wave;
    const int cx0 = X0 + j * L.wcell, cx1 = min(cx0 + L.wcell - 1, X1);
    const int ci = G.row * L.ncols + G.j0 + j;
    if (cx1 < cx0) {
      if (lane == 0) ce[j] = CellEntry{0, 0};
      return;
    }
    uint4 mk = make_uint4(0, 0, 0, 0);
    if (lane < nrow) mk = reinterpret_cast<const uint4 *>(rmask)[lane * FC_MAXK + j];
    const int c_lo = __popc(mk.x) + __popc(mk.y), c_hi = __popc(mk.z) + __popc(mk.w);
    const int cnt2 = (c_hi << 16) | c_lo;  // both scans at once (a cell holds < 2^12 survivors)
    const int incl = wave_incl_scan(cnt2);
    const int tot = __builtin_amdgcn_readlane(incl, 63);
    const bool use_hi = (tot >> 16) > 0;
    const int total = use_hi ? (tot >> 16) : (tot & 0xffff);
    // every cell owns a fixed slot of ceil(wCell/2) * ceil(hCell/2) entries in its level's sub-pool: NMS survivors
    // are never adjacent, so the slot cannot overflow and no atomic is needed
    const int basei = ci * cell_slot;
    if (lane == 0) ce[j] = CellEntry{basei, total};
    if (total == 0) return;
    unsigned long long bits = use_hi ? (((unsigned long long)mk.w << 32) | mk.z) : (((unsigned long long)mk.y << 32) | mk.x);
    int pos = use_hi ? ((incl - cnt2) >> 16) : ((incl - cnt2) & 0xffff);
    unsigned *out = cands + (size_t)frame * g.cands_per_frame + L.cand_off + basei;
    const uint8_t *srow = sb + (lane + 1) * FC_TPB + 4 + (cx0 - xs0);
    const unsigned ybits = (unsigned)(Y0 + lane - ORB_MIN_BORDER) << 12;
    while (bits) {
      const int xx = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      out[pos++] = (unsigned)(cx0 + xx - ORB_MIN_BORDER) | ybits | ((unsigned)srow[xx] << 24);
    }
  }
}

}  // namespace

void launch_fast_score(const orbgpu_ctx *c, int n_frames) {
  dim3 grid(c->geom.ftiles_per_frame * xcd_frames(n_frames));
  hipLaunchKernelGGL(k_fast_score, grid, dim3(256), 0, c->stream, c->d_pyr, c->d_score, c->d_nzmask, c->geom, n_frames, c->lvl0);
}

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

void launch_fast_cells(const orbgpu_ctx *c, int n_frames, uint8_t *score_dump) {
  dim3 grid(c->n_fast_groups * xcd_frames(n_frames));
  const FastGroup *groups = reinterpret_cast<const FastGroup *>(c->d_fast_groups);
  const size_t lds = fast_cells_lds_bytes(c->fast_lds_rows);
  if (score_dump)
    hipLaunchKernelGGL(k_fast_cells<true>, grid, dim3(256), lds, c->stream, c->d_pyr, c->geom, groups, c->n_fast_groups,
                       c->d_cells, c->d_cands, n_frames, c->lvl0, c->fast_lds_rows, score_dump);
  else
    hipLaunchKernelGGL(k_fast_cells<false>, grid, dim3(256), lds, c->stream, c->d_pyr, c->geom, groups, c->n_fast_groups,
                       c->d_cells, c->d_cands, n_frames, c->lvl0, c->fast_lds_rows, (uint8_t *)nullptr);
}

void launch_cell_detect(const orbgpu_ctx *c, int n_frames) {
  int cell_rows = 1;
  for (int l = 0; l < c->geom.levels; l++) cell_rows = std::max(cell_rows, c->geom.lv[l].hcell);
  cell_rows = std::min((cell_rows + 3) & ~3, CELL_MAX);  // hcell <= 64 is checked at create time
  dim3 grid((c->geom.cells_per_frame + CELLS_PER_WG - 1) / CELLS_PER_WG * xcd_frames(n_frames));
  hipLaunchKernelGGL(k_cell_detect, grid, dim3(64 * CELLS_PER_WG), CELLS_PER_WG * cell_lds_bytes(cell_rows), c->stream,
                     c->d_score, c->d_nzmask, c->geom, c->d_cells, c->d_cands, n_frames, cell_rows);
}
