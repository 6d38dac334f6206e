// LDS staging of a 128-pixel-wide image tile with a 4-pixel column apron and a 3-row apron, used by the blur
// kernel.  One wave per staged row (4 waves -> rows w, w+4, ...), lane = aligned dword column
// (34 of 64 lanes), so the row index and its border handling are wave-uniform (scalar ALU) and the vector ALU only
// forms one address per row.  All loads of a wave are issued before the first LDS write.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

// XCD-aware work order for the (item, frame) grids.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs and
// each XCD has its own L2, so XCD k is given whole frames (k, k+8, ...) and walks each frame's items in order: tiles
// that share apron cache lines, and the cells / keypoints that re-read them, then meet in one L2.
// Launch a 1-D grid of items_per_frame * xcd_frames(n_frames) workgroups.
inline int xcd_frames(int n_frames) { return (n_frames + 7) & ~7; }
__device__ __forceinline__ bool xcd_item_frame(int items_per_frame, int n_frames, int &item, int &frame) {
  const unsigned id = blockIdx.x;
  const unsigned j = id >> 3;
  // j / items_per_frame from a float estimate and a two-sided fix-up: the quotient (a frame-group index) is far
  // below 2^22, so the estimate is off by at most one
  int fq = (int)((float)j * (1.0f / (float)items_per_frame));
  int it = (int)j - fq * items_per_frame;
  if (it < 0) { fq--; it += items_per_frame; }
  if (it >= items_per_frame) { fq++; it -= items_per_frame; }
  item = it;
  frame = fq * 8 + (int)(id & 7u);
  return frame < n_frames;
}

constexpr int STAGE_DW = 34;  // dwords per staged row: columns x0-4 .. x0+131

// BORDER_REFLECT_101 on rows and on the 3 columns next to each image edge (further columns of the apron are never used
// and hold unspecified image bytes).
template <int ROWS>
__device__ __forceinline__ void stage_tile(unsigned *__restrict__ tile, const uint8_t *__restrict__ src, int w, int h,
                                           int pitch, int x0, int y0) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (lane >= STAGE_DW) return;
  constexpr int NR = (ROWS + 3) / 4;
  const int gx = x0 - 4 + 4 * lane;
  unsigned v[NR];
  // tiles whose staged rows all lie inside the image (the common case) take a path without per-row border logic:
  // one row pointer, advanced by 4 rows per step
  const bool rows_inside = y0 >= 3 && y0 - 3 + ROWS <= h;  // wave-uniform
  {
    // wave-uniform: does any lane of this tile need a byte from outside [0, w)?
    const bool edge = x0 == 0 || x0 + 128 + 4 > w;
    if (!edge && rows_inside) {
      const uint8_t *row = src + (size_t)(y0 - 3 + wave) * pitch + gx;
#pragma unroll
      for (int k = 0; k < NR; k++) {
        v[k] = 0;
        if (wave + 4 * k < ROWS) v[k] = *reinterpret_cast<const unsigned *>(row + (size_t)(4 * k) * pitch);
      }
    } else if (!edge) {
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int r = wave + 4 * k;
        int gy = min(y0 - 3 + r, h + 2);
        gy = gy < 0 ? -gy : gy;
        gy = gy >= h ? 2 * (h - 1) - gy : gy;
        v[k] = 0;
        if (r < ROWS) v[k] = *reinterpret_cast<const unsigned *>(src + (size_t)gy * pitch + gx);
      }
    } else {
      // every staged dword is a byte permutation of two aligned source dwords A, B of the same row
      int off_a, off_b;
      unsigned sel;
      if (gx >= 0 && gx + 3 < w) {
        off_a = off_b = gx;
        sel = 0x03020100u;
      } else if (gx < 0) {  // columns -4..-1 <- 4 3 2 1 (column -4 is outside the 3-px apron)
        off_a = off_b = 0;
        sel = 0x01020300u;
      } else {  // columns >= w <- 2(w-1) - column; all sources lie in [w-4, w-1]
        const int a = (w - 4) & ~3;
        off_a = a;
        off_b = a + 4 < w ? a + 4 : a;
        sel = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          int col = gx + b;
          if (col >= w) col = 2 * (w - 1) - col;
          col = max(col, a);  // columns beyond the apron: unused, keep the selector in range
          sel |= (unsigned)(col - a) << (8 * b);
        }
      }
#pragma unroll
      for (int k = 0; k < NR; k++) {
        const int r = wave + 4 * k;
        int gy = min(y0 - 3 + r, h + 2);
        gy = gy < 0 ? -gy : gy;
        gy = gy >= h ? 2 * (h - 1) - gy : gy;
        v[k] = 0;
        if (r < ROWS) {
          const uint8_t *row = src + (size_t)gy * pitch;
          const unsigned da = *reinterpret_cast<const unsigned *>(row + off_a);
          const unsigned db = *reinterpret_cast<const unsigned *>(row + off_b);
          v[k] = __builtin_amdgcn_perm(db, da, sel);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NR; k++) {
    const int r = wave + 4 * k;
    if (r < ROWS) tile[r * STAGE_DW + lane] = v[k];
  }
}
