// Work order shared by the per-frame kernels.
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
  // (v_rcp_f32, 1 ulp, instead of the 12-instruction IEEE division: the fix-up below absorbs it)
  int fq = (int)((float)j * __builtin_amdgcn_rcpf((float)items_per_frame));
  int it = (int)j - fq * items_per_frame;
  if (it < 0) { fq--; it += items_per_frame; }
  if (it >= items_per_frame) { fq++; it -= items_per_frame; }
  item = it;
  frame = fq * 8 + (int)(id & 7u);
  return frame < n_frames;
}

// inclusive prefix sum over the 64 lanes of a wave: four row_shr steps inside each 16-lane row, then the row totals
// travel with row_bcast:15 / row_bcast:31 (lanes without a source add 0)
// Written out: from update_dpp + add the compiler makes v_mov 0 / v_mov_dpp / v_add, three instructions a step; the add
// takes the shifted operand itself (a lane without a source is not written and keeps its own value).  s_nop 1: the two
// wait states between a vector write and a DPP read of the same register.
__device__ __forceinline__ int wave_incl_scan(int v) {
  asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(v));
  return v;
}
