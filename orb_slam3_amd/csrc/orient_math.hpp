// Intensity-centroid orientation (IC_Angle, src/features/orb_feature_extractor.cpp:177-202; cv::fastAtan2 :201): the
// weight table, the arc tangent and the per-wave evaluation, shared by k_orient / k_patch_describe (k_describe.hip) and by
// the quadtree kernel's fused tail (k_quadtree.hip).  Each translation unit keeps its own __constant__ copy of the table
// (make_orient_tab) and stages it in LDS.  gfx950 only.
#pragma once

#include <cfloat>

#include "orbgpu_internal.h"
#include "tile_stage.hpp"

namespace orient {

// umax_, orb_feature_extractor.cpp:678-679
constexpr int UMAX[ORB_HALF_PATCH + 1] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// Orientation weights.  The 31 rows of the disc are read as 9 aligned dwords starting at xo = (cx-15) & ~3; with
// sh = (cx-15) - xo, byte b of dword c of row r is the pixel u = 4c + b - 15 - sh, v = r - 15.  Per (sh, r, c):
//   .x = bytes (u + 15) inside the disc, 0 outside   -> v_dot4_u32_u8 gives sum (u+15) I
//   .y = bytes 1 inside the disc, 0 outside           -> v_dot4_u32_u8 gives sum I
// so m10 = sum (u+15) I - 15 sum I and m01 = sum over rows of v * (row's sum I).
constexpr int ODW = 9;                   // aligned dwords covering a 31-byte orientation row
constexpr int OT_STRIDE = 64 * 5;        // entries per shift (31*9 = 279 used; lane + 63k indexing stays inside)
struct OrientTab { uint2 e[4 * OT_STRIDE]; };
constexpr OrientTab make_orient_tab() {
  OrientTab t{};
  for (int sh = 0; sh < 4; sh++)
    for (int r = 0; r < ORB_PATCH; r++)
      for (int c = 0; c < ODW; c++) {
        const int v = r - ORB_HALF_PATCH;
        const int um = UMAX[v < 0 ? -v : v];
        unsigned w = 0, m = 0;
        for (int b = 0; b < 4; b++) {
          const int u = 4 * c + b - ORB_HALF_PATCH - sh;
          if (u >= -um && u <= um) {
            w |= (unsigned)(u + ORB_HALF_PATCH) << (8 * b);
            m |= 1u << (8 * b);
          }
        }
        t.e[sh * OT_STRIDE + r * ODW + c] = uint2{w, m};
      }
  return t;
}

// cv::fastAtan2 scalar path: float32, no contraction, correctly rounded divide.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale;
  const float p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale;
  const float p7 = -0.04432655554792128f * scale;
  const float eps = (float)DBL_EPSILON;
  const float ax = fabsf(x), ay = fabsf(y);
  // the reference's two cases (ax >= ay: c = ay / (ax + eps); else c = ax / (ay + eps), a = 90 - poly) differ only in
  // which magnitude is divided by which: one division and one polynomial on the selected operands are the same
  // operations bit for bit (written as two branches the compiler evaluates both and selects)
  const bool xm = ax >= ay;
  const float c = __fdiv_rn(xm ? ay : ax, __fadd_rn(xm ? ax : ay, eps));
  const float c2 = __fmul_rn(c, c);
  const float poly = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  float a = xm ? poly : __fsub_rn(90.f, poly);
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}


// IC_Angle of NK keypoints side by side by ONE wave (NK times the loads in flight), in two steps so that a caller can
// put a barrier (the table's staging) between them.  lvl[h] / pitch[h] / lbytes[h]: the unblurred level of keypoint h,
// its row pitch and size; (cx, cy) and valid are wave-uniform.  31 rows x 9 aligned dwords, 7 rows = 63 lanes per step.
// Buffer loads: the level is the scalar base, the keypoint's patch origin and the row step of a load scalar offsets, the
// lane's place in a 7-row group one 32-bit register -- no 64-bit address arithmetic per load.  Lane 63, the rows below
// the disc (31 .. 34) and invalid keypoints read 0 through an offset / a range that lets nothing pass.
template <int NK>
__device__ __forceinline__ void orient_load(const uint8_t *const (&lvl)[NK], const int (&pitch)[NK], const int (&lbytes)[NK],
                                            const int (&cx)[NK], const int (&cy)[NK], const bool (&valid)[NK], int lane,
                                            unsigned (&ov)[NK][5]) {
  const int osub = lane / ODW, oc4 = 4 * (lane - osub * ODW);
#pragma unroll
  for (int h = 0; h < NK; h++) {
    const int xo = (cx[h] - ORB_HALF_PATCH) & ~3;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(lvl[h]), 0, valid[h] ? lbytes[h] : 0, 0x00020000);
    const unsigned voff = lane < 63 ? (unsigned)(osub * pitch[h] + oc4) : 0x80000000u;
    const unsigned sbase = (unsigned)((cy[h] - ORB_HALF_PATCH) * pitch[h] + xo);
    // (the last group holds rows 28 .. 34: the lanes of rows 31 .. 34 load nothing -- four cache lines per keypoint less)
    const unsigned voff4 = 7 * 4 + osub < ORB_PATCH ? voff : 0x80000000u;
#pragma unroll
    for (int k = 0; k < 5; k++)
      ov[h][k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, k == 4 ? voff4 : voff, sbase + (unsigned)(7 * k * pitch[h]), 0);
  }
}

// the moments of keypoint h from its loaded rows (wave-uniform results); s_tab: the weight table in LDS
__device__ __forceinline__ void orient_moments(const unsigned (&ov)[5], int cxh, const uint2 *s_tab, int lane, int &m10_out,
                                               int &m01_out) {
  const int osub = lane / ODW;
  const int osh = (cxh - ORB_HALF_PATCH) & 3;
  unsigned a10 = 0, si = 0;
  int m01 = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    // rows beyond the disc read weight 0, lane 63 pixel 0
    const uint2 ow = s_tab[osh * OT_STRIDE + 63 * k + min(lane, 62)];
    const unsigned rs = __builtin_amdgcn_udot4(ov[k], ow.y, 0u, false);
    a10 = __builtin_amdgcn_udot4(ov[k], ow.x, a10, false);
    si += rs;
    m01 += __mul24(7 * k + osub - ORB_HALF_PATCH, (int)rs);  // (rs <= 1020: one v_mad_i32_i24)
  }
  int m10 = (int)a10 - ORB_HALF_PATCH * (int)si;
  // the wave's sums end up in lane 63 (six DPP adds each; a ds_bpermute butterfly is 18 instructions and 6 LDS ops)
  m10_out = __builtin_amdgcn_readlane(wave_incl_scan(m10), 63);
  m01_out = __builtin_amdgcn_readlane(wave_incl_scan(m01), 63);
}

// the NK arc tangents side by side: lane h evaluates keypoint h (one fastAtan2 per wave instead of NK; the moments are
// wave-uniform).  Returns keypoint h's angle on lane h; an invalid keypoint's moments are 0 and so is its angle
template <int NK>
__device__ __forceinline__ float orient_atan_lanes(const int (&m10)[NK], const int (&m01)[NK], int lane) {
  float fx = 0.f, fy = 0.f;
#pragma unroll
  for (int h = 0; h < NK; h++) {
    fx = lane == h ? (float)m10[h] : fx;
    fy = lane == h ? (float)m01[h] : fy;
  }
  return fast_atan2_deg(fy, fx);
}

// angles from the loaded rows: keypoint h's on lane h
template <int NK>
__device__ __forceinline__ float orient_angles(const unsigned (&ov)[NK][5], const int (&cx)[NK], const bool (&valid)[NK],
                                               const uint2 *s_tab, int lane) {
  int m10[NK], m01[NK];
#pragma unroll
  for (int h = 0; h < NK; h++) {
    m10[h] = m01[h] = 0;
    if (valid[h]) orient_moments(ov[h], cx[h], s_tab, lane, m10[h], m01[h]);
  }
  return orient_atan_lanes<NK>(m10, m01, lane);
}

}  // namespace orient
