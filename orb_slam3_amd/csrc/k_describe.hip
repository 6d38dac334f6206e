// Intensity-centroid orientation + 256-bit rotated BRIEF, one wave per keypoint, and the final
// scale / pack / reverse-order epilogue.  gfx950 only.
//
// Reference stages: IC_Angle            src/features/orb_feature_extractor.cpp:177-202 (cv::fastAtan2 :201)
//                   computeOrbDescriptor src/features/orb_feature_extractor.cpp:120-165 (pattern_ :681-852)
//                   Extract epilogue     src/features/orb_feature_extractor.cpp:541-599 (reversed packing)
#include <cfloat>

#include "orbgpu_internal.h"

namespace {

__constant__ signed char c_pattern[256 * 4] = {
#include "orb_pattern.inc"
};

// umax_, orb_feature_extractor.cpp:678-679
__constant__ int c_umax[ORB_HALF_PATCH + 1] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// cv::fastAtan2 scalar path: float32, no contraction, correctly rounded divide.
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale;
  const float p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale;
  const float p7 = -0.04432655554792128f * scale;
  const float eps = (float)DBL_EPSILON;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f,
                  __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

constexpr int PR = 19;              // patch radius staged for the descriptor (taps reach <= 18.4 px)
constexpr int PW = 2 * PR + 1;      // 39
constexpr int PPITCH = 40;

// blockDim = 256 = 4 waves = 4 keypoint slots; grid.x = ceil(kps_per_frame / 4), grid.y = frame
__global__ __launch_bounds__(256) void k_orient_describe(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur,
                                                         DeviceGeom g, const short2 *__restrict__ kp_xy,
                                                         const int *__restrict__ kp_count,
                                                         orbgpu_keypoint *__restrict__ out_kp,
                                                         uint8_t *__restrict__ out_desc, int capacity,
                                                         int *__restrict__ out_n) {
  __shared__ uint8_t patch[4][PW * PPITCH];
  const int f = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + wave;
  const int *cnt = kp_count + f * g.levels;
  // total and per-level prefix (<= 16 levels; every lane computes the same scalars)
  int total = 0;
  for (int l = 0; l < g.levels; l++) total += cnt[l];
  if (blockIdx.x == 0 && threadIdx.x == 0) out_n[f] = total;
  // validity is wave-uniform; invalid waves fall through to the barrier and write nothing
  bool valid = slot < g.kps_per_frame && total <= capacity;
  int l = 0;
#pragma unroll 1
  while (l + 1 < g.levels && slot >= g.lv[l + 1].kp_off) l++;
  const LevelGeom L = g.lv[l];
  const int i = slot - L.kp_off;
  valid = valid && i < cnt[l];
  int before = 0;
  for (int q = 0; q < l; q++) before += cnt[q];
  const int pos = total - 1 - (before + i);  // reversed packing, SURVEY App. B.4
  if (!valid) {
    __syncthreads();
    return;
  }

  const short2 xy = kp_xy[(size_t)f * g.kps_per_frame + slot];
  const int cx = xy.x, cy = xy.y;
  const size_t base = (size_t)f * g.frame_bytes + L.off;

  // ---- orientation on the unblurred level: lanes 0..30 = columns u = -15..15 of rows -15..0,
  //      lanes 32..62 = the same columns of rows 1..15; every row read is 31 contiguous bytes.
  int m10 = 0, m01 = 0;
  {
    const int u = (lane & 31) - ORB_HALF_PATCH;
    const int hi = lane >> 5;
    const uint8_t *center = pyr + base + (size_t)cy * L.pitch + cx;
    if ((lane & 31) < ORB_PATCH) {
      const int au = u < 0 ? -u : u;
      for (int k = 0; k < 16; k++) {
        int v = hi ? (k + 1) : (k - ORB_HALF_PATCH);  // rows 1..16 (16 is masked) / -15..0
        if (v > ORB_HALF_PATCH) break;
        int av = v < 0 ? -v : v;
        if (au <= c_umax[av]) {
          int val = center[v * L.pitch + u];
          m10 += u * val;
          m01 += v * val;
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      m10 += __shfl_xor(m10, off);
      m01 += __shfl_xor(m01, off);
    }
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);

  // ---- stage the blurred patch (39 x 39) in LDS, coalesced row reads
  uint8_t *P = patch[wave];
  {
    const uint8_t *src = blur + base + (size_t)(cy - PR) * L.pitch + (cx - PR);
    for (int k = lane; k < PW * PPITCH; k += 64) {
      int r = k / PPITCH, c = k - r * PPITCH;
      P[k] = c < PW ? src[(size_t)r * L.pitch + c] : 0;
    }
  }
  // ---- rotated BRIEF: float(angle_deg) * (double)(pi/180) -> float; cos/sin evaluated in double and
  //      rounded to float; taps = cvRound(x*b + y*a), cvRound(x*a - y*b), no FMA.
  const double factorPI = 3.14159265358979323846 / 180.;
  const float ang = (float)((double)angle * factorPI);
  const float a = (float)cos((double)ang), b = (float)sin((double)ang);
  __syncthreads();
  unsigned long long bits[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int t = 64 * q + lane;
    const float x0 = (float)c_pattern[4 * t + 0], y0 = (float)c_pattern[4 * t + 1];
    const float x1 = (float)c_pattern[4 * t + 2], y1 = (float)c_pattern[4 * t + 3];
    const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
    const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
    const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
    const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
    const int t0 = P[(PR + r0) * PPITCH + PR + c0];
    const int t1 = P[(PR + r1) * PPITCH + PR + c1];
    bits[q] = __ballot(t0 < t1);  // bit k of byte i = test 8i + k: ballot order is the descriptor's byte order
  }
  if (lane == 0) {
    unsigned long long *d = reinterpret_cast<unsigned long long *>(out_desc + ((size_t)f * capacity + pos) * 32);
    d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
    orbgpu_keypoint k;
    k.x = (double)cx;
    k.y = (double)cy;
    if (l != 0) {  // keypoint->pt *= (float)scale_factors_[level], orb_feature_extractor.cpp:568-577
      k.x *= (double)L.scale_f;
      k.y *= (double)L.scale_f;
    }
    k.angle = angle;
    k.size = L.kp_size;
    k.level = l;
    k.reserved = 0;
    out_kp[(size_t)f * capacity + pos] = k;
  }
}

}  // namespace

void launch_orient_describe(const orbgpu_ctx *c, int n_frames, orbgpu_keypoint *d_kp, uint8_t *d_desc, int capacity,
                            int *d_n_out) {
  dim3 grid((c->geom.kps_per_frame + 3) / 4, n_frames);
  hipLaunchKernelGGL(k_orient_describe, grid, dim3(256), 0, c->stream, c->d_pyr, c->d_blur, c->geom, c->d_kp_xy,
                     c->d_kp_count, d_kp, d_desc, capacity, d_n_out);
}
