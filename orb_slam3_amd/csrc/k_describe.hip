// Intensity-centroid orientation + 256-bit rotated BRIEF, one wave per keypoint, and the final
// scale / pack / reverse-order epilogue.  gfx950 only.
//
// Reference stages: IC_Angle            src/features/orb_feature_extractor.cpp:177-202 (cv::fastAtan2 :201)
//                   computeOrbDescriptor src/features/orb_feature_extractor.cpp:120-165 (pattern_ :681-852)
//                   Extract epilogue     src/features/orb_feature_extractor.cpp:541-599 (reversed packing)
#include <cfloat>

#include "orbgpu_internal.h"
#include "tile_stage.hpp"
#include "blur_math.hpp"
#include "orient_math.hpp"

namespace {

// 256 tests x (x0, y0, x1, y1): pattern_, orb_feature_extractor.cpp:681-852, as floats (one 16-byte load per test)
struct PatternF { float v[256 * 4]; };
constexpr PatternF make_pattern() {
  constexpr int raw[256 * 4] = {
#include "orb_pattern.inc"
  };
  PatternF p{};
  for (int i = 0; i < 256 * 4; i++) p.v[i] = (float)raw[i];
  return p;
}
__constant__ __attribute__((aligned(16))) PatternF c_pattern = make_pattern();

using namespace orient;
__constant__ __attribute__((aligned(16))) OrientTab c_orient = make_orient_tab();

// cos / sin of the descriptor rotation: the reference evaluates them on float(angle * pi / 180) and keeps floats
// (orb_feature_extractor.cpp:123-124); the oracle takes glibc's double cos / sin rounded to float.
// The argument lies in [0, 2 pi], so a two-constant Cody-Waite reduction and the fdlibm kernels are enough, in explicit
// IEEE operations.  tools/sincos_check.c compares this very sequence with glibc for EVERY float in [0, 6.4] (1.09e9
// values): both results round to the same float each time.  About a quarter of the device libm's sincos, which also
// carries the huge-argument reduction and its scratch.
__device__ __forceinline__ void sincos_short(double x, double &sn, double &cs) {
  const double TWO_OVER_PI = 0.63661977236758134308, PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  const double k = rint(__dmul_rn(x, TWO_OVER_PI));
  double r = __fma_rn(-k, PIO2_HI, x);
  r = __fma_rn(-k, PIO2_LO, r);
  const double z = __dmul_rn(r, r);
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double ps = __fma_rn(z, S6, S5);
  ps = __fma_rn(z, ps, S4);
  ps = __fma_rn(z, ps, S3);
  ps = __fma_rn(z, ps, S2);
  ps = __fma_rn(z, ps, S1);
  const double s = __fma_rn(__dmul_rn(r, z), ps, r);
  double pc = __fma_rn(z, C6, C5);
  pc = __fma_rn(z, pc, C4);
  pc = __fma_rn(z, pc, C3);
  pc = __fma_rn(z, pc, C2);
  pc = __fma_rn(z, pc, C1);
  const double hz = __dmul_rn(0.5, z);
  const double w = __dsub_rn(1.0, hz);
  const double c = __dadd_rn(w, __dadd_rn(__dsub_rn(__dsub_rn(1.0, w), hz), __dmul_rn(__dmul_rn(z, z), pc)));
  const int q = (int)k & 3;
  sn = q == 0 ? s : q == 1 ? c : q == 2 ? -s : -c;
  cs = q == 0 ? c : q == 1 ? -s : q == 2 ? -c : s;
}

// ---- one tap of the rotated BRIEF pattern: cvRound(x*b + y*a) rows and cvRound(x*a - y*b) columns from the patch
//      centre (computeOrbDescriptor :120-165; float32 without FMA), read from a byte patch in LDS with rows of PITCH
//      bytes.  Row and column side by side in packed f32 operations (each half is the IEEE operation of the scalar form;
//      x*a - y*b is x*a + y*(-b) bit for bit), and cvRound as the addition of 1.5 * 2^23: the sum's ulp is 1 there, so the
//      addition rounds to the nearest even integer as v_rndne does, and the integer sits in the low mantissa bits --
//      the constant's own bits go into the address with the patch's base (tap_base_of).
typedef float v2f __attribute__((ext_vector_type(2)));
struct TapRot { v2f ba, anb; };  // (sin, cos), (cos, -sin)
__device__ __forceinline__ TapRot tap_rot(float a /* cos */, float b /* sin */) { return TapRot{v2f{b, a}, v2f{a, -b}}; }
// centre: LDS address of the patch's centre byte.  low 24 bits of the row's float = 0x400000 + r, the column's float =
// 0x4B400000 + c
template <int PITCH>
__device__ __forceinline__ unsigned tap_base_of(unsigned centre) { return centre - (0x400000u * PITCH + 0x4B400000u); }
// xy = (x, y) as it lies in the pattern's register pair: the products take x / y for both halves through op_sel
// (written out: for the y broadcast the compiler copies y into another pair first)
template <int PITCH>
__device__ __forceinline__ int brief_tap(v2f xy, const TapRot &rot, unsigned tap_base) {
  const v2f magic = {12582912.0f, 12582912.0f};
  v2f xb, ya;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(xb) : "v"(xy), "v"(rot.ba));                // (x * b, x * a)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(ya) : "v"(xy), "v"(rot.anb));  // (y * a, y * -b)
  const v2f rc = (xb + ya) + magic;
  const float fr = rc.x, fc = rc.y;  // (__builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
  const unsigned addr = tap_base + (unsigned)(__mul24(__float_as_int(fr), PITCH) + __float_as_int(fc));
  return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(addr);
}

constexpr int PR = 18;              // patch radius staged for the descriptor: the pattern's points lie within 18.385 px
                                    // of the centre (13, 13), so a rotated tap rounds to at most 18
constexpr int PW = 2 * PR + 1;      // 37 rows
constexpr int PDW = 10;             // aligned dwords per staged row: 37 bytes + up to 3 bytes of misalignment
constexpr int PPITCH = PDW * 4;     // 40
constexpr int PRI = 64 / PDW;       // rows per load instruction (6: 60 lanes)
constexpr int PNI = (PW + PRI - 1) / PRI;  // load instructions per patch (7)
constexpr int KPB = 4;              // waves per workgroup
constexpr int KPW = 4;              // keypoints per wave, processed side by side: four times the loads in flight, and the
                                    // pattern loads / address math shared (measured per 256-frame batch: 8 waves x 2:
                                    // 0.256 ms, 4 x 4: 0.245, 2 x 4: 0.251, 6 x 4: 0.316, 8 x 4: 0.290)
constexpr int KPG = KPB * KPW;      // keypoint slots per workgroup
constexpr int PATCH_DW = PW * PDW + PDW;

// blockDim = 64 * KPB; 1-D grid over (slot group, frame) in XCD order.  The kernel is latency-bound (keypoint -> patch
// -> orientation -> sin/cos -> taps is a chain of dependent memory round trips), so it is organised for loads in
// flight, not for instruction count: a wave owns four keypoints, the blurred patches go straight from global memory
// to LDS (global_load_lds: no staging registers, 8 waves per SIMD), and everything that depends only on the slot is
// forced wave-uniform (readfirstlane) so it runs on the scalar ALU.
// k_orient: IC_Angle of every keypoint on the unblurred level, one wave per 4 keypoints.  Runs right after the quadtree,
// beside the blur (it is bound by its scattered loads, the blur by the vector ALU), and leaves the angle in
// d_angle[frame][slot] for k_describe.
__global__ __launch_bounds__(64 * KPB) void k_orient(const uint8_t *__restrict__ pyr, DeviceGeom g,
                                                     const short2 *__restrict__ kp_xy, const int *__restrict__ kp_count,
                                                     float *__restrict__ kp_angle, int n_frames, Level0Src lvl0) {
  // the weight table in LDS: read per lane for every row group of every keypoint, and an LDS read costs this
  // gather-bound kernel less than another trip through the texture path (10 load instructions per keypoint -> 5)
  __shared__ uint2 s_tab[4 * OT_STRIDE];
  int f, group;
  if (!xcd_item_frame((g.kps_per_frame + KPG - 1) / KPG, n_frames, group, f)) return;
#pragma unroll
  for (int i = 0; i < 4 * OT_STRIDE / (64 * KPB); i++) s_tab[i * 64 * KPB + threadIdx.x] = c_orient.e[i * 64 * KPB + threadIdx.x];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int *cnt = kp_count + f * g.levels;
  int slot[KPW], lv[KPW], cx[KPW], cy[KPW];
  bool valid[KPW];
  unsigned l_off[KPW];
  int l_pitch[KPW], l_bytes[KPW];
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    slot[h] = group * KPG + wave * KPW + h;
    lv[h] = level_of(g.kp_offs, slot[h]);
    const LevelGeom &L = g.lv[lv[h]];
    l_off[h] = L.off;
    l_pitch[h] = L.pitch;
    l_bytes[h] = L.pitch * L.h;
    valid[h] = slot[h] < g.kps_per_frame && slot[h] - L.kp_off < cnt[lv[h]];
    cx[h] = cy[h] = 0;
  }
  short2 xy[KPW];
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    xy[h] = make_short2(0, 0);
    if (valid[h]) xy[h] = kp_xy[(size_t)f * g.kps_per_frame + slot[h]];
  }
  const uint8_t *lvl[KPW];
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    cx[h] = __builtin_amdgcn_readfirstlane((int)xy[h].x);
    cy[h] = __builtin_amdgcn_readfirstlane((int)xy[h].y);
    lvl[h] = level_src(pyr, g, lvl0, f, lv[h], l_off[h]);
  }
  unsigned ov[KPW][5];
  orient_load<KPW>(lvl, l_pitch, l_bytes, cx, cy, valid, lane, ov);
  __syncthreads();  // s_tab
  const float ang = orient_angles<KPW>(ov, cx, valid, s_tab, lane);  // keypoint h's on lane h
  bool mine = false;
#pragma unroll
  for (int h = 0; h < KPW; h++) mine = lane == h ? valid[h] : mine;
  if (mine) kp_angle[(size_t)f * g.kps_per_frame + slot[0] + lane] = ang;
}

// k_describe: rotated BRIEF on the blurred level + the output records.  Its cost is the patch gather, so it is organised
// for loads in flight: a wave owns four keypoints, the blurred patches go straight from global memory to LDS
// (global_load_lds: no staging registers), and everything that depends only on the slot is forced wave-uniform
// (readfirstlane) so it runs on the scalar ALU.
__global__ __launch_bounds__(64 * KPB) void k_describe(const uint8_t *__restrict__ blur, DeviceGeom g,
                                                       const short2 *__restrict__ kp_xy,
                                                       const int *__restrict__ kp_count,
                                                       const float *__restrict__ kp_angle,
                                                       orbgpu_keypoint *__restrict__ out_kp,
                                                       uint8_t *__restrict__ out_desc, int capacity,
                                                       int *__restrict__ out_n, int n_frames,
                                                       int *__restrict__ error_flag) {
  __shared__ unsigned patch[KPG][PATCH_DW];
  __shared__ float s_cs[KPG][2];
  __shared__ int s_prefix[ORBGPU_MAX_LEVELS];
  int f, group;
  if (!xcd_item_frame((g.kps_per_frame + KPG - 1) / KPG, n_frames, group, f)) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int *cnt = kp_count + f * g.levels;
  // inclusive prefix of the per-level counts, once per workgroup (16 lanes of wave 0); read after the barriers
  if (threadIdx.x < ORBGPU_MAX_LEVELS) {
    int v = (int)threadIdx.x < g.levels ? cnt[threadIdx.x] : 0;
#pragma unroll
    for (int o = 1; o < ORBGPU_MAX_LEVELS; o <<= 1) {
      const int u = __shfl_up(v, o);
      if ((int)threadIdx.x >= o) v += u;
    }
    s_prefix[threadIdx.x] = v;
  }

  int slot[KPW], lv[KPW], idx[KPW], cx[KPW], cy[KPW], sh[KPW];
  bool valid[KPW];  // wave-uniform; invalid ones only take part in the barriers
  unsigned l_off[KPW];
  int l_pitch[KPW];
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    slot[h] = group * KPG + wave * KPW + h;
    lv[h] = level_of(g.kp_offs, slot[h]);
    const LevelGeom &L = g.lv[lv[h]];
    idx[h] = slot[h] - L.kp_off;
    l_off[h] = L.off;
    l_pitch[h] = L.pitch;
    valid[h] = slot[h] < g.kps_per_frame && idx[h] < cnt[lv[h]];
    cx[h] = cy[h] = sh[h] = 0;
  }
  short2 xy[KPW];
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    xy[h] = make_short2(0, 0);
    if (valid[h]) xy[h] = kp_xy[(size_t)f * g.kps_per_frame + slot[h]];
  }
  // ---- blurred patches: PW rows x PDW aligned dwords, PRI rows per step, global -> LDS directly.  The LDS address of
  //      (row PRI k + lane / PDW, dword lane % PDW) is PRI PDW k + lane, which is what the instruction writes
  //      (base + 4 * lane)
  const int psub = lane / PDW, pc4 = 4 * (lane - psub * PDW);
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    if (!valid[h]) continue;
    cx[h] = __builtin_amdgcn_readfirstlane((int)xy[h].x);
    cy[h] = __builtin_amdgcn_readfirstlane((int)xy[h].y);
    const int xb = (cx[h] - PR) & ~3;
    sh[h] = (cx[h] - PR) - xb;
    const uint8_t *bsrc = blur + (size_t)f * g.frame_bytes + l_off[h] + (size_t)(cy[h] - PR) * l_pitch[h] + xb;
    const unsigned poff = (unsigned)(psub * l_pitch[h] + pc4);
    unsigned *P = patch[wave * KPW + h];
#pragma unroll
    for (int k = 0; k < PNI; k++)
      if (lane < PRI * PDW && PRI * k + psub < PW)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned *>(bsrc + (poff + (unsigned)(PRI * k * l_pitch[h]))),
                                         P + PRI * PDW * k, 4, 0, 0);
  }
  // ---- cos/sin of the workgroup's 16 keypoints by 16 lanes of wave 0 while the patches are in flight:
  //      float(angle_deg) * (double)(pi/180) -> float, cos/sin evaluated in double and rounded to float
  if (threadIdx.x < KPG) {
    const int sl = group * KPG + (int)threadIdx.x;
    const float adeg = sl < g.kps_per_frame ? kp_angle[(size_t)f * g.kps_per_frame + sl] : 0.f;
    const double factorPI = 3.14159265358979323846 / 180.;
    const float ang = (float)((double)adeg * factorPI);
#ifdef ORBGPU_TRIG_F32
    // build-time variant (make TRIG=f32): the float overloads, as a reference build that has <math.h> in scope would
    // call.  The device's cosf / sinf are not glibc's bit for bit: this switch measures sensitivity, it claims no parity
    s_cs[threadIdx.x][0] = cosf(ang);
    s_cs[threadIdx.x][1] = sinf(ang);
#else
    double sn, cs;
    sincos_short((double)ang, sn, cs);
    s_cs[threadIdx.x][0] = (float)cs;
    s_cs[threadIdx.x][1] = (float)sn;
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the direct-to-LDS patch loads have landed
  __syncthreads();
  const int total = s_prefix[ORBGPU_MAX_LEVELS - 1];
  if (group == 0 && threadIdx.x == 0) {
    out_n[f] = total;
    // the frame does not fit the caller's arrays: nothing of it is written, the count says how much room it needs, and
    // the context's error flag makes the next orbgpu_synchronize / orbgpu_stage_times return ORBGPU_E_CAPACITY
    if (total > capacity) *(volatile int *)error_flag = ORBGPU_ERR_OUT_CAPACITY;  // mapped host memory: a plain store
  }
  bool any_valid = false;
#pragma unroll
  for (int h = 0; h < KPW; h++) any_valid |= valid[h];
  if (total > capacity || !any_valid) return;
  // the lane's 4 test patterns (pattern_, 16 bytes each), shared by the wave's keypoints
  const float4 *pat = reinterpret_cast<const float4 *>(c_pattern.v);
  float4 ptn[4];
#pragma unroll
  for (int q = 0; q < 4; q++) ptn[q] = pat[64 * q + lane];
#pragma unroll
  for (int h = 0; h < KPW; h++) {
    if (!valid[h]) continue;
    const int l = lv[h];
    const int pos = total - 1 - ((l ? s_prefix[l - 1] : 0) + idx[h]);  // reversed packing, SURVEY App. B.4
    const float a = s_cs[wave * KPW + h][0], b = s_cs[wave * KPW + h][1];
    // ---- rotated BRIEF (brief_tap)
    const TapRot rot = tap_rot(a, b);
    const unsigned tap_base =
        tap_base_of<PPITCH>((unsigned)reinterpret_cast<size_t>(patch[wave * KPW + h]) + PR * PPITCH + PR + sh[h]);
    unsigned long long bits[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 pt = ptn[q];
      const int t0 = brief_tap<PPITCH>(v2f{pt.x, pt.y}, rot, tap_base);
      const int t1 = brief_tap<PPITCH>(v2f{pt.z, pt.w}, rot, tap_base);
      bits[q] = __ballot(t0 < t1);  // bit k of byte i = test 8i + k: ballot order is the descriptor's byte order
    }
    if (lane == 0) {
      unsigned long long *d = reinterpret_cast<unsigned long long *>(out_desc + ((size_t)f * capacity + pos) * 32);
      d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
      const LevelGeom &L = g.lv[l];
      orbgpu_keypoint k;
      k.x = (double)cx[h];
      k.y = (double)cy[h];
      if (l != 0) {  // keypoint->pt *= (float)scale_factors_[level], orb_feature_extractor.cpp:568-577
        k.x *= (double)L.scale_f;
        k.y *= (double)L.scale_f;
      }
      k.angle = kp_angle[(size_t)f * g.kps_per_frame + slot[h]];
      k.size = L.kp_size;
      k.level = l;
      k.reserved = 0;
      out_kp[(size_t)f * capacity + pos] = k;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_patch_describe: orientation + Gaussian blur + rotated BRIEF of a keypoint from ONE unblurred 43 x 43 patch -- for
// contexts whose keypoints' patches cover far fewer pixels than the pyramid (1080p / 4K frames at 5000 keypoints: the
// full-frame blur writes and k_describe re-reads a blurred pyramid of which the descriptors touch a fifth).  No blurred
// slab, no blur kernel, no join of a second stream; the quadtree kernel is followed by this one alone.
//   cv::GaussianBlur applies its 7 x 7 kernel without intermediate rounding (the horizontal sums are exact 16-bit
//   integers, blur_math.hpp), so a blurred pixel is an exact function of its 7 x 7 unblurred neighbourhood: the taps
//   reach 18 px from the keypoint (PR), their blur 3 px further (QR = 21).  Keypoints sit >= 19 px from the image edge
//   (EDGE_THRESHOLD), so a patch can reach 2 px beyond it: those keypoints gather their patch byte by byte with
//   BORDER_REFLECT_101 indices, the others with aligned dword loads straight into LDS.
// A wave owns FKPW keypoints and works on them alone (no workgroup barrier after the weight table is staged):
//   gather (all patches in flight) -> IC_Angle on the 31 x 31 disc of each raw patch (k_orient's weight table) -> cos / sin
//   of the wave's keypoints side by side on as many lanes -> per keypoint: horizontal 7-tap on row pairs (lane = dword of a
//   pair of rows, neighbours by DPP, k_blur's hblur_px) into (row 2m, row 2m+1) u16 pairs in LDS, vertical 7-tap
//   (vblur_px) back over the raw patch's core, the 256 tests (k_describe's), the output record.
constexpr int QR = PR + 3;                 // 21: radius of the raw patch
constexpr int QW = 2 * QR + 1;             // 43 rows / columns
constexpr int QDW = 12;                    // aligned dwords per staged row: 43 bytes + up to 3 of misalignment
constexpr int QPITCH = QDW * 4;            // 48
constexpr int QRI = 64 / QDW;              // rows per load instruction (5: 60 lanes)
constexpr int QNI = (QW + QRI - 1) / QRI;  // load instructions per patch (9)
constexpr int RAW_DW = QNI * QRI * QDW;    // 45 rows of 12 dwords: rows 43, 44 are never loaded (row 43 is read with weight 0)
constexpr int HB_PAIRS = (QW + 1) / 2;     // 22 row pairs
constexpr int HB_PITCH = 48;               // dwords (pixels) per row pair
constexpr int FKPB = 4;                    // waves per workgroup (8: 16 waves per CU instead of 12, same time)
constexpr int FKPW = 2;                    // keypoints per wave
constexpr int FKPG = FKPB * FKPW;

__global__ __launch_bounds__(64 * FKPB) void k_patch_describe(const uint8_t *__restrict__ pyr, DeviceGeom g,
                                                              const short2 *__restrict__ kp_xy,
                                                              const int *__restrict__ kp_count,
                                                              float *__restrict__ kp_angle,
                                                              orbgpu_keypoint *__restrict__ out_kp,
                                                              uint8_t *__restrict__ out_desc, int capacity,
                                                              int *__restrict__ out_n, int n_frames, Level0Src lvl0,
                                                              int *__restrict__ error_flag) {
  __shared__ uint2 s_tab[4 * OT_STRIDE];
  __shared__ __attribute__((aligned(16))) unsigned s_raw[FKPG][RAW_DW];
  __shared__ __attribute__((aligned(16))) unsigned s_hb[FKPB][HB_PAIRS * HB_PITCH];
  int f, group;
  if (!xcd_item_frame((g.kps_per_frame + FKPG - 1) / FKPG, n_frames, group, f)) return;
#pragma unroll
  for (int i = 0; i < 4 * OT_STRIDE / (64 * FKPB); i++) s_tab[i * 64 * FKPB + threadIdx.x] = c_orient.e[i * 64 * FKPB + threadIdx.x];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int *cnt = kp_count + f * g.levels;
  // inclusive prefix of the per-level counts, per wave (16 lanes, no barrier)
  int pre = lane < g.levels ? cnt[lane] : 0;
#pragma unroll
  for (int o = 1; o < ORBGPU_MAX_LEVELS; o <<= 1) {
    const int u = __shfl_up(pre, o);
    if (lane >= o) pre += u;
  }
  const int total = __builtin_amdgcn_readlane(pre, ORBGPU_MAX_LEVELS - 1);
  if (group == 0 && threadIdx.x == 0) {
    out_n[f] = total;
    if (total > capacity) *(volatile int *)error_flag = ORBGPU_ERR_OUT_CAPACITY;  // mapped host memory: a plain store
  }
  int slot[FKPW], lv[FKPW], idx[FKPW], cx[FKPW], cy[FKPW], sh[FKPW];
  bool valid[FKPW];
  short2 xy[FKPW];
#pragma unroll
  for (int h = 0; h < FKPW; h++) {
    slot[h] = group * FKPG + wave * FKPW + h;
    lv[h] = __builtin_amdgcn_readfirstlane(level_of(g.kp_offs, slot[h]));
    idx[h] = slot[h] - g.lv[lv[h]].kp_off;
    valid[h] = slot[h] < g.kps_per_frame && idx[h] < cnt[lv[h]];
    xy[h] = make_short2(0, 0);
    if (valid[h]) xy[h] = kp_xy[(size_t)f * g.kps_per_frame + slot[h]];
    cx[h] = cy[h] = sh[h] = 0;
  }
  // ---- gather the raw patches -------------------------------------------------------------------------------------
  const int psub = lane / QDW, pdw = lane - psub * QDW;
#pragma unroll
  for (int h = 0; h < FKPW; h++) {
    if (!valid[h]) continue;
    cx[h] = __builtin_amdgcn_readfirstlane((int)xy[h].x);
    cy[h] = __builtin_amdgcn_readfirstlane((int)xy[h].y);
    const LevelGeom &L = g.lv[lv[h]];
    const uint8_t *src = level_src(pyr, g, lvl0, f, lv[h], L.off);
    unsigned *R = s_raw[wave * FKPW + h];
    const bool inside = cx[h] - QR >= 0 && cx[h] + QR <= L.w - 1 && cy[h] - QR >= 0 && cy[h] + QR <= L.h - 1;
    if (inside) {
      const int xb = (cx[h] - QR) & ~3;
      sh[h] = (cx[h] - QR) - xb;
      // the row's last dword (bytes 44 .. 47) holds patch bytes only when the patch starts at byte 2 or 3 of its first
      // dword; otherwise it lies wholly beyond the patch (and may lie beyond the image row): not loaded
      const bool want = lane < QRI * QDW && (pdw < QDW - 1 || sh[h] >= 2);
      const uint8_t *bsrc = src + (size_t)(cy[h] - QR) * L.pitch + xb;
      const unsigned poff = (unsigned)(psub * L.pitch + 4 * pdw);
#pragma unroll
      for (int k = 0; k < QNI; k++)
        if (want && QRI * k + psub < QW)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned *>(bsrc + (poff + (unsigned)(QRI * k * L.pitch))),
                                           R + QRI * QDW * k, 4, 0, 0);
    } else {
      // BORDER_REFLECT_101 (cv::GaussianBlur's border, :559): byte by byte, patch column 0 at byte 0 of the staged row
      uint8_t *R8 = reinterpret_cast<uint8_t *>(R);
      for (int i = lane; i < QW * QW; i += 64) {
        const int r = i / QW, c = i - r * QW;
        int gy = cy[h] - QR + r, gx = cx[h] - QR + c;
        gy = gy < 0 ? -gy : (gy > L.h - 1 ? 2 * (L.h - 1) - gy : gy);
        gx = gx < 0 ? -gx : (gx > L.w - 1 ? 2 * (L.w - 1) - gx : gx);
        R8[r * QPITCH + c] = src[(size_t)gy * L.pitch + gx];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's patches have landed in LDS
  __syncthreads();                                             // s_tab (the only workgroup-wide dependency)
  bool any_valid = false;
#pragma unroll
  for (int h = 0; h < FKPW; h++) any_valid |= valid[h];
  if (total > capacity || !any_valid) return;
  // ---- IC_Angle on the raw patches (rows 6 .. 36, columns sh + 6 .. sh + 36 of the staged patch) -----------------------
  const int osub = lane / ODW, oc = lane - osub * ODW;
  int m10s[FKPW], m01s[FKPW];
#pragma unroll
  for (int h = 0; h < FKPW; h++) {
    m10s[h] = m01s[h] = 0;
    if (!valid[h]) continue;
    const unsigned *R = s_raw[wave * FKPW + h];
    const int osh = (sh[h] + (QR - ORB_HALF_PATCH)) & 3, odw = (sh[h] + (QR - ORB_HALF_PATCH)) >> 2;
    unsigned a10 = 0, si = 0;
    int m01 = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int row = 7 * k + osub;  // row of the disc; lanes beyond it read weight 0
      const unsigned px = (lane < 63 && row < ORB_PATCH) ? R[(row + QR - ORB_HALF_PATCH) * QDW + odw + oc] : 0u;
      const uint2 ow = s_tab[osh * OT_STRIDE + 63 * k + min(lane, 62)];
      const unsigned rs = __builtin_amdgcn_udot4(px, ow.y, 0u, false);
      a10 = __builtin_amdgcn_udot4(px, ow.x, a10, false);
      si += rs;
      m01 += __mul24(row - ORB_HALF_PATCH, (int)rs);
    }
    int m10 = (int)a10 - ORB_HALF_PATCH * (int)si;
    m10s[h] = __builtin_amdgcn_readlane(wave_incl_scan(m10), 63);  // (as orient_moments)
    m01s[h] = __builtin_amdgcn_readlane(wave_incl_scan(m01), 63);
  }
  // ---- cos / sin of the wave's keypoints, keypoint h on lane h (float(angle_deg) * (double)(pi / 180) -> float, cos / sin
  //      evaluated in double and rounded to float) ------------------------------------------------------
  const float my_ang = orient_atan_lanes<FKPW>(m10s, m01s, lane);
  float my_cos, my_sin;
  {
    const double factorPI = 3.14159265358979323846 / 180.;
    const float a = (float)((double)my_ang * factorPI);
#ifdef ORBGPU_TRIG_F32
    my_cos = cosf(a);
    my_sin = sinf(a);
#else
    double sn, cs;
    sincos_short((double)a, sn, cs);
    my_cos = (float)cs;
    my_sin = (float)sn;
#endif
  }
  const float4 *pat = reinterpret_cast<const float4 *>(c_pattern.v);
  float4 ptn[4];
#pragma unroll
  for (int q = 0; q < 4; q++) ptn[q] = pat[64 * q + lane];
  unsigned *HB = s_hb[wave];
  const int hpl = lane / QDW, hd = lane - hpl * QDW;  // horizontal pass: row pair within a step, dword
  const bool vodd = lane & 1;                         // vertical pass: the lane's output rows start on a pair's second row
#pragma unroll  // (a rolled loop would index the per-keypoint registers dynamically: scratch memory)
  for (int h = 0; h < FKPW; h++) {
    if (!valid[h]) continue;
    unsigned *R = s_raw[wave * FKPW + h];
    const float ang_h = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_ang), h));
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_cos), h));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_sin), h));
    // ---- horizontal 7-tap, two rows at a time: lane = (row pair, dword); the neighbours' dwords by DPP.  A pair's first
    //      dword takes its left neighbour from the previous pair's last dword and vice versa: that only reaches output
    //      bytes 0 .. 2 and 45 .. 47 of a row, outside the 37 columns sh + 3 .. sh + 39 the vertical pass reads
#pragma unroll
    for (int p = 0; p < (HB_PAIRS + QRI - 1) / QRI; p++) {
      const int m = QRI * p + hpl;
      const bool live = lane < QRI * QDW && m < HB_PAIRS;
      const int mm = live ? m : 0;
      const unsigned r0 = R[(2 * mm) * QDW + hd], r1 = R[(2 * mm + 1) * QDW + hd];
      const unsigned w0[3] = {from_left(r0), r0, from_right(r0)}, w1[3] = {from_left(r1), r1, from_right(r1)};
      uint4 o;
      o.x = hblur_px<0>(w0) | (hblur_px<0>(w1) << 16);
      o.y = hblur_px<1>(w0) | (hblur_px<1>(w1) << 16);
      o.z = hblur_px<2>(w0) | (hblur_px<2>(w1) << 16);
      o.w = hblur_px<3>(w0) | (hblur_px<3>(w1) << 16);
      if (live) *reinterpret_cast<uint4 *>(HB + m * HB_PITCH + 4 * hd) = o;
    }
    // ---- vertical 7-tap over the core: output rows 3 .. 39, dwords 0 .. 10, written over the raw patch.  Even lanes take
    //      the rows whose 7 source rows start on the first row of a pair, odd lanes the others, so a lane's weights never
    //      change; item j = (64 p + lane) / 2 = (row pair jr, dword d) advances by 32 = 2 * 11 + 10 per step ------------------
    {
      int jr = (lane >> 1) / (QDW - 1), d = (lane >> 1) - jr * (QDW - 1);
#pragma unroll
      for (int p = 0; p < (2 * ((PW + 1) / 2) * (QDW - 1) + 63) / 64; p++) {
        const int yr = 2 * jr + (lane & 1);  // output row - 3 = first of its 7 source rows
        if (yr < PW) {
          const uint4 A = *reinterpret_cast<const uint4 *>(HB + (jr + 0) * HB_PITCH + 4 * d);
          const uint4 B = *reinterpret_cast<const uint4 *>(HB + (jr + 1) * HB_PITCH + 4 * d);
          const uint4 C = *reinterpret_cast<const uint4 *>(HB + (jr + 2) * HB_PITCH + 4 * d);
          const uint4 D = *reinterpret_cast<const uint4 *>(HB + (jr + 3) * HB_PITCH + 4 * d);
          const unsigned v0 = vblur_px(A.x, B.x, C.x, D.x, vodd), v1 = vblur_px(A.y, B.y, C.y, D.y, vodd);
          const unsigned v2 = vblur_px(A.z, B.z, C.z, D.z, vodd), v3 = vblur_px(A.w, B.w, C.w, D.w, vodd);
          // every accumulator is < 2^24: the result pixel is its byte 2
          R[(yr + 3) * QDW + d] = __builtin_amdgcn_perm(v1, v0, 0x0c0c0602u) | __builtin_amdgcn_perm(v3, v2, 0x06020c0cu);
        }
        d += 10;
        jr += 2;
        if (d >= QDW - 1) {
          d -= QDW - 1;
          jr++;
        }
      }
    }
    // ---- rotated BRIEF on the blurred core (computeOrbDescriptor :120-165), as k_describe ----------------------------
    const int l = lv[h];
    const int pos = total - 1 - ((l ? __builtin_amdgcn_readlane(pre, max(l - 1, 0)) : 0) + idx[h]);  // reversed packing, App. B.4
    const TapRot rot = tap_rot(a, b);
    const unsigned tap_base = tap_base_of<QPITCH>((unsigned)reinterpret_cast<size_t>(R) + QR * QPITCH + QR + sh[h]);
    unsigned long long bits[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 pt = ptn[q];
      const int t0 = brief_tap<QPITCH>(v2f{pt.x, pt.y}, rot, tap_base);
      const int t1 = brief_tap<QPITCH>(v2f{pt.z, pt.w}, rot, tap_base);
      bits[q] = __ballot(t0 < t1);
    }
    if (lane == 0) {
      unsigned long long *dd = reinterpret_cast<unsigned long long *>(out_desc + ((size_t)f * capacity + pos) * 32);
      dd[0] = bits[0]; dd[1] = bits[1]; dd[2] = bits[2]; dd[3] = bits[3];
      const LevelGeom &L = g.lv[l];
      orbgpu_keypoint k;
      k.x = (double)cx[h];
      k.y = (double)cy[h];
      if (l != 0) {  // keypoint->pt *= (float)scale_factors_[level], orb_feature_extractor.cpp:568-577
        k.x *= (double)L.scale_f;
        k.y *= (double)L.scale_f;
      }
      k.angle = ang_h;
      k.size = L.kp_size;
      k.level = l;
      k.reserved = 0;
      out_kp[(size_t)f * capacity + pos] = k;
      kp_angle[(size_t)f * g.kps_per_frame + slot[h]] = ang_h;
    }
  }
}

}  // namespace

void launch_patch_describe(const orbgpu_ctx *c, FrameSlice fs, orbgpu_keypoint *d_kp, uint8_t *d_desc, int capacity,
                           int *d_n_out) {
  const DeviceGeom &g = c->geom;
  dim3 grid((g.kps_per_frame + FKPG - 1) / FKPG * xcd_frames(fs.n));
  hipLaunchKernelGGL(k_patch_describe, grid, dim3(64 * FKPB), 0, fs.stream, c->d_pyr + (size_t)fs.f0 * g.frame_bytes, g,
                     c->d_kp_xy + (size_t)fs.f0 * g.kps_per_frame, c->d_kp_count + (size_t)fs.f0 * g.levels,
                     c->d_kp_angle + (size_t)fs.f0 * g.kps_per_frame, d_kp + (size_t)fs.f0 * capacity,
                     d_desc + (size_t)fs.f0 * capacity * 32, capacity, d_n_out + fs.f0, fs.n, slice_lvl0(c, fs),
                     c->d_error_flag);
}

void launch_orient(const orbgpu_ctx *c, FrameSlice fs) {
  const DeviceGeom &g = c->geom;
  dim3 grid((g.kps_per_frame + KPG - 1) / KPG * xcd_frames(fs.n));
  hipLaunchKernelGGL(k_orient, grid, dim3(64 * KPB), 0, fs.stream, c->d_pyr + (size_t)fs.f0 * g.frame_bytes, g,
                     c->d_kp_xy + (size_t)fs.f0 * g.kps_per_frame, c->d_kp_count + (size_t)fs.f0 * g.levels,
                     c->d_kp_angle + (size_t)fs.f0 * g.kps_per_frame, fs.n, slice_lvl0(c, fs));
}

void launch_describe(const orbgpu_ctx *c, FrameSlice fs, orbgpu_keypoint *d_kp, uint8_t *d_desc, int capacity,
                     int *d_n_out) {
  const DeviceGeom &g = c->geom;
  dim3 grid((g.kps_per_frame + KPG - 1) / KPG * xcd_frames(fs.n));
  hipLaunchKernelGGL(k_describe, grid, dim3(64 * KPB), 0, fs.stream, c->d_blur + (size_t)fs.f0 * g.frame_bytes, g,
                     c->d_kp_xy + (size_t)fs.f0 * g.kps_per_frame, c->d_kp_count + (size_t)fs.f0 * g.levels,
                     c->d_kp_angle + (size_t)fs.f0 * g.kps_per_frame, d_kp + (size_t)fs.f0 * capacity,
                     d_desc + (size_t)fs.f0 * capacity * 32, capacity, d_n_out + fs.f0, fs.n, c->d_error_flag);
}

// ------------------------------------------------------------------------------------------------
// Post-extract per-keypoint geometry (factories/feature_handler_factory.cpp:53-60): MonocularCamera::UndistortPoint
// and ::UnprojectAndUndistort (camera/monocular_camera.cpp:31-61) with Barrel5::UnDistortPoint
// (camera/distortions/barrel5.cpp:44-70) or FishEye::UnDistortPoint (fish_eye.cpp:68-116).  One lane per keypoint,
// double precision, no contraction; the Barrel5 path is +,-,*,/ only and therefore bit-exact, FishEye calls tan().
namespace {

__device__ __forceinline__ bool cam_undistort3(const orbgpu_camera &c, double dx, double dy, double (&u)[3]) {
  if (c.model == ORBGPU_DISTORTION_BARREL5) {
    const double k1 = c.k[0], k2 = c.k[1], p1 = c.k[2], p2 = c.k[3], k3 = c.k[4];
    double x = dx, y = dy;
    u[2] = 1;
    for (int j = 0; j < 10; j++) {
      const double r2 = x * x + y * y;
      const double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      if (icdist < 0) { u[0] = x; u[1] = y; return false; }
      const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      const double xnew = (dx - deltaX) * icdist;
      const double ynew = (dy - deltaY) * icdist;
      if (fabs(x - xnew) < 1e-7 && fabs(y - ynew) < 1e-7) { u[0] = x; u[1] = y; return true; }
      x = xnew;
      y = ynew;
    }
    u[0] = -10000; u[1] = 10000; u[2] = -1;
    return false;
  }
  const double k1 = c.k[0], k2 = c.k[1], k3 = c.k[2], k4 = c.k[3];
  const double pi_2 = 1.57079632679489661923;
  double theta_d = sqrt(dx * dx + dy * dy);
  theta_d = fmin(pi_2, fmax(-pi_2, theta_d));
  double theta = theta_d;
  bool converged = false;
  for (int i = 0; i < 10; ++i) {
    const double g2 = theta * theta, g3 = g2 * theta, g4 = g3 * theta, g5 = g4 * theta, g6 = g5 * theta, g7 = g6 * theta,
                 g8 = g7 * theta, g9 = g8 * theta;
    const double fp = 1 + 3 * k1 * g2 + 5 * k2 * g4 + 7 * k3 * g6 + 9 * k4 * g8;
    const double theta_fix = (theta + k1 * g3 + k2 * g5 + k3 * g7 + k4 * g9 - theta_d) / fp;
    theta -= theta_fix;
    converged = fabs(theta_fix) < 7e-8;
    if (converged) break;
  }
  if (!converged) { u[0] = -10000; u[1] = -10000; u[2] = -1; return false; }
  const double scale = tan(theta) / theta_d;
  const bool flipped = (theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0);
  if (!flipped && scale > 0) {
    u[0] = dx * scale; u[1] = dy * scale; u[2] = 1;
    return true;
  }
  u[0] = -10000; u[1] = -10000; u[2] = -1;
  return false;
}

// kp: [n_frames][capacity]; n == nullptr: a single flat array of `capacity` keypoints
__global__ __launch_bounds__(256) void k_keypoint_geometry(orbgpu_camera c, const orbgpu_keypoint *__restrict__ kp,
                                                           const int *__restrict__ n, int capacity,
                                                           double *__restrict__ undist, double *__restrict__ unproj,
                                                           uint8_t *__restrict__ ok) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int count = n ? usable_count(n[f], capacity) : capacity;
  if (i >= count) return;
  const size_t s = (size_t)f * capacity + i;
  const double fx_inv = c.fx != 0 ? 1 / c.fx : 0, fy_inv = c.fy != 0 ? 1 / c.fy : 1;
  const double px = kp[s].x, py = kp[s].y;
  const double ux = (px - c.cx) * fx_inv, uy = (py - c.cy) * fy_inv;
  uint8_t flags = 0;
  double u[3];
  if (c.model < 0) {
    undist[2 * s] = px; undist[2 * s + 1] = py;
    unproj[3 * s] = ux; unproj[3 * s + 1] = uy; unproj[3 * s + 2] = 1;
    flags = 3;
  } else {
    const bool good = cam_undistort3(c, ux, uy, u);  // both reference calls run the same undistortion
    if (good) {
      const double z_inv = 1 / u[2];
      undist[2 * s] = u[0] * z_inv * c.fx + c.cx;
      undist[2 * s + 1] = u[1] * z_inv * c.fy + c.cy;
      flags = 3;
    } else {
      undist[2 * s] = 0; undist[2 * s + 1] = 0;
    }
    unproj[3 * s] = u[0]; unproj[3 * s + 1] = u[1]; unproj[3 * s + 2] = u[2];
  }
  ok[s] = flags;
}

}  // namespace

void launch_keypoint_geometry(const orbgpu_ctx *c, const orbgpu_camera &cam, const orbgpu_keypoint *d_kp, const int *d_n,
                              int n_frames, int capacity, double *d_undist, double *d_unproj, uint8_t *d_ok) {
  if (n_frames <= 0 || capacity <= 0) return;
  hipLaunchKernelGGL(k_keypoint_geometry, dim3((capacity + 255) / 256, n_frames), dim3(256), 0, c->stream, cam, d_kp, d_n,
                     capacity, d_undist, d_unproj, d_ok);
}
