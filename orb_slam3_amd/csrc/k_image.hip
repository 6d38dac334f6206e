// Image-plane kernels: import, pyramid level (cv::resize INTER_LINEAR restated in fixed point) and the
// 7x7 Gaussian blur.  gfx950 only.
//
// Reference stages: BuildImagePyramid  src/features/orb_feature_extractor.cpp:504-528 (cv::resize :517)
//                   blur in Extract    src/features/orb_feature_extractor.cpp:558-560 (cv::GaussianBlur)
#include <algorithm>
#include <vector>

#include "orbgpu_internal.h"
#include "tile_stage.hpp"
#include "blur_math.hpp"

namespace {

constexpr int TILE_W = 128;  // output pixels per resize tile row (32 lanes x 4 px)

// ------------------------------------------------------------------------------------------------
// import: user images (arbitrary row stride) -> level 0 of each frame slab
__global__ __launch_bounds__(256) void k_import(const uint8_t *__restrict__ src, long long frame_stride, int stride,
                                                uint8_t *__restrict__ pyr, unsigned frame_bytes, int w, int h,
                                                int pitch, int aligned16) {
  const int f = blockIdx.z;
  const int y = blockIdx.y;
  const int x16 = (blockIdx.x * 256 + threadIdx.x) * 16;
  if (x16 >= pitch) return;
  const uint8_t *s = src + (long long)f * frame_stride + (long long)y * stride;
  uint8_t *d = pyr + (size_t)f * frame_bytes + (size_t)y * pitch + x16;
  if (aligned16 && x16 + 16 <= w) {
    *reinterpret_cast<uint4 *>(d) = *reinterpret_cast<const uint4 *>(s + x16);
  } else {
    unsigned v[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) {
      int x = x16 + i;
      unsigned b = x < w ? s[x] : 0u;
      v[i >> 2] |= b << (8 * (i & 3));
    }
    *reinterpret_cast<uint4 *>(d) = make_uint4(v[0], v[1], v[2], v[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// pyramid level l from level l-1.  Integer arithmetic of OpenCV's VResizeLinear/HResizeLinear for 8U:
//   H = S[sx]*a0 + S[sx+1]*a1 ; dst = (((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2
// The coefficient tables come from the host (exact float/double sequence).
// One workgroup produces a 128 x 64 output tile (measured per 256-frame batch: 32 rows 0.214 ms, 64 rows 0.196, 96 rows
// 0.193, 128 rows 0.223; 64 keeps the footprint of a 2.4x level below 64 KB); its source footprint (rows sy0(first)..sy1(last), columns from the
// 16-byte boundary below sx0(first) to sx0(last)+1) is staged in dynamic LDS whose size the host derives from the
// same tables (rs_pitch x rs_rows), so there is no un-staged path.  A lane owns 4 columns x 8 rows: the column
// coefficients are loaded once and reused for the 8 rows.  Where sx1 clamps to the last column its weight is 0, so
// the kernel always reads column sx0+1 of the staged row.
constexpr int RS_TILE_H = 64;

// full-rate 24-bit multiplies, pinned: the optimiser otherwise widens them to the quarter-rate v_mul_lo_u32
__device__ __forceinline__ unsigned mul24u(unsigned a, unsigned b) {
  unsigned d;
  asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ unsigned mad24u(unsigned a, unsigned b, unsigned c) {
  unsigned d;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__global__ __launch_bounds__(256) void k_resize_level(uint8_t *__restrict__ pyr, DeviceGeom g, int level,
                                                      const XTab *__restrict__ xtab, const YTab *__restrict__ ytab,
                                                      int lds_pitch, int n_frames, Level0Src lvl0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rs_lds[];
  const LevelGeom S = g.lv[level - 1], D = g.lv[level];
  const int tiles_x = (D.w + TILE_W - 1) / TILE_W, tiles_y = (D.h + RS_TILE_H - 1) / RS_TILE_H;
  int t, f;
  if (!xcd_item_frame(tiles_x * tiles_y, n_frames, t, f)) return;
  const int tile_y = t / tiles_x, tile_x = t - tile_y * tiles_x;
  const uint8_t *src = level_src(pyr, g, lvl0, f, level - 1, S.off);
  uint8_t *dst = pyr + (size_t)f * g.frame_bytes + D.off;
  const XTab *xt = xtab + D.xtab_off;
  const YTab *yt = ytab + D.ytab_off;

  const int dx0 = tile_x * TILE_W, dy0 = tile_y * RS_TILE_H;
  const int dx1 = min(dx0 + TILE_W, D.w) - 1, dy1 = min(dy0 + RS_TILE_H, D.h) - 1;
  // source footprint of the tile (tables are monotone); wave-uniform, read through the scalar cache
  const int sxa = xt[dx0].sx0 & ~15;
  const int sxb = xt[dx1].sx0 + 1;
  const int sya = yt[dy0].sy0, syb = yt[dy1].sy1;
  const int ncols16 = (sxb - sxa) / 16 + 1;
  const int nrows = syb - sya + 1;
  {
    // pitch is a multiple of 16, so the 16-byte reads stay inside the row; the chunk that would start at the pitch
    // (column sx0 + 1 of a clamped last column: weight 0, any value will do) is not read -- level 0 may be the caller's
    // buffer, which ends with its last row
    const float inv = 1.0f / (float)ncols16;
    for (int i = threadIdx.x; i < nrows * ncols16; i += 256) {
      int r = (int)((float)i * inv);  // i < 2^15: at most one off, fixed below
      int c = i - r * ncols16;
      if (c < 0) { r--; c += ncols16; }
      if (c >= ncols16) { r++; c -= ncols16; }
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (sxa + 16 * c < S.pitch) v = *reinterpret_cast<const uint4 *>(src + (size_t)(sya + r) * S.pitch + sxa + 16 * c);
      *reinterpret_cast<uint4 *>(&rs_lds[r * lds_pitch + 16 * c]) = v;
    }
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads, 4 columns x 4 rows each
  const int dx = dx0 + 4 * tx;
  // column coefficients (clamped index: columns >= w are computed but masked out of the store)
  int cx[4];
  unsigned ca0[4], ca1[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const XTab X = xt[min(dx + p, D.w - 1)];
    cx[p] = X.sx0 - sxa;
    ca0[p] = X.a0;
    ca1[p] = X.a1;
  }
  YTab Y[RS_TILE_H / 8];
#pragma unroll
  for (int k = 0; k < RS_TILE_H / 8; k++) Y[k] = yt[min(dy0 + ty + 8 * k, D.h - 1)];
  unsigned keep = 0;
#pragma unroll
  for (int p = 0; p < 4; p++)
    if (dx + p < D.w) keep |= 0xffu << (8 * p);
  __syncthreads();
  if (dx >= D.pitch) return;
#pragma unroll
  for (int k = 0; k < RS_TILE_H / 8; k++) {
    const int dy = dy0 + ty + 8 * k;
    if (dy >= D.h) break;
    const uint8_t *r0 = rs_lds + mul24u(Y[k].sy0 - sya, lds_pitch);
    const uint8_t *r1 = rs_lds + mul24u(Y[k].sy1 - sya, lds_pitch);
    const unsigned b0 = (unsigned)Y[k].b0, b1 = (unsigned)Y[k].b1;  // all coefficients are in [0, 2048]
    unsigned out = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const unsigned s00 = r0[cx[p]], s01 = r0[cx[p] + 1], s10 = r1[cx[p]], s11 = r1[cx[p] + 1];
      const unsigned h0 = mad24u(s01, ca1[p], mul24u(s00, ca0[p]));
      const unsigned h1 = mad24u(s11, ca1[p], mul24u(s10, ca0[p]));
      const unsigned v = ((mul24u(b0, h0 >> 4) >> 16) + (mul24u(b1, h1 >> 4) >> 16) + 2) >> 2;
      out |= (v & 255u) << (8 * p);
    }
    *reinterpret_cast<unsigned *>(dst + (size_t)dy * D.pitch + dx) = out & keep;
  }
}

// ------------------------------------------------------------------------------------------------
// The same resize without LDS, for levels whose column tables allow it (every scale factor up to about 2.3): a wave
// owns a STRIP of up to 64 output dwords x 8..32 output rows and walks down the SOURCE rows its output rows need.
// Per source row: lane i loads the strip's source dword i (one coalesced load, 8 rows ahead), every lane fetches the
// three dwords its 4 output pixels touch with ds_bpermute (the LDS crossbar, no LDS memory), aligns them to an 8-byte
// window (v_alignbyte_b32) and forms, per pixel, (s[sx0], s[sx0+1]) with one v_perm_b32 and the horizontal blend with
// one v_dot2_u32_u16 against (a0, a1) << 4.  The blend of the previous source row stays in registers, so an output
// row costs one vertical pass: (b0 * (h0 >> 4)) >> 16 is v_mul_hi_u32_u24 of (b0 << 8) and ((h << 4) & ~255).
// Row control (which output rows a source row completes) is scalar: the wave's lanes share their rows.
// v_mul_hi_u32_u24 with a wave-uniform first factor straight from its scalar register
__device__ __forceinline__ unsigned mulhi24u_s(unsigned a_uniform, unsigned b) {
  unsigned d;
  asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "s"(a_uniform), "v"(b));
  return d;
}

// cols: the level's column table (build_resize_columns): per output dword what the lane that owns it needs, ready made
__global__ __launch_bounds__(256) void k_resize_strips(uint8_t *__restrict__ pyr, DeviceGeom g, int level,
                                                       const RsColumn *__restrict__ cols, const YTab *__restrict__ ytab,
                                                       const short4 *__restrict__ strips, int items_per_frame, int n_frames,
                                                       Level0Src lvl0) {
  typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;
  int item, f;
  if (!xcd_item_frame(items_per_frame, n_frames, item, f)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const short4 st = strips[item * 4 + wave];  // (first output dword, first output row, output dwords, output rows)
  const int d0 = __builtin_amdgcn_readfirstlane((int)st.x), y0 = __builtin_amdgcn_readfirstlane((int)st.y);
  const int nd = __builtin_amdgcn_readfirstlane((int)st.z);
  if (nd == 0) return;  // padding of the strip table
  const LevelGeom &S = g.lv[level - 1], &D = g.lv[level];
  const uint8_t *src = level_src(pyr, g, lvl0, f, level - 1, S.off);
  uint8_t *dst = pyr + (size_t)f * g.frame_bytes + D.off;
  const YTab *yt = ytab + D.ytab_off;
  const int dx = 4 * (d0 + lane);
  // columns: the lane's entry of the level's table (lanes beyond the level's last dword take that one: never stored)
  const int sbase = cols[d0].s0;  // the strip's first source dword (scalar)
  const RsColumn C = cols[min(d0 + lane, D.pitch / 4 - 1)];
  const int k0 = C.s0 - sbase;
  const unsigned o = C.o;
  const unsigned(&sel)[4] = C.sel, (&coef)[4] = C.coef;
  const int i0 = 4 * min(k0, 63), i1 = 4 * min(k0 + 1, 63), i2 = 4 * min(k0 + 2, 63);
  // buffer loads and stores: the level's first byte is the (scalar) base, the row's offset is scalar, the lane's column a
  // 32-bit register -- no vector instruction per row for addresses
  const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, S.h * S.pitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(dst, 0, D.h * D.pitch, 0x00020000);
  const unsigned lcol = 4u * (unsigned)min(sbase + lane, S.pitch / 4 - 1);
  const unsigned keep = C.keep;  // bytes of the lane's dword that are pixels of the level
  const bool writes = lane < nd;
  unsigned ooff = (unsigned)(y0 * D.pitch + dx);  // the lane's byte offset into the level, next output row

  const int y_end = min(y0 + __builtin_amdgcn_readfirstlane((int)st.w), D.h);
  const int sa = yt[y0].sy0, sb = yt[y_end - 1].sy1;
  int y = y0;
  YTab Y = yt[y];
  unsigned H[2][4];  // horizontal blends << 4 (low byte cleared) of the last two source rows, by row parity
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int p = 0; p < 4; p++) H[a][p] = 0;

  auto hrow = [&](unsigned raw, unsigned(&h)[4]) {
    const unsigned w0 = (unsigned)__builtin_amdgcn_ds_bpermute(i0, (int)raw);
    const unsigned w1 = (unsigned)__builtin_amdgcn_ds_bpermute(i1, (int)raw);
    const unsigned w2 = (unsigned)__builtin_amdgcn_ds_bpermute(i2, (int)raw);
    const unsigned a = __builtin_amdgcn_alignbyte(w1, w0, o), b = __builtin_amdgcn_alignbyte(w2, w1, o);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const unsigned pair = __builtin_amdgcn_perm(b, a, sel[p]);
      h[p] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), __builtin_bit_cast(u16x2, coef[p]), 0u, false) & 0xffffff00u;
    }
  };
  auto emit = [&](const unsigned(&h0)[4], const unsigned(&h1)[4]) {
    const unsigned b0 = (unsigned)Y.b0 << 8, b1 = (unsigned)Y.b1 << 8;  // all coefficients are in [0, 2048]
    unsigned v[4];
#pragma unroll
    for (int p = 0; p < 4; p++) v[p] = mulhi24u_s(b0, h0[p]) + mulhi24u_s(b1, h1[p]) + 2u;  // <= 1022
    const u16x2 lo = __builtin_bit_cast(u16x2, v[0] | (v[1] << 16)) >> 2, hi = __builtin_bit_cast(u16x2, v[2] | (v[3] << 16)) >> 2;
    const unsigned out = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x06040200u);
    if (writes) __builtin_amdgcn_raw_buffer_store_b32(out & keep, rsrc_d, ooff, 0u, 0);
    ooff += (unsigned)D.pitch;  // output rows are emitted in order
  };
  // source rows sa .. sb, 8 at a time, the next 8 in flight (rows beyond sb repeat sb: loaded, never used)
  auto load8 = [&](unsigned(&buf)[8], int s0) {
#pragma unroll
    for (int k = 0; k < 8; k++) buf[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc_s, lcol, (unsigned)(min(s0 + k, sb) * S.pitch), 0);
  };
  auto chunk = [&](const unsigned(&cur)[8], unsigned(&nxt)[8], int s0) {
    if (s0 + 8 <= sb) load8(nxt, s0 + 8);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int s = s0 + k;  // parity of s - sa = k & 1
      if (s <= sb) {
        hrow(cur[k], H[k & 1]);
        while (y < y_end && Y.sy1 == s) {
          if (Y.sy0 == s) emit(H[k & 1], H[k & 1]);
          else emit(H[(k & 1) ^ 1], H[k & 1]);
          y++;
          Y = yt[min(y, D.h - 1)];
        }
      }
    }
  };
  unsigned buf[2][8];
  load8(buf[0], sa);
  for (int s0 = sa; s0 <= sb; s0 += 16) {
    chunk(buf[0], buf[1], s0);
    if (s0 + 8 <= sb) chunk(buf[1], buf[0], s0 + 8);
  }
}

// ------------------------------------------------------------------------------------------------
// Gaussian blur 7x7, sigma 2, BORDER_REFLECT_101 -- OpenCV's 8.8 fixed-point path:
//   h = sum k_i * src[x+i-3]            (u16, <= 65280)
//   dst = (sum k_j * h[y+j-3] + 32768) >> 16
// One launch covers all levels of all frames, in packed integer math:
//   horizontal: per pixel the 12-byte window of its dword and two or three v_dot4_u32_u8 against the kernel bytes shifted
//               to where the pixel's taps fall; two rows are interleaved into (row 2m, row 2m+1) u16 pairs per pixel;
//   vertical:   four v_dot2_u32_u16 per pixel on those row pairs (weights chosen by the parity of the output row).
// (hblur_px, the DPP neighbour moves and the packed types: blur_math.hpp)

// ------------------------------------------------------------------------------------------------
// No LDS: a half-wave (32 lanes) owns a STRIP of up to 30 output dwords x 8 * BS_CH - 6 output rows and walks
// down its rows.  Lane j holds the dword at column d0 - 1 + j of the current row (lanes 0 and nd + 1 are the strip's
// halo), gets its neighbours' dwords with wave-shift DPP moves, blurs its 4 pixels horizontally (hblur_px) and keeps
// the last four ROW PAIRS of that in registers (u16 halves); every new pair completes two
// output rows.  The rows of the next chunk are in flight while a chunk is processed.  No barrier, no LDS traffic.
// REFLECT_101: rows by index arithmetic (only in chunks that touch the top / bottom), columns by a per-lane byte
// selector over (left neighbour's raw dword, own raw dword) -- the edge lanes load the dword that completes the pair.
// Both strips of a wave belong to one level, so the level's geometry and row offsets are scalar.
constexpr int BS_CH = ORBGPU_BLUR_STRIP_CHUNKS;  // chunks of 8 staged rows


__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, DeviceGeom g,
                                                     int n_frames, Level0Src lvl0, const short4 *__restrict__ strips,
                                                     int items_per_frame) {
  int item, frame;
  if (!xcd_item_frame(items_per_frame, n_frames, item, frame)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const short4 st = strips[(item * 4 + wave) * 2 + (lane >> 5)];  // (level, first output dword, first output row, dwords)
  const int l = __builtin_amdgcn_readfirstlane((int)st.x);
  const int w = g.lv[l].w, h = g.lv[l].h, pitch = g.lv[l].pitch;
  const uint8_t *src = level_src(pyr, g, lvl0, frame, l, g.lv[l].off);
  uint8_t *dst = blur + (size_t)frame * g.frame_bytes + g.lv[l].off;
  const int j = lane & 31, nd = st.w, y0 = st.z;
  const int d = st.y - 1 + j, gx = 4 * d;
  // which dword the lane loads and how its row dword is put together from (left neighbour's raw dword, own raw dword)
  int ld_off;
  unsigned sel;
  if (gx >= 0 && gx + 3 < w) {
    ld_off = gx;
    sel = 0x03020100u;
  } else if (gx < 0) {  // columns -4..-1 <- 4 3 2 1 (column -4 is never used)
    ld_off = 0;
    sel = 0x01020300u;
  } else {  // columns >= w <- 2(w-1) - column; all sources lie in [w-4, w-1], i.e. in the dwords at a and a + 4
    const int a = (w - 4) & ~3;
    const bool partial = (w & 3) && gx == a + 4;  // the lane of the last, partly filled dword: own = a + 4, left = a
    ld_off = partial ? a + 4 : a;                 // the others: own = a, left = a + 4 (or unused)
    sel = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      int col = gx + b;
      if (col >= w) col = 2 * (w - 1) - col;
      col = max(col, a) - a;  // 0..7; columns beyond the 3-px apron: unused, keep the selector in range
      const int idx = partial ? (col < 4 ? col + 4 : col - 4) : col;
      sel |= (unsigned)idx << (8 * b);
    }
  }
  unsigned keep = 0;  // bytes of this lane's dword that are image pixels
#pragma unroll
  for (int p = 0; p < 4; p++)
    if (gx + p < w) keep |= 0xffu << (8 * p);
  const bool writes = j >= 1 && j <= nd;
  const u16x2 we0 = {18, 34}, we1 = {48, 56}, we2 = {48, 34}, we3 = {18, 0};   // even output row of a pair window
  const u16x2 wo0 = {0, 18}, wo1 = {34, 48}, wo2 = {56, 48}, wo3 = {34, 18};   // odd
  // Buffer addressing: the level is the (scalar) base, a lane carries a 32-bit byte offset, the row step k * pitch of a
  // chunk's loads is a scalar offset -- no 64-bit address arithmetic per row.  Stores: a lane that writes nothing
  // (strip halo, beyond the strip) keeps an offset beyond the level, and so do the rows below the image: the
  // descriptor's range drops them, no compare and no branch per row.
  const unsigned level_bytes = (unsigned)h * (unsigned)pitch;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (int)level_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)level_bytes, 0x00020000);
  unsigned ooff = writes ? (unsigned)gx + (unsigned)y0 * (unsigned)pitch : 0x80000000u;  // the next output row's dword

  // byte offset of staged row r (image row y0 - 3 + r, reflected)
  auto row_off = [&](int r) -> unsigned {
    int gy = min(y0 - 3 + r, h + 2);
    gy = max(gy, -gy);
    gy = min(gy, 2 * (h - 1) - gy);
    return (unsigned)gy * (unsigned)pitch;
  };
  unsigned buf[2][8];  // the chunk being processed and the one in flight (the chunk loop is unrolled: no copies)
#pragma unroll
  for (int k = 0; k < 8; k++) buf[0][k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)ld_off + row_off(k), 0, 0);
  const unsigned c128 = 128u;
  unsigned P[4][4];  // the last four row pairs, horizontally blurred: pixel p of (row 2m, row 2m + 1) in u16 halves
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int p = 0; p < 4; p++) P[a][p] = 0;
#pragma unroll
  for (int c = 0; c < BS_CH; c++) {
    unsigned(&cur)[8] = buf[c & 1], (&nxt)[8] = buf[(c + 1) & 1];
    if (c + 1 < BS_CH) {
      // wave-uniform: does any lane's chunk touch the rows that are reflected?
      const int yb = y0 - 3 + 8 * (c + 1);
      const bool inside = __builtin_amdgcn_readfirstlane(__all(yb >= 0 && yb + 7 < h));
      if (inside) {
        const unsigned voff = (unsigned)ld_off + (unsigned)yb * (unsigned)pitch;
#pragma unroll
        for (int k = 0; k < 8; k++) nxt[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * pitch, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) nxt[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)ld_off + row_off(8 * (c + 1) + k), 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      unsigned hp[2][4];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const unsigned raw = cur[2 * k + q];
        const unsigned v = __builtin_amdgcn_perm(from_left(raw), raw, sel);
        const unsigned win[3] = {from_left(v), v, from_right(v)};
        hp[q][0] = hblur_px<0>(win, c128);
        hp[q][1] = hblur_px<1>(win, c128);
        hp[q][2] = hblur_px<2>(win, c128);
        hp[q][3] = hblur_px<3>(win, c128);
      }
#pragma unroll
      for (int p = 0; p < 4; p++) P[k][p] = hp[0][p] | (hp[1][p] << 16);
      // pair m = 4c + k completes output rows 2m - 6 (even) and 2m - 5 (odd): pairs m - 3 .. m
      const int m = 4 * c + k;
      if (m >= 3) {
        const unsigned(&A)[4] = P[(k + 1) & 3], (&B)[4] = P[(k + 2) & 3], (&C)[4] = P[(k + 3) & 3], (&D)[4] = P[k];
        unsigned ev[4], od[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
          ev[p] = __builtin_amdgcn_udot2(as_u16x2(A[p]), we0, 0u, false);  // the rounding term came with the rows (c128)
          ev[p] = __builtin_amdgcn_udot2(as_u16x2(B[p]), we1, ev[p], false);
          ev[p] = __builtin_amdgcn_udot2(as_u16x2(C[p]), we2, ev[p], false);
          ev[p] = __builtin_amdgcn_udot2(as_u16x2(D[p]), we3, ev[p], false);
          od[p] = __builtin_amdgcn_udot2(as_u16x2(A[p]), wo0, 0u, false);
          od[p] = __builtin_amdgcn_udot2(as_u16x2(B[p]), wo1, od[p], false);
          od[p] = __builtin_amdgcn_udot2(as_u16x2(C[p]), wo2, od[p], false);
          od[p] = __builtin_amdgcn_udot2(as_u16x2(D[p]), wo3, od[p], false);
        }
        // every accumulator is < 2^24: the result pixel is its byte 2
        const unsigned oe = (__builtin_amdgcn_perm(ev[1], ev[0], 0x0c0c0602u) |
                             __builtin_amdgcn_perm(ev[3], ev[2], 0x06020c0cu)) & keep;
        const unsigned oo = (__builtin_amdgcn_perm(od[1], od[0], 0x0c0c0602u) |
                             __builtin_amdgcn_perm(od[3], od[2], 0x06020c0cu)) & keep;
        // output rows y0 + 2m - 6 and the next
        __builtin_amdgcn_raw_buffer_store_b32(oe, rd, ooff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(oo, rd, ooff, pitch, 0);
        ooff += 2u * (unsigned)pitch;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Two or three SMALL pyramid levels in one launch.  The last levels of a VGA pyramid hold 25-70 k
// pixels: their launches are bound by ramp-up and by waiting for the predecessor, not by bytes.  Here a workgroup owns
// a 64 x 32 tile of the LAST level of the nest and everything that tile descends from: it stages the footprint of the
// nest's source level in LDS, builds the region of the next level it needs (plus the part of that level it OWNS: the
// levels in between are partitioned among the tiles by following the tiles' first columns / rows down the tables) into
// LDS and global memory, then the next from that.  Regions of neighbouring tiles overlap by the bilinear halo (one
// or two pixels per level); a pixel is computed from the same table entries whoever computes it, so the overlap is
// redundant work, not a difference.  The regions are computed on the host from the same tables (build_resize_nest).
// Same arithmetic as k_resize_level.
struct NestDev {
  int src_level, nl, ntx, nty;
  int pitch[3], off[3];   // LDS rows of the source region and of the first nl - 1 built regions
  int xoff[4], yoff[4];   // LDS offsets of the staged column / row coefficients of level src + q, q = 1 .. nl
  const short4 *xr, *yr;  // [tile column / row][q = 0 .. nl]: (first, last) of the region and of the owned part, level src + q
};

__global__ __launch_bounds__(256) void k_resize_nest(uint8_t *__restrict__ pyr, DeviceGeom g, NestDev a,
                                                     const XTab *__restrict__ xtab, const YTab *__restrict__ ytab,
                                                     int n_frames, Level0Src lvl0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t nest_lds[];
  int t, f;
  if (!xcd_item_frame(a.ntx * a.nty, n_frames, t, f)) return;
  const int tyi = t / a.ntx, txi = t - tyi * a.ntx;
  const short4 *xr = a.xr + txi * (a.nl + 1), *yr = a.yr + tyi * (a.nl + 1);
  {
    const LevelGeom S = g.lv[a.src_level];
    const uint8_t *src = level_src(pyr, g, lvl0, f, a.src_level, S.off);
    const short4 X = xr[0], Y = yr[0];
    const int ncols16 = (X.y - X.x) / 16 + 1, nrows = Y.y - Y.x + 1;
    const float inv = 1.0f / (float)ncols16;
    for (int i = threadIdx.x; i < nrows * ncols16; i += 256) {
      int r = (int)((float)i * inv);
      int c = i - r * ncols16;
      if (c < 0) { r--; c += ncols16; }
      if (c >= ncols16) { r++; c -= ncols16; }
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (X.x + 16 * c < S.pitch) v = *reinterpret_cast<const uint4 *>(src + (size_t)(Y.x + r) * S.pitch + X.x + 16 * c);
      *reinterpret_cast<uint4 *>(&nest_lds[a.off[0] + r * a.pitch[0] + 16 * c]) = v;
    }
    // the coefficients of every level's region, relative to the region below it: columns (sx0 - first column,
    // a0 | a1 << 16), rows (byte offsets of the two source rows, b0 | b1 << 16)
    for (int q = 1; q <= a.nl; q++) {
      const LevelGeom D = g.lv[a.src_level + q];
      const short4 Xq = xr[q], Yq = yr[q], PX = xr[q - 1], PY = yr[q - 1];
      const int ncol = ((Xq.y - Xq.x) / 4 + 1) * 4, nrow = Yq.y - Yq.x + 1;
      uint2 *xs = reinterpret_cast<uint2 *>(nest_lds + a.xoff[q]);
      uint4 *ys = reinterpret_cast<uint4 *>(nest_lds + a.yoff[q]);
      for (int i = threadIdx.x; i < ncol; i += 256) {
        const XTab e = xtab[D.xtab_off + min(Xq.x + i, D.w - 1)];
        xs[i] = make_uint2((unsigned)(e.sx0 - PX.x), (unsigned)(unsigned short)e.a0 | ((unsigned)(unsigned short)e.a1 << 16));
      }
      for (int i = threadIdx.x; i < nrow; i += 256) {
        const YTab e = ytab[D.ytab_off + Yq.x + i];
        ys[i] = make_uint4((unsigned)((e.sy0 - PY.x) * a.pitch[q - 1]), (unsigned)((e.sy1 - PY.x) * a.pitch[q - 1]),
                           (unsigned)(unsigned short)e.b0 | ((unsigned)(unsigned short)e.b1 << 16), 0u);
      }
    }
  }
  __syncthreads();
  const int cg0 = threadIdx.x & 31, rsub = threadIdx.x >> 5;  // 32 pixel quads x 8 rows per step
  for (int q = 1; q <= a.nl; q++) {
    const LevelGeom D = g.lv[a.src_level + q];
    const short4 X = xr[q], Y = yr[q];
    const uint8_t *in = nest_lds + a.off[q - 1];
    uint8_t *out = q < a.nl ? nest_lds + a.off[q] : nullptr;
    const int op = q < a.nl ? a.pitch[q] : 0;
    uint8_t *dst = pyr + (size_t)f * g.frame_bytes + D.off;
    const int ncg = (X.y - X.x) / 4 + 1, nrows = Y.y - Y.x + 1;  // X.x is a multiple of 4
    const uint2 *xs = reinterpret_cast<const uint2 *>(nest_lds + a.xoff[q]);
    const uint4 *ys = reinterpret_cast<const uint4 *>(nest_lds + a.yoff[q]);
    for (int cg = cg0; cg < ncg; cg += 32) {
      const int dx = X.x + 4 * cg;
      unsigned cx[4], ca0[4], ca1[4], keep = 0;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const uint2 e = xs[4 * cg + p];
        cx[p] = e.x;
        ca0[p] = e.y & 0xffffu;
        ca1[p] = e.y >> 16;
        if (dx + p < D.w) keep |= 0xffu << (8 * p);
      }
      const bool own_x = dx >= X.z && dx <= X.w;
      for (int r = rsub; r < nrows; r += 8) {
        const uint4 yy = ys[r];
        const uint8_t *r0 = in + yy.x, *r1 = in + yy.y;
        const unsigned b0 = yy.z & 0xffffu, b1 = yy.z >> 16;
        unsigned val = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const unsigned s00 = r0[cx[p]], s01 = r0[cx[p] + 1], s10 = r1[cx[p]], s11 = r1[cx[p] + 1];
          const unsigned h0 = mad24u(s01, ca1[p], mul24u(s00, ca0[p]));
          const unsigned h1 = mad24u(s11, ca1[p], mul24u(s10, ca0[p]));
          const unsigned v = ((mul24u(b0, h0 >> 4) >> 16) + (mul24u(b1, h1 >> 4) >> 16) + 2) >> 2;
          val |= (v & 255u) << (8 * p);
        }
        const int dy = Y.x + r;
        if (out) *reinterpret_cast<unsigned *>(out + r * op + 4 * cg) = val;
        if (own_x && dy >= Y.z && dy <= Y.w) *reinterpret_cast<unsigned *>(dst + (size_t)dy * D.pitch + dx) = val & keep;
      }
    }
    __syncthreads();
  }
}

}  // namespace

void launch_import_images(const orbgpu_ctx *c, const uint8_t *d_imgs, int64_t frame_stride, int stride, int n_frames) {
  const LevelGeom &L = c->geom.lv[0];
  int aligned16 = ((reinterpret_cast<uintptr_t>(d_imgs) | (uintptr_t)frame_stride | (uintptr_t)stride) & 15) == 0;
  dim3 grid((L.pitch / 16 + 255) / 256, L.h, n_frames);
  hipLaunchKernelGGL(k_import, grid, dim3(256), 0, c->stream, d_imgs, (long long)frame_stride, stride, c->d_pyr,
                     c->geom.frame_bytes, L.w, L.h, L.pitch, aligned16);
}


// regions of k_resize_nest for the levels src + 1 .. src + nl, tiles of NEST_TW x NEST_TH pixels of the last one.
// Per tile column (and, with the row tables, per tile row) and level q: the region [lo, hi] the workgroup builds and the
// part [own_lo, own_hi] of it that it writes to global memory.  The owned parts partition every level: their first
// columns follow the tiles' first columns down the tables (rounded down to a dword), and a region is the hull of what
// the next level's region reads and of the owned part.  Returns false when a region does not fit the 16-bit fields.
constexpr int NEST_TW = 64, NEST_TH = 32;
static void nest_axis(const DeviceGeom &g, int src, int nl, bool is_x, const XTab *xt, const YTab *yt, int tile,
                      std::vector<short> &tab, int *extent /* [nl + 1]: largest region extent of level src + q */) {
  auto size_of = [&](int q) { return is_x ? g.lv[src + q].w : g.lv[src + q].h; };
  auto lo_of = [&](int q, int v) {  // first source index (level src + q - 1) read by index v of level src + q
    return is_x ? xt[g.lv[src + q].xtab_off + v].sx0 : yt[g.lv[src + q].ytab_off + v].sy0;
  };
  auto hi_of = [&](int q, int v) {
    return is_x ? std::min(xt[g.lv[src + q].xtab_off + v].sx0 + 1, size_of(q - 1) - 1) : yt[g.lv[src + q].ytab_off + v].sy1;
  };
  const int n_last = size_of(nl), nt = (n_last + tile - 1) / tile;
  // chain of first owned indices: c[q][i], i = 0 .. nt
  std::vector<std::vector<int>> c(nl + 1, std::vector<int>(nt + 1));
  for (int i = 0; i <= nt; i++) c[nl][i] = std::min(i * tile, n_last);
  for (int q = nl - 1; q >= 1; q--) {
    for (int i = 0; i < nt; i++) {
      int v = i == 0 ? 0 : lo_of(q + 1, c[q + 1][i]);
      if (is_x) v &= ~3;
      c[q][i] = v;
    }
    c[q][nt] = size_of(q);
  }
  for (int q = 0; q <= nl; q++) extent[q] = 0;
  tab.assign((size_t)nt * (nl + 1) * 4, 0);
  for (int i = 0; i < nt; i++) {
    int lo = c[nl][i], hi = c[nl][i + 1] - 1;
    short *e = &tab[((size_t)i * (nl + 1) + nl) * 4];
    e[0] = (short)lo; e[1] = (short)hi; e[2] = (short)lo; e[3] = (short)hi;
    extent[nl] = std::max(extent[nl], hi - lo + 1);
    for (int q = nl - 1; q >= 0; q--) {
      int nlo = lo_of(q + 1, lo), nhi = hi_of(q + 1, hi);
      int olo = 0, ohi = -1;
      if (q >= 1) {
        olo = c[q][i];
        ohi = c[q][i + 1] - 1;
        if (ohi >= olo) { nlo = std::min(nlo, olo); nhi = std::max(nhi, ohi); }
        if (is_x) nlo &= ~3;
      } else if (is_x) {
        nlo &= ~15;
        nhi = xt[g.lv[src + 1].xtab_off + hi].sx0 + 1;  // unclamped: the staging loop guards the pitch, the weight there is 0
      }
      lo = nlo; hi = nhi;
      e = &tab[((size_t)i * (nl + 1) + q) * 4];
      e[0] = (short)lo; e[1] = (short)hi; e[2] = (short)olo; e[3] = (short)ohi;
      extent[q] = std::max(extent[q], hi - lo + 1);
    }
  }
}

bool build_resize_nest(const DeviceGeom &g, const XTab *xt, const YTab *yt, int src, int nl, ResizeNest &n,
                       std::vector<short> &xr, std::vector<short> &yr) {
  if (nl < 2 || nl > 3 || src < 0 || src + nl > g.levels - 1) return false;
  int ex[4] = {0, 0, 0, 0}, ey[4] = {0, 0, 0, 0};
  nest_axis(g, src, nl, true, xt, yt, NEST_TW, xr, ex);
  nest_axis(g, src, nl, false, xt, yt, NEST_TH, yr, ey);
  n.src_level = src;
  n.nl = nl;
  n.ntx = (g.lv[src + nl].w + NEST_TW - 1) / NEST_TW;
  n.nty = (g.lv[src + nl].h + NEST_TH - 1) / NEST_TH;
  size_t off = 0;
  for (int q = 0; q < nl; q++) {
    // a pixel quad beyond the region's last column still reads its (unused) sources: 16 bytes of slack per row
    n.pitch[q] = ((ex[q] + 15) & ~15) + 16;
    n.off[q] = (int)off;
    off += (size_t)n.pitch[q] * (ey[q] + 1);
  }
  for (int q = 1; q <= nl; q++) {  // staged coefficients: 8 bytes per column (quads rounded up), 16 per row
    off = (off + 15) & ~(size_t)15;
    n.xoff[q] = (int)off;
    off += (size_t)(ex[q] + 8) * 8;
    off = (off + 15) & ~(size_t)15;
    n.yoff[q] = (int)off;
    off += (size_t)(ey[q] + 1) * 16;
  }
  n.lds = off;
  return off <= 60 * 1024;
}

void launch_resize_nest(const orbgpu_ctx *c, FrameSlice fs, const ResizeNest &n) {
  NestDev a{};
  a.src_level = n.src_level; a.nl = n.nl; a.ntx = n.ntx; a.nty = n.nty;
  for (int q = 0; q < 3; q++) { a.pitch[q] = n.pitch[q]; a.off[q] = n.off[q]; }
  for (int q = 0; q < 4; q++) { a.xoff[q] = n.xoff[q]; a.yoff[q] = n.yoff[q]; }
  a.xr = reinterpret_cast<const short4 *>(n.d_xr);
  a.yr = reinterpret_cast<const short4 *>(n.d_yr);
  dim3 grid(n.ntx * n.nty * xcd_frames(fs.n));
  hipLaunchKernelGGL(k_resize_nest, grid, dim3(256), n.lds, fs.stream, c->d_pyr + (size_t)fs.f0 * c->geom.frame_bytes, c->geom, a,
                     c->d_xtab, c->d_ytab, fs.n, slice_lvl0(c, fs));
}

void launch_pyramid_level(const orbgpu_ctx *c, FrameSlice fs, int level) {
  const LevelGeom &D = c->geom.lv[level];
  if (c->rs_nest_of[level] >= 0) {  // built by a nest: launched with the nest's first level
    const ResizeNest &n = c->rs_nest[c->rs_nest_of[level]];
    if (level == n.src_level + 1) launch_resize_nest(c, fs, n);
    return;
  }
  if (c->rs_items[level] > 0) {
    dim3 grid(c->rs_items[level] * xcd_frames(fs.n));
    hipLaunchKernelGGL(k_resize_strips, grid, dim3(256), 0, fs.stream, c->d_pyr + (size_t)fs.f0 * c->geom.frame_bytes, c->geom,
                       level, c->d_rs_cols + c->rs_col_off[level], c->d_ytab,
                       reinterpret_cast<const short4 *>(c->d_rs_strips) + c->rs_strip_off[level],
                       c->rs_items[level], fs.n, slice_lvl0(c, fs));
    return;
  }
  dim3 grid(((D.w + TILE_W - 1) / TILE_W) * ((D.h + RS_TILE_H - 1) / RS_TILE_H) * xcd_frames(fs.n));
  const size_t lds = (size_t)c->rs_pitch[level] * c->rs_rows[level];
  hipLaunchKernelGGL(k_resize_level, grid, dim3(256), lds, fs.stream, c->d_pyr + (size_t)fs.f0 * c->geom.frame_bytes, c->geom,
                     level, c->d_xtab, c->d_ytab, c->rs_pitch[level], fs.n, slice_lvl0(c, fs));
}

// strips of one frame's level for k_resize_strips: (first output dword, first output row, output dwords, output rows),
// four per workgroup (padded with empty strips).  A wave walks its rows one after the other, so the small levels get
// shorter strips: more waves, each done sooner (the last levels of a VGA pyramid are bound by that walk, not by bytes).  Returns the workgroups per frame, or 0 when the level's column table does not
// fit the kernel (a pixel quad reaching more than 8 source bytes, or no strip width >= 8): k_resize_level then.
int build_resize_strips(const DeviceGeom &g, int level, const XTab *xt, std::vector<short> &table, int strip_target) {
  const LevelGeom &S = g.lv[level - 1], &D = g.lv[level];
  const int wd = D.pitch / 4;  // the padding dwords of a row are written too (as zeros)
  auto sx0 = [&](int x) { return xt[std::min(x, D.w - 1)].sx0; };
  for (int d = 0; d < wd; d++)
    if (sx0(4 * d + 3) - sx0(4 * d) > 6) return 0;
  // a strip's source dwords (first pixel's sx0 .. last pixel's sx0 + 1) must be 64 consecutive ones at most
  auto fits = [&](int d0, int nd) { return ((sx0(4 * (d0 + nd) - 1) + 1) >> 2) - (sx0(4 * d0) >> 2) <= 63; };
  int per = 0;
  for (int nsx = (wd + 63) / 64; nsx <= wd; nsx++) {
    per = (wd + nsx - 1) / nsx;
    bool ok = true;
    for (int d0 = 0; d0 < wd && ok; d0 += per) ok = fits(d0, std::min(per, wd - d0));
    if (ok) break;
    per = 0;
  }
  if (per < 8 || S.pitch < 4) return 0;
  const int nsx = (wd + per - 1) / per;
  int rows = 32;  // output rows per strip: at least strip_target strips per frame, down to 8 rows
  while (rows > 8 && nsx * ((D.h + rows - 1) / rows) < strip_target) rows /= 2;
  int n = 0;
  for (int y0 = 0; y0 < D.h; y0 += rows)
    for (int d0 = 0; d0 < wd; d0 += per, n++) {
      table.push_back((short)d0);
      table.push_back((short)y0);
      table.push_back((short)std::min(per, wd - d0));
      table.push_back((short)rows);
    }
  for (; n % 4; n++)
    for (int q = 0; q < 4; q++) table.push_back(0);
  return n / 4;
}

// the column table of k_resize_strips: per output dword of the level (padding dwords included) the first source dword,
// the byte offset of the first source pixel in it, per pixel the byte selector of (s[sx0], s[sx0 + 1]) in the aligned
// 8-byte window and the weights (a0, a1) << 4, and which of the dword's bytes are pixels -- what every strip's lanes
// used to work out from five XTab entries each
void build_resize_columns(const DeviceGeom &g, int level, const XTab *xt, std::vector<RsColumn> &table) {
  const LevelGeom &D = g.lv[level];
  auto X = [&](int x) -> const XTab & { return xt[std::min(x, D.w - 1)]; };
  for (int d = 0; d < D.pitch / 4; d++) {
    RsColumn c{};
    const XTab &X0 = X(4 * d);
    c.s0 = X0.sx0 >> 2;
    c.o = (unsigned)X0.sx0 & 3u;
    for (int p = 0; p < 4; p++) {
      const XTab &Xp = X(4 * d + p);
      const unsigned q = (unsigned)(Xp.sx0 - X0.sx0);  // byte of s[sx0] in the window aligned to the quad's first pixel; <= 6
      c.sel[p] = q | 0x0c00u | ((q + 1) << 16) | 0x0c000000u;
      c.coef[p] = ((unsigned)Xp.a0 << 4) | ((unsigned)Xp.a1 << 20);  // (a0, a1) << 4 as u16 halves, each <= 32768
      if (4 * d + p < D.w) c.keep |= 0xffu << (8 * p);
    }
    table.push_back(c);
  }
}

void launch_blur(const orbgpu_ctx *c, FrameSlice fs) {
  const size_t off = (size_t)fs.f0 * c->geom.frame_bytes;
  dim3 grid(c->n_blur_items * xcd_frames(fs.n));
  hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, fs.stream, c->d_pyr + off, c->d_blur + off, c->geom, fs.n, slice_lvl0(c, fs),
                     reinterpret_cast<const short4 *>(c->d_blur_strips), c->n_blur_items);
}

// strips of one frame for k_blur: (level, first output dword, first output row, output dwords <= 30), two per
// wave and both of one level (levels padded to an even count, the table to whole workgroups of 8, with empty strips)
int build_blur_strips(const DeviceGeom &g, std::vector<short> &table) {
  table.clear();
  const int R = 8 * ORBGPU_BLUR_STRIP_CHUNKS - 6;
  auto push = [&](int l, int d0, int y0, int nd) {
    table.push_back((short)l);
    table.push_back((short)d0);
    table.push_back((short)y0);
    table.push_back((short)nd);
  };
  for (int l = 0; l < g.levels; l++) {
    const int wd = (g.lv[l].w + 3) / 4, nsx = (wd + 29) / 30, per = (wd + nsx - 1) / nsx;
    int n = 0;
    for (int y0 = 0; y0 < g.lv[l].h; y0 += R)
      for (int d0 = 0; d0 < wd; d0 += per, n++) push(l, d0, y0, std::min(per, wd - d0));
    if (n & 1) push(l, 0, 0, 0);
  }
  while ((table.size() / 4) % 8) push(g.levels - 1, 0, 0, 0);
  return (int)(table.size() / 32);
}
