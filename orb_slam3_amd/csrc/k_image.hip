// Image-plane kernels: import, pyramid level (cv::resize INTER_LINEAR restated in fixed point) and the
// 7x7 Gaussian blur.  gfx950 only.
//
// Reference stages: BuildImagePyramid  src/features/orb_feature_extractor.cpp:504-528 (cv::resize :517)
//                   blur in Extract    src/features/orb_feature_extractor.cpp:558-560 (cv::GaussianBlur)
#include "orbgpu_internal.h"
#include "tile_stage.hpp"

namespace {

constexpr int TILE_W = 128;  // output pixels per tile row (32 lanes x 4 px: one LDS row per half-wave)
constexpr int TILE_DW = STAGE_DW;  // dwords per staged row: 4-px apron left and right

__device__ __forceinline__ int reflect101(int i, int n) {
  // BORDER_REFLECT_101: -i -> i, (n-1)+i -> (n-1)-i ; the aprons used here are < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

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
// Gaussian blur 7x7, sigma 2, BORDER_REFLECT_101 -- OpenCV's 8.8 fixed-point path:
//   h = sum k_i * src[x+i-3]            (u16, <= 65280)
//   dst = (sum k_j * h[y+j-3] + 32768) >> 16
// One launch covers all levels of all frames.  Tile 128 x 40, apron 3 rows / 4 columns, both passes in LDS and in
// packed integer math (the integer VALU, not HBM, is what bounds this kernel):
//   horizontal: per pixel two byte-aligned dwords of the row (v_alignbyte_b32) and two v_dot4_u32_u8 against the
//               kernel bytes; two rows per lane are interleaved so that LDS holds (row 2m, row 2m+1) pairs per pixel;
//   vertical:   four v_dot2_u32_u16 per pixel on those row pairs (weights chosen by the parity of the output row).
constexpr int BT_H = ORBGPU_BLUR_TILE_H;
constexpr int BT_ROWS = BT_H + 6;        // 46 staged rows = 23 row pairs
constexpr int BT_PAIRS = BT_ROWS / 2;

typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;
__device__ __forceinline__ u16x2 as_u16x2(unsigned v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ unsigned as_u32(u16x2 v) { return __builtin_bit_cast(unsigned, v); }

// horizontal 7-tap of pixel P (0..3) of the middle dword of the 12-byte window {w0,w1,w2}: the pixel's first tap is byte
// 1 + P of the window, so the 7 taps are spread over two or three ALIGNED dwords; each takes one v_dot4_u32_u8 against
// the kernel (18 34 48 56 48 34 18) shifted to where its bytes fall -- no byte realignment.  <= 65280
template <int P>
__device__ __forceinline__ unsigned hblur_px(const unsigned (&w)[3]) {
  constexpr unsigned long long K = 18ull | (34ull << 8) | (48ull << 16) | (56ull << 24) | (48ull << 32) | (34ull << 40) |
                                   (18ull << 48);
  // byte j of the 12-byte window carries tap j - (1 + P): the kernel as a 96-bit constant shifted left by 1 + P bytes
  constexpr int SH = 8 * (1 + P);
  constexpr unsigned K0 = (unsigned)(K << SH), K1 = (unsigned)(SH == 32 ? K : (K >> (32 - SH))),
                     K2 = (unsigned)(SH == 32 ? (K >> 32) : (K >> (64 - SH)));
  unsigned acc = 0;
  if (K0) acc = __builtin_amdgcn_udot4(w[0], K0, acc, false);
  acc = __builtin_amdgcn_udot4(w[1], K1, acc, false);
  if (K2) acc = __builtin_amdgcn_udot4(w[2], K2, acc, false);
  return acc;
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, DeviceGeom g,
                                              int n_frames, Level0Src lvl0, const short4 *__restrict__ tiles) {
  __shared__ unsigned tile[BT_ROWS * TILE_DW];
  __shared__ __attribute__((aligned(16))) unsigned hbuf[BT_PAIRS * TILE_W];  // (h[2m][x], h[2m+1][x]) per dword
  // tile -> (level, tile x, tile y)
  int t, frame;
  if (!xcd_item_frame(g.btiles_per_frame, n_frames, t, frame)) return;
  const short4 bt = tiles[t];
  const int l = bt.x, x0 = bt.y, y0 = bt.z;
  const LevelGeom L = g.lv[l];
  const size_t base = (size_t)frame * g.frame_bytes + L.off;
  const uint8_t *src = level_src(pyr, g, lvl0, frame, l, L.off);

  stage_tile<BT_ROWS>(tile, src, L.w, L.h, L.pitch, x0, y0);
  __syncthreads();

  // horizontal pass: task = (row pair, 4-pixel group)
  for (int i = threadIdx.x; i < BT_PAIRS * (TILE_W / 4); i += 256) {
    const int rp = i >> 5, c = i & 31;
    unsigned w0[3], w1[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      w0[q] = tile[(2 * rp) * TILE_DW + c + q];
      w1[q] = tile[(2 * rp + 1) * TILE_DW + c + q];
    }
    // window bytes 0..11 = columns 4c-4 .. 4c+7 (relative to x0); pixel p has its centre at byte 4+p, first tap 1+p
    uint4 o;  // per pixel (row 2rp, row 2rp+1) as u16 halves
    o.x = hblur_px<0>(w0) | (hblur_px<0>(w1) << 16);
    o.y = hblur_px<1>(w0) | (hblur_px<1>(w1) << 16);
    o.z = hblur_px<2>(w0) | (hblur_px<2>(w1) << 16);
    o.w = hblur_px<3>(w0) | (hblur_px<3>(w1) << 16);
    *reinterpret_cast<uint4 *>(&hbuf[rp * TILE_W + 4 * c]) = o;
  }
  __syncthreads();

  // vertical pass: staged rows ry .. ry+6 of output row ry.  Even ry: pairs (ry,ry+1)(ry+2,ry+3)(ry+4,ry+5)(ry+6,-);
  // odd ry: (-,ry)(ry+1,ry+2)(ry+3,ry+4)(ry+5,ry+6).  dot2 = a.lo*b.lo + a.hi*b.hi + c with a.lo = even staged row.
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bool odd = ty & 1;
  const u16x2 wt0 = odd ? u16x2{0, 18} : u16x2{18, 34};
  const u16x2 wt1 = odd ? u16x2{34, 48} : u16x2{48, 56};
  const u16x2 wt2 = odd ? u16x2{56, 48} : u16x2{48, 34};
  const u16x2 wt3 = odd ? u16x2{34, 18} : u16x2{18, 0};
  unsigned keep = 0;  // bytes of this lane's dword that are image pixels
#pragma unroll
  for (int p = 0; p < 4; p++)
    if (x0 + 4 * tx + p < L.w) keep |= 0xffu << (8 * p);
#pragma unroll
  for (int part = 0; part < BT_H / 8; part++) {
    const int ry = ty + 8 * part;
    const int y = y0 + ry, x = x0 + 4 * tx;
    if (y >= L.h || x >= L.pitch) continue;
    const int rp0 = ry >> 1;
    const uint4 p0 = *reinterpret_cast<const uint4 *>(&hbuf[(rp0 + 0) * TILE_W + 4 * tx]);
    const uint4 p1 = *reinterpret_cast<const uint4 *>(&hbuf[(rp0 + 1) * TILE_W + 4 * tx]);
    const uint4 p2 = *reinterpret_cast<const uint4 *>(&hbuf[(rp0 + 2) * TILE_W + 4 * tx]);
    const uint4 p3 = *reinterpret_cast<const uint4 *>(&hbuf[(rp0 + 3) * TILE_W + 4 * tx]);
    const unsigned q0[4] = {p0.x, p0.y, p0.z, p0.w}, q1[4] = {p1.x, p1.y, p1.z, p1.w};
    const unsigned q2[4] = {p2.x, p2.y, p2.z, p2.w}, q3[4] = {p3.x, p3.y, p3.z, p3.w};
    unsigned acc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      acc[p] = __builtin_amdgcn_udot2(as_u16x2(q0[p]), wt0, 32768u, false);
      acc[p] = __builtin_amdgcn_udot2(as_u16x2(q1[p]), wt1, acc[p], false);
      acc[p] = __builtin_amdgcn_udot2(as_u16x2(q2[p]), wt2, acc[p], false);
      acc[p] = __builtin_amdgcn_udot2(as_u16x2(q3[p]), wt3, acc[p], false);
    }
    // acc < 2^24: the result pixel is byte 2 of each accumulator
    const unsigned lo = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u);
    const unsigned hi = __builtin_amdgcn_perm(acc[3], acc[2], 0x0c0c0602u);
    const unsigned out = (lo | (hi << 16)) & keep;
    *reinterpret_cast<unsigned *>(blur + base + (size_t)y * L.pitch + x) = out;
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

void launch_pyramid_level(const orbgpu_ctx *c, FrameSlice fs, int level) {
  const LevelGeom &D = c->geom.lv[level];
  dim3 grid(((D.w + TILE_W - 1) / TILE_W) * ((D.h + RS_TILE_H - 1) / RS_TILE_H) * xcd_frames(fs.n));
  const size_t lds = (size_t)c->rs_pitch[level] * c->rs_rows[level];
  hipLaunchKernelGGL(k_resize_level, grid, dim3(256), lds, fs.stream, c->d_pyr + (size_t)fs.f0 * c->geom.frame_bytes, c->geom,
                     level, c->d_xtab, c->d_ytab, c->rs_pitch[level], fs.n, slice_lvl0(c, fs));
}

void launch_blur(const orbgpu_ctx *c, FrameSlice fs) {
  dim3 grid(c->geom.btiles_per_frame * xcd_frames(fs.n));
  const size_t off = (size_t)fs.f0 * c->geom.frame_bytes;
  hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, fs.stream, c->d_pyr + off, c->d_blur + off, c->geom, fs.n, slice_lvl0(c, fs),
                     reinterpret_cast<const short4 *>(c->d_blur_tiles));
}
