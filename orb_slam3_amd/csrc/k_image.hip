// Image-plane kernels: import, pyramid level (cv::resize INTER_LINEAR restated in fixed point) and the
// 7x7 Gaussian blur.  gfx950 only.
//
// Reference stages: BuildImagePyramid  src/features/orb_feature_extractor.cpp:504-528 (cv::resize :517)
//                   blur in Extract    src/features/orb_feature_extractor.cpp:558-560 (cv::GaussianBlur)
#include "orbgpu_internal.h"

namespace {

constexpr int TILE_W = 128;  // output pixels per tile row (32 lanes x 4 px: one LDS row per half-wave)
constexpr int TILE_H = 16;
constexpr int TILE_DW = TILE_W / 4 + 2;  // dwords per staged row: 4-px apron left and right

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
// One workgroup produces a 128 x 8 output tile; the two source-row bands it needs (at most
// ceil(8*1.2)+2 rows of at most 128*1.2+2 bytes) are staged in LDS with coalesced row loads.
constexpr int RS_TILE_H = 8;
constexpr int RS_MAX_ROWS = 24;   // source rows staged per tile (scale <= 2.5)
constexpr int RS_MAX_COLS = 336;  // source bytes staged per row (scale <= 2.5), multiple of 16

__global__ __launch_bounds__(256) void k_resize_level(uint8_t *__restrict__ pyr, DeviceGeom g, int level,
                                                      const XTab *__restrict__ xtab, const YTab *__restrict__ ytab) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[RS_MAX_ROWS * RS_MAX_COLS];
  const LevelGeom S = g.lv[level - 1], D = g.lv[level];
  const int f = blockIdx.z;
  const uint8_t *src = pyr + (size_t)f * g.frame_bytes + S.off;
  uint8_t *dst = pyr + (size_t)f * g.frame_bytes + D.off;
  const XTab *xt = xtab + D.xtab_off;
  const YTab *yt = ytab + D.ytab_off;

  const int dx0 = blockIdx.x * TILE_W, dy0 = blockIdx.y * RS_TILE_H;
  const int dx1 = min(dx0 + TILE_W, D.w) - 1, dy1 = min(dy0 + RS_TILE_H, D.h) - 1;
  // source footprint of the tile (tables are monotone)
  const int sxa = xt[dx0].sx0 & ~15;  // 16-byte aligned start
  const int sxb = xt[dx1].sx1;
  const int sya = yt[dy0].sy0, syb = yt[dy1].sy1;
  const int ncols16 = (sxb - sxa) / 16 + 1;
  const int nrows = syb - sya + 1;
  const bool fits = ncols16 * 16 <= RS_MAX_COLS && nrows <= RS_MAX_ROWS;
  if (fits) {
    for (int i = threadIdx.x; i < nrows * ncols16; i += 256) {
      int r = i / ncols16, c = i - r * ncols16;
      // pitch is a multiple of 16 and slabs are padded, so the 16-byte read stays inside the slab
      uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)(sya + r) * S.pitch + sxa + 16 * c);
      *reinterpret_cast<uint4 *>(&lds[r * RS_MAX_COLS + 16 * c]) = v;
    }
  }
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads, 4 px each
  const int dy = dy0 + ty;
  const int dx = dx0 + 4 * tx;
  if (dy >= D.h || dx >= D.pitch) return;
  const YTab Y = yt[dy];
  unsigned out = 0;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    int x = dx + p;
    if (x < D.w) {
      const XTab X = xt[x];
      int s00, s01, s10, s11;
      if (fits) {
        const int r0 = (Y.sy0 - sya) * RS_MAX_COLS - sxa;
        const int r1 = (Y.sy1 - sya) * RS_MAX_COLS - sxa;
        s00 = lds[r0 + X.sx0]; s01 = lds[r0 + X.sx1]; s10 = lds[r1 + X.sx0]; s11 = lds[r1 + X.sx1];
      } else {
        const uint8_t *r0 = src + (size_t)Y.sy0 * S.pitch;
        const uint8_t *r1 = src + (size_t)Y.sy1 * S.pitch;
        s00 = r0[X.sx0]; s01 = r0[X.sx1]; s10 = r1[X.sx0]; s11 = r1[X.sx1];
      }
      int h0 = s00 * X.a0 + s01 * X.a1;
      int h1 = s10 * X.a0 + s11 * X.a1;
      int v = ((((int)Y.b0 * (h0 >> 4)) >> 16) + (((int)Y.b1 * (h1 >> 4)) >> 16) + 2) >> 2;
      out |= (unsigned)(v & 255) << (8 * p);
    }
  }
  *reinterpret_cast<unsigned *>(dst + (size_t)dy * D.pitch + dx) = out;
}

// ------------------------------------------------------------------------------------------------
// Gaussian blur 7x7, sigma 2, BORDER_REFLECT_101 -- OpenCV's 8.8 fixed-point path:
//   h = sum k_i * src[x+i-3]            (u16, <= 65280)
//   dst = (sum k_j * h[y+j-3] + 32768) >> 16
// One launch covers all levels of all frames.  Tile 128 x 16, apron 3 rows / 4 columns; the u8 tile and
// the u16 horizontal pass both live in LDS.
__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, DeviceGeom g) {
  __shared__ unsigned tile[(TILE_H + 6) * TILE_DW];
  __shared__ __attribute__((aligned(8))) unsigned short hbuf[(TILE_H + 6) * TILE_W];
  // tile -> (level, tile x, tile y)
  int t = blockIdx.x, l = 0;
#pragma unroll 1
  while (l + 1 < g.levels && t >= g.lv[l + 1].tile_off) l++;
  const LevelGeom L = g.lv[l];
  t -= L.tile_off;
  const int tiles_x = (L.w + TILE_W - 1) / TILE_W;
  const int x0 = (t % tiles_x) * TILE_W, y0 = (t / tiles_x) * TILE_H;
  const size_t base = (size_t)blockIdx.y * g.frame_bytes + L.off;
  const uint8_t *src = pyr + base;

  for (int i = threadIdx.x; i < (TILE_H + 6) * TILE_DW; i += 256) {
    int r = i / TILE_DW, c = i - r * TILE_DW;
    int gy = reflect101(y0 - 3 + r, L.h);
    int gx = x0 - 4 + 4 * c;
    unsigned v;
    if (gx >= 0 && gx + 3 < L.w) {
      v = *reinterpret_cast<const unsigned *>(src + (size_t)gy * L.pitch + gx);
    } else {
      v = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        int xx = gx + b;
        // columns further than the 3-px apron from the image are never used; clamp keeps them in range
        xx = reflect101(max(min(xx, L.w + 2), -3), L.w);
        v |= (unsigned)src[(size_t)gy * L.pitch + xx] << (8 * b);
      }
    }
    tile[i] = v;
  }
  __syncthreads();

  // horizontal pass over all 22 staged rows
  for (int i = threadIdx.x; i < (TILE_H + 6) * (TILE_W / 4); i += 256) {
    int r = i >> 5, c = i & 31;
    const unsigned *row = &tile[r * TILE_DW + c];
    unsigned long long lo = row[0] | ((unsigned long long)row[1] << 32);
    unsigned hi = row[2];
    // bytes 0..11 of the window = columns 4c-4 .. 4c+7 (relative to x0); pixel p centre = byte 4+p
    unsigned b[12];
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = (unsigned)(lo >> (8 * k)) & 255u;
#pragma unroll
    for (int k = 0; k < 4; k++) b[8 + k] = (hi >> (8 * k)) & 255u;
    unsigned hv[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      hv[p] = 18u * (b[p + 1] + b[p + 7]) + 34u * (b[p + 2] + b[p + 6]) + 48u * (b[p + 3] + b[p + 5]) + 56u * b[p + 4];
    }
    uint2 w2 = make_uint2(hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16));
    *reinterpret_cast<uint2 *>(&hbuf[r * TILE_W + 4 * c]) = w2;
  }
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    const int ry = ty + 8 * half;
    const int y = y0 + ry, x = x0 + 4 * tx;
    if (y >= L.h || x >= L.pitch) continue;
    unsigned acc[4] = {32768u, 32768u, 32768u, 32768u};
    const unsigned kw[7] = {18u, 34u, 48u, 56u, 48u, 34u, 18u};
#pragma unroll
    for (int j = 0; j < 7; j++) {
      uint2 hv = *reinterpret_cast<const uint2 *>(&hbuf[(ry + j) * TILE_W + 4 * tx]);
      acc[0] += kw[j] * (hv.x & 0xffffu);
      acc[1] += kw[j] * (hv.x >> 16);
      acc[2] += kw[j] * (hv.y & 0xffffu);
      acc[3] += kw[j] * (hv.y >> 16);
    }
    unsigned out = 0;
#pragma unroll
    for (int p = 0; p < 4; p++)
      if (x + p < L.w) out |= (acc[p] >> 16) << (8 * p);
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

void launch_pyramid_level(const orbgpu_ctx *c, int n_frames, int level) {
  const LevelGeom &D = c->geom.lv[level];
  dim3 grid((D.w + TILE_W - 1) / TILE_W, (D.h + RS_TILE_H - 1) / RS_TILE_H, n_frames);
  hipLaunchKernelGGL(k_resize_level, grid, dim3(256), 0, c->stream, c->d_pyr, c->geom, level, c->d_xtab, c->d_ytab);
}

void launch_blur(const orbgpu_ctx *c, int n_frames) {
  dim3 grid(c->geom.tiles_per_frame, n_frames);
  hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, c->stream, c->d_pyr, c->d_blur, c->geom);
}
