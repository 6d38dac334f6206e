// FAST-9/16 kernels: threshold-independent score map, per-cell NMS + threshold fallback + compaction,
// and the dense re-pack of the candidate sub-pools that the host quadtree reads.  gfx950 only.
//
// Reference stage: ComputeKeyPointsOctTree cell loop, src/features/orb_feature_extractor.cpp:227-276
// (cv::FAST :259-266).  Closed form used:
//   d_k = v - ring_k,  M_dark = max over the 16 arcs of min over 9 contiguous d,
//   M_bright = max over arcs of min(-d),  score = max(M_dark, M_bright) - 1,  corner at t  <=>  score >= t.
#include "orbgpu_internal.h"

namespace {

constexpr int TILE_W = 128;
constexpr int TILE_H = 16;
constexpr int TILE_DW = TILE_W / 4 + 2;

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

// One launch covers all levels of all frames: blockIdx.x = tile over (level, ty, tx), blockIdx.y = frame.
// Tile 128 x 16 output pixels, staged with a 3-row / 4-column apron as dwords; each thread scores
// 4 adjacent pixels of 2 rows from 3 dwords x 7 rows held in registers.
__global__ __launch_bounds__(256) void k_fast_score(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ score,
                                                    DeviceGeom g) {
  __shared__ unsigned tile[(TILE_H + 6) * TILE_DW];
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
    int gy = y0 - 3 + r, gx = x0 - 4 + 4 * c;
    unsigned v = 0;
    // pixels outside the image are never part of a scored ring (scores exist for [3,w-4]x[3,h-4] only)
    if (gy >= 0 && gy < L.h && gx >= 0 && gx < L.pitch) v = *reinterpret_cast<const unsigned *>(src + (size_t)gy * L.pitch + gx);
    tile[i] = v;
  }
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int min_thr = g.fast_min;
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
    const int ry = ty + 8 * half;
    const int y = y0 + ry, x = x0 + 4 * tx;
    if (y >= L.h || x >= L.pitch) continue;
    unsigned out = 0;
    if (y >= 3 && y <= L.h - 4) {
      // 12-byte windows (columns x-4 .. x+7) of the 7 rows y-3 .. y+3
      unsigned w[7][3];
#pragma unroll
      for (int j = 0; j < 7; j++) {
#pragma unroll
        for (int q = 0; q < 3; q++) w[j][q] = tile[(ry + j) * TILE_DW + tx + q];
      }
#define PX(j, b) ((w[j][(b) >> 2] >> (8 * ((b) & 3))) & 255u)
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int xx = x + p;
        if (xx < 3 || xx > L.w - 4) continue;
        const int cb = 4 + p;  // byte column of the centre inside the window
        int r[16];
        // ring (dx,dy): (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
        r[0] = PX(6, cb);      r[1] = PX(6, cb + 1);  r[2] = PX(5, cb + 2);  r[3] = PX(4, cb + 3);
        r[4] = PX(3, cb + 3);  r[5] = PX(2, cb + 3);  r[6] = PX(1, cb + 2);  r[7] = PX(0, cb + 1);
        r[8] = PX(0, cb);      r[9] = PX(0, cb - 1);  r[10] = PX(1, cb - 2); r[11] = PX(2, cb - 3);
        r[12] = PX(3, cb - 3); r[13] = PX(4, cb - 3); r[14] = PX(5, cb - 2); r[15] = PX(6, cb - 1);
        int s = fast_score16((int)PX(3, cb), r, min_thr);
        out |= (unsigned)s << (8 * p);
      }
#undef PX
    }
    *reinterpret_cast<unsigned *>(score + base + (size_t)y * L.pitch + x) = out;
  }
}

// ------------------------------------------------------------------------------------------------
// One wave per FAST cell.  The cell's detection rectangle (the pixels cv::FAST scans inside the
// reference's ROI: x in [19 + j*wCell, min(18 + (j+1)*wCell, w-20)], likewise y) is staged
// in LDS with a zero halo: NMS does not see across the seam.  Threshold fallback is decided per cell
// AFTER NMS (orb_feature_extractor.cpp:263-266).  Candidates leave in raster order.
constexpr int CELL_MAX = 64;  // wCell, hCell <= 59 by construction (ceil(w/floor(w/30)))
constexpr int CELL_PITCH = CELL_MAX + 4;

__device__ __forceinline__ bool nms_keep(const uint8_t *s, int idx, int thr) {
  int v = s[idx];
  if (v < thr) return false;
  bool keep = true;
#pragma unroll
  for (int dy = -1; dy <= 1; dy++)
#pragma unroll
    for (int dx = -1; dx <= 1; dx++) {
      if (dx == 0 && dy == 0) continue;
      int n = s[idx + dy * CELL_PITCH + dx];
      n = n >= thr ? n : 0;
      keep = keep && (v > n);
    }
  return keep;
}

__global__ __launch_bounds__(64) void k_cell_detect(const uint8_t *__restrict__ score, DeviceGeom g,
                                                    CellEntry *__restrict__ cells, unsigned *__restrict__ cands,
                                                    int *__restrict__ cand_count) {
  __shared__ uint8_t s[(CELL_MAX + 2) * CELL_PITCH];
  const int f = blockIdx.y;
  int c = blockIdx.x, l = 0;
#pragma unroll 1
  while (l + 1 < g.levels && c >= g.lv[l + 1].cell_off) l++;
  const LevelGeom L = g.lv[l];
  const int ci = c - L.cell_off;
  const int i = ci / L.ncols, j = ci - i * L.ncols;
  const int x0 = ORB_EDGE + j * L.wcell, x1 = min(ORB_EDGE - 1 + (j + 1) * L.wcell, L.w - ORB_EDGE - 1);
  const int y0 = ORB_EDGE + i * L.hcell, y1 = min(ORB_EDGE - 1 + (i + 1) * L.hcell, L.h - ORB_EDGE - 1);
  const int rw = x1 - x0 + 1, rh = y1 - y0 + 1;
  const int lane = threadIdx.x;
  CellEntry *ce = cells + (size_t)f * g.cells_per_frame + c;
  if (rw <= 0 || rh <= 0) {
    if (lane == 0) *ce = CellEntry{0, 0};
    return;
  }
  const uint8_t *src = score + (size_t)f * g.frame_bytes + L.off;
  for (int k = lane; k < (rh + 2) * CELL_PITCH; k += 64) {
    int yy = k / CELL_PITCH, xx = k - yy * CELL_PITCH;
    uint8_t v = 0;
    if (yy >= 1 && yy <= rh && xx >= 1 && xx <= rw) v = src[(size_t)(y0 + yy - 1) * L.pitch + x0 + xx - 1];
    s[k] = v;
  }
  __syncthreads();

  const int npx = rw * rh;
  int thr = g.fast_init;
  int total = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    for (int k0 = 0; k0 < npx; k0 += 64) {
      int k = k0 + lane;
      bool keep = false;
      if (k < npx) {
        int yy = k / rw, xx = k - yy * rw;
        keep = nms_keep(s, (yy + 1) * CELL_PITCH + xx + 1, thr);
      }
      total += __popcll(__ballot(keep));
    }
    if (total > 0 || thr == g.fast_min) break;
    thr = g.fast_min;
  }
  int basei = 0;
  if (lane == 0) {
    if (total > 0) basei = atomicAdd(&cand_count[f * g.levels + l], total);
    *ce = CellEntry{basei, total};
  }
  if (total == 0) return;
  basei = __shfl(basei, 0);
  unsigned *out = cands + (size_t)f * g.cands_per_frame + L.cand_off + basei;
  int run = 0;
  for (int k0 = 0; k0 < npx; k0 += 64) {
    int k = k0 + lane;
    bool keep = false;
    int yy = 0, xx = 0;
    if (k < npx) {
      yy = k / rw;
      xx = k - yy * rw;
      keep = nms_keep(s, (yy + 1) * CELL_PITCH + xx + 1, thr);
    }
    unsigned long long m = __ballot(keep);
    if (keep) {
      int pos = run + __popcll(m & ((1ull << lane) - 1ull));
      out[pos] = pack_cand(x0 + xx - ORB_MIN_BORDER, y0 + yy - ORB_MIN_BORDER, s[(yy + 1) * CELL_PITCH + xx + 1]);
    }
    run += __popcll(m);
  }
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of the (frame, level) candidate counts, then a dense copy of the used part of every
// sub-pool: the host quadtree downloads one contiguous buffer.
__global__ __launch_bounds__(1024) void k_scan_counts(const int *__restrict__ count, int n, int *__restrict__ prefix) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per;
  int sum = 0;
  for (int i = 0; i < per; i++)
    if (b + i < n) sum += count[b + i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - sum;
  for (int i = 0; i < per; i++)
    if (b + i < n) {
      prefix[b + i] = run;
      run += count[b + i];
    }
  if (threadIdx.x == 1023) prefix[n] = part[1023];
}

__global__ __launch_bounds__(256) void k_pack_cands(const unsigned *__restrict__ cands, DeviceGeom g,
                                                    const int *__restrict__ count, const int *__restrict__ prefix,
                                                    unsigned *__restrict__ packed) {
  const int l = blockIdx.x, f = blockIdx.y;
  const int n = count[f * g.levels + l];
  const unsigned *src = cands + (size_t)f * g.cands_per_frame + g.lv[l].cand_off;
  unsigned *dst = packed + prefix[f * g.levels + l];
  for (int i = blockIdx.z * 256 + threadIdx.x; i < n; i += 256 * gridDim.z) dst[i] = src[i];
}

}  // namespace

void launch_fast_score(const orbgpu_ctx *c, int n_frames) {
  dim3 grid(c->geom.tiles_per_frame, n_frames);
  hipLaunchKernelGGL(k_fast_score, grid, dim3(256), 0, c->stream, c->d_pyr, c->d_score, c->geom);
}

void launch_cell_detect(const orbgpu_ctx *c, int n_frames) {
  (void)hipMemsetAsync(c->d_cand_count, 0, sizeof(int) * (size_t)n_frames * c->geom.levels, c->stream);
  dim3 grid(c->geom.cells_per_frame, n_frames);
  hipLaunchKernelGGL(k_cell_detect, grid, dim3(64), 0, c->stream, c->d_score, c->geom, c->d_cells, c->d_cands,
                     c->d_cand_count);
}

void launch_pack_cands(const orbgpu_ctx *c, int n_frames) {
  const int n = n_frames * c->geom.levels;
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, c->stream, c->d_cand_count, n, c->d_cand_prefix);
  dim3 grid(c->geom.levels, n_frames, 4);
  hipLaunchKernelGGL(k_pack_cands, grid, dim3(256), 0, c->stream, c->d_cands, c->geom, c->d_cand_count,
                     c->d_cand_prefix, c->d_cands_packed);
}
