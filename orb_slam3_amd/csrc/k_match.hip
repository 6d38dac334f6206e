// Hamming matching kernels: brute-force / windowed second-nearest-neighbour search, per-'from'
// uniqueness, batched frame-to-frame linking and the rotation-histogram filter.  gfx950 only.
//
// Reference: ComputeDistance            src/features/orb_feature_extractor.cpp:83-98
//            SNNMatcher::MatchWithIterators src/features/matching/second_nearest_neighbor_matcher.hpp:30-80
//            Features::ListFeaturesInArea   src/features/features.cpp:81-125 (linear scan)
//            AreaToIterator/AreaToPointee   src/features/matching/iterators/area_to_iterator.h:24-43, area_to_pointee.h:43-54
//            OrientationValidator::Validate src/features/matching/validators/orientation_validator.cpp:16-97
//
// Layout of the search: every lane owns ONE 'to' descriptor in 8 VGPRs; the 'from' descriptors are
// wave-uniform and arrive through scalar loads (8 SGPRs each), so a pair costs 8 v_xor + 8 v_bcnt
// (popcount-accumulate) and three integer min/max on a packed (distance << 23 | index) key.  Because the
// key orders by distance first and index second, "first candidate wins a tie" (strict < at :50) and
// "a tie with the best becomes the second best" (:54) fall out of plain unsigned min / median.
#include <climits>

#include "orbgpu_internal.h"
#include "tile_stage.hpp"

namespace {

constexpr unsigned KEY_NONE = 0xffffffffu;
constexpr int IDX_BITS = 23;  // n_from < 2^23

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// popcount-accumulate: v_bcnt_u32_b32 d, x, acc  (d = popcount(x) + acc).  Written as asm so that the compiler
// keeps the 8-deep accumulate chain (16 lane-ops per pair) instead of re-associating it into bcnt + add3 trees.
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
  unsigned r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}

__device__ __forceinline__ unsigned hamming8(const unsigned (&t)[8], const uint4 &a, const uint4 &b) {
  unsigned d = __builtin_popcount(t[0] ^ a.x);
  d = bcnt_acc(t[1] ^ a.y, d);
  d = bcnt_acc(t[2] ^ a.z, d);
  d = bcnt_acc(t[3] ^ a.w, d);
  d = bcnt_acc(t[4] ^ b.x, d);
  d = bcnt_acc(t[5] ^ b.y, d);
  d = bcnt_acc(t[6] ^ b.z, d);
  d = bcnt_acc(t[7] ^ b.w, d);
  return d;
}

struct ToWindow {
  double x, y, win;
  int lmin, lmax;
  bool check_levels;
};

// features.cpp:109-119: level rule, then strict |dx| < r and |dy| < r in double
__device__ __forceinline__ bool window_ok(const ToWindow &t, const double2 p, int lv) {
  bool ok = fabs(p.x - t.x) < t.win && fabs(p.y - t.y) < t.win;
  if (t.check_levels) ok = ok && lv >= t.lmin && !(t.lmax >= 0 && lv > t.lmax);
  return ok;
}

constexpr int SNN_U = 4;  // 'from' rows fetched per scalar-load group

// grid: x = 'to' blocks of 256*T rows (T rows per lane, 256 apart), y = 'from' chunks.
// partial[chunk][n_to] = (best, second) packed keys.  The 'from' rows of the next group are fetched (scalar
// loads, wave-uniform) while the current group is being evaluated.
template <bool WINDOW, int T>
__global__ __launch_bounds__(256) void k_snn_partial(orbgpu_window_args a, int chunk, uint2 *__restrict__ partial) {
  const int j0 = blockIdx.y * chunk, j1 = min(j0 + chunk, a.n_from);
  const uint4 *__restrict__ fd = reinterpret_cast<const uint4 *>(a.from_desc);
  const double2 *__restrict__ fxy = reinterpret_cast<const double2 *>(a.from_xy);
  const int *__restrict__ flv = a.from_level;
  unsigned t[T][8];
  ToWindow tw[T];
  int row[T];
#pragma unroll
  for (int q = 0; q < T; q++) {
    row[q] = blockIdx.x * (256 * T) + q * 256 + threadIdx.x;
    const int ii = min(row[q], a.n_to - 1);
    const uint4 *td = reinterpret_cast<const uint4 *>(a.to_desc) + 2 * (size_t)ii;
    const uint4 t0 = td[0], t1 = td[1];
    t[q][0] = t0.x; t[q][1] = t0.y; t[q][2] = t0.z; t[q][3] = t0.w;
    t[q][4] = t1.x; t[q][5] = t1.y; t[q][6] = t1.z; t[q][7] = t1.w;
    if (WINDOW) {
      tw[q].x = a.to_xy[2 * ii];
      tw[q].y = a.to_xy[2 * ii + 1];
      tw[q].win = (double)a.to_window[ii];
      tw[q].lmin = a.to_min_level[ii];
      tw[q].lmax = a.to_max_level[ii];
      tw[q].check_levels = (tw[q].lmin > 0) || (tw[q].lmax >= 0);  // features.cpp:94
    }
  }
  unsigned best[T], second[T];
#pragma unroll
  for (int q = 0; q < T; q++) best[q] = second[q] = KEY_NONE;

  auto eval = [&](int j, const uint4 &fa, const uint4 &fb) {
    double2 p = make_double2(0, 0);
    int lv = 0;
    if (WINDOW) {
      p = fxy[j];
      lv = flv[j];
    }
#pragma unroll
    for (int q = 0; q < T; q++) {
      const unsigned d = hamming8(t[q], fa, fb);
      unsigned key = (d << IDX_BITS) | (unsigned)j;
      if (WINDOW) key = window_ok(tw[q], p, lv) ? key : KEY_NONE;
      second[q] = umed3(best[q], key, second[q]);  // = min(second, max(best, key)) because best <= second
      best[q] = min(best[q], key);
    }
  };

  int j = j0;
  const int jfull = j0 + ((j1 - j0) / SNN_U) * SNN_U;
  if (j < jfull) {
    uint4 ca[SNN_U], cb[SNN_U];
#pragma unroll
    for (int u = 0; u < SNN_U; u++) {
      ca[u] = fd[2 * (j + u)];
      cb[u] = fd[2 * (j + u) + 1];
    }
    for (; j < jfull; j += SNN_U) {
      uint4 na[SNN_U], nb[SNN_U];
      const int jn = min(j + SNN_U, jfull - SNN_U);  // last iteration re-fetches its own group
#pragma unroll
      for (int u = 0; u < SNN_U; u++) {
        na[u] = fd[2 * (jn + u)];
        nb[u] = fd[2 * (jn + u) + 1];
      }
#pragma unroll
      for (int u = 0; u < SNN_U; u++) eval(j + u, ca[u], cb[u]);
#pragma unroll
      for (int u = 0; u < SNN_U; u++) {
        ca[u] = na[u];
        cb[u] = nb[u];
      }
    }
  }
  for (; j < j1; j++) eval(j, fd[2 * j], fd[2 * j + 1]);

#pragma unroll
  for (int q = 0; q < T; q++)
    if (row[q] < a.n_to) partial[(size_t)blockIdx.y * a.n_to + row[q]] = make_uint2(best[q], second[q]);
}

__device__ __forceinline__ void merge_keys(unsigned &best, unsigned &second, unsigned b2, unsigned s2) {
  // union of two (best, second) pairs of disjoint candidate sets
  unsigned hi = max(best, b2);
  best = min(best, b2);
  second = min(min(second, s2), hi);
}

// ------------------------------------------------------------------------------------------------
// EXPLORATORY (the popcount kernels above stay the contract path): brute-force Hamming on the
// matrix cores.  With descriptor bits expanded to bytes +1 / -1, dot(a, b) = 256 - 2 * hamming(a, b), exact in
// v_mfma_i32_32x32x32_i8 (8 of them per 32 x 32 tile of pairs).  A workgroup of 4 waves owns 512 'to' rows (each wave
// MF_T tiles of 32, expanded once into registers as the B operand); the 'from' rows of its chunk stream through LDS 32 at
// a time, expanded by the workgroup once per step (A operand).  In the 32 x 32 result a lane holds one 'to' column and
// 16 'from' rows, so best / second stay per lane; the two lane halves of a column merge at the end.  Only candidates
// within the threshold can take part in the ratio test (second_nearest_neighbor_matcher.hpp:47-48), so a tile whose
// largest dot product is below 256 - 2 * TH is dropped after an 8-instruction maximum -- for descriptor sets with few
// near pairs the epilogue all but disappears and the kernel runs at the rate of the MFMA pipe.
constexpr int MF_T = 4;  // 'to' tiles of 32 rows per wave
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 4 descriptor bits -> 4 bytes, +1 (0x01) for a set bit, -1 (0xff) for a clear one
__device__ __forceinline__ unsigned expand_pm1(unsigned w, int q) {
  const unsigned nib = (w >> (4 * q)) & 15u;
  const unsigned t = __umul24(nib, 0x204081u) & 0x01010101u;  // bit k -> byte k
  const unsigned m = t ^ 0x01010101u;
  return t | ((m << 8) - m);  // m * 255: 0xff in the bytes of clear bits
}

// two waves per SIMD (256 registers each): the accumulators then live in VGPRs, where the threshold check can read them
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_snn_mfma(orbgpu_window_args a,
                                                                                             int chunk,
                                                                                             uint2 *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned abuf[2][8 * 2 * 32 * 4];  // [buffer][m][h][row][4 dwords]
  __shared__ __attribute__((aligned(8))) uint2 lut[256];                       // byte -> its 8 bits as +1 / -1 bytes
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int j0 = blockIdx.y * chunk, j1 = min(j0 + chunk, a.n_from);
  const int to0 = (blockIdx.x * 4 + wv) * (MF_T * 32);
  const unsigned *__restrict__ td = reinterpret_cast<const unsigned *>(a.to_desc);
  const unsigned *__restrict__ fd = reinterpret_cast<const unsigned *>(a.from_desc);
  lut[tid] = make_uint2(expand_pm1(tid, 0), expand_pm1(tid, 1));

  // B operand: lane (col, h) of MFMA m holds dimensions [32 m + 16 h, + 16) of 'to' row col of the tile
  v4i bfrag[MF_T][8];
#pragma unroll
  for (int t = 0; t < MF_T; t++) {
    const int row = min(to0 + t * 32 + col, a.n_to - 1);
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const unsigned w = td[(size_t)row * 8 + m] >> (16 * h);
      bfrag[t][m] = v4i{(int)expand_pm1(w, 0), (int)expand_pm1(w, 1), (int)expand_pm1(w, 2), (int)expand_pm1(w, 3)};
    }
  }
  unsigned best[MF_T], second[MF_T];
#pragma unroll
  for (int t = 0; t < MF_T; t++) best[t] = second[t] = KEY_NONE;
  const int dmin = 256 - 2 * (int)a.threshold;  // dot >= dmin  <=>  distance <= threshold
  __syncthreads();                              // lut

  // staging: thread (r = tid & 31, m = tid >> 5) expands dword m of 'from' row r of the step through the table
  const int sr = tid & 31, sm = tid >> 5;
  auto load_raw = [&](int jb) {
    const int j = jb + sr;
    return j < j1 ? fd[(size_t)j * 8 + sm] : 0u;
  };
  auto store_exp = [&](int buf, unsigned w) {
    const uint2 e0 = lut[w & 255u], e1 = lut[(w >> 8) & 255u], e2 = lut[(w >> 16) & 255u], e3 = lut[w >> 24];
    *reinterpret_cast<uint4 *>(&abuf[buf][((sm * 2 + 0) * 32 + sr) * 4]) = make_uint4(e0.x, e0.y, e1.x, e1.y);
    *reinterpret_cast<uint4 *>(&abuf[buf][((sm * 2 + 1) * 32 + sr) * 4]) = make_uint4(e2.x, e2.y, e3.x, e3.y);
  };
  store_exp(0, load_raw(j0));
  __syncthreads();
  int buf = 0;
  for (int jb = j0; jb < j1; jb += 32, buf ^= 1) {
    const bool more = jb + 32 < j1;
    unsigned raw = 0;
    if (more) raw = load_raw(jb + 32);
    v16i acc[MF_T];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const uint4 av = *reinterpret_cast<const uint4 *>(&abuf[buf][((m * 2 + h) * 32 + col) * 4]);
      const v4i af = v4i{(int)av.x, (int)av.y, (int)av.z, (int)av.w};
#pragma unroll
      for (int t = 0; t < MF_T; t++) {
        if (m == 0) acc[t] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bfrag[t][m], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < MF_T; t++) {
      int mx = acc[t][0];
#pragma unroll
      for (int r = 1; r < 16; r++) mx = max(mx, acc[t][r]);
      if (__ballot(mx >= dmin)) {
        // key = distance << 23 | j with distance = (256 - dot) / 2: (256 - dot) << 22 + j
        const unsigned cb = (1u << 30) + (unsigned)(jb + 4 * h);
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int roff = (r & 3) + 8 * (r >> 2);  // C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 h
          unsigned key = (unsigned)__mul24(acc[t][r], -(1 << 22)) + cb + (unsigned)roff;
          // rows past the end of the chunk (staged as zero bytes); tested on the key so that the test stays in here
          key = (key & ((1u << IDX_BITS) - 1u)) < (unsigned)j1 ? key : KEY_NONE;
          second[t] = umed3(best[t], key, second[t]);
          best[t] = min(best[t], key);
        }
      }
    }
    if (more) store_exp(buf ^ 1, raw);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < MF_T; t++) {
    merge_keys(best[t], second[t], __shfl_xor(best[t], 32), __shfl_xor(second[t], 32));
    const int row = to0 + t * 32 + col;
    if (h == 0 && row < a.n_to) partial[(size_t)blockIdx.y * a.n_to + row] = make_uint2(best[t], second[t]);
  }
}

// ratio test of second_nearest_neighbor_matcher.hpp:57 on packed keys; returns the 'from' index or -1
__device__ __forceinline__ int snn_accept(unsigned best, unsigned second, unsigned threshold, double ratio,
                                          unsigned &dist) {
  if (best == KEY_NONE) return -1;
  const unsigned bd = best >> IDX_BITS;
  if (bd > threshold) return -1;                      // every candidate was skipped at :47-48
  unsigned sd = 0xffffffffu;                          // numeric_limits<unsigned>::max(), :41-42
  if (second != KEY_NONE && (second >> IDX_BITS) <= threshold) sd = second >> IDX_BITS;
  if (!((double)bd < ratio * (double)sd)) return -1;
  dist = bd;
  return (int)(best & ((1u << IDX_BITS) - 1u));
}

// claim[j] = min over accepted 'to' of (distance << 32 | to index): smallest distance, earliest 'to' (:57-70)
__global__ __launch_bounds__(256) void k_snn_finalize(const uint2 *__restrict__ partial, int n_chunks, int n_to,
                                                      const uint8_t *__restrict__ to_active, unsigned threshold,
                                                      double ratio, int *__restrict__ from_idx,
                                                      unsigned *__restrict__ dist,
                                                      unsigned long long *__restrict__ claim) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_to) return;
  unsigned best = KEY_NONE, second = KEY_NONE;
  for (int c = 0; c < n_chunks; c++) {
    uint2 p = partial[(size_t)c * n_to + i];
    merge_keys(best, second, p.x, p.y);
  }
  unsigned d = 0;
  int j = snn_accept(best, second, threshold, ratio, d);
  if (to_active && !to_active[i]) j = -1;
  from_idx[i] = j;
  dist[i] = d;
  if (j >= 0) atomicMin(&claim[j], ((unsigned long long)d << 32) | (unsigned)i);
}

__global__ __launch_bounds__(256) void k_snn_resolve(int n_to, int *__restrict__ from_idx,
                                                     const unsigned *__restrict__ dist,
                                                     const unsigned long long *__restrict__ claim) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_to) return;
  const int j = from_idx[i];
  if (j >= 0 && claim[j] != (((unsigned long long)dist[i] << 32) | (unsigned)i)) from_idx[i] = -1;
}

// ------------------------------------------------------------------------------------------------
// ONE small search (n_to * n_from up to a few million pairs): the call SNNMatcher<ProjectionSearchIterator> /
// SNNMatcher<AreaToIterator> makes once per frame (monocular_frame.cpp:173-189, :285-306).  k_snn_partial gives such a
// problem a handful of workgroups that each walk hundreds of 'from' rows (70 us for 1000 x 1000); here a WAVE owns 64
// 'to' rows x 64 'from' rows, so 1000 x 1000 is 256 independent waves of 64 short iterations.  The wave's 64 'from' rows
// are loaded once, one row per lane (coalesced), and broadcast lane by lane with v_readlane; the window / level
// predicate comes first and rows that no lane can see (__ballot) skip the distance.  Three short launches: partial
// (which also resets the claims of its 'from' rows), finalize (merge of the chunks, threshold, ratio, claims), resolve
// (uniqueness; writes the caller's arrays, which may be mapped host memory).
template <bool WINDOW>
__global__ __launch_bounds__(64) void k_snn_small(orbgpu_window_args a, uint2 *__restrict__ partial,
                                                  unsigned long long *__restrict__ claim) {
  const int lane = threadIdx.x;
  const int j0 = blockIdx.y * 64;
  const int nj = min(64, a.n_from - j0);
  if (blockIdx.x == 0 && lane < nj) claim[j0 + lane] = ~0ull;
  const int row = blockIdx.x * 64 + lane;
  const int ii = min(row, a.n_to - 1);
  const uint4 *td = reinterpret_cast<const uint4 *>(a.to_desc) + 2 * (size_t)ii;
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  ToWindow tw{};
  if (WINDOW) {
    tw.x = a.to_xy[2 * (size_t)ii];
    tw.y = a.to_xy[2 * (size_t)ii + 1];
    tw.win = (double)a.to_window[ii];
    tw.lmin = a.to_min_level[ii];
    tw.lmax = a.to_max_level[ii];
    tw.check_levels = (tw.lmin > 0) || (tw.lmax >= 0);  // features.cpp:94
  }
  const int jj = j0 + min(lane, nj - 1);
  const uint4 *fdp = reinterpret_cast<const uint4 *>(a.from_desc) + 2 * (size_t)jj;
  const uint4 fa = fdp[0], fb = fdp[1];
  double2 fp = make_double2(0, 0);
  int flv = 0;
  if (WINDOW) {
    fp = reinterpret_cast<const double2 *>(a.from_xy)[jj];
    flv = a.from_level[jj];
  }
  auto bcast = [](unsigned v, int u) { return (unsigned)__builtin_amdgcn_readlane((int)v, u); };
  auto bcast_d = [&](double v, int u) {
    const unsigned lo = bcast((unsigned)__double2loint(v), u), hi = bcast((unsigned)__double2hiint(v), u);
    return __hiloint2double((int)hi, (int)lo);
  };
  unsigned best = KEY_NONE, second = KEY_NONE;
  for (int u = 0; u < nj; u++) {
    bool ok = true;
    if (WINDOW) {
      const double2 p = make_double2(bcast_d(fp.x, u), bcast_d(fp.y, u));
      ok = window_ok(tw, p, (int)bcast((unsigned)flv, u));
      if (!__ballot(ok)) continue;
    }
    const uint4 ca = make_uint4(bcast(fa.x, u), bcast(fa.y, u), bcast(fa.z, u), bcast(fa.w, u));
    const uint4 cb = make_uint4(bcast(fb.x, u), bcast(fb.y, u), bcast(fb.z, u), bcast(fb.w, u));
    unsigned key = (hamming8(t, ca, cb) << IDX_BITS) | (unsigned)(j0 + u);
    key = ok ? key : KEY_NONE;
    second = umed3(best, key, second);
    best = min(best, key);
  }
  if (row < a.n_to) partial[(size_t)blockIdx.y * a.n_to + row] = make_uint2(best, second);
}

__global__ __launch_bounds__(256) void k_snn_small_finalize(const uint2 *__restrict__ partial, int n_chunks, int n_to,
                                                            const uint8_t *__restrict__ to_active, unsigned threshold,
                                                            double ratio, uint2 *__restrict__ accepted,
                                                            unsigned long long *__restrict__ claim) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_to) return;
  unsigned best = KEY_NONE, second = KEY_NONE;
  for (int c = 0; c < n_chunks; c++) {
    const uint2 p = partial[(size_t)c * n_to + i];
    merge_keys(best, second, p.x, p.y);
  }
  unsigned d = 0;
  int j = snn_accept(best, second, threshold, ratio, d);
  if (to_active && !to_active[i]) j = -1;
  accepted[i] = make_uint2((unsigned)j, d);
  if (j >= 0) atomicMin(&claim[j], ((unsigned long long)d << 32) | (unsigned)i);
}

__global__ __launch_bounds__(256) void k_snn_small_resolve(int n_to, const uint2 *__restrict__ accepted,
                                                           const unsigned long long *__restrict__ claim,
                                                           int *__restrict__ from_idx, unsigned *__restrict__ dist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_to) return;
  const uint2 acc = accepted[i];
  int j = (int)acc.x;
  if (j >= 0 && claim[j] != (((unsigned long long)acc.y << 32) | (unsigned)i)) j = -1;
  from_idx[i] = j;
  dist[i] = acc.y;
}

// ------------------------------------------------------------------------------------------------
// Many independent window searches in one launch (search by projection over a batch of frames: one
// SNNMatcher<ProjectionSearchIterator>::MatchWithIterators per segment, monocular_frame.cpp:285-306).  Segment s pairs
// the 'to' rows [to_begin, to_begin + to_count) with the 'from' rows [from_begin, from_begin + from_count) of the
// concatenated arrays.  grid: x = 256-row 'to' blocks of a segment, y = segment.  A segment's 'from' rows are few
// (one frame's keypoints), so a lane scans all of them -- wave-uniform scalar loads, window / level predicate before
// the distance -- and applies threshold and ratio itself; claims are keyed by the global 'from' row.
__global__ __launch_bounds__(256) void k_snn_segments(orbgpu_window_args a, const orbgpu_segment *__restrict__ segs,
                                                      int *__restrict__ from_idx, unsigned *__restrict__ dist,
                                                      unsigned long long *__restrict__ claim) {
  const orbgpu_segment sg = segs[blockIdx.y];
  if ((int)(blockIdx.x * 256) >= sg.to_count) return;
  const int li = blockIdx.x * 256 + threadIdx.x;
  const bool live = li < sg.to_count;
  const int i = sg.to_begin + min(li, sg.to_count - 1);
  const uint4 *td = reinterpret_cast<const uint4 *>(a.to_desc) + 2 * (size_t)i;
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  ToWindow tw;
  tw.x = a.to_xy[2 * (size_t)i];
  tw.y = a.to_xy[2 * (size_t)i + 1];
  tw.win = (double)a.to_window[i];
  tw.lmin = a.to_min_level[i];
  tw.lmax = a.to_max_level[i];
  tw.check_levels = (tw.lmin > 0) || (tw.lmax >= 0);  // features.cpp:94
  const uint4 *__restrict__ fd = reinterpret_cast<const uint4 *>(a.from_desc) + 2 * (size_t)sg.from_begin;
  const double2 *__restrict__ fxy = reinterpret_cast<const double2 *>(a.from_xy) + sg.from_begin;
  const int *__restrict__ flv = a.from_level + sg.from_begin;
  unsigned best = KEY_NONE, second = KEY_NONE;
#pragma unroll 2
  for (int j = 0; j < sg.from_count; j++) {
    const bool ok = live && window_ok(tw, fxy[j], flv[j]);
    if (!__ballot(ok)) continue;  // no lane of the wave has this candidate in its window
    unsigned key = (hamming8(t, fd[2 * j], fd[2 * j + 1]) << IDX_BITS) | (unsigned)j;
    key = ok ? key : KEY_NONE;
    second = umed3(best, key, second);
    best = min(best, key);
  }
  if (!live) return;
  unsigned d = 0;
  int j = snn_accept(best, second, a.threshold, a.ratio, d);
  if (a.to_active && !a.to_active[i]) {
    j = -1;
    d = 0;
  }
  from_idx[i] = j;
  dist[i] = d;
  if (j >= 0) atomicMin(&claim[sg.from_begin + j], ((unsigned long long)d << 32) | (unsigned)li);
}

__global__ __launch_bounds__(256) void k_snn_segments_resolve(const orbgpu_segment *__restrict__ segs,
                                                              int *__restrict__ from_idx,
                                                              const unsigned *__restrict__ dist,
                                                              const unsigned long long *__restrict__ claim,
                                                              int max_to, orbgpu_segment seg0) {
  const orbgpu_segment sg = segs ? segs[blockIdx.y] : seg0;
  const int li = blockIdx.x * 256 + threadIdx.x;
  if (li >= min(sg.to_count, max_to)) return;
  const int i = sg.to_begin + li;
  const int j = from_idx[i];
  if (j >= 0 && claim[sg.from_begin + j] != (((unsigned long long)dist[i] << 32) | (unsigned)li)) from_idx[i] = -1;
}

// ------------------------------------------------------------------------------------------------
// The 64 x 48 search grid of Features (features.cpp:127-161: AssignFeaturesToGrid over the body of PosInGrid behind
// its early return) as a device-side index, and the windowed matcher over it.  The grid only narrows the scan: the
// candidate set of a 'to' row is decided by the same window / level predicate as everywhere else and ties are broken
// by the original 'from' index inside the packed key, so the results are those of k_snn_segments bit for bit.
constexpr int GRID_COLS = 64, GRID_ROWS = 48, GRID_CELLS = GRID_COLS * GRID_ROWS, GRID_BINS = 4096;
constexpr int GRID_ITEM_BITS = 20;  // a range holds fewer than 2^20 rows: (bin << 20 | row) fits an int

struct GridDims {
  double w, h, inv_w, inv_h;  // image_width, image_height, grid_element_width_inv_, grid_element_height_inv_
};

// features.cpp:150-158
__device__ __forceinline__ int2 pos_in_grid(double kx, double ky, const GridDims &g) {
  double x = kx > 0.0 ? kx : 0.0;
  x = x < g.w ? x : g.w;
  double y = ky > 0.0 ? ky : 0.0;
  y = y < g.h ? y : g.h;
  return make_int2(min(GRID_COLS - 1, (int)(x * g.inv_w)), min(GRID_ROWS - 1, (int)(y * g.inv_h)));
}

// Position along a 64 x 64 Hilbert curve laid over the image (x, y in 0..63): rows that follow each other on the curve
// are neighbours in the image, so 64 consecutive rows form one compact patch (a Morton order jumps between patches).
__device__ __forceinline__ int hilbert64(int x, int y) {
  int d = 0;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const int rx = (x & s) ? 1 : 0, ry = (y & s) ? 1 : 0;
    d += s * s * ((3 * rx) ^ ry);
    if (ry == 0) {
      if (rx) {
        x = 63 - x;
        y = 63 - y;
      }
      const int tq = x;
      x = y;
      y = tq;
    }
  }
  return d;
}

// One workgroup per range: rows [begin, begin + count) of xy, binned by grid cell (cell = ix * 48 + iy, the flat layout
// of grid[COLS][ROWS]; CURVE: by position along a Hilbert curve over the image), ascending row index inside a bin
// (push_back order).
// ranges: n x range_stride ints, (begin, count) at range_off.  Outputs for range r at out = r * out_stride:
// items[out + k] = row - begin in bin order, cell_start[r * (GRID_CELLS + 1) + c] (not CURVE), and with GATHER the
// rows' descriptor / position / level copied into bin order (the matcher then scans contiguous arrays).
constexpr int GB_THREADS = 1024, GB_PER = GRID_BINS / GB_THREADS;
constexpr int GB_RANK_LIMIT = 4096;  // largest bin that is still ordered by rank counting (quadratic in the bin)

struct GridBuildArgs {
  const double2 *xy;
  const int *ranges;
  int range_stride, range_off, out_stride;
  int *tmp, *items, *cell_start;
  const uint4 *desc;  // GATHER only
  const int *level;
  uint4 *s_desc;
  double2 *s_xy;
  int *s_lv;
  int *error_flag;  // ORBGPU_ERR_SEGMENT_SIZE when a range is larger than out_stride (the surplus is ignored)
  int one[4];       // ranges == nullptr: the single range / segment, by value
};

// count rows, of which keep(i) take part; getxy(i) = position; emit(k, i): row i is the k-th of the binned order.
// Returns the number of rows kept.  hist / cursor: GRID_BINS ints each, wsum: GB_THREADS / 64 + 2.
template <bool CURVE, class GetXY, class Keep, class Emit>
__device__ __forceinline__ int grid_build_core(int count, const GridDims &g, int *__restrict__ tmp, int *cell_start_r,
                                               int *hist, int *cursor, int *wsum, GetXY getxy, Keep keep, Emit emit) {
  const int tid = threadIdx.x;
  for (int b = tid; b < GRID_BINS; b += GB_THREADS) hist[b] = 0;
  if (tid == 0) wsum[GB_THREADS / 64 + 1] = 0;  // size of the largest bin when it is above GB_RANK_LIMIT
  __syncthreads();
  auto bin_of = [&](int i) {
    const double2 p = getxy(i);
    const int2 c = pos_in_grid(p.x, p.y, g);
    if (!CURVE) return c.x * GRID_ROWS + c.y;
    // ordering only: a 64 x 64 lattice over the image, independent of the search grid
    double y = p.y > 0.0 ? p.y : 0.0;
    y = y < g.h ? y : g.h;
    return hilbert64(c.x, min(63, (int)(y * (64.0 / g.h))));
  };
  for (int i = tid; i < count; i += GB_THREADS)
    if (keep(i)) atomicAdd(&hist[bin_of(i)], 1);
  __syncthreads();
  // exclusive scan over the bins: GB_PER consecutive bins per thread, then the thread sums
  int local[GB_PER], sum = 0;
#pragma unroll
  for (int k = 0; k < GB_PER; k++) {
    local[k] = sum;
    sum += hist[tid * GB_PER + k];
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += v;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = incl - sum;
  for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
#pragma unroll
  for (int k = 0; k < GB_PER; k++) cursor[tid * GB_PER + k] = base + local[k];
  if (tid == GB_THREADS - 1) wsum[GB_THREADS / 64] = base + sum;
  // the largest bin decides how rows are ordered inside the bins (below)
  int hmax = 0;
#pragma unroll
  for (int k = 0; k < GB_PER; k++) hmax = max(hmax, hist[tid * GB_PER + k]);
  if (hmax > GB_RANK_LIMIT) atomicMax(&wsum[GB_THREADS / 64 + 1], hmax);
  __syncthreads();
  const int total = wsum[GB_THREADS / 64];
  const bool crowded = wsum[GB_THREADS / 64 + 1] > GB_RANK_LIMIT;
  if (!CURVE && cell_start_r) {
    for (int b = tid; b < GRID_CELLS; b += GB_THREADS) cell_start_r[b] = cursor[b];
    if (tid == 0) cell_start_r[GRID_CELLS] = total;
  }
  if (crowded) {
    // A bin holds thousands of rows (keypoints of a larger image than the grid was laid over all clamp into the last
    // cells; a clustered point set): ranking by counting below is quadratic in the bin.  Stable placement instead, by ONE
    // wave walking the rows in index order, 64 at a time: the lanes of one bin are found with a ballot, their rank is
    // their position among them, the first of them advances the bin's cursor.  Linear in the rows; slow (one wave), but
    // only ever taken by inputs the grid is a poor index for.
    __syncthreads();
    if (tid < 64) {
      const unsigned long long below = (1ull << tid) - 1ull;
      for (int base0 = 0; base0 < count; base0 += 64) {
        const int i = base0 + tid;
        const bool valid = i < count && keep(i);
        const int b = valid ? bin_of(i) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const int bl = __builtin_amdgcn_readlane(b, leader);
          const unsigned long long m = __ballot(b == bl);
          const int start = cursor[bl];
          if (b == bl) emit(start + __popcll(m & below), i);
          if (tid == leader) cursor[bl] = start + __popcll(m);
          todo &= ~m;
        }
      }
    }
    __syncthreads();
    return total;
  }
  for (int i = tid; i < count; i += GB_THREADS) {
    if (!keep(i)) continue;
    const int b = bin_of(i);
    tmp[atomicAdd(&cursor[b], 1)] = (int)(((unsigned)b << GRID_ITEM_BITS) | (unsigned)i);  // arrival order inside the bin
  }
  __syncthreads();  // tmp of this workgroup is complete; cursor[b] = end of bin b, cursor[b] - hist[b] = its start
  for (int k = tid; k < total; k += GB_THREADS) {
    const unsigned e = (unsigned)tmp[k];
    const int b = (int)(e >> GRID_ITEM_BITS), i = (int)(e & ((1u << GRID_ITEM_BITS) - 1u));
    const int b1 = cursor[b], b0 = b1 - hist[b];
    int rank = 0;
    for (int q = b0; q < b1; q++) rank += (tmp[q] & ((1 << GRID_ITEM_BITS) - 1)) < i;
    emit(b0 + rank, i);
  }
  return total;
}

template <bool CURVE, bool GATHER>
__device__ __forceinline__ void grid_build_body(const GridBuildArgs &a, const GridDims &g, int r, int *hist, int *cursor,
                                                int *wsum) {
  const int begin = a.ranges ? a.ranges[r * a.range_stride + a.range_off] : a.one[a.range_off];
  int count = a.ranges ? a.ranges[r * a.range_stride + a.range_off + 1] : a.one[a.range_off + 1];
  if (count > a.out_stride) {  // the caller's max count was too small: stay inside this range's rows and say so
    if (threadIdx.x == 0 && a.error_flag) *(volatile int *)a.error_flag = ORBGPU_ERR_SEGMENT_SIZE;
    count = a.out_stride;
  }
  const size_t out = (size_t)r * a.out_stride;
  const double2 *__restrict__ xy = a.xy + begin;
  grid_build_core<CURVE>(
      count, g, a.tmp + out, (!CURVE && a.cell_start) ? a.cell_start + (size_t)r * (GRID_CELLS + 1) : nullptr, hist, cursor,
      wsum, [&](int i) { return xy[i]; }, [](int) { return true; },
      [&](int k, int i) {
        const size_t dst = out + k;
        a.items[dst] = i;
        if (GATHER) {
          a.s_desc[2 * dst] = a.desc[2 * (size_t)(begin + i)];
          a.s_desc[2 * dst + 1] = a.desc[2 * (size_t)(begin + i) + 1];
          a.s_xy[dst] = xy[i];
          a.s_lv[dst] = a.level[begin + i];
        }
      });
}

// the grid by itself (orbgpu_grid_build_device): grid = ranges
__global__ __launch_bounds__(GB_THREADS) void k_grid_build(GridBuildArgs a, GridDims g) {
  __shared__ int hist[GRID_BINS], cursor[GRID_BINS], wsum[GB_THREADS / 64 + 2];
  grid_build_body<false, false>(a, g, blockIdx.x, hist, cursor, wsum);
}

// both sides of a batch of window searches in one launch: grid = (segments, 2); y = 0 bins the 'from' rows by cell and
// gathers their copies, y = 1 orders the 'to' rows along the curve
__global__ __launch_bounds__(GB_THREADS) void k_grid_build_pair(GridBuildArgs from, GridBuildArgs to, GridDims g) {
  __shared__ int hist[GRID_BINS], cursor[GRID_BINS], wsum[GB_THREADS / 64 + 2];
  if (blockIdx.y == 0)
    grid_build_body<false, true>(from, g, blockIdx.x, hist, cursor, wsum);
  else
    grid_build_body<true, false>(to, g, blockIdx.x, hist, cursor, wsum);
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
  return __builtin_amdgcn_readfirstlane(v);
}

// grid: x = 256-row blocks of a segment's 'to' rows in Hilbert order over the image, y = segment.  A wave's 64 'to' rows
// lie in a compact patch, so the union of their windows covers few grid cells: per grid column ix of that union the
// cells iy_min..iy_max are ONE contiguous run of the binned 'from' copies.  The runs of all columns, laid end to end, are
// fetched 64 candidates at a time (one per lane, the next 64 in flight while these are evaluated) and every candidate is
// broadcast with v_readlane: a scalar load per candidate would pay the full memory latency once per iteration (the
// copies of a batch do not fit L2).
// The scan of one wave over the binned 'from' rows of its frame (every lane of the 256-thread block must call it: it
// holds two barriers).  flv == nullptr: the rows were filtered by level when they were binned.
__device__ __forceinline__ void grid_scan(const ToWindow &tw, bool active, const unsigned (&t)[8], const GridDims &g,
                                          const int *__restrict__ cs, const uint4 *__restrict__ fd,
                                          const double2 *__restrict__ fxy, const int *__restrict__ flv,
                                          const int *__restrict__ fid, unsigned &best, unsigned &second) {
  const int lane = threadIdx.x & 63;
  // features.cpp:91-92
  const int2 lo = pos_in_grid(tw.x - tw.win, tw.y - tw.win, g), hi = pos_in_grid(tw.x + tw.win, tw.y + tw.win, g);
  const int x0 = wave_min_i(active ? lo.x : GRID_COLS), x1 = wave_max_i(active ? hi.x : -1);
  auto rl = [](unsigned v, int c) { return (unsigned)__builtin_amdgcn_readlane((int)v, c); };
  auto rld = [&](double v, int c) {
    const unsigned long long u = __double_as_longlong(v);
    return __longlong_as_double((long long)(((unsigned long long)rl((unsigned)(u >> 32), c) << 32) | rl((unsigned)u, c)));
  };
  // per grid column of the union, the cell rows some lane of this wave needs (a bounding box would also scan the
  // corners nobody looks at)
  __shared__ int col_y0[4][GRID_COLS], col_y1[4][GRID_COLS];
  const int wv = threadIdx.x >> 6;
  col_y0[wv][lane] = GRID_ROWS;
  col_y1[wv][lane] = -1;
  __syncthreads();
  if (active)
    for (int ix = lo.x; ix <= hi.x; ix++) {
      atomicMin(&col_y0[wv][ix], lo.y);
      atomicMax(&col_y1[wv][ix], hi.y);
    }
  __syncthreads();
  // lane l = grid column x0 + l of the union: its run [k0, k1) of binned 'from' rows, and the runs laid end to end
  const int ncols = x1 - x0 + 1;  // <= 64; <= 0 when no lane is active
  int run0 = 0, run_len = 0;
  if (lane < ncols) {
    const int cy0 = col_y0[wv][x0 + lane], cy1 = col_y1[wv][x0 + lane];
    if (cy0 <= cy1) {
      run0 = cs[(x0 + lane) * GRID_ROWS + cy0];
      run_len = cs[(x0 + lane) * GRID_ROWS + cy1 + 1] - run0;
    }
  }
  int run_end = run_len;  // inclusive scan
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const int v = __shfl_up(run_end, dlt);
    if (lane >= dlt) run_end += v;
  }
  const int total = ncols > 0 ? __builtin_amdgcn_readlane(run_end, 63) : 0;
  struct Cand {
    double2 xy;
    int lv;
    unsigned id;
    uint4 a, b;
  };
  // candidate f of the concatenated runs -> its row: the column is the number of runs that end at or before f
  auto fetch = [&](int f0) {
    const int f = min(f0 + lane, total - 1);
    int col = 0;
    for (int l = 0; l < ncols; l++) col += rl((unsigned)run_end, l) <= (unsigned)f ? 1 : 0;
    const int k = __shfl(run0, col) + (f - (__shfl(run_end, col) - __shfl(run_len, col)));
    Cand c;
    c.xy = fxy[k];
    c.lv = flv ? flv[k] : 0;
    c.id = (unsigned)fid[k];
    c.a = fd[2 * k];
    c.b = fd[2 * k + 1];
    return c;
  };
  if (total > 0) {
    Cand cur = fetch(0);
    for (int f0 = 0; f0 < total; f0 += 64) {
      Cand nxt = cur;
      if (f0 + 64 < total) nxt = fetch(f0 + 64);  // in flight while the current 64 are evaluated
      const int cnt = min(64, total - f0);
      for (int c = 0; c < cnt; c++) {
        const bool ok = active && window_ok(tw, make_double2(rld(cur.xy.x, c), rld(cur.xy.y, c)), (int)rl((unsigned)cur.lv, c));
        if (!__ballot(ok)) continue;
        const uint4 fa = make_uint4(rl(cur.a.x, c), rl(cur.a.y, c), rl(cur.a.z, c), rl(cur.a.w, c));
        const uint4 fb4 = make_uint4(rl(cur.b.x, c), rl(cur.b.y, c), rl(cur.b.z, c), rl(cur.b.w, c));
        unsigned key = (hamming8(t, fa, fb4) << IDX_BITS) | rl(cur.id, c);
        key = ok ? key : KEY_NONE;
        second = umed3(best, key, second);
        best = min(best, key);
      }
      cur = nxt;
    }
  }
}

__global__ __launch_bounds__(256) void k_snn_grid(orbgpu_window_args a, const orbgpu_segment *__restrict__ segs,
                                                  orbgpu_segment seg0, GridDims g, int to_stride, int from_stride,
                                                  const int *__restrict__ to_items, const int *__restrict__ cell_start,
                                                  const int *__restrict__ s_idx, const uint4 *__restrict__ s_desc,
                                                  const double2 *__restrict__ s_xy, const int *__restrict__ s_lv,
                                                  int *__restrict__ from_idx, unsigned *__restrict__ dist,
                                                  unsigned long long *__restrict__ claim) {
  orbgpu_segment sg = segs ? segs[blockIdx.y] : seg0;  // segs == nullptr: the one segment, by value
  sg.to_count = min(sg.to_count, to_stride);  // larger: flagged by the build kernel
  if ((int)(blockIdx.x * 256) >= sg.to_count) return;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const bool live = p < sg.to_count;
  const int li = to_items[(size_t)blockIdx.y * to_stride + min(p, sg.to_count - 1)];
  const int i = sg.to_begin + li;
  const uint4 *td = reinterpret_cast<const uint4 *>(a.to_desc) + 2 * (size_t)i;
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  ToWindow tw;
  tw.x = a.to_xy[2 * (size_t)i];
  tw.y = a.to_xy[2 * (size_t)i + 1];
  tw.win = (double)a.to_window[i];
  tw.lmin = a.to_min_level[i];
  tw.lmax = a.to_max_level[i];
  tw.check_levels = (tw.lmin > 0) || (tw.lmax >= 0);  // features.cpp:94
  const bool active = live && !(a.to_active && !a.to_active[i]);
  const int *__restrict__ cs = cell_start + (size_t)blockIdx.y * (GRID_CELLS + 1);
  const size_t fb = (size_t)blockIdx.y * from_stride;
  unsigned best = KEY_NONE, second = KEY_NONE;
  grid_scan(tw, active, t, g, cs, s_desc + 2 * fb, s_xy + fb, s_lv + fb, s_idx + fb, best, second);
  if (!live) return;
  unsigned d = 0;
  int j = snn_accept(best, second, a.threshold, a.ratio, d);
  if (!active) {
    j = -1;
    d = 0;
  }
  from_idx[i] = j;
  dist[i] = d;
  if (j >= 0) atomicMin(&claim[sg.from_begin + j], ((unsigned long long)d << 32) | (unsigned)li);
}

// ------------------------------------------------------------------------------------------------
// Rotation histogram (orientation_validator.cpp:16-97) for one match list, executed by one workgroup.
// angle getters return the float the extractor produced (the reference stores it in a double).
template <class AngTo, class AngFrom>
__device__ void rotation_filter_block(int n_to, int *match, AngTo ang_to, AngFrom ang_from, int *n_discarded,
                                      int *n_left) {
  __shared__ int hist[30];
  __shared__ int keep_idx[3];
  __shared__ int dropped, kept;
  if (threadIdx.x < 30) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) dropped = kept = 0;
  __syncthreads();
  const double factor = (double)(1.0f / 30);  // const precision_t factor = 1.0f / HISTO_LENGTH, :81
  auto bin_of = [&](int i, int j) {
    float rot = (float)((double)ang_to(i) - (double)ang_from(j));  // :87
    if (rot < 0.0) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)round((double)rot * factor);  // std::round: half away from zero, :90
    if (bin == 30) bin = 0;
    return min(max(bin, 0), 29);
  };
  for (int i = threadIdx.x; i < n_to; i += blockDim.x) {
    int j = match[i];
    if (j >= 0) atomicAdd(&hist[bin_of(i, j)], 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // ComputeThreeMaxima :42-76, first-seen wins ties
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int b = 0; b < 30; b++) {
      const int s = hist[b];
      if (s > max1) {
        max3 = max2; max2 = max1; max1 = s;
        ind3 = ind2; ind2 = ind1; ind1 = b;
      } else if (s > max2) {
        max3 = max2; max2 = s;
        ind3 = ind2; ind2 = b;
      } else if (s > max3) {
        max3 = s; ind3 = b;
      }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
      ind2 = -1; ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
      ind3 = -1;
    }
    keep_idx[0] = ind1; keep_idx[1] = ind2; keep_idx[2] = ind3;
  }
  __syncthreads();
  int nd = 0, nk = 0;
  for (int i = threadIdx.x; i < n_to; i += blockDim.x) {
    int j = match[i];
    if (j < 0) continue;
    int b = bin_of(i, j);
    if (b == keep_idx[0] || b == keep_idx[1] || b == keep_idx[2]) {
      nk++;
    } else {
      match[i] = -1;
      nd++;
    }
  }
  if (nd) atomicAdd(&dropped, nd);
  if (nk) atomicAdd(&kept, nk);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (n_discarded) *n_discarded = dropped;
    if (n_left) *n_left = kept;
  }
}

__global__ __launch_bounds__(256) void k_rotation_filter(const float *__restrict__ ang_to, int n_to,
                                                         const float *__restrict__ ang_from, int *match,
                                                         int *n_discarded) {
  rotation_filter_block(
      n_to, match, [&](int i) { return ang_to[i]; }, [&](int j) { return ang_from[j]; }, n_discarded, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Batched frame-to-frame linking (MonocularFrame::ComputeMatchesForLinking, monocular_frame.cpp:173-189) on
// the device-resident output of the extractor.  grid: x = 'to' blocks, y = 'from' chunks, z = pair.
__global__ __launch_bounds__(256) void k_link_partial(const orbgpu_keypoint *__restrict__ kp,
                                                      const uint8_t *__restrict__ desc, const int *__restrict__ n_kp,
                                                      int capacity, const int *__restrict__ to_frame,
                                                      const int *__restrict__ from_frame, double window, int chunk,
                                                      int n_chunks, uint2 *__restrict__ partial) {
  const int p = blockIdx.z;
  const int ft = to_frame[p], ff = from_frame[p];
  const int n_to = usable_count(n_kp[ft], capacity), n_from = usable_count(n_kp[ff], capacity);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if ((int)(blockIdx.x * 256) >= n_to) return;
  const int j0 = blockIdx.y * chunk, j1 = min(j0 + chunk, n_from);
  const int ii = min(i, n_to - 1);
  const orbgpu_keypoint *fk = kp + (size_t)ff * capacity;
  const uint4 *fd = reinterpret_cast<const uint4 *>(desc + (size_t)ff * capacity * 32);
  // AreaToIterator visits 'to' id 0 and the level-0 ids, AreaToPointee level-0 'from' only: a 'to' block or a 'from'
  // chunk (chunk <= 256 = one id per thread) without any of them leaves empty partials.  The extractor packs its
  // output by level, so all but a few blocks of a many-level frame pair end here
  const bool to_live = i < n_to && (i == 0 || kp[(size_t)ft * capacity + ii].level == 0);
  const bool from_live = j0 + (int)threadIdx.x < j1 && fk[j0 + threadIdx.x].level == 0;
  const int any_to = __syncthreads_or(to_live), any_from = __syncthreads_or(from_live);
  if (!any_to || !any_from) {
    if (i < n_to) partial[((size_t)p * n_chunks + blockIdx.y) * capacity + i] = make_uint2(KEY_NONE, KEY_NONE);
    return;
  }
  // only the few live blocks read the keypoint and its descriptor; the chunk's 'from' entries go through LDS once
  // (one per thread, coalesced) instead of being fetched by dependent scalar loads inside the scan
  __shared__ double s_fx[256], s_fy[256];
  __shared__ int s_lv[256];
  __shared__ uint4 s_fd[512];
  {
    const int jt = j0 + (int)threadIdx.x;
    if (jt < j1) {
      s_fx[threadIdx.x] = fk[jt].x;
      s_fy[threadIdx.x] = fk[jt].y;
      s_lv[threadIdx.x] = fk[jt].level;
      s_fd[2 * threadIdx.x] = fd[2 * jt];
      s_fd[2 * threadIdx.x + 1] = fd[2 * jt + 1];
    }
  }
  const orbgpu_keypoint k = kp[(size_t)ft * capacity + ii];
  const uint4 *td = reinterpret_cast<const uint4 *>(desc + ((size_t)ft * capacity + ii) * 32);
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  __syncthreads();
  unsigned best = KEY_NONE, second = KEY_NONE;
  // a wave without a visited 'to' has nothing to scan (block 0 of a many-level frame: only id 0 is visited)
  const int n_scan = __ballot(to_live) ? j1 - j0 : 0;
#pragma unroll 2
  for (int jj = 0; jj < n_scan; jj++) {
    // ListFeaturesInArea(point, window, 0, 0): level-0 'from' only (area_to_pointee.h:44-48); jj is wave-uniform
    if (s_lv[jj] != 0) continue;
    // the window first: when no lane of the wave has this candidate in its window, the distance is not needed
    const bool ok = to_live && fabs(s_fx[jj] - k.x) < window && fabs(s_fy[jj] - k.y) < window;
    if (!__ballot(ok)) continue;
    const uint4 a = s_fd[2 * jj], b = s_fd[2 * jj + 1];
    unsigned d = __builtin_popcount(t[0] ^ a.x);
    d += __builtin_popcount(t[1] ^ a.y);
    d += __builtin_popcount(t[2] ^ a.z);
    d += __builtin_popcount(t[3] ^ a.w);
    d += __builtin_popcount(t[4] ^ b.x);
    d += __builtin_popcount(t[5] ^ b.y);
    d += __builtin_popcount(t[6] ^ b.z);
    d += __builtin_popcount(t[7] ^ b.w);
    unsigned key = (d << IDX_BITS) | (unsigned)(j0 + jj);
    key = ok ? key : KEY_NONE;
    second = umed3(best, key, second);
    best = min(best, key);
  }
  if (i < n_to) partial[((size_t)p * n_chunks + blockIdx.y) * capacity + i] = make_uint2(best, second);
}

__global__ __launch_bounds__(256) void k_link_finalize(const orbgpu_keypoint *__restrict__ kp, const int *__restrict__ n_kp,
                                                       int capacity, const int *__restrict__ to_frame,
                                                       const int *__restrict__ from_frame, int n_chunks,
                                                       const uint2 *__restrict__ partial, unsigned threshold,
                                                       double ratio, int *__restrict__ match,
                                                       unsigned *__restrict__ dist,
                                                       unsigned long long *__restrict__ claim) {
  const int p = blockIdx.y;
  const int ft = to_frame[p];
  const int n_to = usable_count(n_kp[ft], capacity);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= capacity) return;
  int j = -1;
  unsigned d = 0;
  if (i < n_to) {
    unsigned best = KEY_NONE, second = KEY_NONE;
    for (int c = 0; c < n_chunks; c++) {
      uint2 q = partial[((size_t)p * n_chunks + c) * capacity + i];
      merge_keys(best, second, q.x, q.y);
    }
    j = snn_accept(best, second, threshold, ratio, d);
    // AreaToIterator visits id 0 whatever its level, later ids only when level == 0 (area_to_iterator.h:24-43)
    if (!(i == 0 || kp[(size_t)ft * capacity + i].level == 0)) {
      j = -1;
      d = 0;
    }
  }
  match[(size_t)p * capacity + i] = j;
  dist[(size_t)p * capacity + i] = d;
  if (j >= 0) atomicMin(&claim[(size_t)p * capacity + j], ((unsigned long long)d << 32) | (unsigned)i);
}

// one workgroup per pair: uniqueness resolve, then (optionally) the rotation histogram
__global__ __launch_bounds__(256) void k_link_resolve(const orbgpu_keypoint *__restrict__ kp, const int *__restrict__ n_kp,
                                                      int capacity, const int *__restrict__ to_frame,
                                                      const int *__restrict__ from_frame,
                                                      const unsigned long long *__restrict__ claim,
                                                      int check_orientation, int *__restrict__ match,
                                                      const unsigned *__restrict__ dist, int *__restrict__ n_matches) {
  const int p = blockIdx.x;
  const int ft = to_frame[p], ff = from_frame[p];
  const int n_to = usable_count(n_kp[ft], capacity);
  int *m = match + (size_t)p * capacity;
  const unsigned *dd = dist + (size_t)p * capacity;
  const unsigned long long *cl = claim + (size_t)p * capacity;
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = threadIdx.x; i < n_to; i += blockDim.x) {
    int j = m[i];
    if (j >= 0) {
      if (cl[j] != (((unsigned long long)dd[i] << 32) | (unsigned)i)) m[i] = -1;
      else cnt++;
    }
  }
  if (cnt) atomicAdd(&total, cnt);
  __syncthreads();
  if (!check_orientation) {
    if (threadIdx.x == 0) n_matches[p] = total;
    return;
  }
  const orbgpu_keypoint *kt = kp + (size_t)ft * capacity, *kf = kp + (size_t)ff * capacity;
  rotation_filter_block(
      n_to, m, [&](int i) { return kt[i].angle; }, [&](int j) { return kf[j].angle; }, nullptr, &n_matches[p]);
}

// ------------------------------------------------------------------------------------------------
// Linking over the search grid, for keypoint capacities beyond the fused kernel's LDS: replaces k_link_partial (the
// finalize / resolve kernels above stay).  Build, grid = (pair, side): side 0 bins the level-0 'from' rows of the pair
// (the only ones AreaToPointee lists, area_to_pointee.h:44-48) by cell and gathers their copies; side 1 orders the
// visited 'to' rows (id 0 and the level-0 ids, area_to_iterator.h:24-43) along the Hilbert curve.
__global__ __launch_bounds__(GB_THREADS) void k_link_grid_build(const orbgpu_keypoint *__restrict__ kp,
                                                                const uint8_t *__restrict__ desc,
                                                                const int *__restrict__ n_kp, int capacity,
                                                                const int *__restrict__ to_frame,
                                                                const int *__restrict__ from_frame, GridDims g,
                                                                int *__restrict__ tmp, int *__restrict__ to_items,
                                                                int *__restrict__ to_counts, int *__restrict__ cell_start,
                                                                int *__restrict__ s_idx, uint4 *__restrict__ s_desc,
                                                                double2 *__restrict__ s_xy) {
  __shared__ int hist[GRID_BINS], cursor[GRID_BINS], wsum[GB_THREADS / 64 + 2];
  const int p = blockIdx.x;
  const size_t out = (size_t)p * capacity;
  if (blockIdx.y == 0) {
    const int ff = from_frame[p], n = usable_count(n_kp[ff], capacity);
    const orbgpu_keypoint *__restrict__ fk = kp + (size_t)ff * capacity;
    const uint4 *__restrict__ fd = reinterpret_cast<const uint4 *>(desc + (size_t)ff * capacity * 32);
    grid_build_core<false>(
        n, g, tmp + out, cell_start + (size_t)p * (GRID_CELLS + 1), hist, cursor, wsum,
        [&](int i) { return make_double2(fk[i].x, fk[i].y); }, [&](int i) { return fk[i].level == 0; },
        [&](int k, int i) {
          s_idx[out + k] = i;
          s_desc[2 * (out + k)] = fd[2 * i];
          s_desc[2 * (out + k) + 1] = fd[2 * i + 1];
          s_xy[out + k] = make_double2(fk[i].x, fk[i].y);
        });
  } else {
    const int ft = to_frame[p], n = usable_count(n_kp[ft], capacity);
    const orbgpu_keypoint *__restrict__ tk = kp + (size_t)ft * capacity;
    const int total = grid_build_core<true>(
        n, g, tmp + (size_t)(gridDim.x + p) * capacity, nullptr, hist, cursor, wsum,
        [&](int i) { return make_double2(tk[i].x, tk[i].y); }, [&](int i) { return i == 0 || tk[i].level == 0; },
        [&](int k, int i) { to_items[out + k] = i; });
    if (threadIdx.x == 0) to_counts[p] = total;
  }
}

// grid: x = 256-row blocks of the pair's visited 'to' rows, y = pair.  partial[pair][capacity] was filled with KEY_NONE.
__global__ __launch_bounds__(256) void k_link_grid(const orbgpu_keypoint *__restrict__ kp, const uint8_t *__restrict__ desc,
                                                   int capacity, const int *__restrict__ to_frame, double window,
                                                   GridDims g, const int *__restrict__ to_items,
                                                   const int *__restrict__ to_counts, const int *__restrict__ cell_start,
                                                   const int *__restrict__ s_idx, const uint4 *__restrict__ s_desc,
                                                   const double2 *__restrict__ s_xy, uint2 *__restrict__ partial) {
  const int p = blockIdx.y, nt = to_counts[p];
  if ((int)(blockIdx.x * 256) >= nt) return;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool live = q < nt;
  const size_t out = (size_t)p * capacity;
  const int li = to_items[out + min(q, nt - 1)];
  const size_t row = (size_t)to_frame[p] * capacity + li;
  const orbgpu_keypoint k = kp[row];
  const uint4 *td = reinterpret_cast<const uint4 *>(desc + row * 32);
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  ToWindow tw;
  tw.x = k.x;
  tw.y = k.y;
  tw.win = window;
  tw.lmin = tw.lmax = 0;
  tw.check_levels = false;  // the binned rows are the level-0 rows already
  unsigned best = KEY_NONE, second = KEY_NONE;
  grid_scan(tw, live, t, g, cell_start + (size_t)p * (GRID_CELLS + 1), s_desc + 2 * out, s_xy + out, nullptr, s_idx + out,
            best, second);
  if (live) partial[out + li] = make_uint2(best, second);
}

// ------------------------------------------------------------------------------------------------
// Fused frame-to-frame linking: one workgroup per pair.  AreaToIterator / AreaToPointee only ever pair 'to' id 0 or
// level-0 'to' ids with level-0 'from' ids (area_to_iterator.h:24-43, area_to_pointee.h:44-48), so the two index
// lists are compacted first (ascending, which keeps the reference's scan order) and only |to0| x |from0| pairs are
// evaluated -- about 1/20 of N_to x N_from at 8 levels.  Lists, claims and per-'to' results live in LDS.
template <int NT, class Pred>
__device__ __forceinline__ int block_compact_list(int n, int *list, int *s_wtot, Pred pred) {
  // ascending list of i in [0, n) with pred(i); every thread of the NT-thread block calls it; returns the count
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += NT) {
    const int i = i0 + threadIdx.x;
    const bool f = i < n && pred(i);
    const unsigned long long bal = __ballot(f);
    if (lane == 0) s_wtot[wave] = __popcll(bal);
    __syncthreads();
    int off = base, all = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) {
      off += w < wave ? s_wtot[w] : 0;
      all += s_wtot[w];
    }
    if (f) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
    base += all;
    __syncthreads();
  }
  return base;
}

// LK_NT threads: the pair's work is a chain of short phases (two compactions, the scan, the claims, the rotation
// histogram), each a few passes over ~1000 entries, so one workgroup per pair is bound by latency, not by work: 1024
// threads shorten every pass fourfold, and the four 256-thread groups split each staged chunk of 'from' candidates
// between them (46 -> see DESIGN.md for the measured time per 255 pairs).
constexpr int LK_NT = 1024, LK_G = LK_NT / 256;
__global__ __launch_bounds__(LK_NT) void k_link_fused(const orbgpu_keypoint *__restrict__ kp, const uint8_t *__restrict__ desc,
                                                      const int *__restrict__ n_kp, int capacity,
                                                      const int *__restrict__ to_frame, const int *__restrict__ from_frame,
                                                      double window, unsigned threshold, double ratio,
                                                      int check_orientation, int *__restrict__ match,
                                                      unsigned *__restrict__ dist, int *__restrict__ n_matches) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lk_lds[];
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(lk_lds);  // [capacity] per compacted 'from'
  int *to_list = reinterpret_cast<int *>(claim + capacity);
  int *from_list = to_list + capacity;
  int *mj = from_list + capacity;                             // per compacted 'to': compacted 'from' or -1
  unsigned *md = reinterpret_cast<unsigned *>(mj + capacity);
  __shared__ int s_wtot[LK_NT / 64];
  __shared__ int total;
  const int p = blockIdx.x, tid = threadIdx.x;
  const int ft = to_frame[p], ff = from_frame[p];
  const int n_to = usable_count(n_kp[ft], capacity), n_from = usable_count(n_kp[ff], capacity);
  const orbgpu_keypoint *kt = kp + (size_t)ft * capacity, *kf = kp + (size_t)ff * capacity;
  int *m = match + (size_t)p * capacity;
  unsigned *dd = dist + (size_t)p * capacity;
  for (int i = tid; i < capacity; i += LK_NT) {
    m[i] = -1;
    dd[i] = 0;
    claim[i] = ~0ull;
  }
  if (tid == 0) total = 0;
  // AreaToIterator visits id 0 whatever its level, later ids only when level == 0
  const int n_to0 = block_compact_list<LK_NT>(n_to, to_list, s_wtot, [&](int i) { return i == 0 || kt[i].level == 0; });
  const int n_from0 = block_compact_list<LK_NT>(n_from, from_list, s_wtot, [&](int j) { return kf[j].level == 0; });
  const uint4 *__restrict__ fdesc = reinterpret_cast<const uint4 *>(desc + (size_t)ff * capacity * 32);
  // the compacted 'from' set is staged through LDS 256 candidates at a time (descriptor + position), so the scan reads
  // wave-uniform LDS addresses (broadcast) instead of chasing list index -> descriptor through memory per candidate.
  // Thread (group gq = tid / 256, slot tid % 256): 'to' slot s0 + tid % 256 against candidates gq, gq + LK_G, ... of the
  // chunk; the groups' (best, second) pairs are merged through LDS
  __shared__ uint4 s_fdesc[256][2];
  __shared__ double2 s_fxy[256];
  __shared__ uint2 s_part[LK_G - 1][256];
  const int gq = tid >> 8, sl = tid & 255;
  for (int s0 = 0; s0 < n_to0; s0 += 256) {
    const int slot = s0 + sl;
    const int ti = to_list[min(slot, n_to0 - 1)];
    const orbgpu_keypoint k = kt[ti];
    const uint4 *td = reinterpret_cast<const uint4 *>(desc + ((size_t)ft * capacity + ti) * 32);
    const uint4 t0 = td[0], t1 = td[1];
    const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    unsigned best = KEY_NONE, second = KEY_NONE;
    for (int c0 = 0; c0 < n_from0; c0 += 256) {
      const int nc = min(256, n_from0 - c0);
      __syncthreads();  // the previous chunk has been consumed
      if (tid < nc) {
        const int j = from_list[c0 + tid];
        s_fdesc[tid][0] = fdesc[2 * j];
        s_fdesc[tid][1] = fdesc[2 * j + 1];
        s_fxy[tid] = make_double2(kf[j].x, kf[j].y);
      }
      __syncthreads();
#pragma unroll 4
      for (int c = gq; c < nc; c += LK_G) {
        const uint4 a = s_fdesc[c][0], b = s_fdesc[c][1];
        const double2 fxy = s_fxy[c];
        unsigned key = (hamming8(t, a, b) << IDX_BITS) | (unsigned)(c0 + c);
        // ListFeaturesInArea(point, window, 0, 0): strict |dx| < r and |dy| < r in double (features.cpp:109-119)
        const bool ok = fabs(fxy.x - k.x) < window && fabs(fxy.y - k.y) < window;
        key = ok ? key : KEY_NONE;
        second = umed3(best, key, second);
        best = min(best, key);
      }
    }
    if (gq > 0) s_part[gq - 1][sl] = make_uint2(best, second);
    __syncthreads();
    if (gq == 0) {
#pragma unroll
      for (int q = 0; q < LK_G - 1; q++) merge_keys(best, second, s_part[q][sl].x, s_part[q][sl].y);
      if (slot < n_to0) {
        unsigned d = 0;
        const int cm = snn_accept(best, second, threshold, ratio, d);
        mj[slot] = cm;
        md[slot] = d;
        dd[ti] = d;
        if (cm >= 0) atomicMin(&claim[cm], ((unsigned long long)d << 32) | (unsigned)ti);
      }
    }
  }
  __syncthreads();
  int cnt = 0;
  for (int slot = tid; slot < n_to0; slot += LK_NT) {
    const int cm = mj[slot];
    const int ti = to_list[slot];
    if (cm >= 0 && claim[cm] == (((unsigned long long)md[slot] << 32) | (unsigned)ti)) {
      m[ti] = from_list[cm];
      cnt++;
    }
  }
  if (cnt) atomicAdd(&total, cnt);
  __syncthreads();  // also orders this block's global writes to m[] before the filter reads them
  if (!check_orientation) {
    if (tid == 0) n_matches[p] = total;
    return;
  }
  rotation_filter_block(
      n_to, m, [&](int i) { return kt[i].angle; }, [&](int j) { return kf[j].angle; }, nullptr, &n_matches[p]);
}

// ------------------------------------------------------------------------------------------------
// Node-grouped matching (DBoW2Handler::FastMatch, handlers/DBoW2/dbo_w2_handler.cpp:25-72; BowToIterator /
// BowToPointee, bow_to_iterator.cpp:12-47, bow_to_pointee.h:37-45).  The caller has joined the two feature vectors:
// group g holds the 'to' ids to_ids[to_off[g] .. to_off[g+1]) and the 'from' ids from_ids[from_off[g] ..) of one
// vocabulary node, groups in node order.  One lane per visiting-order slot s; the candidates of a slot are its
// group's 'from' ids in list order (key = distance << 23 | list position, so the first candidate wins ties).
// Uniqueness: smallest distance, then the earliest slot -- the order MatchWithIterators visits the 'to' ids.
__global__ __launch_bounds__(256) void k_group_scan(orbgpu_group_args a, int n_slots, int *__restrict__ slot_from,
                                                    unsigned *__restrict__ slot_dist,
                                                    unsigned long long *__restrict__ claim) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n_slots) return;
  // group of the slot: last g with to_off[g] <= s
  int lo = 0, hi = a.n_groups;  // invariant: to_off[lo] <= s < to_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.to_off[mid] <= s) lo = mid; else hi = mid;
  }
  const int c0 = a.from_off[lo], c1 = a.from_off[lo + 1];
  const uint4 *td = reinterpret_cast<const uint4 *>(a.to_desc) + 2 * (size_t)a.to_ids[s];
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  const uint4 *fd = reinterpret_cast<const uint4 *>(a.from_desc);
  unsigned best = KEY_NONE, second = KEY_NONE;
  for (int c = c0; c < c1; c++) {
    const unsigned j = a.from_ids[c];
    const uint4 fa = fd[2 * (size_t)j], fb = fd[2 * (size_t)j + 1];
    const unsigned key = (hamming8(t, fa, fb) << IDX_BITS) | (unsigned)(c - c0);
    second = umed3(best, key, second);
    best = min(best, key);
  }
  unsigned d = 0;
  const int pos = snn_accept(best, second, a.threshold, a.ratio, d);
  const int j = pos >= 0 ? (int)a.from_ids[c0 + pos] : -1;
  slot_from[s] = j;
  slot_dist[s] = d;
  if (j >= 0) atomicMin(&claim[j], ((unsigned long long)d << 32) | (unsigned)s);
}

__global__ __launch_bounds__(256) void k_group_resolve(orbgpu_group_args a, int n_slots, const int *__restrict__ slot_from,
                                                       const unsigned *__restrict__ slot_dist,
                                                       const unsigned long long *__restrict__ claim,
                                                       int *__restrict__ from_idx, unsigned *__restrict__ dist) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n_slots) return;
  const int j = slot_from[s];
  if (j >= 0 && claim[j] == (((unsigned long long)slot_dist[s] << 32) | (unsigned)s)) {
    const unsigned i = a.to_ids[s];
    from_idx[i] = j;
    dist[i] = slot_dist[s];
  }
}

// ------------------------------------------------------------------------------------------------
// Vocabulary tree descent: TemplatedVocabulary<FORB>::transform for one feature
// (DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259) with FORB::distance (FORB.cpp:81-101).  One lane per descriptor; at
// every level the lane scans the children of its current node in Node::children order and keeps the first minimum
// (strict <, :1243).  node_out = the ancestor at level L - levelsup (root when that is <= 0).
__global__ __launch_bounds__(256) void k_bow_transform(const uint8_t *__restrict__ desc, int n,
                                                       const int *__restrict__ child_off,
                                                       const unsigned *__restrict__ children,
                                                       const uint8_t *__restrict__ node_desc,
                                                       const unsigned *__restrict__ word_id, int nid_level, int max_depth,
                                                       unsigned *__restrict__ word_out, unsigned *__restrict__ node_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4 *td = reinterpret_cast<const uint4 *>(desc) + 2 * (size_t)i;
  const uint4 t0 = td[0], t1 = td[1];
  const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  const uint4 *nd = reinterpret_cast<const uint4 *>(node_desc);
  unsigned cur = 0, nid = 0;
  int level = 0;
  int c0 = child_off[0], c1 = child_off[1];
  while (c1 > c0 && level < max_depth) {  // max_depth bounds the walk on a malformed (cyclic) tree
    ++level;
    unsigned best = 0xffffffffu;
    for (int c = c0; c < c1; c++) {
      const unsigned id = children[c];
      const unsigned d = hamming8(t, nd[2 * (size_t)id], nd[2 * (size_t)id + 1]);
      if (d < best) {
        best = d;
        cur = id;
      }
    }
    if (level == nid_level) nid = cur;
    c0 = child_off[cur];
    c1 = child_off[cur + 1];
  }
  word_out[i] = word_id[cur];
  node_out[i] = nid;
}

// ------------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptor (map/map_point.cpp:135-171), batched: one wave per map point, lane = row
// of the distance matrix (rows beyond 64 in further passes).  The element at position (N-1)/2 of the sorted row is
// the smallest value m with #{j : d(i,j) <= m} >= (N-1)/2 + 1; distances are 0..256, so 9 bisection steps over m find
// it, each re-evaluating the row against the wave-uniform (scalar-loaded) descriptors of the group.  The winner is
// the minimum of (median << 16 | row): smallest median, first row on ties (strict <, :162).
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t *__restrict__ desc, const int *__restrict__ offsets,
                                                     int n_points, int *__restrict__ best) {
  const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int lane = threadIdx.x & 63;
  if (p >= n_points) return;
  const int o0 = offsets[p], N = offsets[p + 1] - o0;
  if (N <= 0) {
    if (lane == 0) best[p] = -1;
    return;
  }
  const uint4 *gd = reinterpret_cast<const uint4 *>(desc) + 2 * (size_t)o0;
  const int need = ((N - 1) >> 1) + 1;  // (size_t)(0.5 * (N - 1)) + 1 elements must be <= the median
  unsigned key = 0xffffffffu;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    const uint4 t0 = gd[2 * (size_t)min(i, N - 1)], t1 = gd[2 * (size_t)min(i, N - 1) + 1];
    const unsigned t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    int lo = 0, hi = 256;  // invariant: count(<= hi) >= need; answer in [lo, hi]
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      int cnt = 0;
      for (int j = 0; j < N; j++) cnt += hamming8(t, gd[2 * (size_t)j], gd[2 * (size_t)j + 1]) <= (unsigned)mid ? 1 : 0;
      if (cnt >= need) hi = mid; else lo = mid + 1;
    }
    if (i < N) key = min(key, ((unsigned)lo << 16) | (unsigned)i);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
  if (lane == 0) best[p] = (int)(key & 0xffffu);
}

// ------------------------------------------------------------------------------------------------
// All-pairs keyframe matching: 'to' = keyframe r of the row set, 'from' = keyframe c of the column set,
// SNNMatcher::MatchWithIterators semantics with no candidate restriction (every 'from' in index order), then the
// optional rotation histogram.  One workgroup owns one (r, c) pair from the scan to the match count, so nothing but
// the count (and, on request, the match list) touches global memory: the per-'from' claims and the per-'to' results
// live in LDS.  A lane holds AP_T 'to' descriptors, the 'from' descriptors are wave-uniform scalar loads.
// Work order: XCD k takes rows k, k+8, ... and walks their columns, so a row's descriptors stay in that XCD's L2.
constexpr int AP_T = 4;

// The rotation histogram reads one float per keypoint: ang_rows / ang_cols point at the angle of keypoint 0 of
// keyframe 0 and ang_stride is the distance in bytes between consecutive keypoints' angles (32 for orbgpu_keypoint
// records, 4 for a plain float side array -- what crosses xGMI in the sharded workload).
__global__ __launch_bounds__(256) void k_allpairs(const char *__restrict__ ang_rows,
                                                  const uint8_t *__restrict__ desc_rows, const int *__restrict__ n_rows_kp,
                                                  int n_rows, const char *__restrict__ ang_cols, int ang_stride,
                                                  const uint8_t *__restrict__ desc_cols, const int *__restrict__ n_cols_kp,
                                                  int n_cols, int capacity, int row_offset, unsigned threshold,
                                                  double ratio, int check_orientation, int *__restrict__ n_matches,
                                                  int *__restrict__ match_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ap_lds[];
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(ap_lds);  // [capacity] per 'from'
  int *mj = reinterpret_cast<int *>(claim + capacity);                        // [capacity] per 'to': from index or -1
  unsigned *md = reinterpret_cast<unsigned *>(mj + capacity);                 // [capacity] per 'to': distance
  __shared__ int total;
  int c, r;
  if (!xcd_item_frame(n_cols, n_rows, c, r)) return;
  int *out_n = n_matches + (size_t)r * n_cols + c;
  int *out_m = match_out ? match_out + ((size_t)r * n_cols + c) * capacity : nullptr;
  const int n_to = usable_count(n_rows_kp[r], capacity), n_from = usable_count(n_cols_kp[c], capacity);
  if ((row_offset >= 0 && row_offset + r == c) || n_to == 0 || n_from == 0) {  // a keyframe is not matched with itself
    if (threadIdx.x == 0) *out_n = 0;
    if (out_m)
      for (int i = threadIdx.x; i < capacity; i += 256) out_m[i] = -1;
    return;
  }
  for (int j = threadIdx.x; j < n_from; j += 256) claim[j] = ~0ull;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const uint4 *__restrict__ fd = reinterpret_cast<const uint4 *>(desc_cols + (size_t)c * capacity * 32);
  const uint4 *__restrict__ tdb = reinterpret_cast<const uint4 *>(desc_rows + (size_t)r * capacity * 32);
  for (int i0 = 0; i0 < n_to; i0 += 256 * AP_T) {
    unsigned t[AP_T][8], best[AP_T], second[AP_T];
    int row[AP_T];
#pragma unroll
    for (int q = 0; q < AP_T; q++) {
      row[q] = i0 + q * 256 + threadIdx.x;
      const uint4 *td = tdb + 2 * (size_t)min(row[q], n_to - 1);
      const uint4 t0 = td[0], t1 = td[1];
      t[q][0] = t0.x; t[q][1] = t0.y; t[q][2] = t0.z; t[q][3] = t0.w;
      t[q][4] = t1.x; t[q][5] = t1.y; t[q][6] = t1.z; t[q][7] = t1.w;
      best[q] = second[q] = KEY_NONE;
    }
    auto eval = [&](int j, const uint4 &fa, const uint4 &fb) {
#pragma unroll
      for (int q = 0; q < AP_T; q++) {
        const unsigned key = (hamming8(t[q], fa, fb) << IDX_BITS) | (unsigned)j;
        second[q] = umed3(best[q], key, second[q]);
        best[q] = min(best[q], key);
      }
    };
    int j = 0;
    const int jfull = (n_from / SNN_U) * SNN_U;
    if (j < jfull) {
      uint4 ca[SNN_U], cb[SNN_U];
#pragma unroll
      for (int u = 0; u < SNN_U; u++) {
        ca[u] = fd[2 * u];
        cb[u] = fd[2 * u + 1];
      }
      for (; j < jfull; j += SNN_U) {
        uint4 na[SNN_U], nb[SNN_U];
        const int jn = min(j + SNN_U, jfull - SNN_U);
#pragma unroll
        for (int u = 0; u < SNN_U; u++) {
          na[u] = fd[2 * (jn + u)];
          nb[u] = fd[2 * (jn + u) + 1];
        }
#pragma unroll
        for (int u = 0; u < SNN_U; u++) eval(j + u, ca[u], cb[u]);
#pragma unroll
        for (int u = 0; u < SNN_U; u++) {
          ca[u] = na[u];
          cb[u] = nb[u];
        }
      }
    }
    for (; j < n_from; j++) eval(j, fd[2 * j], fd[2 * j + 1]);
#pragma unroll
    for (int q = 0; q < AP_T; q++) {
      if (row[q] >= n_to) continue;
      unsigned d = 0;
      const int jm = snn_accept(best[q], second[q], threshold, ratio, d);
      mj[row[q]] = jm;
      md[row[q]] = d;
      if (jm >= 0) atomicMin(&claim[jm], ((unsigned long long)d << 32) | (unsigned)row[q]);
    }
  }
  __syncthreads();
  // uniqueness (second_nearest_neighbor_matcher.hpp:57-70): a 'from' keeps its smallest distance, earliest 'to'
  int cnt = 0;
  for (int i = threadIdx.x; i < n_to; i += 256) {
    const int jm = mj[i];
    if (jm >= 0) {
      if (claim[jm] != (((unsigned long long)md[i] << 32) | (unsigned)i)) mj[i] = -1;
      else cnt++;
    }
  }
  if (cnt) atomicAdd(&total, cnt);
  __syncthreads();
  if (check_orientation) {
    const char *kt = ang_rows + (size_t)r * capacity * ang_stride, *kf = ang_cols + (size_t)c * capacity * ang_stride;
    rotation_filter_block(
        n_to, mj, [&](int i) { return *reinterpret_cast<const float *>(kt + (size_t)i * ang_stride); },
        [&](int jj) { return *reinterpret_cast<const float *>(kf + (size_t)jj * ang_stride); }, nullptr, out_n);
    __syncthreads();
  } else if (threadIdx.x == 0) {
    *out_n = total;
  }
  if (out_m)
    for (int i = threadIdx.x; i < capacity; i += 256) out_m[i] = i < n_to ? mj[i] : -1;
}

// ------------------------------------------------------------------------------------------------
// EXPLORATORY twin of k_allpairs on the i8 matrix cores (orbgpu_set_bruteforce_engine; the popcount kernel above stays
// the default and the contract path): the scan of k_snn_mfma -- 'to' rows expanded once into the B operand (4 waves x
// MF_T tiles of 32 = 512 rows per turn), the keyframe's 'from' rows streamed through LDS 32 at a time as +1 / -1 bytes,
// tiles without a pair inside the threshold dropped after an 8-instruction maximum -- followed by k_allpairs' own tail:
// ratio / threshold, per-'from' claims and uniqueness in LDS, rotation histogram, count.  A keyframe of more than 512
// keypoints takes further turns over the 'from' rows.  Same results bit for bit (tests/test_gpu_allpairs.py runs both).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_allpairs_mfma(
    const char *__restrict__ ang_rows, const uint8_t *__restrict__ desc_rows, const int *__restrict__ n_rows_kp, int n_rows,
    const char *__restrict__ ang_cols, int ang_stride, const uint8_t *__restrict__ desc_cols,
    const int *__restrict__ n_cols_kp, int n_cols, int capacity, int row_offset, unsigned threshold, double ratio,
    int check_orientation, int *__restrict__ n_matches, int *__restrict__ match_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ap_lds[];
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(ap_lds);  // [capacity] per 'from'
  int *mj = reinterpret_cast<int *>(claim + capacity);                        // [capacity] per 'to': from index or -1
  unsigned *md = reinterpret_cast<unsigned *>(mj + capacity);                 // [capacity] per 'to': distance
  __shared__ __attribute__((aligned(16))) unsigned abuf[2][8 * 2 * 32 * 4];   // [buffer][m][h][row][4 dwords]
  __shared__ __attribute__((aligned(8))) uint2 lut[256];                      // byte -> its 8 bits as +1 / -1 bytes
  __shared__ int total;
  int c, r;
  if (!xcd_item_frame(n_cols, n_rows, c, r)) return;
  int *out_n = n_matches + (size_t)r * n_cols + c;
  int *out_m = match_out ? match_out + ((size_t)r * n_cols + c) * capacity : nullptr;
  const int n_to = usable_count(n_rows_kp[r], capacity), n_from = usable_count(n_cols_kp[c], capacity);
  if ((row_offset >= 0 && row_offset + r == c) || n_to == 0 || n_from == 0) {  // a keyframe is not matched with itself
    if (threadIdx.x == 0) *out_n = 0;
    if (out_m)
      for (int i = threadIdx.x; i < capacity; i += 256) out_m[i] = -1;
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  for (int j = tid; j < n_from; j += 256) claim[j] = ~0ull;
  if (tid == 0) total = 0;
  lut[tid] = make_uint2(expand_pm1(tid, 0), expand_pm1(tid, 1));
  const unsigned *__restrict__ td = reinterpret_cast<const unsigned *>(desc_rows + (size_t)r * capacity * 32);
  const unsigned *__restrict__ fd = reinterpret_cast<const unsigned *>(desc_cols + (size_t)c * capacity * 32);
  const int dmin = 256 - 2 * (int)threshold;  // dot >= dmin  <=>  distance <= threshold
  const int sr = tid & 31, sm = tid >> 5;     // staging: dword sm of 'from' row sr of the step
  auto load_raw = [&](int jb) {
    const int j = jb + sr;
    return j < n_from ? fd[(size_t)j * 8 + sm] : 0u;
  };
  auto store_exp = [&](int buf, unsigned w) {
    const uint2 e0 = lut[w & 255u], e1 = lut[(w >> 8) & 255u], e2 = lut[(w >> 16) & 255u], e3 = lut[w >> 24];
    *reinterpret_cast<uint4 *>(&abuf[buf][((sm * 2 + 0) * 32 + sr) * 4]) = make_uint4(e0.x, e0.y, e1.x, e1.y);
    *reinterpret_cast<uint4 *>(&abuf[buf][((sm * 2 + 1) * 32 + sr) * 4]) = make_uint4(e2.x, e2.y, e3.x, e3.y);
  };
  __syncthreads();  // lut, claims
  for (int tb = 0; tb < n_to; tb += 4 * MF_T * 32) {  // turns of 512 'to' rows
    const int to0 = tb + wv * (MF_T * 32);
    v4i bfrag[MF_T][8];
#pragma unroll
    for (int t = 0; t < MF_T; t++) {
      const int row = min(to0 + t * 32 + col, n_to - 1);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const unsigned w = td[(size_t)row * 8 + m] >> (16 * h);
        bfrag[t][m] = v4i{(int)expand_pm1(w, 0), (int)expand_pm1(w, 1), (int)expand_pm1(w, 2), (int)expand_pm1(w, 3)};
      }
    }
    unsigned best[MF_T], second[MF_T];
#pragma unroll
    for (int t = 0; t < MF_T; t++) best[t] = second[t] = KEY_NONE;
    store_exp(0, load_raw(0));
    __syncthreads();
    int buf = 0;
    for (int jb = 0; jb < n_from; jb += 32, buf ^= 1) {
      const bool more = jb + 32 < n_from;
      unsigned raw = 0;
      if (more) raw = load_raw(jb + 32);
      v16i acc[MF_T];
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const uint4 av = *reinterpret_cast<const uint4 *>(&abuf[buf][((m * 2 + h) * 32 + col) * 4]);
        const v4i af = v4i{(int)av.x, (int)av.y, (int)av.z, (int)av.w};
#pragma unroll
        for (int t = 0; t < MF_T; t++) {
          if (m == 0) acc[t] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bfrag[t][m], acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < MF_T; t++) {
        int mx = acc[t][0];
#pragma unroll
        for (int q = 1; q < 16; q++) mx = max(mx, acc[t][q]);
        if (__ballot(mx >= dmin)) {
          const unsigned cb = (1u << 30) + (unsigned)(jb + 4 * h);
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const int roff = (q & 3) + 8 * (q >> 2);  // C/D layout: row = (reg & 3) + 8 (reg >> 2) + 4 h
            unsigned key = (unsigned)__mul24(acc[t][q], -(1 << 22)) + cb + (unsigned)roff;
            key = (key & ((1u << IDX_BITS) - 1u)) < (unsigned)n_from ? key : KEY_NONE;  // rows staged as zero bytes
            second[t] = umed3(best[t], key, second[t]);
            best[t] = min(best[t], key);
          }
        }
      }
      if (more) store_exp(buf ^ 1, raw);
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < MF_T; t++) {
      merge_keys(best[t], second[t], __shfl_xor(best[t], 32), __shfl_xor(second[t], 32));
      const int row = to0 + t * 32 + col;
      if (h == 0 && row < n_to) {
        unsigned d = 0;
        const int jm = snn_accept(best[t], second[t], threshold, ratio, d);
        mj[row] = jm;
        md[row] = d;
        if (jm >= 0) atomicMin(&claim[jm], ((unsigned long long)d << 32) | (unsigned)row);
      }
    }
  }
  __syncthreads();
  // uniqueness (second_nearest_neighbor_matcher.hpp:57-70): a 'from' keeps its smallest distance, earliest 'to'
  int cnt = 0;
  for (int i = tid; i < n_to; i += 256) {
    const int jm = mj[i];
    if (jm >= 0) {
      if (claim[jm] != (((unsigned long long)md[i] << 32) | (unsigned)i)) mj[i] = -1;
      else cnt++;
    }
  }
  if (cnt) atomicAdd(&total, cnt);
  __syncthreads();
  if (check_orientation) {
    const char *kt = ang_rows + (size_t)r * capacity * ang_stride, *kf = ang_cols + (size_t)c * capacity * ang_stride;
    rotation_filter_block(
        n_to, mj, [&](int i) { return *reinterpret_cast<const float *>(kt + (size_t)i * ang_stride); },
        [&](int jj) { return *reinterpret_cast<const float *>(kf + (size_t)jj * ang_stride); }, nullptr, out_n);
    __syncthreads();
  } else if (tid == 0) {
    *out_n = total;
  }
  if (out_m)
    for (int i = tid; i < capacity; i += 256) out_m[i] = i < n_to ? mj[i] : -1;
}

// ------------------------------------------------------------------------------------------------
// popcount ceiling: the inner loop of snn_scan with the 'from' words synthesised in SGPRs (no memory)
__global__ __launch_bounds__(256) void k_popcount_peak(unsigned *sink, int iters) {
  unsigned t[8];
#pragma unroll
  for (int k = 0; k < 8; k++) t[k] = (threadIdx.x + 1) * 2654435761u + k * 40503u + blockIdx.x;
  unsigned acc0 = 0, acc1 = 0;
  unsigned s = blockIdx.x * 7u + 1u;
#pragma unroll 1
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      s = s * 1664525u + 1013904223u;  // scalar LCG: wave-uniform 'from' words (stay in SGPRs)
      uint4 a = make_uint4(s, s + 0x9e3779b9u, s + 0x3c6ef372u, s + 0xdaa66d2bu);
      uint4 b = make_uint4(s + 0x78dde6e4u, s + 0x1715609du, s + 0xb54cda56u, s + 0x5384540fu);
      unsigned d = hamming8(t, a, b);
      asm volatile("" : "+v"(d));  // the pair's distance must exist in a VGPR
      if (u & 1) acc1 ^= d; else acc0 ^= d;
    }
  }
  if ((acc0 ^ acc1) == 0x12345u) sink[0] = acc0;
}

}  // namespace

static int pick_chunks(int n_to, int n_from, int &chunk, int max_chunks = INT_MAX) {
  // enough waves to fill 256 CUs several times over, but chunks of at least 256 'from' rows
  const int to_waves = (n_to + 63) / 64;
  int want = (8192 + to_waves - 1) / to_waves;
  int maxc = std::min((n_from + 255) / 256, max_chunks);
  int n_chunks = std::max(1, std::min(want, maxc));
  chunk = (n_from + n_chunks - 1) / n_chunks;
  n_chunks = (n_from + chunk - 1) / std::max(chunk, 1);
  return std::max(n_chunks, 1);
}

// the small-problem path (k_snn_small): scratch = claims, per-chunk partials, accepted pairs
bool match_small_fits(int n_to, int n_from) { return n_to > 0 && n_from > 0 && (long long)n_to * n_from <= (4ll << 20); }
size_t match_small_scratch_bytes(int n_to, int n_from) {
  const size_t n_chunks = (size_t)(n_from + 63) / 64;
  return pad256((size_t)n_from * 8) + pad256(n_chunks * (size_t)n_to * 8) + pad256((size_t)n_to * 8);
}
void launch_match_small(const orbgpu_ctx *c, const orbgpu_window_args &a, int *out_from_idx, unsigned *out_dist, void *scratch) {
  const int n_chunks = (a.n_from + 63) / 64;
  char *p = reinterpret_cast<char *>(scratch);
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(p);
  p += pad256((size_t)a.n_from * 8);
  uint2 *partial = reinterpret_cast<uint2 *>(p);
  p += pad256((size_t)n_chunks * (size_t)a.n_to * 8);
  uint2 *accepted = reinterpret_cast<uint2 *>(p);
  const dim3 grid((a.n_to + 63) / 64, n_chunks);
  if (a.to_xy) hipLaunchKernelGGL((k_snn_small<true>), grid, dim3(64), 0, c->stream, a, partial, claim);
  else hipLaunchKernelGGL((k_snn_small<false>), grid, dim3(64), 0, c->stream, a, partial, claim);
  const unsigned th = std::min(a.threshold, 256u);
  hipLaunchKernelGGL(k_snn_small_finalize, dim3((a.n_to + 255) / 256), dim3(256), 0, c->stream, partial, n_chunks, a.n_to,
                     a.to_active, th, a.ratio, accepted, claim);
  hipLaunchKernelGGL(k_snn_small_resolve, dim3((a.n_to + 255) / 256), dim3(256), 0, c->stream, a.n_to, accepted, claim,
                     out_from_idx, out_dist);
}

size_t match_scratch_size(int n_to, int n_from) {
  int chunk;
  int n_chunks = pick_chunks(n_to, n_from, chunk);
  return (size_t)n_chunks * n_to * sizeof(uint2) + (size_t)n_from * sizeof(unsigned long long) + 256;
}

void launch_match(const orbgpu_ctx *c, const orbgpu_window_args &a, int *d_from_idx, unsigned *d_dist, void *scratch,
                  int max_chunks, bool mfma) {
  int chunk;
  int n_chunks = pick_chunks(a.n_to, a.n_from, chunk, max_chunks);
  if (mfma && !a.to_xy) {
    // 512 'to' rows per workgroup, two workgroups per CU: the number of 'from' chunks that leaves the last round of
    // 512 workgroups fullest (chunks a multiple of the 32-row step and at least 1024 rows; never more chunks than the
    // popcount path would use -- the scratch is sized for those)
    const int to_blocks = (a.n_to + 511) / 512;
    const int maxc = std::max(1, std::min(n_chunks, std::min(16, a.n_from / 1024)));
    int want = 1;
    double best_eff = 0.0;
    for (int cc = 1; cc <= maxc; cc++) {
      const int wg = to_blocks * cc;
      const double eff = (double)wg / (((wg + 511) / 512) * 512.0);
      if (eff > best_eff + 0.02) {
        best_eff = eff;
        want = cc;
      }
    }
    chunk = (((a.n_from + want - 1) / want) + 31) & ~31;
    n_chunks = (a.n_from + chunk - 1) / chunk;
  }
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(scratch);
  uint2 *partial = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(scratch) +
                                             (((size_t)a.n_from * sizeof(unsigned long long) + 255) & ~(size_t)255));
  (void)hipMemsetAsync(claim, 0xff, (size_t)a.n_from * sizeof(unsigned long long), c->stream);
  if (a.to_xy) {
    dim3 grid((a.n_to + 255) / 256, n_chunks);
    hipLaunchKernelGGL((k_snn_partial<true, 1>), grid, dim3(256), 0, c->stream, a, chunk, partial);
  } else if (mfma) {
    orbgpu_window_args b = a;
    b.threshold = std::min(a.threshold, 256u);
    dim3 grid((a.n_to + 511) / 512, n_chunks);
    hipLaunchKernelGGL(k_snn_mfma, grid, dim3(256), 0, c->stream, b, chunk, partial);
  } else if (a.n_to >= 16384) {
    dim3 grid((a.n_to + 511) / 512, n_chunks);
    hipLaunchKernelGGL((k_snn_partial<false, 2>), grid, dim3(256), 0, c->stream, a, chunk, partial);
  } else {
    dim3 grid((a.n_to + 255) / 256, n_chunks);
    hipLaunchKernelGGL((k_snn_partial<false, 1>), grid, dim3(256), 0, c->stream, a, chunk, partial);
  }
  const unsigned th = std::min(a.threshold, 256u);
  hipLaunchKernelGGL(k_snn_finalize, dim3((a.n_to + 255) / 256), dim3(256), 0, c->stream, partial, n_chunks, a.n_to,
                     a.to_active, th, a.ratio, d_from_idx, d_dist, claim);
  hipLaunchKernelGGL(k_snn_resolve, dim3((a.n_to + 255) / 256), dim3(256), 0, c->stream, a.n_to, d_from_idx, d_dist,
                     claim);
}

// ---- scratch of the entry point that allocates nothing (orbgpu_match_bruteforce_noalloc_device) --------------------
// One array per GPU, inside the code object: claims [n_from] u64, then the (best, second) partials [chunks][n_to].
// `scratch` is only a host-side handle of the symbol; the address on a device comes from hipGetSymbolAddress with
// that device current (match_noalloc_init, called by orbgpu_create) and is what every launch and memset receives.
static __device__ __align__(256) unsigned char scratch[ORBGPU_NOALLOC_SCRATCH_BYTES];
static void *g_noalloc_ptr[ORBGPU_NOALLOC_MAX_DEVICES];
static std::mutex g_noalloc_mu;

hipError_t match_noalloc_init(int device) {
  if (device < 0 || device >= ORBGPU_NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_noalloc_mu);
  if (g_noalloc_ptr[device]) return hipSuccess;
  void *p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // the caller made `device` current
  if (e != hipSuccess) return e;
  g_noalloc_ptr[device] = p;
  return hipSuccess;
}

void *match_noalloc_ptr(int device) {
  if (device < 0 || device >= ORBGPU_NOALLOC_MAX_DEVICES) return nullptr;
  std::lock_guard<std::mutex> lock(g_noalloc_mu);
  return g_noalloc_ptr[device];
}

// chunks of the 'from' range that fit the fixed scratch (0: not even one does)
int match_noalloc_chunks(int n_to, int n_from) {
  const size_t claim = ((size_t)n_from * sizeof(unsigned long long) + 255) & ~(size_t)255;
  if (claim >= ORBGPU_NOALLOC_SCRATCH_BYTES) return 0;
  const size_t fit = (ORBGPU_NOALLOC_SCRATCH_BYTES - claim) / ((size_t)n_to * sizeof(uint2));
  return (int)std::min<size_t>(fit, 1u << 20);
}

size_t link_fused_lds_bytes(int capacity) { return (size_t)capacity * 24; }

static GridDims grid_dims(double image_width, double image_height) {
  // features.cpp:13-16
  return GridDims{image_width, image_height, (double)GRID_COLS / image_width, (double)GRID_ROWS / image_height};
}

void launch_match_frames(const orbgpu_ctx *c, const orbgpu_keypoint *d_kp, const uint8_t *d_desc, const int *d_n,
                         int capacity, const int *d_to_frame, const int *d_from_frame, int n_pairs, uint64_t window,
                         unsigned threshold, double ratio, int check_orientation, int *d_match, unsigned *d_dist,
                         int *d_n_matches) {
  if (link_fused_lds_bytes(capacity) + 20 * 1024 <= 64 * 1024) {  // + the kernel's static staging / merge buffers
    hipLaunchKernelGGL(k_link_fused, dim3(n_pairs), dim3(LK_NT), link_fused_lds_bytes(capacity), c->stream, d_kp, d_desc, d_n,
                       capacity, d_to_frame, d_from_frame, (double)window, std::min(threshold, 256u), ratio,
                       check_orientation, d_match, d_dist, d_n_matches);
    return;
  }
  const unsigned th = std::min(threshold, 256u);
  if (capacity < (1 << GRID_ITEM_BITS) && !c->link_plain_scan) {
    // large capacities: the level-0 rows of every pair binned by the 64 x 48 search grid laid over the context's image,
    // then k_link_grid instead of the full predicated scan (same best / second per 'to' row; finalize / resolve as below)
    const size_t rows = (size_t)n_pairs * capacity;
    unsigned char *sp = reinterpret_cast<unsigned char *>(c->d_match_scratch);
    auto take = [&](size_t bytes) {
      void *q = sp;
      sp += pad256(bytes);
      return q;
    };
    unsigned long long *claim = reinterpret_cast<unsigned long long *>(take(rows * 8));
    uint2 *partial = reinterpret_cast<uint2 *>(take(rows * 8));
    int *tmp = reinterpret_cast<int *>(take(2 * rows * 4));
    int *to_items = reinterpret_cast<int *>(take(rows * 4));
    int *to_counts = reinterpret_cast<int *>(take((size_t)n_pairs * 4));
    int *cell_start = reinterpret_cast<int *>(take((size_t)n_pairs * (GRID_CELLS + 1) * 4));
    int *s_idx = reinterpret_cast<int *>(take(rows * 4));
    double2 *s_xy = reinterpret_cast<double2 *>(take(rows * 16));
    uint4 *s_desc = reinterpret_cast<uint4 *>(take(rows * 32));
    (void)hipMemsetAsync(claim, 0xff, 2 * pad256(rows * 8), c->stream);  // claims and the (KEY_NONE, KEY_NONE) partials
    const GridDims g = grid_dims((double)c->geom.lv[0].w, (double)c->geom.lv[0].h);
    hipLaunchKernelGGL(k_link_grid_build, dim3(n_pairs, 2), dim3(GB_THREADS), 0, c->stream, d_kp, d_desc, d_n, capacity,
                       d_to_frame, d_from_frame, g, tmp, to_items, to_counts, cell_start, s_idx, s_desc, s_xy);
    hipLaunchKernelGGL(k_link_grid, dim3((capacity + 255) / 256, n_pairs), dim3(256), 0, c->stream, d_kp, d_desc, capacity,
                       d_to_frame, (double)window, g, to_items, to_counts, cell_start, s_idx, s_desc, s_xy, partial);
    hipLaunchKernelGGL(k_link_finalize, dim3((capacity + 255) / 256, n_pairs), dim3(256), 0, c->stream, d_kp, d_n,
                       capacity, d_to_frame, d_from_frame, 1, partial, th, ratio, d_match, d_dist, claim);
    hipLaunchKernelGGL(k_link_resolve, dim3(n_pairs), dim3(256), 0, c->stream, d_kp, d_n, capacity, d_to_frame,
                       d_from_frame, claim, check_orientation, d_match, d_dist, d_n_matches);
    return;
  }
  // the plain scan (ORBGPU_LINK_SCAN=1, or capacities of 2^20 and more): three kernels through global scratch
  const int chunk = 256;
  const int n_chunks = (capacity + chunk - 1) / chunk;
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(c->d_match_scratch);
  uint2 *partial = reinterpret_cast<uint2 *>(claim + (size_t)n_pairs * capacity);
  (void)hipMemsetAsync(claim, 0xff, (size_t)n_pairs * capacity * sizeof(unsigned long long), c->stream);
  dim3 grid((capacity + 255) / 256, n_chunks, n_pairs);
  hipLaunchKernelGGL(k_link_partial, grid, dim3(256), 0, c->stream, d_kp, d_desc, d_n, capacity, d_to_frame,
                     d_from_frame, (double)window, chunk, n_chunks, partial);
  hipLaunchKernelGGL(k_link_finalize, dim3((capacity + 255) / 256, n_pairs), dim3(256), 0, c->stream, d_kp, d_n,
                     capacity, d_to_frame, d_from_frame, n_chunks, partial, th, ratio, d_match, d_dist, claim);
  hipLaunchKernelGGL(k_link_resolve, dim3(n_pairs), dim3(256), 0, c->stream, d_kp, d_n, capacity, d_to_frame,
                     d_from_frame, claim, check_orientation, d_match, d_dist, d_n_matches);
}

size_t link_scratch_bytes(int n_pairs, int capacity) {
  const size_t rows = (size_t)n_pairs * capacity;
  const size_t scan = rows * 8 + rows * 8 * (size_t)((capacity + 255) / 256) + 256;
  const size_t grid = 2 * pad256(rows * 8) + pad256(2 * rows * 4) + pad256(rows * 4) + pad256((size_t)n_pairs * 4) +
                      pad256((size_t)n_pairs * (GRID_CELLS + 1) * 4) + pad256(rows * 4) + pad256(rows * 16) +
                      pad256(rows * 32) + 256;
  return std::max(scan, grid);
}

// claim: one u64 per 'from' row of the concatenated arrays (scratch holds n_from_total of them)
void launch_match_segments(const orbgpu_ctx *c, const orbgpu_window_args &a, const orbgpu_segment *d_segs, int n_segments,
                           int max_to_count, int n_from_total, int *d_from_idx, unsigned *d_dist, void *scratch) {
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(scratch);
  (void)hipMemsetAsync(claim, 0xff, (size_t)n_from_total * sizeof(unsigned long long), c->stream);
  orbgpu_window_args b = a;
  b.threshold = std::min(a.threshold, 256u);
  const dim3 grid((max_to_count + 255) / 256, n_segments);
  hipLaunchKernelGGL(k_snn_segments, grid, dim3(256), 0, c->stream, b, d_segs, d_from_idx, d_dist, claim);
  hipLaunchKernelGGL(k_snn_segments_resolve, grid, dim3(256), 0, c->stream, d_segs, d_from_idx, d_dist, claim, INT_MAX,
                     orbgpu_segment{});
}

size_t grid_build_scratch_bytes(int n_ranges, int max_count) { return (size_t)n_ranges * max_count * 4 + 256; }

// the grid by itself: ranges = n x (begin, count); items[r * max_count + k] (row - begin), cell_start[r][3073]
void launch_grid_build(const orbgpu_ctx *c, const double *d_xy, const int *d_ranges, int n_ranges, int max_count,
                       double image_width, double image_height, int *d_cell_start, int *d_items, void *scratch) {
  GridBuildArgs ga{};
  ga.xy = reinterpret_cast<const double2 *>(d_xy);
  ga.ranges = d_ranges;
  ga.range_stride = 2;
  ga.range_off = 0;
  ga.out_stride = max_count;
  ga.tmp = reinterpret_cast<int *>(scratch);
  ga.items = d_items;
  ga.cell_start = d_cell_start;
  ga.error_flag = c->d_error_flag;
  hipLaunchKernelGGL(k_grid_build, dim3(n_ranges), dim3(GB_THREADS), 0, c->stream, ga, grid_dims(image_width, image_height));
}

size_t grid_match_scratch_bytes(int n_segments, int max_to_count, int max_from_count, int n_from_total) {
  const size_t nt = (size_t)n_segments * max_to_count, nf = (size_t)n_segments * max_from_count;
  return pad256((size_t)n_from_total * 8) + pad256(nt * 4) + pad256(nt * 4) + pad256(nf * 4) +
         pad256((size_t)n_segments * (GRID_CELLS + 1) * 4) + pad256(nf * 4) + pad256(nf * 4) + pad256(nf * 16) +
         pad256(nf * 32) + 256;
}

void launch_match_grid_segments(const orbgpu_ctx *c, const orbgpu_window_args &a, const orbgpu_segment *d_segs,
                                int n_segments, int max_to_count, int max_from_count, int n_from_total,
                                double image_width, double image_height, int *d_from_idx, unsigned *d_dist, void *scratch,
                                const orbgpu_segment *one) {
  // d_segs == nullptr: the single segment *one (host memory), handed to the kernels by value
  const orbgpu_segment seg0 = one ? *one : orbgpu_segment{};
  const size_t nt = (size_t)n_segments * max_to_count, nf = (size_t)n_segments * max_from_count;
  unsigned char *sp = reinterpret_cast<unsigned char *>(scratch);
  auto take = [&](size_t bytes) {
    void *q = sp;
    sp += pad256(bytes);
    return q;
  };
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(take((size_t)n_from_total * 8));
  int *to_items = reinterpret_cast<int *>(take(nt * 4));
  int *tmp_to = reinterpret_cast<int *>(take(nt * 4));
  int *tmp_from = reinterpret_cast<int *>(take(nf * 4));
  int *cell_start = reinterpret_cast<int *>(take((size_t)n_segments * (GRID_CELLS + 1) * 4));
  int *s_idx = reinterpret_cast<int *>(take(nf * 4));
  int *s_lv = reinterpret_cast<int *>(take(nf * 4));
  double2 *s_xy = reinterpret_cast<double2 *>(take(nf * 16));
  uint4 *s_desc = reinterpret_cast<uint4 *>(take(nf * 32));
  (void)hipMemsetAsync(claim, 0xff, (size_t)n_from_total * sizeof(unsigned long long), c->stream);
  const GridDims g = grid_dims(image_width, image_height);
  const int *ranges = reinterpret_cast<const int *>(d_segs);
  // 'from' rows of every segment: binned by cell, copies in bin order; 'to' rows: Hilbert order over the image (only
  // the order is used)
  GridBuildArgs gf{}, gt{};
  gf.xy = reinterpret_cast<const double2 *>(a.from_xy);
  gf.ranges = ranges;
  gf.range_stride = 4;
  gf.range_off = 2;
  gf.out_stride = max_from_count;
  gf.tmp = tmp_from;
  gf.items = s_idx;
  gf.cell_start = cell_start;
  gf.desc = reinterpret_cast<const uint4 *>(a.from_desc);
  gf.level = a.from_level;
  gf.s_desc = s_desc;
  gf.s_xy = s_xy;
  gf.s_lv = s_lv;
  gf.error_flag = gt.error_flag = c->d_error_flag;
  for (int q = 0; q < 4; q++) gf.one[q] = gt.one[q] = (&seg0.to_begin)[q];
  gt.xy = reinterpret_cast<const double2 *>(a.to_xy);
  gt.ranges = ranges;
  gt.range_stride = 4;
  gt.range_off = 0;
  gt.out_stride = max_to_count;
  gt.tmp = tmp_to;
  gt.items = to_items;
  hipLaunchKernelGGL(k_grid_build_pair, dim3(n_segments, 2), dim3(GB_THREADS), 0, c->stream, gf, gt, g);
  orbgpu_window_args b = a;
  b.threshold = std::min(a.threshold, 256u);
  const dim3 grid((max_to_count + 255) / 256, n_segments);
  hipLaunchKernelGGL(k_snn_grid, grid, dim3(256), 0, c->stream, b, d_segs, seg0, g, max_to_count, max_from_count, to_items,
                     cell_start, s_idx, s_desc, s_xy, s_lv, d_from_idx, d_dist, claim);
  hipLaunchKernelGGL(k_snn_segments_resolve, grid, dim3(256), 0, c->stream, d_segs, d_from_idx, d_dist, claim,
                     max_to_count, seg0);
}

size_t group_scratch_bytes(int n_from, int n_slots) { return (size_t)n_from * 8 + (size_t)n_slots * 8 + 256; }

// d_from_idx / d_dist: n_to entries, set to -1 / 0 here; device_args holds device pointers
void launch_match_groups(const orbgpu_ctx *c, const orbgpu_group_args &a, int n_slots, int *d_from_idx, unsigned *d_dist,
                         void *scratch) {
  unsigned long long *claim = reinterpret_cast<unsigned long long *>(scratch);
  int *slot_from = reinterpret_cast<int *>(claim + a.n_from);
  unsigned *slot_dist = reinterpret_cast<unsigned *>(slot_from + n_slots);
  (void)hipMemsetAsync(d_from_idx, 0xff, (size_t)a.n_to * 4, c->stream);
  (void)hipMemsetAsync(d_dist, 0, (size_t)a.n_to * 4, c->stream);
  if (n_slots == 0 || a.n_from == 0) return;
  (void)hipMemsetAsync(claim, 0xff, (size_t)a.n_from * 8, c->stream);
  orbgpu_group_args b = a;
  b.threshold = std::min(a.threshold, 256u);
  const int blocks = (n_slots + 255) / 256;
  hipLaunchKernelGGL(k_group_scan, dim3(blocks), dim3(256), 0, c->stream, b, n_slots, slot_from, slot_dist, claim);
  hipLaunchKernelGGL(k_group_resolve, dim3(blocks), dim3(256), 0, c->stream, b, n_slots, slot_from, slot_dist, claim,
                     d_from_idx, d_dist);
}

void launch_bow_transform(const orbgpu_ctx *c, const orbgpu_vocab *v, const uint8_t *d_desc, int n, int levelsup,
                          unsigned *d_word, unsigned *d_node) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_desc, n, v->d_child_off,
                     v->d_children, v->d_node_desc, v->d_word_id, v->levels - levelsup, v->n_nodes, d_word, d_node);
}

void launch_distinctive(const orbgpu_ctx *c, const uint8_t *d_desc, const int *d_offsets, int n_points, int *d_best) {
  if (n_points == 0) return;
  hipLaunchKernelGGL(k_distinctive, dim3((n_points + 3) / 4), dim3(256), 0, c->stream, d_desc, d_offsets, n_points, d_best);
}

size_t allpairs_lds_bytes(int capacity) { return (size_t)capacity * 16; }

void launch_match_allpairs(const orbgpu_ctx *c, const void *d_ang_rows, const uint8_t *d_desc_rows, const int *d_n_rows,
                           int n_rows, const void *d_ang_cols, int ang_stride, const uint8_t *d_desc_cols,
                           const int *d_n_cols, int n_cols, int capacity, int row_offset, unsigned threshold, double ratio,
                           int check_orientation, int *d_n_matches, int *d_match) {
  const unsigned th = std::min(threshold, 256u);
  dim3 grid((unsigned)n_cols * (unsigned)xcd_frames(n_rows));
  if (c->bruteforce_mfma)
    hipLaunchKernelGGL(k_allpairs_mfma, grid, dim3(256), allpairs_lds_bytes(capacity), c->stream,
                       reinterpret_cast<const char *>(d_ang_rows), d_desc_rows, d_n_rows, n_rows,
                       reinterpret_cast<const char *>(d_ang_cols), ang_stride, d_desc_cols, d_n_cols, n_cols, capacity,
                       row_offset, th, ratio, check_orientation, d_n_matches, d_match);
  else
    hipLaunchKernelGGL(k_allpairs, grid, dim3(256), allpairs_lds_bytes(capacity), c->stream,
                       reinterpret_cast<const char *>(d_ang_rows), d_desc_rows, d_n_rows, n_rows,
                       reinterpret_cast<const char *>(d_ang_cols), ang_stride, d_desc_cols, d_n_cols, n_cols, capacity,
                       row_offset, th, ratio, check_orientation, d_n_matches, d_match);
}

void launch_rotation_filter(const orbgpu_ctx *c, const float *d_ang_to, int n_to, const float *d_ang_from, int *d_match,
                            int *d_n_discarded) {
  hipLaunchKernelGGL(k_rotation_filter, dim3(1), dim3(256), 0, c->stream, d_ang_to, n_to, d_ang_from, d_match,
                     d_n_discarded);
}

double run_popcount_peak(const orbgpu_ctx *c) {
  unsigned *sink = reinterpret_cast<unsigned *>(c->d_match_scratch);
  const int iters = 1 << 15;
  const int blocks = 256 * 8;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_popcount_peak, dim3(blocks), dim3(256), 0, c->stream, sink, 1024);  // warm-up
  (void)hipEventRecord(e0, c->stream);
  hipLaunchKernelGGL(k_popcount_peak, dim3(blocks), dim3(256), 0, c->stream, sink, iters);
  (void)hipEventRecord(e1, c->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)blocks * 256.0 * (double)iters / ((double)ms * 1e-3);
}
