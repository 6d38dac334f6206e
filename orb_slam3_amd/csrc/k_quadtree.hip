// Quadtree keypoint distribution on the GPU: one workgroup per (frame, level).  gfx950 only.
//
// Reference stage: ORBFeatureExtractor::DistributeOctTree src/features/orb_feature_extractor.cpp:300-502 and
// ExtractorNode::DivideNode :602-652.  Flat-array formulation: the std::list is an array in list order; every sweep
// rebuilds it as
//     reverse(children in creation order) ++ (surviving old nodes in order),
// keys of a node are a contiguous segment of a key buffer and a division is a stable 4-way partition of that
// segment into the OTHER key buffer (segments of live nodes are disjoint, so nothing live is overwritten).
//   phase 1 (:356-423)  every multi-key node is divided per sweep;
//   phase 2 (:424-480)  the multi-key children of the last sweep are divided largest first
//                       (std::sort on (size, node*) -- ties: later-created first) until the list
//                       holds nFeatures nodes: ranks by counting, cut by a prefix sum over the processing order;
//   finally (:486-501)  the first maximum-response key of each node is emitted in list order.
// Parallel decomposition: nodes with many keys are partitioned by a whole wave (ballot prefix compaction),
// small nodes by single lanes; list positions come from workgroup-wide prefix sums.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "orbgpu_internal.h"

namespace {

// workgroup size NT is a template parameter: 256 threads for VGA-class levels (a few thousand keys, short dependent
// rounds), 1024 for large images, where a round moves tens of thousands of keys (launch_quadtree chooses)
constexpr int QT_LDS_KEYS = 2048;   // keys per ping-pong buffer kept in LDS; larger levels use global scratch
constexpr int QT_SMALL = 64;        // nodes up to this many keys are partitioned by a group of 16 lanes (4 nodes per wave)

struct QNode {
  unsigned short x0, x1, y0, y1;  // UL.x, UR.x, UL.y, BR.y
  unsigned beg;                   // first key of the segment
  unsigned cnt;                   // bit 31: key buffer (0 = A, 1 = B); bits 0..30: number of keys
};
static_assert(sizeof(QNode) == 16, "QNode layout");

typedef unsigned long long u64;

__device__ __forceinline__ int key_x(unsigned k) { return (int)(k & 0xfffu); }
__device__ __forceinline__ int key_y(unsigned k) { return (int)((k >> 12) & 0xfffu); }
__device__ __forceinline__ int key_resp(unsigned k) { return (int)(k >> 24); }

// key buffers carry their address space in the type: LDS (3) or global (1), so that the accesses are ds_ / global_
// instructions and not flat ones
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(1))) unsigned glb_u32;

struct QShared {
  QNode *cur, *nxt;
  uint4 *qc;            // quadrant counts of list entry i
  u64 *pre;             // packed prefix sums per entry (3 x 21 bit)
  u64 *pre2;            // survivor positions in phase 2
  unsigned short *vs, *vs_new, *ord;
  unsigned char *flag;
};

// exclusive prefix sum over a[0..n) in place (LDS), all NT threads; returns the total
template <int NT>
__device__ u64 block_excl_scan(u64 *a, int n, u64 *wsum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u64 carry = 0;
  for (int base = 0; base < n; base += NT) {
    const int i = base + threadIdx.x;
    const u64 v = i < n ? a[i] : 0;
    u64 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const u64 o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u64 woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < (NT / 64); w++) {
      const u64 s = wsum[w];
      if (w < wave) woff += s;
      tot += s;
    }
    if (i < n) a[i] = carry + woff + incl - v;
    carry += tot;
    __syncthreads();
  }
  return carry;
}

__device__ __forceinline__ int quadrant(unsigned key, int mx, int my) {
  return (key_x(key) < mx ? 0 : 1) + (key_y(key) < my ? 0 : 2);  // n1 n2 / n3 n4, :637-645
}

// stable 4-way partition of src[beg, beg+cnt) into dst[beg, ...) by a whole wave, cnt <= QT_BIG: every lane loads its
// (up to QT_BIG / 64) keys once, all loads in flight together, and both passes run from registers
template <class KT>
__device__ __forceinline__ uint4 partition_wave(const KT *__restrict__ src, KT *__restrict__ dst, int beg, int cnt, int mx, int my) {
  constexpr int U = 8;  // QT_BIG / 64
  const int lane = threadIdx.x & 63;
  const u64 below = (1ull << lane) - 1ull;
  unsigned key[U];
  int q[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int k = 64 * u + lane;
    key[u] = k < cnt ? src[beg + k] : 0u;
  }
  int c[4] = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; u++) {
    q[u] = 64 * u + lane < cnt ? quadrant(key[u], mx, my) : -1;
    if (64 * u < cnt) {  // uniform
#pragma unroll
      for (int qq = 0; qq < 4; qq++) c[qq] += __popcll(__ballot(q[u] == qq));
    }
  }
  int o[4];
  o[0] = beg;
  o[1] = o[0] + c[0];
  o[2] = o[1] + c[1];
  o[3] = o[2] + c[2];
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (64 * u < cnt) {
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        const u64 bb = __ballot(q[u] == qq);
        if (q[u] == qq) dst[o[qq] + __popcll(bb & below)] = key[u];
        o[qq] += __popcll(bb);
      }
    }
  }
  return make_uint4((unsigned)c[0], (unsigned)c[1], (unsigned)c[2], (unsigned)c[3]);
}

__device__ __forceinline__ void node_mid(const QNode &P, int &mx, int &my) {
  mx = P.x0 + ((P.x1 - P.x0 + 1) >> 1);  // ceil(float(w)/2), :606
  my = P.y0 + ((P.y1 - P.y0 + 1) >> 1);  // :607
}

// stable partition into at most 4 classes of a very large key range by the whole workgroup: every wave takes a
// contiguous quarter of the keys, counts its classes, and scatters behind the waves before it.  Keys are fetched four
// wave-steps at a time so that (global) loads overlap.  classify(key) returns 0..3.  All threads call it together.
constexpr int QT_BIG = 512;
template <int NT, class KT, class F>
__device__ __forceinline__ uint4 partition_block(const KT *__restrict__ src, KT *__restrict__ dst, int beg, int cnt, F classify) {
  __shared__ int wc[(NT / 64)][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u64 below = (1ull << lane) - 1ull;
  const int chunk = (((cnt + (NT / 64) - 1) / (NT / 64)) + 63) & ~63;
  const int b0 = min(wave * chunk, cnt), b1 = min(b0 + chunk, cnt);
  int c[4] = {0, 0, 0, 0};
  for (int k0 = b0; k0 < b1; k0 += 256) {
    unsigned kv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + 64 * u + lane;
      kv[u] = k < b1 ? src[beg + k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = k0 + 64 * u + lane < b1 ? classify(kv[u]) : -1;
#pragma unroll
      for (int qq = 0; qq < 4; qq++) c[qq] += __popcll(__ballot(q == qq));
    }
  }
  if (lane < 4) wc[wave][lane] = lane == 0 ? c[0] : lane == 1 ? c[1] : lane == 2 ? c[2] : c[3];
  __syncthreads();
  int tot[4], o[4];
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    tot[qq] = 0;
    o[qq] = 0;
    for (int w = 0; w < (NT / 64); w++) {
      if (w < wave) o[qq] += wc[w][qq];
      tot[qq] += wc[w][qq];
    }
  }
  o[0] += beg;
  o[1] += beg + tot[0];
  o[2] += beg + tot[0] + tot[1];
  o[3] += beg + tot[0] + tot[1] + tot[2];
  for (int k0 = b0; k0 < b1; k0 += 256) {
    unsigned kv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + 64 * u + lane;
      kv[u] = k < b1 ? src[beg + k] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int q = k0 + 64 * u + lane < b1 ? classify(kv[u]) : -1;
#pragma unroll
      for (int qq = 0; qq < 4; qq++) {
        const u64 bb = __ballot(q == qq);
        if (q == qq) dst[o[qq] + __popcll(bb & below)] = kv[u];
        o[qq] += __popcll(bb);
      }
    }
  }
  __syncthreads();  // wc is reused by the next call
  return make_uint4((unsigned)tot[0], (unsigned)tot[1], (unsigned)tot[2], (unsigned)tot[3]);
}


// divide the keys of the listed nodes (list positions idx[k], or k itself when idx == nullptr)
template <int NT, class KT>
// max_cnt: an upper bound of the listed nodes' key counts (lets whole size classes be skipped without scanning the list)
__device__ __forceinline__ void partition_nodes(const QShared &S, KT *keys0, KT *keys1, const unsigned short *idx, int count,
                                                int max_cnt) {
  const int wave = threadIdx.x >> 6;
  // very large nodes (the first rounds of a big level): the whole workgroup, one node after the other
  for (int k = 0; k < count && max_cnt > QT_BIG; k++) {
    const int i = idx ? idx[k] : k;
    const QNode P = S.cur[i];
    const int cnt = (int)(P.cnt & 0x7fffffffu);
    if (cnt <= QT_BIG) continue;  // uniform: every thread reads the same node
    const int b = (int)(P.cnt >> 31);
    int mx, my;
    node_mid(P, mx, my);
    const uint4 c = partition_block<NT>(b ? keys1 : keys0, b ? keys0 : keys1, (int)P.beg, cnt,
                                    [mx, my](unsigned key) { return quadrant(key, mx, my); });
    if (threadIdx.x == 0) S.qc[i] = c;
  }
  // large nodes: one wave each
  for (int k = wave; k < count && max_cnt > QT_SMALL; k += (NT / 64)) {
    const int i = idx ? idx[k] : k;
    const QNode P = S.cur[i];
    const int cnt = (int)(P.cnt & 0x7fffffffu);
    if (cnt <= QT_SMALL || cnt > QT_BIG) continue;
    const int b = (int)(P.cnt >> 31);
    int mx, my;
    node_mid(P, mx, my);
    const uint4 c = partition_wave<KT>(b ? keys1 : keys0, b ? keys0 : keys1, (int)P.beg, cnt, mx, my);
    if ((threadIdx.x & 63) == 0) S.qc[i] = c;
  }
  // small nodes (2 .. 64 keys): four at a time per wave, 16 lanes each.  A lane loads its (up to four) keys once --
  // all loads of a node quad are in flight together, which is what matters when the keys sit in global memory -- and
  // both passes (count, stable scatter) run from registers with wave ballots cut into 16-bit group fields.
  {
    const int lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
    const unsigned gbelow = (1u << gl) - 1u;
    for (int base = wave * 4; base < count; base += (NT / 64) * 4) {  // uniform per wave
      const int k = base + grp;
      int i = 0, cnt = 0, beg = 0, mx = 0, my = 0;
      const KT *src = keys0;
      KT *dst = keys1;
      if (k < count) {
        i = idx ? idx[k] : k;
        const QNode P = S.cur[i];
        cnt = (int)(P.cnt & 0x7fffffffu);
        beg = (int)P.beg;
        node_mid(P, mx, my);
        if (P.cnt >> 31) { src = keys1; dst = keys0; }
      }
      const bool mine = k < count && cnt <= QT_SMALL;   // larger nodes were handled above
      const int work = (mine && cnt > 1) ? cnt : 0;
      unsigned key[QT_SMALL / 16];
      int q[QT_SMALL / 16];
#pragma unroll
      for (int u = 0; u < QT_SMALL / 16; u++) {
        const int kk = 16 * u + gl;
        key[u] = kk < work ? src[beg + kk] : 0u;
      }
      int c[4] = {0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < QT_SMALL / 16; u++) {
        q[u] = 16 * u + gl < work ? quadrant(key[u], mx, my) : -1;
#pragma unroll
        for (int qq = 0; qq < 4; qq++) c[qq] += __popc((unsigned)(__ballot(q[u] == qq) >> (16 * grp)) & 0xffffu);
      }
      int o[4];
      o[0] = beg;
      o[1] = o[0] + c[0];
      o[2] = o[1] + c[1];
      o[3] = o[2] + c[2];
#pragma unroll
      for (int u = 0; u < QT_SMALL / 16; u++) {
#pragma unroll
        for (int qq = 0; qq < 4; qq++) {
          const unsigned gb = (unsigned)(__ballot(q[u] == qq) >> (16 * grp)) & 0xffffu;
          if (q[u] == qq) dst[o[qq] + __popc(gb & gbelow)] = key[u];
          o[qq] += __popc(gb);
        }
      }
      if (mine && gl == 0) S.qc[i] = make_uint4((unsigned)c[0], (unsigned)c[1], (unsigned)c[2], (unsigned)c[3]);
    }
  }
}

__device__ __forceinline__ u64 pack3(unsigned a, unsigned b, unsigned c) { return (u64)a | ((u64)b << 21) | ((u64)c << 42); }
__device__ __forceinline__ unsigned f0(u64 v) { return (unsigned)(v & 0x1fffffu); }
__device__ __forceinline__ unsigned f1(u64 v) { return (unsigned)((v >> 21) & 0x1fffffu); }
__device__ __forceinline__ unsigned f2(u64 v) { return (unsigned)((v >> 42) & 0x1fffffu); }

// write the children of parent P (quadrant counts c) into the new list: creation index `cidx` onwards, positions
// counted from the back of the `ct` children (push_front order); multi-key children are appended to vs_new at `midx`
// flip: 1 when the division moved the keys into the other key buffer, 0 for the presorted rounds (keys stay where they are)
__device__ void emit_children(const QShared &S, const QNode &P, const uint4 c, int cidx, int ct, int midx, int *max_child,
                              unsigned flip) {
  int mx, my;
  node_mid(P, mx, my);
  const unsigned nb = ((P.cnt >> 31) ^ flip) << 31;
  const unsigned cq[4] = {c.x, c.y, c.z, c.w};
  const unsigned short bx0[4] = {P.x0, (unsigned short)mx, P.x0, (unsigned short)mx};
  const unsigned short bx1[4] = {(unsigned short)mx, P.x1, (unsigned short)mx, P.x1};
  const unsigned short by0[4] = {P.y0, P.y0, (unsigned short)my, (unsigned short)my};
  const unsigned short by1[4] = {(unsigned short)my, (unsigned short)my, P.y1, P.y1};
  if (max_child) atomicMax(max_child, (int)max(max(c.x, c.y), max(c.z, c.w)));
  unsigned beg = P.beg;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    if (cq[q] != 0) {
      const int pos = ct - 1 - cidx;
      QNode n;
      n.x0 = bx0[q]; n.x1 = bx1[q]; n.y0 = by0[q]; n.y1 = by1[q];
      n.beg = beg;
      n.cnt = cq[q] | nb;
      S.nxt[pos] = n;
      if (cq[q] > 1) S.vs_new[midx++] = (unsigned short)pos;
      cidx++;
    }
    beg += cq[q];
  }
}

__device__ __forceinline__ unsigned n_children(const uint4 c) { return (c.x != 0) + (c.y != 0) + (c.z != 0) + (c.w != 0); }
__device__ __forceinline__ unsigned n_multi(const uint4 c) { return (c.x > 1) + (c.y > 1) + (c.z > 1) + (c.w > 1); }

struct QtArgs {
  const CellEntry *cells;      // [frame][cells_per_frame]
  const unsigned *cands;       // [frame][cands_per_frame] candidate sub-pools
  unsigned *buf_a, *buf_b;     // [frame][cands_per_frame] ordered key buffers (global fallback / gather target)
  int *cell_prefix;            // [frame][cells_per_frame]
  short2 *kp_xy;               // [frame][kps_per_frame]
  int *kp_count;               // [frame][levels]
  int *error_flag;
  unsigned char *node_scratch; // node arrays of the workgroups when they do not fit LDS (NODES_GLOBAL), node_bytes each
  size_t node_bytes;
  unsigned long long *prof;    // ORBGPU_QT_PROFILE=1: wall-clock stamps of workgroup (level 0, frame 0), debugging aid
  int node_cap;                // LDS entries per list array
  int budget[ORBGPU_MAX_LEVELS];
  int nini[ORBGPU_MAX_LEVELS];
  unsigned *ps;                // presort tables, PS_STRIDE words per (frame, presorted level)
  int ps_levels;               // levels 0 .. ps_levels - 1 have tables
  int ps_depth[ORBGPU_MAX_LEVELS];  // rounds of phase 1 the presort covers (0: that level is not presorted)
};

// ---- presort: the first rounds of phase 1 by several workgroups --------------------------------------------------
// While nIni * 4^R <= nFeatures the list cannot reach the budget, so rounds 1..R of phase 1 divide EVERY multi-key node
// (:356-423; after round k the list holds at most nIni * 4^k nodes, and phase 2 starts only when size + 3 * multi >
// nFeatures).  R rounds of stable 4-way partitions order the keys by (root, quadrant path of depth R), stable inside a
// path -- one stable counting sort by that code.  Several workgroups do it ahead of k_quadtree: a wave owns a
// contiguous run of cells (candidate order = cells row-major, raster inside a cell), k_qt_presort<false> counts the codes of
// every run, k_qt_presort_scan turns the counts into write positions and the per-code prefix `hpre`, k_qt_presort<true>
// scatters the keys into key buffer B.  k_quadtree then runs its rounds 1..R without touching a key: the quadrant
// counts of a node are differences of hpre over the code range of each child.  Keys are never ordered deeper than the
// divisions that really happen, so the first-maximum rule inside a final node (:486-501) sees the reference's order.
constexpr int PS_CHUNKS = 64;     // cell runs (one wave each) per (frame, level)
constexpr int PS_NB_MAX = 1024;   // codes: nIni * 4^R
constexpr int PS_STRIDE = PS_CHUNKS * PS_NB_MAX + PS_NB_MAX + 64;  // [chunk][code] counts / positions, then hpre[code]

__device__ __forceinline__ int ps_code(unsigned key, float hX, int nIni, int H, int R) {
  const int x = key_x(key), y = key_y(key);
  const int b = nIni == 1 ? 0 : min(max((int)__fdiv_rn((float)x, hX), 0), nIni - 1);  // root strip, :329-333
  int x0 = (int)__fmul_rn(hX, (float)b), x1 = (int)__fmul_rn(hX, (float)(b + 1)), y0 = 0, y1 = H;
  int code = b;
  for (int d = 0; d < R; d++) {
    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);  // node_mid
    const int qx = x < mx ? 0 : 1, qy = y < my ? 0 : 1;
    code = (code << 2) | qx | (qy << 1);
    if (qx) x0 = mx; else x1 = mx;
    if (qy) y0 = my; else y1 = my;
  }
  return code;
}

struct PsArgs {
  const CellEntry *cells;
  const unsigned *cands;
  unsigned *buf_b;
  unsigned *ps;
  int ps_levels;
  int depth[ORBGPU_MAX_LEVELS];
  int nini[ORBGPU_MAX_LEVELS];
};

template <bool SCATTER>
__global__ __launch_bounds__(256) void k_qt_presort(DeviceGeom g, PsArgs a) {
  __shared__ unsigned tab_all[4][PS_NB_MAX];
  const int f = blockIdx.y, l = blockIdx.z;
  const int R = a.depth[l];
  if (R == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, chunk = blockIdx.x * 4 + wave;
  const LevelGeom L = g.lv[l];
  const int nIni = a.nini[l], NB = nIni << (2 * R), nbits = 2 * R + 2;
  const int W = L.w - 2 * ORB_MIN_BORDER, H = L.h - 2 * ORB_MIN_BORDER;
  const float hX = __fdiv_rn((float)W, (float)nIni);
  const int n_cells = L.ncols * L.nrows;
  const int cpc = (n_cells + PS_CHUNKS - 1) / PS_CHUNKS;
  const int c_beg = min(chunk * cpc, n_cells), c_end = min(c_beg + cpc, n_cells);
  const CellEntry *cells = a.cells + (size_t)f * g.cells_per_frame + L.cell_off;
  const size_t goff = (size_t)f * g.cands_per_frame + L.cand_off;
  const unsigned *pool = a.cands + goff;
  unsigned *dst = a.buf_b + goff;
  unsigned *ps = a.ps + ((size_t)f * a.ps_levels + l) * PS_STRIDE;
  unsigned *hist = ps + (size_t)chunk * PS_NB_MAX;
  const unsigned *hpre = ps + (size_t)PS_CHUNKS * PS_NB_MAX;
  volatile unsigned *tab = tab_all[wave];
  for (int b = lane; b < NB; b += 64) tab[b] = SCATTER ? hist[b] + hpre[b] : 0u;
  const u64 below = (1ull << lane) - 1ull;
  const int slot = ((L.wcell + 1) >> 1) * ((L.hcell + 1) >> 1);

  // 64 keys in candidate order (lane order), `have` marks the real ones
  auto process = [&](unsigned key, bool have) {
    const int code = have ? ps_code(key, hX, nIni, H, R) : 0;
    if (!SCATTER) {
      if (have) atomicAdd(&tab_all[wave][code], 1u);
    } else {
      u64 m = __ballot(have);
      for (int bit = 0; bit < nbits; bit++) {
        const bool one = (code >> bit) & 1;
        const u64 bb = __ballot(one);
        m &= one ? bb : ~bb;
      }
      if (have) {
        const unsigned base = tab[code];
        dst[base + (unsigned)__popcll(m & below)] = key;
        if ((m >> lane) == 1ull) tab[code] = base + (unsigned)__popcll(m);  // the last lane of the code advances it
      }
    }
  };

  // 64 cells per step: lane = cell, a wave scan of the counts gives every cell's first position among the step's T
  // candidates; then 64 candidates at a time, lane k finding its cell by bisection over the scanned counts (6 wave
  // shuffles) -- every lane works whatever the cells hold, and the candidate loads of two steps are in flight together
  for (int cb = c_beg; cb < c_end; cb += 64) {
    const int c = cb + lane;
    const int cnt = c < c_end ? cells[c].count : 0;
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    const int T = __shfl(incl, 63);
    for (int k0 = 0; k0 < T; k0 += 128) {
      unsigned key[2];
      bool have[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int k = k0 + 64 * u + lane;
        have[u] = k < T;
        int j = 0;  // number of cells that end at or before k = the cell of candidate k
#pragma unroll
        for (int st = 32; st >= 1; st >>= 1)
          if (__shfl(incl, j + st - 1) <= k) j += st;
        j = min(j, 63);
        const int first = __shfl(incl, j) - __shfl(cnt, j);
        key[u] = have[u] ? pool[(size_t)(cb + j) * slot + (k - first)] : 0u;
      }
      process(key[0], have[0]);
      if (k0 + 64 < T) process(key[1], have[1]);
    }
  }
  if (!SCATTER)
    for (int b = lane; b < NB; b += 64) hist[b] = tab[b];
}

// 256 threads (a 1024-thread workgroup waits for a whole CU's wave slots while the blur runs beside it)
__global__ __launch_bounds__(256) void k_qt_presort_scan(PsArgs a) {
  __shared__ u64 tot[PS_NB_MAX];
  __shared__ u64 wsum[4];
  const int f = blockIdx.x, l = blockIdx.y;
  const int R = a.depth[l];
  if (R == 0) return;
  const int NB = a.nini[l] << (2 * R);
  unsigned *ps = a.ps + ((size_t)f * a.ps_levels + l) * PS_STRIDE;
  unsigned *hpre = ps + (size_t)PS_CHUNKS * PS_NB_MAX;
  for (int b = threadIdx.x; b < PS_NB_MAX; b += 256) {
    unsigned run = 0;
    if (b < NB) {
      unsigned t[PS_CHUNKS];
#pragma unroll
      for (int c = 0; c < PS_CHUNKS; c++) t[c] = ps[(size_t)c * PS_NB_MAX + b];
#pragma unroll
      for (int c = 0; c < PS_CHUNKS; c++) {
        ps[(size_t)c * PS_NB_MAX + b] = run;
        run += t[c];
      }
    }
    tot[b] = run;
  }
  __syncthreads();
  const u64 total = block_excl_scan<256>(tot, PS_NB_MAX, wsum);
  for (int b = threadIdx.x; b < NB; b += 256) hpre[b] = (unsigned)tot[b];
  if (threadIdx.x == 0) hpre[NB] = (unsigned)total;
}

// quadrant counts of the listed nodes (all of depth `rnd`) from the code prefix: no key moves
__device__ __forceinline__ void presorted_counts(const QShared &S, const glb_u32 *keys, int Sz, int rnd, int R,
                                                 const unsigned *hpre, float hX, int nIni, int H) {
  const int sh = 2 * (R - rnd), s2 = sh - 2;
  for (int i = threadIdx.x; i < Sz; i += blockDim.x) {
    const QNode P = S.cur[i];
    if ((P.cnt & 0x7fffffffu) <= 1) continue;
    const int p = ps_code(keys[P.beg], hX, nIni, H, R) >> sh;
    unsigned c[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int lo = ((p << 2) | q) << s2;
      c[q] = hpre[lo + (1 << s2)] - hpre[lo];
    }
    S.qc[i] = make_uint4(c[0], c[1], c[2], c[3]);
  }
}

// Everything after the candidate count is known.  IN_LDS decides at compile time where the two key buffers live, so
// that the partition loops use LDS instructions: with a run-time choice the pointers are generic and every key access
// becomes a flat load / store (measured: 0.5-1 microsecond per dependent step).
#define STAMP(i) do { if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.prof[i] = wall_clock64(); } while (0)
template <int NT, bool IN_LDS>
__device__ __forceinline__ void qt_run(const DeviceGeom &g, const QtArgs &a, const LevelGeom &L, const int l, const int f,
                                       const int n, QShared S, unsigned *lds_keys, u64 *wsum, int *s_misc,
                                       const CellEntry *cells, const int *cpre, const int R = 0,
                                       const unsigned *hpre = nullptr) {
  const int cap = a.node_cap;
  const int nF = a.budget[l];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_cells = L.ncols * L.nrows;
  const size_t goff = (size_t)f * g.cands_per_frame + L.cand_off;
  using KT = typename std::conditional<IN_LDS, lds_u32, glb_u32>::type;
  KT *keys0 = (KT *)(IN_LDS ? lds_keys : a.buf_a + goff);
  KT *keys1 = (KT *)(IN_LDS ? lds_keys + QT_LDS_KEYS : a.buf_b + goff);
  if (R == 0) {  // (a presorted level has its keys in buffer B already)
    const unsigned *pool = a.cands + goff;
    KT *dst = keys0;
    // 16 lanes per cell, 16 cells per step, four steps in flight.  A cell's candidates sit in its fixed slot of the
    // pool (cell index x slot size, k_cell_detect), so the candidate loads do not wait for the cell table
    const int sub = tid & 15;
    const int slot = ((L.wcell + 1) >> 1) * ((L.hcell + 1) >> 1);
    constexpr int CS = NT / 16;
    for (int c0 = tid >> 4; c0 < n_cells; c0 += 4 * CS) {
      int cnt[4], o[4];
      unsigned v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u * CS;
        const bool ok = c < n_cells;
        cnt[u] = ok ? cells[c].count : 0;
        o[u] = ok ? cpre[c] : 0;
        v[u] = ok ? pool[(size_t)c * slot + sub] : 0u;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (sub < cnt[u]) dst[o[u] + sub] = v[u];
        for (int k = 16 + sub; k < cnt[u]; k += 16) dst[o[u] + k] = pool[(size_t)(c0 + u * CS) * slot + k];  // rare
      }
    }
  }
  __syncthreads();

  STAMP(2);
  // ---- root nodes (:310-347): nIni vertical strips, keys bucketed by int(x / hX), stable ----
  const int W = L.w - 2 * ORB_MIN_BORDER, H = L.h - 2 * ORB_MIN_BORDER;
  const int nIni = a.nini[l];
  const float hX = __fdiv_rn((float)W, (float)nIni);
  int Sz = 0;  // list size
  if (!IN_LDS && R > 0) {
    // presorted: the roots' keys are the code ranges [b * 4^R, (b + 1) * 4^R) of buffer B
    if (tid == 0) {
      int nroots = 0;
      unsigned o = 0;
      for (int b = 0; b < nIni; b++) {
        const unsigned cb = hpre[(b + 1) << (2 * R)] - hpre[b << (2 * R)];
        if (cb) {  // empty roots are erased (:343-344)
          QNode r;
          r.x0 = (unsigned short)(int)__fmul_rn(hX, (float)b);
          r.x1 = (unsigned short)(int)__fmul_rn(hX, (float)(b + 1));
          r.y0 = 0; r.y1 = (unsigned short)H;
          r.beg = o; r.cnt = cb | 0x80000000u;
          S.cur[nroots++] = r;
        }
        o += cb;
      }
      s_misc[0] = nroots;
    }
    __syncthreads();
    Sz = s_misc[0];
  } else if (nIni == 1) {
    if (tid == 0 && n > 0) {
      QNode r;
      r.x0 = 0; r.x1 = (unsigned short)(int)__fmul_rn(hX, 1.0f); r.y0 = 0; r.y1 = (unsigned short)H;
      r.beg = 0; r.cnt = (unsigned)n;
      S.cur[0] = r;
    }
    Sz = n > 0 ? 1 : 0;
  } else {
    if (nIni <= 4) {
      // the whole workgroup buckets the keys into buffer B (stable), then the roots live in B
      const uint4 bc = partition_block<NT>(keys0, keys1, 0, n, [hX, nIni](unsigned key) {
        return min(max((int)__fdiv_rn((float)key_x(key), hX), 0), nIni - 1);
      });
      if (tid == 0) {
        const unsigned cnts[4] = {bc.x, bc.y, bc.z, bc.w};
        int nroots = 0;
        unsigned o = 0;
        for (int b = 0; b < nIni; b++) {
          if (cnts[b]) {  // empty roots are erased (:343-344)
            QNode r;
            r.x0 = (unsigned short)(int)__fmul_rn(hX, (float)b);
            r.x1 = (unsigned short)(int)__fmul_rn(hX, (float)(b + 1));
            r.y0 = 0; r.y1 = (unsigned short)H;
            r.beg = o; r.cnt = cnts[b] | 0x80000000u;
            S.cur[nroots++] = r;
          }
          o += cnts[b];
        }
        s_misc[0] = nroots;
      }
    } else if (wave == 0) {  // more than 4 roots (aspect ratio above 4.5): wave 0, bucket by bucket

      const u64 below = (1ull << lane) - 1ull;
      int o = 0, nroots = 0;
      for (int b = 0; b < nIni; b++) {
        const int beg = o;
        for (int k0 = 0; k0 < n; k0 += 64) {
          const int k = k0 + lane;
          const bool ok = k < n;
          const unsigned key = ok ? keys0[k] : 0u;
          int bk = ok ? (int)__fdiv_rn((float)key_x(key), hX) : -1;
          if (ok) bk = min(max(bk, 0), nIni - 1);
          const u64 m = __ballot(bk == b);
          if (bk == b) keys1[o + __popcll(m & below)] = key;
          o += __popcll(m);
        }
        if (o > beg) {  // empty roots are erased (:343-344)
          if (lane == 0) {
            QNode r;
            r.x0 = (unsigned short)(int)__fmul_rn(hX, (float)b);
            r.x1 = (unsigned short)(int)__fmul_rn(hX, (float)(b + 1));
            r.y0 = 0; r.y1 = (unsigned short)H;
            r.beg = (unsigned)beg; r.cnt = (unsigned)(o - beg) | 0x80000000u;
            S.cur[nroots] = r;
          }
          nroots++;
        }
      }
      if (lane == 0) s_misc[0] = nroots;
    }
    __syncthreads();
    Sz = s_misc[0];
  }
  __syncthreads();

  STAMP(3);
  // ---- phase 1 ----------------------------------------------------------------------------------------
  bool finish = false;
  bool phase2 = false;
  int n_vs = 0;
  int max_cnt = n;  // upper bound of the key count of any node of the list
  int rnd = 0;      // rounds done: the multi-key nodes of the list all have this depth
  while (!finish && !phase2) {
    if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.prof[15]++;
    const int prevSize = Sz;
    const int rr_ = a.prof ? min((int)a.prof[15] - 1, 3) : 0;
    if (a.prof) __syncthreads();
    STAMP(16 + 4 * rr_);
    const bool presorted = !IN_LDS && rnd < R;
    if constexpr (!IN_LDS) {
      if (presorted) presorted_counts(S, keys1, Sz, rnd, R, hpre, hX, nIni, H);
      else partition_nodes<NT, KT>(S, keys0, keys1, nullptr, Sz, max_cnt);
    } else {
      partition_nodes<NT, KT>(S, keys0, keys1, nullptr, Sz, max_cnt);
    }
    rnd++;
    __syncthreads();
    STAMP(17 + 4 * rr_);
    for (int i = tid; i < Sz; i += NT) {
      const unsigned cnt = S.cur[i].cnt & 0x7fffffffu;
      const uint4 c = S.qc[i];
      S.pre[i] = cnt > 1 ? pack3(n_children(c), 0, n_multi(c)) : pack3(0, 1, 0);
    }
    if (tid == 0) s_misc[2] = 1;  // max child count of this round (the scan's barriers order it before the emit)
    __syncthreads();
    const u64 tot = block_excl_scan<NT>(S.pre, Sz, wsum);
    const int ct = (int)f0(tot), kt = (int)f1(tot), mt = (int)f2(tot);
    if (ct + kt > cap) {  // cannot happen: the list never exceeds max(nF + 3, 4 nIni)
      if (tid == 0) *(volatile int *)a.error_flag = 1;
      Sz = 0;
      break;
    }
    for (int i = tid; i < Sz; i += NT) {
      const QNode P = S.cur[i];
      const u64 pr = S.pre[i];
      if ((P.cnt & 0x7fffffffu) > 1) emit_children(S, P, S.qc[i], (int)f0(pr), ct, (int)f2(pr), &s_misc[2], presorted ? 0u : 1u);
      else S.nxt[ct + (int)f1(pr)] = P;
    }
    __syncthreads();
    { QNode *t = S.cur; S.cur = S.nxt; S.nxt = t; }
    { unsigned short *t = S.vs; S.vs = S.vs_new; S.vs_new = t; }
    Sz = ct + kt;
    n_vs = mt;
    max_cnt = s_misc[2];
    if (Sz >= nF || Sz == prevSize) finish = true;           // :422
    else if (Sz + n_vs * 3 > nF) phase2 = true;                // :424
  }

  STAMP(4);
  if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.prof[10] = n; a.prof[11] = Sz; a.prof[12] = n_vs; }
  // ---- phase 2 ----------------------------------------------------------------------------------------
  while (!finish && phase2) {
    if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.prof[14]++;
    const int prevSize = Sz;
    const int m = n_vs;
    if (m == 0) break;  // nothing left to divide: size == prevSize (:477)
    if (a.prof) __syncthreads();
    STAMP(32);
    partition_nodes<NT, KT>(S, keys0, keys1, S.vs, m, max_cnt);  // nodes only shrink: the bound of phase 1 still holds
    if (a.prof) __syncthreads();
    STAMP(33);
    // processing order: descending (size, creation order); rank by counting
    // (sizes first gathered into a flat LDS array -- pre2 is free here -- so the O(m^2) count reads one broadcast
    // word per step instead of chasing vs -> node -> cnt)
    unsigned *sz = reinterpret_cast<unsigned *>(S.pre2);
    for (int k = tid; k < m; k += NT) sz[k] = S.cur[S.vs[k]].cnt & 0x7fffffffu;
    __syncthreads();
    for (int k = tid; k < m; k += NT) {
      const unsigned ck = sz[k];
      int rank = 0;
#pragma unroll 8
      for (int k2 = 0; k2 < m; k2++) {
        const unsigned c2 = sz[k2];
        rank += (c2 > ck) || (c2 == ck && k2 > k);
      }
      S.ord[rank] = (unsigned short)k;
    }
    for (int i = tid; i < Sz; i += NT) S.flag[i] = 0;
    __syncthreads();
    // inclusive growth of the list along the processing order -> cut (:474)
    for (int r = tid; r < m; r += NT) {
      const uint4 c = S.qc[S.vs[S.ord[r]]];
      S.pre[r] = pack3(n_children(c), 0, n_multi(c));
    }
    __syncthreads();
    block_excl_scan<NT>(S.pre, m, wsum);
    // first r with prevSize + (children through r) - (r + 1) >= nF
    if (tid == 0) s_misc[1] = m;
    __syncthreads();
    for (int r = tid; r < m; r += NT) {
      const uint4 c = S.qc[S.vs[S.ord[r]]];
      const int grown = prevSize + (int)f0(S.pre[r]) + (int)n_children(c) - (r + 1);
      if (grown >= nF) atomicMin(&s_misc[1], r + 1);
    }
    __syncthreads();
    const int nproc = s_misc[1];  // nodes actually divided this sweep
    // totals over the processed prefix
    int ct, mt;
    {
      const int rl = nproc - 1;
      const uint4 c = S.qc[S.vs[S.ord[rl]]];
      ct = (int)f0(S.pre[rl]) + (int)n_children(c);
      mt = (int)f2(S.pre[rl]) + (int)n_multi(c);
    }
    for (int r = tid; r < nproc; r += NT) S.flag[S.vs[S.ord[r]]] = 1;
    __syncthreads();
    // survivors of the old list keep their order behind the new children
    for (int i = tid; i < Sz; i += NT) S.pre2[i] = S.flag[i] ? 0ull : 1ull;
    __syncthreads();
    const int kt = (int)block_excl_scan<NT>(S.pre2, Sz, wsum);
    if (ct + kt > cap) {
      if (tid == 0) *(volatile int *)a.error_flag = 1;
      Sz = 0;
      break;
    }
    for (int r = tid; r < nproc; r += NT) {
      const int i = S.vs[S.ord[r]];
      emit_children(S, S.cur[i], S.qc[i], (int)f0(S.pre[r]), ct, (int)f2(S.pre[r]), nullptr, 1u);
    }
    for (int i = tid; i < Sz; i += NT)
      if (!S.flag[i]) S.nxt[ct + (int)S.pre2[i]] = S.cur[i];
    __syncthreads();
    { QNode *t = S.cur; S.cur = S.nxt; S.nxt = t; }
    { unsigned short *t = S.vs; S.vs = S.vs_new; S.vs_new = t; }
    Sz = ct + kt;
    n_vs = mt;
    if (Sz >= nF || Sz == prevSize) finish = true;  // :477
  }

  STAMP(5);
  // ---- best point of each node: first maximum, strict > (:486-501) ----
  short2 *out = a.kp_xy + (size_t)f * g.kps_per_frame + L.kp_off;
  if (Sz > L.kp_cap) {
    if (tid == 0) *(volatile int *)a.error_flag = 3;
    Sz = 0;
  }
  // A presorted level orders its keys by the depth-R quadrant code.  When phase 1 stops before round R (the list did not
  // grow: every multi-key node kept all its keys in one quadrant, :422) a node's keys are ordered deeper than the node
  // is, i.e. no longer in candidate order, so "first" among equal responses is then decided by the candidate order
  // itself: cells row-major, raster order inside a cell (:244-276; a cell's keypoints have x - 3 in
  // [j * wCell, (j + 1) * wCell), likewise y)
  const bool by_cand_order = !IN_LDS && R > 0 && rnd < R;
  auto cand_order = [&L](unsigned key) -> u64 {
    const int x = (int)key_x(key), y = (int)key_y(key);
    const int cx = max(x - 3, 0) / L.wcell, cy = max(y - 3, 0) / L.hcell;
    return ((u64)(unsigned)(cy * L.ncols + cx) << 24) | ((u64)(unsigned)y << 12) | (u64)(unsigned)x;
  };
  for (int i = tid; i < Sz; i += NT) {
    const QNode P = S.cur[i];
    const KT *keys = (P.cnt >> 31) ? keys1 : keys0;
    const int cnt = (int)(P.cnt & 0x7fffffffu);
    unsigned best = keys[P.beg];
    if (!by_cand_order) {
#pragma unroll 4
      for (int k = 1; k < cnt; k++) {
        const unsigned key = keys[P.beg + k];
        if (key_resp(key) > key_resp(best)) best = key;
      }
    } else {
      for (int k = 1; k < cnt; k++) {
        const unsigned key = keys[P.beg + k];
        if (key_resp(key) > key_resp(best) || (key_resp(key) == key_resp(best) && cand_order(key) < cand_order(best))) best = key;
      }
    }
    out[i] = make_short2((short)(key_x(best) + ORB_MIN_BORDER), (short)(key_y(best) + ORB_MIN_BORDER));
  }
  if (tid == 0) a.kp_count[f * g.levels + l] = Sz;
  STAMP(6);
  if (a.prof && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.prof[13] = Sz;
}

// NODES_GLOBAL: the per-level node list (cap entries of every array below) is too large for LDS -- a single level
// budgeted more than about 2060 keypoints -- and lives in a per-workgroup slice of global scratch instead; all of a
// workgroup's waves share one CU (one L1), and the barriers between the phases order the accesses
template <int NT, bool NODES_GLOBAL>
__global__ __launch_bounds__(NT) void k_quadtree(DeviceGeom g, QtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ u64 wsum[(NT / 64)];
  __shared__ int s_misc[8];
  STAMP(0);
  // workgroups are handed out in blockIdx.x-fastest order: every frame of level 0 first, then level 1, ... -- the
  // longest jobs (the levels with the most candidates) start first and the short ones fill in behind them
  // a chain of short dependent rounds in few waves, running beside the blur's many: let these waves issue first
  // (1080p: 0.49 -> 0.48 ms; what stretches this kernel beside the blur -- 0.16 ms alone for a level-0 workgroup of a
  // 1080p frame, 0.5 ms next to it -- is not issue arbitration)
  __builtin_amdgcn_s_setprio(3);
  const int f = blockIdx.x, l = blockIdx.y;
  const LevelGeom L = g.lv[l];
  const int cap = a.node_cap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // carve the dynamic LDS
  QShared S;
  unsigned char *p = smem;
  unsigned *lds_keys = reinterpret_cast<unsigned *>(p); p += sizeof(unsigned) * 2 * QT_LDS_KEYS;
  if (NODES_GLOBAL) p = a.node_scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.node_bytes;
  QNode *nodes0 = reinterpret_cast<QNode *>(p); p += sizeof(QNode) * cap;
  QNode *nodes1 = reinterpret_cast<QNode *>(p); p += sizeof(QNode) * cap;
  S.qc = reinterpret_cast<uint4 *>(p); p += sizeof(uint4) * cap;
  S.pre = reinterpret_cast<u64 *>(p); p += sizeof(u64) * cap;
  S.pre2 = reinterpret_cast<u64 *>(p); p += sizeof(u64) * cap;
  S.vs = reinterpret_cast<unsigned short *>(p); p += sizeof(unsigned short) * cap;
  S.vs_new = reinterpret_cast<unsigned short *>(p); p += sizeof(unsigned short) * cap;
  S.ord = reinterpret_cast<unsigned short *>(p); p += sizeof(unsigned short) * cap;
  S.flag = p;
  S.cur = nodes0;
  S.nxt = nodes1;

  // ---- gather the candidates in reference order: cells row-major, raster order inside a cell (:244-276) ----
  const int n_cells = L.ncols * L.nrows;
  const CellEntry *cells = a.cells + (size_t)f * g.cells_per_frame + L.cell_off;
  int *cpre = a.cell_prefix + (size_t)f * g.cells_per_frame + L.cell_off;
  int n = 0;
  const int R = (int)blockIdx.y < a.ps_levels ? a.ps_depth[l] : 0;
  if (R > 0) {
    const unsigned *hpre = a.ps + ((size_t)f * a.ps_levels + l) * PS_STRIDE + (size_t)PS_CHUNKS * PS_NB_MAX;
    n = (int)hpre[a.nini[l] << (2 * R)];
    STAMP(1);
    qt_run<NT, false>(g, a, L, l, f, n, S, lds_keys, wsum, s_misc, cells, cpre, R, hpre);
    return;
  }
  {
    int carry = 0;
    for (int base = 0; base < n_cells; base += NT) {
      const int i = base + tid;
      const int v = i < n_cells ? cells[i].count : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (lane == 63) wsum[wave] = (u64)incl;
      __syncthreads();
      int woff = 0, tot = 0;
#pragma unroll
      for (int w = 0; w < (NT / 64); w++) {
        const int s = (int)wsum[w];
        if (w < wave) woff += s;
        tot += s;
      }
      if (i < n_cells) cpre[i] = carry + woff + incl - v;
      carry += tot;
      __syncthreads();
    }
    n = carry;
  }
  STAMP(1);
  if (n <= QT_LDS_KEYS) qt_run<NT, true>(g, a, L, l, f, n, S, lds_keys, wsum, s_misc, cells, cpre);
  else qt_run<NT, false>(g, a, L, l, f, n, S, lds_keys, wsum, s_misc, cells, cpre);
}

}  // namespace

size_t quadtree_node_bytes(int node_cap) {
  const size_t b = (size_t)node_cap * (2 * sizeof(QNode) + sizeof(uint4) + 2 * sizeof(u64) + 3 * sizeof(unsigned short) + 1) + 64;
  return (b + 255) & ~(size_t)255;
}
size_t quadtree_presort_words() { return PS_STRIDE; }
size_t quadtree_lds_bytes(int node_cap) { return sizeof(unsigned) * 2 * QT_LDS_KEYS + quadtree_node_bytes(node_cap); }

void launch_quadtree_presort(orbgpu_ctx *c, FrameSlice fs, const int *budget) {
  c->qt_ps_active = false;
  if (c->qt_ps_levels <= 0 || !c->d_qt_ps) return;
  const DeviceGeom &g = c->geom;
  PsArgs pa{};
  pa.cells = c->d_cells + (size_t)fs.f0 * g.cells_per_frame;
  pa.cands = c->d_cands + (size_t)fs.f0 * g.cands_per_frame;
  pa.buf_b = c->d_cands_tmp + (size_t)fs.f0 * g.cands_per_frame;
  pa.ps = c->d_qt_ps + (size_t)fs.f0 * c->qt_ps_levels * PS_STRIDE;
  pa.ps_levels = c->qt_ps_levels;
  bool any = false;
  for (int l = 0; l < c->qt_ps_levels; l++) {
    int R = 0;
    const int ni = c->nini[l];
    if (ni <= 4) {  // the first round always runs (:356)
      R = 1;
      while ((ni << (2 * (R + 1))) <= budget[l] && (ni << (2 * (R + 1))) <= PS_NB_MAX) R++;
    }
    pa.depth[l] = c->qt_ps_depth[l] = R;
    pa.nini[l] = ni;
    any = any || R > 0;
  }
  if (!any) return;
  c->qt_ps_active = true;
  const dim3 pg(PS_CHUNKS / 4, fs.n, pa.ps_levels);
  hipLaunchKernelGGL((k_qt_presort<false>), pg, dim3(256), 0, fs.stream, g, pa);
  hipLaunchKernelGGL(k_qt_presort_scan, dim3(fs.n, pa.ps_levels), dim3(256), 0, fs.stream, pa);
  hipLaunchKernelGGL((k_qt_presort<true>), pg, dim3(256), 0, fs.stream, g, pa);
}

int launch_quadtree(const orbgpu_ctx *c, FrameSlice fs, const int *budget) {
  const int n_frames = fs.n;
  const DeviceGeom &g = c->geom;
  QtArgs a{};
  a.cells = c->d_cells + (size_t)fs.f0 * g.cells_per_frame;
  a.cands = c->d_cands + (size_t)fs.f0 * g.cands_per_frame;
  a.buf_a = c->d_cands_packed + (size_t)fs.f0 * g.cands_per_frame;
  a.buf_b = c->d_cands_tmp + (size_t)fs.f0 * g.cands_per_frame;
  a.cell_prefix = c->d_cell_prefix + (size_t)fs.f0 * g.cells_per_frame;
  a.kp_xy = c->d_kp_xy + (size_t)fs.f0 * g.kps_per_frame;
  a.kp_count = c->d_kp_count + (size_t)fs.f0 * g.levels;
  a.error_flag = c->d_error_flag;
  int cap = 0;
  for (int l = 0; l < g.levels; l++) {
    a.budget[l] = budget[l];
    a.nini[l] = c->nini[l];
    cap = std::max(cap, g.lv[l].kp_cap);
  }
  cap = (cap + 7) & ~7;
  a.node_cap = cap;
  size_t lds = quadtree_lds_bytes(cap);
  const bool nodes_global = lds > ORBGPU_QT_LDS_LIMIT;
  if (nodes_global) {
    if (!c->d_qt_nodes) return -1;  // orbgpu_create allocates it for such contexts
    a.node_bytes = quadtree_node_bytes(cap);
    a.node_scratch = c->d_qt_nodes + (size_t)fs.f0 * g.levels * a.node_bytes;
    lds = sizeof(unsigned) * 2 * QT_LDS_KEYS + 64;
  }
  // a level of a VGA-class image holds a few thousand candidates: 256 threads keep its short dependent rounds cheap
  // (measured with 512: 0.15 -> 0.34 ms per 256-frame batch); from about a megapixel on a round moves tens of thousands
  // of keys and 1024 threads win (1080p: 0.72 -> 0.57 ms per 64 frames, 4K: 1.27 -> 0.50 ms per 32 frames)
  const char *force = orbgpu_tune("qt_threads");  // tuning aid: 256 or 1024
  const bool big = force ? atoi(force) > 256 : (long long)g.lv[0].w * g.lv[0].h > 1024 * 768;
  const void *fn = nodes_global ? (big ? reinterpret_cast<const void *>(k_quadtree<1024, true>) : reinterpret_cast<const void *>(k_quadtree<256, true>))
                                : (big ? reinterpret_cast<const void *>(k_quadtree<1024, false>) : reinterpret_cast<const void *>(k_quadtree<256, false>));
  // the raised limit is a per-device property of the function: remembered per context (= per device)
  if (lds > 64 * 1024 && lds > c->qt_lds_configured[big]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    c->qt_lds_configured[big] = lds;
  }
  a.ps = nullptr;
  a.ps_levels = 0;
  if (c->qt_ps_active) {
    a.ps = c->d_qt_ps + (size_t)fs.f0 * c->qt_ps_levels * PS_STRIDE;
    a.ps_levels = c->qt_ps_levels;
    for (int l = 0; l < c->qt_ps_levels; l++) a.ps_depth[l] = c->qt_ps_depth[l];
  }
  a.prof = nullptr;
  const bool profile = orbgpu_tune("qt_profile") != nullptr;
  if (profile && hipMalloc(&a.prof, 64 * 8) == hipSuccess) (void)hipMemset(a.prof, 0, 64 * 8);
  const dim3 grid(n_frames, g.levels);
  if (nodes_global) {
    if (big) hipLaunchKernelGGL((k_quadtree<1024, true>), grid, dim3(1024), lds, fs.stream, g, a);
    else hipLaunchKernelGGL((k_quadtree<256, true>), grid, dim3(256), lds, fs.stream, g, a);
  } else {
    if (big) hipLaunchKernelGGL((k_quadtree<1024, false>), grid, dim3(1024), lds, fs.stream, g, a);
    else hipLaunchKernelGGL((k_quadtree<256, false>), grid, dim3(256), lds, fs.stream, g, a);
  }
  if (a.prof) {  // debugging aid: synchronous, prints the phase times of one workgroup
    unsigned long long h[64];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, a.prof, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "QT level0 frame0 (us): prefix %.1f gather+root %.1f phase1 %.1f phase2 %.1f best %.1f | n=%llu Sz@p2=%llu nvs=%llu final=%llu p1rounds=%llu p2sweeps=%llu\n",
            (h[1] - h[0]) / 100.0, (h[3] - h[1]) / 100.0, (h[4] - h[3]) / 100.0, (h[5] - h[4]) / 100.0,
            (h[6] - h[5]) / 100.0, h[10], h[11], h[12], h[13], h[15], h[14]);
    for (int r = 0; r < 4; r++) fprintf(stderr, "   p1 round %d: partition %.1f\n", r, (h[17 + 4 * r] - h[16 + 4 * r]) / 100.0);
    fprintf(stderr, "   p2 last sweep: partition %.1f\n", (h[33] - h[32]) / 100.0);
    (void)hipFree(a.prof);
  }
  return 0;
}
