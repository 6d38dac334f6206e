// FAST-9/16: threshold-independent score, per-cell NMS + threshold fallback + compaction, fused in one kernel.
// gfx950 only.
//
// Reference stage: ComputeKeyPointsOctTree cell loop, src/features/orb_feature_extractor.cpp:227-276
// (cv::FAST :259-266).  Closed form used:
//   d_k = v - ring_k,  M_dark = max over the 16 arcs of min over 9 contiguous d,
//   M_bright = max over arcs of min(-d),  score = max(M_dark, M_bright) - 1,  corner at t  <=>  score >= t.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "orbgpu_internal.h"
#include "tile_stage.hpp"

namespace {

// three-input max, pinned (the NMS neighbourhood)
__device__ __forceinline__ int vmax3(int a, int b, int c) {
  int r;
  asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Packed 3-input minimum / maximum on the two 16-bit halves of a register (gfx950: v_pk_minimum3_f16 /
// v_pk_maximum3_f16).  The operands are pixel values 0..255 sitting in the halves as integers, i.e. read as f16 they are
// denormals b * 2^-24, which order exactly like the integers (f16 denormals are not flushed: the kernel runs with the
// default mode, denormals on for f16) and come out bit for bit.  SW: bit i set <=> operand i enters with its halves
// exchanged (op_sel modifiers, no instruction).
#define ORBGPU_PK3(NAME, OP)                                                                                          \
  template <int SW>                                                                                                   \
  __device__ __forceinline__ unsigned NAME(unsigned a, unsigned b, unsigned c) {                                      \
    unsigned r;                                                                                                       \
    static_assert(SW == 0 || SW == 2 || SW == 4 || SW == 6, "operand 0 is never exchanged");                          \
    if constexpr (SW == 0) asm(OP " %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));                              \
    else if constexpr (SW == 2) asm(OP " %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); \
    else if constexpr (SW == 4) asm(OP " %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); \
    else asm(OP " %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));               \
    return r;                                                                                                         \
  }
ORBGPU_PK3(pk_min3, "v_pk_minimum3_f16")
ORBGPU_PK3(pk_max3, "v_pk_maximum3_f16")
#undef ORBGPU_PK3

// The score of a pixel that already passed the quick test (no early-out).  With d_k = v - ring_k,
//   M_dark = max over arcs of min over the arc of d = v - min over arcs of (max over the arc of ring),
//   M_bright = (max over arcs of (min over the arc of ring)) - v,
// so the window network runs on the ring pixels themselves, for both polarities, two ring positions per instruction:
// P[k] = ring_k | ring_{k+8} << 16 (k = 0..7).  The 3-windows of positions k and k + 8 are one packed operation on
// P[k], P[k+1], P[k+2], where an index beyond 7 is the register of index - 8 with its halves exchanged; the 9-windows
// likewise from the 3-windows at k, k + 3, k + 6.  8 + 8 operations per polarity, 5 for the reduction over the 16 arcs.
__device__ __forceinline__ int fast_score16_full(int v, const unsigned (&P)[8], int min_thr) {
  unsigned mn3[8], mx3[8];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    mn3[k] = pk_min3<0>(P[k], P[k + 1], P[k + 2]);
    mx3[k] = pk_max3<0>(P[k], P[k + 1], P[k + 2]);
  }
  mn3[6] = pk_min3<4>(P[6], P[7], P[0]);
  mx3[6] = pk_max3<4>(P[6], P[7], P[0]);
  mn3[7] = pk_min3<6>(P[7], P[0], P[1]);
  mx3[7] = pk_max3<6>(P[7], P[0], P[1]);
  unsigned mn9[8], mx9[8];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    mn9[k] = pk_min3<0>(mn3[k], mn3[k + 3], mn3[k + 6]);
    mx9[k] = pk_max3<0>(mx3[k], mx3[k + 3], mx3[k + 6]);
  }
#pragma unroll
  for (int k = 2; k < 5; k++) {
    mn9[k] = pk_min3<4>(mn3[k], mn3[k + 3], mn3[k - 2]);
    mx9[k] = pk_max3<4>(mx3[k], mx3[k + 3], mx3[k - 2]);
  }
#pragma unroll
  for (int k = 5; k < 8; k++) {
    mn9[k] = pk_min3<6>(mn3[k], mn3[k - 5], mn3[k - 2]);
    mx9[k] = pk_max3<6>(mx3[k], mx3[k - 5], mx3[k - 2]);
  }
  unsigned lo = pk_min3<0>(mx9[0], mx9[1], mx9[2]), hi = pk_max3<0>(mn9[0], mn9[1], mn9[2]);
  lo = pk_min3<0>(lo, mx9[3], mx9[4]);
  hi = pk_max3<0>(hi, mn9[3], mn9[4]);
  lo = pk_min3<0>(lo, mx9[5], mx9[6]);
  hi = pk_max3<0>(hi, mn9[5], mn9[6]);
  lo = pk_min3<0>(lo, mx9[7], mx9[7]);
  hi = pk_max3<0>(hi, mn9[7], mn9[7]);
  lo = pk_min3<2>(lo, lo, lo);  // low half: over both halves
  hi = pk_max3<2>(hi, hi, hi);
  const int s = max(v - (int)(lo & 0xffffu), (int)(hi & 0xffffu) - v) - 1;
  return s >= min_thr ? s : 0;
}

// Quick test of the 4 pixels of a quad on bytes, without widening: pixels quantised to 6 bits (p >> 2) leave two
// spare bits per byte, so with a = (t + 1) >> 2 the byte-wise sums v6 + (128 - a) - r6 and r6 + (128 - a) - v6 neither
// borrow nor carry across bytes and have bit 7 set exactly when v6 - r6 >= a / r6 - v6 >= a.  A corner at threshold t has
// 9 contiguous ring pixels all <= v - t - 1 (or all >= v + t + 1), hence one of every opposite pair, and
// r <= v - t - 1 implies r >> 2 <= (v >> 2) - a: the quantised test passes every pixel the exact one passes (and a few
// more: 6.7 % instead of 6.2 % of the pixels at t = 20).  The exact score decides afterwards.  Four opposite pairs (the
// even ring positions); only full-rate operations (shift, and, or, add, sub) but for six v_alignbyte_b32.
// tw -> dword left of the quad in tile row y - 3.  Returns bit 7 of byte p set <=> pixel p may be a corner at `ka`'s
// threshold.
__device__ __forceinline__ unsigned quant6(unsigned x) {
  unsigned m = 0x3f3f3f3fu;
  asm volatile("" : "+s"(m));
  return (x >> 2) & m;
}

// one v_bitop3_b32
__device__ __forceinline__ unsigned or_and(unsigned x, unsigned y, unsigned z) { return (x | y) & z; }

// RAW: the tile holds the pixels themselves (quantised here); else the tile quantised at staging time
template <bool RAW>
__device__ __forceinline__ unsigned quick_quad(const unsigned *tw, unsigned ka) {
  constexpr int TP = 36;  // FC_TP
  auto qz = [](unsigned x) { return RAW ? quant6(x) : x; };
  const unsigned r0 = qz(tw[6 * TP + 1]), r8 = qz(tw[1]);                         // (0, 3), (0, -3)
  const unsigned pa = tw[5 * TP], pb = tw[5 * TP + 1], pc = tw[5 * TP + 2];       // row y + 2: (2, 2), (-2, 2)
  const unsigned r2 = qz(__builtin_amdgcn_alignbyte(pc, pb, 2)), r14 = qz(__builtin_amdgcn_alignbyte(pb, pa, 2));
  const unsigned ma = tw[TP], mb = tw[TP + 1], mc = tw[TP + 2];                   // row y - 2: (2, -2), (-2, -2)
  const unsigned r6 = qz(__builtin_amdgcn_alignbyte(mc, mb, 2)), r10 = qz(__builtin_amdgcn_alignbyte(mb, ma, 2));
  const unsigned ca = tw[3 * TP], cb = tw[3 * TP + 1], cc = tw[3 * TP + 2];       // row y: (3, 0), (-3, 0), centre
  const unsigned r4 = qz(__builtin_amdgcn_alignbyte(cc, cb, 3)), r12 = qz(__builtin_amdgcn_alignbyte(cb, ca, 1));
  const unsigned v = qz(cb);
  const unsigned va = v + ka, vb = ka - v;
  const unsigned dark = or_and(va - r6, va - r14, or_and(va - r4, va - r12, or_and(va - r2, va - r10, (va - r0) | (va - r8))));
  const unsigned bright = or_and(r6 + vb, r14 + vb, or_and(r4 + vb, r12 + vb, or_and(r2 + vb, r10 + vb, (r0 + vb) | (r8 + vb))));
  return dark | bright;
}

// ------------------------------------------------------------------------------------------------
// Fused FAST score + per-cell NMS + threshold fallback + compaction on CELL-ALIGNED tiles.
//
// cv::FAST runs per cell in the reference (orb_feature_extractor.cpp:244-266), so its 3x3 non-maximum suppression is
// blind across cell seams and a workgroup that owns whole cells needs nothing from its neighbours.
// A workgroup takes one "group": up to 4 horizontally adjacent cells of one cell row (<= 128 pixels wide, <= 64 rows),
// stages the pixels of their detection rectangles plus the 3-pixel ring apron, and keeps every intermediate in LDS:
//   stage    the tile with 16-byte buffer loads, once raw and once quantised to 6 bits per pixel;
//   phase 1  quick test at iniThFAST on bytes (quick_quad: 4 pixels per operation on the quantised tile, conservative),
//            a lane keeps its quad column and walks down the rows; survivor bits collect in a register;
//   push     survivors -> LDS queue as (row, column), one wave scan + one LDS atomic per wave;
//   phase 2  full 9-arc score per survivor (fast_score16_full: packed 3-input min / max on the ring pixels) -> score
//            tile in LDS (zero apron);
//   phase 3  3x3 NMS per scored pixel, neighbours outside the pixel's own cell masked; survivors set a bit in their
//            cell's row mask;
//   phase 4  one wave per cell: survivors written in raster order to the cell's fixed slot of the candidate pool;
//   then, only if a cell kept nothing (orb_feature_extractor.cpp:263-266: decided per cell AFTER NMS): a second quick
//            test at minThFAST over the columns of those cells (the queue refilled from its front), score, NMS and
//            emit for them.
// No score map, no bit plane and no second kernel: the only global traffic is the tile read and the candidates.
// DUMP = true additionally scatters the non-zero scores into a global score map (orbgpu_debug_dump's stage test).
struct FastGroup { short level, row, j0, ncell; };

// -DORBGPU_FAST_PROFILE (debugging aid, never in the shipped library): wave 0 of every workgroup adds the 100 MHz
// clock ticks it spent in each phase into its own slot of fc_prof, the launch prints the means; ORBGPU_FAST_STOP=n makes
// every workgroup return after phase n (instruction counts per phase under rocprofv3 --pmc, tools/fast_phase_counts.sh)
#ifdef ORBGPU_FAST_PROFILE
constexpr int FC_PROF_WG = 65536;
__device__ unsigned fc_prof[FC_PROF_WG * 16];  // [workgroup][phase]: ticks; [15]: 1 + (took the minThFAST round)
#define FC_STAMP(i)                                                                   \
  do {                                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memrealtime();                \
    if (tid == 0 && prof_wg < FC_PROF_WG) fc_prof[prof_wg * 16 + (i)] = (unsigned)(t_now - t_prev); \
    t_prev = t_now;                                                                   \
  } while (0)
__device__ volatile int fc_stop;  // phase after which every workgroup returns (0: never): instruction counts per phase under --pmc
#define FC_STOP(n) do { if (fc_stop == (n)) return; } while (0)
#else
#define FC_STAMP(i) do { } while (0)
#define FC_STOP(n) do { } while (0)
#endif  // cells j0 .. j0 + ncell - 1 of cell row `row`

constexpr int FC_TP = 36;               // dwords per staged / score row (<= 33 quads + 2 apron dwords)
constexpr int FC_TPB = FC_TP * 4;
constexpr int FC_MAXK = 4;              // cells per group
constexpr int FC_QPITCH = 128;          // queue entries per row: only pixels of the cells are pushed, <= 128 per row (with 136 a 33-row launch took 20.8 KB, 7 workgroups per CU)
constexpr int FC_MAXROWS = 64;          // rows of a cell (hcell <= 64 is checked at create time)

__host__ __device__ constexpr size_t fast_cells_lds_bytes(int rows) {
  return (size_t)(rows + 6) * FC_TPB       // pixel tile
         + (size_t)(rows + 2) * FC_TPB     // score tile
         + (size_t)std::max(rows * FC_QPITCH * 2, (rows + 6) * FC_TPB)  // survivor queue (u16); before it the quantised tile
         + (size_t)rows * FC_MAXK * 8      // row mask per (row, cell): kept (u64); 8 workgroups per CU need <= 20 KB
         + FC_TP * 4                       // column validity mask per quad
         + 16;
}

// PART only names the instantiation: 0 = the launch the roofline is quoted on (levels >= 1, or all levels), 1 = the
// launch for level 0, which runs on the second stream beside the resize chain (profilers keep the two apart by name)
template <bool DUMP, int PART = 0>
__global__ __launch_bounds__(256) void k_fast_cells(const uint8_t *__restrict__ pyr, DeviceGeom g,
                                                    const FastGroup *__restrict__ groups, int groups_per_frame,
                                                    CellEntry *__restrict__ cells, unsigned *__restrict__ cands,
                                                    int n_frames, Level0Src lvl0, int lds_rows,
                                                    uint8_t *__restrict__ score_dump) {
  extern __shared__ __attribute__((aligned(16))) unsigned fc_lds[];
  unsigned *tile = fc_lds;
  unsigned *stile = tile + (lds_rows + 6) * FC_TP;
  unsigned short *queue = reinterpret_cast<unsigned short *>(stile + (lds_rows + 2) * FC_TP);
  unsigned *qtile = reinterpret_cast<unsigned *>(queue);  // the tile quantised for the quick test, until the first push
  // [row][cell] u64: pixels kept by NMS -- at iniThFAST first; a cell that kept nothing then (its masks are still zero)
  // is the only kind the minThFAST pass writes to, so one mask serves both passes
  unsigned *rmask = reinterpret_cast<unsigned *>(queue + lds_rows * FC_QPITCH);
  unsigned *colmask = rmask + lds_rows * FC_MAXK * 2;
  int *q_count = reinterpret_cast<int *>(colmask + FC_TP);
  int item, frame;
  if (!xcd_item_frame(groups_per_frame, n_frames, item, frame)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const FastGroup G = groups[item];
  const int l = G.level;
  const LevelGeom L = g.lv[l];
#ifdef ORBGPU_FAST_PROFILE
  const unsigned prof_wg = blockIdx.x;
  unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && prof_wg < FC_PROF_WG) fc_prof[prof_wg * 16 + 15] = 1;
#endif
  // detection rectangle of the group: rows of cell row G.row, columns of cells j0 .. j0 + ncell - 1
  const int Y0 = ORB_EDGE + G.row * L.hcell, Y1 = min(ORB_EDGE - 1 + (G.row + 1) * L.hcell, L.h - ORB_EDGE - 1);
  const int X0 = ORB_EDGE + G.j0 * L.wcell, X1 = min(ORB_EDGE - 1 + (G.j0 + G.ncell) * L.wcell, L.w - ORB_EDGE - 1);
  const int nrow = Y1 - Y0 + 1, ncolpx = X1 - X0 + 1;
  const int cell_slot = ((L.wcell + 1) >> 1) * ((L.hcell + 1) >> 1);
  CellEntry *ce = cells + (size_t)frame * g.cells_per_frame + L.cell_off + G.row * L.ncols + G.j0;
  if (nrow <= 0 || ncolpx <= 0) {  // nothing to detect in any cell of the group
    if (tid < G.ncell) ce[tid] = CellEntry{0, 0};
    return;
  }
  const int xs0 = X0 & ~3;                         // first quad column; tile byte column c <-> level x = xs0 - 4 + c
  const int ncg = ((X1 - xs0) >> 2) + 1;           // quads per row, <= 33
  const uint8_t *src = level_src(pyr, g, lvl0, frame, l, L.off) + (size_t)(Y0 - 3) * L.pitch + (xs0 - 4);

  // ---- stage (nrow + 6) rows of 36 dwords (9 x 16 bytes): a lane loads 16 bytes, 7 rows per wave and load, so the
  //      whole tile takes two loads a wave (three for the tallest cells) and lands in LDS in lane order.  Rows are
  //      staged to their full 144 bytes whatever the group's width: the bytes beyond the group's columns belong to the
  //      same or the following image row (Y1 + 3 <= h - 17, so rows follow), are never used and cost no extra
  //      instruction.  Buffer loads: the tile's first byte is the (scalar) base, a lane adds a 32-bit offset, and what
  //      lies beyond the tile's last row is cut off by the descriptor's range (returns 0, touches no memory).  All
  //      loads of a wave are issued before its first LDS write ----
  FC_STAMP(0);  // set-up
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  {
    constexpr int NR = (FC_MAXROWS + 6 + 27) / 28;
    const int nst = nrow + 6;
    // lane / 9, lane % 9; lane 63 is (row 7, column 0): the first 16 bytes of the next wave's first row once more
    const int rs = (lane * 57) >> 9, c4 = lane - 9 * rs;
    const int row0 = 7 * wave + rs;
    const unsigned pitch = (unsigned)L.pitch;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, (nst - 1) * (int)pitch + FC_TPB, 0x00020000);
    const unsigned voff = (unsigned)row0 * pitch + 16u * (unsigned)c4;
    u32x4 v[NR];
#pragma unroll
    // (unconditional: a load wholly beyond the range moves nothing, and a branch around it would make the compiler wait
    //  for the loads in flight where the paths join)
    for (int k = 0; k < NR; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 28u * k * pitch, 0, 0);
    FC_STAMP(1);  // loads issued
    // meanwhile: clear the score tile and the row masks (16 bytes per lane and store), per-quad column masks
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < (nrow + 2) * (FC_TP / 4); i += 256) reinterpret_cast<uint4 *>(stile)[i] = z4;
    if (tid < (nrow * FC_MAXK + 1) / 2) reinterpret_cast<uint4 *>(rmask)[tid] = z4;
    if (tid < 2) q_count[tid] = 0;  // [0] entries in the queue, [1] bit j: cell j falls back to minThFAST
    if (tid < FC_TPB) {  // a byte (0 / 1) per tile column beyond the left apron: a thread per pixel, 36 quads
      const int x = xs0 + tid;
      reinterpret_cast<uint8_t *>(colmask)[tid] = (x >= X0 && x <= X1) ? 1 : 0;
    }
    FC_STAMP(2);  // clears
    u32x4 *t4 = reinterpret_cast<u32x4 *>(tile) + row0 * 9 + c4, *q4 = reinterpret_cast<u32x4 *>(qtile) + row0 * 9 + c4;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      if (row0 + 28 * k < nst) {
        t4[28 * 9 * k] = v[k];
        q4[28 * 9 * k] = (v[k] >> 2) & 0x3f3f3f3fu;
      }
    }
  }
  FC_STAMP(3);  // set-up + staging issued
  __syncthreads();
  FC_STAMP(4);  // staging landed
  FC_STOP(1);

  const int t_lo = g.fast_min, t_hi = g.fast_init;
  // quantised thresholds of the quick test (quick_quad) and their per-byte constants 128 - a
  const int a_lo = (t_lo + 1) >> 2, a_hi = (t_hi + 1) >> 2;
  const unsigned ka_lo = 0x80808080u - 0x01010101u * (unsigned)a_lo, ka_hi = 0x80808080u - 0x01010101u * (unsigned)a_hi;
  // ---- quick test over the quad columns qa .. qa + nw - 1 of every row.  The 256 threads cover R = 256 / nw whole rows
  //      per pass: a lane keeps its quad column and walks down the rows r0, r0 + R, ... (no index arithmetic inside the
  //      loop, the column mask is loaded once); the lanes beyond R * nw idle.  Survivors of up to 8 passes collect in one
  //      register, bit 8 * p + pass for pixel p of the lane's quad; 16 passes cover 64 rows at R >= 7 (nw <= 33).
  //      LO = false: the test at iniThFAST on the tile quantised at staging time.  LO = true (only when a cell of the
  //      group falls back to minThFAST, and then only over the columns of such cells): the test at minThFAST, from the
  //      raw tile (the quantised one is the queue by then).  A pixel scoring below iniThFAST can neither be kept nor
  //      suppress anything in a cell that keeps a corner at iniThFAST (NMS is a strict > against a value >= iniThFAST),
  //      so the first round needs no more.
  struct QuickLane { unsigned acc[2]; unsigned e0; int R; };
  auto quick_round = [&](int qa, int nw, auto lo_tag) -> QuickLane {
    constexpr bool LO = decltype(lo_tag)::value;
    // tid / nw and 256 / nw by a 16-bit reciprocal: with M = floor(65536 / nw) + 1 or + 2 (the float estimate is within
    // 0.02 of 65536 / nw) M * nw - 65536 <= 2 * nw <= 72, and n * M >> 16 == n / nw for every n * 72 < 65536
    const unsigned M = __builtin_amdgcn_readfirstlane((unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)nw)) + 1u);
    const int R = (int)((256u * M) >> 16);
    const int r0 = (int)(((unsigned)tid * M) >> 16), cq = qa + tid - r0 * nw;
    QuickLane ql;
    ql.acc[0] = ql.acc[1] = 0;
    ql.e0 = ((unsigned)r0 << 8) | (unsigned)(4 * cq);
    ql.R = R;
    if (r0 < R) {  // (the lanes beyond R * nw idle)
      const unsigned cm = colmask[cq];
      const unsigned *tw = (LO ? tile : qtile) + r0 * FC_TP + cq;
      const int step = R * FC_TP;
      auto pass = [&](int k, unsigned acc) {
        const unsigned hb = LO ? quick_quad<true>(tw, ka_lo) : quick_quad<false>(tw, ka_hi);
        return acc | (((hb >> 7) & cm) << k);
      };
      // nf passes whose R rows all exist (no row test inside the loops, one exit each), then the last one
      // nf = (nrow - 1) / R, by the same kind of reciprocal (63 * 2 * R < 65536)
      const unsigned MR = __builtin_amdgcn_readfirstlane((unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)R)) + 1u);
      const int nf = (int)(((unsigned)(nrow - 1) * MR) >> 16);
      const int n0 = min(nf, 8);
      unsigned acc = 0;
#pragma unroll 1
      for (int k = 0; k < n0; k++) {
        acc = pass(k, acc);
        tw += step;
      }
      if (nf < 8) {
        if (r0 < nrow - nf * R) acc = pass(nf, acc);
        ql.acc[0] = acc;
      } else {
        ql.acc[0] = acc;
        acc = 0;
#pragma unroll 1
        for (int k = 8; k < nf; k++) {
          acc = pass(k - 8, acc);
          tw += step;
        }
        if (r0 < nrow - nf * R) acc = pass(nf - 8, acc);
        ql.acc[1] = acc;
      }
    }
    return ql;
  };
  // one LDS atomic per wave for the survivors of 8 passes (a wave-wide scan gives every lane its place; an atomic per
  // lane is 14 vector instructions less and keeps the LDS busy for 64 serialised updates: slower); the order inside
  // the queue is free.  Entries are (row << 8 | tile column - 4).  The atomic is written out because the compiler wraps
  // an atomicAdd under `lane == 0` into a second reduction over the (one) active lane.
  const unsigned qc_addr = (unsigned)reinterpret_cast<size_t>(q_count);
  auto push_block = [&](unsigned m, unsigned e0, int R) {
    const int cnt = __popc(m);
    const int incl = wave_incl_scan(cnt);
    const int tot = __builtin_amdgcn_readlane(incl, 63);
    if (tot == 0) return;
    unsigned old = 0;
    if (lane == 0)
      asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(qc_addr), "v"(tot) : "memory");
    old = __builtin_amdgcn_readfirstlane(old);
    int pos = (int)old + incl - cnt;
    const unsigned rstep = (unsigned)R << 8;
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      queue[pos++] = (unsigned short)(e0 + (unsigned)(bit & 7) * rstep + ((unsigned)bit >> 3));
    }
  };
  auto push_round = [&](const QuickLane &ql) {
    push_block(ql.acc[0], ql.e0, ql.R);
    if (8 * ql.R < nrow) push_block(ql.acc[1], ql.e0 + ((unsigned)(8 * ql.R) << 8), ql.R);
  };
  {
    const QuickLane ql = quick_round(0, ncg, std::false_type{});
    FC_STAMP(5);  // quick-test passes
    FC_STOP(2);
    __syncthreads();  // the queue takes the place of the quantised tile
    push_round(ql);
  }
  FC_STAMP(6);  // push
  __syncthreads();
  FC_STAMP(7);
  FC_STOP(3);
  const int n_hi = *q_count;
  const uint8_t *tb = reinterpret_cast<const uint8_t *>(tile);
  uint8_t *sb = reinterpret_cast<uint8_t *>(stile);
  const int bnd0 = X0 - xs0;  // tile column - 4 of the first pixel of cell j0
  // full 9-arc score of queue entry q -> score tile
  auto score_entry = [&](int q) {
    const int e = queue[q];
    const int ey = e >> 8, cx = e & 255;
    const uint8_t *c = tb + (ey + 3) * FC_TPB + cx + 4;
    constexpr int RP = FC_TPB;
    // ring positions k and k + 8 (opposite pixels) side by side in one register
    unsigned P[8];
    P[0] = c[3 * RP] | (unsigned)c[-3 * RP] << 16;          P[1] = c[3 * RP + 1] | (unsigned)c[-3 * RP - 1] << 16;
    P[2] = c[2 * RP + 2] | (unsigned)c[-2 * RP - 2] << 16;  P[3] = c[RP + 3] | (unsigned)c[-RP - 3] << 16;
    P[4] = c[3] | (unsigned)c[-3] << 16;                    P[5] = c[-RP + 3] | (unsigned)c[RP - 3] << 16;
    P[6] = c[-2 * RP + 2] | (unsigned)c[2 * RP - 2] << 16;  P[7] = c[-3 * RP + 1] | (unsigned)c[3 * RP - 1] << 16;
    const int sc = fast_score16_full((int)c[0], P, t_lo);
    if (sc) {
      sb[(ey + 1) * FC_TPB + cx + 4] = (uint8_t)sc;
      if (DUMP) score_dump[(size_t)frame * g.frame_bytes + L.off + (size_t)(Y0 + ey) * L.pitch + (xs0 + cx)] = (uint8_t)sc;
    }
  };
  // 3x3 NMS of (decoded) queue entry q inside the pixel's own cell; survivors with score >= t_keep in the cells of
  // need_mask set their bit in the cell's row mask
  auto nms_entry = [&](int q, int t_keep, unsigned need_mask) {
    const int e = queue[q];
    const int ey = e >> 8, cx = e & 255;
    const uint8_t *c = sb + (ey + 1) * FC_TPB + cx + 4;
    const int v = c[0];
    if (v < t_keep) return;  // the score tile holds 0 below minThFAST
    const int rel = cx - bnd0;
    const int j = (rel >= L.wcell) + (rel >= 2 * L.wcell) + (rel >= 3 * L.wcell);
    if (!((need_mask >> j) & 1u)) return;
    const int xx = rel - j * L.wcell;  // column inside the cell
    int nl = vmax3(c[-FC_TPB - 1], c[-1], c[FC_TPB - 1]);
    int nr = vmax3(c[-FC_TPB + 1], c[1], c[FC_TPB + 1]);
    nl = xx == 0 ? 0 : nl;                  // "NMS does not see across the seam"
    nr = xx == L.wcell - 1 ? 0 : nr;        // (beyond X1 nothing was scored)
    const int nm = vmax3(c[-FC_TPB], c[FC_TPB], max(nl, nr));
    if (v > nm) atomicOr(rmask + ((ey * FC_MAXK + j) << 1) + (xx >> 5), 1u << (xx & 31));
  };
  // wave = cell, lane = row: survivors of the cell's row masks in raster order -> the cell's fixed slot of the
  // candidate pool; returns the number written (wave-uniform)
  auto emit_cell = [&](int j) -> int {
    const int cx0 = X0 + j * L.wcell;
    uint2 mk = make_uint2(0, 0);
    if (lane < nrow) mk = *reinterpret_cast<const uint2 *>(rmask + ((lane * FC_MAXK + j) << 1));
    const int cnt = __popc(mk.x) + __popc(mk.y);
    const int incl = wave_incl_scan(cnt);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return 0;
    // every cell owns a fixed slot of ceil(wCell/2) * ceil(hCell/2) entries in its level's sub-pool: NMS survivors
    // are never adjacent, so the slot cannot overflow and no atomic is needed
    const int basei = (G.row * L.ncols + G.j0 + j) * cell_slot;
    if (lane == 0) ce[j] = CellEntry{basei, total};
    unsigned long long bits = ((unsigned long long)mk.y << 32) | mk.x;
    int pos = incl - cnt;
    unsigned *out = cands + (size_t)frame * g.cands_per_frame + L.cand_off + basei;
    const uint8_t *srow = sb + (lane + 1) * FC_TPB + 4 + (cx0 - xs0);
    const unsigned ybits = (unsigned)(Y0 + lane - ORB_MIN_BORDER) << 12;
    while (bits) {
      const int xx = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      out[pos++] = (unsigned)(cx0 + xx - ORB_MIN_BORDER) | ybits | ((unsigned)srow[xx] << 24);
    }
    return total;
  };

  // ---- iniThFAST: score the front of the queue, NMS, emit ---------------------------------------------------------
  for (int q = tid; q < n_hi; q += 256) score_entry(q);
  FC_STAMP(8);  // full score
  __syncthreads();
  FC_STAMP(9);
  FC_STOP(4);
  for (int q = tid; q < n_hi; q += 256) nms_entry(q, t_hi, 0xfu);
  FC_STAMP(10);  // NMS
  __syncthreads();
  FC_STAMP(11);
  FC_STOP(5);
  if (wave < G.ncell) {
    // a cell without columns (beyond the last detection column) holds nothing at either threshold
    const bool has_cols = X0 + wave * L.wcell <= X1;
    const int total = has_cols ? emit_cell(wave) : 0;
    if (total == 0) {
      if (has_cols) {
        if (lane == 0) atomicOr(reinterpret_cast<unsigned *>(q_count + 1), 1u << wave);  // falls back to minThFAST
      } else if (lane == 0) {
        ce[wave] = CellEntry{0, 0};
      }
    }
  }
  FC_STAMP(12);  // emit
  __syncthreads();
  FC_STAMP(13);
  FC_STOP(6);
  // ---- minThFAST for the cells that kept nothing (orb_feature_extractor.cpp:263-266: decided per cell AFTER NMS) ----
  const unsigned fallback = (unsigned)q_count[1];
  const unsigned score_mask = DUMP ? 0xfu : fallback;
  if (score_mask == 0) return;
  // second quick-test round over the quad columns of the cells in score_mask: per-quad masks narrowed to those cells.
  // The queue starts again from its front (its iniThFAST entries are done with: the cells they belong to have either
  // been emitted or are tested again now -- what passes the quick test at iniThFAST passes it at minThFAST, and scoring
  // a pixel twice writes the same value twice)
  const int j_a = __ffs(score_mask) - 1, j_b = min(31 - __clz(score_mask), (int)G.ncell - 1);
  const int qa_lo = (X0 + j_a * L.wcell - xs0) >> 2;
  const int nw_lo = ((min(X0 + (j_b + 1) * L.wcell - 1, X1) - xs0) >> 2) - qa_lo + 1;
  if (tid < FC_TPB) {
    const int rel = xs0 + tid - X0;
    const int j = (rel >= L.wcell) + (rel >= 2 * L.wcell) + (rel >= 3 * L.wcell);
    reinterpret_cast<uint8_t *>(colmask)[tid] = (rel >= 0 && rel <= X1 - X0 && ((score_mask >> j) & 1u)) ? 1 : 0;
  }
  if (tid == 0) q_count[0] = 0;
  __syncthreads();
  push_round(quick_round(qa_lo, nw_lo, std::true_type{}));
  __syncthreads();
  const int n_lo = *q_count;
  for (int q = tid; q < n_lo; q += 256) score_entry(q);
  if (fallback == 0) return;
  __syncthreads();
  for (int q = tid; q < n_lo; q += 256) nms_entry(q, 1, fallback);
  __syncthreads();
  if (wave < G.ncell && ((fallback >> wave) & 1u)) {
    if (emit_cell(wave) == 0 && lane == 0) ce[wave] = CellEntry{(G.row * L.ncols + G.j0 + wave) * cell_slot, 0};
  }
  FC_STAMP(14);  // the whole minThFAST round
#ifdef ORBGPU_FAST_PROFILE
  if (tid == 0 && prof_wg < FC_PROF_WG) fc_prof[prof_wg * 16 + 15] = 2;
#endif
}

}  // namespace

size_t fast_cells_lds(int rows) { return fast_cells_lds_bytes(rows); }

// groups of up to FC_MAXK cells, <= 128 pixels wide, balanced over a cell row; host side of k_fast_cells
int build_fast_groups(const DeviceGeom &g, std::vector<short> &table) {
  table.clear();
  for (int l = 0; l < g.levels; l++) {
    const LevelGeom &L = g.lv[l];
    const int kmax = std::max(1, std::min(FC_MAXK, 128 / L.wcell));
    const int gcols = (L.ncols + kmax - 1) / kmax;
    for (int i = 0; i < L.nrows; i++)
      for (int q = 0; q < gcols; q++) {
        const int j0 = q * L.ncols / gcols, j1 = (q + 1) * L.ncols / gcols;
        table.push_back((short)l);
        table.push_back((short)i);
        table.push_back((short)j0);
        table.push_back((short)(j1 - j0));
      }
  }
  return (int)(table.size() / 4);
}

// the groups of the levels [la, lb) (the table is level-major); early: k_fast_cells<false, 1>, the launches beside the resize chain
void launch_fast_cells(const orbgpu_ctx *c, FrameSlice fs, uint8_t *score_dump, int la, int lb, bool early) {
  const DeviceGeom &g = c->geom;
  const int first = c->fast_group_first[la];
  const int count = c->fast_group_first[lb] - first;
  if (count <= 0) return;
#ifdef ORBGPU_FAST_PROFILE
  {
    static int stop_set = -1;
    if (stop_set < 0) {
      const char *e = getenv("ORBGPU_FAST_STOP");
      stop_set = e ? atoi(e) : 0;
      if (hipMemcpyToSymbol(HIP_SYMBOL(fc_stop), &stop_set, sizeof(int)) != hipSuccess) fprintf(stderr, "fc_stop: copy failed\n");
    }
  }
#endif
  const FastGroup *groups = reinterpret_cast<const FastGroup *>(c->d_fast_groups) + first;
  // dynamic LDS by the tallest cell of the launch's own levels: 32 rows (level 0 of a VGA frame) fit 8 workgroups per CU,
  // 33 rows and more only 7 -- and the kernel's time follows its occupancy
  int rows = 0;
  for (int l = la; l < lb; l++) rows = std::max(rows, g.lv[l].hcell);
  const size_t lds = fast_cells_lds_bytes(rows);
  dim3 grid(count * xcd_frames(fs.n));
  const uint8_t *pyr = c->d_pyr + (size_t)fs.f0 * g.frame_bytes;
  CellEntry *cells = c->d_cells + (size_t)fs.f0 * g.cells_per_frame;
  unsigned *cands = c->d_cands + (size_t)fs.f0 * g.cands_per_frame;
  if (score_dump)
    hipLaunchKernelGGL((k_fast_cells<true, 0>), grid, dim3(256), lds, fs.stream, pyr, g, groups, count, cells, cands,
                       fs.n, slice_lvl0(c, fs), rows, score_dump);
  else if (early)
    hipLaunchKernelGGL((k_fast_cells<false, 1>), grid, dim3(256), lds, fs.stream, pyr, g, groups, count, cells, cands,
                       fs.n, slice_lvl0(c, fs), rows, (uint8_t *)nullptr);
  else
    hipLaunchKernelGGL((k_fast_cells<false, 0>), grid, dim3(256), lds, fs.stream, pyr, g, groups, count, cells, cands,
                       fs.n, slice_lvl0(c, fs), rows, (uint8_t *)nullptr);
#ifdef ORBGPU_FAST_PROFILE
  if (!early && !getenv("ORBGPU_FAST_STOP")) {
    static std::vector<unsigned> h(FC_PROF_WG * 16);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fc_prof), h.size() * 4);
    static const char *nm[15] = {"setup", "load-issue", "clears", "tile-write", "bar", "quick", "push", "bar", "score", "bar", "nms", "bar", "emit", "bar", "fallback"};
    double sum[15] = {0}, n = 0, nfb = 0;
    for (unsigned w = 0; w < std::min<unsigned>(grid.x, FC_PROF_WG); w++) {
      if (!h[w * 16 + 15]) continue;
      n++;
      const bool fb = h[w * 16 + 15] == 2;
      nfb += fb;
      for (int i = 0; i < (fb ? 15 : 14); i++) sum[i] += h[w * 16 + i];
    }
    fprintf(stderr, "FAST profile, %.0f workgroups (%.0f with fallback), mean ticks (10 ns) per workgroup:", n, nfb);
    double tot = 0;
    for (int i = 0; i < 15; i++) {
      fprintf(stderr, " %s %.1f", nm[i], sum[i] / n);
      tot += sum[i] / n;
    }
    fprintf(stderr, " | total %.1f\n", tot);
    (void)hipMemset(nullptr, 0, 0);
    std::fill(h.begin(), h.end(), 0u);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(fc_prof), h.data(), h.size() * 4);
  }
#endif
}


