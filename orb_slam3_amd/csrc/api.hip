// C ABI of liborbgpu.so (include/orbgpu.h): context set-up, derived geometry, resize tables, the
// extraction pipeline and the host-pointer wrappers of the matching entry points.  gfx950 only.
#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "orbgpu_internal.h"

static thread_local std::string g_create_error;

int fail(orbgpu_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  else g_create_error = msg;
  return code;
}

const char *orbgpu_tune(const char *key) {
  static thread_local std::string value;
  const char *all = std::getenv("ORBGPU_TUNE");
  if (!all || !key) return nullptr;
  const size_t klen = std::strlen(key);
  for (const char *p = all; *p;) {
    const char *end = std::strchr(p, ';');
    const size_t len = end ? (size_t)(end - p) : std::strlen(p);
    if (len > klen && std::strncmp(p, key, klen) == 0 && p[klen] == '=') {
      value.assign(p + klen + 1, len - klen - 1);
      return value.c_str();
    }
    if (!end) break;
    p = end + 1;
  }
  return nullptr;
}

int fail_hip(orbgpu_ctx *ctx, hipError_t e, const char *expr, const char *file, int line) {
  return fail(ctx, ORBGPU_E_HIP,
              std::string("HIP error '") + hipGetErrorString(e) + "' in " + expr + " at " + file + ":" + std::to_string(line));
}

namespace {

inline int cv_round_f(float v) { return (int)lrintf(v); }  // cvRound: nearest even
inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
inline short sat_short(int v) { return (short)std::min(std::max(v, -32768), 32767); }
inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// feature budget per level, orb_feature_extractor.cpp:211-224
void level_budget(double scale_factor, int levels, int n_features, int *budget) {
  const double factor = 1.0 / scale_factor;
  double per = (double)(size_t)n_features * (1 - factor) / (1 - std::pow(factor, (double)levels));
  size_t total = 0;
  for (int l = 0; l < levels - 1; l++) {
    budget[l] = (int)std::round(per);
    total += (size_t)budget[l];
    per *= factor;
  }
  budget[levels - 1] = (int)((size_t)n_features - total);
}

// Geometry of every level and the device slab layout.  Returns 0 or an error code.
int build_geometry(orbgpu_ctx *ctx) {
  const orbgpu_params &p = ctx->params;
  DeviceGeom &g = ctx->geom;
  g.levels = p.levels;
  g.fast_init = p.fast_init;
  g.fast_min = p.fast_min;
  // Initialize(), orb_feature_extractor.cpp:53-64
  ctx->scale[0] = ctx->inv_scale[0] = ctx->sigma2[0] = 1.0;
  for (int i = 1; i < p.levels; i++) {
    ctx->scale[i] = ctx->scale[i - 1] * p.scale_factor;
    ctx->inv_scale[i] = 1.0 / ctx->scale[i];
    ctx->sigma2[i] = ctx->scale[i] * ctx->scale[i];
  }
  int budget[ORBGPU_MAX_LEVELS];
  level_budget(p.scale_factor, p.levels, p.max_features, budget);
  unsigned off = 0;
  int cell_off = 0, cand_off = 0, kp_off = 0, xt = 0, yt = 0;
  for (int l = 0; l < p.levels; l++) {
    LevelGeom &L = g.lv[l];
    const float s = (float)ctx->inv_scale[l];  // :506-508
    L.w = cv_round_f((float)p.width * s);
    L.h = cv_round_f((float)p.height * s);
    if (L.w - 2 * ORB_MIN_BORDER < 30 || L.h - 2 * ORB_MIN_BORDER < 30)
      return fail(ctx, ORBGPU_E_INVALID, "level " + std::to_string(l) + " is smaller than one 30-px FAST cell plus borders");
    if (L.w - 2 * ORB_MIN_BORDER > 4095 || L.h - 2 * ORB_MIN_BORDER > 4095)
      return fail(ctx, ORBGPU_E_INVALID, "images wider or taller than 4127 px are not supported (12-bit candidate coordinates)");
    L.pitch = align_up(L.w, 16);
    L.off = off;
    off += (unsigned)align_up(L.pitch * L.h + 64, 256);
    const float width = (float)(L.w - 2 * ORB_MIN_BORDER), height = (float)(L.h - 2 * ORB_MIN_BORDER);  // :236-242
    L.ncols = (int)(width / 30.f);
    L.nrows = (int)(height / 30.f);
    L.wcell = (int)std::ceil(width / (float)L.ncols);
    L.hcell = (int)std::ceil(height / (float)L.nrows);
    if (L.wcell > 60 || L.hcell > 64) return fail(ctx, ORBGPU_E_INVALID, "FAST cell larger than 60 x 64 px");
    L.cell_off = cell_off;
    cell_off += L.ncols * L.nrows;
    L.cand_off = cand_off;
    L.cand_cap = L.ncols * L.nrows * ((L.wcell + 1) / 2) * ((L.hcell + 1) / 2);  // NMS survivors are never adjacent
    cand_off += align_up(L.cand_cap, 64);
    ctx->nini[l] = (int)std::round((double)(L.w - 2 * ORB_MIN_BORDER) / (double)(L.h - 2 * ORB_MIN_BORDER));  // :310
    if (ctx->nini[l] < 1) return fail(ctx, ORBGPU_E_INVALID, "image taller than twice its width: no quadtree root");
    L.kp_off = kp_off;
    L.kp_cap = std::max(budget[l], 0) + 4 * ctx->nini[l] + 4;  // the node list overshoots the budget by < 3
    kp_off += L.kp_cap;
    L.xtab_off = xt;
    L.ytab_off = yt;
    xt += L.w;
    yt += L.h;
    L.kp_size = (int)(ORB_PATCH * ctx->scale[l]);  // :284
    L.scale_f = (float)ctx->scale[l];              // :568
  }
  g.frame_bytes = off;
  g.cells_per_frame = cell_off;
  g.cands_per_frame = cand_off;
  g.kps_per_frame = kp_off;
  for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) {
    g.kp_offs[l] = l < g.levels ? g.lv[l].kp_off : INT_MAX;
    g.cell_offs[l] = l < g.levels ? g.lv[l].cell_off : INT_MAX;
  }
  return 0;
}

// cv::resize coefficient tables, one entry per destination column / row of levels >= 1
int build_resize_tables(orbgpu_ctx *ctx, std::vector<XTab> &xt, std::vector<YTab> &yt) {
  const DeviceGeom &g = ctx->geom;
  xt.assign((size_t)g.lv[g.levels - 1].xtab_off + g.lv[g.levels - 1].w, XTab{0, 0, 0, 0});
  yt.assign((size_t)g.lv[g.levels - 1].ytab_off + g.lv[g.levels - 1].h, YTab{0, 0, 0, 0});
  for (int l = 1; l < g.levels; l++) {
    const int sw = g.lv[l - 1].w, sh = g.lv[l - 1].h, dw = g.lv[l].w, dh = g.lv[l].h;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
    if (std::fabs(scale_x - 2.0) < 2.3e-16 && std::fabs(scale_y - 2.0) < 2.3e-16)
      return fail(ctx, ORBGPU_E_INVALID, "exact 2x decimation switches cv::resize to INTER_AREA: unsupported scale_factor");
    if (scale_x > 2.4 || scale_y > 2.4) return fail(ctx, ORBGPU_E_INVALID, "scale_factor above 2.4 is not supported");
    for (int dx = 0; dx < dw; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = cv_floor_f(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
      XTab &X = xt[(size_t)g.lv[l].xtab_off + dx];
      X.sx0 = sx;
      X.sx1 = std::min(sx + 1, sw - 1);  // weight is 0 whenever this clamps (the dx >= xmax tail)
      X.a0 = sat_short(cv_round_f((1.f - fx) * 2048.f));
      X.a1 = sat_short(cv_round_f(fx * 2048.f));
    }
    for (int dy = 0; dy < dh; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = cv_floor_f(fy);
      fy -= sy;
      YTab &Y = yt[(size_t)g.lv[l].ytab_off + dy];
      Y.sy0 = std::min(std::max(sy, 0), sh - 1);
      Y.sy1 = std::min(std::max(sy + 1, 0), sh - 1);
      Y.b0 = sat_short(cv_round_f((1.f - fy) * 2048.f));
      Y.b1 = sat_short(cv_round_f(fy * 2048.f));
    }
    // LDS footprint of the largest 128 x 64 output tile of this level (k_resize_level stages exactly this)
    const XTab *X = &xt[(size_t)g.lv[l].xtab_off];
    const YTab *Y = &yt[(size_t)g.lv[l].ytab_off];
    int cols = 0, rows = 0;
    for (int dx0 = 0; dx0 < dw; dx0 += 128) {
      const int dx1 = std::min(dx0 + 128, dw) - 1;
      cols = std::max(cols, ((X[dx1].sx0 + 1 - (X[dx0].sx0 & ~15)) / 16 + 1) * 16);
    }
    for (int dy0 = 0; dy0 < dh; dy0 += 64) {
      const int dy1 = std::min(dy0 + 64, dh) - 1;
      rows = std::max(rows, Y[dy1].sy1 - Y[dy0].sy0 + 1);
    }
    ctx->rs_pitch[l] = cols;
    ctx->rs_rows[l] = rows;
    if ((size_t)cols * rows > 64 * 1024) return fail(ctx, ORBGPU_E_INVALID, "resize tile footprint exceeds LDS");
  }
  return 0;
}

template <class T>
hipError_t dmalloc(T **p, size_t n) {
  return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T));
}
template <class T>
hipError_t hmalloc(T **p, size_t n) {
  return hipHostMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault);
}

int ensure_match_scratch(orbgpu_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->match_scratch_bytes) return 0;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->d_match_scratch) HIP_TRY(hipFree(ctx->d_match_scratch));
  ctx->d_match_scratch = nullptr;
  ctx->match_scratch_bytes = 0;
  bytes = (bytes + (1u << 20)) & ~(size_t)((1u << 20) - 1);
  HIP_TRY(hipMalloc(&ctx->d_match_scratch, bytes));
  ctx->match_scratch_bytes = bytes;
  return 0;
}

int keypoint_capacity(const orbgpu_ctx *ctx, int feature_count) {
  int budget[ORBGPU_MAX_LEVELS];
  level_budget(ctx->params.scale_factor, ctx->params.levels, feature_count, budget);
  int cap = 0;
  for (int l = 0; l < ctx->params.levels; l++) cap += std::max(budget[l], 0) + 4 * ctx->nini[l] + 4;
  return cap;
}

// Side streams and fork / join events of slice s, created on first use: a default context (one slice) then holds three
// streams (its own, level 0's FAST launch, the blur) instead of twelve -- a process has 4 hardware queues to map them on
int ensure_slice(orbgpu_ctx *ctx, int s) {
  if (s > 0 && !ctx->slice_stream[s]) HIP_TRY(hipStreamCreateWithFlags(&ctx->slice_stream[s], hipStreamNonBlocking));
  if (!ctx->slice_l0[s]) HIP_TRY(hipStreamCreateWithFlags(&ctx->slice_l0[s], hipStreamNonBlocking));
  if (!ctx->slice_blur[s]) HIP_TRY(hipStreamCreateWithFlags(&ctx->slice_blur[s], hipStreamNonBlocking));
  for (auto &e : ctx->slice_ev[s])
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

// drains and destroys every side stream and its events (they come back on demand)
void release_side_streams(orbgpu_ctx *ctx) {
  if (ctx->head_stream) {
    (void)hipStreamSynchronize(ctx->head_stream);
    (void)hipStreamDestroy(ctx->head_stream);
    ctx->head_stream = nullptr;
  }
  // (the context's stream is drained by every caller of this function: both working sets are idle, the second goes back)
  ctx->set_state[0] = ctx->set_state[1] = orbgpu_ctx::SET_IDLE;
  void *alt[3] = {ctx->d_pyr_alt, ctx->d_cells_alt, ctx->d_cands_alt};
  if (ctx->d_pyr_alt && ctx->graph_one) {
    // (a one-frame graph captured since the last switch of working sets addresses the current one, which stays; dropped
    //  all the same: one rule -- no graph outlives a change of the sets)
    (void)hipGraphExecDestroy(ctx->graph_one);
    ctx->graph_one = nullptr;
    ctx->graph_one_features = 0;
  }
  for (void *q : alt)
    if (q) (void)hipFree(q);
  ctx->d_pyr_alt = nullptr;
  ctx->d_cells_alt = nullptr;
  ctx->d_cands_alt = nullptr;
  for (int k = 0; k < orbgpu_ctx::MAX_SLICES; k++) {
    hipStream_t *st[3] = {&ctx->slice_stream[k], &ctx->slice_blur[k], &ctx->slice_l0[k]};
    for (hipStream_t *q : st)
      if (*q) (void)hipStreamSynchronize(*q);
    for (auto &e : ctx->slice_ev[k])
      if (e) {
        (void)hipEventDestroy(e);
        e = nullptr;
      }
    for (hipStream_t *q : st)
      if (*q) {
        (void)hipStreamDestroy(*q);
        *q = nullptr;
      }
  }
}

float elapsed(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

// The whole extraction pipeline on device-resident level-0 images already imported into ctx->d_pyr.
// Everything is enqueued on the context's streams; nothing here waits for the GPU.  Stage times and the device
// error flag are collected by finish_pipeline().
//
// The batch is cut into slices of frames that run side by side, slice 0 on the context's stream and the others on
// streams forked from it (and joined back into it at the end): every kernel here is bound by instruction issue or by
// latency, never by HBM, so the dependent, latency-bound stages of one slice (quadtree rounds, orientation ->
// descriptor chains, small pyramid levels) fill issue slots that the instruction-bound stages of another leave.
// Whether an extract call of `feature_count` keypoints per frame blurs per keypoint (k_patch_describe) instead of per
// frame: when the keypoints' 43 x 43 patches cover well under the pyramid's pixels.  ORBGPU_PATCH_BLUR=0 / 1 overrides.
bool patch_mode(const orbgpu_ctx *ctx, int feature_count) {
  if (ctx->patch_blur_forced >= 0) return ctx->patch_blur_forced != 0;
  long long P = 0;
  for (int l = 0; l < ctx->geom.levels; l++) P += (long long)ctx->geom.lv[l].w * ctx->geom.lv[l].h;
  return (double)feature_count * 43.0 * 43.0 < ORBGPU_PATCH_BLUR_RATIO * (double)P;
}

// capture: the call is being recorded into a hipGraph (single-frame host path): no timing events, no event-ring bookkeeping
int run_pipeline(orbgpu_ctx *ctx, int n_frames, int feature_count, orbgpu_keypoint *d_kp, uint8_t *d_desc, int capacity,
                 int *d_n_out, bool capture = false) {
  const DeviceGeom &g = ctx->geom;
  const int L = g.levels;
  int budget[ORBGPU_MAX_LEVELS];
  level_budget(ctx->params.scale_factor, L, feature_count, budget);
  // the device error flag is sticky: it is only cleared by finish_pipeline(), so queued calls cannot hide an error
  hipEvent_t *ev = ctx->ev[ctx->ev_head];
  int n_slices = ctx->n_slices > 0 ? ctx->n_slices : 1;
  n_slices = std::max(1, std::min(std::min(n_slices, (int)orbgpu_ctx::MAX_SLICES), n_frames));
  // an event between two kernels of a stream costs 6-7 us of stream time: by default only the FAST kernel is bracketed
  const bool stage_timing = ctx->stage_timing && !capture;
  const bool ends = stage_timing || n_slices > 1;  // ev[0]: start of the call (and the slices' fork), ev[9]: its end
  for (int s = 0; s < n_slices; s++) {
    const int rc = ensure_slice(ctx, s);
    if (rc) return rc;
  }
  if (ends) HIP_TRY(hipEventRecord(ev[0], ctx->stream));
  for (int s = 0; s < n_slices; s++) {
    const int f0 = (int)((long long)n_frames * s / n_slices), f1 = (int)((long long)n_frames * (s + 1) / n_slices);
    const FrameSlice fs{f0, f1 - f0, s == 0 ? ctx->stream : ctx->slice_stream[s]};
    const FrameSlice fb{f0, f1 - f0, ctx->no_overlap ? fs.stream : ctx->slice_blur[s]};
    hipEvent_t *sev = ctx->slice_ev[s];
    const bool timed = s == 0 && stage_timing;       // the stage timers follow slice 0
    const bool timed_fast = s == 0 && !capture;      // the FAST kernel (the roofline kernel) is always bracketed
    if (s > 0) HIP_TRY(hipStreamWaitEvent(fs.stream, ev[0], 0));  // fork: after everything already on the context's stream
    // FAST of the first E levels on a stream of their own beside the resize chain, each as soon as its level exists
    // (level 0 needs no pyramid at all); the launch on the critical path covers the levels from E on
    const int E = ctx->early_l0 ? std::min(ctx->early_levels, L - 1) : 0;
    const FrameSlice fl{f0, f1 - f0, ctx->no_overlap ? fs.stream : ctx->slice_l0[s]};
    // orbgpu_set_input_ready: the head (pyramid + FAST) of this call on a stream that does not wait for the context's
    // stream -- the previous batch's quadtree, orientation, descriptor and link kernels are chains of short rounds and
    // gathers that wait on memory, the head is vector-ALU work.  It writes the OTHER working set, so it waits only for
    // the caller's input event and for the event recorded where that set was read for the last time (by the call before
    // the previous one, behind its orientation and blur).  Everything from the quadtree on stays on the context's
    // stream, behind the head's end.
    const bool patch = patch_mode(ctx, feature_count);
    bool piped = ctx->call_piped && n_slices == 1 && !capture && !stage_timing && !ctx->no_overlap && !patch &&
                 ctx->set_state[1 - ctx->work_set] != orbgpu_ctx::SET_BUSY;
    if (piped && !ctx->d_pyr_alt) {  // the second working set, on first use (a failed allocation: the call runs unpiped)
      const size_t B = (size_t)ctx->params.max_batch;
      uint8_t *pyr = nullptr;
      CellEntry *cells = nullptr;
      unsigned *cands = nullptr;
      if (dmalloc(&pyr, B * g.frame_bytes + 256) == hipSuccess && dmalloc(&cells, B * g.cells_per_frame) == hipSuccess &&
          dmalloc(&cands, B * g.cands_per_frame) == hipSuccess &&
          hipMemset(pyr, 0, B * g.frame_bytes + 256) == hipSuccess) {
        ctx->d_pyr_alt = pyr;
        ctx->d_cells_alt = cells;
        ctx->d_cands_alt = cands;
      } else {
        (void)hipGetLastError();
        if (pyr) (void)hipFree(pyr);
        if (cells) (void)hipFree(cells);
        if (cands) (void)hipFree(cands);
        piped = false;
      }
    }
    if (piped) {
      if (!ctx->head_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->head_stream, hipStreamNonBlocking));
      if (!ctx->ev_head_done) HIP_TRY(hipEventCreateWithFlags(&ctx->ev_head_done, hipEventDisableTiming));
      if (ctx->graph_one) {  // a captured one-frame graph addresses the set of its time: the next one-frame call captures again
        (void)hipGraphExecDestroy(ctx->graph_one);
        ctx->graph_one = nullptr;
        ctx->graph_one_features = 0;
      }
      std::swap(ctx->d_pyr, ctx->d_pyr_alt);
      std::swap(ctx->d_cells, ctx->d_cells_alt);
      std::swap(ctx->d_cands, ctx->d_cands_alt);
      ctx->work_set = 1 - ctx->work_set;
      if (ctx->input_event) HIP_TRY(hipStreamWaitEvent(ctx->head_stream, ctx->input_event, 0));
      if (ctx->set_state[ctx->work_set] == orbgpu_ctx::SET_EVENT)
        HIP_TRY(hipStreamWaitEvent(ctx->head_stream, ctx->ev_set_free[ctx->work_set], 0));
    }
    // until the event below is recorded the set is in use by whatever this call enqueues
    ctx->set_state[ctx->work_set] = orbgpu_ctx::SET_BUSY;
    const FrameSlice fh{f0, f1 - f0, piped ? ctx->head_stream : fs.stream};
    if (E > 0) {
      HIP_TRY(hipEventRecord(sev[3], fh.stream));  // after everything already on the stream (the caller's images included)
      HIP_TRY(hipStreamWaitEvent(fl.stream, sev[3], 0));
      if (timed) HIP_TRY(hipEventRecord(ev[10], fl.stream));
      launch_fast_cells(ctx, fl, nullptr, 0, 1, true);
      if (timed) HIP_TRY(hipEventRecord(ev[11], fl.stream));
    }
    for (int l = 1; l < L; l++) {
      launch_pyramid_level(ctx, fh, l);
      if (l < E) {
        HIP_TRY(hipEventRecord(sev[4 + l], fh.stream));
        HIP_TRY(hipStreamWaitEvent(fl.stream, sev[4 + l], 0));
        launch_fast_cells(ctx, fl, nullptr, l, l + 1, true);
      }
    }
    if (E > 0) HIP_TRY(hipEventRecord(sev[4], fl.stream));
    if (timed_fast) HIP_TRY(hipEventRecord(ev[1], fh.stream));
    launch_fast_cells(ctx, fh, nullptr, E, L, false);
    if (timed_fast) HIP_TRY(hipEventRecord(ev[2], fh.stream));
    if (E > 0) HIP_TRY(hipStreamWaitEvent(fh.stream, sev[4], 0));  // the quadtree needs every level's cells
    if (piped) {  // join: the context's stream continues behind the head
      HIP_TRY(hipEventRecord(ctx->ev_head_done, fh.stream));
      HIP_TRY(hipStreamWaitEvent(fs.stream, ctx->ev_head_done, 0));
    }
    // the blur's fork point: the FAST kernel's end event when it has just been recorded (one event between two kernels
    // of the critical path instead of two)
    // patch mode (large images, few keypoints per pixel): no full-frame blur at all -- one kernel behind the quadtree takes
    // the orientation, blurs the keypoint's own 43 x 43 patch and runs the tests (k_patch_describe)
    ctx->last_patch_mode = patch;
    // (a piped head may end before the previous call's descriptor kernel has read the blurred slab: the blur then forks
    //  from the context's stream, behind the join, not from the FAST kernel's end event)
    hipEvent_t fork_ev = timed_fast && !piped ? ev[2] : sev[0];
    if (fork_ev == sev[0] && !patch) HIP_TRY(hipEventRecord(sev[0], fs.stream));
    // the blur needs only the pyramid; it runs on its own stream beside the quadtree kernel, which is a chain of short
    // dependent rounds in levels x frames workgroups and leaves most of the chip idle
    // (enqueued after the quadtree so that the quadtree's workgroups are placed first and the blur fills the rest)
    launch_quadtree_presort(ctx, fs, budget);
    if (ctx->qt_ps_active && ctx->qt_ps_blur_late && !patch) {  // the blur starts beside the quadtree kernel, behind the presort
      fork_ev = sev[0];
      HIP_TRY(hipEventRecord(sev[0], fs.stream));
    }
    if (launch_quadtree(ctx, fs, budget) != 0) {
      // earlier slices joined the context's stream at the end of their turn and this slice's level-0 stream has joined
      // fs.stream; a forked fs.stream itself has not: join it, so that the context's stream still covers all enqueued work
      if (s > 0 && hipEventRecord(sev[2], fs.stream) == hipSuccess) (void)hipStreamWaitEvent(ctx->stream, sev[2], 0);
      return fail(ctx, ORBGPU_E_CAPACITY, "feature budget too large for the quadtree kernel's LDS");
    }
    if (timed) HIP_TRY(hipEventRecord(ev[4], fs.stream));
    if (patch) {
      if (timed) {  // the stage timers report orientation and blur as 0: both live inside the descriptor stage
        HIP_TRY(hipEventRecord(ev[3], fs.stream));
        HIP_TRY(hipEventRecord(ev[7], fs.stream));
        HIP_TRY(hipEventRecord(ev[8], fs.stream));
        HIP_TRY(hipEventRecord(ev[5], fs.stream));
      }
      launch_patch_describe(ctx, fs, d_kp, d_desc, capacity, d_n_out);
    } else {
      launch_orient(ctx, fs);  // IC_Angle needs the unblurred levels only: beside the blur, behind the quadtree
      if (timed) HIP_TRY(hipEventRecord(ev[3], fs.stream));
      HIP_TRY(hipStreamWaitEvent(fb.stream, fork_ev, 0));
      if (timed) HIP_TRY(hipEventRecord(ev[7], fb.stream));
      launch_blur(ctx, fb);
      if (timed) HIP_TRY(hipEventRecord(ev[8], fb.stream));
      HIP_TRY(hipEventRecord(sev[1], fb.stream));
      HIP_TRY(hipStreamWaitEvent(fs.stream, sev[1], 0));
      if (timed) HIP_TRY(hipEventRecord(ev[5], fs.stream));
      if (n_slices == 1 && !capture) {  // pyramid, cells and candidates of this call's set have been read for the last time
        hipEvent_t &free_ev = ctx->ev_set_free[ctx->work_set];
        if (!free_ev) HIP_TRY(hipEventCreateWithFlags(&free_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(free_ev, fs.stream));
        ctx->set_state[ctx->work_set] = orbgpu_ctx::SET_EVENT;
      }
      launch_describe(ctx, fs, d_kp, d_desc, capacity, d_n_out);
    }
    if (timed) HIP_TRY(hipEventRecord(ev[6], fs.stream));
    if (s > 0) {  // join
      HIP_TRY(hipEventRecord(sev[2], fs.stream));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, sev[2], 0));
    }
  }
  if (stage_timing) HIP_TRY(hipEventRecord(ev[9], ctx->stream));  // every slice has finished
  HIP_TRY(hipGetLastError());
  ctx->last_frames = n_frames;
  ctx->last_feature_count = feature_count;
  ctx->last_slices = n_slices;
  if (capture) return 0;
  ctx->ev_full[ctx->ev_head] = stage_timing;
  ctx->ev_head = (ctx->ev_head + 1) % ORBGPU_EV_SETS;
  ctx->ev_pending = std::min(ctx->ev_pending + 1, ORBGPU_EV_SETS);
  ctx->last_frames = n_frames;
  ctx->last_feature_count = feature_count;
  ctx->last_slices = n_slices;
  return 0;
}

// waits for the stream, turns the device error flag into a status and reads the stage timers
int finish_pipeline(orbgpu_ctx *ctx) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // every head has been joined into the context's stream: nothing enqueued uses either working set any more
  ctx->set_state[0] = ctx->set_state[1] = orbgpu_ctx::SET_IDLE;
  if (ctx->ev_pending > 0) {
    // stage times: mean over the extract calls enqueued since the last synchronisation (at most ORBGPU_EV_SETS)
    const int n = ctx->ev_pending;
    ctx->ev_pending = 0;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n_full = 0;  // sets that hold every stage boundary (orbgpu_set_stage_timing may have been toggled in between)
    for (int k = 1; k <= n; k++) {
      hipEvent_t *ev = ctx->ev[(ctx->ev_head - k + ORBGPU_EV_SETS) % ORBGPU_EV_SETS];
      acc[1] += elapsed(ev[1], ev[2]);  // FAST score + cells
      if (ctx->ev_full[(ctx->ev_head - k + ORBGPU_EV_SETS) % ORBGPU_EV_SETS]) {
        n_full++;
        acc[0] += elapsed(ev[0], ev[1]);
        acc[6] += elapsed(ev[0], ev[9]);
        acc[2] += elapsed(ev[4], ev[3]);  // orientation, behind the quadtree and beside the blur
        acc[3] += elapsed(ev[2], ev[4]);
        acc[4] += elapsed(ev[7], ev[8]);  // blur, on its own stream (overlaps the quadtree stage)
        acc[5] += elapsed(ev[5], ev[6]);
        if (ctx->early_l0) acc[7] += elapsed(ev[10], ev[11]);  // FAST of level 0, on the second stream beside the pyramid
      }
    }
    for (int i = 0; i < 8; i++) ctx->stage_ms[i] = i == 1 ? acc[i] / (float)n : (n_full ? acc[i] / (float)n_full : 0.f);
  }
  if (*(volatile int *)ctx->h_error_flag != 0) {  // written by the kernels themselves (mapped host memory)
    const int code = *ctx->h_error_flag;
    *ctx->h_error_flag = 0;  // the stream is idle: finish_pipeline has just synchronised it
    if (code == ORBGPU_ERR_OUT_CAPACITY)
      return fail(ctx, ORBGPU_E_CAPACITY, "a frame has more keypoints than the caller's capacity: that frame was not written "
                                          "(its entry of n_out holds the count it needs)");
    if (code == ORBGPU_ERR_SEGMENT_SIZE)
      return fail(ctx, ORBGPU_E_CAPACITY, "a segment or range holds more rows than the max_*_count the call was given: its "
                                          "surplus rows were ignored");
    return fail(ctx, ORBGPU_E_CAPACITY, "quadtree kernel capacity violation (code " + std::to_string(code) + ")");
  }
  return 0;
}

}  // namespace

namespace {
// Host-pointer wrappers stage through ONE persistent pair of buffers per context, grown on demand (no hipMalloc / hipFree
// per call): pinned + mapped host memory and a device buffer of the same size.  put() packs an input array into the
// host buffer and returns the address of its copy in the device buffer; upload() moves everything packed so far with one
// host-to-device copy on the context's stream; regions put(nullptr, n) are outputs.  host_of() / mapped_of() give the
// host twin of a region and the address under which kernels can write that twin directly (small results: no copy back).
// Valid only while the caller holds the context's mutex from reserve() to its last use of the buffers.
int ensure_stage(orbgpu_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->stage_bytes) return 0;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->h_stage) HIP_TRY(hipHostFree(ctx->h_stage));
  if (ctx->d_stage) HIP_TRY(hipFree(ctx->d_stage));
  ctx->h_stage = ctx->h_stage_dev = ctx->d_stage = nullptr;
  ctx->stage_bytes = 0;
  bytes = std::max<size_t>((bytes + (bytes >> 2) + (1u << 20)) & ~(size_t)((1u << 20) - 1), 1u << 20);
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage), bytes, hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_stage_dev), ctx->h_stage, 0));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_stage), bytes));
  ctx->stage_bytes = bytes;
  return 0;
}

struct Staging {
  orbgpu_ctx *ctx;
  size_t used = 0, in_end = 0;
  int reserve(size_t bytes) { return ensure_stage(ctx, bytes + 4096); }
  template <class T>
  T *put(const T *h, size_t n) {
    used = (used + 255) & ~(size_t)255;
    T *d = reinterpret_cast<T *>(ctx->d_stage + used);
    if (h && n) {
      std::memcpy(ctx->h_stage + used, h, n * sizeof(T));
      in_end = used + n * sizeof(T);
    }
    used += n * sizeof(T);
    return d;
  }
  hipError_t upload() {
    return in_end ? hipMemcpyAsync(ctx->d_stage, ctx->h_stage, in_end, hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
  }
  template <class T>
  T *host_of(T *d) const { return reinterpret_cast<T *>(ctx->h_stage + (reinterpret_cast<char *>(d) - ctx->d_stage)); }
  template <class T>
  T *mapped_of(T *d) const { return reinterpret_cast<T *>(ctx->h_stage_dev + (reinterpret_cast<char *>(d) - ctx->d_stage)); }
  // device -> the region's host twin (a pinned copy), enqueued; the caller synchronises and then copies out with fetch()
  template <class T>
  hipError_t download(T *d, size_t n) {
    return n ? hipMemcpyAsync(host_of(d), d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream) : hipSuccess;
  }
  template <class T>
  void fetch(T *dst, T *d, size_t n) const {
    if (n) std::memcpy(dst, host_of(d), n * sizeof(T));
  }
};
}  // namespace

namespace {
// One frame per call, host to host -- how the reference calls Extract (factories/feature_handler_factory.cpp:50).  The eager
// pipeline costs the host ~100 us of launches and event calls for ~17 short dependent kernels, then three small copies
// back; here the whole call (upload of the staged image, every kernel with its forks and joins) is ONE captured hipGraph
// per (feature_count), and the last kernel writes keypoints, descriptors and the count straight into mapped host memory:
// per call one memcpy into pinned memory, one hipGraphLaunch, one stream wait, one memcpy out.
int extract_one_frame_graph(orbgpu_ctx *ctx, const uint8_t *img, int stride, int feature_count, orbgpu_keypoint *kp,
                            uint8_t *desc, int capacity, int32_t *n_out) {
  const LevelGeom &L0 = ctx->geom.lv[0];
  const int w = ctx->params.width, h = ctx->params.height, cap_dev = ctx->out_capacity;
  const size_t kp_bytes = pad256((size_t)cap_dev * sizeof(orbgpu_keypoint)), desc_bytes = pad256((size_t)cap_dev * 32);
  if (!ctx->h_one) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_one), kp_bytes + desc_bytes + 256, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->h_one_dev), ctx->h_one, 0));
  }
  orbgpu_keypoint *m_kp = reinterpret_cast<orbgpu_keypoint *>(ctx->h_one_dev);
  uint8_t *m_desc = reinterpret_cast<uint8_t *>(ctx->h_one_dev + kp_bytes);
  int *m_n = reinterpret_cast<int *>(ctx->h_one_dev + kp_bytes + desc_bytes);
  uint8_t *dst = ctx->h_imgs;
  if (stride == L0.pitch) {
    std::memcpy(dst, img, (size_t)L0.pitch * (h - 1) + w);
  } else {
    for (int y = 0; y < h; y++) {
      std::memcpy(dst + (size_t)y * L0.pitch, img + (size_t)y * stride, (size_t)w);
      std::memset(dst + (size_t)y * L0.pitch + w, 0, (size_t)(L0.pitch - w));
    }
  }
  auto upload = [&]() {
    return hipMemcpyAsync(ctx->d_pyr + L0.off, dst, (size_t)L0.pitch * L0.h, hipMemcpyHostToDevice, ctx->stream);
  };
  ctx->lvl0 = Level0Src{nullptr, 0};
  const int n_slices_saved = ctx->n_slices;
  if (!ctx->graph_one || ctx->graph_one_features != feature_count) {
    if (ctx->graph_one) {
      (void)hipGraphExecDestroy(ctx->graph_one);
      ctx->graph_one = nullptr;
    }
    // one eager pass first: it raises the kernels' LDS limits, sizes lazily created state and surfaces capacity errors
    HIP_TRY(upload());
    ctx->n_slices = 1;
    int rc = run_pipeline(ctx, 1, feature_count, m_kp, m_desc, cap_dev, m_n);
    ctx->n_slices = n_slices_saved;
    if (rc) return rc;
    rc = finish_pipeline(ctx);
    if (rc) return rc;
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    hipError_t e = upload();
    ctx->n_slices = 1;
    rc = e == hipSuccess ? run_pipeline(ctx, 1, feature_count, m_kp, m_desc, cap_dev, m_n, true) : 0;
    ctx->n_slices = n_slices_saved;
    const hipError_t e2 = hipStreamEndCapture(ctx->stream, &graph);  // always ends the capture, also after a failure
    if (rc) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    HIP_TRY(e);
    HIP_TRY(e2);
    const hipError_t e3 = hipGraphInstantiate(&ctx->graph_one, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIP_TRY(e3);
    ctx->graph_one_features = feature_count;
  }
  // (the graph's kernels use the working set of the time of its capture behind no event of the eager path)
  ctx->set_state[0] = ctx->set_state[1] = orbgpu_ctx::SET_BUSY;
  HIP_TRY(hipGraphLaunch(ctx->graph_one, ctx->stream));
  int rc = finish_pipeline(ctx);
  if (rc) return rc;
  const int tot = *reinterpret_cast<volatile int *>(ctx->h_one + kp_bytes + desc_bytes);
  if (tot > cap_dev) return fail(ctx, ORBGPU_E_CAPACITY, "internal keypoint capacity too small: need " + std::to_string(tot));
  if (tot > capacity) return fail(ctx, ORBGPU_E_CAPACITY, "caller capacity too small: need " + std::to_string(tot));
  n_out[0] = tot;
  std::memcpy(kp, ctx->h_one, (size_t)tot * sizeof(orbgpu_keypoint));
  std::memcpy(desc, ctx->h_one + kp_bytes, (size_t)tot * 32);
  return ORBGPU_OK;
}
}  // namespace

extern "C" {

int orbgpu_create(const orbgpu_params *params, orbgpu_ctx **out_ctx) {
  if (!params || !out_ctx) return fail(nullptr, ORBGPU_E_INVALID, "null argument");
  *out_ctx = nullptr;
  if (params->width <= 0 || params->height <= 0) return fail(nullptr, ORBGPU_E_INVALID, "non-positive image size");
  if (params->levels < 1 || params->levels > ORBGPU_MAX_LEVELS) return fail(nullptr, ORBGPU_E_INVALID, "levels out of range");
  if (!(params->scale_factor > 1.0)) return fail(nullptr, ORBGPU_E_INVALID, "scale_factor must be > 1");
  if (params->fast_init < params->fast_min || params->fast_min < 1 || params->fast_init > 255)
    return fail(nullptr, ORBGPU_E_INVALID, "need 1 <= fast_min <= fast_init <= 255");
  if (params->max_batch < 1 || params->max_features < 1) return fail(nullptr, ORBGPU_E_INVALID, "max_batch and max_features must be >= 1");
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0 || params->device < 0 || params->device >= n_dev)
    return fail(nullptr, ORBGPU_E_HIP, "no usable HIP device (this library has no CPU fallback)");
  orbgpu_ctx *ctx = new orbgpu_ctx();
  ctx->params = *params;
  ctx->device = params->device;
  {
    const char *z = orbgpu_tune("no_zero_copy");
    ctx->no_zero_copy = z && z[0] == '1';
    const char *lt = orbgpu_tune("stage_timing");  // 1: start with every stage boundary timed
    ctx->stage_timing = lt && lt[0] == '1';
    const char *ws = orbgpu_tune("window_scan");  // 1: windowed searches by the plain scan, never the grid
    ctx->window_plain_scan = ws && ws[0] == '1';
    const char *wg = orbgpu_tune("window_grid_min");  // tuning / tests: smallest n_from that takes the grid
    if (wg) ctx->window_grid_min_from = std::max(1, std::atoi(wg));
    ctx->window_grid_forced = wg != nullptr;
    const char *ls = orbgpu_tune("link_scan");  // 1: large-capacity frame links by the plain scan, not the grid
    ctx->link_plain_scan = ls && ls[0] == '1';
    const char *pb = orbgpu_tune("patch_blur");  // 0 / 1: full-frame blur / per-keypoint patch blur whatever the size
    ctx->patch_blur_forced = pb ? (pb[0] == '1' ? 1 : 0) : -1;
    const char *ng = orbgpu_tune("no_graph");  // 1: single-frame host calls enqueue eagerly instead of replaying a hipGraph
    ctx->graph_single_frame = !(ng && ng[0] == '1');
    const char *o = orbgpu_tune("slices");  // tuning aid: slices per batch, 0 / unset = automatic
    ctx->n_slices = o ? std::atoi(o) : 0;
  }
  auto bail = [&](int rc) {
    g_create_error = ctx->err;
    orbgpu_destroy(ctx);
    return rc;
  };
  int rc = build_geometry(ctx);
  if (rc) return bail(rc);
  std::vector<XTab> xt;
  std::vector<YTab> yt;
  rc = build_resize_tables(ctx, xt, yt);
  if (rc) return bail(rc);
  const DeviceGeom &g = ctx->geom;
  const size_t B = (size_t)params->max_batch;
  // the quadtree kernel keeps a level's node list in LDS: about 2070 nodes fit the CU's 160 KB.  A context whose
  // max_features budgets more than that to one level (thousands of keypoints on a single level) gets global scratch
  // for the node lists instead
  size_t qt_node_scratch = 0;
  {
    int cap = 0;
    for (int l = 0; l < g.levels; l++) cap = std::max(cap, g.lv[l].kp_cap);
    cap = (cap + 7) & ~7;
    if (cap > 65528) {  // list positions are 16-bit
      fail(ctx, ORBGPU_E_CAPACITY, "max_features budgets more than 65528 keypoints to one pyramid level");
      return bail(ORBGPU_E_CAPACITY);
    }
    if (quadtree_lds_bytes(cap) > ORBGPU_QT_LDS_LIMIT) qt_node_scratch = quadtree_node_bytes(cap) * (size_t)g.levels * B;
  }
  auto alloc_all = [&]() -> int {
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(match_noalloc_init(ctx->device));  // address of this GPU's copy of the fixed scratch array
    HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    // slice / level-0 / blur streams and their events are created by the first extract call that needs them (ensure_slice)
    for (auto &set : ctx->ev)
      for (auto &ev : set) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(dmalloc(&ctx->d_pyr, B * g.frame_bytes + 256));
    HIP_TRY(dmalloc(&ctx->d_blur, B * g.frame_bytes + 256));
    HIP_TRY(hipMemsetAsync(ctx->d_pyr, 0, B * g.frame_bytes + 256, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_blur, 0, B * g.frame_bytes + 256, ctx->stream));
    HIP_TRY(dmalloc(&ctx->d_xtab, xt.size()));
    HIP_TRY(dmalloc(&ctx->d_ytab, yt.size()));
    HIP_TRY(hipMemcpyAsync(ctx->d_xtab, xt.data(), xt.size() * sizeof(XTab), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_ytab, yt.data(), yt.size() * sizeof(YTab), hipMemcpyHostToDevice, ctx->stream));
    {
      std::vector<short> rs;
      std::vector<RsColumn> rc;
      const char *rk = orbgpu_tune("resize");  // ORBGPU_RESIZE=tiles: the LDS-tiled kernel for every level
      const bool tiles = rk && std::string(rk) == "tiles";
      const char *rt = orbgpu_tune("resize_strips");  // tuning aid: strips per frame a level should reach
      // a one-frame context (max_batch == 1: the online call shape) is bound by the latency of the chain, not by
      // throughput: short strips give a level's single frame enough waves (orbgpu_extract 179 -> 162 us host to host)
      const int rs_target = rt ? std::atoi(rt) : (params->max_batch == 1 ? 256 : 32);
      for (int l = 1; l < g.levels; l++) {
        ctx->rs_strip_off[l] = (int)(rs.size() / 4);
        ctx->rs_items[l] = tiles ? 0 : build_resize_strips(g, l, xt.data() + g.lv[l].xtab_off, rs, rs_target);
        ctx->rs_col_off[l] = (int)rc.size();
        if (ctx->rs_items[l] > 0) build_resize_columns(g, l, xt.data() + g.lv[l].xtab_off, rc);
      }
      HIP_TRY(dmalloc(&ctx->d_rs_cols, rc.size()));
      if (!rc.empty()) HIP_TRY(hipMemcpy(ctx->d_rs_cols, rc.data(), rc.size() * sizeof(RsColumn), hipMemcpyHostToDevice));
      HIP_TRY(dmalloc(&ctx->d_rs_strips, rs.size()));
      if (!rs.empty()) HIP_TRY(hipMemcpy(ctx->d_rs_strips, rs.data(), rs.size() * sizeof(short), hipMemcpyHostToDevice));
    }
    {
      // opt-in: small levels at the end of the pyramid built two or three per launch (k_resize_nest): levels below
      // ORBGPU_RESIZE_NEST_MAXPX pixels (default 0: none), or the explicit plan ORBGPU_RESIZE_NEST="src:n,src:n".
      // Measured on the 256-frame VGA batch: levels 5-7 in one launch take as long as their three launches do beside
      // level 0's FAST groups (pyramid stage 0.229 ms either way), so the default stays one launch per level
      for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) ctx->rs_nest_of[l] = -1;
      ctx->n_rs_nest = 0;
      std::vector<std::pair<int, int>> plan;
      if (const char *np = orbgpu_tune("resize_nest")) {
        int a = 0, b = 0, used = 0;
        while (std::sscanf(np, "%d:%d%n", &a, &b, &used) == 2) {
          plan.push_back({a, b});
          np += used;
          if (*np == ',') np++;
        }
      } else {
        const char *mp = orbgpu_tune("resize_nest_maxpx");
        // one-frame contexts nest every level they can (three launches instead of seven: 162 -> 143 us host to host);
        // batches keep one launch per level (beside level 0's FAST groups the nests measured no gain)
        const long long maxpx = mp ? std::atoll(mp) : (params->max_batch == 1 ? LLONG_MAX : 0);
        int s0 = g.levels;
        while (s0 > 1 && (long long)g.lv[s0 - 1].w * g.lv[s0 - 1].h < maxpx) s0--;
        int count = g.levels - s0;
        while (count >= 2 && (int)plan.size() < 4) {
          const int nl = count == 4 ? 2 : std::min(3, count);
          plan.push_back({s0 - 1, nl});
          s0 += nl;
          count -= nl;
        }
      }
      for (auto &pl : plan) {
        if (ctx->n_rs_nest >= 4) break;
        bool clash = pl.first < 0 || pl.second < 2 || pl.first + pl.second > g.levels - 1;
        for (int q = 1; !clash && q <= pl.second; q++) clash = ctx->rs_nest_of[pl.first + q] >= 0;
        if (clash) continue;
        ResizeNest &n = ctx->rs_nest[ctx->n_rs_nest];
        std::vector<short> xr, yr;
        if (!build_resize_nest(g, xt.data(), yt.data(), pl.first, pl.second, n, xr, yr)) continue;
        HIP_TRY(dmalloc(&n.d_xr, xr.size()));
        HIP_TRY(dmalloc(&n.d_yr, yr.size()));
        HIP_TRY(hipMemcpy(n.d_xr, xr.data(), xr.size() * sizeof(short), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(n.d_yr, yr.data(), yr.size() * sizeof(short), hipMemcpyHostToDevice));
        for (int q = 1; q <= pl.second; q++) ctx->rs_nest_of[pl.first + q] = ctx->n_rs_nest;
        ctx->n_rs_nest++;
      }
    }
    {
      std::vector<short> groups;
      ctx->n_fast_groups = build_fast_groups(g, groups);
      for (int l = 0; l < g.levels; l++) ctx->fast_lds_rows = std::max(ctx->fast_lds_rows, g.lv[l].hcell);
      // level 0 needs no pyramid: its FAST groups run on the second stream beside the resize chain (the resize kernels
      // issue 0.59 vector instructions per clock and CU, the FAST kernel is bound by issue: 256-frame 640x480 batch 1.044 ->
      // 0.996 ms).  ORBGPU_EARLY_L0=0 puts every group back into one launch behind the pyramid
      ctx->n_fast_groups_l0 = 0;
      for (size_t q = 0; q < groups.size(); q += 4)
        if (groups[q] == 0) ctx->n_fast_groups_l0++;
      for (int l = 0; l <= g.levels; l++) ctx->fast_group_first[l] = 0;
      for (size_t q = 0; q < groups.size(); q += 4)
        for (int l = groups[q] + 1; l <= g.levels; l++) ctx->fast_group_first[l]++;
      if (const char *el = orbgpu_tune("early_levels")) ctx->early_levels = std::max(1, std::min(3, atoi(el)));  // tuning aid
      const char *nov = orbgpu_tune("no_overlap");
      ctx->no_overlap = nov && nov[0] == '1';
      const char *el0 = orbgpu_tune("early_l0");
      ctx->early_l0 = g.levels >= 2 && !(el0 && el0[0] == '0');
      HIP_TRY(dmalloc(&ctx->d_fast_groups, groups.size()));
      HIP_TRY(hipMemcpy(ctx->d_fast_groups, groups.data(), groups.size() * sizeof(short), hipMemcpyHostToDevice));
    }
    {
      std::vector<short> bs;
      ctx->n_blur_items = build_blur_strips(g, bs);
      HIP_TRY(dmalloc(&ctx->d_blur_strips, bs.size()));
      HIP_TRY(hipMemcpy(ctx->d_blur_strips, bs.data(), bs.size() * sizeof(short), hipMemcpyHostToDevice));
    }
    HIP_TRY(dmalloc(&ctx->d_cells, B * g.cells_per_frame));
    HIP_TRY(dmalloc(&ctx->d_cands, B * g.cands_per_frame));
    HIP_TRY(dmalloc(&ctx->d_cands_packed, B * g.cands_per_frame));
    HIP_TRY(dmalloc(&ctx->d_cands_tmp, B * g.cands_per_frame));
    HIP_TRY(dmalloc(&ctx->d_cell_prefix, B * g.cells_per_frame));
    if (qt_node_scratch) HIP_TRY(dmalloc(&ctx->d_qt_nodes, qt_node_scratch));
    {
      // levels from this many pixels on get the first rounds of their quadtree presorted by several workgroups
      // (k_qt_presort); ORBGPU_QT_PRESORT_MINPX overrides (0: every level, negative: none)
      const char *pm = orbgpu_tune("qt_presort_minpx");
      const long long minpx = pm ? std::atoll(pm) : 400000;
      ctx->qt_ps_levels = 0;
      while (minpx >= 0 && ctx->qt_ps_levels < g.levels &&
             (long long)g.lv[ctx->qt_ps_levels].w * g.lv[ctx->qt_ps_levels].h >= minpx)
        ctx->qt_ps_levels++;
      if (ctx->qt_ps_levels) HIP_TRY(dmalloc(&ctx->d_qt_ps, B * ctx->qt_ps_levels * quadtree_presort_words()));
      // the blur then forks behind the presort launches, beside k_quadtree: forked behind FAST as usual it takes the
      // chip first and the short presort launches wait for it to drain (4K: the scan launch 12 -> 320 us);
      // ORBGPU_PS_BLUR=early restores that order (tuning aid)
      const char *bl = orbgpu_tune("ps_blur");
      ctx->qt_ps_blur_late = !(bl && bl[0] == 'e');
    }
    // the error flag lives in pinned host memory that the device writes directly (only ever in the error case): no copy
    // per call -- a device-to-host copy of one int is a blit kernel of its own, 12 us between two kernels
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_error_flag), sizeof(int), hipHostMallocMapped));
    *ctx->h_error_flag = 0;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->d_error_flag), ctx->h_error_flag, 0));
    HIP_TRY(dmalloc(&ctx->d_kp_xy, B * g.kps_per_frame));
    HIP_TRY(dmalloc(&ctx->d_kp_count, B * g.levels));
    HIP_TRY(dmalloc(&ctx->d_kp_angle, B * g.kps_per_frame));
    ctx->out_capacity = g.kps_per_frame;
    HIP_TRY(dmalloc(&ctx->d_out_kp, B * ctx->out_capacity));
    HIP_TRY(dmalloc(&ctx->d_out_desc, B * ctx->out_capacity * 32));
    HIP_TRY(dmalloc(&ctx->d_out_n, B));
    HIP_TRY(hmalloc(&ctx->h_kp_count, B * g.levels));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // xt/yt are stack-owned
    return 0;
  };
  rc = alloc_all();
  if (rc) return bail(rc);
  rc = ensure_match_scratch(ctx, 1u << 20);
  if (rc) return bail(rc);
  *out_ctx = ctx;
  return ORBGPU_OK;
}

void orbgpu_destroy(orbgpu_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  void *dptrs[] = {ctx->d_pyr, ctx->d_score, ctx->d_blur, ctx->d_xtab, ctx->d_ytab, ctx->d_cells, ctx->d_cands,
                   ctx->d_cands_packed, ctx->d_kp_xy, ctx->d_kp_count, ctx->d_kp_angle,
                   ctx->d_out_kp, ctx->d_out_desc, ctx->d_out_n, ctx->d_match_scratch, ctx->d_cands_tmp, ctx->d_cell_prefix, ctx->d_qt_nodes, ctx->d_qt_ps,
                   ctx->d_fast_groups, ctx->d_blur_strips, ctx->d_rs_strips, ctx->d_rs_cols};
  for (void *p : dptrs)
    if (p) (void)hipFree(p);
  for (auto &n : ctx->rs_nest) {
    if (n.d_xr) (void)hipFree(n.d_xr);
    if (n.d_yr) (void)hipFree(n.d_yr);
  }
  if (ctx->graph_one) (void)hipGraphExecDestroy(ctx->graph_one);
  if (ctx->h_one) (void)hipHostFree(ctx->h_one);
  if (ctx->d_stage) (void)hipFree(ctx->d_stage);
  void *hptrs[] = {ctx->h_kp_count, ctx->h_imgs, ctx->h_out, ctx->h_error_flag, ctx->h_stage};
  for (void *p : hptrs)
    if (p) (void)hipHostFree(p);
  for (auto &set : ctx->ev)
    for (auto &ev : set)
      if (ev) (void)hipEventDestroy(ev);
  release_side_streams(ctx);
  for (hipEvent_t e : ctx->ev_set_free)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_head_done) (void)hipEventDestroy(ctx->ev_head_done);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

const char *orbgpu_last_error(const orbgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int orbgpu_get_geometry(const orbgpu_ctx *ctx, int32_t feature_count, orbgpu_geometry *out) {
  if (!ctx || !out || feature_count < 1) return ORBGPU_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  const DeviceGeom &g = ctx->geom;
  out->levels = g.levels;
  level_budget(ctx->params.scale_factor, g.levels, feature_count, out->budget);
  for (int l = 0; l < g.levels; l++) {
    out->w[l] = g.lv[l].w; out->h[l] = g.lv[l].h;
    out->ncols[l] = g.lv[l].ncols; out->nrows[l] = g.lv[l].nrows;
    out->wcell[l] = g.lv[l].wcell; out->hcell[l] = g.lv[l].hcell;
    out->kp_size[l] = g.lv[l].kp_size;
    out->scale[l] = ctx->scale[l]; out->inv_scale[l] = ctx->inv_scale[l]; out->sigma2[l] = ctx->sigma2[l];
    out->pyramid_pixels += (int64_t)g.lv[l].w * g.lv[l].h;
  }
  out->keypoint_capacity = keypoint_capacity(ctx, feature_count);
  return ORBGPU_OK;
}

void *orbgpu_stream(orbgpu_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int orbgpu_set_stream(orbgpu_ctx *ctx, void *stream) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  // work already enqueued keeps its order: the old stream drains before the switch (this also collects its stage timers
  // and device-side errors)
  int rc = finish_pipeline(ctx);
  ctx->stream = stream ? static_cast<hipStream_t>(stream) : ctx->own_stream;
  return rc;
}

int orbgpu_set_stage_timing(orbgpu_ctx *ctx, int32_t all_stages) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->stage_timing = all_stages != 0;
  return ORBGPU_OK;
}

int orbgpu_set_batch_slices(orbgpu_ctx *ctx, int32_t n_slices) {
  if (!ctx || n_slices < 0 || n_slices > orbgpu_ctx::MAX_SLICES) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->n_slices = n_slices;
  return ORBGPU_OK;
}

int orbgpu_wait_event(orbgpu_ctx *ctx, void *event) {
  if (!ctx || !event) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamWaitEvent(ctx->stream, static_cast<hipEvent_t>(event), 0));
  return ORBGPU_OK;
}

int orbgpu_record_event(orbgpu_ctx *ctx, void *event) {
  if (!ctx || !event) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), ctx->stream));
  return ORBGPU_OK;
}

int orbgpu_synchronize(orbgpu_ctx *ctx) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  return finish_pipeline(ctx);
}

int orbgpu_trim(orbgpu_ctx *ctx) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  const int rc = finish_pipeline(ctx);
  release_side_streams(ctx);
  return rc;
}

int orbgpu_extract_batch_device(orbgpu_ctx *ctx, const uint8_t *d_imgs, int64_t frame_stride, int32_t stride,
                                int32_t n_frames, int32_t feature_count, orbgpu_keypoint *d_kp, uint8_t *d_desc,
                                int32_t capacity, int32_t *d_n_out) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!d_imgs || n_frames <= 0) return fail(ctx, ORBGPU_E_EMPTY, "empty image");
  if (!d_kp || !d_desc || !d_n_out) return fail(ctx, ORBGPU_E_INVALID, "null output");
  if (n_frames > ctx->params.max_batch) return fail(ctx, ORBGPU_E_CAPACITY, "n_frames exceeds max_batch");
  if (feature_count < 1 || feature_count > ctx->params.max_features)
    return fail(ctx, ORBGPU_E_INVALID, "feature_count outside [1, max_features]");
  if (stride < ctx->params.width) return fail(ctx, ORBGPU_E_INVALID, "stride smaller than width");
  if (capacity < 1) return fail(ctx, ORBGPU_E_CAPACITY, "capacity must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  // zero copy: when the caller's rows are laid out like level 0 of the slab (row stride = width rounded up to 16, 16-byte
  // aligned base and frame stride) every kernel reads level 0 where it is; otherwise it is copied into the slab
  const LevelGeom &L0 = ctx->geom.lv[0];
  const bool in_place = stride == L0.pitch && frame_stride >= (int64_t)L0.pitch * L0.h &&
                        ((reinterpret_cast<uintptr_t>(d_imgs) | (uintptr_t)frame_stride) & 15) == 0 && !ctx->no_zero_copy;
  // the caller's input event (orbgpu_set_input_ready): the context's stream waits for it whatever else happens -- a head
  // on its own stream waits for it by itself
  if (ctx->inputs_ready && ctx->input_event) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->input_event, 0));
  if (in_place) {
    ctx->lvl0 = Level0Src{d_imgs, (long long)frame_stride};
  } else {
    ctx->lvl0 = Level0Src{nullptr, 0};
    launch_import_images(ctx, d_imgs, frame_stride, stride, n_frames);
  }
  ctx->call_piped = ctx->inputs_ready && in_place;  // (an import copy runs on the context's stream)
  const int rc = run_pipeline(ctx, n_frames, feature_count, d_kp, d_desc, capacity, d_n_out);
  ctx->call_piped = false;
  return rc;
}

int orbgpu_set_input_ready(orbgpu_ctx *ctx, int32_t ready, void *event) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  ctx->inputs_ready = ready != 0;
  ctx->input_event = ready ? static_cast<hipEvent_t>(event) : nullptr;
  return ORBGPU_OK;
}

int orbgpu_extract_batch(orbgpu_ctx *ctx, const uint8_t *const *imgs, int32_t n_frames, int32_t stride,
                         int32_t feature_count, orbgpu_keypoint *kp, uint8_t *desc, int32_t capacity, int32_t *n_out) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!imgs || n_frames <= 0) return fail(ctx, ORBGPU_E_EMPTY, "empty image");
  for (int f = 0; f < n_frames; f++)
    if (!imgs[f]) return fail(ctx, ORBGPU_E_EMPTY, "empty image");
  if (!kp || !desc || !n_out) return fail(ctx, ORBGPU_E_INVALID, "null output");
  if (n_frames > ctx->params.max_batch) return fail(ctx, ORBGPU_E_CAPACITY, "n_frames exceeds max_batch");
  if (feature_count < 1 || feature_count > ctx->params.max_features)
    return fail(ctx, ORBGPU_E_INVALID, "feature_count outside [1, max_features]");
  const int w = ctx->params.width, h = ctx->params.height;
  if (stride < w) return fail(ctx, ORBGPU_E_INVALID, "stride smaller than width");
  HIP_TRY(hipSetDevice(ctx->device));
  const LevelGeom &L0 = ctx->geom.lv[0];
  if (!ctx->h_imgs) HIP_TRY(hmalloc(&ctx->h_imgs, (size_t)ctx->params.max_batch * L0.pitch * L0.h));
  if (n_frames == 1 && ctx->graph_single_frame) return extract_one_frame_graph(ctx, imgs[0], stride, feature_count, kp, desc, capacity, n_out);
  // host -> pinned (pitch layout) -> level 0 of each frame slab.  The row copies of a batch are spread over a few host
  // threads (a frame's upload is enqueued by the thread that staged it; the kernels are enqueued after the join)
  auto stage_frame = [&](int f) -> hipError_t {
    uint8_t *dst = ctx->h_imgs + (size_t)f * L0.pitch * L0.h;
    if (stride == L0.pitch) {
      std::memcpy(dst, imgs[f], (size_t)L0.pitch * (h - 1) + w);
    } else {
      for (int y = 0; y < h; y++) {
        std::memcpy(dst + (size_t)y * L0.pitch, imgs[f] + (size_t)y * stride, (size_t)w);
        std::memset(dst + (size_t)y * L0.pitch + w, 0, (size_t)(L0.pitch - w));
      }
    }
    return hipMemcpyAsync(ctx->d_pyr + (size_t)f * ctx->geom.frame_bytes + L0.off, dst, (size_t)L0.pitch * L0.h,
                          hipMemcpyHostToDevice, ctx->stream);
  };
  const int n_thr = n_frames >= 8 ? std::min(std::min(8, n_frames / 4), (int)std::max(1u, std::thread::hardware_concurrency())) : 1;
  if (n_thr <= 1) {
    for (int f = 0; f < n_frames; f++) HIP_TRY(stage_frame(f));
  } else {
    std::vector<hipError_t> err((size_t)n_thr, hipSuccess);
    std::vector<std::thread> pool;
    for (int t = 0; t < n_thr; t++)
      pool.emplace_back([&, t] {
        hipError_t e = hipSetDevice(ctx->device);
        for (int f = t; f < n_frames && e == hipSuccess; f += n_thr) e = stage_frame(f);
        err[(size_t)t] = e;
      });
    for (auto &th : pool) th.join();
    for (hipError_t e : err) HIP_TRY(e);
  }
  ctx->lvl0 = Level0Src{nullptr, 0};
  const int cap_dev = ctx->out_capacity;
  int rc = run_pipeline(ctx, n_frames, feature_count, ctx->d_out_kp, ctx->d_out_desc, cap_dev, ctx->d_out_n);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->h_kp_count, ctx->d_out_n, sizeof(int) * n_frames, hipMemcpyDeviceToHost, ctx->stream));
  rc = finish_pipeline(ctx);
  if (rc) return rc;

  for (int f = 0; f < n_frames; f++) {
    const int tot = ctx->h_kp_count[f];
    if (tot > cap_dev) return fail(ctx, ORBGPU_E_CAPACITY, "internal keypoint capacity too small: need " + std::to_string(tot));
    if (tot > capacity) return fail(ctx, ORBGPU_E_CAPACITY, "caller capacity too small: need " + std::to_string(tot));
    n_out[f] = tot;
  }
  if (n_thr <= 1) {
    for (int f = 0; f < n_frames; f++) {
      const int tot = ctx->h_kp_count[f];
      if (!tot) continue;
      HIP_TRY(hipMemcpyAsync(kp + (size_t)f * capacity, ctx->d_out_kp + (size_t)f * cap_dev, sizeof(orbgpu_keypoint) * tot,
                             hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipMemcpyAsync(desc + (size_t)f * capacity * 32, ctx->d_out_desc + (size_t)f * cap_dev * 32, (size_t)tot * 32,
                             hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ORBGPU_OK;
  }
  // batch: two large downloads into pinned memory, then the per-frame pieces are copied out by the same few threads
  const size_t kp_bytes = (size_t)n_frames * cap_dev * sizeof(orbgpu_keypoint), desc_bytes = (size_t)n_frames * cap_dev * 32;
  if (!ctx->h_out) HIP_TRY(hmalloc(&ctx->h_out, (size_t)ctx->params.max_batch * cap_dev * (sizeof(orbgpu_keypoint) + 32)));
  uint8_t *h_desc = ctx->h_out + (size_t)ctx->params.max_batch * cap_dev * sizeof(orbgpu_keypoint);
  HIP_TRY(hipMemcpyAsync(ctx->h_out, ctx->d_out_kp, kp_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(h_desc, ctx->d_out_desc, desc_bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  {
    std::vector<std::thread> pool;
    for (int t = 0; t < n_thr; t++)
      pool.emplace_back([&, t] {
        for (int f = t; f < n_frames; f += n_thr) {
          const size_t tot = (size_t)ctx->h_kp_count[f];
          std::memcpy(kp + (size_t)f * capacity, ctx->h_out + (size_t)f * cap_dev * sizeof(orbgpu_keypoint), tot * sizeof(orbgpu_keypoint));
          std::memcpy(desc + (size_t)f * capacity * 32, h_desc + (size_t)f * cap_dev * 32, tot * 32);
        }
      });
    for (auto &th : pool) th.join();
  }
  return ORBGPU_OK;
}

int orbgpu_extract(orbgpu_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride,
                   int32_t feature_count, orbgpu_keypoint *kp, uint8_t *desc, int32_t capacity, int32_t *n_out) {
  if (!ctx) return ORBGPU_E_INVALID;
  if (!img || width <= 0 || height <= 0) return fail(ctx, ORBGPU_E_EMPTY, "empty image");
  if (width != ctx->params.width || height != ctx->params.height)
    return fail(ctx, ORBGPU_E_INVALID, "image size differs from the size the context was created for");
  const uint8_t *one[1] = {img};
  return orbgpu_extract_batch(ctx, one, 1, stride, feature_count, kp, desc, capacity, n_out);
}

int orbgpu_debug_dump(orbgpu_ctx *ctx, int32_t frame, uint8_t *pyramid, uint8_t *score, uint8_t *blurred,
                      int32_t *cand_x, int32_t *cand_y, int32_t *cand_score, int32_t cand_capacity,
                      int32_t *n_cand_per_level) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (frame < 0 || frame >= ctx->last_frames) return fail(ctx, ORBGPU_E_INVALID, "frame outside the last batch");
  const DeviceGeom &g = ctx->geom;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (score) {
    // the product path keeps no score map: re-run the fused kernel on the (still resident) pyramid of that frame with
    // its DUMP switch, which scatters every non-zero score into a one-frame slab (allocated by the first such dump)
    if (!ctx->d_score) HIP_TRY(dmalloc(&ctx->d_score, (size_t)g.frame_bytes + 256));
    HIP_TRY(hipMemsetAsync(ctx->d_score, 0, g.frame_bytes, ctx->stream));
    launch_fast_cells(ctx, FrameSlice{frame, 1, ctx->stream}, ctx->d_score, 0, ctx->geom.levels, false);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (blurred && ctx->last_patch_mode) {
    // the last call blurred per keypoint: the full-frame blur of that frame is produced here, for the stage tests
    launch_blur(ctx, FrameSlice{frame, 1, ctx->stream});
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  std::vector<uint8_t> slab(g.frame_bytes);
  const uint8_t *srcs[3] = {ctx->d_pyr + (size_t)frame * g.frame_bytes, ctx->d_score, ctx->d_blur + (size_t)frame * g.frame_bytes};
  uint8_t *dsts[3] = {pyramid, score, blurred};
  for (int k = 0; k < 3; k++) {
    if (!dsts[k]) continue;
    HIP_TRY(hipMemcpy(slab.data(), srcs[k], g.frame_bytes, hipMemcpyDeviceToHost));
    if (k == 0 && ctx->lvl0.base)  // level 0 was read in place
      HIP_TRY(hipMemcpy(slab.data() + g.lv[0].off, ctx->lvl0.base + (size_t)frame * (size_t)ctx->lvl0.frame_stride,
                        (size_t)g.lv[0].pitch * g.lv[0].h, hipMemcpyDeviceToHost));
    size_t o = 0;
    for (int l = 0; l < g.levels; l++) {
      const LevelGeom &L = g.lv[l];
      for (int y = 0; y < L.h; y++) std::memcpy(dsts[k] + o + (size_t)y * L.w, slab.data() + L.off + (size_t)y * L.pitch, (size_t)L.w);
      o += (size_t)L.w * L.h;
    }
  }
  if (n_cand_per_level || cand_x) {
    // candidates of the frame: cell table + sub-pools straight from the device, walked in reference order
    std::vector<CellEntry> cells_h((size_t)g.cells_per_frame);
    std::vector<unsigned> pool((size_t)g.cands_per_frame);
    HIP_TRY(hipMemcpy(cells_h.data(), ctx->d_cells + (size_t)frame * g.cells_per_frame, sizeof(CellEntry) * cells_h.size(),
                      hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pool.data(), ctx->d_cands + (size_t)frame * g.cands_per_frame, sizeof(unsigned) * pool.size(),
                      hipMemcpyDeviceToHost));
    int n = 0;
    for (int l = 0; l < g.levels; l++) {
      const LevelGeom &L = g.lv[l];
      const unsigned *c = pool.data() + L.cand_off;
      const CellEntry *cells = cells_h.data() + L.cell_off;
      int nl = 0;
      for (int ci = 0; ci < L.ncols * L.nrows; ci++)
        for (int k = 0; k < cells[ci].count; k++) {
          const unsigned v = c[cells[ci].off + k];
          if (cand_x && n < cand_capacity) {
            cand_x[n] = (int)(v & 0xfffu);
            cand_y[n] = (int)((v >> 12) & 0xfffu);
            cand_score[n] = (int)(v >> 24);
          }
          n++;
          nl++;
        }
      if (n_cand_per_level) n_cand_per_level[l] = nl;
    }
    if (cand_x && n > cand_capacity) return fail(ctx, ORBGPU_E_CAPACITY, "cand_capacity too small: need " + std::to_string(n));
  }
  return ORBGPU_OK;
}

// ---- matching ----------------------------------------------------------------------------------

static bool window_grid_ok(const orbgpu_ctx *ctx, const orbgpu_window_args &a) {
  return a.to_xy && a.n_to >= 64 && a.n_from >= ctx->window_grid_min_from && a.n_to < (1 << 20) && a.n_from < (1 << 20) &&
         !ctx->window_plain_scan;
}
// one small search: the wave-per-64x64-block kernels, unless a test forces the grid or the exploratory MFMA engine is on
static bool takes_small_path(const orbgpu_ctx *ctx, const orbgpu_window_args &a) {
  return match_small_fits(a.n_to, a.n_from) && !(window_grid_ok(ctx, a) && ctx->window_grid_forced) &&
         !(ctx->bruteforce_mfma && !a.to_xy);
}

static int match_device_locked(orbgpu_ctx *ctx, const orbgpu_window_args &a, int32_t *d_from_idx, uint32_t *d_dist) {
  if (a.n_to < 0 || a.n_from < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (a.n_from >= (1 << 23)) return fail(ctx, ORBGPU_E_INVALID, "n_from must be below 2^23");
  if (!(a.ratio >= 0.0)) return fail(ctx, ORBGPU_E_INVALID, "ratio must be >= 0");
  if (a.n_to == 0) return ORBGPU_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (a.n_from == 0) {
    HIP_TRY(hipMemsetAsync(d_from_idx, 0xff, sizeof(int32_t) * (size_t)a.n_to, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_dist, 0, sizeof(uint32_t) * (size_t)a.n_to, ctx->stream));
    return ORBGPU_OK;
  }
  const bool grid_ok = window_grid_ok(ctx, a);
  if (takes_small_path(ctx, a)) {
    // one small search (the per-frame call of the reference's matchers): a wave per 64 x 64 block of pairs
    int rc = ensure_match_scratch(ctx, match_small_scratch_bytes(a.n_to, a.n_from));
    if (rc) return rc;
    launch_match_small(ctx, a, d_from_idx, reinterpret_cast<unsigned *>(d_dist), ctx->d_match_scratch);
    HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
  }
  if (grid_ok) {
    // windowed search among many 'from' rows (below ~2000 a single search has too few wavefronts for the grid to pay): over the 64 x 48 search grid laid over the context's image (one segment, the same
    // kernels as orbgpu_match_window_grid_segments_device; identical results, see there)
    int rc = ensure_match_scratch(ctx, grid_match_scratch_bytes(1, a.n_to, a.n_from, a.n_from));
    if (rc) return rc;
    const orbgpu_segment one{0, a.n_to, 0, a.n_from};
    launch_match_grid_segments(ctx, a, nullptr, 1, a.n_to, a.n_from, a.n_from, (double)ctx->params.width,
                               (double)ctx->params.height, d_from_idx, reinterpret_cast<unsigned *>(d_dist),
                               ctx->d_match_scratch, &one);
    HIP_TRY(hipGetLastError());
    return ORBGPU_OK;
  }
  int rc = ensure_match_scratch(ctx, match_scratch_size(a.n_to, a.n_from));
  if (rc) return rc;
  launch_match(ctx, a, d_from_idx, d_dist, ctx->d_match_scratch, 0x7fffffff, ctx->bruteforce_mfma);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_bruteforce_device(orbgpu_ctx *ctx, const uint8_t *d_to_desc, int32_t n_to, const uint8_t *d_from_desc,
                                   int32_t n_from, uint32_t threshold, double ratio, int32_t *d_from_idx_per_to,
                                   uint32_t *d_dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if ((n_to > 0 && (!d_to_desc || !d_from_idx_per_to || !d_dist)) || (n_from > 0 && !d_from_desc))
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  orbgpu_window_args a{};
  a.to_desc = d_to_desc; a.n_to = n_to; a.from_desc = d_from_desc; a.n_from = n_from;
  a.threshold = threshold; a.ratio = ratio;
  return match_device_locked(ctx, a, d_from_idx_per_to, d_dist);
}

int orbgpu_set_bruteforce_engine(orbgpu_ctx *ctx, int32_t engine) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (engine != ORBGPU_ENGINE_POPCOUNT && engine != ORBGPU_ENGINE_MFMA_I8)
    return fail(ctx, ORBGPU_E_INVALID, "unknown brute-force engine");
  ctx->bruteforce_mfma = engine == ORBGPU_ENGINE_MFMA_I8;
  return ORBGPU_OK;
}

int orbgpu_match_bruteforce_noalloc_device(orbgpu_ctx *ctx, const uint8_t *d_to_desc, int32_t n_to,
                                           const uint8_t *d_from_desc, int32_t n_from, uint32_t threshold, double ratio,
                                           int32_t *d_from_idx_per_to, uint32_t *d_dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (n_to < 0 || n_from < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if ((n_to > 0 && (!d_to_desc || !d_from_idx_per_to || !d_dist)) || (n_from > 0 && !d_from_desc))
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (n_from >= (1 << 23)) return fail(ctx, ORBGPU_E_INVALID, "n_from must be below 2^23");
  if (!(ratio >= 0.0)) return fail(ctx, ORBGPU_E_INVALID, "ratio must be >= 0");
  if (n_to == 0) return ORBGPU_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (n_from == 0) {
    HIP_TRY(hipMemsetAsync(d_from_idx_per_to, 0xff, sizeof(int32_t) * (size_t)n_to, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_dist, 0, sizeof(uint32_t) * (size_t)n_to, ctx->stream));
    return ORBGPU_OK;
  }
  const int max_chunks = match_noalloc_chunks(n_to, n_from);
  if (max_chunks < 1)
    return fail(ctx, ORBGPU_E_CAPACITY, "orbgpu_match_bruteforce_noalloc_device: 8 * (n_from + n_to) bytes exceed the fixed " +
                                            std::to_string(ORBGPU_NOALLOC_SCRATCH_BYTES >> 20) + " MiB scratch");
  void *scratch = match_noalloc_ptr(ctx->device);
  if (!scratch) return fail(ctx, ORBGPU_E_HIP, "fixed scratch array was not resolved on this device");
  orbgpu_window_args a{};
  a.to_desc = d_to_desc; a.n_to = n_to; a.from_desc = d_from_desc; a.n_from = n_from;
  a.threshold = threshold; a.ratio = ratio;
  launch_match(ctx, a, d_from_idx_per_to, reinterpret_cast<unsigned *>(d_dist), scratch, max_chunks);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_window_device(orbgpu_ctx *ctx, const orbgpu_window_args *a, int32_t *d_from_idx_per_to, uint32_t *d_dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_to > 0 && (!a->to_desc || !a->to_xy || !a->to_window || !a->to_min_level || !a->to_max_level ||
                      !d_from_idx_per_to || !d_dist))
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_from > 0 && (!a->from_desc || !a->from_xy || !a->from_level)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  return match_device_locked(ctx, *a, d_from_idx_per_to, d_dist);
}


int orbgpu_match_window_segments_device(orbgpu_ctx *ctx, const orbgpu_window_args *a, const orbgpu_segment *d_segments,
                                        int32_t n_segments, int32_t max_to_count, int32_t *d_from_idx_per_to,
                                        uint32_t *d_dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (n_segments < 0 || max_to_count < 0 || a->n_to < 0 || a->n_from < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n_segments == 0 || max_to_count == 0) return ORBGPU_OK;
  if (!d_segments || !a->to_desc || !a->to_xy || !a->to_window || !a->to_min_level || !a->to_max_level || !d_from_idx_per_to ||
      !d_dist)
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_from > 0 && (!a->from_desc || !a->from_xy || !a->from_level)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_from >= (1 << 23)) return fail(ctx, ORBGPU_E_INVALID, "a segment's from_count must be below 2^23");
  if (!(a->ratio >= 0.0)) return fail(ctx, ORBGPU_E_INVALID, "ratio must be >= 0");
  if (n_segments > 65535) return fail(ctx, ORBGPU_E_INVALID, "more than 65535 segments in one call");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_match_scratch(ctx, (size_t)std::max(a->n_from, 1) * sizeof(unsigned long long) + 256);
  if (rc) return rc;
  launch_match_segments(ctx, *a, d_segments, n_segments, max_to_count, a->n_from, d_from_idx_per_to,
                        reinterpret_cast<unsigned *>(d_dist), ctx->d_match_scratch);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_grid_build_device(orbgpu_ctx *ctx, const double *d_xy, const int32_t *d_ranges, int32_t n_ranges,
                             int32_t max_count, double image_width, double image_height, int32_t *d_cell_start,
                             int32_t *d_items) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (n_ranges < 0 || max_count < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (!(image_width > 0.0) || !(image_height > 0.0)) return fail(ctx, ORBGPU_E_INVALID, "image size must be positive");
  if (max_count >= (1 << 20)) return fail(ctx, ORBGPU_E_INVALID, "a grid range must hold fewer than 2^20 rows");
  if (n_ranges == 0) return ORBGPU_OK;
  if (!d_ranges || !d_cell_start || (max_count > 0 && (!d_xy || !d_items))) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_match_scratch(ctx, grid_build_scratch_bytes(n_ranges, std::max(max_count, 1)));
  if (rc) return rc;
  launch_grid_build(ctx, d_xy, d_ranges, n_ranges, max_count, image_width, image_height, d_cell_start, d_items,
                    ctx->d_match_scratch);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_window_grid_segments_device(orbgpu_ctx *ctx, const orbgpu_window_args *a, const orbgpu_segment *d_segments,
                                             int32_t n_segments, int32_t max_to_count, int32_t max_from_count,
                                             double image_width, double image_height, int32_t *d_from_idx_per_to,
                                             uint32_t *d_dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (n_segments < 0 || max_to_count < 0 || max_from_count < 0 || a->n_to < 0 || a->n_from < 0)
    return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (!(image_width > 0.0) || !(image_height > 0.0)) return fail(ctx, ORBGPU_E_INVALID, "image size must be positive");
  if (n_segments == 0 || max_to_count == 0) return ORBGPU_OK;
  if (!d_segments || !a->to_desc || !a->to_xy || !a->to_window || !a->to_min_level || !a->to_max_level || !d_from_idx_per_to ||
      !d_dist)
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_from > 0 && (!a->from_desc || !a->from_xy || !a->from_level)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_from >= (1 << 23)) return fail(ctx, ORBGPU_E_INVALID, "a segment's from_count must be below 2^23");
  if (max_to_count >= (1 << 20) || max_from_count >= (1 << 20))
    return fail(ctx, ORBGPU_E_INVALID, "a grid range must hold fewer than 2^20 rows");
  if (!(a->ratio >= 0.0)) return fail(ctx, ORBGPU_E_INVALID, "ratio must be >= 0");
  if (n_segments > 65535) return fail(ctx, ORBGPU_E_INVALID, "more than 65535 segments in one call");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_match_scratch(ctx, grid_match_scratch_bytes(n_segments, max_to_count, std::max(max_from_count, 1),
                                                              std::max(a->n_from, 1)));
  if (rc) return rc;
  launch_match_grid_segments(ctx, *a, d_segments, n_segments, max_to_count, std::max(max_from_count, 1), a->n_from,
                             image_width, image_height, d_from_idx_per_to, reinterpret_cast<unsigned *>(d_dist),
                             ctx->d_match_scratch);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_window(orbgpu_ctx *ctx, const orbgpu_window_args *a, int32_t *from_idx_per_to, uint32_t *dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  const bool window = a->to_xy != nullptr;
  if (a->n_to < 0 || a->n_from < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (a->n_to > 0 && (!a->to_desc || !from_idx_per_to || !dist)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_from > 0 && !a->from_desc) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (window && a->n_to > 0 && (!a->to_window || !a->to_min_level || !a->to_max_level)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (window && a->n_from > 0 && (!a->from_xy || !a->from_level)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_to == 0) return ORBGPU_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t nt = (size_t)a->n_to, nf = (size_t)a->n_from;
  Staging st{ctx};
  int rc = st.reserve(pad256(nt * 32) + pad256(nf * 32) + pad256(nt * 16) + pad256(nt * 8) + 2 * pad256(nt * 4) + pad256(nt) +
                      pad256(nf * 16) + pad256(nf * 4) + 2 * pad256(nt * 4) + 4096);
  if (rc) return rc;
  orbgpu_window_args d = *a;
  d.to_desc = st.put(a->to_desc, nt * 32);
  d.from_desc = st.put(a->from_desc, nf * 32);
  if (window) {
    d.to_xy = st.put(a->to_xy, nt * 2);
    d.to_window = st.put(a->to_window, nt);
    d.to_min_level = st.put(a->to_min_level, nt);
    d.to_max_level = st.put(a->to_max_level, nt);
    d.from_xy = st.put(a->from_xy, nf * 2);
    d.from_level = st.put(a->from_level, nf);
  }
  d.to_active = a->to_active ? st.put(a->to_active, nt) : nullptr;
  int32_t *d_idx = st.put<int32_t>(nullptr, nt);
  uint32_t *d_dist = st.put<uint32_t>(nullptr, nt);
  HIP_TRY(st.upload());
  // small results (a few KB): the last kernel writes them straight into the mapped host twin, no copy back
  const bool direct = nt * 8 <= (64u << 10) && takes_small_path(ctx, d);
  rc = match_device_locked(ctx, d, direct ? st.mapped_of(d_idx) : d_idx, direct ? st.mapped_of(d_dist) : d_dist);
  if (rc) return rc;
  if (!direct) {
    HIP_TRY(st.download(d_idx, nt));
    HIP_TRY(st.download(d_dist, nt));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  st.fetch(from_idx_per_to, d_idx, nt);
  st.fetch(dist, d_dist, nt);
  return ORBGPU_OK;
}

int orbgpu_match_bruteforce(orbgpu_ctx *ctx, const uint8_t *to_desc, int32_t n_to, const uint8_t *from_desc, int32_t n_from,
                            uint32_t threshold, double ratio, int32_t *from_idx_per_to, uint32_t *dist) {
  orbgpu_window_args a{};
  a.to_desc = to_desc; a.n_to = n_to; a.from_desc = from_desc; a.n_from = n_from;
  a.threshold = threshold; a.ratio = ratio;
  return orbgpu_match_window(ctx, &a, from_idx_per_to, dist);
}

int orbgpu_match_frames_device(orbgpu_ctx *ctx, const orbgpu_keypoint *d_kp, const uint8_t *d_desc, const int32_t *d_n,
                               int32_t capacity, const int32_t *d_to_frame, const int32_t *d_from_frame, int32_t n_pairs,
                               uint64_t window, uint32_t threshold, double ratio, int32_t check_orientation,
                               int32_t *d_match, uint32_t *d_dist, int32_t *d_n_matches) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (n_pairs < 0 || capacity < 1) return fail(ctx, ORBGPU_E_INVALID, "bad size");
  if (n_pairs == 0) return ORBGPU_OK;
  if (!d_kp || !d_desc || !d_n || !d_to_frame || !d_from_frame || !d_match || !d_dist || !d_n_matches)
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (capacity >= (1 << 23)) return fail(ctx, ORBGPU_E_INVALID, "capacity must be below 2^23");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_match_scratch(ctx, link_scratch_bytes(n_pairs, capacity));
  if (rc) return rc;
  launch_match_frames(ctx, d_kp, d_desc, d_n, capacity, d_to_frame, d_from_frame, n_pairs, window, threshold, ratio,
                      check_orientation, d_match, d_dist, d_n_matches);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_groups_device(orbgpu_ctx *ctx, const orbgpu_group_args *a, int32_t n_slots, int32_t *d_from_idx_per_to,
                               uint32_t *d_dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_to < 0 || a->n_from < 0 || a->n_groups < 0 || n_slots < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (a->n_to == 0) return ORBGPU_OK;
  if (!d_from_idx_per_to || !d_dist) return fail(ctx, ORBGPU_E_INVALID, "null output");
  if (n_slots > 0 && (!a->to_desc || !a->from_desc || !a->to_off || !a->to_ids || !a->from_off || !a->from_ids || a->n_groups == 0))
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc = ensure_match_scratch(ctx, group_scratch_bytes(a->n_from, n_slots));
  if (rc) return rc;
  launch_match_groups(ctx, *a, n_slots, d_from_idx_per_to, d_dist, ctx->d_match_scratch);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_groups(orbgpu_ctx *ctx, const orbgpu_group_args *a, int32_t *from_idx_per_to, uint32_t *dist) {
  if (!ctx) return ORBGPU_E_INVALID;
  int n_slots = 0, n_cand = 0;
  orbgpu_group_args d{};
  std::lock_guard<std::recursive_mutex> hold(ctx->mu);  // the staging buffers are the context's: held to the last copy
  Staging st{ctx};
  int32_t *d_idx = nullptr;
  uint32_t *d_dist = nullptr;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
    if (a->n_to < 0 || a->n_from < 0 || a->n_groups < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
    if (a->n_to == 0) return ORBGPU_OK;
    if (!a->to_desc || !from_idx_per_to || !dist) return fail(ctx, ORBGPU_E_INVALID, "null argument");
    if (a->n_groups > 0) {
      if (!a->to_off || !a->to_ids || !a->from_off || !a->from_ids || (a->n_from > 0 && !a->from_desc))
        return fail(ctx, ORBGPU_E_INVALID, "null argument");
      if (a->to_off[0] != 0 || a->from_off[0] != 0) return fail(ctx, ORBGPU_E_INVALID, "offsets must start at 0");
      for (int g = 0; g < a->n_groups; g++)
        if (a->to_off[g + 1] < a->to_off[g] || a->from_off[g + 1] < a->from_off[g])
          return fail(ctx, ORBGPU_E_INVALID, "offsets must be non-decreasing");
      n_slots = a->to_off[a->n_groups];
      n_cand = a->from_off[a->n_groups];
      for (int s = 0; s < n_slots; s++)
        if (a->to_ids[s] >= (uint32_t)a->n_to) return fail(ctx, ORBGPU_E_INVALID, "'to' id outside [0, n_to)");
      for (int c = 0; c < n_cand; c++)
        if (a->from_ids[c] >= (uint32_t)a->n_from) return fail(ctx, ORBGPU_E_INVALID, "'from' id outside [0, n_from)");
      if (n_cand >= (1 << 23)) return fail(ctx, ORBGPU_E_INVALID, "too many candidates");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nt = (size_t)a->n_to, nf = (size_t)a->n_from, ng = (size_t)a->n_groups;
    int rc = st.reserve(pad256(nt * 32) + pad256(nf * 32) + 2 * pad256((ng + 1) * 4) + pad256((size_t)n_slots * 4) +
                        pad256((size_t)n_cand * 4) + 2 * pad256(nt * 4) + 4096);
    if (rc) return rc;
    d = *a;
    d.to_desc = st.put(a->to_desc, nt * 32);
    d.from_desc = nf ? st.put(a->from_desc, nf * 32) : nullptr;
    if (ng) {
      d.to_off = st.put(a->to_off, ng + 1);
      d.from_off = st.put(a->from_off, ng + 1);
      d.to_ids = st.put(a->to_ids, (size_t)std::max(n_slots, 1));
      d.from_ids = st.put(a->from_ids, (size_t)std::max(n_cand, 1));
    }
    d_idx = st.put<int32_t>(nullptr, nt);
    d_dist = st.put<uint32_t>(nullptr, nt);
    HIP_TRY(st.upload());
  }
  int rc = orbgpu_match_groups_device(ctx, &d, n_slots, d_idx, d_dist);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipMemcpyAsync(from_idx_per_to, d_idx, (size_t)a->n_to * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dist, d_dist, (size_t)a->n_to * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ORBGPU_OK;
}

static int check_camera(orbgpu_ctx *ctx, const orbgpu_camera *cam) {
  if (!cam) return fail(ctx, ORBGPU_E_INVALID, "null camera");
  if (cam->model != ORBGPU_DISTORTION_NONE && cam->model != ORBGPU_DISTORTION_BARREL5 && cam->model != ORBGPU_DISTORTION_FISHEYE)
    return fail(ctx, ORBGPU_E_INVALID, "unsupported distortion model (BARREL8 undistortion is not implemented in the reference either)");
  return 0;
}

int orbgpu_keypoint_geometry_device(orbgpu_ctx *ctx, const orbgpu_camera *camera, const orbgpu_keypoint *d_kp,
                                    const int32_t *d_n, int32_t n_frames, int32_t capacity, double *d_undistorted,
                                    double *d_unprojected, uint8_t *d_ok) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  int rc = check_camera(ctx, camera);
  if (rc) return rc;
  if (n_frames < 0 || capacity < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n_frames == 0 || capacity == 0) return ORBGPU_OK;
  if (!d_kp || !d_n || !d_undistorted || !d_unprojected || !d_ok) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  launch_keypoint_geometry(ctx, *camera, d_kp, d_n, n_frames, capacity, d_undistorted, d_unprojected, d_ok);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_keypoint_geometry(orbgpu_ctx *ctx, const orbgpu_camera *camera, const orbgpu_keypoint *kp, int32_t n,
                             double *undistorted, double *unprojected, uint8_t *ok) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  int rc = check_camera(ctx, camera);
  if (rc) return rc;
  if (n < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n == 0) return ORBGPU_OK;
  if (!kp || !undistorted || !unprojected || !ok) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  Staging st{ctx};
  rc = st.reserve(pad256((size_t)n * sizeof(orbgpu_keypoint)) + pad256((size_t)n * 16) + pad256((size_t)n * 24) + pad256((size_t)n) + 1024);
  if (rc) return rc;
  orbgpu_keypoint *d_kp = st.put(kp, (size_t)n);
  double *d_und = st.put<double>(nullptr, (size_t)n * 2);
  double *d_unp = st.put<double>(nullptr, (size_t)n * 3);
  uint8_t *d_ok = st.put<uint8_t>(nullptr, (size_t)n);
  HIP_TRY(st.upload());
  launch_keypoint_geometry(ctx, *camera, d_kp, nullptr, 1, n, d_und, d_unp, d_ok);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(undistorted, d_und, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(unprojected, d_unp, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ok, d_ok, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ORBGPU_OK;
}

int orbgpu_distinctive_descriptor_device(orbgpu_ctx *ctx, const uint8_t *d_desc, const int32_t *d_offsets, int32_t n_points,
                                         int32_t *d_best_index) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (n_points < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n_points == 0) return ORBGPU_OK;
  if (!d_offsets || !d_best_index) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  launch_distinctive(ctx, d_desc, d_offsets, n_points, d_best_index);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_distinctive_descriptor(orbgpu_ctx *ctx, const uint8_t *desc, const int32_t *offsets, int32_t n_points,
                                  int32_t *best_index) {
  if (!ctx) return ORBGPU_E_INVALID;
  if (n_points < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n_points == 0) return ORBGPU_OK;
  if (!offsets || !best_index) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (offsets[0] != 0) return fail(ctx, ORBGPU_E_INVALID, "offsets must start at 0");
  for (int p = 0; p < n_points; p++) {
    if (offsets[p + 1] < offsets[p]) return fail(ctx, ORBGPU_E_INVALID, "offsets must be non-decreasing");
    if (offsets[p + 1] - offsets[p] > 65535) return fail(ctx, ORBGPU_E_CAPACITY, "more than 65535 descriptors in one group");
  }
  const size_t total = (size_t)offsets[n_points];
  if (total > 0 && !desc) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> hold(ctx->mu);
  Staging st{ctx};
  uint8_t *d_desc;
  int32_t *d_off, *d_best;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = st.reserve(pad256(total * 32) + pad256(((size_t)n_points + 1) * 4) + pad256((size_t)n_points * 4) + 1024);
    if (rc) return rc;
    d_desc = st.put(desc, total * 32);
    d_off = st.put(offsets, (size_t)n_points + 1);
    d_best = st.put<int32_t>(nullptr, (size_t)n_points);
    HIP_TRY(st.upload());
  }
  int rc = orbgpu_distinctive_descriptor_device(ctx, d_desc, d_off, n_points, d_best);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipMemcpyAsync(best_index, d_best, (size_t)n_points * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ORBGPU_OK;
}

int orbgpu_vocab_create(orbgpu_ctx *ctx, int32_t n_nodes, const int32_t *child_off, const uint32_t *children,
                        const uint8_t *node_desc, const uint32_t *word_id, int32_t levels, orbgpu_vocab **out) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!out) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  *out = nullptr;
  if (n_nodes < 1 || levels < 0 || !child_off || !node_desc || !word_id) return fail(ctx, ORBGPU_E_INVALID, "bad vocabulary");
  if (child_off[0] != 0) return fail(ctx, ORBGPU_E_INVALID, "child_off must start at 0");
  for (int v = 0; v < n_nodes; v++)
    if (child_off[v + 1] < child_off[v]) return fail(ctx, ORBGPU_E_INVALID, "child_off must be non-decreasing");
  const int n_children = child_off[n_nodes];
  if (n_children > 0 && !children) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  for (int c = 0; c < n_children; c++)
    if (children[c] == 0 || children[c] >= (uint32_t)n_nodes) return fail(ctx, ORBGPU_E_INVALID, "child id outside (0, n_nodes)");
  HIP_TRY(hipSetDevice(ctx->device));
  orbgpu_vocab *v = new orbgpu_vocab;
  v->ctx = ctx;
  v->n_nodes = n_nodes;
  v->levels = levels;
  auto bail = [&](hipError_t e) {
    orbgpu_vocab_destroy(v);
    return fail_hip(ctx, e, "vocabulary upload", __FILE__, __LINE__);
  };
  hipError_t e;
  if ((e = dmalloc(&v->d_child_off, (size_t)n_nodes + 1)) != hipSuccess) return bail(e);
  if ((e = dmalloc(&v->d_children, (size_t)std::max(n_children, 1))) != hipSuccess) return bail(e);
  if ((e = dmalloc(&v->d_node_desc, (size_t)n_nodes * 32)) != hipSuccess) return bail(e);
  if ((e = dmalloc(&v->d_word_id, (size_t)n_nodes)) != hipSuccess) return bail(e);
  if ((e = hipMemcpy(v->d_child_off, child_off, ((size_t)n_nodes + 1) * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
  if (n_children && (e = hipMemcpy(v->d_children, children, (size_t)n_children * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
  if ((e = hipMemcpy(v->d_node_desc, node_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
  if ((e = hipMemcpy(v->d_word_id, word_id, (size_t)n_nodes * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(e);
  *out = v;
  return ORBGPU_OK;
}

void orbgpu_vocab_destroy(orbgpu_vocab *v) {
  if (!v) return;
  if (v->ctx) (void)hipSetDevice(v->ctx->device);
  void *ptrs[] = {v->d_child_off, v->d_children, v->d_node_desc, v->d_word_id};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  delete v;
}

int orbgpu_bow_transform_device(orbgpu_ctx *ctx, const orbgpu_vocab *vocab, const uint8_t *d_desc, int32_t n, int32_t levelsup,
                                uint32_t *d_word_id_out, uint32_t *d_node_id_out) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!vocab || vocab->ctx != ctx) return fail(ctx, ORBGPU_E_INVALID, "vocabulary belongs to another context");
  if (n < 0 || levelsup < 0) return fail(ctx, ORBGPU_E_INVALID, "negative argument");
  if (n == 0) return ORBGPU_OK;
  if (!d_desc || !d_word_id_out || !d_node_id_out) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  launch_bow_transform(ctx, vocab, d_desc, n, levelsup, d_word_id_out, d_node_id_out);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_bow_transform(orbgpu_ctx *ctx, const orbgpu_vocab *vocab, const uint8_t *desc, int32_t n, int32_t levelsup,
                         uint32_t *word_id_out, uint32_t *node_id_out) {
  if (!ctx) return ORBGPU_E_INVALID;
  if (n < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n == 0) return ORBGPU_OK;
  if (!desc || !word_id_out || !node_id_out) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> hold(ctx->mu);
  Staging st{ctx};
  uint8_t *d_desc;
  uint32_t *d_w, *d_n;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = st.reserve(pad256((size_t)n * 32) + 2 * pad256((size_t)n * 4) + 1024);
    if (rc) return rc;
    d_desc = st.put(desc, (size_t)n * 32);
    d_w = st.put<uint32_t>(nullptr, (size_t)n);
    d_n = st.put<uint32_t>(nullptr, (size_t)n);
    HIP_TRY(st.upload());
  }
  int rc = orbgpu_bow_transform_device(ctx, vocab, d_desc, n, levelsup, d_w, d_n);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipMemcpyAsync(word_id_out, d_w, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(node_id_out, d_n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ORBGPU_OK;
}

int orbgpu_bow_score_device(orbgpu_ctx *ctx, const orbgpu_bow_score_args *a, uint64_t *d_shared_words, double *d_score) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_q < 0 || a->n_keyframes < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (a->scoring < 0 || a->scoring > 5 || a->scoring == 3)
    return fail(ctx, ORBGPU_E_INVALID, "scoring must be L1_NORM, L2_NORM, CHI_SQUARE, BHATTACHARYYA or DOT_PRODUCT");
  if (a->n_keyframes == 0) return ORBGPU_OK;
  if (!a->kf_offsets || !d_shared_words || !d_score || (a->n_q > 0 && (!a->q_words || !a->q_values)))
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  BowScoreDev d{a->q_words, a->q_values, a->q_counts, a->n_q, a->n_keyframes,
                reinterpret_cast<const long long *>(a->kf_offsets), reinterpret_cast<const long long *>(a->kf_offsets) + 1,
                a->kf_words, a->kf_values, a->kf_counts, a->scoring};
  launch_bow_score(ctx, d, reinterpret_cast<unsigned long long *>(d_shared_words), d_score);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_bow_score(orbgpu_ctx *ctx, const orbgpu_bow_score_args *a, uint64_t *shared_words, double *score) {
  if (!ctx) return ORBGPU_E_INVALID;
  if (!a) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (a->n_q < 0 || a->n_keyframes < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (a->n_keyframes == 0) return ORBGPU_OK;
  if (!a->kf_offsets || !shared_words || !score || (a->n_q > 0 && (!a->q_words || !a->q_values)))
    return fail(ctx, ORBGPU_E_INVALID, "null argument");
  for (int i = 1; i < a->n_q; i++)
    if (a->q_words[i] <= a->q_words[i - 1]) return fail(ctx, ORBGPU_E_INVALID, "query words must be strictly ascending");
  if (a->kf_offsets[0] != 0) return fail(ctx, ORBGPU_E_INVALID, "offsets must start at 0");
  for (int k = 0; k < a->n_keyframes; k++) {
    if (a->kf_offsets[k + 1] < a->kf_offsets[k]) return fail(ctx, ORBGPU_E_INVALID, "offsets must be non-decreasing");
    if (a->kf_offsets[k + 1] > a->kf_offsets[k] && (!a->kf_words || !a->kf_values)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
    for (int64_t i = a->kf_offsets[k] + 1; i < a->kf_offsets[k + 1]; i++)
      if (a->kf_words[i] <= a->kf_words[i - 1])
        return fail(ctx, ORBGPU_E_INVALID, "keyframe words must be strictly ascending inside a keyframe");
  }
  const size_t total = (size_t)a->kf_offsets[a->n_keyframes], nq = (size_t)a->n_q, nk = (size_t)a->n_keyframes;
  std::lock_guard<std::recursive_mutex> hold(ctx->mu);
  Staging st{ctx};
  orbgpu_bow_score_args d = *a;
  unsigned long long *d_shared;
  double *d_score;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = st.reserve(pad256(nq * 4) * 2 + pad256(nq * 8) + pad256((nk + 1) * 8) + pad256(total * 4) * 2 + pad256(total * 8) +
                        2 * pad256(nk * 8) + 4096);
    if (rc) return rc;
    d.q_words = nq ? st.put(a->q_words, nq) : nullptr;
    d.q_values = nq ? st.put(a->q_values, nq) : nullptr;
    d.q_counts = (nq && a->q_counts) ? st.put(a->q_counts, nq) : nullptr;
    d.kf_offsets = st.put(a->kf_offsets, nk + 1);
    d.kf_words = total ? st.put(a->kf_words, total) : nullptr;
    d.kf_values = total ? st.put(a->kf_values, total) : nullptr;
    d.kf_counts = (total && a->kf_counts) ? st.put(a->kf_counts, total) : nullptr;
    d_shared = st.put<unsigned long long>(nullptr, nk);
    d_score = st.put<double>(nullptr, nk);
    HIP_TRY(st.upload());
  }
  int rc = orbgpu_bow_score_device(ctx, &d, reinterpret_cast<uint64_t *>(d_shared), d_score);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipMemcpyAsync(shared_words, d_shared, nk * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(score, d_score, nk * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ORBGPU_OK;
}

// ---- resident keyframe database ----------------------------------------------------------------------
namespace {
bool scoring_ok(int s) { return s >= 0 && s <= 5 && s != 3; }

// arena for at least `need` entries; live runs are moved (and holes dropped) when it has to grow or is mostly holes
int bowdb_reserve(orbgpu_bowdb *db, long long need) {
  orbgpu_ctx *ctx = db->ctx;
  const long long live = db->used - db->holes;
  const bool grow = db->used + need > db->cap;
  const bool squeeze = db->holes > (1 << 16) && db->holes > live;
  if (!grow && !squeeze) return 0;
  long long cap = db->cap;
  if (live + need > cap / 2 || cap == 0) cap = std::max<long long>(2 * (live + need), 1 << 16);
  unsigned *nw = nullptr, *nc = nullptr;
  double *nv = nullptr;
  const int n = (int)db->ids.size();
  std::vector<long long> nbeg(n), nend(n);
  long long pos = 0;
  for (int k = 0; k < n; k++) {
    nbeg[k] = pos;
    pos += db->end[k] - db->beg[k];
    nend[k] = pos;
  }
  HIP_TRY(dmalloc(&nw, (size_t)cap));
  hipError_t e = dmalloc(&nv, (size_t)cap);
  if (e == hipSuccess && db->with_counts) e = dmalloc(&nc, (size_t)cap);
  long long *d_tab = nullptr;
  if (e == hipSuccess && n) e = dmalloc(&d_tab, (size_t)3 * n);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_tab, db->beg.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_tab + n, nbeg.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) e = hipMemcpyAsync(d_tab + 2 * n, nend.data(), (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && n) {
    launch_bowdb_compact(ctx, n, d_tab, d_tab + n, d_tab + 2 * n, db->d_words, db->d_values, db->d_counts, nw, nv, nc);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (d_tab) (void)hipFree(d_tab);
  if (e != hipSuccess) {
    for (void *q : {(void *)nw, (void *)nv, (void *)nc})
      if (q) (void)hipFree(q);
    return fail_hip(ctx, e, "keyframe database arena", __FILE__, __LINE__);
  }
  for (void *q : {(void *)db->d_words, (void *)db->d_values, (void *)db->d_counts})
    if (q) (void)hipFree(q);
  db->d_words = nw;
  db->d_values = nv;
  db->d_counts = nc;
  db->cap = cap;
  db->beg.swap(nbeg);
  db->end.swap(nend);
  db->used = pos;
  db->holes = 0;
  db->table_dirty = true;
  return 0;
}
}  // namespace

int orbgpu_bowdb_create(orbgpu_ctx *ctx, int32_t scoring, int32_t with_counts, orbgpu_bowdb **out) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!out) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  *out = nullptr;
  if (!scoring_ok(scoring))
    return fail(ctx, ORBGPU_E_INVALID, "scoring must be L1_NORM, L2_NORM, CHI_SQUARE, BHATTACHARYYA or DOT_PRODUCT");
  orbgpu_bowdb *db = new orbgpu_bowdb;
  db->ctx = ctx;
  db->scoring = scoring;
  db->with_counts = with_counts != 0;
  *out = db;
  return ORBGPU_OK;
}

void orbgpu_bowdb_destroy(orbgpu_bowdb *db) {
  if (!db) return;
  if (db->ctx) {
    (void)hipSetDevice(db->ctx->device);
    (void)hipStreamSynchronize(db->ctx->stream);
  }
  for (void *q : {(void *)db->d_words, (void *)db->d_values, (void *)db->d_counts, (void *)db->d_beg, (void *)db->d_end,
                  (void *)db->d_query})
    if (q) (void)hipFree(q);
  delete db;
}

int orbgpu_bowdb_size(const orbgpu_bowdb *db, int32_t *n_keyframes) {
  if (!db || !n_keyframes) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(db->ctx->mu);
  *n_keyframes = (int32_t)db->ids.size();
  return ORBGPU_OK;
}

int orbgpu_bowdb_erase(orbgpu_bowdb *db, uint64_t keyframe_id) {
  if (!db) return ORBGPU_E_INVALID;
  orbgpu_ctx *ctx = db->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  auto it = db->slot_of.find(keyframe_id);
  if (it == db->slot_of.end()) return ORBGPU_OK;
  const int slot = it->second, last = (int)db->ids.size() - 1;
  db->holes += db->end[slot] - db->beg[slot];
  db->slot_of.erase(it);
  if (slot != last) {  // the last keyframe takes the freed slot
    db->ids[slot] = db->ids[last];
    db->beg[slot] = db->beg[last];
    db->end[slot] = db->end[last];
    db->slot_of[db->ids[slot]] = slot;
  }
  db->ids.pop_back();
  db->beg.pop_back();
  db->end.pop_back();
  db->table_dirty = true;
  return ORBGPU_OK;
}

int orbgpu_bowdb_append(orbgpu_bowdb *db, uint64_t keyframe_id, const uint32_t *words, const double *values,
                        const uint32_t *counts, int32_t n_words) {
  if (!db) return ORBGPU_E_INVALID;
  orbgpu_ctx *ctx = db->ctx;
  {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (n_words < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
    if (n_words > 0 && (!words || !values)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
    if (counts && !db->with_counts) return fail(ctx, ORBGPU_E_INVALID, "this database was created without word counts");
    for (int i = 1; i < n_words; i++)
      if (words[i] <= words[i - 1]) return fail(ctx, ORBGPU_E_INVALID, "keyframe words must be strictly ascending");
  }
  int rc = orbgpu_bowdb_erase(db, keyframe_id);  // an id that is present is replaced
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  rc = bowdb_reserve(db, n_words);
  if (rc) return rc;
  const long long at = db->used;
  if (n_words) {
    HIP_TRY(hipMemcpyAsync(db->d_words + at, words, (size_t)n_words * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db->d_values + at, values, (size_t)n_words * 8, hipMemcpyHostToDevice, ctx->stream));
    if (db->with_counts) {
      std::vector<uint32_t> ones;
      if (!counts) {
        ones.assign((size_t)n_words, 1u);
        counts = ones.data();
      }
      HIP_TRY(hipMemcpyAsync(db->d_counts + at, counts, (size_t)n_words * 4, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
    } else {
      HIP_TRY(hipStreamSynchronize(ctx->stream));  // the caller's arrays may go away after the call
    }
  }
  db->slot_of[keyframe_id] = (int)db->ids.size();
  db->ids.push_back(keyframe_id);
  db->beg.push_back(at);
  db->end.push_back(at + n_words);
  db->used = at + n_words;
  db->table_dirty = true;
  return ORBGPU_OK;
}

int orbgpu_bowdb_query(orbgpu_bowdb *db, const uint32_t *q_words, const double *q_values, const uint32_t *q_counts,
                       int32_t n_q, int32_t capacity, uint64_t *keyframe_ids, uint64_t *shared_words, double *score,
                       int32_t *n_out) {
  if (!db) return ORBGPU_E_INVALID;
  orbgpu_ctx *ctx = db->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (!n_out) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  *n_out = 0;
  if (n_q < 0 || capacity < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n_q > 0 && (!q_words || !q_values)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  for (int i = 1; i < n_q; i++)
    if (q_words[i] <= q_words[i - 1]) return fail(ctx, ORBGPU_E_INVALID, "query words must be strictly ascending");
  const size_t n = db->ids.size();
  if ((size_t)capacity < n) return fail(ctx, ORBGPU_E_CAPACITY, "output capacity below the number of keyframes");
  if (n == 0) return ORBGPU_OK;
  if (!keyframe_ids || !shared_words || !score) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  if (db->table_dirty) {
    if (db->table_cap < n) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (db->d_beg) HIP_TRY(hipFree(db->d_beg));
      if (db->d_end) HIP_TRY(hipFree(db->d_end));
      db->d_beg = db->d_end = nullptr;
      db->table_cap = 0;
      const size_t cap = std::max<size_t>(2 * n, 1024);
      HIP_TRY(dmalloc(&db->d_beg, cap));
      HIP_TRY(dmalloc(&db->d_end, cap));
      db->table_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(db->d_beg, db->beg.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db->d_end, db->end.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    db->table_dirty = false;
  }
  const size_t nq = (size_t)n_q;
  const size_t o_w = 0, o_v = pad256(nq * 4), o_c = o_v + pad256(nq * 8), o_sh = o_c + pad256(nq * 4), o_sc = o_sh + pad256(n * 8),
               bytes = o_sc + pad256(n * 8);
  if (db->query_cap < bytes) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (db->d_query) HIP_TRY(hipFree(db->d_query));
    db->d_query = nullptr;
    db->query_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&db->d_query), 2 * bytes));
    db->query_cap = 2 * bytes;
  }
  char *q = db->d_query;
  if (nq) {
    HIP_TRY(hipMemcpyAsync(q + o_w, q_words, nq * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(q + o_v, q_values, nq * 8, hipMemcpyHostToDevice, ctx->stream));
    if (q_counts) HIP_TRY(hipMemcpyAsync(q + o_c, q_counts, nq * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  BowScoreDev d{reinterpret_cast<unsigned *>(q + o_w), reinterpret_cast<double *>(q + o_v),
                q_counts ? reinterpret_cast<unsigned *>(q + o_c) : nullptr, n_q, (int)n, db->d_beg, db->d_end, db->d_words,
                db->d_values, db->with_counts ? db->d_counts : nullptr, db->scoring};
  launch_bow_score(ctx, d, reinterpret_cast<unsigned long long *>(q + o_sh), reinterpret_cast<double *>(q + o_sc));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(shared_words, q + o_sh, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(score, q + o_sc, n * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (size_t k = 0; k < n; k++) keyframe_ids[k] = db->ids[k];
  *n_out = (int32_t)n;
  return ORBGPU_OK;
}

static int match_allpairs_locked(orbgpu_ctx *ctx, const void *d_ang_rows, const uint8_t *d_desc_rows, const int32_t *d_n_rows,
                                 int32_t n_rows, const void *d_ang_cols, int ang_stride, const uint8_t *d_desc_cols,
                                 const int32_t *d_n_cols, int32_t n_cols, int32_t capacity, int32_t row_offset,
                                 uint32_t threshold, double ratio, int32_t check_orientation, int32_t *d_n_matches,
                                 int32_t *d_match) {
  if (n_rows < 0 || n_cols < 0 || capacity < 1) return fail(ctx, ORBGPU_E_INVALID, "bad size");
  if (n_rows == 0 || n_cols == 0) return ORBGPU_OK;
  if (!d_desc_rows || !d_n_rows || !d_desc_cols || !d_n_cols || !d_n_matches) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  if (check_orientation && (!d_ang_rows || !d_ang_cols))
    return fail(ctx, ORBGPU_E_INVALID, "check_orientation needs the keypoint (or angle) arrays of both sets");
  // dynamic LDS plus the kernel's static __shared__ (rotation histogram, counters: < 256 B) must fit the 64 KB a
  // workgroup gets without raising the limit
  // (the matrix-core engine holds another 18 KB of static LDS: its staging buffers and expansion table)
  if (allpairs_lds_bytes(capacity) + (ctx->bruteforce_mfma ? 19 * 1024 : 256) > 64 * 1024)
    return fail(ctx, ORBGPU_E_CAPACITY, ctx->bruteforce_mfma ? "capacity above 2880 keypoints per keyframe with the MFMA engine"
                                                             : "capacity above 4080 keypoints per keyframe");
  if ((uint64_t)n_cols * (uint64_t)((n_rows + 7) & ~7) >= (1ull << 31)) return fail(ctx, ORBGPU_E_INVALID, "too many pairs for one call");
  HIP_TRY(hipSetDevice(ctx->device));
  launch_match_allpairs(ctx, d_ang_rows, d_desc_rows, d_n_rows, n_rows, d_ang_cols, ang_stride, d_desc_cols, d_n_cols, n_cols,
                        capacity, row_offset, threshold, ratio, check_orientation, d_n_matches, d_match);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

int orbgpu_match_allpairs_device(orbgpu_ctx *ctx, const orbgpu_keypoint *d_kp_rows, const uint8_t *d_desc_rows,
                                 const int32_t *d_n_rows, int32_t n_rows, const orbgpu_keypoint *d_kp_cols,
                                 const uint8_t *d_desc_cols, const int32_t *d_n_cols, int32_t n_cols, int32_t capacity,
                                 int32_t row_offset, uint32_t threshold, double ratio, int32_t check_orientation,
                                 int32_t *d_n_matches, int32_t *d_match) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  return match_allpairs_locked(ctx, d_kp_rows ? &d_kp_rows->angle : nullptr, d_desc_rows, d_n_rows, n_rows,
                               d_kp_cols ? &d_kp_cols->angle : nullptr, (int)sizeof(orbgpu_keypoint), d_desc_cols, d_n_cols,
                               n_cols, capacity, row_offset, threshold, ratio, check_orientation, d_n_matches, d_match);
}

int orbgpu_match_allpairs_angles_device(orbgpu_ctx *ctx, const float *d_angle_rows, const uint8_t *d_desc_rows,
                                        const int32_t *d_n_rows, int32_t n_rows, const float *d_angle_cols,
                                        const uint8_t *d_desc_cols, const int32_t *d_n_cols, int32_t n_cols,
                                        int32_t capacity, int32_t row_offset, uint32_t threshold, double ratio,
                                        int32_t check_orientation, int32_t *d_n_matches, int32_t *d_match) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  return match_allpairs_locked(ctx, d_angle_rows, d_desc_rows, d_n_rows, n_rows, d_angle_cols, (int)sizeof(float), d_desc_cols,
                               d_n_cols, n_cols, capacity, row_offset, threshold, ratio, check_orientation, d_n_matches,
                               d_match);
}

int orbgpu_rotation_filter(orbgpu_ctx *ctx, const float *angle_to, int32_t n_to, const float *angle_from, int32_t n_from,
                           int32_t *match, int32_t *n_discarded) {
  if (!ctx) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  if (n_to < 0 || n_from < 0) return fail(ctx, ORBGPU_E_INVALID, "negative count");
  if (n_discarded) *n_discarded = 0;
  if (n_to == 0) return ORBGPU_OK;
  if (!angle_to || !match || (n_from > 0 && !angle_from)) return fail(ctx, ORBGPU_E_INVALID, "null argument");
  for (int i = 0; i < n_to; i++)
    if (match[i] >= n_from) return fail(ctx, ORBGPU_E_INVALID, "match index outside [0, n_from)");
  HIP_TRY(hipSetDevice(ctx->device));
  Staging st{ctx};
  int rc = st.reserve(pad256((size_t)n_to * 4) * 2 + pad256((size_t)n_from * 4) + 1024);
  if (rc) return rc;
  float *d_to = st.put(angle_to, (size_t)n_to);
  float *d_from = st.put(angle_from, (size_t)n_from);
  int32_t *d_m = st.put(match, (size_t)n_to);
  int32_t *d_nd = st.put<int32_t>(nullptr, 1);
  HIP_TRY(st.upload());
  launch_rotation_filter(ctx, d_to, n_to, d_from, d_m, d_nd);
  HIP_TRY(hipGetLastError());
  int32_t nd = 0;
  HIP_TRY(hipMemcpyAsync(match, d_m, (size_t)n_to * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(&nd, d_nd, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (n_discarded) *n_discarded = nd;
  return ORBGPU_OK;
}

int orbgpu_stage_times(orbgpu_ctx *ctx, float *ms8) {
  if (!ctx || !ms8) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  int rc = finish_pipeline(ctx);
  if (rc) return rc;
  for (int i = 0; i < 8; i++) ms8[i] = ctx->stage_ms[i];
  return ORBGPU_OK;
}

int orbgpu_popcount_peak(orbgpu_ctx *ctx, double *pairs_per_second) {
  if (!ctx || !pairs_per_second) return ORBGPU_E_INVALID;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  HIP_TRY(hipSetDevice(ctx->device));
  *pairs_per_second = run_popcount_peak(ctx);
  HIP_TRY(hipGetLastError());
  return ORBGPU_OK;
}

}  // extern "C"
