/*
 * orb_oracle.c -- CPU restatement of the reference ORB front-end.  TEST INFRASTRUCTURE ONLY:
 * see the header of orb_oracle.h for who may load this and for the parity status
 * ("parity unpinned" for the OpenCV-backed extraction stages; no oracle/_ref build is possible here).
 *
 * Plain C11, compile with -O3 -ffp-contract=off (the reference is built -O3 for baseline x86-64,
 * i.e. no FMA: src/CMakeLists.txt:10-11).  Citations are file:line under /root/reference/.
 */
#include "orb_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* OpenCV scalar helpers: cvRound = round-half-to-even, cvFloor.             */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (i > v);
}
static inline short sat_short(int v) { return (short)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v); }

static const int8_t k_pattern[256 * 4] = {
#include "orb_pattern.inc"
};

/* umax_, orb_feature_extractor.cpp:678-679 */
static const int k_umax[OO_HALF_PATCH + 1] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

/* ------------------------------------------------------------------------------------------ */
/* geometry                                                                                    */
int oo_geometry_compute(const oo_params *p, int n_features, oo_geometry *g) {
  if (p->levels < 1 || p->levels > OO_MAX_LEVELS) return -2;
  memset(g, 0, sizeof(*g));
  g->levels = p->levels;
  /* Initialize(), orb_feature_extractor.cpp:53-64: cumulative double products */
  g->scale[0] = 1.0;
  g->inv_scale[0] = 1.0;
  g->sigma2[0] = 1.0;
  for (int i = 1; i < p->levels; i++) {
    g->scale[i] = g->scale[i - 1] * p->scale_factor;
    g->inv_scale[i] = 1.0 / g->scale[i];
    g->sigma2[i] = g->scale[i] * g->scale[i];
  }
  /* per-level budget, :211-224 */
  {
    double factor = 1.0 / p->scale_factor;
    double per = (double)(size_t)n_features * (1 - factor) / (1 - pow(factor, (double)p->levels));
    size_t total = 0;
    for (int l = 0; l < p->levels - 1; l++) {
      g->budget[l] = (int)round(per);
      total += (size_t)g->budget[l];
      per *= factor;
    }
    size_t rest = (size_t)n_features - total; /* size_t arithmetic as in :224 */
    g->budget[p->levels - 1] = (int)rest;
  }
  for (int l = 0; l < p->levels; l++) {
    /* BuildImagePyramid :506-508 */
    float scale = (float)g->inv_scale[l];
    g->w[l] = cv_round_f((float)p->width * scale);
    g->h[l] = cv_round_f((float)p->height * scale);
    /* cell grid :228-242 */
    const int minB = OO_EDGE - 3;
    const int maxBX = g->w[l] - OO_EDGE + 3, maxBY = g->h[l] - OO_EDGE + 3;
    const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
    const float W = 30;
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return -3; /* reference divides by zero here */
    g->ncols[l] = nCols;
    g->nrows[l] = nRows;
    g->wcell[l] = (int)ceilf(width / nCols);
    g->hcell[l] = (int)ceilf(height / nRows);
    /* DistributeOctTree :310 */
    g->nini[l] = (int)round((double)(maxBX - minB) / (maxBY - minB));
    if (g->nini[l] < 1) return -4;
    /* :284 */
    g->kp_size[l] = (int)(OO_PATCH * g->scale[l]);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::resize INTER_LINEAR CV_8UC1 (OpenCV 4.5.4 generic path)               */
int oo_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                        int dstride) {
  if (sw < 1 || sh < 1 || dw < 1 || dh < 1) return -1;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1.0 / inv_scale_x, scale_y = 1.0 / inv_scale_y;
  {
    /* exact 2x decimation is silently promoted to INTER_AREA by cv::resize; not restated */
    int isx = cv_round_d(scale_x), isy = cv_round_d(scale_y);
    if (fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON && isx == 2 && isy == 2)
      return -2;
  }
  int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
  short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
  int *rows[2];
  rows[0] = (int *)malloc(sizeof(int) * (size_t)dw);
  rows[1] = (int *)malloc(sizeof(int) * (size_t)dw);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) {
      fx = 0;
      sx = 0;
    }
    if (sx + 1 >= sw) {
      if (dx < xmax) xmax = dx;
      if (sx >= sw - 1) {
        fx = 0;
        sx = sw - 1;
      }
    }
    xofs[dx] = sx;
    ialpha[2 * dx] = sat_short(cv_round_f((1.f - fx) * 2048.f));
    ialpha[2 * dx + 1] = sat_short(cv_round_f(fx * 2048.f));
  }
  int prev_sy[2] = {-1, -1};
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy0 = cv_floor_f(fy);
    fy -= sy0;
    const short b0 = sat_short(cv_round_f((1.f - fy) * 2048.f));
    const short b1 = sat_short(cv_round_f(fy * 2048.f));
    for (int k = 0; k < 2; k++) {
      int sy = sy0 + k;
      sy = sy >= 0 ? (sy < sh ? sy : sh - 1) : 0; /* clip(sy, 0, sh) */
      if (prev_sy[k] == sy) continue;
      if (k == 0 && prev_sy[1] == sy) { /* reuse the row computed for the other slot */
        memcpy(rows[0], rows[1], sizeof(int) * (size_t)dw);
        prev_sy[0] = sy;
        continue;
      }
      const uint8_t *S = src + (size_t)sy * sstride;
      int *D = rows[k];
      int dx = 0;
      for (; dx < xmax; dx++) {
        int sx = xofs[dx];
        D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
      }
      for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * 2048;
      prev_sy[k] = sy;
    }
    const int *S0 = rows[0], *S1 = rows[1];
    uint8_t *d = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      d[x] = (uint8_t)((((b0 * (S0[x] >> 4)) >> 16) + ((b1 * (S1[x] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs);
  free(ialpha);
  free(rows[0]);
  free(rows[1]);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::GaussianBlur 8U 7x7 sigma=2 REFLECT_101, fixed-point path (App. A.3)                    */
static const int k_gauss7[7] = {18, 34, 48, 56, 48, 34, 18}; /* 8.8 fixed point, sum 256 */

static inline int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * (n - 1) - i;
  }
  return i;
}

void oo_gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
  uint16_t *hbuf = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * (size_t)h);
  for (int y = 0; y < h; y++) {
    const uint8_t *s = src + (size_t)y * sstride;
    uint16_t *hb = hbuf + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      unsigned acc = 0;
      if (x >= 3 && x < w - 3) {
        for (int i = 0; i < 7; i++) acc += (unsigned)k_gauss7[i] * s[x + i - 3];
      } else {
        for (int i = 0; i < 7; i++) acc += (unsigned)k_gauss7[i] * s[reflect101(x + i - 3, w)];
      }
      hb[x] = (uint16_t)acc; /* <= 255*256 */
    }
  }
  for (int y = 0; y < h; y++) {
    const uint16_t *r[7];
    for (int j = 0; j < 7; j++) r[j] = hbuf + (size_t)reflect101(y + j - 3, h) * w;
    uint8_t *d = dst + (size_t)y * dstride;
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int j = 0; j < 7; j++) acc += (uint32_t)k_gauss7[j] * r[j][x];
      d[x] = (uint8_t)((acc + 32768u) >> 16);
    }
  }
  free(hbuf);
}

/* ------------------------------------------------------------------------------------------ */
/* cv::fastAtan2 scalar (App. A.4). Constants are float products evaluated in float.           */
float oo_fast_atan2(float y, float x) {
  const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * scale;
  const float p3 = -0.3258083974640975f * scale;
  const float p5 = 0.1555786518463281f * scale;
  const float p7 = -0.04432655554792128f * scale;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* IC_Angle, orb_feature_extractor.cpp:177-202 */
float oo_ic_angle(const uint8_t *img, int stride, int cx, int cy) {
  int m_01 = 0, m_10 = 0;
  const uint8_t *center = img + (size_t)cy * stride + cx;
  for (int u = -OO_HALF_PATCH; u <= OO_HALF_PATCH; ++u) m_10 += u * center[u];
  for (int v = 1; v <= OO_HALF_PATCH; ++v) {
    int v_sum = 0;
    int d = k_umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return oo_fast_atan2((float)m_01, (float)m_10);
}

/* cos / sin of the keypoint angle, :124-125: `(float) cos(angle)` on a float.  Which function that names depends on the
 * headers in scope: ::cos(double) (plain <cmath>; the float is widened, the correctly rounded double result rounded to
 * float: variant 0, the default here and in the kernel) or the float overload = cosf (<math.h> in scope; glibc's cosf
 * is within 1 ulp but not always the rounded value: variant 1).  Build with -DOO_TRIG_COSF=1 to make variant 1 the
 * default; oo_set_trig_variant switches at run time (tests/test_oracle.py reports how far the two are apart). */
#ifndef OO_TRIG_COSF
#define OO_TRIG_COSF 0
#endif
static int g_trig_variant = OO_TRIG_COSF;
void oo_set_trig_variant(int v) { g_trig_variant = v ? 1 : 0; }
int oo_get_trig_variant(void) { return g_trig_variant; }
void oo_descriptor_trig(double angle_deg, int variant, float *cos_out, float *sin_out) {
  const double factorPI = (M_PI / 180.); /* :11 */
  float angle = (float)((float)angle_deg * factorPI);
  if (variant) {
    *cos_out = cosf(angle);
    *sin_out = sinf(angle);
  } else {
    *cos_out = (float)cos((double)angle);
    *sin_out = (float)sin((double)angle);
  }
}

/* computeOrbDescriptor, :120-165 (float details SURVEY App. B.5) */
void oo_orb_descriptor(const uint8_t *img, int stride, int cx, int cy, double angle_deg, uint8_t *desc) {
  float a, b;
  oo_descriptor_trig(angle_deg, g_trig_variant, &a, &b);
  const uint8_t *center = img + (ptrdiff_t)cy * stride + cx;
  const int8_t *pat = k_pattern;
  for (int i = 0; i < 32; ++i) {
    int val = 0;
    for (int k = 0; k < 8; ++k, pat += 4) {
      float x0 = (float)pat[0], y0 = (float)pat[1], x1 = (float)pat[2], y1 = (float)pat[3];
      int t0 = center[cv_round_f(x0 * b + y0 * a) * stride + cv_round_f(x0 * a - y0 * b)];
      int t1 = center[cv_round_f(x1 * b + y1 * a) * stride + cv_round_f(x1 * a - y1 * b)];
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* FAST-9/16 (App. A.2)                                                                        */
static const int k_ring[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int oo_fast_score(const uint8_t *img, int stride, int x, int y) {
  const uint8_t *p = img + (ptrdiff_t)y * stride + x;
  int v = p[0];
  int d[16];
  for (int k = 0; k < 16; k++) d[k] = v - p[k_ring[k][0] + k_ring[k][1] * stride];
  int m_dark = INT_MIN, m_bright = INT_MIN;
  for (int s = 0; s < 16; s++) {
    int mn = INT_MAX, mx = INT_MIN;
    for (int j = 0; j < 9; j++) {
      int t = d[(s + j) & 15];
      if (t < mn) mn = t;
      if (t > mx) mx = t;
    }
    if (mn > m_dark) m_dark = mn;    /* max over arcs of min d      */
    if (-mx > m_bright) m_bright = -mx; /* max over arcs of min (-d) */
  }
  return (m_dark > m_bright ? m_dark : m_bright) - 1;
}

void oo_fast_score_map(const uint8_t *img, int w, int h, int stride, int min_thr, uint8_t *score, int ostride) {
  for (int y = 0; y < h; y++) memset(score + (size_t)y * ostride, 0, (size_t)w);
  for (int y = 3; y <= h - 4; y++)
    for (int x = 3; x <= w - 4; x++) {
      int s = oo_fast_score(img, stride, x, y);
      score[(size_t)y * ostride + x] = (uint8_t)(s >= min_thr ? s : 0);
    }
}

/* literal cornerScore<16> of OpenCV (fast_score.cpp) */
static int corner_score16_literal(const uint8_t *ptr, const int pixel[], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[25];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
    a = a < d[k + 3] ? a : d[k + 3];
    if (a <= a0) continue;
    for (int j = 4; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
    int t = a < d[k] ? a : d[k];
    if (t > a0) a0 = t;
    t = a < d[k + 9] ? a : d[k + 9];
    if (t > a0) a0 = t;
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
    for (int j = 3; j <= 5; j++) b = b > d[k + j] ? b : d[k + j];
    if (b >= b0) continue;
    for (int j = 6; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
    int t = b > d[k] ? b : d[k];
    if (t < b0) b0 = t;
    t = b > d[k + 9] ? b : d[k + 9];
    if (t < b0) b0 = t;
  }
  return -b0 - 1;
}

/* literal FAST_t<16> of OpenCV (fast.cpp), scalar path, with the 3-row score buffer NMS */
int oo_fast_literal(const uint8_t *img, int stride, int cols, int rows, int threshold, int nonmax, int cap,
                    int *ox, int *oy, int *oscore) {
  const int K = 8, N = 16 + K + 1;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = k_ring[k][0] + k_ring[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  if (cols < 1 || rows < 1) return 0;
  uint8_t *buf[3];
  int *cpbuf[3];
  for (int i = 0; i < 3; i++) {
    buf[i] = (uint8_t *)calloc((size_t)cols + 8, 1);
    cpbuf[i] = (int *)calloc((size_t)cols + 2, sizeof(int));
  }
  int n = 0;
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t *ptr = img + (ptrdiff_t)i * stride + 3;
    uint8_t *curr = buf[(i - 3) % 3];
    int *cornerpos = cpbuf[(i - 3) % 3] + 1;
    memset(curr, 0, (size_t)cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t *tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)corner_score16_literal(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)corner_score16_literal(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t *prev = buf[(i - 4 + 3) % 3];
    const uint8_t *pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                      score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        if (n < cap) {
          ox[n] = j;
          oy[n] = i - 1;
          oscore[n] = score;
        }
        n++;
      }
    }
  }
  for (int i = 0; i < 3; i++) {
    free(buf[i]);
    free(cpbuf[i]);
  }
  return n;
}

/* per-cell NMS on the threshold-independent score map; rect = detection rectangle (inclusive) */
static int cell_nms(const uint8_t *score, int sstride, int x0, int x1, int y0, int y1, int thr, oo_cand *out,
                    int cap, int n, float offx, float offy) {
  for (int y = y0; y <= y1; y++)
    for (int x = x0; x <= x1; x++) {
      int s = score[(size_t)y * sstride + x];
      if (s < thr) continue;
      int ok = 1;
      for (int dy = -1; dy <= 1 && ok; dy++)
        for (int dx = -1; dx <= 1; dx++) {
          if (!dx && !dy) continue;
          int xx = x + dx, yy = y + dy, nb = 0;
          if (xx >= x0 && xx <= x1 && yy >= y0 && yy <= y1) {
            nb = score[(size_t)yy * sstride + xx];
            if (nb < thr) nb = 0;
          }
          if (!(s > nb)) {
            ok = 0;
            break;
          }
        }
      if (ok) {
        if (n < cap) {
          out[n].x = (float)x - offx;
          out[n].y = (float)y - offy;
          out[n].response = (float)s;
        }
        n++;
      }
    }
  return n;
}

/* ComputeKeyPointsOctTree cell loop, orb_feature_extractor.cpp:227-276 */
int oo_detect_level(const uint8_t *img, int w, int h, int stride, int fast_init, int fast_min, int use_literal,
                    oo_cand *out, int cap) {
  const int minBorderX = OO_EDGE - 3, minBorderY = minBorderX;
  const int maxBorderX = w - OO_EDGE + 3, maxBorderY = h - OO_EDGE + 3;
  const float W = 30;
  const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  if (nCols < 1 || nRows < 1) return -3;
  const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
  uint8_t *score = NULL;
  if (!use_literal) {
    score = (uint8_t *)malloc((size_t)w * h);
    oo_fast_score_map(img, w, h, stride, fast_min, score, w);
  }
  const int tmpcap = 64 * 64;
  int *tx = (int *)malloc(sizeof(int) * 3 * tmpcap), *ty = tx + tmpcap, *ts = ty + tmpcap;
  int n = 0;
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBorderY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBorderY - 3) continue;
    if (maxY > maxBorderY) maxY = (float)maxBorderY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBorderX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBorderX - 6) continue;
      if (maxX > maxBorderX) maxX = (float)maxBorderX;
      const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY; /* ROI [x0,x1) x [y0,y1) */
      if (use_literal) {
        const uint8_t *roi = img + (size_t)y0 * stride + x0;
        int c = oo_fast_literal(roi, stride, x1 - x0, y1 - y0, fast_init, 1, tmpcap, tx, ty, ts);
        if (c == 0) c = oo_fast_literal(roi, stride, x1 - x0, y1 - y0, fast_min, 1, tmpcap, tx, ty, ts);
        if (c > tmpcap) c = tmpcap;
        for (int k = 0; k < c; k++) {
          if (n < cap) {
            out[n].x = (float)tx[k] + (float)(j * wCell); /* :270-271 */
            out[n].y = (float)ty[k] + (float)(i * hCell);
            out[n].response = (float)ts[k];
          }
          n++;
        }
      } else {
        /* cv::FAST scans [3, cw-4] x [3, ch-4] of the ROI */
        int dx0 = x0 + 3, dx1 = x1 - 4, dy0 = y0 + 3, dy1 = y1 - 4;
        if (dx1 < dx0 || dy1 < dy0) continue;
        int before = n;
        n = cell_nms(score, w, dx0, dx1, dy0, dy1, fast_init, out, cap, n, (float)minBorderX, (float)minBorderY);
        if (n == before)
          n = cell_nms(score, w, dx0, dx1, dy0, dy1, fast_min, out, cap, n, (float)minBorderX, (float)minBorderY);
      }
    }
  }
  free(tx);
  free(score);
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* DistributeOctTree :300-502 with an index-linked emulation of std::list<ExtractorNode>       */
typedef struct {
  int x0, x1, y0, y1; /* UL.x, UR.x, UL.y, BR.y */
  int *keys;
  int nkeys;
  int nomore;
  int prev, next;
  int alive;
} qnode;

typedef struct {
  qnode *n;
  int count, cap;
  int head, tail, size;
} qlist;

static int ql_new(qlist *L) {
  if (L->count == L->cap) {
    L->cap = L->cap ? L->cap * 2 : 256;
    L->n = (qnode *)realloc(L->n, sizeof(qnode) * (size_t)L->cap);
  }
  qnode *q = &L->n[L->count];
  memset(q, 0, sizeof(*q));
  q->prev = q->next = -1;
  return L->count++;
}
static void ql_push_front(qlist *L, int i) {
  L->n[i].prev = -1;
  L->n[i].next = L->head;
  if (L->head >= 0) L->n[L->head].prev = i;
  L->head = i;
  if (L->tail < 0) L->tail = i;
  L->n[i].alive = 1;
  L->size++;
}
static void ql_push_back(qlist *L, int i) {
  L->n[i].next = -1;
  L->n[i].prev = L->tail;
  if (L->tail >= 0) L->n[L->tail].next = i;
  L->tail = i;
  if (L->head < 0) L->head = i;
  L->n[i].alive = 1;
  L->size++;
}
static int ql_erase(qlist *L, int i) { /* returns next */
  int p = L->n[i].prev, nx = L->n[i].next;
  if (p >= 0) L->n[p].next = nx; else L->head = nx;
  if (nx >= 0) L->n[nx].prev = p; else L->tail = p;
  L->n[i].alive = 0;
  free(L->n[i].keys);
  L->n[i].keys = NULL;
  L->size--;
  return nx;
}

/* DivideNode :602-652; creates up to 4 children in the pool (not linked). child[k] = -1 when empty */
static void q_divide(qlist *L, int parent, const oo_cand *c, int child[4]) {
  qnode P = L->n[parent];
  const int halfX = (int)ceilf((float)(P.x1 - P.x0) / 2);
  const int halfY = (int)ceilf((float)(P.y1 - P.y0) / 2);
  const int mx = P.x0 + halfX, my = P.y0 + halfY;
  int *k[4], nk[4] = {0, 0, 0, 0};
  for (int q = 0; q < 4; q++) k[q] = (int *)malloc(sizeof(int) * (size_t)(P.nkeys ? P.nkeys : 1));
  for (int i = 0; i < P.nkeys; i++) {
    const oo_cand *kp = &c[P.keys[i]];
    int q;
    if (kp->x < (float)mx)
      q = (kp->y < (float)my) ? 0 : 2;
    else
      q = (kp->y < (float)my) ? 1 : 3;
    k[q][nk[q]++] = P.keys[i];
  }
  const int bx0[4] = {P.x0, mx, P.x0, mx}, bx1[4] = {mx, P.x1, mx, P.x1};
  const int by0[4] = {P.y0, P.y0, my, my}, by1[4] = {my, my, P.y1, P.y1};
  for (int q = 0; q < 4; q++) {
    if (nk[q] == 0) {
      free(k[q]);
      child[q] = -1;
      continue;
    }
    int id = ql_new(L);
    qnode *n = &L->n[id];
    n->x0 = bx0[q];
    n->x1 = bx1[q];
    n->y0 = by0[q];
    n->y1 = by1[q];
    n->keys = k[q];
    n->nkeys = nk[q];
    n->nomore = (nk[q] == 1);
    child[q] = id;
  }
}

typedef struct {
  int size, node;
} qpair;
/* std::sort on pair<int, ExtractorNode*> (:432-433): ties are ordered by heap address in the
 * reference (allocator dependent).  Oracle rule: a later-created node compares
 * greater (pool index stands in for the address). */
static int qpair_cmp(const void *a, const void *b) {
  const qpair *A = (const qpair *)a, *B = (const qpair *)b;
  if (A->size != B->size) return A->size < B->size ? -1 : 1;
  return A->node < B->node ? -1 : (A->node > B->node);
}

int oo_distribute_octtree(const oo_cand *cands, int n, int minX, int maxX, int minY, int maxY, int nFeatures,
                          float *out_x, float *out_y, float *out_resp, int cap, int *tie_ambiguous) {
  if (tie_ambiguous) *tie_ambiguous = 0;
  const int nIni = (int)round((double)(maxX - minX) / (maxY - minY));
  if (nIni < 1) return -4;
  const float hX = (float)(maxX - minX) / nIni;
  qlist L;
  memset(&L, 0, sizeof(L));
  L.head = L.tail = -1;
  int *ini = (int *)malloc(sizeof(int) * (size_t)nIni);
  for (int i = 0; i < nIni; i++) {
    int id = ql_new(&L);
    qnode *q = &L.n[id];
    q->x0 = (int)(hX * (float)i);
    q->x1 = (int)(hX * (float)(i + 1));
    q->y0 = 0;
    q->y1 = maxY - minY;
    q->keys = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    ql_push_back(&L, id);
    ini[i] = id;
  }
  for (int i = 0; i < n; i++) {
    int b = (int)(cands[i].x / hX);
    if (b < 0) b = 0;
    if (b >= nIni) b = nIni - 1; /* reference would index out of bounds */
    qnode *q = &L.n[ini[b]];
    q->keys[q->nkeys++] = i;
  }
  free(ini);
  for (int it = L.head; it >= 0;) {
    if (L.n[it].nkeys == 1) {
      L.n[it].nomore = 1;
      it = L.n[it].next;
    } else if (L.n[it].nkeys == 0)
      it = ql_erase(&L, it);
    else
      it = L.n[it].next;
  }
  int finish = 0;
  qpair *vs = NULL, *vprev = NULL;
  int nvs = 0, vscap = 0, nvprev = 0, vprevcap = 0;
#define VS_PUSH(sz, nd)                                                  \
  do {                                                                   \
    if (nvs == vscap) {                                                  \
      vscap = vscap ? vscap * 2 : 256;                                   \
      vs = (qpair *)realloc(vs, sizeof(qpair) * (size_t)vscap);          \
    }                                                                    \
    vs[nvs].size = (sz);                                                 \
    vs[nvs].node = (nd);                                                 \
    nvs++;                                                               \
  } while (0)
  while (!finish) {
    int prevSize = L.size;
    int nToExpand = 0;
    nvs = 0;
    for (int it = L.head; it >= 0;) {
      if (L.n[it].nomore) {
        it = L.n[it].next;
        continue;
      }
      int ch[4];
      q_divide(&L, it, cands, ch);
      for (int q = 0; q < 4; q++) {
        if (ch[q] < 0) continue;
        ql_push_front(&L, ch[q]);
        if (L.n[ch[q]].nkeys > 1) {
          nToExpand++;
          VS_PUSH(L.n[ch[q]].nkeys, ch[q]);
        }
      }
      it = ql_erase(&L, it);
    }
    if (L.size >= nFeatures || L.size == prevSize) {
      finish = 1;
    } else if (L.size + nToExpand * 3 > nFeatures) {
      while (!finish) {
        prevSize = L.size;
        if (nvs > vprevcap) {
          vprevcap = nvs;
          vprev = (qpair *)realloc(vprev, sizeof(qpair) * (size_t)vprevcap);
        }
        memcpy(vprev, vs, sizeof(qpair) * (size_t)nvs);
        nvprev = nvs;
        nvs = 0;
        qsort(vprev, (size_t)nvprev, sizeof(qpair), qpair_cmp);
        if (tie_ambiguous)
          for (int j = 1; j < nvprev; j++)
            if (vprev[j].size == vprev[j - 1].size) *tie_ambiguous |= 2; /* list order depends on the tie */
        for (int j = nvprev - 1; j >= 0; j--) {
          int ch[4];
          q_divide(&L, vprev[j].node, cands, ch);
          for (int q = 0; q < 4; q++) {
            if (ch[q] < 0) continue;
            ql_push_front(&L, ch[q]);
            if (L.n[ch[q]].nkeys > 1) VS_PUSH(L.n[ch[q]].nkeys, ch[q]);
          }
          ql_erase(&L, vprev[j].node);
          if (L.size >= nFeatures) {
            if (tie_ambiguous && j > 0 && vprev[j - 1].size == vprev[j].size)
              *tie_ambiguous |= 1; /* the SET of keypoints depends on the tie */
            break;
          }
        }
        if (L.size >= nFeatures || L.size == prevSize) finish = 1;
      }
    }
  }
#undef VS_PUSH
  /* retain the best point of each node, :486-501 (first max, strict >) */
  int nout = 0;
  for (int it = L.head; it >= 0; it = L.n[it].next) {
    const qnode *q = &L.n[it];
    int best = q->keys[0];
    float maxResponse = cands[best].response;
    for (int k = 1; k < q->nkeys; k++)
      if (cands[q->keys[k]].response > maxResponse) {
        best = q->keys[k];
        maxResponse = cands[best].response;
      }
    if (nout < cap) {
      out_x[nout] = cands[best].x;
      out_y[nout] = cands[best].y;
      if (out_resp) out_resp[nout] = maxResponse;
    }
    nout++;
  }
  for (int i = 0; i < L.count; i++) free(L.n[i].keys);
  free(L.n);
  free(vs);
  free(vprev);
  return nout;
}

/* ------------------------------------------------------------------------------------------ */
/* Extract, orb_feature_extractor.cpp:530-600                                                  */
static int extract_impl(const oo_params *p, const uint8_t *img, int stride, int n_features, oo_keypoint *kps,
                        uint8_t *desc, int cap, oo_dump *dump, int use_literal) {
  if (!img || p->width <= 0 || p->height <= 0) return -1; /* image.empty(), :533 */
  oo_geometry g;
  int rc = oo_geometry_compute(p, n_features, &g);
  if (rc) return rc;
  const int L = g.levels;
  uint8_t *pyr[OO_MAX_LEVELS];
  size_t total_px = 0;
  for (int l = 0; l < L; l++) total_px += (size_t)g.w[l] * g.h[l];
  uint8_t *store = (uint8_t *)malloc(total_px);
  {
    size_t off = 0;
    for (int l = 0; l < L; l++) {
      pyr[l] = store + off;
      off += (size_t)g.w[l] * g.h[l];
    }
  }
  /* BuildImagePyramid :504-528 (borders are dead data) */
  for (int y = 0; y < g.h[0]; y++) memcpy(pyr[0] + (size_t)y * g.w[0], img + (size_t)y * stride, (size_t)g.w[0]);
  for (int l = 1; l < L; l++) {
    rc = oo_resize_linear_u8(pyr[l - 1], g.w[l - 1], g.h[l - 1], g.w[l - 1], pyr[l], g.w[l], g.h[l], g.w[l]);
    if (rc) {
      free(store);
      return -5;
    }
  }
  if (dump && dump->pyramid) memcpy(dump->pyramid, store, total_px);

  /* ComputeKeyPointsOctTree :204-298 */
  typedef struct {
    float x, y;
    float angle;
  } lkp;
  lkp *lk[OO_MAX_LEVELS];
  int nk[OO_MAX_LEVELS];
  int total = 0;
  size_t cand_off = 0;
  size_t px_off = 0;
  for (int l = 0; l < L; l++) {
    const int w = g.w[l], h = g.h[l];
    const int candcap = ((w + 1) / 2) * ((h + 1) / 2) + 16;
    oo_cand *cands = (oo_cand *)malloc(sizeof(oo_cand) * (size_t)candcap);
    int nc = oo_detect_level(pyr[l], w, h, w, p->fast_init, p->fast_min, use_literal, cands, candcap);
    if (nc < 0) {
      free(cands);
      free(store);
      return nc;
    }
    if (dump) {
      dump->n_cands[l] = nc;
      if (dump->cands) {
        for (int i = 0; i < nc && cand_off + i < (size_t)dump->cands_cap; i++) dump->cands[cand_off + i] = cands[i];
        cand_off += (size_t)nc;
      }
      if (dump->score) oo_fast_score_map(pyr[l], w, h, w, p->fast_min, dump->score + px_off, w);
    }
    const int outcap = nc + 8; /* the list can never hold more nodes than candidates */
    float *ox = (float *)malloc(sizeof(float) * 2 * (size_t)(outcap > 0 ? outcap : 1)), *oy = ox + outcap;
    int tie = 0;
    int m = oo_distribute_octtree(cands, nc, OO_EDGE - 3, w - OO_EDGE + 3, OO_EDGE - 3, h - OO_EDGE + 3,
                                  g.budget[l], ox, oy, NULL, outcap, &tie);
    if (m > outcap) m = outcap;
    if (dump) {
      dump->tie_ambiguous[l] = tie;
      dump->n_kp[l] = m;
    }
    lk[l] = (lkp *)malloc(sizeof(lkp) * (size_t)(m > 0 ? m : 1));
    nk[l] = m;
    for (int i = 0; i < m; i++) {
      lk[l][i].x = ox[i] + (float)(OO_EDGE - 3); /* :288-289 (values are small integers: exact) */
      lk[l][i].y = oy[i] + (float)(OO_EDGE - 3);
      /* computeOrientation :296-297 on the unblurred level */
      lk[l][i].angle = oo_ic_angle(pyr[l], w, cv_round_d((double)lk[l][i].x), cv_round_d((double)lk[l][i].y));
    }
    total += m;
    free(ox);
    free(cands);
    px_off += (size_t)w * h;
  }

  /* descriptors + reversed packing, :541-599 */
  int ret = total;
  if (total > cap) {
    ret = -6;
  } else {
    int stereoIndex = total - 1;
    size_t boff = 0;
    for (int l = 0; l < L; l++) {
      const int w = g.w[l], h = g.h[l];
      if (nk[l] != 0) {
        uint8_t *blur = (uint8_t *)malloc((size_t)w * h);
        oo_gaussian_blur7(pyr[l], w, h, w, blur, w);
        if (dump && dump->blurred) memcpy(dump->blurred + boff, blur, (size_t)w * h);
        const float scale = (float)g.scale[l]; /* :568 */
        for (int i = 0; i < nk[l]; i++) {
          oo_keypoint k;
          k.x = (double)lk[l][i].x;
          k.y = (double)lk[l][i].y;
          k.angle = (double)lk[l][i].angle;
          k.size = (double)g.kp_size[l];
          k.level = l;
          k.pad_ = 0;
          oo_orb_descriptor(blur, w, cv_round_d(k.x), cv_round_d(k.y), k.angle, desc + (size_t)stereoIndex * 32);
          if (l != 0) { /* :575-577, Eigen double vector *= float */
            k.x *= (double)scale;
            k.y *= (double)scale;
          }
          kps[stereoIndex] = k;
          stereoIndex--;
        }
        free(blur);
      } else if (dump && dump->blurred) {
        memset(dump->blurred + boff, 0, (size_t)w * h);
      }
      boff += (size_t)w * h;
    }
  }
  for (int l = 0; l < L; l++) free(lk[l]);
  free(store);
  return ret;
}

int oo_extract(const oo_params *p, const uint8_t *img, int stride, int n_features, oo_keypoint *kps,
               uint8_t *desc, int cap, oo_dump *dump) {
  return extract_impl(p, img, stride, n_features, kps, desc, cap, dump, 0);
}

/* same results, but FAST runs as the reference runs it: literal cv::FAST per cell with its early
 * rejection (this is the variant bench.py times as the CPU baseline) */
int oo_extract_literal(const oo_params *p, const uint8_t *img, int stride, int n_features, oo_keypoint *kps,
                       uint8_t *desc, int cap) {
  return extract_impl(p, img, stride, n_features, kps, desc, cap, NULL, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* matching                                                                                    */
unsigned oo_hamming256(const uint8_t *a, const uint8_t *b) {
  unsigned dist = 0;
  for (unsigned i = 0; i < 8; ++i) {
    uint32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    unsigned v = pa ^ pb;
    v -= ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

static inline int window_pred(double cx, double cy, int level, double px, double py, size_t window, int minLevel,
                              int maxLevel) {
  /* features.cpp:94,109-119 */
  const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  if (bCheckLevels) {
    if (level < minLevel) return 0;
    if (maxLevel >= 0 && level > maxLevel) return 0;
  }
  const double distance_x = cx - px;
  const double distance_y = cy - py;
  return fabs(distance_x) < (double)window && fabs(distance_y) < (double)window;
}

int oo_list_features_in_area(const oo_keypoint *kps, int n, double px, double py, size_t window, int min_level,
                             int max_level, int32_t *out_idx) {
  int m = 0;
  for (int j = 0; j < n; j++)
    if (window_pred(kps[j].x, kps[j].y, kps[j].level, px, py, window, min_level, max_level)) out_idx[m++] = j;
  return m;
}

/* features.cpp:143-161, the body BEHIND the early `return true` of :146-148: the 64 x 48 grid the
 * reference means to have.  constants.h:18-19; grid_element_*_inv_ = features.cpp:13-16. */
void oo_pos_in_grid(double kx, double ky, double image_width, double image_height, int *posX, int *posY) {
  const double grid_element_width_inv = (double)OO_GRID_COLS / image_width;
  const double grid_element_height_inv = (double)OO_GRID_ROWS / image_height;
  double x = kx > 0. ? kx : 0.; /* std::max(kp.x(), 0.) */
  x = x < image_width ? x : image_width;
  double y = ky > 0. ? ky : 0.;
  y = y < image_height ? y : image_height;
  size_t cx = (size_t)(x * grid_element_width_inv);
  size_t cy = (size_t)(y * grid_element_height_inv);
  *posX = (int)(cx < (size_t)(OO_GRID_COLS - 1) ? cx : (size_t)(OO_GRID_COLS - 1));
  *posY = (int)(cy < (size_t)(OO_GRID_ROWS - 1) ? cy : (size_t)(OO_GRID_ROWS - 1));
}

/* features.cpp:127-133 with the working PosInGrid: grid[pos_X][pos_Y].push_back(i) for ascending i.  Flat layout of
 * `std::vector<size_t> grid[COLS][ROWS]` (features.h:36): cell = pos_X * ROWS + pos_Y; items of cell c are
 * items[cell_start[c] .. cell_start[c + 1]). */
void oo_assign_features_to_grid(const double *xy, int n, double image_width, double image_height,
                                int32_t *cell_start, int32_t *items) {
  const int ncell = OO_GRID_COLS * OO_GRID_ROWS;
  int32_t *cell_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
  for (int c = 0; c <= ncell; c++) cell_start[c] = 0;
  for (int i = 0; i < n; i++) {
    int px, py;
    oo_pos_in_grid(xy[2 * i], xy[2 * i + 1], image_width, image_height, &px, &py);
    cell_of[i] = px * OO_GRID_ROWS + py;
    cell_start[cell_of[i] + 1]++;
  }
  for (int c = 0; c < ncell; c++) cell_start[c + 1] += cell_start[c];
  int32_t *cursor = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncell);
  memcpy(cursor, cell_start, sizeof(int32_t) * (size_t)ncell);
  for (int i = 0; i < n; i++) items[cursor[cell_of[i]]++] = i;
  free(cursor);
  free(cell_of);
}

/* features.cpp:81-125 as written, over the working grid: cells ix outer, iy inner, members in push_back order.
 * Same SET as oo_list_features_in_area (the degenerate grid visits ascending j); the order differs. */
int oo_list_features_in_area_grid(const double *xy, const int32_t *level, const int32_t *cell_start,
                                  const int32_t *items, double image_width, double image_height, double px, double py,
                                  size_t window, int min_level, int max_level, int32_t *out_idx) {
  int min_cell_x, min_cell_y, max_cell_x, max_cell_y;
  oo_pos_in_grid(px - (double)window, py - (double)window, image_width, image_height, &min_cell_x, &min_cell_y);
  oo_pos_in_grid(px + (double)window, py + (double)window, image_width, image_height, &max_cell_x, &max_cell_y);
  int m = 0;
  for (int ix = min_cell_x; ix <= max_cell_x; ix++)
    for (int iy = min_cell_y; iy <= max_cell_y; iy++) {
      const int c = ix * OO_GRID_ROWS + iy;
      for (int k = cell_start[c]; k < cell_start[c + 1]; k++) {
        const int j = items[k];
        if (window_pred(xy[2 * j], xy[2 * j + 1], level[j], px, py, window, min_level, max_level)) out_idx[m++] = j;
      }
    }
  return m;
}

int oo_snn_match(const oo_match_args *a, int32_t *match_from, uint32_t *match_dist) {
  /* second_nearest_neighbor_matcher.hpp:30-80 */
  uint32_t *best_from_dist = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(a->n_from ? a->n_from : 1));
  int32_t *from_to = (int32_t *)malloc(sizeof(int32_t) * (size_t)(a->n_from ? a->n_from : 1));
  for (int j = 0; j < a->n_from; j++) from_to[j] = -1;
  for (int i = 0; i < a->n_to; i++) match_from[i] = -1;
  int nmatches = 0;
  for (int i = 0; i < a->n_to; i++) {
    if (a->to_active && !a->to_active[i]) continue;
    unsigned best = UINT_MAX, second = UINT_MAX;
    int best_j = -1;
    const uint8_t *d1 = a->to_desc + (size_t)i * 32;
    for (int j = 0; j < a->n_from; j++) {
      if (a->to_xy) {
        if (!window_pred(a->from_xy[2 * j], a->from_xy[2 * j + 1], a->from_level[j], a->to_xy[2 * i],
                         a->to_xy[2 * i + 1], (size_t)a->to_window[i], a->to_min_level[i], a->to_max_level[i]))
          continue;
      }
      unsigned distance = oo_hamming256(d1, a->from_desc + (size_t)j * 32);
      if (distance > a->threshold) continue;
      if (distance < best) {
        best_j = j;
        second = best;
        best = distance;
      } else if (distance < second)
        second = distance;
    }
    if ((double)best < a->ratio * (double)second) {
      if (from_to[best_j] >= 0) {
        if (best_from_dist[best_j] > best) {
          match_from[from_to[best_j]] = -1;
          nmatches--;
        } else
          continue;
      }
      best_from_dist[best_j] = best;
      from_to[best_j] = i;
      match_from[i] = best_j;
      if (match_dist) match_dist[i] = best;
      nmatches++;
    }
  }
  free(best_from_dist);
  free(from_to);
  return nmatches;
}

static int cam_undistort3(const oo_camera *c, const double d[3], double u[3]) {
  if (c->model == 0) { /* Barrel5::UnDistortPoint, barrel5.cpp:44-70 */
    const double k1 = c->k[0], k2 = c->k[1], p1 = c->k[2], p2 = c->k[3], k3 = c->k[4];
    double x = d[0], y = d[1];
    const double x0 = d[0], y0 = d[1];
    u[0] = x; u[1] = y; u[2] = 1;
    for (int j = 0; j < 10; j++) {
      double r2 = x * x + y * y;
      double icdist = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      if (icdist < 0) { u[0] = x; u[1] = y; return 0; }
      double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      double xnew = (x0 - deltaX) * icdist;
      double ynew = (y0 - deltaY) * icdist;
      if (fabs(x - xnew) < 1e-7 && fabs(y - ynew) < 1e-7) { u[0] = x; u[1] = y; return 1; }
      x = xnew;
      y = ynew;
    }
    u[0] = -10000; u[1] = 10000; u[2] = -1;
    return 0;
  }
  /* FishEye::UnDistortPoint, fish_eye.cpp:68-116 */
  const double k1 = c->k[0], k2 = c->k[1], k3 = c->k[2], k4 = c->k[3];
  double theta_d = sqrt(d[0] * d[0] + d[1] * d[1]);
  theta_d = fmin(M_PI_2, fmax(-M_PI_2, theta_d));
  double theta = theta_d;
  int converged = 0;
  for (int i = 0; i < 10; ++i) {
    double theta_g2 = theta * theta;
    double theta_g3 = theta_g2 * theta;
    double theta_g4 = theta_g3 * theta;
    double theta_g5 = theta_g4 * theta;
    double theta_g6 = theta_g5 * theta;
    double theta_g7 = theta_g6 * theta;
    double theta_g8 = theta_g7 * theta;
    double theta_g9 = theta_g8 * theta;
    double fp = 1 + 3 * k1 * theta_g2 + 5 * k2 * theta_g4 + 7 * k3 * theta_g6 + 9 * k4 * theta_g8;
    double theta_fix = (theta + k1 * theta_g3 + k2 * theta_g5 + k3 * theta_g7 + k4 * theta_g9 - theta_d) / fp;
    theta -= theta_fix;
    converged = fabs(theta_fix) < 7e-8;
    if (converged) break;
  }
  if (!converged) { u[0] = -10000; u[1] = -10000; u[2] = -1; return 0; }
  double scale = tan(theta) / theta_d;
  int flipped = ((theta_d < 0 && theta > 0) || (theta_d > 0 && theta < 0));
  if (converged && !flipped && scale > 0) {
    u[0] = d[0] * scale; u[1] = d[1] * scale; u[2] = 1;
    return 1;
  }
  u[0] = -10000; u[1] = -10000; u[2] = -1;
  return 0;
}

void oo_keypoint_geometry(const oo_keypoint *kp, int n, const oo_camera *c, double *undist, double *unproj, uint8_t *ok) {
  const double fx_inv = c->fx ? 1 / c->fx : 0, fy_inv = c->fy ? 1 / c->fy : 1; /* monocular_camera.h:107-114 */
  for (int i = 0; i < n; i++) {
    const double px = kp[i].x, py = kp[i].y;
    const double up[3] = {(px - c->cx) * fx_inv, (py - c->cy) * fy_inv, 1}; /* UnprojectPoint :31-33 */
    uint8_t flags = 0;
    /* UndistortPoint :49-61 */
    if (c->model < 0) {
      undist[2 * i] = px; undist[2 * i + 1] = py;
      flags |= 1;
    } else {
      double u[3];
      if (cam_undistort3(c, up, u)) {
        double z_inv = 1 / u[2]; /* ProjectPoint :35-38 */
        undist[2 * i] = u[0] * z_inv * c->fx + c->cx;
        undist[2 * i + 1] = u[1] * z_inv * c->fy + c->cy;
        flags |= 1;
      } else {
        undist[2 * i] = 0; undist[2 * i + 1] = 0;
      }
    }
    /* UnprojectAndUndistort :40-47 */
    if (c->model < 0) {
      unproj[3 * i] = up[0]; unproj[3 * i + 1] = up[1]; unproj[3 * i + 2] = up[2];
      flags |= 2;
    } else {
      double u[3];
      if (cam_undistort3(c, up, u)) flags |= 2;
      unproj[3 * i] = u[0]; unproj[3 * i + 1] = u[1]; unproj[3 * i + 2] = u[2];
    }
    ok[i] = flags;
  }
}

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

void oo_distinctive_descriptor(const uint8_t *desc, const int32_t *offsets, int n_points, int32_t *best) {
  for (int p = 0; p < n_points; p++) {
    const int N = offsets[p + 1] - offsets[p];
    best[p] = -1;
    if (N == 0) continue; /* :146-148 */
    const uint8_t *d = desc + (size_t)offsets[p] * 32;
    int *distances = (int *)malloc(sizeof(int) * (size_t)N * (size_t)N);
    for (int i = 0; i < N; i++) {
      distances[i * N + i] = 0;
      for (int j = i + 1; j < N; j++) {
        distances[i * N + j] = (int)oo_hamming256(d + (size_t)i * 32, d + (size_t)j * 32);
        distances[j * N + i] = distances[i * N + j];
      }
    }
    int best_median = INT_MAX, best_idx = 0;
    int *dists = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; i++) {
      memcpy(dists, distances + (size_t)i * N, sizeof(int) * (size_t)N);
      qsort(dists, (size_t)N, sizeof(int), cmp_int);
      const int median = dists[(size_t)(0.5 * (N - 1))]; /* :160 */
      if (median < best_median) {
        best_median = median;
        best_idx = i;
      }
    }
    best[p] = best_idx;
    free(dists);
    free(distances);
  }
}

void oo_bow_transform(const uint8_t *desc, int n, const int32_t *child_off, const uint32_t *children,
                      const uint8_t *node_desc, const uint32_t *word_id, int L, int levelsup, uint32_t *out_word,
                      uint32_t *out_node) {
  const int nid_level = L - levelsup; /* :1226 */
  for (int i = 0; i < n; i++) {
    const uint8_t *f = desc + (size_t)i * 32;
    uint32_t nid = 0; /* root when nid_level <= 0, :1227 */
    uint32_t final_id = 0;
    int current_level = 0;
    do {
      ++current_level;
      const int c0 = child_off[final_id], c1 = child_off[final_id + 1];
      final_id = children[c0];
      double best_d = (double)oo_hamming256(f, node_desc + (size_t)final_id * 32);
      for (int c = c0 + 1; c < c1; c++) {
        const uint32_t id = children[c];
        const double d = (double)oo_hamming256(f, node_desc + (size_t)id * 32);
        if (d < best_d) {
          best_d = d;
          final_id = id;
        }
      }
      if (current_level == nid_level) nid = final_id;
    } while (child_off[final_id + 1] > child_off[final_id]); /* !isLeaf() */
    out_word[i] = word_id[final_id];
    out_node[i] = nid;
  }
}

/* lower_bound over ascending node ids: first k in [lo, n) with nodes[k] >= key (std::map::lower_bound) */
static int node_lower_bound(const uint32_t *nodes, int lo, int n, uint32_t key) {
  int k = lo;
  while (k < n && nodes[k] < key) k++;
  return k;
}

int oo_match_bow(const uint8_t *desc_to, int n_to, const uint8_t *desc_from, int n_from, const uint32_t *to_nodes,
                 const int32_t *to_off, const uint32_t *to_ids, int n_to_nodes, const uint32_t *from_nodes,
                 const int32_t *from_off, const uint32_t *from_ids, int n_from_nodes, unsigned threshold, double ratio,
                 int32_t *match_from, uint32_t *match_dist) {
  uint32_t *best_from_dist = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n_from ? n_from : 1));
  int32_t *from_to = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_from ? n_from : 1));
  for (int j = 0; j < n_from; j++) from_to[j] = -1;
  for (int i = 0; i < n_to; i++) match_from[i] = -1;
  int nmatches = 0;
  int kt = 0, kf = 0; /* bow_it_to_, bow_it_from_ */
  /* BowToIterator::AdvanceUntilSameNode / operator++, bow_to_iterator.cpp:12-47 */
  while (kt < n_to_nodes && kf < n_from_nodes) {
    if (to_nodes[kt] != from_nodes[kf]) {
      if (to_nodes[kt] < from_nodes[kf])
        kt = node_lower_bound(to_nodes, 0, n_to_nodes, from_nodes[kf]);
      else
        kf = node_lower_bound(from_nodes, 0, n_from_nodes, to_nodes[kt]);
      continue;
    }
    /* same node: every 'to' id of the node, candidates = the node's 'from' ids (bow_to_pointee.h:37-45) */
    for (int s = to_off[kt]; s < to_off[kt + 1]; s++) {
      const int i = (int)to_ids[s];
      unsigned best = UINT_MAX, second = UINT_MAX;
      int best_j = -1;
      const uint8_t *d1 = desc_to + (size_t)i * 32;
      for (int c = from_off[kf]; c < from_off[kf + 1]; c++) {
        const int j = (int)from_ids[c];
        unsigned distance = oo_hamming256(d1, desc_from + (size_t)j * 32);
        if (distance > threshold) continue;
        if (distance < best) {
          best_j = j;
          second = best;
          best = distance;
        } else if (distance < second)
          second = distance;
      }
      if ((double)best < ratio * (double)second) {
        if (from_to[best_j] >= 0) {
          if (best_from_dist[best_j] > best) {
            match_from[from_to[best_j]] = -1;
            nmatches--;
          } else
            continue;
        }
        best_from_dist[best_j] = best;
        from_to[best_j] = i;
        match_from[i] = best_j;
        if (match_dist) match_dist[i] = best;
        nmatches++;
      }
    }
    kt++; /* operator++ past the node's last id: both map iterators advance, :21-25 */
    kf++;
  }
  free(best_from_dist);
  free(from_to);
  return nmatches;
}

int oo_match_area(const oo_keypoint *kp_to, const uint8_t *desc_to, int n_to, const oo_keypoint *kp_from,
                  const uint8_t *desc_from, int n_from, size_t window, unsigned threshold, double ratio,
                  int32_t *match_from, uint32_t *match_dist) {
  size_t nt = (size_t)(n_to ? n_to : 1), nf = (size_t)(n_from ? n_from : 1);
  double *to_xy = (double *)malloc(sizeof(double) * 2 * nt);
  double *from_xy = (double *)malloc(sizeof(double) * 2 * nf);
  uint64_t *win = (uint64_t *)malloc(sizeof(uint64_t) * nt);
  int32_t *lmin = (int32_t *)malloc(sizeof(int32_t) * nt), *lmax = (int32_t *)malloc(sizeof(int32_t) * nt);
  int32_t *flev = (int32_t *)malloc(sizeof(int32_t) * nf);
  uint8_t *active = (uint8_t *)malloc(nt);
  for (int i = 0; i < n_to; i++) {
    to_xy[2 * i] = kp_to[i].x;
    to_xy[2 * i + 1] = kp_to[i].y;
    win[i] = window;
    lmin[i] = 0; /* area_to_pointee.h:44-48 */
    lmax[i] = 0;
    /* area_to_iterator.h:24-43: id 0 is visited whatever its level, later ids only when level == 0 */
    active[i] = (i == 0) || (kp_to[i].level == 0);
  }
  for (int j = 0; j < n_from; j++) {
    from_xy[2 * j] = kp_from[j].x;
    from_xy[2 * j + 1] = kp_from[j].y;
    flev[j] = kp_from[j].level;
  }
  oo_match_args a;
  memset(&a, 0, sizeof(a));
  a.to_desc = desc_to; a.n_to = n_to; a.from_desc = desc_from; a.n_from = n_from;
  a.to_xy = to_xy; a.to_window = win; a.to_min_level = lmin; a.to_max_level = lmax; a.to_active = active;
  a.from_xy = from_xy; a.from_level = flev; a.threshold = threshold; a.ratio = ratio;
  int r = oo_snn_match(&a, match_from, match_dist);
  free(to_xy); free(from_xy); free(win); free(lmin); free(lmax); free(flev); free(active);
  return r;
}

int oo_orientation_validate(const double *angle_to, const double *angle_from, int32_t *match_from, int n_to) {
  /* validators/orientation_validator.cpp:16-97 */
  enum { HISTO_LENGTH = 30 };
  const double factor = (double)(1.0f / HISTO_LENGTH); /* :81 */
  int count[HISTO_LENGTH] = {0};
  int *bin_of = (int *)malloc(sizeof(int) * (size_t)(n_to ? n_to : 1));
  for (int i = 0; i < n_to; i++) {
    bin_of[i] = -1;
    if (match_from[i] < 0) continue;
    float rot = (float)(angle_to[i] - angle_from[match_from[i]]);
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)round((double)rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    if (bin < 0 || bin >= HISTO_LENGTH) bin = 0; /* reference asserts */
    bin_of[i] = bin;
    count[bin]++;
  }
  int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
  for (int i = 0; i < HISTO_LENGTH; i++) {
    const int s = count[i];
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s;
      ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s;
      ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s; ind3 = i;
    }
  }
  if ((float)max2 < 0.1f * (float)max1) {
    ind2 = -1; ind3 = -1;
  } else if ((float)max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
  int discarded = 0;
  for (int i = 0; i < n_to; i++) {
    if (bin_of[i] < 0) continue;
    if (bin_of[i] == ind1 || bin_of[i] == ind2 || bin_of[i] == ind3) continue;
    match_from[i] = -1;
    discarded++;
  }
  free(bin_of);
  return discarded;
}

unsigned oo_predict_scale(double distance, double max_distance, double scale_factor, int levels) {
  /* orb_feature_extractor.cpp:864-873 */
  double ratio = max_distance / distance;
  double v = ceil(log(ratio) / log(scale_factor));
  unsigned scale = (unsigned)(v > 0. ? v : 0.);
  if (scale >= (unsigned)levels) scale = (unsigned)levels - 1;
  return scale;
}

/* ---- keyframe database ------------------------------------------------------------------------------------ */
/* std::map<WordId, WordValue>::lower_bound on an ascending array */
static int bow_lower_bound(const uint32_t *words, int n, uint32_t key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = lo + (hi - lo) / 2;
    if (words[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* {L1,L2,ChiSquare,Bhattacharyya,DotProduct}Scoring::score, DBoW2/DBoW2/ScoringObject.cpp */
static double bow_vector_score(const uint32_t *w1, const double *v1, int n1, const uint32_t *w2, const double *v2, int n2,
                               int scoring) {
  int i1 = 0, i2 = 0;
  double score = 0;
  while (i1 != n1 && i2 != n2) {
    const double vi = v1[i1], wi = v2[i2];
    if (w1[i1] == w2[i2]) {
      switch (scoring) {
        case 0: score += fabs(vi - wi) - fabs(vi) - fabs(wi); break; /* :41 */
        case 1: score += vi * wi; break;                              /* :91 */
        case 2: if (vi + wi != 0.0) score += vi * wi / (vi + wi); break; /* :148 */
        case 4: score += sqrt(vi * wi); break;                        /* :245 */
        case 5: score += vi * wi; break;                              /* :290 */
        default: return NAN;
      }
      ++i1;
      ++i2;
    } else if (w1[i1] < w2[i2]) {
      i1 = bow_lower_bound(w1, n1, w2[i2]); /* v1.lower_bound(v2_it->first) */
    } else {
      i2 = bow_lower_bound(w2, n2, w1[i1]);
    }
  }
  switch (scoring) {
    case 0: score = -score / 2.0; break; /* :65 */
    case 1:
      if (score >= 1) score = 1.0; /* :114-117 */
      else score = 1.0 - sqrt(1.0 - score);
      break;
    case 2: score = 2. * score; break; /* :167 */
    default: break;
  }
  return score;
}

typedef struct { uint32_t word; int32_t kf; uint32_t count; } inv_entry;
static int cmp_inv_entry(const void *a, const void *b) {
  const inv_entry *x = (const inv_entry *)a, *y = (const inv_entry *)b;
  if (x->word != y->word) return x->word < y->word ? -1 : 1;
  return (x->kf > y->kf) - (x->kf < y->kf);
}

void oo_bow_score(const uint32_t *q_words, const double *q_values, const uint32_t *q_counts, int n_q,
                  const int64_t *kf_off, const uint32_t *kf_words, const double *kf_values, const uint32_t *kf_counts,
                  int n_kf, int scoring, uint64_t *shared_words, double *score) {
  /* Append, :53-61: inverted_file_[word][keyframe] = count.  Held as one array sorted by (word, keyframe); the
   * reference's inner maps are ordered by KeyFrame* (allocator dependent), which no integer sum depends on. */
  const int64_t total = n_kf > 0 ? kf_off[n_kf] : 0;
  inv_entry *inv = (inv_entry *)malloc(sizeof(inv_entry) * (size_t)(total > 0 ? total : 1));
  for (int k = 0; k < n_kf; k++)
    for (int64_t i = kf_off[k]; i < kf_off[k + 1]; i++) {
      inv[i].word = kf_words[i];
      inv[i].kf = k;
      inv[i].count = kf_counts ? kf_counts[i] : 1u;
    }
  qsort(inv, (size_t)total, sizeof(inv_entry), cmp_inv_entry);
  /* SearchWordSharingKeyFrames, :96-102 */
  for (int k = 0; k < n_kf; k++) shared_words[k] = 0;
  for (int b = 0; b < n_q; b++) {
    int64_t lo = 0, hi = total; /* first entry of inverted_file_[b.first] */
    while (lo < hi) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (inv[mid].word < q_words[b]) lo = mid + 1; else hi = mid;
    }
    for (int64_t e = lo; e < total && inv[e].word == q_words[b]; e++)
      shared_words[inv[e].kf] += (uint64_t)inv[e].count * (uint64_t)(q_counts ? q_counts[b] : 1u);
  }
  free(inv);
  /* :127-128, for every keyframe */
  for (int k = 0; k < n_kf; k++)
    score[k] = bow_vector_score(q_words, q_values, n_q, kf_words + kf_off[k], kf_values + kf_off[k],
                                (int)(kf_off[k + 1] - kf_off[k]), scoring);
}
