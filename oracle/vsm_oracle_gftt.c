/*
 * vsm_oracle_gftt.c -- CPU restatement of visnav::detectKeypoints
 * (/root/reference/include/visnav/keypoints.h:133-149): OpenCV's
 * goodFeaturesToTrack(image, points, num_features, 0.01, 8) followed by the
 * InBounds(x, y, EDGE_THRESHOLD) filter.  SURVEY.md section 8f row N2 (detection part).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as vsm_oracle.c).
 *
 * The algorithm lives in a third-party dependency that is absent here: OpenCV
 * (imgproc), taken by the reference from the system (`libopencv-dev`,
 * install_dependencies.sh:39; CMakeLists.txt:106 accepts any version, none is pinned).
 * Restated from its published algorithm (Shi & Tomasi 1994; cv::goodFeaturesToTrack):
 *   1. Dx, Dy = 3x3 Sobel of the 8-bit image in float, scaled by 1 / (4 * 3 * 255),
 *      BORDER_REFLECT_101;
 *   2. covariance images Dx*Dx, Dx*Dy, Dy*Dy (float), unnormalised 3x3 box sums
 *      (double accumulators, result float), BORDER_REFLECT_101;
 *   3. eig = (a + c) - sqrt((a - c)^2 + b^2), a = 0.5*sxx, b = sxy, c = 0.5*syy (float);
 *   4. threshold at (float)(max(eig) * qualityLevel) (values <= it become 0);
 *   5. candidates: pixels 1 <= x < w-1, 1 <= y < h-1 with eig != 0 equal to the 3x3
 *      maximum (dilate);
 *   6. sort descending by value, ties by higher pixel address first;
 *   7. greedy pass keeping a candidate iff no kept corner is closer than minDistance
 *      (grid of round(minDistance) cells, 3x3 cell neighbourhood), stop at maxCorners.
 *
 * Parity status: PINNED by the reference's golden vectors: the corner lists of
 * test/ex3_test_data/kd{0,1}.json are the output of detectKeypoints on 0_{0,1}.jpg
 * (test_ex3.cpp:76-83, 187-218) -- 537 and 547 corners, reproduced exactly and in order.
 * Not pinned: last-bit float behaviour of a particular OpenCV build (FMA in its SIMD
 * paths, float vs double box sums) -- on the goldens every such variant tried gives the
 * same lists.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int n) { /* BORDER_REFLECT_101, |offset| <= 1 */
  if (n == 1) return 0;
  if (p < 0) return -p;
  if (p >= n) return 2 * n - 2 - p;
  return p;
}

/* Steps 1-3.  eig: float[h * w]. */
__attribute__((optimize("fp-contract=off"))) int vsm_oracle_gftt_mineig(const uint8_t *img, int w, int h,
                                                                         size_t pitch, float *eig) {
  const size_t n = (size_t)w * h;
  float *dx = (float *)malloc(2 * n * sizeof(float));
  if (!dx) return -1;
  float *dy = dx + n;
  const float f1 = (float)(1.0 / (4.0 * 3.0 * 255.0)), f0 = 2.0f * f1;
#define PIX(y, x) ((float)img[(size_t)reflect101((y), h) * pitch + (size_t)reflect101((x), w)])
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      /* Dx: row filter [-1 0 1] (exact), column filter [f1 f0 f1] */
      const float r0 = PIX(y - 1, x + 1) - PIX(y - 1, x - 1), r1 = PIX(y, x + 1) - PIX(y, x - 1),
                  r2 = PIX(y + 1, x + 1) - PIX(y + 1, x - 1);
      const float dxa = r1 * f0, dxb = (r0 + r2) * f1;
      dx[(size_t)y * w + x] = dxa + dxb;
      /* Dy: row filter [f1 f0 f1] in float, column filter [-1 0 1] */
      const float ta = PIX(y - 1, x) * f0, tb = (PIX(y - 1, x - 1) + PIX(y - 1, x + 1)) * f1;
      const float ba = PIX(y + 1, x) * f0, bb = (PIX(y + 1, x - 1) + PIX(y + 1, x + 1)) * f1;
      const float top = ta + tb, bot = ba + bb;
      dy[(size_t)y * w + x] = bot - top;
    }
#undef PIX
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      double sxx = 0, sxy = 0, syy = 0, cxx[3], cxy[3], cyy[3];
      for (int j = 0; j < 3; j++) {
        const int yy = reflect101(y + j - 1, h);
        double rxx = 0, rxy = 0, ryy = 0;
        for (int i = 0; i < 3; i++) {
          const int xx = reflect101(x + i - 1, w);
          const float gx = dx[(size_t)yy * w + xx], gy = dy[(size_t)yy * w + xx];
          const float pxx = gx * gx, pxy = gx * gy, pyy = gy * gy;
          rxx += pxx;
          rxy += pxy;
          ryy += pyy;
        }
        cxx[j] = rxx;
        cxy[j] = rxy;
        cyy[j] = ryy;
      }
      sxx = (cxx[0] + cxx[1]) + cxx[2];
      sxy = (cxy[0] + cxy[1]) + cxy[2];
      syy = (cyy[0] + cyy[1]) + cyy[2];
      const float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
      const float amc = a - c;
      const float t1 = amc * amc, t2 = b * b;
      const float s = sqrtf(t1 + t2);
      eig[(size_t)y * w + x] = (a + c) - s;
    }
  free(dx);
  return 0;
}

typedef struct {
  float v;
  uint32_t idx;
} cand_t;

static int cand_cmp(const void *pa, const void *pb) { /* value descending, then address descending */
  const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return a->idx > b->idx ? -1 : a->idx < b->idx ? 1 : 0;
}

/*
 * cv::goodFeaturesToTrack(image, corners, max_corners, quality, min_distance) with the
 * defaults the reference uses (no mask, blockSize 3, gradientSize 3, no Harris).
 * Writes up to cap (x, y) pairs as doubles and returns the number of corners found
 * (-1 on allocation failure).  max_corners <= 0 = unlimited.
 */
long vsm_oracle_gftt(const uint8_t *img, int w, int h, size_t pitch, int max_corners, double quality,
                     double min_distance, double *out_xy, size_t cap) {
  const size_t n = (size_t)w * h;
  float *eig = (float *)malloc(n * sizeof(float));
  cand_t *cand = (cand_t *)malloc(n * sizeof(cand_t));
  if (!eig || !cand || vsm_oracle_gftt_mineig(img, w, h, pitch, eig) != 0) {
    free(eig);
    free(cand);
    return -1;
  }
  float mx = eig[0];
  for (size_t i = 1; i < n; i++)
    if (eig[i] > mx) mx = eig[i];
  const float thr = (float)((double)mx * quality); /* cv::threshold takes the double, compares in float */
  size_t nc = 0;
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      const float v = eig[(size_t)y * w + x];
      if (!(v > thr) || v == 0) continue; /* THRESH_TOZERO, then `val != 0` */
      int is_max = 1;
      for (int j = -1; j <= 1 && is_max; j++)
        for (int i = -1; i <= 1; i++) {
          float u = eig[(size_t)(y + j) * w + (x + i)];
          if (!(u > thr)) u = 0;
          if (u > v) {
            is_max = 0;
            break;
          }
        }
      if (is_max) {
        cand[nc].v = v;
        cand[nc].idx = (uint32_t)((size_t)y * w + x);
        nc++;
      }
    }
  qsort(cand, nc, sizeof(cand_t), cand_cmp);
  long count = 0;
  if (min_distance >= 1) {
    const int cell = (int)lrint(min_distance); /* cvRound */
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    /* grid cells as linked lists through `next` */
    int *head = (int *)malloc(((size_t)gw * gh + nc + 1) * sizeof(int));
    uint32_t *kept = (uint32_t *)malloc((nc + 1) * sizeof(uint32_t));
    if (!head || !kept) {
      free(head);
      free(kept);
      free(eig);
      free(cand);
      return -1;
    }
    int *next = head + (size_t)gw * gh;
    for (size_t i = 0; i < (size_t)gw * gh; i++) head[i] = -1;
    const double md2 = min_distance * min_distance;
    for (size_t k = 0; k < nc; k++) {
      const int y = (int)(cand[k].idx / (uint32_t)w), x = (int)(cand[k].idx % (uint32_t)w);
      const int xc = x / cell, yc = y / cell;
      const int x1 = xc - 1 < 0 ? 0 : xc - 1, y1 = yc - 1 < 0 ? 0 : yc - 1;
      const int x2 = xc + 1 > gw - 1 ? gw - 1 : xc + 1, y2 = yc + 1 > gh - 1 ? gh - 1 : yc + 1;
      int good = 1;
      for (int yy = y1; yy <= y2 && good; yy++)
        for (int xx = x1; xx <= x2 && good; xx++)
          for (int m = head[yy * gw + xx]; m >= 0; m = next[m]) {
            const float ddx = (float)x - (float)(kept[m] % (uint32_t)w), ddy = (float)y - (float)(kept[m] / (uint32_t)w);
            if (ddx * ddx + ddy * ddy < md2) {
              good = 0;
              break;
            }
          }
      if (!good) continue;
      kept[count] = cand[k].idx;
      next[count] = head[yc * gw + xc];
      head[yc * gw + xc] = (int)count;
      if ((size_t)count < cap) {
        out_xy[2 * count] = (double)x;
        out_xy[2 * count + 1] = (double)y;
      }
      count++;
      if (max_corners > 0 && count == max_corners) break;
    }
    free(head);
    free(kept);
  } else {
    for (size_t k = 0; k < nc; k++) {
      if ((size_t)count < cap) {
        out_xy[2 * count] = (double)(cand[k].idx % (uint32_t)w);
        out_xy[2 * count + 1] = (double)(cand[k].idx / (uint32_t)w);
      }
      count++;
      if (max_corners > 0 && count == max_corners) break;
    }
  }
  free(eig);
  free(cand);
  return count;
}

/* visnav::detectKeypoints, keypoints.h:133-149: goodFeaturesToTrack(.., num_features, 0.01, 8),
 * then keep the points with InBounds(x, y, EDGE_THRESHOLD = 19). */
long vsm_oracle_detect_keypoints(const uint8_t *img, int w, int h, size_t pitch, int num_features,
                                 double *out_xy, size_t cap) {
  const size_t n = (size_t)w * h;
  double *all = (double *)malloc((2 * n + 2) * sizeof(double));
  if (!all) return -1;
  const long found = vsm_oracle_gftt(img, w, h, pitch, num_features, 0.01, 8.0, all, n);
  long count = 0;
  for (long i = 0; i < found; i++) {
    const float x = (float)all[2 * i], y = (float)all[2 * i + 1];
    if (19 <= x && x < (float)(w - 19) && 19 <= y && y < (float)(h - 19)) { /* keypoints.h:145 */
      if ((size_t)count < cap) {
        out_xy[2 * count] = all[2 * i];
        out_xy[2 * count + 1] = all[2 * i + 1];
      }
      count++;
    }
  }
  free(all);
  return found < 0 ? -1 : count;
}
