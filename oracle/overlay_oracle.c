/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's mesh overlay.  Only tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of the benchmarks may load this; the product path (libbodyfit.so) never does.
 *
 * PARITY UNPINNED: the reference holds no golden images, and OpenCV is not in the container, so the polygon fill
 * below restates cv::fillConvexPoly(..., cv::LINE_AA) of OpenCV 4.x (modules/imgproc/src/drawing.cpp:
 * clipLine, LineAA, FillConvexPoly) from its published algorithm [recalled]; agreement with a real OpenCV build is
 * unverified.
 *
 * Follows /root/reference/include/RenderSMPLMesh.h:
 *   :36-46   projection of every vertex (double arithmetic, float result, Z <= 1e-6 is invalid)
 *   :50-88   face list: validity, camera-space normal, backface cull (n.z >= 0), flat shade, painter depth, integer
 *            pixel corners via std::round
 *   :91-92   sort far to near (std::sort there, whose tie order is unspecified: a stable sort by face index here)
 *   :95-104  fill each triangle with gray round(220*shade), cv::LINE_AA
 *   :106-109 wireframe: cv::polylines(img, {p0, p1, p2, p0}, false, Scalar(40,40,40), 1, LINE_AA) after the fill, i.e.
 *            (imgproc drawing.cpp: PolyLine -> ThickLine with thickness 1 and LINE_AA -> LineAA) three more
 *            anti-aliased segments p0-p1, p1-p2, p2-p0 [recalled]
 *
 * Plain sequential C on purpose: one triangle after the other, one scanline after the other, exactly as the
 * reference draws.  Compile with -ffp-contract=off (the shade is a chain of products and sums).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

typedef struct { int64_t x, y; } pt2l;

typedef struct {
  uint8_t* data;
  int width, height;
  size_t step; /* bytes per row; 3 bytes per pixel */
} image3;

/* drawing.cpp: the 3-tap profile of the anti-aliased line and the slope correction [recalled] */
static const uint8_t kFilter[64] = {
    168, 177, 185, 194, 202, 210, 218, 224, 231, 236, 241, 246, 249, 252, 254, 254,
    254, 254, 252, 249, 246, 241, 236, 231, 224, 218, 210, 202, 194, 185, 177, 168,
    158, 149, 140, 131, 122, 114, 105, 97,  89,  82,  75,  68,  62,  56,  50,  45,
    40,  36,  32,  28,  25,  22,  19,  16,  14,  12,  11,  9,   8,   7,   5,   5};
static const uint8_t kSlopeCorr[32] = {181, 181, 181, 182, 182, 183, 184, 185, 187, 188, 190, 192, 194, 196, 198, 201,
                                       203, 206, 209, 211, 214, 218, 221, 224, 227, 231, 235, 238, 242, 246, 250, 254};

const uint8_t* overlay_oracle_filter_table(void) { return kFilter; }
const uint8_t* overlay_oracle_slope_table(void) { return kSlopeCorr; }

/* cv::clipLine(Size2l, Point2l&, Point2l&) */
static int clip_line(int64_t width, int64_t height, pt2l* p1, pt2l* p2) {
  int c1, c2;
  const int64_t right = width - 1, bottom = height - 1;
  if (width <= 0 || height <= 0) return 0;
  int64_t x1 = p1->x, y1 = p1->y, x2 = p2->x, y2 = p2->y;
  c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (int64_t)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (int64_t)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (int64_t)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (int64_t)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  p1->x = x1; p1->y = y1; p2->x = x2; p2->y = y2;
  return (c1 | c2) == 0;
}

/* ICV_PUT_POINT of the 3-channel branch: the blend is applied twice */
static void put_point(image3* im, int x, int y, const int col[3], int a) {
  uint8_t* t = im->data + (size_t)y * im->step + (size_t)x * 3;
  for (int c = 0; c < 3; ++c) {
    int v = t[c];
    v += ((col[c] - v) * a + 127) >> 8;
    v += ((col[c] - v) * a + 127) >> 8;
    t[c] = (uint8_t)v;
  }
}

/* drawing.cpp LineAA, 8-bit 3-channel branch; endpoints in 16.16 fixed point */
static void line_aa(image3* im, pt2l pt1, pt2l pt2, const int col[3]) {
  int64_t dx, dy, ax, ay, x_step, y_step, i, j;
  int ecount, scount = 0, slope;
  int ep_table[9];
  if (!clip_line((int64_t)im->width << XY_SHIFT, (int64_t)im->height << XY_SHIFT, &pt1, &pt2)) return;
  dx = pt2.x - pt1.x;
  dy = pt2.y - pt1.y;
  j = dx < 0 ? -1 : 0;
  ax = (dx ^ j) - j;
  i = dy < 0 ? -1 : 0;
  ay = (dy ^ i) - i;
  if (ax > ay) {
    dy = (dy ^ j) - j;
    pt1.x ^= pt2.x & j; pt2.x ^= pt1.x & j; pt1.x ^= pt2.x & j;
    pt1.y ^= pt2.y & j; pt2.y ^= pt1.y & j; pt1.y ^= pt2.y & j;
    x_step = XY_ONE;
    y_step = (dy * XY_ONE) / (ax | 1);   /* (dy << XY_SHIFT) with a well-defined sign */
    pt2.x += XY_ONE;
    ecount = (int)((pt2.x >> XY_SHIFT) - (pt1.x >> XY_SHIFT));
    j = -(pt1.x & (XY_ONE - 1));
    pt1.y += ((y_step * j) >> XY_SHIFT) + (XY_ONE >> 1);
    slope = (int)((y_step >> (XY_SHIFT - 5)) & 0x3f);
    slope ^= (y_step < 0 ? 0x3f : 0);
    i = (pt1.x >> (XY_SHIFT - 7)) & 0x78;
    j = (pt2.x >> (XY_SHIFT - 7)) & 0x78;
  } else {
    dx = (dx ^ i) - i;
    pt1.x ^= pt2.x & i; pt2.x ^= pt1.x & i; pt1.x ^= pt2.x & i;
    pt1.y ^= pt2.y & i; pt2.y ^= pt1.y & i; pt1.y ^= pt2.y & i;
    x_step = (dx * XY_ONE) / (ay | 1);
    y_step = XY_ONE;
    pt2.y += XY_ONE;
    ecount = (int)((pt2.y >> XY_SHIFT) - (pt1.y >> XY_SHIFT));
    j = -(pt1.y & (XY_ONE - 1));
    pt1.x += ((x_step * j) >> XY_SHIFT) + (XY_ONE >> 1);
    slope = (int)((x_step >> (XY_SHIFT - 5)) & 0x3f);
    slope ^= (x_step < 0 ? 0x3f : 0);
    i = (pt1.y >> (XY_SHIFT - 7)) & 0x78;
    j = (pt2.y >> (XY_SHIFT - 7)) & 0x78;
  }
  slope = (slope & 0x20) ? 0x100 : kSlopeCorr[slope];
  {
    const int t0 = slope << 7;
    const int t1 = ((0x78 - (int)i) | 4) * slope;
    const int t2 = ((int)j | 4) * slope;
    ep_table[0] = 0;
    ep_table[8] = slope;
    ep_table[1] = ep_table[3] = ((((int)(j - i) & 0x78) | 4) * slope >> 8) & 0x1ff;
    ep_table[2] = (t1 >> 8) & 0x1ff;
    ep_table[4] = ((((int)(j - i) + 0x80) | 4) * slope >> 8) & 0x1ff;
    ep_table[5] = ((t1 + t0) >> 8) & 0x1ff;
    ep_table[6] = (t2 >> 8) & 0x1ff;
    ep_table[7] = ((t2 + t0) >> 8) & 0x1ff;
  }
  if (ax > ay) {
    int x = (int)(pt1.x >> XY_SHIFT);
    for (; ecount >= 0; x++, pt1.y += y_step, scount++, ecount--) {
      if ((unsigned)x >= (unsigned)im->width) continue;
      const int y = (int)((pt1.y >> XY_SHIFT) - 1);
      const int ep = ep_table[(((scount >= 2) + 1) & (scount | 2)) * 3 + (((ecount >= 2) + 1) & (ecount | 2))];
      const int dist = (int)((pt1.y >> (XY_SHIFT - 5)) & 31);
      int a = (ep * kFilter[dist + 32] >> 8) & 0xff;
      if ((unsigned)y < (unsigned)im->height) put_point(im, x, y, col, a);
      a = (ep * kFilter[dist] >> 8) & 0xff;
      if ((unsigned)(y + 1) < (unsigned)im->height) put_point(im, x, y + 1, col, a);
      a = (ep * kFilter[63 - dist] >> 8) & 0xff;
      if ((unsigned)(y + 2) < (unsigned)im->height) put_point(im, x, y + 2, col, a);
    }
  } else {
    int y = (int)(pt1.y >> XY_SHIFT);
    for (; ecount >= 0; y++, pt1.x += x_step, scount++, ecount--) {
      if ((unsigned)y >= (unsigned)im->height) continue;
      const int x = (int)((pt1.x >> XY_SHIFT) - 1);
      const int ep = ep_table[(((scount >= 2) + 1) & (scount | 2)) * 3 + (((ecount >= 2) + 1) & (ecount | 2))];
      const int dist = (int)((pt1.x >> (XY_SHIFT - 5)) & 31);
      int a = (ep * kFilter[dist + 32] >> 8) & 0xff;
      if ((unsigned)x < (unsigned)im->width) put_point(im, x, y, col, a);
      a = (ep * kFilter[dist] >> 8) & 0xff;
      if ((unsigned)(x + 1) < (unsigned)im->width) put_point(im, x + 1, y, col, a);
      a = (ep * kFilter[63 - dist] >> 8) & 0xff;
      if ((unsigned)(x + 2) < (unsigned)im->width) put_point(im, x + 2, y, col, a);
    }
  }
}

/* drawing.cpp FillConvexPoly with line_type = LINE_AA, shift = 0, npts = 3 */
static void fill_convex_tri_aa(image3* im, const pt2l v[3], const int col[3]) {
  struct { int idx, di; int64_t x, dx; int ye; } edge[2];
  const int npts = 3;
  int i, y, imin = 0, edges = npts;
  int64_t xmin, xmax, ymin, ymax;
  const int delta1 = XY_ONE - 1, delta2 = 0;   /* LINE_AA: strictly interior span; the edges are the AA lines */
  pt2l p0;
  xmin = xmax = v[0].x;
  ymin = ymax = v[0].y;
  p0 = v[npts - 1];
  p0.x *= XY_ONE; p0.y *= XY_ONE;
  for (i = 0; i < npts; i++) {
    pt2l p = v[i];
    if (p.y < ymin) { ymin = p.y; imin = i; }
    if (p.y > ymax) ymax = p.y;
    if (p.x > xmax) xmax = p.x;
    if (p.x < xmin) xmin = p.x;
    p.x *= XY_ONE; p.y *= XY_ONE;
    line_aa(im, p0, p, col);
    p0 = p;
  }
  if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= im->width || (int)ymin >= im->height) return;
  if (ymax > im->height - 1) ymax = im->height - 1;
  edge[0].idx = edge[1].idx = imin;
  edge[0].ye = edge[1].ye = y = (int)ymin;
  edge[0].di = 1;
  edge[1].di = npts - 1;
  edge[0].x = edge[1].x = -XY_ONE;
  edge[0].dx = edge[1].dx = 0;
  do {
    if (y < (int)ymax || y == (int)ymin) {
      for (i = 0; i < 2; i++) {
        if (y >= edge[i].ye) {
          int idx0 = edge[i].idx, di = edge[i].di;
          int idx = idx0 + di;
          if (idx >= npts) idx -= npts;
          int ty = 0;
          for (; edges-- > 0;) {
            ty = (int)v[idx].y;
            if (ty > y) {
              const int64_t xs = v[idx0].x * XY_ONE, xe = v[idx].x * XY_ONE;
              edge[i].ye = ty;
              edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
              edge[i].x = xs;
              edge[i].idx = idx;
              break;
            }
            idx0 = idx;
            idx += di;
            if (idx >= npts) idx -= npts;
          }
        }
      }
    }
    if (edges < 0) break;
    if (y >= 0) {
      int left = 0, right = 1;
      if (edge[0].x > edge[1].x) { left = 1; right = 0; }
      int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
      int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
      if (xx2 >= 0 && xx1 < im->width) {
        if (xx1 < 0) xx1 = 0;
        if (xx2 >= im->width) xx2 = im->width - 1;
        uint8_t* row = im->data + (size_t)y * im->step;
        for (int x = xx1; x <= xx2; ++x) {
          row[3 * x + 0] = (uint8_t)col[0];
          row[3 * x + 1] = (uint8_t)col[1];
          row[3 * x + 2] = (uint8_t)col[2];
        }
      }
    }
    edge[0].x += edge[0].dx;
    edge[1].x += edge[1].dx;
  } while (++y <= (int)ymax);
}

/* static_cast<int>(std::round(float)); out-of-range values (undefined in C++) saturate here, as the device does */
static int round_to_int(float f) {
  const float r = roundf(f);
  if (!(r > -2147483648.0f)) return r != r ? 0 : INT32_MIN;
  if (r >= 2147483648.0f) return INT32_MAX;
  return (int)r;
}

typedef struct {
  int face;
  double depth;
  int px[3], py[3];
  int gray;
} face_item;

static int by_depth_desc(const void* a, const void* b) {
  const face_item* fa = (const face_item*)a;
  const face_item* fb = (const face_item*)b;
  if (fa->depth > fb->depth) return -1;
  if (fa->depth < fb->depth) return 1;
  return (fa->face > fb->face) - (fa->face < fb->face);   /* stable: ties in face order */
}

/*
 * Face list of RenderSMPLMesh.h:36-92 for one frame.  cloud: n_vertices x 3 doubles (x, y, z per vertex, the memory
 * order of the reference's 3xN column-major matrix).  Writes up to n_faces items in draw order (far to near) and
 * returns how many.  items_* may be NULL.
 */
int overlay_oracle_drawlist(const double* cloud, int n_vertices, const int32_t* faces, int n_faces, double fx, double fy,
                            double cx, double cy, int backface_cull, int32_t* item_face, double* item_depth,
                            int32_t* item_pts /* [n][6] x0 y0 x1 y1 x2 y2 */, int32_t* item_gray) {
  float* pu = (float*)malloc(sizeof(float) * (size_t)n_vertices);
  float* pv = (float*)malloc(sizeof(float) * (size_t)n_vertices);
  uint8_t* valid = (uint8_t*)calloc((size_t)n_vertices, 1);
  face_item* list = (face_item*)malloc(sizeof(face_item) * (size_t)(n_faces > 0 ? n_faces : 1));
  int n = 0;
  for (int i = 0; i < n_vertices; ++i) {
    const double X = cloud[3 * i], Y = cloud[3 * i + 1], Z = cloud[3 * i + 2];
    pu[i] = pv[i] = -9999.f;
    if (Z <= 1e-6) continue;   /* NaN compares false and goes on, as in the reference */
    pu[i] = (float)(fx * X / Z + cx);
    pv[i] = (float)(fy * Y / Z + cy);
    valid[i] = 1;
  }
  for (int f = 0; f < n_faces; ++f) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    if (!valid[i0] || !valid[i1] || !valid[i2]) continue;
    const double* v0 = cloud + 3 * i0;
    const double* v1 = cloud + 3 * i1;
    const double* v2 = cloud + 3 * i2;
    const double e1[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const double e2[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const double n3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    if (backface_cull && n3[2] >= 0.0) continue;
    const double c[3] = {(v0[0] + v1[0] + v2[0]) / 3.0, (v0[1] + v1[1] + v2[1]) / 3.0, (v0[2] + v1[2] + v2[2]) / 3.0};
    /* Eigen normalized(): v / sqrt(v.v) when v.v > 0, else v unchanged */
    double w[3] = {-c[0], -c[1], -c[2]};
    const double w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (w2 > 0.0) { const double s = sqrt(w2); w[0] /= s; w[1] /= s; w[2] /= s; }
    double m[3] = {n3[0], n3[1], n3[2]};
    const double m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    if (m2 > 0.0) { const double s = sqrt(m2); m[0] /= s; m[1] /= s; m[2] /= s; }
    double shade = m[0] * w[0] + m[1] * w[1] + m[2] * w[2];
    shade = shade < 0.0 ? 0.0 : (shade > 1.0 ? 1.0 : shade);   /* std::clamp; NaN passes through both tests */
    face_item* it = &list[n++];
    it->face = f;
    it->depth = (v0[2] + v1[2] + v2[2]) / 3.0;
    it->px[0] = round_to_int(pu[i0]); it->py[0] = round_to_int(pv[i0]);
    it->px[1] = round_to_int(pu[i1]); it->py[1] = round_to_int(pv[i1]);
    it->px[2] = round_to_int(pu[i2]); it->py[2] = round_to_int(pv[i2]);
    const double g = round(220 * shade);
    it->gray = g != g ? 0 : (int)g;
  }
  qsort(list, (size_t)n, sizeof(face_item), by_depth_desc);
  for (int k = 0; k < n; ++k) {
    if (item_face) item_face[k] = list[k].face;
    if (item_depth) item_depth[k] = list[k].depth;
    if (item_gray) item_gray[k] = list[k].gray;
    if (item_pts)
      for (int c = 0; c < 3; ++c) { item_pts[6 * k + 2 * c] = list[k].px[c]; item_pts[6 * k + 2 * c + 1] = list[k].py[c]; }
  }
  free(pu); free(pv); free(valid); free(list);
  return n;
}

/* cv::polylines(img, {p0, p1, p2, p0}, false, Scalar(40, 40, 40), 1, cv::LINE_AA) */
static void wire_triangle(image3* im, const pt2l v[3]) {
  const int col[3] = {40, 40, 40};
  for (int i = 0; i < 3; ++i) {
    pt2l a = v[i], b = v[(i + 1) % 3];
    a.x *= XY_ONE; a.y *= XY_ONE; b.x *= XY_ONE; b.y *= XY_ONE;
    line_aa(im, a, b, col);
  }
}

/* cv::fillConvexPoly(img, pts, 3, Scalar(g,g,g), cv::LINE_AA) for one triangle (unit-test entry) */
void overlay_oracle_fill_triangle(uint8_t* img, int width, int height, size_t step, const int32_t pts[6], int gray) {
  image3 im = {img, width, height, step};
  const pt2l v[3] = {{pts[0], pts[1]}, {pts[2], pts[3]}, {pts[4], pts[5]}};
  /* saturate_cast<uchar> of the Scalar */
  const int g = gray < 0 ? 0 : (gray > 255 ? 255 : gray);
  const int col[3] = {g, g, g};
  fill_convex_tri_aa(&im, v, col);
}

/* renderSMPLMesh(cloud, faces, img, fx, fy, cx, cy, fill, backface_cull, wireframe) on one 8UC3 image */
int overlay_oracle_render(const double* cloud, int n_vertices, const int32_t* faces, int n_faces, uint8_t* img, int width,
                          int height, size_t step, double fx, double fy, double cx, double cy, int fill,
                          int backface_cull, int wireframe) {
  int32_t* pts = (int32_t*)malloc(sizeof(int32_t) * 6 * (size_t)(n_faces > 0 ? n_faces : 1));
  int32_t* gray = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_faces > 0 ? n_faces : 1));
  const int n = overlay_oracle_drawlist(cloud, n_vertices, faces, n_faces, fx, fy, cx, cy, backface_cull, NULL, NULL,
                                        pts, gray);
  for (int k = 0; k < n; ++k) {
    if (fill) overlay_oracle_fill_triangle(img, width, height, step, pts + 6 * k, gray[k]);
    if (wireframe) {
      image3 im = {img, width, height, step};
      const pt2l v[3] = {{pts[6 * k], pts[6 * k + 1]}, {pts[6 * k + 2], pts[6 * k + 3]}, {pts[6 * k + 4], pts[6 * k + 5]}};
      wire_triangle(&im, v);
    }
  }
  free(pts); free(gray);
  return n;
}
