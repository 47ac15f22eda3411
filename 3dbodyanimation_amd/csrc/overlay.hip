// Mesh overlay on the device: the batched replacement of smpl::render::renderSMPLMesh
// (/root/reference/include/RenderSMPLMesh.h:16-110).  Declared in include/bodyfit.h (bodyfit_overlay_*).
//
// The reference draws the triangles of one frame one after the other (painter's order) with OpenCV's
// anti-aliased convex fill, so a pixel's final value is a fold over the triangles that touch it, in draw order.
// That fold is kept, and made parallel over pixels instead:
//
//   k_ov_faces        one thread per (frame, face): projection, cull, shade, depth key, integer corners
//                     (RenderSMPLMesh.h:36-88)
//   k_ov_sort_chunks  one workgroup per 8192 faces of a frame: bitonic sort of (depth key, face) in LDS
//   k_ov_rank         rank of every face in the frame's draw order (own position + binary searches in the
//                     other chunks); the face records are scattered into draw order (:91-92)
//   k_ov_bin_count / k_ov_scan{1,2,3} / k_ov_bin_fill
//                     16x16-pixel tiles: which ranks touch which tile (bounding box + the AA fringe)
//   k_ov_tiles        one workgroup per non-empty tile, one thread per pixel: the tile's ranks are put in order
//                     through an LDS bitmap, 32 triangles at a time are set up (clipLine / LineAA / FillConvexPoly
//                     state, closed form per row and column) by 128 threads, and every pixel folds them in order
//                     (:95-104)
//
// Scan conversion is integer work (16.16 fixed point); the face stage is a short f64 chain compiled without
// contraction so that it matches the C restatement bit for bit.  Everything is HBM/latency-bound byte work: no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bodyfit.h"
#include "bodyfit_device.h"
#include "solver_view.h"

#pragma clang fp contract(off)

namespace {

constexpr int kTile = 16;               // pixels per tile side; one thread per pixel
constexpr int kChunk = 8192;            // faces per LDS sort
constexpr int kSortThreads = 1024;
constexpr int kMaxFaces = 65536;        // LDS bitmap of the tile kernel
constexpr int kFringe = 2;              // pixels around a triangle's bounding box the AA lines can touch
constexpr int XY_SHIFT = 16;
constexpr int XY_ONE = 1 << XY_SHIFT;
constexpr unsigned long long kDeadKey = ~0ull;

// drawing.cpp tables [recalled, see oracle/overlay_oracle.c]
__constant__ unsigned char cFilter[64] = {
    168, 177, 185, 194, 202, 210, 218, 224, 231, 236, 241, 246, 249, 252, 254, 254,
    254, 254, 252, 249, 246, 241, 236, 231, 224, 218, 210, 202, 194, 185, 177, 168,
    158, 149, 140, 131, 122, 114, 105, 97,  89,  82,  75,  68,  62,  56,  50,  45,
    40,  36,  32,  28,  25,  22,  19,  16,  14,  12,  11,  9,   8,   7,   5,   5};
__constant__ unsigned char cSlopeCorr[32] = {181, 181, 181, 182, 182, 183, 184, 185, 187, 188, 190,
                                             192, 194, 196, 198, 201, 203, 206, 209, 211, 214, 218,
                                             221, 224, 227, 231, 235, 238, 242, 246, 250, 254};

struct OvFace {      // 32 bytes
  int px[3], py[3];
  int gray;          // -1: not drawn
  int face;
};

struct OvLine {      // one anti-aliased edge after clipping, closed form along its major axis (48 bytes)
  long long m0;      // minor coordinate at step 0, 16.16
  int step;          // minor increment per step, |step| <= 1 << 16
  int c0, E;         // first major coordinate, last step index (steps 0..E)
  int flags;         // bit 0 drawn, bit 1 x-major
  unsigned epk[3];   // end-point correction table, row min(step, 2): three 9-bit entries, column min(E - step, 2)
  unsigned pad[3];
};

struct OvFill {      // the two edge walkers of FillConvexPoly, at most two linear pieces each
  long long xs[2][2], dx[2][2];
  int ys[2][2];      // first row of the piece (second piece: INT_MAX when absent)
  int y0, y1;        // rows drawn: y0..y1 (empty when y1 < y0)
  int gray;
  int bx0, by0, bx1, by1;   // bounding box + fringe (quick reject)
  int pad;
};

struct OvTri {       // everything the per-pixel fold needs of one triangle: 256 bytes, made once by k_ov_setup
  OvLine l[3];
  OvFill f;
};
static_assert(sizeof(OvLine) == 48 && sizeof(OvFill) == 112 && sizeof(OvTri) == 256, "OvTri layout");
static_assert(offsetof(OvLine, step) == 8 && offsetof(OvLine, flags) == 20 && offsetof(OvLine, epk) == 24 &&
                  offsetof(OvTri, f) == 144 && offsetof(OvFill, dx) == 32 && offsetof(OvFill, ys) == 64 &&
                  offsetof(OvFill, y0) == 80 && offsetof(OvFill, gray) == 88 && offsetof(OvFill, bx0) == 92,
              "k_ov_tiles reads the record by dword index");

__device__ __forceinline__ int round_to_int(float f) {
  const float r = roundf(f);
  if (!(r > -2147483648.0f)) return r != r ? 0 : INT32_MIN;
  if (r >= 2147483648.0f) return INT32_MAX;
  return (int)r;
}

// ---- faces --------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_ov_faces(const T* __restrict__ cloud, size_t frame_stride,
                                                  const int* __restrict__ faces, int nF, int nV, double fx, double fy,
                                                  double cx, double cy, int cull, OvFace* __restrict__ out,
                                                  unsigned long long* __restrict__ keys) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  const int frame = blockIdx.y;
  if (f >= nF) return;
  const T* c = cloud + (size_t)frame * frame_stride;
  const int id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
  double v[3][3];
  bool ok = true;
  int px[3], py[3];
  for (int k = 0; k < 3; ++k) {
    v[k][0] = (double)c[3 * (size_t)id[k]];
    v[k][1] = (double)c[3 * (size_t)id[k] + 1];
    v[k][2] = (double)c[3 * (size_t)id[k] + 2];
    if (v[k][2] <= 1e-6) ok = false;                     // RenderSMPLMesh.h:42
    const float u = (float)(fx * v[k][0] / v[k][2] + cx);   // :43-44
    const float w = (float)(fy * v[k][1] / v[k][2] + cy);
    px[k] = round_to_int(u);                             // :79-84
    py[k] = round_to_int(w);
  }
  OvFace o;
  o.face = f;
  o.gray = -1;
  unsigned long long key = kDeadKey;
  for (int k = 0; k < 3; ++k) { o.px[k] = px[k]; o.py[k] = py[k]; }
  if (ok) {
    const double e1[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
    const double e2[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    if (!(cull && n[2] >= 0.0)) {                        // :65
      const double ctr[3] = {(v[0][0] + v[1][0] + v[2][0]) / 3.0, (v[0][1] + v[1][1] + v[2][1]) / 3.0,
                             (v[0][2] + v[1][2] + v[2][2]) / 3.0};
      double w[3] = {-ctr[0], -ctr[1], -ctr[2]};
      const double w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
      if (w2 > 0.0) { const double s = sqrt(w2); w[0] /= s; w[1] /= s; w[2] /= s; }
      double m[3] = {n[0], n[1], n[2]};
      const double m2 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
      if (m2 > 0.0) { const double s = sqrt(m2); m[0] /= s; m[1] /= s; m[2] /= s; }
      double shade = m[0] * w[0] + m[1] * w[1] + m[2] * w[2];   // :70
      shade = shade < 0.0 ? 0.0 : (shade > 1.0 ? 1.0 : shade);
      const double g = round(220 * shade);                 // :99
      int gi = g != g ? 0 : (int)g;
      o.gray = gi < 0 ? 0 : (gi > 255 ? 255 : gi);
      const double depth = (v[0][2] + v[1][2] + v[2][2]) / 3.0;   // :74, > 0 for a valid face
      // far to near: larger depth first.  depth > 0, so its bit pattern orders like the value.
      key = ~(unsigned long long)__double_as_longlong(depth);
      if (key == kDeadKey) key = kDeadKey - 1;
    }
  }
  out[(size_t)frame * nF + f] = o;
  keys[(size_t)frame * nF + f] = key;
}

// ---- draw order ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool key_less(unsigned long long ka, unsigned ia, unsigned long long kb, unsigned ib) {
  return ka < kb || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(kSortThreads) void k_ov_sort_chunks(const unsigned long long* __restrict__ keys, int nF,
                                                                 int nChunks, unsigned long long* __restrict__ skey,
                                                                 unsigned* __restrict__ sidx) {
  extern __shared__ unsigned long long sm[];
  unsigned long long* k = sm;
  unsigned* ix = reinterpret_cast<unsigned*>(sm + kChunk);
  const int chunk = blockIdx.x, frame = blockIdx.y;
  const int base = chunk * kChunk;
  for (int i = threadIdx.x; i < kChunk; i += kSortThreads) {
    const int f = base + i;
    k[i] = f < nF ? keys[(size_t)frame * nF + f] : kDeadKey;
    ix[i] = f < nF ? (unsigned)f : 0xFFFFFFFFu;
  }
  __syncthreads();
  for (int size = 2; size <= kChunk; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < kChunk / 2; t += kSortThreads) {
        const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1));
        const int hi = lo | stride;
        const bool up = (lo & size) == 0;
        const unsigned long long ka = k[lo], kb = k[hi];
        const unsigned ia = ix[lo], ib = ix[hi];
        const bool swap = up ? key_less(kb, ib, ka, ia) : key_less(ka, ia, kb, ib);
        if (swap) { k[lo] = kb; k[hi] = ka; ix[lo] = ib; ix[hi] = ia; }
      }
      __syncthreads();
    }
  }
  const size_t o = ((size_t)frame * nChunks + chunk) * kChunk;
  for (int i = threadIdx.x; i < kChunk; i += kSortThreads) { skey[o + i] = k[i]; sidx[o + i] = ix[i]; }
}

__global__ __launch_bounds__(256) void k_ov_rank(const unsigned long long* __restrict__ skey,
                                                 const unsigned* __restrict__ sidx, int nF, int nChunks,
                                                 const OvFace* __restrict__ in, OvFace* __restrict__ sorted,
                                                 int* __restrict__ n_alive) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int frame = blockIdx.y;
  if (e >= nChunks * kChunk) return;
  const size_t fo = (size_t)frame * nChunks * kChunk;
  const unsigned idx = sidx[fo + e];
  if (idx == 0xFFFFFFFFu) return;          // padding of the last chunk
  const unsigned long long key = skey[fo + e];
  const int c = e / kChunk;
  int rank = e - c * kChunk;
  for (int o = 0; o < nChunks; ++o) {
    if (o == c) continue;
    const unsigned long long* ok = skey + fo + (size_t)o * kChunk;
    const unsigned* oi = sidx + fo + (size_t)o * kChunk;
    int lo = 0, hi = kChunk;             // first position whose (key, idx) is not less than mine
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (key_less(ok[mid], oi[mid], key, idx)) lo = mid + 1; else hi = mid;
    }
    rank += lo;
  }
  sorted[(size_t)frame * nF + rank] = in[(size_t)frame * nF + idx];
  if (key != kDeadKey) atomicAdd(&n_alive[frame], 1);
}

// ---- binning ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tile_range(const OvFace& o, int W, int H, int tilesX, int tilesY, int& tx0, int& ty0,
                                           int& tx1, int& ty1) {
  long long x0 = min(o.px[0], min(o.px[1], o.px[2])), x1 = max(o.px[0], max(o.px[1], o.px[2]));
  long long y0 = min(o.py[0], min(o.py[1], o.py[2])), y1 = max(o.py[0], max(o.py[1], o.py[2]));
  x0 -= kFringe; y0 -= kFringe; x1 += kFringe; y1 += kFringe;
  if (x1 < 0 || y1 < 0 || x0 >= W || y0 >= H) return false;
  x0 = max(x0, 0ll); y0 = max(y0, 0ll); x1 = min(x1, (long long)W - 1); y1 = min(y1, (long long)H - 1);
  tx0 = (int)x0 / kTile; ty0 = (int)y0 / kTile; tx1 = (int)x1 / kTile; ty1 = (int)y1 / kTile;
  return true;
}

template <bool kFillPass>
__global__ __launch_bounds__(256) void k_ov_bin(const OvFace* __restrict__ sorted, int nF, int W, int H, int tilesX,
                                                int tilesY, unsigned* __restrict__ count,
                                                const unsigned* __restrict__ offset, unsigned* __restrict__ entries) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int frame = blockIdx.y;
  if (r >= nF) return;
  const OvFace o = sorted[(size_t)frame * nF + r];
  if (o.gray < 0) return;
  int tx0, ty0, tx1, ty1;
  if (!tile_range(o, W, H, tilesX, tilesY, tx0, ty0, tx1, ty1)) return;
  const size_t tb = (size_t)frame * tilesX * tilesY;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) {
      const size_t t = tb + (size_t)ty * tilesX + tx;
      if (kFillPass) {
        const unsigned slot = atomicSub(&count[t], 1u) - 1u;   // the count pass left the tile's total here
        entries[offset[t] + slot] = (unsigned)r;
      } else {
        atomicAdd(&count[t], 1u);
      }
    }
}

// exclusive scan of the tile counts in three small launches: block sums, their scan, offsets + compaction
constexpr int kScanBlock = 1024;   // tiles per block (256 threads x 4)

__global__ __launch_bounds__(256) void k_ov_scan1(const unsigned* __restrict__ count, size_t n,
                                                  unsigned* __restrict__ blockSum) {
  __shared__ unsigned red[4];
  const size_t i0 = (size_t)blockIdx.x * kScanBlock + threadIdx.x * 4;
  unsigned s = 0;
  for (int k = 0; k < 4; ++k) if (i0 + k < n) s += count[i0 + k];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) blockSum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void k_ov_scan2(unsigned* __restrict__ blockSum, int nBlocks,
                                                   unsigned* __restrict__ totals /* [0] entries */) {
  __shared__ unsigned part[1024];
  const int per = (nBlocks + 1023) / 1024;
  const int b0 = threadIdx.x * per;
  unsigned s = 0;
  for (int k = 0; k < per; ++k) if (b0 + k < nBlocks) s += blockSum[b0 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = part[threadIdx.x] - s;   // exclusive
  for (int k = 0; k < per; ++k)
    if (b0 + k < nBlocks) { const unsigned v = blockSum[b0 + k]; blockSum[b0 + k] = run; run += v; }
  if (threadIdx.x == 1023) totals[0] = part[1023];
}

__global__ __launch_bounds__(256) void k_ov_scan3(const unsigned* __restrict__ count, size_t n,
                                                  const unsigned* __restrict__ blockBase, unsigned* __restrict__ offset,
                                                  unsigned* __restrict__ active, unsigned* __restrict__ totals) {
  __shared__ unsigned part[256], nz[256];
  __shared__ unsigned slotBase;
  const size_t i0 = (size_t)blockIdx.x * kScanBlock + threadIdx.x * 4;
  unsigned c[4], s = 0, z = 0;
  for (int k = 0; k < 4; ++k) { c[k] = i0 + k < n ? count[i0 + k] : 0; s += c[k]; z += c[k] ? 1u : 0u; }
  part[threadIdx.x] = s;
  nz[threadIdx.x] = z;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned v = threadIdx.x >= o ? part[threadIdx.x - o] : 0, w = threadIdx.x >= o ? nz[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    nz[threadIdx.x] += w;
    __syncthreads();
  }
  // one atomic per block reserves the block's slots in the list of non-empty tiles (its order is immaterial)
  if (threadIdx.x == 255) slotBase = nz[255] ? atomicAdd(&totals[1], nz[255]) : 0u;
  __syncthreads();
  unsigned run = blockBase[blockIdx.x] + part[threadIdx.x] - s;
  unsigned slot = slotBase + nz[threadIdx.x] - z;
  for (int k = 0; k < 4; ++k)
    if (i0 + k < n) {
      offset[i0 + k] = run;
      run += c[k];
      if (c[k]) active[slot++] = (unsigned)(i0 + k);
    }
}

// ---- triangle set-up (one lane per edge / per fill) ---------------------------------------------------------
struct Pt { long long x, y; };

// cv::clipLine(Size2l, Point2l&, Point2l&)
__device__ bool clip_line(long long width, long long height, Pt& p1, Pt& p2) {
  const long long right = width - 1, bottom = height - 1;
  long long x1 = p1.x, y1 = p1.y, x2 = p2.x, y2 = p2.y;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) {
      a = c1 < 8 ? 0 : bottom;
      x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
      y1 = a;
      c1 = (x1 < 0) + (x1 > right) * 2;
    }
    if (c2 & 12) {
      a = c2 < 8 ? 0 : bottom;
      x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
      y2 = a;
      c2 = (x2 < 0) + (x2 > right) * 2;
    }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) {
        a = c1 == 1 ? 0 : right;
        y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
        x1 = a;
        c1 = 0;
      }
      if (c2) {
        a = c2 == 1 ? 0 : right;
        y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
        x2 = a;
        c2 = 0;
      }
    }
  }
  p1.x = x1; p1.y = y1; p2.x = x2; p2.y = y2;
  return (c1 | c2) == 0;
}

// drawing.cpp LineAA up to its pixel loop: the loop itself becomes a closed form in the step index
__device__ void setup_line(OvLine& L, int xa, int ya, int xb, int yb, int W, int H) {
  Pt pt1{(long long)xa * XY_ONE, (long long)ya * XY_ONE}, pt2{(long long)xb * XY_ONE, (long long)yb * XY_ONE};
  L.flags = 0;
  if (!clip_line((long long)W << XY_SHIFT, (long long)H << XY_SHIFT, pt1, pt2)) return;
  long long dx = pt2.x - pt1.x, dy = pt2.y - pt1.y;
  long long j = dx < 0 ? -1 : 0, i = dy < 0 ? -1 : 0;
  const long long ax = (dx ^ j) - j, ay = (dy ^ i) - i;
  long long step;
  int slope, ecount;
  if (ax > ay) {
    dy = (dy ^ j) - j;
    if (j) { const Pt t = pt1; pt1 = pt2; pt2 = t; }
    step = (dy * XY_ONE) / (ax | 1);
    pt2.x += XY_ONE;
    ecount = (int)((pt2.x >> XY_SHIFT) - (pt1.x >> XY_SHIFT));
    j = -(pt1.x & (XY_ONE - 1));
    pt1.y += ((step * j) >> XY_SHIFT) + (XY_ONE >> 1);
    slope = (int)((step >> (XY_SHIFT - 5)) & 0x3f);
    slope ^= (step < 0 ? 0x3f : 0);
    i = (pt1.x >> (XY_SHIFT - 7)) & 0x78;
    j = (pt2.x >> (XY_SHIFT - 7)) & 0x78;
    L.flags = 3;
    L.c0 = (int)(pt1.x >> XY_SHIFT);
    L.m0 = pt1.y;
  } else {
    dx = (dx ^ i) - i;
    if (i) { const Pt t = pt1; pt1 = pt2; pt2 = t; }
    step = (dx * XY_ONE) / (ay | 1);
    pt2.y += XY_ONE;
    ecount = (int)((pt2.y >> XY_SHIFT) - (pt1.y >> XY_SHIFT));
    j = -(pt1.y & (XY_ONE - 1));
    pt1.x += ((step * j) >> XY_SHIFT) + (XY_ONE >> 1);
    slope = (int)((step >> (XY_SHIFT - 5)) & 0x3f);
    slope ^= (step < 0 ? 0x3f : 0);
    i = (pt1.y >> (XY_SHIFT - 7)) & 0x78;
    j = (pt2.y >> (XY_SHIFT - 7)) & 0x78;
    L.flags = 1;
    L.c0 = (int)(pt1.y >> XY_SHIFT);
    L.m0 = pt1.x;
  }
  L.step = (int)step;   // |dy| <= ax (resp. |dx| <= ay): at most one minor pixel per step
  L.E = ecount;
  slope = (slope & 0x20) ? 0x100 : cSlopeCorr[slope];
  const int t0 = slope << 7;
  const int t1 = ((0x78 - (int)i) | 4) * slope;
  const int t2 = ((int)j | 4) * slope;
  int ep[9];
  ep[0] = 0;
  ep[8] = slope;
  ep[1] = ep[3] = ((((int)(j - i) & 0x78) | 4) * slope >> 8) & 0x1ff;
  ep[2] = (t1 >> 8) & 0x1ff;
  ep[4] = (((((int)(j - i) + 0x80) | 4) * slope) >> 8) & 0x1ff;
  ep[5] = ((t1 + t0) >> 8) & 0x1ff;
  ep[6] = (t2 >> 8) & 0x1ff;
  ep[7] = ((t2 + t0) >> 8) & 0x1ff;
  for (int r = 0; r < 3; ++r) L.epk[r] = (unsigned)ep[3 * r] | ((unsigned)ep[3 * r + 1] << 9) | ((unsigned)ep[3 * r + 2] << 18);
}

// drawing.cpp FillConvexPoly (LINE_AA, shift 0, three points): the row loop only changes state at vertex rows, so
// the same state machine is run from event row to event row and every walker piece is kept as (first row, x, dx)
__device__ __forceinline__ int sel3(int a, int b, int c, int i) { return i == 0 ? a : (i == 1 ? b : c); }

__device__ void setup_fill(OvFill& Fl, const OvFace& o, int W, int H) {
  const int vx0 = o.px[0], vx1 = o.px[1], vx2 = o.px[2], vy0 = o.py[0], vy1 = o.py[1], vy2 = o.py[2];
  int imin = 0;
  long long ymin = vy0;
  if (vy1 < ymin) { ymin = vy1; imin = 1; }
  if (vy2 < ymin) { ymin = vy2; imin = 2; }
  long long ymax = max(vy0, max(vy1, vy2));
  const long long xmin = min(vx0, min(vx1, vx2)), xmax = max(vx0, max(vx1, vx2));
  Fl.gray = o.gray;
  Fl.pad = 0;
  Fl.bx0 = (int)max(xmin - kFringe, (long long)INT32_MIN); Fl.bx1 = (int)min(xmax + kFringe, (long long)INT32_MAX);
  Fl.by0 = (int)max(ymin - kFringe, (long long)INT32_MIN); Fl.by1 = (int)min(ymax + kFringe, (long long)INT32_MAX);
  Fl.y0 = 0; Fl.y1 = -1;
  long long xs[2][2] = {{-XY_ONE, 0}, {-XY_ONE, 0}}, dxs[2][2] = {{0, 0}, {0, 0}};
  int ys[2][2] = {{(int)ymin, INT32_MAX}, {(int)ymin, INT32_MAX}};
  auto store = [&]() {
    for (int w = 0; w < 2; ++w)
      for (int q = 0; q < 2; ++q) { Fl.xs[w][q] = xs[w][q]; Fl.dx[w][q] = dxs[w][q]; Fl.ys[w][q] = ys[w][q]; }
  };
  if (xmax < 0 || ymax < 0 || xmin >= W || ymin >= H) { store(); return; }
  if (ymax > H - 1) ymax = H - 1;
  int edges = 3;
  int idx_a = imin, idx_b = imin, ye_a = (int)ymin, ye_b = (int)ymin, np_a = 0, np_b = 0;
  int y = (int)ymin;
  int y_end = (int)ymax;
  for (;;) {
    // an event row: (y < ymax || y == ymin) holds here by construction
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ye = i ? ye_b : ye_a;
      if (y >= ye) {
        int idx0 = i ? idx_b : idx_a;
        const int di = i ? 2 : 1;
        int idx = idx0 + di;
        if (idx >= 3) idx -= 3;
        for (; edges-- > 0;) {
          const int ty = sel3(vy0, vy1, vy2, idx);
          if (ty > y) {
            const long long x0 = (long long)sel3(vx0, vx1, vx2, idx0) * XY_ONE;
            const long long x1 = (long long)sel3(vx0, vx1, vx2, idx) * XY_ONE;
            const long long d = ((x1 - x0) * 2 + (ty - y)) / (2 * (long long)(ty - y));
            const int q = min(i ? np_b : np_a, 1);
            if (i) { ye_b = ty; idx_b = idx; np_b = q + 1; } else { ye_a = ty; idx_a = idx; np_a = q + 1; }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
              if (qq == q) { ys[i][qq] = y; xs[i][qq] = x0; dxs[i][qq] = d; }
            break;
          }
          idx0 = idx;
          idx += di;
          if (idx >= 3) idx -= 3;
        }
      }
    }
    if (edges < 0) { y_end = y - 1; break; }
    const int next = min(ye_a, ye_b);         // > y: every walker that was due has moved on (or edges < 0 above)
    if (next <= y || next >= (int)ymax) break;   // no further update row before the last one
    y = next;
  }
  store();
  Fl.y0 = max((int)ymin, 0);
  Fl.y1 = y_end;
}

// one thread per (frame, rank): the triangle's three AA edges and its fill walkers, once, for every tile it touches
__global__ __launch_bounds__(256) void k_ov_setup(const OvFace* __restrict__ sorted, int nF, int W, int H,
                                                  OvTri* __restrict__ tris) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  const int frame = blockIdx.y;
  if (r >= nF) return;
  const OvFace o = sorted[(size_t)frame * nF + r];
  if (o.gray < 0) return;
  OvTri t = {};
  // FillConvexPoly draws p0 = v[2] -> v[0], v[0] -> v[1], v[1] -> v[2]
  setup_line(t.l[0], o.px[2], o.py[2], o.px[0], o.py[0], W, H);
  setup_line(t.l[1], o.px[0], o.py[0], o.px[1], o.py[1], W, H);
  setup_line(t.l[2], o.px[1], o.py[1], o.px[2], o.py[2], W, H);
  setup_fill(t.f, o, W, H);
  tris[(size_t)frame * nF + r] = t;
}

// ---- tiles --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void blend2(int& v, int col, int a) {   // ICV_PUT_POINT, applied twice
  v += (__mul24(col - v, a) + 127) >> 8;   // |col - v|, a <= 255: the full-rate 24-bit multiply is exact
  v += (__mul24(col - v, a) + 127) >> 8;
}

// One wave per tile.  The tile's pixels live in LDS as packed 0x00RRGGBB words; the triangles of the tile are taken in
// draw order, and for each one the LANES SPAN THE TRIANGLE'S OWN WORK: an edge's (step, tap) pairs inside the tile
// (<= 16 steps x 3 taps), then the fill's (row, column) pairs four rows at a time.  Within one edge no two taps share
// a pixel and a wave's LDS operations execute in program order, so the fold over triangles, edges and fill is the
// sequential one.  A set-up record is exactly one dword per lane: it is fetched with one coalesced load several
// triangles ahead, its fields are read with v_readlane, and the per-triangle bookkeeping (bounding box, step range, row range)
// runs on the scalar unit.
constexpr int kTileThreads = 64;
constexpr int kTileList = 256;    // ordered ranks held in LDS per pass
constexpr int kAhead = 8;         // set-up records in flight per wave

__device__ __forceinline__ unsigned wave_excl_scan(unsigned v, unsigned& total) {
  unsigned incl = v;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned u = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += u;
  }
  total = __shfl(incl, 63);
  return incl - v;
}

__global__ __launch_bounds__(kTileThreads) void k_ov_tiles(const OvTri* __restrict__ tris, int nF, int W, int H,
                                                           int tilesX, int tilesY, const unsigned* __restrict__ offset,
                                                           const unsigned* __restrict__ entries,
                                                           const unsigned* __restrict__ active,
                                                           const unsigned* __restrict__ totals,
                                                           unsigned char* __restrict__ images, size_t row_stride,
                                                           size_t frame_stride, int mode /* 1 fill, 2 wireframe */) {
  extern __shared__ unsigned char smem[];
  const int nWords = (nF + 31) >> 5;
  unsigned* pix = reinterpret_cast<unsigned*>(smem);                    // [256] the tile, row-major
  unsigned* list = pix + 256;                                           // [kTileList] ranks in draw order
  unsigned* raw = list + kTileList;                                     // [256] short lists before ordering
  unsigned* span = raw + 256;                                           // [16] a triangle's span per tile row
  unsigned* bitmap = span + kTile;                                      // [nWords]  long lists only
  unsigned* wordPos = bitmap + nWords;                                  // [nWords]
  unsigned char* filt = reinterpret_cast<unsigned char*>(wordPos + nWords);   // [64]
  const int lane = threadIdx.x;
  filt[lane] = cFilter[lane];
  const unsigned nActive = totals[1];
  const int tilesPerFrame = tilesX * tilesY;
  const int per = (nWords + 63) / 64;
  // edge lanes: 16 steps x 3 taps on lanes 0..47; fill lanes: 4 rows x 16 columns
  const int eStep = lane / 3, eTap = lane - eStep * 3;
  const int fRow = lane >> 4, fCol = lane & 15;

  for (unsigned a = blockIdx.x; a < nActive; a += gridDim.x) {
    const unsigned t = active[a];
    const int frame = (int)(t / tilesPerFrame);
    const int tt = (int)(t - (unsigned)frame * tilesPerFrame);
    const int tx0 = (tt % tilesX) * kTile, ty0 = (tt / tilesX) * kTile;
    const unsigned e0 = offset[t], e1 = offset[t + 1];
    const unsigned nList = e1 - e0;
    const OvTri* ft = tris + (size_t)frame * nF;
    unsigned char* img = images + (size_t)frame * frame_stride;
    __syncthreads();   // the previous tile's LDS is no longer in use
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + fRow + 4 * i, x = tx0 + fCol;
      unsigned v = 0;
      if (x < W && y < H) {
        const unsigned char* q = img + (size_t)y * row_stride + (size_t)x * 3;
        v = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16);
      }
      pix[(fRow + 4 * i) * kTile + fCol] = v;
    }
    const bool shortList = nList <= 256;
    if (shortList) {
      // order by counting: an entry's position is the number of smaller ranks (they are distinct)
      unsigned mine[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned j = lane + 64 * i;
        mine[i] = j < nList ? entries[e0 + j] : 0xFFFFFFFFu;
        if (j < nList) raw[j] = mine[i];
      }
      __syncthreads();
      unsigned pos[4] = {0, 0, 0, 0};
      for (unsigned j = 0; j < nList; ++j) {
        const unsigned r = raw[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) pos[i] += r < mine[i] ? 1u : 0u;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < (int)nList) list[pos[i]] = mine[i];
    } else {
      for (int w = lane; w < nWords; w += 64) bitmap[w] = 0;
      __syncthreads();
      for (unsigned e = e0 + lane; e < e1; e += 64) {
        const unsigned r = entries[e];
        atomicOr(&bitmap[r >> 5], 1u << (r & 31));
      }
      __syncthreads();
      unsigned sum = 0;
      for (int k = 0; k < per; ++k) { const int w = lane * per + k; if (w < nWords) sum += __popc(bitmap[w]); }
      unsigned total;
      unsigned run = wave_excl_scan(sum, total);
      for (int k = 0; k < per; ++k) {
        const int w = lane * per + k;
        if (w < nWords) { wordPos[w] = run; run += __popc(bitmap[w]); }
      }
    }
    __syncthreads();
    for (unsigned pass0 = 0; pass0 < nList; pass0 += kTileList) {
      if (!shortList) {
        __syncthreads();
        for (int w = lane; w < nWords; w += 64) {   // ranks with ordinal in [pass0, pass0 + kTileList), ascending
          unsigned bits = bitmap[w], pos = wordPos[w];
          while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            if (pos >= pass0 && pos < pass0 + kTileList) list[pos - pass0] = (unsigned)(w * 32 + b);
            ++pos;
          }
        }
        __syncthreads();
      }
      const unsigned nPass = min((unsigned)kTileList, nList - pass0);
      // a set-up record is 64 dwords: one dword per lane, one load per triangle, two triangles ahead of the fold;
      // its fields are then picked out with v_readlane (dword offsets of OvTri: edges at 0 / 12 / 24, fill at 36)
      auto fetch = [&](unsigned i) -> unsigned {
        return i < nPass ? reinterpret_cast<const unsigned*>(ft + list[i])[lane] : 0u;
      };
      unsigned ring[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) ring[j] = fetch(j);
      for (unsigned it = 0; it < nPass; ++it) {
        const unsigned rec = ring[0];
#pragma unroll
        for (int j = 0; j + 1 < kAhead; ++j) ring[j] = ring[j + 1];
        ring[kAhead - 1] = fetch(it + kAhead);
#define RL(i) ((int)__builtin_amdgcn_readlane(rec, (i)))
#define RL64(i) ((long long)(((unsigned long long)(unsigned)RL((i) + 1) << 32) | (unsigned)RL(i)))
        if (RL(61) < tx0 || RL(59) > tx0 + kTile - 1 || RL(62) < ty0 || RL(60) > ty0 + kTile - 1) continue;
        const int g = RL(58);
        // one anti-aliased segment (record dwords o .. o + 8) in colour `col`
        auto draw_edge = [&](int o, int col) {
          const int flags = RL(o + 5);
          if (!(flags & 1)) return;
          const bool xm = (flags & 2) != 0;
          const int major0 = xm ? tx0 : ty0, minor0 = xm ? ty0 : tx0;
          const int c0 = RL(o + 3), E = RL(o + 4);
          const int kA = max(0, major0 - c0), kB = min(E, major0 + kTile - 1 - c0);
          if (kA > kB) return;
          const int k = kA + eStep;
          const int m = RL(o) + __mul24(k, RL(o + 2));   // k < 2^15, |step| <= 2^16; the true value fits 32 bits
          const int mi = (m >> XY_SHIFT) - 1 + eTap - minor0;
          const int ma = c0 + k - major0;
          const unsigned ep0 = (unsigned)RL(o + 6), ep1 = (unsigned)RL(o + 7), ep2 = (unsigned)RL(o + 8);
          if (lane < 48 && k <= kB && (unsigned)mi < (unsigned)kTile) {
            const int dist = (m >> (XY_SHIFT - 5)) & 31;
            const int f = filt[eTap == 0 ? dist + 32 : (eTap == 1 ? dist : 63 - dist)];
            const int row = min(k, 2), cl = min(E - k, 2);
            const unsigned er = row == 0 ? ep0 : (row == 1 ? ep1 : ep2);
            const int ep = (int)((er >> (9 * cl)) & 0x1ffu);
            const int al = (__mul24(ep, f) >> 8) & 0xff;
            const int idx = xm ? mi * kTile + ma : ma * kTile + mi;
            const unsigned v = pix[idx];
            int c0v = (int)(v & 0xff), c1v = (int)((v >> 8) & 0xff), c2v = (int)((v >> 16) & 0xff);
            blend2(c0v, col, al); blend2(c1v, col, al); blend2(c2v, col, al);
            pix[idx] = (unsigned)c0v | ((unsigned)c1v << 8) | ((unsigned)c2v << 16);
          }
        };
        if (mode & 1) {
        // ---- cv::fillConvexPoly: the three anti-aliased edges v2-v0, v0-v1, v1-v2, then the spans ----
        draw_edge(0, g);
        draw_edge(12, g);
        draw_edge(24, g);
        // ---- the opaque span of every row: the two walkers once per tile row (lanes 0..15), then 4 rows per pass ----
        const int ya = max(RL(56), ty0), yb = min(RL(57), ty0 + kTile - 1);
        if (ya <= yb) {
          if (lane < kTile) {
            const int y = ty0 + lane;
            const int ys0b = RL(53), ys1b = RL(55);
            const bool pb0 = y >= ys0b, pb1 = y >= ys1b;
            const long long xs0 = pb0 ? RL64(38) : RL64(36), dx0 = pb0 ? RL64(46) : RL64(44);
            const long long xs1 = pb1 ? RL64(42) : RL64(40), dx1 = pb1 ? RL64(50) : RL64(48);
            const int yy0 = pb0 ? ys0b : RL(52), yy1 = pb1 ? ys1b : RL(54);
            long long xa = xs0 + dx0 * (long long)(y - yy0);
            long long xb = xs1 + dx1 * (long long)(y - yy1);
            if (xa > xb) { const long long tmp = xa; xa = xb; xb = tmp; }
            long long xx1 = ((xa + (XY_ONE - 1)) >> XY_SHIFT) - tx0, xx2 = (xb >> XY_SHIFT) - tx0;   // tile-relative
            const bool on = y >= ya && y <= yb && xx2 >= 0 && xx1 <= kTile - 1 && xx1 <= xx2;
            const int lo = (int)max(xx1, 0ll), hi = (int)min(xx2, (long long)kTile - 1);
            span[lane] = on ? (unsigned)lo | ((unsigned)hi << 8) : 0x00ffu;   // empty: lo = 255 > hi = 0
          }
          const unsigned gg = (unsigned)g * 0x010101u;
          for (int r0 = (ya - ty0) & ~3; r0 <= yb - ty0; r0 += 4) {
            const unsigned sp = span[r0 + fRow];
            if (fCol >= (int)(sp & 0xff) && fCol <= (int)(sp >> 8)) pix[(r0 + fRow) * kTile + fCol] = gg;
          }
        }
        }
        if (mode & 2) {
          // ---- cv::polylines({v0, v1, v2, v0}, open, gray 40, LINE_AA): the same three segments, after the fill ----
          draw_edge(12, 40);
          draw_edge(24, 40);
          draw_edge(0, 40);
        }
#undef RL
#undef RL64
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = ty0 + fRow + 4 * i, x = tx0 + fCol;
      if (x < W && y < H) {
        unsigned char* q = img + (size_t)y * row_stride + (size_t)x * 3;
        const unsigned v = pix[(fRow + 4 * i) * kTile + fCol];
        q[0] = (unsigned char)v; q[1] = (unsigned char)(v >> 8); q[2] = (unsigned char)(v >> 16);
      }
    }
  }
}

size_t tile_lds_bytes(int nF) {
  const int nWords = (nF + 31) >> 5;
  return 256 * 4 + kTileList * 4 + 256 * 4 + kTile * 4 + (size_t)nWords * 8 + 64;
}

#define OV_TRY(expr)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess)                                                                                     \
      return bodyfit_internal_fail(BODYFIT_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
  } while (0)

}  // namespace

struct bodyfit_overlay {
  int device = 0, nV = 0, nF = 0, W = 0, H = 0, maxFrames = 0;
  int tilesX = 0, tilesY = 0, nChunks = 0;
  int* d_faces = nullptr;
  OvFace *d_tmp = nullptr, *d_sorted = nullptr;
  OvTri* d_tris = nullptr;
  unsigned long long *d_key = nullptr, *d_skey = nullptr;
  unsigned *d_sidx = nullptr, *d_count = nullptr, *d_offset = nullptr, *d_blockSum = nullptr, *d_active = nullptr,
           *d_totals = nullptr, *d_entries = nullptr;
  int* d_alive = nullptr;
  size_t entriesCap = 0;
  void* d_cloud = nullptr;          // staging of the host form
  size_t cloudCap = 0;
  unsigned char* d_images = nullptr;
  size_t imagesCap = 0;
  int lastFrames = 0;
  hipEvent_t ev[5] = {};
  bool timed = false;
  std::vector<void*> owned;
  template <typename T>
  hipError_t alloc(T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) { owned.push_back(q); *p = static_cast<T*>(q); }
    return e;
  }
  ~bodyfit_overlay() {
    for (void* q : owned) (void)hipFree(q);
    if (d_entries) (void)hipFree(d_entries);
    if (d_cloud) (void)hipFree(d_cloud);
    if (d_images) (void)hipFree(d_images);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
  }
};

extern "C" {

int bodyfit_overlay_create(const bodyfit_overlay_desc* desc, bodyfit_overlay** out) {
  if (!desc || !out || !desc->faces) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_create: null argument");
  if (desc->n_faces < 1 || desc->n_faces > kMaxFaces || desc->n_vertices < 1 || desc->width < 1 || desc->height < 1 ||
      desc->max_frames < 1 || desc->width > 16384 || desc->height > 16384)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_create: sizes out of range (n_faces <= 65536, images <= 16384 x 16384)");
  for (int i = 0; i < desc->n_faces * 3; ++i)
    if (desc->faces[i] < 0 || desc->faces[i] >= desc->n_vertices)
      return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_create: face refers to a vertex out of range");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || desc->device < 0 || desc->device >= ndev)
    return bodyfit_internal_fail(BODYFIT_ERR_HIP, "bodyfit_overlay_create: no such HIP device (there is no CPU path)");
  OV_TRY(hipSetDevice(desc->device));
  auto* ov = new bodyfit_overlay;
  ov->device = desc->device; ov->nV = desc->n_vertices; ov->nF = desc->n_faces; ov->W = desc->width; ov->H = desc->height;
  ov->maxFrames = desc->max_frames;
  ov->tilesX = (ov->W + kTile - 1) / kTile; ov->tilesY = (ov->H + kTile - 1) / kTile;
  ov->nChunks = (ov->nF + kChunk - 1) / kChunk;
  const size_t F = ov->maxFrames, nT = F * ov->tilesX * ov->tilesY;
  const size_t nBlocks = (nT + 1 + kScanBlock - 1) / kScanBlock;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  chk(ov->alloc(&ov->d_faces, (size_t)ov->nF * 3));
  chk(ov->alloc(&ov->d_tmp, F * ov->nF));
  chk(ov->alloc(&ov->d_sorted, F * ov->nF));
  chk(ov->alloc(&ov->d_tris, F * ov->nF));
  chk(ov->alloc(&ov->d_key, F * ov->nF));
  chk(ov->alloc(&ov->d_skey, F * ov->nChunks * kChunk));
  chk(ov->alloc(&ov->d_sidx, F * ov->nChunks * kChunk));
  chk(ov->alloc(&ov->d_count, nT + 1));
  chk(ov->alloc(&ov->d_offset, nT + 1));
  chk(ov->alloc(&ov->d_blockSum, nBlocks));
  chk(ov->alloc(&ov->d_active, nT));
  chk(ov->alloc(&ov->d_totals, 4));
  chk(ov->alloc(&ov->d_alive, F));
  if (e == hipSuccess) e = hipMemcpy(ov->d_faces, desc->faces, (size_t)ov->nF * 3 * sizeof(int), hipMemcpyHostToDevice);
  for (auto& evn : ov->ev) if (e == hipSuccess) e = hipEventCreate(&evn);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ov_sort_chunks), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kChunk * 12);
  if (e != hipSuccess) {
    delete ov;
    return bodyfit_internal_fail(BODYFIT_ERR_HIP, (std::string("bodyfit_overlay_create: ") + hipGetErrorString(e)).c_str());
  }
  *out = ov;
  return BODYFIT_OK;
}

void bodyfit_overlay_destroy(bodyfit_overlay* ov) {
  if (!ov) return;
  (void)hipSetDevice(ov->device);
  delete ov;
}

int bodyfit_overlay_render_device(bodyfit_overlay* ov, const void* d_cloud, int cloud_is_f64,
                                  size_t cloud_frame_stride_elems, int n_frames, uint8_t* d_images, size_t row_stride,
                                  size_t frame_stride, double fx, double fy, double cx, double cy, int fill,
                                  int backface_cull, int wireframe, void* stream) {
  if (!ov || !d_cloud || !d_images) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_render: null argument");
  if (n_frames < 1 || n_frames > ov->maxFrames)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_render: n_frames exceeds max_frames");
  if ((n_frames > 1 && cloud_frame_stride_elems < (size_t)ov->nV * 3) || row_stride < (size_t)ov->W * 3 ||
      (n_frames > 1 && frame_stride < row_stride * (size_t)ov->H))
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_render: strides smaller than the data");
  OV_TRY(hipSetDevice(ov->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int F = n_frames, nF = ov->nF;
  const size_t nT = (size_t)F * ov->tilesX * ov->tilesY;
  // the scan runs over nT + 1 counts (the last one is zero), so that offset[nT] is the total
  const int nBlocks = (int)((nT + 1 + kScanBlock - 1) / kScanBlock);
  ov->lastFrames = F;
  ov->timed = false;
  OV_TRY(hipEventRecord(ov->ev[0], st));
  {
    dim3 grid((nF + 255) / 256, F);
    if (cloud_is_f64)
      BODYFIT_LAUNCH(k_ov_faces<double>, grid, dim3(256), 0, st, static_cast<const double*>(d_cloud),
                         cloud_frame_stride_elems, ov->d_faces, nF, ov->nV, fx, fy, cx, cy, backface_cull, ov->d_tmp,
                         ov->d_key);
    else
      BODYFIT_LAUNCH(k_ov_faces<float>, grid, dim3(256), 0, st, static_cast<const float*>(d_cloud),
                         cloud_frame_stride_elems, ov->d_faces, nF, ov->nV, fx, fy, cx, cy, backface_cull, ov->d_tmp,
                         ov->d_key);
  }
  OV_TRY(hipEventRecord(ov->ev[1], st));
  OV_TRY(hipMemsetAsync(ov->d_alive, 0, sizeof(int) * F, st));
  BODYFIT_LAUNCH(k_ov_sort_chunks, dim3(ov->nChunks, F), dim3(kSortThreads), kChunk * 12, st, ov->d_key, nF,
                     ov->nChunks, ov->d_skey, ov->d_sidx);
  BODYFIT_LAUNCH(k_ov_rank, dim3((ov->nChunks * kChunk + 255) / 256, F), dim3(256), 0, st, ov->d_skey, ov->d_sidx, nF,
                     ov->nChunks, ov->d_tmp, ov->d_sorted, ov->d_alive);
  OV_TRY(hipEventRecord(ov->ev[2], st));
  if (!fill && !wireframe) {   // the reference draws nothing (RenderSMPLMesh.h:97,106)
    OV_TRY(hipEventRecord(ov->ev[3], st));
    OV_TRY(hipEventRecord(ov->ev[4], st));
    OV_TRY(hipGetLastError());
    ov->timed = true;
    return BODYFIT_OK;
  }
  OV_TRY(hipMemsetAsync(ov->d_count, 0, sizeof(unsigned) * (nT + 1), st));
  OV_TRY(hipMemsetAsync(ov->d_totals, 0, sizeof(unsigned) * 4, st));
  BODYFIT_LAUNCH(k_ov_bin<false>, dim3((nF + 255) / 256, F), dim3(256), 0, st, ov->d_sorted, nF, ov->W, ov->H,
                     ov->tilesX, ov->tilesY, ov->d_count, ov->d_offset, ov->d_entries);
  BODYFIT_LAUNCH(k_ov_scan1, dim3(nBlocks), dim3(256), 0, st, ov->d_count, nT + 1, ov->d_blockSum);
  BODYFIT_LAUNCH(k_ov_scan2, dim3(1), dim3(1024), 0, st, ov->d_blockSum, nBlocks, ov->d_totals);
  BODYFIT_LAUNCH(k_ov_scan3, dim3(nBlocks), dim3(256), 0, st, ov->d_count, nT + 1, ov->d_blockSum, ov->d_offset,
                     ov->d_active, ov->d_totals);
  unsigned totals[2] = {0, 0};
  OV_TRY(hipMemcpyAsync(totals, ov->d_totals, sizeof(totals), hipMemcpyDeviceToHost, st));
  OV_TRY(hipStreamSynchronize(st));
  if (totals[0] > ov->entriesCap) {
    if (ov->d_entries) OV_TRY(hipFree(ov->d_entries));
    ov->d_entries = nullptr;
    ov->entriesCap = 0;
    const size_t want = (size_t)totals[0] + totals[0] / 4 + 1024;
    void* q = nullptr;
    OV_TRY(hipMalloc(&q, want * sizeof(unsigned)));
    ov->d_entries = static_cast<unsigned*>(q);
    ov->entriesCap = want;
  }
  if (totals[0]) {
    BODYFIT_LAUNCH(k_ov_setup, dim3((nF + 255) / 256, F), dim3(256), 0, st, ov->d_sorted, nF, ov->W, ov->H, ov->d_tris);
    BODYFIT_LAUNCH(k_ov_bin<true>, dim3((nF + 255) / 256, F), dim3(256), 0, st, ov->d_sorted, nF, ov->W, ov->H,
                       ov->tilesX, ov->tilesY, ov->d_count, ov->d_offset, ov->d_entries);
  }
  OV_TRY(hipEventRecord(ov->ev[3], st));
  if (totals[1]) {
    const int grid = (int)std::min<unsigned>(totals[1], 256u * 64u);
    BODYFIT_LAUNCH(k_ov_tiles, dim3(grid), dim3(kTileThreads), tile_lds_bytes(nF), st, ov->d_tris, nF, ov->W, ov->H,
                       ov->tilesX, ov->tilesY, ov->d_offset, ov->d_entries, ov->d_active, ov->d_totals, d_images,
                       row_stride, frame_stride, (fill ? 1 : 0) | (wireframe ? 2 : 0));
  }
  OV_TRY(hipEventRecord(ov->ev[4], st));
  OV_TRY(hipGetLastError());
  ov->timed = true;
  return BODYFIT_OK;
}

int bodyfit_overlay_render(bodyfit_overlay* ov, const void* cloud, int cloud_is_f64, size_t cloud_frame_stride_elems,
                           int n_frames, uint8_t* images, size_t row_stride, size_t frame_stride, double fx, double fy,
                           double cx, double cy, int fill, int backface_cull, int wireframe) {
  if (!ov || !cloud || !images) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_render: null argument");
  if (n_frames < 1 || n_frames > ov->maxFrames)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_render: n_frames exceeds max_frames");
  if (row_stride < (size_t)ov->W * 3 || (n_frames > 1 && frame_stride < row_stride * (size_t)ov->H))
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_render: strides smaller than the data");
  OV_TRY(hipSetDevice(ov->device));
  const size_t esz = cloud_is_f64 ? 8 : 4;
  const size_t cbytes = ((size_t)(n_frames - 1) * cloud_frame_stride_elems + (size_t)ov->nV * 3) * esz;
  const size_t ibytes = (size_t)(n_frames - 1) * frame_stride + row_stride * (size_t)ov->H;
  if (cbytes > ov->cloudCap) {
    if (ov->d_cloud) OV_TRY(hipFree(ov->d_cloud));
    ov->d_cloud = nullptr; ov->cloudCap = 0;
    OV_TRY(hipMalloc(&ov->d_cloud, cbytes));
    ov->cloudCap = cbytes;
  }
  if (ibytes > ov->imagesCap) {
    if (ov->d_images) OV_TRY(hipFree(ov->d_images));
    ov->d_images = nullptr; ov->imagesCap = 0;
    void* q = nullptr;
    OV_TRY(hipMalloc(&q, ibytes));
    ov->d_images = static_cast<unsigned char*>(q);
    ov->imagesCap = ibytes;
  }
  OV_TRY(hipMemcpy(ov->d_cloud, cloud, cbytes, hipMemcpyHostToDevice));
  OV_TRY(hipMemcpy(ov->d_images, images, ibytes, hipMemcpyHostToDevice));
  const int rc = bodyfit_overlay_render_device(ov, ov->d_cloud, cloud_is_f64, cloud_frame_stride_elems, n_frames,
                                               ov->d_images, row_stride, frame_stride, fx, fy, cx, cy, fill,
                                               backface_cull, wireframe, nullptr);
  if (rc) return rc;
  OV_TRY(hipDeviceSynchronize());
  OV_TRY(hipMemcpy(images, ov->d_images, ibytes, hipMemcpyDeviceToHost));
  return BODYFIT_OK;
}

int bodyfit_overlay_drawlist(bodyfit_overlay* ov, int frame, int* n_items, int32_t* face, int32_t* corners,
                             int32_t* gray) {
  if (!ov || !n_items) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_drawlist: null argument");
  if (frame < 0 || frame >= ov->lastFrames)
    return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_drawlist: frame was not part of the latest render");
  OV_TRY(hipSetDevice(ov->device));
  OV_TRY(hipDeviceSynchronize());
  int n = 0;
  OV_TRY(hipMemcpy(&n, ov->d_alive + frame, sizeof(int), hipMemcpyDeviceToHost));
  std::vector<OvFace> h((size_t)std::max(n, 1));
  if (n) OV_TRY(hipMemcpy(h.data(), ov->d_sorted + (size_t)frame * ov->nF, sizeof(OvFace) * n, hipMemcpyDeviceToHost));
  for (int k = 0; k < n; ++k) {
    if (face) face[k] = h[k].face;
    if (gray) gray[k] = h[k].gray;
    if (corners)
      for (int c = 0; c < 3; ++c) { corners[6 * k + 2 * c] = h[k].px[c]; corners[6 * k + 2 * c + 1] = h[k].py[c]; }
  }
  *n_items = n;
  return BODYFIT_OK;
}

int bodyfit_overlay_last_timing(bodyfit_overlay* ov, float ms[4]) {
  if (!ov || !ms) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_last_timing: null argument");
  if (!ov->timed) return bodyfit_internal_fail(BODYFIT_ERR_INVALID, "bodyfit_overlay_last_timing: nothing rendered yet");
  OV_TRY(hipSetDevice(ov->device));
  OV_TRY(hipEventSynchronize(ov->ev[4]));
  for (int i = 0; i < 4; ++i) OV_TRY(hipEventElapsedTime(&ms[i], ov->ev[i], ov->ev[i + 1]));
  return BODYFIT_OK;
}

}  // extern "C"
