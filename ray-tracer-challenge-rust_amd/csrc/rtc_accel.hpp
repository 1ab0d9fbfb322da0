// rtc_accel.hpp — flatten-time acceleration data for triangle runs ("pre-compute bounding boxes", the TODO of the
// reference's README.md:25), built by rtc_scene_create.
//
// It never changes a result.  The exact test of a triangle is still the reference's f64 Möller–Trumbore
// (shape.rs:438-459, rtc_device.hpp::intersect_triangle); the structures here only decide, conservatively,
// which (ray, triangle) pairs can be skipped because the exact test would certainly report no intersection:
//
//   tile (<= 64 clusters)              -- f32 box   \  tested 64 at a time, one box per LANE, against the bundle of
//   cluster (<= CLUSTER_TRIS triangles) -- f32 box   /  the wave's rays (bundle_slabs); clusters also per ray (box_filter)
//   triangle filter record (DTriF)     -- f32 copy of p1,e1,e2 + error coefficients.  One triangle per LANE: its box
//                                         is rebuilt from the record and tested against the bundle (bundle_tri); then
//                                         per ray the f32 evaluation of det,u,v with a rigorous error bound rejects a
//                                         triangle only when the f64 evaluation is certain to reject it too
//                                         (tri_filter).
//
// Triangles of a run are reordered spatially (median splits) so that clusters are compact; the original
// depth-first leaf index travels with every triangle because the reference's tie rules are stated in that order.
// Every cluster of a run but the last holds exactly CLUSTER_TRIS records (the device computes record addresses from
// cluster numbers); the last one is padded to a multiple of TRI_BATCH with all-zero records that are never read as
// triangles (index >= the run's triangle count).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace rtc {

#ifndef RTC_SAH
#define RTC_SAH 1  // cluster build: 1 = surface-area heuristic over all cluster-aligned split positions, 0 = median splits
#endif
#ifndef RTC_CLUSTER_TRIS
#define RTC_CLUSTER_TRIS 8  // measured on MI355X (profiles/r01/v1_variants_cluster_tile.txt): 8 beats 16 by ~15% on C3/C4, loses ~8% on C5
#endif
constexpr int TRI_BATCH = 4;                     // records fetched per (scalar) load batch
constexpr int TRI9X_STRIDE = 10;                // doubles per exact record of the accelerated sweep: p1, e1, e2, depth-first leaf index (as bits)
constexpr int CLUSTER_TRIS = RTC_CLUSTER_TRIS;   // triangles per cluster (upper bound, multiple of TRI_BATCH)
constexpr int TILE_CLUSTERS = 64;                // clusters per tile = lanes per wave (one box per lane)
static_assert(CLUSTER_TRIS % TRI_BATCH == 0, "clusters are read in batches");
constexpr float F32_EPS = 5.9604645e-08f;  // 2^-24
constexpr double FILTER_K = 128.0;         // error-bound constant, >= 2x the derived bound
constexpr uint32_t NO_TRIANGLE = 0xffffffffu;

struct DTriF {  // 48 B: one triangle's filter record
  float p1[3], e1[3], e2[3];
  float c1;  // |det error| <= D * c1                 (D = max |d_i|)
  float c2;  // 2 * |x error| <= (P * D) * c2         (P = max|o_i| + max|p1_i| over the run)
  float c3;  // 2 * |y error| <= (P * D) * c3
};
struct DBox {  // 32 B: cluster or tile
  float mn[3];
  uint32_t first;
  float mx[3];
  uint32_t count;
};

// ---- host-side builder ------------------------------------------------------------------------------------
inline float f32_down(double x) {
  float f = (float)x;
  if ((double)f > x) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
  return f;
}
inline float f32_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafter(f, std::numeric_limits<float>::infinity());
  return f;
}

struct AccelRun {  // result for one triangle run
  std::vector<uint32_t> order;  // order[k] = index (within the run) of the k-th record, NO_TRIANGLE for padding
  std::vector<DTriF> trif;      // spatial order, padded
  std::vector<DBox> clusters;   // first/count index trif (relative to the run's first record); count is padded
  std::vector<DBox> tiles;      // first/count index clusters (relative to the run's first cluster)
  float pmax = 0.f;             // max |p1_i| over the run (rounded up)
  float ext = 0.f;              // max |coordinate| over all boxes (rounded up)
  int root_axis = 0;            // axis of the first split: records run from low to high coordinates on it, roughly
};

// tri9: n x 9 doubles (p1, e1, e2), object space
inline AccelRun build_accel_run(const double* tri9, uint32_t n) {
  AccelRun R;
  struct Item {
    double c[3], mn[3], mx[3];
    uint32_t idx;
  };
  std::vector<Item> items(n);
  for (uint32_t i = 0; i < n; ++i) {
    const double* t = tri9 + (size_t)i * 9;
    Item& it = items[i];
    it.idx = i;
    for (int a = 0; a < 3; ++a) {
      const double v0 = t[a], v1 = t[a] + t[3 + a], v2 = t[a] + t[6 + a];
      it.mn[a] = std::fmin(v0, std::fmin(v1, v2));
      it.mx[a] = std::fmax(v0, std::fmax(v1, v2));
      it.c[a] = (v0 + v1 + v2) / 3.0;
      if (!std::isfinite(it.c[a])) it.c[a] = 0.0;
    }
  }
  // recursive balanced splits (axis chosen by box area, below); left parts hold whole clusters
  struct Range {
    uint32_t b, e;
  };
  std::vector<Range> stack{{0, n}}, leaves;
  while (!stack.empty()) {
    Range r = stack.back();
    stack.pop_back();
    const uint32_t cnt = r.e - r.b;
    if (cnt <= (uint32_t)CLUSTER_TRIS) {
      if (cnt) leaves.push_back(r);
      continue;
    }
    const uint32_t ncl = (cnt + CLUSTER_TRIS - 1) / CLUSTER_TRIS;
    uint32_t mid = r.b + (ncl / 2) * CLUSTER_TRIS;
    auto by_axis = [](int axis) {
      return [axis](const Item& x, const Item& y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.idx < y.idx); };
    };
    auto box_area = [](const double mn[3], const double mx[3]) {
      const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
      const double area = dx * dy + dy * dz + dz * dx;
      return std::isfinite(area) ? area : INFINITY;
    };
    int axis = 0;
#if RTC_SAH
    // Surface-area heuristic over every cluster-aligned split position of the three centroid orders: the cost of a
    // split is area(left) * count(left) + area(right) * count(right), what a ray pays for visiting the two halves.
    // (Round 1 only compared the three MEDIAN splits: balanced, but its clusters straddle gaps in the mesh.)
    double best = INFINITY;
    uint32_t best_pos = mid;
    std::vector<double> right_area(ncl + 1);
    for (int a = 0; a < 3; ++a) {
      std::sort(items.begin() + r.b, items.begin() + r.e, by_axis(a));
      double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t i = r.e; i-- > r.b;) {  // suffix boxes at cluster-aligned positions
        for (int k = 0; k < 3; ++k) mn[k] = std::fmin(mn[k], items[i].mn[k]), mx[k] = std::fmax(mx[k], items[i].mx[k]);
        if ((i - r.b) % CLUSTER_TRIS == 0) right_area[(i - r.b) / CLUSTER_TRIS] = box_area(mn, mx);
      }
      double lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t i = r.b; i < r.e; ++i) {
        for (int k = 0; k < 3; ++k) lmn[k] = std::fmin(lmn[k], items[i].mn[k]), lmx[k] = std::fmax(lmx[k], items[i].mx[k]);
        const uint32_t done = i + 1 - r.b;
        if (done % CLUSTER_TRIS == 0 && done < cnt) {
          const double cost = box_area(lmn, lmx) * (double)done + right_area[done / CLUSTER_TRIS] * (double)(cnt - done);
          if (cost < best) best = cost, axis = a, best_pos = r.b + done;
        }
      }
    }
    mid = best_pos;
    std::sort(items.begin() + r.b, items.begin() + r.e, by_axis(axis));
#else
    // Split at the median (cluster-aligned) of the axis whose two halves have the smallest total box area: the
    // halves' areas are what a ray pays for (surface-area heuristic restricted to the three balanced splits).
    // (First version: axis of the widest centroid extent; 1-4 % slower, profiles/r01/v15_variants_split_axis.txt.)
    auto split = [&](int ax) { std::nth_element(items.begin() + r.b, items.begin() + mid, items.begin() + r.e, by_axis(ax)); };
    auto half_area = [&](uint32_t b0, uint32_t e0) {
      double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t i = b0; i < e0; ++i)
        for (int a = 0; a < 3; ++a) mn[a] = std::fmin(mn[a], items[i].mn[a]), mx[a] = std::fmax(mx[a], items[i].mx[a]);
      return box_area(mn, mx);
    };
    double best = INFINITY;
    for (int a = 0; a < 3; ++a) {
      split(a);
      const double cost = half_area(r.b, mid) * (double)(mid - r.b) + half_area(mid, r.e) * (double)(r.e - mid);
      if (cost < best) best = cost, axis = a;
    }
    split(axis);
#endif
    if (r.b == 0 && r.e == n) R.root_axis = axis;
    stack.push_back({mid, r.e});  // right first so that the left part is emitted first (depth-first order)
    stack.push_back({r.b, mid});
  }
  // `leaves` is in depth-first (spatially coherent) order because the left range is always popped first
  double pmax = 0.0, ext = 0.0;
  for (const Range& lf : leaves) {
    DBox cb;
    cb.first = (uint32_t)R.trif.size();
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = lf.b; i < lf.e; ++i) {
      const Item& it = items[i];
      const double* t = tri9 + (size_t)it.idx * 9;
      DTriF f;
      double E1 = 0, E2 = 0;
      for (int a = 0; a < 3; ++a) {
        f.p1[a] = (float)t[a], f.e1[a] = (float)t[3 + a], f.e2[a] = (float)t[6 + a];
        E1 = std::fmax(E1, std::fabs(t[3 + a]));
        E2 = std::fmax(E2, std::fabs(t[6 + a]));
        pmax = std::fmax(pmax, std::fabs(t[a]));
        mn[a] = std::fmin(mn[a], it.mn[a]);
        mx[a] = std::fmax(mx[a], it.mx[a]);
      }
      const double ke = FILTER_K * (double)F32_EPS * 1.001;
      f.c1 = f32_up(ke * E1 * E2);
      f.c2 = f32_up(2.0 * ke * E2);
      f.c3 = f32_up(2.0 * ke * E1);
      R.trif.push_back(f);
      R.order.push_back(it.idx);
    }
    while (R.trif.size() % TRI_BATCH) {  // padding: det = 0 => "|det| < EPSILON", always rejected
      R.trif.push_back(DTriF{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, 0, 0, 0});
      R.order.push_back(NO_TRIANGLE);
    }
    cb.count = (uint32_t)R.trif.size() - cb.first;
    for (int a = 0; a < 3; ++a) {
      cb.mn[a] = f32_down(mn[a]), cb.mx[a] = f32_up(mx[a]);
      if (std::isfinite(mn[a])) ext = std::fmax(ext, std::fabs(mn[a]));
      if (std::isfinite(mx[a])) ext = std::fmax(ext, std::fabs(mx[a]));
    }
    R.clusters.push_back(cb);
  }
  for (uint32_t c0 = 0; c0 < R.clusters.size(); c0 += TILE_CLUSTERS) {
    DBox tb;
    tb.first = c0;
    tb.count = std::min<uint32_t>(TILE_CLUSTERS, (uint32_t)R.clusters.size() - c0);
    for (int a = 0; a < 3; ++a) tb.mn[a] = INFINITY, tb.mx[a] = -INFINITY;
    for (uint32_t c = c0; c < c0 + tb.count; ++c)
      for (int a = 0; a < 3; ++a) tb.mn[a] = std::fmin(tb.mn[a], R.clusters[c].mn[a]), tb.mx[a] = std::fmax(tb.mx[a], R.clusters[c].mx[a]);
    R.tiles.push_back(tb);
  }
  R.pmax = f32_up(pmax);
  R.ext = f32_up(ext);
  return R;
}

}  // namespace rtc
