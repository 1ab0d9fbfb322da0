// CPU test of the flatten-time acceleration builder (csrc/rtc_accel.hpp): the device code computes record addresses
// from cluster numbers and trusts every box to contain its triangles.  Compiled and run by tests/test_accel_build.py.
//   g++ -O2 -std=c++17 -I ray-tracer-challenge-rust_amd/csrc tests/cpp/accel_invariants.cpp -o accel_invariants
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "rtc_accel.hpp"

using namespace rtc;

static int fails = 0;
#define CHECK(cond, ...)                     \
  do {                                       \
    if (!(cond)) {                           \
      if (fails < 20) {                      \
        std::printf("FAIL %s: ", #cond);     \
        std::printf(__VA_ARGS__);            \
        std::printf("\n");                   \
      }                                      \
      ++fails;                               \
    }                                        \
  } while (0)

static void check_run(const std::vector<double>& tri9, uint32_t n, const char* what) {
  AccelRun R = build_accel_run(tri9.data(), n);
  const size_t nrec = R.trif.size();
  CHECK(R.order.size() == nrec, "%s: order/trif sizes", what);
  CHECK(nrec % TRI_BATCH == 0 || n == 0, "%s: records are padded to the batch size", what);
  // order is a permutation of 0..n-1 followed by padding only at the very end
  std::vector<int> seen(n, 0);
  for (size_t k = 0; k < nrec; ++k) {
    if (k < n) {
      CHECK(R.order[k] != NO_TRIANGLE && R.order[k] < n, "%s: record %zu is padding inside the triangle range", what, k);
      if (R.order[k] < n) ++seen[R.order[k]];
    } else {
      CHECK(R.order[k] == NO_TRIANGLE, "%s: record %zu beyond the triangle count is not padding", what, k);
    }
  }
  for (uint32_t i = 0; i < n; ++i) CHECK(seen[i] == 1, "%s: triangle %u appears %d times", what, i, seen[i]);
  // dense packing: cluster c starts at c * CLUSTER_TRIS, all but the last are full
  for (size_t c = 0; c < R.clusters.size(); ++c) {
    const bool last = c + 1 == R.clusters.size();
    CHECK(R.clusters[c].first == c * (size_t)CLUSTER_TRIS, "%s: cluster %zu starts at %u", what, c, R.clusters[c].first);
    CHECK(last || R.clusters[c].count == (uint32_t)CLUSTER_TRIS, "%s: cluster %zu holds %u records", what, c, R.clusters[c].count);
    CHECK(R.clusters[c].first + R.clusters[c].count <= nrec, "%s: cluster %zu overruns the records", what, c);
  }
  CHECK(R.clusters.size() == (n + CLUSTER_TRIS - 1) / CLUSTER_TRIS, "%s: %zu clusters for %u triangles", what, R.clusters.size(), n);
  // tiles partition the clusters in order, 64 at a time
  size_t next = 0;
  for (const DBox& t : R.tiles) {
    CHECK(t.first == next, "%s: tile starts at cluster %u, expected %zu", what, t.first, next);
    CHECK(t.count >= 1 && t.count <= (uint32_t)TILE_CLUSTERS, "%s: tile with %u clusters", what, t.count);
    next += t.count;
  }
  CHECK(next == R.clusters.size(), "%s: tiles cover %zu of %zu clusters", what, next, R.clusters.size());
  // every triangle vertex (f64) lies inside its cluster's and tile's f32 box; filter records are the f32 roundings
  for (size_t ti = 0; ti < R.tiles.size(); ++ti) {
    const DBox& T = R.tiles[ti];
    for (uint32_t c = T.first; c < T.first + T.count; ++c) {
      const DBox& C = R.clusters[c];
      for (int a = 0; a < 3; ++a) CHECK(T.mn[a] <= C.mn[a] && T.mx[a] >= C.mx[a], "%s: tile %zu does not contain cluster %u", what, ti, c);
      for (uint32_t k = C.first; k < C.first + C.count; ++k) {
        if (R.order[k] == NO_TRIANGLE) continue;
        const double* t = &tri9[(size_t)R.order[k] * 9];
        for (int a = 0; a < 3; ++a) {
          const double v[3] = {t[a], t[a] + t[3 + a], t[a] + t[6 + a]};
          for (double x : v) CHECK((double)C.mn[a] <= x && x <= (double)C.mx[a], "%s: vertex outside cluster %u on axis %d", what, c, a);
          CHECK(R.trif[k].p1[a] == (float)t[a] && R.trif[k].e1[a] == (float)t[3 + a] && R.trif[k].e2[a] == (float)t[6 + a],
                "%s: filter record %u is not the f32 rounding of the triangle", what, k);
          CHECK(std::fabs(t[a]) <= (double)R.pmax, "%s: pmax", what);
        }
      }
    }
  }
  for (const DBox& C : R.clusters)
    for (int a = 0; a < 3; ++a) CHECK(std::fabs(C.mn[a]) <= R.ext && std::fabs(C.mx[a]) <= R.ext, "%s: ext does not bound the boxes", what);
  CHECK(R.root_axis >= 0 && R.root_axis <= 2, "%s: root axis", what);
}

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const uint32_t sizes[] = {0, 1, 3, 4, 7, 8, 9, 15, 16, 17, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 6320, 33000};
  for (uint32_t n : sizes) {
    for (int kind = 0; kind < 4; ++kind) {
      std::vector<double> t((size_t)n * 9);
      for (uint32_t i = 0; i < n; ++i) {
        double* p = &t[(size_t)i * 9];
        const double scale = kind == 1 ? 1e6 : kind == 2 ? 1e-6 : 1.0;
        for (int a = 0; a < 3; ++a) {
          p[a] = U(rng) * 10.0 * scale;
          p[3 + a] = U(rng) * scale * (kind == 3 ? 0.0 : 1.0);  // kind 3: degenerate (zero-area) triangles
          p[6 + a] = U(rng) * scale;
        }
        if (kind == 3 && (i & 1)) p[0] = p[1] = p[2] = 5.0;  // many coincident centroids: ties in the median split
      }
      char what[64];
      std::snprintf(what, sizeof what, "n=%u kind=%d", n, kind);
      check_run(t, n, what);
    }
  }
  if (fails) {
    std::printf("%d checks failed\n", fails);
    return 1;
  }
  std::printf("ok\n");
  return 0;
}
