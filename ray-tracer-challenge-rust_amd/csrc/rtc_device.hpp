// rtc_device.hpp — device-side data layout and the per-ray arithmetic of the gfx950 renderer.
//
// Everything geometric is IEEE f64 in the reference's own operation order (this file is compiled with
// -ffp-contract=off): identical predicates => identical hit / shadow / pattern decisions as the CPU path,
// so colour parity does not depend on luck near discontinuities.  Only pow() differs from the host libm
// (smooth, ~1e-16 relative).  Citations are reference file:line under /root/reference/src.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtc_accel.hpp"

#pragma clang fp contract(off)

namespace rtc {

constexpr double EPSILON = 0.00001;  // utils.rs:2
#ifndef RTC_TILE_TRIS
#define RTC_TILE_TRIS 128
#endif
constexpr int TILE_TRIS = RTC_TILE_TRIS;  // triangles staged in LDS per tile of the brute-force sweep (9 f64 each = 9 KiB; with the
                                          // four waves. trace state 33 / 49 KiB per workgroup: at least three workgroups per CU)
// Workgroup shapes.  The brute-force sweep shares LDS triangle tiles between four waves (256 threads, 16x16
// pixels, each wave an 8x8 quadrant).  The accelerated sweep never synchronises waves, so its workgroup is ONE
// wave (64 threads, 8x8 pixels): a finished wave frees its slot at once instead of waiting for three siblings
// (measured on MI355X, profiles/r01/v6_*: +4..16 % over 256-thread workgroups).
constexpr int BLOCK_EXACT = 256, BLOCK_ACCEL = 64;
constexpr int block_threads(bool accel) { return accel ? BLOCK_ACCEL : BLOCK_EXACT; }
constexpr int tile_side(bool accel) { return accel ? 8 : 16; }
constexpr int MAX_TOP = 4;           // top-level group boxes carried in the kernel arguments (DScene::top_*)

enum : uint32_t { K_SPHERE = 0, K_PLANE = 1, K_CUBE = 2, K_CYLINDER = 3, K_CONE = 4, K_TRIANGLE = 5 };

struct DMaterial {
  double color[3];
  double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
  int32_t pattern;
  int32_t _pad;
};
struct DPattern {
  int32_t kind;
  int32_t _pad;
  double a[3], b[3];
  double inv[12];  // rows 0..2 of Pattern::transform_inverse
};
struct DGroup {
  double mn[3], mx[3];
  int32_t parent;
  int32_t _pad;
  float fmn[3], fmx[3];  // the same box rounded outward to f32 (-inf/+inf where the f64 bound is not finite)
  float fext;            // max |finite coordinate| of the box
  float _pad2;
};

#ifndef RTC_CONST_LOADS
#define RTC_CONST_LOADS 1
#endif
#ifndef RTC_GROUP_PREFILTER
#define RTC_GROUP_PREFILTER 1  // f32 pass / fail / undecided classification in front of the exact f64 group-box test
#endif
// A segment is a maximal run of consecutive depth-first leaves that the sweep handles as one unit:
// either one non-triangle leaf, or a run of triangles sharing one transform and one group.
struct DSegment {
  uint32_t is_tri;
  uint32_t first_leaf;
  uint32_t count;
  uint32_t xform;
  int32_t group;
  uint32_t first_tri;  // index into tri9 for triangle segments
  // acceleration data of a triangle run (rtc_accel.hpp); indices are absolute into DScene::tiles / trif
  uint32_t first_tile, n_tiles;
  // every cluster of a run except the last holds exactly CLUSTER_TRIS records, so record r of cluster c (both
  // relative to the run) is DScene::trif[first_rec + c * CLUSTER_TRIS + r], and it is a real triangle (not padding)
  // exactly when c * CLUSTER_TRIS + r < count
  uint32_t first_cluster, first_rec;
  float pmax;  // max |p1_i| over the run
  float ext;   // max |box coordinate| over the run
  // groups above this segment, root first: DScene::seg_chain[chain_off .. chain_off + chain_len); the first
  // chain_common entries are the same groups as the previous segment's (their pass bits are still valid)
  uint32_t chain_off, chain_len, chain_common;
  uint32_t kind;  // bits 0-7: leaf kind of a non-triangle segment (K_TRIANGLE for a run); bits 8-9: axis of the run's
                  // first spatial split (records ascend along it).  Makes the record 64 bytes.
};

// What a sweep needs to start on a segment, in ONE record, so that its loads do not chase pointers (segment ->
// chain index -> group box, segment -> transform: dependent L2-latency round trips at the start of every sweep of every
// wave): the segment, its transform_inverse rows and the boxes of the first SEG_INLINE_GROUPS groups of its chain.
constexpr int SEG_INLINE_GROUPS = 2;  // an OBJ mesh sits two groups deep (obj_file.rs:120-128)
struct DSegRec {
  DSegment s;
  double xf[12];                   // = DScene::xf[s.xform]
  DGroup g[SEG_INLINE_GROUPS];     // = DScene::groups[seg_chain[s.chain_off + d]] for d < min(s.chain_len, SEG_INLINE_GROUPS)
};

struct DScene {
  double light_pos[3], light_int[3];
  const double* xf;  // [n_xf][12] rows 0..2 of Shape::transform_inverse (bottom row is exactly 0,0,0,1)
  const DMaterial* mats;
  const DPattern* pats;
  const DGroup* groups;
  const uint32_t* seg_chain;  // group chains of the segments (see DSegment)
  const DSegRec* segs;
  // per leaf, depth-first order
  const uint8_t* leaf_kind;
  const uint32_t* leaf_xf;
  const uint32_t* leaf_mat;
  const uint8_t* leaf_cflags; // null unless some leaves are value-equal duplicates: bit0 last copy, bit1 odd class size
  const uint32_t* leaf_class; // with leaf_cflags: first leaf of the value-equality class
  const double* leaf_params;  // [n_leaves][3]
  // packed triangles, depth-first order
  const double* tri9;    // [n_tri][9] p1, e1, e2 (object space)
  const double* leaf_nw; // [n_leaves][3] TRIANGLE leaves: world normal = Shape::normal_at for any point of the triangle
                         // (indexed by leaf so that shading needs no leaf -> triangle lookup first)
  // spatially ordered copies for the accelerated sweep (rtc_accel.hpp)
  const DTriF* trif;          // [n_tri] f32 filter records
  const DBox* clusters;       // DBox::first indexes trif
  const DBox* tiles;          // DBox::first indexes clusters
  const DBox* blocks;         // [n_tiles][8]: boxes of the aligned runs of 8 clusters inside each tile (first/count unused)
  const double* tri9x;        // [n_rec][TRI9X_STRIDE] exact records in the same spatial order as trif: p1, e1, e2, leaf index
  uint32_t n_groups, n_segs, n_leaves, n_tri;
  uint32_t has_refraction;  // any material with transparency != 0
  // Background early-out (render kernels): if every leaf sits under one of at most MAX_TOP top-level groups, their
  // f32 boxes travel with the kernel arguments and a wave whose 64 primary rays certainly miss all of them writes
  // BLACK without generating or tracing rays.  n_top = 0: not applicable.
  uint32_t n_top;
  float top_mn[4][3], top_mx[4][3];
  // parked rays beyond the 64 a wave keeps in LDS (trace(), scenes with a transparent material): pool_cap entries per
  // wave, [wave][field 0..7][entry]; pool_overflow is set if a wave ever needed more (the host reports it)
  double* pool;
  uint32_t pool_cap;
  unsigned int* pool_overflow;
  unsigned long long* counters;  // [6] radiance, shadow, triangle tests, primitive tests, filter tests, box tests; may be null
};

struct DCamera {
  uint32_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double inv[16];
  // pixel-invariant terms of ray_for_pixel, evaluated on the host in the same order as the device would (IEEE f64):
  // origin = inv * point(0,0,0); zt[i] = inv[4i+2] * -1.0 and wt[i] = inv[4i+3] * 1.0 (both exact)
  double origin[3], zt[3], wt[3];
};

// Scene tables never change while a kernel runs.  Reading them through the constant address space says so: a load at
// a wave-uniform address can then be a scalar load (s_load, SGPR result) even in the kernels whose loop also stores to
// global memory (the ray pool's overflow area), where the compiler otherwise has to assume the table was clobbered and
// falls back to a per-lane vector load plus v_readfirstlane.
#if RTC_CONST_LOADS
template <class T>
__device__ __forceinline__ T ldc(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword-sized records only");
  union U {
    T v;
    uint32_t w[sizeof(T) / 4];
    __device__ U() {}
  } u;
  const __attribute__((address_space(4))) uint32_t* q = (const __attribute__((address_space(4))) uint32_t*)p;
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; ++i) u.w[i] = q[i];
  return u.v;
}
#else
template <class T>
__device__ __forceinline__ T ldc(const T* p) { return *p; }
#endif

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
// tuple.rs:68-73 (the w*w term is +0.0 for vectors and is dropped: it can only change the sign of a zero)
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// tuple.rs:75-83
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// tuple.rs:43-66
__device__ __forceinline__ double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) {
  double m = magnitude(a);
  if (m == 0.0) return {0., 0., 0.};
  return {a.x / m, a.y / m, a.z / m};
}
// tuple.rs:86-90: v - (n * 2.) * v.dot(n)
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (n * 2.) * dot(v, n); }

// matrix.rs:207-227 for a point (w = 1) and a vector (w = 0); m = rows 0..2 (12 doubles)
__device__ __forceinline__ V3 xf_point(const double* m, V3 p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * 1.0, m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * 1.0,
          m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * 1.0};
}
__device__ __forceinline__ V3 xf_vector(const double* m, V3 v) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * 0.0, m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7] * 0.0,
          m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11] * 0.0};
}
// transform_inverse_transpose * n with w := 0 (shape.rs:623-626): columns of the inverse
__device__ __forceinline__ V3 xf_normal(const double* m, V3 n) {
  return {m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
          m[2] * n.x + m[6] * n.y + m[10] * n.z};
}

struct Ray {
  V3 o, d;
};

// ---- per-ray sweep state -------------------------------------------------------------------------
constexpr int NO_LEAF = 0x7fffffff;  // "no hit yet": larger than any leaf index
struct RadianceState {
  double best_t;
  int best_leaf;
  int best_parity;  // parity of the hit leaf's negative-t intersections (is it in `containers`?)
  // the two odd-parity leaves with the greatest negative t (ties: later depth-first leaf), i.e. the last
  // two entries of prepare_computations' `containers` list just before the hit (intersection.rs:27-62)
  double neg_t1, neg_t2;
  int neg_leaf1, neg_leaf2;
  // value-equal duplicate leaves (DScene::leaf_cflags, null when the scene has none)
  const uint8_t* cflags;
};

template <bool REFR>
struct LeafAcc {  // collects one leaf's intersections in the radiance sweep
  RadianceState& s;
  int leaf;
  int nneg = 0;
  double negmax = -INFINITY;
  __device__ __forceinline__ LeafAcc(RadianceState& st, int l) : s(st), leaf(l) {}
  __device__ __forceinline__ void emit(double t) {
    if (t >= 0.) {  // Intersection::hit filter (intersection.rs:81); -0.0 passes, NaN does not
      // The first minimum in depth-first order wins (sort_by is stable, min_by keeps the first).  Leaves may be
      // visited in any order here, so the tie rule is spelled out: smaller t, then smaller leaf index.
      if (t < s.best_t || (t == s.best_t && leaf < s.best_leaf)) {
        s.best_t = t;
        s.best_leaf = leaf;
      }
    } else if (REFR && t < 0.) {
      nneg++;
      negmax = fmax(negmax, t);
    }
  }
  __device__ __forceinline__ void end() {
    if (REFR) {
      // Value-equal duplicates (m leaves that are `==`, shape.rs:638-646) have identical intersection lists, and the
      // containers list toggles once per copy: an even m never leaves an entry, an odd m behaves like one object whose
      // list position is that of its LAST copy.  cflags bit 1 = "m is odd", bit 0 = "this leaf is the last copy".
      bool odd = (nneg & 1) != 0, rep = true;
      if (s.cflags) {
        const uint8_t f = s.cflags[leaf];
        odd = odd && (f & 2);
        rep = (f & 1) != 0;
      }
      if (s.best_leaf == leaf) s.best_parity = odd ? 1 : 0;
      if (odd && rep) {
        // greatest (t, leaf): among equal t the later depth-first leaf was pushed last
        if (negmax > s.neg_t1 || (negmax == s.neg_t1 && leaf > s.neg_leaf1)) {
          s.neg_t2 = s.neg_t1, s.neg_leaf2 = s.neg_leaf1;
          s.neg_t1 = negmax, s.neg_leaf1 = leaf;
        } else if (negmax > s.neg_t2 || (negmax == s.neg_t2 && leaf > s.neg_leaf2)) {
          s.neg_t2 = negmax, s.neg_leaf2 = leaf;
        }
      }
    }
  }
};
// Transparent scenes, accelerated kernels with triangle runs: the radiance sweep of a triangle run only looks FORWARD (t >= 0,
// with the nearest-hit cut of opaque scenes); the intersections behind the origin, which only feed n1 / n2, are collected
// by a second, BACKWARD sweep (t < 0) and only for rays whose hit turns out to be transparent (sweep_accel<.., BACK>).  A line
// meets a triangle once, so every intersection lands in exactly one of the two accumulators and RadianceState ends up as the
// single two-sided sweep leaves it.
struct LeafAccFwd {
  RadianceState& s;
  int leaf;
  __device__ __forceinline__ LeafAccFwd(RadianceState& st, int l) : s(st), leaf(l) {}
  __device__ __forceinline__ void emit(double t) {
    if (t >= 0.) {
      if (t < s.best_t || (t == s.best_t && leaf < s.best_leaf)) {
        s.best_t = t;
        s.best_leaf = leaf;
        s.best_parity = 0;  // a triangle that is hit in front has no intersection behind: LeafAcc::end() would store 0 too
      }
    }
  }
  __device__ __forceinline__ void end() {}
};
struct LeafAccBack {
  RadianceState& s;
  int leaf;
  int nneg = 0;
  double negmax = -INFINITY;
  __device__ __forceinline__ LeafAccBack(RadianceState& st, int l) : s(st), leaf(l) {}
  __device__ __forceinline__ void emit(double t) {
    if (t < 0.) {
      nneg++;
      negmax = fmax(negmax, t);
    }
  }
  __device__ __forceinline__ void end() {  // LeafAcc<true>::end() without the hit leaf's parity (set by the forward sweep)
    bool odd = (nneg & 1) != 0, rep = true;
    if (s.cflags) {
      const uint8_t f = s.cflags[leaf];
      odd = odd && (f & 2);
      rep = (f & 1) != 0;
    }
    if (odd && rep) {
      if (negmax > s.neg_t1 || (negmax == s.neg_t1 && leaf > s.neg_leaf1)) {
        s.neg_t2 = s.neg_t1, s.neg_leaf2 = s.neg_leaf1;
        s.neg_t1 = negmax, s.neg_leaf1 = leaf;
      } else if (negmax > s.neg_t2 || (negmax == s.neg_t2 && leaf > s.neg_leaf2)) {
        s.neg_t2 = negmax, s.neg_leaf2 = leaf;
      }
    }
  }
};
struct ShadowAcc {  // is_shadowed: nearest t >= 0 is < distance  <=>  some t in [0, distance)  (world.rs:109-113)
  bool& shadowed;
  double dist;
  __device__ __forceinline__ void emit(double t) {
    if (t >= 0. && t < dist) shadowed = true;
  }
  __device__ __forceinline__ void end() {}
};

// shape.rs:587-606
__device__ __forceinline__ void check_axis(double mn, double mx, double origin, double direction, double& tmin,
                                           double& tmax) {
  double tmin_numerator = mn - origin;
  double tmax_numerator = mx - origin;
  if (fabs(direction) >= EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INFINITY;
    tmax = tmax_numerator * INFINITY;
  }
  if (tmin > tmax) {
    double t = tmin;
    tmin = tmax;
    tmax = t;
  }
}
// shape.rs:579-585
__device__ __forceinline__ bool check_cap(V3 o, V3 d, double t) {
  double x = o.x + t * d.x;
  double y = o.y + t * d.y;
  double z = o.z + t * d.z;
  return x * x + z * z <= fabs(y);
}

// Non-triangle leaf kinds of Shape::intersect (shape.rs:258-398), local (object-space) ray.
template <class Acc>
__device__ __forceinline__ void intersect_primitive(uint32_t kind, const double* params, V3 o, V3 d, Acc& acc) {
  switch (kind) {
    case K_SPHERE: {  // :258-273
      double a = dot(d, d);
      double b = 2. * dot(d, o);
      double c = dot(o, o) - 1.;
      double disc = b * b - 4. * a * c;
      if (disc >= 0.) {
        double sq = sqrt(disc);
        acc.emit((-b - sq) / (2. * a));
        acc.emit((-b + sq) / (2. * a));
      }
      break;
    }
    case K_PLANE: {  // :274-282
      if (fabs(d.y) >= EPSILON) acc.emit(-o.y / d.y);
      break;
    }
    case K_CUBE: {  // :283-319
      double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
      check_axis(-1.0, 1.0, o.x, d.x, xtmin, xtmax);
      check_axis(-1.0, 1.0, o.y, d.y, ytmin, ytmax);
      check_axis(-1.0, 1.0, o.z, d.z, ztmin, ztmax);
      double tmin = fmax(fmax(xtmin, ytmin), ztmin);
      double tmax = fmin(fmin(xtmax, ytmax), ztmax);
      if (tmax >= tmin) {
        acc.emit(tmin);
        acc.emit(tmax);
      }
      break;
    }
    case K_CYLINDER:
    case K_CONE: {  // :320-398
      const double minimum = params[0], maximum = params[1];
      const bool capped = params[2] != 0.0;
      double a, b, c;
      bool walls = true;
      if (kind == K_CYLINDER) {
        a = d.x * d.x + d.z * d.z;
        if (fabs(a - 0.0) < EPSILON) {
          walls = false;
        } else {
          b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
          c = o.x * o.x + o.z * o.z - 1.0;
        }
      } else {
        a = d.x * d.x - d.y * d.y + d.z * d.z;
        b = 2.0 * o.x * d.x - 2.0 * o.y * d.y + 2.0 * o.z * d.z;
        c = o.x * o.x - o.y * o.y + o.z * o.z;
        if (fabs(a - 0.0) < EPSILON) {
          walls = false;
          if (!(fabs(b - 0.0) < EPSILON)) acc.emit(-c / (2.0 * b));
        }
      }
      if (walls) {
        double disc = b * b - 4.0 * a * c;
        if (disc >= 0.0) {
          double sq = sqrt(disc);
          double t0 = (-b - sq) / (2. * a);
          double t1 = (-b + sq) / (2. * a);
          if (t0 > t1) {
            double t = t0;
            t0 = t1;
            t1 = t;
          }
          double y0 = o.y + t0 * d.y;
          if (minimum < y0 && y0 < maximum) acc.emit(t0);
          double y1 = o.y + t1 * d.y;
          if (minimum < y1 && y1 < maximum) acc.emit(t1);
        }
      }
      // intersect_caps :537-573
      if (capped && !(fabs(d.y - 0.0) < EPSILON)) {
        double t = (minimum - o.y) / d.y;
        if (check_cap(o, d, t)) acc.emit(t);
        t = (maximum - o.y) / d.y;
        if (check_cap(o, d, t)) acc.emit(t);
      }
      break;
    }
    default:
      break;
  }
}

// 1 / x for the slab tests of the filters: v_rcp_f32 (one instruction, 1 ulp, +-inf for zero and for a denormal) instead of the
// ten-instruction correctly rounded quotient (`1.0f / x`, which is also what __frcp_rn expands to).  Its extra relative error of
// 1.2e-7 moves a slab plane by < 1.2e-7 * (|o| + extent), a hundredth of the inflation every such test applies (>= 1e-5 * scale).
#ifndef RTC_FAST_RCP
#define RTC_FAST_RCP 1
#endif
#ifndef RTC_REACH_PAIR
#define RTC_REACH_PAIR 1  // C3 -2 %, C3zoom -2 % (profiles/r02/v20_*)
#endif
#ifndef RTC_STRAIGHT_CLASS
#define RTC_STRAIGHT_CLASS 1  // (C3 -2 %, C5 -2 %) the f32 classifications (group boxes, cube filter, background exit, triangle boxes) without exec-mask branches
#endif
__device__ __forceinline__ float rcp_approx(float x) {
#if RTC_FAST_RCP
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
#ifndef RTC_CUBE_FILTER
#define RTC_CUBE_FILTER 1  // f32 "certainly no intersection" test in front of the Cube arm (accelerated sweep)
#endif
// Conservative f32 test in front of the Cube arm of intersect_primitive, which always pays six correctly rounded f64
// divisions (check_axis): false only if the f64 evaluation is certain to report no intersection with t in
// [tlo, thi].  o, d = the f64 object-space ray (computed anyway).  Same construction as GroupGate::box_class: slab
// parameters in f32, every plane of [-1,1]^3 moved outwards by mb = 1e-5 (max|o| + 1), i.e. by e = mb / |d| in t,
// > 50x the f32 evaluation error (< 5 * 2^-24 (max|o| + 1) / |d|); an axis with |d| <= 2e-5 may be one that
// check_axis treats as parallel (|d| < EPSILON, absolute, shape.rs:593) and gives no constraint.
// (The same test in front of Sphere and Cylinder was measured and dropped: their arms reject a miss after ~17 f64
// operations, cheaper than this filter -- profiles/r01/v14_variants_prim_filter.txt.)
__device__ __forceinline__ bool cube_maybe(V3 o, V3 d, float tlo, float thi) {
  const float fo[3] = {(float)o.x, (float)o.y, (float)o.z};
  const float fd[3] = {(float)d.x, (float)d.y, (float)d.z};
  const float omax = fmaxf(fmaxf(fabsf(fo[0]), fabsf(fo[1])), fabsf(fo[2]));
  const float sum = fabsf(fo[0]) + fabsf(fo[1]) + fabsf(fo[2]) + fabsf(fd[0]) + fabsf(fd[1]) + fabsf(fd[2]);
  if (!(sum < 1e30f)) return true;  // NaN / huge: no f32 statement
  const float mb = 1e-5f * (omax + 1.0f);
  float tmin = tlo, tmax = thi;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#if RTC_STRAIGHT_CLASS
    const bool use = fabsf(fd[a]) > 2e-5f;
    const float inv = rcp_approx(fd[a]);
    const float t0 = (-1.0f - fo[a]) * inv, t1 = (1.0f - fo[a]) * inv;
    const float e = mb * fabsf(inv);
    tmin = use ? fmaxf(tmin, fminf(t0, t1) - e) : tmin;
    tmax = use ? fminf(tmax, fmaxf(t0, t1) + e) : tmax;
#else
    if (fabsf(fd[a]) > 2e-5f) {
      const float inv = rcp_approx(fd[a]);
      const float t0 = (-1.0f - fo[a]) * inv, t1 = (1.0f - fo[a]) * inv;
      const float e = mb * fabsf(inv);
      tmin = fmaxf(tmin, fminf(t0, t1) - e);
      tmax = fminf(tmax, fmaxf(t0, t1) + e);
    }
#endif
  }
  return !(tmin > tmax);
}

// Möller–Trumbore exactly as shape.rs:438-459 evaluates it; T = p1, e1, e2 (9 doubles).
template <class Acc>
__device__ __forceinline__ void intersect_triangle(const double* T, V3 o, V3 d, Acc& acc) {
  const V3 p1{T[0], T[1], T[2]}, e1{T[3], T[4], T[5]}, e2{T[6], T[7], T[8]};
  V3 dce2 = cross(d, e2);
  double det = dot(e1, dce2);
  if (!(fabs(det) < EPSILON)) {
    double f = 1.0 / det;
    V3 p1o = o - p1;
    double u = f * dot(p1o, dce2);
    if (!(u < 0.0 || u > 1.0)) {
      V3 oce1 = cross(p1o, e1);
      double v = f * dot(d, oce1);
      if (!(v < 0.0 || (u + v) > 1.0)) {
        double t = f * dot(e2, oce1);
        acc.emit(t);
      }
    }
  }
}

// ---- conservative f32 filters (rtc_accel.hpp) -----------------------------------------------------------
// They may only answer "certainly no intersection" (false) or "run the exact f64 test" (true).  Every reject
// condition is written so that a NaN makes it false (=> exact test).  FMA is used explicitly here (the
// translation unit is compiled with contraction off) because the error bounds below assume nothing about
// operation order beyond |rounding| <= 2^-24 per operation.
struct RayF {            // object-space ray of one triangle run, in f32, plus the per-ray bound factors
  float ox, oy, oz, dx, dy, dz;
  float D;               // >= max |d_i|
  float PD;              // >= (max |o_i| + pmax) * D
  float mb;              // box inflation: 1e-5 * (max |o_i| + ext)  (>= 80x the f32 error of (b - o) and of the slab products)
  float tlo, thi;        // parameter range of interest, already widened
};

__device__ __forceinline__ bool box_filter(const DBox& b, const RayF& r) {
  const float oxp = r.ox + r.mb, oxm = r.ox - r.mb;
  const float oyp = r.oy + r.mb, oym = r.oy - r.mb;
  const float ozp = r.oz + r.mb, ozm = r.oz - r.mb;
  // 1 / d where it is used, not held in three registers through the sweep
  const float ix = rcp_approx(r.dx), iy = rcp_approx(r.dy), iz = rcp_approx(r.dz);
  const float ax = (b.mn[0] - oxp) * ix, bx = (b.mx[0] - oxm) * ix;
  const float ay = (b.mn[1] - oyp) * iy, by = (b.mx[1] - oym) * iy;
  const float az = (b.mn[2] - ozp) * iz, bz = (b.mx[2] - ozm) * iz;
  // fminf/fmaxf ignore a NaN operand (0 * inf when the ray is parallel to a slab it starts on)
  const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  const bool miss = (tmin > tmax) || (tmax < r.tlo) || (tmin > r.thi);
  return !miss;
}

// ---- wave-level culling: one box per LANE against a bundle that encloses all of the wave's rays ----------
// A box is certainly missed by EVERY ray of the wave when the bundle test fails; the test costs about one ray-box
// test, but 64 boxes are tested at once (lane = box) instead of 64 rays against one box at a time.
#ifndef RTC_WAVE_REDUCE_DPP
#define RTC_WAVE_REDUCE_DPP 1
#endif
#if RTC_WAVE_REDUCE_DPP
// Wave-wide min / max without LDS: four row_shr steps leave each row's result in its lane 15, row_bcast:15 / row_bcast:31
// carry it to lanes 31 and 63, v_readlane(63) makes it wave-uniform (an SGPR).  Four independent chains are interleaved so
// that no step waits for the one before it (a DPP operand needs two wait states after the VALU write); the leading s_nop
// covers values and an EXEC mask written just before the block.  v_min_f32 / v_max_f32 return the other operand when one is a
// NaN, as fminf / fmaxf do.  Every lane of the wave must be active (all callers are in wave-uniform control flow).
// (The butterfly of ds_bpermute steps it replaces cost the sweep 16 dependent LDS round trips per ray bundle.)
#define RTC_DPP_STAGE(A, B, C, D, CTRL)                                                                  \
  A "_dpp %0, %0, %0 " CTRL "\n\t" B "_dpp %1, %1, %1 " CTRL "\n\t" C "_dpp %2, %2, %2 " CTRL "\n\t" \
    D "_dpp %3, %3, %3 " CTRL "\n\t"
#define RTC_DPP_REDUCE4(A, B, C, D)                                         \
  "s_nop 4\n\t" RTC_DPP_STAGE(A, B, C, D, "row_shr:1 row_mask:0xf bank_mask:0xf") \
  RTC_DPP_STAGE(A, B, C, D, "row_shr:2 row_mask:0xf bank_mask:0xf")        \
  RTC_DPP_STAGE(A, B, C, D, "row_shr:4 row_mask:0xf bank_mask:0xf")        \
  RTC_DPP_STAGE(A, B, C, D, "row_shr:8 row_mask:0xf bank_mask:0xf")        \
  RTC_DPP_STAGE(A, B, C, D, "row_bcast:15 row_mask:0xa bank_mask:0xf")     \
  RTC_DPP_STAGE(A, B, C, D, "row_bcast:31 row_mask:0xc bank_mask:0xf")
__device__ __forceinline__ float wave_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
// (a, b, c) -> their minima, d -> its maximum
__device__ __forceinline__ void wave_min3_max1(float& a, float& b, float& c, float& d) {
  asm volatile(RTC_DPP_REDUCE4("v_min_f32", "v_min_f32", "v_min_f32", "v_max_f32") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  a = wave_lane63(a), b = wave_lane63(b), c = wave_lane63(c), d = wave_lane63(d);
}
// (lo0, lo1) -> minima, (hi0, hi1) -> maxima
__device__ __forceinline__ void wave_min2_max2(float& lo0, float& hi0, float& lo1, float& hi1) {
  asm volatile(RTC_DPP_REDUCE4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32") : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1));
  lo0 = wave_lane63(lo0), hi0 = wave_lane63(hi0), lo1 = wave_lane63(lo1), hi1 = wave_lane63(hi1);
}
// one value (one register; the steps wait for one another: two wait states after each)
#define RTC_DPP_REDUCE1(OP)                                                                                                   \
  "s_nop 4\n\t" OP "_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP                                   \
  "_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP "_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP "_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t" OP                                   \
  "_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t" OP "_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
__device__ __forceinline__ float wave_min(float v) {
  asm volatile(RTC_DPP_REDUCE1("v_min_f32") : "+v"(v));
  return wave_lane63(v);
}
__device__ __forceinline__ float wave_max(float v) {
  asm volatile(RTC_DPP_REDUCE1("v_max_f32") : "+v"(v));
  return wave_lane63(v);
}
#else
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
#endif
// Anchor-and-slope bundle.  The wave picks the axis k on which all of its directions have the same sign and are
// never small (|d_k| >= 1/4 of the ray's largest component, so the slopes s_a = d_a / d_k are bounded by 4).  Each
// ray is the line  x_a = A_a + s_a * (x_k - A_k)  through an anchor point A on it: the origin for radiance rays,
// the far end point for shadow rays (the light, which all shadow rays of a wave share, so that both kinds of bundle
// are pyramids with a common apex and the intervals below are tight).  The bundle keeps the intervals of A and s
// over the wave and the extent [xlo, xhi] of the rays' parameter ranges along k.  A box is missed by every ray if it
// lies outside [xlo, xhi] along k, or if on one of the other two axes the interval  A_a + s_a * ([box_k] - A_k)
// does not meet the box.  Unlike intervals of 1/d this stays sharp when a direction component is near zero.
// Margins: anchors are inflated by mb = 1e-5 * (max|o| + max|A| + ext), which is > 6x the accumulated f32 error of
// the evaluation (25 * 2^-24 * (max|A| + ext): conversions of o and d, the slope quotient, |s| <= 4, product and
// sum roundings); slopes are widened by 1e-6 on top.  Operands are finite: huge rays disable the bundle (k = -1)
// and rtc_scene_create keeps coordinates beyond 1e30 off the accelerated path.
struct Bundle {
  int k;  // -1: no usable axis, every box passes
  float aklo, akhi, a1lo, a1hi, a2lo, a2hi;  // anchor intervals on axes k, k+1, k+2 (cyclic), inflated
  float s1lo, s1hi, s2lo, s2hi;              // slope intervals of axes k+1, k+2
  float xlo, xhi;                            // extent of the rays along k
  uint32_t neg;                              // bit a: every ray's direction is negative on axis a (front-to-back order)
};
// The bundle is wave-uniform: it waits in 64 bytes of the wave's LDS between its uses (one tile step, one test per visited
// tile and per cluster group) instead of occupying 16 VGPRs of every lane through the whole run.
typedef __attribute__((address_space(3))) volatile float lds_vf32;
__device__ __forceinline__ void store_bundle(lds_vf32* p, const Bundle& B, uint32_t lane) {
  if (lane == 0) {
    p[0] = B.aklo, p[1] = B.akhi, p[2] = B.a1lo, p[3] = B.a1hi, p[4] = B.a2lo, p[5] = B.a2hi;
    p[6] = B.s1lo, p[7] = B.s1hi, p[8] = B.s2lo, p[9] = B.s2hi, p[10] = B.xlo, p[11] = B.xhi;
  }
}
__device__ __forceinline__ Bundle load_bundle(lds_vf32* p, int k) {
  Bundle B;
  B.k = k, B.neg = 0;
  B.aklo = p[0], B.akhi = p[1], B.a1lo = p[2], B.a1hi = p[3], B.a2lo = p[4], B.a2hi = p[5];
  B.s1lo = p[6], B.s1hi = p[7], B.s2lo = p[8], B.s2hi = p[9], B.xlo = p[10], B.xhi = p[11];
  return B;
}
__device__ __forceinline__ float cyc(const float v[3], int k, int j) {  // v[(k + j) % 3] without dynamic indexing
  const float a = j == 0 ? v[0] : j == 1 ? v[1] : v[2];
  const float b = j == 0 ? v[1] : j == 1 ? v[2] : v[0];
  const float c = j == 0 ? v[2] : j == 1 ? v[0] : v[1];
  return k == 0 ? a : k == 1 ? b : c;
}
__device__ __forceinline__ Bundle make_bundle(const RayF& r, bool need, bool far_anchor) {
  Bundle B;
  B.neg = 0;
  const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
  const float ad[3] = {fabsf(r.dx), fabsf(r.dy), fabsf(r.dz)};
  const float dmax = fmaxf(fmaxf(ad[0], ad[1]), ad[2]);
  const float idm = 1.0f / dmax;
  float score[3];
#if RTC_WAVE_REDUCE_DPP
  // the margin does not depend on the axis chosen below (amax is symmetric in the axes): reduced together with the scores
  const float tf = far_anchor ? r.thi : 0.f;
  const float amax = far_anchor ? fmaxf(fmaxf(fabsf(fmaf(d[0], tf, o[0])), fabsf(fmaf(d[1], tf, o[1]))), fabsf(fmaf(d[2], tf, o[2])))
                                : fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fabsf(o[2]));
  float mb = need ? r.mb + 1e-5f * amax : 0.f;
  {
    float q0 = need ? ad[0] * idm : 1.0f, q1 = need ? ad[1] * idm : 1.0f, q2 = need ? ad[2] * idm : 1.0f;
    wave_min3_max1(q0, q1, q2, mb);  // NaN (d = 0) is dropped by v_min_f32: caught by `bad`
    score[0] = q0, score[1] = q1, score[2] = q2;
  }
#endif
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#if !RTC_WAVE_REDUCE_DPP
    const float q = wave_min(need ? ad[a] * idm : 1.0f);  // NaN (d = 0) is dropped by fminf: caught by `bad`
#else
    const float q = score[a];
#endif
    const bool all_pos = __ballot((need && !(d[a] > 0.f)) ? 1 : 0) == 0;
    const bool all_neg = __ballot((need && !(d[a] < 0.f)) ? 1 : 0) == 0;
    score[a] = (all_pos || all_neg) ? q : -1.f;
    if (all_neg) B.neg |= 1u << a;
  }
  int k = 0;
  if (score[1] > score[k]) k = 1;
  if (score[2] > score[k]) k = 2;
  const float best = k == 0 ? score[0] : k == 1 ? score[1] : score[2];
  const float ok_ = cyc(o, k, 0), o1 = cyc(o, k, 1), o2 = cyc(o, k, 2);
  const float dk = cyc(d, k, 0), d1 = cyc(d, k, 1), d2 = cyc(d, k, 2);
#if RTC_WAVE_REDUCE_DPP
  const float ak = far_anchor ? fmaf(dk, tf, ok_) : ok_;
  const float a1 = far_anchor ? fmaf(d1, tf, o1) : o1;
  const float a2 = far_anchor ? fmaf(d2, tf, o2) : o2;
#else
  const float tf = far_anchor ? r.thi : 0.f;
  const float ak = far_anchor ? fmaf(dk, tf, ok_) : ok_;
  const float a1 = far_anchor ? fmaf(d1, tf, o1) : o1;
  const float a2 = far_anchor ? fmaf(d2, tf, o2) : o2;
  const float amax = fmaxf(fmaxf(fabsf(ak), fabsf(a1)), fabsf(a2));
#endif
  const float big = fabsf(r.ox) + fabsf(r.oy) + fabsf(r.oz) + dmax + amax;
  const bool bad = need && !(big < 1e30f && dmax > 1e-30f);
  B.k = (best >= 0.25f && __ballot(bad ? 1 : 0) == 0) ? k : -1;
  if (B.k < 0) return B;
  const float ik = 1.0f / dk;
  const float s1 = d1 * ik, s2 = d2 * ik;
  const float xa = fmaf(dk, r.tlo, ok_), xb = fmaf(dk, r.thi, ok_);  // +-inf allowed (dk != 0)
#if RTC_WAVE_REDUCE_DPP
  // the margins are applied per lane, before the reductions (rounding is monotonic: min_i(x_i - m) == (min_i x_i) - m exactly),
  // so that the results leave v_readlane as finished wave-uniform values
  B.aklo = need ? ak - mb : INFINITY, B.akhi = need ? ak + mb : -INFINITY;
  B.a1lo = need ? a1 - mb : INFINITY, B.a1hi = need ? a1 + mb : -INFINITY;
  wave_min2_max2(B.aklo, B.akhi, B.a1lo, B.a1hi);
  B.a2lo = need ? a2 - mb : INFINITY, B.a2hi = need ? a2 + mb : -INFINITY;
  B.s1lo = need ? s1 - 1e-6f : INFINITY, B.s1hi = need ? s1 + 1e-6f : -INFINITY;
  wave_min2_max2(B.a2lo, B.a2hi, B.s1lo, B.s1hi);
  B.s2lo = need ? s2 - 1e-6f : INFINITY, B.s2hi = need ? s2 + 1e-6f : -INFINITY;
  B.xlo = need ? fminf(xa, xb) - mb : INFINITY, B.xhi = need ? fmaxf(xa, xb) + mb : -INFINITY;
  wave_min2_max2(B.s2lo, B.s2hi, B.xlo, B.xhi);
#else
  const float mb = wave_max(need ? r.mb + 1e-5f * amax : 0.f);
  B.aklo = wave_min(need ? ak : INFINITY) - mb, B.akhi = wave_max(need ? ak : -INFINITY) + mb;
  B.a1lo = wave_min(need ? a1 : INFINITY) - mb, B.a1hi = wave_max(need ? a1 : -INFINITY) + mb;
  B.a2lo = wave_min(need ? a2 : INFINITY) - mb, B.a2hi = wave_max(need ? a2 : -INFINITY) + mb;
  B.s1lo = wave_min(need ? s1 : INFINITY) - 1e-6f, B.s1hi = wave_max(need ? s1 : -INFINITY) + 1e-6f;
  B.s2lo = wave_min(need ? s2 : INFINITY) - 1e-6f, B.s2hi = wave_max(need ? s2 : -INFINITY) + 1e-6f;
  B.xlo = wave_min(need ? fminf(xa, xb) : INFINITY) - mb;
  B.xhi = wave_max(need ? fmaxf(xa, xb) : -INFINITY) + mb;
#endif
  return B;
}
#ifndef RTC_FAST_BUNDLE
#define RTC_FAST_BUNDLE 1  // make_bundle2 below instead of make_bundle for rays that share a point: two groups of wave reductions
                           // instead of four, one ballot instead of three reductions and six ballots for the choice of the axis
#endif
#if RTC_FAST_BUNDLE && RTC_WAVE_REDUCE_DPP
__device__ __forceinline__ float lane_f32(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
// The same bundle with less wave-level work (the construction above is 16 reductions and 6 ballots per sweep of a triangle run),
// for rays whose anchors are (nearly) one point: the light for shadow rays, whose far ends o + d * thi differ by rounding and the
// widening of thi; the common origin of a wave's primary rays, bit for bit.
// * Axis: the dominant axis of the first ray's direction, then the other two in turn; an axis is taken if every ray has the same
//   sign on it and at least a quarter of its largest component there (one ballot per candidate; the construction above reduces
//   three scores and takes the best axis, this one takes the first that qualifies: the same set of usable axes).
// * Anchors: the first ray's anchor A0 stands for all, and the three anchor intervals are A0 +- (margin + dev), dev = the greatest
//   distance |A_lane - A0| on any axis (one reduction, exact): they contain the intervals the construction above builds.
//   (`common` = false keeps per-axis minima and maxima; rays with scattered origins use make_bundle, whose best axis pays there.)
// * Slopes, extent along k, margin: reduced as above.
__device__ __forceinline__ Bundle make_bundle2(const RayF& r, bool need, bool far_anchor, bool common, uint32_t root_axis) {
  Bundle B;
  B.neg = 0, B.k = -1;
  const unsigned long long nm = __ballot(need ? 1 : 0);
  if (!nm) return B;
  const int l0 = __ffsll((long long)nm) - 1;
  const float o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
  const float ad[3] = {fabsf(r.dx), fabsf(r.dy), fabsf(r.dz)};
  const float dmax = fmaxf(fmaxf(ad[0], ad[1]), ad[2]);
  const float tf = far_anchor ? r.thi : 0.f;
  const float A[3] = {far_anchor ? fmaf(d[0], tf, o[0]) : o[0], far_anchor ? fmaf(d[1], tf, o[1]) : o[1], far_anchor ? fmaf(d[2], tf, o[2]) : o[2]};
  const float amax = fmaxf(fmaxf(fabsf(A[0]), fabsf(A[1])), fabsf(A[2]));
  const float big = fabsf(r.ox) + fabsf(r.oy) + fabsf(r.oz) + dmax + amax;
  if (__ballot((need && !(big < 1e30f && dmax > 1e-30f)) ? 1 : 0)) return B;  // huge or degenerate rays: no bundle, every box passes
  int k = __builtin_amdgcn_readlane(ad[0] >= ad[1] ? (ad[0] >= ad[2] ? 0 : 2) : (ad[1] >= ad[2] ? 1 : 2), l0);
  bool found = false;
#pragma unroll 1
  for (int tries = 0; tries < 3; ++tries) {
    const float dk_ = cyc(d, k, 0);
    const bool pos0 = lane_f32(dk_, l0) > 0.f;
    if (__ballot((need && !(((dk_ > 0.f) == pos0) && (dk_ != 0.f) && fabsf(dk_) * 4.f >= dmax)) ? 1 : 0) == 0) {
      found = true;
      break;
    }
    k = k == 2 ? 0 : k + 1;
  }
  if (!found) return B;
  B.k = k;
  B.neg = __ballot((need && !(cyc(d, (int)root_axis, 0) < 0.f)) ? 1 : 0) == 0 ? 1u << root_axis : 0u;
  const float ok_ = cyc(o, k, 0);
  const float dk = cyc(d, k, 0), d1 = cyc(d, k, 1), d2 = cyc(d, k, 2);
  const float ak = cyc(A, k, 0), a1 = cyc(A, k, 1), a2 = cyc(A, k, 2);
  const float ik = 1.0f / dk;
  const float s1 = d1 * ik, s2 = d2 * ik;
  B.s1lo = need ? s1 - 1e-6f : INFINITY, B.s1hi = need ? s1 + 1e-6f : -INFINITY;
  B.s2lo = need ? s2 - 1e-6f : INFINITY, B.s2hi = need ? s2 + 1e-6f : -INFINITY;
  wave_min2_max2(B.s1lo, B.s1hi, B.s2lo, B.s2hi);
  const float xa = fmaf(dk, r.tlo, ok_), xb = fmaf(dk, r.thi, ok_);  // +-inf allowed (dk != 0)
  float xlo = need ? fminf(xa, xb) : INFINITY, xhi = need ? fmaxf(xa, xb) : -INFINITY;
  float nmb = need ? -(r.mb + 1e-5f * amax) : 0.f;  // (minus the margin: its minimum is the greatest margin)
  if (common) {
    const float ak0 = lane_f32(ak, l0), a10 = lane_f32(a1, l0), a20 = lane_f32(a2, l0);
    float dev = need ? fmaxf(fmaxf(fabsf(ak - ak0), fabsf(a1 - a10)), fabsf(a2 - a20)) : 0.f;
    wave_min2_max2(xlo, xhi, nmb, dev);
    const float mb = -nmb, ma = mb + dev * 1.000001f;
    B.aklo = ak0 - ma, B.akhi = ak0 + ma, B.a1lo = a10 - ma, B.a1hi = a10 + ma, B.a2lo = a20 - ma, B.a2hi = a20 + ma;
    B.xlo = xlo - mb, B.xhi = xhi + mb;
  } else {
    float unused = 0.f;
    wave_min2_max2(xlo, xhi, nmb, unused);
    const float mb = -nmb;
    B.xlo = xlo - mb, B.xhi = xhi + mb;
    B.aklo = need ? ak - mb : INFINITY, B.akhi = need ? ak + mb : -INFINITY;
    B.a1lo = need ? a1 - mb : INFINITY, B.a1hi = need ? a1 + mb : -INFINITY;
    wave_min2_max2(B.aklo, B.akhi, B.a1lo, B.a1hi);
    B.a2lo = need ? a2 - mb : INFINITY, B.a2hi = need ? a2 + mb : -INFINITY;
    float u0 = 0.f, u1 = 0.f;
    wave_min2_max2(B.a2lo, B.a2hi, u0, u1);
  }
  return B;
}
#endif
// true unless every ray of the bundle certainly misses the box
__device__ __forceinline__ bool bundle_slabs(const float mn[3], const float mx[3], const Bundle& B) {
  if (B.k < 0) return true;
  const float lo = fmaxf(cyc(mn, B.k, 0), B.xlo), hi = fminf(cyc(mx, B.k, 0), B.xhi);
  const float ulo = lo - B.akhi, uhi = hi - B.aklo;  // x_k - A_k over the box's extent and all anchors
  const float p0 = B.s1lo * ulo, p1 = B.s1lo * uhi, p2 = B.s1hi * ulo, p3 = B.s1hi * uhi;
  const float r1lo = B.a1lo + fminf(fminf(p0, p1), fminf(p2, p3));
  const float r1hi = B.a1hi + fmaxf(fmaxf(p0, p1), fmaxf(p2, p3));
  const float q0 = B.s2lo * ulo, q1 = B.s2lo * uhi, q2 = B.s2hi * ulo, q3 = B.s2hi * uhi;
  const float r2lo = B.a2lo + fminf(fminf(q0, q1), fminf(q2, q3));
  const float r2hi = B.a2hi + fmaxf(fmaxf(q0, q1), fmaxf(q2, q3));
  const bool miss = (lo > hi) || (r1lo > cyc(mx, B.k, 1)) || (r1hi < cyc(mn, B.k, 1)) || (r2lo > cyc(mx, B.k, 2)) ||
                    (r2hi < cyc(mn, B.k, 2));
  return !miss;
}
// (First version: per axis, the interval of the origins and of the reciprocal directions -- loose whenever a direction
// component is near zero: C5 5.04 -> 4.24 ms with the form above, profiles/r01/v10_*.)
__device__ __forceinline__ bool bundle_box(const DBox& b, const Bundle& B) { return bundle_slabs(b.mn, b.mx, B); }
// The same test for ONE triangle per lane: its box is rebuilt from the f32 filter record (p1, p1+e1, p1+e2) and
// padded by `pad` = 1e-6 * (max |coordinate| of the run), four times the worst-case f32 error of those three
// points against the f64 vertices (<= 4.1 * 2^-24 * ext: two roundings of the record, one of the sum).  Any
// intersection the exact test can report lies on the f64 triangle, hence inside this box.  A record that
// overflowed f32 gives no usable box and passes.
__device__ __forceinline__ bool bundle_tri(const DTriF& t, float pad, const Bundle& B) {
  float mn[3], mx[3], s = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float v0 = t.p1[a], v1 = t.p1[a] + t.e1[a], v2 = t.p1[a] + t.e2[a];
    mn[a] = fminf(fminf(v0, v1), v2) - pad;
    mx[a] = fmaxf(fmaxf(v0, v1), v2) + pad;
    s += fabsf(v0) + fabsf(t.e1[a]) + fabsf(t.e2[a]);
  }
#if RTC_STRAIGHT_CLASS
  return !(s < INFINITY) | bundle_slabs(mn, mx, B);
#else
  if (!(s < INFINITY)) return true;
  return bundle_slabs(mn, mx, B);
#endif
}

// f32 evaluation of det, x = p1o.dce2, y = d.oce1 (the numerators of u and v) with absolute error bounds
//   |det' - det| <= D*c1,  2|x' - x| <= PD*c2,  2|y' - y| <= PD*c3        (DESIGN.md §3.3 derives the constants)
// Reject only if the reference's f64 evaluation (shape.rs:438-459) is certain to reject:
//   |det| < EPSILON,  u < 0,  u > 1,  v < 0,  u + v > 1.
#ifndef RTC_FILTER_STRAIGHT
#define RTC_FILTER_STRAIGHT 1  // C3 -4 %, C4 -8 %, C5 -4 % against the version with early exits below (profiles/r02/v18_*)
#endif
#if RTC_FILTER_STRAIGHT
// the same tests without early exits: every product is evaluated, one predicate at the end (no exec-mask branches)
__device__ __forceinline__ bool tri_filter(const DTriF& t, const RayF& r) {
  const float cx = fmaf(r.dy, t.e2[2], -(r.dz * t.e2[1]));
  const float cy = fmaf(r.dz, t.e2[0], -(r.dx * t.e2[2]));
  const float cz = fmaf(r.dx, t.e2[1], -(r.dy * t.e2[0]));
  const float det = fmaf(t.e1[2], cz, fmaf(t.e1[1], cy, t.e1[0] * cx));
  const float ad = fabsf(det);
  const float Edet = r.D * t.c1;
  const float px = r.ox - t.p1[0], py = r.oy - t.p1[1], pz = r.oz - t.p1[2];
  const float x = fmaf(pz, cz, fmaf(py, cy, px * cx));
  const float qx = fmaf(py, t.e1[2], -(pz * t.e1[1]));
  const float qy = fmaf(pz, t.e1[0], -(px * t.e1[2]));
  const float qz = fmaf(px, t.e1[1], -(py * t.e1[0]));
  const float y = fmaf(r.dz, qz, fmaf(r.dy, qy, r.dx * qx));
  const float xs = det < 0.f ? -x : x, ys = det < 0.f ? -y : y;
  const float Ex2 = r.PD * t.c2, Ey2 = r.PD * t.c3;
  const float lim = ad + 2.f * Edet;
  const float s = xs + ys;
  const bool parallel = ad + Edet < 0.999e-5f;
  const bool unsure = !(ad > Edet);
  const bool rej = (xs < -Ex2) | (xs - Ex2 > lim) | (ys < -Ey2) | (s - (Ex2 + Ey2) - fabsf(s) * 1e-6f > lim);
  return !parallel & (unsure | !rej);
}
#else
__device__ __forceinline__ bool tri_filter(const DTriF& t, const RayF& r) {
  const float cx = fmaf(r.dy, t.e2[2], -(r.dz * t.e2[1]));
  const float cy = fmaf(r.dz, t.e2[0], -(r.dx * t.e2[2]));
  const float cz = fmaf(r.dx, t.e2[1], -(r.dy * t.e2[0]));
  const float det = fmaf(t.e1[2], cz, fmaf(t.e1[1], cy, t.e1[0] * cx));
  const float ad = fabsf(det);
  const float Edet = r.D * t.c1;
  if (ad + Edet < 0.999e-5f) return false;  // |det| < EPSILON for sure: parallel, misses
  if (!(ad > Edet)) return true;            // sign of det unknown: let the exact test decide
  const float px = r.ox - t.p1[0], py = r.oy - t.p1[1], pz = r.oz - t.p1[2];
  const float x = fmaf(pz, cz, fmaf(py, cy, px * cx));
  const float xs = det < 0.f ? -x : x;      // x * sign(det): u = xs / |det|
  const float Ex2 = r.PD * t.c2;
  if (xs < -Ex2) return false;              // u < 0 for sure (and by more than f64 noise)
  const float lim = ad + 2.f * Edet;
  if (xs - Ex2 > lim) return false;         // u > 1 for sure
  const float qx = fmaf(py, t.e1[2], -(pz * t.e1[1]));
  const float qy = fmaf(pz, t.e1[0], -(px * t.e1[2]));
  const float qz = fmaf(px, t.e1[1], -(py * t.e1[0]));
  const float y = fmaf(r.dz, qz, fmaf(r.dy, qy, r.dx * qx));
  const float ys = det < 0.f ? -y : y;
  const float Ey2 = r.PD * t.c3;
  if (ys < -Ey2) return false;              // v < 0 for sure
  const float s = xs + ys;
  if (s - (Ex2 + Ey2) - fabsf(s) * 1e-6f > lim) return false;  // u + v > 1 for sure
  return true;
}
#endif

// Group bounding-box gate of Shape::intersect (shape.rs:399-425).  Segments are visited in depth-first order and
// every segment carries the chain of groups above it (root first) plus the number of leading entries it shares with
// the previous segment, so each lane keeps one pass bit per nesting DEPTH (bit d = "the ray passed the boxes of the
// chain's first d+1 groups") and a group's box is evaluated once per visit, only for rays that passed its parent —
// exactly what the reference's recursive Group arm does.  No limit on the number of groups; nesting depth <= 64.
struct GroupGate {
  uint64_t bits = 0;
  __device__ __forceinline__ void begin(const Ray&) { bits = 0; }
  // shape.rs:399-425 for one group box, exactly
  __device__ __forceinline__ bool box_exact(const DGroup& G, const Ray& r) const {
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(G.mn[0], G.mx[0], r.o.x, r.d.x, xtmin, xtmax);
    check_axis(G.mn[1], G.mx[1], r.o.y, r.d.y, ytmin, ytmax);
    check_axis(G.mn[2], G.mx[2], r.o.z, r.d.z, ztmin, ztmax);
    const double tmin = fmax(fmax(xtmin, ytmin), ztmin);
    const double tmax = fmin(fmin(xtmax, ytmax), ztmax);
    return tmax > tmin;  // strict, as the reference (shape.rs:425)
  }
  // f32 classification in front of the exact test (which costs six correctly rounded f64 divisions): 0 = the exact
  // test certainly fails, 1 = it certainly passes, 2 = undecided.  The slab parameters are evaluated once in f32;
  // moving every slab plane by mb = 1e-5 * (max|o| + max|box|) outwards (inwards) changes them by e = mb / |d|,
  // which is > 50x the f32 evaluation error (conversions of box, o and d, one subtraction, one quotient:
  // < 5 * 2^-24 * (max|o| + max|box|) / |d|), so an empty intersection of the widened intervals proves the exact
  // intervals do not overlap, and a non-empty intersection of the narrowed intervals (strict, like the reference's
  // tmax > tmin) proves they do.  Axes the reference may treat as parallel (|d| < EPSILON, absolute, shape.rs:593)
  // give no constraint for the first answer and forbid the second.  The test covers the whole line, as the
  // reference's does (no t >= 0 restriction).
  __device__ __forceinline__ int box_class(const DGroup& G, const Ray& r) const {
    const float fo[3] = {(float)r.o.x, (float)r.o.y, (float)r.o.z};
    const float fd[3] = {(float)r.d.x, (float)r.d.y, (float)r.d.z};
    const float omax = fmaxf(fmaxf(fabsf(fo[0]), fabsf(fo[1])), fabsf(fo[2]));
    const float mb = 1e-5f * (omax + G.fext);
    const float sum = fabsf(fo[0]) + fabsf(fo[1]) + fabsf(fo[2]) + fabsf(fd[0]) + fabsf(fd[1]) + fabsf(fd[2]) + G.fext;
    const bool bad = !(sum < 1e30f);  // NaN or huge operands: no f32 statement, and fminf/fmaxf below would hide a NaN
    float tminI = -INFINITY, tmaxI = INFINITY, tminD = -INFINITY, tmaxD = INFINITY;
    bool all_axes = true;
#if RTC_STRAIGHT_CLASS
    // no exec-mask branches: an axis that gives no statement is evaluated too (its infinities / NaNs are dropped by the selects)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const bool use = fabsf(fd[a]) > 2e-5f;
      const float inv = rcp_approx(fd[a]);
      const float t0 = (G.fmn[a] - fo[a]) * inv, t1 = (G.fmx[a] - fo[a]) * inv;
      const float lo = fminf(t0, t1), hi = fmaxf(t0, t1), e = mb * fabsf(inv);
      tminI = use ? fmaxf(tminI, lo - e) : tminI, tmaxI = use ? fminf(tmaxI, hi + e) : tmaxI;
      tminD = use ? fmaxf(tminD, lo + e) : tminD, tmaxD = use ? fminf(tmaxD, hi - e) : tmaxD;
      all_axes = all_axes & use;
    }
    int c = 2;  // includes every NaN / infinite case
    c = (all_axes & (tmaxD > tminD)) ? 1 : c;
    c = (tminI > tmaxI) ? 0 : c;
    return bad ? 2 : c;
#else
    if (bad) return 2;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (fabsf(fd[a]) > 2e-5f) {
        const float inv = rcp_approx(fd[a]);
        const float t0 = (G.fmn[a] - fo[a]) * inv, t1 = (G.fmx[a] - fo[a]) * inv;
        const float lo = fminf(t0, t1), hi = fmaxf(t0, t1), e = mb * fabsf(inv);
        tminI = fmaxf(tminI, lo - e), tmaxI = fminf(tmaxI, hi + e);
        tminD = fmaxf(tminD, lo + e), tmaxD = fminf(tmaxD, hi - e);
      } else {
        all_axes = false;
      }
    }
    if (tminI > tmaxI) return 0;
    if (all_axes && tmaxD > tminD) return 1;
    return 2;  // includes every NaN / infinite case
#endif
  }
  __device__ __forceinline__ bool box(const DGroup& G, const Ray& r) const {
#if RTC_GROUP_PREFILTER
    const int c = box_class(G, r);
    if (c != 2) return c == 1;
#endif
    return box_exact(G, r);
  }
  // Does the ray reach this segment?  Must be called for every segment, in order.
  // CL: read the record through the constant address space (ldc; pays in the mesh kernels, not in those of primitive-only
  // scenes, which visit many small segments: C3 / C4 / C5 -1.5 %, C2 +2.5 %, profiles/r02/v10_*)
  template <bool CL>
  __device__ __forceinline__ bool reach(const DScene& S, const DSegRec& rec, const DSegment& seg, const Ray& r, bool active) {
#if RTC_REACH_PAIR
    if (CL && seg.chain_common == 0 && seg.chain_len == 2) {  // wave-uniform: a mesh under two groups (obj_file.rs:116-126)
      // both records are requested together and both classifications are evaluated for every lane (they are pure); the
      // exact test still runs only where the reference's would: box 1 for the rays that passed box 0
      const DGroup g0 = ldc(&rec.g[0]), g1 = ldc(&rec.g[1]);
#if RTC_GROUP_PREFILTER
      const int c0 = box_class(g0, r), c1 = box_class(g1, r);
#else
      const int c0 = 2, c1 = 2;
#endif
      bool ok0 = active & (c0 == 1);
      if (active && c0 == 2) ok0 = box_exact(g0, r);
      bool ok1 = ok0 & (c1 == 1);
      if (ok0 && c1 == 2) ok1 = box_exact(g1, r);
      bits = (ok0 ? 1ull : 0ull) | (ok1 ? 2ull : 0ull) | (bits & ~3ull);
      return ok1;
    }
#endif
    for (uint32_t d = seg.chain_common; d < seg.chain_len; ++d) {  // wave-uniform
      bool ok = active && (d == 0 || ((bits >> (d - 1)) & 1ull));
      if (ok) ok = d < (uint32_t)SEG_INLINE_GROUPS ? box(CL ? ldc(&rec.g[d]) : rec.g[d], r) : box(S.groups[S.seg_chain[seg.chain_off + d]], r);
      bits = ok ? (bits | (1ull << d)) : (bits & ~(1ull << d));
    }
    return seg.chain_len == 0 ? active : (((bits >> (seg.chain_len - 1)) & 1ull) != 0);
  }
};

// Shape::normal_at (shape.rs:466-519) for a non-triangle leaf.
__device__ __forceinline__ V3 normal_at_primitive(uint32_t kind, const double* params, const double* inv, V3 wp) {
  V3 lp = xf_point(inv, wp);
  V3 ln;
  switch (kind) {
    case K_SPHERE:
      ln = lp;  // local_point - point(0,0,0)
      break;
    case K_PLANE:
      ln = {0.0, 1.0, 0.0};
      break;
    case K_CUBE: {
      double xa = fabs(lp.x), ya = fabs(lp.y), za = fabs(lp.z);
      double maxc = fmax(fmax(xa, ya), za);
      if (maxc == xa)
        ln = {lp.x, 0.0, 0.0};
      else if (maxc == ya)
        ln = {0.0, lp.y, 0.0};
      else
        ln = {0.0, 0.0, lp.z};
      break;
    }
    case K_CYLINDER: {
      double dist = lp.x * lp.x + lp.z * lp.z;
      if (dist < 1.0 && lp.y >= params[1] - EPSILON)
        ln = {0.0, 1.0, 0.0};
      else if (dist < 1.0 && lp.y <= params[0] + EPSILON)
        ln = {0.0, -1.0, 0.0};
      else
        ln = {lp.x, 0.0, lp.z};
      break;
    }
    case K_CONE: {
      double y = sqrt(lp.x * lp.x + lp.z * lp.z);
      if (lp.y > 0.0) y = -y;
      ln = {lp.x, y, lp.z};
      break;
    }
    default:
      ln = {0., 0., 0.};
      break;
  }
  V3 wn = normalize(xf_normal(inv, ln));  // normal_to_world (:623-635)
  return normalize(wn);                   // normal_at normalises again (:513-518)
}

// Pattern::color_at_shape + color_at (pattern.rs:68-103)
__device__ __forceinline__ void pattern_color(const DPattern& P, const double* shape_inv, V3 wp, double out[3]) {
  V3 op = xf_point(shape_inv, wp);
  V3 pp = xf_point(P.inv, op);
  bool first;
  switch (P.kind) {
    case 1:  // Stripe
      first = fmod(floor(pp.x), 2.0) == 0.0;
      break;
    case 2: {  // Gradient: from + (to - from) * (x - floor(x))
      double f = pp.x - floor(pp.x);
      out[0] = P.a[0] + (P.b[0] - P.a[0]) * f;
      out[1] = P.a[1] + (P.b[1] - P.a[1]) * f;
      out[2] = P.a[2] + (P.b[2] - P.a[2]) * f;
      return;
    }
    case 3:  // Ring
      first = fmod(floor(sqrt(pp.x * pp.x + pp.z * pp.z)), 2.0) == 0.0;
      break;
    case 4:  // Checkers
      first = fmod(floor(pp.x) + floor(pp.y) + floor(pp.z), 2.0) == 0.0;
      break;
    default:  // Test
      out[0] = pp.x, out[1] = pp.y, out[2] = pp.z;
      return;
  }
  const double* c = first ? P.a : P.b;
  out[0] = c[0], out[1] = c[1], out[2] = c[2];
}

// powf(shininess) of the specular term (material.rs:69).  Kept out of line on purpose: when pow() is inlined
// into the megakernel hipcc hoists its ~12 polynomial constants into VGPRs at kernel entry and then SPILLS them to
// scratch for the whole sweep (88 B written per thread, 180 MB of HBM writes per 1080p frame, seen as WRITE_SIZE
// in rocprofv3) to reload them once per shaded hit.  One call per shaded hit is nothing next to a sweep.
#if RTC_POW_INLINE
__device__ __forceinline__ double specular_pow(double base, double exponent) { return pow(base, exponent); }
#else
__device__ __attribute__((noinline)) double specular_pow(double base, double exponent) { return pow(base, exponent); }
#endif

// Material::lighting (material.rs:32-75)
__device__ __forceinline__ void lighting(const DScene& S, const DMaterial& m, const double* shape_inv, V3 point,
                                         V3 eyev, V3 normalv, bool in_shadow, double out[3]) {
  double color[3];
  if (m.pattern >= 0)
    pattern_color(S.pats[m.pattern], shape_inv, point, color);
  else
    color[0] = m.color[0], color[1] = m.color[1], color[2] = m.color[2];
  double eff[3] = {color[0] * S.light_int[0], color[1] * S.light_int[1], color[2] * S.light_int[2]};
  V3 lightv = normalize(V3{S.light_pos[0], S.light_pos[1], S.light_pos[2]} - point);
  double amb[3] = {eff[0] * m.ambient, eff[1] * m.ambient, eff[2] * m.ambient};
  double dif[3] = {0., 0., 0.}, spe[3] = {0., 0., 0.};
  if (!in_shadow) {
    double light_dot_normal = dot(lightv, normalv);
    if (light_dot_normal >= 0.) {
      for (int k = 0; k < 3; ++k) dif[k] = eff[k] * m.diffuse * light_dot_normal;
      V3 reflectv = reflect(-lightv, normalv);
      double reflect_dot_eye = dot(reflectv, eyev);
      if (reflect_dot_eye > 0.) {
        double factor = specular_pow(reflect_dot_eye, m.shininess);
        for (int k = 0; k < 3; ++k) spe[k] = S.light_int[k] * m.specular * factor;
      }
    }
  }
  for (int k = 0; k < 3; ++k) out[k] = amb[k] + dif[k] + spe[k];
}

}  // namespace rtc
