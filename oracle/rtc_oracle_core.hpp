// rtc_oracle_core.hpp — CPU oracle internals (TEST INFRASTRUCTURE ONLY; see rtc_oracle.h).
//
// A statement-for-statement f64 restatement of the reference's per-pixel path.  Every
// function cites the reference file:line it follows (paths are under /root/reference/src).
// Operation order inside dot / mat*vec / sums is kept exactly as the reference writes it,
// so that on an IEEE-754 machine without FMA contraction (-ffp-contract=off) results are
// the same doubles the Rust code produces, up to libm differences in pow/tan/sin/cos.
//
// Rust -> C++ notes
//   x.powi(2), x.powf(2.)  -> x*x        (LLVM folds both to one fmul in release builds)
//   x.powi(5)              -> x*((x*x)*(x*x))  (LLVM ExpandPowI binary decomposition)
//   f64::max/min           -> fmax/fmin   (NaN-ignoring, same as Rust)
//   a % b on f64           -> fmod
//   sort_by(partial_cmp.unwrap_or(Equal)) -> std::stable_sort on t <
//   usize `remaining - 1`  -> wraps (release semantics) and raises a flag (debug: panic)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <string>
#include <vector>

namespace orc {

// utils.rs:2-6
constexpr double EPSILON = 0.00001;
inline bool is_almost_equal(double a, double b) { return std::fabs(a - b) < EPSILON; }

// ---------------------------------------------------------------- tuple.rs
struct Tuple {
  double x, y, z, w;
  static Tuple zero() { return {0., 0., 0., 0.}; }                      // tuple.rs:19-26
  static Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }   // :36-38
  static Tuple vector(double x, double y, double z) { return {x, y, z, 0.0}; }  // :40-42
  bool is_point() const { return w == 1.0; }
  bool is_vector() const { return w == 0.0; }
  // tuple.rs:43-48 (".w really?!")
  double magnitude() const { return std::sqrt(x * x + y * y + z * z + w * w); }
  // tuple.rs:50-66
  Tuple normalize() const {
    double m = magnitude();
    if (m == 0.0) return zero();
    return {x / m, y / m, z / m, w / m};
  }
  // tuple.rs:68-73
  double dot(const Tuple& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
  // tuple.rs:75-83
  Tuple cross(const Tuple& o) const {
    return vector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
  }
};
inline Tuple operator+(Tuple a, Tuple b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }  // :102-113
inline Tuple operator-(Tuple a, Tuple b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }  // :115-126
inline Tuple operator-(Tuple a) { return {-a.x, -a.y, -a.z, -a.w}; }                              // :128-134
inline Tuple operator*(Tuple a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }        // :136-143
inline Tuple operator/(Tuple a, double s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }        // :145-152
// tuple.rs:86-90  (*self) - (*normal) * 2. * self.dot(normal)
inline Tuple reflect(Tuple v, Tuple n) { return v - (n * 2.) * v.dot(n); }
// tuple.rs:93-100 (approximate PartialEq)
inline bool approx_eq(const Tuple& a, const Tuple& b) {
  return is_almost_equal(a.x, b.x) && is_almost_equal(a.y, b.y) && is_almost_equal(a.z, b.z) &&
         is_almost_equal(a.w, b.w);
}

// ---------------------------------------------------------------- color.rs
struct Color {
  double red, green, blue;
};
inline Color operator+(Color a, Color b) { return {a.red + b.red, a.green + b.green, a.blue + b.blue}; }  // :55-65
inline Color operator-(Color a, Color b) { return {a.red - b.red, a.green - b.green, a.blue - b.blue}; }  // :67-77
inline Color operator*(Color a, double s) { return {a.red * s, a.green * s, a.blue * s}; }                // :79-85
inline Color operator*(Color a, Color b) { return {a.red * b.red, a.green * b.green, a.blue * b.blue}; }  // :88-98
inline bool approx_eq(const Color& a, const Color& b) {                                                   // :47-53
  return is_almost_equal(a.red, b.red) && is_almost_equal(a.green, b.green) && is_almost_equal(a.blue, b.blue);
}
constexpr Color BLACK{0., 0., 0.};
constexpr Color WHITE{1., 1., 1.};

// ---------------------------------------------------------------- matrix.rs
struct Matrix4 {
  double v[4][4];
  static Matrix4 zero() {
    Matrix4 m;
    std::memset(m.v, 0, sizeof m.v);
    return m;
  }
  static Matrix4 identity() {  // :19-27
    Matrix4 m = zero();
    for (int i = 0; i < 4; ++i) m.v[i][i] = 1.;
    return m;
  }
  Matrix4 transpose() const {  // :29-39
    Matrix4 r = zero();
    for (int row = 0; row < 4; ++row)
      for (int col = 0; col < 4; ++col) r.v[col][row] = v[row][col];
    return r;
  }
};

namespace detail {
// matrix.rs:41-52 with SIZE==2
inline double det2(const double m[2][2]) { return m[0][0] * m[1][1] - m[0][1] * m[1][0]; }
// submatrix2 (:84-113) + minor/cofactor (:115-137) + determinant (:41-52) for SIZE==3
inline double det3(const double m[3][3]) {
  double result = 0.;
  for (int column = 0; column < 3; ++column) {
    double s[2][2];
    int rx = 0;
    for (int x = 0; x < 3; ++x) {
      if (x == 0) continue;
      int ry = 0;
      for (int y = 0; y < 3; ++y) {
        if (y == column) continue;
        s[rx][ry++] = m[x][y];
      }
      ++rx;
    }
    double minor = det2(s);
    double cof = ((0 + column) % 2 == 0) ? minor : -minor;
    result += m[0][column] * cof;
  }
  return result;
}
}  // namespace detail

// submatrix3 (:55-82) + minor (:115-128) + cofactor (:131-137) for SIZE==4
inline double cofactor4(const Matrix4& m, int row, int col) {
  double s[3][3];
  int rx = 0;
  for (int x = 0; x < 4; ++x) {
    if (x == row) continue;
    int ry = 0;
    for (int y = 0; y < 4; ++y) {
      if (y == col) continue;
      s[rx][ry++] = m.v[x][y];
    }
    ++rx;
  }
  double minor = detail::det3(s);
  return ((row + col) % 2 == 0) ? minor : -minor;
}
// matrix.rs:41-52 SIZE==4
inline double determinant4(const Matrix4& m) {
  double result = 0.;
  for (int column = 0; column < 4; ++column) result += m.v[0][column] * cofactor4(m, 0, column);
  return result;
}
// matrix.rs:138-157 — note determinant() is re-evaluated for every element, as in the reference.
inline std::optional<Matrix4> inverse(const Matrix4& m) {
  if (is_almost_equal(determinant4(m), 0.)) return std::nullopt;
  Matrix4 r = Matrix4::zero();
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) {
      double c = cofactor4(m, row, col);
      r.v[col][row] = c / determinant4(m);
    }
  return r;
}
// matrix.rs:174-185
inline bool approx_eq(const Matrix4& a, const Matrix4& b) {
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x)
      if (!is_almost_equal(a.v[x][y], b.v[x][y])) return false;
  return true;
}
// matrix.rs:187-205
inline Matrix4 operator*(const Matrix4& a, const Matrix4& b) {
  Matrix4 r = Matrix4::zero();
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) {
      double val = 0.;
      for (int n = 0; n < 4; ++n) val += a.v[row][n] * b.v[n][col];
      r.v[row][col] = val;
    }
  return r;
}
// matrix.rs:207-227
inline Tuple operator*(const Matrix4& m, const Tuple& t) {
  Tuple r = Tuple::zero();
  r.x = m.v[0][0] * t.x + m.v[0][1] * t.y + m.v[0][2] * t.z + m.v[0][3] * t.w;
  r.y = m.v[1][0] * t.x + m.v[1][1] * t.y + m.v[1][2] * t.z + m.v[1][3] * t.w;
  r.z = m.v[2][0] * t.x + m.v[2][1] * t.y + m.v[2][2] * t.z + m.v[2][3] * t.w;
  r.w = m.v[3][0] * t.x + m.v[3][1] * t.y + m.v[3][2] * t.z + m.v[3][3] * t.w;
  return r;
}

// ---------------------------------------------------------------- transformations.rs
inline Matrix4 translation(double x, double y, double z) {  // :4-11
  Matrix4 r = Matrix4::identity();
  r.v[0][3] = x;
  r.v[1][3] = y;
  r.v[2][3] = z;
  return r;
}
inline Matrix4 scaling(double x, double y, double z) {  // :13-21
  Matrix4 r = Matrix4::identity();
  r.v[0][0] = x;
  r.v[1][1] = y;
  r.v[2][2] = z;
  return r;
}
inline Matrix4 rotation_x(double rad) {  // :23-35
  Matrix4 r = Matrix4::identity();
  double c = std::cos(rad);
  r.v[1][1] = c;
  r.v[2][2] = c;
  double s = std::sin(rad);
  r.v[1][2] = -s;
  r.v[2][1] = s;
  return r;
}
inline Matrix4 rotation_y(double rad) {  // :37-49
  Matrix4 r = Matrix4::identity();
  double c = std::cos(rad);
  r.v[0][0] = c;
  r.v[2][2] = c;
  double s = std::sin(rad);
  r.v[0][2] = s;
  r.v[2][0] = -s;
  return r;
}
inline Matrix4 rotation_z(double rad) {  // :51-63
  Matrix4 r = Matrix4::identity();
  double c = std::cos(rad);
  r.v[0][0] = c;
  r.v[1][1] = c;
  double s = std::sin(rad);
  r.v[0][1] = -s;
  r.v[1][0] = s;
  return r;
}
inline Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {  // :65-78
  Matrix4 r = Matrix4::identity();
  r.v[0][1] = xy;
  r.v[0][2] = xz;
  r.v[1][0] = yx;
  r.v[1][2] = yz;
  r.v[2][0] = zx;
  r.v[2][1] = zy;
  return r;
}
inline Matrix4 view_transform(Tuple from, Tuple to, Tuple up) {  // :80-93
  Tuple forward = (to - from).normalize();
  Tuple upn = up.normalize();
  Tuple left = forward.cross(upn);
  Tuple true_up = left.cross(forward);
  Matrix4 o = Matrix4::zero();
  o.v[0][0] = left.x, o.v[0][1] = left.y, o.v[0][2] = left.z, o.v[0][3] = 0.;
  o.v[1][0] = true_up.x, o.v[1][1] = true_up.y, o.v[1][2] = true_up.z, o.v[1][3] = 0.;
  o.v[2][0] = -forward.x, o.v[2][1] = -forward.y, o.v[2][2] = -forward.z, o.v[2][3] = 0.;
  o.v[3][0] = 0., o.v[3][1] = 0., o.v[3][2] = 0., o.v[3][3] = 1.;
  return o * translation(-from.x, -from.y, -from.z);
}

// ---------------------------------------------------------------- ray.rs
struct Ray {
  Tuple origin, direction;
  Tuple position(double t) const { return origin + direction * t; }               // :15-17
  Ray transform(const Matrix4& m) const { return {m * origin, m * direction}; }    // :19-24
};

// ---------------------------------------------------------------- light.rs
struct Light {
  Tuple position;
  Color intensity;
};

// ---------------------------------------------------------------- pattern.rs
enum class PatternKind { Stripe = 1, Gradient = 2, Ring = 3, Checkers = 4, Test = 5 };
struct Pattern {
  Matrix4 transform = Matrix4::identity();
  Matrix4 transform_inverse = Matrix4::identity();
  PatternKind kind = PatternKind::Test;
  Color a{0, 0, 0}, b{0, 0, 0};
  // pattern.rs:63-66
  bool set_transform(const Matrix4& t) {
    transform = t;
    auto inv = inverse(t);
    if (!inv) return false;
    transform_inverse = *inv;
    return true;
  }
  // pattern.rs:68-95
  Color color_at(const Tuple& p) const {
    switch (kind) {
      case PatternKind::Stripe:
        return (std::fmod(std::floor(p.x), 2.0) == 0.0) ? a : b;
      case PatternKind::Gradient:
        return a + (b - a) * (p.x - std::floor(p.x));
      case PatternKind::Ring:
        return (std::fmod(std::floor(std::sqrt(p.x * p.x + p.z * p.z)), 2.0) == 0.0) ? a : b;
      case PatternKind::Checkers:
        return (std::fmod(std::floor(p.x) + std::floor(p.y) + std::floor(p.z), 2.0) == 0.0) ? a : b;
      case PatternKind::Test:
        return {p.x, p.y, p.z};
    }
    return BLACK;
  }
};
// derived PartialEq (pattern.rs:3,14): Matrix approx, kind + colours approx
inline bool pattern_eq(const Pattern& x, const Pattern& y) {
  if (!approx_eq(x.transform, y.transform) || !approx_eq(x.transform_inverse, y.transform_inverse)) return false;
  if (x.kind != y.kind) return false;
  if (x.kind == PatternKind::Test) return true;
  return approx_eq(x.a, y.a) && approx_eq(x.b, y.b);
}

// ---------------------------------------------------------------- material.rs
struct Material {  // :4-29
  Color color = WHITE;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0, reflective = 0.0;
  std::optional<Pattern> pattern;
  double transparency = 0.0, refractive_index = 1.0;
};
// derived PartialEq (material.rs:3): Color approx, f64 exact
inline bool material_eq(const Material& a, const Material& b) {
  if (!approx_eq(a.color, b.color)) return false;
  if (!(a.ambient == b.ambient && a.diffuse == b.diffuse && a.specular == b.specular &&
        a.shininess == b.shininess && a.reflective == b.reflective))
    return false;
  if (a.pattern.has_value() != b.pattern.has_value()) return false;
  if (a.pattern && !pattern_eq(*a.pattern, *b.pattern)) return false;
  return a.transparency == b.transparency && a.refractive_index == b.refractive_index;
}

// ---------------------------------------------------------------- shape.rs
enum class Kind { Sphere, Plane, Cube, Cylinder, Cone, Group, Triangle };

struct Bounds {
  Tuple min, max;
};

struct Shape {
  Kind kind = Kind::Sphere;
  // Cylinder / Cone
  double minimum = 0, maximum = 0;
  bool capped = false;
  // Group
  std::vector<Shape> shapes;
  // Triangle (shape.rs:31-38)
  Tuple p1{}, p2{}, p3{}, e1{}, e2{}, normal{};

  Matrix4 transform = Matrix4::identity();
  Matrix4 transform_inverse = Matrix4::identity();
  Matrix4 transform_inverse_transpose = Matrix4::identity();
  bool transformed = false;
  Material material;

  // oracle-only cache for ORC_MODE_CACHED (never consulted in faithful mode)
  mutable bool bounds_cached = false;
  mutable Bounds bounds_cache{};
};

inline Shape make_kind(Kind k) {
  Shape s;
  s.kind = k;
  return s;
}
inline Shape sphere() { return make_kind(Kind::Sphere); }  // shape.rs:52-61
inline Shape glass_sphere() {                               // shape.rs:63-76
  Shape s = sphere();
  s.material.transparency = 1.0;
  s.material.refractive_index = 1.5;
  return s;
}
inline Shape plane() { return make_kind(Kind::Plane); }  // :78-87
inline Shape cube() { return make_kind(Kind::Cube); }    // :89-98
inline Shape cylinder(double mn, double mx, bool capped) {  // :100-128
  Shape s = make_kind(Kind::Cylinder);
  s.minimum = mn, s.maximum = mx, s.capped = capped;
  return s;
}
inline Shape cone(double mn, double mx, bool capped) {  // :130-158
  Shape s = make_kind(Kind::Cone);
  s.minimum = mn, s.maximum = mx, s.capped = capped;
  return s;
}
inline Shape group() { return make_kind(Kind::Group); }  // :160-169
inline Shape triangle(Tuple p1, Tuple p2, Tuple p3) {    // :171-193
  Shape s = make_kind(Kind::Triangle);
  s.p1 = p1, s.p2 = p2, s.p3 = p3;
  s.e1 = p2 - p1;
  s.e2 = p3 - p1;
  s.normal = s.e2.cross(s.e1).normalize();
  return s;
}

// shape.rs:207-218
inline bool set_transform_internal(Shape& s, const Matrix4& t) {
  if (s.kind == Kind::Group) {
    for (auto& c : s.shapes)
      if (!set_transform_internal(c, t)) return false;
    return true;
  }
  s.transform = t * s.transform;
  auto inv = inverse(s.transform);
  if (!inv) return false;
  s.transform_inverse = *inv;
  s.transform_inverse_transpose = s.transform_inverse.transpose();
  s.bounds_cached = false;
  return true;
}
// shape.rs:196-205; returns 0 ok, -2 called twice, -1 not invertible
inline int set_transform(Shape& s, const Matrix4& t) {
  if (s.transformed) return -2;
  s.transformed = true;
  return set_transform_internal(s, t) ? 0 : -1;
}
// shape.rs:220-229
inline void set_material(Shape& s, const Material& m) {
  if (s.kind == Kind::Group) {
    for (auto& c : s.shapes) set_material(c, m);
  } else {
    s.material = m;
  }
}

// impl PartialEq for Shape (shape.rs:638-646) + derived ShapeKind eq (:13)
inline bool shape_eq(const Shape& a, const Shape& b);
inline bool kind_eq(const Shape& a, const Shape& b) {
  if (a.kind != b.kind) return false;
  switch (a.kind) {
    case Kind::Sphere:
    case Kind::Plane:
    case Kind::Cube:
      return true;
    case Kind::Cylinder:
    case Kind::Cone:
      return a.minimum == b.minimum && a.maximum == b.maximum && a.capped == b.capped;
    case Kind::Group:
      if (a.shapes.size() != b.shapes.size()) return false;
      for (size_t i = 0; i < a.shapes.size(); ++i)
        if (!shape_eq(a.shapes[i], b.shapes[i])) return false;
      return true;
    case Kind::Triangle:
      return approx_eq(a.p1, b.p1) && approx_eq(a.p2, b.p2) && approx_eq(a.p3, b.p3) && approx_eq(a.e1, b.e1) &&
             approx_eq(a.e2, b.e2) && approx_eq(a.normal, b.normal);
  }
  return false;
}
inline bool shape_eq(const Shape& a, const Shape& b) {
  return kind_eq(a, b) && approx_eq(a.transform, b.transform) && material_eq(a.material, b.material);
}

// ---------------------------------------------------------------- bounds.rs
inline void bounds_add(Bounds& b, const Tuple& p) {  // :142-151
  b.min.x = std::fmin(b.min.x, p.x);
  b.min.y = std::fmin(b.min.y, p.y);
  b.min.z = std::fmin(b.min.z, p.z);
  b.max.x = std::fmax(b.max.x, p.x);
  b.max.y = std::fmax(b.max.y, p.y);
  b.max.z = std::fmax(b.max.z, p.z);
}
inline Bounds bounds_new(const Shape& s) {  // :11-140
  constexpr double INF = std::numeric_limits<double>::infinity();
  switch (s.kind) {
    case Kind::Sphere:
    case Kind::Cube:
      return {Tuple::point(-1., -1., -1.), Tuple::point(1., 1., 1.)};
    case Kind::Plane:
      return {Tuple::point(-1., -1., 0.), Tuple::point(1., 1., 0.)};
    case Kind::Cylinder:
    case Kind::Cone:
      if (s.capped) return {Tuple::point(-1., s.minimum, -1.), Tuple::point(1., s.maximum, 1.)};
      return {Tuple::point(-1., -INF, -1.), Tuple::point(1., INF, 1.)};
    case Kind::Group: {
      Bounds out{Tuple::point(0., 0., 0.), Tuple::point(0., 0., 0.)};
      for (const auto& c : s.shapes) {
        Bounds pb = bounds_new(c);
        const Matrix4& t = c.transform;
        Tuple q1 = t * Tuple::point(pb.min.x, pb.min.y, pb.min.z);
        Tuple q2 = t * Tuple::point(pb.min.x, pb.min.y, pb.max.z);
        Tuple q3 = t * Tuple::point(pb.min.x, pb.max.y, pb.min.z);
        Tuple q4 = t * Tuple::point(pb.min.x, pb.max.y, pb.max.z);
        Tuple q5 = t * Tuple::point(pb.max.x, pb.min.y, pb.min.z);
        Tuple q6 = t * Tuple::point(pb.max.x, pb.min.y, pb.max.z);
        Tuple q7 = t * Tuple::point(pb.max.x, pb.max.y, pb.min.z);
        Tuple q8 = t * pb.max;
        bounds_add(out, q1);
        bounds_add(out, q2);
        bounds_add(out, q3);
        bounds_add(out, q4);
        bounds_add(out, q5);
        bounds_add(out, q6);
        bounds_add(out, q7);
        bounds_add(out, q8);
      }
      return out;
    }
    case Kind::Triangle: {
      Bounds b{Tuple::point(0., 0., 0.), Tuple::point(0., 0., 0.)};
      bounds_add(b, s.p1);
      bounds_add(b, s.p2);
      bounds_add(b, s.p3);
      return b;
    }
  }
  return {};
}

// ---------------------------------------------------------------- intersection.rs (types)
struct Intersection {
  double t;
  const Shape* object;
};
using Intersections = std::vector<Intersection>;

inline void sort_by_t(Intersections& xs) {  // world.rs:51 / shape.rs:431-432 (stable, partial_cmp)
  std::stable_sort(xs.begin(), xs.end(), [](const Intersection& a, const Intersection& b) { return a.t < b.t; });
}

// shape.rs:587-606
inline void check_axis(double mn, double mx, double origin, double direction, double& tmin, double& tmax) {
  double tmin_numerator = mn - origin;
  double tmax_numerator = mx - origin;
  if (std::fabs(direction) >= EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * std::numeric_limits<double>::infinity();
    tmax = tmax_numerator * std::numeric_limits<double>::infinity();
  }
  if (tmin > tmax) std::swap(tmin, tmax);
}
// shape.rs:579-585 (radius^2 compared with |y|: reference quirk kept)
inline bool check_cap(const Ray& ray, double t) {
  double x = ray.origin.x + t * ray.direction.x;
  double y = ray.origin.y + t * ray.direction.y;
  double z = ray.origin.z + t * ray.direction.z;
  return x * x + z * z <= std::fabs(y);
}
// shape.rs:537-573
inline void intersect_caps(const Shape& s, Intersections& out, const Ray& local_ray) {
  if (!s.capped) return;
  if (is_almost_equal(local_ray.direction.y, 0.0)) return;
  double t = (s.minimum - local_ray.origin.y) / local_ray.direction.y;
  if (check_cap(local_ray, t)) out.push_back({t, &s});
  t = (s.maximum - local_ray.origin.y) / local_ray.direction.y;
  if (check_cap(local_ray, t)) out.push_back({t, &s});
}

struct Ctx {
  bool faithful = false;     // ORC_MODE_FAITHFUL: per-call inverse() and per-ray Bounds::new
  bool invert_failed = false;
};

// shape.rs:248-463
inline Intersections intersect(const Shape& self, const Ray& world_ray, Ctx& ctx) {
  Ray local_ray;
  if (ctx.faithful) {
    auto inv = inverse(self.transform);  // shape.rs:249-253: recomputed every call
    if (!inv) {
      ctx.invert_failed = true;
      return {};
    }
    local_ray = world_ray.transform(*inv);
  } else {
    local_ray = world_ray.transform(self.transform_inverse);  // same doubles (shape.rs:215)
  }
  Intersections result;
  switch (self.kind) {
    case Kind::Sphere: {  // :258-273
      Tuple sphere_to_ray = local_ray.origin - Tuple::point(0., 0., 0.);
      double a = local_ray.direction.dot(local_ray.direction);
      double b = 2. * local_ray.direction.dot(sphere_to_ray);
      double c = sphere_to_ray.dot(sphere_to_ray) - 1.;
      double discriminant = b * b - 4. * a * c;
      if (discriminant >= 0.) {
        double sq = std::sqrt(discriminant);
        result.push_back({(-b - sq) / (2. * a), &self});
        result.push_back({(-b + sq) / (2. * a), &self});
      }
      break;
    }
    case Kind::Plane: {  // :274-282
      if (std::fabs(local_ray.direction.y) >= EPSILON)
        result.push_back({-local_ray.origin.y / local_ray.direction.y, &self});
      break;
    }
    case Kind::Cube: {  // :283-319
      double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
      check_axis(-1.0, 1.0, local_ray.origin.x, local_ray.direction.x, xtmin, xtmax);
      check_axis(-1.0, 1.0, local_ray.origin.y, local_ray.direction.y, ytmin, ytmax);
      check_axis(-1.0, 1.0, local_ray.origin.z, local_ray.direction.z, ztmin, ztmax);
      double tmin = std::fmax(std::fmax(xtmin, ytmin), ztmin);
      double tmax = std::fmin(std::fmin(xtmax, ytmax), ztmax);
      if (tmax >= tmin) {
        result.push_back({tmin, &self});
        result.push_back({tmax, &self});
      }
      break;
    }
    case Kind::Cylinder: {  // :320-355
      const Tuple& o = local_ray.origin;
      const Tuple& d = local_ray.direction;
      double a = d.x * d.x + d.z * d.z;
      if (!is_almost_equal(a, 0.0)) {
        double b = 2.0 * o.x * d.x + 2.0 * o.z * d.z;
        double c = o.x * o.x + o.z * o.z - 1.0;
        double discriminant = b * b - 4.0 * a * c;
        if (discriminant >= 0.0) {
          double sq = std::sqrt(discriminant);
          double t0 = (-b - sq) / (2. * a);
          double t1 = (-b + sq) / (2. * a);
          if (t0 > t1) std::swap(t0, t1);
          double y0 = o.y + t0 * d.y;
          if (self.minimum < y0 && y0 < self.maximum) result.push_back({t0, &self});
          double y1 = o.y + t1 * d.y;
          if (self.minimum < y1 && y1 < self.maximum) result.push_back({t1, &self});
        }
      }
      intersect_caps(self, result, local_ray);
      break;
    }
    case Kind::Cone: {  // :356-398
      const Tuple& o = local_ray.origin;
      const Tuple& d = local_ray.direction;
      double a = d.x * d.x - d.y * d.y + d.z * d.z;
      double b = 2.0 * o.x * d.x - 2.0 * o.y * d.y + 2.0 * o.z * d.z;
      double c = o.x * o.x - o.y * o.y + o.z * o.z;
      if (is_almost_equal(a, 0.0)) {
        if (!is_almost_equal(b, 0.0)) {
          double t = -c / (2.0 * b);
          result.push_back({t, &self});
        }
      } else {
        double discriminant = b * b - 4.0 * a * c;
        if (discriminant >= 0.0) {
          double sq = std::sqrt(discriminant);
          double t0 = (-b - sq) / (2. * a);
          double t1 = (-b + sq) / (2. * a);
          if (t0 > t1) std::swap(t0, t1);
          double y0 = o.y + t0 * d.y;
          if (self.minimum < y0 && y0 < self.maximum) result.push_back({t0, &self});
          double y1 = o.y + t1 * d.y;
          if (self.minimum < y1 && y1 < self.maximum) result.push_back({t1, &self});
        }
      }
      intersect_caps(self, result, local_ray);
      break;
    }
    case Kind::Group: {  // :399-436
      Bounds bounds;
      if (ctx.faithful) {
        bounds = bounds_new(self);  // rebuilt per ray, as the reference does
      } else {
        if (!self.bounds_cached) {
          self.bounds_cache = bounds_new(self);
          self.bounds_cached = true;
        }
        bounds = self.bounds_cache;
      }
      double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
      check_axis(bounds.min.x, bounds.max.x, local_ray.origin.x, local_ray.direction.x, xtmin, xtmax);
      check_axis(bounds.min.y, bounds.max.y, local_ray.origin.y, local_ray.direction.y, ytmin, ytmax);
      check_axis(bounds.min.z, bounds.max.z, local_ray.origin.z, local_ray.direction.z, ztmin, ztmax);
      double tmin = std::fmax(std::fmax(xtmin, ytmin), ztmin);
      double tmax = std::fmin(std::fmin(xtmax, ytmax), ztmax);
      if (tmax > tmin) {
        Intersections shape_results;
        for (const auto& c : self.shapes) {
          Intersections xs = intersect(c, world_ray, ctx);  // world ray: transforms live on the leaves
          shape_results.insert(shape_results.end(), xs.begin(), xs.end());
        }
        sort_by_t(shape_results);
        result.insert(result.end(), shape_results.begin(), shape_results.end());
      }
      break;
    }
    case Kind::Triangle: {  // :438-459 Möller–Trumbore
      Tuple dir_cross_e2 = local_ray.direction.cross(self.e2);
      double det = self.e1.dot(dir_cross_e2);
      if (!(std::fabs(det) < EPSILON)) {
        double f = 1.0 / det;
        Tuple p1_to_origin = local_ray.origin - self.p1;
        double u = f * p1_to_origin.dot(dir_cross_e2);
        if (!(u < 0.0 || u > 1.0)) {
          Tuple origin_cross_e1 = p1_to_origin.cross(self.e1);
          double v = f * local_ray.direction.dot(origin_cross_e1);
          if (!(v < 0.0 || (u + v) > 1.0)) {
            double t = f * self.e2.dot(origin_cross_e1);
            result.push_back({t, &self});
          }
        }
      }
      break;
    }
  }
  return result;
}

// shape.rs:608-635
inline Tuple world_to_object(const Shape& s, const Tuple& p) { return s.transform_inverse * p; }
inline Tuple normal_to_world(const Shape& s, const Tuple& n) {
  Tuple r = s.transform_inverse_transpose * n;
  r.w = 0.0;
  return r.normalize();
}
// shape.rs:466-519
inline Tuple normal_at(const Shape& s, const Tuple& world_point) {
  Tuple lp = world_to_object(s, world_point);
  Tuple ln{};
  switch (s.kind) {
    case Kind::Sphere:
      ln = lp - Tuple::point(0.0, 0.0, 0.0);
      break;
    case Kind::Plane:
      ln = Tuple::vector(0.0, 1.0, 0.0);
      break;
    case Kind::Cube: {
      double xa = std::fabs(lp.x), ya = std::fabs(lp.y), za = std::fabs(lp.z);
      double maxc = std::fmax(std::fmax(xa, ya), za);
      if (maxc == xa)
        ln = Tuple::vector(lp.x, 0.0, 0.0);
      else if (maxc == ya)
        ln = Tuple::vector(0.0, lp.y, 0.0);
      else
        ln = Tuple::vector(0.0, 0.0, lp.z);
      break;
    }
    case Kind::Cylinder: {
      double dist = lp.x * lp.x + lp.z * lp.z;
      if (dist < 1.0 && lp.y >= s.maximum - EPSILON)
        ln = Tuple::vector(0.0, 1.0, 0.0);
      else if (dist < 1.0 && lp.y <= s.minimum + EPSILON)
        ln = Tuple::vector(0.0, -1.0, 0.0);
      else
        ln = Tuple::vector(lp.x, 0.0, lp.z);
      break;
    }
    case Kind::Cone: {
      double y = std::sqrt(lp.x * lp.x + lp.z * lp.z);
      if (lp.y > 0.0) y = -y;
      ln = Tuple::vector(lp.x, y, lp.z);
      break;
    }
    case Kind::Group:
      ln = Tuple::zero();  // unreachable!() in the reference
      break;
    case Kind::Triangle:
      ln = s.normal;
      break;
  }
  Tuple wn = normal_to_world(s, ln);
  wn.w = 0.;
  return wn.normalize();
}

// pattern.rs:98-103
inline Color pattern_color_at_shape(const Pattern& p, const Shape& object, const Tuple& world_point) {
  Tuple object_point = object.transform_inverse * world_point;
  Tuple pattern_point = p.transform_inverse * object_point;
  return p.color_at(pattern_point);
}

// material.rs:32-75
inline Color lighting(const Material& m, const Light& light, const Shape& object, const Tuple& point,
                      const Tuple& eyev, const Tuple& normalv, bool in_shadow) {
  Color color = m.pattern ? pattern_color_at_shape(*m.pattern, object, point) : m.color;
  Color effective_color = color * light.intensity;
  Tuple lightv = (light.position - point).normalize();
  Color ambient = effective_color * m.ambient;
  Color diffuse = BLACK, specular = BLACK;
  if (!in_shadow) {
    double light_dot_normal = lightv.dot(normalv);
    if (light_dot_normal >= 0.) {
      diffuse = effective_color * m.diffuse * light_dot_normal;
      Tuple reflectv = reflect(-lightv, normalv);
      double reflect_dot_eye = reflectv.dot(eyev);
      if (reflect_dot_eye > 0.) {
        double factor = std::pow(reflect_dot_eye, m.shininess);
        specular = light.intensity * m.specular * factor;
      }
    }
  }
  return ambient + diffuse + specular;
}

// ---------------------------------------------------------------- intersection.rs
struct Computations {  // :88-100
  double t;
  const Shape* object;
  Tuple point, over_point, under_point, eyev;
  bool inside;
  Tuple normalv, reflectv;
  double n1, n2;
};

// derived PartialEq on Intersection (:6): t == t && *object == *object (Shape value equality)
inline bool intersection_eq(const Intersection& a, const Intersection& b) {
  return a.t == b.t && shape_eq(*a.object, *b.object);
}

// intersection.rs:79-83 — first minimum over t >= 0
inline const Intersection* hit(const Intersections& xs) {
  const Intersection* best = nullptr;
  for (const auto& x : xs) {
    if (!(x.t >= 0.)) continue;
    if (!best || x.t < best->t) best = &x;
  }
  return best;
}

// intersection.rs:17-77
inline Computations prepare_computations(const Intersection& self, const Ray& ray, const Intersections& xs) {
  Tuple point = ray.position(self.t);
  Tuple eyev = -ray.direction;
  Tuple normalv = normal_at(*self.object, point);
  bool inside = normalv.dot(eyev) < 0.0;
  if (inside) normalv = -normalv;
  Tuple reflectv = reflect(ray.direction, normalv);

  std::vector<const Shape*> containers;
  double n1 = 1.0, n2 = 1.0;
  for (const auto& i : xs) {
    bool is_self = intersection_eq(i, self);
    if (is_self) {
      if (!containers.empty()) n1 = containers.back()->material.refractive_index;
    }
    auto pos = std::find_if(containers.begin(), containers.end(),
                            [&](const Shape* o) { return shape_eq(*o, *i.object); });
    if (pos != containers.end())
      containers.erase(pos);
    else
      containers.push_back(i.object);
    if (is_self) {
      if (!containers.empty()) n2 = containers.back()->material.refractive_index;
      break;
    }
  }
  Computations c;
  c.t = self.t;
  c.object = self.object;
  c.point = point;
  c.over_point = point + normalv * EPSILON;
  c.under_point = point - normalv * EPSILON;
  c.eyev = eyev;
  c.inside = inside;
  c.normalv = normalv;
  c.reflectv = reflectv;
  c.n1 = n1;
  c.n2 = n2;
  return c;
}

inline double powi5(double x) {
  double x2 = x * x;
  double x4 = x2 * x2;
  return x * x4;
}
// intersection.rs:107-128
inline double schlick(const Computations& c) {
  double cosv = c.eyev.dot(c.normalv);
  if (c.n1 > c.n2) {
    double n = c.n1 / c.n2;
    double sin2_t = (n * n) * (1.0 - cosv * cosv);
    if (sin2_t > 1.0) return 1.0;
    double cos_t = std::sqrt(1.0 - sin2_t);
    cosv = cos_t;
  }
  double q = (c.n1 - c.n2) / (c.n1 + c.n2);
  double r0 = q * q;
  return r0 + (1.0 - r0) * powi5(1.0 - cosv);
}

// ---------------------------------------------------------------- world.rs
struct World {
  std::vector<Shape> objects;
  Light light{Tuple::point(0, 0, 0), WHITE};
};
struct TraceCtx : Ctx {
  int64_t radiance_rays = 0, shadow_rays = 0;  // one World::intersect call == one ray
  bool depth_underflow = false;                // world.rs:68 `remaining - 1` with remaining == 0
  int frames = 0;
  bool runaway = false;
};
constexpr int MAX_FRAMES = 256;  // guard for the release-mode wrap (unbounded recursion in the reference)

inline World default_world() {  // world.rs:26-41
  World w;
  w.light = {Tuple::point(-10.0, 10.0, -10.0), WHITE};
  Shape s1 = sphere();
  s1.material.color = {0.8, 1.0, 0.6};
  s1.material.diffuse = 0.7;
  s1.material.specular = 0.2;
  Shape s2 = sphere();
  set_transform(s2, scaling(0.5, 0.5, 0.5));
  w.objects = {s1, s2};
  return w;
}

// world.rs:43-54
inline Intersections world_intersect(const World& w, const Ray& ray, TraceCtx& ctx) {
  Intersections result;
  for (const auto& object : w.objects) {
    Intersections xs = intersect(object, ray, ctx);
    result.insert(result.end(), xs.begin(), xs.end());
  }
  sort_by_t(result);
  return result;
}

inline Color internal_color_at(const World& w, const Ray& ray, uint64_t remaining, TraceCtx& ctx);

// world.rs:100-114
inline bool is_shadowed(const World& w, const Tuple& point, TraceCtx& ctx) {
  Tuple vector = w.light.position - point;
  double distance = vector.magnitude();
  Tuple direction = vector.normalize();
  Ray ray{point, direction};
  ctx.shadow_rays++;
  Intersections xs = world_intersect(w, ray, ctx);
  const Intersection* h = hit(xs);
  return h ? (h->t < distance) : false;
}
// world.rs:116-129
inline Color reflected_color(const World& w, const Computations& comps, uint64_t remaining, TraceCtx& ctx) {
  if (remaining < 1) return BLACK;
  const Shape* object = comps.object;
  if (object->material.reflective == 0.0) return BLACK;
  Ray reflect_ray{comps.over_point, comps.reflectv};
  Color color = internal_color_at(w, reflect_ray, remaining - 1, ctx);
  return color * object->material.reflective;
}
// world.rs:131-163
inline Color refracted_color(const World& w, const Computations& comps, uint64_t remaining, TraceCtx& ctx) {
  if (remaining == 0) return BLACK;
  const Shape* object = comps.object;
  if (object->material.transparency == 0.0) return BLACK;
  double n_ratio = comps.n1 / comps.n2;
  double cos_i = comps.eyev.dot(comps.normalv);
  double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
  if (sin2_t > 1.0) return BLACK;
  double cos_t = std::sqrt(1.0 - sin2_t);
  Tuple direction = comps.normalv * (n_ratio * cos_i - cos_t) - comps.eyev * n_ratio;
  Ray refract_ray{comps.under_point, direction};
  Color color = internal_color_at(w, refract_ray, remaining - 1, ctx) * object->material.transparency;
  return color;
}
// world.rs:56-78
inline Color shade_hit(const World& w, const Computations& comps, uint64_t remaining, TraceCtx& ctx) {
  const Shape* object = comps.object;
  const Material& material = object->material;
  Color surface = lighting(material, w.light, *object, comps.point, comps.eyev, comps.normalv,
                           is_shadowed(w, comps.over_point, ctx));
  // debug build: "attempt to subtract with overflow".  In release the wrap is harmless unless the
  // material then actually recurses (reflected_color / refracted_color get remaining = 2^64-1).
  if (remaining == 0 && (material.reflective != 0.0 || material.transparency != 0.0)) ctx.depth_underflow = true;
  uint64_t next = remaining - 1;                   // release build: wraps
  Color reflected = reflected_color(w, comps, next, ctx);
  Color refracted = refracted_color(w, comps, next, ctx);
  if (material.reflective > 0.0 && material.transparency > 0.0) {
    double reflectance = schlick(comps);
    return surface + reflected * reflectance + refracted * (1.0 - reflectance);
  }
  return surface + reflected + refracted;
}
// world.rs:84-98
inline Color internal_color_at(const World& w, const Ray& ray, uint64_t remaining, TraceCtx& ctx) {
  if (remaining < 1) return BLACK;
  if (++ctx.frames > MAX_FRAMES) {
    ctx.runaway = true;
    --ctx.frames;
    return BLACK;
  }
  ctx.radiance_rays++;
  Intersections xs = world_intersect(w, ray, ctx);
  const Intersection* h = hit(xs);
  Color out = BLACK;
  if (h) out = shade_hit(w, prepare_computations(*h, ray, xs), remaining - 1, ctx);
  --ctx.frames;
  return out;
}

// ---------------------------------------------------------------- camera.rs
struct Camera {
  int64_t hsize = 0, vsize = 0;
  double field_of_view = 0;
  Matrix4 transform = Matrix4::identity(), transform_inverse = Matrix4::identity();
  double pixel_size = 0, half_width = 0, half_height = 0;
};
inline Camera camera_new(int64_t hsize, int64_t vsize, double fov) {  // :16-41
  Camera c;
  c.hsize = hsize, c.vsize = vsize, c.field_of_view = fov;
  double half_view = std::tan(fov / 2.0);
  double aspect = (double)hsize / (double)vsize;
  if (aspect >= 1.0) {
    c.half_width = half_view;
    c.half_height = half_view / aspect;
  } else {
    c.half_width = half_view * aspect;
    c.half_height = half_view;
  }
  c.pixel_size = (c.half_width * 2.0) / (double)hsize;
  return c;
}
inline bool camera_set_transform(Camera& c, const Matrix4& t) {  // :43-46
  c.transform = t;
  auto inv = inverse(t);
  if (!inv) return false;
  c.transform_inverse = *inv;
  return true;
}
inline Ray ray_for_pixel(const Camera& c, int64_t px, int64_t py) {  // :48-65
  double xoffset = ((double)px + 0.5) * c.pixel_size;
  double yoffset = ((double)py + 0.5) * c.pixel_size;
  double world_x = c.half_width - xoffset;
  double world_y = c.half_height - yoffset;
  Tuple pixel = c.transform_inverse * Tuple::point(world_x, world_y, -1.0);
  Tuple origin = c.transform_inverse * Tuple::point(0.0, 0.0, 0.0);
  Tuple direction = (pixel - origin).normalize();
  return {origin, direction};
}

// ---------------------------------------------------------------- obj_file.rs
struct ObjResult {
  Shape group;            // obj_to_group(): group[default_group, named groups...]
  int64_t ignored_lines = 0;
  size_t n_vertices = 0;
  std::string error;      // non-empty <=> the reference would have panicked
};
ObjResult parse_obj(const char* text, size_t len);

// ---------------------------------------------------------------- canvas.rs
std::string canvas_to_ppm(int64_t width, int64_t height, const double* rgb);

}  // namespace orc
