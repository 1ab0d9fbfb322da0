// rtc_host.cpp — scene model, OBJ loader, Canvas/PPM and the Camera::render / World::color_at entry points.
// See rtc_host.hpp.  Compiled with -ffp-contract=off: flatten-time arithmetic (matrix inverses, group bounds,
// triangle edges/normals, camera derivations) follows the reference's operation order so that the numbers
// handed to the device are the doubles the reference itself would hold.
#include "rtc_host.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>

namespace rtc::host {

static bool almost(double a, double b) { return std::fabs(a - b) < EPSILON; }  // utils.rs:4-6

// ------------------------------------------------------------------ tuple.rs
double Tuple::magnitude() const { return std::sqrt(x * x + y * y + z * z + w * w); }  // :43-48
Tuple Tuple::normalize() const {                                                       // :50-66
  double mag = magnitude();
  if (mag == 0.0) return Tuple{};
  return {x / mag, y / mag, z / mag, w / mag};
}
double Tuple::dot(const Tuple& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }  // :68-73
Tuple Tuple::cross(const Tuple& o) const {                                                  // :75-83
  return vector(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x);
}
Tuple operator+(const Tuple& a, const Tuple& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
Tuple operator-(const Tuple& a, const Tuple& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
Tuple operator-(const Tuple& a) { return {-a.x, -a.y, -a.z, -a.w}; }
Tuple operator*(const Tuple& a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
bool approx_eq(const Tuple& a, const Tuple& b) {  // :93-100
  return almost(a.x, b.x) && almost(a.y, b.y) && almost(a.z, b.z) && almost(a.w, b.w);
}
bool approx_eq(const Color& a, const Color& b) {  // color.rs:47-53
  return almost(a.red, b.red) && almost(a.green, b.green) && almost(a.blue, b.blue);
}

// ------------------------------------------------------------------ matrix.rs
Matrix4 Matrix4::identity() {
  Matrix4 r;
  for (int i = 0; i < 4; ++i) r.at(i, i) = 1.;
  return r;
}
Matrix4 Matrix4::transpose() const {
  Matrix4 r;
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) r.at(col, row) = at(row, col);
  return r;
}
namespace {
// Laplace expansion along the first row, the way Matrix::determinant / cofactor / minor / submatrixN do it
// (matrix.rs:41-137): `result = 0.; result += m[0][c] * cofactor(0, c)`, 2x2 base case a*d - b*c.
double det_n(const double* a, int n) {  // a: n x n row-major
  if (n == 2) return a[0] * a[3] - a[1] * a[2];
  double result = 0.;
  double sub[9];
  for (int column = 0; column < n; ++column) {
    int k = 0;
    for (int r = 1; r < n; ++r)
      for (int c = 0; c < n; ++c)
        if (c != column) sub[k++] = a[r * n + c];
    double minor = det_n(sub, n - 1);
    result += a[column] * ((column % 2 == 0) ? minor : -minor);
  }
  return result;
}
double cofactor4(const Matrix4& m, int row, int col) {
  double sub[9];
  int k = 0;
  for (int r = 0; r < 4; ++r) {
    if (r == row) continue;
    for (int c = 0; c < 4; ++c)
      if (c != col) sub[k++] = m.at(r, c);
  }
  double minor = det_n(sub, 3);
  return ((row + col) % 2 == 0) ? minor : -minor;
}
}  // namespace
double Matrix4::determinant() const { return det_n(m.data(), 4); }
std::optional<Matrix4> Matrix4::inverse() const {  // :138-157
  const double det = determinant();
  if (almost(det, 0.)) return std::nullopt;
  Matrix4 r;
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) r.at(col, row) = cofactor4(*this, row, col) / det;
  return r;
}
Matrix4 operator*(const Matrix4& a, const Matrix4& b) {  // :187-205
  Matrix4 r;
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 4; ++col) {
      double val = 0.;
      for (int n = 0; n < 4; ++n) val += a.at(row, n) * b.at(n, col);
      r.at(row, col) = val;
    }
  return r;
}
Tuple operator*(const Matrix4& a, const Tuple& t) {  // :207-227
  Tuple r;
  r.x = a.at(0, 0) * t.x + a.at(0, 1) * t.y + a.at(0, 2) * t.z + a.at(0, 3) * t.w;
  r.y = a.at(1, 0) * t.x + a.at(1, 1) * t.y + a.at(1, 2) * t.z + a.at(1, 3) * t.w;
  r.z = a.at(2, 0) * t.x + a.at(2, 1) * t.y + a.at(2, 2) * t.z + a.at(2, 3) * t.w;
  r.w = a.at(3, 0) * t.x + a.at(3, 1) * t.y + a.at(3, 2) * t.z + a.at(3, 3) * t.w;
  return r;
}
bool approx_eq(const Matrix4& a, const Matrix4& b) {  // :174-185
  for (size_t i = 0; i < 16; ++i)
    if (!almost(a.m[i], b.m[i])) return false;
  return true;
}

// ------------------------------------------------------------------ transformations.rs
Matrix4 translation(double x, double y, double z) {
  Matrix4 r = Matrix4::identity();
  r.at(0, 3) = x, r.at(1, 3) = y, r.at(2, 3) = z;
  return r;
}
Matrix4 scaling(double x, double y, double z) {
  Matrix4 r = Matrix4::identity();
  r.at(0, 0) = x, r.at(1, 1) = y, r.at(2, 2) = z;
  return r;
}
Matrix4 rotation_x(double rad) {
  Matrix4 r = Matrix4::identity();
  const double c = std::cos(rad), s = std::sin(rad);
  r.at(1, 1) = c, r.at(2, 2) = c, r.at(1, 2) = -s, r.at(2, 1) = s;
  return r;
}
Matrix4 rotation_y(double rad) {
  Matrix4 r = Matrix4::identity();
  const double c = std::cos(rad), s = std::sin(rad);
  r.at(0, 0) = c, r.at(2, 2) = c, r.at(0, 2) = s, r.at(2, 0) = -s;
  return r;
}
Matrix4 rotation_z(double rad) {
  Matrix4 r = Matrix4::identity();
  const double c = std::cos(rad), s = std::sin(rad);
  r.at(0, 0) = c, r.at(1, 1) = c, r.at(0, 1) = -s, r.at(1, 0) = s;
  return r;
}
Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
  Matrix4 r = Matrix4::identity();
  r.at(0, 1) = xy, r.at(0, 2) = xz, r.at(1, 0) = yx, r.at(1, 2) = yz, r.at(2, 0) = zx, r.at(2, 1) = zy;
  return r;
}
Matrix4 view_transform(const Tuple& from, const Tuple& to, const Tuple& up) {  // :80-93
  const Tuple forward = (to - from).normalize();
  const Tuple left = forward.cross(up.normalize());
  const Tuple true_up = left.cross(forward);
  Matrix4 o = Matrix4::identity();
  o.at(0, 0) = left.x, o.at(0, 1) = left.y, o.at(0, 2) = left.z;
  o.at(1, 0) = true_up.x, o.at(1, 1) = true_up.y, o.at(1, 2) = true_up.z;
  o.at(2, 0) = -forward.x, o.at(2, 1) = -forward.y, o.at(2, 2) = -forward.z;
  return o * translation(-from.x, -from.y, -from.z);
}

// ------------------------------------------------------------------ pattern.rs / material.rs
void Pattern::set_transform(const Matrix4& t) {
  auto inv = t.inverse();
  if (!inv) throw Error(HOST_ERR_NOT_INVERTIBLE, "pattern transform should be invertible");
  transform_ = t;
  transform_inverse_ = *inv;
}
bool Pattern::operator==(const Pattern& o) const {
  if (!approx_eq(transform_, o.transform_) || !approx_eq(transform_inverse_, o.transform_inverse_)) return false;
  if (kind_ != o.kind_) return false;
  if (kind_ == RTC_PATTERN_TEST) return true;
  return approx_eq(a_, o.a_) && approx_eq(b_, o.b_);
}
bool Material::operator==(const Material& o) const {
  return approx_eq(color, o.color) && ambient == o.ambient && diffuse == o.diffuse && specular == o.specular &&
         shininess == o.shininess && reflective == o.reflective && pattern.has_value() == o.pattern.has_value() &&
         (!pattern || *pattern == *o.pattern) && transparency == o.transparency &&
         refractive_index == o.refractive_index;
}

// ------------------------------------------------------------------ shape.rs
Shape Shape::glass_sphere() {
  Shape s = sphere();
  s.material_.transparency = 1.0;
  s.material_.refractive_index = 1.5;
  return s;
}
Shape Shape::infinite_cylinder() {
  constexpr double inf = std::numeric_limits<double>::infinity();
  return cylinder(-inf, inf, false);
}
Shape Shape::cylinder(double mn, double mx, bool cap) {
  Shape s(ShapeKind::Cylinder);
  s.minimum = mn, s.maximum = mx, s.capped = cap;
  return s;
}
Shape Shape::infinite_cone() {
  constexpr double inf = std::numeric_limits<double>::infinity();
  return cone(-inf, inf, false);
}
Shape Shape::cone(double mn, double mx, bool cap) {
  Shape s(ShapeKind::Cone);
  s.minimum = mn, s.maximum = mx, s.capped = cap;
  return s;
}
Shape Shape::triangle(const Tuple& a, const Tuple& b, const Tuple& c) {  // :171-193
  if (!(a.w == 1.0 && b.w == 1.0 && c.w == 1.0)) throw Error(HOST_ERR_ARG, "triangle corners must be points");
  Shape s(ShapeKind::Triangle);
  s.p1 = a, s.p2 = b, s.p3 = c;
  s.e1 = b - a;
  s.e2 = c - a;
  s.normal = s.e2.cross(s.e1).normalize();
  return s;
}
void Shape::set_transform(const Matrix4& t) {
  if (transformed_) throw Error(HOST_ERR_TRANSFORM_TWICE, "Can't call set_transform more than once.");
  transformed_ = true;
  set_transform_internal(t);
}
void Shape::set_transform_internal(const Matrix4& t) {  // :207-218
  if (kind_ == ShapeKind::Group) {
    for (auto& c : children_) c.set_transform_internal(t);
    return;
  }
  transform_ = t * transform_;
  auto inv = transform_.inverse();
  if (!inv) throw Error(HOST_ERR_NOT_INVERTIBLE, "should be invertible");
  transform_inverse_ = *inv;
}
void Shape::set_material(const Material& m) {
  if (kind_ == ShapeKind::Group) {
    for (auto& c : children_) c.set_material(m);
  } else {
    material_ = m;
  }
}
void Shape::push_shape(Shape s) {
  if (kind_ != ShapeKind::Group) throw Error(HOST_ERR_NOT_GROUP, "push_shape was called on something that isn't a group");
  children_.push_back(std::move(s));
}
size_t Shape::leaf_count() const {
  if (kind_ != ShapeKind::Group) return 1;
  size_t n = 0;
  for (const auto& c : children_) n += c.leaf_count();
  return n;
}
bool Shape::kind_eq(const Shape& o) const {  // derived PartialEq on ShapeKind
  if (kind_ != o.kind_) return false;
  switch (kind_) {
    case ShapeKind::Cylinder:
    case ShapeKind::Cone:
      return minimum == o.minimum && maximum == o.maximum && capped == o.capped;
    case ShapeKind::Group:
      return children_ == o.children_;
    case ShapeKind::Triangle:
      return approx_eq(p1, o.p1) && approx_eq(p2, o.p2) && approx_eq(p3, o.p3) && approx_eq(e1, o.e1) &&
             approx_eq(e2, o.e2) && approx_eq(normal, o.normal);
    default:
      return true;
  }
}
bool Shape::operator==(const Shape& o) const {
  return kind_eq(o) && approx_eq(transform_, o.transform_) && material_ == o.material_;
}

// ------------------------------------------------------------------ bounds.rs
static void grow(Bounds& b, const Tuple& p) {  // Bounds::add :142-151
  b.min.x = std::fmin(b.min.x, p.x), b.min.y = std::fmin(b.min.y, p.y), b.min.z = std::fmin(b.min.z, p.z);
  b.max.x = std::fmax(b.max.x, p.x), b.max.y = std::fmax(b.max.y, p.y), b.max.z = std::fmax(b.max.z, p.z);
}
Bounds Shape::bounds() const {
  constexpr double inf = std::numeric_limits<double>::infinity();
  switch (kind_) {
    case ShapeKind::Sphere:
    case ShapeKind::Cube:
      return {Tuple::point(-1, -1, -1), Tuple::point(1, 1, 1)};
    case ShapeKind::Plane:
      return {Tuple::point(-1, -1, 0), Tuple::point(1, 1, 0)};  // reference quirk: an xy square
    case ShapeKind::Cylinder:
    case ShapeKind::Cone:
      return capped ? Bounds{Tuple::point(-1, minimum, -1), Tuple::point(1, maximum, 1)}
                    : Bounds{Tuple::point(-1, -inf, -1), Tuple::point(1, inf, 1)};
    case ShapeKind::Triangle: {
      Bounds b{Tuple::point(0, 0, 0), Tuple::point(0, 0, 0)};  // seeded with the origin (reference quirk)
      grow(b, p1), grow(b, p2), grow(b, p3);
      return b;
    }
    case ShapeKind::Group: {
      Bounds out{Tuple::point(0, 0, 0), Tuple::point(0, 0, 0)};
      for (const auto& c : children_) {
        const Bounds pb = c.bounds();
        const Matrix4& t = c.transform_;
        // the eight corners in the reference's order (bounds.rs:62-120)
        const double xs[2] = {pb.min.x, pb.max.x}, ys[2] = {pb.min.y, pb.max.y}, zs[2] = {pb.min.z, pb.max.z};
        for (int ix = 0; ix < 2; ++ix)
          for (int iy = 0; iy < 2; ++iy)
            for (int iz = 0; iz < 2; ++iz) grow(out, t * Tuple::point(xs[ix], ys[iy], zs[iz]));
      }
      return out;
    }
  }
  return {};
}

// ------------------------------------------------------------------ canvas.rs:28-63
std::string Canvas::to_ppm() const {
  // canvas.rs:28-63: "P3", "W H", "255", then per row the channel values round(clamp(c, 0, 1) * 255) separated by single
  // spaces, a line break wherever the next value would take the line past 70 characters, and a newline after every row.
  // One pass into a preallocated buffer, decimal digits from a table (a 1080p canvas is 6.2 M values).
  static const struct Digits {
    char text[256][3];
    unsigned char len[256];
    Digits() {
      for (int v = 0; v < 256; ++v) {
        const std::string t = std::to_string(v);
        len[v] = (unsigned char)t.size();
        for (size_t i = 0; i < t.size(); ++i) text[v][i] = t[i];
      }
    }
  } digits;
  auto chan = [](double c) -> int {
    if (!(c > 0.)) return 0;  // clamp(0, 1) keeps a NaN, `as u8`-style conversion of the rounded NaN gives 0; -0.0 and negatives: 0
    const double x = (c > 1. ? 1. : c) * 255.;  // in (0, 255]
    const int r = (int)x;                        // round(): half away from zero, evaluated exactly (x - r is exact)
    return r + ((x - (double)r) >= 0.5 ? 1 : 0);
  };
  std::string out = "P3\n" + std::to_string(width_) + " " + std::to_string(height_) + "\n255\n";
  const size_t head = out.size();
  out.resize(head + height_ * (width_ * 3 * 4 + 1));  // at most "255" + one separator per value, one newline per row
  char* p = &out[head];
  for (size_t y = 0; y < height_; ++y) {
    size_t len = 0;
    for (size_t x = 0; x < width_; ++x) {
      const Color c = get_pixel(x, y);
      const int v3[3] = {chan(c.red), chan(c.green), chan(c.blue)};
      for (int k = 0; k < 3; ++k) {
        const size_t n = digits.len[v3[k]];
        if (len + n + 1 > 70) {
          *p++ = '\n';
          len = 0;
        }
        if (len > 0) {
          *p++ = ' ';
          ++len;
        }
        for (size_t i = 0; i < n; ++i) *p++ = digits.text[v3[k]][i];
        len += n;
      }
    }
    *p++ = '\n';
  }
  out.resize((size_t)(p - out.data()));
  return out;
}

// ------------------------------------------------------------------ camera.rs
Camera::Camera(size_t hsize, size_t vsize, double fov) : hsize_(hsize), vsize_(vsize), field_of_view_(fov) {
  const double half_view = std::tan(fov / 2.0);
  const double aspect = (double)hsize / (double)vsize;
  if (aspect >= 1.0) {
    half_width_ = half_view;
    half_height_ = half_view / aspect;
  } else {
    half_width_ = half_view * aspect;
    half_height_ = half_view;
  }
  pixel_size_ = (half_width_ * 2.0) / (double)hsize;
}
void Camera::set_transform(const Matrix4& t) {
  auto inv = t.inverse();
  if (!inv) throw Error(HOST_ERR_NOT_INVERTIBLE, "should be invertible");
  transform_ = t;
  transform_inverse_ = *inv;
}
Ray Camera::ray_for_pixel(size_t px, size_t py) const {
  const double xoffset = ((double)px + 0.5) * pixel_size_;
  const double yoffset = ((double)py + 0.5) * pixel_size_;
  const double world_x = half_width_ - xoffset;
  const double world_y = half_height_ - yoffset;
  const Tuple pixel = transform_inverse_ * Tuple::point(world_x, world_y, -1.0);
  const Tuple origin = transform_inverse_ * Tuple::point(0.0, 0.0, 0.0);
  return {origin, (pixel - origin).normalize()};
}
RtcCameraDesc Camera::desc() const {
  RtcCameraDesc d{};
  d.hsize = (uint32_t)hsize_, d.vsize = (uint32_t)vsize_;
  d.half_width = half_width_, d.half_height = half_height_, d.pixel_size = pixel_size_;
  std::memcpy(d.transform_inverse, transform_inverse_.m.data(), sizeof d.transform_inverse);
  return d;
}

// ------------------------------------------------------------------ device library
const DeviceApi& device_api() {
  static DeviceApi api{};
  static bool loaded = false;
  if (loaded) return api;
  std::string path;
  if (const char* env = std::getenv("RTC_HIP_LIB")) {
    path = env;
  } else {
    Dl_info info;
    if (dladdr((void*)&device_api, &info) && info.dli_fname) {
      std::string self = info.dli_fname;
      size_t slash = self.rfind('/');
      path = (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/../csrc/librtc_hip.so";
    } else {
      path = "librtc_hip.so";
    }
  }
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) throw Error(HOST_ERR_NO_DEVICE_LIB, std::string("cannot load the HIP renderer (") + path + "): " + dlerror());
  auto sym = [&](const char* n) {
    void* p = dlsym(h, n);
    if (!p) throw Error(HOST_ERR_NO_DEVICE_LIB, std::string("missing symbol ") + n + " in " + path);
    return p;
  };
  api.render = (decltype(api.render))sym("rtc_render");
  api.render_multi = (decltype(api.render_multi))sym("rtc_render_multi");
  api.scene_create = (decltype(api.scene_create))sym("rtc_scene_create");
  api.scene_destroy = (decltype(api.scene_destroy))sym("rtc_scene_destroy");
  api.color_at = (decltype(api.color_at))sym("rtc_color_at");
  api.last_error_string = (decltype(api.last_error_string))sym("rtc_last_error_string");
  api.device_count = (decltype(api.device_count))sym("rtc_device_count");
  loaded = true;
  return api;
}

static void check(const DeviceApi& api, RtcStatus st, const char* what) {
  if (st != RTC_OK) throw Error((int)st, std::string(what) + ": " + api.last_error_string());
}

// ------------------------------------------------------------------ world.rs
World World::default_world() {
  World w(Light{Tuple::point(-10.0, 10.0, -10.0), WHITE});
  Shape s1 = Shape::sphere();
  s1.get_material_mut().color = Color{0.8, 1.0, 0.6};
  s1.get_material_mut().diffuse = 0.7;
  s1.get_material_mut().specular = 0.2;
  Shape s2 = Shape::sphere();
  s2.set_transform(scaling(0.5, 0.5, 0.5));
  w.objects.push_back(std::move(s1));
  w.objects.push_back(std::move(s2));
  return w;
}

std::vector<Color> World::color_at(const std::vector<Ray>& rays, std::vector<float>* f32_out) const {
  const DeviceApi& api = device_api();
  FlatScene flat = flatten();
  RtcSceneDesc d = flat.desc();
  RtcScene* sc = nullptr;
  check(api, api.scene_create(&d, device, &sc), "rtc_scene_create");
  std::vector<double> o(rays.size() * 3), dir(rays.size() * 3), out(rays.size() * 3);
  std::vector<float> out32(rays.size() * 3);
  for (size_t i = 0; i < rays.size(); ++i) {
    o[3 * i] = rays[i].origin.x, o[3 * i + 1] = rays[i].origin.y, o[3 * i + 2] = rays[i].origin.z;
    dir[3 * i] = rays[i].direction.x, dir[3 * i + 1] = rays[i].direction.y, dir[3 * i + 2] = rays[i].direction.z;
  }
  RtcRenderOpts opts{};
  opts.recursion_limit = recursion_limit;
  opts.flags = render_flags;
  RtcStatus st = api.color_at(sc, rays.size(), o.data(), dir.data(), &opts, out32.data(), out.data());
  std::string msg = st == RTC_OK ? "" : api.last_error_string();
  api.scene_destroy(sc);
  if (st != RTC_OK) throw Error((int)st, "rtc_color_at: " + msg);
  std::vector<Color> res(rays.size());
  for (size_t i = 0; i < rays.size(); ++i) res[i] = {out[3 * i], out[3 * i + 1], out[3 * i + 2]};
  if (f32_out) *f32_out = std::move(out32);
  return res;
}

void Camera::render_into(const World& world, float* rgb32, double* rgb64, RenderStats* stats, int recursion_limit,
                         int64_t render_flags, int device, const std::vector<int>* devices) const {
  const DeviceApi& api = device_api();
  FlatScene flat = world.flatten();
  RtcSceneDesc d = flat.desc();
  RtcCameraDesc c = desc();
  RtcRenderOpts opts{};
  opts.recursion_limit = recursion_limit >= 0 ? (uint32_t)recursion_limit : world.recursion_limit;
  opts.flags = render_flags >= 0 ? (uint32_t)render_flags : world.render_flags;
  RtcStats s{};
  const std::vector<int>& devs = devices ? *devices : world.devices;
  if (devs.size() > 1 || devices)
    check(api, api.render_multi(&d, &c, &opts, (int)devs.size(), devs.data(), rgb32, rgb64, stats ? &s : nullptr), "rtc_render_multi");
  else
    check(api, api.render(&d, &c, &opts, device >= 0 ? device : (devs.size() == 1 ? devs[0] : world.device), rgb32, rgb64,
                          stats ? &s : nullptr), "rtc_render");
  if (stats) {
    stats->radiance_rays = s.radiance_rays, stats->shadow_rays = s.shadow_rays;
    stats->triangle_tests = s.triangle_tests, stats->primitive_tests = s.primitive_tests;
    stats->filter_tests = s.filter_tests, stats->box_tests = s.box_tests, stats->culled_rays = s.culled_rays, stats->unswept_shadow_rays = s.unswept_shadow_rays;
    stats->kernel_ms = s.kernel_ms, stats->upload_ms = s.upload_ms, stats->download_ms = s.download_ms;
  }
}

Canvas Camera::render(const World& world, RenderStats* stats) const {
  Canvas image(hsize_, vsize_);
  // Color is three doubles: the Canvas's pixel array IS the device's [n][3] f64 framebuffer layout
  render_into(world, image.pixels_f32_mut().data(), reinterpret_cast<double*>(image.pixels_mut().data()), stats);
  return image;
}

// ------------------------------------------------------------------ obj_file.rs:29-128
namespace {
std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  size_t i = 0;
  auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; };
  while (i < s.size()) {
    while (i < s.size() && ws(s[i])) ++i;
    size_t b = i;
    while (i < s.size() && !ws(s[i])) ++i;
    if (i > b) out.emplace_back(s, b, i - b);
  }
  return out;
}
// f64::from_str grammar (no hex, no trailing junk)
bool to_f64(const std::string& tok, double& out) {
  size_t i = 0;
  const size_t n = tok.size();
  if (i < n && (tok[i] == '+' || tok[i] == '-')) ++i;
  std::string lower;
  for (size_t k = i; k < n; ++k) lower += (char)std::tolower((unsigned char)tok[k]);
  if (lower == "inf" || lower == "infinity" || lower == "nan") {
    out = std::strtod(tok.c_str(), nullptr);
    return true;
  }
  size_t digits = 0;
  while (i < n && std::isdigit((unsigned char)tok[i])) ++i, ++digits;
  if (i < n && tok[i] == '.') {
    ++i;
    while (i < n && std::isdigit((unsigned char)tok[i])) ++i, ++digits;
  }
  if (!digits) return false;
  if (i < n && (tok[i] == 'e' || tok[i] == 'E')) {
    ++i;
    if (i < n && (tok[i] == '+' || tok[i] == '-')) ++i;
    size_t ed = 0;
    while (i < n && std::isdigit((unsigned char)tok[i])) ++i, ++ed;
    if (!ed) return false;
  }
  if (i != n) return false;
  out = std::strtod(tok.c_str(), nullptr);
  return true;
}
bool to_usize(const std::string& tok, size_t& out) {
  size_t i = (!tok.empty() && tok[0] == '+') ? 1 : 0;
  if (i == tok.size()) return false;
  size_t v = 0;
  for (; i < tok.size(); ++i) {
    if (!std::isdigit((unsigned char)tok[i])) return false;
    size_t dgt = (size_t)(tok[i] - '0');
    if (v > (std::numeric_limits<size_t>::max() - dgt) / 10) return false;
    v = v * 10 + dgt;
  }
  out = v;
  return true;
}
}  // namespace

Tuple Parser::vertices(size_t one_based_index) const {
  if (one_based_index == 0 || one_based_index > vertices_.size()) throw Error(HOST_ERR_PARSE, "vertex index out of range");
  return vertices_[one_based_index - 1];
}
Parser Parser::from_obj_file(const std::string& filename) {
  std::ifstream f(filename, std::ios::binary);
  if (!f) throw Error(HOST_ERR_PARSE, "something went wrong reading " + filename + ".");
  std::stringstream ss;
  ss << f.rdbuf();
  return from_obj_str(ss.str());
}
Parser Parser::from_obj_str(const std::string& text) {
  Parser p;
  int current = -1;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    std::string line = text.substr(pos, eol - pos);
    pos = eol + 1;
    auto tok = split_ws(line);
    if (tok.empty()) continue;
    auto bad = [&](const std::string& what) { return Error(HOST_ERR_PARSE, what + " in \"" + line + "\""); };
    if (tok[0] == "v") {
      double c[3];
      const char* nm[3] = {"x", "y", "z"};
      for (size_t k = 0; k < 3; ++k) {
        if (tok.size() < k + 2) throw bad(std::string("vertex token to have a ") + nm[k]);
        if (!to_f64(tok[k + 1], c[k])) throw bad(std::string("vertex ") + nm[k] + " should be an f64");
      }
      p.vertices_.push_back(Tuple::point(c[0], c[1], c[2]));
    } else if (tok[0] == "f") {
      size_t v1, v2, v3;
      if (tok.size() < 2) throw bad("face should have a v1");
      if (!to_usize(tok[1], v1)) throw bad("face v1 should be a usize");
      if (tok.size() < 3) throw bad("face should have a v2");
      if (!to_usize(tok[2], v2)) throw bad("face v2 should be a usize");
      for (size_t k = 3; k < tok.size(); ++k) {  // fan triangulation (:70-94)
        if (!to_usize(tok[k], v3)) throw bad("face v3 should be a usize");
        Shape tri = Shape::triangle(p.vertices(v1), p.vertices(v2), p.vertices(v3));
        if (current >= 0)
          p.named_groups[(size_t)current].second.push_shape(std::move(tri));
        else
          p.default_group.push_shape(std::move(tri));
        v2 = v3;
      }
    } else if (tok[0] == "g") {
      if (tok.size() < 2) throw bad("group should have a name");
      current = -1;
      for (size_t k = 0; k < p.named_groups.size(); ++k)
        if (p.named_groups[k].first == tok[1]) current = (int)k;
      if (current >= 0) {
        p.named_groups[(size_t)current].second = Shape::group();  // HashMap::insert replaces (:101-103)
      } else {
        p.named_groups.emplace_back(tok[1], Shape::group());
        current = (int)p.named_groups.size() - 1;
      }
    } else {
      p.ignored_lines += 1;
    }
  }
  return p;
}
Shape Parser::obj_to_group() && {
  Shape g = Shape::group();
  g.push_shape(std::move(default_group));
  for (auto& kv : named_groups) g.push_shape(std::move(kv.second));
  return g;
}

}  // namespace rtc::host
