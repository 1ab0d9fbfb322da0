// rtc_host.hpp — C++ host side of the MI355X renderer: the reference's public scene API
// (Tuple / Matrix / transformations / Color / Light / Pattern / Material / Shape / World / Camera / Canvas /
// obj Parser) with `Camera::render` and `World::color_at` routed through the C ABI in include/rtc_hip.h.
//
// The Rust toolchain is not available in this image, so the host that `north_star` asks for in Rust is written
// in C++ with the same names, argument meaning and error behaviour (panics become rtc::host::Error).  This
// library holds NO renderer: it builds the scene tree, flattens it (flatten.cpp) and calls librtc_hip.so.
// If the HIP library or a GPU is missing, render() throws — there is no CPU fallback.
//
// Citations are reference file:line under /root/reference/src.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtc_hip.h"

namespace rtc::host {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
// error codes of the flat C API (negative); RtcStatus values from the device library pass through unchanged
enum : int {
  HOST_ERR_NOT_INVERTIBLE = -101,   // Matrix::inverse() == None -> expect() panic (shape.rs:215, camera.rs:45, pattern.rs:65)
  HOST_ERR_TRANSFORM_TWICE = -102,  // shape.rs:199-201
  HOST_ERR_NOT_GROUP = -103,        // shape.rs:533
  HOST_ERR_PARSE = -104,            // obj_file.rs panics
  HOST_ERR_ARG = -105,
  HOST_ERR_NO_DEVICE_LIB = -106     // librtc_hip.so could not be loaded
};

constexpr double EPSILON = 0.00001;  // utils.rs:2

// ------------------------------------------------------------------ tuple.rs / color.rs
struct Tuple {
  double x = 0, y = 0, z = 0, w = 0;
  static Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
  static Tuple vector(double x, double y, double z) { return {x, y, z, 0.0}; }
  double magnitude() const;
  Tuple normalize() const;
  double dot(const Tuple& o) const;
  Tuple cross(const Tuple& o) const;
};
Tuple operator+(const Tuple& a, const Tuple& b);
Tuple operator-(const Tuple& a, const Tuple& b);
Tuple operator-(const Tuple& a);
Tuple operator*(const Tuple& a, double s);
bool approx_eq(const Tuple& a, const Tuple& b);

struct Color {
  double red = 0, green = 0, blue = 0;
};
static_assert(sizeof(Color) == 3 * sizeof(double), "Canvas hands its Color array to the device as [n][3] doubles");
bool approx_eq(const Color& a, const Color& b);
inline constexpr Color BLACK{0, 0, 0}, WHITE{1, 1, 1}, RED{1, 0, 0}, GREEN{0, 1, 0}, BLUE{0, 0, 1};

// ------------------------------------------------------------------ matrix.rs / transformations.rs
struct Matrix4 {
  std::array<double, 16> m{};  // row-major
  double& at(int r, int c) { return m[(size_t)(r * 4 + c)]; }
  double at(int r, int c) const { return m[(size_t)(r * 4 + c)]; }
  static Matrix4 identity();
  Matrix4 transpose() const;
  double determinant() const;
  std::optional<Matrix4> inverse() const;  // matrix.rs:138-157 (None when |det| < EPSILON)
};
Matrix4 operator*(const Matrix4& a, const Matrix4& b);
Tuple operator*(const Matrix4& a, const Tuple& t);
bool approx_eq(const Matrix4& a, const Matrix4& b);

Matrix4 translation(double x, double y, double z);
Matrix4 scaling(double x, double y, double z);
Matrix4 rotation_x(double rad);
Matrix4 rotation_y(double rad);
Matrix4 rotation_z(double rad);
Matrix4 shearing(double xy, double xz, double yx, double yz, double zx, double zy);
Matrix4 view_transform(const Tuple& from, const Tuple& to, const Tuple& up);

// ------------------------------------------------------------------ ray.rs / light.rs
struct Ray {
  Tuple origin, direction;
};
struct Light {
  Tuple position = Tuple::point(0, 0, 0);
  Color intensity = WHITE;
};

// ------------------------------------------------------------------ pattern.rs
class Pattern {
 public:
  static Pattern stripe(Color a, Color b) { return Pattern(RTC_PATTERN_STRIPE, a, b); }
  static Pattern gradient(Color from, Color to) { return Pattern(RTC_PATTERN_GRADIENT, from, to); }
  static Pattern ring(Color a, Color b) { return Pattern(RTC_PATTERN_RING, a, b); }
  static Pattern checkers(Color a, Color b) { return Pattern(RTC_PATTERN_CHECKERS, a, b); }
  static Pattern test_pattern() { return Pattern(RTC_PATTERN_TEST, BLACK, BLACK); }
  void set_transform(const Matrix4& t);  // throws HOST_ERR_NOT_INVERTIBLE (pattern.rs:63-66)
  int kind() const { return kind_; }
  const Color& a() const { return a_; }
  const Color& b() const { return b_; }
  const Matrix4& transform() const { return transform_; }
  const Matrix4& transform_inverse() const { return transform_inverse_; }
  bool operator==(const Pattern& o) const;  // derived PartialEq (approximate on matrices / colours)

 private:
  Pattern(int k, Color a, Color b) : kind_(k), a_(a), b_(b) {}
  int kind_;
  Color a_, b_;
  Matrix4 transform_ = Matrix4::identity(), transform_inverse_ = Matrix4::identity();
};

// ------------------------------------------------------------------ material.rs:4-29
struct Material {
  Color color = WHITE;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0, reflective = 0.0;
  std::optional<Pattern> pattern;
  double transparency = 0.0, refractive_index = 1.0;
  bool operator==(const Material& o) const;  // derived PartialEq
};

// ------------------------------------------------------------------ shape.rs / bounds.rs
enum class ShapeKind { Sphere, Plane, Cube, Cylinder, Cone, Group, Triangle };

struct Bounds {
  Tuple min, max;
};

class Shape {
 public:
  static Shape sphere() { return Shape(ShapeKind::Sphere); }
  static Shape glass_sphere();
  static Shape plane() { return Shape(ShapeKind::Plane); }
  static Shape cube() { return Shape(ShapeKind::Cube); }
  static Shape infinite_cylinder();
  static Shape cylinder(double minimum, double maximum, bool capped);
  static Shape infinite_cone();
  static Shape cone(double minimum, double maximum, bool capped);
  static Shape group() { return Shape(ShapeKind::Group); }
  static Shape triangle(const Tuple& p1, const Tuple& p2, const Tuple& p3);

  void set_transform(const Matrix4& t);  // throws TRANSFORM_TWICE / NOT_INVERTIBLE (shape.rs:196-218)
  void set_material(const Material& m);  // recursive on groups (shape.rs:220-229)
  const Material& get_material() const { return material_; }
  Material& get_material_mut() { return material_; }
  const Matrix4& transform() const { return transform_; }
  const Matrix4& transform_inverse() const { return transform_inverse_; }
  const std::vector<Shape>* shapes() const { return kind_ == ShapeKind::Group ? &children_ : nullptr; }
  void push_shape(Shape s);  // throws NOT_GROUP (shape.rs:526-535)
  bool operator==(const Shape& o) const;  // shape.rs:638-646 value equality
  Bounds bounds() const;                  // Bounds::new (bounds.rs:11-140)

  ShapeKind kind() const { return kind_; }
  double minimum = 0, maximum = 0;  // Cylinder / Cone
  bool capped = false;
  Tuple p1, p2, p3, e1, e2, normal;  // Triangle (shape.rs:31-38)
  size_t leaf_count() const;

 private:
  explicit Shape(ShapeKind k) : kind_(k) {}
  void set_transform_internal(const Matrix4& t);
  bool kind_eq(const Shape& o) const;
  ShapeKind kind_;
  std::vector<Shape> children_;
  Matrix4 transform_ = Matrix4::identity(), transform_inverse_ = Matrix4::identity();
  bool transformed_ = false;
  Material material_;
};

// ------------------------------------------------------------------ canvas.rs
class Canvas {
 public:
  Canvas(size_t width, size_t height) : width_(width), height_(height), pixels_(width * height), f32_(width * height * 3, 0.f) {}
  size_t width() const { return width_; }
  size_t height() const { return height_; }
  Color get_pixel(size_t x, size_t y) const { return pixels_[x + y * width_]; }
  void set_pixel(size_t x, size_t y, Color c) {
    pixels_[x + y * width_] = c;
    float* p = &f32_[3 * (x + y * width_)];
    p[0] = (float)c.red, p[1] = (float)c.green, p[2] = (float)c.blue;
  }
  std::string to_ppm() const;  // canvas.rs:28-63, byte for byte
  const std::vector<Color>& pixels() const { return pixels_; }
  // the device's f32 framebuffer (the values the parity metric is measured on)
  const std::vector<float>& pixels_f32() const { return f32_; }
  std::vector<Color>& pixels_mut() { return pixels_; }
  std::vector<float>& pixels_f32_mut() { return f32_; }

 private:
  size_t width_, height_;
  std::vector<Color> pixels_;
  std::vector<float> f32_;
};

// ------------------------------------------------------------------ flatten (flatten.cpp)
struct FlatScene {  // owns the SoA arrays an RtcSceneDesc points into
  std::vector<double> transform_inverse;
  std::vector<RtcMaterial> materials;
  std::vector<RtcPattern> patterns;
  std::vector<int32_t> group_parent;
  std::vector<double> group_bounds_min, group_bounds_max;
  std::vector<uint8_t> leaf_kind;
  std::vector<uint32_t> leaf_transform, leaf_material, leaf_class, leaf_triangle;
  std::vector<int32_t> leaf_group;
  std::vector<double> leaf_params;
  std::vector<double> tri_p1, tri_e1, tri_e2, tri_normal;
  double light_position[3]{}, light_intensity[3]{};
  RtcSceneDesc desc() const;
};

// ------------------------------------------------------------------ world.rs
struct RenderStats {
  uint64_t radiance_rays = 0, shadow_rays = 0;
  uint64_t triangle_tests = 0, primitive_tests = 0, filter_tests = 0, box_tests = 0, culled_rays = 0, unswept_shadow_rays = 0;
  double kernel_ms = 0, upload_ms = 0, download_ms = 0;
};

class World {
 public:
  explicit World(Light l) : light(std::move(l)) {}
  static World default_world();  // world.rs:26-41
  std::vector<Shape> objects;
  Light light;
  int recursion_limit = 5;  // RECURSION_LIMIT (world.rs:11): a constant in the reference, a knob here
  int device = 0;
  std::vector<int> devices;  // more than one entry: Camera::render shards the frame's rows over these devices
                             // (rtc_render_multi: interleaved 16-row blocks, gathered to devices[0] over xGMI)
  uint32_t render_flags = 0;  // RTC_FLAG_* passed through (e.g. RTC_FLAG_EXACT_SWEEP for the brute-force sweep)
  FlatScene flatten(bool always_classes = false) const;  // leaf_class is left empty unless a material is transparent
  // World::color_at (world.rs:80-82), batched; returns f64 colours computed on the GPU
  std::vector<Color> color_at(const std::vector<Ray>& rays, std::vector<float>* f32_out = nullptr) const;
  Color color_at(const Ray& r) const { return color_at(std::vector<Ray>{r})[0]; }
};

// ------------------------------------------------------------------ camera.rs
class Camera {
 public:
  Camera(size_t hsize, size_t vsize, double field_of_view);  // camera.rs:16-41
  void set_transform(const Matrix4& t);                      // camera.rs:43-46
  Ray ray_for_pixel(size_t px, size_t py) const;             // camera.rs:48-65 (host copy, for tests/tools)
  Canvas render(const World& world, RenderStats* stats = nullptr) const;  // camera.rs:67-80 -> rtc_render
  // The same, straight into caller storage (rgb32: hsize*vsize*3 floats; rgb64: as many doubles or null), with the
  // world's recursion limit / flags / device optionally overridden (negative = keep): no intermediate image copies.
  void render_into(const World& world, float* rgb32, double* rgb64, RenderStats* stats = nullptr, int recursion_limit = -1,
                   int64_t render_flags = -1, int device = -1, const std::vector<int>* devices = nullptr) const;
  RtcCameraDesc desc() const;
  size_t hsize() const { return hsize_; }
  size_t vsize() const { return vsize_; }
  double field_of_view() const { return field_of_view_; }
  double pixel_size() const { return pixel_size_; }
  double half_width() const { return half_width_; }
  double half_height() const { return half_height_; }
  const Matrix4& transform() const { return transform_; }
  const Matrix4& transform_inverse() const { return transform_inverse_; }

 private:
  size_t hsize_, vsize_;
  double field_of_view_;
  Matrix4 transform_ = Matrix4::identity(), transform_inverse_ = Matrix4::identity();
  double pixel_size_ = 0, half_width_ = 0, half_height_ = 0;
};

// ------------------------------------------------------------------ obj_file.rs
class Parser {
 public:
  static Parser from_obj_file(const std::string& filename);
  static Parser from_obj_str(const std::string& text);
  Tuple vertices(size_t one_based_index) const;
  size_t ignored_lines = 0;
  Shape default_group = Shape::group();
  std::vector<std::pair<std::string, Shape>> named_groups;  // HashMap in the reference (order unspecified there;
                                                            // first-declaration order here)
  Shape obj_to_group() &&;  // consumes the parser (obj_file.rs:120-128)

 private:
  std::vector<Tuple> vertices_;
};

// device library binding (dlopen of librtc_hip.so next to this library); throws HOST_ERR_NO_DEVICE_LIB
struct DeviceApi {
  decltype(&rtc_render) render;
  decltype(&rtc_render_multi) render_multi;
  decltype(&rtc_scene_create) scene_create;
  decltype(&rtc_scene_destroy) scene_destroy;
  decltype(&rtc_color_at) color_at;
  decltype(&rtc_last_error_string) last_error_string;
  decltype(&rtc_device_count) device_count;
};
const DeviceApi& device_api();

}  // namespace rtc::host
