// rtc_host_capi.cpp — flat C API over rtc_host.hpp (see rtc_host_capi.h).  Exceptions never cross the
// boundary: every entry point converts rtc::host::Error into a status code + thread-local message.
#include "rtc_host_capi.h"

#include <cstring>
#include <string>

#include "rtc_host.hpp"

using namespace rtc::host;

struct RtchShape {
  Shape s;
};
struct RtchWorld {
  World w;
};
struct RtchCamera {
  Camera c;
};

namespace {
thread_local std::string g_err;
int fail(const Error& e) {
  g_err = e.what();
  return e.code;
}
Matrix4 min_(const double* m) {
  Matrix4 r;
  std::memcpy(r.m.data(), m, 16 * sizeof(double));
  return r;
}
void mout(const Matrix4& r, double* m) { std::memcpy(m, r.m.data(), 16 * sizeof(double)); }

Material mat_in(const RtchMaterial* m) {
  Material r;
  r.color = {m->color[0], m->color[1], m->color[2]};
  r.ambient = m->ambient, r.diffuse = m->diffuse, r.specular = m->specular, r.shininess = m->shininess;
  r.reflective = m->reflective, r.transparency = m->transparency, r.refractive_index = m->refractive_index;
  if (m->pattern_kind != 0) {
    Color a{m->pattern_a[0], m->pattern_a[1], m->pattern_a[2]}, b{m->pattern_b[0], m->pattern_b[1], m->pattern_b[2]};
    Pattern p = Pattern::test_pattern();
    switch (m->pattern_kind) {
      case RTC_PATTERN_STRIPE: p = Pattern::stripe(a, b); break;
      case RTC_PATTERN_GRADIENT: p = Pattern::gradient(a, b); break;
      case RTC_PATTERN_RING: p = Pattern::ring(a, b); break;
      case RTC_PATTERN_CHECKERS: p = Pattern::checkers(a, b); break;
      default: break;
    }
    p.set_transform(min_(m->pattern_transform));
    r.pattern = p;
  }
  return r;
}
void mat_out(const Material& r, RtchMaterial* m) {
  std::memset(m, 0, sizeof *m);
  m->color[0] = r.color.red, m->color[1] = r.color.green, m->color[2] = r.color.blue;
  m->ambient = r.ambient, m->diffuse = r.diffuse, m->specular = r.specular, m->shininess = r.shininess;
  m->reflective = r.reflective, m->transparency = r.transparency, m->refractive_index = r.refractive_index;
  mout(Matrix4::identity(), m->pattern_transform);
  if (r.pattern) {
    m->pattern_kind = r.pattern->kind();
    m->pattern_a[0] = r.pattern->a().red, m->pattern_a[1] = r.pattern->a().green, m->pattern_a[2] = r.pattern->a().blue;
    m->pattern_b[0] = r.pattern->b().red, m->pattern_b[1] = r.pattern->b().green, m->pattern_b[2] = r.pattern->b().blue;
    mout(r.pattern->transform(), m->pattern_transform);
  }
}
}  // namespace

extern "C" {

const char* rtch_last_error(void) { return g_err.c_str(); }

void rtch_identity(double* m) { mout(Matrix4::identity(), m); }
void rtch_translation(double x, double y, double z, double* m) { mout(translation(x, y, z), m); }
void rtch_scaling(double x, double y, double z, double* m) { mout(scaling(x, y, z), m); }
void rtch_rotation_x(double r, double* m) { mout(rotation_x(r), m); }
void rtch_rotation_y(double r, double* m) { mout(rotation_y(r), m); }
void rtch_rotation_z(double r, double* m) { mout(rotation_z(r), m); }
void rtch_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double* m) {
  mout(shearing(xy, xz, yx, yz, zx, zy), m);
}
void rtch_view_transform(const double* f, const double* t, const double* u, double* m) {
  mout(view_transform(Tuple::point(f[0], f[1], f[2]), Tuple::point(t[0], t[1], t[2]), Tuple::vector(u[0], u[1], u[2])), m);
}
void rtch_mat_mul(const double* a, const double* b, double* out) { mout(min_(a) * min_(b), out); }
int rtch_mat_inverse(const double* a, double* out) {
  auto inv = min_(a).inverse();
  if (!inv) {
    g_err = "matrix is not invertible";
    return HOST_ERR_NOT_INVERTIBLE;
  }
  mout(*inv, out);
  return 0;
}

void rtch_material_default(RtchMaterial* m) { mat_out(Material{}, m); }

static RtchShape* wrap(Shape s) { return new RtchShape{std::move(s)}; }
RtchShape* rtch_shape_sphere(void) { return wrap(Shape::sphere()); }
RtchShape* rtch_shape_glass_sphere(void) { return wrap(Shape::glass_sphere()); }
RtchShape* rtch_shape_plane(void) { return wrap(Shape::plane()); }
RtchShape* rtch_shape_cube(void) { return wrap(Shape::cube()); }
RtchShape* rtch_shape_cylinder(double mn, double mx, int capped) { return wrap(Shape::cylinder(mn, mx, capped != 0)); }
RtchShape* rtch_shape_cone(double mn, double mx, int capped) { return wrap(Shape::cone(mn, mx, capped != 0)); }
RtchShape* rtch_shape_group(void) { return wrap(Shape::group()); }
RtchShape* rtch_shape_triangle(const double* a, const double* b, const double* c) {
  return wrap(Shape::triangle(Tuple::point(a[0], a[1], a[2]), Tuple::point(b[0], b[1], b[2]), Tuple::point(c[0], c[1], c[2])));
}
void rtch_shape_free(RtchShape* s) { delete s; }
int rtch_shape_set_transform(RtchShape* s, const double* m) {
  try {
    s->s.set_transform(min_(m));
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}
void rtch_shape_set_material(RtchShape* s, const RtchMaterial* m) {
  try {
    s->s.set_material(mat_in(m));
  } catch (const Error& e) {
    fail(e);
  }
}
void rtch_shape_get_material(const RtchShape* s, RtchMaterial* m) { mat_out(s->s.get_material(), m); }
int rtch_shape_push(RtchShape* g, const RtchShape* child) {
  try {
    g->s.push_shape(child->s);
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}
int64_t rtch_shape_leaf_count(const RtchShape* s) { return (int64_t)s->s.leaf_count(); }
void rtch_shape_bounds(const RtchShape* s, double* mn, double* mx) {
  Bounds b = s->s.bounds();
  mn[0] = b.min.x, mn[1] = b.min.y, mn[2] = b.min.z;
  mx[0] = b.max.x, mx[1] = b.max.y, mx[2] = b.max.z;
}

RtchShape* rtch_obj_parse(const char* text, size_t len, int64_t* ignored, int* err) {
  try {
    Parser p = Parser::from_obj_str(std::string(text, len));
    if (ignored) *ignored = (int64_t)p.ignored_lines;
    if (err) *err = 0;
    return wrap(std::move(p).obj_to_group());
  } catch (const Error& e) {
    if (err) *err = fail(e);
    return nullptr;
  }
}

RtchWorld* rtch_world_new(const double* lp, const double* li) {
  return new RtchWorld{World(Light{Tuple::point(lp[0], lp[1], lp[2]), Color{li[0], li[1], li[2]}})};
}
RtchWorld* rtch_world_default(void) { return new RtchWorld{World::default_world()}; }
void rtch_world_free(RtchWorld* w) { delete w; }
void rtch_world_push(RtchWorld* w, const RtchShape* s) { w->w.objects.push_back(s->s); }
int rtch_world_flat_info(const RtchWorld* w, RtchFlatInfo* info) {
  try {
    FlatScene f = w->w.flatten(true);
    info->n_transforms = (uint32_t)(f.transform_inverse.size() / 16);
    info->n_materials = (uint32_t)f.materials.size();
    info->n_patterns = (uint32_t)f.patterns.size();
    info->n_groups = (uint32_t)f.group_parent.size();
    info->n_leaves = (uint32_t)f.leaf_kind.size();
    info->n_triangles = (uint32_t)(f.tri_p1.size() / 3);
    uint32_t shared = 0;
    for (size_t i = 0; i < f.leaf_class.size(); ++i) shared += f.leaf_class[i] != i;
    info->n_classes_shared = shared;
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}
int rtch_color_at(const RtchWorld* w, int64_t n, const double* o, const double* d, int limit, uint32_t flags, int device,
                  double* out64, float* out32) {
  try {
    World world = w->w;
    if (limit >= 0) world.recursion_limit = limit;  // negative: the world's own RECURSION_LIMIT
    world.render_flags = flags;
    world.device = device;
    std::vector<Ray> rays((size_t)n);
    for (int64_t i = 0; i < n; ++i)
      rays[(size_t)i] = {Tuple::point(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Tuple::vector(d[3 * i], d[3 * i + 1], d[3 * i + 2])};
    std::vector<float> f32;
    std::vector<Color> c = world.color_at(rays, &f32);
    for (int64_t i = 0; i < n; ++i) {
      if (out64) out64[3 * i] = c[(size_t)i].red, out64[3 * i + 1] = c[(size_t)i].green, out64[3 * i + 2] = c[(size_t)i].blue;
    }
    if (out32 && n) std::memcpy(out32, f32.data(), (size_t)n * 3 * sizeof(float));
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}

RtchCamera* rtch_camera_new(int64_t h, int64_t v, double fov) { return new RtchCamera{Camera((size_t)h, (size_t)v, fov)}; }
void rtch_camera_free(RtchCamera* c) { delete c; }
int rtch_camera_set_transform(RtchCamera* c, const double* m) {
  try {
    c->c.set_transform(min_(m));
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}
void rtch_camera_get(const RtchCamera* c, double* ps, double* hw, double* hh, double* tinv) {
  if (ps) *ps = c->c.pixel_size();
  if (hw) *hw = c->c.half_width();
  if (hh) *hh = c->c.half_height();
  if (tinv) mout(c->c.transform_inverse(), tinv);
}
void rtch_camera_ray_for_pixel(const RtchCamera* c, int64_t px, int64_t py, double* o, double* d) {
  Ray r = c->c.ray_for_pixel((size_t)px, (size_t)py);
  o[0] = r.origin.x, o[1] = r.origin.y, o[2] = r.origin.z;
  d[0] = r.direction.x, d[1] = r.direction.y, d[2] = r.direction.z;
}
static void stats_out(const RenderStats& rs, RtchStats* stats) {
  stats->radiance_rays = rs.radiance_rays, stats->shadow_rays = rs.shadow_rays;
  stats->triangle_tests = rs.triangle_tests, stats->primitive_tests = rs.primitive_tests;
  stats->filter_tests = rs.filter_tests, stats->box_tests = rs.box_tests, stats->culled_rays = rs.culled_rays, stats->unswept_shadow_rays = rs.unswept_shadow_rays;
  stats->kernel_ms = rs.kernel_ms, stats->upload_ms = rs.upload_ms, stats->download_ms = rs.download_ms;
}
int rtch_render(const RtchCamera* c, const RtchWorld* w, int limit, uint32_t flags, int device, float* out32,
                double* out64, RtchStats* stats) {
  try {
    RenderStats rs;
    // stats == NULL must reach rtc_render as NULL: only then does the uncounted, untimed kernel run
    c->c.render_into(w->w, out32, out64, stats ? &rs : nullptr, limit, (int64_t)flags, device);
    if (stats) stats_out(rs, stats);
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}
int rtch_render_multi(const RtchCamera* c, const RtchWorld* w, int limit, uint32_t flags, int n_devices, const int* devices,
                      float* out32, double* out64, RtchStats* stats) {
  try {
    RenderStats rs;
    std::vector<int> devs(devices, devices + (n_devices > 0 ? n_devices : 0));
    c->c.render_into(w->w, out32, out64, stats ? &rs : nullptr, limit, (int64_t)flags, -1, &devs);
    if (stats) stats_out(rs, stats);
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}

int rtch_world_scene_create(const RtchWorld* w, int device, void** out_scene) {
  try {
    const DeviceApi& api = device_api();
    FlatScene flat = w->w.flatten();
    RtcSceneDesc d = flat.desc();
    RtcScene* sc = nullptr;
    RtcStatus st = api.scene_create(&d, device, &sc);
    if (st != RTC_OK) {
      g_err = std::string("rtc_scene_create: ") + api.last_error_string();
      return (int)st;
    }
    *out_scene = sc;
    return 0;
  } catch (const Error& e) {
    return fail(e);
  }
}
void rtch_camera_desc(const RtchCamera* c, void* out) {
  RtcCameraDesc d = c->c.desc();
  std::memcpy(out, &d, sizeof d);
}

size_t rtch_canvas_to_ppm(int64_t width, int64_t height, const double* rgb, char* out, size_t cap) {
  Canvas c((size_t)width, (size_t)height);
  for (int64_t y = 0; y < height; ++y)
    for (int64_t x = 0; x < width; ++x) {
      const double* p = rgb + 3 * (x + y * width);
      c.set_pixel((size_t)x, (size_t)y, Color{p[0], p[1], p[2]});
    }
  std::string s = c.to_ppm();
  if (out && cap) std::memcpy(out, s.data(), std::min(cap, s.size()));
  return s.size();
}

}  // extern "C"
