// rtc_oracle.cpp — CPU oracle: OBJ parser, PPM writer and the flat C API (TEST INFRASTRUCTURE ONLY).
// See rtc_oracle.h for the role of this library and rtc_oracle_core.hpp for the restated path.
#include "rtc_oracle.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "rtc_oracle_core.hpp"

namespace orc {

// ---------------------------------------------------------------- obj_file.rs:29-128
namespace {
struct Tok {
  const char* p;
  size_t n;
};
bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// Rust's `str::parse::<f64>` grammar: [+-]? (digits [. digits?]? | . digits) ([eE][+-]?digits)? | inf | infinity | nan
bool parse_f64(Tok t, double& out) {
  std::string s(t.p, t.n);
  size_t i = 0;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) ++i;
  std::string rest = s.substr(i);
  for (auto& c : rest) c = (char)tolower((unsigned char)c);
  if (rest == "inf" || rest == "infinity" || rest == "nan") {
    out = std::strtod(s.c_str(), nullptr);
    return true;
  }
  size_t j = i, nd = 0;
  while (j < s.size() && isdigit((unsigned char)s[j])) ++j, ++nd;
  if (j < s.size() && s[j] == '.') {
    ++j;
    while (j < s.size() && isdigit((unsigned char)s[j])) ++j, ++nd;
  }
  if (nd == 0) return false;
  if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
    ++j;
    if (j < s.size() && (s[j] == '+' || s[j] == '-')) ++j;
    size_t ne = 0;
    while (j < s.size() && isdigit((unsigned char)s[j])) ++j, ++ne;
    if (ne == 0) return false;
  }
  if (j != s.size()) return false;
  out = std::strtod(s.c_str(), nullptr);  // correctly rounded, like Rust's parser
  return true;
}
// `parse::<usize>`: [+]? digits, no overflow
bool parse_usize(Tok t, uint64_t& out) {
  size_t i = 0;
  if (i < t.n && t.p[i] == '+') ++i;
  if (i == t.n) return false;
  uint64_t v = 0;
  for (; i < t.n; ++i) {
    if (!isdigit((unsigned char)t.p[i])) return false;
    uint64_t d = (uint64_t)(t.p[i] - '0');
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  out = v;
  return true;
}
}  // namespace

ObjResult parse_obj(const char* text, size_t len) {
  ObjResult res;
  std::vector<Tuple> vertices;
  Shape default_group = group();
  std::vector<std::pair<std::string, Shape>> named;  // HashMap in the reference: iteration order unspecified;
                                                     // the oracle emits first-declaration order.
  int current = -1;                                  // index into `named`, -1 = default group
  size_t pos = 0;
  auto fail = [&](const std::string& msg, const std::string& line) {
    res.error = msg + " in \"" + line + "\"";
    return res;
  };
  while (pos < len) {
    size_t eol = pos;
    while (eol < len && text[eol] != '\n') ++eol;
    std::string line(text + pos, eol - pos);
    pos = eol + 1;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::vector<Tok> toks;
    size_t i = 0;
    while (i < line.size()) {
      while (i < line.size() && is_ws(line[i])) ++i;
      size_t b = i;
      while (i < line.size() && !is_ws(line[i])) ++i;
      if (i > b) toks.push_back({line.data() + b, i - b});
    }
    if (toks.empty()) continue;  // blank lines are not "ignored lines" (obj_file.rs:37)
    std::string head(toks[0].p, toks[0].n);
    if (head == "v") {
      double c[3];
      static const char* nm[3] = {"x", "y", "z"};
      for (int k = 0; k < 3; ++k) {
        if (toks.size() < (size_t)k + 2) return fail(std::string("vertex token to have a ") + nm[k], line);
        if (!parse_f64(toks[k + 1], c[k])) return fail(std::string("vertex ") + nm[k] + " should be an f64", line);
      }
      vertices.push_back(Tuple::point(c[0], c[1], c[2]));
    } else if (head == "f") {
      uint64_t v1, v2;
      if (toks.size() < 2) return fail("face should have a v1", line);
      if (!parse_usize(toks[1], v1)) return fail("face v1 should be a usize", line);
      if (toks.size() < 3) return fail("face should have a v2", line);
      if (!parse_usize(toks[2], v2)) return fail("face v2 should be a usize", line);
      for (size_t k = 3; k < toks.size(); ++k) {
        uint64_t v3;
        if (!parse_usize(toks[k], v3)) return fail("face v3 should be a usize", line);
        auto vert = [&](uint64_t one_based, Tuple& out) {
          if (one_based == 0 || one_based > vertices.size()) return false;  // index panic in the reference
          out = vertices[one_based - 1];
          return true;
        };
        Tuple p1, p2, p3;
        if (!vert(v1, p1) || !vert(v2, p2) || !vert(v3, p3)) return fail("vertex index out of range", line);
        Shape tri = triangle(p1, p2, p3);
        if (current >= 0)
          named[(size_t)current].second.shapes.push_back(tri);
        else
          default_group.shapes.push_back(tri);
        v2 = v3;
      }
    } else if (head == "g") {
      if (toks.size() < 2) return fail("group should have a name", line);
      std::string name(toks[1].p, toks[1].n);
      int found = -1;
      for (size_t k = 0; k < named.size(); ++k)
        if (named[k].first == name) found = (int)k;
      if (found >= 0) {
        named[(size_t)found].second = group();  // HashMap::insert replaces the earlier group
        current = found;
      } else {
        named.emplace_back(name, group());
        current = (int)named.size() - 1;
      }
    } else {
      res.ignored_lines += 1;
    }
  }
  res.n_vertices = vertices.size();
  Shape g = group();  // obj_to_group :120-128
  g.shapes.push_back(default_group);
  for (auto& kv : named) g.shapes.push_back(kv.second);
  res.group = std::move(g);
  return res;
}

// ---------------------------------------------------------------- canvas.rs:28-63
static std::string color_to_str(double c) {
  double cl = c;  // f64::clamp keeps NaN
  if (cl < 0.) cl = 0.;
  if (cl > 1.) cl = 1.;
  double r = std::round(cl * 255.);
  int v = std::isnan(r) ? 0 : (int)r;  // `as i32` saturates, NaN -> 0
  return std::to_string(v);
}
std::string canvas_to_ppm(int64_t width, int64_t height, const double* rgb) {
  std::string out;
  out += "P3\n";
  out += std::to_string(width) + " " + std::to_string(height) + "\n";
  out += "255\n";
  for (int64_t y = 0; y < height; ++y) {
    size_t len = 0;
    for (int64_t x = 0; x < width; ++x) {
      const double* c = rgb + 3 * (x + y * width);
      for (int k = 0; k < 3; ++k) {
        std::string s = color_to_str(c[k]);
        if (len + s.size() + 1 > 70) {
          out += "\n";
          len = 0;
        }
        if (len > 0) {
          out += " ";
          len += 1;
        }
        out += s;
        len += s.size();
      }
    }
    out += "\n";
  }
  return out;
}

static void warm_bounds(const Shape& s) {
  if (s.kind != Kind::Group) return;
  if (!s.bounds_cached) {
    s.bounds_cache = bounds_new(s);
    s.bounds_cached = true;
  }
  for (const auto& c : s.shapes) warm_bounds(c);
}

}  // namespace orc

// ================================================================= C API
using namespace orc;

struct OrcShape {
  Shape s;
};
struct OrcWorld {
  World w;
};
struct OrcCamera {
  Camera c;
};

static Matrix4 m_in(const double* m) {
  Matrix4 r;
  std::memcpy(r.v, m, sizeof r.v);
  return r;
}
static void m_out(const Matrix4& r, double* m) { std::memcpy(m, r.v, sizeof r.v); }

extern "C" {

void orc_identity(double* m) { m_out(Matrix4::identity(), m); }
void orc_translation(double x, double y, double z, double* m) { m_out(translation(x, y, z), m); }
void orc_scaling(double x, double y, double z, double* m) { m_out(scaling(x, y, z), m); }
void orc_rotation_x(double r, double* m) { m_out(rotation_x(r), m); }
void orc_rotation_y(double r, double* m) { m_out(rotation_y(r), m); }
void orc_rotation_z(double r, double* m) { m_out(rotation_z(r), m); }
void orc_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double* m) {
  m_out(shearing(xy, xz, yx, yz, zx, zy), m);
}
void orc_view_transform(const double* f, const double* t, const double* u, double* m) {
  m_out(view_transform(Tuple::point(f[0], f[1], f[2]), Tuple::point(t[0], t[1], t[2]),
                       Tuple::vector(u[0], u[1], u[2])),
        m);
}
void orc_mat_mul(const double* a, const double* b, double* out) { m_out(m_in(a) * m_in(b), out); }
int orc_mat_inverse(const double* a, double* out) {
  auto inv = inverse(m_in(a));
  if (!inv) return ORC_ERR_NOT_INVERTIBLE;
  m_out(*inv, out);
  return ORC_OK;
}

static Material mat_in(const OrcMaterial* m) {
  Material r;
  r.color = {m->color[0], m->color[1], m->color[2]};
  r.ambient = m->ambient, r.diffuse = m->diffuse, r.specular = m->specular, r.shininess = m->shininess;
  r.reflective = m->reflective, r.transparency = m->transparency, r.refractive_index = m->refractive_index;
  if (m->pattern_kind != 0) {
    Pattern p;
    p.kind = (PatternKind)m->pattern_kind;
    p.a = {m->pattern_a[0], m->pattern_a[1], m->pattern_a[2]};
    p.b = {m->pattern_b[0], m->pattern_b[1], m->pattern_b[2]};
    p.set_transform(m_in(m->pattern_transform));  // identity inverts to identity
    r.pattern = p;
  }
  return r;
}
static void mat_out(const Material& r, OrcMaterial* m) {
  std::memset(m, 0, sizeof *m);
  m->color[0] = r.color.red, m->color[1] = r.color.green, m->color[2] = r.color.blue;
  m->ambient = r.ambient, m->diffuse = r.diffuse, m->specular = r.specular, m->shininess = r.shininess;
  m->reflective = r.reflective, m->transparency = r.transparency, m->refractive_index = r.refractive_index;
  m_out(Matrix4::identity(), m->pattern_transform);
  if (r.pattern) {
    m->pattern_kind = (int)r.pattern->kind;
    m->pattern_a[0] = r.pattern->a.red, m->pattern_a[1] = r.pattern->a.green, m->pattern_a[2] = r.pattern->a.blue;
    m->pattern_b[0] = r.pattern->b.red, m->pattern_b[1] = r.pattern->b.green, m->pattern_b[2] = r.pattern->b.blue;
    m_out(r.pattern->transform, m->pattern_transform);
  }
}
void orc_material_default(OrcMaterial* m) { mat_out(Material{}, m); }

static OrcShape* wrap(Shape s) { return new OrcShape{std::move(s)}; }
OrcShape* orc_shape_sphere(void) { return wrap(sphere()); }
OrcShape* orc_shape_glass_sphere(void) { return wrap(glass_sphere()); }
OrcShape* orc_shape_plane(void) { return wrap(plane()); }
OrcShape* orc_shape_cube(void) { return wrap(cube()); }
OrcShape* orc_shape_cylinder(double mn, double mx, int capped) { return wrap(cylinder(mn, mx, capped != 0)); }
OrcShape* orc_shape_cone(double mn, double mx, int capped) { return wrap(cone(mn, mx, capped != 0)); }
OrcShape* orc_shape_group(void) { return wrap(group()); }
OrcShape* orc_shape_triangle(const double* a, const double* b, const double* c) {
  return wrap(triangle(Tuple::point(a[0], a[1], a[2]), Tuple::point(b[0], b[1], b[2]), Tuple::point(c[0], c[1], c[2])));
}
void orc_shape_free(OrcShape* s) { delete s; }
int orc_shape_set_transform(OrcShape* s, const double* m) {
  int rc = set_transform(s->s, m_in(m));
  return rc == 0 ? ORC_OK : (rc == -2 ? ORC_ERR_TRANSFORM_TWICE : ORC_ERR_NOT_INVERTIBLE);
}
void orc_shape_set_material(OrcShape* s, const OrcMaterial* m) { set_material(s->s, mat_in(m)); }
void orc_shape_get_material(const OrcShape* s, OrcMaterial* m) { mat_out(s->s.material, m); }
int orc_shape_push(OrcShape* g, const OrcShape* child) {
  if (g->s.kind != Kind::Group) return ORC_ERR_NOT_GROUP;
  g->s.shapes.push_back(child->s);
  g->s.bounds_cached = false;
  return ORC_OK;
}
static int64_t leaf_count(const Shape& s) {
  if (s.kind != Kind::Group) return 1;
  int64_t n = 0;
  for (const auto& c : s.shapes) n += leaf_count(c);
  return n;
}
int64_t orc_shape_leaf_count(const OrcShape* s) { return leaf_count(s->s); }

OrcShape* orc_obj_parse(const char* text, size_t len, int64_t* ignored, int* err) {
  ObjResult r = parse_obj(text, len);
  if (!r.error.empty()) {
    if (err) *err = ORC_ERR_PARSE;
    return nullptr;
  }
  if (err) *err = ORC_OK;
  if (ignored) *ignored = r.ignored_lines;
  return wrap(std::move(r.group));
}

OrcWorld* orc_world_new(const double* lp, const double* li) {
  auto* w = new OrcWorld;
  w->w.light = {Tuple::point(lp[0], lp[1], lp[2]), Color{li[0], li[1], li[2]}};
  return w;
}
OrcWorld* orc_world_default(void) {
  auto* w = new OrcWorld;
  w->w = default_world();
  return w;
}
void orc_world_free(OrcWorld* w) { delete w; }
void orc_world_push(OrcWorld* w, const OrcShape* s) { w->w.objects.push_back(s->s); }

static int finish(const TraceCtx& ctx) {
  if (ctx.invert_failed) return ORC_ERR_NOT_INVERTIBLE;
  if (ctx.depth_underflow && ctx.runaway) return ORC_ERR_DEPTH_UNDERFLOW;
  return ORC_OK;
}

int orc_color_at(const OrcWorld* w, int64_t n, const double* o, const double* d, int limit, int mode, double* out,
                 int64_t* counts) {
  if (limit < 0) return ORC_ERR_ARG;
  TraceCtx ctx;
  ctx.faithful = (mode == ORC_MODE_FAITHFUL);
  if (!ctx.faithful)
    for (const auto& s : w->w.objects) warm_bounds(s);
  for (int64_t i = 0; i < n; ++i) {
    Ray r{Tuple::point(o[3 * i], o[3 * i + 1], o[3 * i + 2]), Tuple::vector(d[3 * i], d[3 * i + 1], d[3 * i + 2])};
    Color c = internal_color_at(w->w, r, (uint64_t)limit, ctx);
    out[3 * i] = c.red, out[3 * i + 1] = c.green, out[3 * i + 2] = c.blue;
  }
  if (counts) counts[0] = ctx.radiance_rays, counts[1] = ctx.shadow_rays;
  return finish(ctx);
}

OrcCamera* orc_camera_new(int64_t h, int64_t v, double fov) { return new OrcCamera{camera_new(h, v, fov)}; }
void orc_camera_free(OrcCamera* c) { delete c; }
int orc_camera_set_transform(OrcCamera* c, const double* m) {
  return camera_set_transform(c->c, m_in(m)) ? ORC_OK : ORC_ERR_NOT_INVERTIBLE;
}
void orc_camera_get(const OrcCamera* c, double* ps, double* hw, double* hh, double* tinv) {
  if (ps) *ps = c->c.pixel_size;
  if (hw) *hw = c->c.half_width;
  if (hh) *hh = c->c.half_height;
  if (tinv) m_out(c->c.transform_inverse, tinv);
}
void orc_camera_ray_for_pixel(const OrcCamera* c, int64_t px, int64_t py, double* o, double* d) {
  Ray r = ray_for_pixel(c->c, px, py);
  o[0] = r.origin.x, o[1] = r.origin.y, o[2] = r.origin.z;
  d[0] = r.direction.x, d[1] = r.direction.y, d[2] = r.direction.z;
}

int orc_render_pixels(const OrcCamera* c, const OrcWorld* w, int limit, int mode, int64_t n_pix, const int32_t* xs,
                      const int32_t* ys, int nthreads, double* out, int64_t* counts, double* seconds) {
  if (limit < 0) return ORC_ERR_ARG;
  const Camera& cam = c->c;
  const bool all = n_pix < 0;
  const int64_t n = all ? cam.hsize * cam.vsize : n_pix;
  const bool faithful = (mode == ORC_MODE_FAITHFUL);
  if (!faithful)
    for (const auto& s : w->w.objects) warm_bounds(s);
  if (nthreads < 1) nthreads = 1;
  std::vector<TraceCtx> ctxs((size_t)nthreads);
  auto work = [&](int tid) {
    TraceCtx& ctx = ctxs[(size_t)tid];
    ctx.faithful = faithful;
    // camera.rs:70-76 — y outer, x inner; here over the requested pixel list (interleaved across threads)
    for (int64_t i = tid; i < n; i += nthreads) {
      int64_t x = all ? (i % cam.hsize) : xs[i];
      int64_t y = all ? (i / cam.hsize) : ys[i];
      Ray ray = ray_for_pixel(cam, x, y);
      Color col = internal_color_at(w->w, ray, (uint64_t)limit, ctx);  // World::color_at with explicit limit
      out[3 * i] = col.red, out[3 * i + 1] = col.green, out[3 * i + 2] = col.blue;
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  if (nthreads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  int rc = ORC_OK;
  int64_t rr = 0, sr = 0;
  for (auto& ctx : ctxs) {
    rr += ctx.radiance_rays, sr += ctx.shadow_rays;
    int f = finish(ctx);
    if (f != ORC_OK) rc = f;
  }
  if (counts) counts[0] = rr, counts[1] = sr;
  return rc;
}

size_t orc_canvas_to_ppm(int64_t width, int64_t height, const double* rgb, char* out, size_t cap) {
  std::string s = canvas_to_ppm(width, height, rgb);
  if (out && cap) std::memcpy(out, s.data(), std::min(cap, s.size()));
  return s.size();
}

}  // extern "C"
