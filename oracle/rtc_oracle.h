/*
 * rtc_oracle.h — flat C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the *checker*: a single-threaded
 * IEEE-f64 restatement of the reference renderer's per-pixel path
 * (camera.rs / world.rs / intersection.rs / shape.rs / bounds.rs / material.rs /
 * pattern.rs / tuple.rs / matrix.rs / transformations.rs / canvas.rs /
 * obj_file.rs of antoinehebert/ray-tracer-challenge-rust).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (ray-tracer-challenge-rust_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED by the reference's own unit-test vectors (the KAT
 * suite in oracle/kat.cpp restates the goldens listed in SURVEY.md §8c).
 * The reference itself cannot be built here (Rust toolchain absent), so there
 * is no oracle/_ref.
 *
 * All matrices are double[16], row-major.  Points/vectors are double[3].
 * Shapes have value semantics, like the reference: pushing a shape into a
 * group or a world stores a deep copy; the handle stays valid and owned by
 * the caller (free with orc_shape_free).
 */
#ifndef RTC_ORACLE_H
#define RTC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order as reference Material (material.rs:4-14) + flattened Pattern. */
typedef struct {
  double color[3];
  double ambient, diffuse, specular, shininess, reflective;
  double transparency, refractive_index;
  int32_t pattern_kind; /* 0 none, 1 stripe, 2 gradient, 3 ring, 4 checkers, 5 test */
  int32_t _pad;
  double pattern_a[3], pattern_b[3];
  double pattern_transform[16];
} OrcMaterial;

typedef struct OrcShape OrcShape;
typedef struct OrcWorld OrcWorld;
typedef struct OrcCamera OrcCamera;

enum { ORC_MODE_CACHED = 0, ORC_MODE_FAITHFUL = 1 };

/* error codes (negative) */
enum {
  ORC_OK = 0,
  ORC_ERR_NOT_INVERTIBLE = -1,   /* Matrix::inverse -> None -> expect() panic */
  ORC_ERR_TRANSFORM_TWICE = -2,  /* shape.rs:199-201 */
  ORC_ERR_NOT_GROUP = -3,        /* shape.rs:533 */
  ORC_ERR_PARSE = -4,            /* obj_file.rs panics */
  ORC_ERR_DEPTH_UNDERFLOW = -5,  /* world.rs:68 usize underflow (L = 1 mod 3) reached */
  ORC_ERR_ARG = -6
};

/* ---- matrix.rs / transformations.rs ---- */
void orc_identity(double* m);
void orc_translation(double x, double y, double z, double* m);
void orc_scaling(double x, double y, double z, double* m);
void orc_rotation_x(double rad, double* m);
void orc_rotation_y(double rad, double* m);
void orc_rotation_z(double rad, double* m);
void orc_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double* m);
void orc_view_transform(const double* from, const double* to, const double* up, double* m);
void orc_mat_mul(const double* a, const double* b, double* out);
int orc_mat_inverse(const double* a, double* out);

/* ---- material.rs defaults ---- */
void orc_material_default(OrcMaterial* m);

/* ---- shape.rs ---- */
OrcShape* orc_shape_sphere(void);
OrcShape* orc_shape_glass_sphere(void);
OrcShape* orc_shape_plane(void);
OrcShape* orc_shape_cube(void);
OrcShape* orc_shape_cylinder(double minimum, double maximum, int capped);
OrcShape* orc_shape_cone(double minimum, double maximum, int capped);
OrcShape* orc_shape_group(void);
OrcShape* orc_shape_triangle(const double* p1, const double* p2, const double* p3);
void orc_shape_free(OrcShape* s);
int orc_shape_set_transform(OrcShape* s, const double* m);
void orc_shape_set_material(OrcShape* s, const OrcMaterial* m);
void orc_shape_get_material(const OrcShape* s, OrcMaterial* m);
int orc_shape_push(OrcShape* group, const OrcShape* child);
int64_t orc_shape_leaf_count(const OrcShape* s);

/* ---- obj_file.rs ---- */
OrcShape* orc_obj_parse(const char* text, size_t len, int64_t* ignored_lines, int* err);

/* ---- world.rs ---- */
OrcWorld* orc_world_new(const double* light_pos, const double* light_intensity);
OrcWorld* orc_world_default(void);
void orc_world_free(OrcWorld* w);
void orc_world_push(OrcWorld* w, const OrcShape* s);
/* World::color_at generalised to an explicit RECURSION_LIMIT (world.rs:11). out = 3 doubles per ray. */
int orc_color_at(const OrcWorld* w, int64_t n, const double* origins, const double* dirs,
                 int recursion_limit, int mode, double* out_rgb, int64_t* ray_counts /*[2] radiance, shadow; may be NULL*/);

/* ---- camera.rs ---- */
OrcCamera* orc_camera_new(int64_t hsize, int64_t vsize, double fov);
void orc_camera_free(OrcCamera* c);
int orc_camera_set_transform(OrcCamera* c, const double* m);
void orc_camera_get(const OrcCamera* c, double* pixel_size, double* half_width, double* half_height,
                    double* transform_inverse);
void orc_camera_ray_for_pixel(const OrcCamera* c, int64_t px, int64_t py, double* origin, double* dir);
/* Camera::render restricted to the given pixels (all pixels when n_pix < 0 : row-major, out has hsize*vsize*3).
 * nthreads > 1 splits the pixel list across std::threads (cached mode goldens only; baseline timing uses 1). */
int orc_render_pixels(const OrcCamera* c, const OrcWorld* w, int recursion_limit, int mode, int64_t n_pix,
                      const int32_t* xs, const int32_t* ys, int nthreads, double* out_rgb,
                      int64_t* ray_counts /*[2]*/, double* seconds);

/* ---- canvas.rs ---- */
/* P3 text as Canvas::to_ppm writes it.  Returns bytes needed; writes at most cap bytes. */
size_t orc_canvas_to_ppm(int64_t width, int64_t height, const double* rgb, char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
