/*
 * rtc_host_capi.h — flat C view of the C++ host library (rtc_host.hpp) for language bindings
 * (the Python package rtc_amd binds it with ctypes; a Rust crate would bind it the same way).
 *
 * Naming follows the reference's scene API (Shape::sphere, set_transform, World::new, Camera::new,
 * Parser::from_obj_str ...).  Shapes have value semantics as in the reference: pushing a shape into a
 * group or a world stores a copy; the caller keeps and frees its handle.  Matrices are double[16]
 * row-major, points/vectors double[3].  Functions that can fail return 0 or a negative code (host codes
 * -101..-106, or an RtcStatus from the device library); rtch_last_error() has the message.
 */
#ifndef RTC_HOST_CAPI_H
#define RTC_HOST_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RtchMaterial { /* Material (material.rs:4-14) with its optional Pattern flattened */
  double color[3];
  double ambient, diffuse, specular, shininess, reflective;
  double transparency, refractive_index;
  int32_t pattern_kind; /* 0 = None, else RtcPatternKind */
  int32_t _pad;
  double pattern_a[3], pattern_b[3];
  double pattern_transform[16]; /* Pattern::transform (the inverse is derived) */
} RtchMaterial;

typedef struct RtchShape RtchShape;
typedef struct RtchWorld RtchWorld;
typedef struct RtchCamera RtchCamera;

typedef struct RtchStats {
  uint64_t radiance_rays, shadow_rays;
  uint64_t triangle_tests, primitive_tests, filter_tests, box_tests, culled_rays, unswept_shadow_rays; /* as RtcStats (include/rtc_hip.h) */
  double kernel_ms, upload_ms, download_ms;
} RtchStats;

typedef struct RtchFlatInfo { /* sizes of the flattened scene, for tests and tools */
  uint32_t n_transforms, n_materials, n_patterns, n_groups, n_leaves, n_triangles, n_classes_shared;
} RtchFlatInfo;

const char* rtch_last_error(void);

/* transformations.rs / matrix.rs */
void rtch_identity(double* m);
void rtch_translation(double x, double y, double z, double* m);
void rtch_scaling(double x, double y, double z, double* m);
void rtch_rotation_x(double rad, double* m);
void rtch_rotation_y(double rad, double* m);
void rtch_rotation_z(double rad, double* m);
void rtch_shearing(double xy, double xz, double yx, double yz, double zx, double zy, double* m);
void rtch_view_transform(const double* from, const double* to, const double* up, double* m);
void rtch_mat_mul(const double* a, const double* b, double* out);
int rtch_mat_inverse(const double* a, double* out);

void rtch_material_default(RtchMaterial* m);

/* shape.rs */
RtchShape* rtch_shape_sphere(void);
RtchShape* rtch_shape_glass_sphere(void);
RtchShape* rtch_shape_plane(void);
RtchShape* rtch_shape_cube(void);
RtchShape* rtch_shape_cylinder(double minimum, double maximum, int capped);
RtchShape* rtch_shape_cone(double minimum, double maximum, int capped);
RtchShape* rtch_shape_group(void);
RtchShape* rtch_shape_triangle(const double* p1, const double* p2, const double* p3);
void rtch_shape_free(RtchShape* s);
int rtch_shape_set_transform(RtchShape* s, const double* m);
void rtch_shape_set_material(RtchShape* s, const RtchMaterial* m);
void rtch_shape_get_material(const RtchShape* s, RtchMaterial* m);
int rtch_shape_push(RtchShape* group, const RtchShape* child);
int64_t rtch_shape_leaf_count(const RtchShape* s);
/* Bounds::new (bounds.rs:11) */
void rtch_shape_bounds(const RtchShape* s, double* mn, double* mx);

/* obj_file.rs: Parser::from_obj_str(text).obj_to_group() */
RtchShape* rtch_obj_parse(const char* text, size_t len, int64_t* ignored_lines, int* err);

/* world.rs */
RtchWorld* rtch_world_new(const double* light_pos, const double* light_intensity);
RtchWorld* rtch_world_default(void);
void rtch_world_free(RtchWorld* w);
void rtch_world_push(RtchWorld* w, const RtchShape* s);
int rtch_world_flat_info(const RtchWorld* w, RtchFlatInfo* info);
/* World::color_at on the GPU (through rtc_color_at).  out_rgb64 / out_rgb32: [n][3], either may be NULL.
 * recursion_limit: RECURSION_LIMIT, negative = the world's own (5, world.rs:11); flags: RTC_FLAG_* for
 * RtcRenderOpts.flags (also rtch_render). */
int rtch_color_at(const RtchWorld* w, int64_t n, const double* origins, const double* dirs, int recursion_limit,
                  uint32_t flags, int device, double* out_rgb64, float* out_rgb32);

/* camera.rs */
RtchCamera* rtch_camera_new(int64_t hsize, int64_t vsize, double fov);
void rtch_camera_free(RtchCamera* c);
int rtch_camera_set_transform(RtchCamera* c, const double* m);
void rtch_camera_get(const RtchCamera* c, double* pixel_size, double* half_width, double* half_height,
                     double* transform_inverse);
void rtch_camera_ray_for_pixel(const RtchCamera* c, int64_t px, int64_t py, double* origin, double* dir);
/* Camera::render on the GPU (through rtc_render).  out_rgb32: hsize*vsize*3 floats (required);
 * out_rgb64 optional; stats optional: NULL selects the uncounted, untimed kernel (the counting one is several
 * times slower). */
int rtch_render(const RtchCamera* c, const RtchWorld* w, int recursion_limit, uint32_t flags, int device,
                float* out_rgb32, double* out_rgb64, RtchStats* stats);

/* The same frame rendered by several devices of the node (through rtc_render_multi): interleaved 16-row blocks,
 * gathered to devices[0] over xGMI (RCCL; RTC_FLAG_GATHER_P2P in flags selects peer DMA).  Bit-identical to rtch_render. */
int rtch_render_multi(const RtchCamera* c, const RtchWorld* w, int recursion_limit, uint32_t flags, int n_devices,
                      const int* devices, float* out_rgb32, double* out_rgb64, RtchStats* stats);

/* Flatten + upload once (rtc_scene_create); the returned RtcScene* is used with rtc_render_device /
 * rtc_color_at / rtc_scene_stats of librtc_hip.so directly and released with rtc_scene_destroy. */
int rtch_world_scene_create(const RtchWorld* w, int device, void** out_scene);
/* The RtcCameraDesc (include/rtc_hip.h) of this camera; out points to an RtcCameraDesc. */
void rtch_camera_desc(const RtchCamera* c, void* out_rtc_camera_desc);

/* canvas.rs: Canvas::to_ppm.  Returns the byte count; writes at most cap bytes. */
size_t rtch_canvas_to_ppm(int64_t width, int64_t height, const double* rgb, char* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
