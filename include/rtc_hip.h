/*
 * rtc_hip.h — C ABI of the MI355X (gfx950) renderer for the per-pixel hot path of
 * antoinehebert/ray-tracer-challenge-rust.
 *
 * The reference has no FFI of its own (zero dependencies, no `extern`): the seam is the public
 * method `Camera::render(&self, world: &World) -> Canvas` (camera.rs:67-80) and, below it,
 * `World::color_at(&self, &Ray) -> Color` (world.rs:80-82).  This header is what a Rust host would
 * bind with an `extern "C"` block to replace those two calls (INTEGRATION.md shows the binding).
 *
 * Everything here is plain C: pointers, sizes and POD structs.  All scene pointers are borrowed for
 * the duration of the call only (like `&World`); nothing is retained after rtc_scene_create returns.
 * No function unwinds across the boundary: where the reference panics, a negative RtcStatus comes back.
 *
 * Numerics: geometry, intersection predicates, hit selection, shading and recursion are evaluated in
 * IEEE f64 in the reference's own operation order (no FMA contraction); only the framebuffer store
 * narrows to f32.  An optional f64 framebuffer is available for parity checks.
 */
#ifndef RTC_HIP_H
#define RTC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTC_ABI_VERSION 3 /* 2: RtcStats.culled_rays / unswept_shadow_rays, RtcRenderOpts.block_count, rtc_multi_*, rtc_render_multi, rtc_release_cached
                             3: RTC_ERR_RAY_POOL */

typedef enum RtcStatus {
  RTC_OK = 0,
  RTC_ERR_INVALID_ARG = -1,     /* null pointer, index out of range, zero-sized image ... */
  RTC_ERR_NO_DEVICE = -2,       /* no HIP device / device index out of range */
  RTC_ERR_HIP = -3,             /* a HIP runtime call failed; see rtc_last_error_string() */
  RTC_ERR_NON_AFFINE = -4,      /* a transform_inverse whose bottom row is not exactly (0,0,0,1) */
  RTC_ERR_DEPTH_UNDERFLOW = -5, /* recursion_limit = 1 (mod 3) with a reflective/transparent material:
                                   the reference evaluates `remaining - 1` on usize 0 (world.rs:68): panic in
                                   debug builds, unbounded recursion in release builds */
  RTC_ERR_UNSUPPORTED = -6,     /* scene uses a feature outside this build (see message) */
  RTC_ERR_TOO_MANY_GROUPS = -7, /* a leaf nested under more than RTC_MAX_GROUP_DEPTH groups */
  RTC_ERR_COLLECTIVE = -8,      /* RCCL could not be loaded or a collective call failed (rtc_multi_*) */
  RTC_ERR_RAY_POOL = -9         /* rtc_scene_stats after rtc_render_device only: a wave of the last launch had more rays parked
                                   than its pool holds (deep recursion through materials that are reflective AND transparent);
                                   that frame is incomplete, the scene's pools have been doubled: launch the same frame again.
                                   rtc_render, rtc_render_multi, rtc_multi_render and rtc_color_at do that themselves and never
                                   return this code */
} RtcStatus;

#define RTC_MAX_GROUP_DEPTH 64 /* nesting depth; the number of groups is not limited */
#define RTC_MAX_RECURSION_LIMIT 24

/* ShapeKind (shape.rs:14-39).  Groups are not leaves; they appear in the group table. */
typedef enum RtcShapeKind {
  RTC_SPHERE = 0,
  RTC_PLANE = 1,
  RTC_CUBE = 2,
  RTC_CYLINDER = 3,
  RTC_CONE = 4,
  RTC_TRIANGLE = 5
} RtcShapeKind;

/* PatternKind (pattern.rs:4-12) */
typedef enum RtcPatternKind {
  RTC_PATTERN_STRIPE = 1,
  RTC_PATTERN_GRADIENT = 2,
  RTC_PATTERN_RING = 3,
  RTC_PATTERN_CHECKERS = 4,
  RTC_PATTERN_TEST = 5
} RtcPatternKind;

/* Pattern (pattern.rs:15-19): the two colours, the kind and the cached transform_inverse. */
typedef struct RtcPattern {
  int32_t kind; /* RtcPatternKind */
  int32_t _pad;
  double a[3];
  double b[3];
  double transform_inverse[16]; /* row-major, Pattern::transform_inverse */
} RtcPattern;

/* Material (material.rs:4-14) */
typedef struct RtcMaterial {
  double color[3];
  double ambient, diffuse, specular, shininess, reflective;
  double transparency, refractive_index;
  int32_t pattern; /* index into RtcSceneDesc.patterns, -1 = None */
  int32_t _pad;
} RtcMaterial;

/*
 * A flattened `World` (world.rs:13-16).  `Shape::set_transform` pushes group transforms down to the
 * leaves (shape.rs:207-218), so the world is a list of leaves in depth-first order, each with its own
 * world-from-object matrix; groups keep only their place in the tree and their bounding box.
 */
typedef struct RtcSceneDesc {
  /* Light (light.rs:5-8) */
  double light_position[3];
  double light_intensity[3];

  /* distinct Shape::transform_inverse matrices (shape.rs:45), row-major 4x4 */
  uint32_t n_transforms;
  const double* transform_inverse; /* [n_transforms][16] */

  uint32_t n_materials;
  const RtcMaterial* materials;
  uint32_t n_patterns;
  const RtcPattern* patterns;

  /* Groups in depth-first pre-order (a parent precedes its children; -1 = child of the world).
   * bounds = Bounds::new(group) exactly as bounds.rs:50-125 computes it (the host flattener does). */
  uint32_t n_groups;
  const int32_t* group_parent;    /* [n_groups] */
  const double* group_bounds_min; /* [n_groups][3] */
  const double* group_bounds_max; /* [n_groups][3] */

  /* Leaves in depth-first order: world.objects order, then ShapeKind::Group::shapes order. */
  uint32_t n_leaves;
  const uint8_t* leaf_kind;       /* [n_leaves] RtcShapeKind */
  const uint32_t* leaf_transform; /* [n_leaves] index into transform_inverse */
  const uint32_t* leaf_material;  /* [n_leaves] index into materials */
  const int32_t* leaf_group;      /* [n_leaves] innermost enclosing group, -1 = top-level object */
  const uint32_t* leaf_class;     /* [n_leaves] or NULL.  Index of the first leaf that is `==` to this one under
                                     the reference's value equality (shape.rs:638-646); prepare_computations'
                                     container list compares shapes that way (intersection.rs:42).  NULL means
                                     every leaf is its own class. */
  const double* leaf_params;      /* [n_leaves][3] Cylinder/Cone: minimum, maximum, capped (0/1); else ignored */
  const uint32_t* leaf_triangle;  /* [n_leaves] index into the triangle arrays for RTC_TRIANGLE; else ignored */

  /* ShapeKind::Triangle payload (shape.rs:31-38), object space */
  uint32_t n_triangles;
  const double* tri_p1;     /* [n_triangles][3] */
  const double* tri_e1;     /* [n_triangles][3] p2 - p1 */
  const double* tri_e2;     /* [n_triangles][3] p3 - p1 */
  const double* tri_normal; /* [n_triangles][3] normalize(e2 x e1) */
} RtcSceneDesc;

/* Camera (camera.rs:4-13) after Camera::new + set_transform */
typedef struct RtcCameraDesc {
  uint32_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double transform_inverse[16]; /* row-major */
} RtcCameraDesc;

typedef struct RtcRenderOpts {
  int32_t recursion_limit; /* RECURSION_LIMIT (world.rs:11) with the reference's semantics: ceil(L/3) ray
                              generations are shaded; the reference hard-codes 5 */
  /* Row sharding for one-process-per-GPU rendering: the image is cut into blocks of `block_rows` rows; this
   * call renders blocks block_first, block_first+block_step, ... and stores them densely, in that order, in the
   * output buffer (rtc_local_rows() rows).  block_rows = 0 means the whole image (first 0, step 1). */
  uint32_t block_rows;
  uint32_t block_first;
  uint32_t block_step;
  uint32_t flags; /* RTC_FLAG_* */
  uint32_t block_count; /* with block_rows != 0: render only the first block_count of those blocks (0 = all of them),
                           so that a shard can be rendered, and sent on its way, in pieces */
} RtcRenderOpts;

#define RTC_FLAG_TIMING 1u    /* record HIP events around the kernel on the launch stream */
#define RTC_FLAG_COUNT_RAYS 2u /* maintain device-side ray / test counters (atomics; not for timed launches) */
#define RTC_FLAG_EXACT_SWEEP 4u /* brute force: every ray tests every leaf with the exact f64 arithmetic, triangle
                                   runs staged through LDS.  Default is the accelerated sweep (spatial clusters +
                                   conservative f32 filters in front of the same exact test): identical results. */
#define RTC_FLAG_GATHER_P2P 8u  /* rtc_multi_render: move the row blocks to the root device with hipMemcpyPeerAsync
                                   instead of RCCL send/recv (same bytes over the same xGMI links, no communicator) */

typedef struct RtcStats {
  uint64_t radiance_rays; /* World::intersect calls made from internal_color_at (world.rs:89) */
  uint64_t shadow_rays;   /* World::intersect calls made from is_shadowed (world.rs:107) */
  uint64_t triangle_tests;  /* exact f64 Shape::intersect evaluations on Triangle leaves (shape.rs:438-459) */
  uint64_t primitive_tests; /* Shape::intersect evaluations on non-triangle leaves */
  uint64_t filter_tests;    /* conservative f32 triangle filter evaluations (accelerated sweep only) */
  uint64_t box_tests;       /* conservative f32 tile/cluster box tests (accelerated sweep only) */
  uint64_t culled_rays;     /* of radiance_rays: primary rays answered by the wave-level background exit (all 64
                               rays of a wave miss every top-level group box, shape.rs:399-425) without a sweep */
  uint64_t unswept_shadow_rays; /* of shadow_rays: cast by the reference, but Material::lighting ignores their answer
                               (the hit faces away from the light, material.rs:57-60), so no sweep was run for them */
  double kernel_ms;       /* HIP-event time of the last render kernel (RTC_FLAG_TIMING), else 0 */
  double upload_ms;       /* host wall time of the scene upload in rtc_render / rtc_scene_create */
  double download_ms;     /* host wall time of the framebuffer copy in rtc_render */
} RtcStats;

typedef struct RtcScene RtcScene; /* device-resident flattened scene */

int rtc_abi_version(void);
int rtc_device_count(void);
const char* rtc_last_error_string(void); /* thread-local, never NULL */

/* Number of rows a call with these opts stores (the height of its dense output buffer). */
uint32_t rtc_local_rows(uint32_t vsize, const RtcRenderOpts* opts);

/* Validate + upload a scene to `device`.  Replaces the per-frame borrow of `&World`. */
RtcStatus rtc_scene_create(const RtcSceneDesc* desc, int device, RtcScene** out);
void rtc_scene_destroy(RtcScene* scene);

/*
 * Camera::render (camera.rs:67-80) on the device, asynchronously on `hip_stream` (a hipStream_t, NULL = default
 * stream).  d_rgb: device pointer, rtc_local_rows()*hsize*3 floats, pixel-major, row-major, top row first,
 * RGB interleaved — the Canvas layout (canvas.rs:20-26).  d_rgb64: optional device pointer to the same
 * layout in f64 (NULL to skip).  Colours are raw and unclamped, like Canvas pixels before to_ppm.
 */
RtcStatus rtc_render_device(RtcScene* scene, const RtcCameraDesc* camera, const RtcRenderOpts* opts, float* d_rgb,
                            double* d_rgb64, void* hip_stream);

/* Blocks until the scene's stream work is done and returns counters/timing of the last launch. */
RtcStatus rtc_scene_stats(RtcScene* scene, RtcStats* stats);
RtcStatus rtc_scene_reset_counters(RtcScene* scene);

/*
 * Camera::render with host buffers: upload, render, copy back.  h_rgb: rtc_local_rows()*hsize*3 floats;
 * h_rgb64: as many doubles; either may be NULL (not both).  stats optional: when it is NULL (and RTC_FLAG_COUNT_RAYS is not set) the uncounted kernel runs,
 * which is several times faster than the counting one.
 * The device copy of the scene is kept after the call, keyed by a hash of everything `desc` points to (plus a second,
 * sampled key that is compared word for word on a hash match: a hash collision alone never selects another scene), so the next
 * call with an identical scene (Camera::render borrows the same `&World` every frame, camera.rs:67) skips the
 * upload and the acceleration build; the framebuffer comes back through a pinned staging buffer.  The cache holds
 * a few scenes per device; rtc_release_cached() empties it.
 */
RtcStatus rtc_render(const RtcSceneDesc* desc, const RtcCameraDesc* camera, const RtcRenderOpts* opts, int device,
                     float* h_rgb, double* h_rgb64, RtcStats* stats);

/*
 * Camera::render across several devices of one node, in one process: the scene is replicated, the
 * image is cut into blocks of 16 rows dealt round-robin to the devices, every device renders its blocks in a few
 * chunks, and each finished chunk travels to devices[0] over xGMI while the next one renders: grouped
 * ncclSend / ncclRecv on an in-process RCCL communicator (ncclCommInitAll; librccl is loaded on first use), or
 * hipMemcpyPeerAsync with RTC_FLAG_GATHER_P2P.  devices[0] puts the blocks back in Canvas order and the frame is
 * copied to the host buffers.  The result is bit-identical to a single-device render.
 * The RCCL hop is UNVERIFIED ON HARDWARE: the boxes this library was developed on have one GPU and RCCL refuses a
 * communicator with one device listed twice; everything else on the path has run through the peer-DMA transport (one device
 * listed several times).  A failure inside the exchange closes the open RCCL group and waits for every stream of the call
 * before returning, so nothing stays in flight and the next call starts clean.
 * rtc_multi_create replicates the scene once (replaces the borrow of `&World` for many frames); rtc_render_multi is
 * the one-call form and keeps its replicas for the next call with an identical scene (see rtc_render).
 * opts->block_rows / block_first / block_step must be 0 here.  stats (optional): counters summed over the devices,
 * kernel_ms = the longest device's render time, download_ms = gather + copy to the host.
 */
typedef struct RtcMulti RtcMulti;
RtcStatus rtc_multi_create(const RtcSceneDesc* desc, int n_devices, const int* devices, RtcMulti** out);
void rtc_multi_destroy(RtcMulti* multi);
RtcStatus rtc_multi_render(RtcMulti* multi, const RtcCameraDesc* camera, const RtcRenderOpts* opts, float* h_rgb,
                           double* h_rgb64, RtcStats* stats);
RtcStatus rtc_render_multi(const RtcSceneDesc* desc, const RtcCameraDesc* camera, const RtcRenderOpts* opts,
                           int n_devices, const int* devices, float* h_rgb, double* h_rgb64, RtcStats* stats);

/* Frees what rtc_render / rtc_render_multi retain between calls (device scenes keyed by their content, pinned
 * staging buffers, RCCL communicators).  Never required for correctness. */
void rtc_release_cached(void);

/*
 * World::color_at (world.rs:80-82) for a batch of rays, with an explicit recursion limit.
 * origins/directions: host pointers, [n][3] doubles.  h_rgb / h_rgb64: host pointers [n][3], either may be NULL.
 */
RtcStatus rtc_color_at(RtcScene* scene, uint64_t n, const double* origins, const double* directions,
                       const RtcRenderOpts* opts, float* h_rgb, double* h_rgb64);

/*
 * Device arithmetic self-check used by the parity tests: evaluates op on n f64 operand pairs on the GPU.
 * op: 0 = a/b, 1 = sqrt(a), 2 = pow(a,b), 3 = floor(a), 4 = fmod(a,b); 5 / 6 = minimum / maximum of (float)a over
 * each aligned run of 64 elements (the wave reductions of the ray-bundle construction).  Host pointers.
 */
RtcStatus rtc_selftest_math(int device, int op, uint64_t n, const double* a, const double* b, double* out);

#ifdef __cplusplus
}
#endif
#endif /* RTC_HIP_H */
