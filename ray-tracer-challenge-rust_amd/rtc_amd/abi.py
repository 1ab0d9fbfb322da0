"""ctypes mirror of include/rtc_hip.h (the C ABI of csrc/librtc_hip.so)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
HIP_LIB_PATH = os.environ.get("RTC_HIP_LIB", os.path.join(PKG, "csrc", "librtc_hip.so"))
HOST_LIB_PATH = os.path.join(PKG, "host", "librtc_host.so")

RTC_OK = 0
RTC_ERR_INVALID_ARG = -1
RTC_ERR_NO_DEVICE = -2
RTC_ERR_HIP = -3
RTC_ERR_NON_AFFINE = -4
RTC_ERR_DEPTH_UNDERFLOW = -5
RTC_ERR_UNSUPPORTED = -6
RTC_ERR_TOO_MANY_GROUPS = -7
RTC_ERR_COLLECTIVE = -8
RTC_ERR_RAY_POOL = -9

RTC_FLAG_TIMING = 1
RTC_FLAG_COUNT_RAYS = 2
RTC_FLAG_EXACT_SWEEP = 4
RTC_FLAG_GATHER_P2P = 8
RTC_ABI_VERSION = 3

# every symbol include/rtc_hip.h declares
ABI_SYMBOLS = [
    "rtc_abi_version", "rtc_device_count", "rtc_last_error_string", "rtc_local_rows", "rtc_scene_create",
    "rtc_scene_destroy", "rtc_render_device", "rtc_scene_stats", "rtc_scene_reset_counters", "rtc_render",
    "rtc_color_at", "rtc_selftest_math", "rtc_multi_create", "rtc_multi_destroy", "rtc_multi_render", "rtc_render_multi",
    "rtc_release_cached",
]


class RtcPattern(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("a", C.c_double * 3), ("b", C.c_double * 3),
                ("transform_inverse", C.c_double * 16)]


class RtcMaterial(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double), ("pattern", C.c_int32), ("_pad", C.c_int32)]


class RtcSceneDesc(C.Structure):
    _fields_ = [
        ("light_position", C.c_double * 3), ("light_intensity", C.c_double * 3),
        ("n_transforms", C.c_uint32), ("transform_inverse", C.POINTER(C.c_double)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(RtcMaterial)),
        ("n_patterns", C.c_uint32), ("patterns", C.POINTER(RtcPattern)),
        ("n_groups", C.c_uint32), ("group_parent", C.POINTER(C.c_int32)),
        ("group_bounds_min", C.POINTER(C.c_double)), ("group_bounds_max", C.POINTER(C.c_double)),
        ("n_leaves", C.c_uint32), ("leaf_kind", C.POINTER(C.c_uint8)), ("leaf_transform", C.POINTER(C.c_uint32)),
        ("leaf_material", C.POINTER(C.c_uint32)), ("leaf_group", C.POINTER(C.c_int32)),
        ("leaf_class", C.POINTER(C.c_uint32)), ("leaf_params", C.POINTER(C.c_double)),
        ("leaf_triangle", C.POINTER(C.c_uint32)),
        ("n_triangles", C.c_uint32), ("tri_p1", C.POINTER(C.c_double)), ("tri_e1", C.POINTER(C.c_double)),
        ("tri_e2", C.POINTER(C.c_double)), ("tri_normal", C.POINTER(C.c_double)),
    ]


class RtcCameraDesc(C.Structure):
    _fields_ = [("hsize", C.c_uint32), ("vsize", C.c_uint32), ("half_width", C.c_double), ("half_height", C.c_double),
                ("pixel_size", C.c_double), ("transform_inverse", C.c_double * 16)]


class RtcRenderOpts(C.Structure):
    _fields_ = [("recursion_limit", C.c_int32), ("block_rows", C.c_uint32), ("block_first", C.c_uint32),
                ("block_step", C.c_uint32), ("flags", C.c_uint32), ("block_count", C.c_uint32)]


class RtcStats(C.Structure):
    _fields_ = [("radiance_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("triangle_tests", C.c_uint64),
                ("primitive_tests", C.c_uint64), ("filter_tests", C.c_uint64), ("box_tests", C.c_uint64), ("culled_rays", C.c_uint64), ("unswept_shadow_rays", C.c_uint64),
                ("kernel_ms", C.c_double), ("upload_ms", C.c_double), ("download_ms", C.c_double)]


def load_hip_library(path: str | None = None) -> C.CDLL:
    """Load csrc/librtc_hip.so and declare its prototypes.  Raises if the library is missing: there is no
    fallback path in this package."""
    path = path or HIP_LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python ray-tracer-challenge-rust_amd/build.py` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
    lib.rtc_abi_version.restype = C.c_int
    lib.rtc_device_count.restype = C.c_int
    lib.rtc_last_error_string.restype = C.c_char_p
    lib.rtc_local_rows.restype = C.c_uint32
    lib.rtc_local_rows.argtypes = [C.c_uint32, C.POINTER(RtcRenderOpts)]
    lib.rtc_scene_create.restype = C.c_int
    lib.rtc_scene_create.argtypes = [C.POINTER(RtcSceneDesc), C.c_int, C.POINTER(vp)]
    lib.rtc_scene_destroy.restype = None
    lib.rtc_scene_destroy.argtypes = [vp]
    lib.rtc_render_device.restype = C.c_int
    lib.rtc_render_device.argtypes = [vp, C.POINTER(RtcCameraDesc), C.POINTER(RtcRenderOpts), vp, vp, vp]
    lib.rtc_scene_stats.restype = C.c_int
    lib.rtc_scene_stats.argtypes = [vp, C.POINTER(RtcStats)]
    lib.rtc_scene_reset_counters.restype = C.c_int
    lib.rtc_scene_reset_counters.argtypes = [vp]
    lib.rtc_render.restype = C.c_int
    lib.rtc_render.argtypes = [C.POINTER(RtcSceneDesc), C.POINTER(RtcCameraDesc), C.POINTER(RtcRenderOpts), C.c_int,
                               fp, dp, C.POINTER(RtcStats)]
    lib.rtc_color_at.restype = C.c_int
    lib.rtc_color_at.argtypes = [vp, C.c_uint64, dp, dp, C.POINTER(RtcRenderOpts), fp, dp]
    ip = C.POINTER(C.c_int)
    lib.rtc_multi_create.restype = C.c_int
    lib.rtc_multi_create.argtypes = [C.POINTER(RtcSceneDesc), C.c_int, ip, C.POINTER(vp)]
    lib.rtc_multi_destroy.restype = None
    lib.rtc_multi_destroy.argtypes = [vp]
    lib.rtc_multi_render.restype = C.c_int
    lib.rtc_multi_render.argtypes = [vp, C.POINTER(RtcCameraDesc), C.POINTER(RtcRenderOpts), fp, dp, C.POINTER(RtcStats)]
    lib.rtc_render_multi.restype = C.c_int
    lib.rtc_render_multi.argtypes = [C.POINTER(RtcSceneDesc), C.POINTER(RtcCameraDesc), C.POINTER(RtcRenderOpts), C.c_int, ip,
                                     fp, dp, C.POINTER(RtcStats)]
    lib.rtc_release_cached.restype = None
    lib.rtc_release_cached.argtypes = []
    lib.rtc_selftest_math.restype = C.c_int
    lib.rtc_selftest_math.argtypes = [C.c_int, C.c_int, C.c_uint64, dp, dp, dp]
    return lib
