"""rtc_amd — Python binding of the MI355X-native renderer.

The scene API mirrors the reference crate (Shape::sphere, set_transform, World::new, Camera::new,
Camera::render, World::color_at, Parser::from_obj_str, Canvas::to_ppm ...).  Everything numeric happens in
the native libraries:

  host/librtc_host.so   C++ scene model + flattener (this module binds its flat C API with ctypes)
  csrc/librtc_hip.so    gfx950 kernels behind the C ABI of include/rtc_hip.h

Importing this package fails loudly if the libraries are not built; rendering fails loudly if there is no
GPU.  There is no CPU rendering path in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

# PyTorch wheels bundle their own libamdhip64.so.7.  If torch and librtc_hip.so end up with two different HIP
# runtimes in one process the second one cannot initialise the GPU, so when torch is available it is imported
# FIRST: the dynamic loader then resolves librtc_hip.so's libamdhip64.so.7 to the copy torch already mapped.
# (A host that never uses torch - e.g. a Rust binary - simply gets the system ROCm runtime.)
if "torch" not in sys.modules and not os.environ.get("RTC_AMD_NO_TORCH"):
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing for streams / torch.distributed, not a requirement of the renderer
        pass

from . import abi
from ._capi import Camera as _Camera
from ._capi import FlatSceneApi, Material, NativeError, Pattern, Shape
from ._capi import World as _World
from ._capi import _dptr

__all__ = ["api", "Material", "Pattern", "Shape", "World", "Camera", "Canvas", "DeviceScene", "NativeError",
           "hip", "device_count", "abi"]

if not os.path.exists(abi.HOST_LIB_PATH):
    raise ImportError(f"{abi.HOST_LIB_PATH} not found: run `python ray-tracer-challenge-rust_amd/build.py`")

_host = C.CDLL(abi.HOST_LIB_PATH)
_host.rtch_last_error.restype = C.c_char_p


def _last_error() -> str:
    return (_host.rtch_last_error() or b"").decode(errors="replace")


class RtchStats(C.Structure):
    _fields_ = [("radiance_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("triangle_tests", C.c_uint64),
                ("primitive_tests", C.c_uint64), ("filter_tests", C.c_uint64), ("box_tests", C.c_uint64),
                ("culled_rays", C.c_uint64), ("unswept_shadow_rays", C.c_uint64), ("kernel_ms", C.c_double), ("upload_ms", C.c_double), ("download_ms", C.c_double)]


class RtchFlatInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_transforms", "n_materials", "n_patterns", "n_groups", "n_leaves",
                                           "n_triangles", "n_classes_shared")]


_vp, _dp, _fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
_host.rtch_color_at.restype = C.c_int
_host.rtch_color_at.argtypes = [_vp, C.c_int64, _dp, _dp, C.c_int, C.c_uint32, C.c_int, _dp, _fp]
_host.rtch_render.restype = C.c_int
_host.rtch_render.argtypes = [_vp, _vp, C.c_int, C.c_uint32, C.c_int, _fp, _dp, C.POINTER(RtchStats)]
_host.rtch_render_multi.restype = C.c_int
_host.rtch_render_multi.argtypes = [_vp, _vp, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_int), _fp, _dp, C.POINTER(RtchStats)]
_host.rtch_world_flat_info.restype = C.c_int
_host.rtch_world_flat_info.argtypes = [_vp, C.POINTER(RtchFlatInfo)]
_host.rtch_world_scene_create.restype = C.c_int
_host.rtch_world_scene_create.argtypes = [_vp, C.c_int, C.POINTER(_vp)]
_host.rtch_camera_desc.restype = None
_host.rtch_camera_desc.argtypes = [_vp, C.POINTER(abi.RtcCameraDesc)]
_host.rtch_shape_bounds.restype = None
_host.rtch_shape_bounds.argtypes = [_vp, _dp, _dp]

_hip_lib = None


def hip() -> C.CDLL:
    """The raw C ABI (csrc/librtc_hip.so)."""
    global _hip_lib
    if _hip_lib is None:
        _hip_lib = abi.load_hip_library()
    return _hip_lib


def device_count() -> int:
    return int(hip().rtc_device_count())


def _hip_check(rc: int, what: str):
    if rc != abi.RTC_OK:
        raise NativeError(rc, f"{what}: {(hip().rtc_last_error_string() or b'').decode(errors='replace')}")


@dataclass
class Canvas:
    """canvas.rs Canvas: width, height, pixels (row-major, top row first).  rgb64 are the f64 colours the GPU
    computed, rgb32 the f32 framebuffer it stored (the values the parity metric is measured on)."""

    width: int
    height: int
    rgb32: np.ndarray
    rgb64: np.ndarray
    stats: dict

    def get_pixel(self, x: int, y: int) -> np.ndarray:
        return self.rgb64[y, x]

    def to_ppm(self) -> str:
        return api.canvas_to_ppm(self.width, self.height, self.rgb64)


class DeviceScene:
    """A flattened World resident in HBM (RtcScene*): render many frames without re-uploading."""

    def __init__(self, handle: int, device: int):
        self._h, self.device = handle, device

    def close(self):
        if self._h:
            hip().rtc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_device(self, cam_desc: abi.RtcCameraDesc, opts: abi.RtcRenderOpts, d_rgb: int, d_rgb64: int = 0,
                      stream: int = 0):
        """rtc_render_device: asynchronous launch on `stream`; d_rgb/d_rgb64 are device pointers."""
        _hip_check(hip().rtc_render_device(self._h, C.byref(cam_desc), C.byref(opts), d_rgb, d_rgb64 or None,
                                           stream or None), "rtc_render_device")

    def stats(self) -> dict:
        s = abi.RtcStats()
        _hip_check(hip().rtc_scene_stats(self._h, C.byref(s)), "rtc_scene_stats")
        return {"radiance_rays": int(s.radiance_rays), "shadow_rays": int(s.shadow_rays),
                "triangle_tests": int(s.triangle_tests), "primitive_tests": int(s.primitive_tests),
                "filter_tests": int(s.filter_tests), "box_tests": int(s.box_tests), "culled_rays": int(s.culled_rays), "unswept_shadow_rays": int(s.unswept_shadow_rays),
                "kernel_ms": s.kernel_ms, "upload_ms": s.upload_ms}

    def reset_counters(self):
        _hip_check(hip().rtc_scene_reset_counters(self._h), "rtc_scene_reset_counters")

    def color_at(self, origins, directions, recursion_limit: int = 5, count_rays: bool = False, exact_sweep: bool = False):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        out64 = np.zeros((n, 3), dtype=np.float64)
        out32 = np.zeros((n, 3), dtype=np.float32)
        flags = (abi.RTC_FLAG_COUNT_RAYS if count_rays else 0) | (abi.RTC_FLAG_EXACT_SWEEP if exact_sweep else 0)
        opts = abi.RtcRenderOpts(recursion_limit, 0, 0, 0, flags)
        _hip_check(hip().rtc_color_at(self._h, n, _dptr(o), _dptr(d), C.byref(opts),
                                      out32.ctypes.data_as(_fp), _dptr(out64)), "rtc_color_at")
        return out64, out32


class World(_World):
    """world.rs World {objects, light}; color_at runs on the GPU."""

    def color_at(self, origins, directions, recursion_limit: int = 5, device: int = 0, exact_sweep: bool = False):
        """World::color_at (world.rs:80) for a batch of rays -> (rgb64 [n,3], rgb32 [n,3])."""
        flags = abi.RTC_FLAG_EXACT_SWEEP if exact_sweep else 0
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        out64 = np.zeros((n, 3), dtype=np.float64)
        out32 = np.zeros((n, 3), dtype=np.float32)
        rc = _host.rtch_color_at(self._h, n, _dptr(o), _dptr(d), int(recursion_limit), flags, int(device), _dptr(out64),
                                 out32.ctypes.data_as(_fp))
        if rc != 0:
            raise NativeError(rc, _last_error())
        return out64, out32

    def flat_info(self) -> dict:
        info = RtchFlatInfo()
        rc = _host.rtch_world_flat_info(self._h, C.byref(info))
        if rc != 0:
            raise NativeError(rc, _last_error())
        return {n: int(getattr(info, n)) for n, _ in RtchFlatInfo._fields_}

    def upload(self, device: int = 0) -> DeviceScene:
        """Flatten + rtc_scene_create: the scene stays resident in HBM until the DeviceScene is closed."""
        h = _vp()
        rc = _host.rtch_world_scene_create(self._h, int(device), C.byref(h))
        if rc != 0:
            raise NativeError(rc, _last_error())
        return DeviceScene(h.value, device)


class Camera(_Camera):
    """camera.rs Camera; render runs on the GPU through rtc_render."""

    def desc(self) -> abi.RtcCameraDesc:
        d = abi.RtcCameraDesc()
        _host.rtch_camera_desc(self._h, C.byref(d))
        return d

    def render(self, world: World, recursion_limit: int = 5, device: int = 0, exact_sweep: bool = False,
               stats: bool = False, devices=None, p2p: bool = False) -> Canvas:
        """Camera::render (camera.rs:67-80).  exact_sweep selects the brute-force sweep (RTC_FLAG_EXACT_SWEEP);
        the default accelerated sweep returns bit-identical pixels.  stats=True also returns the ray / leaf-test
        counters and the kernel time (Canvas.stats); that runs the counting kernel, which is several times slower.
        devices=[d0, d1, ...] shards the frame's rows over those devices in this one process (rtc_render_multi:
        interleaved 16-row blocks gathered to d0 over xGMI with RCCL, or peer DMA with p2p=True)."""
        flags = (abi.RTC_FLAG_EXACT_SWEEP if exact_sweep else 0) | (abi.RTC_FLAG_GATHER_P2P if p2p else 0)
        rgb32 = np.empty((self.vsize, self.hsize, 3), dtype=np.float32)  # every value is written by the call
        rgb64 = np.empty((self.vsize, self.hsize, 3), dtype=np.float64)
        st = RtchStats()
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = _host.rtch_render_multi(self._h, world._h, int(recursion_limit), flags, len(devices), devs,
                                         rgb32.ctypes.data_as(_fp), _dptr(rgb64), C.byref(st) if stats else None)
        else:
            rc = _host.rtch_render(self._h, world._h, int(recursion_limit), flags, int(device), rgb32.ctypes.data_as(_fp),
                                   _dptr(rgb64), C.byref(st) if stats else None)
        if rc != 0:
            raise NativeError(rc, _last_error())
        if not stats:
            return Canvas(self.hsize, self.vsize, rgb32, rgb64, {})
        return Canvas(self.hsize, self.vsize, rgb32, rgb64,
                      {n: (float(getattr(st, n)) if n.endswith("_ms") else int(getattr(st, n))) for n, _ in RtchStats._fields_})


class _Api(FlatSceneApi):
    def world(self, light_position, light_intensity=(1.0, 1.0, 1.0), cls=World):
        return super().world(light_position, light_intensity, cls)

    def default_world(self, cls=World):
        return super().default_world(cls)

    def camera(self, hsize, vsize, fov, cls=Camera):
        return super().camera(hsize, vsize, fov, cls)

    def bounds(self, shape: Shape):
        """Bounds::new (bounds.rs:11)."""
        mn, mx = np.zeros(3), np.zeros(3)
        _host.rtch_shape_bounds(shape._h, _dptr(mn), _dptr(mx))
        return mn, mx


api = _Api(_host, "rtch_", _last_error)
