"""ctypes binding of the CPU oracle (oracle/_build/librtc_oracle.so) — TEST INFRASTRUCTURE ONLY.

Uses the same generic scene-API wrapper as the product binding (rtc_amd._capi.FlatSceneApi) with the `orc_`
prefix, so one scene function builds identical inputs for both sides.  Nothing under
ray-tracer-challenge-rust_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rtc_amd._capi import Camera as _Camera
from rtc_amd._capi import FlatSceneApi, NativeError
from rtc_amd._capi import World as _World
from rtc_amd._capi import _dptr

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "librtc_oracle.so")
KAT = os.path.join(ORACLE_DIR, "_build", "kat")

MODE_CACHED, MODE_FAITHFUL = 0, 1


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _load():
    if not (os.path.exists(LIB) and os.path.exists(KAT)):
        build()
    lib = C.CDLL(LIB)
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    lib.orc_color_at.restype = C.c_int
    lib.orc_color_at.argtypes = [vp, C.c_int64, dp, dp, C.c_int, C.c_int, dp, C.POINTER(C.c_int64)]
    lib.orc_render_pixels.restype = C.c_int
    lib.orc_render_pixels.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                      C.c_int, dp, C.POINTER(C.c_int64), dp]
    return lib


_lib = _load()


class World(_World):
    def color_at(self, origins, directions, recursion_limit=5, mode=MODE_CACHED):
        """-> (rgb float64 [n,3], (radiance_rays, shadow_rays))"""
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        out = np.zeros_like(o)
        counts = (C.c_int64 * 2)()
        rc = _lib.orc_color_at(self._h, o.shape[0], _dptr(o), _dptr(d), int(recursion_limit), int(mode), _dptr(out), counts)
        if rc != 0:
            raise NativeError(rc, "oracle color_at failed")
        return out, (counts[0], counts[1])


class Camera(_Camera):
    def render(self, world, recursion_limit=5, mode=MODE_CACHED, nthreads=8):
        """Full frame -> (rgb float64 [vsize,hsize,3], info)"""
        out = np.zeros((self.vsize, self.hsize, 3), dtype=np.float64)
        counts, secs = (C.c_int64 * 2)(), C.c_double()
        rc = _lib.orc_render_pixels(self._h, world._h, int(recursion_limit), int(mode), -1, None, None, int(nthreads),
                                    _dptr(out), counts, C.byref(secs))
        if rc != 0:
            raise NativeError(rc, "oracle render failed")
        return out, {"radiance_rays": counts[0], "shadow_rays": counts[1], "seconds": secs.value}

    def render_pixels(self, world, xs, ys, recursion_limit=5, mode=MODE_CACHED, nthreads=1):
        """Selected pixels -> (rgb float64 [n,3], info)"""
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        ys = np.ascontiguousarray(ys, dtype=np.int32)
        out = np.zeros((xs.size, 3), dtype=np.float64)
        counts, secs = (C.c_int64 * 2)(), C.c_double()
        rc = _lib.orc_render_pixels(self._h, world._h, int(recursion_limit), int(mode), xs.size,
                                    xs.ctypes.data_as(C.POINTER(C.c_int32)), ys.ctypes.data_as(C.POINTER(C.c_int32)),
                                    int(nthreads), _dptr(out), counts, C.byref(secs))
        if rc != 0:
            raise NativeError(rc, "oracle render_pixels failed")
        return out, {"radiance_rays": counts[0], "shadow_rays": counts[1], "seconds": secs.value}


class _Api(FlatSceneApi):
    def world(self, light_position, light_intensity=(1.0, 1.0, 1.0), cls=World):
        return super().world(light_position, light_intensity, cls)

    def default_world(self, cls=World):
        return super().default_world(cls)

    def camera(self, hsize, vsize, fov, cls=Camera):
        return super().camera(hsize, vsize, fov, cls)


api = _Api(_lib, "orc_")
