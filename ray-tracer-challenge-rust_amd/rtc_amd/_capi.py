"""ctypes binding for a "flat scene C API" (see host/rtc_host_capi.h).

The binding is generic over the symbol prefix so that the test-suite can drive its CPU checker, which
exports the same API shape under another prefix, with the same scene-building code.  This module contains
no arithmetic: every number comes from the native library behind it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field, replace
from typing import Iterable, Optional, Sequence

import numpy as np

PATTERN_KINDS = {"stripe": 1, "gradient": 2, "ring": 3, "checkers": 4, "test": 5}
_PATTERN_NAMES = {v: k for k, v in PATTERN_KINDS.items()}


class CMaterial(C.Structure):
    """RtchMaterial / same layout in the checker."""

    _fields_ = [
        ("color", C.c_double * 3),
        ("ambient", C.c_double),
        ("diffuse", C.c_double),
        ("specular", C.c_double),
        ("shininess", C.c_double),
        ("reflective", C.c_double),
        ("transparency", C.c_double),
        ("refractive_index", C.c_double),
        ("pattern_kind", C.c_int32),
        ("_pad", C.c_int32),
        ("pattern_a", C.c_double * 3),
        ("pattern_b", C.c_double * 3),
        ("pattern_transform", C.c_double * 16),
    ]


@dataclass
class Pattern:
    """pattern.rs Pattern: kind, two colours, transform."""

    kind: str
    a: Sequence[float] = (0.0, 0.0, 0.0)
    b: Sequence[float] = (0.0, 0.0, 0.0)
    transform: Optional[np.ndarray] = None

    def set_transform(self, m: np.ndarray) -> "Pattern":
        self.transform = np.array(m, dtype=np.float64)
        return self


@dataclass
class Material:
    """material.rs Material, defaults from Material::new (material.rs:17-29)."""

    color: Sequence[float] = (1.0, 1.0, 1.0)
    ambient: float = 0.1
    diffuse: float = 0.9
    specular: float = 0.9
    shininess: float = 200.0
    reflective: float = 0.0
    pattern: Optional[Pattern] = None
    transparency: float = 0.0
    refractive_index: float = 1.0

    def to_c(self) -> CMaterial:
        m = CMaterial()
        m.color[:] = [float(x) for x in self.color]
        m.ambient, m.diffuse, m.specular = self.ambient, self.diffuse, self.specular
        m.shininess, m.reflective = self.shininess, self.reflective
        m.transparency, m.refractive_index = self.transparency, self.refractive_index
        ident = np.eye(4).ravel()
        m.pattern_transform[:] = ident.tolist()
        if self.pattern is not None:
            m.pattern_kind = PATTERN_KINDS[self.pattern.kind]
            m.pattern_a[:] = [float(x) for x in self.pattern.a]
            m.pattern_b[:] = [float(x) for x in self.pattern.b]
            if self.pattern.transform is not None:
                m.pattern_transform[:] = np.asarray(self.pattern.transform, dtype=np.float64).ravel().tolist()
        return m

    @staticmethod
    def from_c(m: CMaterial) -> "Material":
        pat = None
        if m.pattern_kind != 0:
            pat = Pattern(_PATTERN_NAMES[m.pattern_kind], tuple(m.pattern_a), tuple(m.pattern_b),
                          np.array(m.pattern_transform[:], dtype=np.float64).reshape(4, 4))
        return Material(tuple(m.color), m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, pat,
                        m.transparency, m.refractive_index)


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _vec3(v) -> "C.Array":
    a = (C.c_double * 3)()
    a[:] = [float(v[0]), float(v[1]), float(v[2])]
    return a


def _mat(m) -> "C.Array":
    a = (C.c_double * 16)()
    a[:] = np.asarray(m, dtype=np.float64).reshape(16).tolist()
    return a


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Shape:
    """Handle to a native Shape (value semantics: push/copy)."""

    def __init__(self, api: "FlatSceneApi", handle):
        if not handle:
            raise NativeError(-1, "null shape handle")
        self._api, self._h = api, handle

    def __del__(self):
        try:
            if self._h:
                self._api._f("shape_free")(self._h)
        except Exception:
            pass

    def set_transform(self, m) -> "Shape":
        self._api._check(self._api._f("shape_set_transform")(self._h, _mat(m)))
        return self

    def set_material(self, mat: Material) -> "Shape":
        cm = mat.to_c()
        self._api._f("shape_set_material")(self._h, C.byref(cm))
        return self

    @property
    def material(self) -> Material:
        cm = CMaterial()
        self._api._f("shape_get_material")(self._h, C.byref(cm))
        return Material.from_c(cm)

    def update_material(self, **kw) -> "Shape":
        """get_material_mut().field = value"""
        return self.set_material(replace(self.material, **kw))

    def push_shape(self, child: "Shape") -> "Shape":
        self._api._check(self._api._f("shape_push")(self._h, child._h))
        return self

    def leaf_count(self) -> int:
        return int(self._api._f("shape_leaf_count")(self._h))


class World:
    def __init__(self, api: "FlatSceneApi", handle):
        self._api, self._h = api, handle
        self.shapes: list = []

    def __del__(self):
        try:
            if self._h:
                self._api._f("world_free")(self._h)
        except Exception:
            pass

    def push(self, shape: Shape) -> "World":
        self._api._f("world_push")(self._h, shape._h)
        return self


class Camera:
    def __init__(self, api: "FlatSceneApi", handle, hsize: int, vsize: int, fov: float):
        self._api, self._h = api, handle
        self.hsize, self.vsize, self.field_of_view = hsize, vsize, fov

    def __del__(self):
        try:
            if self._h:
                self._api._f("camera_free")(self._h)
        except Exception:
            pass

    def set_transform(self, m) -> "Camera":
        self._api._check(self._api._f("camera_set_transform")(self._h, _mat(m)))
        return self

    def _get(self):
        ps, hw, hh = C.c_double(), C.c_double(), C.c_double()
        inv = (C.c_double * 16)()
        self._api._f("camera_get")(self._h, C.byref(ps), C.byref(hw), C.byref(hh), inv)
        return ps.value, hw.value, hh.value, np.array(inv[:]).reshape(4, 4)

    @property
    def pixel_size(self) -> float:
        return self._get()[0]

    @property
    def half_width(self) -> float:
        return self._get()[1]

    @property
    def half_height(self) -> float:
        return self._get()[2]

    @property
    def transform_inverse(self) -> np.ndarray:
        return self._get()[3]

    def ray_for_pixel(self, px: int, py: int):
        o, d = (C.c_double * 3)(), (C.c_double * 3)()
        self._api._f("camera_ray_for_pixel")(self._h, px, py, o, d)
        return np.array(o[:]), np.array(d[:])


class FlatSceneApi:
    """Scene-building half of the API (names follow the reference: Shape::sphere, World::new, ...)."""

    def __init__(self, lib: C.CDLL, prefix: str, last_error=None):
        self.lib, self.prefix = lib, prefix
        self._last_error = last_error
        self._declare()

    # ---- plumbing
    def _f(self, name: str):
        return getattr(self.lib, self.prefix + name)

    def _check(self, rc: int):
        if rc != 0:
            msg = self._last_error() if self._last_error else ""
            raise NativeError(rc, msg or f"{self.prefix} call failed")

    def _declare(self):
        D, P = C.c_double, C.POINTER(C.c_double)
        vp = C.c_void_p
        sig = {
            "identity": (None, [P]),
            "translation": (None, [D, D, D, P]),
            "scaling": (None, [D, D, D, P]),
            "rotation_x": (None, [D, P]),
            "rotation_y": (None, [D, P]),
            "rotation_z": (None, [D, P]),
            "shearing": (None, [D, D, D, D, D, D, P]),
            "view_transform": (None, [P, P, P, P]),
            "mat_mul": (None, [P, P, P]),
            "mat_inverse": (C.c_int, [P, P]),
            "material_default": (None, [C.POINTER(CMaterial)]),
            "shape_sphere": (vp, []),
            "shape_glass_sphere": (vp, []),
            "shape_plane": (vp, []),
            "shape_cube": (vp, []),
            "shape_cylinder": (vp, [D, D, C.c_int]),
            "shape_cone": (vp, [D, D, C.c_int]),
            "shape_group": (vp, []),
            "shape_triangle": (vp, [P, P, P]),
            "shape_free": (None, [vp]),
            "shape_set_transform": (C.c_int, [vp, P]),
            "shape_set_material": (None, [vp, C.POINTER(CMaterial)]),
            "shape_get_material": (None, [vp, C.POINTER(CMaterial)]),
            "shape_push": (C.c_int, [vp, vp]),
            "shape_leaf_count": (C.c_int64, [vp]),
            "obj_parse": (vp, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
            "world_new": (vp, [P, P]),
            "world_default": (vp, []),
            "world_free": (None, [vp]),
            "world_push": (None, [vp, vp]),
            "camera_new": (vp, [C.c_int64, C.c_int64, D]),
            "camera_free": (None, [vp]),
            "camera_set_transform": (C.c_int, [vp, P]),
            "camera_get": (None, [vp, P, P, P, P]),
            "camera_ray_for_pixel": (None, [vp, C.c_int64, C.c_int64, P, P]),
            "canvas_to_ppm": (C.c_size_t, [C.c_int64, C.c_int64, P, C.c_char_p, C.c_size_t]),
        }
        for name, (res, args) in sig.items():
            fn = self._f(name)
            fn.restype, fn.argtypes = res, args

    def _m(self, name: str, *args) -> np.ndarray:
        out = (C.c_double * 16)()
        self._f(name)(*args, out)
        return np.array(out[:], dtype=np.float64).reshape(4, 4)

    # ---- transformations.rs / matrix.rs
    def identity(self):
        return self._m("identity")

    def translation(self, x, y, z):
        return self._m("translation", float(x), float(y), float(z))

    def scaling(self, x, y, z):
        return self._m("scaling", float(x), float(y), float(z))

    def rotation_x(self, r):
        return self._m("rotation_x", float(r))

    def rotation_y(self, r):
        return self._m("rotation_y", float(r))

    def rotation_z(self, r):
        return self._m("rotation_z", float(r))

    def shearing(self, xy, xz, yx, yz, zx, zy):
        return self._m("shearing", *[float(v) for v in (xy, xz, yx, yz, zx, zy)])

    def view_transform(self, frm, to, up):
        return self._m("view_transform", _vec3(frm), _vec3(to), _vec3(up))

    def mul(self, *ms) -> np.ndarray:
        """a * b * c ... with the library's own Matrix * Matrix (left to right, like the Rust expression)."""
        acc = np.asarray(ms[0], dtype=np.float64)
        for m in ms[1:]:
            acc = self._m("mat_mul", _mat(acc), _mat(m))
        return acc

    def inverse(self, m) -> np.ndarray:
        out = (C.c_double * 16)()
        self._check(self._f("mat_inverse")(_mat(m), out))
        return np.array(out[:]).reshape(4, 4)

    def default_material(self) -> Material:
        cm = CMaterial()
        self._f("material_default")(C.byref(cm))
        return Material.from_c(cm)

    # ---- shape.rs
    def sphere(self):
        return Shape(self, self._f("shape_sphere")())

    def glass_sphere(self):
        return Shape(self, self._f("shape_glass_sphere")())

    def plane(self):
        return Shape(self, self._f("shape_plane")())

    def cube(self):
        return Shape(self, self._f("shape_cube")())

    def cylinder(self, minimum=-np.inf, maximum=np.inf, capped=False):
        return Shape(self, self._f("shape_cylinder")(float(minimum), float(maximum), int(bool(capped))))

    def cone(self, minimum=-np.inf, maximum=np.inf, capped=False):
        return Shape(self, self._f("shape_cone")(float(minimum), float(maximum), int(bool(capped))))

    def group(self, children: Iterable[Shape] = ()):
        g = Shape(self, self._f("shape_group")())
        for c in children:
            g.push_shape(c)
        return g

    def triangle(self, p1, p2, p3):
        return Shape(self, self._f("shape_triangle")(_vec3(p1), _vec3(p2), _vec3(p3)))

    # ---- obj_file.rs
    def obj_parse(self, text: str | bytes):
        data = text.encode() if isinstance(text, str) else text
        ignored, err = C.c_int64(0), C.c_int(0)
        h = self._f("obj_parse")(data, len(data), C.byref(ignored), C.byref(err))
        if not h:
            msg = self._last_error() if self._last_error else "OBJ parse error"
            raise NativeError(err.value, msg)
        s = Shape(self, h)
        s.ignored_lines = ignored.value
        return s

    # ---- world.rs / camera.rs
    def world(self, light_position, light_intensity=(1.0, 1.0, 1.0), cls=World):
        return cls(self, self._f("world_new")(_vec3(light_position), _vec3(light_intensity)))

    def default_world(self, cls=World):
        return cls(self, self._f("world_default")())

    def camera(self, hsize: int, vsize: int, fov: float, cls=Camera):
        return cls(self, self._f("camera_new")(int(hsize), int(vsize), float(fov)), int(hsize), int(vsize), float(fov))

    # ---- canvas.rs
    def canvas_to_ppm(self, width: int, height: int, rgb: np.ndarray) -> str:
        rgb = np.ascontiguousarray(rgb, dtype=np.float64).reshape(-1)
        assert rgb.size == width * height * 3
        n = self._f("canvas_to_ppm")(width, height, _dptr(rgb), None, 0)
        buf = C.create_string_buffer(n)
        self._f("canvas_to_ppm")(width, height, _dptr(rgb), buf, n)
        return buf.raw[:n].decode()
