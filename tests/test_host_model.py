"""CPU tests of the product's host side (C++ scene model, flattener, OBJ loader, PPM writer) — no GPU.

The host library must hand the device the same doubles the reference would hold, so its flatten-time
arithmetic is compared BIT-FOR-BIT with the oracle's independent restatement, and against the reference's
golden values where the reference's tests pin them (1e-5, the reference's own tolerance).
"""
import math

import numpy as np
import pytest

from rtc_amd import meshes, scenes
from rtc_amd._capi import Material, NativeError, Pattern

PI = math.pi


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_transform_builders_bit_equal(prod, orc):
    cases = [
        ("translation", (5.0, -3.0, 2.0)), ("scaling", (2.0, 3.0, 4.0)), ("rotation_x", (PI / 4,)),
        ("rotation_y", (PI / 6,)), ("rotation_z", (-PI / 5,)), ("shearing", (1.0, 2.0, 3.0, 4.0, 5.0, 6.0)),
    ]
    for name, args in cases:
        assert (bits(getattr(prod, name)(*args)) == bits(getattr(orc, name)(*args))).all(), name
    a = prod.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0))
    b = orc.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0))
    assert (bits(a) == bits(b)).all()
    # transformations.rs:305-319
    exp = np.array([[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
                    [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1.0]])
    assert np.abs(a - exp).max() < 1e-5


def test_inverse_matches_reference_goldens_and_oracle(prod, orc):
    a = np.array([[-5., 2., 6., -8.], [1., -5., 1., 8.], [7., 7., -6., -7.], [1., -3., 7., 4.]])
    exp = np.array([[0.21805, 0.45113, 0.24060, -0.04511], [-0.80827, -1.45677, -0.44361, 0.52068],
                    [-0.07895, -0.22368, -0.05263, 0.19737], [-0.52256, -0.81391, -0.30075, 0.30639]])  # matrix.rs:480-503
    inv = prod.inverse(a)
    assert np.abs(inv - exp).max() < 1e-5
    assert (bits(inv) == bits(orc.inverse(a))).all()
    rng = np.random.default_rng(7)
    for _ in range(50):
        m = prod.mul(prod.translation(*rng.normal(size=3)), prod.rotation_y(rng.normal()), prod.rotation_z(rng.normal()),
                     prod.scaling(*(0.2 + rng.random(3))))
        i1, i2 = prod.inverse(m), orc.inverse(m)
        assert (bits(i1) == bits(i2)).all()
        # the device layout stores 3x4: bottom row must be exactly (0,0,0,1)
        assert (i1[3] == [0.0, 0.0, 0.0, 1.0]).all()
    with pytest.raises(NativeError):
        prod.inverse(np.array([[-4., 2., -2., -3.], [9., 6., 2., 6.], [0., -5., 1., -5.], [0., 0., 0., 0.]]))  # matrix.rs:466-477


def test_camera_fields(prod, orc):
    # camera.rs:105-114
    assert abs(prod.camera(200, 125, PI / 2).pixel_size - 0.01) < 1e-5
    assert abs(prod.camera(125, 200, PI / 2).pixel_size - 0.01) < 1e-5
    for api_pair in [(1920, 1080, 0.785), (201, 101, PI / 2), (100, 200, 1.1)]:
        cp, co = prod.camera(*api_pair), orc.camera(*api_pair)
        t = lambda api: api.mul(api.rotation_y(PI / 4), api.translation(0, -2, 5))
        cp.set_transform(t(prod)), co.set_transform(t(orc))
        assert cp.pixel_size == co.pixel_size and cp.half_width == co.half_width and cp.half_height == co.half_height
        assert (bits(cp.transform_inverse) == bits(co.transform_inverse)).all()
        for px, py in [(0, 0), (100, 50), (api_pair[0] - 1, api_pair[1] - 1)]:
            op, dp = cp.ray_for_pixel(px, py)
            oo, do = co.ray_for_pixel(px, py)
            assert (bits(op) == bits(oo)).all() and (bits(dp) == bits(do)).all()
    # camera.rs:125-142
    c = prod.camera(201, 101, PI / 2)
    o, d = c.ray_for_pixel(0, 0)
    assert np.abs(d - [0.66519, 0.33259, -0.66851]).max() < 1e-5 and np.abs(o).max() == 0
    c.set_transform(prod.mul(prod.rotation_y(PI / 4), prod.translation(0, -2, 5)))
    o, d = c.ray_for_pixel(100, 50)
    assert np.abs(o - [0, 2, -5]).max() < 1e-5 and np.abs(d - [math.sqrt(2) / 2, 0, -math.sqrt(2) / 2]).max() < 1e-5


def test_shape_errors(prod):
    s = prod.sphere().set_transform(prod.translation(1, 0, 0))
    with pytest.raises(NativeError) as e:
        s.set_transform(prod.translation(1, 0, 0))  # shape.rs:199-201
    assert e.value.code == -102
    with pytest.raises(NativeError) as e:
        prod.sphere().set_transform(np.zeros((4, 4)))  # shape.rs:215 expect("should be invertible")
    assert e.value.code == -101
    with pytest.raises(NativeError) as e:
        prod.sphere().push_shape(prod.sphere())  # shape.rs:533
    assert e.value.code == -103
    with pytest.raises(NativeError):
        prod.camera(10, 10, 1.0).set_transform(np.zeros((4, 4)))  # camera.rs:45


def test_material_roundtrip_and_group_push_down(prod):
    m = prod.default_material()  # material.rs:17-29
    assert (tuple(m.color), m.ambient, m.diffuse, m.specular, m.shininess) == ((1.0, 1.0, 1.0), 0.1, 0.9, 0.9, 200.0)
    assert (m.reflective, m.transparency, m.refractive_index, m.pattern) == (0.0, 0.0, 1.0, None)
    g = prod.glass_sphere().material  # shape.rs:63-76
    assert g.transparency == 1.0 and g.refractive_index == 1.5
    grp = prod.group([prod.sphere(), prod.group([prod.cube()])])
    mat = Material(color=(0.1, 0.2, 0.3), reflective=0.25, pattern=Pattern("ring", (1, 0, 0), (0, 1, 0)).set_transform(prod.scaling(2, 2, 2)))
    grp.set_material(mat)  # recursive (shape.rs:220-229)
    assert grp.leaf_count() == 2
    w = prod.world((0, 0, 0))
    w.push(grp)
    info = w.flat_info()
    assert info["n_leaves"] == 2 and info["n_groups"] == 2 and info["n_materials"] == 1 and info["n_patterns"] == 1
    assert info["n_transforms"] == 1


def test_bounds_match_oracle_quirks(prod):
    # bounds.rs: group/triangle boxes are seeded with the origin; planes are an xy square
    tri = prod.triangle((1, 1, 1), (2, 1, 1), (1, 3, 1))
    mn, mx = prod.bounds(tri)
    assert (mn == [0, 0, 0]).all() and (mx == [2, 3, 1]).all()
    mn, mx = prod.bounds(prod.plane())
    assert (mn == [-1, -1, 0]).all() and (mx == [1, 1, 0]).all()
    s = prod.sphere().set_transform(prod.translation(5, 0, 0))
    g = prod.group([s])
    g.set_transform(prod.scaling(2, 2, 2))
    mn, mx = prod.bounds(g)
    assert (mn == [0, -2, -2]).all() and (mx == [12, 2, 2]).all()
    mn, mx = prod.bounds(prod.cylinder())
    assert mn[1] == -np.inf and mx[1] == np.inf


def test_obj_parser_reference_cases(prod):
    gib = ("\n            There was a young lady named Bright\n            who traveled much faster than light.\n"
           "            She set out one day\n            in a relative way,\n            and came back the previous night.\n            ")
    assert prod.obj_parse(gib).ignored_lines == 5  # obj_file.rs:139-153
    g = prod.obj_parse("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\nf 1 2 3 4 5\n")  # obj_file.rs:196-236 fan
    assert g.leaf_count() == 3
    g = prod.obj_parse("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\n\ng FirstGroup\nf 1 2 3\ng SecondGroup\nf 1 3 4\n")  # triangles.obj
    w = prod.world((0, 0, 0))
    w.push(g)
    info = w.flat_info()
    assert info["n_groups"] == 4 and info["n_leaves"] == 2  # outer, default, FirstGroup, SecondGroup
    for bad in ("v 1 2\n", "v a b c\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1 2/2 3/3\n", "v 0 0 0\nf 1 2 3\n", "g\n", "f 0 1 2 3\n"):
        with pytest.raises(NativeError):
            prod.obj_parse(bad)


@pytest.mark.parametrize("name,ntri", [("teapot", 6320), ("cow", 5804), ("pumpkin", 10000), ("teddy", 3192)])
def test_mesh_flatten(prod, orc, name, ntri):
    text = meshes.obj_text(name)
    g = prod.obj_parse(text)
    assert g.leaf_count() == ntri == orc.obj_parse(text).leaf_count()
    g.set_transform(prod.mul(prod.translation(0, 1, 0), prod.scaling(0.5, 0.5, 0.5)))
    w = prod.world((0, 0, 0))
    w.push(g)
    info = w.flat_info()
    # group[default_group[N triangles]]: one transform, one material for the whole mesh
    assert info == {"n_transforms": 1, "n_materials": 1, "n_patterns": 0, "n_groups": 2, "n_leaves": ntri,
                    "n_triangles": ntri, "n_classes_shared": info["n_classes_shared"]}
    assert info["n_classes_shared"] == 0, "mesh has value-equal duplicate triangles"


def test_ppm_reference_goldens(prod, orc):
    px = np.zeros((3, 5, 3))
    px[0, 0] = (1.5, 0, 0)
    px[1, 2] = (0, 0.5, 0)
    px[2, 4] = (-0.5, 0, 1)
    exp = ("P3\n5 3\n255\n255 0 0 0 0 0 0 0 0 0 0 0 0 0 0\n0 0 0 0 0 0 0 128 0 0 0 0 0 0 0\n"
           "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255\n")  # canvas.rs:110-131
    assert prod.canvas_to_ppm(5, 3, px) == exp
    q = np.tile(np.array([1.0, 0.8, 0.6]), (2, 10, 1))
    l1 = "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204\n"
    l2 = "153 255 204 153 255 204 153 255 204 153 255 204 153\n"
    assert prod.canvas_to_ppm(10, 2, q) == "P3\n10 2\n255\n" + l1 + l2 + l1 + l2  # canvas.rs:134-163
    rng = np.random.default_rng(3)
    r = rng.uniform(-0.3, 1.4, size=(7, 23, 3))
    r[0, 0, 0] = np.nan
    assert prod.canvas_to_ppm(23, 7, r) == orc.canvas_to_ppm(23, 7, r)
    assert prod.canvas_to_ppm(23, 7, r).endswith("\n")


def test_scene_builders_run_on_both_apis(prod, orc):
    for name, fn in scenes.CONFIGS.items():
        kw = {"hsize": 32, "vsize": 18}
        if name in ("C3", "C4", "C5"):
            kw["max_faces"] = 50
        for api in (prod, orc):
            cam, w, L = fn(api, **kw)
            assert cam.hsize == 32 and L in (1, 5, 8)
    cam, w, L = scenes.hexagon(prod)
    assert w.flat_info()["n_groups"] == 7 and w.flat_info()["n_leaves"] == 12
    assert scenes.table(prod)[1].flat_info()["n_leaves"] == 18
