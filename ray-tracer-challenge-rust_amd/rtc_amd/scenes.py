"""Scene definitions = measurement inputs (BASELINE.json configs C1..C5) plus the reference's two
other hard-coded scenes.  Each builder takes a scene API object (rtc_amd.api, or the test-suite's checker
binding, which exposes the same names) and returns (camera, world, recursion_limit).

C3 / C4 / hexagon / table follow reference main.rs:84-397 line by line (same constructor order, same matrix
products); C1, C2 and C5 are authored for this build exactly as SURVEY.md §8d writes them down (the reference
has no such scene functions).
"""
from __future__ import annotations

import math

from . import meshes
from ._capi import Material, Pattern

PI = math.pi
GREEN, BLUE, BLACK, WHITE = (0.0, 1.0, 0.0), (0.0, 0.0, 1.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)


def _three_spheres(api, glass_right: bool, reflective_middle: float):
    middle = api.sphere().set_transform(api.translation(-0.5, 1, 0.5))
    middle.set_material(Material(color=(0.1, 1, 0.5), diffuse=0.7, specular=0.3, reflective=reflective_middle))
    right = api.sphere().set_transform(api.mul(api.translation(1.5, 0.5, -0.5), api.scaling(0.5, 0.5, 0.5)))
    if glass_right:
        right.set_material(Material(color=(0.1, 0.1, 0.1), transparency=0.9, reflective=0.9, refractive_index=1.5,
                                    diffuse=0.1, ambient=0.1))
    else:
        right.set_material(Material(color=(0.5, 1, 0.1), diffuse=0.7, specular=0.3))
    left = api.sphere().set_transform(api.mul(api.translation(-1.5, 0.33, -0.75), api.scaling(0.33, 0.33, 0.33)))
    left.set_material(Material(color=(1, 0.8, 0.1), diffuse=0.7, specular=0.3))
    return middle, right, left


def c1_chapter7(api, hsize=200, vsize=100):
    """C1: 3 spheres + plane, 200x100, depth 1 (no reflective/transparent material, so L=1 == L=2)."""
    cam = api.camera(hsize, vsize, PI / 3)
    cam.set_transform(api.view_transform((0, 1.5, -5), (0, 1, 0), (0, 1, 0)))
    w = api.world((-10, 10, -10), WHITE)
    floor = api.plane().set_material(Material(color=(1, 0.9, 0.9), specular=0.0))
    w.push(floor)
    for s in _three_spheres(api, glass_right=False, reflective_middle=0.0):
        w.push(s)
    return cam, w, 1


def c2_book_cover(api, hsize=1920, vsize=1080):
    """C2: checkered reflective floor, reflective middle sphere, glass right sphere; depth 5."""
    cam = api.camera(hsize, vsize, PI / 3)
    cam.set_transform(api.view_transform((0, 1.5, -5), (0, 1, 0), (0, 1, 0)))
    w = api.world((-10, 10, -10), WHITE)
    floor = api.plane().set_material(Material(color=(1, 0.9, 0.9), specular=0.0, reflective=0.4,
                                              pattern=Pattern("checkers", (0.35, 0.35, 0.35), (0.65, 0.65, 0.65))))
    w.push(floor)
    for s in _three_spheres(api, glass_right=True, reflective_middle=0.3):
        w.push(s)
    return cam, w, 5


def _mesh_group(api, name, max_faces=None):
    return api.obj_parse(meshes.obj_text(name, max_faces))


def c3_teapot(api, hsize=1920, vsize=1080, max_faces=None):
    """C3 = putting_it_together_teapot (main.rs:368-397) at hsize x vsize."""
    cam = api.camera(hsize, vsize, 0.785)
    cam.set_transform(api.view_transform((0.0, 4.0, -12.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    w = api.world((0.0, 6.9, -5.0), (1.0, 1.0, 0.9))
    teapot = _mesh_group(api, "teapot", max_faces)
    teapot.set_transform(api.translation(0.0, -1.5, 0.0))
    teapot.set_material(Material(pattern=Pattern("gradient", GREEN, BLUE)))
    w.push(teapot)
    return cam, w, 5


def c4_cow(api, hsize=3840, vsize=2160, max_faces=None):
    """C4 = putting_it_together_cow (main.rs:328-363) at hsize x vsize."""
    cam = api.camera(hsize, vsize, 0.785)
    cam.set_transform(api.view_transform((8.0, 6.0, -8.0), (0.0, 3.0, 0.0), (0.0, 1.0, 0.0)))
    w = api.world((0.0, 6.9, -5.0), (1.0, 1.0, 0.9))
    cow = _mesh_group(api, "cow", max_faces)
    cow.set_transform(api.mul(api.translation(0.0, 3.5, 0.0), api.scaling(0.5, 0.5, 0.5)))
    cow.set_material(Material(color=WHITE, ambient=0.1, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.2))
    w.push(cow)
    return cam, w, 5


def c5_pumpkin(api, hsize=3840, vsize=2160, max_faces=None):
    """C5: glass pumpkin over a checkered floor, depth 8."""
    cam = api.camera(hsize, vsize, 0.785)
    cam.set_transform(api.view_transform((0.0, 3.0, -12.0), (0.0, 2.0, 0.0), (0.0, 1.0, 0.0)))
    w = api.world((0.0, 6.9, -5.0), (1.0, 1.0, 0.9))
    floor = api.plane().set_material(Material(color=(1, 0.9, 0.9), specular=0.0, reflective=0.4,
                                              pattern=Pattern("checkers", (0.35, 0.35, 0.35), (0.65, 0.65, 0.65))))
    w.push(floor)
    pumpkin = _mesh_group(api, "pumpkin", max_faces)
    pumpkin.set_transform(api.mul(api.translation(0.1312, 1.9565, 5.5008), api.scaling(0.05, 0.05, 0.05)))
    pumpkin.set_material(Material(transparency=0.9, reflective=0.9, refractive_index=1.5, diffuse=0.1, ambient=0.1,
                                  specular=0.9, shininess=300.0))
    w.push(pumpkin)
    return cam, w, 8


def hexagon(api, hsize=400, vsize=200):
    """putting_it_together_hexagon (main.rs:84-146): nested groups, spheres and capped cylinders."""
    cam = api.camera(hsize, vsize, 0.785)
    cam.set_transform(api.view_transform((8.0, 6.0, -8.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    w = api.world((0.0, 6.9, -5.0), (1.0, 1.0, 0.9))

    def corner():
        return api.sphere().set_transform(api.mul(api.translation(0.0, 0.0, -1.0), api.scaling(0.25, 0.25, 0.25)))

    def edge():
        return api.cylinder(0.0, 1.0, True).set_transform(
            api.mul(api.translation(0.0, 0.0, -1.0), api.rotation_y(-PI / 6.0), api.rotation_z(-PI / 2.0),
                    api.scaling(0.25, 1.0, 0.25)))

    hexa = api.group()
    for i in range(6):
        side = api.group([corner(), edge()])
        side.set_transform(api.rotation_y(float(i) * PI / 3.0))
        hexa.push_shape(side)
    hexa.set_transform(api.scaling(2.5, 2.5, 2.5))
    w.push(hexa)
    return cam, w, 5


def table(api, hsize=400, vsize=200):
    """putting_it_together_table_scene (main.rs:151-323): 19 cubes, three patterns, a glass cube, a mirror."""
    cam = api.camera(hsize, vsize, 0.785)
    cam.set_transform(api.view_transform((8.0, 6.0, -8.0), (0.0, 3.0, 0.0), (0.0, 1.0, 0.0)))
    w = api.world((0.0, 6.9, -5.0), (1.0, 1.0, 0.9))
    T, S, RY = api.translation, api.scaling, api.rotation_y

    def cube(transform, **mat):
        c = api.cube().set_transform(transform)
        c.set_material(Material(**mat))
        w.push(c)

    cube(api.mul(S(20.0, 7.0, 20.0), T(0.0, 1.0, 0.1)),
         pattern=Pattern("checkers", BLACK, (0.25, 0.25, 0.25)).set_transform(S(0.07, 0.07, 0.07)),
         ambient=0.25, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.1)
    cube(S(10.0, 10.0, 10.0),
         pattern=Pattern("checkers", (0.4863, 0.3765, 0.2941), (0.3725, 0.2902, 0.2275)).set_transform(S(0.05, 20.0, 0.05)),
         ambient=0.1, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.1)
    cube(api.mul(T(0.0, 3.1, 0.0), S(3.0, 0.1, 2.0)),
         pattern=Pattern("stripe", (0.5529, 0.4235, 0.3255), (0.6588, 0.5098, 0.4000)).set_transform(
             api.mul(S(0.05, 0.05, 0.05), RY(0.1))),
         ambient=0.1, diffuse=0.7, specular=0.9, shininess=300.0, reflective=0.2)
    for x, z in ((2.7, -1.7), (2.7, 1.7), (-2.7, -1.7), (-2.7, 1.7)):
        cube(api.mul(T(x, 1.5, z), S(0.1, 1.5, 0.1)), color=(0.5529, 0.4235, 0.3255), ambient=0.2, diffuse=0.7)
    cube(api.mul(T(0.0, 3.45001, 0.0), RY(0.2), S(0.25, 0.25, 0.25)), color=(1.0, 1.0, 0.8), ambient=0.0, diffuse=0.3,
         specular=0.9, shininess=300.0, reflective=0.1, transparency=0.7, refractive_index=1.5)
    cube(api.mul(T(1.0, 3.35, -0.9), RY(-0.4), S(0.15, 0.15, 0.15)), color=(1.0, 0.5, 0.5), reflective=0.6, diffuse=0.4)
    cube(api.mul(T(-1.5, 3.27, 0.3), RY(0.4), S(0.15, 0.7, 0.15)), color=(1.0, 1.0, 0.5))
    cube(api.mul(T(0.0, 3.25, 1.0), RY(0.4), S(0.2, 0.05, 0.05)), color=(0.5, 1.0, 0.5))
    cube(api.mul(T(-0.6, 3.4, -1.0), RY(0.8), S(0.05, 0.2, 0.05)), color=(0.5, 0.5, 1.0))
    cube(api.mul(T(2.0, 3.4, 1.0), RY(0.8), S(0.05, 0.2, 0.05)), color=(0.5, 1.0, 1.0))
    cube(api.mul(T(-10.0, 4.0, 1.0), S(0.05, 1.0, 1.0)), color=(0.7098, 0.2471, 0.2196), diffuse=0.6)
    cube(api.mul(T(-10.0, 3.4, 2.7), S(0.05, 0.4, 0.4)), color=(0.2667, 0.2706, 0.6902), diffuse=0.6)
    cube(api.mul(T(-10.0, 4.6, 2.7), S(0.05, 0.4, 0.4)), color=(0.3098, 0.5961, 0.3098), diffuse=0.6)
    cube(api.mul(T(-2.0, 3.5, 9.95), S(5.0, 1.5, 0.05)), color=(0.3882, 0.2627, 0.1882), diffuse=0.7)
    cube(api.mul(T(-2.0, 3.5, 9.95), S(4.8, 1.4, 0.06)), color=BLACK, diffuse=0.0, ambient=0.0, specular=0.0,
         shininess=300.0, reflective=1.0)
    return cam, w, 5


CONFIGS = {"C1": c1_chapter7, "C2": c2_book_cover, "C3": c3_teapot, "C4": c4_cow, "C5": c5_pumpkin,
           "hexagon": hexagon, "table": table}
