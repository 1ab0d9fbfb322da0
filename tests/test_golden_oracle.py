"""CPU: the oracle against the committed golden framebuffers (tests/golden/*.npz, made by make_goldens.py), and the
oracle's two cost modes against each other.  Keeps the checker honest between rounds: an accidental change to
oracle/ or to a scene definition shows up here, without a GPU."""
import glob
import os

import numpy as np
import pytest

import golden_util
import oracle_binding as ob

GOLDENS = sorted(glob.glob(os.path.join(golden_util.GOLDEN_DIR, "*.npz")))


def test_golden_set_is_complete():
    names = {os.path.basename(g)[:-4] for g in GOLDENS}
    assert {"default_world_11x11", "c1_200x100", "c2_96x54", "c3_64x36", "c4_64x36", "c5_48x27", "hexagon_80x40",
            "table_80x40", "c3_1920x1080_s40", "c4_3840x2160_s80", "c5_3840x2160_s120"} <= names


@pytest.mark.parametrize("path", GOLDENS, ids=[os.path.basename(g)[:-4] for g in GOLDENS])
def test_oracle_reproduces_golden_bit_for_bit(orc, path):
    g = golden_util.load(path)
    cam, world, L = golden_util.build(g, orc)
    assert L == g["recursion_limit"]
    if g["xs"] is None:
        rgb, info = cam.render(world, L, nthreads=8)
    else:
        rgb, info = cam.render_pixels(world, g["xs"], g["ys"], L, nthreads=8)
    assert (rgb.view(np.uint64) == g["rgb"].view(np.uint64)).all()
    assert info["radiance_rays"] == g["radiance_rays"] and info["shadow_rays"] == g["shadow_rays"]


def test_camera_rs_golden_pixel_in_fixture():
    g = golden_util.load(os.path.join(golden_util.GOLDEN_DIR, "default_world_11x11.npz"))
    assert np.abs(g["rgb"][5, 5] - [0.38066, 0.47583, 0.2855]).max() < 1e-5  # camera.rs:145-155


@pytest.mark.parametrize("name", ["c2_96x54", "c3_64x36", "hexagon_80x40", "table_80x40"])
def test_faithful_cost_mode_gives_the_same_doubles(orc, name):
    """ORC_MODE_FAITHFUL (per-ray cofactor inverse per leaf + per-ray Bounds::new: the reference's cost model,
    used as bench.py's cpu_baseline) vs ORC_MODE_CACHED (used for goldens): identical colours and ray counts, on a
    pixel subsample (faithful mode is thousands of times slower on meshes)."""
    g = golden_util.load(os.path.join(golden_util.GOLDEN_DIR, name + ".npz"))
    cam, world, L = golden_util.build(g, orc)
    step = 7 if name.startswith(("c3", "c4")) else 3
    xs, ys = np.meshgrid(np.arange(1, cam.hsize, step), np.arange(1, cam.vsize, step))
    xs, ys = xs.ravel(), ys.ravel()
    a, ia = cam.render_pixels(world, xs, ys, L, mode=ob.MODE_CACHED)
    b, ib = cam.render_pixels(world, xs, ys, L, mode=ob.MODE_FAITHFUL, nthreads=8)
    assert (a.view(np.uint64) == b.view(np.uint64)).all()
    assert (ia["radiance_rays"], ia["shadow_rays"]) == (ib["radiance_rays"], ib["shadow_rays"])
    assert (a.view(np.uint64) == g["rgb"][ys, xs].view(np.uint64)).all()
