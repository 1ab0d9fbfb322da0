"""Helpers for the golden framebuffers in tests/golden/ (see make_goldens.py)."""
import math
import os
import re

import numpy as np

from rtc_amd import scenes

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(path):
    z = np.load(path, allow_pickle=False)
    return {"path": path, "rgb": z["rgb"], "xs": z["xs"] if "xs" in z else None, "ys": z["ys"] if "ys" in z else None,
            "recursion_limit": int(z["recursion_limit"]), "radiance_rays": int(z["radiance_rays"]),
            "shadow_rays": int(z["shadow_rays"]), "scene": str(z["scene"])}


def build(g, api):
    """Rebuild the scene a golden was rendered from, on the given scene API."""
    stem = os.path.basename(g["path"])[:-4]
    if stem == "default_world_11x11":
        cam = api.camera(11, 11, math.pi / 2.0)
        cam.set_transform(api.view_transform((0.0, 0.0, -5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
        return cam, api.default_world(), 5
    m = re.match(r"(c[1-5]|hexagon|table)_(\d+)x(\d+)(_s\d+)?$", stem)
    assert m, stem
    name = m.group(1).upper() if m.group(1).startswith("c") else m.group(1)
    return scenes.CONFIGS[name](api, hsize=int(m.group(2)), vsize=int(m.group(3)))
