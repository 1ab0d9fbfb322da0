"""Mesh assets (assets/meshes/*.npz, made by assets/make_meshes.py) -> OBJ text for Parser::from_obj_str."""
from __future__ import annotations

import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MESH_DIR = os.path.join(REPO, "assets", "meshes")
NAMES = ("teapot", "cow", "pumpkin", "teddy")


def load(name: str):
    """-> (v float64 [n,3], f int32 [m,k] one-based)"""
    z = np.load(os.path.join(MESH_DIR, name + ".npz"))
    return z["v"], z["f"]


def obj_text(name: str, max_faces: int | None = None) -> str:
    """OBJ text with only `v` and `f` records; repr(float) round-trips every double exactly."""
    v, f = load(name)
    if max_faces is not None:
        f = f[:max_faces]
    lines = [f"v {float(x)!r} {float(y)!r} {float(z)!r}" for x, y, z in v]
    lines += ["f " + " ".join(str(int(i)) for i in face) for face in f]
    return "\n".join(lines) + "\n"
