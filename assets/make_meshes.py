"""Regenerate assets/meshes/*.npz from the reference checkout's OBJ model data.

The four meshes under /root/reference/objs are plain vertex/face data (inputs of the reference's cow and
teapot scenes, main.rs:340,380, plus two unreferenced assets).  The reference tree does not exist on the GPU
box, so the numbers are stored here as compact arrays: `v` float64 [n,3] (each coordinate parsed with Python's
correctly-rounded float(), the same double Rust's `parse::<f64>` yields) and `f` int32 [m,k] one-based vertex
indices per face record, in file order.  rtc_amd/meshes.py turns them back into OBJ text (repr() round-trips
doubles exactly), so the product's OBJ parser sees the same records the reference's parser would.

Only `v` and `f` records exist in these files (no g/vn/vt, no '/'): checked below.
"""
import os
import sys

import numpy as np

SRC = "/root/reference/objs"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "meshes")
NAMES = {"teapot.obj": "teapot", "cow-nonormals.obj": "cow", "pumpkin_tall_10k.obj": "pumpkin", "teddy.obj": "teddy"}


def main():
    os.makedirs(DST, exist_ok=True)
    for fname, name in NAMES.items():
        verts, faces, other = [], [], 0
        with open(os.path.join(SRC, fname)) as fh:
            for line in fh:
                tok = line.split()
                if not tok:
                    continue
                if tok[0] == "v":
                    verts.append([float(t) for t in tok[1:4]])
                elif tok[0] == "f":
                    assert all("/" not in t for t in tok[1:]), line
                    faces.append([int(t) for t in tok[1:]])
                else:
                    other += 1
        k = {len(f) for f in faces}
        assert len(k) == 1, f"{fname}: mixed face arity {k}"
        v = np.array(verts, dtype=np.float64)
        f = np.array(faces, dtype=np.int32)
        np.savez_compressed(os.path.join(DST, name + ".npz"), v=v, f=f, ignored_lines=np.int64(other))
        print(f"{name}: {len(v)} vertices, {len(f)} faces (arity {k.pop()}), {other} other records,"
              f" bbox {v.min(0)} .. {v.max(0)}")


if __name__ == "__main__":
    sys.exit(main())
