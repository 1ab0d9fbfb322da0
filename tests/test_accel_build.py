"""CPU test of the flatten-time acceleration builder (csrc/rtc_accel.hpp, host code of the HIP library).

The device addresses triangle records arithmetically (cluster c of a run starts at c * CLUSTER_TRIS, real triangles
come before padding) and trusts every cluster / tile box to contain its triangles; tests/cpp/accel_invariants.cpp
checks exactly that on random, huge, tiny and degenerate triangle soups of many sizes."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_accel_builder_invariants(tmp_path):
    exe = tmp_path / "accel_invariants"
    src = os.path.join(REPO, "tests", "cpp", "accel_invariants.cpp")
    inc = os.path.join(REPO, "ray-tracer-challenge-rust_amd", "csrc")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I", inc, src, "-o", str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:]
