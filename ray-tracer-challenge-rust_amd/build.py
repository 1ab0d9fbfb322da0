"""Build the native libraries of the MI355X renderer in-tree.

  csrc/librtc_hip.so   hipcc --offload-arch=gfx950   (kernels + the C ABI of include/rtc_hip.h)
  host/librtc_host.so  g++                            (C++ scene API, flattener, OBJ/PPM; dlopens librtc_hip.so)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build container; the .so files travel
to the GPU box with the snapshot.  -ffp-contract=off everywhere: the f64 arithmetic must keep the reference's
operation order (no FMA contraction) for the parity guarantee described in DESIGN.md.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")

HIP_LIB = os.path.join(CSRC, "librtc_hip.so")
HOST_LIB = os.path.join(HOST, "librtc_host.so")

HIP_SOURCES = ["rtc_hip.hip", "rtc_device.hpp", "rtc_accel.hpp", "rtc_multi.hpp", os.path.join("..", "..", "include", "rtc_hip.h")]
HOST_SOURCES = ["rtc_host.cpp", "flatten.cpp", "rtc_host_capi.cpp"]
HOST_HEADERS = ["rtc_host.hpp", "rtc_host_capi.h", os.path.join("..", "..", "include", "rtc_hip.h")]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd: list[str], cwd: str) -> None:
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=cwd, check=True)


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def build_hip(force: bool = False, extra_flags: list[str] | None = None) -> str:
    deps = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    if force or _stale(HIP_LIB, deps):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function", "-o", HIP_LIB, "rtc_hip.hip"] + (extra_flags or [])
        _run(cmd, CSRC)
    return HIP_LIB


def build_host(force: bool = False) -> str:
    deps = [os.path.join(HOST, s) for s in HOST_SOURCES + HOST_HEADERS]
    if force or _stale(HOST_LIB, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared", "-Wall", "-Wextra",
               "-o", HOST_LIB] + HOST_SOURCES + ["-ldl"]
        _run(cmd, HOST)
    return HOST_LIB


EXAMPLE_BIN = os.path.join(HOST, "rtc_example")


def build_example(force: bool = False) -> str:
    """host/rtc_example: a C++ program written against rtc_host.hpp (links librtc_host.so, rpath $ORIGIN)."""
    deps = [os.path.join(HOST, "example_main.cpp"), os.path.join(HOST, "rtc_host.hpp"), HOST_LIB]
    if force or _stale(EXAMPLE_BIN, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wextra", "-o", EXAMPLE_BIN, "example_main.cpp",
               "-L.", "-lrtc_host", "-Wl,-rpath,$ORIGIN", "-ldl"]
        _run(cmd, HOST)
    return EXAMPLE_BIN


def anything_stale() -> bool:
    """True if a library is missing or older than one of its sources."""
    return (_stale(HIP_LIB, [os.path.join(CSRC, s) for s in HIP_SOURCES]) or
            _stale(HOST_LIB, [os.path.join(HOST, s) for s in HOST_SOURCES + HOST_HEADERS]))


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_host(force)
    build_example(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
