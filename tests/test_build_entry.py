"""__graft_entry__.ensure_built() must never let a test or bench run against libraries older than their sources
(round 1: a failed rebuild was swallowed as long as some .so existed)."""
import importlib.util
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "ray-tracer-challenge-rust_amd")


def _build_module():
    spec = importlib.util.spec_from_file_location("rtc_amd_build_t", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_staleness_is_noticed_for_every_source_of_both_libraries():
    b = _build_module()
    assert not b.anything_stale(), "the session fixture has just built everything"
    for rel in ("csrc/rtc_hip.hip", "csrc/rtc_device.hpp", "csrc/rtc_accel.hpp", "csrc/rtc_multi.hpp", "host/flatten.cpp",
                "host/rtc_host.hpp", os.path.join("..", "include", "rtc_hip.h")):
        path = os.path.join(PKG, rel)
        st = os.stat(path)
        try:
            newest = max(os.path.getmtime(b.HIP_LIB), os.path.getmtime(b.HOST_LIB)) + 10
            os.utime(path, (newest, newest))
            assert b.anything_stale(), f"{rel} newer than the libraries went unnoticed"
        finally:
            os.utime(path, (st.st_atime, st.st_mtime))
    assert not b.anything_stale()


def test_ensure_built_reraises_a_compile_error(monkeypatch):
    import __graft_entry__ as ge

    def boom(force=False):
        raise RuntimeError("hipcc: error: simulated compile failure")

    class FakeBuild:
        HIP_LIB, HOST_LIB = "x", "y"
        build_all = staticmethod(boom)
        anything_stale = staticmethod(lambda: False)

    monkeypatch.setattr(ge, "_load_build_module", lambda: FakeBuild)
    try:
        ge.ensure_built()
    except RuntimeError as e:
        assert "simulated compile failure" in str(e)
    else:
        raise AssertionError("a compile error was swallowed")
    # the only tolerated failure: no compiler on a box that received up-to-date prebuilt libraries

    def no_compiler(force=False):
        raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")

    FakeBuild.build_all = staticmethod(no_compiler)
    ge.ensure_built()  # does not raise: nothing is stale
    FakeBuild.anything_stale = staticmethod(lambda: True)
    try:
        ge.ensure_built()
    except RuntimeError:
        pass
    else:
        raise AssertionError("stale libraries without a compiler were accepted")
