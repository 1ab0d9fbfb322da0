"""The C-ABI library loads and exports every symbol include/rtc_hip.h declares (no compute calls: CPU-only)."""
import os
import re

import rtc_amd
from rtc_amd import abi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    declared = _declared(os.path.join(REPO, "include", "rtc_hip.h"))
    assert declared == sorted(abi.ABI_SYMBOLS), "abi.ABI_SYMBOLS is out of date with include/rtc_hip.h"
    lib = rtc_amd.hip()
    for sym in declared:
        assert hasattr(lib, sym), f"librtc_hip.so does not export {sym}"
    assert lib.rtc_abi_version() == abi.RTC_ABI_VERSION == 3
    assert isinstance(lib.rtc_last_error_string(), bytes)


def test_host_capi_symbols_are_exported():
    import ctypes as C

    text = open(os.path.join(REPO, "ray-tracer-challenge-rust_amd", "host", "rtc_host_capi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rtch_[a-z0-9_]+)\s*\(", text)))
    lib = C.CDLL(abi.HOST_LIB_PATH)
    assert len(declared) >= 40
    for sym in declared:
        assert hasattr(lib, sym), f"librtc_host.so does not export {sym}"


def test_struct_layouts_match_the_header():
    """ctypes mirrors must have the C sizes (guards against a silent field drift)."""
    import ctypes as C

    assert C.sizeof(abi.RtcMaterial) == 11 * 8
    assert C.sizeof(abi.RtcPattern) == 8 + 6 * 8 + 16 * 8
    assert C.sizeof(abi.RtcCameraDesc) == 8 + 3 * 8 + 16 * 8
    assert C.sizeof(abi.RtcRenderOpts) == 24
    assert C.sizeof(abi.RtcStats) == 11 * 8


def test_no_device_is_reported_not_faked():
    """On a machine without a GPU the product must say so (no CPU fallback).  On a GPU box this test only checks
    that device_count is positive."""
    n = rtc_amd.device_count()
    if n == 0:
        import pytest

        with pytest.raises(rtc_amd.NativeError) as e:
            rtc_amd.api.default_world().upload(0)
        assert e.value.code == abi.RTC_ERR_NO_DEVICE
        cam = rtc_amd.api.camera(4, 4, 1.0)
        with pytest.raises(rtc_amd.NativeError):
            cam.render(rtc_amd.api.default_world())
    else:
        assert n >= 1
