"""rtc_render's resident-scene cache and the in-process multi-device entry points (rtc_multi_* / rtc_render_multi).

One GPU box: the multi-device path is exercised with the SAME device listed several times, which the peer-DMA
transport (RTC_FLAG_GATHER_P2P) accepts — every part of the path runs (block dealing, chunked renders, per-"device"
buffers, copies to the root, block-wise download in Canvas order) except the xGMI hop itself.  RCCL refuses a
communicator with one device twice, so its transport is covered by what can be checked without a second GPU: the
library loads and every entry point used resolves (CPU test), duplicates are reported as such, and a one-device
"multi" render works with the RCCL flag (no communicator is needed for one device)."""
import ctypes as C

import numpy as np
import pytest

import rtc_amd
from rtc_amd import abi, scenes


def test_rccl_loads_and_every_entry_point_resolves():
    """CPU: no communicator is created and no device is touched."""
    lib = rtc_amd.hip()
    lib.rtc_debug_rccl_symbols.restype = C.c_int
    assert lib.rtc_debug_rccl_symbols() == 7, lib.rtc_last_error_string()


def test_multi_entry_points_validate_arguments_without_a_gpu():
    lib = rtc_amd.hip()
    out = C.c_void_p()
    assert lib.rtc_multi_create(None, 1, (C.c_int * 1)(0), C.byref(out)) == abi.RTC_ERR_INVALID_ARG
    assert lib.rtc_render_multi(None, None, None, 0, None, None, None, None) == abi.RTC_ERR_INVALID_ARG
    lib.rtc_release_cached()  # nothing cached: must be a no-op
    lib.rtc_multi_destroy(None)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,devices", [
    ("C3", dict(hsize=320, vsize=180), [0]),
    ("C3", dict(hsize=320, vsize=180), [0, 0]),
    ("C3", dict(hsize=161, vsize=97), [0, 0, 0]),       # 7 blocks, the last one partial, over 3 "devices"
    ("C5", dict(hsize=240, vsize=135, max_faces=2000), [0, 0, 0, 0]),
    ("hexagon", dict(hsize=200, vsize=40), [0, 0, 0, 0]),  # 3 blocks over 4 "devices": one gets nothing
    ("C2", dict(hsize=64, vsize=8), [0, 0]),               # a single partial block
])
def test_multi_device_frame_is_bit_identical_to_one_device(gpu_device, name, kw, devices):
    cam, w, L = scenes.CONFIGS[name](rtc_amd.api, **kw)
    one = cam.render(w, L, 0, stats=True)
    multi = cam.render(w, L, devices=devices, p2p=True, stats=True)
    assert np.array_equal(one.rgb64.view(np.uint64), multi.rgb64.view(np.uint64))
    assert np.array_equal(one.rgb32.view(np.uint32), multi.rgb32.view(np.uint32))
    for k in ("radiance_rays", "shadow_rays", "culled_rays"):
        assert one.stats[k] == multi.stats[k], k
    quiet = cam.render(w, L, devices=devices, p2p=True)  # the uncounted kernels, replicas now cached
    assert np.array_equal(one.rgb64.view(np.uint64), quiet.rgb64.view(np.uint64))
    rtc_amd.hip().rtc_release_cached()


@pytest.mark.gpu
def test_rccl_transport_with_one_device_and_its_duplicate_check(gpu_device):
    cam, w, L = scenes.c3_teapot(rtc_amd.api, 160, 90)
    one = cam.render(w, L, 0)
    multi = cam.render(w, L, devices=[0])  # RCCL transport selected; one device needs no communicator
    assert np.array_equal(one.rgb64.view(np.uint64), multi.rgb64.view(np.uint64))
    with pytest.raises(rtc_amd.NativeError) as e:
        cam.render(w, L, devices=[0, 0])
    assert e.value.code == abi.RTC_ERR_INVALID_ARG and "twice" in str(e.value)
    with pytest.raises(rtc_amd.NativeError) as e:
        cam.render(w, L, devices=[0, 99], p2p=True)
    assert e.value.code == abi.RTC_ERR_NO_DEVICE
    rtc_amd.hip().rtc_release_cached()


@pytest.mark.gpu
def test_rtc_render_keeps_the_scene_resident_and_notices_a_changed_scene(gpu_device):
    from rtc_amd._capi import Material

    cam, w, L = scenes.c3_teapot(rtc_amd.api, 256, 144)
    rtc_amd.hip().rtc_release_cached()
    a = cam.render(w, L, 0, stats=True)
    b = cam.render(w, L, 0, stats=True)
    assert a.stats["upload_ms"] > 0.0 and b.stats["upload_ms"] == 0.0  # the second call found the scene in HBM
    assert np.array_equal(a.rgb64.view(np.uint64), b.rgb64.view(np.uint64))
    # the same world with another light is another scene: must not be served from the cache
    from rtc_amd import meshes
    api = rtc_amd.api
    w2 = api.world((3.0, 6.9, -5.0), (1.0, 1.0, 0.9))
    teapot = api.obj_parse(meshes.obj_text("teapot"))
    teapot.set_transform(api.translation(0.0, -1.5, 0.0))
    teapot.set_material(Material(pattern=rtc_amd.Pattern("gradient", (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))))
    w2.push(teapot)
    c = cam.render(w2, L, 0, stats=True)
    assert c.stats["upload_ms"] > 0.0 and not np.array_equal(a.rgb64, c.rgb64)
    # more scenes than the cache holds, then the first one again: evicted, re-uploaded, same pixels
    for i in range(5):
        wi = rtc_amd.api.world((float(i), 5.0, -5.0))
        wi.push(rtc_amd.api.sphere().set_material(Material(color=(0.1 * i, 0.5, 0.5))))
        rtc_amd.api.camera(32, 32, 1.0).render(wi, 5, 0)
    d = cam.render(w, L, 0, stats=True)
    assert d.stats["upload_ms"] > 0.0 and np.array_equal(a.rgb64.view(np.uint64), d.rgb64.view(np.uint64))
    rtc_amd.hip().rtc_release_cached()
    e = cam.render(w, L, 0, stats=True)
    assert e.stats["upload_ms"] > 0.0 and np.array_equal(a.rgb64.view(np.uint64), e.rgb64.view(np.uint64))


@pytest.mark.gpu
def test_rtc_render_with_only_one_of_the_two_framebuffers(gpu_device):
    lib = rtc_amd.hip()
    cam, w, L = scenes.c3_teapot(rtc_amd.api, 96, 54)
    ref = cam.render(w, L, 0)
    sc = w.upload(0)  # only to obtain the flattened description the same way the host library does
    sc.close()
    # through the host library: f64 only / f32 only
    import rtc_amd as R
    st = None
    rgb64 = np.zeros((54, 96, 3))
    rc = R._host.rtch_render(cam._h, w._h, L, 0, 0, None, R._dptr(rgb64), st)
    assert rc == 0 and np.array_equal(rgb64.view(np.uint64), ref.rgb64.view(np.uint64))
    rgb32 = np.zeros((54, 96, 3), dtype=np.float32)
    rc = R._host.rtch_render(cam._h, w._h, L, 0, 0, rgb32.ctypes.data_as(R._fp), None, st)
    assert rc == 0 and np.array_equal(rgb32, ref.rgb32)
    lib.rtc_release_cached()
