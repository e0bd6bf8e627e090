"""The C-ABI library loads on a GPU-less host and exports every symbol include/rtx.h declares;
struct layouts match the reference's PODs; compute entry points fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, SCENES


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(rt):
    lib = C.CDLL(rt.host.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "librtx.so does not export %s" % n
        assert n in rt.host.SYMBOLS, "host.py does not bind %s" % n
    assert sorted(rt.host.SYMBOLS) == names


def test_struct_layouts_match_reference(rt):
    h = rt.host
    # raytracing.cpp:15-68: vec 12, matrix 48, sphere 120, plane 40, triangle 88, light 24
    assert [C.sizeof(t) for t in (h.Vec3, h.XMatrix, h.Sphere, h.Plane, h.Triangle, h.Light)] == [12, 48, 120, 40, 88, 24]
    assert C.sizeof(h.Polygon) == 84  # rt.cpp:71-80 with the vertex vector flattened
    assert h.Sphere.istransformed.offset == 12 and h.Sphere.radius.offset == 16 and h.Sphere.mat.offset == 72
    assert h.Triangle.color.offset == 36 and h.Triangle.reflectivity.offset == 72


def test_abi_version_and_errors(rt):
    L = rt.lib()
    assert L.rtx_abi_version() == 0x00030300  # 3.3: rtx_stats.shadow_form, RTX_F_SHADOW_TABLES / _PACKETS, RTX_F_MULTI_ON_ONE_DEVICE; 3.0: capacities on the diagnostic entry points, rtx_render_rgb8_noalloc, executed-work counters in rtx_stats; 3.1: rtx_release_cached_context, RTX_F_NO_CACHE; 3.2: rtx_cluster_records
    assert b"no HIP device" in L.rtx_strerror(-2)
    assert L.rtx_strerror(0) == b"ok"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_compute_fails_loudly_without_device(rt):
    assert rt.device_count() == 0
    sc = rt.load_scene(os.path.join(SCENES, "probe.txt"))
    with pytest.raises(rt.RtxError) as e:
        rt.render_rgb8(sc, rt.Params.make(16, 8, 4))
    assert e.value.code == -2
    with pytest.raises(rt.RtxError):
        rt.Context(sc, rt.Params.make(16, 8, 4))


def test_invalid_arguments(rt):
    sc = rt.load_scene(os.path.join(SCENES, "probe.txt"))
    for bad in (rt.Params.make(0, 8, 4), rt.Params.make(8, -1, 4), rt.Params.make(8, 8, -1), rt.Params.make(8, 8, 4, nranks=2, rank=2)):
        with pytest.raises(rt.RtxError) as e:
            rt.render_f32(sc, bad) if bad.nranks <= 1 else rt.Context(sc, bad)
        assert e.value.code == -1
    with pytest.raises(rt.RtxError) as e:
        rt.Context(sc, rt.Params.make(8, 8, 40))
    assert e.value.code == -6
    with pytest.raises(rt.RtxError) as e:
        rt.load_scene("/nonexistent/scene.txt")
    assert e.value.code == -5
