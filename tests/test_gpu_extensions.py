"""-m gpu: the build's two opt-in extensions, through the C-ABI, against the oracle running
the same extension.  Neither is reference behaviour: with the flags off (every other test) planes are inert and
triangles take the reference's area test."""
import os

import numpy as np
import pytest

from conftest import SCENES
from parity import assert_rgb8_parity, raw_equal_mask

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_depth", [4, 0])
def test_planes_opt_in_matches_oracle(rt, oracle, gpu, max_depth):
    """RTX_F_TRACE_PLANES: intersectplane (raytracing.cpp:275-287) after the polygons, the triangle's facing rule and
    shading forms, no shadows from planes -- the same definition in oracle/rt_oracle.c (ORC_TRACE_PLANES)."""
    sc = rt.load_scene(os.path.join(SCENES, "planes.txt"))
    assert sc.view.nplanes == 3
    W, H = 160, 120
    p = rt.Params.make(W, H, max_depth, flags=rt.host.F_TRACE_PLANES)
    g, st = rt.render_f32(sc, p)
    with oracle.flags(oracle.TRACE_PLANES):
        c, cnt = oracle.render_f32(sc, W, H, max_depth)
        rc, _ = oracle.render_rgb8(sc, W, H, max_depth)
    assert st.rays_primary + st.rays_secondary == cnt.trace_calls and st.rays_shadow == cnt.shadow_rays and st.hits == cnt.hits
    assert st.hits >= W * H                      # a plane behind everything: no primary ray misses
    same = raw_equal_mask(g, c)
    assert same.mean() >= 0.999, int((~same).sum())
    rg, _ = rt.render_rgb8(sc, p)
    assert_rgb8_parity(rg, rc, "planes")
    # off by default: the planes are inert, as in the reference
    g0, st0 = rt.render_f32(sc, rt.Params.make(W, H, max_depth))
    c0, _ = oracle.render_f32(sc, W, H, max_depth)
    assert raw_equal_mask(g0, c0).mean() >= 0.999 and st0.hits < W * H


def test_planes_culled_equals_exact(rt, gpu):
    sc = rt.load_scene(os.path.join(SCENES, "planes.txt"))
    for (W, H) in ((800, 450), (200, 120)):
        a, sa = rt.render_f32(sc, rt.Params.make(W, H, 4, flags=rt.host.F_TRACE_PLANES))
        b, sb = rt.render_f32(sc, rt.Params.make(W, H, 4, flags=rt.host.F_TRACE_PLANES | rt.host.F_EXACT_ONLY))
        assert raw_equal_mask(a, b).all()
        assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


@pytest.mark.parametrize("name", ["room.txt", "probe.txt"])
def test_moller_trumbore_mode_matches_its_oracle(rt, oracle, gpu, name):
    """RTX_F_TRI_MOLLER_TRUMBORE against the oracle's restatement of the same arithmetic (ORC_TRI_MT)."""
    sc = rt.load_scene(os.path.join(SCENES, name))
    W, H = 160, 120
    p = rt.Params.make(W, H, 4, flags=rt.host.F_TRI_MOLLER_TRUMBORE)
    g, st = rt.render_f32(sc, p)
    with oracle.flags(oracle.TRI_MT):
        c, cnt = oracle.render_f32(sc, W, H, 4)
        rc, _ = oracle.render_rgb8(sc, W, H, 4)
    assert st.rays_primary + st.rays_secondary == cnt.trace_calls and st.rays_shadow == cnt.shadow_rays
    assert raw_equal_mask(g, c).mean() >= 0.999
    rg, _ = rt.render_rgb8(sc, p)
    assert_rgb8_parity(rg, rc, name + " MT")


def test_moller_trumbore_is_a_non_parity_mode(rt, gpu, tmp_path):
    """The mode is opt-in because it is NOT the reference's test (area sum within an absolute 0.01, raytracing.cpp:362):
    report how many pixels of the 10k-primitive scene change, and check its own culling (bounding spheres sized for the
    reference's larger accept region stay conservative) against its exact-only kernels."""
    p = tmp_path / "c4.txt"
    rt.scenegen.write_c4(str(p))
    sc = rt.load_scene(str(p))
    W, H = 960, 540
    ref, _ = rt.render_rgb8(sc, rt.Params.make(W, H, 4))
    mt, sm = rt.render_rgb8(sc, rt.Params.make(W, H, 4, flags=rt.host.F_TRI_MOLLER_TRUMBORE))
    mtx, sx = rt.render_rgb8(sc, rt.Params.make(W, H, 4, flags=rt.host.F_TRI_MOLLER_TRUMBORE | rt.host.F_EXACT_ONLY))
    assert np.array_equal(mt, mtx) and (sm.rays_secondary, sm.rays_shadow, sm.hits) == (sx.rays_secondary, sx.rays_shadow, sx.hits)
    differ = float((ref != mt).any(axis=2).mean())
    print("Moller-Trumbore vs the reference's area test, C4 at %dx%d: %.3f %% of pixels differ (NON-PARITY mode)" % (W, H, differ * 100))
    assert 0.0 < differ < 0.10


@pytest.mark.parametrize("name", ["inp.txt", "inp-refl.txt", "inpaffine.txt", "inp1.txt", "inp01.txt"])
def test_legacy_grammar_scenes_match_oracle(rt, oracle, gpu, name):
    """The bundled scenes as their authors wrote them (opt-in rtx_scene_load_legacy): spheres, the affine sphere of
    inpaffine.txt, the triangle room, the cuboid, the polygon, reflection and refraction all really exercised, GPU
    against the oracle on the same records; inp1.txt also with its six planes traced (RTX_F_TRACE_PLANES)."""
    from conftest import GOLDEN
    sc = rt.load_scene(os.path.join(GOLDEN, "bundled", name), legacy=True)
    assert sc.view.nspheres >= 1 and sc.view.nlights == 1
    W, H = 160, 120
    for flags, oflags in ((0, 0),) + (((rt.host.F_TRACE_PLANES, oracle.TRACE_PLANES),) if sc.view.nplanes else ()):
        g, st = rt.render_f32(sc, rt.Params.make(W, H, 4, flags=flags))
        with oracle.flags(oflags):
            c, cnt = oracle.render_f32(sc, W, H, 4)
            rc, _ = oracle.render_rgb8(sc, W, H, 4)
        assert st.rays_primary + st.rays_secondary == cnt.trace_calls and st.rays_shadow == cnt.shadow_rays and st.hits == cnt.hits
        assert st.hits > 500
        same = raw_equal_mask(g, c)
        assert same.mean() >= 0.999, (name, int((~same).sum()))
        rg, _ = rt.render_rgb8(sc, rt.Params.make(W, H, 4, flags=flags))
        assert_rgb8_parity(rg, rc, name + " (legacy grammar)")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 3, 5])
def test_pooled_shadow_stage_equals_exact(rt, gpu, tmp_path, monkeypatch, seed):
    """The pooled plain-primitive stage of the shadow pass (rtx_shadow.hip.h, shadow_plain_pool: generations >= 1 by
    default, here EVERY generation with RTX_SHADOW_POOL_FROM=0) gives the image of the exact-only kernels bit for bit:
    random scenes with affine spheres (fused thin-generation kernel) and a triangle-only scene above 2^20 shadow rays
    (unsplit launch); then the same scenes with the stage off everywhere (the one-at-a-time walk in every generation)."""
    import numpy as np
    from parity import raw_equal_mask
    monkeypatch.setenv("RTX_SHADOW_POOL_FROM", "0")
    rng = np.random.default_rng(100 + seed)
    p = tmp_path / ("pool_%d.txt" % seed)
    lights = [(float(rng.uniform(-45, 45)), float(rng.uniform(-30, 45)), float(rng.uniform(-130, 20)), 0.3, 0.3, 0.3) for _ in range(3)]
    rt.scenegen.write_scene(str(p), 0 if seed == 5 else 1500, 3000, lights, seed=2000 + seed, frac_affine=0.0 if seed == 5 else 0.3,
                            frac_reflective=0.5, frac_refractive=0.1, affine="general", radius=(0.3, 3.0), edge=4.0)
    sc = rt.load_scene(str(p))
    W, H = (960, 540) if seed == 5 else (480, 270)
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, 3))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, 3, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    c, sc_ = rt.render_f32(sc, rt.Params.make(W, H, 3, flags=rt.host.F_COUNT_EXEC))  # counting instantiation of the pooled kernels
    assert raw_equal_mask(a, c).all()
    monkeypatch.setenv("RTX_SHADOW_POOL_FROM", "99")
    w, sw = rt.render_f32(sc, rt.Params.make(W, H, 3))
    assert raw_equal_mask(a, w).all() and sw.rays_shadow == sa.rays_shadow


def test_calibrated_queues_equal_worst_case_queues(rt, gpu, tmp_path):
    """Ray queues sized by the calibration pass (the default) against queues sized for fanout^k rays per pixel
    (RTX_F_WORST_CASE_QUEUES): same image, same ray counts, nothing dropped -- on a scene whose primitives both reflect
    and refract (fan-out 2), at a depth where the worst case is 31 x the pixel count."""
    from parity import raw_equal_mask
    p = tmp_path / "fan2.txt"
    rt.scenegen.write_scene(str(p), 600, 600, rt.scenegen.lights_c4(), seed=77, frac_affine=0.2, frac_reflective=0.7,
                            frac_refractive=0.5, affine="diag", radius=(1.0, 4.0), edge=8.0)
    sc = rt.load_scene(str(p))
    W, H = 480, 270
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, 4))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, 4, flags=rt.host.F_WORST_CASE_QUEUES))
    assert raw_equal_mask(a, b).all()
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert sa.queue_overflow == 0 and sb.queue_overflow == 0
    assert sa.rays_secondary > W * H  # the scene really fans out
