"""-m gpu: the HIP path, called through the C-ABI, against the CPU oracle on the same inputs."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT, SCENES
from parity import assert_rgb8_parity, raw_equal_mask

pytestmark = pytest.mark.gpu

SCENE_FILES = sorted(glob.glob(os.path.join(SCENES, "*.txt")))


@pytest.mark.parametrize("path", SCENE_FILES, ids=[os.path.basename(p) for p in SCENE_FILES])
@pytest.mark.parametrize("max_depth", [4, 0, 2])
def test_scene_matches_oracle(rt, oracle, gpu, path, max_depth):
    W, H = 160, 120
    sc = rt.load_scene(path)
    p = rt.Params.make(W, H, max_depth)
    raw_gpu, st = rt.render_f32(sc, p)
    raw_cpu, cnt = oracle.render_f32(sc, W, H, max_depth)
    same = raw_equal_mask(raw_gpu, raw_cpu)
    # ray accounting must agree exactly with the recursion's
    assert st.rays_primary + st.rays_secondary == cnt.trace_calls
    assert st.rays_shadow == cnt.shadow_rays
    assert st.hits == cnt.hits
    rgb_gpu, _ = rt.render_rgb8(sc, p)
    rgb_cpu, _ = oracle.render_rgb8(sc, W, H, max_depth)
    r = assert_rgb8_parity(rgb_gpu, rgb_cpu, os.path.basename(path))
    print(f"{os.path.basename(path)} md={max_depth}: raw bit-identical {same.mean()*100:.4f}% "
          f"({int((~same).sum())} px differ), rgb8 {r}")
    # float image: anything but the powf last-ulp cases must be bit-identical
    assert same.mean() >= 0.999, (os.path.basename(path), int((~same).sum()))


@pytest.fixture(scope="module")
def c4_scene(rt, tmp_path_factory):
    """The benchmark's 10k-primitive scene (BASELINE.json configs[3]) with a 5 % refractive share,
    so both kinds of secondary ray exist."""
    p = tmp_path_factory.mktemp("c4") / "c4.txt"
    rt.scenegen.write_c4(str(p), frac_refractive=0.05)
    return rt.load_scene(str(p))


def test_filter_is_exact_on_c4(rt, gpu, c4_scene):
    """Size-independent property: the bounding-sphere filter may only skip work, never change a
    result.  Filtered and exact-only renders of the 10k-primitive scene must agree bit for bit."""
    W, H = 960, 540
    a, sa = rt.render_f32(c4_scene, rt.Params.make(W, H, 4))
    b, sb = rt.render_f32(c4_scene, rt.Params.make(W, H, 4, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "filtered and exact-only float images differ on %d pixels" % int((~same).sum())
    assert (sa.rays_primary, sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_primary, sb.rays_secondary, sb.rays_shadow, sb.hits)
    ra, _ = rt.render_rgb8(c4_scene, rt.Params.make(W, H, 4))
    rb, _ = rt.render_rgb8(c4_scene, rt.Params.make(W, H, 4, flags=rt.host.F_EXACT_ONLY))
    assert np.array_equal(ra, rb)


def test_c4_crop_matches_oracle(rt, oracle, gpu, c4_scene):
    """The 10k-primitive scene at 4K geometry, rows 1000..1004 (5 x 3840 rays), against the oracle."""
    W, H = 3840, 2160
    row0, row1 = 1000, 1004
    raw_cpu, cnt = oracle.render_f32(c4_scene, W, H, 4, row0, row1)
    # one band that contains exactly those rows: band_rows = 4, rank owning band 250 of 540
    p = rt.Params.make(W, H, 4, nranks=540, rank=250, band_rows=4)
    raw_gpu, st = rt.render_f32(c4_scene, p)
    assert raw_gpu.shape == raw_cpu.shape
    same = raw_equal_mask(raw_gpu, raw_cpu)
    print("c4 crop: %d of %d pixels bit-identical" % (int(same.sum()), same.size))
    assert same.mean() >= 0.999


def test_culling_is_exact_on_4k_frame(rt, gpu, c4_scene):
    """The benchmark frame itself (3840x2160, depth 5): packets, per-ray traversal, light-space binning and
    side streams against the exact-only kernels that walk every primitive in file order."""
    W, H = 3840, 2160
    ra, sa = rt.render_rgb8(c4_scene, rt.Params.make(W, H, 4))
    rb, sb = rt.render_rgb8(c4_scene, rt.Params.make(W, H, 4, flags=rt.host.F_EXACT_ONLY))
    assert (sa.rays_primary, sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_primary, sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert np.array_equal(ra, rb), "%d bytes differ" % int((ra != rb).sum())
    # some primitives both reflect and refract (fan-out 2): the queues are sized by the calibration pass, not by 2^k --
    # the whole 4K frame is ONE chunk and no ray was dropped; the exact-only context keeps the worst case and chunks
    assert sa.chunks == 1 and sa.queue_overflow == 0 and sb.queue_overflow == 0 and sb.chunks >= 1


@pytest.fixture(scope="module")
def c4_headline_scene(rt, tmp_path_factory):
    """The UNMODIFIED benchmark scene -- what bench.py renders (scenegen.write_c4 with its defaults)."""
    p = tmp_path_factory.mktemp("c4h") / "c4.txt"
    rt.scenegen.write_c4(str(p))
    return rt.load_scene(str(p))


def test_c4_headline_scene_matches_the_reference_fixture(rt, gpu, c4_headline_scene):
    """The real reference's output on the unmodified C4 scene at depth 5, 48x27 (tests/golden/c4_48x27_d4.*, made by
    make_golden.py from oracle/_ref/rt_ref): float image, PPM and ray counters."""
    import json
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    run = [r for r in json.load(open(os.path.join(golden, "index.json")))["runs"] if r["scene"] == "scenegen:c4"][0]
    W, H, D = run["width"], run["height"], run["max_depth"]
    raw_ref = np.fromfile(os.path.join(golden, run["raw"]), dtype=np.float32).reshape(H, W, 3)
    g, st = rt.render_f32(c4_headline_scene, rt.Params.make(W, H, D))
    same = raw_equal_mask(g, raw_ref)
    assert same.mean() >= 0.999, int((~same).sum())
    rgb, _ = rt.render_rgb8(c4_headline_scene, rt.Params.make(W, H, D))
    data = open(os.path.join(golden, run["ppm"]), "rb").read()
    ppm = np.frombuffer(data[data.index(b"255\n") + 4:], dtype=np.uint8).reshape(H, W, 3)
    assert_rgb8_parity(rgb, ppm, "c4 48x27")


def test_c4_headline_scene_crop_matches_oracle_at_4k(rt, oracle, gpu, c4_headline_scene):
    """The benchmark frame's own geometry (3840x2160, depth 5): rows 1200..1203 against the oracle, and the whole
    frame culled against exact-only, byte for byte."""
    W, H = 3840, 2160
    row0, row1 = 1200, 1204
    raw_cpu, cnt = oracle.render_f32(c4_headline_scene, W, H, 4, row0, row1)
    p = rt.Params.make(W, H, 4, nranks=540, rank=300, band_rows=4)   # the band that is exactly those rows
    raw_gpu, st = rt.render_f32(c4_headline_scene, p)
    assert raw_gpu.shape == raw_cpu.shape
    same = raw_equal_mask(raw_gpu, raw_cpu)
    print("c4 headline crop: %d of %d pixels bit-identical" % (int(same.sum()), same.size))
    assert same.mean() >= 0.999
    ra, sa = rt.render_rgb8(c4_headline_scene, rt.Params.make(W, H, 4))
    rb, sb = rt.render_rgb8(c4_headline_scene, rt.Params.make(W, H, 4, flags=rt.host.F_EXACT_ONLY))
    assert (sa.rays_primary, sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_primary, sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert np.array_equal(ra, rb), "%d bytes differ" % int((ra != rb).sum())


def test_culling_is_exact_across_segments(rt, gpu, tmp_path):
    """45,000 primitives = two 32,768-primitive segments of the closest-hit hierarchy; 800x450 > 2^18 rays
    so generation 0 runs the packet kernel."""
    p = tmp_path / "seg.txt"
    rt.scenegen.write_scene(str(p), 20000, 25000, rt.scenegen.lights_c4(), seed=77, frac_refractive=0.05,
                            radius=(0.2, 0.7), edge=1.2)
    sc = rt.load_scene(str(p))
    W, H = 800, 450
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, 2))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, 2, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


@pytest.mark.parametrize("size", [(960, 540), (200, 120)])
def test_culling_is_exact_with_general_affine_matrices(rt, gpu, tmp_path, size):
    """Half of the spheres affine with rotation x scale matrices, some with a translation row (never bounded:
    always pass), some singular (zero-direction children): the world-space ellipsoid bounds, the strip test
    of the shadow pass and the sticky-flag list of the per-ray traversal, against the exact-only kernels."""
    p = tmp_path / "aff.txt"
    rt.scenegen.write_scene(str(p), 3000, 3000, rt.scenegen.lights_c4(), seed=5, frac_affine=0.5,
                            frac_reflective=0.5, frac_refractive=0.1, affine="general")
    sc = rt.load_scene(str(p))
    W, H = size
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, 3))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, 3, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


def test_culling_is_exact_beyond_the_affine_table(rt, gpu, tmp_path):
    """Shadow passes without the (light, direction cell) table of affine spheres: 5,000 affine spheres are more
    than the table holds (4,096), so every wave strip-tests all of them; and a light INSIDE a scene whose hit
    points lie further from it than the far corner of the box of the bounded primitives cannot happen by
    construction, so the table's range check is exercised the other way: a second, tiny scene whose only
    bounded primitives are small and near the light while the rays hit a huge polygon (polygons have no bounding
    sphere and are not part of the box) far outside it."""
    p = tmp_path / "many_affine.txt"
    rt.scenegen.write_scene(str(p), 5000, 200, rt.scenegen.lights_c4(), seed=11, frac_affine=1.0, frac_reflective=0.3)
    sc = rt.load_scene(str(p))
    a, sa = rt.render_f32(sc, rt.Params.make(240, 135, 2))
    b, sb = rt.render_f32(sc, rt.Params.make(240, 135, 2, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    q = tmp_path / "far_hits.txt"
    q.write_text("6\ncamera 0 0 -1\nvrp 0 0 0\nvpn 0 0 -1\nup 0 1 0\nambient 0.05 0.05 0.05\n"
                 "light 0 30 -20 0.6 0.6 0.6\n"
                 "sphere 0 0 -20 1  1 0 0  1 0 1  .4 .4 .4  .5 0 1 0 3\n"
                 "sphere 3 1 -22 1.5  0 1 0  .2 .9 .2  .5 .5 .5  0 0 1 1 12  2 0 0  0 1 0  0 0 .5  0 0 0\n"
                 "polygon 4 -900 -12 100  900 -12 100  900 -12 -3000  -900 -12 -3000   0.8 0.8 0.9  .5 .5 .55  .3 .3 .3  0.4 0 1 12\n"
                 "triangle -2 4 -18  2 4 -18  0 4 -24  .8 .6 .7  .5 .4 .5  .1 .1 .1  0 0 1 4\n")
    sc2 = rt.load_scene(str(q))
    a, sa = rt.render_f32(sc2, rt.Params.make(320, 180, 3))
    b, sb = rt.render_f32(sc2, rt.Params.make(320, 180, 3, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_culled_equals_exact(rt, gpu, tmp_path, seed):
    """Random scenes (counts, lights -- also inside the scene --, affine share and kind, radii, edge lengths, depth
    drawn per seed): every culling stage of the default kernels against the exact-only kernels, bit for bit."""
    rng = np.random.default_rng(seed)
    ns, nt, nl = int(rng.integers(50, 4000)), int(rng.integers(0, 4000)), int(rng.integers(1, 6))
    lights = [(float(rng.uniform(-45, 45)), float(rng.uniform(-30, 45)), float(rng.uniform(-130, 20)), 0.3, 0.3, 0.3) for _ in range(nl)]
    p = tmp_path / ("fuzz_%d.txt" % seed)
    rt.scenegen.write_scene(str(p), ns, nt, lights, seed=1000 + seed, frac_affine=float(rng.uniform(0, 0.6)),
                            frac_reflective=float(rng.uniform(0.1, 0.8)), frac_refractive=float(rng.uniform(0, 0.4)),
                            affine="general" if seed % 2 else "diag", radius=(0.2, float(rng.uniform(0.5, 5.0))),
                            edge=float(rng.uniform(0.5, 10.0)))
    sc = rt.load_scene(str(p))
    md = int(rng.integers(1, 6))
    a, sa = rt.render_f32(sc, rt.Params.make(480, 270, md))
    b, sb = rt.render_f32(sc, rt.Params.make(480, 270, md, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


def test_general_affine_scene_matches_oracle(rt, oracle, gpu, tmp_path):
    """A generated scene the reference's authors never saw: 300 spheres (half affine, general matrices,
    translation rows, singular ones) + 300 triangles, reflective and refractive, against the CPU oracle."""
    p = tmp_path / "affo.txt"
    rt.scenegen.write_scene(str(p), 300, 300, rt.scenegen.lights_c4(), seed=9, frac_affine=0.5, frac_reflective=0.5,
                            frac_refractive=0.2, affine="general", radius=(1.0, 4.0), edge=8.0)
    sc = rt.load_scene(str(p))
    W, H = 160, 120
    raw_gpu, st = rt.render_f32(sc, rt.Params.make(W, H, 4))
    raw_cpu, cnt = oracle.render_f32(sc, W, H, 4)
    assert st.rays_primary + st.rays_secondary == cnt.trace_calls
    assert st.rays_shadow == cnt.shadow_rays and st.hits == cnt.hits
    same = raw_equal_mask(raw_gpu, raw_cpu)
    assert same.mean() >= 0.999, int((~same).sum())
    rgb_gpu, _ = rt.render_rgb8(sc, rt.Params.make(W, H, 4))
    rgb_cpu, _ = oracle.render_rgb8(sc, W, H, 4)
    assert_rgb8_parity(rgb_gpu, rgb_cpu, "general affine")


@pytest.mark.parametrize("name", ["poly.txt", "quirks.txt", "affine.txt"])
def test_packet_kernel_on_small_scenes(rt, gpu, name):
    """800x450 > 2^18 rays sends generation 0 through the packet kernel (bitmap in file order) also for the
    hand-written scenes: polygons (records that always pass), the sticky-flag and matrix quirks."""
    sc = rt.load_scene(os.path.join(SCENES, name))
    W, H = 800, 450
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, 4))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, 4, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


HEADER = "camera 0 0 -1\nvrp 0 0 0\nvpn 0 0 -1\nup 0 1 0\nambient 0.1 0.2 0.3\n"
EDGE_SCENES = {
    "empty": "0\n" + HEADER,
    "lights_only": "2\n" + HEADER + "light 1 2 3 1 1 1\nlight -1 2 3 1 1 1\n",
    "no_lights": "2\n" + HEADER + "sphere 0 0 -10 3  1 1 1  1 1 1  .5 .5 .5  0.5 0 1 0 8\n"
                 "triangle -5 -4 -8  5 -4 -8  0 -4 -30  1 1 1  1 1 1  .5 .5 .5  0.5 0 1 8\n",
    "polygon_only": "2\n" + HEADER + "light 3 8 -2 1 1 1\n"
                    "polygon 4 -6 -3 -6  6 -3 -6  6 -3 -40  -6 -3 -40  1 1 1  .7 .7 .7  .3 .3 .3  0.5 0 1 8\n",
    "one_affine_sphere": "2\n" + HEADER + "light 3 8 -2 1 1 1\n"
                         "sphere 0 0 -10 2  1 1 1  1 1 1  .5 .5 .5  0.5 0.5 1.3 1 8  2 0 0  0 1 0  0 0 0.5  0 0 0\n",
}


@pytest.mark.parametrize("name", sorted(EDGE_SCENES))
@pytest.mark.parametrize("size", [(1, 1), (67, 33), (800, 450)])
def test_edge_scenes(rt, oracle, gpu, tmp_path, name, size):
    """Empty and ragged inputs: no primitives, no lights, a single record of one kind; a 1 x 1 image, an odd
    size (partial waves, thin kernels) and one above 2^18 rays (packet kernel)."""
    p = tmp_path / (name + ".txt")
    p.write_text(EDGE_SCENES[name])
    sc = rt.load_scene(str(p))
    W, H = size
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, 4))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, 4, flags=rt.host.F_EXACT_ONLY))
    assert raw_equal_mask(a, b).all()
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    if W * H <= 67 * 33:
        raw_cpu, cnt = oracle.render_f32(sc, W, H, 4)
        assert sa.rays_primary + sa.rays_secondary == cnt.trace_calls and sa.rays_shadow == cnt.shadow_rays
        assert raw_equal_mask(a, raw_cpu).mean() >= 0.999
        rgb_gpu, _ = rt.render_rgb8(sc, rt.Params.make(W, H, 4))
        rgb_cpu, _ = oracle.render_rgb8(sc, W, H, 4)
        assert_rgb8_parity(rgb_gpu, rgb_cpu, name)


def _camera_case(kind):
    """Scene text for test_camera_table_cases (HEAD grammar: first line = number of records)."""
    rng = np.random.default_rng(77)
    cam = "0 0 -0.1" if kind == "wide" else "0 0 -1"
    head = "camera %s\nvrp 0 0 0\nvpn 0 0 -1\nup 0 1 0\nambient 0.1 0.2 0.3\n" % cam
    rec = ["light 3 8 -2 1 1 1", "light -20 5 30 0.5 0.5 0.5"]
    mat = "1 1 1  .8 .8 .8  .4 .4 .4"
    if kind in ("inside", "wide"):
        # primitives on every side of the camera, also behind it and around it; a few affine spheres, reflective ones
        for i in range(400):
            c = rng.uniform(-50, 50, 3)
            r = float(rng.uniform(0.5, 4.0))
            xf = i % 7 == 0
            s = "sphere %.4f %.4f %.4f %.4f  %s  %s 1.3 %d 8" % (c[0], c[1], c[2], r, mat, "0.5 0" if i % 3 == 0 else "0 0", 1 if xf else 0)
            if xf:
                s += "  %.3f 0 0  0 %.3f 0  0 0 %.3f  0 0 0" % tuple(rng.uniform(0.6, 1.8, 3))
            rec.append(s)
        for i in range(300):
            v = rng.uniform(-50, 50, 3)
            e = rng.uniform(-6, 6, (2, 3))
            rec.append("triangle %.4f %.4f %.4f  %.4f %.4f %.4f  %.4f %.4f %.4f  %s  %s 1 8" %
                       (v[0], v[1], v[2], *(v + e[0]), *(v + e[1]), mat, "0.5 0" if i % 4 == 0 else "0 0"))
        rec.append("sphere 0 0 0 30  %s  0 0 1 0 8" % mat)   # the camera sits inside this one
        rec.append("polygon 4 -6 -3 -6  6 -3 -6  6 -3 -40  -6 -3 -40  %s  0.5 0 1 8" % mat)
    elif kind == "long_list":
        # 2,500 small spheres behind one another: the middle cells' lists exceed what a build wave sorts (their rays take the scan);
        # a third light on their line: ITS cells along the line overflow too (a light's list is then stored unsorted, dneed = -1)
        rec.append("light 0.01 0.02 8 0.6 0.6 0.6")
        for i in range(2500):
            rec.append("sphere %.4f %.4f %.4f 0.3  %s  %s 1 0 8" % (rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), -5.0 - 0.1 * i, mat, "0.5 0" if i % 50 == 0 else "0 0"))
    elif kind == "many_polygons":
        # 300 records no filter bounds: the camera table is not built, generation 0 keeps the packet kernel
        for i in range(300):
            x, y, z = rng.uniform(-30, 30), rng.uniform(-20, 20), rng.uniform(-90, -20)
            rec.append("polygon 4 %.3f %.3f %.3f  %.3f %.3f %.3f  %.3f %.3f %.3f  %.3f %.3f %.3f  %s  %s 1 8" %
                       (x, y, z, x + 2, y, z, x + 2, y + 2, z, x, y + 2, z, mat, "0.5 0" if i % 5 == 0 else "0 0"))
        for i in range(50):
            c = rng.uniform(-30, 30, 3)
            rec.append("sphere %.4f %.4f %.4f 2  %s  0.5 0 1 0 8" % (c[0], c[1], c[2] - 50, mat))
    return "%d\n" % len(rec) + head + "\n".join(rec) + "\n"


@pytest.mark.parametrize("kind", ["inside", "wide", "long_list", "many_polygons"])
def test_camera_table_cases(rt, gpu, tmp_path, kind):
    """The primary rays' direction-cell table (k_primary_cells) where its construction has a case of its own:
    the camera inside the scene and inside a sphere (primitives behind the apex, the front cull), a view cone of more than
    150 degrees, a cell whose list is too long to sort (its rays take the scan over every primitive), a scene with hundreds of
    always-pass records (no table: the packet kernel).  Default kernels against the exact-only kernels, bit for bit, whole
    frames and a rank's bands."""
    p = tmp_path / ("cam_%s.txt" % kind)
    p.write_text(_camera_case(kind))
    sc = rt.load_scene(str(p))
    W, H = (96, 54) if kind == "long_list" else (480, 270)
    for extra in ({}, {"nranks": 3, "rank": 1, "band_rows": 5}):
        a, sa = rt.render_f32(sc, rt.Params.make(W, H, 3, **extra))
        b, sb = rt.render_f32(sc, rt.Params.make(W, H, 3, flags=rt.host.F_EXACT_ONLY, **extra))
        same = raw_equal_mask(a, b)
        assert same.all(), "%s %s: %d pixels differ" % (kind, extra, int((~same).sum()))
        assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
        assert sa.hits > 0 or extra


def test_many_lights_take_a_coarser_map(rt, gpu, tmp_path):
    """200 lights: one list per (light, cell) would pass what one build launch can hold at 512 cells a side, so the tables take
    a coarser map (rtx_device.hip: kMaxLists).  Default kernels against the exact-only kernels, bit for bit."""
    rng = np.random.default_rng(5)
    rec = ["light %.3f %.3f %.3f 0.02 0.02 0.02" % (rng.uniform(-40, 40), rng.uniform(-10, 40), rng.uniform(-100, 10)) for _ in range(200)]
    mat = "1 1 1  .8 .8 .8  .4 .4 .4"
    for i in range(60):
        c = rng.uniform(-20, 20, 3)
        xf = i % 6 == 0
        s = "sphere %.4f %.4f %.4f %.4f  %s  %s 1.3 %d 8" % (c[0], c[1], c[2] - 50, rng.uniform(1, 4), mat, "0.5 0" if i % 3 == 0 else "0 0", 1 if xf else 0)
        if xf:
            s += "  1.5 0 0  0 0.8 0  0 0 1.2  0 0 0"
        rec.append(s)
    for i in range(40):
        v = rng.uniform(-20, 20, 3) + np.array([0, 0, -50])
        e = rng.uniform(-8, 8, (2, 3))
        rec.append("triangle %.4f %.4f %.4f  %.4f %.4f %.4f  %.4f %.4f %.4f  %s  0 0 1 8" % (v[0], v[1], v[2], *(v + e[0]), *(v + e[1]), mat))
    p = tmp_path / "lights200.txt"
    p.write_text("%d\n" % len(rec) + HEADER + "\n".join(rec) + "\n")
    sc = rt.load_scene(str(p))
    a, sa = rt.render_f32(sc, rt.Params.make(96, 54, 2))
    b, sb = rt.render_f32(sc, rt.Params.make(96, 54, 2, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert sa.shadow_form == 1 and sa.rays_shadow == 200 * sa.hits


@pytest.mark.parametrize("seed", range(6))
def test_random_cameras_culled_equals_exact(rt, gpu, tmp_path, seed):
    """Random scenes under random cameras (inside or outside the cloud, any view direction, an up vector not orthogonal to it,
    fields of view from a few degrees to more than 150 -- tools/dbg/fuzz_camera.py runs more seeds): the primary rays' table
    and everything behind it against the exact-only kernels, bit for bit."""
    rng = np.random.default_rng(5000 + seed)
    ns, nt, nl = int(rng.integers(50, 2500)), int(rng.integers(0, 2500)), int(rng.integers(1, 4))
    lights = [(float(rng.uniform(-45, 45)), float(rng.uniform(-30, 45)), float(rng.uniform(-130, 20)), 0.3, 0.3, 0.3) for _ in range(nl)]
    p = tmp_path / ("fuzzcam_%d.txt" % seed)
    rt.scenegen.write_scene(str(p), ns, nt, lights, seed=3000 + seed, frac_affine=float(rng.uniform(0, 0.5)),
                            frac_reflective=float(rng.uniform(0.1, 0.8)), frac_refractive=float(rng.uniform(0, 0.3)),
                            affine="general" if seed % 2 else "diag", radius=(0.2, float(rng.uniform(0.5, 5.0))),
                            edge=float(rng.uniform(0.5, 10.0)))
    cam = rng.uniform(-1, 1, 3) * float(rng.choice([0.1, 0.5, 1.0, 4.0]))
    if abs(cam[2]) < 0.05:
        cam[2] = 0.3
    vrp = np.array([rng.uniform(-45, 45), rng.uniform(-30, 30), rng.uniform(-130, 10)]) if seed % 3 else np.zeros(3)
    vpn = rng.normal(size=3); vpn /= np.linalg.norm(vpn)
    up = rng.normal(size=3); up /= np.linalg.norm(up)
    repl = {"camera": "camera %.5f %.5f %.5f" % tuple(cam), "vrp": "vrp %.4f %.4f %.4f" % tuple(vrp),
            "vpn": "vpn %.6f %.6f %.6f" % tuple(vpn), "up": "up %.6f %.6f %.6f" % tuple(up)}
    lines = [repl.get(l.split()[0], l) if l.split() else l for l in p.read_text().splitlines()]
    p.write_text("\n".join(lines) + "\n")
    sc = rt.load_scene(str(p))
    md = int(rng.integers(1, 5))
    a, sa = rt.render_f32(sc, rt.Params.make(400, 225, md))
    b, sb = rt.render_f32(sc, rt.Params.make(400, 225, md, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)


@pytest.mark.parametrize("seed", range(4))
def test_random_lights_shadow_stage_culled_equals_exact(rt, gpu, tmp_path, seed):
    """Random scenes in front of two wall triangles (every primary ray hits something), 24 random lights each -- inside the cloud,
    far outside, behind the walls, at the camera -- depth 1 (tools/dbg/fuzz_shadow.py, which runs more seeds): the (light, cell)
    tables and their walk against the exact-only kernels, bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_shadow", os.path.join(ROOT, "tools", "dbg", "fuzz_shadow.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    p = tmp_path / ("fuzzsh_%d.txt" % seed)
    fz.make_scene(str(p), seed)
    sc = rt.load_scene(str(p))
    a, sa = rt.render_f32(sc, rt.Params.make(480, 270, 1))
    b, sb = rt.render_f32(sc, rt.Params.make(480, 270, 1, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert sa.shadow_form == 1 and sa.hits > 100000


@pytest.mark.parametrize("affine", ["rotated", "diag"])
def test_ordered_scan_of_flagged_rays(rt, gpu, tmp_path, affine):
    """Affine spheres overlapping many deep (3,000 spheres, four in five of them affine, in a box of 20 x 12 x 30): rays whose list
    of affine-sphere hits overflows (LexAcc, four entries) take the ordered scan over every primitive -- one ray at a time with the
    wave's lanes over the primitives when few lanes of a wave are flagged (ordered_scan, rtx_closest.hip.h), each lane for itself
    when many are.  Culled == exact-only, bit for bit, and the flagged rays exist (rtx_stats.packet_stats[1], low half: 20 secondary
    rays with rotated ellipsoids, 1,386 with axis-aligned ones; tools/dbg/flagged_probe.py)."""
    p = tmp_path / "dense_affine.txt"
    rt.scenegen.write_scene(str(p), 3000, 500, rt.scenegen.lights_c4(), seed=77, box=(10.0, 6.0), zfar=-60.0, znear=-30.0,
                            frac_affine=0.8, frac_reflective=0.6, radius=(0.5, 3.0), affine=affine)
    sc = rt.load_scene(str(p))
    a, sa = rt.render_f32(sc, rt.Params.make(640, 360, 3, flags=rt.host.F_PACKET_STATS))
    b, sb = rt.render_f32(sc, rt.Params.make(640, 360, 3, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    flagged = int(sa.packet_stats[1]) & 0xffffffff
    assert flagged >= 10 and sa.rays_secondary > 10000, (flagged, sa.rays_secondary)
    # ... and primary rays whose list of affine hits overflowed: they take the ordered scan over their camera cell's own list,
    # merged by primitive index from its two parts (k_primary_cells)
    assert (int(sa.packet_stats[2]) & 0xffffffff) >= 10, int(sa.packet_stats[2])


@pytest.mark.parametrize("seed", [1, 7])
def test_dense_scenes_culled_equals_exact(rt, gpu, tmp_path, seed):
    """Two seeds of tools/dbg/fuzz_dense.py (thousands of mostly affine spheres overlapping many deep, reflective and refractive,
    depth 3 - 4): hundreds to thousands of secondary AND primary rays take the ordered scans; culled == exact-only, bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_dense", os.path.join(ROOT, "tools", "dbg", "fuzz_dense.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    p = tmp_path / ("fuzzde_%d.txt" % seed)
    _, _, md = fz.make_scene(str(p), seed)
    sc = rt.load_scene(str(p))
    a, sa = rt.render_f32(sc, rt.Params.make(480, 270, md, flags=rt.host.F_PACKET_STATS))
    b, sb = rt.render_f32(sc, rt.Params.make(480, 270, md, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert (int(sa.packet_stats[1]) & 0xffffffff) >= 100 and (int(sa.packet_stats[2]) & 0xffffffff) >= 1000


@pytest.mark.parametrize("seed", [0, 4, 13])
def test_scaled_scenes_culled_equals_exact(rt, gpu, tmp_path, seed):
    """Random scenes with every length multiplied by 1e-2, 1e3 and 1e2 (tools/dbg/fuzz_scale.py, which runs more seeds and scales):
    the library's absolute constants -- the reference's own 1e-4 origin bias, the tables' delta, the guard box -- on scenes far
    smaller and far larger than the benchmark's, against the exact-only kernels, bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_scale", os.path.join(ROOT, "tools", "dbg", "fuzz_scale.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    p = tmp_path / ("fuzzsc_%d.txt" % seed)
    _, _, _, md = fz.make_scene(str(p), seed)
    sc = rt.load_scene(str(p))
    a, sa = rt.render_f32(sc, rt.Params.make(400, 225, md))
    b, sb = rt.render_f32(sc, rt.Params.make(400, 225, md, flags=rt.host.F_EXACT_ONLY))
    same = raw_equal_mask(a, b)
    assert same.all(), "%d pixels differ" % int((~same).sum())
    assert (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    assert sa.hits > 10000
