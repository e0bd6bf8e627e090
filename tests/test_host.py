"""Host side kept from the reference: the scene reader's stream semantics (raytracing.cpp:851-979,
SURVEY.md section 4 table), the PPM writer (:786-797) and the band arithmetic."""
import os

import numpy as np

from conftest import GOLDEN, ROOT, SCENES


def test_bundled_scenes_parse_as_the_head_grammar_does(rt):
    b = os.path.join(GOLDEN, "bundled")
    # inp-refl / inp1: no header -> the stream dies before the object loop: nothing survives
    for name in ("inp-refl.txt", "inp1.txt"):
        sc = rt.load_scene(os.path.join(b, name)); v = sc.view
        assert (v.nspheres, v.ntriangles, v.nlights, v.nplanes) == (0, 0, 0, 0)
    # inpaffine: the `ambient` line eats the light; one sphere survives, untransformed (int read of "0.0")
    sc = rt.load_scene(os.path.join(b, "inpaffine.txt")); v = sc.view
    assert (v.nspheres, v.ntriangles, v.nlights) == (1, 0, 0)
    s = v.spheres[0]
    assert (v.ambient.x, v.ambient.y, v.ambient.z) == (0.0, 10.0, -30.0)
    assert (s.center.x, s.center.y, s.center.z, s.radius) == (2.0, -2.0, -30.0, 4.0)
    assert (s.color.x, s.color.y, s.color.z) == (1.0, 0.0, 0.0)
    assert (s.diffcolor.x, s.diffcolor.y, s.diffcolor.z) == (1.0, 0.0, 1.0)
    assert (s.speccoeff.x, s.speccoeff.y, s.speccoeff.z) == (1.0, 3.0, 0.0)
    assert s.istransformed == 0 and s.specexpo == 0.0 and s.refractive_index == 1.0
    # inp.txt: ambient <- the light line; slot 4 is a sphere whose record fails at diffcolor.x and stays zero-filled
    sc = rt.load_scene(os.path.join(b, "inp.txt")); v = sc.view
    assert (v.nspheres, v.ntriangles, v.nlights) == (1, 0, 0)
    assert (v.ambient.x, v.ambient.y, v.ambient.z) == (10.0, 10.0, -5.0)
    s = v.spheres[0]
    assert (s.center.y, s.center.z, s.radius, s.color.x) == (-7.0, -15.0, 4.0, 1.0)
    assert s.diffcolor.x == 0.0 and s.reflectivity == 0.0 and s.istransformed == 0
    sc = rt.load_scene(os.path.join(b, "inp01.txt")); v = sc.view
    assert (v.nspheres, v.ntriangles, v.nlights) == (1, 0, 0)
    assert (v.ambient.x, v.ambient.y, v.ambient.z) == (-20.0, 40.0, 8.0)


def test_head_grammar_scene(rt):
    sc = rt.load_scene(os.path.join(SCENES, "quirks.txt")); v = sc.view
    # 3 lights + 6 spheres + 4 triangles + one unknown keyword that burns the 14th slot
    assert (v.nspheres, v.ntriangles, v.nlights) == (6, 4, 3)
    assert v.spheres[0].istransformed == 1 and abs(v.spheres[0].mat.a.x - 1.2) < 1e-6 and v.spheres[0].mat.d.z == -1.0
    assert v.spheres[1].istransformed == 0 and v.spheres[1].mat.a.x == 0.0   # zero-filled, never parsed
    assert abs(v.triangles[0].specexpo - 3.5) < 1e-6
    sc = rt.load_scene(os.path.join(SCENES, "room.txt")); v = sc.view
    assert (v.nspheres, v.ntriangles, v.nlights, v.nplanes) == (2, 11, 2, 1)


def test_polygon_records(rt, tmp_path):
    """parseinput's polygon rules (rt.cpp:910-967): normal from the first three vertices, centroid, exact-equality
    coplanarity check that DROPS the record, and the colour / diffuse / specular order of that record type."""
    p = tmp_path / "poly.txt"
    p.write_text("5\ncamera 0 0 -1\nvrp 0 0 0\nvpn 0 0 -1\nup 0 1 0\nambient 0 0 0\n"
                 "polygon 4 0 0 -5 1 0 -5 1 1 -5 0 1 -5  1 1 1 .5 .5 .5 .1 .1 .1 0.25 0 1 5\n"
                 "polygon 4 0 0 -20  1 0 -20  1 1 -21  0 1 -20   1 0 0  1 0 0  1 1 1  0 0 1 5\n"
                 "polygon 3 2 0 -7 2 2 -7 2 0 -9  0 1 0 0 0 1 1 0 0 0 0.5 1.5 9\n"
                 "sphere 0 0 -10 1  1 1 1  1 1 1  0 0 0  0 0 1 0 1\nlight 1 2 3 1 1 1\n")
    sc = rt.load_scene(str(p)); v = sc.view
    assert (v.nspheres, v.nlights, v.npolygons, v.npolygons_dropped, v.npolygon_vertices) == (1, 1, 2, 1, 7)
    assert v.spheres[0].center.z == -10.0 and v.lights[0].point.y == 2.0
    a, b = v.polygons[0], v.polygons[1]
    assert (a.numvertices, a.first_vertex, b.numvertices, b.first_vertex) == (4, 0, 3, 4)   # the dropped one left no vertices
    assert (a.normal.x, a.normal.y, abs(a.normal.z)) == (0.0, 0.0, 1.0)
    assert (a.centerPoint.x, a.centerPoint.y, a.centerPoint.z) == (0.5, 0.5, -5.0)
    assert (a.diffcolor.x, a.speccoeff.x, a.reflectivity, a.specexpo) == (0.5, np.float32(0.1), 0.25, 5.0)
    assert (abs(b.normal.x), b.normal.y, b.normal.z) == (1.0, 0.0, 0.0)
    assert (b.color.y, b.diffcolor.z, b.speccoeff.x, b.transparency, b.refractive_index) == (1.0, 1.0, 1.0, 0.5, 1.5)
    assert v.polygon_vertices[6].z == -9.0
    assert sc.nprims == 3


def test_ppm_writer(rt, oracle, tmp_path):
    img = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    p = tmp_path / "o.ppm"
    rt.write_ppm(str(p), img)
    data = p.read_bytes()
    assert data.startswith(b"P6\n7 5 255\n") and len(data) == len(b"P6\n7 5 255\n") + 105
    assert np.array_equal(oracle.read_ppm(str(p)), img)


def test_band_rows_partition(rt):
    for (h, n, b) in [(2160, 8, 30), (2160, 8, 0), (480, 3, 16), (7, 4, 2), (5, 8, 1), (1, 1, 0), (4320, 8, 30)]:
        seen = np.zeros(h, dtype=int)
        for r in range(n):
            rows = rt.band_rows(h, n, r, b)
            assert np.all(np.diff(rows) > 0)
            seen[rows] += 1
            bw = b if b > 0 else -(-h // n)
            assert all((y // bw) % n == r for y in rows)
        assert np.all(seen == 1)
    assert len(rt.band_rows(10, 1, 0, 0)) == 10


def test_bench_launcher_honours_gpus():
    """bench.py --gpus N (the driver's command): N > 1 without WORLD_SIZE spawns N ranks under
    torch.distributed.run on 127.0.0.1 with the same arguments; a WORLD_SIZE that disagrees is an error;
    the ranks themselves (WORLD_SIZE == N) and N == 1 just run."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launch_plan(1, {}, []) == ("run", None)
    assert bench.launch_plan(4, {"WORLD_SIZE": "4"}, []) == ("run", None)
    what, cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--steps", "3"])
    assert what == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert bench.launch_plan(8, {"WORLD_SIZE": "2"}, [])[0] == "error"
    assert bench.launch_plan(1, {"WORLD_SIZE": "2"}, [])[0] == "error"
    assert bench.launch_plan(0, {}, [])[0] == "error"
    a = bench.parse_args(["--gpus", "2", "--workload", "c5"])
    assert a.gpus == 2 and a.workload == "c5" and a.steps == 5 and not a.in_process
    # --in-process: ONE process times the C-ABI's own multi-GPU call; nothing is spawned (main() branches before launch_plan)
    a = bench.parse_args(["--in-process", "--gpus", "8"])
    assert a.in_process and a.gpus == 8


def test_legacy_grammar_loader_reads_the_bundled_scenes_as_written(rt):
    """rtx_scene_load_legacy (opt-in): the older line-per-record grammars documented by the bundled
    files' own comment lines.  Record counts and a few fields of every bundled scene; the default loader is unchanged
    (test_bundled_scenes_parse_as_the_head_grammar_does above)."""
    b = os.path.join(GOLDEN, "bundled")
    want = {  # spheres, triangles, planes, lights, polygons
        "inp.txt": (5, 2, 0, 1, 0), "inp-refl.txt": (1, 3, 0, 1, 0), "inpaffine.txt": (5, 5, 0, 1, 0),
        "inp1.txt": (9, 14 + 12, 6, 1, 0), "inp01.txt": (7, 12, 0, 1, 1)}
    for name, counts in want.items():
        sc = rt.load_scene(os.path.join(b, name), legacy=True); v = sc.view   # (the view lives as long as sc does)
        assert (v.nspheres, v.ntriangles, v.nplanes, v.nlights, v.npolygons) == counts, name
        assert (v.camera.x, v.camera.y, v.camera.z) == (0.0, 0.0, -1.0) and (v.up.x, v.up.y, v.up.z) == (0.0, 1.0, 0.0)
    sc = rt.load_scene(os.path.join(b, "inpaffine.txt"), legacy=True); v = sc.view
    s = v.spheres[0]   # sphere 2 -2 -30 4  1 0 0  1.0 0.0 1.0  1  3 0 0 0 1 0 0 0 1 0 0 0
    assert (s.center.x, s.center.y, s.center.z, s.radius) == (2.0, -2.0, -30.0, 4.0)
    assert (s.reflectivity, s.transparency, s.refractive_index, s.istransformed) == (1.0, 0.0, 1.0, 1)
    assert (s.mat.a.x, s.mat.b.y, s.mat.c.z, s.mat.d.x) == (3.0, 1.0, 1.0, 0.0)
    assert (s.diffcolor.x, s.diffcolor.y, s.diffcolor.z) == (s.color.x, s.color.y, s.color.z) == (1.0, 0.0, 0.0)
    assert v.spheres[1].istransformed == 0 and v.spheres[1].reflectivity == 0.5
    t = v.triangles[0]  # ... 0.80 0.60 0.70  1.5 0.0 0.0
    assert (t.color.x, t.reflectivity, t.transparency, t.refractive_index) == (np.float32(0.8), 1.5, 0.0, 0.0)
    l = v.lights[0]
    assert (l.point.x, l.point.y, l.point.z, l.color.x) == (0.0, 10.0, -30.0, np.float32(0.33))
    # inp.txt (the oldest grammar): colour only, everything else defaulted
    sc = rt.load_scene(os.path.join(b, "inp.txt"), legacy=True); v = sc.view
    s = v.spheres[0]
    assert (s.reflectivity, s.transparency, s.refractive_index, s.istransformed, s.specexpo) == (0.0, 0.0, 1.0, 0, 20.0)
    # inp1.txt: plane 2 0 0 200  1 1 0  0.4 0 1.5 (normal, d, colour, material), the cuboid as 12 triangles
    sc = rt.load_scene(os.path.join(b, "inp1.txt"), legacy=True); v = sc.view
    p = v.planes[0]
    assert (p.normal.x, p.d, p.color.z, p.reflectivity, p.refractive_index) == (2.0, 200.0, 0.0, np.float32(0.4), 1.5)
    xs = sorted({(tr.u.x, tr.u.y, tr.u.z) for tr in v.triangles[12:24]} | {(tr.w.x, tr.w.y, tr.w.z) for tr in v.triangles[12:24]}
                | {(tr.v.x, tr.v.y, tr.v.z) for tr in v.triangles[12:24]})
    assert len(xs) == 8 and xs[0] == (0.0, -10.0, -10.0) and xs[-1] == (10.0, 0.0, 0.0)
    # inp01.txt: polygon 4 ... with parseinput's normal / centroid rules
    sc = rt.load_scene(os.path.join(b, "inp01.txt"), legacy=True); v = sc.view
    g = v.polygons[0]
    assert g.numvertices == 4 and (g.centerPoint.x, g.centerPoint.y, g.centerPoint.z) == (0.0, 25.0, -40.0) and abs(g.normal.z) == 1.0
