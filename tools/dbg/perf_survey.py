"""Performance survey over scenes the thresholds were NOT tuned on ("nothing guards against a cliff on a third
scene"): 4K, depth 5, one context each; frame time, rays, Mrays/s, context creation, the form chosen -- and at 480x270 the culled
kernels against the exact-only kernels, bit for bit.  usage: perf_survey.py [name ...]"""
import sys, os, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt

G = rt.scenegen
L4 = G.lights_c4()
MAT = "1 1 1  .8 .8 .8  .4 .4 .4  %s 0 1 8"


def append(path, records):
    lines = open(path).read().splitlines()
    lines[0] = str(int(lines[0]) + len(records))
    open(path, "w").write("\n".join(lines + records) + "\n")


def floor(path):  # C4 over a mirror floor and in front of a wall: every primary ray hits, the floor's reflections look up into the cloud
    G.write_c4(path)
    append(path, ["triangle -400 -26 20  400 -26 20  400 -26 -400  " + MAT % ".5", "triangle -400 -26 20  400 -26 -400  -400 -26 -400  " + MAT % ".5",
                  "triangle -400 -300 -130  400 -300 -130  400 300 -130  " + MAT % "0", "triangle -400 -300 -130  400 300 -130  -400 300 -130  " + MAT % "0"])


def room(path):  # C4 inside a closed box of twelve wall triangles, the camera in it, two of the lights inside: no ray misses
    G.write_scene(path, 5000, 5000, [(-30.0, 20.0, -20.0, .25, .25, .25), (30.0, 20.0, -100.0, .25, .25, .25), (0.0, 24.0, -60.0, .25, .25, .25), (0.0, -20.0, -40.0, .25, .25, .25)])
    x0, x1, y0, y1, z0, z1 = -45.0, 45.0, -28.0, 28.0, -125.0, 5.0
    c = [(x, y, z) for x in (x0, x1) for y in (y0, y1) for z in (z0, z1)]   # corner index = 4 ix + 2 iy + iz
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    recs = []
    for qi, (a, b, cc, d) in enumerate(quads):
        refl = ".3" if qi == 2 else "0"   # the floor reflects
        for tri in ((a, b, cc), (a, cc, d)):
            recs.append("triangle " + "  ".join("%g %g %g" % c[v] for v in tri) + "  " + MAT % refl)
    append(path, recs)


SCENES = {
    "c4": G.write_c4,
    "floor": floor,
    "room": room,
    "dense": lambda p: G.write_scene(p, 5000, 5000, L4, box=(10.0, 6.0), zfar=-60.0, znear=-30.0),                 # everything overlaps
    "bigspheres": lambda p: G.write_scene(p, 2000, 0, L4, radius=(2.0, 8.0)),
    "longtri": lambda p: G.write_scene(p, 0, 5000, L4, edge=20.0),
    "affine": lambda p: G.write_scene(p, 3000, 0, L4, frac_affine=1.0, affine="rotated"),                            # every sphere an ellipsoid
    "affine_odd": lambda p: G.write_scene(p, 3000, 0, L4, frac_affine=1.0, affine="general"),  # + 1 in 16 with a translation row, 1 in 32 singular: ~280 records no bound holds for
    "glass": lambda p: G.write_scene(p, 5000, 5000, L4, frac_reflective=0.5, frac_refractive=0.5),                   # binary ray trees
    "lights_in": lambda p: G.write_scene(p, 5000, 5000, [(x, y, z, .1, .1, .1) for x in (-20.0, 20.0) for y in (-10.0, 10.0) for z in (-40.0, -90.0)]),
    "few": lambda p: G.write_scene(p, 6, 6, L4, radius=(3.0, 8.0), edge=20.0, box=(20.0, 12.0), zfar=-60.0),
    "s50k": lambda p: G.write_scene(p, 50000, 0, L4, radius=(0.1, 0.5)),
    "m1": lambda p: G.write_scene(p, 500000, 500000, L4, radius=(0.05, 0.2), edge=0.4),                              # a million primitives (not in the default list)
    "mixed_sizes": lambda p: (G.write_scene(p, 5000, 5000, L4), append(p, ["sphere 0 0 -70 18  1 1 1  .8 .8 .8  .4 .4 .4  .5 0 1.2 0 8",
                                                                           "sphere 25 5 -50 9  1 1 1  .8 .8 .8  .4 .4 .4  .5 .5 1.2 0 8"])),
}

if __name__ == "__main__":
    names = sys.argv[1:] or [n for n in SCENES if n != "m1"]
    bad = 0
    for name in names:
        p = os.path.join(tempfile.gettempdir(), "survey_%s.txt" % name)
        SCENES[name](p)
        sc = rt.load_scene(p)
        a, sa = rt.render_f32(sc, rt.Params.make(480, 270, 4))
        b, sb = rt.render_f32(sc, rt.Params.make(480, 270, 4, flags=rt.host.F_EXACT_ONLY))
        same = bool((a.view(np.uint32) == b.view(np.uint32)).all()) and (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
        bad += 0 if same else 1
        t0 = time.perf_counter()
        c = rt.Context(sc, rt.Params.make(3840, 2160, 4))
        c.render(); c.stats()
        t1 = time.perf_counter()
        ms = []
        for _ in range(6):
            c.render(); ms.append(c.stats().ms_total)
        s = c.stats()
        m = sorted(ms)[len(ms) // 2]
        print("%-12s %s  frame %7.3f ms  %6.2f Mrays (%5.2f secondary, %6.2f shadow)  %7.0f Mrays/s  first frame incl. creation %6.1f ms  form %d chunks %d" %
              (name, "exact" if same else "MISMATCH", m, s.rays_total / 1e6, s.rays_secondary / 1e6, s.rays_shadow / 1e6, s.rays_total / m / 1e3, (t1 - t0) * 1e3, s.shadow_form, s.chunks), flush=True)
        c.close()
        if os.environ.get("SURVEY_COUNTS"):  # executed tests per ray, by kernel (the counting instantiations, in the form the timed context chose)
            form = rt.host.F_SHADOW_TABLES if s.shadow_form == 1 else rt.host.F_SHADOW_PACKETS
            cc = rt.Context(sc, rt.Params.make(3840, 2160, 4, flags=rt.host.F_COUNT_EXEC | form))
            cc.render(); e = cc.stats(); cc.close()
            for kn, kinds in e.exec_dict().items():
                rays = max(1, kinds.get("rays", 0))
                print("    %-16s rays %9d  per ray: %s" % (kn, kinds.get("rays", 0), "  ".join("%s %.2f" % (kd, v / rays) for kd, v in kinds.items() if kd != "rays" and v)), flush=True)
        sc.close(); os.remove(p)
    sys.exit(1 if bad else 0)
