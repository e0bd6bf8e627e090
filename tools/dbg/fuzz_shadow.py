"""Shadow-stage fuzz: random scenes in front of two big wall triangles (every primary ray hits something, so shadow rays start all
over the scene and on the walls behind it), 24 random lights each -- inside the cloud, outside, behind the walls, next to
primitives -- depth 1: the (light, cell) tables and the walk against the exact-only kernels, bit for bit.  usage: fuzz_shadow.py [seeds]"""
import sys, os, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt


def make_scene(path, seed):
    rng = np.random.default_rng(7000 + seed)
    ns, nt = int(rng.integers(20, 1500)), int(rng.integers(0, 1500))
    lights = []
    for _ in range(24):
        kind = rng.integers(0, 4)
        if kind == 0: p = (rng.uniform(-45, 45), rng.uniform(-30, 30), rng.uniform(-125, -15))      # inside the cloud
        elif kind == 1: p = (rng.uniform(-300, 300), rng.uniform(-300, 300), rng.uniform(-300, 300))  # far outside
        elif kind == 2: p = (rng.uniform(-45, 45), rng.uniform(-30, 30), rng.uniform(-200, -135))     # behind the walls
        else: p = (rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-5, 5))                        # at the camera
        lights.append((float(p[0]), float(p[1]), float(p[2]), 0.05, 0.05, 0.05))
    rt.scenegen.write_scene(path, ns, nt, lights, seed=4000 + seed, frac_affine=float(rng.uniform(0, 0.5)),
                            frac_reflective=0.0, frac_refractive=0.0, affine="general" if seed % 2 else "diag",
                            radius=(0.2, float(rng.uniform(0.5, 6.0))), edge=float(rng.uniform(0.5, 12.0)))
    # two wall triangles behind everything (z = -130), appended as the last records
    lines = open(path).read().splitlines()
    mat = "1 1 1  .8 .8 .8  .4 .4 .4  0 0 1 8"
    lines.append("triangle -400 -300 -130  400 -300 -130  400 300 -130  " + mat)
    lines.append("triangle -400 -300 -130  400 300 -130  -400 300 -130  " + mat)
    lines[0] = str(int(lines[0]) + 2)
    open(path, "w").write("\n".join(lines) + "\n")
    return ns, nt


if __name__ == "__main__":
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        p = os.path.join(tempfile.gettempdir(), "fuzzsh_%d.txt" % seed)
        ns, nt = make_scene(p, seed)
        sc = rt.load_scene(p)
        a, sa = rt.render_f32(sc, rt.Params.make(480, 270, 1))
        b, sb = rt.render_f32(sc, rt.Params.make(480, 270, 1, flags=rt.host.F_EXACT_ONLY))
        same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1)
        ok = bool(same.all()) and (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
        bad += 0 if ok else 1
        print("seed %d: %d spheres %d triangles 24 lights: %s (%d px differ; hits %d, shadow rays %d, form %d)" %
              (seed, ns, nt, "ok" if ok else "MISMATCH", int((~same).sum()), sa.hits, sa.rays_shadow, sa.shadow_form), flush=True)
        sc.close(); os.remove(p)
    sys.exit(1 if bad else 0)
