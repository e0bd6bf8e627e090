"""Random scenes: the culled kernels against the exact-only kernels (every primitive, file order), bit for bit."""
import sys, os, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    rng = np.random.default_rng(seed)
    ns, nt = int(rng.integers(50, 4000)), int(rng.integers(0, 4000))
    nl = int(rng.integers(1, 6))
    lights = [(float(rng.uniform(-45, 45)), float(rng.uniform(-30, 45)), float(rng.uniform(-130, 20)), 0.3, 0.3, 0.3) for _ in range(nl)]
    p = os.path.join(tempfile.gettempdir(), "fuzz_%d.txt" % seed)
    rt.scenegen.write_scene(p, ns, nt, lights, seed=1000 + seed, frac_affine=float(rng.uniform(0, 0.6)),
                            frac_reflective=float(rng.uniform(0.1, 0.8)), frac_refractive=float(rng.uniform(0, 0.4)),
                            affine="general" if seed % 2 else "diag", radius=(0.2, float(rng.uniform(0.5, 5.0))), edge=float(rng.uniform(0.5, 10.0)))
    sc = rt.load_scene(p)
    W, H, md = 480, 270, int(rng.integers(1, 6))
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, md))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, md, flags=rt.host.F_EXACT_ONLY))
    same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1) if a.ndim == 3 else (a.view(np.uint32) == b.view(np.uint32))
    ok = bool(same.all()) and (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    bad += 0 if ok else 1
    print("seed %d: %d spheres %d triangles %d lights depth %d: %s (%d px differ; rays %d/%d)" % (seed, ns, nt, nl, md, "ok" if ok else "MISMATCH", int((~same).sum()), sa.rays_total, sb.rays_total), flush=True)
    sc.close(); os.remove(p)
sys.exit(1 if bad else 0)
