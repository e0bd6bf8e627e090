"""Random scenes under random CAMERAS (position inside or outside the cloud, any view direction, up vector not orthogonal to it,
field of view from a few degrees to > 150): the culled kernels -- the primary rays' direction-cell table first of all -- against
the exact-only kernels, bit for bit.  usage: fuzz_camera.py [seeds]"""
import sys, os, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    rng = np.random.default_rng(5000 + seed)
    ns, nt = int(rng.integers(50, 2500)), int(rng.integers(0, 2500))
    nl = int(rng.integers(1, 4))
    lights = [(float(rng.uniform(-45, 45)), float(rng.uniform(-30, 45)), float(rng.uniform(-130, 20)), 0.3, 0.3, 0.3) for _ in range(nl)]
    p = os.path.join(tempfile.gettempdir(), "fuzzcam_%d.txt" % seed)
    rt.scenegen.write_scene(p, ns, nt, lights, seed=3000 + seed, frac_affine=float(rng.uniform(0, 0.5)),
                            frac_reflective=float(rng.uniform(0.1, 0.8)), frac_refractive=float(rng.uniform(0, 0.3)),
                            affine="general" if seed % 2 else "diag", radius=(0.2, float(rng.uniform(0.5, 5.0))), edge=float(rng.uniform(0.5, 10.0)))
    # a camera of this seed's own in place of the generator's
    cam = rng.uniform(-1, 1, 3) * float(rng.choice([0.1, 0.5, 1.0, 4.0]))
    if abs(cam[2]) < 0.05: cam[2] = 0.3
    vrp = np.array([rng.uniform(-45, 45), rng.uniform(-30, 30), rng.uniform(-130, 10)]) if seed % 3 else np.array([0.0, 0.0, 0.0])
    vpn = rng.normal(size=3); vpn /= np.linalg.norm(vpn)
    up = rng.normal(size=3); up /= np.linalg.norm(up)
    out = []
    for line in open(p).read().splitlines():
        w = line.split()
        if w and w[0] == "camera": line = "camera %.5f %.5f %.5f" % tuple(cam)
        elif w and w[0] == "vrp": line = "vrp %.4f %.4f %.4f" % tuple(vrp)
        elif w and w[0] == "vpn": line = "vpn %.6f %.6f %.6f" % tuple(vpn)
        elif w and w[0] == "up": line = "up %.6f %.6f %.6f" % tuple(up)
        out.append(line)
    open(p, "w").write("\n".join(out) + "\n")
    sc = rt.load_scene(p)
    W, H, md = 400, 225, int(rng.integers(1, 5))
    a, sa = rt.render_f32(sc, rt.Params.make(W, H, md))
    b, sb = rt.render_f32(sc, rt.Params.make(W, H, md, flags=rt.host.F_EXACT_ONLY))
    same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1) if a.ndim == 3 else (a.view(np.uint32) == b.view(np.uint32))
    ok = bool(same.all()) and (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
    bad += 0 if ok else 1
    print("seed %d: %d spheres %d triangles, camera %s vrp %s depth %d: %s (%d px differ; hits %d, rays %d/%d)" %
          (seed, ns, nt, np.round(cam, 2).tolist(), np.round(vrp, 1).tolist(), md, "ok" if ok else "MISMATCH", int((~same).sum()), sa.hits, sa.rays_total, sb.rays_total), flush=True)
    sc.close(); os.remove(p)
sys.exit(1 if bad else 0)
