"""Scale fuzz: the random scenes of fuzz_exact.py with every length multiplied by 1e-2 ... 1e3 (the library's absolute constants --
the reference's own 1e-4 origin bias, the tables' delta, the guard box -- meet scenes 100 times smaller and 1,000 times larger than
the benchmark's): culled kernels against the exact-only kernels, bit for bit.  usage: fuzz_scale.py [seeds]"""
import sys, os, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt


def make_scene(path, seed):
    rng = np.random.default_rng(9000 + seed)
    s = float([1e-2, 1e-1, 10.0, 1e2, 1e3][seed % 5])
    ns, nt, nl = int(rng.integers(50, 2500)), int(rng.integers(0, 2500)), int(rng.integers(1, 5))
    lights = [(float(rng.uniform(-45, 45)) * s, float(rng.uniform(-30, 45)) * s, float(rng.uniform(-130, 20)) * s, 0.3, 0.3, 0.3) for _ in range(nl)]
    rt.scenegen.write_scene(path, ns, nt, lights, seed=6000 + seed, zfar=-120.0 * s, znear=-20.0 * s, box=(40.0 * s, 25.0 * s),
                            frac_affine=float(rng.uniform(0, 0.5)), frac_reflective=float(rng.uniform(0.1, 0.8)),
                            frac_refractive=float(rng.uniform(0, 0.3)), affine="general" if seed % 2 else "diag",
                            radius=(0.2 * s, float(rng.uniform(0.5, 5.0)) * s), edge=float(rng.uniform(0.5, 10.0)) * s)
    return s, ns, nt, int(rng.integers(1, 5))


if __name__ == "__main__":
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        p = os.path.join(tempfile.gettempdir(), "fuzzsc_%d.txt" % seed)
        s, ns, nt, md = make_scene(p, seed)
        sc = rt.load_scene(p)
        a, sa = rt.render_f32(sc, rt.Params.make(400, 225, md))
        b, sb = rt.render_f32(sc, rt.Params.make(400, 225, md, flags=rt.host.F_EXACT_ONLY))
        same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1)
        ok = bool(same.all()) and (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
        bad += 0 if ok else 1
        print("seed %d: scale %g, %d spheres %d triangles depth %d: %s (%d px differ; hits %d, rays %d/%d, form %d)" %
              (seed, s, ns, nt, md, "ok" if ok else "MISMATCH", int((~same).sum()), sa.hits, sa.rays_total, sb.rays_total, sa.shadow_form), flush=True)
        sc.close(); os.remove(p)
    sys.exit(1 if bad else 0)
