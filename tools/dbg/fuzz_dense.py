"""Dense-scene fuzz: thousands of (mostly affine) spheres and some triangles overlapping many deep in a small box, reflective and
refractive, depth 5 -- the scenes whose rays overflow LexAcc's list of affine hits and take the ordered scans (ordered_scan for
secondary rays, the camera cell's own list for primary rays): culled kernels against the exact-only kernels, bit for bit, with the
counts of flagged rays.  usage: fuzz_dense.py [seeds]"""
import sys, os, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt


def make_scene(path, seed):
    rng = np.random.default_rng(11000 + seed)
    ns, nt, nl = int(rng.integers(1500, 4000)), int(rng.integers(0, 1500)), int(rng.integers(1, 4))
    lights = [(float(rng.uniform(-30, 30)), float(rng.uniform(-20, 40)), float(rng.uniform(-80, 10)), 0.3, 0.3, 0.3) for _ in range(nl)]
    rt.scenegen.write_scene(path, ns, nt, lights, seed=8000 + seed, box=(float(rng.uniform(6, 14)), float(rng.uniform(4, 9))), zfar=-60.0, znear=-30.0,
                            frac_affine=float(rng.uniform(0.5, 1.0)), frac_reflective=float(rng.uniform(0.3, 0.8)), frac_refractive=float(rng.uniform(0.0, 0.3)),
                            affine=("rotated", "diag", "general")[seed % 3], radius=(0.4, float(rng.uniform(2.0, 4.0))), edge=float(rng.uniform(1.0, 6.0)))
    return ns, nt, int(rng.integers(2, 5))


if __name__ == "__main__":
    bad = 0
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 9):
        p = os.path.join(tempfile.gettempdir(), "fuzzde_%d.txt" % seed)
        ns, nt, md = make_scene(p, seed)
        sc = rt.load_scene(p)
        a, sa = rt.render_f32(sc, rt.Params.make(480, 270, md, flags=rt.host.F_PACKET_STATS))
        b, sb = rt.render_f32(sc, rt.Params.make(480, 270, md, flags=rt.host.F_EXACT_ONLY))
        same = (a.view(np.uint32) == b.view(np.uint32)).all(axis=-1)
        ok = bool(same.all()) and (sa.rays_secondary, sa.rays_shadow, sa.hits) == (sb.rays_secondary, sb.rays_shadow, sb.hits)
        bad += 0 if ok else 1
        ps = [int(v) for v in sa.packet_stats]
        print("seed %d: %d spheres %d triangles depth %d: %s (%d px differ; secondary %d; flagged secondary %d, primary by list %d / by scan %d; %.1f ms)" %
              (seed, ns, nt, md, "ok" if ok else "MISMATCH", int((~same).sum()), sa.rays_secondary, ps[1] & 0xffffffff, ps[2] & 0xffffffff, ps[2] >> 32, sa.ms_total), flush=True)
        sc.close(); os.remove(p)
    sys.exit(1 if bad else 0)
