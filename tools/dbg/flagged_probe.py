"""How many rays of a dense affine scene take the ordered scan (rtx_stats.packet_stats[1]): picks the test's configuration."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "flagged.txt")
for (ns, fa, rmax, aff, w, h) in ((1500, 0.5, 2.5, "rotated", 640, 360), (3000, 0.8, 3.0, "rotated", 640, 360), (3000, 0.8, 3.0, "diag", 640, 360), (3000, 1.0, 4.0, "diag", 480, 270), (5000, 1.0, 4.0, "rotated", 480, 270)):
    rt.scenegen.write_scene(p, ns, 500, rt.scenegen.lights_c4(), seed=77, box=(10.0, 6.0), zfar=-60.0, znear=-30.0, frac_affine=fa, frac_reflective=0.6, radius=(0.5, rmax), affine=aff)
    sc = rt.load_scene(p)
    a, sa = rt.render_f32(sc, rt.Params.make(w, h, 3, flags=rt.host.F_PACKET_STATS))
    print(ns, fa, rmax, aff, w, h, "secondary", sa.rays_secondary, "flagged", int(sa.packet_stats[1]) & 0xffffffff, "abnormal", int(sa.packet_stats[1]) >> 32, "ms", sa.ms_total, flush=True)
    sc.close()
