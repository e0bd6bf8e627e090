"""Executed tests per shadow ray of the all-ellipsoid survey scene in either form of the shadow stage."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "affprobe.txt")
for aff in ("rotated", "diag"):
    rt.scenegen.write_scene(p, 3000, 0, rt.scenegen.lights_c4(), frac_affine=1.0, affine=aff)
    sc = rt.load_scene(p)
    for name, form in (("tables", rt.host.F_SHADOW_TABLES), ("packets", rt.host.F_SHADOW_PACKETS)):
        c = rt.Context(sc, rt.Params.make(3840, 2160, 4, flags=form))
        for _ in range(3): c.render()
        ms = c.stats().ms_total; c.close()
        cc = rt.Context(sc, rt.Params.make(3840, 2160, 4, flags=rt.host.F_COUNT_EXEC | form))
        cc.render(); e = cc.stats(); cc.close()
        k = e.exec_dict()["k_shadow_pass"]; rays = max(1, k["rays"])
        print(aff, name, "frame %.2f ms" % ms, "shadow rays", k["rays"], " ".join("%s %.1f" % (a, v / rays) for a, v in k.items() if v and a != "rays"), flush=True)
    sc.close()
