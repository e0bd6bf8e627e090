"""Executed-work counters and serial kernel times of the shadow stage on a workload.  usage: r3_counts.py [workload]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
F = rt.host
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
p = os.path.join(tempfile.gettempdir(), "cntdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
c = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], flags=F.F_COUNT_EXEC))
c.render(); ex = c.stats().exec_dict(); st = c.stats(); c.close()
rays = st.rays_shadow
print("shadow rays %d" % rays)
for k, v in ex["k_shadow_pass"].items():
    if v: print("  %-12s %12d  %.2f / ray" % (k, v, v / max(rays, 1)))
sec = st.rays_secondary
print("secondary rays %d" % sec)
for k, v in ex["k_closest_lanes"].items():
    if v: print("  %-12s %12d  %.2f / ray" % (k, v, v / max(sec, 1)))
c = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], flags=F.F_SERIAL | F.F_TIME_KERNELS))
for _ in range(3): c.render()
s = c.stats(); c.close()
print("serial: frame %.3f ms, shadow brackets %.3f ms (%d), closest_f %.3f, closest_lanes %.3f" % (s.ms_total, s.ms_shadow, s.launches_shadow, s.ms_closest, s.ms_closest_lanes))
