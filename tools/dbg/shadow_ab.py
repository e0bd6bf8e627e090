"""Shadow-pass A/B aid: serial (non-overlapped) kernel times and the pass's executed-test counts for the library
RTX_LIB names.  usage: shadow_ab.py [workload] [width height]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
F = rt.host
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
W = int(sys.argv[2]) if len(sys.argv) > 3 else wl["width"]
H = int(sys.argv[3]) if len(sys.argv) > 3 else wl["height"]
p = os.path.join(tempfile.gettempdir(), "sabdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(W, H, wl["max_depth"], flags=F.F_SERIAL | F.F_TIME_KERNELS))
best = None
for _ in range(6):
    ctx.render()
    s = ctx.stats()
    if best is None or s.ms_shadow < best[0]:
        best = (s.ms_shadow, s.ms_closest, s.ms_closest_lanes, s.ms_total)
ctx.close()
ctx = rt.Context(sc, rt.Params.make(W, H, wl["max_depth"], flags=F.F_COUNT_EXEC))
ctx.render()
ex = ctx.stats().exec_dict()["k_shadow_pass"]
ctx.close()
print("serial ms: shadow %.3f closest_f %.3f closest_lanes %.3f frame %.3f | " % best + " ".join("%s=%.4g" % (k, v) for k, v in ex.items() if v))
