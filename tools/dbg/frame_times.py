"""Per-kernel event times of the 4K C4 frame at depth 1 and depth 5 (RTX_F_TIME_KERNELS)."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
for md in (0, 4):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, md, flags=rt.host.F_TIME_KERNELS))
    for _ in range(4): ctx.render()
    st = ctx.stats()
    print("  depth %d: frame %.3f ms  shadow passes %.3f ms  lanes %.3f  closest_f %.3f" % (md, st.ms_total, st.ms_shadow, st.ms_closest_lanes, st.ms_closest))
    ctx.close()
