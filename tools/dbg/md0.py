"""Per-kernel event times of C4 at four frame sizes (640x480 ... 4K) for one depth.  usage: md0.py [max_depth]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
md = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for (W, H) in ((640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    ctx = rt.Context(sc, rt.Params.make(W, H, md, flags=rt.host.F_TIME_KERNELS))
    for _ in range(3):
        ctx.render()
    st = ctx.stats()
    print("%dx%d md %d: frame %.3f ms, closest_f %.3f, lanes %.3f (%d launches), shadow %.3f (%d launches), rays %d" % (
        W, H, md, st.ms_total, st.ms_closest, st.ms_closest_lanes, st.launches_closest_lanes, st.ms_shadow, st.launches_shadow, st.rays_total))
    ctx.close()
