import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
prev = None
for md in (0, 1):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, md, flags=rt.host.F_PACKET_STATS))
    ctx.render(); st = ctx.stats()
    ps = list(st.packet_stats)
    if prev: print("L1 rays", st.rays_level[1], "waves", (st.rays_level[1] + 63) // 64, "packet waves", ps[0] - prev[0], "batches/packet wave", (ps[2] - prev[2]) / max(ps[0] - prev[0], 1), "kept/packet wave", (ps[3] - prev[3]) / max(ps[0] - prev[0], 1))
    prev = ps
    ctx.close()
