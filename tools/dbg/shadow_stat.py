import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
for md in (0, 4):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, md, flags=rt.host.F_PACKET_STATS))
    ctx.render(); ctx.render()
    st = ctx.stats()
    ps = list(st.packet_stats)
    w = ps[4] & 0xffffff
    print("depth %d: shadow waves %d listed %d avg list %.1f | batches/wave %.1f aff survivors/wave %.2f" % (md, w, (ps[4] >> 24) & 0xffff, (ps[4] >> 40) / w, ps[6] / w, (ps[7] >> 32) / w))
    ctx.close()
