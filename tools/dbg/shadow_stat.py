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
    w = ps[4] & 0xffffffff
    print("depth %d: sampled shadow waves %d; per wave: batches %.1f survivors %.1f affine survivors %.1f exact runs %.1f affine exact runs %.1f lanes per exact run %.1f fallback rays %.3f" % (
        md, w, (ps[4] >> 32) / w, (ps[5] & 0xffffffff) / w, (ps[5] >> 32) / w, (ps[6] & 0xffffffff) / w, (ps[6] >> 32) / w,
        (ps[7] & ((1 << 40) - 1)) / max(1, (ps[6] & 0xffffffff) + (ps[6] >> 32)), (ps[7] >> 40) / w))
    ctx.close()
