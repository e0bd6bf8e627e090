import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(3840, 2160, 0, flags=rt.host.F_PACKET_STATS))
ctx.render(); st = ctx.stats()
ps = list(st.packet_stats)
print("blocks", ps[4], "batches/block", ps[6] / ps[4], "main survivors/block", (ps[7] & 0xffffffff) / ps[4], "affine survivors/block", (ps[7] >> 32) / ps[4])
ctx.close()
