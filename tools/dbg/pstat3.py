import sys, os, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(3840, 2160, 0, flags=rt.host.F_PACKET_STATS))
ctx.render(); st = ctx.stats()
ctx.close()
