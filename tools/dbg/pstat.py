import sys, os, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ctx = rt.Context(sc, rt.Params.make(W, H, 4, flags=rt.host.F_TIME_KERNELS | rt.host.F_PACKET_STATS))
ctx.render(); st = ctx.stats()
ps = list(st.packet_stats)
print("frame ms", st.ms_total, "closest", st.ms_closest, "shadow", st.ms_shadow)
print("closest: waves %d ok %d (%.1f%%) batches %d survivors %d (%.2f per batch)" % (ps[0], ps[1], 100.0*ps[1]/max(ps[0],1), ps[2], ps[3], ps[3]/max(ps[2],1)))
print("shadow : waves %d ok %d (%.1f%%) batches %d survivors %d (%.2f per batch)" % (ps[4], ps[5], 100.0*ps[5]/max(ps[4],1), ps[6], ps[7], ps[7]/max(ps[6],1)))
