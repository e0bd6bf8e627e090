import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
W, H = 3840, 2160
prev = [0] * 8
for md in (0, 1, 2, 3, 4):
    ctx = rt.Context(sc, rt.Params.make(W, H, md, flags=rt.host.F_TIME_KERNELS | rt.host.F_PACKET_STATS))
    ctx.render(); st = ctx.stats()
    ps = list(st.packet_stats)
    d = [a - b for a, b in zip(ps, prev)]; prev = ps
    print("level %d rays %d: frame ms %.2f" % (md, st.rays_level[md], st.ms_total))
    print("  closest: waves %d slow %d abn %d tested/ray %.1f" % (d[0], d[1] & 0xffffffff, d[1] >> 32, d[3]/max(st.rays_level[md],1)))
    print("  shadow : blocks %d ok %d fallback rays %d batches/block %.1f survivors/block %.1f" % (d[4], d[5] & 0xffffffff, d[5] >> 32, d[6]/max(d[4],1), d[7]/max(d[4],1)))
    ctx.close()
