import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
for H in (8, 4, 2, 1):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, 0, nranks=H, rank=H // 2, band_rows=30 if H > 1 else 0, flags=rt.host.F_PACKET_STATS | rt.host.F_TIME_KERNELS))
    for _ in range(3):
        ctx.render()
    st = ctx.stats()
    ps = list(st.packet_stats)
    w = ps[0]
    print("N=%d closest %.3f ms, waves %d, wave time max %.1f us avg %.1f us, max(surv,exact)=(%d,%d) avg surv %.1f exact %.1f batches %.1f" % (
        H, st.ms_closest, w, ps[1] / 100.0, ps[2] / 100.0 / w, ps[3] >> 32, ps[3] & 0xffffffff, ps[4] / w, ps[5] / w, ps[6] / w))
    ctx.close()
