import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
prev = None
for N in (8,):
  for md in (1, 2, 3, 4):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, md, nranks=N, rank=N // 2, band_rows=30 if N > 1 else 0, flags=rt.host.F_PACKET_STATS | rt.host.F_TIME_KERNELS))
    for _ in range(2):
        ctx.render()
    st = ctx.stats()
    ps = list(st.packet_stats)
    w = ps[0] & 0xfffff
    print("N=%d depth %d: lanes %.3f ms in %d launches; sampled waves %d (sums over generations) max wave %.1f us; avg us: total %.1f phase0 %.1f | cluster %.1f sub %.1f members %.1f flush %.1f | iters sub %.1f members %.1f exact %.1f" % (
        N, md, st.ms_closest_lanes, st.launches_closest_lanes, w, ps[1] / 100., (ps[2] & ((1 << 40) - 1)) / 100. / w, ps[3] / 100. / w, ps[4] / 100. / w, ps[5] / 100. / w, ps[6] / 100. / w, ps[7] / 100. / w,
        (ps[0] >> 40) / w, ((ps[0] >> 20) & 0xfffff) / w, (ps[2] >> 40) / w))
    ctx.close()
