"""Packet statistics of the C5 scene at 1080p (packet form of the shadow stage): blocks, fallbacks, batches and survivors per block."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
wl = rt.scenegen.WORKLOADS["c5"]
p = os.path.join(tempfile.gettempdir(), "c5dbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
for md in (0, 2):
    ctx = rt.Context(sc, rt.Params.make(1920, 1080, md, flags=rt.host.F_TIME_KERNELS | rt.host.F_PACKET_STATS))
    ctx.render(); st = ctx.stats(); ps = list(st.packet_stats)
    print("depth %d: frame %.2f ms shadow %.2f closest %.2f lanes %.2f | rays shadow %d" % (md, st.ms_total, st.ms_shadow, st.ms_closest, st.ms_closest_lanes, st.rays_shadow))
    print("  shadow: blocks %d ok %d fallback %d batches/block %.1f survivors/block %.1f" % (ps[4], ps[5] & 0xffffffff, ps[5] >> 32, ps[6] / max(ps[4], 1), (ps[7] & 0xffffffff) / max(ps[4], 1)))
    print("  closest gen0: waves %d ok %d batches/wave %.1f kept/wave %.1f" % (ps[0], ps[1] & 0xffffffff, ps[2] / max(ps[0], 1), ps[3] / max(ps[0], 1)))
    ctx.close()
