"""Per-wave durations of k_closest_lanes for one rank's share of C4 (library built with EXTRA=-DRTX_WAVE_CLOCK): is a thin launch
bound by every wave's own chain or by a few long waves?  usage: wave_clock.py [nranks] [lanes16] [lanes4]"""
import sys, os, tempfile, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = os.path.join(tempfile.gettempdir(), "wclk.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
lib = rt.host.lib()
lib.rtx_debug_wave_clock.restype = C.c_int
lib.rtx_debug_wave_clock.argtypes = [C.c_void_p, C.c_int]
ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=N, rank=N // 2, band_rows=30 if N > 1 else 0))
buf = np.zeros((1 << 17, 3), dtype=np.uint64)
for _ in range(3): ctx.render()
ctx.stats()
lib.rtx_debug_wave_clock(buf.ctypes.data, buf.shape[0])   # discard the warm-up frames
ctx.render(); st = ctx.stats()
n = lib.rtx_debug_wave_clock(buf.ctypes.data, buf.shape[0])
rec = buf[:n].astype(np.int64)
print("frame %.3f ms, %d wave records" % (st.ms_total, n))
for nr in sorted(set(rec[:, 0]), reverse=True):
    r = rec[rec[:, 0] == nr]
    dur = (r[:, 2] - r[:, 1]) * 0.01   # us
    t0 = r[:, 1].min()
    start = (r[:, 1] - t0) * 0.01; end = (r[:, 2] - t0) * 0.01
    q = np.percentile(dur, [0, 10, 50, 90, 99, 100])
    print("launch of %7d rays: %5d waves; wave duration us min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f; last start %.1f us, last end %.1f us; sum of durations %.0f us" %
          (nr, len(r), q[0], q[1], q[2], q[3], q[4], q[5], start.max(), end.max(), dur.sum()))
ctx.close()
