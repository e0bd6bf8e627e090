"""Render a workload's frame a few times on one stream (RTX_F_SERIAL) and print the rays of each generation: run under
rocprofv3 --kernel-trace, tools/dbg/r3_timeline.sh prints the last frame's launches.  usage: serial_frames.py [workload] [nranks] [serial 1|0]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
F = rt.host
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
p = os.path.join(tempfile.gettempdir(), "serdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ser = F.F_SERIAL if (len(sys.argv) <= 3 or sys.argv[3] == "1") else 0
c = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], nranks=N, rank=N // 2, band_rows=30 if N > 1 else 0, flags=ser | F.F_SHADOW_TABLES))
for _ in range(4): c.render()
s = c.stats()
print("rays per generation:", [int(s.rays_level[k]) for k in range(s.levels)], "frame %.3f ms" % s.ms_total)
c.close()
