"""Three RTX_F_SERIAL frames of a workload (for a kernel trace whose launches do not overlap).  usage: serial_frames.py [workload]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
p = os.path.join(tempfile.gettempdir(), "serdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], flags=rt.host.F_SERIAL))
for _ in range(3):
    ctx.render()
print(ctx.stats().ms_total)
ctx.close()
