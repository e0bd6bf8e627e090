"""Render a benchmark frame eight times (C5: twice) (for counter passes; the context's one calibration pass -- a closest-hit +
set-up sweep at creation -- is then a few per cent of the closest-hit kernels' totals)."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]   # usage: one_frame.py [c4 | c4r | c5]
p = os.path.join(tempfile.gettempdir(), "pmcdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
# the form the measured choice takes for this frame (bench.py: config.shadow_form), forced so that no timing frames of the other form run under the counters
ctx = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], flags=rt.host.F_SHADOW_TABLES))
for _ in range(8 if wl["width"] <= 3840 else 2):
    ctx.render()
print(ctx.stats().ms_total)
ctx.close()
