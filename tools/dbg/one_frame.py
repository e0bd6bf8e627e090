"""Render the 4K benchmark frame eight times (for counter passes; the context's one calibration pass -- a closest-hit +
set-up sweep at creation -- is then a few per cent of the closest-hit kernels' totals)."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
# the form the measured choice takes for this frame (bench.py: config.shadow_form), forced so that no timing frames of the other form run under the counters
ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, flags=rt.host.F_SHADOW_TABLES))
for _ in range(8):
    ctx.render()
print(ctx.stats().ms_total)
ctx.close()
