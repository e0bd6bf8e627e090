"""Render rank 4's share of the 4K frame at N = 8 a few times (for kernel traces)."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=8, rank=4, band_rows=30))
for _ in range(4):
    ctx.render()
print(ctx.stats().ms_total)
ctx.close()
