"""Frame time of the whole 4K frame and of ranks 0 / 4 / 7 of 8 (bands of 30 rows), one GPU."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
out = []
for (N, r) in ((1, 0), (2, 0), (4, 2), (8, 0), (8, 4), (8, 7)):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=N, rank=r, band_rows=30 if N > 1 else 0))
    ms = []
    for _ in range(8):
        ctx.render(); ms.append(ctx.stats().ms_total)
    out.append("N=%d r%d %.3f" % (N, r, min(ms[2:])))
    ctx.close()
print("  ".join(out))
