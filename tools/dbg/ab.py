"""A/B aid: frame time of the 4K benchmark frame and of one rank's share of it (N = 2, 4, 8; bands of 30 rows), with
optional extra flags.  usage: ab.py [flags_int] [workload]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
wl = rt.scenegen.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "c4"]
p = os.path.join(tempfile.gettempdir(), "abdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
out = []
for (N, band) in ((1, 0), (2, 30), (4, 30), (8, 30)):
    best = 0.0
    for r in sorted({0, N // 2}):
        ctx = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], nranks=N, rank=r, band_rows=band, flags=flags))
        ms = []
        for _ in range(8):
            ctx.render()
            ms.append(ctx.stats().ms_total)
        ctx.close()
        best = max(best, sorted(ms)[len(ms) // 2])
    out.append("N=%d %.3f ms" % (N, best))
print("flags=%d:" % flags, ", ".join(out))
