"""Frame time of ONE rank's share of the 4K benchmark frame (bands of `band_rows` rows, round-robin over N ranks),
measured on a single GPU: what strong scaling over N GPUs can reach at best (no gather)."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
for (N, band) in ((1, 0), (2, 30), (4, 30), (8, 30), (8, 90), (8, 270)):
    tt = []
    for r in sorted({0, N // 2, N - 1}):
        ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=N, rank=r, band_rows=band))
        for _ in range(4):
            ctx.render()
        st = ctx.stats()
        tt.append((r, st.ms_total, st.rays_total))
        ctx.close()
    print("N=%d band_rows=%d:" % (N, band), ", ".join("rank %d %.3f ms (%.1f Mrays)" % (r, ms, rays / 1e6) for r, ms, rays in tt))
