import sys, os, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
W, H = 3840, 2160
for md in (0, 4):
    ctx = rt.Context(sc, rt.Params.make(W, H, md, flags=rt.host.F_TIME_KERNELS))
    ctx.render(); ctx.render(); st = ctx.stats()
    print("md", md, "ms", st.ms_total, "closest", st.ms_closest, "shadow", st.ms_shadow)
    print(" rays", st.rays_primary, st.rays_secondary, st.rays_shadow, "hits", st.hits, "levels", list(st.rays_level)[:6])
    print(" shadow kinds", list(st.tests_shadow_kind), "affine rays", st.tests_shadow_kind[1] / 500.0)
    ctx.close()
