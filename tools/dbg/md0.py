import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
md = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = rt.Context(sc, rt.Params.make(3840, 2160, md, flags=rt.host.F_TIME_KERNELS))
for _ in range(3):
    ctx.render()
st = ctx.stats()
print("md", md, "ms", st.ms_total, "closest", st.ms_closest, "shadow", st.ms_shadow)
ctx.close()
