"""Context creation, phase by phase (library built with EXTRA=-DRTX_AB_KNOBS; RTX_VERBOSE=1).  usage: r3_create.py [workload]"""
import sys, os, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
p = os.path.join(tempfile.gettempdir(), "crdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
for i in range(2):
    t0 = time.perf_counter()
    ctx = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"]))
    print("create %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    ctx.close()
