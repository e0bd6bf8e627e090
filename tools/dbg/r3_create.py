"""Context creation, phase by phase (library built with EXTRA=-DRTX_AB_KNOBS; RTX_VERBOSE=1).  usage: r3_create.py [workload | a scene of perf_survey.py]"""
import sys, os, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
p = os.path.join(tempfile.gettempdir(), "crdbg.txt")
if name in rt.scenegen.WORKLOADS:
    wl = rt.scenegen.WORKLOADS[name]
    wl["writer"](p)
else:
    import importlib.util
    spec = importlib.util.spec_from_file_location("perf_survey", os.path.join(ROOT, "tools", "dbg", "perf_survey.py"))
    ps = importlib.util.module_from_spec(spec); spec.loader.exec_module(ps)
    ps.SCENES[name](p)
    wl = dict(width=3840, height=2160, max_depth=4)
sc = rt.load_scene(p)
for i in range(2):
    t0 = time.perf_counter()
    ctx = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"]))
    print("create %d: %.1f ms" % (i, (time.perf_counter() - t0) * 1e3), flush=True)
    ctx.close()
