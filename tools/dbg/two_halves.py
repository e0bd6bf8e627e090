"""Experiment: the 4K frame as ONE context vs as two half-frame contexts rendered concurrently on two streams
(how much the front end of one half hides behind the shadow pass of the other)."""
import sys, os, tempfile, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
hip = C.CDLL("libamdhip64.so")
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
p = os.path.join(tempfile.gettempdir(), "abdbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
W, H = 3840, 2160
def streams(n):
    out = []
    for _ in range(n):
        s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; out.append(s)
    return out
def timeit(ctxs, ss, reps=10):
    ts = []
    for _ in range(reps):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for c, s in zip(ctxs, ss): c.render(s, None)
        for s in ss: hip.hipStreamSynchronize(s)
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]
full = rt.Context(sc, rt.Params.make(W, H, 4))
print("GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES"))
print("one context: %.3f ms" % timeit([full], streams(1)))
for parts, band in ((2, 1080), (2, 30), (3, 720), (4, 540)):
    ctxs = [rt.Context(sc, rt.Params.make(W, H, 4, nranks=parts, rank=r, band_rows=band)) for r in range(parts)]
    print("%d contexts (band %d) concurrently: %.3f ms; one after the other on one stream: %.3f ms" % (parts, band, timeit(ctxs, streams(parts)), timeit(ctxs, streams(1) * parts)))
    for c in ctxs: c.close()
