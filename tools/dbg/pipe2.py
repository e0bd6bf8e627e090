"""Throughput of the 4K frame (N = 1) and of rank 4's share at N = 8 with 1, 2 or 3 frames in flight
(one context and one stream per frame in flight, frames issued round-robin), one GPU."""
import sys, os, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rtxamd as rt
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
def mkstream():
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return s
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
for (N, r) in ((1, 0), (2, 1), (4, 2), (8, 4)):
    line = []
    for depth in (1, 2, 3):
        ctxs = [rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=N, rank=r, band_rows=30 if N > 1 else 0)) for _ in range(depth)]
        streams = [None] + [mkstream() for _ in range(depth - 1)]
        best = 1e9
        for rep in range(3):
            for i in range(2 * depth):
                ctxs[i % depth].render(streams[i % depth], None)
            torch.cuda.synchronize()
            K = 24
            t0 = time.perf_counter()
            for i in range(K):
                ctxs[i % depth].render(streams[i % depth], None)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
        line.append("depth %d: %.3f ms" % (depth, best))
        for c in ctxs: c.close()
    print("N=%d rank %d  " % (N, r) + "  ".join(line), flush=True)
