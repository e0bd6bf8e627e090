"""Host time to enqueue one frame (rtx_context_render returns before the GPU is done) against the GPU frame time,
whole frame and a rank's share of 8: the enqueue must stay below the GPU time or the host bounds the frame rate."""
import sys, os, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
for (N, r) in ((1, 0), (8, 4)):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=N, rank=r, band_rows=30 if N > 1 else 0))
    for _ in range(3):
        ctx.render()
    ctx.stats()
    t0 = time.perf_counter()
    K = 20
    enq = []
    for _ in range(K):
        a = time.perf_counter(); ctx.render(); enq.append(time.perf_counter() - a)
    st = ctx.stats()   # waits for the last frame
    t1 = time.perf_counter()
    print("N=%d: enqueue %.3f ms per frame (min %.3f), %d frames back to back %.3f ms per frame, GPU frame %.3f ms" % (N, sum(enq) / K * 1e3, min(enq) * 1e3, K, (t1 - t0) / K * 1e3, st.ms_total))
    ctx.close()
