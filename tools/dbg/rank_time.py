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

# Render + GATHER, emulated on one device (round 3): the in-process multi-device call with every rank on device 0
# (RTX_F_MULTI_ON_ONE_DEVICE: N resident contexts, device-to-device copies where a node has RCCL, one copy to the host).  The ranks
# run one after the other here, so   gather + host copy  ~  t(call) - sum of the ranks' own frame times;   on N GPUs the ranks
# overlap and a frame costs  max rank + gather.  Still an emulation: no xGMI link is exercised.
import time
import numpy as np
one, _ = rt.render_rgb8(sc, rt.Params.make(3840, 2160, 4))
for N in (2, 4, 8):
    p = rt.Params.make(3840, 2160, 4, band_rows=30, flags=rt.host.F_MULTI_ON_ONE_DEVICE)
    img, st = rt.render_rgb8(sc, p, ngpus=N)          # creates the N contexts
    assert np.array_equal(img, one)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        rt.render_rgb8(sc, p, ngpus=N)
        ts.append((time.perf_counter() - t0) * 1e3)
    ranks = []
    for r in range(N):
        ctx = rt.Context(sc, rt.Params.make(3840, 2160, 4, nranks=N, rank=r, band_rows=30))
        for _ in range(4):
            ctx.render()
        ranks.append(ctx.stats().ms_total)
        ctx.close()
    t = sorted(ts)[len(ts) // 2]
    print("N=%d on one device: call %.2f ms (resident), ranks' frames sum %.2f ms (max %.2f) -> gather + 24.9 MB host copy + call overhead %.2f ms" %
          (N, t, sum(ranks), max(ranks), t - sum(ranks)))
rt.host.release_cached_context()
