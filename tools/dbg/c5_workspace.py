"""C5 frame time against the workspace budget (row chunks per frame), one GPU."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
wl = rt.scenegen.WORKLOADS["c5"]
p = os.path.join(tempfile.gettempdir(), "c5dbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
for gb in (16, 32, 64, 128):
    ctx = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], workspace_bytes=gb << 30))
    ms = []
    for _ in range(3):
        ctx.render(); ms.append(ctx.stats().ms_total)
    print("workspace %d GiB: %.1f ms (rays %.1f M)" % (gb, min(ms[1:]), ctx.stats().rays_total / 1e6), flush=True)
    ctx.close()
