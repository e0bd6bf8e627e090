import sys, os, tempfile
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "c4dbg.txt")
rt.scenegen.write_c4(p)
sc = rt.load_scene(p)
prev = [0]*8
for md in (0, 1, 2, 3):
    ctx = rt.Context(sc, rt.Params.make(3840, 2160, md, flags=rt.host.F_PACKET_STATS))
    ctx.render(); st = ctx.stats()
    ps = list(st.packet_stats); d = [a-b for a, b in zip(ps, prev)]; prev = ps
    print(md, st.rays_level[md], "waves", d[0], "slow", d[1] & 0xffffffff, "abn", d[1] >> 32, "flushes/wave", d[2]/max(d[0],1), "tested/ray", d[3]/max(st.rays_level[md],1))
    ctx.close()
