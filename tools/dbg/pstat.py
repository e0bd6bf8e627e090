"""Packet statistics of a workload's frame (RTX_F_PACKET_STATS).  usage: pstat.py [workload]"""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
F = rt.host
wl = rt.scenegen.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
p = os.path.join(tempfile.gettempdir(), "psdbg.txt")
wl["writer"](p)
sc = rt.load_scene(p)
c = rt.Context(sc, rt.Params.make(wl["width"], wl["height"], wl["max_depth"], flags=F.F_PACKET_STATS | F.F_SHADOW_TABLES))
c.render(); s = c.stats(); c.close()
ps = [int(v) for v in s.packet_stats]
print("waves %d, ok/slow %d, abnormal lanes %d, batches %d, tested %d" % (ps[0], ps[1] & 0xffffffff, ps[1] >> 32, ps[2], ps[3]))
print("rays per generation", [int(s.rays_level[k]) for k in range(s.levels)])
