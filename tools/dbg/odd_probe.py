"""affine_odd survey scene: how many secondary rays are abnormal / flagged, and executed tests per ray."""
import sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rtxamd as rt
p = os.path.join(tempfile.gettempdir(), "odd.txt")
rt.scenegen.write_scene(p, 3000, 0, rt.scenegen.lights_c4(), frac_affine=1.0, affine="general")
sc = rt.load_scene(p)
rec, guard = rt.filter_records(sc)
import numpy as np
print("guard", guard, "unbounded closest records", int(np.isneginf(rec[:, 3]).sum()), "of", rec.shape[0], "largest finite R", float(np.sqrt(-rec[np.isfinite(rec[:, 3]), 3]).max()))
a, sa = rt.render_f32(sc, rt.Params.make(960, 540, 4, flags=rt.host.F_PACKET_STATS))
print("secondary", sa.rays_secondary, "flagged", int(sa.packet_stats[1]) & 0xffffffff, "abnormal", int(sa.packet_stats[1]) >> 32, "levels", [int(sa.rays_level[k]) for k in range(sa.levels)], "ms", sa.ms_total)
