import sys, glob, os, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rtxamd as rt, oracle_lib as O
from parity import raw_equal_mask
for p in sorted(glob.glob("tests/scenes/*.txt")):
    sc = rt.load_scene(p)
    W,H=160,120
    cpu,_ = O.render_f32(sc, W,H,4)
    for name, fl in (("filt",0),("exact",rt.host.F_EXACT_ONLY)):
        g, st = rt.render_f32(sc, rt.Params.make(W,H,4,flags=fl))
        same = raw_equal_mask(g,cpu)
        bad = np.argwhere(~same)
        print(os.path.basename(p), name, "differ", len(bad), "first", bad[:3].tolist(), "hits", st.hits, "rays", st.rays_primary+st.rays_secondary)
