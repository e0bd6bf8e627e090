"""What the cluster culling rests on (rtx_pack.cpp: build_clusters, exported by rtx_cluster_records): the packet and
cone tests reject a whole cluster or sub-cluster on its sphere alone, so every member's own filter sphere must lie
inside its sub-cluster's sphere and its cluster's, and every primitive that can pass at all must sit in exactly one
member slot.  Checked in float64 on the binary32 records the kernels read (host code only: no GPU)."""
import glob
import os
import tempfile

import numpy as np
import pytest

import rtxamd as rt
from conftest import SCENES


def _scene_paths():
    return sorted(glob.glob(os.path.join(SCENES, "*.txt")))


def _check_set(scene, which):
    cl, sub, mem, idx = rt.cluster_records(scene, which)
    rec, _guard = rt.filter_records(scene)
    v = scene.view
    ns, nt = v.nspheres, v.ntriangles
    ncl = cl.shape[0]
    # --- every primitive that can pass sits in exactly one slot -----------------------------------------------------
    slots = idx[idx >= 0]
    assert len(np.unique(slots)) == len(slots), "a primitive sits in two member slots"
    can_pass = ~(rec[:, 3] == np.inf)  # -R^2 = +inf: the record rejects every ray, the primitive is in no cluster
    if which != 1:
        want = set(np.nonzero(can_pass)[0].tolist())
    else:  # shadow casters: plain spheres and triangles (affine spheres have their own records, polygons cast none)
        plain = np.array([not scene.view.spheres[i].istransformed for i in range(ns)], dtype=bool)
        want = set(i for i in range(ns + nt) if (i >= ns or plain[i]))
        want = set(i for i in want if can_pass[i]) | (set(slots.tolist()) & set(i for i in want if not can_pass[i]))
    assert set(slots.tolist()) == want
    # padding slots never pass
    pad = idx < 0
    assert np.all(mem[pad][:, 3] == np.inf)
    # --- containment: member sphere inside its sub-cluster sphere inside ... its cluster sphere -------------------------
    m = mem.astype(np.float64)
    with np.errstate(all="ignore"):
        Rm = np.sqrt(-m[..., 3])                       # [ncl, 64]; padding -> nan
    bounded = np.isfinite(m[..., :3]).all(-1) & np.isfinite(m[..., 3]) & (m[..., 3] <= 0) & ~pad
    unbounded = ~pad & ~bounded
    c = cl.astype(np.float64)
    s = sub.astype(np.float64)
    for j in range(ncl):
        for g in range(8):
            lanes = slice(8 * g, 8 * g + 8)
            if unbounded[j, lanes].any():
                assert s[j, g, 3] == np.inf and c[j, 3] == np.inf
            b = bounded[j, lanes]
            if not b.any():
                continue
            cm = m[j, lanes, :3][b]
            rm = Rm[j, lanes][b]
            for sph in (s[j, g], c[j]):
                if sph[3] == np.inf:
                    continue
                reach = np.sqrt(((cm - sph[:3]) ** 2).sum(-1)) + rm
                assert np.all(reach <= sph[3]), (which, j, g, float(reach.max()), float(sph[3]))
    return ncl, len(slots)


@pytest.mark.parametrize("path", _scene_paths(), ids=os.path.basename)
def test_cluster_sets_of_the_test_scenes(path):
    sc = rt.load_scene(path)
    for which in (0, 1):
        _check_set(sc, which)
    sc.close()


def test_cluster_sets_of_a_generated_scene():
    """2,000 spheres (10 % affine) + 2,000 triangles: full clusters, k-d order, both sets."""
    p = os.path.join(tempfile.gettempdir(), "rtx_clusters_%d.txt" % os.getpid())
    rt.scenegen.write_c4(p, nspheres=2000, ntriangles=2000)
    sc = rt.load_scene(p)
    ncl0, n0 = _check_set(sc, 0)
    ncl1, n1 = _check_set(sc, 1)
    assert n0 == 4000 and 3700 <= n1 <= 3900
    # k-d order leaves at most one partly filled cluster per group of records (small / large / unbounded, per class)
    assert ncl0 <= (n0 + 63) // 64 + 6 and ncl1 <= (n1 + 63) // 64 + 3
    sc.close()
    os.remove(p)


def test_degenerate_inputs():
    """Coincident centres, zero-size and huge primitives: the build terminates and the invariants hold."""
    p = os.path.join(tempfile.gettempdir(), "rtx_clusters_deg_%d.txt" % os.getpid())
    mat = "1 1 1  .5 .5 .5  .3 .3 .3"
    lines = ["light 0 10 0 1 1 1"]
    lines += ["sphere 1 2 -30 0.5 %s 0 0 1.2 0 10" % mat] * 200                      # 200 identical spheres
    lines += ["sphere %d 0 -40 0 %s 0 0 1.2 0 10" % (i, mat) for i in range(70)]      # zero radius
    lines += ["sphere 0 0 -50 1e6 %s 0 0 1.2 0 10" % mat]                             # one huge sphere
    lines += ["sphere 3 3 -25 1 %s 0 0 1.2 1 10  2 0 0  0 2 0  0 0 2  0 0 0" % mat] * 3  # affine, coincident
    lines += ["triangle 0 0 -20  0 0 -20  0 0 -20  %s 0 0 1.2 10" % mat] * 100        # degenerate triangles
    with open(p, "w") as f:
        f.write("%d\n" % len(lines))
        f.write(rt.scenegen.HEADER)
        f.write("ambient 0.05 0.05 0.05\n")
        f.write("\n".join(lines) + "\n")
    sc = rt.load_scene(p)
    for which in (0, 1):
        _check_set(sc, which)
    sc.close()
    os.remove(p)


def _check_runs(scene, which):
    """The run spheres the per-ray traversal culls on (rtx_cluster_runs): `order` is a permutation of the clusters with the
    clusters that hold affine spheres first, and every cluster's sphere lies inside the sphere of its run of 8 (an unbounded
    cluster makes its run unbounded)."""
    cl, _sub, _mem, idx = rt.cluster_records(scene, which)
    order, run, nfirst = rt.cluster_runs(scene, which)
    ncl = cl.shape[0]
    assert sorted(order.tolist()) == list(range(ncl))
    ns = scene.view.nspheres
    aff = np.array([bool(scene.view.spheres[i].istransformed) for i in range(ns)], dtype=bool)
    if which != 1:   # (the shadow set holds no affine spheres: nfirst = 0)
        for p, c in enumerate(order):
            members = idx[c][idx[c] >= 0]
            has_aff = bool(len(members)) and bool(np.any((members < ns) & aff[np.minimum(members, max(ns - 1, 0))]))
            assert has_aff == (p < nfirst), (which, p, int(c))
    nruns0 = (nfirst + 7) // 8
    assert run.shape[0] == nruns0 + (ncl - nfirst + 7) // 8
    c64 = cl.astype(np.float64)
    r64 = run.astype(np.float64)
    for p, c in enumerate(order):
        r = p // 8 if p < nfirst else nruns0 + (p - nfirst) // 8
        if not np.isfinite(c64[c, 3]):
            assert r64[r, 3] == np.inf, (which, p)
            continue
        if r64[r, 3] == np.inf:
            continue
        reach = np.sqrt(((c64[c, :3] - r64[r, :3]) ** 2).sum()) + c64[c, 3]
        assert reach <= r64[r, 3], (which, p, float(reach), float(r64[r, 3]))
    return ncl


@pytest.mark.parametrize("path", _scene_paths(), ids=os.path.basename)
def test_run_spheres_of_the_test_scenes(path):
    sc = rt.load_scene(path)
    for which in (0, 1, 2):
        _check_runs(sc, which)
    sc.close()


def test_per_ray_set_of_a_scene_of_several_segments():
    """More than 16,384 primitives: the closest-hit set is cut into segments and the per-ray traversal walks a set of its own
    (which = 2) -- every invariant on that one too, and it is not the segmented set."""
    p = os.path.join(tempfile.gettempdir(), "rtx_clusters_big_%d.txt" % os.getpid())
    rt.scenegen.write_c4(p, nspheres=9000, ntriangles=9000)
    sc = rt.load_scene(p)
    n2, slots2 = _check_set(sc, 2)
    n0 = rt.cluster_records(sc, 0)[0].shape[0]
    assert slots2 == 18000 and n2 <= n0
    _check_runs(sc, 2)
    _check_runs(sc, 0)
    sc.close()
    os.remove(p)
