"""The bounding-sphere filter of the intersection kernels (rtx_bounds.cpp + filt() in
rtx_device.hip) must never reject a (ray, primitive) pair the reference's own test accepts.
The kernel's binary32 FMA arithmetic is emulated here with numpy (float64 product + sum rounded
to float32) on the records the library exports, and compared with the oracle's exact tests on
rays aimed at, around and away from every primitive."""
import glob
import os

import numpy as np
import pytest

from conftest import SCENES

f32 = np.float32


def fma(a, b, c):
    with np.errstate(all="ignore"):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def filter_pass(rec, o, d):
    with np.errstate(all="ignore"):
        n = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(f32)).astype(f32)
        nd = np.where(n[:, None] > 0, (d / n[:, None]).astype(f32), d)
        fd = (nd * f32(1.0 + 1.0 / 262144.0)).astype(f32)
        P = np.ones((o.shape[0], rec.shape[0]), dtype=bool)
        for k in range(rec.shape[0]):
            X = (o[:, 0] - rec[k, 0]).astype(f32)
            Y = (o[:, 1] - rec[k, 1]).astype(f32)
            Z = (o[:, 2] - rec[k, 2]).astype(f32)
            bh = (fd[:, 2] * Z).astype(f32)
            bh = fma(fd[:, 1], Y, bh)
            bh = fma(fd[:, 0], X, bh)
            cc = fma(Z, Z, np.full_like(Z, rec[k, 3]))
            cc = fma(Y, Y, cc)
            cc = fma(X, X, cc)
            P[:, k] = ~(fma(bh, bh, -cc) < 0)
    return P


def aimed_rays(sc, rng, n_per_prim):
    """Rays through points on / near every primitive, from origins near and far."""
    v = sc.view
    targets = []
    for i in range(v.nspheres):
        s = v.spheres[i]
        c = np.array([s.center.x, s.center.y, s.center.z])
        u = rng.normal(size=(n_per_prim, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        rad = abs(s.radius) * rng.choice([0.0, 0.5, 0.999, 1.0, 1.001, 1.5, 3.0], size=(n_per_prim, 1))
        targets.append(c + u * rad)
    for i in range(v.ntriangles):
        t = v.triangles[i]
        A, B, Cc = (np.array([p.x, p.y, p.z]) for p in (t.u, t.v, t.w))
        w = rng.uniform(-0.3, 1.3, size=(n_per_prim, 2))
        w[: n_per_prim // 4] = rng.choice([0.0, 1.0, 0.5], size=(n_per_prim // 4, 2))   # vertices, edges
        targets.append(A + w[:, :1] * (B - A) + w[:, 1:] * (Cc - A))
    T = np.concatenate(targets) if targets else np.zeros((0, 3))
    o = rng.uniform(-60, 60, size=T.shape)
    o[::3] = T[::3] + rng.normal(size=T[::3].shape) * 0.5        # origins very close to the target
    o[1::7] = [0.0, 0.0, 1.0]                                    # the bundled camera
    d = T - o
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    d[::2] = -d[::2] if False else d[::2]
    # half of the rays also point away (t < 0 side): the line still meets the primitive
    d[5::11] = -d[5::11]
    return o.astype(f32), d.astype(f32)


SCENE_FILES = sorted(glob.glob(os.path.join(SCENES, "*.txt")))


@pytest.mark.parametrize("path", SCENE_FILES, ids=[os.path.basename(p) for p in SCENE_FILES])
def test_filter_never_rejects_an_accepted_pair(rt, oracle, path):
    sc = rt.load_scene(path)
    rec, guard = rt.filter_records(sc)
    assert rec.shape == (sc.nprims, 4) and guard >= 16.0
    rng = np.random.default_rng(7)
    o, d = aimed_rays(sc, rng, 600)
    acc = oracle.accept_matrix(sc, o, d).astype(bool)
    nst = sc.view.nspheres + sc.view.ntriangles
    assert np.isneginf(rec[nst:, 3]).all(), "polygons are not bounded: their records always pass"
    rec = rec[:nst]
    P = filter_pass(rec, o, d)
    assert acc.sum() > 1000, "test rays do not exercise the primitives"
    viol = acc & ~P
    assert not viol.any(), "filter rejected %d accepted pairs, primitives %s" % (viol.sum(), np.unique(np.argwhere(viol)[:, 1]))
    # and it is a filter: far-away primitives are mostly rejected
    finite = np.isfinite(rec[:, 3])
    if finite.any():
        assert P[:, finite].mean() < 0.9


def test_filter_on_synthetic_c4(rt, oracle, tmp_path):
    """The benchmark scene generator at reduced primitive count: 300 spheres (10 % affine) + 300 triangles."""
    p = tmp_path / "c4s.txt"
    rt.scenegen.write_c4(str(p), nspheres=300, ntriangles=300)
    sc = rt.load_scene(str(p))
    rec, guard = rt.filter_records(sc)
    rng = np.random.default_rng(11)
    o, d = aimed_rays(sc, rng, 24)
    acc = oracle.accept_matrix(sc, o, d).astype(bool)
    P = filter_pass(rec, o, d)
    assert acc.sum() > 5000
    assert not (acc & ~P).any()
    # affine spheres of this generator have no translation row: they get a finite world-space bound
    v = sc.view
    xf = np.array([v.spheres[i].istransformed for i in range(v.nspheres)], dtype=bool)
    assert xf.any() and np.isfinite(rec[: v.nspheres][xf, 3]).all()
    # the filter's selectivity on this scene: a random pair passes rarely
    assert P.mean() < 0.05


def test_zero_direction_is_a_point_query(rt, oracle, tmp_path):
    """Rays whose direction is exactly (0,0,0) (children of hits on spheres with a singular matrix) go
    through the filter with fd = 0: it must keep every primitive the exact test accepts for them."""
    p = tmp_path / "c4s.txt"
    rt.scenegen.write_c4(str(p), nspheres=300, ntriangles=300)
    sc = rt.load_scene(str(p))
    rec, guard = rt.filter_records(sc)
    rng = np.random.default_rng(5)
    v = sc.view
    pts = []
    for i in range(v.nspheres):
        s = v.spheres[i]
        c = np.array([s.center.x, s.center.y, s.center.z])
        u = rng.normal(size=(40, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        pts.append(c + u * abs(s.radius) * rng.choice([0.0, 0.5, 0.99999, 1.0, 1.00001, 1.2], size=(40, 1)))
    for i in range(v.ntriangles):
        t = v.triangles[i]
        pts.append(np.array([[t.u.x, t.u.y, t.u.z], [t.v.x, t.v.y, t.v.z]]))
    o = np.concatenate(pts).astype(f32)
    d = np.zeros_like(o)
    acc = oracle.accept_matrix(sc, o, d).astype(bool)
    P = filter_pass(rec, o, d)
    assert acc.sum() > 3000
    assert not acc[:, v.nspheres:].any(), "a zero direction never hits a triangle (vd == 0)"
    assert not (acc & ~P).any()
    assert P.mean() < 0.05


def test_degenerate_primitives_always_pass(rt, tmp_path):
    p = tmp_path / "deg.txt"
    p.write_text("5\ncamera 0 0 -1\nvrp 0 0 0\nvpn 0 0 -1\nup 0 1 0\nambient 0 0 0\n"
                 "triangle 0 0 -5 0 0 -5 0 0 -5  1 1 1 1 1 1 1 1 1 0 0 1 1\n"         # a point
                 "triangle 0 0 -5 1 0 -5 2 0 -5  1 1 1 1 1 1 1 1 1 0 0 1 1\n"         # collinear
                 "sphere 0 0 -9 1  1 1 1 1 1 1 0 0 0 0 0 1 1 1  1 2 3 2 4 6 0 1 0 0 0 0\n"   # singular matrix
                 "sphere 0 0 -9 1  1 1 1 1 1 1 0 0 0 0 0 1 1 1  1 0 0 0 1 0 0 0 1 .5 0 0\n"  # translation row
                 "sphere 1 2 -9 2  1 1 1 1 1 1 0 0 0 0 0 1 0 1\n")
    sc = rt.load_scene(str(p))
    rec, _ = rt.filter_records(sc)
    # order: spheres first (3), then triangles (2)
    assert np.isneginf(rec[0, 3]) and np.isneginf(rec[1, 3]) and np.isfinite(rec[2, 3])
    assert np.isneginf(rec[3, 3]) and np.isneginf(rec[4, 3])
    assert rec[2, 3] <= -4.0 and rec[2, 3] > -4.01


def affine_shadow_pass(rec, o, d):
    """numpy emulation of the affine-sphere loop of k_shadow_f."""
    P = np.ones((o.shape[0], rec.shape[0]), dtype=bool)
    K = f32(1.0 + 1.0 / 8192.0)
    col = lambda v: np.full(o.shape[0], v, dtype=f32)
    with np.errstate(all="ignore"):
        for k in range(rec.shape[0]):
            m = rec[k]
            qx = fma(d[:, 0], col(m[0]), fma(d[:, 1], col(m[3]), (d[:, 2] * m[6]).astype(f32)))
            qy = fma(d[:, 0], col(m[1]), fma(d[:, 1], col(m[4]), (d[:, 2] * m[7]).astype(f32)))
            qz = fma(d[:, 0], col(m[2]), fma(d[:, 1], col(m[5]), (d[:, 2] * m[8]).astype(f32)))
            X = (o[:, 0] - m[9]).astype(f32); Y = (o[:, 1] - m[10]).astype(f32); Z = (o[:, 2] - m[11]).astype(f32)
            bq = fma(qx, X, fma(qy, Y, (qz * Z).astype(f32)))
            qq = fma(qx, qx, fma(qy, qy, (qz * qz).astype(f32)))
            cc = fma(X, X, fma(Y, Y, fma(Z, Z, col(m[12]))))
            F = fma((bq * K).astype(f32), bq, -(qq * cc).astype(f32))
            P[:, k] = ~(F < 0)
    return P


def test_affine_shadow_filter(rt, oracle, tmp_path):
    """Shadow rays meet affine spheres with only the direction inverse-transformed (rt.cpp:571-576)."""
    p = tmp_path / "aff.txt"
    rt.scenegen.write_c4(str(p), nspheres=200, ntriangles=0)
    for path in (str(p), os.path.join(SCENES, "affine.txt"), os.path.join(SCENES, "quirks.txt")):
        sc = rt.load_scene(path)
        v = sc.view
        rec = rt.affine_shadow_records(sc)
        idx = rec[:, 13].view(np.int32)
        xf = [i for i in range(v.nspheres) if v.spheres[i].istransformed]
        assert list(idx) == xf
        main, _ = rt.filter_records(sc)   # closest-hit records; shadow main-loop records mark these never-pass
        rng = np.random.default_rng(3)
        # shadow-pattern targets: since the origin stays in world space, aim at the sphere's file centre
        n = 4000
        o = rng.uniform(-50, 50, (n, 3)).astype(f32)
        d = np.zeros((n, 3))
        for i in range(n):
            s = v.spheres[xf[i % len(xf)]]
            c = np.array([s.center.x, s.center.y, s.center.z]) + rng.normal(size=3) * abs(s.radius) * rng.choice([0.2, 1.0, 3.0])
            d[i] = c - o[i]
        d += rng.normal(size=d.shape) * 0.05 * np.linalg.norm(d, axis=1, keepdims=True) * (rng.uniform(size=(n, 1)) < 0.5)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
        acc = oracle.accept_matrix(sc, o, d, shadow=True).astype(bool)[:, xf]
        P = affine_shadow_pass(rec, o, d)
        assert acc.sum() > 50
        assert not (acc & ~P).any()


@pytest.mark.parametrize("affine", ["rotated", "general"])
def test_affine_shadow_filter_rotated_matrices(rt, oracle, tmp_path, affine):
    """The same for ROTATED ellipsoids (rotation x scale, scale 0.5 .. 2: condition number <= 4).  The record's "cond <= 8" test
    takes spectral norms (rtx_bounds.cpp: spectral_norm_up) -- with sqrt(||.||_1 ||.||_inf) for both factors most of these were
    records every shadow ray tests -- so every rotated sphere must now be bounded, and the filter must still keep every pair the
    reference accepts: rays are aimed in OBJECT space (td = normalize(ld * A) towards the centre, ld = td * A^-1)."""
    p = tmp_path / "rot.txt"
    rt.scenegen.write_scene(str(p), 300, 0, rt.scenegen.lights_c4(), frac_affine=1.0, affine=affine, seed=321)
    sc = rt.load_scene(str(p))
    v = sc.view
    rec = rt.affine_shadow_records(sc)
    assert rec.shape[0] == v.nspheres == 300
    bounded = np.isfinite(rec[:, 12])
    # (the matrix a record holds is what the reference's inverseMatrix makes of the file's -- not always its inverse, so a few
    # records do exceed the bound; "general": one in 16 has a translation row, one in 32 is singular)
    assert bounded.mean() > (0.9 if affine == "rotated" else 0.8)
    for k in range(300):
        A = rec[k, :9].astype(np.float64).reshape(3, 3)
        sv = np.linalg.svd(A, compute_uv=False)
        if sv[2] > 0 and np.linalg.svd(np.abs(A), compute_uv=False)[0] / sv[2] <= 7.99 and not np.any(rec[k, 9:12] != rec[k, 9:12]):
            xm = v.spheres[k].mat
            if xm.d.x == 0 and xm.d.y == 0 and xm.d.z == 0:
                assert bounded[k], "record %d: || |A| ||_2 / sigma_min = %.3f <= 8 but unbounded" % (k, np.linalg.svd(np.abs(A), compute_uv=False)[0] / sv[2])
    rng = np.random.default_rng(9)
    n = 6000
    o = rng.uniform(-50, 50, (n, 3)).astype(f32)
    d = np.zeros((n, 3))
    for i in range(n):
        k = i % 300
        A = rec[k, :9].astype(np.float64).reshape(3, 3)
        s = v.spheres[k]
        c = np.array([s.center.x, s.center.y, s.center.z]) + rng.normal(size=3) * abs(s.radius) * rng.choice([0.2, 0.9, 1.0, 1.1, 3.0])
        q = c - o[i]
        try:
            ld = q @ np.linalg.inv(A)
        except np.linalg.LinAlgError:
            ld = q
        d[i] = ld / max(np.linalg.norm(ld), 1e-30)
    d = d.astype(f32)
    acc = oracle.accept_matrix(sc, o, d, shadow=True).astype(bool)
    P = affine_shadow_pass(rec, o, d)
    assert acc.sum() > 2000
    assert not (acc & ~P).any(), "filter rejected %d accepted pairs" % int((acc & ~P).sum())
    assert P[:, bounded].mean() < 0.2   # and it is a filter
    # closest-hit records (origin AND direction mapped, rt.cpp:387-398): the world-space bounding sphere takes its radius from
    # ||A^-1||_2 (spectral too); rays aimed at points of the ellipsoid as it lies in the world, p = (c + r u) A^-1
    main, _ = rt.filter_records(sc)
    tg, og = [], []
    for k in range(300):
        A = rec[k, :9].astype(np.float64).reshape(3, 3)
        if abs(np.linalg.det(A)) < 1e-9:
            continue
        s = v.spheres[k]
        u = rng.normal(size=(12, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        pts = (np.array([s.center.x, s.center.y, s.center.z]) + u * abs(s.radius) * rng.choice([0.0, 0.9, 0.999, 1.001, 1.3], size=(12, 1))) @ np.linalg.inv(A)
        tg.append(pts); og.append(pts + rng.normal(size=pts.shape) * rng.choice([0.5, 5.0, 40.0], size=(12, 1)))
    tg = np.concatenate(tg); og = np.concatenate(og)
    ok = np.all(np.abs(og) < 1e3, axis=1) & np.all(np.abs(tg) < 1e3, axis=1)
    og, tg = og[ok], tg[ok]
    dg = tg - og
    dg /= np.maximum(np.linalg.norm(dg, axis=1, keepdims=True), 1e-30)
    acc2 = oracle.accept_matrix(sc, og.astype(f32), dg.astype(f32)).astype(bool)
    P2 = filter_pass(main, og.astype(f32), dg.astype(f32))
    assert acc2.sum() > 1500
    assert not (acc2 & ~P2).any(), "closest-hit filter rejected %d accepted pairs" % int((acc2 & ~P2).sum())


def test_direction_cell_chord_bound():
    """k_affine_cells (rtx_shadow.hip.h) takes every unit direction of a cell of the G x G octahedral map to lie within
    3 sqrt(2) / G of the direction of the cell's centre.  Checked here on a numpy restatement of light_cell and of
    the kernel's decode, on random directions and on directions pushed onto cell borders and folds."""
    rng = np.random.default_rng(7)
    for G in (16, 64, 256):
        d = rng.normal(size=(200000, 3)).astype(np.float32)
        # a share of the samples on the octahedron's folds and on axis planes
        d[:20000, 2] = 0.0
        d[20000:40000, 0] = 0.0
        d[40000:60000, 2] *= 1e-4
        d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
        s = np.abs(d).sum(axis=1)
        px, py = d[:, 0] / s, d[:, 1] / s
        neg = d[:, 2] < 0
        qx = np.where(neg, (1 - np.abs(py)) * np.where(px >= 0, 1.0, -1.0), px)
        qy = np.where(neg, (1 - np.abs(px)) * np.where(py >= 0, 1.0, -1.0), py)
        cx = np.clip(np.floor((qx * 0.5 + 0.5) * G), 0, G - 1)
        cy = np.clip(np.floor((qy * 0.5 + 0.5) * G), 0, G - 1)
        # the kernel's decode of the cell centre
        ux, uy = (cx + 0.5) / G * 2 - 1, (cy + 0.5) / G * 2 - 1
        inner = np.abs(ux) + np.abs(uy) <= 1
        vx = np.where(inner, ux, (1 - np.abs(uy)) * np.where(ux >= 0, 1.0, -1.0))
        vy = np.where(inner, uy, (1 - np.abs(ux)) * np.where(uy >= 0, 1.0, -1.0))
        vz = 1 - np.abs(ux) - np.abs(uy)
        v = np.stack([vx, vy, vz], axis=1)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        chord = np.linalg.norm(d.astype(np.float64) - v, axis=1)
        assert chord.max() <= 4.2427 / G + 1e-5, (G, chord.max() * G)


def edge_filter_reject(sc, erec, o, d):
    """The kernel's edge filter (triangle_test_edge, rtx_math.hip.h) in binary32, one operation per operator:
    reject[ray, triangle].  The plane intersection is the reference's own (raytracing.cpp:330-352)."""
    v = sc.view
    n_t = v.ntriangles
    rej = np.zeros((o.shape[0], n_t), dtype=bool)
    ox, oy, oz = (o[:, k] for k in range(3))
    dx, dy, dz = (d[:, k] for k in range(3))
    with np.errstate(all="ignore"):
        for i in range(n_t):
            t = v.triangles[i]
            U, V, Wv = ([f32(p.x), f32(p.y), f32(p.z)] for p in (t.u, t.v, t.w))
            v1 = [U[k] - V[k] for k in range(3)]
            v2 = [Wv[k] - V[k] for k in range(3)]
            n = [v1[1] * v2[2] - v2[1] * v1[2], v2[0] * v1[2] - v1[0] * v2[2], v1[0] * v2[1] - v2[0] * v1[1]]
            norm = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
            if norm > 0:
                n = [c / norm for c in n]
            dd = -(n[0] * U[0] + n[1] * U[1] + n[2] * U[2])
            vd = n[0] * dx + n[1] * dy + n[2] * dz
            tt = -((n[0] * ox + n[1] * oy + n[2] * oz) + dd) / vd
            ax = (ox + dx * tt) - V[0]
            ay = (oy + dy * tt) - V[1]
            az = (oz + dz * tt) - V[2]
            e = erec[i]
            q = np.abs(ax) + np.abs(ay) + np.abs(az)
            b1 = e[0] * ax + e[1] * ay + e[2] * az
            b2 = e[4] * ax + e[5] * ay + e[6] * az
            b0 = (f32(1.0) - b1) - b2
            nm = -e[3]
            rej[:, i] = (q <= e[7]) & ((b1 < nm) | (b2 < nm) | (b0 < nm + nm))
    return rej


def edge_rays(sc, rng, n_per_tri):
    """Rays through points on, just inside and just outside every triangle's edges and vertices (the band the
    reference's 0.01 area tolerance accepts is 0.01 / edge length wide), head-on and grazing."""
    v = sc.view
    T, N = [], []
    for i in range(v.ntriangles):
        t = v.triangles[i]
        A, B, Cc = (np.array([p.x, p.y, p.z], dtype=np.float64) for p in (t.u, t.v, t.w))
        w = rng.dirichlet([1.0, 1.0, 1.0], size=n_per_tri)
        w[: n_per_tri // 2, rng.integers(0, 3)] = 0.0                     # on an edge line
        w[: n_per_tri // 2] /= np.maximum(w[: n_per_tri // 2].sum(axis=1, keepdims=True), 1e-9)
        P = w[:, :1] * A + w[:, 1:2] * B + w[:, 2:] * Cc
        nn = np.cross(B - A, Cc - A)
        ln = np.linalg.norm(nn)
        nn = nn / ln if ln > 0 else np.array([0.0, 0.0, 1.0])
        c = (A + B + Cc) / 3.0
        out = P - c
        out /= np.maximum(np.linalg.norm(out, axis=1, keepdims=True), 1e-12)
        emax = max(np.linalg.norm(B - A), np.linalg.norm(Cc - A), np.linalg.norm(Cc - B), 1e-6)
        P = P + out * rng.choice([0.0, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 1.0], size=(n_per_tri, 1)) / emax * rng.choice([-1.0, 1.0, 1.0], size=(n_per_tri, 1))
        T.append(P)
        N.append(np.tile(nn, (n_per_tri, 1)))
    T, N = np.concatenate(T), np.concatenate(N)
    side = rng.normal(size=T.shape)
    tilt = rng.choice([0.0, 0.3, 3.0, 30.0, 300.0], size=(T.shape[0], 1))   # up to grazing incidence
    dirn = N * rng.choice([-1.0, 1.0], size=(T.shape[0], 1)) + side * tilt
    dirn /= np.maximum(np.linalg.norm(dirn, axis=1, keepdims=True), 1e-30)
    o = T - dirn * rng.choice([0.01, 1.0, 30.0, 300.0], size=(T.shape[0], 1))
    d = (T - o)
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    o32 = o.astype(f32)
    d32 = d.astype(f32)
    d32 /= np.sqrt((d32 * d32).sum(axis=1, keepdims=True)).astype(f32)
    return o32, d32.astype(f32)


@pytest.mark.parametrize("edge", [0.3, 2.0, 12.0])
def test_triangle_edge_filter_never_rejects_an_accepted_pair(rt, oracle, tmp_path, edge):
    p = str(tmp_path / "tris.txt")
    rt.scenegen.write_scene(p, 4, 120, [(0.0, 40.0, 10.0, 0.3, 0.3, 0.3)], seed=77, edge=edge)
    sc = rt.load_scene(p)
    erec = rt.triangle_edge_records(sc)
    ns, nt = sc.view.nspheres, sc.view.ntriangles
    assert erec.shape == (nt, 8) and (erec[:, 7] > 0).all() and (erec[:, 3] > 0).all()
    rng = np.random.default_rng(11)
    o, d = edge_rays(sc, rng, 300)
    acc = oracle.accept_matrix(sc, o, d).astype(bool)[:, ns:ns + nt]
    rej = edge_filter_reject(sc, erec, o, d)
    assert acc.sum() > 5000, "test rays do not exercise the triangles"
    viol = acc & rej
    assert not viol.any(), "edge filter rejected %d accepted pairs, triangles %s" % (viol.sum(), np.unique(np.argwhere(viol)[:, 1]))
    # and it is a filter: of the rays aimed at a triangle's own neighbourhood that the reference rejects, most are cut
    own = np.zeros_like(acc)
    for i in range(nt):
        own[i * 300:(i + 1) * 300, i] = True
    assert rej[own & ~acc].mean() > 0.25   # (the rest: t < 0, hits inside the certified margin, grazing hits beyond Q)


def test_triangle_edge_filter_degenerate_never_rejects(rt, tmp_path):
    """Zero-area and non-finite triangles get Q = -1 (no point has |P - v|_1 <= -1)."""
    p = str(tmp_path / "deg.txt")
    with open(p, "w") as f:
        f.write("camera 0 0 -1\nvrp 0 0 0\nvpn 0 0 -1\nup 0 1 0\nambient 0.05 0.05 0.05\n")
        mat = " 0.5 0.5 0.5  0.5 0.5 0.5  0.5 0.5 0.5  10  0 0 1.2"
        f.write("triangle 0 0 -5  1 0 -5  2 0 -5" + mat + "\n")          # collinear
        f.write("triangle 0 0 -5  0 0 -5  0 0 -5" + mat + "\n")          # a point
        f.write("triangle 0 0 -5  1e-9 0 -5  0 1 -5" + mat + "\n")       # one edge 1e-9 of the others
        f.write("triangle 0 0 -5  1 0 -5  0 1 -5" + mat + "\n")          # a proper one
    sc = rt.load_scene(p)
    erec = rt.triangle_edge_records(sc)
    assert erec.shape == (sc.view.ntriangles, 8) and np.isfinite(erec).all()
    if sc.view.ntriangles == 4:
        assert (erec[:3, 7] == -1.0).all() and erec[3, 7] > 0
