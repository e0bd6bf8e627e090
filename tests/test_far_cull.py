"""The far cull of the per-ray traversal (rtx_closest.hip.h: cone_slack_cull) against the reference's own arithmetic,
on the CPU: k_closest_lanes skips a sphere { c, R } of its hierarchy -- and every primitive inside -- when the sphere lies wholly
beyond lr = 1.015625 L + 1e-5 guard, L being the largest t at which a hit could still change the ray's state.  So no hit the
reference's tests accept at t <= L may lie inside a skipped sphere.  Checked with the kernel's own functions compiled for the host
(rtx_cone_tests) on hits computed in binary32 exactly as raytracing.cpp:205-224 computes them (plain spheres: near and far roots,
grazing rays, origins inside, tiny and huge radii), and on points of a primitive's bounding sphere (triangles, whose accepted
point P = o + t d lies in it: rtx_bounds.cpp) -- each against the primitive's own sphere and against enclosing ones."""
import numpy as np
import pytest

import rtxamd as rt

F = np.float32
GUARD = F(1600.0)            # 16 x an extent of 100 (rtx_pack.cpp)
CULL_ABS = F(1e-5) * GUARD   # closest_pool: cull_abs


def _normalize(v):
    """raytracing.cpp:116-125 in binary32."""
    v = v.astype(F)
    n2 = F(v[0] * v[0]) + F(v[1] * v[1])
    n2 = F(n2 + F(v[2] * v[2]))
    ln = F(np.sqrt(n2))
    return np.array([v[0] / ln, v[1] / ln, v[2] / ln], dtype=F)


def _sphere_test(o, d, c, r2):
    """raytracing.cpp:205-224 in binary32, operation by operation (rtx_math.hip.h: sphere_test)."""
    X, Y, Z = F(o[0] - c[0]), F(o[1] - c[1]), F(o[2] - c[2])
    b = F(F(2.0) * F(F(F(d[0] * X) + F(d[1] * Y)) + F(d[2] * Z)))
    cc = F(F(F(F(X * X) + F(Y * Y)) + F(Z * Z)) - r2)
    disc = F(F(b * b) - F(F(4.0) * cc))
    if disc < 0:
        return None
    sq = F(np.sqrt(disc))
    t0 = F(F(-b - sq) / F(2.0))
    t1 = F(F(-b + sq) / F(2.0))
    if t0 < 0 and t1 < 0:
        return None
    if t0 < 0:
        t0 = t1
    if t1 > 0:
        t0 = t1 if t1 < t0 else t0
    return t0


def _far_limit(t):
    return F(np.float64(t) * 1.015625 + np.float64(CULL_ABS))   # __builtin_fmaf(L, 1.015625f, cull_abs)


def _enclosing(rng, c, R):
    """The sphere itself, or a sphere around it (centre moved, radius grown by at least the move)."""
    kind = rng.integers(0, 3)
    if kind == 0:
        return np.array([c[0], c[1], c[2], R], dtype=F)
    delta = rng.normal(size=3) * R * rng.choice([0.1, 1.0, 5.0])
    ce = (c.astype(np.float64) + delta).astype(F)
    Re = np.linalg.norm(ce.astype(np.float64) - c.astype(np.float64)) + np.float64(R)
    Re = F(Re * (1.0 + 1e-6) + rng.choice([0.0, 0.5, 10.0]))
    Re = np.nextafter(Re, F(np.inf))
    return np.array([ce[0], ce[1], ce[2], Re], dtype=F)


def test_no_accepted_sphere_hit_lies_in_a_culled_sphere():
    rng = np.random.default_rng(11)
    checked = hits = 0
    for _ in range(30000):
        c = rng.uniform(-90, 90, 3).astype(F)
        r = F(10.0 ** rng.uniform(-2, 1.3))
        mode = rng.integers(0, 4)
        if mode == 0:   # origin close to the surface, either side
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            o = (c + u * r * rng.choice([0.5, 0.999, 1.001, 1.1, 2.0])).astype(F)
        else:
            o = rng.uniform(-100, 100, 3).astype(F)
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        target = c.astype(np.float64) + u * np.float64(r) * rng.choice([0.0, 0.5, 0.99, 0.9999, 1.0, 1.0001, 1.02])
        v = (target - o.astype(np.float64)).astype(F)
        if not np.any(v):
            continue
        nd = _normalize(v)
        t = _sphere_test(o, nd, c, F(r * r))
        if t is None:
            continue
        hits += 1
        Rb = F(np.sqrt(np.float64(r) * np.float64(r) * (1.0 + 1.0 / 16384.0)) * (1.0 + 1e-6))   # the filter record's radius (rtx_bounds.cpp)
        sph = _enclosing(rng, c, Rb)
        slack, front, cull = rt.cone_tests(o, nd, sph, _far_limit(t))
        # the hit is inside `sph`: the cone test and the front half-space test must pass, and at L = t the far cull must not skip it
        assert not np.signbit(slack), (o, nd, sph, t)
        assert not np.signbit(front), (o, nd, sph, t)
        assert not np.signbit(cull), (o, nd, sph, t, float(_far_limit(t)))
        # ... nor at any larger L
        assert not np.signbit(rt.cone_tests(o, nd, sph, _far_limit(F(t * F(1.5) + F(1.0))))[2])
        checked += 1
    assert hits > 15000 and checked == hits


def test_no_point_of_a_bounding_sphere_is_culled_at_its_own_distance():
    """Triangles (and anything else whose accepted point P = o + t d, as computed, lies in its bounding sphere)."""
    rng = np.random.default_rng(12)
    checked = 0
    for _ in range(20000):
        c = rng.uniform(-90, 90, 3).astype(F)
        R = F(10.0 ** rng.uniform(-2, 1.5))
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        P = c.astype(np.float64) + u * np.float64(R) * rng.choice([0.0, 0.5, 0.99, 0.999999])
        o = rng.uniform(-100, 100, 3).astype(F)
        v = (P - o.astype(np.float64)).astype(F)
        if not np.any(v):
            continue
        d = _normalize(v)
        t = F(np.linalg.norm(P - o.astype(np.float64)))
        Pc = np.array([F(o[k] + F(d[k] * t)) for k in range(3)], dtype=F)   # raytracing.cpp:352
        if np.linalg.norm(Pc.astype(np.float64) - c.astype(np.float64)) > np.float64(R):
            continue   # (rounded out of the ball: not a point the bound speaks about)
        sph = _enclosing(rng, c, R)
        slack, front, cull = rt.cone_tests(o, d, sph, _far_limit(t))
        assert not np.signbit(front), (o, d, sph, t)
        assert not np.signbit(cull), (o, d, sph, t, float(_far_limit(t)))
        checked += 1
    assert checked > 15000


def test_the_cull_does_cull():
    """A sphere well beyond the limit is skipped, one that always passes (R = +inf) never, and NaN never."""
    o = np.zeros(3, dtype=F); n = np.array([0, 0, -1], dtype=F)
    far = np.array([0, 0, -100, 1], dtype=F)
    assert np.signbit(rt.cone_tests(o, n, far, _far_limit(F(50.0)))[2])
    assert not np.signbit(rt.cone_tests(o, n, far, _far_limit(F(99.5)))[2])
    assert not np.signbit(rt.cone_tests(o, n, np.array([0, 0, -100, np.inf], dtype=F), _far_limit(F(1.0)))[2])
    assert not np.signbit(rt.cone_tests(o, n, far, F(np.nan))[2])


def test_the_margins_are_needed():
    """The plain statement "the sphere starts beyond L" (n.(c - o) - R > L) is NOT safe: for a small sphere far away the reference's
    discriminant cancels (b*b - 4c with |b| ~ 2 |o - c|) and its near root comes out up to ~1e-3 |o - c| in front of the true one
    -- further in front of the centre's projection than the sphere's radius.  Emulated by handing the kernel's test the lr at which
    its own margins cancel (lr = 0.9921875 t - 0.0234375 R  <=>  n.(c - o) - R > t): some accepted hits would be skipped, which is
    what the factors 1 - 1/128, 1 + 1/64 and the absolute term are there for."""
    rng = np.random.default_rng(3)
    skipped = hits = 0
    for _ in range(4000):
        c = rng.uniform(-90, 90, 3).astype(F)
        r = F(10.0 ** rng.uniform(-2, -1))
        o = rng.uniform(-100, 100, 3).astype(F)
        u = rng.normal(size=3); u /= np.linalg.norm(u)
        v = (c.astype(np.float64) + u * np.float64(r) * 0.9999 - o.astype(np.float64)).astype(F)
        nd = _normalize(v)
        t = _sphere_test(o, nd, c, F(r * r))
        if t is None:
            continue
        hits += 1
        sph = np.array([c[0], c[1], c[2], r], dtype=F)
        skipped += bool(np.signbit(rt.cone_tests(o, nd, sph, F(np.float64(t) * 0.9921875 - 0.0234375 * np.float64(r)))[2]))
        assert not np.signbit(rt.cone_tests(o, nd, sph, _far_limit(t))[2])
    assert hits > 1000 and skipped > 10, (hits, skipped)
