"""Synthetic scene files in the reference's HEAD grammar (parseinput, raytracing.cpp:851-979;
SURVEY.md Appendix A), so that the kept parser, the oracle and the GPU path all read the same
bytes.  Seeded splitmix64 (never a library distribution), floats printed ``%.9g`` so they
round-trip exactly through ``operator>>`` into ``float``.

Workloads:
  C4  3840x2160, 5,000 spheres (10% affine) + 5,000 triangles, 4 lights, MAX_depth 4
  C5  7680x4320, 100,000 triangles, 16 lights, MAX_depth 7
"""
import numpy as np

SEED = 0x5EED2026
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


class SplitMix64:
    """Vectorised splitmix64: the i-th output depends only on (seed, i)."""

    def __init__(self, seed=SEED):
        self.seed = np.uint64(seed)
        self.pos = 0

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(self.pos + 1, self.pos + n + 1, dtype=np.uint64)
            z = self.seed + idx * _GOLDEN
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        self.pos += n
        return z

    def uniform(self, lo, hi, shape):
        n = int(np.prod(shape))
        u = (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
        return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)

    def choice(self, values, n):
        v = np.asarray(values, dtype=np.float32)
        return v[(self.u64(n) % np.uint64(len(v))).astype(np.int64)]


def _f(x):
    return "%.9g" % float(np.float32(x))


def _fs(xs):
    return " ".join(_f(x) for x in np.ravel(xs))


HEADER = ("camera 0 0 -1\n" "vrp 0 0 0\n" "vpn 0 0 -1\n" "up 0 1 0\n")  # bundled camera (inp.txt:2-5)


def write_scene(path, nspheres, ntriangles, lights, seed=SEED, zfar=-120.0, frac_affine=0.10,
                frac_reflective=0.30, frac_refractive=0.0, ambient=(0.05, 0.05, 0.05),
                radius=(0.3, 1.2), edge=2.0, box=(40.0, 25.0), znear=-20.0, affine="diag"):
    """Write one scene; returns the number of records.  `lights` is a list of (x,y,z,r,g,b).
    affine = "diag": scale matrices (the benchmark scenes); "general": rotation x scale matrices, one in 16
    with a translation row and one in 32 singular (parity stress scenes; drawn from a second generator so
    the other records do not change); "rotated": rotation x scale matrices, none of those."""
    rng = SplitMix64(seed)
    rng2 = SplitMix64(seed ^ 0x5DEECE66D)
    lines = []
    for L in lights:
        lines.append("light " + _fs(L))
    bx, by = box
    if nspheres:
        c = np.stack([rng.uniform(-bx, bx, nspheres), rng.uniform(-by, by, nspheres),
                      rng.uniform(zfar, znear, nspheres)], axis=1)
        r = rng.uniform(radius[0], radius[1], nspheres)
        col = rng.uniform(0, 1, (nspheres, 3))
        dif = rng.uniform(0, 1, (nspheres, 3))
        spe = rng.uniform(0, 1, (nspheres, 3))
        expo = rng.choice([5, 10, 20, 50], nspheres)
        refl = np.where(rng.uniform(0, 1, nspheres) < frac_reflective, 0.5, 0.0)
        refr = np.where(rng.uniform(0, 1, nspheres) < frac_refractive, 0.8, 0.0)
        xf = rng.uniform(0, 1, nspheres) < frac_affine
        sc = rng.uniform(0.5, 2.0, (nspheres, 3))
        for i in range(nspheres):
            s = ("sphere %s %s %s %s %s %s %s 1.2 %d %s" %
                 (_fs(c[i]), _f(r[i]), _fs(col[i]), _fs(dif[i]), _fs(spe[i]), _f(refl[i]), _f(refr[i]),
                  1 if xf[i] else 0, _f(expo[i])))
            if xf[i] and affine in ("general", "rotated"):
                q = rng2.uniform(-1, 1, 4)
                q = q / max(float(np.linalg.norm(q)), 1e-6)
                a, b, cc, d = (float(v) for v in q)
                R = np.array([[a*a+b*b-cc*cc-d*d, 2*(b*cc-a*d), 2*(b*d+a*cc)],
                              [2*(b*cc+a*d), a*a-b*b+cc*cc-d*d, 2*(cc*d-a*b)],
                              [2*(b*d-a*cc), 2*(cc*d+a*b), a*a-b*b-cc*cc+d*d]])
                M = R * np.asarray(sc[i], dtype=np.float64)[None, :]
                kind = float(rng2.uniform(0, 1, 1)[0])
                t = np.zeros(3)
                if affine == "rotated":
                    pass
                elif kind < 1.0 / 16:
                    t = np.asarray(rng2.uniform(-0.5, 0.5, 3), dtype=np.float64)
                elif kind < 1.0 / 16 + 1.0 / 32:
                    M[2] = M[0] * 2.0  # singular
                s += " " + " ".join(_f(v) for v in list(M.reshape(-1)) + list(t))
            elif xf[i]:
                s += " %s 0 0 0 %s 0 0 0 %s 0 0 0" % (_f(sc[i, 0]), _f(sc[i, 1]), _f(sc[i, 2]))
            lines.append(s)
    if ntriangles:
        g = np.stack([rng.uniform(-bx, bx, ntriangles), rng.uniform(-by, by, ntriangles),
                      rng.uniform(zfar, znear, ntriangles)], axis=1)
        e1 = rng.uniform(-edge, edge, (ntriangles, 3))
        e2 = rng.uniform(-edge, edge, (ntriangles, 3))
        # centroid g, edges e1 = v - u, e2 = w - u
        u = g - (e1 + e2) / np.float32(3.0)
        v = u + e1
        w = u + e2
        col = rng.uniform(0, 1, (ntriangles, 3))
        dif = rng.uniform(0, 1, (ntriangles, 3))
        spe = rng.uniform(0, 1, (ntriangles, 3))
        expo = rng.choice([5, 10, 20, 50], ntriangles)
        refl = np.where(rng.uniform(0, 1, ntriangles) < frac_reflective, 0.5, 0.0)
        refr = np.where(rng.uniform(0, 1, ntriangles) < frac_refractive, 0.8, 0.0)
        for i in range(ntriangles):
            lines.append("triangle %s %s %s %s %s %s %s %s 1.2 %s" %
                         (_fs(u[i]), _fs(v[i]), _fs(w[i]), _fs(col[i]), _fs(dif[i]), _fs(spe[i]),
                          _f(refl[i]), _f(refr[i]), _f(expo[i])))
    with open(path, "w") as f:
        f.write("%d\n" % len(lines))
        f.write(HEADER)
        f.write("ambient %s\n" % _fs(ambient))
        f.write("\n".join(lines))
        f.write("\n")
    return len(lines)


def lights_c4():
    return [(x, 40.0, z, 0.25, 0.25, 0.25) for z in (10.0, -60.0) for x in (-30.0, 30.0)]


def lights_c5():
    out = []
    for iz in range(4):
        for ix in range(4):
            out.append((-45.0 + 30.0 * ix, 40.0, 10.0 - 100.0 * iz, 1.0 / 16, 1.0 / 16, 1.0 / 16))
    return out


def write_c4(path, nspheres=5000, ntriangles=5000, seed=SEED, frac_refractive=0.0):
    """BASELINE.json configs[3]: 4K, 10k spheres+triangles, 4 lights, depth 5 (MAX_depth 4)."""
    return write_scene(path, nspheres, ntriangles, lights_c4(), seed=seed, zfar=-120.0,
                       frac_refractive=frac_refractive)


def write_c5(path, ntriangles=100000, seed=SEED):
    """BASELINE.json configs[4]: 8K, 100k triangles, 16 lights, depth 8 (MAX_depth 7)."""
    return write_scene(path, 0, ntriangles, lights_c5(), seed=seed, zfar=-300.0)


WORKLOADS = {
    "c4": dict(width=3840, height=2160, max_depth=4, writer=write_c4,
               name="synthetic 4K, 5000 spheres (10% affine) + 5000 triangles, 4 lights, depth 5"),
    # SURVEY.md 8d's second run: the same scene with 5 % of the primitives refractive (both kinds of secondary ray)
    "c4r": dict(width=3840, height=2160, max_depth=4, writer=lambda path: write_c4(path, frac_refractive=0.05),
                name="synthetic 4K, 5000 spheres (10% affine) + 5000 triangles, 5% refractive, 4 lights, depth 5"),
    "c5": dict(width=7680, height=4320, max_depth=7, writer=write_c5,
               name="synthetic 8K, 100000 triangles, 16 lights, depth 8"),
}
