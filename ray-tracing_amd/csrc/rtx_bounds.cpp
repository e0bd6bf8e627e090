// rtx_bounds.cpp -- conservative bounding spheres for the filter pass (see rtx_pack.h).
//
// The hot kernels do not run the reference's intersection test on every (ray, primitive) pair.
// They first run an 11-instruction FMA line-vs-bounding-sphere test and run the reference's
// exact, contraction-free test only for pairs that pass it.  The images stay identical as long
// as the filter never rejects a pair the exact test would accept; this file computes, in double
// precision, bounding spheres that make that a theorem rather than a hope:
//
//   filter, per ray and primitive (all FMA, binary32):
//       X = fo - c;  bh = fd.X;  F = bh*bh - (X.X - R2)        reject  <=>  F < 0
//   with fd = (1 + 2^-18) * normalised ray direction.
//
//   Rays the analysis does not cover (non-finite or huge origin, direction not unit, a direction
//   component in (0, 1e-10)) bypass the filter in the kernel ("abnormal" lanes).  Primitives it
//   does not cover get R2 = +inf here and always pass.
//   A direction of exactly zero IS covered: the reference's normalize leaves it zero, a plain sphere
//   then accepts iff fl(|X|^2 - r^2) <= 0 (b = 0, disc = -4c, t1 = sqrt(disc)/2 >= 0), an affine
//   sphere without translation row the same in object space (its world-space bound contains the
//   ellipsoid), a triangle never (vd == 0); the filter with fd = 0 computes R2 - |X|^2 with R2 >=
//   r^2 (1 + 2^-14), so every accepted origin passes.
//
// Margins (u = 2^-24):
//   * evaluation error of F in binary32 FMA arithmetic: <= 8u (2|X|^2 + R2); the exact sphere
//     test's own discriminant (raytracing.cpp:205-212) differs from the real value by
//     <= 12u (2|X|^2 + r^2).  Scaling the direction by (1 + 2^-18) raises the |X|^2 coefficient by
//     >= 2^-18 and inflating r^2 by (1 + 2^-16) covers the rest, so the threshold stays 0.
//   * triangles: the exact test accepts a point P (as computed in binary32, :352) only if the
//     three sub-areas sum to the area within tau = 0.01 + delta, delta bounding the binary32
//     evaluation error of the four areas.  Projected on the plane that puts P inside the
//     triangle whose edges are pushed out by tau / edge_length; off the plane |height| <=
//     2 (A + tau) / e_max.  P itself is within eta of the ray's line, eta bounding the rounding
//     of o + t*d for origins inside the guard box.  The sphere around all that, inflated by
//     2^-16 relatively, is what the filter sees.
//   * affine spheres whose inverse matrix has no translation (im.d == 0) map world points
//     linearly, so the object-space sphere is a world-space ellipsoid with a bounding sphere
//     of radius r * ||A^-1||; with a translation row the reference adds it to DIRECTIONS as well
//     (:185-187), the hit set is no longer an ellipsoid, and the sphere always passes.
#include <algorithm>
#include <cmath>
#include <limits>

#include "rtx_pack.h"

namespace rtx {

namespace {

struct D3 { double x, y, z; };
inline D3 d3(const rtx_vec3& v) { return {v.x, v.y, v.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(D3 a) { return std::sqrt(dot(a, a)); }
inline bool finite3(D3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

const float kInf = std::numeric_limits<float>::infinity();

inline BoundSphere always_pass() { return {0.0f, 0.0f, 0.0f, -kInf}; }

// Round a double up to the next binary32 (so the stored radius^2 is never below the bound).
inline float up(double v)
{
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, kInf);
    return f;
}

inline BoundSphere make(D3 c, double R)
{
    if (!finite3(c) || !std::isfinite(R) || R < 0) return always_pass();
    // centre is rounded to binary32: cover the rounding, then the relative inflation and the
    // absolute floor discussed above
    double cr = (std::fabs(c.x) + std::fabs(c.y) + std::fabs(c.z)) * 1.2e-7;
    double Rh = (R + cr) * (1.0 + 1.0 / 65536.0);
    double R2 = Rh * Rh * (1.0 + 1.0 / 65536.0) + 1e-30;
    if (!(R2 < 1e37)) return always_pass();
    return {(float)c.x, (float)c.y, (float)c.z, -up(R2)};
}

// smallest enclosing circle of three points in the plane (2-D coordinates)
void enclosing_circle(const double p[3][2], double& cx, double& cy, double& r)
{
    // try the three diameter circles, else the circumcircle
    double best = -1;
    for (int i = 0; i < 3; i++) {
        int j = (i + 1) % 3, k = (i + 2) % 3;
        double mx = 0.5 * (p[i][0] + p[j][0]), my = 0.5 * (p[i][1] + p[j][1]);
        double rr = 0.5 * std::hypot(p[i][0] - p[j][0], p[i][1] - p[j][1]);
        if (std::hypot(p[k][0] - mx, p[k][1] - my) <= rr * (1 + 1e-12)) {
            if (best < 0 || rr < best) { best = rr; cx = mx; cy = my; }
        }
    }
    if (best >= 0) { r = best; return; }
    double ax = p[0][0], ay = p[0][1], bx = p[1][0], by = p[1][1], qx = p[2][0], qy = p[2][1];
    double dd = 2 * (ax * (by - qy) + bx * (qy - ay) + qx * (ay - by));
    cx = ((ax * ax + ay * ay) * (by - qy) + (bx * bx + by * by) * (qy - ay) + (qx * qx + qy * qy) * (ay - by)) / dd;
    cy = ((ax * ax + ay * ay) * (qx - bx) + (bx * bx + by * by) * (ax - qx) + (qx * qx + qy * qy) * (bx - ax)) / dd;
    r = std::hypot(ax - cx, ay - cy);
}

// ||M||_2 of a 3 x 3 matrix, never below it: the largest eigenvalue of M^T M by cyclic Jacobi rotations (double; converged to
// ~1e-15, returned with 1e-9 on top).  sqrt(||M||_1 ||M||_inf) -- what the bounds used before -- is up to sqrt(3) above it for a
// rotated matrix: with both factors of a condition number that loose, most ROTATED ellipsoids failed "cond <= 8" and became
// records every shadow ray tests (3,000 of them at 4K: 324 exact tests a shadow ray, tools/dbg/perf_survey.py "affine").
double spectral_norm_up(const double M[3][3])
{
    double S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[i][j] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (!std::isfinite(S[i][j])) return kInf;
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = std::fabs(S[0][1]) + std::fabs(S[0][2]) + std::fabs(S[1][2]);
        if (off <= 1e-300 || off <= 1e-17 * (std::fabs(S[0][0]) + std::fabs(S[1][1]) + std::fabs(S[2][2]))) break;
        for (int pp = 0; pp < 2; pp++)
            for (int q = pp + 1; q < 3; q++) {
                if (S[pp][q] == 0.0) continue;
                const double theta = (S[q][q] - S[pp][pp]) / (2.0 * S[pp][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                const int r = 3 - pp - q;
                const double spq = S[pp][q], srp = S[r][pp], srq = S[r][q];
                S[pp][pp] -= t * spq; S[q][q] += t * spq;
                S[pp][q] = S[q][pp] = 0.0;
                S[r][pp] = S[pp][r] = c * srp - sn * srq;
                S[r][q] = S[q][r] = sn * srp + c * srq;
            }
    }
    // (Gershgorin on what is left of the off-diagonal part: an upper bound whether or not the sweeps converged)
    double lam = 0;
    for (int i = 0; i < 3; i++) {
        double g = S[i][i];
        for (int j = 0; j < 3; j++) if (j != i) g += std::fabs(S[i][j]);
        lam = std::max(lam, g);
    }
    return std::sqrt(std::max(lam, 0.0)) * (1.0 + 1e-9);
}

}  // namespace

BoundSphere bound_sphere_plain(const rtx_sphere& s)
{
    D3 c = d3(s.center);
    double r = std::fabs((double)s.radius);
    if (!finite3(c) || !std::isfinite(r)) return always_pass();
    // no centre rounding here: the filter and the exact test subtract the very same binary32 centre
    double R2 = r * r * (1.0 + 1.0 / 16384.0) + 1e-30;
    if (!(R2 < 1e37)) return always_pass();
    return {s.center.x, s.center.y, s.center.z, -up(R2)};
}

BoundSphere bound_sphere_affine(const rtx_sphere& s, const rtx_xmatrix& im, double guard)
{
    // object point p' = p*A + b with A = rows (im.a, im.b, im.c), b = im.d  (raytracing.cpp:185-187)
    if (im.d.x != 0 || im.d.y != 0 || im.d.z != 0) return always_pass();
    double A[3][3] = {{im.a.x, im.a.y, im.a.z}, {im.b.x, im.b.y, im.b.z}, {im.c.x, im.c.y, im.c.z}};
    for (auto& row : A)
        for (double v : row)
            if (!std::isfinite(v)) return always_pass();
    double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                 A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    // ||M||_2 <= sqrt(||M||_1 ||M||_inf): exact for diagonal matrices, never below the spectral norm
    auto norm2_bound = [](const double M[3][3]) {
        double n1 = 0, ninf = 0;
        for (int i = 0; i < 3; i++) {
            double r = std::fabs(M[i][0]) + std::fabs(M[i][1]) + std::fabs(M[i][2]);
            double c = std::fabs(M[0][i]) + std::fabs(M[1][i]) + std::fabs(M[2][i]);
            ninf = std::max(ninf, r);
            n1 = std::max(n1, c);
        }
        return std::sqrt(n1 * ninf);
    };
    double nA = norm2_bound(A);
    if (!(std::fabs(det) > 1e-12 * nA * nA * nA) || nA == 0) return always_pass();
    double B[3][3];  // B = A^-1 (row-vector convention: p = p' * B)
    B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
    B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
    B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
    B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
    B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
    B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
    B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    double nB = std::min(norm2_bound(B), spectral_norm_up(B));  // the bounding sphere's radius is r ||A^-1||_2
    if (!std::isfinite(nB) || nA * nB > 1e3) return always_pass();  // ill-conditioned: not worth a bound
    D3 c = d3(s.center);
    double r = std::fabs((double)s.radius);
    if (!finite3(c) || !std::isfinite(r)) return always_pass();
    D3 cw = {c.x * B[0][0] + c.y * B[1][0] + c.z * B[2][0], c.x * B[0][1] + c.y * B[1][1] + c.z * B[2][1],
             c.x * B[0][2] + c.y * B[1][2] + c.z * B[2][2]};
    // object-space slack: binary32 evaluation of o*A and d*A for origins inside the guard box
    // (<= 4u * guard * ||A|| each), of the normalisation and of the discriminant (relative 2^-18 of
    // the object-space distances, themselves <= guard*||A|| + |c| + r)
    double obj = guard * nA + norm(c) + r;
    double slack = obj * (1.0 / 131072.0);
    return make(cw, (r * (1.0 + 1.0 / 16384.0) + slack) * nB);
}

// Shadow rays test an affine sphere with the ORIGIN left in world space and only the direction
// mapped, td = normalize(ld*A + b) (raytracing.cpp:571-576).  The kernel's filter for that call
// pattern recomputes q = ld*A with FMAs and tests the line (o, q) against the sphere (c, R):
//     (1 + 2^-13) (q.X)^2 >= |q|^2 (|X|^2 - R^2)
// Covered only when b == 0 and cond(A) <= 8, which bounds the angle between the binary32
// direction the reference normalises and the filter's q by 2^-17; the 2^-13 factor and the
// 2^-12 inflation of r^2 absorb that and the evaluation error.  Anything else always passes.
AffineShadowRec affine_shadow_record(const rtx_sphere& s, const rtx_xmatrix& im, int index)
{
    AffineShadowRec r = {im.a.x, im.a.y, im.a.z, im.b.x, im.b.y, im.b.z, im.c.x, im.c.y, im.c.z,
                         s.center.x, s.center.y, s.center.z, -kInf, index, kInf, kInf};
    if (im.d.x != 0 || im.d.y != 0 || im.d.z != 0) return r;
    double A[3][3] = {{im.a.x, im.a.y, im.a.z}, {im.b.x, im.b.y, im.b.z}, {im.c.x, im.c.y, im.c.z}};
    double n1 = 0, ninf = 0;
    for (int i = 0; i < 3; i++) {
        if (!std::isfinite(A[i][0]) || !std::isfinite(A[i][1]) || !std::isfinite(A[i][2])) return r;
        ninf = std::max(ninf, std::fabs(A[i][0]) + std::fabs(A[i][1]) + std::fabs(A[i][2]));
        n1 = std::max(n1, std::fabs(A[0][i]) + std::fabs(A[1][i]) + std::fabs(A[2][i]));
    }
    double nA = std::sqrt(n1 * ninf);
    double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                 A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (!(std::fabs(det) > 1e-9 * nA * nA * nA) || nA == 0) return r;
    // ||A^-1||_2 <= ||A||_F^2 / |det| for 3x3 (adjugate bound); use the cheap Frobenius form
    double fro2 = 0;
    for (auto& row : A)
        for (double v : row) fro2 += v * v;
    double nB = fro2 / std::fabs(det);  // >= ||adj(A)||_2 / |det| >= ||A^-1||_2
    if (!(nA * nB <= 8.0)) {
        // retry with the exact inverse norm bound before giving up
        double B[3][3];
        B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
        B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det; B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
        B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
        B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
        B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
        double b1 = 0, binf = 0;
        for (int i = 0; i < 3; i++) {
            binf = std::max(binf, std::fabs(B[i][0]) + std::fabs(B[i][1]) + std::fabs(B[i][2]));
            b1 = std::max(b1, std::fabs(B[0][i]) + std::fabs(B[1][i]) + std::fabs(B[2][i]));
        }
        // nA must bound the matrix of ABSOLUTE values (the rounding of ld*A is entry by entry); A^-1 enters as 1 / sigma_min(A) alone
        double absA[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) absA[i][j] = std::fabs(A[i][j]);
        nA = std::min(nA, spectral_norm_up(absA));
        nB = std::min(std::sqrt(b1 * binf), spectral_norm_up(B));
        if (!(nA * nB <= 8.0)) return r;
    }
    double rad = std::fabs((double)s.radius);
    if (!std::isfinite(rad) || !finite3(d3(s.center))) return r;
    double R2 = rad * rad * (1.0 + 1.0 / 4096.0) + 1e-30;
    if (!(R2 < 1e37)) return r;
    r.neg_r2 = -up(R2);
    r.r_up = up(std::sqrt(R2) * 1.00001);          // what the packet test of the shadow kernel would compute per wave
    r.fro_up = up(std::sqrt(fro2) * 1.00001);      // >= ||A||_2
    return r;
}

BoundSphere bound_sphere_triangle(const rtx_triangle& t, double guard)
{
    D3 u = d3(t.u), v = d3(t.v), w = d3(t.w);
    if (!finite3(u) || !finite3(v) || !finite3(w)) return always_pass();
    D3 e1 = v - u, e2 = w - u;
    D3 n = cross(e1, e2);
    double nn = norm(n);
    double A = 0.5 * nn;
    double l[3] = {norm(w - v), norm(u - w), norm(v - u)};  // edge opposite u, v, w
    double emax = std::max(l[0], std::max(l[1], l[2])), emin = std::min(l[0], std::min(l[1], l[2]));
    if (!(emin > 1e-6 * emax) || !(A > 1e-7 * emax * emax) || !(emax < 1e8)) return always_pass();
    // the binary32 unit normal the reference uses must be a unit vector with no tiny non-zero
    // component (see the vd analysis in the kernel's abnormal-ray guard)
    D3 nh = n * (1.0 / nn);
    for (double comp : {nh.x, nh.y, nh.z})
        if (comp != 0 && std::fabs(comp) < 1e-10) return always_pass();
    // plane frame
    D3 ex = e1 * (1.0 / norm(e1));
    D3 ey = cross(nh, ex);
    double P[3][2] = {{0, 0}, {dot(e1, ex), dot(e1, ey)}, {dot(e2, ex), dot(e2, ey)}};
    double Rg = emax;  // first guess of the accept region's radius
    double R = 0, cx = 0, cy = 0, hmax = 0;
    for (int it = 0; it < 2; it++) {
        // binary32 error of the reference's areas for points within Rg of the triangle:
        // each cross-product component errs by <= 2u * 2 * |vec||edge|, each area by a few of those
        double delta = (Rg + emax) * emax * (64.0 / 16777216.0) + A * (16.0 / 16777216.0);
        double tau = (0.01 + delta) * 1.001;
        // push every edge line outward by tau / its length and intersect neighbours
        double Q[3][2];
        bool ok = true;
        // edge i joins P[(i+1)%3] and P[(i+2)%3]; outward normal points away from P[i]
        double ln[3][3];  // a*x + b*y = c, unit (a,b)
        for (int i = 0; i < 3; i++) {
            const double* a = P[(i + 1) % 3];
            const double* b = P[(i + 2) % 3];
            double dx = b[0] - a[0], dy = b[1] - a[1];
            double len = std::hypot(dx, dy);
            double nx = dy / len, ny = -dx / len;
            double side = nx * (P[i][0] - a[0]) + ny * (P[i][1] - a[1]);
            if (side > 0) { nx = -nx; ny = -ny; }
            ln[i][0] = nx; ln[i][1] = ny; ln[i][2] = nx * a[0] + ny * a[1] + tau / len;
        }
        for (int i = 0; i < 3; i++) {  // vertex i of the grown triangle = lines (i+1) x (i+2)
            const double* L1 = ln[(i + 1) % 3];
            const double* L2 = ln[(i + 2) % 3];
            double dd = L1[0] * L2[1] - L1[1] * L2[0];
            if (!(std::fabs(dd) > 1e-12)) { ok = false; break; }
            Q[i][0] = (L1[2] * L2[1] - L1[1] * L2[2]) / dd;
            Q[i][1] = (L1[0] * L2[2] - L1[2] * L2[0]) / dd;
        }
        if (!ok) return always_pass();
        enclosing_circle(Q, cx, cy, R);
        hmax = 2.0 * (A + tau) / emax;
        if (!std::isfinite(R) || !std::isfinite(hmax)) return always_pass();
        Rg = std::sqrt(R * R + hmax * hmax) + emax;
    }
    D3 c = u + ex * cx + ey * cy;
    double Rs = std::sqrt(R * R + hmax * hmax);
    // P as computed in binary32 (o + t*d, :352) lies within eta of the ray's line
    double eta = (guard + norm(c) + Rs) * (8.0 / 16777216.0);
    return make(c, Rs + eta);
}

// Edge filter.  The exact test (raytracing.cpp:338-365) computes P = o + dir*t and vec1 = P - v in binary32
// and accepts only if the three sub-areas exceed the area by <= tau = 0.01 + delta in real arithmetic (delta:
// the binary32 error of the four areas, as above, here for every P with |P - v| <= Q).  Each sub-area is at
// least its projection on the triangle's plane, A |b_i| with b_i the barycentric coordinates of P's
// projection, and sum b_i = 1, so the excess is >= 2 A sum_{b_i < 0} |b_i|: an accepted P has every
// b_i >= -tau / (2 A).  The kernel evaluates b_1 = a1 . vec1 (weight of u), b_2 = a2 . vec1 (weight of w)
// with the in-plane dual vectors a1, a2 of the edges u - v, w - v and b_0 = (1 - b_1) - b_2, from the SAME
// binary32 vec1 the exact test goes on to use, and rejects when |vec1|_1 <= Q and some b_i < -m (-2m for b_0).
// Its own error: a_i rounded to binary32 (u |a|_inf |p|_1), vec1 rounded (the same), the dot product
// (3u |a|_inf |p|_1): < 8u amax Q for |p|_1 <= Q; b_0 adds both plus 4u (1 + |b_1| + |b_2|).  With
// m = tau / (2A) + 32u amax Q + 1e-6 all of that is covered.  Anything not covered (degenerate or huge
// triangles, non-finite numbers) gets Q = -1 and is never rejected; NaN compares false and is never rejected.
TriEdge tri_edge_filter(const rtx_triangle& t)
{
    const TriEdge never = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, -1.0f};
    D3 u = d3(t.u), v = d3(t.v), w = d3(t.w);
    if (!finite3(u) || !finite3(v) || !finite3(w)) return never;
    D3 e1 = u - v, e2 = w - v;
    D3 n = cross(e1, e2);
    double nn2 = dot(n, n);
    double A = 0.5 * std::sqrt(nn2);
    double l[3] = {norm(w - v), norm(u - w), norm(v - u)};
    double emax = std::max(l[0], std::max(l[1], l[2])), emin = std::min(l[0], std::min(l[1], l[2]));
    if (!(emin > 1e-6 * emax) || !(A > 1e-7 * emax * emax) || !(emax < 1e8) || !(emax > 1e-8)) return never;
    const double uu = 1.0 / 16777216.0;
    double Q = 32.0 * emax;
    double delta = (Q + emax) * emax * (64.0 * uu) + A * (16.0 * uu);
    double tau = (0.01 + delta) * 1.001;
    D3 a1 = cross(e2, n) * (1.0 / nn2);
    D3 a2 = cross(n, e1) * (1.0 / nn2);
    double amax = 0;
    for (double c : {a1.x, a1.y, a1.z, a2.x, a2.y, a2.z}) amax = std::max(amax, std::fabs(c));
    double m = tau / (2.0 * A) * (1.0 + 1e-6) + 32.0 * uu * amax * Q + 1e-6;
    if (!std::isfinite(m) || !(m < 1e30) || !(amax < 1e30)) return never;
    float Qf = (float)(Q * (1.0 - 1e-6));
    if ((double)Qf > Q) return never;
    return {(float)a1.x, (float)a1.y, (float)a1.z, up(m), (float)a2.x, (float)a2.y, (float)a2.z, Qf};
}

}  // namespace rtx
