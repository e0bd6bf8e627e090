// rtx_math.hip.h -- device-side scalar math with the reference's exact IEEE-754 semantics.
//
// Parity contract (BASELINE.json north_star): FP contraction off in both builds.  This file is
// compiled with -ffp-contract=off and without any fast-math flag, so every * + - / sqrt below is
// one correctly rounded binary32 (or, where the reference promotes through a double literal,
// binary64) operation, evaluated in the reference's order.  hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt keeps `/` and sqrtf correctly rounded and the default
// gfx950 float mode keeps denormals, matching x86-64 SSE2.
#ifndef RTX_MATH_HIP_H_
#define RTX_MATH_HIP_H_

#include <hip/hip_runtime.h>

#include "rtx.h"

namespace rtx {

struct F3 { float x, y, z; };

__host__ __device__ __forceinline__ F3 f3(float x, float y, float z) { F3 v; v.x = x; v.y = y; v.z = z; return v; }
__host__ __device__ __forceinline__ F3 f3(const rtx_vec3& a) { return f3(a.x, a.y, a.z); }

// raytracing.cpp:127-130
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// raytracing.cpp:116-125: divide only when the norm is > 0 (NaN and 0 leave v untouched)
__device__ __forceinline__ F3 normalized(F3 v)
{
    float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (n > 0.0f) {
        v.x = v.x / n;
        v.y = v.y / n;
        v.z = v.z / n;
    }
    return v;
}

// raytracing.cpp:185-187 with the inverse `im` precomputed on the host (rtx_pack.cpp).  The
// translation row is added to directions as well as points, as the reference does.
__device__ __forceinline__ F3 inv_apply(F3 v, const rtx_xmatrix& im)
{
    F3 r;
    r.x = v.x * im.a.x + v.y * im.b.x + v.z * im.c.x + im.d.x;
    r.y = v.x * im.a.y + v.y * im.b.y + v.z * im.c.y + im.d.y;
    r.z = v.x * im.a.z + v.y * im.b.z + v.z * im.c.z + im.d.z;
    return r;
}

// raytracing.cpp:189-196 (the "transpose" mode used for normals)
__device__ __forceinline__ F3 inv_apply_normal(F3 v, const rtx_xmatrix& im)
{
    float val = v.x * im.d.x + v.y * im.d.y + v.z * im.d.z + 1.0f;
    if (val == 0.0f) return v;
    F3 r;
    r.x = (v.x * im.a.x + v.y * im.a.y + v.z * im.a.z) / val;
    r.y = (v.x * im.b.x + v.y * im.b.y + v.z * im.b.z) / val;
    r.z = (v.x * im.c.x + v.y * im.c.y + v.z * im.c.z) / val;
    return r;
}

// raytracing.cpp:169-175: the freshly written x feeds y, and x,y feed z
__device__ __forceinline__ F3 fwd_apply(F3 v, const rtx_xmatrix& m)
{
    v.x = v.x * m.a.x + v.y * m.b.x + v.z * m.c.x + m.d.x;
    v.y = v.x * m.a.y + v.y * m.b.y + v.z * m.c.y + m.d.y;
    v.z = v.x * m.a.z + v.y * m.b.z + v.z * m.c.z + m.d.z;
    return v;
}

// raytracing.cpp:202-228.  Returns what intersectsphere returns (true also when the discriminant
// is NaN: every comparison is then false), t = the root the reference leaves in t0.
__device__ __forceinline__ bool sphere_test(F3 o, F3 d, float cx, float cy, float cz, float r2, float& t)
{
    float X = o.x - cx, Y = o.y - cy, Z = o.z - cz;
    float b = 2.0f * (d.x * X + d.y * Y + d.z * Z);
    float c = X * X + Y * Y + Z * Z - r2;
    float disc = b * b - 4.0f * c;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float t0 = (-b - sq) / 2.0f;
    float t1 = (-b + sq) / 2.0f;
    if (t0 < 0.0f && t1 < 0.0f) return false;
    if (t0 < 0.0f) t0 = t1;
    if (t1 > 0.0f) t0 = (t1 < t0) ? t1 : t0;  // std::min(t0, t1)
    t = t0;
    return true;
}

// raytracing.cpp:229-231
__device__ __forceinline__ F3 sphere_normal(F3 o, F3 d, float t, const rtx_sphere& s)
{
    F3 n;
    n.x = (o.x + d.x * t - s.center.x) / s.radius;
    n.y = (o.y + d.y * t - s.center.y) / s.radius;
    n.z = (o.z + d.z * t - s.center.z) / s.radius;
    return n;
}

// Triangle hot record as packed by rtx_pack.cpp (TriHot): 5 float4.
struct TriRec {
    float nx, ny, nz, d;
    float vx, vy, vz, area;
    float wx, wy, wz, v1x;
    float v1y, v1z, v2x, v2y;
    float v2z, ex, ey, ez;
};

// raytracing.cpp:338-365 with the ray-independent head precomputed.  Returns what
// intersecttriangle returns (true also on NaN); vd is handed back for the facing normal (:343-347).
// RTX_F_TRI_MOLLER_TRUMBORE (opt-in, NOT the reference's test: SURVEY.md F8): Moller-Trumbore on the record packed for it
// (rtx_pack.cpp: u in {vx,vy,vz}, e1 = v - u in {wx,wy,wz}, e2 = w - u in {v1x,v1y,v1z}; q0 keeps the reference's unit normal
// for the facing rule).  No epsilon: det == 0 is the only parallel case, as vd == 0 is the reference's; hits need t >= 0 as
// the reference's do.  Plain binary32 operations in this order, so oracle/rt_oracle.c (ORC_TRI_MT) gives the same bits.
__device__ __forceinline__ bool triangle_test_mt(F3 o, F3 dir, const TriRec& T, float& t, float& vd)
{
    vd = T.nx * dir.x + T.ny * dir.y + T.nz * dir.z;
    const float e1x = T.wx, e1y = T.wy, e1z = T.wz, e2x = T.v1x, e2y = T.v1y, e2z = T.v1z;
    const float px = dir.y * e2z - dir.z * e2y, py = dir.z * e2x - dir.x * e2z, pz = dir.x * e2y - dir.y * e2x;
    const float det = e1x * px + e1y * py + e1z * pz;
    if (det == 0.0f) return false;
    const float inv = 1.0f / det;
    const float tx = o.x - T.vx, ty = o.y - T.vy, tz = o.z - T.vz;
    const float uu = (tx * px + ty * py + tz * pz) * inv;
    if (uu < 0.0f || uu > 1.0f) return false;
    const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    const float vv = (dir.x * qx + dir.y * qy + dir.z * qz) * inv;
    if (vv < 0.0f || uu + vv > 1.0f) return false;
    t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    if (t < 0.0f) return false;
    return true;
}

// exitp (counting builds only; the pointer is a compile-time null elsewhere): where the test left --
// 0 at vd == 0 or t < 0 (13 FLOP), 1 at the nearest-t cull (13), 3 at the edge filter (38), 2 after the areas (76).
__device__ __forceinline__ bool triangle_test(F3 o, F3 dir, const TriRec& T, float& t, float& vd, int* exitp = nullptr, const bool mt = false)
{
    if (exitp) *exitp = 0;
    if (mt) { if (exitp) *exitp = 2; return triangle_test_mt(o, dir, T, t, vd); }  // wave-uniform
    vd = T.nx * dir.x + T.ny * dir.y + T.nz * dir.z;
    if (vd == 0.0f) return false;
    t = -((T.nx * o.x + T.ny * o.y + T.nz * o.z) + T.d) / vd;
    if (t < 0.0f) return false;
    if (exitp) *exitp = 2;
    float px = o.x + dir.x * t, py = o.y + dir.y * t, pz = o.z + dir.z * t;
    float ax = px - T.vx, ay = py - T.vy, az = pz - T.vz;  // vec1
    float bx = px - T.wx, by = py - T.wy, bz = pz - T.wz;  // vec2
    // cross1 (:356), cross2 (:357), cross3 (:358), component orders as written
    float c1x = ay * T.v1z - T.v1y * az, c1y = T.v1x * az - T.v1z * ax, c1z = T.v1y * ax - T.v1x * ay;
    float c2x = az * T.v2y - ay * T.v2z, c2y = ax * T.v2z - az * T.v2x, c2z = ay * T.v2x - ax * T.v2y;
    float c3x = bz * T.ey - by * T.ez, c3y = bx * T.ez - bz * T.ex, c3z = by * T.ex - bx * T.ey;
    float areau = 0.5f * sqrtf(c2x * c2x + c2y * c2y + c2z * c2z);
    float areav = 0.5f * sqrtf(c3x * c3x + c3y * c3y + c3z * c3z);
    float areaw = 0.5f * sqrtf(c1x * c1x + c1y * c1y + c1z * c1z);
    // (double)f > 0.01  <=>  f > 0.01f, because 0.01f < 0.01 < nextafterf(0.01f)
    if (fabsf(areau + areav + areaw - T.area) > 0.01f) return false;
    return true;
}

// The same test with the edge filter (rtx_bounds.cpp, tri_edge_filter) between the plane intersection and the
// area square roots: e0 = { a1, m }, e1 = { a2, Q }.  Returns exactly what triangle_test returns.
// tmax: a closest-hit caller that only takes hits with t < nearest passes nearest, and a hit it would drop
// is reported as a miss before the areas are computed (any-hit callers pass +inf: NaN and every t stay in).
__device__ __forceinline__ bool triangle_test_edge(F3 o, F3 dir, const TriRec& T, const float4 e0, const float4 e1, float& t, float& vd,
                                                   const float tmax = __builtin_inff(), int* exitp = nullptr, const bool mt = false)
{
    if (exitp) *exitp = 0;
    if (mt) {  // wave-uniform; the edge records belong to the area test and are not used
        if (exitp) *exitp = 2;
        return triangle_test_mt(o, dir, T, t, vd) && !(t >= tmax);
    }
    vd = T.nx * dir.x + T.ny * dir.y + T.nz * dir.z;
    if (vd == 0.0f) return false;
    t = -((T.nx * o.x + T.ny * o.y + T.nz * o.z) + T.d) / vd;
    if (t < 0.0f) return false;
    if (exitp) *exitp = 1;
    if (t >= tmax) return false;
    float px = o.x + dir.x * t, py = o.y + dir.y * t, pz = o.z + dir.z * t;
    float ax = px - T.vx, ay = py - T.vy, az = pz - T.vz;  // vec1
    {
        const float q = fabsf(ax) + fabsf(ay) + fabsf(az);
        const float b1 = e0.x * ax + e0.y * ay + e0.z * az;
        const float b2 = e1.x * ax + e1.y * ay + e1.z * az;
        const float b0 = (1.0f - b1) - b2;
        const float nm = -e0.w;
        if (exitp) *exitp = 3;
        if (q <= e1.w && (b1 < nm || b2 < nm || b0 < nm + nm)) return false;
    }
    if (exitp) *exitp = 2;
    float bx = px - T.wx, by = py - T.wy, bz = pz - T.wz;  // vec2
    float c1x = ay * T.v1z - T.v1y * az, c1y = T.v1x * az - T.v1z * ax, c1z = T.v1y * ax - T.v1x * ay;
    float c2x = az * T.v2y - ay * T.v2z, c2y = ax * T.v2z - az * T.v2x, c2z = ay * T.v2x - ax * T.v2y;
    float c3x = bz * T.ey - by * T.ez, c3y = bx * T.ez - bz * T.ex, c3z = by * T.ex - bx * T.ey;
    float areau = 0.5f * sqrtf(c2x * c2x + c2y * c2y + c2z * c2z);
    float areav = 0.5f * sqrtf(c3x * c3x + c3y * c3y + c3z * c3z);
    float areaw = 0.5f * sqrtf(c1x * c1x + c1y * c1y + c1z * c1z);
    if (fabsf(areau + areav + areaw - T.area) > 0.01f) return false;
    return true;
}

// triangle_test_edge cut in two at the edge filter, for callers that pool the two halves separately (the shadow
// pass: nine out of ten pairs leave in the first half, and the three square roots of the second run on lanes
// collected from many pairs instead of on the few of one).  The same binary32 operations in the same order:
// stage 1 returns 0 = miss (vd == 0, t < 0, edge filter), 1 = the areas decide (t set); stage 2 recomputes
// P = o + dir t from that t -- the very value triangle_test_edge goes on with -- and returns the test's answer.
__device__ __forceinline__ int triangle_stage1(F3 o, F3 dir, const float4 q0, const float4 q1, const float4 e0, const float4 e1,
                                               float& t, int* exitp = nullptr)
{
    if (exitp) *exitp = 0;
    const float vd = q0.x * dir.x + q0.y * dir.y + q0.z * dir.z;
    if (vd == 0.0f) return 0;
    t = -((q0.x * o.x + q0.y * o.y + q0.z * o.z) + q0.w) / vd;
    if (t < 0.0f) return 0;
    const float px = o.x + dir.x * t, py = o.y + dir.y * t, pz = o.z + dir.z * t;
    const float ax = px - q1.x, ay = py - q1.y, az = pz - q1.z;  // vec1
    const float q = fabsf(ax) + fabsf(ay) + fabsf(az);
    const float b1 = e0.x * ax + e0.y * ay + e0.z * az;
    const float b2 = e1.x * ax + e1.y * ay + e1.z * az;
    const float b0 = (1.0f - b1) - b2;
    const float nm = -e0.w;
    if (exitp) *exitp = 3;
    if (q <= e1.w && (b1 < nm || b2 < nm || b0 < nm + nm)) return 0;
    if (exitp) *exitp = 2;
    return 1;
}

__device__ __forceinline__ bool triangle_stage2(F3 o, F3 dir, const float t, const float4 q1, const float4 q2, const float4 q3, const float4 q4)
{
    const float px = o.x + dir.x * t, py = o.y + dir.y * t, pz = o.z + dir.z * t;
    const float ax = px - q1.x, ay = py - q1.y, az = pz - q1.z;  // vec1
    const float bx = px - q2.x, by = py - q2.y, bz = pz - q2.z;  // vec2
    const float v1x = q2.w, v1y = q3.x, v1z = q3.y, v2x = q3.z, v2y = q3.w, v2z = q4.x, ex = q4.y, ey = q4.z, ez = q4.w;
    const float c1x = ay * v1z - v1y * az, c1y = v1x * az - v1z * ax, c1z = v1y * ax - v1x * ay;
    const float c2x = az * v2y - ay * v2z, c2y = ax * v2z - az * v2x, c2z = ay * v2x - ax * v2y;
    const float c3x = bz * ey - by * ez, c3y = bx * ez - bz * ex, c3z = by * ex - bx * ey;
    const float areau = 0.5f * sqrtf(c2x * c2x + c2y * c2y + c2z * c2z);
    const float areav = 0.5f * sqrtf(c3x * c3x + c3y * c3y + c3z * c3z);
    const float areaw = 0.5f * sqrtf(c1x * c1x + c1y * c1y + c1z * c1z);
    if (fabsf(areau + areav + areaw - q1.w) > 0.01f) return false;
    return true;
}

// x*x*x rounded once (error-free products; the tail sum is far below half an ulp), standing in
// for glibc's pow(x, 3.0) (raytracing.cpp:465) which is correctly rounded in all but
// vanishingly rare cases; the result is rounded to float right after (see fresnel_term).
__device__ __forceinline__ double cube_rn(double x)
{
    double p = x * x;
    double pe = __builtin_fma(x, x, -p);
    double q = p * x;
    double qe = __builtin_fma(p, x, -q);
    return q + (qe + pe * x);
}

// raytracing.cpp:465 / :481 / :499 / :511:  float fresnel = 0.1*1 + 0.9*pow(1.0 - fratio, 3);
__device__ __forceinline__ float fresnel_term(float fratio)
{
    double x = 1.0 - (double)fratio;
    return (float)(0.1 + 0.9 * cube_rn(x));
}

// raytracing.cpp:606: pow(float, float) = glibc powf.  Evaluated through the double-precision
// pow and rounded once to float: the correctly rounded binary32 power except with probability
// ~2^-29.  glibc's powf is itself within a few hundredths of an ulp of that before rounding, so
// the two agree except for operands whose exact power lies that close to a rounding boundary
// (measured mismatch rate: DESIGN.md "numerics"); the north-star tolerance (+-1 LSB) covers it.
// Integer exponents 0 .. 64 -- a Phong exponent as scene files write it -- take square-and-multiply in double instead (round 3:
// the double pow was a third of k_unwind's instructions): <= 12 products, relative error < 64 * 2^-53 = 2^-47 before the one
// rounding to binary32, so the result is the correctly rounded power except with probability ~2^-22 per evaluation -- four orders
// of magnitude below the rate at which glibc's own powf departs from it.  Same special values as powf: x^0 = 1 also for NaN and
// inf, NaN^n = NaN, (-0)^odd = -0, negative bases by the parity of n, overflow to +-inf and underflow to +-0 through the double.
__device__ __forceinline__ float powf_rn(float x, float y)
{
    const int n = (int)y;
    if ((float)n == y && n >= 0 && n <= 64) {
        double r = 1.0, b = (double)x;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if ((n >> i) & 1) r *= b;
            b *= b;
        }
        return (float)r;
    }
    return (float)pow((double)x, (double)y);
}

// raytracing.cpp:792-794: fputc(min(float(1.0), v)*255, f).  std::min(1.0f, v) is 1.0f for NaN;
// float -> int truncates; below-range values give INT_MIN on x86-64 (low byte 0), which is also
// what v_cvt_i32_f32 saturates to.
__device__ __forceinline__ unsigned char quantise(float v)
{
    float m = (v < 1.0f) ? v : 1.0f;
    float p = m * 255.0f;
    int i = (p > -2147483648.0f) ? (int)p : (int)0x80000000u;
    return (unsigned char)(i & 0xff);
}

}  // namespace rtx
#endif
