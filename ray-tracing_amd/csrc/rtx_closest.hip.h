// rtx_closest.hip.h -- closest hit: bounding-sphere filter, packet test, k_closest_f (coherent waves), k_closest_lanes (per-ray traversal), secondary-ray binning
// Part of the one translation unit rtx_device.hip (device code, namespace rtx); not a stand-alone header.
#ifndef RTX_CLOSEST_HIP_H_
#define RTX_CLOSEST_HIP_H_

namespace rtx {

// ------------------------------------------------------------------------------------------------
// Filtered intersection kernels (the default hot path).
//
// Every (ray, primitive) pair first takes an 11-instruction FMA test of the ray's line against a
// conservative bounding sphere of the primitive (rtx_bounds.cpp proves it never rejects a pair
// the reference's test accepts); only survivors run the exact contraction-free test above.
// Bounding-sphere records are staged through LDS in tiles of kTile primitives and read back
// with broadcast ds_read_b128 (all lanes, same address): primitive operands then sit in VGPRs,
// which on gfx950 issue at twice the rate of SGPR / constant-bus operands (tools/ubench_valu2).
// The sign bits of F over a group of kGroup primitives are AND-ed (v_and_b32, full rate) and
// tested once per group, instead of one half-rate v_cmp per primitive.
// ------------------------------------------------------------------------------------------------
constexpr int kTile = 1024;   // primitives per LDS tile (16 KiB)
constexpr int kGroup = 8;     // primitives per sign-AND group

__device__ __forceinline__ float filt(F3 fo, F3 fd, float4 b)
{
    float X = fo.x - b.x, Y = fo.y - b.y, Z = fo.z - b.z;
    float bh = fd.z * Z;
    bh = __builtin_fmaf(fd.y, Y, bh);
    bh = __builtin_fmaf(fd.x, X, bh);
    float cc = __builtin_fmaf(Z, Z, b.w);
    cc = __builtin_fmaf(Y, Y, cc);
    cc = __builtin_fmaf(X, X, cc);
    return __builtin_fmaf(bh, bh, -cc);
}

// The filter with the half-space behind the ray's origin taken out (plain spheres, triangles and polygons only:
// NOT affine spheres, whose object-space ray the reference builds with the translation added to the direction).
// The reference's tests accept a hit only at t >= 0 (sphere: both roots negative is a miss, :217-222; triangle:
// t < 0 is a miss, :347) and the hit point p = o + t n lies in the primitive's bounding sphere (c, R)
// (rtx_bounds.cpp), so n.(p - o) = t >= 0; a sphere entirely behind the origin, n.(o - c) > R, has
// n.(p - o) <= n.(c - o) + R < 0 for each of its points: no hit.  Taken with a margin: bh^2 > 1.01 R^2 + 1e-6 |X|^2
// puts n.(o - c) at least 5e-5 |o - c| beyond R for every ratio |o - c| / R, fifty times the rounding of the
// reference's b, c and t at that distance.  NaN operands and R = inf (records that always pass) compare false:
// no cull.  A zero direction (point query) has bh = 0: no cull.
__device__ __forceinline__ bool filt_front(F3 fo, F3 fd, float4 b)  // true: the exact test may accept
{
    float X = fo.x - b.x, Y = fo.y - b.y, Z = fo.z - b.z;
    float bh = fd.z * Z;
    bh = __builtin_fmaf(fd.y, Y, bh);
    bh = __builtin_fmaf(fd.x, X, bh);
    float cc = __builtin_fmaf(Z, Z, b.w);
    cc = __builtin_fmaf(Y, Y, cc);
    cc = __builtin_fmaf(X, X, cc);
    const float F = __builtin_fmaf(bh, bh, -cc);
    const bool behind = bh > 0.0f && bh * bh > __builtin_fmaf(-1.01f, b.w, 1e-6f * (cc - b.w));
    return !(F < 0.0f) && !behind;
}

__device__ __forceinline__ bool finite3(F3 v)
{
    return (fabsf(v.x) <= 3.0e38f) && (fabsf(v.y) <= 3.0e38f) && (fabsf(v.z) <= 3.0e38f);
}
__device__ __forceinline__ bool tiny_comp(float c) { return c != 0.0f && fabsf(c) < 1e-10f; }

// A ray the filter's error analysis does not cover: it skips the filter and takes the exact test
// against every primitive (rtx_bounds.cpp, "abnormal" lanes).
__device__ __forceinline__ bool ray_abnormal(F3 o, F3 d, F3 nd, float guard)
{
    bool ok = finite3(o) && finite3(d) && finite3(nd);
    float n2 = nd.x * nd.x + nd.y * nd.y + nd.z * nd.z;
    bool dir = (fabsf(n2 - 1.0f) <= 1e-3f);
    dir = dir && (fabsf(d.x - nd.x) <= 1e-5f) && (fabsf(d.y - nd.y) <= 1e-5f) && (fabsf(d.z - nd.z) <= 1e-5f);
    dir = dir && !(tiny_comp(d.x) || tiny_comp(d.y) || tiny_comp(d.z) || tiny_comp(nd.x) || tiny_comp(nd.y) || tiny_comp(nd.z));
    // A direction of exactly (0,0,0) -- what the reference's normalize (:116-125) leaves behind when a
    // singular sphere matrix has flattened the incoming ray -- is a POINT: with fd = 0 the filter
    // reads R^2 - |X|^2 >= 0, "the origin lies inside the bounding sphere", which is what the exact
    // tests need too (sphere: b = 0, disc = -4c >= 0; triangle: vd == 0 rejects) -- rtx_bounds.cpp.
    dir = dir || (d.x == 0.0f && d.y == 0.0f && d.z == 0.0f);
    ok = ok && dir;
    ok = ok && (fabsf(o.x) <= guard) && (fabsf(o.y) <= guard) && (fabsf(o.z) <= guard);
    return !ok;
}

// A ray with a NaN in its origin or direction -- what total internal reflection leaves behind: sqrtf(k < 0) at
// raytracing.cpp:483-485 makes every component of the refracted direction NaN -- hits nothing in the reference's
// closest-hit loops: every t it computes is NaN (sphere: b and c are sums with the NaN in them, :209-216; triangle and
// polygon: t = .../vd, :345, :313; 0 * NaN is NaN too, so no zero coefficient saves a term) and `t < nearest` (:392,
// :411, :426) is false for NaN.  Such rays are answered "miss" without a single test instead of taking the
// every-primitive path of the rays the filter cannot analyse (5 %-refractive C4: 84 ms of a 119 ms frame was that).
__device__ __forceinline__ bool ray_dead(F3 o, F3 d)
{
    return (o.x != o.x) || (o.y != o.y) || (o.z != o.z) || (d.x != d.x) || (d.y != d.y) || (d.z != d.z);
}

constexpr float kDirScale = 1.0f + 1.0f / 262144.0f;  // (1 + 2^-18), see rtx_bounds.cpp

// Block-cooperative staging of one tile of filter records; the tail of the last group is padded
// with records no ray can pass (R^2 = -inf).
__device__ __forceinline__ int stage_tile(float4* tile, const float4* bounds, int t0, int nprim)
{
    int cnt = min(kTile, nprim - t0);
    for (int j = threadIdx.x; j < cnt; j += kBlock) tile[j] = bounds[t0 + j];
    int padded = (cnt + kGroup - 1) / kGroup * kGroup;
    for (int j = cnt + threadIdx.x; j < padded; j += kBlock) tile[j] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());
    return cnt;
}

struct HitAcc { float nearest; int best; int tflag; float txf; };  // txf: t of the last affine-sphere record (thin mode)

// exact closest-hit update for primitive k (wave-uniform), raytracing.cpp:385-403 / :407-418
__device__ __forceinline__ void exact_closest(const SceneDev& S, int k, F3 o, F3 d, F3 nd, HitAcc& h, Exec* ex = nullptr)
{
    if (k < S.ns) {
        float4 hh = S.sph_hot[k];
        int xf = S.sph_xf[k];
        ex_add(ex, xf ? EX_SPHX : EX_SPH);
        F3 to = o, td = nd;
        if (xf) {
            const rtx_xmatrix& im = S.sph_im[k];
            to = inv_apply(o, im);
            td = normalized(inv_apply(d, im));
        }
        float t;
        if (sphere_test(to, td, hh.x, hh.y, hh.z, hh.w, t)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; if (xf) { h.tflag = kTFlag; h.txf = t; } }
        }
    } else if (k < S.ns + S.nt) {
        TriRec T = load_tri(S.tri_hot, k - S.ns);
        const float4 e0 = S.tri_edge[(size_t)(k - S.ns) * 2], e1 = S.tri_edge[(size_t)(k - S.ns) * 2 + 1];
        float t, vd;
        int exitp = 0;
        const bool hit = triangle_test_edge(o, d, T, e0, e1, t, vd, h.nearest, ex ? &exitp : nullptr, S.mt != 0);
        if (ex) ex_tri(ex, exitp);
        if (hit) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    } else {
        const float4* p = S.phot + (size_t)k * 5;
        float t;
        ex_add(ex, EX_POLY, (unsigned)max(__float_as_int(p[1].z), 1));
        if (polygon_test(o, d, p[0], p[1], S.poly_verts, t)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    }
}


// The ordered scan over every primitive (raytracing.cpp:385-418 as it stands) for the FLAGGED lanes of a wave: rays whose list of
// affine-sphere hits overflowed (LexAcc), rays no table or filter covers.  Few flagged lanes -- the usual case: one ray in a wave
// of a scene whose affine spheres overlap five deep -- go one RAY at a time: the wave's 64 lanes filter 64 primitives a step and the
// survivors take the exact test in index order, wave-uniform (157 steps for 10,000 primitives where the flagged lane alone took
// 10,000: one such ray held its whole launch for 2 ms, tools/dbg/perf_survey.py "dense").  Many flagged lanes (a camera cell without
// a list: all 64) scan each for itself, 64 rays a step.  Same tests on the same operands in the same order either way.
__device__ __forceinline__ void ordered_scan(const SceneDev& S, const bool slow, const bool abn, const F3 o, const F3 d, const F3 nd,
                                             float& nearest, int& best, int& tflag, Exec* ex)
{
    const unsigned long long sm0 = __ballot(slow);
    if (sm0 == 0ull) return;  // wave-uniform
    const int nprim = S.ns + S.nt + S.np;
    const int lane = threadIdx.x & 63;
    if (__popcll(sm0) > 16) {
        const F3 fo = abn ? f3(1e18f, 0.0f, 0.0f) : o;
        const F3 fd = abn ? f3(0.0f, 0.0f, 0.0f) : f3(nd.x * kDirScale, nd.y * kDirScale, nd.z * kDirScale);
        HitAcc hs;
        hs.nearest = kMaxT; hs.best = -1; hs.tflag = 0; hs.txf = kMaxT;
        for (int k = 0; k < nprim; k++) {
            const float4 b = S.bounds[k];
            if (slow && !abn) ex_add(ex, EX_FILT);
            if (slow && (abn || !(filt(fo, fd, b) < 0.0f))) exact_closest(S, k, o, d, nd, hs, ex);
        }
        if (slow) { nearest = hs.nearest; best = hs.best; tflag = hs.tflag; }
        return;
    }
    const unsigned long long am = __ballot(abn);
    auto bc = [](float v, int r) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
    for (unsigned long long sm = sm0; sm != 0ull; sm &= sm - 1) {
        const int r = __ffsll((long long)sm) - 1;
        const bool rabn = ((am >> r) & 1ull) != 0ull;
        const F3 ro = f3(bc(o.x, r), bc(o.y, r), bc(o.z, r)), rd = f3(bc(d.x, r), bc(d.y, r), bc(d.z, r)), rn = f3(bc(nd.x, r), bc(nd.y, r), bc(nd.z, r));
        const F3 fo = ro, fd = f3(rn.x * kDirScale, rn.y * kDirScale, rn.z * kDirScale);
        HitAcc hs;
        hs.nearest = kMaxT; hs.best = -1; hs.tflag = 0; hs.txf = kMaxT;
        Exec tmp;
        Exec* const te = ex ? &tmp : nullptr;  // every lane runs the survivors' tests: counted once, for the ray's own lane
        ex_zero(te);
        for (int k0 = 0; k0 < nprim; k0 += 64) {
            const int k = k0 + lane;
            bool pass = k < nprim;
            if (pass && !rabn) pass = !(filt(fo, fd, S.bounds[k]) < 0.0f);
            for (unsigned long long pm = __ballot(pass); pm != 0ull; pm &= pm - 1)
                exact_closest(S, k0 + (__ffsll((long long)pm) - 1), ro, rd, rn, hs, te);
        }
        if (lane == r) {
            nearest = hs.nearest; best = hs.best; tflag = hs.tflag;
            if (ex) {
                if (!rabn) ex->n[EX_FILT] += (unsigned)nprim;
                for (int q = 0; q < EX_KINDS; q++) ex->n[q] += tmp.n[q];
            }
        }
    }
}

// exact_closest for a wave-uniform primitive k whose hot record sits in lane u's registers
__device__ __forceinline__ void exact_closest_bcast(const SceneDev& S, int k, const PrimRec& pr, int u, bool pass, F3 o, F3 d, F3 nd, HitAcc& h, Exec* ex = nullptr)
{
    const float4 q0 = bcast4(pr.q0, u);
    if (k < S.ns) {
        const float4 q4 = bcast4(pr.q4, u);
        const bool xf = q4.x != 0.0f;
        if (pass) ex_add(ex, xf ? EX_SPHX : EX_SPH);
        F3 to = o, td = nd;
        if (xf) {  // wave-uniform
            const float4 q1 = bcast4(pr.q1, u), q2 = bcast4(pr.q2, u), q3 = bcast4(pr.q3, u);
            rtx_xmatrix im;
            im.a.x = q1.x; im.a.y = q1.y; im.a.z = q1.z; im.b.x = q1.w;
            im.b.y = q2.x; im.b.z = q2.y; im.c.x = q2.z; im.c.y = q2.w;
            im.c.z = q3.x; im.d.x = q3.y; im.d.y = q3.z; im.d.z = q3.w;
            to = inv_apply(o, im);
            td = normalized(inv_apply(d, im));
        }
        float t;
        if (pass && sphere_test(to, td, q0.x, q0.y, q0.z, q0.w, t)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; if (xf) { h.tflag = kTFlag; h.txf = t; } }
        }
    } else if (k < S.ns + S.nt) {
        const float4 q1 = bcast4(pr.q1, u), q2 = bcast4(pr.q2, u), q3 = bcast4(pr.q3, u), q4 = bcast4(pr.q4, u);
        TriRec T;
        T.nx = q0.x; T.ny = q0.y; T.nz = q0.z; T.d = q0.w;
        T.vx = q1.x; T.vy = q1.y; T.vz = q1.z; T.area = q1.w;
        T.wx = q2.x; T.wy = q2.y; T.wz = q2.z; T.v1x = q2.w;
        T.v1y = q3.x; T.v1z = q3.y; T.v2x = q3.z; T.v2y = q3.w;
        T.v2z = q4.x; T.ex = q4.y; T.ey = q4.z; T.ez = q4.w;
        float t, vd;
        int exitp = 0;
        const bool hit = pass && triangle_test(o, d, T, t, vd, ex ? &exitp : nullptr, S.mt != 0);
        if (ex && pass) ex_tri(ex, exitp);
        if (hit) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    } else {
        const float4 q1 = bcast4(pr.q1, u);
        float t;
        if (pass) ex_add(ex, EX_POLY, (unsigned)max(__float_as_int(q1.z), 1));
        if (pass && polygon_test(o, d, q0, q1, S.poly_verts, t)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Packet test.  A wave's 64 rays are bounded by a sphere around their origins (co, rho) and a cone
// around their directions (axis w, half-angle theta).  Each LANE then tests a DIFFERENT primitive's
// bounding sphere against that packet (64 primitives per instruction instead of one), a ballot turns
// the answers into a mask, and only primitives in the mask go through the per-ray filter (and, if
// that passes, the exact test), in index order.  Coherent waves -- primary rays, shadow rays of
// neighbouring pixels towards one light -- keep a handful of the 10,000 primitives.
//
// Conservative by construction: the per-ray filter passes only if the primitive's centre lies within
// R of the double cone (apex o_i, axis d_i, half-angle atan sqrt(eps/(1-eps))), eps = 1.2e-5 bounding
// the filter's 2^-18 direction scale and binary32 evaluation error relative to |X|^2; that cone is
// inside the packet's cone widened by the same angle and thickened by rho.  The test below is
//      dist(c - co, cone(w, theta')) <= R + rho     i.e.     h cos(theta') <= R + rho + a sin(theta')
// with a = |w.C|, h^2 = |C|^2 - a^2, squared, every rounding pushed to the passing side.  Waves with an
// abnormal lane, or a cone wider than ~70 degrees, do not use it (they take the grouped loop).
// ------------------------------------------------------------------------------------------------
// For BOUNDS only (never in an exact test): the hardware's 1-ulp square root and reciprocal instead of the
// correctly rounded sequences (10+ instructions each); every use carries a relative margin of 1e-5 or more.
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32

struct Packet { F3 co; float rho; F3 w; float cosT2, sinT; float c0; bool ok; };  // c0 = cos(theta), rounded down, before the widening by E

// Wave reductions for the BOUNDS (packets): four DPP steps reduce each row of 16 lanes in place (quad swaps, half-row
// and row mirrors: register-to-register, no LDS round trip), the four row results are read with v_readlane and
// combined -- the result is wave-uniform and lands in an SGPR.  (The ds_bpermute butterfly this replaces was six
// dependent LDS-pipeline round trips per reduction, five reductions per packet.)  All 64 lanes must be active.
#ifndef RTX_DPP_REDUCE
#define RTX_DPP_REDUCE 1
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
#if RTX_DPP_REDUCE
__device__ __forceinline__ float wave_min(float v)
{
    v = fminf(v, dpp_f<0xB1>(v)); v = fminf(v, dpp_f<0x4E>(v)); v = fminf(v, dpp_f<0x141>(v)); v = fminf(v, dpp_f<0x140>(v));
    return fminf(fminf(lane_f(v, 0), lane_f(v, 16)), fminf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float wave_max(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); v += dpp_f<0x141>(v); v += dpp_f<0x140>(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
#else
__device__ __forceinline__ float wave_min(float v) { for (int m = 32; m; m >>= 1) v = fminf(v, __shfl_xor(v, m)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int m = 32; m; m >>= 1) v = fmaxf(v, __shfl_xor(v, m)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m); return v; }
#endif

// apex_ok: every line of the wave passes through `apex` (shadow rays: their light); the origin sphere then
// shrinks to the slack that claim needs (the 1e-4 origin bias, the filter's angular tolerance and the
// direction rounding, both proportional to the distance from the light).
__device__ __forceinline__ Packet make_packet(F3 o, F3 nd, bool live, bool abn, bool apex_ok = false, F3 apex = F3{0.0f, 0.0f, 0.0f})
{
    Packet P;
    unsigned long long lm = __ballot(live);
    bool bad = __any(live && abn) || lm == 0;
    int src = lm ? (__ffsll((long long)lm) - 1) : 0;
    // dead lanes borrow a live lane's ray so they do not widen the bounds
    // (the shuffles are evaluated by EVERY lane, outside the selects: a ds_bpermute under a divergent branch
    // reads 0 from the lanes that did not take the branch)
    // (src is wave-uniform: v_readlane, no LDS round trip)
    const F3 so = f3(lane_f(o.x, src), lane_f(o.y, src), lane_f(o.z, src));
    const F3 sd = f3(lane_f(nd.x, src), lane_f(nd.y, src), lane_f(nd.z, src));
    F3 oo = f3(live ? o.x : so.x, live ? o.y : so.y, live ? o.z : so.z);
    F3 dd = f3(live ? nd.x : sd.x, live ? nd.y : sd.y, live ? nd.z : sd.z);
    if (apex_ok) {  // wave-uniform: one reduction instead of the seven of the origins' bounding sphere
        float ax = oo.x - apex.x, ay = oo.y - apex.y, az = oo.z - apex.z;
        float maxD = wave_max(fsqrt(ax * ax + ay * ay + az * az));
        P.rho = 1.3e-4f + 3.8e-3f * maxD;
        P.co = apex;
    } else {
        float lx = wave_min(oo.x), hx = wave_max(oo.x), ly = wave_min(oo.y), hy = wave_max(oo.y), lz = wave_min(oo.z), hz = wave_max(oo.z);
        P.co = f3(0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz));
        float ex = oo.x - P.co.x, ey = oo.y - P.co.y, ez = oo.z - P.co.z;
        P.rho = wave_max(fsqrt(ex * ex + ey * ey + ez * ez)) * 1.00002f + 1e-30f;
    }
    F3 sw = f3(wave_sum(dd.x), wave_sum(dd.y), wave_sum(dd.z));
    float sn = fsqrt(sw.x * sw.x + sw.y * sw.y + sw.z * sw.z);
    bad = bad || !(sn > 1e-3f) || !(P.rho < 1e30f);
    float inv = frcp(fmaxf(sn, 1e-3f));
    P.w = f3(sw.x * inv, sw.y * inv, sw.z * inv);
    float c = wave_min(P.w.x * dd.x + P.w.y * dd.y + P.w.z * dd.z) - 4e-6f;  // cos(theta), rounded down
    bad = bad || !(c > 0.35f);
    c = fminf(c, 1.0f);
    P.c0 = c;
    float sT = fsqrt(fmaxf(0.0f, 1.0f - c * c)) + 4e-6f;
    // widen by the filter's own angular tolerance (3.47e-3 rad) plus slack: E = 3.7e-3 rad
    const float cosE = 0.99999316f, sinE = 3.7e-3f;
    float cT = c * cosE - sT * sinE - 4e-6f;           // cos(theta + E), rounded down
    float sT2 = sT * cosE + c * sinE + 4e-6f;          // sin(theta + E), rounded up
    bad = bad || !(cT > 0.3f);
    P.cosT2 = cT * cT;
    P.sinT = sT2;
    P.ok = !bad;
    return P;
}

// true: the primitive may pass some lane's filter.  b = { c.xyz, -R^2 }.
__device__ __forceinline__ bool packet_test(const Packet& P, float4 b)
{
    float Cx = b.x - P.co.x, Cy = b.y - P.co.y, Cz = b.z - P.co.z;
    float C2 = Cx * Cx + Cy * Cy + Cz * Cz;
    float a = fabsf(P.w.x * Cx + P.w.y * Cy + P.w.z * Cz);
    float h2 = (C2 - a * a) - 4e-6f * C2;              // rounded down (cancellation, |w| != 1)
    float Rp = (fsqrt(-b.w) * 1.00001f + P.rho) * 1.00001f + 1e-20f;
    float rhs = Rp + a * P.sinT;
    return !(h2 * P.cosT2 > rhs * rhs * 1.00001f);     // NaN (R = inf, never-bounded primitives) passes
}

// The same test for ONE ray (origin o, unit or zero direction n): can the sphere b = { c, R } contain
// the bounding sphere of a primitive whose filter passes for this ray?  The filter's pass region lies
// inside { c : dist(c, double cone(o, +-n, E)) <= R_prim } (above), so a sphere around such primitives
// passes if its centre is within its radius of that cone.  Returns the slack, sign bit clear = pass
// (R = +inf gives +inf).  Used on cluster and sub-cluster spheres by the per-ray traversal; the plain
// filter is NOT valid on enclosing spheres: its hyperboloid lacks the cone's cross term 2 R a tan(E).
// (Round 3: written with FMAs -- 15 instructions instead of 28.  These are BOUNDS, not the reference's arithmetic: every term
// keeps its slack -- 4e-6 C^2 under the cancellation, 1.00001 on the radius and on the right-hand side, cosE2 rounded down,
// sinE rounded up -- and a fused product only drops a rounding of 6e-8 relative, a sixtieth of the smallest of them.)
__host__ __device__ __forceinline__ float cone_slack(F3 o, F3 n, float4 b)
{
    const float cosE2 = 0.9999783f;  // (cos(3.7e-3) - 4e-6)^2 rounded down
    const float sinE = 3.705e-3f;    // sin(3.7e-3) rounded up, + 4e-6
    const float Cx = b.x - o.x, Cy = b.y - o.y, Cz = b.z - o.z;
    const float C2 = __builtin_fmaf(Cx, Cx, __builtin_fmaf(Cy, Cy, Cz * Cz));
    const float a = fabsf(__builtin_fmaf(n.x, Cx, __builtin_fmaf(n.y, Cy, n.z * Cz)));
    const float h2 = __builtin_fmaf(-a, a, C2 * (1.0f - 4e-6f));
    const float rhs = __builtin_fmaf(a, sinE, b.w * 1.00001f + 1e-20f);
    return __builtin_fmaf(rhs * 1.00001f, rhs, -(h2 * cosE2));
}

// cone_slack with the half-space behind the origin taken out (see filt_front; b = { c, R } encloses bounding
// spheres of plain primitives only): sign bit set = no member can be hit.
__host__ __device__ __forceinline__ float cone_slack_front(F3 o, F3 n, float4 b)
{
    const float cosE2 = 0.9999783f;
    const float sinE = 3.705e-3f;
    const float Cx = b.x - o.x, Cy = b.y - o.y, Cz = b.z - o.z;
    const float C2 = __builtin_fmaf(Cx, Cx, __builtin_fmaf(Cy, Cy, Cz * Cz));
    const float as = __builtin_fmaf(n.x, Cx, __builtin_fmaf(n.y, Cy, n.z * Cz));
    const float a = fabsf(as);
    const float h2 = __builtin_fmaf(-a, a, C2 * (1.0f - 4e-6f));
    const float rhs = __builtin_fmaf(a, sinE, b.w * 1.00001f + 1e-20f);
    const bool behind = as < 0.0f && as * as > __builtin_fmaf(1.01f * b.w, b.w, 1e-6f * C2);
    return behind ? -1.0f : __builtin_fmaf(rhs * 1.00001f, rhs, -(h2 * cosE2));
}

// cone_slack_front with what lies wholly BEYOND a distance taken out as well (round 3): lr = 1.015625 L + 1e-5 guard for a ray
// whose state cannot be changed by a hit at t > L (closest_pool).  b = { c, R } encloses bounding spheres of plain spheres,
// triangles and polygons only.  A hit the exact tests accept has its point in the primitive's bounding sphere (rtx_bounds.cpp:
// the triangle's sphere holds P = o + t d AS COMPUTED; a plain sphere's root lies on it up to the rounding below), so
// n.(P - o) >= n.(c - o) - |n| R = as - R for every sphere b around it, and
//   triangle, polygon: t (n.d) = n.(P - o) - n.eps, |eps| <= 8u (guard + t), n.d in [0.998, 1.002] (ray_abnormal admits
//     |d - n| <= 1e-5 per component, |n|^2 = 1 +- 8u from normalized()): t >= (as - R - 1e-6 guard) / 1.0021;
//   plain sphere (sphere_test: t >= t0 = (-b - sqrt(disc)) / 2, -b / 2 = n.(c - o) +- 4u |X|, disc / 4 <= r^2 (1 + 12u) + 32u |X|^2
//     as computed, and an accepted hit has |X| <= 1.00001 (as + r)): t >= as (1 - 1.5e-3) - r (1 + 1.5e-3) - 1e-6 guard,
//     and for an enclosing sphere (as_m <= as + R, r_m <= R): t >= as (1 - 1.5e-3) - R (1 + 3e-3) - 1e-6 guard.
// The test as * (1 - 1/128) - R * (1 + 1/64) > 1.015625 L + 1e-5 guard therefore puts every such t strictly above L, with the
// rounding of as itself (4u |c - o| <= 1e-6 guard) a tenth of the absolute term.  R = +inf and NaN compare false: no cull.
__host__ __device__ __forceinline__ float cone_slack_cull(F3 o, F3 n, float4 b, float lr)
{
    const float cosE2 = 0.9999783f;
    const float sinE = 3.705e-3f;
    const float Cx = b.x - o.x, Cy = b.y - o.y, Cz = b.z - o.z;
    const float C2 = __builtin_fmaf(Cx, Cx, __builtin_fmaf(Cy, Cy, Cz * Cz));
    const float as = __builtin_fmaf(n.x, Cx, __builtin_fmaf(n.y, Cy, n.z * Cz));
    const float a = fabsf(as);
    const float h2 = __builtin_fmaf(-a, a, C2 * (1.0f - 4e-6f));
    const float rhs = __builtin_fmaf(a, sinE, b.w * 1.00001f + 1e-20f);
    const bool behind = as < 0.0f && as * as > __builtin_fmaf(1.01f * b.w, b.w, 1e-6f * C2);
    const bool far = __builtin_fmaf(as, 0.9921875f, -1.015625f * b.w) > lr;
    return (behind || far) ? -1.0f : __builtin_fmaf(rhs * 1.00001f, rhs, -(h2 * cosE2));
}

// packet_test for a sphere given as { c, R } (cluster and sub-cluster spheres of the per-ray hierarchy, clR / subR)
__device__ __forceinline__ bool packet_test_R(const Packet& P, float4 b)
{
    float Cx = b.x - P.co.x, Cy = b.y - P.co.y, Cz = b.z - P.co.z;
    float C2 = Cx * Cx + Cy * Cy + Cz * Cz;
    float a = fabsf(P.w.x * Cx + P.w.y * Cy + P.w.z * Cz);
    float h2 = (C2 - a * a) - 4e-6f * C2;
    float Rp = (b.w * 1.00001f + P.rho) * 1.00001f + 1e-20f;
    float rhs = Rp + a * P.sinT;
    return !(h2 * P.cosT2 > rhs * rhs * 1.00001f);     // R = +inf (clusters of never-bounded primitives) passes
}
__device__ __forceinline__ int lanes_below_(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <bool PREFETCH>
__device__ __forceinline__ void closest_packet(const SceneDev& S, const LevelDev& L, const int n, const int base, float4* tile, Exec* ex)
{
    const int lane = threadIdx.x & 63;
    const int q = base + threadIdx.x;
    const bool inq = q < n;
    const int i = inq ? q : 0;
    const int ii = i;
    F3 o = f3(L.ox[ii], L.oy[ii], L.oz[ii]);
    F3 d = f3(L.dx[ii], L.dy[ii], L.dz[ii]);
    F3 nd = normalized(d);
    const bool live = inq && !ray_dead(o, d);  // NaN rays miss everything: they sit the kernel out and are written as misses
    const bool abn = ray_abnormal(o, d, nd, S.guard);
    const bool use = live && !abn;
    F3 fo = use ? o : f3(1e18f, 0.0f, 0.0f);
    F3 fd = use ? f3(nd.x * kDirScale, nd.y * kDirScale, nd.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
    HitAcc h;
    h.nearest = kMaxT; h.best = -1; h.tflag = 0; h.txf = kMaxT;
    // generation 0 only: every ray leaves the camera, the packet's apex (one wave reduction instead of seven)
    const unsigned long long lm0 = __ballot(live);
    const int src0 = lm0 ? (__ffsll((long long)lm0) - 1) : 0;
    const F3 cam = f3(lane_f(o.x, src0), lane_f(o.y, src0), lane_f(o.z, src0));
    const bool one_origin = __all(!live || (o.x == cam.x && o.y == cam.y && o.z == cam.z));
    const Packet P = make_packet(o, nd, live, abn, one_origin, cam);
    if (lane == 0) ex_add(ex, EX_WAVES);
    if (live) ex_add(ex, EX_RAYS);
    int pk_batches = 0, pk_surv = 0;
    const int nprim = S.ns + S.nt + S.np;
    // this wave's bitmap of candidate primitives of the current segment (wave-private LDS: the wave's
    // DS operations execute in issue order, the fences only pin the compiler's ordering)
    volatile unsigned* bm = reinterpret_cast<volatile unsigned*>(tile) + (threadIdx.x >> 6) * (kSegPrims / 32 + 512);  // + 1024-entry candidate list
    for (int seg = 0; seg < S.cnseg; seg++) {
        const int p0 = seg * kSegPrims;
        const int pn = min(kSegPrims, nprim - p0);
        const int words = (pn + 31) >> 5;
        if (P.ok) {  // wave-uniform
            for (int j = lane; j < words; j += 64) bm[j] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            const int c0 = S.cseg[seg], c1 = S.cseg[seg + 1];
            for (int cb = c0; cb < c1; cb += 64) {
                const int c = cb + lane;
                unsigned long long cm = __ballot(c < c1 && packet_test(P, S.ccl[min(c, c1 - 1)]));
                if (c < c1) ex_add(ex, EX_PACKET);
                pk_batches++;
                if constexpr (!PREFETCH) {
                    // Frames that fill the chip: two-level member selection (as shadow_select in rtx_shadow.hip.h).  The
                    // passing clusters are listed and taken eight at a time, lane = (cluster, sub-cluster) against the
                    // sphere around members 8s .. 8s+7 (csubR); the passing sub-clusters are listed and taken eight at
                    // a time, lane = (sub-cluster, member).  19 member steps per wave become 2 + 4 on C4.  The lists use
                    // the wave's candidate-list space, which this walk does not need.
                    volatile int* clist = reinterpret_cast<volatile int*>(const_cast<unsigned*>(bm) + kSegPrims / 32);
                    volatile int* slist = clist + 64;
                    const int j8 = lane >> 3;
                    if (cm != 0ull) {
                        const int nc = __popcll(cm);
                        if ((cm >> lane) & 1ull) clist[lanes_below_(cm)] = c;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        for (int q0 = 0; q0 < nc; q0 += 8) {
                            const bool v1 = q0 + j8 < nc;
                            const int sid0 = clist[min(q0 + j8, nc - 1)] * 8 + (lane & 7);
                            const bool spass = v1 && packet_test_R(P, S.csubR[sid0]);
                            if (v1) ex_add(ex, EX_PACKET);
                            const unsigned long long sm = __ballot(spass);
                            pk_batches++;
                            if (sm == 0ull) continue;  // wave-uniform
                            const int nsb = __popcll(sm);
                            if (spass) slist[lanes_below_(sm)] = sid0;
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            for (int s0 = 0; s0 < nsb; s0 += 8) {
                                const bool v2 = s0 + j8 < nsb;
                                const size_t m = (size_t)slist[min(s0 + j8, nsb - 1)] * 8 + (lane & 7);
                                const float4 mb = S.cmem[m];
                                const int k = v2 ? S.cidx[m] : -1;
                                const bool pass = k >= 0 && packet_test(P, mb);
                                if (k >= 0) ex_add(ex, EX_PACKET);
                                if (pass) atomicOr(const_cast<unsigned*>(bm) + ((k - p0) >> 5), 1u << ((k - p0) & 31));
                                pk_batches++; pk_surv += __popcll(__ballot(pass));
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    cm = 0ull;
                }
                while (cm) {
                    if constexpr (PREFETCH) {
                        // four clusters' members in flight at a time: one memory round trip per four
                        int jj[4];
#pragma unroll
                        for (int a = 0; a < 4; a++) { jj[a] = cm ? (__ffsll((long long)cm) - 1) : -1; cm &= cm - 1; }
                        float4 mb[4];
                        int kk[4];
#pragma unroll
                        for (int a = 0; a < 4; a++) {
                            const size_t m = (size_t)(cb + max(jj[a], 0)) * 64 + lane;
                            mb[a] = S.cmem[m];
                            kk[a] = S.cidx[m];
                        }
#pragma unroll
                        for (int a = 0; a < 4; a++) {
                            const bool pass = jj[a] >= 0 && kk[a] >= 0 && packet_test(P, mb[a]);
                            if (jj[a] >= 0 && kk[a] >= 0) ex_add(ex, EX_PACKET);
                            if (pass) atomicOr(const_cast<unsigned*>(bm) + ((kk[a] - p0) >> 5), 1u << ((kk[a] - p0) & 31));
                            if (jj[a] >= 0) { pk_batches++; pk_surv += __popcll(__ballot(pass)); }
                        }
                    } else {
                        const int j = __ffsll((long long)cm) - 1;
                        cm &= cm - 1;
                        const size_t m = (size_t)(cb + j) * 64 + lane;
                        const float4 mb = S.cmem[m];
                        const int k = S.cidx[m];
                        const bool pass = k >= 0 && packet_test(P, mb);
                        if (k >= 0) ex_add(ex, EX_PACKET);
                        if (pass) atomicOr(const_cast<unsigned*>(bm) + ((k - p0) >> 5), 1u << ((k - p0) & 31));
                        pk_batches++; pk_surv += __popcll(__ballot(pass));
                    }
                }
            }
        } else {
            for (int j = lane; j < words; j += 64) bm[j] = (j * 32 + 32 <= pn) ? 0xFFFFFFFFu : ((1u << (pn & 31)) - 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if constexpr (PREFETCH) {
        // Walk the bitmap in index order: file order, as the reference's loops (:385-418).
        // Few waves per SIMD (frames below ~1 M rays): a wave's serial chain of dependent loads IS the run time.
        // The set bits of 32 words are first written out as a list (wave-level prefix sum of the popcounts keeps
        // the order), then 64 candidates at a time fetch their filter record and hot record with independent
        // loads -- one memory round trip for the lot -- and the ordered loop reads them from registers.
        volatile unsigned short* list = reinterpret_cast<volatile unsigned short*>(const_cast<unsigned*>(bm) + kSegPrims / 32);
        for (int w0 = 0; w0 < words; w0 += 32) {
            const unsigned wv = (lane < 32 && w0 + lane < words) ? bm[w0 + lane] : 0u;
            const int cnt = __popc(wv);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(incl, off);
                if (lane >= off) incl += up;
            }
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            if (tot == 0) continue;  // wave-uniform
            {
                unsigned bits = wv;
                int pos = incl - cnt;
                while (bits) {
                    const int bpos = __ffs(bits) - 1;
                    bits &= bits - 1;
                    list[pos++] = (unsigned short)(lane * 32 + bpos);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int c0 = 0; c0 < tot; c0 += 64) {
                const int m = min(64, tot - c0);
                int k = 0;
                float4 br = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                PrimRec pr;
                pr.q0 = pr.q1 = pr.q2 = pr.q3 = pr.q4 = br;
                if (lane < m) {
                    k = p0 + (w0 << 5) + (int)list[c0 + lane];
                    br = S.bounds[k];
                    pr = load_prim(S.phot, k);
                }
                for (int u = 0; u < m; u++) {
                    const float4 b = bcast4(br, u);
                    const bool pass = live && (abn || !(filt(fo, fd, b) < 0.0f));
                    if (use) ex_add(ex, EX_FILT);
                    if (__any(pass)) exact_closest_bcast(S, __builtin_amdgcn_readlane(k, u), pr, u, pass, o, d, nd, h, ex);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        } else {
        // Many waves per SIMD (a 4K frame): latency hides itself and the plain ordered loop costs fewer
        // instructions -- one scalar-loaded filter record and, on a pass, one exact test per set bit.
        for (int w0 = 0; w0 < words; w0 += 64) {
            const unsigned wv = (w0 + lane < words) ? bm[w0 + lane] : 0u;
            unsigned long long nz = __ballot(wv != 0u);
            while (nz) {
                const int j = __ffsll((long long)nz) - 1;
                nz &= nz - 1;
                unsigned bits = __builtin_amdgcn_readlane(wv, j);
                while (bits) {
                    const int bpos = __ffs(bits) - 1;
                    bits &= bits - 1;
                    const int k = p0 + ((w0 + j) << 5) + bpos;
                    const float4 br = S.bounds[k];
                    if (use) ex_add(ex, EX_FILT);
                    if (live && (abn || !(filt(fo, fd, br) < 0.0f))) exact_closest(S, k, o, d, nd, h, ex);
                }
            }
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (S.pstat && lane == 0) {  // RTX_F_PACKET_STATS only: same-address atomics cost ~12 ns each
        atomicAdd(S.pstat + 0, 1ull); atomicAdd(S.pstat + 1, P.ok ? 1ull : 0ull);
        atomicAdd(S.pstat + 2, (unsigned long long)pk_batches); atomicAdd(S.pstat + 3, (unsigned long long)pk_surv);
    }
    if (inq) {
        L.t[i] = h.nearest;
        L.prim[i] = (h.best < 0) ? -1 : (h.best | h.tflag);
    }
}


// Closest hit for the generations after the first: reflection and refraction rays of neighbouring pixels
// leave curved surfaces in all directions, a wave of them is no packet (measured on the benchmark scene: the
// packet test kept 1,171 of 10,000 primitives per wave at level 1, against 17 at level 0), so every RAY walks
// the cluster hierarchy (closest_pool below).  Its candidates arrive in cluster order, not file order, so the
// reference's sequential scan (:385-418) is restated order-independently.  With H = the accepted hits (t_k, k):
//   * nearest/best = the lexicographic minimum of (t_k, k) over hits with t_k < Max_t: the scan keeps
//     the first of equal t's because it updates on strict `<` only;
//   * a triangle that wins resets the `transformed` flag and only triangles follow it, so the flag is
//     0 when best is a triangle; otherwise it is 1 iff some affine sphere k was a record when the scan
//     reached it: t_k < Max_t and t_j > t_k for every hit j < k.  Affine-sphere clusters are traversed
//     FIRST; their hits form a list in which (j < k, t_j <= t_k) removes k, and every later hit of a
//     plain sphere removes entries the same way.  The flag is "list not empty".  The list holds the
//     staircase of affine hits (index up, t down), 4 entries; a fifth sends the ray to the ordered
//     scan over every primitive at the end (ordered_scan above).
// Rays the filter does not cover (ray_abnormal) take every cluster and every member.
constexpr int kAffList = 4;

struct LexAcc {
    float nearest; int best;
    float at[kAffList]; int ak[kAffList];
    bool overflow;
};

// exact test of primitive k (per lane): 0 miss | 1 plain sphere | 2 triangle or polygon | 3 affine sphere, t set on a hit.
// The primitive's 80-byte hot record (SceneDev::phot) comes in with five independent loads: one memory
// round trip, where sph_hot -> sph_xf -> sph_im was a chain of three.
__device__ __forceinline__ int exact_hit(const SceneDev& S, int k, F3 o, F3 d, F3 nd, float& t, Exec* ex = nullptr)
{
    const float4* p = S.phot + (size_t)k * 5;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
    if (k < S.ns) {
        const bool xf = q4.x != 0.0f;
        ex_add(ex, xf ? EX_SPHX : EX_SPH);
        F3 to = o, td = nd;
        if (xf) {
            rtx_xmatrix im;
            im.a.x = q1.x; im.a.y = q1.y; im.a.z = q1.z; im.b.x = q1.w;
            im.b.y = q2.x; im.b.z = q2.y; im.c.x = q2.z; im.c.y = q2.w;
            im.c.z = q3.x; im.d.x = q3.y; im.d.y = q3.z; im.d.z = q3.w;
            to = inv_apply(o, im);
            td = normalized(inv_apply(d, im));
        }
        return sphere_test(to, td, q0.x, q0.y, q0.z, q0.w, t) ? (xf ? 3 : 1) : 0;
    } else if (k >= S.ns + S.nt) {
        ex_add(ex, EX_POLY, (unsigned)max(__float_as_int(q1.z), 1));
        return polygon_test(o, d, q0, q1, S.poly_verts, t) ? 2 : 0;
    } else {
        TriRec T;
        T.nx = q0.x; T.ny = q0.y; T.nz = q0.z; T.d = q0.w;
        T.vx = q1.x; T.vy = q1.y; T.vz = q1.z; T.area = q1.w;
        T.wx = q2.x; T.wy = q2.y; T.wz = q2.z; T.v1x = q2.w;
        T.v1y = q3.x; T.v1z = q3.y; T.v2x = q3.z; T.v2y = q3.w;
        T.v2z = q4.x; T.ex = q4.y; T.ey = q4.z; T.ez = q4.w;
        float vd;
        int exitp = 0;
        const bool hit = triangle_test(o, d, T, t, vd, ex ? &exitp : nullptr, S.mt != 0);
        if (ex) ex_tri(ex, exitp);
        return hit ? 2 : 0;
    }
}

// exact_hit with the triangle's edge filter and a bound on t (a triangle hit at t >= tmax is a miss before its three square roots)
__device__ __forceinline__ int exact_hit_edge(const SceneDev& S, int k, F3 o, F3 d, F3 nd, float& t, const float tmax, Exec* ex)
{
    if (k < S.ns || k >= S.ns + S.nt) return exact_hit(S, k, o, d, nd, t, ex);
    const float4* p = S.phot + (size_t)k * 5;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
    const float4 e0 = S.tri_edge[(size_t)(k - S.ns) * 2], e1 = S.tri_edge[(size_t)(k - S.ns) * 2 + 1];
    TriRec T;
    T.nx = q0.x; T.ny = q0.y; T.nz = q0.z; T.d = q0.w;
    T.vx = q1.x; T.vy = q1.y; T.vz = q1.z; T.area = q1.w;
    T.wx = q2.x; T.wy = q2.y; T.wz = q2.z; T.v1x = q2.w;
    T.v1y = q3.x; T.v1z = q3.y; T.v2x = q3.z; T.v2y = q3.w;
    T.v2z = q4.x; T.ex = q4.y; T.ey = q4.z; T.ez = q4.w;
    float vd;
    int exitp = 0;
    const bool hit = triangle_test_edge(o, d, T, e0, e1, t, vd, tmax, ex ? &exitp : nullptr, S.mt != 0);
    if (ex) ex_tri(ex, exitp);
    return hit ? 2 : 0;
}

// the order-independent update of a ray's state with the hit (k, t) of kind 1 / 2 / 3 (see above)
__device__ __forceinline__ void lex_update(LexAcc& h, int k, float t, int kind)
{
    if (t < h.nearest || (t == h.nearest && k < h.best)) { h.nearest = t; h.best = k; }
    if (kind == 3) {
        if (t < kMaxT) {
            bool killed = false;
#pragma unroll
            for (int e = 0; e < kAffList; e++) killed = killed || (h.ak[e] >= 0 && h.ak[e] < k && h.at[e] <= t);
            if (!killed) {
                bool placed = false;
#pragma unroll
                for (int e = 0; e < kAffList; e++) {
                    if (h.ak[e] >= 0 && k < h.ak[e] && t <= h.at[e]) h.ak[e] = -1;
                    if (!placed && h.ak[e] < 0) { h.ak[e] = k; h.at[e] = t; placed = true; }
                }
                if (!placed) h.overflow = true;
            }
        }
    } else if (kind == 1) {
#pragma unroll
        for (int e = 0; e < kAffList; e++)
            if (h.ak[e] >= 0 && k < h.ak[e] && t <= h.at[e]) h.ak[e] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// Pooled traversal: the WORK, not the rays, is spread over the lanes.
//
// Giving each lane its own ray to walk (the first version of this kernel) makes every loop run as long as the
// wave's busiest lane needs: measured on the benchmark frame, 147 member steps per wave where the average lane
// needs 55, each step a round trip of per-lane gathers.  Here a wave owns R rays (64, or 16 for
// generations too small to fill the chip) whose data sit in LDS, and every stage is a list of independent
// (ray, item) pairs in LDS that the 64 lanes consume 64 at a time, whoever the rays belong to:
//   A  (ray, cluster sphere)          cone test                 64/R lanes per ray, kPoolQ clusters a round
//   B  (ray, passing cluster)         its 8 sub-cluster spheres  8 lanes per pair   -> (ray, cluster, group)
//   C  (ray, cluster, passing group)  its 8 members' filter      8 lanes per pair   -> (ray, primitive)
//   D  (ray, primitive)               the exact test             1 lane per pair    -> hits
//   E  hits are handed to the lane that owns the ray (one per ray per step) and folded into its LexAcc.
// Loads are coalesced (8 lanes read 128 consecutive bytes), a list's addresses are known ahead so several
// steps' loads are in flight together, and no lane idles while another works.  What is tested is what a
// per-lane walk tests -- the same cone tests on the same spheres, the same filter, every surviving
// (ray, primitive) pair through the same exact test -- and LexAcc does not depend on the order of arrival,
// except that affine-sphere clusters come first (phase 0 is drained before phase 1 starts).
// ------------------------------------------------------------------------------------------------
constexpr int kPoolPrimBits = 26;  // (ray, primitive) entries: primitive | ray << 26 (rtx_context_create checks the scene fits)
constexpr int kPoolB = 512;        // (ray, cluster) pairs of one round: R * clusters per round
constexpr int kPoolQMax = 128;     // clusters per round at most (their spheres are staged in LDS)
constexpr int kPoolC = 64 + 64;    // (ray, cluster, group) entries: fewer than 64 wait, one push adds up to 64
constexpr int kPoolD = 64 + 64;    // (ray, primitive) entries: the same

struct PoolLds {
    float4 ray[3 * 64];            // R rays: [0, R) origin, w = flag bits (1 live, 2 abnormal); [R, 2R) normalised direction, w = the
                                   // ray's far limit; [2R, 3R) direction as stored (the exact tests take both).  Forms with fewer than
                                   // 64 rays keep the round's cluster spheres { c, R } and cluster numbers behind their rays:
                                   // R = 16: spheres [48, 80), numbers [80, 88);  R = 4: numbers [16, 48), spheres [64, 192).
                                   // (5,696 bytes a wave: seven workgroups a CU where 6,912 gave five)
    unsigned short listB[kPoolB];  // ray << 7 | cluster slot of the round
    unsigned int listC[kPoolC];    // cluster << 9 | ray << 3 | group
    unsigned int listD[kPoolD];    // ray << 26 | primitive
    signed char slot[64];          // E: which lane delivers to ray r this step (-1 none)
    int inK[64];                   // kind << 28 | primitive
    float inT[64];
};

__device__ __forceinline__ void lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lanes_below(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <int R>
__device__ __forceinline__ void closest_pool(const SceneDev& S, const LevelDev& L, const int n, const int q0, PoolLds& W, Exec* ex)
{
    constexpr int kPer = 64 / R;                                  // lanes per ray in stage A
    constexpr int kPoolQ = (kPoolB / R < kPoolQMax) ? kPoolB / R : kPoolQMax;   // clusters per round: 8 (R = 64), 32 (R = 16), 128 (R = 4)
    const int lane = threadIdx.x & 63;
    const int myray = lane & (R - 1), part = lane / R;
    float4* const rayO = W.ray; float4* const rayN = W.ray + R; float4* const rayD = W.ray + 2 * R;
    float4* const csph = (R == 4) ? W.ray + 64 : W.ray + 48;
    int* const ccid = reinterpret_cast<int*>((R == 4) ? W.ray + 16 : W.ray + 80);
    static_assert(R == 64 || (R == 16 && kPoolQ == 32) || (R == 4 && kPoolQ == 128), "the round's cluster spheres and numbers fit behind the rays");
    // ---- the wave's rays: lanes < R own one each ----
    const int q = q0 + lane;
    const bool inq = lane < R && q < n;
    const int i = inq ? q : 0;
    F3 o = f3(L.ox[i], L.oy[i], L.oz[i]);
    F3 d = f3(L.dx[i], L.dy[i], L.dz[i]);
    F3 nd = normalized(d);
    const bool live = inq && !ray_dead(o, d);  // NaN rays (total internal reflection) miss everything: no traversal
    const bool abn = live && ray_abnormal(o, d, nd, S.guard);
    if (inq) ex_add(ex, EX_RAYS);
    if (lane < R) {
        rayO[lane] = make_float4(o.x, o.y, o.z, __int_as_float((live ? 1 : 0) | (abn ? 2 : 0)));
        rayN[lane] = make_float4(nd.x, nd.y, nd.z, __builtin_inff());  // .w: the ray's far limit (cone_slack_cull), phase 1
        rayD[lane] = make_float4(d.x, d.y, d.z, 0.0f);
    }
    W.slot[lane] = -1;
    lds_fence();
    const float4 aO = rayO[myray], aN = rayN[myray];  // stage A's ray of this lane
    const F3 ao = f3(aO.x, aO.y, aO.z), an = f3(aN.x, aN.y, aN.z);
    const bool alive = (__float_as_int(aO.w) & 1) != 0, aabn = (__float_as_int(aO.w) & 2) != 0;
    LexAcc h;
    h.nearest = kMaxT; h.best = -1; h.overflow = false;
#pragma unroll
    for (int e = 0; e < kAffList; e++) { h.ak[e] = -1; h.at[e] = 0.0f; }
    int nC = 0, nD = 0, tested = 0;
    bool front = false;  // phase 1 (no affine spheres): primitives behind the ray's origin are skipped

    // What a hit beyond L cannot change (phase 1: plain spheres, triangles, polygons): the nearest hit needs t < nearest or
    // t == nearest with a lower index, an affine entry of the staircase is killed by a plain sphere at t <= its own; so
    // L = max(nearest, the live entries' t).  L only falls as hits arrive, a stale value in LDS is a larger one.
    const float cull_abs = 1e-5f * S.guard;
    // The ray's own parent first: a ray that left a PLAIN sphere (LevelDev::src) -- refracted into it, or reflected inside it -- hits
    // that sphere again a chord away, and the traversal will find that hit when it reaches the sphere's cluster; knowing its t now
    // bounds the nearest hit from the start (it is the same test on the same operands, so the same t), and phase 1 culls beyond it
    // from its first step.  Not a hit yet: LexAcc takes it in phase 1 like any other (plain spheres must not act before the affine
    // spheres' hits are in).  Rays that left convex spheres outwards miss it: one exact test spent.
    float tself = kMaxT;
    if (lane < R && live && !abn) {
        const int ks = L.src[i];
        float ts = 0.0f;
        if (ks >= 0 && exact_hit(S, ks, o, d, nd, ts, ex) == 1 && ts < kMaxT) tself = ts;
    }
    auto far_limit = [&]() {
        float lim = fminf(h.nearest, tself);
#pragma unroll
        for (int e = 0; e < kAffList; e++) lim = (h.ak[e] >= 0) ? fmaxf(lim, h.at[e]) : lim;
        return __builtin_fmaf(lim, 1.015625f, cull_abs);
    };
    // E: hand the hits of a D step to the owners of their rays, one hit per ray per round
    auto deliver = [&](bool pending, int rr, int k, float t, int kind) {
        while (__any(pending)) {
            if (pending) W.slot[rr] = (signed char)lane;
            lds_fence();
            const bool win = pending && W.slot[rr] == lane;
            if (win) { W.inK[rr] = k | (kind << 28); W.inT[rr] = t; pending = false; }
            lds_fence();
            if (lane < R && W.slot[lane] >= 0) {
                const int e = W.inK[lane];
                lex_update(h, e & ((1 << 28) - 1), W.inT[lane], e >> 28);
                W.slot[lane] = -1;
                if (front) rayN[lane].w = far_limit();
            }
            lds_fence();
        }
    };
    // D: exact tests of the last `cnt` (<= 64) list entries, one pair per lane
    auto run_D = [&](int cnt) {
        const int base = nD - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listD[base + (has ? lane : 0)];
        const int k = (int)(e & ((1u << kPoolPrimBits) - 1u));
        const int rr = (int)(e >> kPoolPrimBits);
        const float4 rO = rayO[rr], rN = rayN[rr], rD = rayD[rr];
        float t = 0.0f;
        int kind = 0;
        // (triangles through the edge filter, and none beyond the ray's far limit rN.w -- infinite in phase 0: a sliver's bounding
        // sphere passes rays far from the sliver, 186 full tests per secondary ray in a scene of long triangles, 3.5 with this)
        if (has) kind = exact_hit_edge(S, k, f3(rO.x, rO.y, rO.z), f3(rD.x, rD.y, rD.z), f3(rN.x, rN.y, rN.z), t, rN.w, ex);
        tested += cnt;
        nD = base;
        deliver(kind != 0, rr, k, t, kind);
    };
    // C: one (ray, cluster, group) per lane: its 8 members against the ray's filter
    auto run_C = [&](int cnt) {
        const int base = nC - cnt;
        const bool has = lane < cnt;
        const unsigned e = W.listC[base + (has ? lane : 0)];
        const int rr = (int)((e >> 3) & 63u);
        const size_t m = (size_t)(e >> 9) * 64 + (e & 7u) * 8;
        float4 mb[kGroup];
#pragma unroll
        for (int v = 0; v < kGroup; v++) mb[v] = S.cmem[m + v];
        const int4 k0 = *reinterpret_cast<const int4*>(S.cidx + m), k1 = *reinterpret_cast<const int4*>(S.cidx + m + 4);
        const int kk[kGroup] = { k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w };
        const float4 rO = rayO[rr], rN = rayN[rr];
        const bool use = (__float_as_int(rO.w) & 2) == 0;
        const F3 fo = use ? f3(rO.x, rO.y, rO.z) : f3(1e18f, 0.0f, 0.0f);
        const F3 fd = use ? f3(rN.x * kDirScale, rN.y * kDirScale, rN.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
        nC = base;
        unsigned pmask = 0u;
#pragma unroll
        for (int v = 0; v < kGroup; v++) {
            if (has && use && kk[v] >= 0) ex_add(ex, EX_FILT);
            if (kk[v] >= 0 && (!use || (front ? filt_front(fo, fd, mb[v]) : !(filt(fo, fd, mb[v]) < 0.0f)))) pmask |= 1u << v;
        }
        if (!has) pmask = 0u;
        // every lane pushes its lowest passing member, as many times as the busiest lane has members (0.35 of 8 pass)
        for (;;) {
            const bool pass = pmask != 0u;
            const unsigned long long bal = __ballot(pass);
            if (bal == 0ull) break;  // wave-uniform
            const int v = __builtin_ctz(pmask | 0x100u);
            int kv = kk[0];
#pragma unroll
            for (int u = 1; u < kGroup; u++) kv = (u == v) ? kk[u] : kv;  // (a select chain: no dynamically indexed register array)
            if (pass) W.listD[nD + lanes_below(bal)] = ((unsigned)rr << kPoolPrimBits) | (unsigned)kv;
            pmask &= pmask - 1u;
            nD += __popcll(bal);
            lds_fence();
            if (nD >= 64) run_D(64);
        }
    };
    // B: one (ray, cluster) per lane: its 8 sub-cluster spheres against the ray's cone
    auto do_B = [&](const bool has, const int rr, const int cid) {
        const float4* srec = S.csubR + (size_t)cid * 8;
        float4 sb[8];
#pragma unroll
        for (int v = 0; v < 8; v++) sb[v] = srec[v];
        const float4 rO = rayO[rr], rN = rayN[rr];
        const bool rabn = (__float_as_int(rO.w) & 2) != 0;
        const F3 ro = f3(rO.x, rO.y, rO.z), rn = f3(rN.x, rN.y, rN.z);
        unsigned gmask = 0u;
#pragma unroll
        for (int v = 0; v < 8; v++)
            if (rabn || !((front ? cone_slack_cull(ro, rn, sb[v], rN.w) : cone_slack(ro, rn, sb[v])) < 0.0f)) gmask |= 1u << v;
        if (has && !rabn) ex_add(ex, EX_CONE, 8u);
        if (!has) gmask = 0u;
        // every lane pushes its lowest passing group, as many times as the busiest lane has groups (1.2 of 8 pass on the
        // benchmark frame: 3 or 4 rounds where a loop over the 8 groups took 8); a C step as soon as 64 entries wait.
        // The lanes are in cluster order, so the 64 entries of a C step still name few clusters' members.
        for (;;) {
            const bool pass = gmask != 0u;
            const unsigned long long bal = __ballot(pass);
            if (bal == 0ull) break;  // wave-uniform
            const unsigned v = (unsigned)__builtin_ctz(gmask | 0x100u);
            if (pass) W.listC[nC + lanes_below(bal)] = ((unsigned)cid << 9) | ((unsigned)rr << 3) | v;
            gmask &= gmask - 1u;
            nC += __popcll(bal);
            lds_fence();
            if (nC >= 64) run_C(64);
        }
    };
    auto run_B = [&](int base, int cnt) {
        const bool has = lane < cnt;
        const unsigned e = W.listB[base + (has ? lane : 0)];
        do_B(has, (int)(e >> 7), ccid[e & 127u]);
    };

    const int ncl = S.cseg[S.cnseg];
    static_assert(kPoolQ % kSuperRun == 0, "a round of stage A is a whole number of ClusterSet::supR runs");
    const int nsup0 = (S.cnfirst + kSuperRun - 1) / kSuperRun;
    if constexpr (R == 64) {
      // ---- 64 rays a wave: the RUN level is pooled too (round 3).  A ray passes ~6 of C4's 20 runs of 8 clusters; testing every
      //      cluster of every round for every ray (lanes in lockstep: a round cost its 8 cone tests whenever ANY ray passed its run)
      //      was 157 of the 225 cone tests per ray.  Now a lane tests its ray against the run's sphere only, the passing (ray, run)
      //      pairs collect in LDS in run order, and whenever 64 wait each lane takes one pair and tests its 8 cluster spheres. ----
      unsigned* const listR = reinterpret_cast<unsigned*>(W.listB);          // [128]  ray << 20 | run
      unsigned* const listB32 = reinterpret_cast<unsigned*>(W.listB) + 128;  // [128]  ray << 26 | cluster
      static_assert(kPoolB * sizeof(unsigned short) >= 256 * sizeof(unsigned), "the two lists fit listB's space");
      constexpr unsigned kPrimMaskB = (1u << kPoolPrimBits) - 1u;
      int nR = 0, nB = 0;
      int cendv = 0, cbegv = 0, sbase = 0;
      auto run_B32 = [&](const int want) {
          const int cnt = min(want, nB), base = nB - cnt;
          const bool has = lane < cnt;
          const unsigned e = listB32[base + (has ? lane : 0)];
          nB = base;
          do_B(has, (int)(e >> kPoolPrimBits), (int)(e & kPrimMaskB));
      };
      auto run_R = [&](const int want) {
          const int cnt = min(want, nR), base = nR - cnt;
          const bool has = lane < cnt;
          const unsigned e = listR[base + (has ? lane : 0)];
          nR = base;
          const int rr = (int)(e >> 20);
          const int p0 = cbegv + ((int)(e & 0xfffffu) - sbase) * kSuperRun;  // the run's first cluster in traversal order
          const int pc = min(kSuperRun, cendv - p0);
          const float4 rO = rayO[rr], rN = rayN[rr];
          const bool rabn = (__float_as_int(rO.w) & 2) != 0;
          const F3 ro = f3(rO.x, rO.y, rO.z), rn = f3(rN.x, rN.y, rN.z);
          int cid[kSuperRun];
          float4 sb[kSuperRun];
#pragma unroll
          for (int v = 0; v < kSuperRun; v++) cid[v] = S.corder[min(p0 + v, cendv - 1)];
#pragma unroll
          for (int v = 0; v < kSuperRun; v++) sb[v] = S.cclRo[min(p0 + v, cendv - 1)];  // (in traversal order: no load waits on another)
          unsigned cm = 0u;
#pragma unroll
          for (int v = 0; v < kSuperRun; v++)
              if (v < pc && (rabn || !((front ? cone_slack_cull(ro, rn, sb[v], rN.w) : cone_slack(ro, rn, sb[v])) < 0.0f))) cm |= 1u << v;
          if (has && !rabn) ex_add(ex, EX_CONE, (unsigned)pc);
          if (!has) cm = 0u;
          // every lane pushes its lowest passing cluster, as many times as the busiest lane has clusters
          // (a loop, not unrolled: every copy of its body holds a whole B -> C -> D nest)
          for (;;) {
              const bool pass = cm != 0u;
              const unsigned long long bal = __ballot(pass);
              if (bal == 0ull) break;  // wave-uniform
              const int v = __builtin_ctz(cm | 0x100u);
              int cv = cid[0];
#pragma unroll
              for (int u = 1; u < kSuperRun; u++) cv = (u == v) ? cid[u] : cv;  // (a select chain: no dynamically indexed register array)
              if (pass) listB32[nB + lanes_below(bal)] = ((unsigned)rr << kPoolPrimBits) | (unsigned)cv;
              cm &= cm - 1u;
              nB += __popcll(bal);
              lds_fence();
              if (nB >= 64) run_B32(64);
          }
      };
      for (int phase = 0; phase < 2; phase++) {
          cbegv = phase ? S.cnfirst : 0; cendv = phase ? ncl : S.cnfirst;
          sbase = phase ? nsup0 : 0;
          front = phase == 1;
          const int nruns = (cendv - cbegv + kSuperRun - 1) / kSuperRun;
          for (int sr = 0; sr < nruns; sr++) {
              const float4 sup = S.csupR[sbase + sr];
              if (alive && !aabn) ex_add(ex, EX_CONE);
              const bool pass = alive && (aabn || !((front ? cone_slack_cull(ao, an, sup, far_limit()) : cone_slack(ao, an, sup)) < 0.0f));
              const unsigned long long bal = __ballot(pass);
              if (bal == 0ull) continue;  // wave-uniform: no ray of the wave passes the run
              if (pass) listR[nR + lanes_below(bal)] = ((unsigned)lane << 20) | (unsigned)(sbase + sr);
              nR += __popcll(bal);
              lds_fence();
              if (nR >= 64) run_R(64);
          }
          // phase 0: every affine-sphere hit is known before the first plain sphere is tested
          while (nR > 0) run_R(nR);
          while (nB > 0) run_B32(nB);
          if (nC > 0) run_C(nC);
          if (nD > 0) run_D(nD);
          if (lane < R) rayN[lane].w = far_limit();  // phase 1 culls beyond what phase 0 found
          lds_fence();
      }
    } else {
    float alr = __builtin_inff();
    for (int phase = 0; phase < 2; phase++) {
          const int cbeg = phase ? S.cnfirst : 0, cend = phase ? ncl : S.cnfirst;
          front = phase == 1;
          for (int cb = cbeg; cb < cend; cb += kPoolQ) {
              const int ce = min(kPoolQ, cend - cb);
              // ---- the spheres around the round's runs of 8 clusters first: a wave none of whose rays can pass one
              //      skips the round ----
              {
                  const int s0 = (phase ? nsup0 : 0) + (cb - cbeg) / kSuperRun, sn = (ce + kSuperRun - 1) / kSuperRun;
                  bool spass = aabn;
                  alr = rayN[myray].w;  // this round's far limit of the lane's ray (its owner lowers it as hits arrive)
                  for (int v = 0; v < sn; v++) {
                      const float4 sup = S.csupR[s0 + v];
                      if (alive && !aabn) ex_add(ex, EX_CONE);
                      spass = spass || !((front ? cone_slack_cull(ao, an, sup, alr) : cone_slack(ao, an, sup)) < 0.0f);
                  }
                  if (!__any(alive && spass)) continue;  // wave-uniform
              }
              // ---- A: this round's cluster spheres into LDS, each lane tests its ray against its share of them ----
              for (int u = lane; u < ce; u += 64) {
                  ccid[u] = S.corder[cb + u];
                  csph[u] = S.cclRo[cb + u];
              }
              lds_fence();
              // (bit i of cmask: this lane's i-th cluster of the round, slot part + i * kPer; at most kPoolQ / kPer <= 32 of them)
              unsigned cmask = 0u;
              for (int u = part, it = 0; u < ce; u += kPer, it++) {
                  const float4 b = csph[u];
                  if (alive && !aabn) ex_add(ex, EX_CONE);
                  if (aabn || !((front ? cone_slack_cull(ao, an, b, alr) : cone_slack(ao, an, b)) < 0.0f)) cmask |= 1u << it;
              }
              if (!alive) cmask = 0u;
              // the pairs that passed, as a list: neighbouring lanes of a B step should read the same cluster's sub-spheres
              // (one cache line for the lot, where a list in ray order touches 64)
              // (every lane pushes its lowest passing cluster, as many times as the busiest lane has clusters -- 2 or 3 rounds
              // where a loop over the round's clusters took 32 or 128; the R lanes of a part test the same cluster, so
              // neighbouring entries of a push still name one cluster)
              int nB = 0;
              for (;;) {
                  const bool bit = cmask != 0u;
                  const unsigned long long bal = __ballot(bit);
                  if (bal == 0ull) break;  // wave-uniform
                  const int u = part + __builtin_ctz(cmask | 0x80000000u) * kPer;
                  if (bit) W.listB[nB + lanes_below(bal)] = (unsigned short)((myray << 7) | u);
                  cmask &= cmask - 1u;
                  nB += __popcll(bal);
              }
              lds_fence();
              for (int p0 = 0; p0 < nB; p0 += 64) run_B(p0, min(64, nB - p0));
          }
          // phase 0: every affine-sphere hit is known before the first plain sphere is tested
          if (nC > 0) run_C(nC);
          if (nD > 0) run_D(nD);
          if (lane < R) rayN[lane].w = far_limit();  // phase 1 culls beyond what phase 0 found
          lds_fence();
      }
    }
    bool aliveList = false;
#pragma unroll
    for (int e = 0; e < kAffList; e++) aliveList = aliveList || h.ak[e] >= 0;
    float nearest = h.nearest;
    int best = h.best;
    int tflag = (best >= 0 && best < S.ns && aliveList) ? kTFlag : 0;
    const bool slow = live && h.overflow;
    ordered_scan(S, slow, abn, o, d, nd, nearest, best, tflag, ex);
    if (S.pstat) {
        const unsigned long long sl = __ballot(slow);
        const unsigned long long ab = __ballot(abn);
        if (lane == 0) {
            atomicAdd(S.pstat + 0, 1ull); atomicAdd(S.pstat + 1, (unsigned long long)__popcll(sl) + ((unsigned long long)__popcll(ab) << 32));
            atomicAdd(S.pstat + 3, (unsigned long long)tested);
        }
    }
    if (inq) {
        L.t[i] = nearest;
        L.prim[i] = (best < 0) ? -1 : (best | tflag);
    }
}

// Thin generations (deep bounces: few, incoherent rays).  A workgroup takes only 64 rays and its
// four waves split the primitive list into four contiguous slices, so a generation of a few
// thousand rays still puts a wave on every SIMD and the serial walk per wave is 4x shorter.
// The slices are merged in index order, which reproduces the sequential scan exactly: slice p's
// nearest hit wins iff it is strictly nearer than everything before it, and the reference's sticky
// `transformed` flag (:396-399) is set iff some slice's last affine-sphere record beats the minimum
// of the slices before it.
constexpr int kThinRays = 1 << 18;
constexpr int kThinTile = kTile / 4;

__device__ __forceinline__ int stage_thin(float4* mt, const float4* bounds, int t0, int end, int lane)
{
    int cnt = max(0, min(kThinTile, end - t0));
    for (int j = lane; j < cnt; j += 64) mt[j] = bounds[t0 + j];
    int padded = (cnt + kGroup - 1) / kGroup * kGroup;
    for (int j = cnt + lane; j < padded; j += 64) mt[j] = make_float4(0.0f, 0.0f, 0.0f, __builtin_inff());
    return cnt;
}

__device__ __forceinline__ void closest_thin(const SceneDev& S, const LevelDev& L, const int n, float4* tile, Exec* ex)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;  // the four waves hold the same 64 rays
    const bool inq = i < n;
    const int ii = inq ? i : 0;
    F3 o = f3(L.ox[ii], L.oy[ii], L.oz[ii]);
    F3 d = f3(L.dx[ii], L.dy[ii], L.dz[ii]);
    F3 nd = normalized(d);
    const bool live = inq && !ray_dead(o, d);
    const bool abn = ray_abnormal(o, d, nd, S.guard);
    const bool use = live && !abn;
    if (inq && w == 0) ex_add(ex, EX_RAYS);
    F3 fo = use ? o : f3(1e18f, 0.0f, 0.0f);
    F3 fd = use ? f3(nd.x * kDirScale, nd.y * kDirScale, nd.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
    unsigned accinit = (live && abn) ? 0u : 0xFFFFFFFFu;
    asm volatile("" : "+v"(accinit));
    HitAcc h;
    h.nearest = kMaxT; h.best = -1; h.tflag = 0; h.txf = kMaxT;
    const int nprim = S.ns + S.nt + S.np;
    const int SL = ((nprim + kGroup - 1) / kGroup + 3) / 4 * kGroup;  // primitives per slice, whole groups
    const int p0 = min(w * SL, nprim), p1 = min(p0 + SL, nprim);
    const int T = (SL + kThinTile - 1) / kThinTile;
    float4* mt = tile + w * kThinTile;
    for (int tt = 0; tt < T; tt++) {
        __syncthreads();
        const int t0 = p0 + tt * kThinTile;
        const int cnt = stage_thin(mt, S.bounds, t0, p1, lane);
        __syncthreads();
        for (int g = 0; g < cnt; g += kGroup) {
            unsigned acc = accinit;
#pragma unroll
            for (int u = 0; u < kGroup; u++) acc &= __float_as_uint(filt(fo, fd, mt[g + u]));
            if (use) ex_add(ex, EX_FILT, (unsigned)min(kGroup, cnt - g));
            if ((int)acc >= 0) {
                int gend = min(kGroup, cnt - g);
                for (int u = 0; u < gend; u++)
                    if (live && (abn || !(filt(fo, fd, mt[g + u]) < 0.0f))) exact_closest(S, t0 + g + u, o, d, nd, h, ex);
            }
        }
    }
    __syncthreads();
    float* rt = reinterpret_cast<float*>(tile);
    int* rb = reinterpret_cast<int*>(tile) + 256;
    float* rx = reinterpret_cast<float*>(tile) + 512;
    rt[threadIdx.x] = h.nearest; rb[threadIdx.x] = h.best; rx[threadIdx.x] = h.txf;
    __syncthreads();
    if (w == 0 && inq) {
        float m = kMaxT;
        int best = -1, flag = 0;
        for (int p = 0; p < 4; p++) {
            if (rx[p * 64 + lane] < m) flag = kTFlag;
            float tp = rt[p * 64 + lane];
            if (tp < m) { m = tp; best = rb[p * 64 + lane]; }
        }
        if (best >= S.ns) flag = 0;  // a triangle won: :416
        L.t[i] = m;
        L.prim[i] = (best < 0) ? -1 : (best | flag);
    }
}

template <bool PREFETCH, bool EX = false>
__global__ __launch_bounds__(kBlock) void k_closest_f(SceneDev S, LevelDev L, const int* nrays)
{
    __shared__ float4 tile[kTile];  // 16 KiB: thin tiles / the four waves' bitmaps (2 KiB) and candidate lists (2 KiB)
    static_assert((kSegPrims / 32 + 512) * 4 * (kBlock / 64) <= kTile * 16, "bitmaps and lists fit");
    const int n = min(*nrays, L.cap);
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
    if (n > kThinRays) {
        const int base = blockIdx.x * kBlock;
        if (base >= n) return;  // block-uniform
        closest_packet<PREFETCH>(S, L, n, base, tile, ex);
    } else {
        if (blockIdx.x * 64 >= n) return;
        closest_thin(S, L, n, tile, ex);
    }
    ex_flush(ex, S.exec + 0 * EX_KINDS);
}

// Generations of at most kSplitClosest rays give a wave 16 rays instead of 64 (64 rays per workgroup): a thin
// generation has fewer waves than the chip has SIMDs, its run time is one wave's serial walk, and the pooled
// stages keep all 64 lanes busy whatever the number of rays.
#ifndef RTX_SPLIT_CLOSEST_LOG2
#define RTX_SPLIT_CLOSEST_LOG2 17
#endif
constexpr int kSplitClosest = 1 << RTX_SPLIT_CLOSEST_LOG2;
// ... and 4 rays (16 lanes a ray in stage A, 128 clusters a round: two rounds where 16 rays take five) when even that leaves
// most SIMDs without a wave: the deep generations of a rank's share hold a few thousand rays and cost a full per-wave chain each
#ifndef RTX_SPLIT_CLOSEST4_LOG2
#define RTX_SPLIT_CLOSEST4_LOG2 15
#endif
constexpr int kSplitClosest4 = 1 << RTX_SPLIT_CLOSEST4_LOG2;
constexpr size_t kPrefetchRays = (size_t)1 << 20;  // generation-0 frames below this take the latency-oriented walk (k_closest_f<true>)

#ifndef RTX_WAVES_LANES
#define RTX_WAVES_LANES 5
#endif
#ifdef RTX_WAVE_CLOCK
constexpr unsigned kWaveClockCap = 1u << 17;
__device__ unsigned long long g_wclk[3 * kWaveClockCap];
__device__ unsigned int g_wclk_n;
#endif
template <bool EX = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RTX_WAVES_LANES))) void k_closest_lanes(SceneDev S, LevelDev L, const int* nrays, int split16, int split4)
{
    __shared__ PoolLds pool[kBlock / 64];
    const int n = min(*nrays, L.cap);
    const int w = threadIdx.x >> 6;
    Exec exs;
    Exec* ex = EX ? &exs : nullptr;
    ex_zero(ex);
#ifdef RTX_WAVE_CLOCK
    // diagnosis build (tools/dbg/wave_clock.py): every wave that traces rays leaves { rays of the launch, start, end } in 10 ns ticks
    struct WaveClock {
        unsigned long long t0; int n; bool on;
        __device__ ~WaveClock()
        {
            if (!on || (threadIdx.x & 63) != 0) return;
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            const unsigned slot = atomicAdd(&g_wclk_n, 1u);
            if (slot < kWaveClockCap) { g_wclk[3 * slot] = (unsigned long long)n; g_wclk[3 * slot + 1] = t0; g_wclk[3 * slot + 2] = t1; }
        }
    } wc{__builtin_amdgcn_s_memrealtime(), n, false};
    {
        const int per = (n <= split4) ? 16 : (n <= split16) ? 64 : kBlock;
        wc.on = (long long)blockIdx.x * per + (long long)w * (per / 4) < n;
    }
#endif
    if (n <= split4) {
        if (blockIdx.x * 16 >= n) return;  // block-uniform
        closest_pool<4>(S, L, n, blockIdx.x * 16 + w * 4, pool[w], ex);
    } else if (n <= split16) {
        if (blockIdx.x * 64 >= n) return;  // block-uniform
        closest_pool<16>(S, L, n, blockIdx.x * 64 + w * 16, pool[w], ex);
    } else {
        if (blockIdx.x * kBlock >= n) return;
        closest_pool<64>(S, L, n, blockIdx.x * kBlock + w * 64, pool[w], ex);
    }
    ex_flush(ex, S.exec + 1 * EX_KINDS);
}


}  // namespace rtx
#endif
