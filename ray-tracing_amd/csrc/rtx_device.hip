// rtx_device.hip -- gfx950 kernels and the resident render context behind include/rtx.h.
//
// The reference's recursive raytrace() (raytracing.cpp:371-647) becomes an iterative wavefront
// over ray generations ("levels"):
//
//   raygen                       level-0 rays, one thread per pixel            (:753-759)
//   for level k = 0 .. max_depth
//     closest_hit(k)             spheres, triangles, polygons in file order     (:382-434)
//                                level 0: packets of 64 coherent rays against a cluster hierarchy;
//                                level >= 1: every ray walks the hierarchy itself; both reproduce the
//                                reference's sequential scan exactly
//     hit_setup(k)               hit point / normal, spawn reflect + refract
//                                rays into level k+1 (queue slots per workgroup) (:436-526)
//     shadow_any(k)              one any-hit ray per (hit, light), sorted by light and direction,
//                                on a side stream                               (:560-589)
//   for level k = max_depth .. 0
//     unwind(k)                  children first, then lights in file order,
//                                then ambient: the reference's add order        (:446-646)
//   aa_quantise                  3x3 weighted filter + 8-bit quantisation       (:738-783,:792-794)
//
// Queue sizes live in device memory; every launch covers the queue's capacity and threads past
// the live count exit, so a frame is enqueued without a single host round trip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rtx.h"
#include "rtx_math.hip.h"
#include "rtx_pack.h"

namespace rtx {

static thread_local std::string g_last_hip_error;

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_last_hip_error = std::string(#expr) + ": " + hipGetErrorString(e_);             \
            return (e_ == hipErrorOutOfMemory) ? RTX_E_OOM : RTX_E_HIP;                       \
        }                                                                                     \
    } while (0)

constexpr int kBlock = 256;
constexpr float kBias = 1e-4f;       // raytracing.cpp:447
constexpr float kMaxT = 1e8f;        // raytracing.cpp:10 (float nearest = Max_t)
constexpr int kTFlag = 1 << 30;      // the reference's `transformed` flag, carried in the prim word

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct SceneDev {
    int ns, nt, nl;
    int np;                       // polygons: primitive index ns + nt + i, traced after the triangles (:421-434)
    const rtx_polygon* poly;      // [np] raw records (materials)
    const float4* poly_verts;     // vertices, x y z 0; a polygon's run starts at its hot record's first_vertex
    const float4* sph_hot;        // [ns]  cx cy cz r^2
    const int* sph_xf;            // [ns]
    const rtx_xmatrix* sph_im;    // [ns]  precomputed inverseMatrix(mat)
    const rtx_sphere* sph;        // [ns]
    const float4* tri_hot;        // [nt*5]
    const float4* phot;           // [(ns+nt)*5] one 80-byte hot record per primitive (rtx_pack.h, prim_hot)
    const rtx_triangle* tri;      // [nt]
    const rtx_light* lights;      // [nl]
    F3 ambient;
    const float4* bounds;         // [ns+nt] filter records { c, -R^2 } (rtx_bounds.cpp)
    const float4* aff_shadow;     // [naff*4] AffineShadowRec: affine spheres for shadow rays
    int naff;
    F3 box_lo, box_inv;           // scene box for ray binning: cell = (o - box_lo) * box_inv, 16 cells per axis
    unsigned long long* pstat;    // [8] packet statistics: closest {waves, waves ok, batches, survivors}, shadow {same}
    const float4* sbounds;        // [nsb] filter records of the shadow casters except affine spheres (transposed scan of
    const int* sorder;            // [nsb]  unpacketable rays), and the primitive index of each
    int nsb;
    float guard;                  // |origin component| beyond this: the ray bypasses the filter
    // two-level packet test (rtx_pack.h, ClusterSet): cluster spheres, their 64 members, member -> primitive
    const float4* ccl; const float4* cmem; const int* cidx; const int* cseg; int cnseg;  // closest-hit, per segment
    const int* corder; int cnfirst;  // per-ray traversal order of the closest-hit clusters: affine-sphere clusters first
    const float4* cclR; const float4* csubR;  // { c, R } cluster spheres [ncl] and 8 sub-cluster spheres each [ncl*8]
    const float4* scl; const float4* smem; const int* sidx; int sncl;                    // shadow rays
};

struct CamDev {
    F3 ncrossv, up, vpn, camera, orig;
    const float* imagex;  // [width]
    const float* imagey;  // [height]
};

// One ray generation.  Rays are SoA (coalesced: lane i touches element i of each array).
struct LevelDev {
    int cap;
    float *ox, *oy, *oz, *dx, *dy, *dz;  // rays
    float* t;                            // nearest
    int* prim;                           // -1 miss | sphere i | ns + triangle i, | kTFlag
    float* val;                          // [cap*3] colour returned by raytrace() for this ray
    // hit records ("nodes"), compacted
    int* n_ray;                          // owning ray slot
    float *n_nx, *n_ny, *n_nz;           // shading normal (after the :458-461 flip)
    float *n_px, *n_py, *n_pz;           // intpoint
    int *n_crefl, *n_crefr;              // child ray slots in the next level, -1 = none
    float* n_fres;                       // fresnel (:465)
    unsigned char* lit;                  // [nl*cap] 1 = light reaches the hit (not in shadow)
};

__device__ __forceinline__ TriRec load_tri(const float4* hot, int i)
{
    const float4* p = hot + (size_t)i * 5;
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    TriRec T;
    T.nx = a.x; T.ny = a.y; T.nz = a.z; T.d = a.w;
    T.vx = b.x; T.vy = b.y; T.vz = b.z; T.area = b.w;
    T.wx = c.x; T.wy = c.y; T.wz = c.z; T.v1x = c.w;
    T.v1y = d.x; T.v1z = d.y; T.v2x = d.z; T.v2y = d.w;
    T.v2z = e.x; T.ex = e.y; T.ey = e.z; T.ez = e.w;
    return T;
}

// The 80-byte hot record of one primitive in registers (SceneDev::phot).  The lanes that need records
// fetch them with five independent loads -- one memory round trip for up to 64 primitives -- and a
// wave-uniform exact test then reads its record by v_readlane instead of a chain of scalar loads.
struct PrimRec { float4 q0, q1, q2, q3, q4; };

__device__ __forceinline__ PrimRec load_prim(const float4* phot, int k)
{
    const float4* p = phot + (size_t)k * 5;
    PrimRec r;
    r.q0 = p[0]; r.q1 = p[1]; r.q2 = p[2]; r.q3 = p[3]; r.q4 = p[4];
    return r;
}

__device__ __forceinline__ float4 bcast4(float4 v, int u)
{
    float4 r;
    r.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), u));
    r.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), u));
    r.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.z), u));
    r.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.w), u));
    return r;
}

// raytracing.cpp:303-325 with the ray-independent head precomputed (rtx_pack.cpp): q0 = { n, d } with n the
// stored normal NEGATED (always: :308 is `true` once it compiles; :305 compares a function
// with 0 and never fires), q1 = { area of the centroid fan, first vertex, vertex count, - }.
__device__ __forceinline__ bool polygon_test(F3 o, F3 dir, float4 q0, float4 q1, const float4* verts, float& t)
{
    const int first = __float_as_int(q1.y), cnt = __float_as_int(q1.z);
    if (cnt < 1) return false;
    t = -((q0.x * o.x + q0.y * o.y + q0.z * o.z) + q0.w) / (q0.x * dir.x + q0.y * dir.y + q0.z * dir.z);  // :313
    if (t < 0.0f) return false;
    const float px = o.x + dir.x * t, py = o.y + dir.y * t, pz = o.z + dir.z * t;
    float sum = 0.0f;  // findAreaPolygon(p, point), :289-300
    float4 vj = verts[first + cnt - 1];
    for (int i = 0; i < cnt; i++) {
        const float4 vi = verts[first + i];
        const float ux = px - vi.x, uy = py - vi.y, uz = pz - vi.z;
        const float vx = px - vj.x, vy = py - vj.y, vz = pz - vj.z;
        const float cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
        const float temparea = 0.5f * sqrtf(cx * cx + cy * cy + cz * cz);
        sum += temparea;
        vj = vi;
    }
    if ((double)fabsf(q1.x - sum) > 0.001) return false;  // float abs, compared as double (:322)
    return true;
}

// Slot allocation for a compacted queue: one atomicAdd per wavefront (64-bit ballot + mbcnt
// prefix), so queue order is wave-granular and the counter sees 1/64th of the traffic.
__device__ __forceinline__ int wave_alloc(bool want, int* counter)
{
    unsigned long long m = __ballot(want);
    if (m == 0) return -1;
    unsigned int lo = (unsigned int)m, hi = (unsigned int)(m >> 32);
    int prefix = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0));
    int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader);
    return want ? base + prefix : -1;
}

// The same for a whole workgroup of up to 1024 threads: slots keep the threads' order and the shared counter
// takes ONE atomic per workgroup.  Same-address atomics are serialised at ~10 ns each on this chip: the set-up
// kernel of a 4K frame spent 0.33 ms per queue on 32,400 of them with 256-thread workgroups, which is why it
// runs 1024-thread workgroups at generation 0.  sh: blockDim.x / 64 + 1 ints.
constexpr int kSetupBlock = 1024;
__device__ __forceinline__ int block_alloc(bool want, int* counter, int* sh)
{
    unsigned long long m = __ballot(want);
    int lp = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int k = 0; k < nw; k++) { if (k < w) woff += sh[k]; tot += sh[k]; }
    __syncthreads();
    if (threadIdx.x == 0) sh[nw] = tot ? atomicAdd(counter, tot) : 0;
    __syncthreads();
    const int base = sh[nw];
    __syncthreads();
    return want ? base + woff + lp : -1;
}

// Test counters (RTX_F_COUNT_TESTS): per-lane tallies, summed over the wave, one atomic per wave
// and kind.  kind: 0 sphere, 1 affine sphere, 2 triangle full path, 3 triangle early exit.
__device__ __forceinline__ void flush_counts(const unsigned int (&n)[4], unsigned long long* dst)
{
    for (int k = 0; k < 4; k++) {
        unsigned int v = n[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst + k, (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 ray generation (raytracing.cpp:753-759); origin is one constant (:735-737) but is stored
// per ray so that every level has the same layout.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_raygen(LevelDev L, CamDev C, const int* trace_rows, int ti0,
                                                   int nrows, int width, int* nrays)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    int n = nrows * width;
    if (i == 0) *nrays = n;
    if (i >= n) return;
    int r = i / width, x = i - r * width;
    int y = trace_rows[ti0 + r];
    float imagex = C.imagex[x], imagey = C.imagey[y];
    float a = imagex - C.camera.x, b = imagey - C.camera.y, c = -C.camera.z;
    F3 d;
    d.x = a * C.ncrossv.x + b * C.up.x + c * C.vpn.x;
    d.y = a * C.ncrossv.y + b * C.up.y + c * C.vpn.y;
    d.z = a * C.ncrossv.z + b * C.up.z + c * C.vpn.z;
    d = normalized(d);
    L.ox[i] = C.orig.x; L.oy[i] = C.orig.y; L.oz[i] = C.orig.z;
    L.dx[i] = d.x; L.dy[i] = d.y; L.dz[i] = d.z;
}

// ------------------------------------------------------------------------------------------------
// K2 closest hit (raytracing.cpp:382-419).  One ray per lane; the primitive index is
// wave-uniform, so primitive records arrive through scalar loads and sit in SGPRs.
// Spheres first, in file order, strict `<` (lowest index wins ties); then triangles.
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_closest(SceneDev S, LevelDev L, const int* nrays, unsigned long long* tests)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    bool live = i < *nrays;
    if (!COUNT && !live) return;
    unsigned int cnt[4] = {0, 0, 0, 0};
    if (live) {
    F3 o = f3(L.ox[i], L.oy[i], L.oz[i]);
    F3 d = f3(L.dx[i], L.dy[i], L.dz[i]);
    F3 nd = normalized(d);  // :390 for untransformed spheres: same input every iteration
    float nearest = kMaxT;
    int best = -1;
    int tflag = 0;
    for (int s = 0; s < S.ns; s++) {
        float4 h = S.sph_hot[s];
        F3 to = o, td = nd;
        int xf = S.sph_xf[s];
        if (xf) {  // wave-uniform branch
            const rtx_xmatrix& im = S.sph_im[s];
            to = inv_apply(o, im);
            td = normalized(inv_apply(d, im));
        }
        float t;
        if (COUNT) cnt[xf ? 1 : 0]++;
        if (sphere_test(to, td, h.x, h.y, h.z, h.w, t)) {
            if (t < nearest) {
                nearest = t;
                best = s;
                if (xf) tflag = kTFlag;  // sticky (:396-399)
            }
        }
    }
    for (int k = 0; k < S.nt; k++) {
        TriRec T = load_tri(S.tri_hot, k);
        float t = 0.0f, vd;
        bool h = triangle_test(o, d, T, t, vd);
        if (COUNT) cnt[(vd == 0.0f || t < 0.0f) ? 3 : 2]++;
        if (h) {
            if (t < nearest) {
                nearest = t;
                best = S.ns + k;
                tflag = 0;  // :416
            }
        }
    }
    for (int k = 0; k < S.np; k++) {  // :421-434
        const float4* p = S.phot + (size_t)(S.ns + S.nt + k) * 5;
        float t = 0.0f;
        bool h = polygon_test(o, d, p[0], p[1], S.poly_verts, t);
        if (COUNT) cnt[2]++;  // counted with the full-path triangle tests
        if (h) {
            if (t < nearest) {
                nearest = t;
                best = S.ns + S.nt + k;
                tflag = 0;  // :432
            }
        }
    }
    L.t[i] = nearest;
    L.prim[i] = (best < 0) ? -1 : (best | tflag);
    }
    if (COUNT) flush_counts(cnt, tests);
}


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
__device__ __forceinline__ void exact_closest(const SceneDev& S, int k, F3 o, F3 d, F3 nd, HitAcc& h)
{
    if (k < S.ns) {
        float4 hh = S.sph_hot[k];
        int xf = S.sph_xf[k];
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
        float t, vd;
        if (triangle_test(o, d, T, t, vd)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    } else {
        const float4* p = S.phot + (size_t)k * 5;
        float t;
        if (polygon_test(o, d, p[0], p[1], S.poly_verts, t)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    }
}


// exact_closest for a wave-uniform primitive k whose hot record sits in lane u's registers
__device__ __forceinline__ void exact_closest_bcast(const SceneDev& S, int k, const PrimRec& pr, int u, bool pass, F3 o, F3 d, F3 nd, HitAcc& h)
{
    const float4 q0 = bcast4(pr.q0, u);
    if (k < S.ns) {
        const float4 q4 = bcast4(pr.q4, u);
        const bool xf = q4.x != 0.0f;
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
        if (pass && triangle_test(o, d, T, t, vd)) {
            if (t < h.nearest) { h.nearest = t; h.best = k; h.tflag = 0; }
        }
    } else {
        const float4 q1 = bcast4(pr.q1, u);
        float t;
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
struct Packet { F3 co; float rho; F3 w; float cosT2, sinT; float c0; bool ok; };  // c0 = cos(theta), rounded down, before the widening by E

__device__ __forceinline__ float wave_min(float v) { for (int m = 32; m; m >>= 1) v = fminf(v, __shfl_xor(v, m)); return v; }
__device__ __forceinline__ float wave_max(float v) { for (int m = 32; m; m >>= 1) v = fmaxf(v, __shfl_xor(v, m)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int m = 32; m; m >>= 1) v += __shfl_xor(v, m); return v; }

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
    const F3 so = f3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
    const F3 sd = f3(__shfl(nd.x, src), __shfl(nd.y, src), __shfl(nd.z, src));
    F3 oo = f3(live ? o.x : so.x, live ? o.y : so.y, live ? o.z : so.z);
    F3 dd = f3(live ? nd.x : sd.x, live ? nd.y : sd.y, live ? nd.z : sd.z);
    float lx = wave_min(oo.x), hx = wave_max(oo.x), ly = wave_min(oo.y), hy = wave_max(oo.y), lz = wave_min(oo.z), hz = wave_max(oo.z);
    P.co = f3(0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz));
    float ex = oo.x - P.co.x, ey = oo.y - P.co.y, ez = oo.z - P.co.z;
    P.rho = wave_max(sqrtf(ex * ex + ey * ey + ez * ez)) * 1.00002f + 1e-30f;
    if (apex_ok) {  // wave-uniform
        float ax = oo.x - apex.x, ay = oo.y - apex.y, az = oo.z - apex.z;
        float maxD = wave_max(sqrtf(ax * ax + ay * ay + az * az));
        float rho_apex = 1.3e-4f + 3.8e-3f * maxD;
        if (rho_apex < P.rho) { P.rho = rho_apex; P.co = apex; }
    }
    F3 sw = f3(wave_sum(dd.x), wave_sum(dd.y), wave_sum(dd.z));
    float sn = sqrtf(sw.x * sw.x + sw.y * sw.y + sw.z * sw.z);
    bad = bad || !(sn > 1e-3f) || !(P.rho < 1e30f);
    float inv = 1.0f / fmaxf(sn, 1e-3f);
    P.w = f3(sw.x * inv, sw.y * inv, sw.z * inv);
    float c = wave_min(P.w.x * dd.x + P.w.y * dd.y + P.w.z * dd.z) - 4e-6f;  // cos(theta), rounded down
    bad = bad || !(c > 0.35f);
    c = fminf(c, 1.0f);
    P.c0 = c;
    float sT = sqrtf(fmaxf(0.0f, 1.0f - c * c)) + 4e-6f;
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
    float Rp = (sqrtf(-b.w) * 1.00001f + P.rho) * 1.00001f + 1e-20f;
    float rhs = Rp + a * P.sinT;
    return !(h2 * P.cosT2 > rhs * rhs * 1.00001f);     // NaN (R = inf, never-bounded primitives) passes
}

// The same test for ONE ray (origin o, unit or zero direction n): can the sphere b = { c, R } contain
// the bounding sphere of a primitive whose filter passes for this ray?  The filter's pass region lies
// inside { c : dist(c, double cone(o, +-n, E)) <= R_prim } (above), so a sphere around such primitives
// passes if its centre is within its radius of that cone.  Returns the slack, sign bit clear = pass
// (R = +inf gives +inf).  Used on cluster and sub-cluster spheres by the per-ray traversal; the plain
// filter is NOT valid on enclosing spheres: its hyperboloid lacks the cone's cross term 2 R a tan(E).
__device__ __forceinline__ float cone_slack(F3 o, F3 n, float4 b)
{
    const float cosE2 = 0.9999783f;  // (cos(3.7e-3) - 4e-6)^2 rounded down
    const float sinE = 3.705e-3f;    // sin(3.7e-3) rounded up, + 4e-6
    float Cx = b.x - o.x, Cy = b.y - o.y, Cz = b.z - o.z;
    float C2 = Cx * Cx + Cy * Cy + Cz * Cz;
    float a = fabsf(n.x * Cx + n.y * Cy + n.z * Cz);
    float h2 = (C2 - a * a) - 4e-6f * C2;
    float rhs = (b.w * 1.00001f + 1e-20f) + a * sinE;
    return rhs * rhs * 1.00001f - h2 * cosE2;
}

template <bool PREFETCH>
__device__ __forceinline__ void closest_packet(const SceneDev& S, const LevelDev& L, const int n, const int base, float4* tile, const int* perm)
{
    const int lane = threadIdx.x & 63;
    const int q = base + threadIdx.x;
    const bool live = q < n;
    // secondary generations are traced in binned order (k_ray_bin): perm maps sorted position -> ray slot
    const int i = live ? (perm ? perm[q] : q) : 0;
    const int ii = i;
    F3 o = f3(L.ox[ii], L.oy[ii], L.oz[ii]);
    F3 d = f3(L.dx[ii], L.dy[ii], L.dz[ii]);
    F3 nd = normalized(d);
    const bool abn = ray_abnormal(o, d, nd, S.guard);
    const bool use = live && !abn;
    F3 fo = use ? o : f3(1e18f, 0.0f, 0.0f);
    F3 fd = use ? f3(nd.x * kDirScale, nd.y * kDirScale, nd.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
    HitAcc h;
    h.nearest = kMaxT; h.best = -1; h.tflag = 0; h.txf = kMaxT;
    const Packet P = make_packet(o, nd, live, abn);
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
                pk_batches++;
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
        // Few waves per SIMD (frames below ~4 M rays): a wave's serial chain of dependent loads IS the run time.
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
            const int tot = __shfl(incl, 63);
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
                    if (__any(pass)) exact_closest_bcast(S, __builtin_amdgcn_readlane(k, u), pr, u, pass, o, d, nd, h);
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
                    if (live && (abn || !(filt(fo, fd, br) < 0.0f))) exact_closest(S, k, o, d, nd, h);
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
    if (live) {
        L.t[i] = h.nearest;
        L.prim[i] = (h.best < 0) ? -1 : (h.best | h.tflag);
    }
}


// Per-ray traversal of the cluster hierarchy, for the generations after the first: reflection and
// refraction rays of neighbouring pixels leave curved surfaces in all directions, a wave of them is no
// packet (measured on the benchmark scene: the packet test kept 1,171 of 10,000 primitives per wave at
// level 1, against 17 at level 0).  Every LANE walks the hierarchy for its own ray instead:
//   1. the filter against every cluster sphere (wave-uniform loop, 64 clusters per round, a bit per pass),
//   2. lanes step through their own passing clusters in lockstep and run the filter against the 64
//      members (per-lane gathers, sign bits AND-ed over groups of 8); members that pass are appended to
//      the lane's candidate list in LDS,
//   3. whenever a list is nearly full, and at the end, all lanes run the exact tests of their lists in
//      lockstep.
// The candidates of a ray arrive in cluster order, not file order, so the reference's sequential scan
// (:385-418) is restated order-independently.  With H = the accepted hits (t_k, k):
//   * nearest/best = the lexicographic minimum of (t_k, k) over hits with t_k < Max_t: the scan keeps
//     the first of equal t's because it updates on strict `<` only;
//   * a triangle that wins resets the `transformed` flag and only triangles follow it, so the flag is
//     0 when best is a triangle; otherwise it is 1 iff some affine sphere k was a record when the scan
//     reached it: t_k < Max_t and t_j > t_k for every hit j < k.  Affine-sphere clusters are traversed
//     FIRST; their hits form a list in which (j < k, t_j <= t_k) removes k, and every later hit of a
//     plain sphere removes entries the same way.  The flag is "list not empty".  The list holds the
//     staircase of affine hits (index up, t down), 4 entries; a fifth sends the lane to the ordered
//     scan over every primitive at the end (wave-uniform loop, only the flagged lanes active).
// Rays the filter does not cover (ray_abnormal) take every cluster and every member.
constexpr int kCandCap = 24;   // candidate list entries per lane (LDS)
constexpr int kQBatch = 32;    // clusters per round; the lane's queue of passing clusters holds as many
constexpr int kAffList = 4;

constexpr int kLaneLds = kCandCap * 64 * 4 + kQBatch * 64 * 2;  // LDS bytes per wave: candidate lists + cluster queues

struct LexAcc {
    float nearest; int best;
    float at[kAffList]; int ak[kAffList];
    bool overflow;
};

// exact test of primitive k (per lane) with the order-independent update.  The primitive's 80-byte hot
// record (SceneDev::phot) comes in with five independent loads: one memory round trip per round of the
// lockstep loop, where sph_hot -> sph_xf -> sph_im was a chain of three.
__device__ __forceinline__ void exact_lex(const SceneDev& S, int k, F3 o, F3 d, F3 nd, LexAcc& h)
{
    const float4* p = S.phot + (size_t)k * 5;
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
    if (k < S.ns) {
        const bool xf = q4.x != 0.0f;
        F3 to = o, td = nd;
        if (xf) {
            rtx_xmatrix im;
            im.a.x = q1.x; im.a.y = q1.y; im.a.z = q1.z; im.b.x = q1.w;
            im.b.y = q2.x; im.b.z = q2.y; im.c.x = q2.z; im.c.y = q2.w;
            im.c.z = q3.x; im.d.x = q3.y; im.d.y = q3.z; im.d.z = q3.w;
            to = inv_apply(o, im);
            td = normalized(inv_apply(d, im));
        }
        float t;
        if (sphere_test(to, td, q0.x, q0.y, q0.z, q0.w, t)) {
            if (t < h.nearest || (t == h.nearest && k < h.best)) { h.nearest = t; h.best = k; }
            if (xf) {
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
            } else {
#pragma unroll
                for (int e = 0; e < kAffList; e++)
                    if (h.ak[e] >= 0 && k < h.ak[e] && t <= h.at[e]) h.ak[e] = -1;
            }
        }
    } else if (k >= S.ns + S.nt) {
        float t;
        if (polygon_test(o, d, q0, q1, S.poly_verts, t)) {
            if (t < h.nearest || (t == h.nearest && k < h.best)) { h.nearest = t; h.best = k; }
        }
    } else {
        TriRec T;
        T.nx = q0.x; T.ny = q0.y; T.nz = q0.z; T.d = q0.w;
        T.vx = q1.x; T.vy = q1.y; T.vz = q1.z; T.area = q1.w;
        T.wx = q2.x; T.wy = q2.y; T.wz = q2.z; T.v1x = q2.w;
        T.v1y = q3.x; T.v1z = q3.y; T.v2x = q3.z; T.v2y = q3.w;
        T.v2z = q4.x; T.ex = q4.y; T.ey = q4.z; T.ez = q4.w;
        float t, vd;
        if (triangle_test(o, d, T, t, vd)) {
            if (t < h.nearest || (t == h.nearest && k < h.best)) { h.nearest = t; h.best = k; }
        }
    }
}

__device__ __forceinline__ void closest_lanes(const SceneDev& S, const LevelDev& L, const int n, const int q, float4* tile, const int* perm,
                                              const int sub, const int nsub)
{
    const int lane = threadIdx.x & 63;
    const bool live = q < n;
    const int i = live ? (perm ? perm[q] : q) : 0;
    F3 o = f3(L.ox[i], L.oy[i], L.oz[i]);
    F3 d = f3(L.dx[i], L.dy[i], L.dz[i]);
    F3 nd = normalized(d);
    const bool abn = live && ray_abnormal(o, d, nd, S.guard);
    const bool use = live && !abn;
    F3 fo = use ? o : f3(1e18f, 0.0f, 0.0f);
    F3 fd = use ? f3(nd.x * kDirScale, nd.y * kDirScale, nd.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
    LexAcc h;
    h.nearest = kMaxT; h.best = -1; h.overflow = false;
#pragma unroll
    for (int e = 0; e < kAffList; e++) { h.ak[e] = -1; h.at[e] = 0.0f; }
    int* cand = reinterpret_cast<int*>(tile) + (threadIdx.x >> 6) * (kLaneLds / 4) + lane;  // cand[slot * 64]
    unsigned short* queue = reinterpret_cast<unsigned short*>(reinterpret_cast<int*>(tile) + (threadIdx.x >> 6) * (kLaneLds / 4) + kCandCap * 64) + lane;  // queue[slot * 64]
    int ncand = 0, flushes = 0, tested = 0;
    auto flush = [&]() {
        const int maxn = (int)wave_max((float)ncand);
        for (int a = 0; a < maxn; a++)
            if (a < ncand) exact_lex(S, cand[a * 64], o, d, nd, h);
        tested += ncand; flushes++;
        ncand = 0;
    };
    const int ncl = S.cseg[S.cnseg];
    for (int phase = 0; phase < 2; phase++) {
        const int cbeg = phase ? S.cnfirst : 0, cend = phase ? ncl : S.cnfirst;
        for (int cb = cbeg; cb < cend; cb += kQBatch) {
            // (1) this round's cluster spheres: one coalesced load, then lane u's record is broadcast
            unsigned cmask = 0u;
            const int ce = min(kQBatch, cend - cb);
            const int myc = S.corder[min(cb + (lane & (kQBatch - 1)), cend - 1)];
            const float4 mine = S.cclR[myc];
            for (int u = 0; u < ce; u++) {
                float4 b;
                b.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.x), u));
                b.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.y), u));
                b.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.z), u));
                b.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.w), u));
                if (!(cone_slack(o, nd, b) < 0.0f)) cmask |= 1u << u;
            }
            if (abn) cmask = (ce == 32) ? ~0u : ((1u << ce) - 1u);
            if (!live) cmask = 0u;
            if (phase == 1 && nsub == 4) cmask &= 0x11111111u << sub;  // split generation: this lane's quarter of the clusters
            // (2) sub-cluster spheres of the lane's own passing clusters -> queue of (cluster slot, group mask)
            int nq = 0;
            while (__any(cmask != 0u)) {
                const bool has = cmask != 0u;
                const int u = has ? (__ffs(cmask) - 1) : 0;
                cmask &= cmask - 1u;  // 0 stays 0
                const int cid = __shfl(myc, u);
                const float4* srec = S.csubR + (size_t)cid * 8;
                unsigned gmask = 0u;
#pragma unroll
                for (int v = 0; v < 8; v++)
                    if (!(cone_slack(o, nd, srec[v]) < 0.0f)) gmask |= 1u << v;
                if (abn) gmask = 0xFFu;
                if (has && gmask) { queue[nq * 64] = (unsigned short)((u << 8) | gmask); nq++; }
            }
            // (3) the members of the lane's own passing groups, one group of 8 per step
            int qi = 0, cid = 0;
            unsigned gm = 0u;
            while (true) {
                const bool refill = gm == 0u && qi < nq;
                const unsigned e = refill ? (unsigned)queue[qi * 64] : 0u;
                const int cnext = __shfl(myc, (int)(e >> 8));  // every lane takes part: ds_bpermute reads 0 from inactive lanes
                if (refill) { qi++; gm = e & 0xFFu; cid = cnext; }
                if (!__any(gm != 0u)) break;
                const bool hasg = gm != 0u;
                const int g = hasg ? (__ffs(gm) - 1) : 0;
                gm &= gm - 1u;
                const float4* mrec = S.cmem + (size_t)cid * 64 + g * 8;
                const int* midx = S.cidx + (size_t)cid * 64 + g * 8;
                float4 mb[kGroup];
#pragma unroll
                for (int v = 0; v < kGroup; v++) mb[v] = mrec[v];
                unsigned acc = 0xFFFFFFFFu;
#pragma unroll
                for (int v = 0; v < kGroup; v++) acc &= __float_as_uint(filt(fo, fd, mb[v]));
                if (hasg && (abn || (int)acc >= 0)) {
#pragma unroll
                    for (int v = 0; v < kGroup; v++) {
                        if (abn || !(filt(fo, fd, mb[v]) < 0.0f)) {
                            const int k = midx[v];
                            if (k >= 0) { cand[ncand * 64] = k; ncand++; }
                        }
                    }
                }
                if (__any(ncand > kCandCap - kGroup)) flush();
            }
        }
        flush();  // phase 0: every affine-sphere hit is known before the first plain sphere is tested
    }
    if (nsub == 4) {
        // the four lanes of a ray hold the same affine list (phase 0 is not split) with their own kills, and
        // their own nearest hit: an entry lives if no lane removed it, the lexicographic minimum wins
#pragma unroll
        for (int m = 1; m <= 2; m <<= 1) {
            const float t2 = __shfl_xor(h.nearest, m);
            const int b2 = __shfl_xor(h.best, m);
            if (t2 < h.nearest || (t2 == h.nearest && b2 < h.best)) { h.nearest = t2; h.best = b2; }
#pragma unroll
            for (int e = 0; e < kAffList; e++) h.ak[e] = min(h.ak[e], __shfl_xor(h.ak[e], m));
            h.overflow = h.overflow || (__shfl_xor((int)h.overflow, m) != 0);
        }
    }
    bool alive = false;
#pragma unroll
    for (int e = 0; e < kAffList; e++) alive = alive || h.ak[e] >= 0;
    float nearest = h.nearest;
    int best = h.best;
    int tflag = (best >= 0 && best < S.ns && alive) ? kTFlag : 0;
    const bool slow = live && h.overflow;
    if (__any(slow)) {
        HitAcc hs;
        hs.nearest = kMaxT; hs.best = -1; hs.tflag = 0; hs.txf = kMaxT;
        const int nprim = S.ns + S.nt + S.np;
        for (int k = 0; k < nprim; k++) {
            const float4 b = S.bounds[k];
            if (slow && (abn || !(filt(fo, fd, b) < 0.0f))) exact_closest(S, k, o, d, nd, hs);
        }
        if (slow) { nearest = hs.nearest; best = hs.best; tflag = hs.tflag; }
    }
    if (S.pstat) {
        const unsigned long long sl = __ballot(slow);
        const unsigned long long ab = __ballot(abn);
        const float tot = wave_sum(live ? (float)tested : 0.0f);
        if (lane == 0) {
            atomicAdd(S.pstat + 0, 1ull); atomicAdd(S.pstat + 1, (unsigned long long)__popcll(sl) + ((unsigned long long)__popcll(ab) << 32));
            atomicAdd(S.pstat + 2, (unsigned long long)flushes); atomicAdd(S.pstat + 3, (unsigned long long)tot);
        }
    }
    if (live && sub == 0) {
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

__device__ __forceinline__ void closest_thin(const SceneDev& S, const LevelDev& L, const int n, float4* tile)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + lane;  // the four waves hold the same 64 rays
    const bool live = i < n;
    const int ii = live ? i : 0;
    F3 o = f3(L.ox[ii], L.oy[ii], L.oz[ii]);
    F3 d = f3(L.dx[ii], L.dy[ii], L.dz[ii]);
    F3 nd = normalized(d);
    const bool abn = ray_abnormal(o, d, nd, S.guard);
    const bool use = live && !abn;
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
            if ((int)acc >= 0) {
                int gend = min(kGroup, cnt - g);
                for (int u = 0; u < gend; u++)
                    if (live && (abn || !(filt(fo, fd, mt[g + u]) < 0.0f))) exact_closest(S, t0 + g + u, o, d, nd, h);
            }
        }
    }
    __syncthreads();
    float* rt = reinterpret_cast<float*>(tile);
    int* rb = reinterpret_cast<int*>(tile) + 256;
    float* rx = reinterpret_cast<float*>(tile) + 512;
    rt[threadIdx.x] = h.nearest; rb[threadIdx.x] = h.best; rx[threadIdx.x] = h.txf;
    __syncthreads();
    if (w == 0 && live) {
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

template <bool PREFETCH>
__global__ __launch_bounds__(kBlock) void k_closest_f(SceneDev S, LevelDev L, const int* nrays, const int* perm)
{
    __shared__ float4 tile[kTile];  // 16 KiB: thin tiles / the four waves' bitmaps (2 KiB) and candidate lists (2 KiB)
    static_assert((kSegPrims / 32 + 512) * 4 * (kBlock / 64) <= kTile * 16, "bitmaps and lists fit");
    const int n = *nrays;
    if (n > kThinRays) {
        const int base = blockIdx.x * kBlock;
        if (base >= n) return;  // block-uniform
        closest_packet<PREFETCH>(S, L, n, base, tile, perm);
    } else {
        if (blockIdx.x * 64 >= n) return;
        closest_thin(S, L, n, tile);
    }
}

// Generations of at most kSplitClosest rays put FOUR lanes on every ray (64 rays per workgroup): the lanes
// share the affine-sphere clusters and take a quarter of the others each, and merge at the end.  A thin
// generation has fewer waves than the chip has SIMDs; its run time is one wave's serial walk, now a
// third as long.
constexpr int kSplitClosest = 1 << 17;
constexpr size_t kPrefetchRays = (size_t)1 << 22;  // generation-0 frames below this take the latency-oriented walk (k_closest_f<true>)

__global__ __launch_bounds__(kBlock) void k_closest_lanes(SceneDev S, LevelDev L, const int* nrays, const int* perm)
{
    __shared__ float4 tile[kLaneLds * (kBlock / 64) / 16];  // the four waves' candidate lists and cluster queues (40 KiB)
    const int n = *nrays;
    const int* pm = n > kThinRays ? perm : nullptr;
    if (n <= kSplitClosest) {
        if (blockIdx.x * 64 >= n) return;  // block-uniform
        closest_lanes(S, L, n, blockIdx.x * 64 + (threadIdx.x >> 2), tile, pm, threadIdx.x & 3, 4);
    } else {
        if (blockIdx.x * kBlock >= n) return;
        closest_lanes(S, L, n, blockIdx.x * kBlock + threadIdx.x, tile, pm, 0, 1);
    }
}

// exact any-hit for primitive k (wave-uniform), raytracing.cpp:569-589
__device__ __forceinline__ bool exact_any(const SceneDev& S, int k, F3 o, F3 ld, F3 nld)
{
    if (k < S.ns) {
        float4 hh = S.sph_hot[k];
        F3 td = nld;
        if (S.sph_xf[k]) td = normalized(inv_apply(ld, S.sph_im[k]));
        float t;
        return sphere_test(o, td, hh.x, hh.y, hh.z, hh.w, t);
    }
    TriRec T = load_tri(S.tri_hot, k - S.ns);
    float t, vd;
    return triangle_test(o, ld, T, t, vd);
}

// Shadow rays (raytracing.cpp:560-589) are any-hit: which primitive blocks the light first does not
// matter, only whether one does.  So they are not traced in file order in one long loop.  They are
// materialised once per generation into a queue sorted by (light, direction cell) (k_shadow_bin, k_scan_*,
// k_shadow_scatter) and traced by k_shadow_pass as packets with their apex at the light: plain spheres
// and triangles through the shadow cluster hierarchy, affine spheres through a strip test of their own.
struct ShadowQ { float *ox, *oy, *oz, *dx, *dy, *dz; int* id; };

// Light-space binning.  Every shadow ray's line passes through its light, so rays towards one light
// that leave it in nearly the same direction are nearly the same line, wherever their hit points
// are.  The rays of a generation are therefore counting-sorted by (light, direction cell) -- an
// octahedral G x G map of the direction, cells in Morton order -- before the passes: a wave then
// holds 64 rays inside a cone of a degree or so with its apex AT the light, the packet test keeps a
// few dozen of the 10,000 primitives, and the 64 rays tend to be blocked by the same primitive.
__device__ __forceinline__ unsigned morton8(unsigned x)
{
    x &= 0xffu;
    x = (x | (x << 4)) & 0x0f0fu;
    x = (x | (x << 2)) & 0x3333u;
    x = (x | (x << 1)) & 0x5555u;
    return x;
}

__device__ __forceinline__ int light_cell(F3 v, int G)
{
    float s = fabsf(v.x) + fabsf(v.y) + fabsf(v.z);
    float inv = 1.0f / fmaxf(s, 1e-30f);
    float px = v.x * inv, py = v.y * inv;
    if (v.z < 0.0f) {
        float qx = (1.0f - fabsf(py)) * (px >= 0.0f ? 1.0f : -1.0f);
        float qy = (1.0f - fabsf(px)) * (py >= 0.0f ? 1.0f : -1.0f);
        px = qx; py = qy;
    }
    // fminf/fmaxf drop NaN operands, so a NaN direction lands in cell 0
    int cx = (int)fminf(fmaxf((px * 0.5f + 0.5f) * (float)G, 0.0f), (float)(G - 1));
    int cy = (int)fminf(fmaxf((py * 0.5f + 0.5f) * (float)G, 0.0f), (float)(G - 1));
    return (int)(morton8((unsigned)cx) | (morton8((unsigned)cy) << 1));
}


// Secondary rays (reflection / refraction) of neighbouring pixels leave from different surfaces in
// different directions.  Before a generation k >= 1 is traced its rays are counting-sorted by
// (origin cell of a 16^3 grid over the scene box, 16 x 16 octahedral direction cell), Morton order,
// and traced in that order through a permutation, so that a wave is again a packet with a small
// origin sphere and a narrow cone.  Generations too thin for packet mode skip the sort.
constexpr int kRayBins = 1 << 20;

__device__ __forceinline__ unsigned morton3_4(unsigned x)  // 4 bits -> every third bit
{
    x &= 0xfu;
    x = (x | (x << 4)) & 0x0c3u;
    x = (x | (x << 2)) & 0x249u;
    return x;
}

__device__ __forceinline__ int ray_key(const SceneDev& S, F3 o, F3 d)
{
    int cx = (int)fminf(fmaxf((o.x - S.box_lo.x) * S.box_inv.x, 0.0f), 15.0f);
    int cy = (int)fminf(fmaxf((o.y - S.box_lo.y) * S.box_inv.y, 0.0f), 15.0f);
    int cz = (int)fminf(fmaxf((o.z - S.box_lo.z) * S.box_inv.z, 0.0f), 15.0f);
    unsigned oc = morton3_4((unsigned)cx) | (morton3_4((unsigned)cy) << 1) | (morton3_4((unsigned)cz) << 2);
    return (int)((oc << 8) | (unsigned)light_cell(d, 16));
}

__global__ __launch_bounds__(kBlock) void k_ray_bin(SceneDev S, LevelDev L, const int* nrays, int* hist, int* rank)
{
    const int n = *nrays;
    if (n <= kThinRays) return;  // thin generations are not traced in packet mode
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    int key = -1;
    if (live) key = ray_key(S, f3(L.ox[i], L.oy[i], L.oz[i]), f3(L.dx[i], L.dy[i], L.dz[i]));
    int myrank = 0;
    unsigned long long pending = __ballot(live);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(live && key == k0) & pending;
        const int prefix = __popcll(same & ((1ull << (threadIdx.x & 63)) - 1ull));
        int base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(hist + k0, __popcll(same));
        base = __shfl(base, leader);
        if ((same >> (threadIdx.x & 63)) & 1ull) myrank = base + prefix;
        pending &= ~same;
    }
    if (live) rank[i] = myrank;
}

__global__ __launch_bounds__(kBlock) void k_ray_scatter(SceneDev S, LevelDev L, const int* nrays, const int* offs, const int* rank, int* perm)
{
    const int n = *nrays;
    if (n <= kThinRays) return;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int key = ray_key(S, f3(L.ox[i], L.oy[i], L.oz[i]), f3(L.dx[i], L.dy[i], L.dz[i]));
    perm[offs[key] + rank[i]] = i;
}

struct ShadowRay { F3 o, ld; int key; };

__device__ __forceinline__ ShadowRay make_shadow_ray(const SceneDev& S, const LevelDev& L, int j, int li, int G)
{
    const rtx_light& Lt = S.lights[li];
    F3 ip = f3(L.n_px[j], L.n_py[j], L.n_pz[j]);
    F3 nn = f3(L.n_nx[j], L.n_ny[j], L.n_nz[j]);
    ShadowRay r;
    r.ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));                  // :561-562
    r.o = f3(ip.x + nn.x * kBias, ip.y + nn.y * kBias, ip.z + nn.z * kBias);                       // :568
    r.key = li * G * G + light_cell(r.ld, G);
    return r;
}

__global__ __launch_bounds__(kBlock) void k_shadow_bin(SceneDev S, LevelDev L, const int* nnodes, int G, int* hist, int* rank)
{
    const int n = *nnodes;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int li = blockIdx.y;
    const bool live = j < n;
    int key = -1;
    if (live) {
        key = make_shadow_ray(S, L, j, li, G).key;
        L.lit[li * L.cap + j] = 1;  // until some primitive blocks it
    }
    // Neighbouring hits mostly fall into the same few cells: one returning atomic per distinct key
    // per wave (leader election on the first pending lane's key) instead of one per ray.
    int myrank = 0;
    unsigned long long pending = __ballot(live);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int k0 = __shfl(key, leader);
        const unsigned long long same = __ballot(live && key == k0) & pending;
        const int prefix = __popcll(same & ((1ull << (threadIdx.x & 63)) - 1ull));
        int base = 0;
        if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(hist + k0, __popcll(same));
        base = __shfl(base, leader);
        if (live && key == k0 && ((same >> (threadIdx.x & 63)) & 1ull)) myrank = base + prefix;
        pending &= ~same;
    }
    if (live) rank[li * n + j] = myrank;
}

// Exclusive scan of the bin counts in three small launches: per-block scan of 1024 bins, scan of the
// block sums, add-back.  nbins <= 4 Mi, so at most 4096 block sums.
// align > 1 (shadow queues): every group of 1024 bins -- a 32 x 32 square of one light's direction map --
// starts at a slot that is a multiple of `align` (64), so no wave straddles two lights or a long jump of
// the Morton curve; the slots skipped are marked as padding (id = -1) by k_scan_add.
__global__ __launch_bounds__(1024) void k_scan_blocks(const int* hist, int* offs, int nbins, int* bsum, int* braw, int align)
{
    __shared__ int part[1024];
    const int t = threadIdx.x, b = blockIdx.x * 1024 + t;
    const int v = (b < nbins) ? hist[b] : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int x = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    if (b < nbins) offs[b] = part[t] - v;
    if (t == 1023) {
        braw[blockIdx.x] = part[1023];
        bsum[blockIdx.x] = (part[1023] + align - 1) / align * align;
    }
}

__global__ __launch_bounds__(1024) void k_scan_sums(int* bsum, int nblocks, int* qcount)
{
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = min(t * per, nblocks), b1 = min(b0 + per, nblocks);
    int sum = 0;
    for (int b = b0; b < b1; b++) sum += bsum[b];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int x = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int b = b0; b < b1; b++) { int v = bsum[b]; bsum[b] = run; run += v; }
    if (t == 1023) *qcount = part[1023];
}

__global__ __launch_bounds__(1024) void k_scan_add(int* offs, int nbins, const int* bsum, const int* braw, int align, int* pad_ids)
{
    const int b = blockIdx.x * 1024 + threadIdx.x;
    const int start = bsum[blockIdx.x];
    if (b < nbins) offs[b] += start;
    if (pad_ids) {
        const int raw = braw[blockIdx.x];
        const int pad = (raw + align - 1) / align * align - raw;
        if ((int)threadIdx.x < pad) pad_ids[start + raw + threadIdx.x] = -1;
    }
}

__global__ __launch_bounds__(kBlock) void k_shadow_scatter(SceneDev S, LevelDev L, const int* nnodes, int G, const int* offs,
                                                           const int* rank, ShadowQ Q)
{
    const int n = *nnodes;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int li = blockIdx.y;
    if (j >= n) return;
    ShadowRay r = make_shadow_ray(S, L, j, li, G);
    const int q = offs[r.key] + rank[li * n + j];
    Q.ox[q] = r.o.x; Q.oy[q] = r.o.y; Q.oz[q] = r.o.z;
    Q.dx[q] = r.ld.x; Q.dy[q] = r.ld.y; Q.dz[q] = r.ld.z;
    Q.id[q] = li * L.cap + j;
}

// Affine spheres.  The reference maps only the
// DIRECTION into object space for shadow rays (:571-576); filter = line (o, q = ld*A) against the
// sphere, (1 + 2^-13)(q.X)^2 >= |q|^2 (|X|^2 - R^2)  (rtx_bounds.cpp, affine_shadow_record).
__device__ __forceinline__ float filt_affine(F3 o, F3 l, float4 m0, float4 m1, float4 m2, float4 m3)
{
    // m0 = a.x a.y a.z b.x | m1 = b.y b.z c.x c.y | m2 = c.z o.x o.y o.z | m3 = -R^2 index . .
    float qx = __builtin_fmaf(l.x, m0.x, __builtin_fmaf(l.y, m0.w, l.z * m1.z));
    float qy = __builtin_fmaf(l.x, m0.y, __builtin_fmaf(l.y, m1.x, l.z * m1.w));
    float qz = __builtin_fmaf(l.x, m0.z, __builtin_fmaf(l.y, m1.y, l.z * m2.x));
    float X = o.x - m2.y, Y = o.y - m2.z, Z = o.z - m2.w;
    float bq = __builtin_fmaf(qx, X, __builtin_fmaf(qy, Y, qz * Z));
    float qq = __builtin_fmaf(qx, qx, __builtin_fmaf(qy, qy, qz * qz));
    float cc = __builtin_fmaf(X, X, __builtin_fmaf(Y, Y, __builtin_fmaf(Z, Z, m3.x)));
    return __builtin_fmaf(bq * (1.0f + 1.0f / 8192.0f), bq, -(qq * cc));
}

// One shadow pass with the packet test in front of the per-ray filter (see closest_packet).
__device__ __forceinline__ void shadow_pass_packet(const SceneDev& S, const ShadowQ& Qi, const int n, const int base,
                                                   unsigned char* lit, const int lit_cap, const int part, const int parts)
{
    const int c0 = 0, c1 = S.nsb;
    const int lane = threadIdx.x & 63;
    const int i = base + threadIdx.x;
    const int ii = (i < n) ? i : 0;
    const int id = Qi.id[ii];
    const bool live = (i < n) && id >= 0;  // id < 0: alignment padding of the bin groups
    F3 o = f3(Qi.ox[ii], Qi.oy[ii], Qi.oz[ii]);
    F3 ld = f3(Qi.dx[ii], Qi.dy[ii], Qi.dz[ii]);
    F3 nld = normalized(ld);
    const bool abn = ray_abnormal(o, ld, nld, S.guard);
    const bool use = live && !abn;
    F3 fo = use ? o : f3(1e18f, 0.0f, 0.0f);
    F3 fd = use ? f3(nld.x * kDirScale, nld.y * kDirScale, nld.z * kDirScale) : f3(0.0f, 0.0f, 0.0f);
    bool done = !live;
    int pk_batches = 0, pk_surv = 0, pk_blocks = 0, pk_ok = 0, pk_aff = 0;
    auto bcast4 = [](float4 v, int u) {
        float4 r;
        r.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.x), u));
        r.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.y), u));
        r.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.z), u));
        r.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v.w), u));
        return r;
    };
    // A wave is traced as ONE packet when its 64 rays make a usable one.  At a seam -- the last rays towards
    // one light followed by the first towards the next, a jump of the Morton curve over the direction map,
    // an abnormal ray -- they do not: the lanes are then split into aligned blocks of 32, 16, ... 1, each
    // traced as a packet of its own with the other lanes switched off.  A single ray that makes no packet
    // (an abnormal one) takes the transposed scan below.
    const int li = id / lit_cap;
    unsigned long long remaining = __ballot(live), fallback = 0ull;
    int size = 64;
    while (remaining) {
        const int first = __ffsll((long long)remaining) - 1;
        const int lo = first & ~(size - 1);
        const unsigned long long blk = ((size == 64) ? ~0ull : (((1ull << size) - 1ull) << lo)) & remaining;
        const bool sel = (blk >> lane) & 1ull;
        // all of the block's lanes aimed at the same light?  then their lines share that point
        const int li0 = __builtin_amdgcn_readlane(li, first);
        const bool one_light = __all(!sel || li == li0) && li0 >= 0 && li0 < S.nl;
        F3 apex = f3(0.0f, 0.0f, 0.0f);
        if (one_light) apex = f3(S.lights[li0].point);
        const Packet P = make_packet(o, nld, sel, abn, one_light, apex);
        if (!P.ok && size > 1) { size >>= 1; continue; }
        remaining &= ~blk;
        if (!P.ok) { fallback |= blk; continue; }
        pk_blocks++; pk_ok += P.ok ? 1 : 0;
        // lanes outside the block sit this round out
        const bool done_keep = done;
        const F3 fo_keep = fo, fd_keep = fd;
        if (!sel) { done = true; fo = f3(1e18f, 0.0f, 0.0f); fd = f3(0.0f, 0.0f, 0.0f); }
        if (P.ok) {  // wave-uniform: two-level packet test, any order (any-hit)
            bool out = __all(done);
            for (int cb = 0; cb < S.sncl && !out; cb += 64) {
                const int c = cb + lane;
                unsigned long long cm = __ballot(c < S.sncl && (c % parts) == part && packet_test(P, S.scl[min(c, S.sncl - 1)]));
                pk_batches++;
                while (cm && !out) {
                    const int j = __ffsll((long long)cm) - 1;
                    cm &= cm - 1;
                    const size_t m = (size_t)(cb + j) * 64 + lane;
                    const float4 mb = S.smem[m];
                    const int k = S.sidx[m];
                    unsigned long long mm = __ballot(k >= 0 && packet_test(P, mb));
                    pk_batches++; pk_surv += __popcll(mm);
                    while (mm) {
                        const int u = __ffsll((long long)mm) - 1;
                        mm &= mm - 1;
                        const float4 br = bcast4(mb, u);
                        if (!done && !(filt(fo, fd, br) < 0.0f)) {
                            if (exact_any(S, __builtin_amdgcn_readlane(k, u), o, ld, nld)) {
                                done = true;
                                fo = f3(1e18f, 0.0f, 0.0f);
                                fd = f3(0.0f, 0.0f, 0.0f);
                            }
                        }
                    }
                    out = __all(done);
                }
            }
            // Affine spheres (raytracing.cpp:571-576: only the DIRECTION is mapped, q = ld*A, the origin stays
            // in world space).  Packet test, one sphere per lane: with a reference point Lr (the light, or the
            // packet's origin centre) every origin is Lr - D w + e, D in [Dmin, Dmax], |e| <= rperp, and every
            // direction is (w + delta) A with |delta| <= chord.  For a unit axis a orthogonal to q_w = w A the
            // point p of a line nearest the sphere centre c has a.(p - Lr) = -D (a.w) + a.e + s a.(delta A),
            // |s| |q| = |p - o| <= |c - o|, and the per-ray filter (1 + 2^-13)(q.X)^2 >= |q|^2 (|X|^2 - R^2)
            // passes only if |p - c| <= R + tan(alpha) |p - o|, tan^2(alpha) <= 2^-13 + 2^-19.  So a sphere some
            // ray can pass has, on BOTH axes of an orthonormal pair around q_w,
            //     a.(c - Lr) within [min, max](-D (a.w)) widened by T,
            //     T = R + rperp + (chord |A a^T| / min|ld A| + 0.0112) far + rounding slack,   far >= |c - o|,
            // (a.(delta A) = delta.(A a^T); |s| <= |p - o| / |ld A|).  Everything else is skipped for the whole block.
            if (!out && S.naff > 0) {
                const F3 Lr = one_light ? apex : P.co;
                const F3 so = f3(__shfl(o.x, first), __shfl(o.y, first), __shfl(o.z, first));
                const F3 oo = f3(sel ? o.x : so.x, sel ? o.y : so.y, sel ? o.z : so.z);  // lanes outside the block borrow a ray of it
                const F3 rel = f3(Lr.x - oo.x, Lr.y - oo.y, Lr.z - oo.z);
                const float D = P.w.x * rel.x + P.w.y * rel.y + P.w.z * rel.z;
                const F3 e = f3(D * P.w.x - rel.x, D * P.w.y - rel.y, D * P.w.z - rel.z);
                const float Dmin = wave_min(D), Dmax = wave_max(D);
                const float Dfar = wave_max(sqrtf(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z)) * 1.00001f;
                const float rperp = wave_max(sqrtf(e.x * e.x + e.y * e.y + e.z * e.z)) * 1.0001f + 1e-5f * Dfar;
                const float chord = sqrtf(fmaxf(0.0f, 2.0f - 2.0f * P.c0)) * 1.0001f + 2e-5f;  // |ld - w| over the block
                for (int a0 = part * 64; a0 < S.naff && !out; a0 += 64 * parts) {
                    const int ai = min(a0 + lane, S.naff - 1);
                    const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
                    const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
                    bool pass = a0 + lane < S.naff;
                    {
                        const F3 w = P.w;
                        // rows of A: ra = (m0.x m0.y m0.z), rb = (m0.w m1.x m1.y), rc = (m1.z m1.w m2.x); q = w A
                        float qx = w.x * m0.x + w.y * m0.w + w.z * m1.z;
                        float qy = w.x * m0.y + w.y * m1.x + w.z * m1.w;
                        float qz = w.x * m0.z + w.y * m1.y + w.z * m2.x;
                        const float qn = sqrtf(qx * qx + qy * qy + qz * qz);
                        const float R = sqrtf(-m3.x) * 1.00001f;   // +inf for records that always pass, NaN never arises (m3.x <= 0)
                        const float nF = sqrtf(m0.x * m0.x + m0.y * m0.y + m0.z * m0.z + m0.w * m0.w + m1.x * m1.x + m1.y * m1.y +
                                               m1.z * m1.z + m1.w * m1.w + m2.x * m2.x) * 1.00001f;  // >= ||A||_2
                        const float qlow = qn * 0.99999f - chord * nF;  // <= |ld A| for every ray of the block
                        if (qn < 1e18f && R < 1e18f && nF < 1e18f && qlow > 0.25f * qn && qlow > 1e-18f) {
                            const float inv = 1.0f / qn;
                            qx *= inv; qy *= inv; qz *= inv;
                            // orthonormal pair around q (Duff et al.), good to ~1e-6
                            const float sg = (qz >= 0.0f) ? 1.0f : -1.0f;
                            const float aa = -1.0f / (sg + qz);
                            const float bb = qx * qy * aa;
                            const F3 ax1 = f3(1.0f + sg * qx * qx * aa, sg * bb, -sg * qx);
                            const F3 ax2 = f3(bb, sg + qy * qy * aa, -qy);
                            const F3 E = f3(m2.y - Lr.x, m2.z - Lr.y, m2.w - Lr.z);
                            // |c - o| <= far for every ray of the block: o = Lr - D w + e
                            const F3 f1 = f3(E.x + Dmin * w.x, E.y + Dmin * w.y, E.z + Dmin * w.z);
                            const F3 f2 = f3(E.x + Dmax * w.x, E.y + Dmax * w.y, E.z + Dmax * w.z);
                            const float far = sqrtf(fmaxf(f1.x * f1.x + f1.y * f1.y + f1.z * f1.z, f2.x * f2.x + f2.y * f2.y + f2.z * f2.z)) * 1.0001f + rperp;
                            // a.(delta A) = delta.(A a^T): |.| <= chord |A a^T|; 1e-5 covers the pair's own orthogonality error
                            const F3 v1 = f3(m0.x * ax1.x + m0.y * ax1.y + m0.z * ax1.z, m0.w * ax1.x + m1.x * ax1.y + m1.y * ax1.z, m1.z * ax1.x + m1.w * ax1.y + m2.x * ax1.z);
                            const F3 v2 = f3(m0.x * ax2.x + m0.y * ax2.y + m0.z * ax2.z, m0.w * ax2.x + m1.x * ax2.y + m1.y * ax2.z, m1.z * ax2.x + m1.w * ax2.y + m2.x * ax2.z);
                            const float g1 = (chord * sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z) + 1e-5f * nF) / qlow;
                            const float g2 = (chord * sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z) + 1e-5f * nF) / qlow;
                            const float base_T = (R + rperp + 0.0112f * far) * 1.0001f + 1e-5f * (far + Dfar);
                            const float T1 = base_T + g1 * far * 1.0001f, T2 = base_T + g2 * far * 1.0001f;
                            const float t1 = ax1.x * w.x + ax1.y * w.y + ax1.z * w.z, e1 = ax1.x * E.x + ax1.y * E.y + ax1.z * E.z;
                            const float t2 = ax2.x * w.x + ax2.y * w.y + ax2.z * w.z, e2 = ax2.x * E.x + ax2.y * E.y + ax2.z * E.z;
                            const float lo1 = fminf(-Dmin * t1, -Dmax * t1), hi1 = fmaxf(-Dmin * t1, -Dmax * t1);
                            const float lo2 = fminf(-Dmin * t2, -Dmax * t2), hi2 = fmaxf(-Dmin * t2, -Dmax * t2);
                            if (e1 < lo1 - T1 || e1 > hi1 + T1 || e2 < lo2 - T2 || e2 > hi2 + T2) pass = false;
                        }
                    }
                    unsigned long long am = __ballot(pass);
                    pk_batches++; pk_aff += __popcll(am);
                    while (am) {
                        const int u = __ffsll((long long)am) - 1;
                        am &= am - 1;
                        const float4 b0 = bcast4(m0, u), b1 = bcast4(m1, u), b2 = bcast4(m2, u), b3 = bcast4(m3, u);
                        if (!done && !(filt_affine(o, ld, b0, b1, b2, b3) < 0.0f)) {
                            if (exact_any(S, __float_as_int(b3.y), o, ld, nld)) {
                                done = true;
                                fo = f3(1e18f, 0.0f, 0.0f);
                                fd = f3(0.0f, 0.0f, 0.0f);
                            }
                        }
                    }
                    out = __all(done);
                }
            }
        }
        if (!sel) { done = done_keep; fo = fo_keep; fd = fd_keep; }
    }
    // Rays that fit no packet (abnormal ones included) are traced one at a time, TRANSPOSED: the ray is
    // broadcast and the 64 lanes take 64 different primitives each round, filter and exact test.
    const int pk_fb = __popcll(fallback);
    while (fallback) {
        const int r = __ffsll((long long)fallback) - 1;
        fallback &= fallback - 1;
        auto bc = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); };
        const F3 ro = f3(bc(o.x), bc(o.y), bc(o.z)), rl = f3(bc(ld.x), bc(ld.y), bc(ld.z)), rn = f3(bc(nld.x), bc(nld.y), bc(nld.z));
        const F3 rfo = f3(bc(fo.x), bc(fo.y), bc(fo.z)), rfd = f3(bc(fd.x), bc(fd.y), bc(fd.z));
        const bool rabn = __builtin_amdgcn_readlane((int)abn, r) != 0;
        bool blocked = false;
        for (int g = c0 + part * 64; g < c1 && !blocked; g += 64 * parts) {
            const int gi = g + lane;
            bool hit = false;
            if (gi < c1) {
                const float4 rec = S.sbounds[gi];
                if (rabn || !(filt(rfo, rfd, rec) < 0.0f)) hit = exact_any(S, S.sorder[gi], ro, rl, rn);
            }
            blocked = __any(hit);
        }
        for (int a0 = part * 64; a0 < S.naff && !blocked; a0 += 64 * parts) {
            const int ai = a0 + lane;
            bool hit = false;
            if (ai < S.naff) {
                const float4 m0 = S.aff_shadow[(size_t)ai * 4], m1 = S.aff_shadow[(size_t)ai * 4 + 1];
                const float4 m2 = S.aff_shadow[(size_t)ai * 4 + 2], m3 = S.aff_shadow[(size_t)ai * 4 + 3];
                if (rabn || !(filt_affine(ro, rl, m0, m1, m2, m3) < 0.0f)) hit = exact_any(S, __float_as_int(m3.y), ro, rl, rn);
            }
            blocked = __any(hit);
        }
        if (lane == r && blocked) done = true;
    }
    if (S.pstat && lane == 0) {
        atomicAdd(S.pstat + 4, (unsigned long long)pk_blocks); atomicAdd(S.pstat + 5, (unsigned long long)pk_ok + ((unsigned long long)pk_fb << 32));
        atomicAdd(S.pstat + 6, (unsigned long long)pk_batches); atomicAdd(S.pstat + 7, (unsigned long long)pk_surv + ((unsigned long long)pk_aff << 32));
    }
    if (live && done) lit[id] = 0;
}

// A thin generation (<= kSplitRays shadow rays) is traced by kSplit workgroups per 256 rays, each taking every
// kSplit-th cluster and affine batch: a ray is blocked if ANY part finds a blocker (each just clears lit[id]),
// and the serial walk per wave -- what bounds the run time when there are fewer waves than SIMDs can hold --
// is kSplit times shorter.
constexpr int kSplit = 4;
constexpr int kSplitRays = 1 << 21;

__global__ __launch_bounds__(kBlock) void k_shadow_pass(SceneDev S, ShadowQ Qi, const int* cnt_in, unsigned char* lit, int lit_cap)
{
    const int n = *cnt_in;
    const int parts = (n <= kSplitRays) ? kSplit : 1;
    const int base = (blockIdx.x / parts) * kBlock;
    if (base >= n) return;  // block-uniform
    shadow_pass_packet(S, Qi, n, base, lit, lit_cap, blockIdx.x % parts, parts);
}

// ------------------------------------------------------------------------------------------------
// K3 hit set-up and spawn (raytracing.cpp:436-526)
// ------------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_setup(SceneDev S, LevelDev L, LevelDev N, const int* nrays,
                                                 int* nnodes, int* nnext, int spawn)
{
    int i = blockIdx.x * BLOCK + threadIdx.x;
    bool live = i < *nrays;
    int pw = live ? L.prim[i] : -1;
    bool hit = pw >= 0;
    if (live && !hit) {  // :436-438: background, no ambient
        L.val[3 * i + 0] = 0.4f; L.val[3 * i + 1] = 0.4f; L.val[3 * i + 2] = 0.4f;
    }
    __shared__ int sh_alloc[2 * (BLOCK / 64) + 1];
    int node = block_alloc(hit, nnodes, sh_alloc);
    bool want_refl = false, want_refr = false;
    F3 o = f3(0, 0, 0), d = o, normal = o, ip = o, rd = o;
    float fres = 0.0f, fratio = 0.0f, eta = 1.0f;
    if (hit) {
        o = f3(L.ox[i], L.oy[i], L.oz[i]);
        d = f3(L.dx[i], L.dy[i], L.dz[i]);
        float nearest = L.t[i];
        bool tflag = (pw & kTFlag) != 0;
        int prim = pw & ~kTFlag;
        bool is_sphere = prim < S.ns;
        float reflectivity, transparency, ri;
        if (is_sphere) {
            const rtx_sphere& s = S.sph[prim];
            const rtx_xmatrix& im = S.sph_im[prim];
            // recompute the winner's object-space normal (:229-231, :395-398)
            F3 to = o, td = d;
            if (s.istransformed) { to = inv_apply(o, im); td = inv_apply(d, im); }
            td = normalized(td);
            normal = sphere_normal(to, td, nearest, s);
            if (s.istransformed) normal = inv_apply_normal(normal, im);
            reflectivity = s.reflectivity; transparency = s.transparency; ri = s.refractive_index;
        } else if (prim < S.ns + S.nt) {
            const float4* p = S.tri_hot + (size_t)(prim - S.ns) * 5;
            float4 q0 = p[0];
            float vd = q0.x * d.x + q0.y * d.y + q0.z * d.z;
            normal = (vd > 0.0f) ? f3(-q0.x, -q0.y, -q0.z) : f3(q0.x, q0.y, q0.z);  // :343-347
            const rtx_triangle& t = S.tri[prim - S.ns];
            reflectivity = t.reflectivity; transparency = t.transparency; ri = t.refractive_index;
        } else {
            const float4 q0 = S.phot[(size_t)prim * 5];
            normal = f3(q0.x, q0.y, q0.z);  // :311: the stored normal, negated
            const rtx_polygon& g = S.poly[prim - S.ns - S.nt];
            reflectivity = g.reflectivity; transparency = g.transparency; ri = g.refractive_index;
        }
        ip = f3(o.x + nearest * d.x, o.y + nearest * d.y, o.z + nearest * d.z);  // :440
        if (tflag) ip = fwd_apply(ip, S.sph[prim].mat);                          // :441-444
        normal = normalized(normal);                                             // :445
        if (spawn) {  // depth < MAX_depth (:449)
            bool insphere = false;
            rd = d;
            if (is_sphere) {
                if (tflag) rd = inv_apply(d, S.sph_im[prim]);  // :452-455
                rd = normalized(rd);
                if (dot3(normal, rd) > 0.0f) {                 // :458-461
                    normal = f3(-normal.x, -normal.y, -normal.z);
                    insphere = true;
                }
            }
            want_refl = reflectivity > 0.0f;
            want_refr = transparency > 0.0f;
            if (want_refl || want_refr) {
                fratio = -(dot3(rd, normal));
                fres = fresnel_term(fratio);
                eta = ri;
                if (insphere) eta = (float)(1.0 / (double)ri);  // :477-479 (spheres only)
            }
        }
        L.n_ray[node] = i;
        L.n_nx[node] = normal.x; L.n_ny[node] = normal.y; L.n_nz[node] = normal.z;
        L.n_px[node] = ip.x; L.n_py[node] = ip.y; L.n_pz[node] = ip.z;
        L.n_fres[node] = fres;
    }
    if (!spawn) {
        if (hit) { L.n_crefl[node] = -1; L.n_crefr[node] = -1; }
        return;
    }
    // both kinds of child from ONE allocation per workgroup: reflection rays take the first slots of the
    // workgroup's run, refraction rays the rest (two barriers instead of six)
    int crefl = -1, crefr = -1;
    {
        const unsigned long long m1 = __ballot(want_refl), m2 = __ballot(want_refr);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        constexpr int nw = BLOCK / 64;
        const int p1 = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0));
        const int p2 = __builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0));
        if (lane == 0) { sh_alloc[w] = __popcll(m1); sh_alloc[nw + w] = __popcll(m2); }  // free: block_alloc ended with a barrier
        __syncthreads();
        int off1 = 0, off2 = 0, tot1 = 0, tot2 = 0;
        for (int k = 0; k < nw; k++) {
            if (k < w) { off1 += sh_alloc[k]; off2 += sh_alloc[nw + k]; }
            tot1 += sh_alloc[k]; tot2 += sh_alloc[nw + k];
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_alloc[2 * nw] = (tot1 + tot2) ? atomicAdd(nnext, tot1 + tot2) : 0;
        __syncthreads();
        const int base = sh_alloc[2 * nw];
        if (want_refl) crefl = base + off1 + p1;
        if (want_refr) crefr = base + tot1 + off2 + p2;
    }
    if (hit) { L.n_crefl[node] = crefl; L.n_crefr[node] = crefr; }
    if (want_refl) {  // :464-468
        F3 r = f3(rd.x + 2.0f * fratio * normal.x, rd.y + 2.0f * fratio * normal.y, rd.z + 2.0f * fratio * normal.z);
        r = normalized(r);
        N.ox[crefl] = ip.x + normal.x * kBias; N.oy[crefl] = ip.y + normal.y * kBias; N.oz[crefl] = ip.z + normal.z * kBias;
        N.dx[crefl] = r.x; N.dy[crefl] = r.y; N.dz[crefl] = r.z;
    }
    if (want_refr) {  // :476-488
        float raydotn = fratio;
        float k = (float)(1.0 - (double)(eta * eta) * (1.0 - (double)(raydotn * raydotn)));
        float coef = eta * raydotn - sqrtf(k);  // NaN when k < 0: the child then misses everything
        F3 r = f3(rd.x * eta + normal.x * coef, rd.y * eta + normal.y * coef, rd.z * eta + normal.z * coef);
        r = normalized(r);
        N.ox[crefr] = ip.x - normal.x * kBias; N.oy[crefr] = ip.y - normal.y * kBias; N.oz[crefr] = ip.z - normal.z * kBias;
        N.dx[crefr] = r.x; N.dy[crefr] = r.y; N.dz[crefr] = r.z;
    }
}

// ------------------------------------------------------------------------------------------------
// K4 shadow any-hit (raytracing.cpp:560-589): one thread per (hit, light); blockIdx.y = light.
// No distance limit: any primitive at t >= 0 along the ray blocks the light.
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_shadow(SceneDev S, LevelDev L, const int* nnodes, unsigned long long* tests)
{
    int j = blockIdx.x * kBlock + threadIdx.x;
    bool live = j < *nnodes;
    if (!COUNT && !live) return;
    unsigned int cnt[4] = {0, 0, 0, 0};
    if (live) {
    int li = blockIdx.y;
    F3 ip = f3(L.n_px[j], L.n_py[j], L.n_pz[j]);
    F3 n = f3(L.n_nx[j], L.n_ny[j], L.n_nz[j]);
    const rtx_light& Lt = S.lights[li];
    F3 ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));  // :561-562
    F3 o = f3(ip.x + n.x * kBias, ip.y + n.y * kBias, ip.z + n.z * kBias);            // :568
    F3 nld = normalized(ld);  // :575 for untransformed spheres
    bool shadow = false;
    for (int s = 0; s < S.ns && !shadow; s++) {
        float4 h = S.sph_hot[s];
        F3 td = nld;
        if (S.sph_xf[s]) td = normalized(inv_apply(ld, S.sph_im[s]));  // :572-575; origin NOT transformed
        float t;
        if (COUNT) cnt[S.sph_xf[s] ? 1 : 0]++;
        shadow = sphere_test(o, td, h.x, h.y, h.z, h.w, t);
    }
    for (int k = 0; k < S.nt && !shadow; k++) {
        TriRec T = load_tri(S.tri_hot, k);
        float t = 0.0f, vd;
        shadow = triangle_test(o, ld, T, t, vd);
        if (COUNT) cnt[(vd == 0.0f || t < 0.0f) ? 3 : 2]++;
    }
    L.lit[(size_t)li * L.cap + j] = shadow ? 0 : 1;
    }
    if (COUNT) flush_counts(cnt, tests);
}

// ------------------------------------------------------------------------------------------------
// K5 unwind (raytracing.cpp:446, :470-472, :490-492, :590-646): colour of one hit from its
// children's colours, in the reference's add order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_unwind(SceneDev S, LevelDev L, LevelDev N, const int* nnodes)
{
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= *nnodes) return;
    int i = L.n_ray[j];
    int pw = L.prim[i];
    int prim = pw & ~kTFlag;
    bool is_sphere = prim < S.ns;
    F3 d = f3(L.dx[i], L.dy[i], L.dz[i]);
    F3 n = f3(L.n_nx[j], L.n_ny[j], L.n_nz[j]);
    F3 ip = f3(L.n_px[j], L.n_py[j], L.n_pz[j]);
    F3 color, spec, diff;
    float transparency, specexpo;
    bool xf = false;
    if (is_sphere) {
        const rtx_sphere& s = S.sph[prim];
        color = f3(s.color); spec = f3(s.speccoeff); diff = f3(s.diffcolor);
        transparency = s.transparency; specexpo = s.specexpo;
        xf = s.istransformed != 0;
    } else if (prim < S.ns + S.nt) {
        const rtx_triangle& t = S.tri[prim - S.ns];
        color = f3(t.color); spec = f3(t.speccoeff); diff = f3(t.diffcolor);
        transparency = t.transparency; specexpo = t.specexpo;
    } else {  // :527-556, :627-639: the triangle forms with the polygon's material
        const rtx_polygon& g = S.poly[prim - S.ns - S.nt];
        color = f3(g.color); spec = f3(g.speccoeff); diff = f3(g.diffcolor);
        transparency = g.transparency; specexpo = g.specexpo;
    }
    F3 pix = f3(0.0f, 0.0f, 0.0f);
    float fres = L.n_fres[j];
    int crefl = L.n_crefl[j], crefr = L.n_crefr[j];
    if (crefl >= 0) {  // :470-472
        pix.x += N.val[3 * crefl + 0] * fres * color.x;
        pix.y += N.val[3 * crefl + 1] * fres * color.y;
        pix.z += N.val[3 * crefl + 2] * fres * color.z;
    }
    if (crefr >= 0) {  // :490-492: (1.0 - fresnel) makes the whole right-hand side double
        double w = 1.0 - (double)fres;
        pix.x = (float)((double)pix.x + (double)N.val[3 * crefr + 0] * w * (double)transparency * (double)color.x);
        pix.y = (float)((double)pix.y + (double)N.val[3 * crefr + 1] * w * (double)transparency * (double)color.y);
        pix.z = (float)((double)pix.z + (double)N.val[3 * crefr + 2] * w * (double)transparency * (double)color.z);
    }
    // view direction used by the specular term (:592-599 spheres, :619 triangles)
    F3 vdir = d;
    if (is_sphere) {
        if (xf) vdir = inv_apply(d, S.sph_im[prim]);
        vdir = normalized(vdir);
    }
    for (int li = 0; li < S.nl; li++) {
        if (!L.lit[(size_t)li * L.cap + j]) continue;
        const rtx_light& Lt = S.lights[li];
        F3 lc = f3(Lt.color);
        F3 ld = normalized(f3(Lt.point.x - ip.x, Lt.point.y - ip.y, Lt.point.z - ip.z));
        float dp = dot3(n, ld);
        float dotprod = (0.0f < dp) ? dp : 0.0f;  // std::max(float(0.0), dp): NaN -> 0
        F3 ldir = ld;
        if (is_sphere) {
            if (xf) ldir = inv_apply(ld, S.sph_im[prim]);
            ldir = normalized(ldir);
        }
        float ndoti2 = 2.0f * (dot3(ldir, n));
        F3 rr = normalized(f3(ndoti2 * n.x - ldir.x, ndoti2 * n.y - ldir.y, ndoti2 * n.z - ldir.z));
        float rdotv = -(dot3(vdir, rr));
        float pw_ = powf_rn(rdotv, specexpo);
        pix.x += spec.x * pw_ * lc.x;
        pix.y += spec.y * pw_ * lc.y;
        pix.z += spec.z * pw_ * lc.z;
        pix.x += dotprod * diff.x * lc.x;
        pix.y += dotprod * diff.y * lc.y;
        pix.z += dotprod * diff.z * lc.z;
    }
    pix.x += S.ambient.x; pix.y += S.ambient.y; pix.z += S.ambient.z;  // :643-645
    L.val[3 * i + 0] = pix.x; L.val[3 * i + 1] = pix.y; L.val[3 * i + 2] = pix.z;
}

// ------------------------------------------------------------------------------------------------
// K6 3x3 antialias + quantise (raytracing.cpp:745-780, :792-794).  The reference re-traces the
// neighbour rays; raytrace() is a pure function of the pixel, so filtering the 1-spp image in
// the same row-major add order is byte-identical.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_aa_quantise(const float* fimg, const int* row2ti, const int* own_rows,
                                                        int nown, int width, int height, unsigned char* out)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= nown * width) return;
    int r = i / width, x = i - r * width;
    int y = own_rows[r];
    float counter = 0.0f, ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int yv = y - 1; yv <= y + 1; yv++)
        for (int xv = x - 1; xv <= x + 1; xv++)
            if (xv >= 0 && yv >= 0 && xv < width && yv < height) {
                const float* p = fimg + ((size_t)row2ti[yv] * width + xv) * 3;
                counter = counter + 1.0f;
                if (yv == y && xv == x) { ax += 2.0f * p[0]; ay += 2.0f * p[1]; az += 2.0f * p[2]; }
                else { ax += p[0]; ay += p[1]; az += p[2]; }
            }
    float dv = counter + 1.0f;
    ax = ax / dv; ay = ay / dv; az = az / dv;
    out[3 * (size_t)i + 0] = quantise(ax);
    out[3 * (size_t)i + 1] = quantise(ay);
    out[3 * (size_t)i + 2] = quantise(az);
}

// counters of one chunk -> 64-bit frame totals
__global__ void k_accumulate(const int* cnt, unsigned long long* totals, int levels)
{
    int k = threadIdx.x;
    if (k < 2 * levels) totals[k] += (unsigned long long)cnt[k];
}

}  // namespace rtx

// ================================================================================================
// host side: resident context
// ================================================================================================
using namespace rtx;

struct rtx_context {
    int device = 0;
    rtx_params params{};
    int width = 0, height = 0, levels = 1;
    uint64_t singular = 0;
    int fanout = 0;
    std::vector<int> own_rows, trace_rows, row2ti;
    std::vector<void*> allocs;
    SceneDev S{};
    CamDev C{};
    std::vector<LevelDev> L;
    int chunk_rows = 0;
    int* d_trace_rows = nullptr; int* d_row2ti = nullptr; int* d_own_rows = nullptr;
    float* d_fimg = nullptr;
    unsigned char* d_rgb8 = nullptr;
    int* d_cnt = nullptr;                   // [2*levels]: rays per level, then nodes per level
    unsigned long long* d_totals = nullptr; // [2*levels]
    unsigned long long* d_tests = nullptr;  // [8] closest kinds, shadow kinds (RTX_F_COUNT_TESTS)
    int binG = 256, nbins = 0;              // light-space bins: nl * G * G
    unsigned long long* d_pstat = nullptr;
    int* d_perm = nullptr; int* d_dummy = nullptr;
    // The shadow rays of generation k and the closest-hit pass of generation k+1 depend on nothing but
    // k_setup(k): generation k's shadow chain (bin, scan, scatter, pass) runs on side stream k % kSide with
    // that stream's own queue and scan buffers (fork after every set-up, one join per side stream before
    // the unwind), so the thin, latency-bound kernels of the deep generations overlap instead of queueing.
    // The ray binning of the closest-hit pass has scan buffers of its own for the same reason.
    static constexpr int kSide = 3;
    struct ShadowLane {
        hipStream_t stream = nullptr;
        hipEvent_t join = nullptr;
        ShadowQ q = {};
        int* hist = nullptr; int* offs = nullptr; int* rank = nullptr; int* bsum = nullptr; int* braw = nullptr;
        bool used = false;
    } side[kSide];
    std::vector<hipEvent_t> ev_fork;
    int* r_hist = nullptr; int* r_offs = nullptr; int* r_rank = nullptr; int* r_bsum = nullptr; int* r_braw = nullptr;
    size_t qpad = 0;                        // slots the 64-alignment of the shadow bin groups can add to a queue
    int* d_scnt = nullptr;                  // [nchunks*levels] slots (rays + alignment padding) of each generation's shadow queue
    int nframe_chunks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;            // RTX_F_TIME_KERNELS: (start, stop) per hot launch
    std::vector<int> kev_kind;              // 0 closest (packet / exact), 1 shadow, 2 closest per-ray; per pair used by the last render
    hipStream_t last_stream = nullptr;
    bool rendered = false;
    int nchunks = 0;

    template <typename T> int dalloc(T** p, size_t n)
    {
        void* q = nullptr;
        if (n == 0) n = 1;
        HIPCHK(hipMalloc(&q, n * sizeof(T)));
        allocs.push_back(q);
        *p = (T*)q;
        return RTX_OK;
    }
    template <typename T> int upload(T** p, const std::vector<T>& v)
    {
        int rc = dalloc(p, v.size());
        if (rc) return rc;
        if (!v.empty()) HIPCHK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return RTX_OK;
    }
};

static int validate(const rtx_scene* sc, const rtx_params* p)
{
    if (!sc || !p) return RTX_E_INVALID_ARG;
    if (p->width <= 0 || p->height <= 0 || p->max_depth < 0) return RTX_E_INVALID_ARG;
    if (p->max_depth + 1 > RTX_MAX_LEVELS) return RTX_E_TOO_LARGE;
    if ((int64_t)p->width * p->height > (int64_t)1 << 30) return RTX_E_TOO_LARGE;
    if (sc->nspheres < 0 || sc->ntriangles < 0 || sc->nlights < 0) return RTX_E_INVALID_ARG;
    if ((sc->nspheres && !sc->spheres) || (sc->ntriangles && !sc->triangles) || (sc->nlights && !sc->lights))
        return RTX_E_INVALID_ARG;
    if (sc->npolygons < 0 || sc->npolygon_vertices < 0) return RTX_E_INVALID_ARG;
    if ((sc->npolygons && !sc->polygons) || (sc->npolygon_vertices && !sc->polygon_vertices)) return RTX_E_INVALID_ARG;
    for (int i = 0; i < sc->npolygons; i++) {
        const rtx_polygon& g = sc->polygons[i];
        if (g.numvertices < 0 || g.first_vertex < 0 || (int64_t)g.first_vertex + g.numvertices > (int64_t)sc->npolygon_vertices)
            return RTX_E_INVALID_ARG;
    }
    if ((int64_t)sc->nspheres + sc->ntriangles + sc->npolygons >= (int64_t)kTFlag) return RTX_E_TOO_LARGE;
    if (sc->nlights > 65535) return RTX_E_TOO_LARGE;  // gridDim.y
    int nr = p->nranks <= 1 ? 1 : p->nranks;
    int rk = p->nranks <= 1 ? 0 : p->rank;
    if (rk < 0 || rk >= nr) return RTX_E_INVALID_ARG;
    return RTX_OK;
}

static int context_create_impl(const rtx_scene* scene, const rtx_params* params, rtx_context* c)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RTX_E_NO_DEVICE;
    int dev = params->device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev) return RTX_E_INVALID_ARG;
    HIPCHK(hipSetDevice(dev));
    c->device = dev;
    c->params = *params;
    c->width = params->width;
    c->height = params->height;
    const int W = c->width, H = c->height;

    // rows: owned rows, plus a one-row halo either side so the 3x3 filter needs no exchange
    int nown = rtx_band_rows(H, params->nranks, params->rank, params->band_rows, nullptr, 0);
    if (nown < 0) return nown;
    c->own_rows.resize(nown);
    if (nown) rtx_band_rows(H, params->nranks, params->rank, params->band_rows, c->own_rows.data(), nown);
    c->row2ti.assign(H, -1);
    std::vector<char> need(H, 0);
    const int halo = (params->flags & RTX_F_NO_HALO) ? 0 : 1;
    for (int y : c->own_rows)
        for (int yy = y - halo; yy <= y + halo; yy++)
            if (yy >= 0 && yy < H) need[yy] = 1;
    for (int y = 0; y < H; y++)
        if (need[y]) { c->row2ti[y] = (int)c->trace_rows.size(); c->trace_rows.push_back(y); }
    const int ntrace = (int)c->trace_rows.size();

    PackedScene P;
    pack_scene(*scene, W, H, P);
    c->singular = P.singular;
    c->fanout = P.max_fanout;
    c->levels = (P.max_fanout == 0) ? 1 : params->max_depth + 1;
    const int levels = c->levels;

    // scene upload
    int rc;
    {
        float4* hot = nullptr;
        static_assert(sizeof(SphHot) == sizeof(float4), "sphere hot record is one float4");
        static_assert(sizeof(TriHot) == 5 * sizeof(float4), "triangle hot record is five float4");
        if ((rc = c->dalloc(&hot, P.sph_hot.size()))) return rc;
        if (!P.sph_hot.empty()) HIPCHK(hipMemcpy(hot, P.sph_hot.data(), P.sph_hot.size() * sizeof(float4), hipMemcpyHostToDevice));
        c->S.sph_hot = hot;
        float4* thot = nullptr;
        if ((rc = c->dalloc(&thot, P.tri_hot.size() * 5))) return rc;
        if (!P.tri_hot.empty()) HIPCHK(hipMemcpy(thot, P.tri_hot.data(), P.tri_hot.size() * sizeof(TriHot), hipMemcpyHostToDevice));
        c->S.tri_hot = thot;
        float4* ph = nullptr;
        if ((rc = c->dalloc(&ph, P.prim_hot.size() * 5))) return rc;
        if (!P.prim_hot.empty()) HIPCHK(hipMemcpy(ph, P.prim_hot.data(), P.prim_hot.size() * sizeof(TriHot), hipMemcpyHostToDevice));
        c->S.phot = ph;
        int* xf = nullptr; rtx_xmatrix* im = nullptr; rtx_sphere* sp = nullptr; rtx_triangle* tr = nullptr; rtx_light* lt = nullptr;
        if ((rc = c->upload(&xf, P.sph_xf))) return rc;
        if ((rc = c->upload(&im, P.sph_im))) return rc;
        if ((rc = c->upload(&sp, P.sph))) return rc;
        if ((rc = c->upload(&tr, P.tri))) return rc;
        if ((rc = c->upload(&lt, P.lights))) return rc;
        c->S.sph_xf = xf; c->S.sph_im = im; c->S.sph = sp; c->S.tri = tr; c->S.lights = lt;
        {
            rtx_polygon* pg = nullptr; float* pv = nullptr;
            if ((rc = c->upload(&pg, P.poly))) return rc;
            if ((rc = c->upload(&pv, P.poly_verts))) return rc;
            c->S.poly = pg; c->S.poly_verts = reinterpret_cast<const float4*>(pv); c->S.np = (int)P.poly.size();
        }
        c->S.ns = (int)P.sph.size(); c->S.nt = (int)P.tri.size(); c->S.nl = (int)P.lights.size();
        c->S.ambient = {P.ambient.x, P.ambient.y, P.ambient.z};
        static_assert(sizeof(BoundSphere) == sizeof(float4), "filter record is one float4");
        float4* bnd = nullptr;
        if ((rc = c->dalloc(&bnd, P.bounds.size()))) return rc;
        if (!P.bounds.empty()) HIPCHK(hipMemcpy(bnd, P.bounds.data(), P.bounds.size() * sizeof(float4), hipMemcpyHostToDevice));
        c->S.bounds = bnd;
        static_assert(sizeof(AffineShadowRec) == 4 * sizeof(float4), "affine shadow record is four float4");
        float4* aff = nullptr;
        if ((rc = c->dalloc(&aff, P.aff_shadow.size() * 4))) return rc;
        if (!P.aff_shadow.empty()) HIPCHK(hipMemcpy(aff, P.aff_shadow.data(), P.aff_shadow.size() * sizeof(AffineShadowRec), hipMemcpyHostToDevice));
        c->S.aff_shadow = aff;
        c->S.naff = (int)P.aff_shadow.size();
        float4* sb = nullptr; int* so = nullptr;
        if ((rc = c->dalloc(&sb, P.sbounds.size()))) return rc;
        if (!P.sbounds.empty()) HIPCHK(hipMemcpy(sb, P.sbounds.data(), P.sbounds.size() * sizeof(float4), hipMemcpyHostToDevice));
        if ((rc = c->upload(&so, P.sorder))) return rc;
        c->S.sbounds = sb; c->S.sorder = so; c->S.nsb = (int)P.sbounds.size();
        c->S.guard = P.guard;
        {
            auto up4 = [&](const std::vector<BoundSphere>& v, const float4** dst) -> int {
                float4* q = nullptr;
                int r = c->dalloc(&q, v.size());
                if (r) return r;
                if (!v.empty()) HIPCHK(hipMemcpy(q, v.data(), v.size() * sizeof(float4), hipMemcpyHostToDevice));
                *dst = q;
                return RTX_OK;
            };
            int* q = nullptr;
            if ((rc = up4(P.cc.cl, &c->S.ccl))) return rc;
            if ((rc = up4(P.cc.mem, &c->S.cmem))) return rc;
            if ((rc = c->upload(&q, P.cc.idx))) return rc;
            c->S.cidx = q;
            if ((rc = c->upload(&q, P.cc.seg_first))) return rc;
            c->S.cseg = q;
            c->S.cnseg = (int)P.cc.seg_first.size() - 1;
            if ((rc = up4(P.cc.clR, &c->S.cclR))) return rc;
            if ((rc = up4(P.cc.subR, &c->S.csubR))) return rc;
            if ((rc = c->upload(&q, P.cc.order))) return rc;
            c->S.corder = q;
            c->S.cnfirst = P.cc.nfirst;
            if ((rc = up4(P.sc.cl, &c->S.scl))) return rc;
            if ((rc = up4(P.sc.mem, &c->S.smem))) return rc;
            if ((rc = c->upload(&q, P.sc.idx))) return rc;
            c->S.sidx = q;
            c->S.sncl = (int)P.sc.cl.size();
        }
        {   // scene box for ray binning: every finite bounding sphere, 16 cells per axis
            float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            for (const BoundSphere& b : P.bounds) {
                if (!std::isfinite(b.neg_r2)) continue;
                float r = std::sqrt(-b.neg_r2);
                const float cc[3] = {b.cx, b.cy, b.cz};
                for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], cc[a] - r); hi[a] = std::max(hi[a], cc[a] + r); }
            }
            float inv[3];
            for (int a = 0; a < 3; a++) {
                if (!(hi[a] > lo[a])) { lo[a] = 0.0f; hi[a] = 1.0f; }
                inv[a] = 16.0f / (hi[a] - lo[a]);
            }
            c->S.box_lo = {lo[0], lo[1], lo[2]};
            c->S.box_inv = {inv[0], inv[1], inv[2]};
        }
        float* ix = nullptr; float* iy = nullptr;
        if ((rc = c->upload(&ix, P.imagex))) return rc;
        if ((rc = c->upload(&iy, P.imagey))) return rc;
        c->C.imagex = ix; c->C.imagey = iy;
        c->C.ncrossv = {P.ncrossv.x, P.ncrossv.y, P.ncrossv.z};
        c->C.up = {P.up.x, P.up.y, P.up.z};
        c->C.vpn = {P.vpn.x, P.vpn.y, P.vpn.z};
        c->C.camera = {P.camera.x, P.camera.y, P.camera.z};
        c->C.orig = {P.orig.x, P.orig.y, P.orig.z};
    }
    if ((rc = c->upload(&c->d_trace_rows, c->trace_rows))) return rc;
    if ((rc = c->upload(&c->d_row2ti, c->row2ti))) return rc;
    if ((rc = c->upload(&c->d_own_rows, c->own_rows))) return rc;
    if ((rc = c->dalloc(&c->d_fimg, (size_t)ntrace * W * 3))) return rc;
    if ((rc = c->dalloc(&c->d_rgb8, (size_t)nown * W * 3))) return rc;
    if ((rc = c->dalloc(&c->d_cnt, (size_t)2 * levels))) return rc;
    if ((rc = c->dalloc(&c->d_totals, (size_t)2 * levels))) return rc;
    if ((rc = c->dalloc(&c->d_tests, (size_t)8))) return rc;
    if ((rc = c->dalloc(&c->d_pstat, (size_t)8))) return rc;
    c->S.pstat = (params->flags & RTX_F_PACKET_STATS) ? c->d_pstat : nullptr;

    // chunking: level k holds at most fanout^k rays per level-0 ray
    const size_t slot_bytes = 24 + 4 + 4 + 12 + 4 + 24 + 8 + 4 + (size_t)c->S.nl;
    double mult = 0, m = 1;
    for (int k = 0; k < levels; k++) { mult += m; m *= (c->fanout >= 2 ? 2.0 : 1.0); }
    uint64_t budget = params->workspace_bytes ? params->workspace_bytes : ((uint64_t)32 << 30);
    // per level-0 ray: every level's slots, plus the side streams' shadow-ray queues (28 B a ray + 4 B rank) sized for the largest level
    double per_row = (double)W * (mult * (double)slot_bytes + (m / (c->fanout >= 2 ? 2.0 : 1.0)) * (double)std::min(levels, (int)rtx_context::kSide) * 32.0 * (double)c->S.nl);
    int64_t rows = (int64_t)((double)budget / per_row);
    if (rows < 1) rows = 1;
    if (rows > ntrace) rows = ntrace;
    // the deepest level's capacity must stay indexable by int
    while (rows > 1 && (double)rows * W * (m / (c->fanout >= 2 ? 2.0 : 1.0)) * std::max(c->S.nl, 1) > (double)((int64_t)1 << 30)) rows /= 2;
    c->chunk_rows = (int)std::max<int64_t>(rows, 1);
    if (ntrace == 0) c->chunk_rows = 1;

    c->L.resize(levels);
    size_t cap = (size_t)c->chunk_rows * W;
    for (int k = 0; k < levels; k++) {
        LevelDev& L = c->L[k];
        if (cap > ((size_t)1 << 30)) return RTX_E_TOO_LARGE;
        L.cap = (int)cap;
        float** fl[] = {&L.ox, &L.oy, &L.oz, &L.dx, &L.dy, &L.dz, &L.t, &L.n_nx, &L.n_ny, &L.n_nz, &L.n_px, &L.n_py, &L.n_pz, &L.n_fres};
        for (float** f : fl)
            if ((rc = c->dalloc(f, cap))) return rc;
        int** il[] = {&L.prim, &L.n_ray, &L.n_crefl, &L.n_crefr};
        for (int** f : il)
            if ((rc = c->dalloc(f, cap))) return rc;
        if (k == 0) L.val = nullptr;  // level-0 colours are written straight into the float image
        else if ((rc = c->dalloc(&L.val, cap * 3))) return rc;
        if ((rc = c->dalloc(&L.lit, cap * (size_t)std::max(c->S.nl, 1)))) return rc;
        if (c->fanout >= 2) cap *= 2;
    }
    {   // shadow-ray queues: the largest generation's hits x lights
        size_t qcap = (size_t)c->L[levels - 1].cap * (size_t)std::max(c->S.nl, 1);
        c->binG = 256;
        while ((size_t)std::max(c->S.nl, 1) * c->binG * c->binG > ((size_t)1 << 22) && c->binG > 8) c->binG /= 2;
        c->nbins = std::max(c->S.nl, 1) * c->binG * c->binG;
        c->qpad = (size_t)64 * ((size_t)(c->nbins + 1023) / 1024 + 1);
        qcap += c->qpad;
        if (qcap > ((size_t)1 << 31) - 1) return RTX_E_TOO_LARGE;
        const size_t maxbins = (size_t)c->nbins;
        const int nside = std::min(levels, (int)rtx_context::kSide);
        for (int i = 0; i < nside; i++) {
            rtx_context::ShadowLane& sl = c->side[i];
            float** fl[] = {&sl.q.ox, &sl.q.oy, &sl.q.oz, &sl.q.dx, &sl.q.dy, &sl.q.dz};
            for (float** f : fl)
                if ((rc = c->dalloc(f, qcap))) return rc;
            if ((rc = c->dalloc(&sl.q.id, qcap))) return rc;
            if ((rc = c->dalloc(&sl.hist, maxbins))) return rc;
            if ((rc = c->dalloc(&sl.offs, maxbins))) return rc;
            if ((rc = c->dalloc(&sl.rank, qcap))) return rc;
            if ((rc = c->dalloc(&sl.bsum, (maxbins + 1023) / 1024))) return rc;
            if ((rc = c->dalloc(&sl.braw, (maxbins + 1023) / 1024))) return rc;
        }
        if ((rc = c->dalloc(&c->r_hist, (size_t)kRayBins))) return rc;
        if ((rc = c->dalloc(&c->r_offs, (size_t)kRayBins))) return rc;
        if ((rc = c->dalloc(&c->r_rank, (size_t)c->L[levels - 1].cap))) return rc;
        if ((rc = c->dalloc(&c->r_bsum, (size_t)kRayBins / 1024))) return rc;
        if ((rc = c->dalloc(&c->r_braw, (size_t)kRayBins / 1024))) return rc;
        if ((rc = c->dalloc(&c->d_perm, (size_t)c->L[levels - 1].cap))) return rc;
        if ((rc = c->dalloc(&c->d_dummy, (size_t)1))) return rc;
        c->nframe_chunks = ntrace ? (ntrace + c->chunk_rows - 1) / c->chunk_rows : 0;
        if ((rc = c->dalloc(&c->d_scnt, (size_t)std::max(c->nframe_chunks, 1) * levels))) return rc;
    }
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    for (int i = 0; i < std::min(levels, (int)rtx_context::kSide); i++) {
        HIPCHK(hipStreamCreateWithFlags(&c->side[i].stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->side[i].join, hipEventDisableTiming));
    }
    c->ev_fork.resize(levels);
    for (auto& e : c->ev_fork) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (params->flags & RTX_F_TIME_KERNELS) {
        int nchunks = ntrace ? (ntrace + c->chunk_rows - 1) / c->chunk_rows : 0;
        c->kev.resize((size_t)nchunks * levels * 2 * 2);  // (start, stop) x (closest, shadow) per generation and chunk
        for (auto& e : c->kev) HIPCHK(hipEventCreate(&e));
    }
    return RTX_OK;
}

static inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
// intersection kernels: thin generations (<= kThinRays live rays) use 64 rays per workgroup
static inline unsigned grid_for_isect(size_t cap) { return std::max(grid_for(cap), (unsigned)((std::min<size_t>(cap, kThinRays) + 63) / 64)); }

static int context_render_impl(rtx_context* c, hipStream_t st, uint8_t* d_out)
{
    HIPCHK(hipSetDevice(c->device));
    const int W = c->width, levels = c->levels;
    const int ntrace = (int)c->trace_rows.size(), nown = (int)c->own_rows.size();
    HIPCHK(hipEventRecord(c->ev0, st));
    HIPCHK(hipMemsetAsync(c->d_totals, 0, sizeof(unsigned long long) * 2 * levels, st));
    HIPCHK(hipMemsetAsync(c->d_tests, 0, sizeof(unsigned long long) * 8, st));
    HIPCHK(hipMemsetAsync(c->d_pstat, 0, sizeof(unsigned long long) * 8, st));
    HIPCHK(hipMemsetAsync(c->d_scnt, 0, sizeof(int) * (size_t)std::max(c->nframe_chunks, 1) * levels, st));
    const bool count = (c->params.flags & RTX_F_COUNT_TESTS) != 0;
    const bool exact_only = (c->params.flags & RTX_F_EXACT_ONLY) != 0;
    const bool timek = !c->kev.empty();
    size_t kev_used = 0;
    c->kev_kind.clear();
    int nchunks = 0;
    for (int ti0 = 0; ti0 < ntrace; ti0 += c->chunk_rows, nchunks++) {
        int nrows = std::min(c->chunk_rows, ntrace - ti0);
        size_t npix = (size_t)nrows * W;
        int* rays = c->d_cnt;
        int* nodes = c->d_cnt + levels;
        HIPCHK(hipMemsetAsync(c->d_cnt, 0, sizeof(int) * 2 * levels, st));
        LevelDev L0 = c->L[0];
        L0.val = c->d_fimg + (size_t)ti0 * W * 3;
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(npix)), dim3(kBlock), 0, st, L0, c->C, c->d_trace_rows, ti0, nrows, W, rays + 0);
        size_t live_cap = npix;  // upper bound on live rays at this level
        std::vector<size_t> bound(levels);
        for (int k = 0; k < levels; k++) {
            LevelDev Lk = (k == 0) ? L0 : c->L[k];
            LevelDev Ln = (k + 1 < levels) ? c->L[k + 1] : Lk;
            bound[k] = live_cap;
            int spawn = (k < c->params.max_depth && k + 1 < levels) ? 1 : 0;
            unsigned g = grid_for(live_cap);
            const int* perm = nullptr;
            if (k > 0 && !count && !exact_only) {
                // bin the secondary rays by (origin cell, direction cell) so that waves are coherent packets
                HIPCHK(hipMemsetAsync(c->r_hist, 0, sizeof(int) * (size_t)kRayBins, st));
                hipLaunchKernelGGL(k_ray_bin, dim3(g), dim3(kBlock), 0, st, c->S, Lk, rays + k, c->r_hist, c->r_rank);
                const int nb = kRayBins / 1024;
                hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, st, c->r_hist, c->r_offs, kRayBins, c->r_bsum, c->r_braw, 1);
                hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, c->r_bsum, nb, c->d_dummy);
                hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, st, c->r_offs, kRayBins, c->r_bsum, c->r_braw, 1, (int*)nullptr);
                hipLaunchKernelGGL(k_ray_scatter, dim3(g), dim3(kBlock), 0, st, c->S, Lk, rays + k, c->r_offs, c->r_rank, c->d_perm);
                perm = c->d_perm;
            }
            if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used], st)); }
            if (count) hipLaunchKernelGGL(k_closest<true>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, rays + k, c->d_tests);
            else if (exact_only) hipLaunchKernelGGL(k_closest<false>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, rays + k, c->d_tests);
            else if (k > 0) hipLaunchKernelGGL(k_closest_lanes, dim3(std::max(g, (unsigned)((std::min<size_t>(live_cap, (size_t)kSplitClosest) + 63) / 64))), dim3(kBlock), 0, st, c->S, Lk, rays + k, perm);
            else if (live_cap < kPrefetchRays) hipLaunchKernelGGL(k_closest_f<true>, dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k, perm);
            else hipLaunchKernelGGL(k_closest_f<false>, dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k, perm);
            if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used + 1], st)); kev_used += 2; c->kev_kind.push_back((k > 0 && !count && !exact_only) ? 2 : 0); }
            // generation 0 runs alone and every workgroup allocates: 1024-thread workgroups quarter its same-address
            // atomics; later generations share the chip with the shadow passes, where a 1024-thread workgroup waits
            // for a whole CU to drain
            if (k == 0) hipLaunchKernelGGL(k_setup<kSetupBlock>, dim3((unsigned)((live_cap + kSetupBlock - 1) / kSetupBlock)), dim3(kSetupBlock), 0, st, c->S, Lk, Ln, rays + k, nodes + k,
                               spawn ? rays + k + 1 : rays + k, spawn);
            else hipLaunchKernelGGL(k_setup<kBlock>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, Ln, rays + k, nodes + k,
                               spawn ? rays + k + 1 : rays + k, spawn);
            if (c->S.nl > 0 && (count || exact_only)) {
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used], st)); }
                if (count) hipLaunchKernelGGL(k_shadow<true>, dim3(g, c->S.nl), dim3(kBlock), 0, st, c->S, Lk, nodes + k, c->d_tests + 4);
                else hipLaunchKernelGGL(k_shadow<false>, dim3(g, c->S.nl), dim3(kBlock), 0, st, c->S, Lk, nodes + k, c->d_tests + 4);
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used + 1], st)); kev_used += 2; c->kev_kind.push_back(1); }
            } else if (c->S.nl > 0) {
                // materialise and sort the shadow rays, then one any-hit pass -- on a side stream
                rtx_context::ShadowLane& sl = c->side[k % rtx_context::kSide];
                hipStream_t ss = sl.stream;
                HIPCHK(hipEventRecord(c->ev_fork[k], st));
                HIPCHK(hipStreamWaitEvent(ss, c->ev_fork[k], 0));
                sl.used = true;
                int* sc = c->d_scnt + ((size_t)nchunks * levels + k);
                HIPCHK(hipMemsetAsync(sl.hist, 0, sizeof(int) * (size_t)c->nbins, ss));
                hipLaunchKernelGGL(k_shadow_bin, dim3(g, c->S.nl), dim3(kBlock), 0, ss, c->S, Lk, nodes + k, c->binG, sl.hist, sl.rank);
                {
                    const int nb = (c->nbins + 1023) / 1024;
                    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, ss, sl.hist, sl.offs, c->nbins, sl.bsum, sl.braw, 64);
                    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ss, sl.bsum, nb, sc + 0);
                    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, ss, sl.offs, c->nbins, sl.bsum, sl.braw, 64, sl.q.id);
                }
                hipLaunchKernelGGL(k_shadow_scatter, dim3(g, c->S.nl), dim3(kBlock), 0, ss, c->S, Lk, nodes + k, c->binG, sl.offs, sl.rank, sl.q);
                unsigned gq = grid_for(live_cap * (size_t)c->S.nl + c->qpad);  // + the alignment padding of the bin groups
                gq = std::max(gq, (unsigned)(grid_for(std::min<size_t>(live_cap * (size_t)c->S.nl + c->qpad, (size_t)kSplitRays)) * kSplit));
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used], ss)); }
                hipLaunchKernelGGL(k_shadow_pass, dim3(gq), dim3(kBlock), 0, ss, c->S, sl.q, sc + 0, Lk.lit, Lk.cap);
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used + 1], ss)); kev_used += 2; c->kev_kind.push_back(1); }
            }
            if (c->fanout >= 2) live_cap *= 2;
        }
        for (rtx_context::ShadowLane& sl : c->side)
            if (sl.used) {
                HIPCHK(hipEventRecord(sl.join, sl.stream));
                HIPCHK(hipStreamWaitEvent(st, sl.join, 0));
                sl.used = false;
            }
        for (int k = levels - 1; k >= 0; k--) {
            LevelDev Lk = (k == 0) ? L0 : c->L[k];
            LevelDev Ln = (k + 1 < levels) ? c->L[k + 1] : Lk;
            hipLaunchKernelGGL(k_unwind, dim3(grid_for(bound[k])), dim3(kBlock), 0, st, c->S, Lk, Ln, nodes + k);
        }
        hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(64), 0, st, c->d_cnt, c->d_totals, levels);
    }
    if (nown > 0 && !(c->params.flags & RTX_F_NO_HALO))
        hipLaunchKernelGGL(k_aa_quantise, dim3(grid_for((size_t)nown * W)), dim3(kBlock), 0, st, c->d_fimg, c->d_row2ti,
                           c->d_own_rows, nown, W, c->height, d_out ? d_out : c->d_rgb8);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev1, st));
    c->last_stream = st;
    c->rendered = true;
    c->nchunks = nchunks;
    return RTX_OK;
}

extern "C" {

int rtx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

const char* rtx_last_hip_error(void) { return g_last_hip_error.c_str(); }

int rtx_context_create(const rtx_scene* scene, const rtx_params* params, rtx_context** out)
{
    if (!out) return RTX_E_INVALID_ARG;
    *out = nullptr;
    int rc = validate(scene, params);
    if (rc) return rc;
    rtx_context* c = new (std::nothrow) rtx_context();
    if (!c) return RTX_E_OOM;
    rc = context_create_impl(scene, params, c);
    if (rc) { rtx_context_destroy(c); return rc; }
    *out = c;
    return RTX_OK;
}

void rtx_context_destroy(rtx_context* c)
{
    if (!c) return;
    if (!c->allocs.empty() || c->ev0 || c->ev1 || !c->kev.empty()) {
        (void)hipSetDevice(c->device);
        if (c->rendered) (void)hipStreamSynchronize(c->last_stream);
        for (void* p : c->allocs) (void)hipFree(p);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        for (auto e : c->kev) if (e) (void)hipEventDestroy(e);
        for (auto e : c->ev_fork) if (e) (void)hipEventDestroy(e);
        for (rtx_context::ShadowLane& sl : c->side) {
            if (sl.join) (void)hipEventDestroy(sl.join);
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
        }
    }
    delete c;
}

int rtx_context_rows(const rtx_context* c, int32_t* rows_out, int32_t capacity)
{
    if (!c) return RTX_E_INVALID_ARG;
    int n = (int)c->own_rows.size();
    if (rows_out) {
        if (capacity < n) return RTX_E_INVALID_ARG;
        std::memcpy(rows_out, c->own_rows.data(), sizeof(int32_t) * n);
    }
    return n;
}

int rtx_context_render(rtx_context* c, void* hip_stream, uint8_t* d_rgb8_out)
{
    if (!c) return RTX_E_INVALID_ARG;
    return context_render_impl(c, (hipStream_t)hip_stream, d_rgb8_out);
}

int rtx_context_read_rgb8(rtx_context* c, uint8_t* host)
{
    if (!c || !host || !c->rendered) return RTX_E_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    size_t n = c->own_rows.size() * (size_t)c->width * 3;
    if (n) HIPCHK(hipMemcpy(host, c->d_rgb8, n, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_context_read_f32(rtx_context* c, float* host)
{
    if (!c || !host || !c->rendered) return RTX_E_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    size_t rowf = (size_t)c->width * 3;
    for (size_t r = 0; r < c->own_rows.size(); r++) {
        int ti = c->row2ti[c->own_rows[r]];
        HIPCHK(hipMemcpy(host + r * rowf, c->d_fimg + (size_t)ti * rowf, rowf * sizeof(float), hipMemcpyDeviceToHost));
    }
    return RTX_OK;
}

int rtx_context_stats(rtx_context* c, rtx_stats* s)
{
    if (!c || !s || !c->rendered) return RTX_E_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    std::memset(s, 0, sizeof *s);
    std::vector<unsigned long long> t(2 * c->levels);
    HIPCHK(hipMemcpy(t.data(), c->d_totals, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint64_t nprims = (uint64_t)c->S.ns + c->S.nt;
    for (int k = 0; k < c->levels; k++) {
        s->rays_level[k] = t[k];
        if (k == 0) s->rays_primary = t[k]; else s->rays_secondary += t[k];
        s->hits += t[c->levels + k];
    }
    s->rays_shadow = s->hits * (uint64_t)c->S.nl;
    s->tests_closest = (s->rays_primary + s->rays_secondary) * nprims;
    s->tests_shadow_max = s->rays_shadow * nprims;
    s->singular_matrices = c->singular;
    s->chunks = c->nchunks;
    s->levels = c->levels;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    s->ms_total = ms;
    unsigned long long tk[8];
    HIPCHK(hipMemcpy(tk, c->d_pstat, sizeof tk, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; k++) s->packet_stats[k] = tk[k];
    HIPCHK(hipMemcpy(tk, c->d_tests, sizeof tk, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) { s->tests_closest_kind[k] = tk[k]; s->tests_shadow_kind[k] = tk[4 + k]; }
    for (size_t e = 0; e < c->kev_kind.size(); e++) {
        float m = 0;
        HIPCHK(hipEventElapsedTime(&m, c->kev[2 * e], c->kev[2 * e + 1]));
        if (c->kev_kind[e] == 0) { s->ms_closest += m; s->launches_closest++; }
        else if (c->kev_kind[e] == 2) { s->ms_closest_lanes += m; s->launches_closest_lanes++; }
        else { s->ms_shadow += m; s->launches_shadow++; }
    }
    return RTX_OK;
}

static int render_one(const rtx_scene* scene, const rtx_params* params, uint8_t* rgb, float* f32, rtx_stats* stats)
{
    rtx_context* c = nullptr;
    int rc = rtx_context_create(scene, params, &c);
    if (rc) return rc;
    rc = rtx_context_render(c, nullptr, nullptr);
    if (!rc && rgb) rc = rtx_context_read_rgb8(c, rgb);
    if (!rc && f32) rc = rtx_context_read_f32(c, f32);
    if (!rc && stats) rc = rtx_context_stats(c, stats);
    rtx_context_destroy(c);
    return rc;
}

int rtx_render_f32(const rtx_scene* scene, const rtx_params* params, float* out, rtx_stats* stats)
{
    if (!out) return RTX_E_INVALID_ARG;
    int rc = validate(scene, params);
    if (rc) return rc;
    return render_one(scene, params, nullptr, out, stats);
}

int rtx_render_rgb8(const rtx_scene* scene, const rtx_params* params, int ngpus, uint8_t* out, rtx_stats* stats)
{
    if (!out) return RTX_E_INVALID_ARG;
    int rc = validate(scene, params);
    if (rc) return rc;
    if (params->nranks > 1 || ngpus <= 1) return render_one(scene, params, out, nullptr, stats);
    // In-process multi-GPU: one host thread per device, interleaved bands, each device copies
    // its own rows to the host over its own PCIe link; rows are placed in the caller's frame.
    int ndev = rtx_device_count();
    if (ndev <= 0) return RTX_E_NO_DEVICE;
    if (ngpus > ndev) ngpus = ndev;
    std::vector<int> rcs(ngpus, 0);
    std::vector<rtx_stats> sts(ngpus);
    std::vector<std::thread> th;
    const size_t rowb = (size_t)params->width * 3;
    for (int g = 0; g < ngpus; g++)
        th.emplace_back([&, g]() {
            rtx_params p = *params;
            p.nranks = ngpus; p.rank = g; p.device = g;
            if (p.band_rows <= 0) p.band_rows = 16;
            int n = rtx_band_rows(p.height, p.nranks, p.rank, p.band_rows, nullptr, 0);
            if (n <= 0) { rcs[g] = n < 0 ? n : 0; std::memset(&sts[g], 0, sizeof(rtx_stats)); return; }
            std::vector<int32_t> rows(n);
            rtx_band_rows(p.height, p.nranks, p.rank, p.band_rows, rows.data(), n);
            std::vector<uint8_t> tmp((size_t)n * rowb);
            rcs[g] = render_one(scene, &p, tmp.data(), nullptr, &sts[g]);
            if (rcs[g]) return;
            for (int r = 0; r < n; r++) std::memcpy(out + (size_t)rows[r] * rowb, tmp.data() + (size_t)r * rowb, rowb);
        });
    for (auto& t : th) t.join();
    for (int g = 0; g < ngpus; g++)
        if (rcs[g]) return rcs[g];
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (int g = 0; g < ngpus; g++) {
            const rtx_stats& s = sts[g];
            stats->rays_primary += s.rays_primary; stats->rays_secondary += s.rays_secondary;
            stats->rays_shadow += s.rays_shadow; stats->hits += s.hits;
            for (int k = 0; k < RTX_MAX_LEVELS; k++) stats->rays_level[k] += s.rays_level[k];
            stats->tests_closest += s.tests_closest; stats->tests_shadow_max += s.tests_shadow_max;
            stats->singular_matrices = s.singular_matrices;
            stats->chunks += s.chunks; stats->levels = std::max(stats->levels, s.levels);
            stats->ms_total = std::max(stats->ms_total, s.ms_total);
        }
    }
    return RTX_OK;
}

}  // extern "C"
