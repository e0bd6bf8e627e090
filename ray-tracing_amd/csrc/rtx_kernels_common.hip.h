// rtx_kernels_common.hip.h -- device-side views of the scene and ray queues, queue-slot allocation, hot-record helpers, ray generation, and the exact-only closest-hit kernel (the reference's loops as written)
// Part of the one translation unit rtx_device.hip (device code, namespace rtx); not a stand-alone header.
#ifndef RTX_KERNELS_COMMON_HIP_H_
#define RTX_KERNELS_COMMON_HIP_H_

namespace rtx {

constexpr int kBlock = 256;
constexpr float kBias = 1e-4f;       // raytracing.cpp:447
constexpr float kMaxT = 1e8f;        // raytracing.cpp:10 (float nearest = Max_t)
constexpr int kTFlag = 1 << 30;      // the reference's `transformed` flag, carried in the prim word

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct SceneDev {
    int ns, nt, nl;
    int mt;                       // RTX_F_TRI_MOLLER_TRUMBORE: tri_hot / phot hold the Moller-Trumbore records (rtx_math.hip.h)
    int np;                       // polygons: primitive index ns + nt + i, traced after the triangles (:421-434); with
                                  // RTX_F_TRACE_PLANES the planes follow them as records with vertex count -1
    const rtx_polygon* poly;      // [np] raw records (materials)
    const float4* mat4;           // [(ns + nt + np) * 4] packed materials (rtx_pack.h: mat4): what set-up and shading read per hit
    const float4* poly_verts;     // vertices, x y z 0; a polygon's run starts at its hot record's first_vertex
    const float4* sph_hot;        // [ns]  cx cy cz r^2
    const int* sph_xf;            // [ns]
    const rtx_xmatrix* sph_im;    // [ns]  precomputed inverseMatrix(mat)
    const rtx_sphere* sph;        // [ns]
    const float4* tri_hot;        // [nt*5]
    const float4* tri_edge;       // [nt*2] edge filter { a1, m }, { a2, Q } (rtx_bounds.cpp, tri_edge_filter)
    const float4* phot;           // [(ns+nt)*5] one 80-byte hot record per primitive (rtx_pack.h, prim_hot)
    const rtx_triangle* tri;      // [nt]
    const rtx_light* lights;      // [nl]
    F3 ambient;
    const float4* bounds;         // [ns+nt] filter records { c, -R^2 } (rtx_bounds.cpp)
    const float4* aff_shadow;     // [naff*4] AffineShadowRec: affine spheres for shadow rays
    const unsigned long long* aff_cells;  // [nl*G*G][aff_words] per (light, direction cell): which affine spheres a shadow ray of that cell can pass (nullptr: no table)
    const float* aff_dt;          // [nl] the table covers origins within this distance of the light
    int aff_words, aff_G;
    int naff;
    F3 box_lo, box_inv;           // scene box for ray binning: cell = (o - box_lo) * box_inv, 16 cells per axis
    unsigned long long* pstat;    // [8] packet statistics: closest {waves, waves ok, batches, survivors}, shadow {same}
    unsigned long long* exec;     // [3 * EX_KINDS] executed-work counters of k_closest_f / k_closest_lanes / k_shadow_pass (RTX_F_COUNT_EXEC)
    const float4* sbounds;        // [nsb] filter records of the shadow casters except affine spheres (transposed scan of
    const int* sorder;            // [nsb]  unpacketable rays), and the primitive index of each
    int nsb;
    float guard;                  // |origin component| beyond this: the ray bypasses the filter
    // two-level packet test (rtx_pack.h, ClusterSet): cluster spheres, their 64 members, member -> primitive
    const float4* ccl; const float4* cmem; const int* cidx; const int* cseg; int cnseg;  // closest-hit, per segment
    const int* corder; int cnfirst;  // per-ray traversal order of the closest-hit clusters: affine-sphere clusters first
    const float4* cclR; const float4* csubR;  // { c, R } cluster spheres [ncl] and 8 sub-cluster spheres each [ncl*8]
    const float4* csupR;          // { c, R } around every run of 8 clusters in traversal order (ClusterSet::supR)
    const float4* cclRo;          // cclR in traversal order (cclRo[p] = cclR[corder[p]]): the per-ray traversal reads sphere and number side by side
    const float4* scl; const float4* smem; const int* sidx; int sncl;                    // shadow rays
    int sel2;                     // shadow pass: two-level member selection (frames that fill the chip; rtx_device.hip)
    const float4* ssubR;          // [sncl*8] { c, R } spheres around members 8g .. 8g+7 of every shadow cluster
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
    int* src;                            // per RAY (generations >= 1): the plain sphere its parent hit left, -1 = none (k_closest_lanes: far limit)
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
    if (cnt == -1) {
        // RTX_F_TRACE_PLANES: intersectplane, raytracing.cpp:275-287 with normalize(p.normal) (:277) done on the host;
        // q0 = { unit normal, d as given }
        const float vd = q0.x * dir.x + q0.y * dir.y + q0.z * dir.z;
        if (vd == 0.0f) return false;
        t = -((q0.x * o.x + q0.y * o.y + q0.z * o.z) + q0.w) / vd;
        if (t < 0.0f) return false;
        return true;
    }
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
    base = __builtin_amdgcn_readlane(base, leader);
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

// Executed-work counters (RTX_F_COUNT_EXEC): what the PRODUCTION kernels actually evaluate, lane by lane -- one
// count per (ray or packet, sphere / primitive) pair a lane works on whose result is used.  Per-lane tallies in
// registers, one wave reduction and one atomic per kind at the end of the kernel; the counting instantiations
// are separate kernels (template flag), the timed ones carry a compile-time null pointer and none of this.
enum {
    EX_PACKET = 0,   // packet_test: one bounding sphere against a wave's packet (28 FLOP, bench.py EXEC_FLOPS)
    EX_CONE,         // cone_slack / cone_slack_front: one enclosing sphere against one ray's cone (25)
    EX_FILT,         // filt / filt_front: one primitive's bounding sphere against one ray's line (16)
    EX_AFILT,        // filt_affine: affine sphere, shadow-ray form (39)
    EX_STRIP,        // affine_strip_pass: one affine sphere against a wave's strip packet (150)
    EX_SPH,          // exact sphere test, raytracing.cpp:202-228 (20)
    EX_SPHX,         // exact affine-sphere test incl. the inverse transform(s) (65; shadow form 47)
    EX_TRI_EARLY,    // exact triangle test leaving at vd == 0 or t < 0 (13)
    EX_TRI_CULL,     // ... leaving at the nearest-t cull, t >= nearest (13)
    EX_TRI_FULL,     // ... through the three areas (76; the edge filter's 16 in front of them are not credited)
    EX_POLY,         // exact polygon test (counted as a full triangle test per vertex)
    EX_WAVES,        // packets built (make_packet)
    EX_RAYS,         // rays the kernel traced
    EX_TRI_EDGE,     // exact triangle test leaving at the edge filter (13 + P, vec1 and the filter: 38)
    EX_KINDS
};
struct Exec { unsigned n[EX_KINDS]; };
__device__ __forceinline__ void ex_add(Exec* ex, int kind, unsigned v = 1u) { if (ex) ex->n[kind] += v; }
__device__ __forceinline__ void ex_zero(Exec* ex) { if (ex) for (int k = 0; k < EX_KINDS; k++) ex->n[k] = 0u; }
__device__ __forceinline__ void ex_tri(Exec* ex, int exitp) { ex_add(ex, exitp == 0 ? EX_TRI_EARLY : exitp == 1 ? EX_TRI_CULL : exitp == 3 ? EX_TRI_EDGE : EX_TRI_FULL); }
__device__ __forceinline__ void ex_flush(Exec* ex, unsigned long long* dst)
{
    if (!ex) return;
    for (int k = 0; k < EX_KINDS; k++) {
        unsigned int v = ex->n[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst + k, (unsigned long long)v);
    }
}

// ------------------------------------------------------------------------------------------------
// K1 ray generation (raytracing.cpp:753-759); origin is one constant (:735-737) but is stored
// per ray so that every level has the same layout.
// ------------------------------------------------------------------------------------------------
#ifndef RTX_LANES_TU  // (a plain kernel has one definition: rtx_device.hip's; rtx_lanes.hip holds k_closest_lanes only)
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
#endif

// ------------------------------------------------------------------------------------------------
// K2 closest hit (raytracing.cpp:382-419).  One ray per lane; the primitive index is
// wave-uniform, so primitive records arrive through scalar loads and sit in SGPRs.
// Spheres first, in file order, strict `<` (lowest index wins ties); then triangles.
// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void k_closest(SceneDev S, LevelDev L, const int* nrays, unsigned long long* tests)
{
    int i = blockIdx.x * kBlock + threadIdx.x;
    bool live = i < min(*nrays, L.cap);
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
        bool h = triangle_test(o, d, T, t, vd, nullptr, S.mt != 0);
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



}  // namespace rtx
#endif
