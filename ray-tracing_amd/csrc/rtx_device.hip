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
//
// One translation unit: the kernels live in rtx_kernels_common.hip.h (views, helpers, raygen, the
// exact-only closest-hit kernel), rtx_closest.hip.h, rtx_shadow.hip.h and rtx_shade.hip.h (set-up,
// unwind, antialias); this file holds the resident context and the C-ABI entry points.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is loaded with dlopen on first multi-GPU use
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rtx.h"
#include "rtx_math.hip.h"
#include "rtx_pack.h"

namespace rtx {

static thread_local std::string g_last_hip_error;

// A/B aids (tools/dbg): environment variables that move a threshold or force a kernel form -- compiled in only with
// -DRTX_AB_KNOBS (make EXTRA=-DRTX_AB_KNOBS).  The shipped library reads no environment at all.
#ifdef RTX_AB_KNOBS
static inline const char* ab_knob(const char* name) { return std::getenv(name); }
#else
static inline const char* ab_knob(const char*) { return nullptr; }
#endif

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_last_hip_error = std::string(#expr) + ": " + hipGetErrorString(e_);             \
            return (e_ == hipErrorOutOfMemory) ? RTX_E_OOM : RTX_E_HIP;                       \
        }                                                                                     \
    } while (0)

}  // namespace rtx

#include "rtx_kernels_common.hip.h"
#include "rtx_closest.hip.h"
namespace rtx {  // compiled in rtx_lanes.hip (the Makefile builds the two files side by side)
extern template __global__ void k_closest_lanes<false>(SceneDev, LevelDev, const int*, int, int);
extern template __global__ void k_closest_lanes<true>(SceneDev, LevelDev, const int*, int, int);
}
#include "rtx_shadow.hip.h"
#include "rtx_cells.hip.h"
#include "rtx_shade.hip.h"


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
    int* d_cnt = nullptr;                   // [2*levels + 2]: rays of level k at [2k + 1], nodes of level k at [2k + 2] -- nodes(k) and rays(k + 1)
                                            // are one aligned 64-bit word: k_setup takes both with ONE atomic per workgroup
    unsigned long long* d_totals = nullptr; // [2*levels]
    unsigned long long* d_tests = nullptr;  // [8] closest kinds, shadow kinds (RTX_F_COUNT_TESTS)
    int binG = 256, nbins = 0;              // light-space bins: nl * G * G
    unsigned long long* d_pstat = nullptr;
    // The shadow rays of generation k and the closest-hit pass of generation k+1 depend on nothing but
    // k_setup(k): generation k's shadow chain (bin, scan, scatter, pass) runs on side stream side_of(k) with
    // that stream's own queue and scan buffers (fork after every set-up, one join per generation before
    // its own unwind), so the thin, latency-bound kernels of the deep generations overlap instead of queueing.
    static constexpr int kSide = 3;
    // generation 0's pass is the long one (it ends last at every frame size measured): it keeps a stream to
    // itself, the deeper generations alternate on the other two (generation 3 queued behind generation 0's
    // pass cost a rank of 8 a tenth of its frame)
    static int side_of(int k) { return k == 0 ? 0 : 1 + (k - 1) % (kSide - 1); }
    struct ShadowLane {
        hipStream_t stream = nullptr;
        ShadowQ q = {};
        int* hist = nullptr; int* offs = nullptr; int* rank = nullptr; int* bsum = nullptr; int* braw = nullptr;
    } side[kSide + 1];
    // The LAST generation's chain runs on the main stream itself, with the extra lane's buffers: after that
    // generation's set-up the main stream has nothing left but the unwinds, which wait for the passes anyway,
    // and the chain need not queue behind an earlier generation's pass (measured: 5.88 -> 5.81 ms per 4K frame,
    // 1.36 -> 1.34 for a rank of 8; the last TWO generations there: slower, they delay the closest-hit chain).
    int main_from = 1 << 30;
    int lanes16 = kSplitClosest, lanes4 = kSplitClosest4;  // k_closest_lanes: generations up to this many rays give a wave 16 / 4 rays instead of 64
    int split4 = kSplitRays, split8 = kSplitRays8;  // shadow generations up to this many rays: 4 / 8 workgroups per 256 rays
    std::vector<hipEvent_t> ev_fork, ev_pass;  // per generation: set-up done (main stream), shadow pass done (side stream)
    // k_closest_lanes' own cluster set (PackedScene::pc) for scenes of more than one segment; has_pc = false: S's own
    struct LanesSet {
        bool has_pc = false;
        const float4 *cmem = nullptr, *cclR = nullptr, *csubR = nullptr, *csupR = nullptr, *cclRo = nullptr;
        const int *cidx = nullptr, *cseg = nullptr, *corder = nullptr;
        int cnfirst = 0;
    } lanes;
    SceneDev lanes_scene() const
    {
        SceneDev SL = S;
        if (lanes.has_pc) {
            SL.cmem = lanes.cmem; SL.cclR = lanes.cclR; SL.csubR = lanes.csubR; SL.csupR = lanes.csupR; SL.cclRo = lanes.cclRo;
            SL.cidx = lanes.cidx; SL.cseg = lanes.cseg; SL.corder = lanes.corder;
            SL.cnseg = 1; SL.cnfirst = lanes.cnfirst;
        }
        return SL;
    }
    size_t qpad = 0;                        // slots the 64-alignment of the shadow bin groups can add to a queue
    int* d_scnt = nullptr;                  // [nchunks*levels] slots (rays + alignment padding) of each generation's shadow queue
    int nframe_chunks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kev;            // RTX_F_TIME_KERNELS: (start, stop) per hot launch
    std::vector<int> kev_kind;              // 0 closest (packet / exact), 1 shadow, 2 closest per-ray; per pair used by the last render
    hipStream_t last_stream = nullptr;
    CamTab cam{};                           // camera direction cell -> primitives (k_primary_cells, rtx_cells.hip.h); off == nullptr: generation 0
                                            // takes k_closest_f
    CellTab cells{};                        // (light, direction cell) -> shadow casters (rtx_cells.hip.h); off == nullptr: no table,
                                            // the packet kernels (k_shadow_pass) trace the plain primitives
    bool shadow_direct = false;             // the table kernels read the hit records themselves: no shadow queue, no sort (k_shadow_direct)
    bool use_cells = false;                 // this context's shadow stage goes through the tables (false: the packet kernels, k_shadow_pass)
    bool form_early = false;                // the tables were chosen before the queues were allocated (short lists or RTX_F_SHADOW_TABLES):
                                            // the packet form's bit masks and shadow queues are never built
    std::vector<void*> table_allocs, packet_allocs;  // what only one of the two forms needs (the other's is freed once the form is chosen)
    int pool_from = 1;                      // first generation whose shadow pass takes the pooled stages (shadow_plain_pool): generation 0's
                                            // coherent rays keep the one-survivor-at-a-time walk; RTX_SHADOW_POOL_FROM=k: A/B aid
    uint8_t* h_stage = nullptr;             // pinned host staging of rtx_context_read_rgb8 (allocated on its first call; not in arena mode)
    unsigned char* last_dst = nullptr;      // where the last render put its rgb8 rows (d_rgb8 or the caller's buffer)
    bool rendered = false;
    int nchunks = 0;
    // Arena mode (rtx_render_rgb8_noalloc): every device buffer is carved out of a caller-provided block
    // (the library's static scratch array) instead of hipMalloc; nothing is created or freed on the device.
    char* arena = nullptr;
    size_t arena_cap = 0, arena_off = 0;
    unsigned long long* d_cursor = nullptr; // the table builds' cursor (build_cell_table)
    int* d_overflow = nullptr;              // rays k_setup could not place (a level's capacity exceeded): 0 by construction
    bool calibrated = false;                // level capacities come from the calibration pass (fan-out 2 scenes)
    bool serial = false;                    // RTX_F_SERIAL / arena mode: everything on the caller's stream, no side streams, no fork/join events

    // Slab: between slab_begin and slab_end every dalloc that fits is carved out of ONE hipMalloc (256-byte aligned).  The ray
    // records of the generations are ~100 arrays: as separate blocks they cost 3 ms to allocate and 18 ms to free
    // (tools/ubench/malloc_cost.hip: 4 GiB in 96 blocks), as one block 1.5 + 0.6 ms -- the calibration pass allocates and frees
    // its worst-case set inside context creation.  A slab is freed as a whole (it is one entry of `allocs`).
    char* slab = nullptr;
    size_t slab_cap = 0, slab_off = 0;
    int slab_begin(size_t bytes)
    {
        if (arena) return RTX_OK;
        void* q = nullptr;
        HIPCHK(hipMalloc(&q, std::max<size_t>(bytes, 256)));
        allocs.push_back(q);
        slab = (char*)q; slab_cap = std::max<size_t>(bytes, 256); slab_off = 0;
        return RTX_OK;
    }
    void slab_end() { slab = nullptr; slab_cap = slab_off = 0; }
    static size_t round256(size_t b) { return (std::max<size_t>(b, 1) + 255) & ~(size_t)255; }

    template <typename T> int dalloc(T** p, size_t n)
    {
        void* q = nullptr;
        if (n == 0) n = 1;
        if (slab) {
            const size_t bytes = round256(n * sizeof(T));
            if (bytes <= slab_cap - slab_off) {
                *p = (T*)(slab + slab_off);
                slab_off += bytes;
                return RTX_OK;
            }
        }
        if (arena) {
            const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
            if (bytes > arena_cap - arena_off) return RTX_E_TOO_LARGE;
            q = arena + arena_off;
            arena_off += bytes;
            *p = (T*)q;
            return RTX_OK;
        }
        HIPCHK(hipMalloc(&q, n * sizeof(T)));
        allocs.push_back(q);
        *p = (T*)q;
        return RTX_OK;
    }
    void dfree_group(std::vector<void*>& grp)
    {
        if (arena) { grp.clear(); return; }
        for (void* q : grp) {
            auto it = std::find(allocs.begin(), allocs.end(), q);
            if (it != allocs.end()) { (void)hipFree(q); allocs.erase(it); }
        }
        grp.clear();
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
    if (int rc = validate_scene(sc)) return rc;
    if ((p->flags & RTX_F_TRACE_PLANES) && sc->nplanes && !sc->planes) return RTX_E_INVALID_ARG;
    const int64_t npl = (p->flags & RTX_F_TRACE_PLANES) ? sc->nplanes : 0;
    if ((int64_t)sc->nspheres + sc->ntriangles + sc->npolygons + npl >= (int64_t)kTFlag) return RTX_E_TOO_LARGE;
    if ((int64_t)sc->nspheres + sc->ntriangles + sc->npolygons + npl >= ((int64_t)1 << kPoolPrimBits)) return RTX_E_TOO_LARGE;  // closest_pool's list entries
    if (sc->nlights > 65535) return RTX_E_TOO_LARGE;  // gridDim.y
    int nr = p->nranks <= 1 ? 1 : p->nranks;
    int rk = p->nranks <= 1 ? 0 : p->rank;
    if (rk < 0 || rk >= nr) return RTX_E_INVALID_ARG;
    return RTX_OK;
}

// the counters of generation k (rtx_context::d_cnt): `rays + k`, `nodes + k` as the launch sites write them
struct CntPtr { int* p; int* operator+(int k) const { return p + 2 * k; } };
static int calibrate_counts(rtx_context* c, int* d_blk);
static int context_render_impl(rtx_context* c, hipStream_t st, uint8_t* d_out);

// One direction-cell table (rtx_cells.hip.h), built in ONE pass: a build workgroup takes its cells' range of the table with an
// atomic add on a cursor, so the table is allocated for a GUESS of its size and built again at the size the cursor reports if the
// guess was short; a table beyond the budget takes a coarser map (a quarter of the cells), as does one the device has no room
// for.  kind 0: plain shadow casters per (light, cell), entries { primitive, dneed }; 1: affine spheres per (light, cell, slab),
// entries = record indices; 2: every primitive per camera cell.  off == nullptr on return: no map is coarse enough, no table.
struct TableSpec {
    int kind, napex, G0, Gmin;
    size_t max_lists;
    uint64_t guess, max_entries;
    const float* d_range;
    int tight;
    CamCells cam;
};
struct TableOut { const int2* off = nullptr; const void* ent = nullptr; int G = 0, attempt = 0; uint64_t entries = 0, lists = 0, room = 0; };

static int build_cell_table(rtx_context* c, const TableSpec& T, TableOut* out)
{
    *out = TableOut{};
    int rc = RTX_OK;
    if (!c->d_cursor && (rc = c->dalloc(&c->d_cursor, 1))) return rc;
    const int mult = T.kind == 1 ? kAffSlabs : 1;
    const size_t esz = T.kind == 1 ? sizeof(int) : sizeof(int2);
    for (int G = T.G0; G >= T.Gmin; G /= 2) {
        if ((size_t)T.napex * G * G * mult > T.max_lists) continue;
        const int nwave = T.napex * G * G;       // one build wave per (apex, cell)
        const int ncell = nwave * mult;          // lists
        uint64_t want = std::min<uint64_t>(std::max<uint64_t>(T.guess, (uint64_t)1 << 16), T.max_entries);
        int2* off = nullptr;
        if ((rc = c->dalloc(&off, (size_t)ncell))) return rc;
        bool coarser = false;
        for (int attempt = 0; attempt < 2; attempt++) {
            char* ent = nullptr;
            rc = c->dalloc(&ent, ((size_t)want + kCellPad) * esz);
            if (rc == RTX_E_OOM) { (void)hipGetLastError(); coarser = true; break; }  // no room for this map: a coarser one
            if (rc) return rc;
            HIPCHK(hipMemsetAsync(c->d_cursor, 0, sizeof(unsigned long long), nullptr));
            if (T.kind == 1) hipLaunchKernelGGL(k_aff_build, dim3((unsigned)nwave), dim3(64), 0, 0, c->S, G, T.d_range, off, reinterpret_cast<int*>(ent), c->d_cursor, (unsigned long long)want);
            else hipLaunchKernelGGL(k_cells_build, dim3((unsigned)(nwave / kCellsPerGroup)), dim3(64 * kCellsPerGroup), 0, 0, c->S, G, T.d_range, off, reinterpret_cast<int2*>(ent),
                                    c->d_cursor, (unsigned long long)want, T.tight, T.cam);
            HIPCHK(hipGetLastError());
            unsigned long long sum = 0;
            HIPCHK(hipMemcpy(&sum, c->d_cursor, sizeof sum, hipMemcpyDeviceToHost));  // (waits for the build)
            if (sum <= want) {
                // the walks read a few entries past a list's end: { primitive -1, dneed NaN } (never taken), affine record 0 (masked)
                HIPCHK(hipMemset(ent + (size_t)sum * esz, T.kind == 1 ? 0 : 0xff, esz * kCellPad));
                // a generous guess must not stay in HBM for the life of the context: more than 256 MiB of unused room moves the
                // entries into an allocation of their own size (C5's plain table: room for 512 M entries, 136 M used: 1 ms of copy)
                if (want > 2 * sum && (want - sum) * esz > ((uint64_t)256 << 20) && !c->allocs.empty() && c->allocs.back() == (void*)ent) {
                    char* fit = nullptr;
                    const int rf = c->dalloc(&fit, ((size_t)sum + kCellPad) * esz);
                    if (rf == RTX_OK) {
                        HIPCHK(hipMemcpy(fit, ent, ((size_t)sum + kCellPad) * esz, hipMemcpyDeviceToDevice));
                        const auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)ent);
                        if (it != c->allocs.end()) { (void)hipFree(ent); c->allocs.erase(it); }
                        ent = fit;
                    } else if (rf == RTX_E_OOM) (void)hipGetLastError();  // no room for the copy: keep the large one
                    else return rf;
                }
                out->off = off; out->ent = ent; out->G = G; out->attempt = attempt; out->entries = sum; out->lists = (uint64_t)ncell; out->room = want;
                return RTX_OK;
            }
            // short: give the entries back (the last allocation) and build again at the size it needs, or coarser
            (void)hipFree(c->allocs.back());
            c->allocs.pop_back();
            if (sum > T.max_entries || attempt == 1) { coarser = true; break; }
            want = sum;
        }
        if (coarser) {
            (void)hipFree(c->allocs.back());  // the offsets of this map
            c->allocs.pop_back();
        }
    }
    return RTX_OK;
}

static int context_create_impl(const rtx_scene* scene, const rtx_params* params, rtx_context* c)
{
    const bool verbose = ab_knob("RTX_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!verbose) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "rtx: create %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RTX_E_NO_DEVICE;
    int dev = params->device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev) return RTX_E_INVALID_ARG;
    HIPCHK(hipSetDevice(dev));
    c->device = dev;
    c->params = *params;
    c->serial = (params->flags & RTX_F_SERIAL) != 0 || c->arena != nullptr;
    if (const char* e = ab_knob("RTX_LANES16")) c->lanes16 = std::atoi(e);
    if (const char* e = ab_knob("RTX_LANES4")) c->lanes4 = std::atoi(e);
    if (const char* e4 = ab_knob("RTX_SPLIT4")) c->split4 = std::atoi(e4);  // A/B aids for the split thresholds (same image)
    if (const char* e8 = ab_knob("RTX_SPLIT8")) c->split8 = std::atoi(e8);
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

    phase("rows");
    PackedScene P;
    pack_scene(*scene, W, H, P, params->flags);
    phase("pack_scene (host)");
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
        float4* tedge = nullptr;
        static_assert(sizeof(TriEdge) == 2 * sizeof(float4), "triangle edge record is two float4");
        if ((rc = c->dalloc(&tedge, P.tri_edge.size() * 2))) return rc;
        if (!P.tri_edge.empty()) HIPCHK(hipMemcpy(tedge, P.tri_edge.data(), P.tri_edge.size() * sizeof(TriEdge), hipMemcpyHostToDevice));
        c->S.tri_edge = tedge;
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
            c->S.poly = pg; c->S.poly_verts = reinterpret_cast<const float4*>(pv); c->S.np = (int)P.poly.size();  // + planes, if traced
            c->S.mt = (params->flags & RTX_F_TRI_MOLLER_TRUMBORE) ? 1 : 0;
        }
        {
            float* m4 = nullptr;
            if ((rc = c->upload(&m4, P.mat4))) return rc;
            c->S.mat4 = reinterpret_cast<const float4*>(m4);
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
            if ((rc = up4(P.cc.supR, &c->S.csupR))) return rc;
            if ((rc = c->upload(&q, P.cc.order))) return rc;
            c->S.corder = q;
            c->S.cnfirst = P.cc.nfirst;
            auto ordered = [](const ClusterSet& cs) {
                std::vector<BoundSphere> v(cs.order.size());
                for (size_t p = 0; p < v.size(); p++) v[p] = cs.clR[cs.order[p]];
                return v;
            };
            if ((rc = up4(ordered(P.cc), &c->S.cclRo))) return rc;
            if (!P.pc.clR.empty()) {
                if ((rc = up4(P.pc.mem, &c->lanes.cmem))) return rc;
                if ((rc = up4(P.pc.clR, &c->lanes.cclR))) return rc;
                if ((rc = up4(P.pc.subR, &c->lanes.csubR))) return rc;
                if ((rc = up4(P.pc.supR, &c->lanes.csupR))) return rc;
                if ((rc = c->upload(&q, P.pc.idx))) return rc;
                c->lanes.cidx = q;
                if ((rc = c->upload(&q, P.pc.seg_first))) return rc;
                c->lanes.cseg = q;
                if ((rc = c->upload(&q, P.pc.order))) return rc;
                c->lanes.corder = q;
                if ((rc = up4(ordered(P.pc), &c->lanes.cclRo))) return rc;
                c->lanes.cnfirst = P.pc.nfirst;
                c->lanes.has_pc = true;
            }
            if ((rc = up4(P.sc.cl, &c->S.scl))) return rc;
            if ((rc = up4(P.sc.mem, &c->S.smem))) return rc;
            if ((rc = up4(P.sc.subR, &c->S.ssubR))) return rc;
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
        phase("scene upload");
        // The (light, direction cell) tables cover origins up to the far corner of the scene box (rays from further away,
        // possible only through the reference's odd hit points on affine spheres, are traced against every sphere).
        std::vector<float> dt(c->S.nl);
        for (int l = 0; l < c->S.nl; l++) {
            double far2 = 0;
            const float lp[3] = {P.lights[l].point.x, P.lights[l].point.y, P.lights[l].point.z};
            const float lo[3] = {c->S.box_lo.x, c->S.box_lo.y, c->S.box_lo.z};
            const float inv[3] = {c->S.box_inv.x, c->S.box_inv.y, c->S.box_inv.z};
            for (int a = 0; a < 3; a++) {
                const double hi = (double)lo[a] + 16.0 / (double)inv[a];
                const double d = std::max(std::fabs((double)lp[a] - lo[a]), std::fabs((double)lp[a] - hi));
                far2 += d * d;
            }
            const double v = std::sqrt(far2) * 1.01 + 1.0;
            dt[l] = std::isfinite(v) ? (float)v : 0.0f;
        }
        float* d_dt = nullptr;
        if ((rc = c->upload(&d_dt, dt))) return rc;
        // (light, direction cell) -> plain shadow casters, CSR lists sorted by the distance from the light a ray must
        // start beyond to pass the entry (rtx_cells.hip.h): count, scan, fill.  Not in arena mode (its size is unknown
        // before the count) and not for the exact-only / counting frames, which trace no culled shadow rays.
        c->cells = CellTab{};
        // (a scene whose shadow casters are ALL affine spheres still takes the tables: its plain lists are empty, and the packet form's
        // strip test over a bit-mask union of thousands of affine spheres is what it was never meant for -- 3,000 general affine
        // spheres at 4K: 239 ms a frame in the packet form, tools/dbg/perf_survey.py)
        bool want_cells = !c->arena && c->S.nl > 0 && (c->S.sncl > 0 || c->S.naff > 0) && !(params->flags & (RTX_F_EXACT_ONLY | RTX_F_COUNT_TESTS));
        if (const char* e = ab_knob("RTX_CELLS")) want_cells = want_cells && std::atoi(e) != 0;
        if (want_cells) {
            // The tables' range: the far corner of the box of the shadow casters themselves (an affine sphere as it sits for a
            // shadow ray: its centre and radius as given, raytracing.cpp:571-576), twice over -- not of every bounded
            // primitive: an affine sphere's world-space ellipsoid can be thousands of units long, rho = 3.8e-3 D then lets
            // every primitive pass every cell, and hit points that far out are a handful of pixels.  Rays from beyond it take
            // the transposed scan (shadow_table_trace).
            std::vector<float> dc(c->S.nl);
            {
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                auto grow = [&](double x, double y, double z, double r) {
                    if (!std::isfinite(x) || !std::isfinite(y) || !std::isfinite(z) || !std::isfinite(r)) return;
                    const double cc[3] = {x, y, z};
                    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], cc[a] - r); hi[a] = std::max(hi[a], cc[a] + r); }
                };
                for (const BoundSphere& b : P.sbounds) grow(b.cx, b.cy, b.cz, std::sqrt(-(double)b.neg_r2));
                for (const AffineShadowRec& a : P.aff_shadow) grow(a.ox, a.oy, a.oz, std::sqrt(-(double)a.neg_r2));
                for (int l = 0; l < c->S.nl; l++) {
                    const double lp[3] = {P.lights[l].point.x, P.lights[l].point.y, P.lights[l].point.z};
                    double far2 = 0;
                    for (int a = 0; a < 3; a++) {
                        const double d = (hi[a] >= lo[a]) ? std::max(std::fabs(lp[a] - lo[a]), std::fabs(lp[a] - hi[a])) : 0.0;
                        far2 += d * d;
                    }
                    const double v = 2.0 * std::sqrt(far2) + 1.0;
                    dc[l] = std::isfinite(v) ? std::min((float)v, dt[l]) : 0.0f;
                }
            }
            float* d_dc = nullptr;
            if ((rc = c->upload(&d_dc, dc))) return rc;
            int G0 = 512;
            if (const char* e = ab_knob("RTX_CELL_G")) G0 = std::max(8, std::min(4096, std::atoi(e)));
            // a finer map for dense scenes, as the sort grid of the packet form takes (C5, 100,000 triangles: 78.5 / 56.6 / 47.0 ms
            // per frame with 256 / 512 / 1024 cells a side; C4, 10,000 primitives: 2.88 / 2.76 / 2.82)
            if (c->S.nsb >= 32768 && !ab_knob("RTX_CELL_G")) G0 = 1024;
            constexpr size_t kMaxLists = (size_t)1 << 25;  // (one build wave per list: a launch holds fewer than 2^32 threads)
            while ((size_t)c->S.nl * G0 * G0 > kMaxLists && G0 > 8) G0 /= 2;
            // entries a table may hold: 2^31 by its int offsets, an eighth of the workspace budget by size (8 B each)
            const uint64_t wbytes = params->workspace_bytes ? params->workspace_bytes : ((uint64_t)32 << 30);
            const uint64_t max_entries = std::min<uint64_t>(((uint64_t)1 << 31) - 64, wbytes / 8 / sizeof(int2));
            int cells_tight = 1;  // entries by what the exact tests can accept (rtx_cells.hip.h); 0: by what the per-ray filter can pass
            if (const char* e = ab_knob("RTX_CELL_TIGHT")) cells_tight = std::atoi(e) ? 1 : 0;
            // the guesses: 512 entries per caster and light for the plain table (C4 has 60, C5 85; lights inside the scene, large or
            // long primitives take 200 ... 2,600 -- tools/dbg/perf_survey.py -- and a table that outgrows its guess is built twice:
            // room that stays empty costs address space, a second build costs the first call 10 ... 30 ms), 16,384 per affine sphere
            // and light (C4: 14,150).  (The affine lists gain nothing from cells finer than 256 a side -- what they list is set by the slabs
            // of origin distance, 13.8 M filter tests per C4 frame either way -- and 33 M lists take 29 ms to build against 6.)
            uint64_t tab_entries[2] = {0, 0}, tab_lists[2] = {0, 0};
            const bool verbose_tab = ab_knob("RTX_VERBOSE") != nullptr;
            if (!c->d_cursor && (rc = c->dalloc(&c->d_cursor, 1))) return rc;  // (ahead of tab_mark: the camera's table uses it too)
            const size_t tab_mark = c->allocs.size();
            {
                TableOut to;
                TableSpec ts{0, c->S.nl, G0, 8, kMaxLists, (uint64_t)c->S.nl * (uint64_t)c->S.nsb * 512, max_entries, d_dc, cells_tight, CamCells{}};
                rc = build_cell_table(c, ts, &to);
                c->cells.off = to.off; c->cells.ent = static_cast<const int2*>(to.ent); c->cells.G = to.G;
                tab_entries[0] = to.entries; tab_lists[0] = to.lists;
                if (verbose_tab && to.off) std::fprintf(stderr, "rtx: plain table G = %d, %llu lists, %llu entries (room for %llu, attempt %d)\n", to.G, (unsigned long long)to.lists, (unsigned long long)to.entries, (unsigned long long)to.room, to.attempt);
                phase("plain table");
                // (without the plain table the packet kernels trace everything, with their own affine stage)
                if (!rc && c->cells.off && c->S.naff > 0) {
                    ts.kind = 1; ts.max_lists = (size_t)1 << 24; ts.guess = (uint64_t)c->S.nl * (uint64_t)c->S.naff * 16384;
                    rc = build_cell_table(c, ts, &to);
                    c->cells.aoff = to.off; c->cells.aent = static_cast<const int*>(to.ent); c->cells.aG = to.G;
                    tab_entries[1] = to.entries; tab_lists[1] = to.lists;
                    if (verbose_tab && to.off) std::fprintf(stderr, "rtx: affine table G = %d, %llu lists, %llu entries (room for %llu, attempt %d)\n", to.G, (unsigned long long)to.lists, (unsigned long long)to.entries, (unsigned long long)to.room, to.attempt);
                    phase("affine table");
                }
            }
            if (rc) return rc;
            c->cells.dmax = d_dc;
            // tight entries go straight to the exact stages: the per-ray filter in front of them rejected one pair in five and cost
            // a gather and eleven instructions per pair (C4 2.85 -> 2.75 ms, C5 47.1 -> 43.9 ms); filter-conservative entries keep it
            c->cells.nofilter = cells_tight;
            if (const char* e = ab_knob("RTX_CELL_NOFILTER")) c->cells.nofilter = std::atoi(e) ? 1 : 0;
            c->table_allocs.assign(c->allocs.begin() + tab_mark, c->allocs.end());
            c->use_cells = c->cells.off != nullptr;
            c->shadow_direct = c->cells.off && (c->S.naff == 0 || c->cells.aoff);
            if (const char* e = ab_knob("RTX_SHADOW_DIRECT")) c->shadow_direct = c->shadow_direct && std::atoi(e) != 0;
            // Which form the shadow stage takes is measured at the end of creation (both forms render this context's frame) --
            // unless the answer is plain: forced by a flag, or lists so short that the walk cannot lose.  What a shadow ray pays in
            // the table form is its list (4 entries a step); the packet form lost on every frame measured whose lists average
            // below 30 entries (C4: 2.3 plain and 3.4 affine, 2.5 against 4.2 ms; C5: 8 per list, 40 against 85 ms), and on every
            // scene of tools/dbg/perf_survey.py that was still measured with lists of 16 ... 50 (a floor and walls: 22 per affine
            // list; long triangles: 50 per plain list) -- the one scene the packets won, 3,000 rotated ellipsoids a seventh of
            // which no record bounds, averages above 250.  So lists of <= 64 on average skip the measurement, the packet form's
            // bit masks and its queues: 18 ms of C4's 74 ms creation at the time, 26 ... 71 ms of those scenes' 70 ... 130.
            {
                int force = -1;
                if (params->flags & RTX_F_SHADOW_PACKETS) force = 0;
                if (params->flags & RTX_F_SHADOW_TABLES) force = 1;
                if (const char* e = ab_knob("RTX_SHADOW_FORM")) force = std::atoi(e) ? 1 : 0;
                const bool shortl = tab_entries[0] <= 64 * tab_lists[0] && tab_entries[1] <= 64 * tab_lists[1];
                c->form_early = c->use_cells && c->shadow_direct && (force == 1 || (force < 0 && shortl));
                if (ab_knob("RTX_FORM_MEASURE") && force < 0) c->form_early = false;
            }
        }
        c->S.aff_cells = nullptr; c->S.aff_dt = nullptr; c->S.aff_words = 0; c->S.aff_G = 0;
        const size_t bitmask_mark = c->allocs.size();
        if (c->S.naff > 0 && c->S.naff <= 4096 && c->S.nl > 0 && !c->form_early) {
            // (light, direction cell) -> affine spheres bit masks for the packet form's k_shadow_affine (k_affine_cells)
            const int words = (c->S.naff + 63) / 64;
            int G = 256;
            const size_t tab_limit = c->arena ? c->arena_cap / 8 : ((size_t)256 << 20);  // arena mode: the table may take an eighth of the block
            while ((size_t)c->S.nl * G * G * words * 8 > tab_limit && G > 16) G /= 2;
            if ((size_t)c->S.nl * G * G * words * 8 <= tab_limit) {
                unsigned long long* tab = nullptr;
                if ((rc = c->dalloc(&tab, (size_t)c->S.nl * G * G * words))) return rc;
                const int ncell = c->S.nl * G * G;
                hipLaunchKernelGGL(k_affine_cells, dim3((unsigned)((ncell + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, 0, c->S, G, d_dt, tab, words);
                HIPCHK(hipGetLastError());
                HIPCHK(hipDeviceSynchronize());
                c->S.aff_cells = tab; c->S.aff_dt = d_dt; c->S.aff_words = words; c->S.aff_G = G;
            }
        }
        c->packet_allocs.assign(c->allocs.begin() + bitmask_mark, c->allocs.end());
        phase("affine bit masks");
        float* ix = nullptr; float* iy = nullptr;
        if ((rc = c->upload(&ix, P.imagex))) return rc;
        if ((rc = c->upload(&iy, P.imagey))) return rc;
        c->C.imagex = ix; c->C.imagey = iy;
        c->C.ncrossv = {P.ncrossv.x, P.ncrossv.y, P.ncrossv.z};
        c->C.up = {P.up.x, P.up.y, P.up.z};
        c->C.vpn = {P.vpn.x, P.vpn.y, P.vpn.z};
        c->C.camera = {P.camera.x, P.camera.y, P.camera.z};
        c->C.orig = {P.orig.x, P.orig.y, P.orig.z};
        // ---- the camera's table: primary rays all leave C.orig, so which primitives one of them can hit is a function of its
        //      direction alone (rtx_cells.hip.h, camera mode).  Built for the cells inside the frame's view cone only. ----
        c->cam = CamTab{};
        bool want_cam = !c->arena && ntrace > 0 && !(params->flags & (RTX_F_EXACT_ONLY | RTX_F_COUNT_TESTS)) && c->S.cseg != nullptr && !P.cc.cl.empty();
        {   // records that always pass (polygons, planes, affine spheres with a translation row) sit in EVERY list: a scene with
            // hundreds of them keeps the packet kernel, whose hierarchy holds them once
            size_t unbounded = 0;
            for (const BoundSphere& b : P.bounds) unbounded += (b.neg_r2 == -std::numeric_limits<float>::infinity()) ? 1 : 0;
            if (unbounded > 256) want_cam = false;
        }
        if (const char* e = ab_knob("RTX_CAM")) want_cam = want_cam && std::atoi(e) != 0;
        if (want_cam) {
            // the view cone: the four corner rays as k_raygen forms them (double precision here, 1e-4 of slack on the cosine)
            auto dir_of = [&](int x, int y, double out[3]) {
                const double a = (double)P.imagex[x] - P.camera.x, b = (double)P.imagey[y] - P.camera.y, cz = -(double)P.camera.z;
                double v[3] = {a * P.ncrossv.x + b * P.up.x + cz * P.vpn.x, a * P.ncrossv.y + b * P.up.y + cz * P.vpn.y, a * P.ncrossv.z + b * P.up.z + cz * P.vpn.z};
                const double nn = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                for (int q = 0; q < 3; q++) out[q] = nn > 0 ? v[q] / nn : 0.0;
            };
            double cr[4][3], ax[3] = {0, 0, 0};
            dir_of(0, 0, cr[0]); dir_of(W - 1, 0, cr[1]); dir_of(0, H - 1, cr[2]); dir_of(W - 1, H - 1, cr[3]);
            for (auto& v : cr) for (int q = 0; q < 3; q++) ax[q] += v[q];
            const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            double cosv = 1.0;
            for (auto& v : cr) cosv = std::min(cosv, (v[0] * ax[0] + v[1] * ax[1] + v[2] * ax[2]) / (an > 0 ? an : 1.0));
            cosv -= 1e-4;
            if (!(an > 1e-6) || !std::isfinite(cosv) || cosv < -1.0) { cosv = -1.0; ax[0] = 0; ax[1] = 0; ax[2] = an > 1e-6 ? an : 1.0; }  // no usable cone: every cell
            CamCells cm{};
            cm.on = 1;
            cm.apex = {P.orig.x, P.orig.y, P.orig.z};
            cm.axis = {(float)(ax[0] / (an > 1e-6 ? an : 1.0)), (float)(ax[1] / (an > 1e-6 ? an : 1.0)), (float)(ax[2] / (an > 1e-6 ? an : 1.0))};
            cm.cosv = (float)cosv;
            cm.sinv = (float)std::sqrt(std::max(0.0, 1.0 - cosv * cosv)) + 1e-4f;
            // range: the far corner of the scene box from the camera (the entries' rho is per entry; this scales the absolute slack)
            double far2 = 0;
            {
                const float lp[3] = {P.orig.x, P.orig.y, P.orig.z};
                const float lo[3] = {c->S.box_lo.x, c->S.box_lo.y, c->S.box_lo.z};
                const float inv[3] = {c->S.box_inv.x, c->S.box_inv.y, c->S.box_inv.z};
                for (int a = 0; a < 3; a++) {
                    const double hi = (double)lo[a] + 16.0 / (double)inv[a];
                    const double dd = std::max(std::fabs((double)lp[a] - lo[a]), std::fabs((double)lp[a] - hi));
                    far2 += dd * dd;
                }
            }
            const double rng = std::sqrt(far2) * 1.01 + 1.0;
            const bool apex_ok = std::isfinite(rng) && rng < 1e15 && std::isfinite(P.orig.x) && std::isfinite(P.orig.y) && std::isfinite(P.orig.z);
            std::vector<float> dcam(1, apex_ok ? (float)rng : 0.0f);
            float* d_dcam = nullptr;
            if (apex_ok && (rc = c->upload(&d_dcam, dcam))) return rc;
            // cells a side: 512, and 1,024 for dense scenes as the lights' tables take (C4: 2.271 / 2.268 / 2.254 ms per frame with 512 /
            // 1,024 / 2,048, building them 1.6 / 4.9 / 17 ms; C5, 100,000 triangles: 33.2 / 32.7 / 32.6 ms per frame)
            int Gc = (c->S.nsb >= 32768) ? 1024 : 512;
            if (const char* e = ab_knob("RTX_CAM_G")) Gc = std::max(8, std::min(4096, std::atoi(e)));
            const uint64_t wbytes = params->workspace_bytes ? params->workspace_bytes : ((uint64_t)32 << 30);
            const uint64_t max_entries = std::min<uint64_t>(((uint64_t)1 << 31) - 64, wbytes / 8 / sizeof(int2));
            const uint64_t nprim_all = (uint64_t)P.bounds.size();
            if (apex_ok) {
                TableOut to;
                const TableSpec ts{2, 1, Gc, 64, (size_t)1 << 25, nprim_all * 256, max_entries, d_dcam, 1, cm};
                if ((rc = build_cell_table(c, ts, &to))) return rc;
                if (to.off) {
                    c->cam.off = to.off; c->cam.ent = static_cast<const int2*>(to.ent); c->cam.G = to.G; c->cam.apex = cm.apex;
                    if (ab_knob("RTX_VERBOSE")) std::fprintf(stderr, "rtx: camera table G = %d, %llu entries (room for %llu, attempt %d), view cone cos %.4f\n", to.G, (unsigned long long)to.entries, (unsigned long long)to.room, to.attempt, cosv);
                }
            }
            phase("camera table");
        }
    }
    if ((rc = c->upload(&c->d_trace_rows, c->trace_rows))) return rc;
    if ((rc = c->upload(&c->d_row2ti, c->row2ti))) return rc;
    if ((rc = c->upload(&c->d_own_rows, c->own_rows))) return rc;
    if ((rc = c->dalloc(&c->d_fimg, (size_t)ntrace * W * 3))) return rc;
    if ((rc = c->dalloc(&c->d_rgb8, (size_t)nown * W * 3))) return rc;
    if ((rc = c->dalloc(&c->d_cnt, (size_t)2 * levels + 2))) return rc;
    // frame totals, test counters and packet statistics in one block: one memset per frame
    if ((rc = c->dalloc(&c->d_totals, (size_t)2 * levels + 16 + 3 * EX_KINDS))) return rc;
    c->d_tests = c->d_totals + 2 * levels;
    c->d_pstat = c->d_tests + 8;
    c->S.pstat = (params->flags & RTX_F_PACKET_STATS) ? c->d_pstat : nullptr;
    c->S.exec = c->d_pstat + 8;  // written by the counting instantiations only (RTX_F_COUNT_EXEC)

    if (c->serial) c->main_from = 0;  // every generation's shadow chain on the caller's stream, one set of queue buffers
    else if (levels >= 2) c->main_from = levels - 1;
    const int nside = c->serial ? 0 : std::min(levels, (int)rtx_context::kSide);
    const int nlanes = nside + (c->main_from < levels ? 1 : 0);
    const double fan = c->fanout >= 2 ? 2.0 : 1.0;
    const size_t nl1 = (size_t)std::max(c->S.nl, 1);
    c->binG = 256;
    // Finer maps where a cell of the 256 map would hold several waves of rays, each as wide as the cell, and where the
    // scene is dense (the denser the scene, the more a narrower packet saves): 512 x 512 cells per light for frames of
    // >= 24 M primary rays (8K), 1024 x 1024 in scenes of >= 32,768 shadow casters at any frame size.  Measured on C5
    // (100,000 triangles, 16 lights): 119.5 ms with 256, 106.2 with 512, 99.0 with 1024, 102.7 with 2048; a rank of 8 of
    // it 22.5 / 18.8 / 16.3 ms.  The 4K frame of C4 (10,000 primitives) has 126 rays per cell of the 256 map and loses
    // with a finer one (4.02 -> 4.10 ms).  The block of the no-allocation entry point keeps 256 (the histograms' size).
    if (!c->arena) {
        if ((int64_t)ntrace * W >= (int64_t)24 << 20) c->binG = 512;
        if (c->S.nsb >= 32768) c->binG = 1024;
    }
    if (const char* bg = ab_knob("RTX_BIN_G")) c->binG = std::max(8, std::min(4096, std::atoi(bg)));  // A/B aid (same image)
    while (nl1 * c->binG * c->binG > ((size_t)1 << 24) && c->binG > 8) c->binG /= 2;
    c->nbins = (int)nl1 * c->binG * c->binG;
    c->qpad = (size_t)64 * ((size_t)(c->nbins + 1023) / 1024 + 1);
    // ---- ray-queue capacities -----------------------------------------------------------------------------------
    // Level k of a chunk holds at most fanout^k rays per level-0 ray.  With fan-out 1 that bound is the pixel count and
    // is used as it is.  With fan-out 2 (some primitive both reflects and refracts) it is 31x the pixels at depth 5 and
    // 255x at depth 8 while the rays that exist are a few tenths of a pixel count per level: there the capacities come
    // from a CALIBRATION pass at creation -- a context renders one fixed frame, so its ray counts per level and row
    // block are a property of the context: trace closest hit + set-up (no shadows, no shading) once in worst-case
    // sized blocks of r_cal rows, keep the per-block counts, and size every level for the largest chunk of the final
    // layout (chunks are whole numbers of calibration blocks, so their counts are sums of measured ones: exact).
    // k_setup still refuses to write past a level's capacity and reports it (rtx_stats.queue_overflow), and every
    // consumer clamps the live count to the capacity: a wrong count costs rays, never memory safety.
    struct Caps { std::vector<size_t> cap; };
    auto worst_caps = [&](int64_t rows) {
        Caps cp; cp.cap.resize(levels);
        double cap = (double)rows * W;
        for (int k = 0; k < levels; k++) { cp.cap[k] = (size_t)cap; cap *= fan; }
        return cp;
    };
    // Device bytes of the ray queues for these capacities, allocation by allocation as alloc_levels / alloc_shadow do
    // (arena mode rounds every allocation up to 256 bytes); nchunks only sizes the per-chunk counters.
    auto plan_levels = [&](const Caps& cp, bool with_lit) -> double {
        auto al = [&](double b) { return std::ceil(std::max(b, 1.0) / 256.0) * 256.0; };  // (arena and slab both round up)
        double tot = 0;
        for (int k = 0; k < levels; k++) {
            const double cap = (double)cp.cap[k];
            tot += (k ? 19.0 : 18.0) * al(cap * 4.0) + (k ? al(cap * 12.0) : 0.0) + (with_lit ? al(cap * (double)nl1) : 0.0);  // (LevelDev::src: generations >= 1)
        }
        return tot;
    };
    auto plan_shadow = [&](const Caps& cp, double nch) -> double {
        if (c->form_early) return std::ceil(std::max(nch * levels * 4.0 * 2.0, 1.0) / 256.0) * 256.0;  // tables: no shadow queues
        auto al = [&](double b) { return std::ceil(std::max(b, 1.0) / 256.0) * 256.0; };
        size_t mx = 0;
        for (size_t v : cp.cap) mx = std::max(mx, v);
        const double qcap = (double)mx * (double)nl1 + (double)c->qpad;
        const double nb = (double)c->nbins, ng = std::ceil(nb / 1024.0);
        return nlanes * (2.0 * al(qcap * 16.0) + 2.0 * al(nb * 4.0) + al(qcap * 4.0) + 2.0 * al(ng * 4.0)) + al(nch * levels * 4.0 * 2.0);
    };
    auto too_big = [&](const Caps& cp) {
        size_t mx = 0;
        for (size_t v : cp.cap) mx = std::max(mx, v);
        return (double)mx * (double)nl1 + (double)c->qpad > (double)(((int64_t)1 << 30));
    };
    auto alloc_levels = [&](const Caps& cp, bool with_lit, size_t extra = 0) -> int {
        c->L.assign(levels, LevelDev{});
        {   // one block for the lot (plan_levels counts what the loop below carves, cap >= 1 each)
            Caps c1 = cp;
            for (size_t& v : c1.cap) v = std::max<size_t>(v, 1);
            if ((rc = c->slab_begin((size_t)plan_levels(c1, with_lit) + rtx_context::round256(extra)))) return rc;
        }
        for (int k = 0; k < levels; k++) {
            LevelDev& L = c->L[k];
            const size_t cap = std::max<size_t>(cp.cap[k], 1);
            if (cap > ((size_t)1 << 30)) return RTX_E_TOO_LARGE;
            L.cap = (int)cap;
            float** fl[] = {&L.ox, &L.oy, &L.oz, &L.dx, &L.dy, &L.dz, &L.t, &L.n_nx, &L.n_ny, &L.n_nz, &L.n_px, &L.n_py, &L.n_pz, &L.n_fres};
            for (float** f : fl)
                if ((rc = c->dalloc(f, cap))) return rc;
            int** il[] = {&L.prim, &L.n_ray, &L.n_crefl, &L.n_crefr};
            for (int** f : il)
                if ((rc = c->dalloc(f, cap))) return rc;
            if (k > 0 && (rc = c->dalloc(&L.src, cap))) return rc;  // (primary rays have no parent; only k_closest_lanes reads it)
            if (k == 0) L.val = nullptr;  // level-0 colours are written straight into the float image
            else if ((rc = c->dalloc(&L.val, cap * 3))) return rc;
            if (with_lit && (rc = c->dalloc(&L.lit, cap * nl1))) return rc;
        }
        return RTX_OK;
    };
    uint64_t budget = params->workspace_bytes ? params->workspace_bytes : ((uint64_t)32 << 30);
    if (c->arena) budget = c->arena_cap - c->arena_off;
    if ((rc = c->dalloc(&c->d_overflow, 1))) return rc;
    HIPCHK(hipMemset(c->d_overflow, 0, sizeof(int)));

    Caps caps;
    int64_t rows = ntrace > 0 ? ntrace : 1;
    // (fan-out 1 scenes are calibrated too: their bound is the pixel count at EVERY level while the deep generations hold a
    // few per cent of that, and every launch covers a level's capacity: 5.77 -> 5.59 ms per 4K frame, 2 GB less memory)
    const bool calibrate = c->fanout >= 1 && levels >= 2 && ntrace > 0 && !(params->flags & (RTX_F_EXACT_ONLY | RTX_F_COUNT_TESTS | RTX_F_WORST_CASE_QUEUES));
    if (!calibrate) {
        // worst case = what exists (fan-out 1), or a mode that is not worth a calibration pass
        auto fits = [&](int64_t r) { const Caps cp = worst_caps(r); return !(plan_levels(cp, true) + plan_shadow(cp, ntrace ? std::ceil((double)ntrace / (double)r) : 1.0) > (double)budget || too_big(cp)); };
        if (!fits(rows)) {
            int64_t lo = 0, hi = rows;  // lo fits (0 rows: nothing), hi does not; fits() is monotone
            while (hi - lo > 1) {
                const int64_t mid = lo + (hi - lo) / 2;
                if (fits(mid)) lo = mid; else hi = mid;
            }
            rows = lo;
        }
        if (rows < 1) {
            if (c->arena || too_big(worst_caps(1))) return RTX_E_TOO_LARGE;  // not even one row of queues fits
            rows = 1;  // a workspace_bytes cap below one row is a hint, not a limit
        }
        caps = worst_caps(rows);
    } else {
        // (a) calibration blocks: worst-case capacities for r_cal rows within a bounded scratch budget
        const uint64_t cal_budget = std::min<uint64_t>(budget, (uint64_t)4 << 30);
        int64_t r_cal = ntrace;
        // a block also fits the final layout at its worst case, so one block per chunk (m = 1 below) always fits the budget
        auto cal_fits = [&](int64_t r) {
            const Caps cp = worst_caps(r);
            return !(plan_levels(cp, false) > (double)cal_budget || too_big(cp) ||
                     plan_levels(cp, true) + plan_shadow(cp, std::ceil((double)ntrace / (double)r)) > (double)budget);
        };
        if (!cal_fits(r_cal)) {
            int64_t lo = 0, hi = r_cal;
            while (hi - lo > 1) {
                const int64_t mid = lo + (hi - lo) / 2;
                if (cal_fits(mid)) lo = mid; else hi = mid;
            }
            r_cal = lo;
        }
        if (r_cal < 1) {
            if (c->arena || too_big(worst_caps(1))) return RTX_E_TOO_LARGE;
            r_cal = 1;
        }
        const int nblk = (int)((ntrace + r_cal - 1) / r_cal);
        std::vector<int> blk((size_t)nblk * levels, 0);
        {
            const size_t mark_allocs = c->allocs.size(), mark_arena = c->arena_off;
            int* d_blk = nullptr;
            if ((rc = alloc_levels(worst_caps(r_cal), false, blk.size() * sizeof(int)))) return rc;
            if ((rc = c->dalloc(&d_blk, blk.size()))) return rc;
            c->slab_end();
            c->chunk_rows = (int)r_cal;
            if ((rc = calibrate_counts(c, d_blk))) return rc;
            HIPCHK(hipMemcpy(blk.data(), d_blk, blk.size() * sizeof(int), hipMemcpyDeviceToHost));
            // give the blocks back: hipFree in reverse order / rewind the arena
            for (size_t i = c->allocs.size(); i > mark_allocs; i--) HIPCHK(hipFree(c->allocs[i - 1]));
            c->allocs.resize(mark_allocs);
            c->arena_off = mark_arena;
            c->L.clear();
        }
        // (b) the largest chunk (a whole number m of calibration blocks) whose measured capacities fit the budget
        auto caps_for = [&](int m) {
            Caps cp; cp.cap.assign(levels, 0);
            for (int b0 = 0; b0 < nblk; b0 += m)
                for (int k = 0; k < levels; k++) {
                    size_t sum = 0;
                    for (int b = b0; b < std::min(nblk, b0 + m); b++) sum += (size_t)blk[(size_t)b * levels + k];
                    cp.cap[k] = std::max(cp.cap[k], sum);
                }
            for (int k = 0; k < levels; k++) cp.cap[k] = (cp.cap[k] + 255) / 256 * 256;  // whole workgroups
            return cp;
        };
        auto fits = [&](int m) { const Caps cp = caps_for(m); return !(plan_levels(cp, true) + plan_shadow(cp, std::ceil((double)nblk / (double)m)) > (double)budget || too_big(cp)); };
        int m = nblk;
        if (!fits(m)) {
            int lo = 0, hi = m;  // more blocks per chunk never needs less memory
            while (hi - lo > 1) {
                const int mid = lo + (hi - lo) / 2;
                if (fits(mid)) lo = mid; else hi = mid;
            }
            m = lo;
        }
        if (m < 1) {
            if (c->arena) return RTX_E_TOO_LARGE;
            m = 1;
        }
        rows = std::min<int64_t>((int64_t)m * r_cal, ntrace);
        caps = caps_for(m);
        c->calibrated = true;
    }
    phase("calibration");
    c->chunk_rows = (int)rows;
    // Two-level member selection in the shadow pass (rtx_shadow.hip.h, shadow_select) for frames that fill the chip:
    // chunks of >= 2^22 primary rays (a whole 4K frame or half of one; any 8K share).  Thinner frames -- a rank of 4 or 8
    // of the 4K frame -- are bound by a wave's chain of dependent steps, which the extra level lengthens.  RTX_TWO_LEVEL=0/1
    // forces it (A/B aid; same image).
    c->S.sel2 = ((int64_t)rows * W >= ((int64_t)1 << 22)) ? 1 : 0;
    if (const char* e2 = ab_knob("RTX_TWO_LEVEL")) c->S.sel2 = std::atoi(e2) ? 1 : 0;
    // Generation 0's shadow rays take the walk when they fill the chip and the pooled stages when they do not (chunks of
    // <= 3 * 2^20 primary rays: a rank's share of 4 or 8 of the 4K frame, halo rows included; measured 1.20 -> 1.04 ms for a rank of 8, 1.59 -> 1.55
    // for a rank of 4, 2.59 -> 2.67 for a rank of 2, 4.03 -> 4.18 for the whole frame).
    if ((int64_t)rows * W <= ((int64_t)3 << 20)) c->pool_from = 0;
    if (const char* pf = ab_knob("RTX_SHADOW_POOL_FROM")) c->pool_from = std::atoi(pf);  // A/B aid; same image either way
    if ((rc = alloc_levels(caps, true))) return rc;
    c->slab_end();
    c->nframe_chunks = ntrace ? (ntrace + c->chunk_rows - 1) / c->chunk_rows : 0;
    if (c->form_early) {
        if ((rc = c->dalloc(&c->d_scnt, (size_t)std::max(c->nframe_chunks, 1) * levels * 2))) return rc;
    } else {   // shadow-ray queues: the largest generation's hits x lights
        size_t mx = 0;
        for (int k = 0; k < levels; k++) mx = std::max(mx, (size_t)c->L[k].cap);
        size_t qcap = mx * nl1 + c->qpad;
        if (qcap > ((size_t)1 << 31) - 1) return RTX_E_TOO_LARGE;
        const size_t maxbins = (size_t)c->nbins;
        const size_t qmark = c->allocs.size();
        {
            const size_t ng = (maxbins + 1023) / 1024;
            auto r256 = rtx_context::round256;
            const size_t per = 2 * r256(qcap * sizeof(*c->side[0].q.a)) + 2 * r256(maxbins * 4) + r256(qcap * 4) + 2 * r256(ng * 4);
            if ((rc = c->slab_begin(per * (size_t)nlanes))) return rc;
        }
        for (int i = 0; i <= nside; i++) {
            if (i == nside && c->main_from >= levels) break;
            rtx_context::ShadowLane& sl = c->side[i == nside ? (int)rtx_context::kSide : i];
            if ((rc = c->dalloc(&sl.q.a, qcap))) return rc;
            if ((rc = c->dalloc(&sl.q.b, qcap))) return rc;
            if ((rc = c->dalloc(&sl.hist, maxbins))) return rc;
            if ((rc = c->dalloc(&sl.offs, maxbins))) return rc;
            if ((rc = c->dalloc(&sl.rank, qcap))) return rc;
            if ((rc = c->dalloc(&sl.bsum, (maxbins + 1023) / 1024))) return rc;
            if ((rc = c->dalloc(&sl.braw, (maxbins + 1023) / 1024))) return rc;
        }
        c->slab_end();
        c->packet_allocs.insert(c->packet_allocs.end(), c->allocs.begin() + qmark, c->allocs.end());
        c->nframe_chunks = ntrace ? (ntrace + c->chunk_rows - 1) / c->chunk_rows : 0;
        if ((rc = c->dalloc(&c->d_scnt, (size_t)std::max(c->nframe_chunks, 1) * levels * 2))) return rc;
    }
    phase("queue allocation");
    if (c->arena) return RTX_OK;  // arena mode creates no streams and no events
    HIPCHK(hipEventCreate(&c->ev0));
    HIPCHK(hipEventCreate(&c->ev1));
    for (int i = 0; i < nside; i++) {
        HIPCHK(hipStreamCreateWithFlags(&c->side[i].stream, hipStreamNonBlocking));
    }
    c->ev_fork.resize(levels);
    for (auto& e : c->ev_fork) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_pass.resize(levels);
    for (auto& e : c->ev_pass) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (params->flags & RTX_F_TIME_KERNELS) {
        int nchunks = ntrace ? (ntrace + c->chunk_rows - 1) / c->chunk_rows : 0;
        c->kev.resize((size_t)nchunks * levels * 2 * 2);  // (start, stop) x (closest, shadow) per generation and chunk
        for (auto& e : c->kev) HIPCHK(hipEventCreate(&e));
    }
    // ---- which form of the shadow stage this context runs: measured, not guessed -------------------------------------
    // Two forms exist for the plain primitives and the affine spheres: the (light, cell) tables walked straight off the hit
    // records (rtx_cells.hip.h) and the sorted queue traced by packets (k_shadow_pass).  Which is faster depends on how
    // many casters a shadow ray's line meets (C4, 10,000 primitives: tables 3.1 against 4.2 ms; C5, 100,000 triangles in
    // a box three times as deep: packets 100 against 135 ms), so a context whose chunks hold >= 2^18 primary rays renders
    // its own frame twice in either form (the second one timed) and keeps the faster; the other form's buffers are freed.
    // Smaller frames take the tables (the shorter chain of launches).  The images are the same either way.
    {
        int force = -1;
        if (params->flags & RTX_F_SHADOW_PACKETS) force = 0;
        if (params->flags & RTX_F_SHADOW_TABLES) force = 1;
        if (const char* e = ab_knob("RTX_SHADOW_FORM")) force = std::atoi(e) ? 1 : 0;  // 0 packets, 1 tables
        const bool can_tables = c->use_cells && c->shadow_direct;
        bool tables = can_tables;
        if (c->form_early) {
            tables = true;
        } else if (can_tables && force < 0 && ntrace > 0 && (int64_t)c->chunk_rows * W >= ((int64_t)1 << 18) && c->S.nl > 0) {
            float ms[2] = {0.0f, 0.0f};
            for (int form = 0; form < 2; form++) {
                c->use_cells = form == 1;
                for (int it = 0; it < 2; it++)
                    if ((rc = context_render_impl(c, nullptr, nullptr))) return rc;
                HIPCHK(hipStreamSynchronize(nullptr));
                HIPCHK(hipEventElapsedTime(&ms[form], c->ev0, c->ev1));
            }
            tables = ms[1] <= ms[0];
            c->rendered = false;
            phase("timing the two forms");
        } else if (force >= 0) {
            tables = can_tables && force == 1;
        }
        if (tables) {
            c->use_cells = true;
            c->dfree_group(c->packet_allocs);
            for (rtx_context::ShadowLane& sl : c->side) { sl.q = {}; sl.hist = sl.offs = sl.rank = sl.bsum = sl.braw = nullptr; }
            c->S.aff_cells = nullptr;
        } else if (c->cells.off && !can_tables && force != 0) {
            // tables for the plain primitives but none for this scene's affine spheres: k_shadow_cells over the sorted
            // queue, k_shadow_affine behind it -- everything stays
            c->use_cells = true;
            c->shadow_direct = false;
        } else {
            c->use_cells = false;  // packets: the tables go
            c->shadow_direct = false;
            c->dfree_group(c->table_allocs);
            c->cells = CellTab{};
        }
        phase("freeing the other form");
    }
    return RTX_OK;
}

static inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static inline unsigned grid_for_isect(size_t cap);
// k_closest_lanes: 64 rays a workgroup for generations of <= kSplitClosest rays, 16 for <= kSplitClosest4 (the kernel reads the live count)
static inline unsigned grid_lanes_thin(size_t cap, int s16, int s4) { return std::max((unsigned)((std::min<size_t>(cap, (size_t)s16) + 63) / 64), (unsigned)((std::min<size_t>(cap, (size_t)s4) + 15) / 16)); }

// Calibration pass (context_create_impl): closest hit + set-up of every generation, block of chunk_rows rows by block,
// on the default stream; d_blk[block * levels + k] = rays of generation k in that block.  The levels hold worst-case
// capacities here; no shadow rays, no shading.
static int calibrate_counts(rtx_context* c, int* d_blk)
{
    const int W = c->width, levels = c->levels, ntrace = (int)c->trace_rows.size();
    hipStream_t st = nullptr;
    const CntPtr rays{c->d_cnt + 1}, nodes{c->d_cnt + 2};
    int blk = 0;
    for (int ti0 = 0; ti0 < ntrace; ti0 += c->chunk_rows, blk++) {
        const int nrows = std::min(c->chunk_rows, ntrace - ti0);
        const size_t npix = (size_t)nrows * W;
        HIPCHK(hipMemsetAsync(c->d_cnt, 0, sizeof(int) * (2 * (size_t)levels + 2), st));
        LevelDev L0 = c->L[0];
        L0.val = c->d_fimg + (size_t)ti0 * W * 3;
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(npix)), dim3(kBlock), 0, st, L0, c->C, c->d_trace_rows, ti0, nrows, W, rays + 0);
        size_t live_cap = npix;
        for (int k = 0; k < levels; k++) {
            LevelDev Lk = (k == 0) ? L0 : c->L[k];
            LevelDev Ln = (k + 1 < levels) ? c->L[k + 1] : Lk;
            const int spawn = (k < c->params.max_depth && k + 1 < levels) ? 1 : 0;
            if (!spawn) break;  // the last generation spawns nothing: its count is already known
            const unsigned g = grid_for(live_cap);
            if (k > 0) hipLaunchKernelGGL(k_closest_lanes<false>, dim3(std::max(g, grid_lanes_thin(live_cap, c->lanes16, c->lanes4))), dim3(kBlock), 0, st, c->lanes_scene(), Lk, rays + k, c->lanes16, c->lanes4);
            else if (live_cap < kPrefetchRays) hipLaunchKernelGGL((k_closest_f<true, false>), dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k);
            else hipLaunchKernelGGL((k_closest_f<false, false>), dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k);
            hipLaunchKernelGGL(k_setup<kBlock>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, Ln, rays + k, nodes + k, spawn, c->d_overflow);
            live_cap = std::min(live_cap * 2, (size_t)c->L[k + 1].cap);
        }
        HIPCHK(hipMemcpy2DAsync(d_blk + (size_t)blk * levels, sizeof(int), rays + 0, 2 * sizeof(int), sizeof(int), (size_t)levels, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return RTX_OK;
}
// intersection kernels: thin generations (<= kThinRays live rays) use 64 rays per workgroup
static inline unsigned grid_for_isect(size_t cap) { return std::max(grid_for(cap), (unsigned)((std::min<size_t>(cap, kThinRays) + 63) / 64)); }

static int context_render_impl(rtx_context* c, hipStream_t st, uint8_t* d_out)
{
    HIPCHK(hipSetDevice(c->device));
    const int W = c->width, levels = c->levels;
    const int ntrace = (int)c->trace_rows.size(), nown = (int)c->own_rows.size();
    if (c->ev0) HIPCHK(hipEventRecord(c->ev0, st));
    HIPCHK(hipMemsetAsync(c->d_totals, 0, sizeof(unsigned long long) * (2 * levels + 16 + 3 * EX_KINDS), st));  // + d_tests, d_pstat, exec
    // (d_scnt needs no clearing: k_scan_sums writes a generation's queue size before its pass reads it)
    const bool count = (c->params.flags & RTX_F_COUNT_TESTS) != 0;
    const bool exact_only = (c->params.flags & RTX_F_EXACT_ONLY) != 0;
    const bool exec = (c->params.flags & RTX_F_COUNT_EXEC) != 0;
    const bool timek = !c->kev.empty();
    size_t kev_used = 0;
    c->kev_kind.clear();
    int nchunks = 0;
    for (int ti0 = 0; ti0 < ntrace; ti0 += c->chunk_rows, nchunks++) {
        int nrows = std::min(c->chunk_rows, ntrace - ti0);
        size_t npix = (size_t)nrows * W;
        const CntPtr rays{c->d_cnt + 1}, nodes{c->d_cnt + 2};
        HIPCHK(hipMemsetAsync(c->d_cnt, 0, sizeof(int) * (2 * (size_t)levels + 2), st));
        LevelDev L0 = c->L[0];
        L0.val = c->d_fimg + (size_t)ti0 * W * 3;
        hipLaunchKernelGGL(k_raygen, dim3(grid_for(npix)), dim3(kBlock), 0, st, L0, c->C, c->d_trace_rows, ti0, nrows, W, rays + 0);
        size_t live_cap = npix;  // upper bound on live rays at this level
        std::vector<size_t> bound(levels);
        std::vector<char> forked(levels, 0);
        for (int k = 0; k < levels; k++) {
            LevelDev Lk = (k == 0) ? L0 : c->L[k];
            LevelDev Ln = (k + 1 < levels) ? c->L[k + 1] : Lk;
            bound[k] = live_cap;
            int spawn = (k < c->params.max_depth && k + 1 < levels) ? 1 : 0;
            unsigned g = grid_for(live_cap);
            if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used], st)); }
            if (count) hipLaunchKernelGGL(k_closest<true>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, rays + k, c->d_tests);
            else if (exact_only) hipLaunchKernelGGL(k_closest<false>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, rays + k, c->d_tests);
            else if (k > 0) {
                const dim3 gl(std::max(g, grid_lanes_thin(live_cap, c->lanes16, c->lanes4)));
                if (exec) hipLaunchKernelGGL(k_closest_lanes<true>, gl, dim3(kBlock), 0, st, c->lanes_scene(), Lk, rays + k, c->lanes16, c->lanes4);
                else hipLaunchKernelGGL(k_closest_lanes<false>, gl, dim3(kBlock), 0, st, c->lanes_scene(), Lk, rays + k, c->lanes16, c->lanes4);
            }
            else if (c->cam.off) {
                if (exec) hipLaunchKernelGGL(k_primary_cells<true>, dim3(g), dim3(kBlock), 0, st, c->S, c->cam, Lk, rays + k);
                else hipLaunchKernelGGL(k_primary_cells<false>, dim3(g), dim3(kBlock), 0, st, c->S, c->cam, Lk, rays + k);
            }
            else if (live_cap < kPrefetchRays) {
                if (exec) hipLaunchKernelGGL((k_closest_f<true, true>), dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k);
                else hipLaunchKernelGGL((k_closest_f<true, false>), dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k);
            } else {
                if (exec) hipLaunchKernelGGL((k_closest_f<false, true>), dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k);
                else hipLaunchKernelGGL((k_closest_f<false, false>), dim3(grid_for_isect(live_cap)), dim3(kBlock), 0, st, c->S, Lk, rays + k);
            }
            if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used + 1], st)); kev_used += 2; c->kev_kind.push_back((k > 0 && !count && !exact_only) ? 2 : 0); }
            // generation 0 runs alone and every workgroup allocates: 1024-thread workgroups quarter its same-address
            // atomics; later generations share the chip with the shadow passes, where a 1024-thread workgroup waits
            // for a whole CU to drain
            if (k == 0) hipLaunchKernelGGL(k_setup<kSetupBlock>, dim3((unsigned)((live_cap + kSetupBlock - 1) / kSetupBlock)), dim3(kSetupBlock), 0, st, c->S, Lk, Ln, rays + k, nodes + k, spawn, c->d_overflow);
            else hipLaunchKernelGGL(k_setup<kBlock>, dim3(g), dim3(kBlock), 0, st, c->S, Lk, Ln, rays + k, nodes + k, spawn, c->d_overflow);
            if (c->S.nl > 0 && (count || exact_only)) {
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used], st)); }
                if (count) hipLaunchKernelGGL(k_shadow<true>, dim3(g, c->S.nl), dim3(kBlock), 0, st, c->S, Lk, nodes + k, c->d_tests + 4);
                else hipLaunchKernelGGL(k_shadow<false>, dim3(g, c->S.nl), dim3(kBlock), 0, st, c->S, Lk, nodes + k, c->d_tests + 4);
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used + 1], st)); kev_used += 2; c->kev_kind.push_back(1); }
            } else if (c->S.nl > 0) {
                // materialise and sort the shadow rays, then one any-hit pass -- on a side stream
                const bool on_main = k >= c->main_from;
                rtx_context::ShadowLane& sl = c->side[on_main ? (int)rtx_context::kSide : rtx_context::side_of(k)];
                hipStream_t ss = on_main ? st : sl.stream;
                if (!on_main) {
                    HIPCHK(hipEventRecord(c->ev_fork[k], st));
                    HIPCHK(hipStreamWaitEvent(ss, c->ev_fork[k], 0));
                    forked[k] = true;
                }
                int* sc = c->d_scnt + ((size_t)nchunks * levels + k);
                if (!(c->use_cells && c->shadow_direct)) {
                    HIPCHK(hipMemsetAsync(sl.hist, 0, sizeof(int) * (size_t)c->nbins, ss));
                    hipLaunchKernelGGL(k_shadow_bin, dim3(g), dim3(kBlock), 0, ss, c->S, Lk, nodes + k, c->binG, sl.hist, sl.rank);
                    {
                        const int nb = (c->nbins + 1023) / 1024;
                        hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, ss, sl.hist, sl.offs, c->nbins, sl.bsum, sl.braw, 64);
                        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ss, sl.bsum, nb, sc + 0);
                        hipLaunchKernelGGL(k_scan_add, dim3(nb * (1024 / kBlock)), dim3(kBlock), 0, ss, sl.offs, c->nbins, sl.bsum, sl.braw, 64, sl.q.b);
                    }
                    hipLaunchKernelGGL(k_shadow_scatter, dim3(g), dim3(kBlock), 0, ss, c->S, Lk, nodes + k, c->binG, sl.offs, sl.rank, sl.q);
                }
                unsigned gq = grid_for(live_cap * (size_t)c->S.nl + c->qpad);  // + the alignment padding of the bin groups
                gq = std::max(gq, (unsigned)(grid_for(std::min<size_t>(live_cap * (size_t)c->S.nl + c->qpad, (size_t)c->split4)) * kSplit));
                gq = std::max(gq, (unsigned)(grid_for(std::min<size_t>(live_cap * (size_t)c->S.nl + c->qpad, (size_t)c->split8)) * 2 * kSplit));
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used], ss)); }
                // scenes with affine spheres: a generation that can fill the chip runs the traversal of the plain primitives
                // and the affine spheres as two kernels (registers: 65 / 93 VGPRs instead of 104 together), a thin one as one
                const bool fused = c->S.naff > 0 && live_cap * (size_t)c->S.nl <= (size_t)c->split4;
                // generation 0's shadow rays (hits of primary rays) are coherent enough for the one-survivor-at-a-time walk;
                // later generations take the pooled stages (rtx_shadow.hip.h, shadow_plain_pool)
                const bool pool = k >= c->pool_from;
#define RTX_SHADOW_ARGS dim3(gq), dim3(kBlock), 0, ss, c->S, sl.q, sc + 0, Lk.lit, Lk.cap, c->split4, c->split8
                if (c->use_cells && c->shadow_direct) {
                    // the whole shadow stage of this generation in ONE launch: every (hit, light) pair walks the tables
                    // (rtx_cells.hip.h) straight off the hit records -- no queue, no sort
                    const dim3 gd(g, (unsigned)c->S.nl);
                    if (exec) hipLaunchKernelGGL(k_shadow_direct<true>, gd, dim3(kBlock), 0, ss, c->S, c->cells, Lk, nodes + k);
                    else hipLaunchKernelGGL(k_shadow_direct<false>, gd, dim3(kBlock), 0, ss, c->S, c->cells, Lk, nodes + k);
                } else if (c->use_cells) {
                    // the sorted queue through the tables, one workgroup per 256 queue slots at every size; without an affine
                    // table the affine spheres behind it for the rays still lit
                    const dim3 gc(grid_for(live_cap * (size_t)c->S.nl + c->qpad));
                    if (exec) hipLaunchKernelGGL(k_shadow_cells<true>, gc, dim3(kBlock), 0, ss, c->S, c->cells, sl.q, sc + 0, Lk.lit, Lk.cap);
                    else hipLaunchKernelGGL(k_shadow_cells<false>, gc, dim3(kBlock), 0, ss, c->S, c->cells, sl.q, sc + 0, Lk.lit, Lk.cap);
                    if (c->S.naff > 0 && !c->cells.aoff) {
                        if (exec) hipLaunchKernelGGL((k_shadow_affine<true, false, false>), RTX_SHADOW_ARGS);
                        else hipLaunchKernelGGL((k_shadow_affine<false, false, false>), RTX_SHADOW_ARGS);
                    }
                } else if (fused) {
                    if (exec) { if (pool) hipLaunchKernelGGL((k_shadow_affine<true, true, true>), RTX_SHADOW_ARGS); else hipLaunchKernelGGL((k_shadow_affine<true, true, false>), RTX_SHADOW_ARGS); }
                    else { if (pool) hipLaunchKernelGGL((k_shadow_affine<false, true, true>), RTX_SHADOW_ARGS); else hipLaunchKernelGGL((k_shadow_affine<false, true, false>), RTX_SHADOW_ARGS); }
                } else {
                    if (exec) { if (pool) hipLaunchKernelGGL((k_shadow_pass<true, true>), RTX_SHADOW_ARGS); else hipLaunchKernelGGL((k_shadow_pass<true, false>), RTX_SHADOW_ARGS); }
                    else { if (pool) hipLaunchKernelGGL((k_shadow_pass<false, true>), RTX_SHADOW_ARGS); else hipLaunchKernelGGL((k_shadow_pass<false, false>), RTX_SHADOW_ARGS); }
                    if (c->S.naff > 0) {
                        if (exec) hipLaunchKernelGGL((k_shadow_affine<true, false, false>), RTX_SHADOW_ARGS);
                        else hipLaunchKernelGGL((k_shadow_affine<false, false, false>), RTX_SHADOW_ARGS);
                    }
                }
#undef RTX_SHADOW_ARGS
                if (timek) { HIPCHK(hipEventRecord(c->kev[kev_used + 1], ss)); kev_used += 2; c->kev_kind.push_back(1); }
                if (!on_main) HIPCHK(hipEventRecord(c->ev_pass[k], ss));
            }
            if (c->fanout >= 2) live_cap *= 2;
            if (k + 1 < levels) live_cap = std::min(live_cap, (size_t)c->L[k + 1].cap);  // calibrated capacities (fan-out 2)
        }
        for (int k = levels - 1; k >= 0; k--) {
            LevelDev Lk = (k == 0) ? L0 : c->L[k];
            LevelDev Ln = (k + 1 < levels) ? c->L[k + 1] : Lk;
            if (forked[k]) HIPCHK(hipStreamWaitEvent(st, c->ev_pass[k], 0));  // generation k's lit flags
            hipLaunchKernelGGL(k_unwind, dim3(grid_for(bound[k])), dim3(kBlock), 0, st, c->S, Lk, Ln, nodes + k);
        }
        hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(64), 0, st, c->d_cnt, c->d_totals, levels);
    }
    if (nown > 0 && !(c->params.flags & RTX_F_NO_HALO)) {
        unsigned char* dst = d_out ? d_out : c->d_rgb8;
        if ((W & 3) == 0 && ((uintptr_t)dst & 3) == 0 && ((uintptr_t)c->d_fimg & 15) == 0)
            hipLaunchKernelGGL(k_aa_quantise4, dim3(grid_for((size_t)nown * (W / 4))), dim3(kBlock), 0, st, c->d_fimg, c->d_row2ti,
                               c->d_own_rows, nown, W, c->height, dst);
        else
            hipLaunchKernelGGL(k_aa_quantise, dim3(grid_for((size_t)nown * W)), dim3(kBlock), 0, st, c->d_fimg, c->d_row2ti,
                               c->d_own_rows, nown, W, c->height, dst);
    }
    HIPCHK(hipGetLastError());
    if (c->ev1) HIPCHK(hipEventRecord(c->ev1, st));
    c->last_stream = st;
    c->last_dst = d_out ? d_out : c->d_rgb8;
    c->rendered = true;
    c->nchunks = nchunks;
    return RTX_OK;
}

static int context_stats_impl(rtx_context* c, rtx_stats* s)
{
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    std::memset(s, 0, sizeof *s);
    // frame totals, test counters, packet statistics and executed-work counters are ONE block (context_create_impl): one copy
    std::vector<unsigned long long> t((size_t)2 * c->levels + 16 + 3 * EX_KINDS);
    HIPCHK(hipMemcpy(t.data(), c->d_totals, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long* const h_tests = t.data() + (c->d_tests - c->d_totals);
    const unsigned long long* const h_pstat = t.data() + (c->d_pstat - c->d_totals);
    const unsigned long long* const h_exec = t.data() + (c->S.exec - c->d_totals);
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
    {
        int ov = 0;
        HIPCHK(hipMemcpy(&ov, c->d_overflow, sizeof ov, hipMemcpyDeviceToHost));
        s->queue_overflow = (uint64_t)(ov < 0 ? 0 : ov);
    }
    s->chunks = c->nchunks;
    s->levels = c->levels;
    s->shadow_form = c->use_cells ? 1 : 0;
    float ms = 0;
    if (c->ev0 && c->ev1) HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));  // arena mode keeps no events: 0
    s->ms_total = ms;
    for (int k = 0; k < 8; k++) s->packet_stats[k] = h_pstat[k];
    for (int k = 0; k < 4; k++) { s->tests_closest_kind[k] = h_tests[k]; s->tests_shadow_kind[k] = h_tests[4 + k]; }
    {
        static_assert(EX_KINDS <= RTX_EXEC_KINDS, "rtx_stats.exec_counts holds every kind");
        for (int kc = 0; kc < 3; kc++)
            for (int k = 0; k < EX_KINDS; k++) s->exec_counts[kc][k] = h_exec[kc * EX_KINDS + k];
    }
    for (size_t e = 0; e < c->kev_kind.size(); e++) {
        float m = 0;
        HIPCHK(hipEventElapsedTime(&m, c->kev[2 * e], c->kev[2 * e + 1]));
        if (c->kev_kind[e] == 0) { s->ms_closest += m; s->launches_closest++; }
        else if (c->kev_kind[e] == 2) { s->ms_closest_lanes += m; s->launches_closest_lanes++; }
        else { s->ms_shadow += m; s->launches_shadow++; }
    }
    return RTX_OK;
}

// The one-shot entry points keep their LAST context: a drop-in host calls rtx_render_rgb8 once per frame with the same
// scene and size, and creating a context (scene pack + upload, the affine cell table, queue allocation, the calibration
// pass) is 45 of the 55 ms such a call takes at 4K.  The context is reused only when a 128-bit hash of EVERYTHING the call
// reads -- every record of every array of the scene view, its scalars, rtx_params, the device -- equals the cached one
// (hashing C4's 1.2 MB of records: 0.2 ms), so a caller that edits a record between calls gets a new context.  One entry,
// held (with its HBM) until a different call replaces it, rtx_release_cached_context() drops it, or the process ends;
// RTX_F_NO_CACHE bypasses it.  Calls through the cache are serialised.
struct Hash128 {
    uint64_t a = 0x9e3779b97f4a7c15ull, b = 0xc2b2ae3d27d4eb4full;
    void word(uint64_t w)
    {
        a = (a ^ w) * 0x100000001b3ull; a ^= a >> 29;
        b = (b + w) * 0xff51afd7ed558ccdull; b ^= b >> 32;
    }
    void bytes(const void* p, size_t n)
    {
        const unsigned char* q = static_cast<const unsigned char*>(p);
        word(n);
        if (n >= 256) {
            // four independent lanes over 32-byte blocks (the two multiply chains of word() are latency-bound: C4's 1.2 MB of
            // records took 0.3 ms of a 3.9 ms call), folded into the state in lane order
            Hash128 l[4];
            for (int k = 0; k < 4; k++) { l[k].a ^= (uint64_t)(k + 1) * 0x9e3779b97f4a7c15ull; l[k].b += (uint64_t)(k + 1) * 0xc2b2ae3d27d4eb4full; }
            for (; n >= 32; n -= 32, q += 32) {
                uint64_t w[4];
                std::memcpy(w, q, 32);
                l[0].word(w[0]); l[1].word(w[1]); l[2].word(w[2]); l[3].word(w[3]);
            }
            for (int k = 0; k < 4; k++) { word(l[k].a); word(l[k].b); }
        }
        for (; n >= 8; n -= 8, q += 8) { uint64_t w; std::memcpy(&w, q, 8); word(w); }
        if (n) { uint64_t w = 0; std::memcpy(&w, q, n); word(w); }
    }
};

static void hash_call(const rtx_scene* sc, const rtx_params* p, int dev, uint64_t key[2])
{
    Hash128 h;
    h.bytes(sc->spheres, sizeof(rtx_sphere) * (size_t)sc->nspheres);
    h.bytes(sc->triangles, sizeof(rtx_triangle) * (size_t)sc->ntriangles);
    h.bytes(sc->planes, sc->planes ? sizeof(rtx_plane) * (size_t)sc->nplanes : 0);
    h.bytes(sc->lights, sizeof(rtx_light) * (size_t)sc->nlights);
    h.bytes(sc->polygons, sizeof(rtx_polygon) * (size_t)sc->npolygons);
    h.bytes(sc->polygon_vertices, sizeof(rtx_vec3) * (size_t)sc->npolygon_vertices);
    const rtx_vec3 v[5] = { sc->camera, sc->vrp, sc->vpn, sc->up, sc->ambient };
    h.bytes(v, sizeof v);
    const int32_t n[8] = { sc->nspheres, sc->ntriangles, sc->nplanes, sc->nlights, sc->npolygons, sc->npolygon_vertices, sc->npolygons_dropped, dev };
    h.bytes(n, sizeof n);
    const int64_t q[9] = { p->width, p->height, p->max_depth, p->nranks, p->rank, p->band_rows, (int64_t)p->flags, (int64_t)p->workspace_bytes, 0 };
    h.bytes(q, sizeof q);
    for (const char* name : {"RTX_SHADOW_POOL_FROM", "RTX_SPLIT4", "RTX_SPLIT8", "RTX_TWO_LEVEL", "RTX_BIN_G", "RTX_CELLS", "RTX_CELL_G", "RTX_SHADOW_DIRECT", "RTX_SHADOW_FORM", "RTX_LANES16", "RTX_LANES4", "RTX_CELL_TIGHT", "RTX_CELL_NOFILTER"})  // the A/B aids context creation reads
        if (const char* v = ab_knob(name)) { h.bytes(name, std::strlen(name)); h.bytes(v, std::strlen(v)); }
    key[0] = h.a; key[1] = h.b;
}

struct OneShotCache { std::mutex m; rtx_context* c = nullptr; uint64_t key[2] = {0, 0}; };
static OneShotCache g_one_shot;

static int render_one(const rtx_scene* scene, const rtx_params* params, uint8_t* rgb, float* f32, rtx_stats* stats)
{
    if (params->flags & RTX_F_NO_CACHE) {
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
    int dev = params->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;  // (no device: rtx_context_create reports it)
    uint64_t key[2];
    hash_call(scene, params, dev, key);
    std::lock_guard<std::mutex> lk(g_one_shot.m);
    OneShotCache& C = g_one_shot;
    if (!C.c || C.key[0] != key[0] || C.key[1] != key[1]) {
        if (C.c) { rtx_context_destroy(C.c); C.c = nullptr; }
        int rc = rtx_context_create(scene, params, &C.c);
        if (rc) { C.c = nullptr; return rc; }
        C.key[0] = key[0]; C.key[1] = key[1];
    }
    int rc = rtx_context_render(C.c, nullptr, nullptr);
    if (!rc && rgb) rc = rtx_context_read_rgb8(C.c, rgb);
    if (!rc && f32) rc = rtx_context_read_f32(C.c, f32);
    if (!rc && stats) rc = rtx_context_stats(C.c, stats);
    if (rc) { rtx_context_destroy(C.c); C.c = nullptr; }  // never keep a context a call failed on
    return rc;
}

// ------------------------------------------------------------------------------------------------
// In-process multi-GPU (rtx_render_rgb8 with ngpus > 1; the CLI's --gpus=N): scanline bands round-robin over the
// devices, ONE context per device, every band rendered straight into the buffer it is sent from, the 8-bit bands
// gathered into device 0 with RCCL (grouped ncclSend / ncclRecv: each peer has its own xGMI link into device 0, so
// the fan-in is point to point, not a ring), one device-to-host copy of the gathered bands, rows put in place.
// RCCL is loaded on first use (dlopen: the library does not depend on it at load time, and a process that already
// holds an RCCL -- PyTorch's -- shares it); failing that: RTX_E_RCCL.
// RTX_F_MULTI_ON_ONE_DEVICE (test aid for one-GPU machines): the ranks all run on one device and the exchange is
// device-to-device copies instead of RCCL (which refuses two ranks on one device); everything else is the same code.
// ------------------------------------------------------------------------------------------------
struct Rccl {
    void* h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

static Rccl& rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.h) break;
        }
        if (!R.h) return;
        R.CommInitAll = (decltype(R.CommInitAll))dlsym(R.h, "ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.h, "ncclCommDestroy");
        R.GroupStart = (decltype(R.GroupStart))dlsym(R.h, "ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))dlsym(R.h, "ncclGroupEnd");
        R.Send = (decltype(R.Send))dlsym(R.h, "ncclSend");
        R.Recv = (decltype(R.Recv))dlsym(R.h, "ncclRecv");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.h, "ncclGetErrorString");
        R.ok = R.CommInitAll && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv && R.GetErrorString;
    });
    return R;
}

#define NCCLCHK(expr)                                                                         \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            g_last_hip_error = std::string(#expr) + ": " + rccl().GetErrorString(r_);         \
            rc = RTX_E_RCCL;                                                                  \
            goto done;                                                                        \
        }                                                                                     \
    } while (0)
#define HIPCHK_GOTO(expr)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_last_hip_error = std::string(#expr) + ": " + hipGetErrorString(e_);             \
            rc = (e_ == hipErrorOutOfMemory) ? RTX_E_OOM : RTX_E_HIP;                         \
            goto done;                                                                        \
        }                                                                                     \
    } while (0)

// The multi-device form keeps what it built -- one context and one stream per device, the RCCL communicators, the
// gather buffer on device 0 and its pinned host staging -- between calls that read the same bytes (the one-shot cache's
// 128-bit hash, plus the device count): ncclCommInitAll and N context creations are hundreds of ms around a frame of a
// millisecond.  rtx_release_cached_context() frees it, RTX_F_NO_CACHE never keeps it, a failed call drops it.
struct MultiCache {
    std::mutex m;
    bool valid = false;
    uint64_t key[2] = {0, 0};
    int G = 0;
    bool on_one = false;
    std::vector<rtx_context*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<int> dev;
    std::vector<std::vector<int32_t>> rows;
    std::vector<ncclComm_t> comms;
    bool have_comms = false;
    unsigned char* d_gather = nullptr;   // on device 0: G slots of `slot` bytes, slot g = rank g's packed rows
    uint8_t* h_gather = nullptr;         // pinned
    size_t slot = 0;
    void release()
    {
        if (have_comms)
            for (ncclComm_t cm : comms)
                if (cm) (void)rccl().CommDestroy(cm);
        for (size_t g = 0; g < ctx.size(); g++) {
            if (ctx[g]) rtx_context_destroy(ctx[g]);
            if (g < stream.size() && stream[g]) { (void)hipSetDevice(dev[g]); (void)hipStreamDestroy(stream[g]); }
        }
        if (d_gather) { (void)hipSetDevice(dev.empty() ? 0 : dev[0]); (void)hipFree(d_gather); }
        if (h_gather) (void)hipHostFree(h_gather);
        ctx.clear(); stream.clear(); dev.clear(); rows.clear(); comms.clear();
        have_comms = false; d_gather = nullptr; h_gather = nullptr; slot = 0; valid = false; G = 0;
    }
};
static MultiCache g_multi;

static int multi_create(MultiCache& M, const rtx_scene* scene, const rtx_params* params, int G, bool on_one)
{
    const size_t rowb = (size_t)params->width * 3;
    const int band = params->band_rows > 0 ? params->band_rows : 16;
    int rc = RTX_OK;
    M.G = G; M.on_one = on_one;
    M.ctx.assign(G, nullptr); M.stream.assign(G, nullptr); M.dev.assign(G, 0); M.rows.assign(G, {}); M.comms.assign(G, nullptr);
    std::vector<int> rcs(G, 0);
    std::vector<std::string> errs(G);
    size_t slot = 0;
    for (int g = 0; g < G; g++) {
        M.dev[g] = on_one ? (params->device >= 0 ? params->device : 0) : g;
        int n = rtx_band_rows(params->height, G, g, band, nullptr, 0);
        if (n < 0) return n;
        M.rows[g].resize(n);
        if (n) rtx_band_rows(params->height, G, g, band, M.rows[g].data(), n);
        slot = std::max(slot, (size_t)n * rowb);
    }
    M.slot = (slot + 255) & ~(size_t)255;
    {   // one context per device, created concurrently (scene packing and upload are per device)
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++)
            th.emplace_back([&, g]() {
                try {
                    if (M.rows[g].empty()) return;
                    rtx_params p = *params;
                    p.nranks = G; p.rank = g; p.device = M.dev[g]; p.band_rows = band;
                    p.flags &= ~(uint32_t)(RTX_F_MULTI_ON_ONE_DEVICE | RTX_F_NO_CACHE);
                    rcs[g] = rtx_context_create(scene, &p, &M.ctx[g]);
                    if (!rcs[g] && hipStreamCreateWithFlags(&M.stream[g], hipStreamNonBlocking) != hipSuccess) rcs[g] = RTX_E_HIP;
                } catch (const std::bad_alloc&) { rcs[g] = RTX_E_OOM; }
                catch (...) { rcs[g] = RTX_E_HIP; }
                if (rcs[g]) errs[g] = g_last_hip_error;  // the worker's own thread-local text
            });
        for (auto& t : th) t.join();
    }
    for (int g = 0; g < G; g++)
        if (rcs[g]) { g_last_hip_error = "device " + std::to_string(M.dev[g]) + ": " + errs[g]; return rcs[g]; }
    HIPCHK(hipSetDevice(M.dev[0]));
    HIPCHK(hipMalloc((void**)&M.d_gather, M.slot * (size_t)G));
    HIPCHK(hipHostMalloc((void**)&M.h_gather, M.slot * (size_t)G, hipHostMallocDefault));
    if (!M.stream[0]) HIPCHK(hipStreamCreateWithFlags(&M.stream[0], hipStreamNonBlocking));
    if (!on_one) {
        if (!rccl().ok) { g_last_hip_error = "librccl.so could not be loaded (dlopen)"; return RTX_E_RCCL; }
        ncclResult_t r = rccl().CommInitAll(M.comms.data(), G, M.dev.data());
        if (r != ncclSuccess) { g_last_hip_error = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r); return RTX_E_RCCL; }
        M.have_comms = true;
    }
    (void)rc;
    return RTX_OK;
}

static int multi_frame(MultiCache& M, const rtx_params* params, uint8_t* out, rtx_stats* stats)
{
    const int G = M.G;
    const size_t rowb = (size_t)params->width * 3;
    int rc = RTX_OK;
    bool in_group = false;
    // every rank renders into the buffer its band is sent from: rank 0 straight into its slot of the gather buffer
    for (int g = 0; g < G; g++) {
        if (!M.ctx[g]) continue;
        int r = rtx_context_render(M.ctx[g], M.stream[g], g == 0 ? M.d_gather : nullptr);
        if (r) { rc = r; goto done; }
    }
    if (!M.on_one) {
        // RCCL behind the C-ABI: "parity unpinned" until a machine with >= 2 GPUs has run it (every box so far had one)
        NCCLCHK(rccl().GroupStart());
        in_group = true;
        for (int g = 1; g < G; g++) {
            if (!M.ctx[g]) continue;
            const size_t bytes = M.rows[g].size() * rowb;
            HIPCHK_GOTO(hipSetDevice(M.dev[g]));
            NCCLCHK(rccl().Send(M.ctx[g]->d_rgb8, bytes, ncclUint8, 0, M.comms[g], M.stream[g]));
            HIPCHK_GOTO(hipSetDevice(M.dev[0]));
            NCCLCHK(rccl().Recv(M.d_gather + (size_t)g * M.slot, bytes, ncclUint8, g, M.comms[0], M.stream[0]));
        }
        in_group = false;
        NCCLCHK(rccl().GroupEnd());
    } else {
        for (int g = 1; g < G; g++) {
            if (!M.ctx[g]) continue;
            HIPCHK_GOTO(hipSetDevice(M.dev[g]));
            HIPCHK_GOTO(hipMemcpyAsync(M.d_gather + (size_t)g * M.slot, M.ctx[g]->d_rgb8, M.rows[g].size() * rowb, hipMemcpyDeviceToDevice, M.stream[g]));
        }
    }
    for (int g = G - 1; g >= 1; g--) {
        if (!M.stream[g]) continue;
        HIPCHK_GOTO(hipSetDevice(M.dev[g]));
        HIPCHK_GOTO(hipStreamSynchronize(M.stream[g]));
    }
    // ONE device-to-host copy of the gathered bands (pinned staging), then the rows go to their places in the caller's frame
    HIPCHK_GOTO(hipSetDevice(M.dev[0]));
    HIPCHK_GOTO(hipMemcpyAsync(M.h_gather, M.d_gather, M.slot * (size_t)G, hipMemcpyDeviceToHost, M.stream[0]));
    HIPCHK_GOTO(hipStreamSynchronize(M.stream[0]));
    for (int g = 0; g < G; g++)
        for (size_t r = 0; r < M.rows[g].size(); r++)
            std::memcpy(out + (size_t)M.rows[g][r] * rowb, M.h_gather + (size_t)g * M.slot + r * rowb, rowb);
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (int g = 0; g < G; g++) {
            if (!M.ctx[g]) continue;
            rtx_stats s;
            int r = rtx_context_stats(M.ctx[g], &s);
            if (r) { rc = r; goto done; }
            stats->rays_primary += s.rays_primary; stats->rays_secondary += s.rays_secondary;
            stats->rays_shadow += s.rays_shadow; stats->hits += s.hits;
            for (int k = 0; k < RTX_MAX_LEVELS; k++) stats->rays_level[k] += s.rays_level[k];
            stats->tests_closest += s.tests_closest; stats->tests_shadow_max += s.tests_shadow_max;
            stats->singular_matrices = s.singular_matrices; stats->queue_overflow += s.queue_overflow;
            stats->chunks += s.chunks; stats->levels = std::max(stats->levels, s.levels);
            stats->ms_total = std::max(stats->ms_total, s.ms_total);
        }
    }
done:
    if (in_group) {  // a Send / Recv / hipSetDevice failed inside the group: close it before anything else touches RCCL
        const std::string keep = g_last_hip_error;
        (void)rccl().GroupEnd();
        g_last_hip_error = keep;
    }
    return rc;
}

static int render_multi(const rtx_scene* scene, const rtx_params* params, int ngpus, uint8_t* out, rtx_stats* stats)
{
    const int ndev = rtx_device_count();
    if (ndev <= 0) return RTX_E_NO_DEVICE;
    const bool on_one = (params->flags & RTX_F_MULTI_ON_ONE_DEVICE) != 0;
    if (!on_one && ngpus > ndev) ngpus = ndev;
    if (ngpus <= 1) return render_one(scene, params, out, nullptr, stats);
    if (ngpus > 64) ngpus = 64;
    uint64_t key[2];
    hash_call(scene, params, -1 - ngpus, key);
    std::lock_guard<std::mutex> lk(g_multi.m);
    MultiCache& M = g_multi;
    int rc = RTX_OK;
    if (!M.valid || M.key[0] != key[0] || M.key[1] != key[1] || M.G != ngpus || M.on_one != on_one) {
        M.release();
        rc = multi_create(M, scene, params, ngpus, on_one);
        if (rc) { const std::string keep = g_last_hip_error; M.release(); g_last_hip_error = keep; return rc; }
        M.key[0] = key[0]; M.key[1] = key[1];
        M.valid = true;
    }
    rc = multi_frame(M, params, out, stats);
    if (rc || (params->flags & RTX_F_NO_CACHE)) {
        const std::string keep = g_last_hip_error;
        M.release();
        if (rc) g_last_hip_error = keep;
    }
    return rc;
}

// No entry point lets a C++ exception cross the C-ABI (include/rtx.h): host-side allocations can throw.
#define RTX_TRY try {
#define RTX_CATCH                                                                             \
    } catch (const std::bad_alloc&) { return RTX_E_OOM; }                                     \
    catch (...) { g_last_hip_error = "unexpected C++ exception"; return RTX_E_HIP; }

// ------------------------------------------------------------------------------------------------
// rtx_render_rgb8_noalloc: the one-shot render with every device buffer in a static array
// ------------------------------------------------------------------------------------------------
constexpr size_t kScratchBytes = (size_t)RTX_NOALLOC_SCRATCH_MIB << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// `scratch` above is a host-side stand-in in host code: its device address is asked for once per device
// (each GPU holds its own copy of the array at its own address) and only that pointer is ever used.
constexpr int kMaxScratchDevices = 64;
struct ScratchDev { void* ptr = nullptr; bool ready = false; std::mutex busy; };
static ScratchDev g_scratch[kMaxScratchDevices];
static std::mutex g_scratch_init;

static int scratch_for_device(int dev, ScratchDev** out)
{
    if (dev < 0 || dev >= kMaxScratchDevices) return RTX_E_INVALID_ARG;
    std::lock_guard<std::mutex> lk(g_scratch_init);
    ScratchDev& sd = g_scratch[dev];
    if (!sd.ready) {
        HIPCHK(hipSetDevice(dev));
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p || ((uintptr_t)p & 255) != 0) { g_last_hip_error = "hipGetSymbolAddress(scratch): null or misaligned address"; return RTX_E_HIP; }
        sd.ptr = p;
        sd.ready = true;
    }
    *out = &sd;
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
    rtx_context* c = nullptr;
    RTX_TRY
    c = new (std::nothrow) rtx_context();
    if (!c) return RTX_E_OOM;
    try { rc = context_create_impl(scene, params, c); }
    catch (...) { rtx_context_destroy(c); throw; }
    if (rc) { rtx_context_destroy(c); return rc; }
    *out = c;
    return RTX_OK;
    RTX_CATCH
}

void rtx_context_destroy(rtx_context* c)
{
    if (!c) return;
    if (c->rendered) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->last_stream);
        for (rtx_context::ShadowLane& sl : c->side)
            if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    }
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (!c->allocs.empty() || c->ev0 || c->ev1 || !c->kev.empty() || !c->ev_fork.empty()) {
        (void)hipSetDevice(c->device);
        for (void* p : c->allocs) (void)hipFree(p);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        for (auto e : c->kev) if (e) (void)hipEventDestroy(e);
        for (auto e : c->ev_fork) if (e) (void)hipEventDestroy(e);
        for (auto e : c->ev_pass) if (e) (void)hipEventDestroy(e);
        for (rtx_context::ShadowLane& sl : c->side) {
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
    RTX_TRY
    return context_render_impl(c, (hipStream_t)hip_stream, d_rgb8_out);
    RTX_CATCH
}

int rtx_context_read_rgb8(rtx_context* c, uint8_t* host)
{
    if (!c || !host || !c->rendered) return RTX_E_INVALID_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->last_stream));
    size_t n = c->own_rows.size() * (size_t)c->width * 3;
    // the last frame's rows: in the context's own framebuffer, or in the device buffer the caller gave that render.
    // Through a pinned staging buffer kept by the context: a copy into the caller's pageable memory runs at a third of the
    // link's rate (3.2 ms for a 4K frame); pinned it takes 0.5 ms, and four host threads move it on in under a millisecond.
    if (!n) return RTX_OK;
    if (!c->arena && !c->h_stage && n >= ((size_t)1 << 20)) {
        void* q = nullptr;
        if (hipHostMalloc(&q, n, hipHostMallocDefault) == hipSuccess) c->h_stage = static_cast<uint8_t*>(q);
        else (void)hipGetLastError();  // no pinned memory to be had: the plain copy below
    }
    if (!c->h_stage) {
        HIPCHK(hipMemcpy(host, c->last_dst, n, hipMemcpyDeviceToHost));
        return RTX_OK;
    }
    constexpr int kParts = 4;
    const size_t part = ((n + kParts - 1) / kParts + 63) & ~(size_t)63;
    // part p is copied to the host while part p - 1 is moved on by a thread of its own
    std::thread th[kParts];
    int started = 0, rc = RTX_OK;
    for (int p = 0; p < kParts && !rc; p++) {
        const size_t o0 = std::min(n, (size_t)p * part), o1 = std::min(n, o0 + part);
        if (o1 == o0) break;
        if (hipMemcpyAsync(c->h_stage + o0, c->last_dst + o0, o1 - o0, hipMemcpyDeviceToHost, c->last_stream) != hipSuccess ||
            hipStreamSynchronize(c->last_stream) != hipSuccess) {
            g_last_hip_error = std::string("rtx_context_read_rgb8: ") + hipGetErrorString(hipGetLastError());
            rc = RTX_E_HIP;
            break;
        }
        try { th[started] = std::thread([=]() { std::memcpy(host + o0, c->h_stage + o0, o1 - o0); }); started++; }
        catch (...) { std::memcpy(host + o0, c->h_stage + o0, o1 - o0); }  // no thread to be had: move it on here
    }
    for (int p = 0; p < started; p++) th[p].join();
    return rc;
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
    RTX_TRY
    return context_stats_impl(c, s);
    RTX_CATCH
}

int rtx_render_f32(const rtx_scene* scene, const rtx_params* params, float* out, rtx_stats* stats)
{
    if (!out) return RTX_E_INVALID_ARG;
    int rc = validate(scene, params);
    if (rc) return rc;
    RTX_TRY
    return render_one(scene, params, nullptr, out, stats);
    RTX_CATCH
}

int rtx_render_rgb8_noalloc(const rtx_scene* scene, const rtx_params* params, uint8_t* out, rtx_stats* stats)
{
    if (!out) return RTX_E_INVALID_ARG;
    int rc = validate(scene, params);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return RTX_E_NO_DEVICE;
    int dev = params->device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev) return RTX_E_INVALID_ARG;
    RTX_TRY
    ScratchDev* sd = nullptr;
    if ((rc = scratch_for_device(dev, &sd))) return rc;
    // one call at a time per device: every call on a GPU works in that GPU's one array
    std::lock_guard<std::mutex> lk(sd->busy);
    rtx_context* c = new (std::nothrow) rtx_context();  // host object only
    if (!c) return RTX_E_OOM;
    c->arena = static_cast<char*>(sd->ptr);
    c->arena_cap = kScratchBytes;
    rtx_params p = *params;
    p.device = dev;
    p.flags &= ~(uint32_t)RTX_F_TIME_KERNELS;  // per-kernel timing needs events: not in this variant
    try {
        rc = context_create_impl(scene, &p, c);
        if (rc == RTX_E_TOO_LARGE)
            g_last_hip_error = "rtx_render_rgb8_noalloc: scene records + float image + one row of ray queues exceed the " +
                               std::to_string(RTX_NOALLOC_SCRATCH_MIB) + " MiB static scratch";
        if (!rc) rc = context_render_impl(c, nullptr, nullptr);
        if (!rc) rc = rtx_context_read_rgb8(c, out);
        if (!rc && stats) rc = context_stats_impl(c, stats);
    } catch (...) { rtx_context_destroy(c); throw; }
    rtx_context_destroy(c);  // waits for the stream; nothing on the device to free
    return rc;
    RTX_CATCH
}

// The direction map of the shadow stage, as the kernels compute it (the same functions, compiled for the host): the cell of
// a direction, and the cone the (light, cell) tables are built with.  Pure host functions, no device needed.
int rtx_light_cell(const float* dir3, int32_t G)
{
    if (!dir3 || G < 1 || G > 4096) return RTX_E_INVALID_ARG;
    return light_cell(f3(dir3[0], dir3[1], dir3[2]), G);
}

int rtx_cell_cone(int32_t cell, int32_t G, float* out7)
{
    if (!out7 || G < 1 || G > 4096 || cell < 0 || (int64_t)cell >= (int64_t)G * G) return RTX_E_INVALID_ARG;
    const CellCone K = cell_cone((unsigned)cell, G);
    out7[0] = K.w.x; out7[1] = K.w.y; out7[2] = K.w.z; out7[3] = K.sT; out7[4] = K.cLo;
    out7[5] = (float)demorton12((unsigned)cell); out7[6] = (float)demorton12((unsigned)cell >> 1);
    return RTX_OK;
}

// The per-ray traversal's tests on an enclosing sphere { c, R }, the kernel's own functions compiled for the host: out3 = { cone_slack,
// cone_slack_front, cone_slack_cull(lr) } (sign bit set = the sphere is skipped).  Pure host function.
int rtx_cone_tests(const float* origin3, const float* dir3, const float* sphere4, float lr, float* out3)
{
    if (!origin3 || !dir3 || !sphere4 || !out3) return RTX_E_INVALID_ARG;
    const F3 o = f3(origin3[0], origin3[1], origin3[2]), n = f3(dir3[0], dir3[1], dir3[2]);
    const float4 b = make_float4(sphere4[0], sphere4[1], sphere4[2], sphere4[3]);
    out3[0] = cone_slack(o, n, b);
    out3[1] = cone_slack_front(o, n, b);
    out3[2] = cone_slack_cull(o, n, b, lr);
    return RTX_OK;
}

void rtx_release_cached_context(void)
{
    {
        std::lock_guard<std::mutex> lk(g_one_shot.m);
        if (g_one_shot.c) { rtx_context_destroy(g_one_shot.c); g_one_shot.c = nullptr; }
    }
    std::lock_guard<std::mutex> lk(g_multi.m);
    g_multi.release();
}

int rtx_render_rgb8(const rtx_scene* scene, const rtx_params* params, int ngpus, uint8_t* out, rtx_stats* stats)
{
    if (!out) return RTX_E_INVALID_ARG;
    int rc = validate(scene, params);
    if (rc) return rc;
    RTX_TRY
    if (params->nranks > 1 || ngpus <= 1) return render_one(scene, params, out, nullptr, stats);
    return render_multi(scene, params, ngpus, out, stats);
    RTX_CATCH
}

}  // extern "C"
