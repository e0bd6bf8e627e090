/*
 * rtx.h -- C-ABI of the MI355X (gfx950) render path.
 *
 * Drop-in boundary for ONE hot path of ChandanY1011/ray-tracing: the per-pixel render loop
 * `render2Dantialiasing` (raytracing.cpp:719-783) together with everything it calls
 * (`raytrace` :371-647, the intersection and vector helpers :116-366).  The reference has no
 * plugin / FFI interface: its only seam is that function, which communicates
 * through mutable globals (:98-113).  This header restates that seam as plain C: the globals
 * become one `rtx_scene` view, the compile-time constants (:12, :112-113) become `rtx_params`,
 * and the image comes back in a caller-owned buffer that the host then writes out as the
 * reference does (:786-797).
 *
 * Everything here is `extern "C"`, plain pointers and sizes.  No C++ or torch types cross it.
 * All entry points return 0 on success or a negative RTX_E_* code; none of them exits, throws
 * or prints (the reference's per-ray prints at :149 / :191 are dropped; counts go to rtx_stats).
 * The library has NO CPU fallback: without a HIP device every compute entry point fails with
 * RTX_E_NO_DEVICE.
 */
#ifndef RTX_H_
#define RTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 0x00030300u

/* ---- data contract: field-for-field the reference's PODs, so a host that keeps the ----------
 * ---- reference's std::vector<sphere> etc. can pass &v[0] with zero conversion.      ---------- */

/* raytracing.cpp:15-20  `struct vector3f` (12 B) */
typedef struct rtx_vec3 { float x, y, z; } rtx_vec3;

/* raytracing.cpp:23-28  `struct xmatrix` (48 B).  Row-vector convention: rows a,b,c are the
 * linear part, d the translation (:171-173). */
typedef struct rtx_xmatrix { rtx_vec3 a, b, c, d; } rtx_xmatrix;

/* raytracing.cpp:31-40  `struct sphere` (120 B) */
typedef struct rtx_sphere {
    rtx_vec3 center;
    int32_t  istransformed;
    float    radius, reflectivity, transparency, refractive_index, specexpo;
    rtx_vec3 color, speccoeff, diffcolor;
    rtx_xmatrix mat;
} rtx_sphere;

/* raytracing.cpp:43-49  `struct plane` (40 B).  Parsed (:886-895) but never traced by the
 * reference: accepted and ignored here too, unless RTX_F_TRACE_PLANES asks for them. */
typedef struct rtx_plane {
    rtx_vec3 normal, color;
    float    d, reflectivity, transparency, refractive_index;
} rtx_plane;

/* raytracing.cpp:52-59  `struct triangle` (88 B) */
typedef struct rtx_triangle {
    rtx_vec3 u, v, w;
    rtx_vec3 color, speccoeff, diffcolor;
    float    reflectivity, transparency, refractive_index, specexpo;
} rtx_triangle;

/* raytracing.cpp:62-68  `struct light` (24 B) */
typedef struct rtx_light { rtx_vec3 point, color; } rtx_light;

/* raytracing.cpp:71-80  `struct polygon` with its std::vector<V> flattened (84 B): the vertices of
 * polygon i are rtx_scene.polygon_vertices[first_vertex .. first_vertex + numvertices).  `normal` and
 * `centerPoint` are what parseinput computes (:929-933, :957-959).  Traced after the triangles
 * (:421-434); never a shadow caster (:569-589 test spheres and triangles only). */
typedef struct rtx_polygon {
    int32_t  numvertices;
    int32_t  first_vertex;
    rtx_vec3 color, speccoeff, diffcolor;
    rtx_vec3 normal;
    rtx_vec3 centerPoint;
    float    reflectivity, transparency, refractive_index, specexpo;
} rtx_polygon;

/* The reference's globals (:98-110) as one read-only view.  The caller owns every pointer;
 * the library copies what it needs during the call that receives the view. */
typedef struct rtx_scene {
    const rtx_sphere*   spheres;    int32_t nspheres;     /* :98,  :106 */
    const rtx_triangle* triangles;  int32_t ntriangles;   /* :100, :108 */
    const rtx_plane*    planes;     int32_t nplanes;      /* :99,  :107  (ignored, F7; traced with RTX_F_TRACE_PLANES) */
    const rtx_light*    lights;     int32_t nlights;      /* :102, :110 */
    rtx_vec3 camera, vrp, vpn, up;                        /* :103  `vcs` */
    rtx_vec3 ambient;                                     /* :104 */
    const rtx_polygon*  polygons;   int32_t npolygons;    /* :101, :109 */
    const rtx_vec3*     polygon_vertices; int32_t npolygon_vertices;
    int32_t  npolygons_dropped;  /* polygon records the loader consumed but did not keep, as parseinput does
                                    when the coplanarity check fails (:945-948, :962) */
} rtx_scene;

/* The reference's compile-time constants as run-time parameters, plus the band decomposition.
 *
 * Rows are grouped into bands of `band_rows` consecutive scanlines; band b belongs to rank
 * (b % nranks).  A rank's framebuffer is its own rows, in increasing row order, packed.
 * nranks = 1 (or 0) means "this call renders every row".  band_rows <= 0 selects contiguous
 * bands, ceil(height / nranks) rows each. */
typedef struct rtx_params {
    int32_t  width;        /* raytracing.cpp:112 (640) */
    int32_t  height;       /* raytracing.cpp:113 (480) */
    int32_t  max_depth;    /* raytracing.cpp:12  MAX_depth (4): rays of generation > max_depth are never spawned */
    int32_t  nranks;       /* number of band owners (processes / GPUs) */
    int32_t  rank;         /* which owner this call renders for */
    int32_t  band_rows;    /* rows per band */
    int32_t  device;       /* HIP device ordinal; -1 = current device */
    uint32_t flags;        /* RTX_F_* */
    uint64_t workspace_bytes; /* cap on per-context ray-queue memory; 0 = default (32 GiB) */
} rtx_params;

#define RTX_F_NONE          0u
#define RTX_F_TIME_KERNELS  1u  /* HIP events around every closest-hit and shadow launch -> rtx_stats.ms_closest/ms_shadow */
#define RTX_F_EXACT_ONLY    4u  /* skip the bounding-sphere filter: run the reference's exact test on every
                                   (ray, primitive) pair.  Must give the identical image; used to check that. */
#define RTX_F_NO_HALO       8u  /* accounting only: trace the owned rows without the one-row halo and skip the
                                   antialias step (framebuffer undefined).  Gives a rank's UNIQUE ray counts. */
#define RTX_F_PACKET_STATS 16u  /* fill rtx_stats.packet_stats (adds atomics to the intersection kernels: slower) */
#define RTX_F_SERIAL       32u  /* enqueue every kernel of the frame on the caller's stream, in order: no side streams, no
                                   fork / join events, one set of shadow-queue buffers.  Same image; kernel durations do
                                   not overlap, which is what a per-kernel roofline needs (bench.py). */
#define RTX_F_TRACE_PLANES 128u /* OPT-IN EXTENSION, off by default: also trace the scene's planes.  The reference parses planes
                                   (:886-895) and never traces them (:382-434), so parity images need this off.
                                   With it, intersectplane (:275-287) runs after the polygons in file order (strict <, as
                                   every kind); the hit takes the triangle's facing normal and shading forms with, a plane
                                   having no such fields (:43-49), diffcolor = color, speccoeff = 0, specexpo = 1; like
                                   polygons, planes cast no shadows.  oracle/rt_oracle.c implements the same (ORC_TRACE_PLANES). */
#define RTX_F_TRI_MOLLER_TRUMBORE 256u /* OPT-IN, NON-PARITY fast mode: triangles are tested with Moller-Trumbore instead of
                                   the reference's area test with its absolute 0.01 tolerance (:329-366).  Edge
                                   pixels differ from the reference's; the mismatch rate is reported by the tests, never
                                   counted as parity. */
#define RTX_F_WORST_CASE_QUEUES 512u /* size every generation's ray queue for fanout^k rays per pixel instead of the counts a
                                   calibration pass measures at creation (A/B aid; same image) */
#define RTX_F_NO_CACHE   1024u /* one-shot entry points (rtx_render_rgb8 / _f32): create and destroy a context inside the call instead of
                                   keeping the last one (see rtx_render_rgb8) */
#define RTX_F_MULTI_ON_ONE_DEVICE 2048u /* test aid for one-GPU machines, rtx_render_rgb8 with ngpus > 1 only: every rank runs on ONE device (params->device,
                                   or device 0) and the bands are gathered with device-to-device copies instead of RCCL, which refuses
                                   two ranks on one device; everything else is the code a multi-GPU machine runs */
#define RTX_F_SHADOW_TABLES  4096u /* the shadow stage has two forms with the same result: (light, direction cell) tables walked straight off the hit
                                   records, and packets over a queue sorted by light and direction.  By default a context renders its own
                                   frame in both forms at creation and keeps the faster (rtx_stats.shadow_form says which); these two flags
                                   skip the measurement and take the tables (where the scene allows them) / the packets. */
#define RTX_F_SHADOW_PACKETS 8192u
#define RTX_F_COUNT_EXEC   64u  /* run the counting instantiations of the PRODUCTION intersection kernels (slower: atomics):
                                   what they actually evaluate, by kind -> rtx_stats.exec_counts.  Same image. */
#define RTX_F_COUNT_TESTS   2u  /* run the counting variants of the two intersection kernels (slower):
                                   exact ray x primitive tests executed, by kind -> rtx_stats.tests_*_kind */

/* Work actually done by the last render of a context (all counts are for this rank's rows,
 * halo rows included). */
#define RTX_MAX_LEVELS 32
#define RTX_EXEC_KINDS 16
typedef struct rtx_stats {
    uint64_t rays_primary;      /* generation-0 rays traced */
    uint64_t rays_secondary;    /* reflection + refraction rays traced */
    uint64_t rays_shadow;       /* shadow rays traced (one per hit per light) */
    uint64_t hits;              /* rays that found a primitive */
    uint64_t rays_level[RTX_MAX_LEVELS]; /* rays per generation */
    uint64_t tests_closest;     /* ray x primitive tests executed by closest-hit (rays * nprims) */
    uint64_t tests_shadow_max;  /* upper bound for shadow rays (no early exit): rays * nprims */
    /* RTX_F_COUNT_TESTS only (0 otherwise): the reference algorithm's ray x primitive tests by kind
     * {sphere, affine sphere, triangle full path, triangle leaving at vd == 0 or t < 0}, counted by the
     * exact kernels that walk every primitive in file order as raytracing.cpp does (shadow rays: per-ray
     * early exit).  The ALGORITHMIC work of the frame; the production kernels cull most of it. */
    uint64_t tests_closest_kind[4];
    uint64_t tests_shadow_kind[4];
    uint64_t singular_matrices; /* transformed spheres whose matrix has det == 0 (:148) */
    int32_t  chunks;            /* row chunks the frame was traced in */
    int32_t  levels;            /* generations run (max_depth + 1) */
    float    ms_total;          /* device time of the last render, HIP events on its stream */
    float    ms_closest;        /* RTX_F_TIME_KERNELS only: summed device time of the generation-0 closest-hit launches
                                   (k_closest_f; with RTX_F_COUNT_TESTS / RTX_F_EXACT_ONLY: of every generation's) */
    float    ms_shadow;         /*                          ... of the shadow any-hit launches (k_shadow_pass) */
    /* RTX_F_PACKET_STATS only.  [0..3] closest-hit: generation 0 adds {waves, waves that made a packet,
     * 64-record batches tested, primitives kept}, later generations {waves, rays sent to the ordered scan
     * | abnormal rays << 32, unused, exact tests}; generation 0 on the camera's table (k_primary_cells) adds to [2] only:
     * {primary rays sent to the ordered scan over their own list | over every primitive << 32}.  [4..7] shadow pass: {lane blocks traced, blocks
     * that made a packet | rays traced transposed << 32, 64-record batches tested, plain primitives kept | affine spheres kept << 32} */
    uint64_t packet_stats[8];
    int32_t  launches_closest;  /* launches behind ms_closest / ms_shadow */
    int32_t  launches_shadow;
    float    ms_closest_lanes;  /* RTX_F_TIME_KERNELS only: summed device time of the per-ray closest-hit launches of
                                   generations >= 1 (k_closest_lanes) */
    int32_t  launches_closest_lanes;
    /* RTX_F_COUNT_EXEC only (0 otherwise): EXECUTED work of the production kernels, [kernel][kind].  kernel: 0 k_closest_f
     * (generation 0), 1 k_closest_lanes (generations >= 1), 2 k_shadow_pass.  kind (one count per (ray or packet, sphere
     * or primitive) pair a lane evaluated): 0 packet tests (a bounding sphere against a wave's packet), 1 cone tests (an
     * enclosing sphere against one ray), 2 per-ray bounding-sphere filter tests, 3 the same for affine spheres on shadow
     * rays, 4 affine strip tests, 5 exact sphere tests (raytracing.cpp:202-228), 6 exact affine-sphere tests, 7 exact
     * triangle tests that left at vd == 0 or t < 0 (:340-350), 8 ... that left at the nearest-t cull (t >= nearest),
     * 9 ... that ran the three areas (:352-365), 10 polygon fan triangles tested, 11 packets built, 12 rays traced,
     * 13 exact triangle tests that left at the edge filter. */
    uint64_t exec_counts[3][RTX_EXEC_KINDS];
    int32_t  shadow_form;       /* the form of the shadow stage this context runs: 1 tables, 0 packets (RTX_F_SHADOW_TABLES) */
    int32_t  reserved0;
    uint64_t queue_overflow;    /* secondary rays dropped because a generation's queue was full, since the context was created.
                                   0 by construction: capacities are the worst case, or (scenes with primitives that both
                                   reflect and refract) measured by a calibration pass over the very frame the context renders */
} rtx_stats;

enum {
    RTX_OK            =  0,
    RTX_E_INVALID_ARG = -1,
    RTX_E_NO_DEVICE   = -2,
    RTX_E_HIP         = -3,
    RTX_E_OOM         = -4,
    RTX_E_IO          = -5,
    RTX_E_TOO_LARGE   = -6,
    RTX_E_RCCL        = -7
};

/* ---- library ---------------------------------------------------------------------------- */
uint32_t    rtx_abi_version(void);
const char* rtx_strerror(int code);
/* Number of HIP devices visible, 0 if none (never an error code). */
int         rtx_device_count(void);
/* Last HIP error string seen by the calling thread ("" if none). */
const char* rtx_last_hip_error(void);

/* ---- one-shot drop-in: replaces raytracing.cpp:725-783 --------------------------------------
 * Renders the rows `params` selects (all of them for nranks <= 1) with 3x3 antialiasing and
 * the reference's 8-bit quantisation (:792-794) into rgb_out[rows * width * 3].  With
 * params->nranks <= 1 and ngpus > 1 the frame is split into bands of params->band_rows (default 16) rows round-robin
 * over the first `ngpus` devices of this process, one context per device; the 8-bit bands are gathered into device 0
 * with RCCL (grouped ncclSend / ncclRecv over xGMI; librccl.so is loaded on first use, RTX_E_RCCL if that or a
 * collective fails), copied to the host once and put in place in rgb_out.  ngpus beyond the visible devices is clamped.
 *
 * The single-device form keeps its context (scene records, cell table, ray queues: HBM) after the call and reuses it when the
 * next call reads exactly the same bytes -- a 128-bit hash over every record of the scene view, its scalars, the parameters
 * and the device -- so a host that renders the same scene once per frame pays context creation (45 of 55 ms at 4K) once.  An
 * edited record gives a new context.  rtx_release_cached_context() frees it; RTX_F_NO_CACHE never keeps one. */
int rtx_render_rgb8(const rtx_scene* scene, const rtx_params* params, int ngpus,
                    uint8_t* rgb_out, rtx_stats* stats_or_null);
/* Frees the context the one-shot entry points keep (no-op if none).  No counterpart in the reference. */
void rtx_release_cached_context(void);

/* The same call (ngpus = 1) for callers that cannot allocate device memory -- tight serving loops, harnesses that forbid
 * hipMalloc.  (NOT for code under HIP graph capture: the call uploads the scene with synchronous copies, reads the frame
 * back and waits for the default stream.)  NO counterpart in the reference.  Scene records, ray queues, the
 * float image and the 8-bit rows all live in one file-scope `static __device__` array of RTX_NOALLOC_SCRATCH_MIB MiB
 * that every GPU the library runs on holds until the process ends; the call makes no hipMalloc / hipFree, creates no
 * stream and no event, and traces the frame in as many row chunks as that block needs (everything on the default
 * stream, as RTX_F_SERIAL).  Same bytes and the same ray counts as rtx_render_rgb8; rtx_stats.ms_* stay 0.
 * RTX_E_TOO_LARGE (text in rtx_last_hip_error) when the scene's device records, the (rows + halo) x width float image
 * and one row of ray queues do not fit the block.  Every call on a GPU shares that GPU's array, which keeps the last
 * call's contents: calls are serialised per device by the library; a second PROCESS has its own array. */
#define RTX_NOALLOC_SCRATCH_MIB 64
int rtx_render_rgb8_noalloc(const rtx_scene* scene, const rtx_params* params,
                            uint8_t* rgb_out, rtx_stats* stats_or_null);

/* Same rows, but the pre-antialiasing 1-sample-per-pixel float image: what the reference's
 * render2d() (:801-833) holds in image[][] before quantisation.  out[rows * width * 3]. */
int rtx_render_f32(const rtx_scene* scene, const rtx_params* params,
                   float* rgb_out, rtx_stats* stats_or_null);

/* ---- resident context: scene + all work buffers live in HBM across frames ------------------
 * (what a serving loop or the benchmark uses; create allocates, render does not). */
typedef struct rtx_context rtx_context;

int  rtx_context_create(const rtx_scene* scene, const rtx_params* params, rtx_context** out);
void rtx_context_destroy(rtx_context* ctx);
/* Rows this context renders (its bands), and their global row numbers (rows_out may be NULL). */
int  rtx_context_rows(const rtx_context* ctx, int32_t* rows_out, int32_t capacity);
/* Enqueue one full render of the context's rows on `hip_stream` (a hipStream_t, NULL = default
 * stream).  Asynchronous.  If d_rgb8_out is non-NULL it is a DEVICE pointer to rows*width*3
 * bytes that receives the quantised framebuffer (e.g. the send buffer of an RCCL gather);
 * otherwise the context's own device framebuffer is used. */
int  rtx_context_render(rtx_context* ctx, void* hip_stream, uint8_t* d_rgb8_out);
/* Wait for the stream, then copy out.  host buffers: rows*width*3 elements. */
int  rtx_context_read_rgb8(rtx_context* ctx, uint8_t* host_rgb8_out);
int  rtx_context_read_f32(rtx_context* ctx, float* host_rgb_f32_out);
int  rtx_context_stats(rtx_context* ctx, rtx_stats* stats_out);

/* ---- band arithmetic (pure host functions; no device needed) -------------------------------- */
/* Number of rows rank `rank` owns; fills rows_out[0..n) with their global indices if non-NULL. */
int  rtx_band_rows(int32_t height, int32_t nranks, int32_t rank, int32_t band_rows,
                   int32_t* rows_out, int32_t capacity);

/* ---- diagnostics (pure host functions; no counterpart in the reference) -----------------------
 * All of them validate the scene view like rtx_context_create, return the number of records (or a negative
 * RTX_E_* code), accept a NULL output pointer to query that number, and fail with RTX_E_INVALID_ARG when
 * `capacity` (in records) is smaller.
 *
 * rtx_filter_records: the conservative bounding spheres the intersection kernels filter with, one { c.x, c.y,
 * c.z, -R^2 } record per primitive -- spheres first, then triangles, then one always-pass record (-R^2 = -inf)
 * per polygon: nspheres + ntriangles + npolygons records of 4 floats -- and the origin guard beyond which a ray
 * bypasses the filter.  Lets the CPU test suite prove the filter never rejects a pair the reference's test accepts. */
int  rtx_filter_records(const rtx_scene* scene, float* out4, int32_t capacity, float* guard_out);
/* The shadow-ray filter records of the affine spheres (16 floats each: rows a,b,c of the inverse's
 * linear part, centre, -R^2, sphere index as int bits, R and ||A||_F rounded up). */
int  rtx_affine_shadow_records(const rtx_scene* scene, float* out16, int32_t capacity);

/* The triangles' edge-filter records (8 floats each: a1.xyz, m, a2.xyz, Q; rtx_bounds.cpp): between the plane
 * intersection and the area square roots of raytracing.cpp:338-365 the kernels reject a point P with
 * |P - v|_1 <= Q whose barycentric coordinate a1.(P - v), a2.(P - v) or 1 - both is below -m (-2m for the
 * third).  Q = -1: never rejected.  Exported so the CPU suite can check that bound against the reference's test. */
int  rtx_triangle_edge_records(const rtx_scene* scene, float* out8, int32_t capacity);

/* The cluster hierarchy the traversal kernels walk (rtx_pack.cpp: build_clusters): which = 0, the closest-hit set over
 * the records of rtx_filter_records; which = 1, the shadow set (plain spheres and triangles); which = 2, the set the per-ray
 * traversal walks (its own unsegmented one in scenes of more than 16,384 primitives, else set 0).  Returns the number of
 * clusters ncl; with non-NULL outputs (capacity in clusters, >= ncl) it fills
 *   cluster4 [ncl * 4]       { c.x, c.y, c.z, R }   the cluster spheres (R = +inf: always passes),
 *   sub4     [ncl * 8 * 4]   { c, R }               the spheres around members 8s .. 8s+7 of every cluster,
 *   member4  [ncl * 64 * 4]  { c, -R^2 }            the members' own filter records, padding = { 0, 0, 0, +inf },
 *   index    [ncl * 64]      primitive index of every member slot (sphere i | nspheres + triangle i | ...), -1 = padding.
 * Any of the four may be NULL.  Exported so the CPU suite can check what the culling argument rests on: every primitive
 * whose record can pass sits in exactly one slot, and every member's sphere lies inside its sub-cluster's and its
 * cluster's.  No counterpart in the reference. */
int  rtx_cluster_records(const rtx_scene* scene, int32_t which, float* cluster4, float* sub4, float* member4,
                         int32_t* index, int32_t capacity);

/* The per-ray traversal's order and run spheres of cluster set `which` (as rtx_cluster_records): order [ncl] = the clusters in
 * traversal order, the first *nfirst_out of them the clusters of affine spheres; run4 [*nruns_out * 4] = { c, R } around every
 * run of 8 consecutive clusters of that order, the affine-first part and the rest counted apart (R = +inf: always passes).
 * k_closest_lanes skips a run whose sphere a ray's cone misses or that lies wholly beyond the ray's far limit, so every
 * cluster sphere must lie inside its run's (tests/test_clusters.py).  capacity in entries, >= ncl; returns ncl.  No
 * counterpart in the reference. */
int  rtx_cluster_runs(const rtx_scene* scene, int32_t which, int32_t* order, float* run4, int32_t capacity, int32_t* nfirst_out,
                      int32_t* nruns_out);

/* The direction map of the shadow stage (rtx_shadow.hip.h: light_cell; rtx_cells.hip.h: cell_cone), the kernels' own functions
 * compiled for the host.  rtx_light_cell: the cell (Morton index, 0 .. G*G-1) of the G x G octahedral map a direction falls
 * into (NaN components land in cell 0), G <= 4096.  rtx_cell_cone: out7 = { axis x y z, sin(half-angle) rounded up,
 * cos(half-angle) rounded down, cell column, cell row }: every direction rtx_light_cell maps to `cell` lies within that
 * half-angle of the axis, which is what the (light, cell) tables are built on.  Exported so the CPU suite can check both at
 * every map size in use.  No counterpart in the reference. */
int  rtx_light_cell(const float* dir3, int32_t G);
int  rtx_cell_cone(int32_t cell, int32_t G, float* out7);

/* The tests k_closest_lanes runs on the spheres { c, R } of its hierarchy (rtx_closest.hip.h), the kernel's own functions compiled
 * for the host: for a ray (origin3, unit dir3) out3 = { cone_slack, cone_slack_front, cone_slack_cull with the far limit lr }; a
 * value with the sign bit set means the sphere and everything in it is skipped.  cone_slack_cull skips what lies wholly beyond
 * lr = 1.015625 L + 1e-5 guard, L = the largest t at which a hit could still change the ray's state; tests/test_far_cull.py
 * checks against the reference's own sphere arithmetic that no hit at t <= L is ever inside a skipped sphere.  No counterpart
 * in the reference. */
int  rtx_cone_tests(const float* origin3, const float* dir3, const float* sphere4, float lr, float* out3);

/* ---- host side kept from the reference (C++ inside, C outside) ------------------------------ */
/* Scene-file reader with the grammar and stream semantics of parseinput (raytracing.cpp:851-979),
 * records zero-filled before parsing.  Polygons keep parseinput's rules: normal from the
 * first three vertices, exact-equality coplanarity check, centroid (:913-966).
 * The returned scene owns its arrays; release with rtx_scene_free. */
int  rtx_scene_load(const char* path, rtx_scene** out);
/* OPT-IN, not reference behaviour: reads the OLDER, line-per-record grammars the five bundled inp*.txt
 * were written in, as their own trailing comments document them (inp.txt:15-17, inp-refl.txt:9-11, inp1.txt:46-50,
 * inp01.txt:36-39), into the same records, so those scenes render as their authors meant them instead of the grey frame
 * / single disc parseinput makes of them at HEAD.  Fields those grammars never had are defined by this
 * loader (rtx_host.cpp: diffcolor = color, speccoeff 0.5, specexpo 20, ambient 0.05); a `cuboid` becomes 12 triangles.
 * rtx_scene_load stays the contract; the CLI selects this one with --legacy-grammar.  Release with rtx_scene_free. */
int  rtx_scene_load_legacy(const char* path, rtx_scene** out);
void rtx_scene_free(rtx_scene* scene);
/* "P6\n%i %i 255\n" + bytes, as raytracing.cpp:786-797. */
int  rtx_write_ppm(const char* path, int32_t width, int32_t height, const uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H_ */
