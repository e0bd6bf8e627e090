/*
 * rt_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, scalar, recursive) of the hot path of ChandanY1011/ray-tracing,
 * raytracing.cpp ("rt.cpp" below).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker.  The product library (librtx.so)
 * never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement bit-for-bit
 * (raw float image) and byte-for-byte (PPM) against outputs of the reference itself, compiled
 * in the build container by oracle/Makefile (`make ref`) and committed as small fixtures under
 * tests/golden/ by tests/golden/make_golden.py.
 *
 * Scene/record layouts are the reference's (rt.cpp:15-68), shared with include/rtx.h.
 */
#ifndef RT_ORACLE_H_
#define RT_ORACLE_H_
#include "../include/rtx.h"

typedef struct orc_counts {
    unsigned long long trace_calls;   /* raytrace() invocations (rt.cpp:371) */
    unsigned long long shadow_rays;   /* light-loop iterations (rt.cpp:560)  */
    unsigned long long hits;
} orc_counts;

/* rt.cpp:116-199 */
void        orc_normalize(rtx_vec3* v);
float       orc_determinant(rtx_vec3 a, rtx_vec3 b, rtx_vec3 c);
rtx_xmatrix orc_inverse_matrix(rtx_xmatrix m, int* singular);
rtx_vec3    orc_transform(rtx_vec3 v, rtx_xmatrix m);
rtx_vec3    orc_invtransform(rtx_vec3 v, rtx_xmatrix m, int transpose);
/* rt.cpp:202-233, :329-366 */
int orc_intersect_sphere(rtx_vec3 o, rtx_vec3 d, const rtx_sphere* s, rtx_vec3* normal, float* t0, float* t1);
int orc_intersect_triangle(rtx_vec3 o, rtx_vec3 d, const rtx_triangle* s, rtx_vec3* normal, float* t);
/* rt.cpp:289-325 */
float orc_find_area_polygon(const rtx_polygon* p, const rtx_vec3* verts, rtx_vec3 point);
int orc_intersect_polygon(rtx_vec3 o, rtx_vec3 d, const rtx_polygon* p, const rtx_vec3* verts, rtx_vec3* normal, float* t);
/* rt.cpp:275-287 (dead code in the reference: planes are parsed, never traced) */
int orc_intersect_plane(rtx_vec3 o, rtx_vec3 d, const rtx_plane* p, float* t, float* vd_out);
/* Opt-in extensions the build defines beyond the reference (include/rtx.h RTX_F_TRACE_PLANES,
 * RTX_F_TRI_MOLLER_TRUMBORE), restated here so the GPU path can be checked in those modes too.  Process-wide; 0 =
 * the reference's behaviour, which is what every parity test and fixture uses. */
#define ORC_TRACE_PLANES 1
#define ORC_TRI_MT 2
void orc_set_flags(int flags);
int  orc_get_flags(void);
/* rt.cpp:371-647 */
rtx_vec3 orc_raytrace(const rtx_scene* sc, int max_depth, rtx_vec3 o, rtx_vec3 d, int depth, orc_counts* cnt);
/* rt.cpp:792-794: one channel */
unsigned char orc_quantise(float v);

/* rt.cpp:801-833 (render2d): 1 sample per pixel, rows [row0,row1), out[(row1-row0)*width*3] */
int orc_render_f32(const rtx_scene* sc, int width, int height, int max_depth,
                   int row0, int row1, float* out, orc_counts* cnt);
/* the same for the pixel rectangle rows [row0,row1) x columns [col0,col1) of the width x height frame (crops of
 * frames too large for one CPU core): out[(row1-row0)*(col1-col0)*3] */
int orc_render_f32_rect(const rtx_scene* sc, int width, int height, int max_depth,
                        int row0, int row1, int col0, int col1, float* out, orc_counts* cnt);
/* rt.cpp:719-797 (render2Dantialiasing).  literal_9x != 0 re-traces every neighbour sample as
 * the reference does (:745-776); 0 traces each pixel centre once and filters (SURVEY.md F9,
 * byte-identical).  out[height*width*3]. */
int orc_render_rgb8(const rtx_scene* sc, int width, int height, int max_depth,
                    int literal_9x, unsigned char* out, orc_counts* cnt);
/* The AA filter + quantiser alone on a full 1-spp float image (rt.cpp:738-783, :792-794). */
void orc_filter_quantise(const float* img, int width, int height, unsigned char* out);

/* test helper, see rt_oracle.c */
void orc_accept_matrix(const rtx_scene* sc, int nrays, const float* o, const float* d, int shadow, unsigned char* accept);

#endif
