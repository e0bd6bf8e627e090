/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See rt_oracle.h.
 *
 * Scalar, recursive CPU restatement of raytracing.cpp ("rt.cpp") for the render hot path.
 * Every function cites the reference lines it follows.  It is written to be BIT-identical to
 * the reference built with `g++ -O2 -ffp-contract=off` on x86-64 (SSE2 scalar float, no FMA):
 *   - C++ overload resolution is made explicit: sqrt(float) -> sqrtf, pow(float,float) -> powf,
 *     pow(double,int) -> pow(double,double), abs(float) -> fabsf, std::min/max -> the (b<a)?b:a
 *     forms (they differ from fminf/fmaxf on NaN);
 *   - wherever a double literal promotes a float expression in the reference, the expression is
 *     evaluated in double here too and rounded to float at the same place.
 * Build: gcc -O2 -ffp-contract=off -std=c99 (oracle/Makefile).
 */
#include "rt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_T 1e8 /* rt.cpp:10 */

typedef rtx_vec3 V;

static V mkv(float x, float y, float z) { V v; v.x = x; v.y = y; v.z = z; return v; }
static float min_f(float a, float b) { return (b < a) ? b : a; } /* std::min */
static float max_f(float a, float b) { return (a < b) ? b : a; } /* std::max */

/* rt.cpp:116-125 */
void orc_normalize(V* v)
{
    float norm = sqrtf((v->x) * (v->x) + (v->y) * (v->y) + (v->z) * (v->z));
    if (norm > 0.0) {
        v->x = v->x / norm;
        v->y = v->y / norm;
        v->z = v->z / norm;
    }
}

/* rt.cpp:127-130 */
static float dot(V a, V b) { float r = a.x * b.x + a.y * b.y + a.z * b.z; return r; }

/* rt.cpp:137-142 */
float orc_determinant(V a, V b, V c)
{
    float x = a.x * (b.y * c.z - b.z * c.y);
    float y = -1 * a.y * (b.x * c.z - b.z * c.x);
    float z = a.z * (b.x * c.y - b.y * c.x);
    return (x + y + z);
}

/* rt.cpp:144-165, bugs included (wrong cofactor at :160; singular => returns m itself :148-151) */
rtx_xmatrix orc_inverse_matrix(rtx_xmatrix m, int* singular)
{
    V a = m.a, b = m.b, c = m.c, d = m.d;
    rtx_xmatrix mm = m;
    float det = orc_determinant(m.a, m.b, m.c);
    if (singular) *singular = 0;
    if (det == 0) {
        if (singular) *singular = 1;
        return m;
    }
    mm.a.x = ((b.y * c.z - b.z * c.y)) / det;
    mm.a.y = (-1 * (a.y * c.z - c.y * a.z)) / det;
    mm.a.z = ((a.y * b.z - b.y * a.z)) / det;
    mm.b.x = (-1 * (b.x * c.z - c.x * b.z)) / det;
    mm.b.y = ((a.x * c.z - a.z * c.x)) / det;
    mm.b.z = (-1 * (a.x * b.z - a.z * b.x)) / det;
    mm.c.x = ((b.x * c.y - c.x * b.y)) / det;
    mm.c.y = (-1 * (a.x * c.y - a.y * c.x)) / det;
    mm.c.z = ((a.x * b.y - a.y * b.z)) / det; /* sic, rt.cpp:160 */
    mm.d.x = -1 * orc_determinant(b, c, d) / det;
    mm.d.y = orc_determinant(a, c, d) / det;
    mm.d.z = -1 * orc_determinant(a, b, d) / det;
    return mm;
}

/* rt.cpp:169-175: v.x is overwritten and then reused for v.y and v.z */
V orc_transform(V v, rtx_xmatrix m)
{
    v.x = v.x * m.a.x + v.y * m.b.x + v.z * m.c.x + m.d.x;
    v.y = v.x * m.a.y + v.y * m.b.y + v.z * m.c.y + m.d.y;
    v.z = v.x * m.a.z + v.y * m.b.z + v.z * m.c.z + m.d.z;
    return v;
}

/* rt.cpp:181-199 */
V orc_invtransform(V v, rtx_xmatrix m, int transpose)
{
    rtx_xmatrix im = orc_inverse_matrix(m, NULL);
    V ret = mkv(0, 0, 0);
    if (transpose == 0) {
        ret.x = v.x * im.a.x + v.y * im.b.x + v.z * im.c.x + im.d.x;
        ret.y = v.x * im.a.y + v.y * im.b.y + v.z * im.c.y + im.d.y;
        ret.z = v.x * im.a.z + v.y * im.b.z + v.z * im.c.z + im.d.z;
    } else {
        float val = v.x * im.d.x + v.y * im.d.y + v.z * im.d.z + 1;
        if (val == 0) return v;
        ret.x = (v.x * im.a.x + v.y * im.a.y + v.z * im.a.z) / val;
        ret.y = (v.x * im.b.x + v.y * im.b.y + v.z * im.b.z) / val;
        ret.z = (v.x * im.c.x + v.y * im.c.y + v.z * im.c.z) / val;
    }
    return ret;
}

/* rt.cpp:202-233 */
int orc_intersect_sphere(V o, V d, const rtx_sphere* s, V* normal, float* t0, float* t1)
{
    float a = 1.0;
    float b = 2 * (d.x * (o.x - s->center.x) +
                   d.y * (o.y - s->center.y) +
                   d.z * (o.z - s->center.z));
    float c = (o.x - s->center.x) * (o.x - s->center.x) +
              (o.y - s->center.y) * (o.y - s->center.y) +
              (o.z - s->center.z) * (o.z - s->center.z) -
              (s->radius) * (s->radius);
    float b2m4ac = b * b - 4 * a * c;
    if (b2m4ac < 0) return 0;
    *t0 = (-b - sqrtf(b2m4ac)) / (2 * a);
    *t1 = (-b + sqrtf(b2m4ac)) / (2 * a);
    if (*t0 < 0 && *t1 < 0) return 0;
    if (*t0 < 0) *t0 = *t1;
    float t = *t0;
    if (*t1 > 0) {
        *t0 = min_f(*t0, *t1);
        t = *t0;
    }
    normal->x = (o.x + d.x * t - s->center.x) / s->radius;
    normal->y = (o.y + d.y * t - s->center.y) / s->radius;
    normal->z = (o.z + d.z * t - s->center.z) / s->radius;
    return 1;
}

/* Opt-in extensions of the build (NOT reference behaviour; both off by default, see rt_oracle.h) */
static int g_flags = 0;
void orc_set_flags(int flags) { g_flags = flags; }
int orc_get_flags(void) { return g_flags; }

/* ORC_TRI_MT: Moller-Trumbore in plain binary32 operations, the order of triangle_test_mt (rtx_math.hip.h);
 * the facing normal stays the reference's (unit `cross`, sign of vd, rt.cpp:343-347).  No reference counterpart. */
static int intersect_triangle_mt(V o, V dir, const rtx_triangle* s, V* normal, float* t)
{
    V v1 = mkv(s->u.x - s->v.x, s->u.y - s->v.y, s->u.z - s->v.z);
    V v2 = mkv(s->w.x - s->v.x, s->w.y - s->v.y, s->w.z - s->v.z);
    V cr = mkv(v1.y * v2.z - v2.y * v1.z, v2.x * v1.z - v1.x * v2.z, v1.x * v2.y - v2.x * v1.y);
    orc_normalize(&cr);
    float vd = dot(cr, dir);
    if (vd > 0) { normal->x = -cr.x; normal->y = -cr.y; normal->z = -cr.z; } else *normal = cr;
    float e1x = s->v.x - s->u.x, e1y = s->v.y - s->u.y, e1z = s->v.z - s->u.z;
    float e2x = s->w.x - s->u.x, e2y = s->w.y - s->u.y, e2z = s->w.z - s->u.z;
    float px = dir.y * e2z - dir.z * e2y, py = dir.z * e2x - dir.x * e2z, pz = dir.x * e2y - dir.y * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    if (det == 0.0f) return 0;
    float inv = 1.0f / det;
    float tx = o.x - s->u.x, ty = o.y - s->u.y, tz = o.z - s->u.z;
    float uu = (tx * px + ty * py + tz * pz) * inv;
    if (uu < 0.0f || uu > 1.0f) return 0;
    float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    float vv = (dir.x * qx + dir.y * qy + dir.z * qz) * inv;
    if (vv < 0.0f || uu + vv > 1.0f) return 0;
    *t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    if (*t < 0.0f) return 0;
    return 1;
}

/* rt.cpp:275-287 intersectplane (dead code in the reference; traced only with ORC_TRACE_PLANES) */
int orc_intersect_plane(V o, V dir, const rtx_plane* p, float* t, float* vd_out)
{
    V n = p->normal;
    orc_normalize(&n);
    float vd = dot(n, dir);
    if (vd_out) *vd_out = vd;
    if (vd == 0) return 0;
    *t = -(dot(n, o) + p->d) / (vd);
    if (*t < 0) return 0;
    return 1;
}

/* rt.cpp:329-366: plane hit + "three sub-areas sum to the area within 0.01" (not Moller-Trumbore) */
int orc_intersect_triangle(V o, V dir, const rtx_triangle* s, V* normal, float* t)
{
    if (g_flags & ORC_TRI_MT) return intersect_triangle_mt(o, dir, s, normal, t);
    V v1 = mkv(s->u.x - s->v.x, s->u.y - s->v.y, s->u.z - s->v.z);
    V v2 = mkv(s->w.x - s->v.x, s->w.y - s->v.y, s->w.z - s->v.z);
    V cross = mkv(v1.y * v2.z - v2.y * v1.z, v2.x * v1.z - v1.x * v2.z, v1.x * v2.y - v2.x * v1.y);
    float area = 0.5 * sqrtf(dot(cross, cross));
    orc_normalize(&cross);
    float d = -(dot(cross, s->u));
    float vd = dot(cross, dir);
    if (vd == 0) return 0;
    if (vd > 0) {
        normal->x = -cross.x; normal->y = -cross.y; normal->z = -cross.z;
    } else
        *normal = cross;
    *t = -(dot(cross, o) + d) / (vd);
    if (*t < 0) return 0;
    V point = mkv(o.x + dir.x * *t, o.y + dir.y * *t, o.z + dir.z * *t);
    V vec1 = mkv(point.x - s->v.x, point.y - s->v.y, point.z - s->v.z);
    V vec2 = mkv(point.x - s->w.x, point.y - s->w.y, point.z - s->w.z);
    V vec3 = mkv(s->u.x - s->w.x, s->u.y - s->w.y, s->u.z - s->w.z);
    V cross1 = mkv(vec1.y * v1.z - v1.y * vec1.z, v1.x * vec1.z - v1.z * vec1.x, v1.y * vec1.x - v1.x * vec1.y);
    V cross2 = mkv(vec1.z * v2.y - vec1.y * v2.z, vec1.x * v2.z - vec1.z * v2.x, vec1.y * v2.x - vec1.x * v2.y);
    V cross3 = mkv(vec2.z * vec3.y - vec2.y * vec3.z, vec2.x * vec3.z - vec2.z * vec3.x, vec2.y * vec3.x - vec2.x * vec3.y);
    float areau = 0.5 * sqrtf(dot(cross2, cross2));
    float areav = 0.5 * sqrtf(dot(cross3, cross3));
    float areaw = 0.5 * sqrtf(dot(cross1, cross1));
    if (fabsf(areau + areav + areaw - area) > 0.01) return 0;
    return 1;
}

/* rt.cpp:132-135 */
static V cross(V u, V v) { return mkv(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }

/* rt.cpp:289-300: fan of triangles (point, v_i, v_prev), areas summed in float */
float orc_find_area_polygon(const rtx_polygon* p, const rtx_vec3* verts, V point)
{
    float area = 0.0;
    int n = p->numvertices;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        V u = mkv(point.x - verts[i].x, point.y - verts[i].y, point.z - verts[i].z);
        V v = mkv(point.x - verts[j].x, point.y - verts[j].y, point.z - verts[j].z);
        V c = cross(u, v);
        float temparea = 0.5 * sqrtf(dot(c, c));
        area += temparea;
    }
    return area;
}

/* rt.cpp:303-325 as it behaves once it compiles: `dotproduct == 0` (:305) compares the
 * FUNCTION with 0 and is never true; `dotproduct > 0` (:308) does not compile and is `true` in the
 * one-token fix oracle/Makefile applies (what gcc 4.8 made of it), so the stored normal is ALWAYS negated. */
int orc_intersect_polygon(V o, V dir, const rtx_polygon* p, const rtx_vec3* verts, V* normal, float* t)
{
    if (p->numvertices < 1) return 0; /* vertices[0] does not exist (:312 would be out of bounds) */
    V n = p->normal;
    n.x *= -1; n.y *= -1; n.z *= -1;
    *normal = n;
    float d = -1 * (dot(n, verts[0]));
    *t = -(dot(n, o) + d) / dot(n, dir);
    if (*t < 0) return 0;
    V point = mkv(o.x + dir.x * *t, o.y + dir.y * *t, o.z + dir.z * *t);
    float totalArea = orc_find_area_polygon(p, verts, p->centerPoint);
    float sumArea = orc_find_area_polygon(p, verts, point);
    if (fabsf(totalArea - sumArea) > 0.001) return 0; /* float abs, double compare */
    return 1;
}

/* The reflect / refract block shared (textually duplicated in the reference) by spheres
 * (rt.cpp:462-494, with rd = traydir) and triangles (:497-525, with rd = raydir).
 * invert_eta is the sphere-only `insphere` rule (:477-479). */
static void secondary(const rtx_scene* sc, int max_depth, int depth, orc_counts* cnt,
                      V rd, V normal, V intpoint, float bias,
                      float reflectivity, float transparency, float refractive_index, V color,
                      int invert_eta, V* pix)
{
    if (reflectivity > 0) {
        float fratio = -1 * (dot(rd, normal));
        float fresnel = 0.1 * 1 + 0.9 * pow(1.0 - fratio, 3);
        V reflectdir = mkv(rd.x + 2 * fratio * normal.x, rd.y + 2 * fratio * normal.y, rd.z + 2 * fratio * normal.z);
        orc_normalize(&reflectdir);
        V biased = mkv(intpoint.x + normal.x * bias, intpoint.y + normal.y * bias, intpoint.z + normal.z * bias);
        V r = orc_raytrace(sc, max_depth, biased, reflectdir, depth + 1, cnt);
        pix->x += r.x * fresnel * color.x;
        pix->y += r.y * fresnel * color.y;
        pix->z += r.z * fresnel * color.z;
    }
    if (transparency > 0) {
        float refractindex = refractive_index;
        if (invert_eta) refractindex = 1.0 / refractindex;
        float fratio = -1 * (rd.x * normal.x + rd.y * normal.y + rd.z * normal.z);
        float fresnel = 0.1 * 1 + 0.9 * pow(1.0 - fratio, 3);
        float raydotn = -1 * (normal.x * rd.x + normal.y * rd.y + normal.z * rd.z);
        float k = 1.0 - refractindex * refractindex * (1.0 - raydotn * raydotn);
        V refractdir = mkv(rd.x * refractindex + normal.x * (refractindex * raydotn - sqrtf(k)),
                           rd.y * refractindex + normal.y * (refractindex * raydotn - sqrtf(k)),
                           rd.z * refractindex + normal.z * (refractindex * raydotn - sqrtf(k)));
        orc_normalize(&refractdir);
        V biased = mkv(intpoint.x - normal.x * bias, intpoint.y - normal.y * bias, intpoint.z - normal.z * bias);
        V r = orc_raytrace(sc, max_depth, biased, refractdir, depth + 1, cnt);
        /* (1.0 - fresnel) is double: the whole product and the += are evaluated in double */
        pix->x += r.x * (1.0 - fresnel) * transparency * color.x;
        pix->y += r.y * (1.0 - fresnel) * transparency * color.y;
        pix->z += r.z * (1.0 - fresnel) * transparency * color.z;
    }
}

/* rt.cpp:371-647 */
V orc_raytrace(const rtx_scene* sc, int max_depth, V rayorigin, V raydir, int depth, orc_counts* cnt)
{
    float nearest = MAX_T;
    const rtx_sphere* s = NULL;
    const rtx_triangle* tr = NULL;
    const rtx_polygon* pg = NULL;
    const rtx_plane* pl = NULL;
    V pix, normal = mkv(0, 0, 0);
    int objectfound = 0, transformed = 0;
    int ob = 0; /* 0 none, 1 sphere, 2 triangle, 3 polygon */
    if (cnt) cnt->trace_calls++;
    pix.x = pix.y = pix.z = 0.4;
    /* :382-404 */
    for (int i = 0; i < sc->nspheres; i++) {
        float t0 = MAX_T, t1 = MAX_T;
        V normaltemp, trayorigin = rayorigin, traydir = raydir;
        if (sc->spheres[i].istransformed) {
            trayorigin = orc_invtransform(rayorigin, sc->spheres[i].mat, 0);
            traydir = orc_invtransform(raydir, sc->spheres[i].mat, 0);
        }
        orc_normalize(&traydir);
        if (orc_intersect_sphere(trayorigin, traydir, &sc->spheres[i], &normaltemp, &t0, &t1)) {
            if (t0 < nearest) {
                nearest = t0;
                s = &sc->spheres[i];
                normal = normaltemp;
                if (s->istransformed) {
                    normal = orc_invtransform(normal, s->mat, 1);
                    transformed = 1; /* never reset by a later, nearer untransformed sphere */
                }
                ob = 1;
                objectfound = 1;
            }
        }
    }
    /* :406-419 */
    for (int i = 0; i < sc->ntriangles; i++) {
        float t = MAX_T;
        V normaltemp;
        if (orc_intersect_triangle(rayorigin, raydir, &sc->triangles[i], &normaltemp, &t)) {
            if (t < nearest) {
                nearest = t;
                tr = &sc->triangles[i];
                normal = normaltemp;
                ob = 2;
                objectfound = 1;
                transformed = 0;
            }
        }
    }
    /* :421-434 */
    for (int i = 0; i < sc->npolygons; i++) {
        float t = MAX_T;
        V normaltemp;
        const rtx_polygon* p = &sc->polygons[i];
        if (orc_intersect_polygon(rayorigin, raydir, p, sc->polygon_vertices + p->first_vertex, &normaltemp, &t)) {
            if (t < nearest) {
                nearest = t;
                pg = p;
                normal = normaltemp;
                ob = 3;
                objectfound = 1;
                transformed = 0;
            }
        }
    }
    /* ORC_TRACE_PLANES (extension; the reference never traces planes): after the polygons, file order, strict <;
     * intersectplane returns no normal: the triangle's facing rule (:343-347) on the normalised plane normal */
    for (int i = 0; (g_flags & ORC_TRACE_PLANES) && i < sc->nplanes; i++) {
        float t = MAX_T, vd = 0;
        if (orc_intersect_plane(rayorigin, raydir, &sc->planes[i], &t, &vd)) {
            if (t < nearest) {
                nearest = t;
                pl = &sc->planes[i];
                normal = pl->normal;
                orc_normalize(&normal);
                if (vd > 0) { normal.x = -normal.x; normal.y = -normal.y; normal.z = -normal.z; }
                ob = 4;
                objectfound = 1;
                transformed = 0;
            }
        }
    }
    if (!objectfound) return pix; /* :436-438, no ambient */
    if (cnt) cnt->hits++;
    V intpoint = mkv(rayorigin.x + nearest * raydir.x, rayorigin.y + nearest * raydir.y, rayorigin.z + nearest * raydir.z);
    if (transformed) intpoint = orc_transform(intpoint, s->mat); /* :441-444 */
    orc_normalize(&normal);
    pix.x = pix.y = pix.z = 0.0;
    float bias = 1e-4;
    if (depth < max_depth) { /* :449 */
        if (ob == 1) {
            V traydir = raydir;
            if (transformed) traydir = orc_invtransform(raydir, s->mat, 0);
            orc_normalize(&traydir);
            int insphere = 0;
            if (dot(normal, traydir) > 0) {
                normal.x *= -1; normal.y *= -1; normal.z *= -1;
                insphere = 1;
            }
            secondary(sc, max_depth, depth, cnt, traydir, normal, intpoint, bias,
                      s->reflectivity, s->transparency, s->refractive_index, s->color, insphere, &pix);
        } else if (ob == 2) {
            secondary(sc, max_depth, depth, cnt, raydir, normal, intpoint, bias,
                      tr->reflectivity, tr->transparency, tr->refractive_index, tr->color, 0, &pix);
        } else if (ob == 3) { /* :527-556: the triangle block again with the polygon's material */
            secondary(sc, max_depth, depth, cnt, raydir, normal, intpoint, bias,
                      pg->reflectivity, pg->transparency, pg->refractive_index, pg->color, 0, &pix);
        } else { /* extension: the triangle block with the plane's material */
            secondary(sc, max_depth, depth, cnt, raydir, normal, intpoint, bias,
                      pl->reflectivity, pl->transparency, pl->refractive_index, pl->color, 0, &pix);
        }
    }
    /* :560-641 */
    for (int li = 0; li < sc->nlights; li++) {
        const rtx_light* L = &sc->lights[li];
        V lightdir = mkv(L->point.x - intpoint.x, L->point.y - intpoint.y, L->point.z - intpoint.z);
        orc_normalize(&lightdir);
        float dotprod = max_f((float)0.0, dot(normal, lightdir));
        int inshadow = 0;
        V normaltemp;
        float t0, t1, t;
        if (cnt) cnt->shadow_rays++;
        V biased = mkv(intpoint.x + normal.x * bias, intpoint.y + normal.y * bias, intpoint.z + normal.z * bias);
        for (int i = 0; i < sc->nspheres; i++) {
            V tlightdir = lightdir;
            if (sc->spheres[i].istransformed) tlightdir = orc_invtransform(lightdir, sc->spheres[i].mat, 0);
            orc_normalize(&tlightdir);
            if (orc_intersect_sphere(biased, tlightdir, &sc->spheres[i], &normaltemp, &t0, &t1)) {
                inshadow = 1;
                break;
            }
        }
        if (inshadow) continue;
        for (int i = 0; i < sc->ntriangles; i++) {
            if (orc_intersect_triangle(biased, lightdir, &sc->triangles[i], &normaltemp, &t)) {
                inshadow = 1;
                break;
            }
        }
        if (!inshadow) {
            V vdir = raydir, ldir = lightdir, speccoeff, diffcolor;
            float specexpo;
            if (ob == 1) { /* :591-612 */
                if (s->istransformed) {
                    ldir = orc_invtransform(lightdir, s->mat, 0);
                    vdir = orc_invtransform(raydir, s->mat, 0);
                }
                orc_normalize(&ldir);
                orc_normalize(&vdir);
                speccoeff = s->speccoeff; diffcolor = s->diffcolor; specexpo = s->specexpo;
            } else if (ob == 2) { /* :613-626: no normalisation of either direction */
                speccoeff = tr->speccoeff; diffcolor = tr->diffcolor; specexpo = tr->specexpo;
            } else if (ob == 3) { /* :627-639: the same form for polygons */
                speccoeff = pg->speccoeff; diffcolor = pg->diffcolor; specexpo = pg->specexpo;
            } else { /* extension: a plane has no such fields (:43-49): diffcolor = color, speccoeff = 0, specexpo = 1 */
                speccoeff = mkv(0, 0, 0); diffcolor = pl->color; specexpo = 1.0f;
            }
            float ndoti2 = 2 * (dot(ldir, normal));
            V reflectedray = mkv(ndoti2 * normal.x - ldir.x, ndoti2 * normal.y - ldir.y, ndoti2 * normal.z - ldir.z);
            orc_normalize(&reflectedray);
            float rdotv = -1 * (dot(vdir, reflectedray));
            pix.x += speccoeff.x * powf(rdotv, specexpo) * L->color.x;
            pix.y += speccoeff.y * powf(rdotv, specexpo) * L->color.y;
            pix.z += speccoeff.z * powf(rdotv, specexpo) * L->color.z;
            pix.x += dotprod * diffcolor.x * L->color.x;
            pix.y += dotprod * diffcolor.y * L->color.y;
            pix.z += dotprod * diffcolor.z * L->color.z;
        }
    }
    /* :643-645 */
    pix.x += sc->ambient.x;
    pix.y += sc->ambient.y;
    pix.z += sc->ambient.z;
    return pix;
}

/* Camera of render2Dantialiasing / render2d (rt.cpp:725-737 == :807-819) */
typedef struct cam {
    V ncrossv, orig;
    float whratio, dwidth, dheight, angleofview;
} cam;

static cam make_cam(const rtx_scene* sc, int width, int height)
{
    cam c;
    c.ncrossv.x = -1 * (sc->vpn.y * sc->up.z - sc->vpn.z * sc->up.y);
    c.ncrossv.y = -1 * (sc->up.x * sc->vpn.z - sc->vpn.x * sc->up.z);
    c.ncrossv.z = -1 * (sc->vpn.x * sc->up.y - sc->vpn.y * sc->up.x);
    orc_normalize(&c.ncrossv);
    c.whratio = (float)width / (float)height;
    c.dwidth = 1.0 / (float)width;
    c.dheight = 1.0 / (float)height;
    c.angleofview = 1.0 / sqrtf(dot(sc->camera, sc->camera));
    c.orig.x = sc->camera.x * c.ncrossv.x + sc->camera.y * sc->up.x + sc->camera.z * sc->vpn.x + sc->vrp.x;
    c.orig.y = sc->camera.x * c.ncrossv.y + sc->camera.y * sc->up.y + sc->camera.z * sc->vpn.y + sc->vrp.y;
    c.orig.z = sc->camera.x * c.ncrossv.z + sc->camera.y * sc->up.z + sc->camera.z * sc->vpn.z + sc->vrp.z;
    return c;
}

/* rt.cpp:753-760 */
static V sample(const rtx_scene* sc, const cam* c, int max_depth, int xval, int yval, orc_counts* cnt)
{
    float imagex = (2 * ((xval + 0.5) * c->dwidth) - 1) * c->angleofview * c->whratio;
    float imagey = (1 - 2 * ((yval + 0.5) * c->dheight)) * c->angleofview;
    V raydir;
    raydir.x = (imagex - sc->camera.x) * c->ncrossv.x + (imagey - sc->camera.y) * sc->up.x + (-sc->camera.z) * sc->vpn.x;
    raydir.y = (imagex - sc->camera.x) * c->ncrossv.y + (imagey - sc->camera.y) * sc->up.y + (-sc->camera.z) * sc->vpn.y;
    raydir.z = (imagex - sc->camera.x) * c->ncrossv.z + (imagey - sc->camera.y) * sc->up.z + (-sc->camera.z) * sc->vpn.z;
    orc_normalize(&raydir);
    return orc_raytrace(sc, max_depth, c->orig, raydir, 0, cnt);
}

/* rt.cpp:792-794: fputc(min(float(1.0), v)*255, f).  float -> int truncates; out-of-range and
 * NaN conversions give INT_MIN on x86-64 (cvttss2si), whose low byte is 0; made explicit here. */
unsigned char orc_quantise(float v)
{
    float m = min_f((float)1.0, v); /* NaN -> 1.0 */
    float p = m * 255;
    int i;
    if (!(p > -2147483648.0f)) i = (int)0x80000000u; /* below range, -inf */
    else i = (int)p;
    return (unsigned char)i;
}

int orc_render_f32_rect(const rtx_scene* sc, int width, int height, int max_depth,
                        int row0, int row1, int col0, int col1, float* out, orc_counts* cnt)
{
    if (!sc || !out || width <= 0 || height <= 0 || row0 < 0 || row1 > height || row0 > row1) return -1;
    if (col0 < 0 || col1 > width || col0 > col1) return -1;
    cam c = make_cam(sc, width, height);
    for (int y = row0; y < row1; y++)
        for (int x = col0; x < col1; x++) {
            V p = sample(sc, &c, max_depth, x, y, cnt);
            float* o = out + ((size_t)(y - row0) * (col1 - col0) + (x - col0)) * 3;
            o[0] = p.x; o[1] = p.y; o[2] = p.z;
        }
    return 0;
}

int orc_render_f32(const rtx_scene* sc, int width, int height, int max_depth,
                   int row0, int row1, float* out, orc_counts* cnt)
{
    return orc_render_f32_rect(sc, width, height, max_depth, row0, row1, 0, width, out, cnt);
}

/* rt.cpp:738-783 on an existing 1-spp image + :792-794 */
void orc_filter_quantise(const float* img, int width, int height, unsigned char* out)
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float counter = 0.0;
            float ax = 0.0f, ay = 0.0f, az = 0.0f; /* the reference's accumulator is uninitialised; +0 assumed */
            for (int yv = y - 1; yv <= y + 1; yv++)
                for (int xv = x - 1; xv <= x + 1; xv++)
                    if (xv >= 0 && yv >= 0 && xv < width && yv < height) {
                        const float* p = img + ((size_t)yv * width + xv) * 3;
                        counter = counter + 1.0;
                        if (yv == y && xv == x) { ax += 2 * p[0]; ay += 2 * p[1]; az += 2 * p[2]; }
                        else { ax += p[0]; ay += p[1]; az += p[2]; }
                    }
            ax = ax / (counter + 1); ay = ay / (counter + 1); az = az / (counter + 1);
            unsigned char* o = out + ((size_t)y * width + x) * 3;
            o[0] = orc_quantise(ax); o[1] = orc_quantise(ay); o[2] = orc_quantise(az);
        }
}

int orc_render_rgb8(const rtx_scene* sc, int width, int height, int max_depth,
                    int literal_9x, unsigned char* out, orc_counts* cnt)
{
    if (!sc || !out || width <= 0 || height <= 0) return -1;
    if (!literal_9x) {
        float* img = (float*)malloc((size_t)width * height * 3 * sizeof(float));
        if (!img) return -4;
        orc_render_f32(sc, width, height, max_depth, 0, height, img, cnt);
        orc_filter_quantise(img, width, height, out);
        free(img);
        return 0;
    }
    cam c = make_cam(sc, width, height);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            float counter = 0.0;
            V acc = mkv(0.0f, 0.0f, 0.0f);
            for (int yv = y - 1; yv <= y + 1; yv++)
                for (int xv = x - 1; xv <= x + 1; xv++)
                    if (xv >= 0 && yv >= 0 && xv < width && yv < height) {
                        counter = counter + 1.0;
                        V t = sample(sc, &c, max_depth, xv, yv, cnt);
                        if (yv == y && xv == x) { acc.x += 2 * t.x; acc.y += 2 * t.y; acc.z += 2 * t.z; }
                        else { acc.x += t.x; acc.y += t.y; acc.z += t.z; }
                    }
            acc.x = acc.x / (counter + 1); acc.y = acc.y / (counter + 1); acc.z = acc.z / (counter + 1);
            unsigned char* o = out + ((size_t)y * width + x) * 3;
            o[0] = orc_quantise(acc.x); o[1] = orc_quantise(acc.y); o[2] = orc_quantise(acc.z);
        }
    return 0;
}

/* Test helper: for every (ray, primitive) pair, does the reference's intersection routine return
 * true?  Spheres follow the closest-hit call pattern (rt.cpp:385-391: inverse-transformed origin
 * and direction for affine spheres, direction normalised); triangles rt.cpp:409.
 * accept[ray * (ns + nt) + prim] = 0/1.  shadow != 0 uses the shadow-ray call pattern instead
 * (rt.cpp:571-576: only the direction is inverse-transformed). */
void orc_accept_matrix(const rtx_scene* sc, int nrays, const float* o, const float* d, int shadow, unsigned char* accept)
{
    int np = sc->nspheres + sc->ntriangles;
    for (int r = 0; r < nrays; r++) {
        V ro = mkv(o[3 * r], o[3 * r + 1], o[3 * r + 2]);
        V rd = mkv(d[3 * r], d[3 * r + 1], d[3 * r + 2]);
        for (int i = 0; i < sc->nspheres; i++) {
            V nt, to = ro, td = rd;
            float t0 = MAX_T, t1 = MAX_T;
            if (sc->spheres[i].istransformed) {
                if (!shadow) to = orc_invtransform(ro, sc->spheres[i].mat, 0);
                td = orc_invtransform(rd, sc->spheres[i].mat, 0);
            }
            orc_normalize(&td);
            accept[(size_t)r * np + i] = (unsigned char)orc_intersect_sphere(to, td, &sc->spheres[i], &nt, &t0, &t1);
        }
        for (int i = 0; i < sc->ntriangles; i++) {
            V nt;
            float t = MAX_T;
            accept[(size_t)r * np + sc->nspheres + i] = (unsigned char)orc_intersect_triangle(ro, rd, &sc->triangles[i], &nt, &t);
        }
    }
}
