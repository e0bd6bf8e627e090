// rtx_pack.cpp -- see rtx_pack.h.  Build with -ffp-contract=off (no FMA on the host either).
#include "rtx_pack.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <utility>
#include <cstring>
#include <limits>

namespace rtx {

namespace {

inline float det3(const rtx_vec3& a, const rtx_vec3& b, const rtx_vec3& c)
{
    // raytracing.cpp:137-142; rows a,b,c.  `-1*a.y*(..)` is (-a.y)*(..)
    float x = a.x * (b.y * c.z - b.z * c.y);
    float y = (-a.y) * (b.x * c.z - b.z * c.x);
    float z = a.z * (b.x * c.y - b.y * c.x);
    return (x + y) + z;
}

inline void normalize(rtx_vec3& v)
{
    // raytracing.cpp:116-125: untouched when the norm is 0 or NaN
    float n = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    if (n > 0.0f) {
        v.x = v.x / n;
        v.y = v.y / n;
        v.z = v.z / n;
    }
}

inline float dot(const rtx_vec3& a, const rtx_vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

}  // namespace

rtx_xmatrix inverse_matrix(const rtx_xmatrix& m, bool* singular)
{
    // raytracing.cpp:144-165.  Adjugate / det with the reference's entry formulas kept as
    // written -- including c.z, whose second product is a.y*b.z where a true cofactor has
    // a.y*b.x (:160) -- and "singular => hand back m itself" (:148-151).
    const rtx_vec3 &a = m.a, &b = m.b, &c = m.c, &d = m.d;
    float det = det3(a, b, c);
    if (singular) *singular = (det == 0);
    if (det == 0) return m;
    rtx_xmatrix r = m;
    r.a.x = (b.y * c.z - b.z * c.y) / det;
    r.a.y = (-(a.y * c.z - c.y * a.z)) / det;
    r.a.z = (a.y * b.z - b.y * a.z) / det;
    r.b.x = (-(b.x * c.z - c.x * b.z)) / det;
    r.b.y = (a.x * c.z - a.z * c.x) / det;
    r.b.z = (-(a.x * b.z - a.z * b.x)) / det;
    r.c.x = (b.x * c.y - c.x * b.y) / det;
    r.c.y = (-(a.x * c.y - a.y * c.x)) / det;
    r.c.z = (a.x * b.y - a.y * b.z) / det;
    r.d.x = (-det3(b, c, d)) / det;
    r.d.y = det3(a, c, d) / det;
    r.d.z = (-det3(a, b, d)) / det;
    return r;
}

namespace {

struct Item { double c[3]; double R; int32_t slot; };

// f(i) for i in [0, n) on up to 8 host threads (chunks of `grain`); every i writes its own outputs, so the result does not
// depend on the schedule.  Context creation is inside the first rtx_render_rgb8 call of a scene: the enclosing spheres below
// are 65 sweeps over every cluster's members, 10 of the 12 ms pack_scene took for 10,000 primitives on one thread.
template <typename F>
void parallel_for(size_t n, size_t grain, F f)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::min<size_t>(std::min<size_t>(8, hw ? hw : 1), (n + grain - 1) / grain);
    if (nt <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t i0 = next.fetch_add(grain);
            if (i0 >= n) break;
            for (size_t i = i0; i < std::min(n, i0 + grain); i++) f(i);
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
}

// items [lo, hi) -> k-d order with leaves of `leaf` consecutive items (see emit_clusters)
void kd_order(std::vector<Item>& items, size_t lo, size_t hi, size_t leaf)
{
    while (hi - lo > leaf) {
        double bl[3] = {1e300, 1e300, 1e300}, bh[3] = {-1e300, -1e300, -1e300};
        for (size_t i = lo; i < hi; i++)
            for (int a = 0; a < 3; a++) { bl[a] = std::min(bl[a], items[i].c[a]); bh[a] = std::max(bh[a], items[i].c[a]); }
        int ax = 0;
        for (int a = 1; a < 3; a++) if (bh[a] - bl[a] > bh[ax] - bl[ax]) ax = a;
        const size_t leaves = (hi - lo + leaf - 1) / leaf;
        const size_t mid = lo + (leaves / 2) * leaf;
        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, [ax](const Item& a, const Item& b) {
            return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.slot < b.slot);
        });
        kd_order(items, lo, mid, leaf);
        lo = mid;
    }
}

// one group of items -> clusters of <= 64 consecutive items in k-d order of their centres
void emit_clusters(std::vector<Item>& items, bool bounded, const std::vector<BoundSphere>& rec,
                   const std::vector<int32_t>& ids, ClusterSet& out)
{
    if (items.empty()) return;
    if (bounded) {
        // k-d order: the run is cut at the median of its widest axis into two runs whose sizes are multiples of the
        // leaf size, down to leaves of 64 (clusters: every cluster but the last is full), then every cluster again
        // down to leaves of 8 (its sub-clusters).  (A surface-area sweep over all three axes and every admissible cut
        // was measured: same frame times on C4 and C5, twice the build time.)  Cells of a k-d partition are compact where runs of a space-filling
        // curve straddle its jumps: fewer cluster and sub-cluster spheres pass a ray or a packet.
        kd_order(items, 0, items.size(), kClusterSize);
        parallel_for((items.size() + kClusterSize - 1) / kClusterSize, 8, [&](size_t ci) {
            kd_order(items, ci * kClusterSize, std::min(items.size(), (ci + 1) * kClusterSize), 8);
        });
    }
    const float inf = std::numeric_limits<float>::infinity();
    // sphere around items [j0, j1): centre rounded to binary32 FIRST, radius measured from the rounded
    // centre and rounded up; { c, R }.  Not bounded: { 0, 0, 0, +inf }.
    auto enclose = [&](size_t j0, size_t j1) -> BoundSphere {
        if (!bounded || j0 >= j1) return {0.0f, 0.0f, 0.0f, (j0 >= j1) ? 0.0f : inf};
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = j0; i < j1; i++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::min(lo[a], items[i].c[a] - items[i].R);
                hi[a] = std::max(hi[a], items[i].c[a] + items[i].R);
            }
        // centre: from the middle of the box, a few dozen steps towards whichever member reaches furthest out
        // (Badoiu-Clarkson; converges on the smallest enclosing ball), keeping the best centre seen
        double cd[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
        double best[3] = {cd[0], cd[1], cd[2]}, bestR = 1e300;
        for (int it = 0; it <= 64; it++) {
            double far = -1.0; size_t fi = j0;
            for (size_t i = j0; i < j1; i++) {
                const double dx = items[i].c[0] - cd[0], dy = items[i].c[1] - cd[1], dz = items[i].c[2] - cd[2];
                const double d = std::sqrt(dx * dx + dy * dy + dz * dz) + items[i].R;
                if (d > far) { far = d; fi = i; }
            }
            if (far < bestR) { bestR = far; for (int a = 0; a < 3; a++) best[a] = cd[a]; }
            const double dx = items[fi].c[0] - cd[0], dy = items[fi].c[1] - cd[1], dz = items[fi].c[2] - cd[2];
            const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (!(dist > 0.0)) break;  // the furthest reach is a member around the centre itself (or not a number)
            // towards the far point of that member's own sphere, 1 / (it + 2) of the way
            const double step = (far / dist) / (double)(it + 2);
            cd[0] += dx * step; cd[1] += dy * step; cd[2] += dz * step;
        }
        float cf[3];
        for (int a = 0; a < 3; a++) cf[a] = (float)best[a];
        double R = 0.0;
        for (size_t i = j0; i < j1; i++) {
            double dx = items[i].c[0] - cf[0], dy = items[i].c[1] - cf[1], dz = items[i].c[2] - cf[2];
            R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + items[i].R);
        }
        float Rf = std::nextafter((float)(R * (1.0 + 1e-6)), inf);
        if (!std::isfinite(Rf) || !(Rf < 1e15f) || !std::isfinite(cf[0]) || !std::isfinite(cf[1]) || !std::isfinite(cf[2]))
            return {0.0f, 0.0f, 0.0f, inf};
        return {cf[0], cf[1], cf[2], Rf};
    };
    const size_t nclu = (items.size() + kClusterSize - 1) / kClusterSize;
    std::vector<BoundSphere> encl(nclu * 9);  // per cluster: its sphere, then its 8 sub-cluster spheres
    parallel_for(nclu, 4, [&](size_t ci) {
        const size_t i0 = ci * kClusterSize, i1 = std::min(items.size(), i0 + kClusterSize);
        encl[ci * 9] = enclose(i0, i1);
        for (size_t g = 0; g < 8; g++) encl[ci * 9 + 1 + g] = enclose(std::min(i0 + g * 8, i1), std::min(i0 + g * 8 + 8, i1));
    });
    for (size_t i0 = 0; i0 < items.size(); i0 += kClusterSize) {
        size_t i1 = std::min(items.size(), i0 + kClusterSize);
        const size_t ci = i0 / kClusterSize;
        BoundSphere cr = encl[ci * 9];
        BoundSphere cs = {0.0f, 0.0f, 0.0f, -inf};
        if (std::isfinite(cr.neg_r2)) {
            float nr2 = std::nextafter((float)(-((double)cr.neg_r2 * (double)cr.neg_r2)), -inf);
            if (std::isfinite(nr2)) cs = {cr.cx, cr.cy, cr.cz, nr2};
            else cr = {0.0f, 0.0f, 0.0f, inf};
        }
        out.cl.push_back(cs);
        out.clR.push_back(cr);
        for (size_t g = 0; g < 8; g++) out.subR.push_back(encl[ci * 9 + 1 + g]);
        for (size_t i = i0; i < i0 + kClusterSize; i++) {
            if (i < i1) { out.mem.push_back(rec[items[i].slot]); out.idx.push_back(ids[items[i].slot]); }
            else { out.mem.push_back({0.0f, 0.0f, 0.0f, inf}); out.idx.push_back(-1); }
        }
    }
}

}  // namespace

void build_clusters(const std::vector<BoundSphere>& rec, const std::vector<int32_t>& ids, int seg_prims,
                    const std::vector<uint8_t>* cls, ClusterSet& out)
{
    out = ClusterSet();
    const size_t n = rec.size();
    int32_t maxid = -1;
    for (int32_t v : ids) maxid = std::max(maxid, v);
    const int nseg = (seg_prims > 0) ? std::max(1, (maxid + seg_prims) / seg_prims) : 1;
    std::vector<uint8_t> cl_class;
    for (int seg = 0; seg < nseg; seg++) {
        out.seg_first.push_back((int32_t)out.cl.size());
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && !cls) break;
            std::vector<Item> small, big, unb;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (size_t i = 0; i < n; i++) {
                if (seg_prims > 0 && ids[i] / seg_prims != seg) continue;
                if (cls && ((*cls)[i] != 0) != (pass == 1)) continue;
                const BoundSphere& b = rec[i];
                if (b.neg_r2 == std::numeric_limits<float>::infinity()) continue;  // can never pass: in no cluster
                Item it = {{b.cx, b.cy, b.cz}, 0.0, (int32_t)i};
                bool ok = std::isfinite(b.cx) && std::isfinite(b.cy) && std::isfinite(b.cz) && std::isfinite(b.neg_r2) && b.neg_r2 <= 0.0f;
                if (!ok) { unb.push_back(it); continue; }
                it.R = std::sqrt(-(double)b.neg_r2);
                for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], it.c[a]); hi[a] = std::max(hi[a], it.c[a]); }
                small.push_back(it);
            }
            // primitives that are large against the cloud of centres would blow up their neighbours' cluster
            double ext = 0.0;
            for (int a = 0; a < 3; a++) ext = std::max(ext, hi[a] - lo[a]);
            std::vector<Item> keep;
            for (const Item& it : small) (it.R > 0.125 * ext && small.size() > (size_t)kClusterSize ? big : keep).push_back(it);
            emit_clusters(keep, true, rec, ids, out);
            emit_clusters(big, true, rec, ids, out);
            emit_clusters(unb, false, rec, ids, out);
            cl_class.resize(out.cl.size(), (uint8_t)pass);
        }
    }
    for (int pass = 1; pass >= 0; pass--) {
        for (size_t c = 0; c < cl_class.size(); c++)
            if (cl_class[c] == pass) out.order.push_back((int32_t)c);
        if (pass == 1) out.nfirst = (int32_t)out.order.size();
    }
    out.seg_first.push_back((int32_t)out.cl.size());
    // spheres around the rounds of the per-ray traversal
    const float inf = std::numeric_limits<float>::infinity();
    const int32_t ncl = (int32_t)out.order.size();
    for (int phase = 0; phase < 2; phase++) {
        const int32_t beg = phase ? out.nfirst : 0, end = phase ? ncl : out.nfirst;
        for (int32_t cb = beg; cb < end; cb += kSuperRun) {
            const int32_t ce = std::min(end, cb + kSuperRun);
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool bounded = true;
            for (int32_t u = cb; u < ce; u++) {
                const BoundSphere& b = out.clR[out.order[u]];
                const float cc[3] = {b.cx, b.cy, b.cz};
                if (!(b.neg_r2 < 1e15f)) { bounded = false; break; }  // +inf or NaN radius
                for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], (double)cc[a] - b.neg_r2); hi[a] = std::max(hi[a], (double)cc[a] + b.neg_r2); }
            }
            BoundSphere sp = {0.0f, 0.0f, 0.0f, inf};
            if (bounded) {
                const float cf[3] = {(float)(0.5 * (lo[0] + hi[0])), (float)(0.5 * (lo[1] + hi[1])), (float)(0.5 * (lo[2] + hi[2]))};
                double R = 0.0;
                for (int32_t u = cb; u < ce; u++) {
                    const BoundSphere& b = out.clR[out.order[u]];
                    const double dx = b.cx - cf[0], dy = b.cy - cf[1], dz = b.cz - cf[2];
                    R = std::max(R, std::sqrt(dx * dx + dy * dy + dz * dz) + (double)b.neg_r2);
                }
                const float Rf = std::nextafter((float)(R * (1.0 + 1e-6)), inf);
                if (std::isfinite(Rf) && Rf < 1e15f && std::isfinite(cf[0]) && std::isfinite(cf[1]) && std::isfinite(cf[2]))
                    sp = {cf[0], cf[1], cf[2], Rf};
            }
            out.supR.push_back(sp);
        }
    }
}

void pack_scene(const rtx_scene& sc, int width, int height, PackedScene& P, uint32_t flags)
{
    P = PackedScene();
    const bool mt = (flags & RTX_F_TRI_MOLLER_TRUMBORE) != 0, planes = (flags & RTX_F_TRACE_PLANES) != 0;
    int fan = 0;
    for (int i = 0; i < sc.nspheres; i++) {
        const rtx_sphere& s = sc.spheres[i];
        SphHot h = { s.center.x, s.center.y, s.center.z, s.radius * s.radius };
        P.sph_hot.push_back(h);
        P.sph_xf.push_back(s.istransformed ? 1 : 0);
        bool sing = false;
        P.sph_im.push_back(inverse_matrix(s.mat, &sing));
        if (s.istransformed && sing) P.singular++;
        P.sph.push_back(s);
        int f = (s.reflectivity > 0 ? 1 : 0) + (s.transparency > 0 ? 1 : 0);
        if (f > fan) fan = f;
    }
    for (int i = 0; i < sc.ntriangles; i++) {
        const rtx_triangle& t = sc.triangles[i];
        // raytracing.cpp:330-337
        rtx_vec3 v1 = { t.u.x - t.v.x, t.u.y - t.v.y, t.u.z - t.v.z };
        rtx_vec3 v2 = { t.w.x - t.v.x, t.w.y - t.v.y, t.w.z - t.v.z };
        rtx_vec3 n = { v1.y * v2.z - v2.y * v1.z, v2.x * v1.z - v1.x * v2.z, v1.x * v2.y - v2.x * v1.y };
        // 0.5*sqrt(float) is evaluated in double by the reference and rounded back: the product
        // with 0.5 is exact either way, so the float product is the same number.
        float area = 0.5f * std::sqrt(dot(n, n));
        normalize(n);
        float d = -(dot(n, t.u));
        rtx_vec3 e = { t.u.x - t.w.x, t.u.y - t.w.y, t.u.z - t.w.z };  // :355
        TriHot h = { { n.x, n.y, n.z, d,
                       t.v.x, t.v.y, t.v.z, area,
                       t.w.x, t.w.y, t.w.z, v1.x,
                       v1.y, v1.z, v2.x, v2.y,
                       v2.z, e.x, e.y, e.z } };
        if (mt) {
            // opt-in Moller-Trumbore record (rtx_math.hip.h, triangle_test_mt): u, e1 = v - u, e2 = w - u; the unit
            // normal, d and the area keep their places (facing rule of the set-up kernel, shadow order below)
            TriHot m = { { n.x, n.y, n.z, d,
                           t.u.x, t.u.y, t.u.z, area,
                           t.v.x - t.u.x, t.v.y - t.u.y, t.v.z - t.u.z, t.w.x - t.u.x,
                           t.w.y - t.u.y, t.w.z - t.u.z, 0.0f, 0.0f,
                           0.0f, 0.0f, 0.0f, 0.0f } };
            h = m;
        }
        P.tri_hot.push_back(h);
        P.tri_edge.push_back(tri_edge_filter(t));
        P.tri.push_back(t);
        int f = (t.reflectivity > 0 ? 1 : 0) + (t.transparency > 0 ? 1 : 0);
        if (f > fan) fan = f;
    }
    for (int i = 0; i < sc.nlights; i++) P.lights.push_back(sc.lights[i]);
    P.max_fanout = fan;
    for (size_t i = 0; i < P.sph.size(); i++) {
        const SphHot& h = P.sph_hot[i];
        const rtx_xmatrix& m = P.sph_im[i];
        TriHot r = { { h.cx, h.cy, h.cz, h.r2,
                       m.a.x, m.a.y, m.a.z, m.b.x,
                       m.b.y, m.b.z, m.c.x, m.c.y,
                       m.c.z, m.d.x, m.d.y, m.d.z,
                       P.sph_xf[i] ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f } };
        P.prim_hot.push_back(r);
    }
    for (const TriHot& t : P.tri_hot) P.prim_hot.push_back(t);
    for (int i = 0; i < sc.npolygons; i++) {
        const rtx_polygon& pg = sc.polygons[i];
        P.poly.push_back(pg);
        const rtx_vec3* vv = sc.polygon_vertices + pg.first_vertex;
        const int nv = pg.numvertices;
        // intersectpolygon's ray-independent head (:309-312, :320), in its float expression order.  With the
        // one-token fix that makes :308 compile the stored normal is ALWAYS negated.
        rtx_vec3 n = pg.normal;
        n.x *= -1; n.y *= -1; n.z *= -1;
        float d = 0.0f, total = 0.0f;
        if (nv >= 1) {
            d = -1 * (dot(n, vv[0]));
            for (int a = 0, b = nv - 1; a < nv; b = a++) {  // findAreaPolygon(p, p.centerPoint), :289-300
                rtx_vec3 u = { pg.centerPoint.x - vv[a].x, pg.centerPoint.y - vv[a].y, pg.centerPoint.z - vv[a].z };
                rtx_vec3 v = { pg.centerPoint.x - vv[b].x, pg.centerPoint.y - vv[b].y, pg.centerPoint.z - vv[b].z };
                rtx_vec3 c = { u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x };
                float temparea = 0.5f * std::sqrt(dot(c, c));
                total += temparea;
            }
        }
        int32_t first = (int32_t)(P.poly_verts.size() / 4), cnt = nv;
        float ff, fc;
        std::memcpy(&ff, &first, 4);
        std::memcpy(&fc, &cnt, 4);
        TriHot r = { { n.x, n.y, n.z, d, total, ff, fc, 0.0f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 } };
        P.prim_hot.push_back(r);
        for (int a = 0; a < nv; a++) { P.poly_verts.push_back(vv[a].x); P.poly_verts.push_back(vv[a].y); P.poly_verts.push_back(vv[a].z); P.poly_verts.push_back(0.0f); }
        int f = (pg.reflectivity > 0 ? 1 : 0) + (pg.transparency > 0 ? 1 : 0);
        if (f > P.max_fanout) P.max_fanout = f;
    }

    // RTX_F_TRACE_PLANES (opt-in; the reference parses planes and never traces them): each plane rides
    // behind the polygons as a record with vertex count -1: q0 = { normalize(normal) (:277), d as given }, and a
    // polygon-shaped material record -- a plane has no diffcolor / speccoeff / specexpo (:43-49), so this build
    // DEFINES diffcolor = color, speccoeff = 0, specexpo = 1.  Like polygons, planes cast no shadows.
    for (int i = 0; planes && i < sc.nplanes; i++) {
        const rtx_plane& pl = sc.planes[i];
        rtx_vec3 n = pl.normal;
        normalize(n);
        rtx_polygon g;
        std::memset(&g, 0, sizeof g);
        g.numvertices = -1;
        g.color = pl.color; g.diffcolor = pl.color; g.specexpo = 1.0f;
        g.normal = n;
        g.reflectivity = pl.reflectivity; g.transparency = pl.transparency; g.refractive_index = pl.refractive_index;
        P.poly.push_back(g);
        int32_t first = 0, cnt = -1;
        float ff, fc;
        std::memcpy(&ff, &first, 4);
        std::memcpy(&fc, &cnt, 4);
        TriHot r = { { n.x, n.y, n.z, pl.d, 0.0f, ff, fc, 0.0f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 } };
        P.prim_hot.push_back(r);
        int f = (pl.reflectivity > 0 ? 1 : 0) + (pl.transparency > 0 ? 1 : 0);
        if (f > P.max_fanout) P.max_fanout = f;
    }

    {   // packed materials, in primitive order: spheres, triangles, polygons (+ planes)
        auto put = [&](float refl, float tran, float ri, float radius, const rtx_vec3& col, float expo, const rtx_vec3& spec, float xf, const rtx_vec3& dif) {
            const float r[16] = {refl, tran, ri, radius, col.x, col.y, col.z, expo, spec.x, spec.y, spec.z, xf, dif.x, dif.y, dif.z, 0.0f};
            P.mat4.insert(P.mat4.end(), r, r + 16);
        };
        for (const rtx_sphere& q : P.sph) put(q.reflectivity, q.transparency, q.refractive_index, q.radius, q.color, q.specexpo, q.speccoeff, q.istransformed ? 1.0f : 0.0f, q.diffcolor);
        for (const rtx_triangle& q : P.tri) put(q.reflectivity, q.transparency, q.refractive_index, 0.0f, q.color, q.specexpo, q.speccoeff, 0.0f, q.diffcolor);
        for (const rtx_polygon& q : P.poly) put(q.reflectivity, q.transparency, q.refractive_index, 0.0f, q.color, q.specexpo, q.speccoeff, 0.0f, q.diffcolor);
    }

    // filter records.  guard = 16 x the extent of everything finite in the scene.
    double ext = 1.0;
    auto grow = [&](double v) { if (std::isfinite(v) && std::fabs(v) < 1e30 && std::fabs(v) > ext) ext = std::fabs(v); };
    for (const rtx_sphere& s : P.sph) {
        double r = std::fabs((double)s.radius);
        grow(std::fabs(s.center.x) + r); grow(std::fabs(s.center.y) + r); grow(std::fabs(s.center.z) + r);
    }
    for (const rtx_triangle& t : P.tri)
        for (const rtx_vec3* v : {&t.u, &t.v, &t.w}) { grow(v->x); grow(v->y); grow(v->z); }
    for (const rtx_light& l : P.lights) { grow(l.point.x); grow(l.point.y); grow(l.point.z); }
    for (float v : P.poly_verts) grow(v);
    grow(sc.camera.x); grow(sc.camera.y); grow(sc.camera.z);
    grow(sc.vrp.x); grow(sc.vrp.y); grow(sc.vrp.z);
    double guard = 16.0 * ext;
    P.guard = (float)guard;
    guard = (double)P.guard * 1.0001;
    for (size_t i = 0; i < P.sph.size(); i++)
        P.bounds.push_back(P.sph[i].istransformed ? bound_sphere_affine(P.sph[i], P.sph_im[i], guard)
                                                  : bound_sphere_plain(P.sph[i]));
    for (size_t i = 0; i < P.tri.size(); i++) P.bounds.push_back(bound_sphere_triangle(P.tri[i], guard));
    // polygons are not bounded yet: every ray takes their exact test (records that always pass)
    for (size_t i = 0; i < P.poly.size(); i++) P.bounds.push_back({0.0f, 0.0f, 0.0f, -std::numeric_limits<float>::infinity()});
    for (size_t i = 0; i < P.sph.size(); i++)
        if (P.sph[i].istransformed) P.aff_shadow.push_back(affine_shadow_record(P.sph[i], P.sph_im[i], (int)i));

    {   // shadow order: mean projected area, descending (sphere pi r^2, triangle area / 2); primitives the
        // filter cannot bound (always pass: an exact test every time) go last
        std::vector<std::pair<double, int32_t>> w;
        const int ns = (int)P.sph.size();
        for (int i = 0; i < ns; i++) {
            if (P.sph[i].istransformed) continue;
            double r = std::fabs((double)P.sph[i].radius);
            bool bounded = std::isfinite(P.bounds[i].neg_r2);
            w.push_back({bounded ? 3.141592653589793 * r * r : -1.0, i});
        }
        for (int i = 0; i < (int)P.tri.size(); i++) {
            bool bounded = std::isfinite(P.bounds[ns + i].neg_r2);
            w.push_back({bounded ? 0.5 * (double)P.tri_hot[i].q[7] : -1.0, ns + i});
        }
        std::stable_sort(w.begin(), w.end(), [](const std::pair<double, int32_t>& a, const std::pair<double, int32_t>& b) { return a.first > b.first; });
        for (auto& e : w) { P.sorder.push_back(e.second); P.sbounds.push_back(P.bounds[e.second]); }
    }

    {
        std::vector<int32_t> all(P.bounds.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = (int32_t)i;
        std::vector<uint8_t> cls(P.bounds.size(), 0);
        for (size_t i = 0; i < P.sph_xf.size(); i++) cls[i] = P.sph_xf[i] ? 1 : 0;
        std::thread shadow_set([&]() { build_clusters(P.sbounds, P.sorder, 0, nullptr, P.sc); });  // (independent of the other two)
        build_clusters(P.bounds, all, kSegPrims, &cls, P.cc);
        if (P.cc.seg_first.size() > 2) build_clusters(P.bounds, all, 0, &cls, P.pc);
        shadow_set.join();
    }

    // camera basis, raytracing.cpp:725-737
    P.camera = sc.camera;
    P.up = sc.up;
    P.vpn = sc.vpn;
    P.ambient = sc.ambient;
    P.ncrossv.x = -(sc.vpn.y * sc.up.z - sc.vpn.z * sc.up.y);
    P.ncrossv.y = -(sc.up.x * sc.vpn.z - sc.vpn.x * sc.up.z);
    P.ncrossv.z = -(sc.vpn.x * sc.up.y - sc.vpn.y * sc.up.x);
    normalize(P.ncrossv);
    float whratio = (float)width / (float)height;
    float dwidth = (float)(1.0 / (double)(float)width);
    float dheight = (float)(1.0 / (double)(float)height);
    float angleofview = (float)(1.0 / (double)std::sqrt(dot(sc.camera, sc.camera)));
    P.orig.x = sc.camera.x * P.ncrossv.x + sc.camera.y * sc.up.x + sc.camera.z * sc.vpn.x + sc.vrp.x;
    P.orig.y = sc.camera.x * P.ncrossv.y + sc.camera.y * sc.up.y + sc.camera.z * sc.vpn.y + sc.vrp.y;
    P.orig.z = sc.camera.x * P.ncrossv.z + sc.camera.y * sc.up.z + sc.camera.z * sc.vpn.z + sc.vrp.z;
    // :753-754 are double expressions stored to float; imagex depends only on the column and
    // imagey only on the row, so two small tables reproduce them exactly.
    P.imagex.resize(width);
    P.imagey.resize(height);
    for (int x = 0; x < width; x++)
        P.imagex[x] = (float)((2 * ((x + 0.5) * (double)dwidth) - 1) * (double)angleofview * (double)whratio);
    for (int y = 0; y < height; y++)
        P.imagey[y] = (float)((1 - 2 * ((y + 0.5) * (double)dheight)) * (double)angleofview);
}

}  // namespace rtx

namespace rtx {
// Pointer / count / range checks on a caller's scene view, shared by every entry point that reads one.
int validate_scene(const rtx_scene* sc)
{
    if (!sc) return RTX_E_INVALID_ARG;
    if (sc->nspheres < 0 || sc->ntriangles < 0 || sc->nlights < 0 || sc->nplanes < 0) return RTX_E_INVALID_ARG;
    if ((sc->nspheres && !sc->spheres) || (sc->ntriangles && !sc->triangles) || (sc->nlights && !sc->lights))
        return RTX_E_INVALID_ARG;
    if (sc->npolygons < 0 || sc->npolygon_vertices < 0) return RTX_E_INVALID_ARG;
    if ((sc->npolygons && !sc->polygons) || (sc->npolygon_vertices && !sc->polygon_vertices)) return RTX_E_INVALID_ARG;
    for (int i = 0; i < sc->npolygons; i++) {
        const rtx_polygon& g = sc->polygons[i];
        if (g.numvertices < 0 || g.first_vertex < 0 || (int64_t)g.first_vertex + g.numvertices > (int64_t)sc->npolygon_vertices)
            return RTX_E_INVALID_ARG;
    }
    return RTX_OK;
}
}  // namespace rtx

#define RTX_PACK_TRY try {
#define RTX_PACK_CATCH } catch (const std::bad_alloc&) { return RTX_E_OOM; } catch (...) { return RTX_E_INVALID_ARG; }

extern "C" int rtx_filter_records(const rtx_scene* scene, float* out4, int32_t capacity, float* guard_out)
{
    int rc = rtx::validate_scene(scene);
    if (rc) return rc;
    RTX_PACK_TRY
    rtx::PackedScene P;
    rtx::pack_scene(*scene, 1, 1, P, 0u);
    const int n = (int)P.bounds.size();  // spheres, triangles, then one always-pass record per polygon
    if (out4) {
        if (capacity < n) return RTX_E_INVALID_ARG;
        for (int i = 0; i < n; i++) {
            out4[4 * i + 0] = P.bounds[i].cx; out4[4 * i + 1] = P.bounds[i].cy;
            out4[4 * i + 2] = P.bounds[i].cz; out4[4 * i + 3] = P.bounds[i].neg_r2;
        }
    }
    if (guard_out) *guard_out = P.guard;
    return n;
    RTX_PACK_CATCH
}

extern "C" int rtx_affine_shadow_records(const rtx_scene* scene, float* out16, int32_t capacity)
{
    int rc = rtx::validate_scene(scene);
    if (rc) return rc;
    RTX_PACK_TRY
    rtx::PackedScene P;
    rtx::pack_scene(*scene, 1, 1, P, 0u);
    int n = (int)P.aff_shadow.size();
    if (out16) {
        if (capacity < n) return RTX_E_INVALID_ARG;
        static_assert(sizeof(rtx::AffineShadowRec) == 64, "16 floats");
        if (n) std::memcpy(out16, P.aff_shadow.data(), (size_t)n * 64);
    }
    return n;
    RTX_PACK_CATCH
}

extern "C" int rtx_cluster_records(const rtx_scene* scene, int32_t which, float* cluster4, float* sub4, float* member4,
                                   int32_t* index, int32_t capacity)
{
    int rc = rtx::validate_scene(scene);
    if (rc) return rc;
    if (which < 0 || which > 2) return RTX_E_INVALID_ARG;
    RTX_PACK_TRY
    rtx::PackedScene P;
    rtx::pack_scene(*scene, 1, 1, P, 0u);
    // which = 2: the set k_closest_lanes walks -- its own unsegmented one for scenes of more than one segment, else the closest-hit set
    const rtx::ClusterSet& cs = which == 1 ? P.sc : (which == 2 && !P.pc.clR.empty()) ? P.pc : P.cc;
    const int ncl = (int)cs.clR.size();
    if ((cluster4 || sub4 || member4 || index) && capacity < ncl) return RTX_E_INVALID_ARG;
    static_assert(sizeof(rtx::BoundSphere) == 16, "4 floats");
    if (cluster4 && ncl) std::memcpy(cluster4, cs.clR.data(), (size_t)ncl * 16);
    if (sub4 && ncl) std::memcpy(sub4, cs.subR.data(), (size_t)ncl * 8 * 16);
    if (member4 && ncl) std::memcpy(member4, cs.mem.data(), (size_t)ncl * rtx::kClusterSize * 16);
    if (index && ncl) std::memcpy(index, cs.idx.data(), (size_t)ncl * rtx::kClusterSize * sizeof(int32_t));
    return ncl;
    RTX_PACK_CATCH
}

extern "C" int rtx_cluster_runs(const rtx_scene* scene, int32_t which, int32_t* order, float* run4, int32_t capacity, int32_t* nfirst_out,
                                int32_t* nruns_out)
{
    int rc = rtx::validate_scene(scene);
    if (rc) return rc;
    if (which < 0 || which > 2) return RTX_E_INVALID_ARG;
    RTX_PACK_TRY
    rtx::PackedScene P;
    rtx::pack_scene(*scene, 1, 1, P, 0u);
    const rtx::ClusterSet& cs = which == 1 ? P.sc : (which == 2 && !P.pc.clR.empty()) ? P.pc : P.cc;
    const int ncl = (int)cs.order.size(), nruns = (int)cs.supR.size();
    if ((order || run4) && (capacity < ncl || capacity < nruns)) return RTX_E_INVALID_ARG;
    if (order && ncl) std::memcpy(order, cs.order.data(), (size_t)ncl * sizeof(int32_t));
    if (run4 && nruns) std::memcpy(run4, cs.supR.data(), (size_t)nruns * 16);
    if (nfirst_out) *nfirst_out = cs.nfirst;
    if (nruns_out) *nruns_out = nruns;
    return ncl;
    RTX_PACK_CATCH
}

extern "C" int rtx_triangle_edge_records(const rtx_scene* scene, float* out8, int32_t capacity)
{
    int rc = rtx::validate_scene(scene);
    if (rc) return rc;
    if (!out8) return scene->ntriangles;
    if (capacity < scene->ntriangles) return RTX_E_INVALID_ARG;
    static_assert(sizeof(rtx::TriEdge) == 32, "8 floats");
    for (int i = 0; i < scene->ntriangles; i++) {
        const rtx::TriEdge e = rtx::tri_edge_filter(scene->triangles[i]);
        std::memcpy(out8 + (size_t)8 * i, &e, sizeof(e));
    }
    return scene->ntriangles;
}
