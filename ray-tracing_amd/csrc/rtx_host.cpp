// rtx_host.cpp -- host-side pieces kept from the reference, behind the C-ABI (include/rtx.h):
// scene-file reader (grammar + stream semantics of parseinput, raytracing.cpp:851-979),
// PPM writer (:786-797), band arithmetic, error strings.  No device code here.
#include "rtx.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

namespace {

// The scene file is a whitespace-separated token stream read with operator>> (raytracing.cpp:853).
// Using the same iostream extraction keeps every quirk SURVEY.md section 4 documents: a word where a
// float is expected sets failbit and turns all later reads into no-ops; an int field given as
// "0.0" leaves ".0" for the next float; an unknown record keyword burns one record slot.
struct Reader {
    std::ifstream in;
    explicit Reader(const char* path) : in(path) {}
    void f(float& v) { in >> v; }
    void v3(rtx_vec3& v) { in >> v.x >> v.y >> v.z; }
    void named_v3(rtx_vec3& v) { std::string word; in >> word >> v.x >> v.y >> v.z; }
};

struct OwnedScene {
    rtx_scene view;  // must be first: rtx_scene* handed out aliases the OwnedScene
    std::vector<rtx_sphere> spheres;
    std::vector<rtx_triangle> triangles;
    std::vector<rtx_plane> planes;
    std::vector<rtx_light> lights;
    std::vector<rtx_polygon> polygons;
    std::vector<rtx_vec3> polygon_vertices;
};

// raytracing.cpp:116-125
inline void normalize_v(rtx_vec3& v)
{
    float n = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    if (n > 0.0f) { v.x = v.x / n; v.y = v.y / n; v.z = v.z / n; }
}

void publish(OwnedScene* sc, int dropped_polygons)
{
    sc->view.spheres = sc->spheres.empty() ? nullptr : sc->spheres.data();
    sc->view.nspheres = (int32_t)sc->spheres.size();
    sc->view.triangles = sc->triangles.empty() ? nullptr : sc->triangles.data();
    sc->view.ntriangles = (int32_t)sc->triangles.size();
    sc->view.planes = sc->planes.empty() ? nullptr : sc->planes.data();
    sc->view.nplanes = (int32_t)sc->planes.size();
    sc->view.lights = sc->lights.empty() ? nullptr : sc->lights.data();
    sc->view.nlights = (int32_t)sc->lights.size();
    sc->view.polygons = sc->polygons.empty() ? nullptr : sc->polygons.data();
    sc->view.npolygons = (int32_t)sc->polygons.size();
    sc->view.polygon_vertices = sc->polygon_vertices.empty() ? nullptr : sc->polygon_vertices.data();
    sc->view.npolygon_vertices = (int32_t)sc->polygon_vertices.size();
    sc->view.npolygons_dropped = dropped_polygons;
}

// parseinput's polygon rules on a complete vertex list (raytracing.cpp:913-959): normal from the first three vertices,
// exact-equality coplanarity check of every later one, centroid.  Returns false when the check drops the record.
bool finish_polygon(const std::vector<rtx_vec3>& vv, rtx_polygon& p)
{
    const int n = (int)vv.size();
    rtx_vec3 center = {0.0f, 0.0f, 0.0f}, normal = {0.0f, 0.0f, 0.0f};
    bool correct = true;
    for (int i = 0; i < n; i++) {
        center.x += vv[i].x; center.y += vv[i].y; center.z += vv[i].z;
        if (i == 2) {
            rtx_vec3 u = {vv[0].x - vv[1].x, vv[0].y - vv[1].y, vv[0].z - vv[1].z};
            rtx_vec3 v = {vv[0].x - vv[2].x, vv[0].y - vv[2].y, vv[0].z - vv[2].z};
            normal = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
            normalize_v(normal);
        }
        if (i > 2) {
            rtx_vec3 u = {vv[0].x - vv[1].x, vv[0].y - vv[1].y, vv[0].z - vv[1].z};
            rtx_vec3 v = {vv[0].x - vv[i].x, vv[0].y - vv[i].y, vv[0].z - vv[i].z};
            rtx_vec3 tn = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
            normalize_v(tn);
            bool same = (normal.x == tn.x && normal.y == tn.y && normal.z == tn.z) ||
                        (-1 * normal.x == tn.x && -1 * normal.y == tn.y && -1 * normal.z == tn.z);
            if (!same) correct = false;
        }
    }
    center.x = center.x / n; center.y = center.y / n; center.z = center.z / n;
    p.numvertices = n;
    p.centerPoint = center;
    p.normal = normal;
    return correct;
}

}  // namespace

extern "C" {

uint32_t rtx_abi_version(void) { return RTX_ABI_VERSION; }

const char* rtx_strerror(int code)
{
    switch (code) {
        case RTX_OK: return "ok";
        case RTX_E_INVALID_ARG: return "invalid argument";
        case RTX_E_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case RTX_E_HIP: return "HIP runtime error (see rtx_last_hip_error)";
        case RTX_E_OOM: return "out of memory";
        case RTX_E_IO: return "file I/O error";
        case RTX_E_TOO_LARGE: return "request exceeds a fixed limit";
        case RTX_E_RCCL: return "RCCL error (see rtx_last_hip_error)";
        default: return "unknown error";
    }
}

int rtx_scene_load(const char* path, rtx_scene** out)
{
    if (!path || !out) return RTX_E_INVALID_ARG;
    *out = nullptr;
    Reader r(path);
    if (!r.in.is_open()) return RTX_E_IO;
    OwnedScene* sc = new (std::nothrow) OwnedScene();
    if (!sc) return RTX_E_OOM;
    try {
    std::memset(&sc->view, 0, sizeof sc->view);  // the reference's globals are zero-initialised statics

    int numobjects = 0;
    r.in >> numobjects;                 // :855
    r.named_v3(sc->view.camera);        // :858
    r.named_v3(sc->view.vrp);           // :859
    r.named_v3(sc->view.vpn);           // :860
    r.named_v3(sc->view.up);            // :861
    r.named_v3(sc->view.ambient);       // :862
    int dropped_polygons = 0;
    for (int slot = 0; slot < numobjects; slot++) {  // :863
        std::string kind;
        r.in >> kind;
        // a stream that has failed stays failed: every later read is a no-op and every later slot matches no branch
        // (SURVEY.md section 4), so the remaining slots -- up to 2^31 of them for a corrupt count -- change nothing
        if (!r.in) break;
        if (kind == "sphere") {  // :866-885
            rtx_sphere s;
            std::memset(&s, 0, sizeof s);  // deterministic stand-in for the reference's indeterminate record (F5)
            r.v3(s.center);
            r.f(s.radius);
            r.v3(s.color);
            r.v3(s.diffcolor);
            r.v3(s.speccoeff);
            r.in >> s.reflectivity >> s.transparency >> s.refractive_index >> s.istransformed >> s.specexpo;
            if (s.istransformed) {
                r.v3(s.mat.a);
                r.v3(s.mat.b);
                r.v3(s.mat.c);
                r.v3(s.mat.d);
            }
            sc->spheres.push_back(s);
        } else if (kind == "plane") {  // :886-895 (stored, never traced)
            rtx_plane p;
            std::memset(&p, 0, sizeof p);
            r.v3(p.normal);
            r.v3(p.color);
            r.in >> p.d >> p.reflectivity >> p.transparency >> p.refractive_index;
            sc->planes.push_back(p);
        } else if (kind == "triangle") {  // :896-909
            rtx_triangle t;
            std::memset(&t, 0, sizeof t);
            r.v3(t.u);
            r.v3(t.v);
            r.v3(t.w);
            r.v3(t.color);
            r.v3(t.diffcolor);
            r.v3(t.speccoeff);
            r.in >> t.reflectivity >> t.transparency >> t.refractive_index >> t.specexpo;
            sc->triangles.push_back(t);
        } else if (kind == "polygon") {  // :910-967
            rtx_polygon p;
            std::memset(&p, 0, sizeof p);
            r.in >> p.numvertices;
            // the reference's `while (n--)` never ends for a negative count and allocates without bound for a
            // huge one: both are refused here instead
            if (p.numvertices < 0 || p.numvertices > (1 << 20)) { delete sc; return RTX_E_TOO_LARGE; }
            int n = p.numvertices;
            p.first_vertex = (int32_t)sc->polygon_vertices.size();
            rtx_vec3 center = {0.0f, 0.0f, 0.0f};
            rtx_vec3 normal = {0.0f, 0.0f, 0.0f};  // indeterminate in the reference below 3 vertices (F5 rule: zero)
            bool correct = true;
            const size_t v0 = sc->polygon_vertices.size();
            while (n--) {
                rtx_vec3 side = {0.0f, 0.0f, 0.0f};
                r.v3(side);
                center.x += side.x; center.y += side.y; center.z += side.z;
                sc->polygon_vertices.push_back(side);
                const rtx_vec3* vv = sc->polygon_vertices.data() + v0;
                const size_t cnt = sc->polygon_vertices.size() - v0;
                if (n == p.numvertices - 3) {  // :925-934
                    rtx_vec3 u = {vv[0].x - vv[1].x, vv[0].y - vv[1].y, vv[0].z - vv[1].z};
                    rtx_vec3 v = {vv[0].x - vv[2].x, vv[0].y - vv[2].y, vv[0].z - vv[2].z};
                    normal.x = u.y * v.z - u.z * v.y;
                    normal.y = u.z * v.x - u.x * v.z;
                    normal.z = u.x * v.y - u.y * v.x;
                    normalize_v(normal);
                }
                if (n < p.numvertices - 3) {   // :936-949: exact float equality of the two unit normals, or of one and the other's negation
                    rtx_vec3 u = {vv[0].x - vv[1].x, vv[0].y - vv[1].y, vv[0].z - vv[1].z};
                    rtx_vec3 v = {vv[0].x - vv[cnt - 1].x, vv[0].y - vv[cnt - 1].y, vv[0].z - vv[cnt - 1].z};
                    rtx_vec3 tn = {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
                    normalize_v(tn);
                    bool same = (normal.x == tn.x && normal.y == tn.y && normal.z == tn.z) ||
                                (-1 * normal.x == tn.x && -1 * normal.y == tn.y && -1 * normal.z == tn.z);
                    if (!same) correct = false;
                }
            }
            center.x = center.x / p.numvertices; center.y = center.y / p.numvertices; center.z = center.z / p.numvertices;  // :957
            p.centerPoint = center;
            p.normal = normal;
            r.v3(p.color);
            r.v3(p.diffcolor);
            r.v3(p.speccoeff);
            r.in >> p.reflectivity >> p.transparency >> p.refractive_index >> p.specexpo;
            if (correct) {
                sc->polygons.push_back(p);
            } else {  // :962: the record is consumed but not kept
                sc->polygon_vertices.resize(v0);
                dropped_polygons++;
            }
        } else if (kind == "light") {  // :968-976
            rtx_light l;
            std::memset(&l, 0, sizeof l);
            r.v3(l.point);
            r.v3(l.color);
            sc->lights.push_back(l);
        }
    }
    publish(sc, dropped_polygons);
    } catch (const std::bad_alloc&) {  // no exception crosses the C-ABI
        delete sc;
        return RTX_E_OOM;
    } catch (...) {
        delete sc;
        return RTX_E_IO;
    }
    *out = &sc->view;
    return RTX_OK;
}

// ---- opt-in: the OLDER grammars the bundled inp*.txt were written in ----------------------------------------------
// The five bundled scenes do not match parseinput's grammar at HEAD (they render as a grey frame or one
// flat disc).  Their trailing comment lines document what their authors wrote them for (inp.txt:15-17, inp-refl.txt:9-11,
// inpaffine.txt:20-22, inp1.txt:46-50, inp01.txt:36-39): ONE RECORD PER LINE, a keyword and numbers, with the later
// material fields added revision by revision -- and lines of different revisions live side by side in one file, so a
// record is read by its own length:
//   camera | vrp | vpn | up   x y z                      (all optional: the bundled camera, inp.txt:2-5, is the default)
//   light     x y z  r g b
//   sphere    x y z rad  r g b  [reflectivity [transparency [refractive_index [isAffine [12 matrix floats]]]]]
//   plane     a b c d    r g b  [reflectivity [transparency [refractive_index]]]
//   triangle  9 coords   r g b  [reflectivity [transparency [refractive_index]]]
//   polygon   n  3n coords  r g b  [reflectivity [transparency [refractive_index]]]
//   cuboid    ax ay az  bx by bz  r g b  [reflectivity [transparency [refractive_index]]]     (two opposite corners)
//   "//..." and blank lines are skipped; the leading object count is read and ignored (inp1.txt's is wrong).
// The HEAD records have material fields these grammars never had; this loader DEFINES them: diffcolor = color,
// speccoeff = (0.5, 0.5, 0.5), specexpo = 20, ambient = (0.05, 0.05, 0.05); missing trailing fields: reflectivity 0,
// transparency 0, refractive_index 1, isAffine 0.  A cuboid becomes 12 triangles.  No reference output exists for these
// scenes as their authors meant them (the bundled out.ppm is from an older, unavailable revision): the
// loader is checked structurally and its scenes render through the same kernels, compared with the oracle.
int rtx_scene_load_legacy(const char* path, rtx_scene** out)
{
    if (!path || !out) return RTX_E_INVALID_ARG;
    *out = nullptr;
    std::ifstream in(path);
    if (!in.is_open()) return RTX_E_IO;
    OwnedScene* sc = nullptr;
    try {
        sc = new OwnedScene();
        std::memset(&sc->view, 0, sizeof sc->view);
        sc->view.camera = {0.0f, 0.0f, -1.0f};
        sc->view.vrp = {0.0f, 0.0f, 0.0f};
        sc->view.vpn = {0.0f, 0.0f, -1.0f};
        sc->view.up = {0.0f, 1.0f, 0.0f};
        sc->view.ambient = {0.05f, 0.05f, 0.05f};
        int dropped = 0;
        std::string line;
        bool first = true;
        auto material = [](const std::vector<float>& v, size_t at, float& refl, float& transp, float& ri) {
            refl = v.size() > at ? v[at] : 0.0f;
            transp = v.size() > at + 1 ? v[at + 1] : 0.0f;
            ri = v.size() > at + 2 ? v[at + 2] : 1.0f;
        };
        const rtx_vec3 spec = {0.5f, 0.5f, 0.5f};
        auto add_triangle = [&](rtx_vec3 a, rtx_vec3 b, rtx_vec3 c, rtx_vec3 col, float refl, float transp, float ri) {
            rtx_triangle t;
            std::memset(&t, 0, sizeof t);
            t.u = a; t.v = b; t.w = c;
            t.color = col; t.diffcolor = col; t.speccoeff = spec; t.specexpo = 20.0f;
            t.reflectivity = refl; t.transparency = transp; t.refractive_index = ri;
            sc->triangles.push_back(t);
        };
        while (std::getline(in, line)) {
            std::istringstream ls(line);
            std::string kind;
            if (!(ls >> kind)) continue;                       // blank
            if (kind.compare(0, 2, "//") == 0) continue;       // comment
            std::vector<float> v;
            if (kind == "polygon") {
                // the vertex count is an int token; everything after it are floats
                int n = 0;
                if (!(ls >> n) || n < 0 || n > (1 << 20)) continue;
                v.push_back((float)n);
            }
            for (float x; ls >> x;) v.push_back(x);
            if (first && v.empty()) {                          // the object count: a lone integer on the first record line
                char* end = nullptr;
                (void)std::strtol(kind.c_str(), &end, 10);
                first = false;
                if (end && *end == 0) continue;
            }
            first = false;
            auto v3 = [&](size_t at) { rtx_vec3 r = {v[at], v[at + 1], v[at + 2]}; return r; };
            if ((kind == "camera" || kind == "vrp" || kind == "vpn" || kind == "up") && v.size() >= 3) {
                rtx_vec3& dst = kind == "camera" ? sc->view.camera : kind == "vrp" ? sc->view.vrp : kind == "vpn" ? sc->view.vpn : sc->view.up;
                dst = v3(0);
            } else if (kind == "light" && v.size() >= 6) {
                rtx_light l;
                l.point = v3(0); l.color = v3(3);
                sc->lights.push_back(l);
            } else if (kind == "sphere" && v.size() >= 7) {
                rtx_sphere s;
                std::memset(&s, 0, sizeof s);
                s.center = v3(0); s.radius = v[3];
                s.color = v3(4); s.diffcolor = s.color; s.speccoeff = spec; s.specexpo = 20.0f;
                material(v, 7, s.reflectivity, s.transparency, s.refractive_index);
                s.istransformed = (v.size() > 10 && v[10] != 0.0f && v.size() >= 23) ? 1 : 0;
                if (s.istransformed) { s.mat.a = v3(11); s.mat.b = v3(14); s.mat.c = v3(17); s.mat.d = v3(20); }
                sc->spheres.push_back(s);
            } else if (kind == "plane" && v.size() >= 7) {
                rtx_plane p;
                std::memset(&p, 0, sizeof p);
                p.normal = v3(0); p.d = v[3]; p.color = v3(4);
                material(v, 7, p.reflectivity, p.transparency, p.refractive_index);
                sc->planes.push_back(p);
            } else if (kind == "triangle" && v.size() >= 12) {
                float refl, transp, ri;
                material(v, 12, refl, transp, ri);
                add_triangle(v3(0), v3(3), v3(6), v3(9), refl, transp, ri);
            } else if (kind == "polygon" && !v.empty()) {
                const int n = (int)v[0];
                if (n < 1 || v.size() < (size_t)(1 + 3 * n + 3)) continue;
                std::vector<rtx_vec3> vv;
                for (int i = 0; i < n; i++) vv.push_back(v3(1 + 3 * (size_t)i));
                rtx_polygon p;
                std::memset(&p, 0, sizeof p);
                const size_t at = 1 + 3 * (size_t)n;
                p.color = v3(at); p.diffcolor = p.color; p.speccoeff = spec; p.specexpo = 20.0f;
                material(v, at + 3, p.reflectivity, p.transparency, p.refractive_index);
                p.first_vertex = (int32_t)sc->polygon_vertices.size();
                if (finish_polygon(vv, p)) {
                    sc->polygon_vertices.insert(sc->polygon_vertices.end(), vv.begin(), vv.end());
                    sc->polygons.push_back(p);
                } else {
                    dropped++;
                }
            } else if (kind == "cuboid" && v.size() >= 9) {
                const rtx_vec3 a = v3(0), b = v3(3), col = v3(6);
                float refl, transp, ri;
                material(v, 9, refl, transp, ri);
                rtx_vec3 c[8];
                for (int i = 0; i < 8; i++) c[i] = {(i & 1) ? b.x : a.x, (i & 2) ? b.y : a.y, (i & 4) ? b.z : a.z};
                static const int quad[6][4] = {{0, 1, 3, 2}, {4, 6, 7, 5}, {0, 4, 5, 1}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 5, 7, 3}};
                for (const auto& q : quad) {
                    add_triangle(c[q[0]], c[q[1]], c[q[2]], col, refl, transp, ri);
                    add_triangle(c[q[0]], c[q[2]], c[q[3]], col, refl, transp, ri);
                }
            }
            // anything else (an unknown keyword, a record too short for its kind) is skipped
        }
        publish(sc, dropped);
    } catch (const std::bad_alloc&) {
        delete sc;
        return RTX_E_OOM;
    } catch (...) {
        delete sc;
        return RTX_E_IO;
    }
    *out = &sc->view;
    return RTX_OK;
}

void rtx_scene_free(rtx_scene* scene)
{
    if (scene) delete reinterpret_cast<OwnedScene*>(scene);
}

int rtx_write_ppm(const char* path, int32_t width, int32_t height, const uint8_t* rgb)
{
    if (!path || !rgb || width <= 0 || height <= 0) return RTX_E_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return RTX_E_IO;
    std::fprintf(f, "P6\n%i %i 255\n", width, height);  // :787
    size_t n = (size_t)width * (size_t)height * 3;
    size_t w = std::fwrite(rgb, 1, n, f);
    int rc = std::fclose(f);
    return (w == n && rc == 0) ? RTX_OK : RTX_E_IO;
}

int rtx_band_rows(int32_t height, int32_t nranks, int32_t rank, int32_t band_rows,
                  int32_t* rows_out, int32_t capacity)
{
    if (height <= 0) return RTX_E_INVALID_ARG;
    if (nranks <= 1) { nranks = 1; rank = 0; }
    if (rank < 0 || rank >= nranks) return RTX_E_INVALID_ARG;
    if (band_rows <= 0) band_rows = (height + nranks - 1) / nranks;
    int n = 0;
    for (int y = 0; y < height; y++) {
        if ((y / band_rows) % nranks != rank) continue;
        if (rows_out) {
            if (n >= capacity) return RTX_E_INVALID_ARG;
            rows_out[n] = y;
        }
        n++;
    }
    return n;
}

}  // extern "C"
