// rtx_host.cpp -- host-side pieces kept from the reference, behind the C-ABI (include/rtx.h):
// scene-file reader (grammar + stream semantics of parseinput, raytracing.cpp:851-979),
// PPM writer (:786-797), band arithmetic, error strings.  No device code here.
#include "rtx.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
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
};

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
        } else if (kind == "polygon") {  // :910-967: tokens consumed so later records line up; record dropped
            int nv = 0;
            r.in >> nv;
            rtx_vec3 skip;
            for (int k = 0; k < nv && r.in; k++) r.v3(skip);
            float m;
            for (int k = 0; k < 13; k++) r.f(m);
            dropped_polygons++;
        } else if (kind == "light") {  // :968-976
            rtx_light l;
            std::memset(&l, 0, sizeof l);
            r.v3(l.point);
            r.v3(l.color);
            sc->lights.push_back(l);
        }
    }
    sc->view.spheres = sc->spheres.empty() ? nullptr : sc->spheres.data();
    sc->view.nspheres = (int32_t)sc->spheres.size();
    sc->view.triangles = sc->triangles.empty() ? nullptr : sc->triangles.data();
    sc->view.ntriangles = (int32_t)sc->triangles.size();
    sc->view.planes = sc->planes.empty() ? nullptr : sc->planes.data();
    sc->view.nplanes = (int32_t)sc->planes.size();
    sc->view.lights = sc->lights.empty() ? nullptr : sc->lights.data();
    sc->view.nlights = (int32_t)sc->lights.size();
    sc->view.npolygons_ignored = dropped_polygons;
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
