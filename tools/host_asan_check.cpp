// Sanitizer harness for the HOST side of the library ("new host code compiled with ASan"): the two
// scene loaders (rtx_host.cpp: untrusted text in two grammars), scene validation, packing and the k-d cluster build
// (rtx_pack.cpp, rtx_bounds.cpp), the band arithmetic and the PPM writer -- compiled from the very sources of librtx.so
// with -fsanitize=address,undefined (make -C ray-tracing_amd/csrc asan) and run on every scene file given on the command
// line plus N seeded mutations of each (truncations, swapped / corrupted tokens, huge and negative counts, NaN / inf
// literals, raw bytes).  No device code is involved; exit status 0 = every input was handled without a report.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <unistd.h>

#include "rtx.h"
#include "rtx_pack.h"

static uint64_t g_state = 0x5EED2026ull;
static uint64_t rnd()
{
    g_state += 0x9e3779b97f4a7c15ull;
    uint64_t z = g_state;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static std::string mutate(const std::string& in)
{
    std::string s = in;
    static const char* junk[] = {"nan", "inf", "-inf", "1e39", "-1e39", "2147483647", "-2147483648", "4294967296", "-1", "0",
                                 "sphere", "triangle", "polygon", "plane", "light", "cuboid", "1e-45", "0x10", "", "\n"};
    const int edits = 1 + (int)(rnd() % 6);
    for (int e = 0; e < edits && !s.empty(); e++) {
        const size_t pos = rnd() % s.size();
        switch (rnd() % 6) {
        case 0: s.resize(pos); break;                                                  // truncate
        case 1: s[pos] = (char)(rnd() & 0xff); break;                                  // raw byte
        case 2: {                                                                      // replace the token at pos
            size_t a = s.find_last_of(" \n\t", pos), b = s.find_first_of(" \n\t", pos);
            a = (a == std::string::npos) ? 0 : a + 1;
            if (b == std::string::npos) b = s.size();
            s.replace(a, b - a, junk[rnd() % (sizeof junk / sizeof *junk)]);
            break;
        }
        case 3: s.insert(pos, std::string(" ") + junk[rnd() % (sizeof junk / sizeof *junk)] + " "); break;
        case 4: s.erase(pos, 1 + rnd() % 16); break;
        default: s.insert(pos, s.substr(rnd() % s.size(), rnd() % 64)); break;         // duplicate a run
        }
    }
    return s;
}

static void exercise(const rtx_scene* sc)
{
    if (rtx::validate_scene(sc) != RTX_OK) return;
    if ((int64_t)sc->nspheres + sc->ntriangles + sc->npolygons > 200000) return;  // (keeps a fuzzed count from costing minutes)
    for (uint32_t flags : {0u, (uint32_t)RTX_F_TRACE_PLANES, (uint32_t)RTX_F_TRI_MOLLER_TRUMBORE}) {
        rtx::PackedScene P;
        rtx::pack_scene(*sc, 64, 48, P, flags);
    }
    const int n = rtx_filter_records(sc, nullptr, 0, nullptr);
    if (n > 0) {
        std::vector<float> rec((size_t)n * 4);
        float guard = 0;
        rtx_filter_records(sc, rec.data(), n, &guard);
    }
    for (int which = 0; which < 2; which++) {
        const int ncl = rtx_cluster_records(sc, which, nullptr, nullptr, nullptr, nullptr, 0);
        if (ncl > 0) {
            std::vector<float> c4((size_t)ncl * 4), s4((size_t)ncl * 32), m4((size_t)ncl * 256);
            std::vector<int32_t> idx((size_t)ncl * 64);
            rtx_cluster_records(sc, which, c4.data(), s4.data(), m4.data(), idx.data(), ncl);
        }
    }
    const int na = rtx_affine_shadow_records(sc, nullptr, 0);
    if (na > 0) { std::vector<float> a((size_t)na * 16); rtx_affine_shadow_records(sc, a.data(), na); }
    const int nt = rtx_triangle_edge_records(sc, nullptr, 0);
    if (nt > 0) { std::vector<float> t((size_t)nt * 8); rtx_triangle_edge_records(sc, t.data(), nt); }
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: host_asan_check <mutations per file> <scene.txt>...\n"); return 2; }
    const int nmut = std::atoi(argv[1]);
    const std::string tmp = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/rtx_asan_" + std::to_string((long)getpid()) + ".txt";
    long loaded = 0, refused = 0;
    for (int a = 2; a < argc; a++) {
        std::ifstream f(argv[a], std::ios::binary);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string text = ss.str();
        for (int m = 0; m <= nmut; m++) {
            const std::string body = m ? mutate(text) : text;
            { std::ofstream o(tmp, std::ios::binary); o << body; }
            for (int legacy = 0; legacy < 2; legacy++) {
                rtx_scene* sc = nullptr;
                const int rc = legacy ? rtx_scene_load_legacy(tmp.c_str(), &sc) : rtx_scene_load(tmp.c_str(), &sc);
                if (rc == RTX_OK && sc) { loaded++; exercise(sc); rtx_scene_free(sc); }
                else refused++;
            }
        }
    }
    std::remove(tmp.c_str());
    // band arithmetic and the PPM writer
    for (int h : {0, 1, 7, 480, 2160})
        for (int n : {0, 1, 3, 8, 5000})
            for (int b : {-1, 0, 1, 16, 100000}) {
                std::vector<int32_t> rows(h > 0 ? h : 1);
                for (int r = 0; r < (n > 0 ? n : 1) && r < 9; r++) rtx_band_rows(h, n, r, b, rows.data(), (int32_t)rows.size());
            }
    {
        std::vector<uint8_t> px(5 * 3 * 3, 7);
        rtx_write_ppm((tmp + ".ppm").c_str(), 5, 3, px.data());
        std::remove((tmp + ".ppm").c_str());
        rtx_write_ppm("/nonexistent-dir/x.ppm", 5, 3, px.data());
    }
    std::printf("host_asan_check: %ld scenes loaded and packed, %ld inputs refused, no sanitizer report\n", loaded, refused);
    return 0;
}
