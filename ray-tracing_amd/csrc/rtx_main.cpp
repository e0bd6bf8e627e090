// rtx_main.cpp -- drop-in command line for the reference's `a.out <scene.txt> <out.ppm>`
// (raytracing.cpp:981-995): same two positional arguments, same usage text and exit status on a
// wrong argument count, same PPM bytes.  The per-pixel render loop (:992 render2Dantialiasing)
// is the one call into the HIP library.  Optional flags extend, never change, the default run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "rtx.h"

int main(int argc, char** argv)
{
    rtx_params p;
    std::memset(&p, 0, sizeof p);
    p.width = 640;       // raytracing.cpp:112
    p.height = 480;      // raytracing.cpp:113
    p.max_depth = 4;     // raytracing.cpp:12
    p.device = -1;
    int ngpus = 1;
    bool legacy = false;
    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        auto val = [&](const char* name) -> const char* {
            size_t n = std::strlen(name);
            if (std::strncmp(argv[i], name, n) == 0 && argv[i][n] == '=') return argv[i] + n + 1;
            return nullptr;
        };
        const char* v;
        if ((v = val("--width"))) p.width = std::atoi(v);
        else if ((v = val("--height"))) p.height = std::atoi(v);
        else if ((v = val("--max-depth"))) p.max_depth = std::atoi(v);
        else if ((v = val("--gpus"))) ngpus = std::atoi(v);
        else if ((v = val("--band-rows"))) p.band_rows = std::atoi(v);
        else if (std::strcmp(argv[i], "--legacy-grammar") == 0) legacy = true;          // opt-in: rtx_scene_load_legacy
        else if (std::strcmp(argv[i], "--trace-planes") == 0) p.flags |= RTX_F_TRACE_PLANES;   // opt-in extension
        else if (std::strcmp(argv[i], "--moller-trumbore") == 0) p.flags |= RTX_F_TRI_MOLLER_TRUMBORE;  // opt-in, non-parity
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 2) {  // :983-988
        std::cout << "Please follow the specifications!\nUnexpected arguments given.\n";
        std::cout << "Please run as: a.exe <inputfile>\n";
        std::exit(0);
    }
    rtx_scene* scene = nullptr;
    int rc = legacy ? rtx_scene_load_legacy(pos[0], &scene) : rtx_scene_load(pos[0], &scene);
    if (rc) { std::fprintf(stderr, "rtx: cannot read %s: %s\n", pos[0], rtx_strerror(rc)); return 1; }
    std::vector<uint8_t> rgb((size_t)p.width * p.height * 3);
    rtx_stats st;
    rc = rtx_render_rgb8(scene, &p, ngpus, rgb.data(), &st);
    if (rc) {
        std::fprintf(stderr, "rtx: render failed: %s %s\n", rtx_strerror(rc), rtx_last_hip_error());
        rtx_scene_free(scene);
        return 1;
    }
    rc = rtx_write_ppm(pos[1], p.width, p.height, rgb.data());
    rtx_scene_free(scene);
    if (rc) { std::fprintf(stderr, "rtx: cannot write %s\n", pos[1]); return 1; }
    return 0;
}
