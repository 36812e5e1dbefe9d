// rt_render_main.cpp -- headless stand-in for the reference's src/main.cpp: initialize(), then the 'T' key
// (main.cpp:69-70 -> Flyscene::raytraceScene()).  Reads the same two stdin switches (flyscene.cpp:31-34).
//   usage: rt_render [--scene path.obj] [--size W H] [--samples U V] [--depth D] [--out result.ppm]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "flyscene.hpp"

int main(int argc, char **argv) {
    int w = 1000, h = 1000;                    // WINDOW_WIDTH / WINDOW_HEIGHT, main.cpp:8-9
    rtamd::Flyscene scene;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--scene") && i + 1 < argc) scene.setScenePath(argv[++i]);
        else if (!std::strcmp(argv[i], "--size") && i + 2 < argc) { w = std::atoi(argv[++i]); h = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--samples") && i + 2 < argc) { const int u = std::atoi(argv[++i]); scene.setAreaGrid(u, std::atoi(argv[++i])); }
        else if (!std::strcmp(argv[i], "--depth") && i + 1 < argc) scene.setMaxDepth(std::atoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) scene.setOutputPath(argv[++i]);
        else { std::fprintf(stderr, "usage: %s [--scene obj] [--size W H] [--samples U V] [--depth D] [--out ppm]\n", argv[0]); return 2; }
    }
    if (w <= 0 || h <= 0) return 2;
    scene.initialize(w, h);
    scene.raytraceScene();
    if (scene.lastStatus() != RT_OK) return 1;          // no result.ppm was written: say so with the exit code
    const rt_stats &st = scene.lastStats();
    std::printf("device ms: trace %.3f shadow %.3f shade %.3f resolve %.3f total %.3f\n", st.ms_trace, st.ms_shadow, st.ms_shade, st.ms_resolve, st.ms_total);
    return 0;
}
