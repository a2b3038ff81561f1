#include "qgd_mesh.hpp"
#include "qgd_setup.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace qgd;
int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 128;
    const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    const int32_t pt[6] = {0, 0, 0, 0, 0, 0};
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* w) { auto t = std::chrono::steady_clock::now(); std::printf("%-40s %8.2f s\n", w, std::chrono::duration<double>(t - t0).count()); t0 = t; };
    HostMesh m = makeBox(n, n, n, 0, n, lo, hi, pt); lap("makeBox");
    StaticData s = buildStaticData(m); lap("buildStaticData");
    FaceTiles t = buildFaceTiles(s, 128); lap("buildFaceTiles");
    FusedBlocks b = buildFusedBlocks(s); lap("buildFusedBlocks");
    std::printf("%d blocks\n", b.nBlocks);
    return 0;
}
