// qt_probe.hip - where the device quadtree spends its time: quadtree_kernel built with -DQT_TIMING on a synthetic
// 752x480 pyramid (random keep bitmap of a given density, random scores), shader-clock stamps of the level-0 workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQT_TIMING -I swarmmap_amd/csrc tools/probe/qt_probe.hip -o tools/probe/qt_probe_bin
//   tools/probe/qt_probe_bin [density per mille, default 12] [nfeatures, default 1000]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "quadtree_kernel.hip"

using namespace so;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int density = argc > 1 ? atoi(argv[1]) : 12;
    const int nfeatures = argc > 2 ? atoi(argv[2]) : 1000;
    const int w = 752, h = 480, nl = 8;
    PyramidParams P{};
    P.nlevels = nl;
    P.th_high = 20;
    P.th_low = 7;
    std::mt19937 rng(7);
    int tile_base = 0, row_base = 0;
    for (int l = 0; l < nl; l++) {
        LevelDesc& L = P.lv[l];
        const float inv = 1.0f / powf(1.2f, (float)l);
        L.w = (int)lrintf((float)w * inv);
        L.h = (int)lrintf((float)h * inv);
        const int rw = L.w - 2 * kFastBorder, rh = L.h - 2 * kFastBorder;
        L.ntx = (rw - 6 + kTile - 1) / kTile;
        L.nty = (rh - 6 + kTile - 1) / kTile;
        L.tile_base = tile_base;
        L.row_base = row_base;
        tile_base += L.ntx * L.nty;
        row_base += L.nty * kTile;
        const int nt = L.ntx * L.nty;
        std::vector<uint32_t> bm((size_t)nt * kTile, 0u);
        std::vector<uint8_t> sc((size_t)nt * kScoreBlock);
        for (auto& v : sc) v = (uint8_t)(rng() % 200 + 7);
        for (int t = 0; t < nt; t++) {
            const int ty = t / L.ntx, tx = t % L.ntx;
            for (int r = 0; r < kTile; r++)
                for (int c = 0; c < kTile; c++) {
                    const int x = 32 * tx + c, y = 32 * ty + r;  // ROI pixel - 3
                    if (x >= rw - 6 || y >= rh - 6) continue;
                    if ((int)(rng() % 1000) < density) bm[(size_t)t * kTile + r] |= 1u << c;
                }
        }
        CK(hipMalloc((void**)&L.bitmap, bm.size() * 4));
        CK(hipMalloc((void**)&L.score, sc.size()));
        CK(hipMemcpy(L.bitmap, bm.data(), bm.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(L.score, sc.data(), sc.size(), hipMemcpyHostToDevice));
    }
    P.total_tiles = tile_base;
    P.total_rows = row_base;
    int n_target[8];
    {
        const float factor = 1.0f / 1.2f;
        float desired = (float)nfeatures * (1 - factor) / (1 - powf(factor, (float)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) {
            n_target[l] = (int)lrintf(desired);
            sum += n_target[l];
            desired *= factor;
        }
        n_target[nl - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
    const int stride = 1024;
    SelectedKp* d_sel;
    int32_t* d_count;
    CK(hipMalloc((void**)&d_sel, sizeof(SelectedKp) * stride * nl));
    CK(hipMalloc((void**)&d_count, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int it = 0; it < 5; it++) launch_quadtree(P, n_target, stride, d_sel, d_count, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, nullptr));
    for (int it = 0; it < 20; it++) launch_quadtree(P, n_target, stride, d_sel, d_count, nullptr);
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long st[96];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(qt_stamps), sizeof(st)));
    int32_t cnt[8];
    CK(hipMemcpy(cnt, d_count, 32, hipMemcpyDeviceToHost));
    printf("kernel %.2f us per launch (20 back to back); level 0: n = %lld keys, N = %d, nodes out = %lld (count %d)\n", ms * 50.0, st[90],
           n_target[0], st[7], cnt[0]);
    // s_memtime ticks with the shader clock: scale the stamps so that the whole workgroup = the measured launch time
    const double tick_us = (double)ms * 50.0 / (double)(st[6] - st[0]);
    auto us = [&](int a, int b) { return (double)(st[b] - st[a]) * tick_us; };
    printf("word cache %.2f | count+scan %.2f | keys %.2f | roots %.2f | steps %.2f | best+out %.2f | total %.2f us\n", us(0, 1), us(1, 2),
           us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(0, 6));
    for (int g = 0; g < 20; g++) {
        if (st[8 + 4 * g + 3] <= 0 || (g > 0 && st[8 + 4 * g + 3] < st[8 + 4 * (g - 1) + 3])) break;
        const long long prev = g == 0 ? st[4] : st[8 + 4 * (g - 1) + 3];
        printf("  step %2d: quadrants+scan %.2f | order %.2f | nodes %.2f | partition %.2f\n", g, (double)(st[8 + 4 * g] - prev) * tick_us,
               us(8 + 4 * g, 8 + 4 * g + 1), us(8 + 4 * g + 1, 8 + 4 * g + 2), us(8 + 4 * g + 2, 8 + 4 * g + 3));
    }
    return 0;
}
