// adapter_smoke.cpp — exercises the C++ adapter classes (the reference's signatures) end to end on the GPU and
// prints checksums that tests/test_cpp_adapters_gpu.py compares with the same inputs through the Python binding.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../swarmmap_amd/host/ORBextractor.h"
#include "../../swarmmap_amd/host/ORBmatcher.h"
#include "../../swarmmap_amd/host/Frame.h"
#include "../../swarmmap_amd/host/Optimizer.h"

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    std::vector<uint8_t> img((size_t)w * h);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    fclose(f);
    ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<swarmorb::KeyPoint> kps;
    swarmorb::Descriptors desc;
    swarmorb::ImageView view{img.data(), w, h, w}, mask;
    ex(view, mask, kps, desc);
    ex(view, mask, kps, desc);  // the context is reused frame after frame
    printf("levels %d scale7 %.6f\n", ex.GetLevels(), ex.GetScaleFactors()[7]);
    printf("keypoints %zu %016llx\n", kps.size(), (unsigned long long)fnv(kps.data(), kps.size() * sizeof(kps[0])));
    printf("descriptors %d %016llx\n", desc.rows, (unsigned long long)fnv(desc.data.data(), desc.data.size()));
    printf("distance %d\n", ORB_SLAM2::ORBmatcher::DescriptorDistance(desc.ptr(0), desc.ptr(1)));
    // match the frame against itself shifted by one pixel: every level-0 keypoint must find its twin
    std::vector<float> x(kps.size()), y(kps.size()), ang(kps.size());
    std::vector<int32_t> oct(kps.size());
    for (size_t i = 0; i < kps.size(); i++) {
        x[i] = kps[i].pt.x; y[i] = kps[i].pt.y; ang[i] = kps[i].angle; oct[i] = kps[i].octave;
    }
    std::vector<float> sf = ex.GetScaleFactors();
    so_frame_view F{};
    F.n = (int32_t)kps.size(); F.x = x.data(); F.y = y.data(); F.octave = oct.data(); F.angle = ang.data();
    F.desc = desc.data.data(); F.min_x = 0; F.max_x = (float)w; F.min_y = 0; F.max_y = (float)h;
    F.grid_inv_w = 64.0f / (float)w; F.grid_inv_h = 48.0f / (float)h; F.scale_factors = sf.data(); F.nlevels = 8;
    ORB_SLAM2::ORBmatcher matcher(0.9f, true);
    std::vector<float> prev(2 * kps.size());
    for (size_t i = 0; i < kps.size(); i++) { prev[2 * i] = x[i] + 1.0f; prev[2 * i + 1] = y[i]; }
    std::vector<int32_t> m12;
    const int nm = matcher.SearchForInitialization(F, F, prev, m12, 20);
    int self = 0, lvl0 = 0;
    for (size_t i = 0; i < m12.size(); i++) { self += m12[i] == (int)i; lvl0 += oct[i] == 0; }
    printf("init_matches %d self %d level0 %d\n", nm, self, lvl0);
    // a tiny BA window: 2 keyframes (one fixed), 4 points seen by both, consistent observations
    ORB_SLAM2::BAWindow win;
    const float T0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, T1[12] = {1, 0, 0, -0.5f, 0, 1, 0, 0, 0, 0, 1, 0};
    win.Tcw.assign(T0, T0 + 12); win.Tcw.insert(win.Tcw.end(), T1, T1 + 12);
    win.fixed = {1, 0};
    win.intr = {458.654f, 457.296f, 367.215f, 248.375f, 458.654f, 457.296f, 367.215f, 248.375f};
    const float P[4][3] = {{-1, -0.5f, 5}, {1, 0.5f, 6}, {0.5f, -1, 4}, {-0.7f, 0.8f, 7}};
    for (int j = 0; j < 4; j++) {
        for (int k = 0; k < 3; k++) win.Xw.push_back(P[j][k] + 0.05f * (float)(j - k));
        for (int i = 0; i < 2; i++) {
            const float tx = i == 0 ? 0.f : -0.5f;
            const float u = 458.654f * (P[j][0] + tx) / P[j][2] + 367.215f, v = 457.296f * P[j][1] / P[j][2] + 248.375f;
            win.edge_kf.push_back(i); win.edge_mp.push_back(j); win.obs.push_back(u); win.obs.push_back(v);
            win.inv_sigma2.push_back(1.0f);
        }
    }
    ORB_SLAM2::Optimizer& optimizer = ORB_SLAM2::Optimizer::ThreadInstance();  // static call sites, as in the reference
    ORB_SLAM2::BAResult res;
    bool stop = false;
    optimizer.LocalBundleAdjustment(win, &stop, res);
    printf("ba chi2 %.3e -> %.3e its %d+%d outliers %d\n", res.info.chi2_initial, res.info.chi2_final,
           res.info.iterations_stage1, res.info.iterations_stage2, res.info.n_outliers);
    // Frame post-processing on the extracted keypoints with EuRoC's lens
    const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
    ORB_SLAM2::Frame frame(K, {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f});
    std::vector<swarmorb::KeyPoint> keysUn;
    frame.UndistortAndAssign(kps, w, h, keysUn);
    std::vector<float> un(2 * keysUn.size());
    for (size_t i = 0; i < keysUn.size(); i++) { un[2 * i] = keysUn[i].pt.x; un[2 * i + 1] = keysUn[i].pt.y; }
    printf("undistorted %zu %016llx bounds %.9g %.9g %.9g %.9g grid %zu %016llx\n", keysUn.size(),
           (unsigned long long)fnv(un.data(), un.size() * 4), frame.mnMinX, frame.mnMaxX, frame.mnMinY, frame.mnMaxY,
           frame.GridItems().size(), (unsigned long long)fnv(frame.GridItems().data(), frame.GridItems().size() * 4));
    // PoseOptimization on exact observations of the 4 BA points + 4 more from a perturbed pose
    std::vector<float> X, ob, iw;
    std::vector<uint8_t> outl;
    for (int j = 0; j < 40; j++) {
        const float px = -2.f + 0.1f * (float)j, py = -1.f + 0.05f * (float)((j * 7) % 40), pz = 4.f + 0.1f * (float)((j * 3) % 30);
        X.insert(X.end(), {px, py, pz});
        ob.push_back(K[0] * px / pz + K[2]);
        ob.push_back(K[1] * py / pz + K[3]);
        iw.push_back(1.0f);
    }
    float Tcw[12] = {1, 0, 0, 0.05f, 0, 1, 0, -0.03f, 0, 0, 1, 0.02f};  // start off the true identity pose
    const int inl = optimizer.PoseOptimization(Tcw, K, X, ob, iw, outl);
    printf("pose inliers %d t %.4f %.4f %.4f\n", inl, Tcw[3], Tcw[7], Tcw[11]);
    ORB_SLAM2::DistinctiveDescriptors dd;
    std::vector<int32_t> off = {0, 3, 3, 8};
    std::vector<uint8_t> dsc(desc.data.begin(), desc.data.begin() + 8 * 32);
    const std::vector<int32_t> best = dd.Compute(off, dsc);
    printf("distinctive %d %d %d\n", best[0], best[1], best[2]);
    return 0;
}
