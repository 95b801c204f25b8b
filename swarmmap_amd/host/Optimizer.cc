// Optimizer.cc — see Optimizer.h.
#include "Optimizer.h"

#include <stdexcept>
#include <string>

namespace ORB_SLAM2 {

static void check(int status, const char* what) {
    if (status != SO_OK)
        throw std::runtime_error(std::string(what) + ": " + so_status_string(status) + " (" + so_last_error() + ")");
}

Optimizer::Optimizer(int device) { check(so_ba_create(device, &handle_), "so_ba_create"); }
Optimizer::~Optimizer() { so_ba_destroy(handle_); }

namespace {
constexpr int kMaxDevices = 16;
thread_local Optimizer* t_instance[kMaxDevices] = {};
}  // namespace

Optimizer& Optimizer::ThreadInstance(int device) {
    if (device < 0 || device >= kMaxDevices) throw std::runtime_error("Optimizer: device index out of range");
    if (!t_instance[device]) t_instance[device] = new Optimizer(device);
    return *t_instance[device];
}

void Optimizer::ReleaseThreadInstance() {
    for (Optimizer*& o : t_instance) {
        delete o;
        o = nullptr;
    }
}

void Optimizer::solve(const BAWindow& w, const so_ba_options& opt, bool* pbStopFlag, BAResult& out) {
    static_assert(sizeof(bool) == 1, "pbStopFlag is polled as a byte");
    so_ba_problem p{};
    p.n_poses = (int32_t)w.fixed.size();
    p.Tcw = w.Tcw.data();
    p.fixed = w.fixed.data();
    p.intr = w.intr.data();
    p.n_points = (int32_t)(w.Xw.size() / 3);
    p.Xw = w.Xw.data();
    p.n_edges = (int32_t)w.edge_kf.size();
    p.edge_pose = w.edge_kf.data();
    p.edge_point = w.edge_mp.data();
    p.obs = w.obs.data();
    p.inv_sigma2 = w.inv_sigma2.data();
    out.Tcw.assign(w.Tcw.size(), 0.f);
    out.Xw.assign(w.Xw.size(), 0.f);
    out.edge_outlier.assign(w.edge_kf.size(), 0);
    check(so_bundle_adjust(handle_, &p, &opt, reinterpret_cast<const volatile uint8_t*>(pbStopFlag), out.Tcw.data(),
                           out.Xw.data(), out.edge_outlier.data(), nullptr, &out.info),
          "so_bundle_adjust");
}

void Optimizer::LocalBundleAdjustment(const BAWindow& window, bool* pbStopFlag, BAResult& out) {
    so_ba_options opt;
    so_ba_options_local(&opt);
    solve(window, opt, pbStopFlag, out);
}

void Optimizer::BundleAdjustment(const BAWindow& map, int nIterations, bool* pbStopFlag, bool bRobust, BAResult& out) {
    so_ba_options opt;
    so_ba_options_global(&opt, nIterations, bRobust ? 1 : 0);
    solve(map, opt, pbStopFlag, out);
}

int Optimizer::PoseOptimization(float Tcw[12], const float K[4], const std::vector<float>& X, const std::vector<float>& obs,
                                const std::vector<float>& w, std::vector<uint8_t>& outl) {
    const int n = (int)w.size();
    outl.assign((size_t)n, 0);
    float Tout[12];
    int32_t inliers = 0, info[2] = {0, 0};
    check(so_pose_optimization(handle_, Tcw, K, n, X.data(), obs.data(), w.data(), Tout, outl.data(), &inliers, info),
          "so_pose_optimization");
    if (n >= 3) for (int i = 0; i < 12; i++) Tcw[i] = Tout[i];  // pFrame->SetPose(pose)
    return inliers;
}

}  // namespace ORB_SLAM2
