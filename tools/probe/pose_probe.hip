// Phase-level cycle accounting of the LDS-resident PoseOptimization kernel (developer tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I swarmmap_amd/csrc tools/probe/pose_probe.hip -o /tmp/pose_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long* g_ticks;
#define SO_POSE_TICK_DECL long long so_tk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long so_t_ = clock64(); int so_np_ = 0
#define SO_POSE_TICK(i) do { const long long now_ = clock64(); so_tk_[i] += now_ - so_t_; so_t_ = now_; if ((i) == 0) so_np_++; } while (0)
#define SO_POSE_TICK_FLUSH do { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 8; q_++) g_ticks[q_] = so_tk_[q_]; g_ticks[8] = so_np_; } } while (0)
#include "ba_kernels.hip"
#include "ba_dense.hip"
using namespace so;
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 500;
    srand(7);
    auto rnd = []() { return rand() / (double)RAND_MAX; };
    std::vector<float> X(3 * n), obs(2 * n), w(n);
    const double K[4] = {458.654, 457.296, 367.215, 248.375};
    for (int i = 0; i < n; i++) {
        const double x = (rnd() - 0.5) * 6, y = (rnd() - 0.5) * 4, z = 3 + 6 * rnd();
        X[3 * i] = (float)x; X[3 * i + 1] = (float)y; X[3 * i + 2] = (float)z;
        double u = K[0] * x / z + K[2] + (rnd() - 0.5) * 2, v = K[1] * y / z + K[3] + (rnd() - 0.5) * 2;
        if (rnd() < 0.1) { u += (rnd() - 0.5) * 80; v += (rnd() - 0.5) * 80; }
        obs[2 * i] = (float)u; obs[2 * i + 1] = (float)v; w[i] = 1.0f;
    }
    float *dX, *dobs, *dw; uint8_t* dout; BaPose* dpose; int* dinfo; double* derr; long long* dt;
    hipMalloc(&dX, 12 * n); hipMalloc(&dobs, 8 * n); hipMalloc(&dw, 4 * n); hipMalloc(&dout, n); hipMalloc(&dpose, sizeof(BaPose));
    hipMalloc(&dinfo, 16); hipMalloc(&derr, 16 * n); hipMalloc(&dt, 16 * 8);
    hipMemcpy(dX, X.data(), 12 * n, hipMemcpyHostToDevice); hipMemcpy(dobs, obs.data(), 8 * n, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), &dt, sizeof(dt));
    PoseOptArgs a{};
    a.Xw = dX; a.obs = dobs; a.inv_sigma2 = dw; a.pose_out = dpose; a.info = dinfo; a.outlier = dout; a.err = derr; a.trace = nullptr;
    for (int k = 0; k < 4; k++) a.K[k] = K[k];
    a.init.q[0] = 0.004; a.init.q[1] = -0.003; a.init.q[2] = 0.002; a.init.q[3] = 1.0;
    const double qn = sqrt(a.init.q[0]*a.init.q[0] + a.init.q[1]*a.init.q[1] + a.init.q[2]*a.init.q[2] + 1.0);
    for (int k = 0; k < 4; k++) a.init.q[k] /= qn;
    a.init.t[0] = 0.02; a.init.t[1] = -0.015; a.init.t[2] = 0.01; a.init.pad = 0;
    a.n = n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 10; it++) {
        hipEventRecord(e0, 0); launch_pose_opt(a, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    long long t[9]; int info[4];
    hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost); hipMemcpy(info, dinfo, 16, hipMemcpyDeviceToHost);
    const double np = (double)t[8];
    printf("n %d: kernel %.1f us, outliers %d iterations %d trials %d, edge passes %lld\n", n, best * 1e3, info[0], info[1], info[2], t[8]);
    printf("per pass (cycles): edge loop %.0f  wave reduction %.0f  barrier %.0f  cross-wave sum %.0f  decision + solve + barrier + round work %.0f\n",
           t[0] / np, t[1] / np, t[2] / np, t[3] / np, t[6] / np);
    printf("  register-resident kernel only: decision %.0f  6x6 solve %.0f  SE3 update %.0f  (rest of the serial part %.0f)\n",
           t[4] / np, t[5] / np, t[7] / np, t[6] / np);
    printf("total cycles %lld\n", t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7]);
    return 0;
}
