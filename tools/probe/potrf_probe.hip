// Phase-level cycle probe of the blocked solver's diagonal-block kernel (developer tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I swarmmap_amd/csrc tools/probe/potrf_probe.hip -o gpurun_out/potrf_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long* g_marks;
#define SO_POTRF_MARK(i) do { if (threadIdx.x == 0) g_marks[(i)] = clock64(); } while (0)
#include "ba_dense.hip"
using namespace so;
int main() {
    const int n = 96;
    std::vector<double> M((size_t)n * n), S((size_t)n * n);
    srand(1);
    for (auto& v : M) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double a = 0;
            for (int k = 0; k < n; k++) a += M[(size_t)i * n + k] * M[(size_t)j * n + k];
            S[(size_t)i * n + j] = a + (i == j ? n : 0);
        }
    double *dS, *dS0, *dW, *dp;
    long long* dm;
    hipMalloc(&dS, sizeof(double) * n * n); hipMalloc(&dS0, sizeof(double) * n * n); hipMalloc(&dW, sizeof(double) * n * n);
    hipMalloc(&dp, sizeof(double) * kBaPartialCount); hipMalloc(&dm, sizeof(long long) * 64);
    hipMemcpy(dS0, S.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipMemset(dm, 0, sizeof(long long) * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &dm, sizeof(dm));
    BaLm hl{}; hl.active = 1;
    BaLm* dl; hipMalloc(&dl, sizeof(BaLm)); hipMemcpy(dl, &hl, sizeof(BaLm), hipMemcpyHostToDevice);
    BaDev d{}; d.stage = 1; d.n_free = 16; d.S = dS; d.partial = dp; d.lm = dl; d.ldS = n; d.dense_ws = dW;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 20; it++) {
        hipMemcpy(dS, dS0, sizeof(double) * n * n, hipMemcpyDeviceToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(dense_potrf_kernel, dim3(1), dim3(256), 0, 0, d, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<double> W((size_t)n * n); hipMemcpy(W.data(), dW, sizeof(double) * n * n, hipMemcpyDeviceToHost);
    // check: W^-T W^-1 ... cheaper: || W S W^T - I ||
    double err = 0;
    std::vector<double> T((size_t)n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double a = 0; for (int k = 0; k < n; k++) a += W[(size_t)i*n+k] * S[(size_t)k*n+j]; T[(size_t)i*n+j] = a; }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double a = 0; for (int k = 0; k < n; k++) a += T[(size_t)i*n+k] * W[(size_t)j*n+k]; err = fmax(err, fabs(a - (i == j))); }
    std::vector<long long> m(64); hipMemcpy(m.data(), dm, sizeof(long long) * 64, hipMemcpyDeviceToHost);
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < W.size() * sizeof(double); i++) h = (h ^ ((const unsigned char*)W.data())[i]) * 1099511628211ull;
    printf("kernel %.2f us, |W S W^T - I| = %.3g, FNV-1a of the inverse factor's bytes %016llx\n", best * 1e3, err, h);
    printf("diag0 %lld\n", m[1] - m[0]);
    for (int jb = 0; jb < 6; jb++)
        printf("jb %d: panel %lld  diag-tile %lld  diag16 %lld  tail+barrier %lld\n", jb, m[2 + 4*jb] - m[1 + 4*jb], jb < 5 ? m[3 + 4*jb] - m[2 + 4*jb] : 0,
               jb < 5 ? m[4 + 4*jb] - m[3 + 4*jb] : 0, (jb < 5 ? m[5 + 4*jb] : m[25]) - (jb < 5 ? m[4 + 4*jb] : m[2 + 4*jb]));
    printf("last row's store %lld  total(marks) %lld  (shader clock cycles)\n", m[26] - m[25], m[26] - m[0]);
    return 0;
}
