// Phase-level cycle probe of the single-workgroup MFMA solver (developer tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I swarmmap_amd/csrc tools/probe/solve_mfma_probe.hip -o /tmp/solve_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long* g_marks;
#define SO_POTRF_MARK(i) do { if (threadIdx.x == 0) g_marks[(i)] = clock64(); } while (0)
#include "ba_dense.hip"
using namespace so;
int main(int argc, char** argv) {
    const int nf = argc > 1 ? atoi(argv[1]) : 25, n = 6 * nf;
    std::vector<double> M((size_t)n * n), S((size_t)n * n), b(n);
    srand(1);
    for (auto& v : M) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double a = 0;
            for (int k = 0; k < n; k++) a += M[(size_t)i * n + k] * M[(size_t)j * n + k];
            S[(size_t)i * n + j] = a + (i == j ? n : 0);
        }
    for (auto& v : b) v = (rand() / (double)RAND_MAX) - 0.5;
    if (argc > 2) S[(size_t)atoi(argv[2]) * n + atoi(argv[2])] = -1.0;  // argv[2]: make this pivot fail
    double *dS, *db, *db0, *dp;
    long long* dm;
    hipMalloc(&dS, sizeof(double) * n * n); hipMalloc(&db, sizeof(double) * n); hipMalloc(&db0, sizeof(double) * n);
    hipMalloc(&dp, sizeof(double) * kBaPartialCount); hipMalloc(&dm, sizeof(long long) * 64);  // marks 0..63
    hipMemcpy(dS, S.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    hipMemcpy(db0, b.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    hipMemset(dm, 0, sizeof(long long) * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &dm, sizeof(dm));
    BaLm hl{}; hl.active = 1;
    BaLm* dl; hipMalloc(&dl, sizeof(BaLm)); hipMemcpy(dl, &hl, sizeof(BaLm), hipMemcpyHostToDevice);
    BaDev d{}; d.stage = 1; d.n_free = nf; d.S = dS; d.bs = db; d.partial = dp; d.lm = dl; d.ldS = n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 20; it++) {
        hipMemcpy(db, db0, sizeof(double) * n, hipMemcpyDeviceToDevice);
        hipEventRecord(e0, 0);
        if (!launch_ba_solve_mfma(d, 0)) { printf("does not apply\n"); return 1; }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<double> x(n); hipMemcpy(x.data(), db, sizeof(double) * n, hipMemcpyDeviceToHost);
    double res = 0;
    for (int i = 0; i < n; i++) { double a = -b[i]; for (int j = 0; j < n; j++) a += S[(size_t)i * n + j] * x[j]; res = fmax(res, fabs(a)); }
    std::vector<long long> m(64); hipMemcpy(m.data(), dm, sizeof(long long) * 64, hipMemcpyDeviceToHost);
    const int NT = (n + 1 + 15) / 16;
    double okflag = -1; hipMemcpy(&okflag, dp + kBaSolveOk, sizeof(double), hipMemcpyDeviceToHost);
    printf("nf %d n %d tiles %d: kernel %.2f us, |S x - b| = %.3g, solve_ok %.1f\n", nf, n, NT, best * 1e3, res, okflag);
    if (NT <= 11) printf("diag0 %lld\n", m[1] - m[0]);
    for (int jb = 0; NT <= 11 && jb + 1 < NT; jb++)
        printf("jb %d: panel+barrier %lld  wave0 tile+diag16 %lld  wait %lld\n", jb, m[2 + 3*jb] - m[1 + 3*jb], m[3 + 3*jb] - m[2 + 3*jb], m[4 + 3*jb] - m[3 + 3*jb]);
    if (NT > 11) {  // register-resident kernel: marks of wave 0
        for (int jb = 0; jb + 1 < NT; jb += 5)
            printf("jb %d: wait for panel %lld  tile+diag16 %lld  wait for updates %lld\n", jb, m[2 + 3*jb] - m[1 + 3*jb], m[3 + 3*jb] - m[2 + 3*jb], m[4 + 3*jb] - m[3 + 3*jb]);
        printf("factor (from first panel wait) %lld  backward %lld  (shader clock cycles)\n", m[58] - m[1], m[59] - m[58]);
        return 0;
    }
    printf("factor total %lld  backward %lld  (shader clock cycles)\n", m[40] - m[0], m[41] - m[40]);
    return 0;
}
