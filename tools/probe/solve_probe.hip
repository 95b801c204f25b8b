// Phase-level cycle probe of the reduced-camera-system solver (developer tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I swarmmap_amd/csrc tools/probe/solve_probe.hip -o /tmp/solve_probe
//   (ba_dense.hip is included as well: launch_ba_solve dispatches into it; SWARMORB_BA_NO_MFMA_SOLVER=1 times the
//   register-resident solvers for sizes the MFMA kernel would take)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__device__ long long* g_marks;
__device__ int g_mark_tid;
#define SO_SOLVE_MARK_DECL long long* const so_marks_ = ((int)threadIdx.x == g_mark_tid) ? g_marks : nullptr
#define SO_SOLVE_MARK(k, phase) do { if (so_marks_) so_marks_[(k) * 8 + (phase)] = clock64(); } while (0)
#ifdef SO_SOLVE_DEBUG
__device__ double* g_dbgL;
#endif
#include "ba_kernels.hip"
#include "ba_dense.hip"
using namespace so;
int main(int argc, char** argv) {
    const int nf = argc > 1 ? atoi(argv[1]) : 25, n = 6 * nf;
    const int mark_tid = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<double> M((size_t)n * n), S((size_t)n * n), b(n);
    srand(1);
    for (auto& v : M) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double a = 0; for (int k = 0; k < n; k++) a += M[(size_t)i*n+k]*M[(size_t)j*n+k]; S[(size_t)i*n+j] = a + (i==j ? n : 0); }
    for (auto& v : b) v = (rand() / (double)RAND_MAX) - 0.5;
    double *dS, *dS0, *db, *db0, *dp; long long* dm;
    hipMalloc(&dS, sizeof(double)*n*n); hipMalloc(&dS0, sizeof(double)*n*n); hipMalloc(&db, sizeof(double)*n); hipMalloc(&db0, sizeof(double)*n);
    hipMalloc(&dp, sizeof(double)*kBaPartialCount); hipMalloc(&dm, sizeof(long long)*8*64);
    hipMemcpy(dS0, S.data(), sizeof(double)*n*n, hipMemcpyHostToDevice); hipMemcpy(db0, b.data(), sizeof(double)*n, hipMemcpyHostToDevice);
    hipMemset(dm, 0, sizeof(long long)*8*64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &dm, sizeof(dm)); hipMemcpyToSymbol(HIP_SYMBOL(g_mark_tid), &mark_tid, sizeof(int));
    BaLm hl{}; hl.active = 1; BaLm* dl; hipMalloc(&dl, sizeof(BaLm)); hipMemcpy(dl, &hl, sizeof(BaLm), hipMemcpyHostToDevice);
    BaDev d{}; d.n_free = nf; d.S = dS; d.bs = db; d.partial = dp; d.lm = dl; d.ldS = n;
#ifdef SO_SOLVE_DEBUG
    double* dL; hipMalloc(&dL, sizeof(double) * (nf + 1) * (nf + 1) * 36);
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbgL), &dL, sizeof(dL));
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 20; it++) {
        hipMemcpy(dS, dS0, sizeof(double)*n*n, hipMemcpyDeviceToDevice); hipMemcpy(db, db0, sizeof(double)*n, hipMemcpyDeviceToDevice);
        hipEventRecord(e0, 0); launch_ba_solve(d, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<double> x(n); hipMemcpy(x.data(), db, sizeof(double)*n, hipMemcpyDeviceToHost);
    double res = 0; for (int i = 0; i < n; i++) { double a = -b[i]; for (int j = 0; j < n; j++) a += S[(size_t)i*n+j]*x[j]; res = fmax(res, fabs(a)); }
    std::vector<long long> m(8*64); hipMemcpy(m.data(), dm, sizeof(long long)*8*64, hipMemcpyDeviceToHost);
#ifdef SO_SOLVE_DEBUG
    {
        const int NB = nf + 1;
        hipMemset(dL, 0, sizeof(double) * NB * NB * 36);
        hipMemcpy(dS, dS0, sizeof(double)*n*n, hipMemcpyDeviceToDevice); hipMemcpy(db, db0, sizeof(double)*n, hipMemcpyDeviceToDevice);
        launch_ba_solve(d, 0); hipDeviceSynchronize();
        std::vector<double> Lg((size_t)NB * NB * 36); hipMemcpy(Lg.data(), dL, sizeof(double) * NB * NB * 36, hipMemcpyDeviceToHost);
        // host Cholesky of [S | b]
        std::vector<double> A((size_t)(n + 1) * n);
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[(size_t)i * n + j] = S[(size_t)i * n + j];
        for (int j = 0; j < n; j++) A[(size_t)n * n + j] = b[j];
        for (int j = 0; j < n; j++) {
            double dj = sqrt(A[(size_t)j * n + j]); A[(size_t)j * n + j] = dj;
            for (int i = j + 1; i <= n; i++) A[(size_t)i * n + j] /= dj;
            for (int i = j + 1; i <= n; i++) for (int k2 = j + 1; k2 <= (i < n ? i : n - 1); k2++) A[(size_t)i * n + k2] -= A[(size_t)i * n + j] * A[(size_t)k2 * n + j];
        }
        int shown = 0;
        for (int J = 0; J < nf && shown < 12; J++) for (int I = J; I <= nf && shown < 12; I++) {
            double worst = 0;
            for (int r = 0; r < (I == nf ? 1 : 6); r++) for (int c = 0; c < 6; c++) {
                if (I == J && c > r) continue;
                const double ref = A[(size_t)(6 * I + r) * n + 6 * J + c], got = Lg[(size_t)(I * NB + J) * 36 + r * 6 + c];
                worst = fmax(worst, fabs(ref - got));
            }
            if (worst > 1e-9) { printf("  block (%d,%d) max err %.3e\n", I, J, worst); shown++; }
        }
    }
#endif
    printf("nf %d best %.1f us residual %.3e\n", nf, best*1e3, res);
    if (mark_tid < (argc > 3 ? atoi(argv[3]) : 192)) {  // a panel-team lane (argv[3] = first update-team thread)
        long long w_col=0,w_pan=0,apply=0,factor=0,w_upd=0,publish=0,total=0;
        for (int k = 0; k < nf; k++) { const long long* t = &m[k*8];
            w_col += t[1]-t[0]; if (k>0) { w_pan += t[2]-t[1]; apply += t[3]-t[2]; } else apply += t[3]-t[1];
            factor += t[4]-t[3]; w_upd += t[5]-t[4]; publish += t[6]-t[5]; total += t[6]-t[0];
            if (k < 3 || k == nf/2 || k == nf-1) printf(" panel step %d: wait col %lld wait panel %lld load+apply %lld factor %lld wait upd %lld publish %lld\n", k, t[1]-t[0], k>0?t[2]-t[1]:0, k>0?t[3]-t[2]:t[3]-t[1], t[4]-t[3], t[5]-t[4], t[6]-t[5]); }
        printf("panel lane totals: wait col %lld wait panel %lld load+apply %lld factor %lld wait upd %lld publish %lld | loop %lld\n", w_col,w_pan,apply,factor,w_upd,publish,total);
    } else {
        long long wait=0, work=0;
        for (int k = 0; k < nf; k++) { const long long* t = &m[k*8]; wait += t[1]-t[0]; work += t[2]-t[1];
            if (k < 3 || k == nf/2 || k == nf-1) printf(" update step %d: wait panel %lld work %lld\n", k, t[1]-t[0], t[2]-t[1]); }
        printf("update lane totals: wait %lld work %lld\n", wait, work);
    }
    printf("kernel: setup->loop %lld (start mark %lld) backward %lld\n", 0LL, m[40*8+2], m[40*8+1]-m[40*8+0]);
    return 0;
}
