// Phase-level cycle probe of the reduced-camera-system solver (developer tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I swarmmap_amd/csrc tools/probe/solve_probe.hip -o gpurun_out/solve_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
__device__ long long* g_marks;
__device__ int g_mark_tid;
#define SO_SOLVE_MARK(k, phase) do { if ((int)threadIdx.x == g_mark_tid) g_marks[(k) * 8 + (phase)] = clock64(); } while (0)
#include "ba_kernels.hip"
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
    BaDev d{}; d.n_free = nf; d.S = dS; d.bs = db; d.partial = dp;
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
    printf("nf %d best %.1f us residual %.3e\n", nf, best*1e3, res);
    long long diag=0, panel=0, upd=0; 
    for (int k = 0; k < nf; k++) { diag += m[k*8+2]-m[k*8+1]; panel += m[k*8+3]-m[k*8+2]; upd += m[k*8+4]-m[k*8+3];
        if (k < 4 || k == nf-1) printf(" step %d: diag+wait %lld panel %lld update(own) %lld\n", k, m[k*8+2]-m[k*8+1], m[k*8+3]-m[k*8+2], m[k*8+4]-m[k*8+3]); }
    printf("cycles: load->%lld factor total %lld (diag %lld panel %lld upd %lld) backward %lld\n", 0LL, m[5]-m[0], diag, panel, upd, m[6]-m[5]);
    return 0;
}
