// FP64 issue / latency microbenchmark (developer tool): cycles per v_fma_f64 in a dependent chain and with 2 / 4 / 8
// independent chains, one wave and two waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/dp_latency_probe.hip -o /tmp/dp_probe && /tmp/dp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void chain_kernel(double* out, long long* cyc, double a, double b, int iters) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = a + c + threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fma(x[c], b, a);
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void rcp_chain_kernel(double* out, long long* cyc, double a, int iters) {
    double x = a + threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) x = __builtin_amdgcn_rcp(x) + a;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void f32_chain_kernel(float* out, long long* cyc, float a, float b, int iters) {
    float x = a + threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) x = __builtin_fmaf(x, b, a);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CHAINS>
void run(int threads, const char* what) {
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    const int iters = 1000;
    hipLaunchKernelGGL(chain_kernel<CHAINS>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.5, 0.999, iters);
    hipLaunchKernelGGL(chain_kernel<CHAINS>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.5, 0.999, iters);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %d chains, %d threads: %.2f cycles per fma instruction (%.2f per chain step)\n", what, CHAINS, threads,
           (double)c / (iters * 8.0 * CHAINS), (double)c / (iters * 8.0));
}
int main() {
    run<1>(64, "fp64"); run<2>(64, "fp64"); run<4>(64, "fp64"); run<8>(64, "fp64");
    run<1>(256, "fp64"); run<4>(256, "fp64"); run<1>(512, "fp64"); run<4>(512, "fp64"); run<8>(512, "fp64");
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(rcp_chain_kernel, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, 1000);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("fp64 rcp + add chain: %.2f cycles per (rcp, add) pair\n", (double)c / 8000.0);
    hipLaunchKernelGGL(f32_chain_kernel, dim3(1), dim3(64), 0, 0, (float*)out, cyc, 1.5f, 0.999f, 1000);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("fp32 fma chain: %.2f cycles per step\n", (double)c / 8000.0);
    return 0;
}
