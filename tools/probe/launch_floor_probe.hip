// Dependent-launch floor as a function of the kernel-argument size (developer tool): N back-to-back launches of an empty
// kernel on one stream, timed with HIP events; argument = 8 bytes, a 528-byte struct by value (sizeof(BaDev)), or a
// pointer to the same struct in device memory.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/launch_floor_probe.hip -o tools/probe/launch_floor_bin
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double v[66]; };  // 528 bytes
__global__ void k_small(int* p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 999) *p = (int)b.v[65]; }
__global__ void k_ptr(const Big* b, int* p) { if (p && threadIdx.x == 999) *p = (int)b->v[65]; }
template <typename F>
static float run(F launch, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, s);
        for (int i = 0; i < n; i++) launch(s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3f / n;
}
int main() {
    Big b{}; Big* db; hipMalloc(&db, sizeof(Big)); hipMemcpy(db, &b, sizeof(Big), hipMemcpyHostToDevice);
    const int n = 2000;
    printf("empty kernel, 64 threads, 1 workgroup, %d dependent launches on one stream:\n", n);
    printf("  8-byte argument:            %.2f us per launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, nullptr); }, n));
    printf("  528-byte struct by value:   %.2f us per launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b, nullptr); }, n));
    printf("  pointer to the struct:      %.2f us per launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_ptr, dim3(1), dim3(64), 0, s, db, nullptr); }, n));
    printf("  528-byte struct, 100 workgroups of 256: %.2f us per launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(100), dim3(256), 0, s, b, nullptr); }, n));
    printf("  8-byte argument, 100 workgroups of 256: %.2f us per launch\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_small, dim3(100), dim3(256), 0, s, nullptr); }, n));
    return 0;
}
