// What several chains of small dependent kernels cost each other on one GPU (developer tool; NOTES.md G.10).
// K host threads, each with its own stream, launch a chain of N dependent kernels; the time per link of a chain is printed for
// K = 1, 2, 4, 8, 16 and for three kinds of link:
//   empty    64 workgroups x 256 threads that touch no memory
//   stream   the same grid reading + writing the chain's own 2 MB buffer once (a window's working set: stays in L2 when alone)
//   resident ONE launch per chain whose 64 workgroups do the same N passes over the buffer with a grid barrier (device-scope
//            release + acquire) between passes instead of a launch boundary
// If `stream` stretches with K while `empty` does not, the chains lose their cached working sets to each other's launch-boundary
// cache maintenance (every XCD's L2 is written back / invalidated per hand-over) rather than queueing at the command processor;
// `resident` tells whether barriers inside one launch are any cheaper than the boundaries they replace.
//   hipcc -O3 --offload-arch=gfx950 -pthread tools/probe/multi_chain_probe.hip -o tools/probe/multi_chain_bin
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

constexpr int kWgs = 64, kThreads = 256;
constexpr size_t kWords = 2u << 20 >> 2;  // 2 MB of uint32

__global__ __launch_bounds__(kThreads) void k_empty(unsigned* p) {
    if (p && threadIdx.x == 9999) *p = 1;
}
__device__ __forceinline__ void pass(unsigned* buf, unsigned add) {
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < kWords; i += (size_t)kWgs * kThreads) buf[i] += add;
}
__global__ __launch_bounds__(kThreads) void k_stream(unsigned* buf, unsigned add) { pass(buf, add); }
__global__ __launch_bounds__(kThreads) void k_resident(unsigned* buf, unsigned* sync, int n, unsigned epoch) {
    for (int t = 0; t < n; t++) {
        pass(buf, 1u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = epoch + (unsigned)t + 1u;
            if (__hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kWgs - 1) {
                __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sync[16], target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                unsigned long long polls = 0;
                while (__hip_atomic_load(&sync[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target && ++polls < (1ull << 26))
                    __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

struct Chain {
    hipStream_t s = nullptr;
    unsigned* buf = nullptr;
    unsigned* sync = nullptr;
};

static double run(int kind, int K, int n) {
    std::vector<Chain> ch((size_t)K);
    for (auto& c : ch) {
        hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking);
        hipMalloc((void**)&c.buf, kWords * 4);
        hipMemset(c.buf, 0, kWords * 4);
        hipMalloc((void**)&c.sync, 256);
        hipMemset(c.sync, 0, 256);
    }
    hipDeviceSynchronize();
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<double> ms((size_t)K, 0.0);
    std::vector<std::thread> th;
    for (int k = 0; k < K; k++)
        th.emplace_back([&, k] {
            Chain& c = ch[(size_t)k];
            auto chain = [&](int links, unsigned epoch) {
                if (kind == 2) {
                    hipLaunchKernelGGL(k_resident, dim3(kWgs), dim3(kThreads), 0, c.s, c.buf, c.sync, links, epoch);
                } else {
                    for (int i = 0; i < links; i++) {
                        if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(kWgs), dim3(kThreads), 0, c.s, (unsigned*)nullptr);
                        else hipLaunchKernelGGL(k_stream, dim3(kWgs), dim3(kThreads), 0, c.s, c.buf, 1u);
                    }
                }
                hipStreamSynchronize(c.s);
            };
            chain(50, 1u << 20);  // warm-up
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            const auto t0 = std::chrono::steady_clock::now();
            chain(n, 2u << 20);
            ms[(size_t)k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
    while (ready.load() < K) std::this_thread::yield();
    go.store(true);
    for (auto& t : th) t.join();
    double worst = 0.0;
    for (double v : ms) worst = v > worst ? v : worst;
    for (auto& c : ch) {
        hipFree(c.buf);
        hipFree(c.sync);
        hipStreamDestroy(c.s);
    }
    return worst * 1e3 / n;  // us per link of the slowest chain
}

int main() {
    const int n = 2000;
    const char* names[3] = {"empty   ", "stream  ", "resident"};
    printf("us per link of the slowest of K concurrent chains (%d links, %d workgroups x %d threads per link, 2 MB per chain)\n", n, kWgs, kThreads);
    printf("kind      K=1     K=2     K=4     K=8     K=16\n");
    for (int kind = 0; kind < 3; kind++) {
        printf("%s", names[kind]);
        for (int K : {1, 2, 4, 8, 16}) printf(" %7.2f", run(kind, K, n));
        printf("\n");
    }
    return 0;
}
