// Cost of a grid-wide barrier among co-resident workgroups (developer tool; what a persistent local-window kernel would
// pay instead of a dependent launch): N workgroups of 512 threads spin on one counter in device memory (sense reversal,
// agent-scope atomics; or groups of eight workgroups with a counter each and one counter above them), 2000 barriers, time
// per barrier.  Run under `timeout`: the workgroups must all be resident.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/grid_barrier_probe.hip -o tools/probe/grid_barrier_bin
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool TWO_LEVEL, int POLL>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* counter, unsigned* sense, int iters, unsigned long long* sink, size_t lds_pad) {
    (void)lds_pad;
    unsigned local = 0;
    double acc = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        acc = acc * 1.0000001 + 1.0;  // a little work between barriers
        __syncthreads();
        if (threadIdx.x == 0) {
            local ^= 1u;
            __threadfence();  // what a phase wrote is visible to the agent before the arrival is
            bool last;
            if (TWO_LEVEL) {  // groups of 8 workgroups count on their own word (64 B apart), the last of a group counts for it
                const unsigned g = blockIdx.x >> 3, gsize = min(8u, gridDim.x - 8u * g), ngroups = (gridDim.x + 7u) >> 3;
                unsigned* gc = counter + 16u * (1u + g);
                last = false;
                if (__hip_atomic_fetch_add(gc, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gsize) {
                    __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == ngroups;
                }
            } else {
                last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
            }
            if (last) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sense, local, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (POLL == 0) {
                    while (__hip_atomic_load(sense, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local) __builtin_amdgcn_s_sleep(1);
                } else if (POLL == 1) {  // no sleep between polls
                    while (__hip_atomic_load(sense, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != local) {
                    }
                } else {                 // relaxed polls, one acquire fence at the end
                    while (__hip_atomic_load(sense, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != local) {
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = (unsigned long long)acc;
}
template <bool TWO_LEVEL, int POLL>
static void sweep(unsigned* counter, unsigned* sense, unsigned long long* sink, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int n : {1, 8, 32, 64, 96, 128, 192}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(counter, 0, 64 * 64); hipMemset(sense, 0, 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((barrier_kernel<TWO_LEVEL, POLL>), dim3(n), dim3(512), 0, 0, counter, sense, iters, sink, (size_t)0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s, %3d workgroups of 512 threads: %.2f us per grid barrier\n", what, n, best * 1e3f / iters);
    }
}
int main() {
    unsigned *counter, *sense; unsigned long long* sink;
    hipMalloc(&counter, 64 * 64); hipMalloc(&sense, 64); hipMalloc(&sink, 8);
    sweep<false, 0>(counter, sense, sink, "one counter");
    sweep<true, 0>(counter, sense, sink, "groups of 8 + one counter");
    sweep<false, 1>(counter, sense, sink, "one counter, polls without s_sleep");
    sweep<false, 2>(counter, sense, sink, "one counter, relaxed polls + one acquire fence");
    return 0;
}
