// Integer VALU issue-rate microbenchmark (developer tool): chip-wide lane-ops/s of the plain 32-bit vector instructions
// the Hamming kernels are made of (v_xor_b32, v_bcnt_u32_b32, v_min_i32, v_med3_i32, v_add_u32), each alone and in the
// scan's own mix - the ceiling the detection scan of the candidate search (kfstore_kernels.hip: kf_scan_kernel) is
// priced against.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// MODE 0: v_xor  1: v_bcnt (accumulating)  2: v_add_u32  3: v_min_i32  4: v_med3_i32  5: the scan's mix (8 xor + 8 bcnt + med3 + min)
template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t* out, uint32_t a, int iters) {
    constexpr int CHAINS = 8;
    uint32_t x[CHAINS], d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { x[c] = a * (c + 1) + threadIdx.x; d[c] = c; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (MODE == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(d[c]) : "s"(a));
                if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(d[c]) : "v"(x[c]));
                if (MODE == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(d[c]) : "v"(x[c]));
                if (MODE == 3) asm volatile("v_min_i32 %0, %1, %0" : "+v"(d[c]) : "v"(x[c]));
                if (MODE == 4) asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(d[c]) : "v"(x[c]), "v"(x[(c + 1) % CHAINS]));
            }
        if (MODE == 5) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                uint32_t t, acc = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t) : "s"(a + w), "v"(x[c]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(t));
                }
                asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(d[c]) : "v"(x[c]), "v"(acc));
                asm volatile("v_min_i32 %0, %1, %0" : "+v"(x[c]) : "v"(acc));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += d[c] + x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what, int blocks, int threads) {
    uint32_t* out;
    hipMalloc(&out, sizeof(uint32_t) * (size_t)blocks * threads);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 0x9e3779b9u, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 0x9e3779b9u, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * (MODE == 5 ? 8 * 18 : 16);
    const double lane_ops = instr_per_wave * (double)threads * blocks;
    const double t = lane_ops / (ms * 1e-3) / 1e12;
    // cycles a SIMD spends per wave64 instruction if the clock were 2.4 GHz: 1024 SIMDs
    printf("%-34s %5d x %3d threads: %6.1f T lane-ops/s  (%.2f cycles per wave64 instruction per SIMD at 2.4 GHz)\n", what, blocks,
           threads, t, 64.0 * 1024 * 2.4e9 / (t * 1e12));
    (void)hipFree(out);
}

int main() {
    run<0>("v_xor_b32 (SGPR operand)", 4096, 256);
    run<1>("v_bcnt_u32_b32 (accumulating)", 4096, 256);
    run<2>("v_add_u32", 4096, 256);
    run<3>("v_min_i32", 4096, 256);
    run<4>("v_med3_i32", 4096, 256);
    run<5>("scan mix: 8 xor + 8 bcnt + med3 + min", 4096, 256);
    run<5>("scan mix, one workgroup per CU", 256, 256);
    run<5>("scan mix, two workgroups per CU", 512, 256);
    return 0;
}
