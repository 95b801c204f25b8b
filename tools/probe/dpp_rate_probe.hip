// Issue cost of the instructions of the 16x16 pivot tile (developer tool): cycles per instruction for 16 independent
// accumulators, one wave.   hipcc -O3 --offload-arch=gfx950 tools/probe/dpp_rate_probe.hip -o tools/probe/dpp_rate_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ void rate_kernel(double* out, long long* cyc, double a, double b, int iters) {
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = a + c + threadIdx.x;
    double col = b + threadIdx.x, own = b * 0.5;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#define X(n) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(x[n]) : "v"(col), "v"(own));
            REP16(X)
#undef X
        } else if (MODE == 1) {
#define X(n) asm volatile("v_fmac_f64 %0, -%1, %2" : "+v"(x[n]) : "v"(col), "v"(own));
            REP16(X)
#undef X
        } else if (MODE == 2) {  // 64-bit DPP move + plain FMA
#define X(n) { double t; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(col)); \
               asm volatile("v_fmac_f64 %0, -%1, %2" : "+v"(x[n]) : "v"(t), "v"(own)); }
            REP16(X)
#undef X
        } else if (MODE == 3) {  // scalar multiplier (v_readlane'd beforehand): FMA with an SGPR operand
            double s = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(col)), __builtin_amdgcn_readfirstlane(__double2loint(col)));
#define X(n) asm volatile("v_fmac_f64 %0, -%1, %2" : "+v"(x[n]) : "s"(s), "v"(own));
            REP16(X)
#undef X
        } else if (MODE == 4) {  // 32-bit DPP FMA for comparison
#define X(n) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(((float*)x)[2 * n]) : "v"((float)col), "v"((float)own));
            REP16(X)
#undef X
        } else if (MODE == 5) {  // readlane pair + FMA with the scalar
#define X(n) { int lo, hi; asm volatile("v_readlane_b32 %0, %2, " #n "\n\tv_readlane_b32 %1, %3, " #n : "=s"(lo), "=s"(hi) : "v"(__double2loint(col)), "v"(__double2hiint(col))); \
               double s = __hiloint2double(hi, lo); asm volatile("v_fmac_f64 %0, -%1, %2" : "+v"(x[n]) : "s"(s), "v"(own)); }
            REP16(X)
#undef X
        } else if (MODE == 6) {  // dependent chain of plain FMAs
#define X(n) asm volatile("v_fmac_f64 %0, -%1, %2" : "+v"(x[0]) : "v"(col), "v"(own));
            REP16(X)
#undef X
        } else if (MODE == 7) {  // dependent chain of rsq
#define X(n) asm volatile("v_rsq_f64 %0, %0\n\ts_nop 0" : "+v"(x[0]));
            REP16(X)
#undef X
        } else if (MODE == 12 || MODE == 13) {  // FP64 MFMA 16x16x4: one accumulator (dependent) / four (independent)
            typedef double d4 __attribute__((ext_vector_type(4)));
            d4 acc[4];
            for (int q = 0; q < 4; q++) acc[q] = d4{x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
#pragma unroll
            for (int n = 0; n < 16; n++) { const int q = MODE == 12 ? 0 : (n & 3); acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(col, own, acc[q], 0, 0, 0); }
            for (int q = 0; q < 4; q++) { x[4 * q] = acc[q][0]; x[4 * q + 1] = acc[q][1]; x[4 * q + 2] = acc[q][2]; x[4 * q + 3] = acc[q][3]; }
        } else if (MODE >= 8) {  // the pivot chain itself, 16 times: variants knock out one link each
#define X(n) { double y0, t, e, p, ye, y, xs, pn; \
            if (MODE == 9) asm volatile("v_mul_f64 %0, %1, %1" : "=v"(y0) : "v"(x[0])); else asm volatile("v_rsq_f64 %0, %1\n\ts_nop 0" : "=v"(y0) : "v"(x[0])); \
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(x[0]), "v"(y0)); \
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(t), "v"(y0)); \
            asm volatile("v_fma_f64 %0, %1, %2, 0.5" : "=v"(p) : "v"(own), "v"(e)); \
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(ye) : "v"(y0), "v"(e)); \
            asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "v"(ye), "v"(p), "v"(y0)); \
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(xs) : "v"(x[1]), "v"(y)); \
            asm volatile("v_fma_f64 %0, -%1, %1, %2" : "=v"(pn) : "v"(xs), "v"(x[2])); \
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x[3]) : "v"(x[3]), "v"(y)); \
            if (MODE == 10) asm volatile("v_mov_b64 %0, %1" : "=v"(x[0]) : "v"(pn)); \
            else if (MODE == 11) { int lo, hi; asm volatile("s_nop 0\n\tv_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(pn)), "v"(__double2hiint(pn))); x[0] = __hiloint2double(hi, lo); } \
            else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x[0]) : "v"(pn)); }
            REP16(X)
#undef X
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) s += x[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* what, int per_iter) {
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    const int iters = 500;
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 0.5, 1e-3, iters);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %.2f cycles per instruction\n", what, (double)c / (iters * 16.0));
}
int main() {
    run<0>("v_fmac_f64_dpp row_newbcast, 16 independent:", 16);
    run<1>("v_fmac_f64, 16 independent:", 16);
    run<2>("v_mov_b64_dpp + v_fmac_f64, 16 independent pairs:", 32);
    run<3>("v_fmac_f64 with a scalar multiplier, 16 independent:", 16);
    run<4>("v_fmac_f32_dpp row_newbcast, 16 independent:", 16);
    run<5>("2 x v_readlane_b32 + v_fmac_f64 (scalar), 16 independent:", 48);
    run<6>("v_fmac_f64 dependent chain:", 16);
    run<7>("v_rsq_f64 dependent chain:", 16);
    run<8>("pivot chain (rsq, 6 levels, dpp broadcast), per pivot:", 16);
    run<9>("pivot chain with a multiplication in place of the rsq:", 16);
    run<10>("pivot chain with a plain move in place of the dpp move:", 16);
    run<11>("pivot chain with two v_readlane in place of the dpp move:", 16);
    run<12>("v_mfma_f64_16x16x4f64, one accumulator (dependent):", 16);
    run<13>("v_mfma_f64_16x16x4f64, four accumulators in turn:", 16);
    return 0;
}
