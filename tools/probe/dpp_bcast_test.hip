// Checks v_fmac_f64_dpp with row_newbcast and a negated DPP operand on the device (developer tool):
//   hipcc -O3 --offload-arch=gfx950 tools/probe/dpp_bcast_test.hip -o /tmp/t && /tmp/t     -> "bad 0"
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, const double* in) {
    double x = in[threadIdx.x], a = in[64 + threadIdx.x], acc = in[128 + threadIdx.x];
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(a));
    out[threadIdx.x] = acc;
}
int main() {
    double h[192], o[64]; for (int i = 0; i < 192; i++) h[i] = i * 0.5 + 1;
    double *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dout, di); hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) { double ref = h[128 + i] - h[(i & ~15) + 3] * h[64 + i]; if (o[i] != ref) bad++; }
    printf("bad %d (o[0]=%g ref=%g)\n", bad, o[0], h[128] - h[3] * h[64]);
    return 0;
}
