// pcg_spmv_probe.hip - what ONE iteration of block-Jacobi PCG on a reduced camera system costs on the GPU (round 4's verdict,
// item 5; tools/pcg_study.py has the iteration counts): y = S p with S in 6 x 6 block-sparse rows (FP64, full symmetric
// storage) fused with the partial sums of p.Sp, then the vector updates x += a p, r -= a Sp, z = M^-1 r (6 x 6 blocks) with
// the partial sums of r.z, then p = z + b p - three launches per iteration, the scalars staying on the device.
// Usage: pcg_spmv_probe <block rows> <blocks per row> <band: 0 = spread over all columns, w = within +-w of the diagonal> [reps]
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/pcg_spmv_probe.hip -o /tmp/pcg_spmv_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one wavefront per block row: lane l works on block (l / 6) of a group of ten, row (l % 6) of it: 48 contiguous bytes per lane,
// a block's 288 bytes over six neighbouring lanes
__global__ __launch_bounds__(256) void bsr_spmv_dot_kernel(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                           const double* __restrict__ blocks, const double* __restrict__ p,
                                                           double* __restrict__ Sp, double* __restrict__ partial, int n_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    double acc = 0.0;
    const int sub = lane / 6, r = lane % 6;
    if (row < n_rows && lane < 60) {
        const int lo = indptr[row], hi = indptr[row + 1];
        for (int k = lo + sub; k < hi; k += 10) {
            const double* B = blocks + 36 * (size_t)k + 6 * r;
            const double* x = p + 6 * (size_t)indices[k];
            acc += B[0] * x[0] + B[1] * x[1] + B[2] * x[2] + B[3] * x[3] + B[4] * x[4] + B[5] * x[5];
        }
    }
    // sum over the ten sub-blocks: lanes r, r + 6, ..., r + 54
    double tot = acc;
    for (int s = 1; s < 10; s++) tot += __shfl(acc, (lane % 6) + 6 * s >= 60 ? lane : r + 6 * ((sub + s) % 10));
    double dot = 0.0;
    if (row < n_rows && lane < 6) {
        Sp[6 * (size_t)row + lane] = tot;
        dot = tot * p[6 * (size_t)row + lane];
    }
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    __shared__ double wsum[4];
    if (lane == 0) wsum[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// alpha = rz / sum(partial); x += alpha p; r -= alpha Sp; z = Minv r; partial2 = r.z
__global__ __launch_bounds__(256) void update_kernel(const double* __restrict__ partial, int n_partial, const double* __restrict__ rz,
                                                     const double* __restrict__ Minv, const double* __restrict__ p,
                                                     const double* __restrict__ Sp, double* __restrict__ x, double* __restrict__ r,
                                                     double* __restrict__ z, double* __restrict__ partial2, int n_rows) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double alpha = rz[0] / sh[0];
    __syncthreads();
    const int row = blockIdx.x * 256 + threadIdx.x;  // one block row per thread
    double d = 0.0;
    if (row < n_rows) {
        double rr[6];
        for (int i = 0; i < 6; i++) {
            const size_t q = 6 * (size_t)row + i;
            x[q] += alpha * p[q];
            rr[i] = r[q] - alpha * Sp[q];
            r[q] = rr[i];
        }
        const double* M = Minv + 36 * (size_t)row;
        for (int i = 0; i < 6; i++) {
            double v = 0.0;
            for (int j = 0; j < 6; j++) v += M[6 * i + j] * rr[j];
            z[6 * (size_t)row + i] = v;
            d += v * rr[i];
        }
    }
    sh[threadIdx.x] = d;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial2[blockIdx.x] = sh[0];
}

// beta = sum(partial2) / rz; p = z + beta p; rz <- sum(partial2)
__global__ __launch_bounds__(256) void direction_kernel(const double* __restrict__ partial2, int n_partial2, double* __restrict__ rz,
                                                        const double* __restrict__ z, double* __restrict__ p, double* __restrict__ rz_next,
                                                        int n) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partial2; i += 256) s += partial2[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    const double rz_new = sh[0], beta = rz_new / rz[0];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = z[i] + beta * p[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) rz_next[0] = rz_new;
}

int main(int argc, char** argv) {
    const int nf = argc > 1 ? atoi(argv[1]) : 1500, per_row = argc > 2 ? atoi(argv[2]) : 300, band = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 200;
    std::mt19937 rng(7);
    std::vector<int> indptr(nf + 1, 0), indices;
    for (int i = 0; i < nf; i++) {
        std::vector<int> cols;
        if (per_row >= nf) {
            for (int j = 0; j < nf; j++) cols.push_back(j);
        } else {
            std::vector<uint8_t> used(nf, 0);
            used[i] = 1;
            cols.push_back(i);
            const int avail = band > 0 ? std::min(nf - 1, i + band) - std::max(0, i - band) + 1 : nf;
            const int want = std::min(per_row, avail);
            while ((int)cols.size() < want) {
                const int j = band > 0 ? i + (int)(rng() % (2 * band + 1)) - band : (int)(rng() % nf);
                if (j < 0 || j >= nf || used[j]) continue;
                used[j] = 1;
                cols.push_back(j);
            }
            std::sort(cols.begin(), cols.end());
        }
        indices.insert(indices.end(), cols.begin(), cols.end());
        indptr[i + 1] = (int)indices.size();
    }
    const size_t nnz = indices.size(), n = 6 * (size_t)nf;
    std::vector<double> blocks(36 * nnz), vec(n, 1.0), Minv(36 * (size_t)nf, 0.0);
    for (auto& v : blocks) v = (double)(rng() % 1000) * 1e-3;
    for (int i = 0; i < nf; i++)
        for (int d = 0; d < 6; d++) Minv[36 * (size_t)i + 7 * d] = 1.0;
    int *d_indptr, *d_indices;
    double *d_blocks, *d_p, *d_Sp, *d_x, *d_r, *d_z, *d_Minv, *d_part, *d_part2, *d_rz;
    const int g1 = (nf + 3) / 4, g2 = (nf + 255) / 256, g3 = (int)((n + 255) / 256);
    CK(hipMalloc(&d_indptr, 4 * (nf + 1))); CK(hipMalloc(&d_indices, 4 * nnz)); CK(hipMalloc(&d_blocks, 8 * 36 * nnz));
    CK(hipMalloc(&d_p, 8 * n)); CK(hipMalloc(&d_Sp, 8 * n)); CK(hipMalloc(&d_x, 8 * n)); CK(hipMalloc(&d_r, 8 * n)); CK(hipMalloc(&d_z, 8 * n));
    CK(hipMalloc(&d_Minv, 8 * 36 * (size_t)nf)); CK(hipMalloc(&d_part, 8 * g1)); CK(hipMalloc(&d_part2, 8 * g2)); CK(hipMalloc(&d_rz, 16));
    CK(hipMemcpy(d_indptr, indptr.data(), 4 * (nf + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_indices, indices.data(), 4 * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_blocks, blocks.data(), 8 * 36 * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_Minv, Minv.data(), 8 * 36 * (size_t)nf, hipMemcpyHostToDevice));
    for (double* v : {d_p, d_x, d_r, d_z}) CK(hipMemcpy(v, vec.data(), 8 * n, hipMemcpyHostToDevice));
    const double one[2] = {1.0, 1.0};
    CK(hipMemcpy(d_rz, one, 16, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto iteration = [&](bool spmv_only) {
        hipLaunchKernelGGL(bsr_spmv_dot_kernel, dim3(g1), dim3(256), 0, s, d_indptr, d_indices, d_blocks, d_p, d_Sp, d_part, nf);
        if (spmv_only) return;
        hipLaunchKernelGGL(update_kernel, dim3(g2), dim3(256), 0, s, d_part, g1, d_rz, d_Minv, d_p, d_Sp, d_x, d_r, d_z, d_part2, nf);
        hipLaunchKernelGGL(direction_kernel, dim3(g3), dim3(256), 0, s, d_part2, g2, d_rz, d_z, d_p, d_rz, (int)n);
    };
    float ms_all = 0.f, ms_spmv = 0.f;
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < 20; i++) iteration(pass == 1);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; i++) iteration(pass == 1);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(pass == 0 ? &ms_all : &ms_spmv, e0, e1));
    }
    const double bytes = 288.0 * (double)nnz + 4.0 * (double)nnz + 96.0 * (double)nf;
    printf("{\"block_rows\": %d, \"nonzero_blocks\": %zu, \"band\": %d, \"us_per_iteration\": %.2f, \"us_per_spmv\": %.2f, "
           "\"spmv_algorithmic_GBs\": %.1f, \"matrix_MB\": %.1f}\n",
           nf, nnz, band, 1e3 * ms_all / reps, 1e3 * ms_spmv / reps, bytes / (1e-3 * ms_spmv / reps) / 1e9, 288.0 * (double)nnz / 1e6);
    return 0;
}
