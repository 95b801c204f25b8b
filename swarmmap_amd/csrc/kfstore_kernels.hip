// kfstore_kernels.hip — gfx950 kernels of the keyframe store and the cross-agent candidate search.
//
//   kf_append_kernel         a gathered keyframe record -> its ring slot + the compacted rows of the keypoints that
//                            carry a map point (the only ones SearchByBoW(KF, KF) compares, ORBmatcher.cc:517-541)
//   kf_scan_kernel<Q>        phase 1, the one throughput kernel of the path: every lane keeps Q query descriptors in
//                            registers (8 VGPRs each), the store's rows stream through the SCALAR unit (one
//                            s_load_dwordx8 per row, the row is an SGPR operand of the vector XORs), per pair 8 v_xor +
//                            8 v_bcnt (the popcount accumulates) + v_med3 / v_min for best / second: 18 VALU
//                            instructions, no LDS, no cross-lane traffic until the vote count at the end.  Bound: the
//                            integer VALU rate (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s); HBM
//                            traffic is the store read once (32 B per row against >= 64 x 18 lane-ops).
//   kf_pair_topk_kernel      phase 2: K smallest (distance, keypoint) keys of every query row against ONE candidate
//                            keyframe, a wave per row, keys staged in LDS
//   kf_pair_rerun_kernel     phase 2, exhausted K-list: best two among the rows the host's resolve has not taken yet
//   kf_pack_record_kernel    a version-2 record assembled in the exchange slot from a device-resident frame
#include "kfstore_device.h"

namespace so {

namespace {

constexpr int kHdrBytes = 128;
// so_keyframe_header field offsets (include/swarmorb.h; static_assert'ed in kfstore.cpp)
constexpr int kOffAgent = 8, kOffN = 12, kOffFlags = 104, kOffBound = 108, kOffChecksum = 32;

typedef const uint32_t __attribute__((address_space(4))) * ConstU32Ptr;  // constant address space: uniform reads go through s_load

__device__ __forceinline__ int med3_i32(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int bcnt_acc(uint32_t x, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}

__device__ __forceinline__ int hamming_row(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Compaction of a record's bound keypoints, in keypoint order, by the 256 threads of a workgroup.
// Returns the number of rows written (uniform).  s_cnt: 4 ints of LDS.
__device__ int compact_rows(const uint8_t* __restrict__ r, int n, bool has_mp, uint32_t* __restrict__ out_desc,
                            uint16_t* __restrict__ out_idx, float* __restrict__ out_angle, int* s_cnt) {
    const uint4* desc = reinterpret_cast<const uint4*>(r + kHdrBytes);
    const float4* geo = reinterpret_cast<const float4*>(r + kHdrBytes + (size_t)n * 32);
    const int32_t* mp = reinterpret_cast<const int32_t*>(r + kHdrBytes + (size_t)n * 48);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int total = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const bool in = i < n;
        const bool valid = in && (!has_mp || mp[i] >= 0);
        const unsigned long long b = __ballot(valid);
        if (lane == 0) s_cnt[wave] = __popcll(b);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wave; w++) off += s_cnt[w];
        const int round_total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
        if (in && out_angle) out_angle[i] = geo[i].z;
        if (valid) {
            const int pos = off + __popcll(b & ((1ull << lane) - 1ull));
            uint4* o = reinterpret_cast<uint4*>(out_desc + (size_t)pos * 8);
            o[0] = desc[2 * i];
            o[1] = desc[2 * i + 1];
            out_idx[pos] = (uint16_t)i;
        }
        total += round_total;
    }
    return total;
}

__global__ __launch_bounds__(256) void kf_append_kernel(KfStoreDev S, const uint8_t* __restrict__ src, size_t src_stride,
                                                        const int32_t* __restrict__ job_slot) {
    __shared__ int s_cnt[4];
    const int slot = job_slot[blockIdx.x];
    if (slot < 0) return;
    const uint8_t* r = src + (size_t)blockIdx.x * src_stride;
    const int n = *reinterpret_cast<const int32_t*>(r + kOffN);  // validated by the host: 0 <= n <= S.kp
    const uint32_t flags = *reinterpret_cast<const uint32_t*>(r + kOffFlags);
    const bool has_mp = (flags & 1u) != 0;
    // the record as received
    const size_t bytes = ((size_t)kHdrBytes + (size_t)n * (has_mp ? 52 : 48) + 15) / 16 * 16;
    uint4* dst = reinterpret_cast<uint4*>(S.rec + (size_t)slot * S.rec_stride);
    const uint4* s4 = reinterpret_cast<const uint4*>(r);
    for (size_t k = threadIdx.x; k < bytes / 16; k += 256) dst[k] = s4[k];
    const int nv = compact_rows(r, n, has_mp, S.vdesc + (size_t)slot * S.kp * 8, S.vidx + (size_t)slot * S.kp,
                                S.angle + (size_t)slot * S.kp, s_cnt);
    if (threadIdx.x == 0) {
        S.nv[slot] = nv;
        S.agent[slot] = *reinterpret_cast<const int32_t*>(r + kOffAgent);
    }
}

__global__ __launch_bounds__(256) void kf_compact_query_kernel(const uint8_t* __restrict__ r, int kp, uint32_t* __restrict__ qdesc,
                                                               uint16_t* __restrict__ qidx, int32_t* __restrict__ nqv) {
    __shared__ int s_cnt[4];
    int n = *reinterpret_cast<const int32_t*>(r + kOffN);
    n = n < 0 ? 0 : (n > kp ? kp : n);
    const uint32_t flags = *reinterpret_cast<const uint32_t*>(r + kOffFlags);
    const int nv = compact_rows(r, n, (flags & 1u) != 0, qdesc, qidx, nullptr, s_cnt);
    if (threadIdx.x == 0) *nqv = nv;
}

// Phase 1.  grid: (slots x query chunks) rounded up to a multiple of 8, re-mapped so that consecutive logical
// workgroups - the chunks of one keyframe, then the next keyframe - run on ONE XCD and share its L2.
template <int Q>
__global__ __launch_bounds__(256) void kf_scan_kernel(const uint32_t* __restrict__ vdesc, const int32_t* __restrict__ nv,
                                                      const int32_t* __restrict__ agent, int kp, int n_slots, int chunks,
                                                      const uint32_t* __restrict__ qdesc, int nqv, int query_agent,
                                                      int th_low, float nn_ratio, int32_t* __restrict__ votes) {
    const int per_xcd = gridDim.x >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= n_slots * chunks) return;
    const int slot = logical / chunks, chunk = logical - slot * chunks;
    const int a = agent[slot];
    const int n = nv[slot];
    if (a < 0 || a == query_agent || n <= 0) return;
    const int q0 = chunk * 256 * Q + threadIdx.x;
    if (chunk * 256 * Q + (int)(threadIdx.x & ~63u) >= nqv) return;  // a wave without a single query row
    uint32_t q[Q][8];
#pragma unroll
    for (int j = 0; j < Q; j++) {
        const int qi = q0 + j * 256;
        uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
        if (qi < nqv) {
            lo = reinterpret_cast<const uint4*>(qdesc)[2 * (size_t)qi];
            hi = reinterpret_cast<const uint4*>(qdesc)[2 * (size_t)qi + 1];
        }
        q[j][0] = lo.x; q[j][1] = lo.y; q[j][2] = lo.z; q[j][3] = lo.w;
        q[j][4] = hi.x; q[j][5] = hi.y; q[j][6] = hi.z; q[j][7] = hi.w;
    }
    int best[Q], second[Q];
#pragma unroll
    for (int j = 0; j < Q; j++) best[j] = second[j] = 256;
    ConstU32Ptr T = (ConstU32Ptr)(uintptr_t)(vdesc + (size_t)slot * kp * 8);
    // one store row against the lane's Q query rows; the row's eight words are wave-uniform (SGPR operands of the XORs)
    auto step = [&](const uint32_t w0, const uint32_t w1, const uint32_t w2, const uint32_t w3, const uint32_t w4,
                    const uint32_t w5, const uint32_t w6, const uint32_t w7) {
#pragma unroll
        for (int j = 0; j < Q; j++) {
            // v_bcnt_u32_b32 adds its second operand: a chain of eight is the whole distance (written out because the
            // optimizer re-associates `popc + popc + ...` into a tree of plain popcounts and three-way adds: 23
            // instructions per pair instead of 18)
            int d = bcnt_acc(q[j][0] ^ w0, 0);
            d = bcnt_acc(q[j][1] ^ w1, d);
            d = bcnt_acc(q[j][2] ^ w2, d);
            d = bcnt_acc(q[j][3] ^ w3, d);
            d = bcnt_acc(q[j][4] ^ w4, d);
            d = bcnt_acc(q[j][5] ^ w5, d);
            d = bcnt_acc(q[j][6] ^ w6, d);
            d = bcnt_acc(q[j][7] ^ w7, d);
            // best <= second always: the new second is the median of (best, second, d), the new best the minimum -
            // the scan of ORBmatcher.cc:543-549 on distances alone
            second[j] = med3_i32(best[j], second[j], d);
            best[j] = min(best[j], d);
        }
    };
    int t = 0;
    constexpr int kRows = Q == 1 ? 8 : 4;  // rows per trip: 64 / 32 dwords of scalar loads in flight under 144 / 72 Q vector instructions
    for (; t + kRows <= n; t += kRows) {
        ConstU32Ptr r = T + (size_t)t * 8;
        uint32_t w[8 * kRows];
#pragma unroll
        for (int k = 0; k < 8 * kRows; k++) w[k] = r[k];
#pragma unroll
        for (int u = 0; u < kRows; u++)
            step(w[8 * u], w[8 * u + 1], w[8 * u + 2], w[8 * u + 3], w[8 * u + 4], w[8 * u + 5], w[8 * u + 6], w[8 * u + 7]);
    }
    for (; t < n; t++) {
        ConstU32Ptr r = T + (size_t)t * 8;
        step(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    }
    int c = 0;
#pragma unroll
    for (int j = 0; j < Q; j++) {
        const bool v = (q0 + j * 256 < nqv) && best[j] < th_low && (float)best[j] < nn_ratio * (float)second[j];
        c += __popcll(__ballot(v));
    }
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&votes[slot], c);
}

// Phase 2.  grid (ceil(nqv / 4), n_cand), 4 waves, a query row per wave; dynamic LDS: 4 x kp keys.
__global__ __launch_bounds__(256) void kf_pair_topk_kernel(KfStoreDev S, const int32_t* __restrict__ cand,
                                                           const uint32_t* __restrict__ qdesc, int nqv, int K,
                                                           uint32_t* __restrict__ keys) {
    extern __shared__ uint32_t s_keys[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= nqv) return;  // wave-uniform; the workgroup never meets at a barrier
    const int c = blockIdx.y;
    const int slot = cand[c];
    const int n = S.nv[slot];
    uint32_t* L = s_keys + (size_t)wave * S.kp;
    const uint4 q0 = reinterpret_cast<const uint4*>(qdesc)[2 * (size_t)q], q1 = reinterpret_cast<const uint4*>(qdesc)[2 * (size_t)q + 1];
    const uint4* rows = reinterpret_cast<const uint4*>(S.vdesc + (size_t)slot * S.kp * 8);
    const uint16_t* idx = S.vidx + (size_t)slot * S.kp;
    for (int t = lane; t < n; t += 64) {
        const int d = hamming_row(q0, q1, rows[2 * t], rows[2 * t + 1]);
        L[t] = ((uint32_t)d << 16) | (uint32_t)idx[t];  // keypoint indices ascend with t: ties go to the first row scanned
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t* out = keys + ((size_t)c * nqv + q) * K;
    uint32_t prev = 0;
    for (int k = 0; k < K; k++) {
        uint32_t m = 0xFFFFFFFFu;
        for (int t = lane; t < n; t += 64) {
            const uint32_t v = L[t];
            if ((k == 0 || v > prev) && v < m) m = v;
        }
        m = wave_min_u32(m);
        if (lane == 0) out[k] = m;
        if (m == 0xFFFFFFFFu) {
            if (lane == 0)
                for (int kk = k + 1; kk < K; kk++) out[kk] = 0xFFFFFFFFu;
            break;
        }
        prev = m;
    }
}

__global__ __launch_bounds__(64) void kf_pair_rerun_kernel(KfStoreDev S, int slot, const uint32_t* __restrict__ qdesc, int q,
                                                           const uint32_t* __restrict__ taken, uint32_t* __restrict__ keys2) {
    const int lane = threadIdx.x;
    const int n = S.nv[slot];
    const uint4 q0 = reinterpret_cast<const uint4*>(qdesc)[2 * (size_t)q], q1 = reinterpret_cast<const uint4*>(qdesc)[2 * (size_t)q + 1];
    const uint4* rows = reinterpret_cast<const uint4*>(S.vdesc + (size_t)slot * S.kp * 8);
    const uint16_t* idx = S.vidx + (size_t)slot * S.kp;
    uint32_t lb = 0xFFFFFFFFu, ls = 0xFFFFFFFFu;
    for (int t = lane; t < n; t += 64) {
        const uint32_t i = idx[t];
        if ((taken[i >> 5] >> (i & 31)) & 1u) continue;  // vbMatched2[idx2], ORBmatcher.cc:537
        const uint32_t key = ((uint32_t)hamming_row(q0, q1, rows[2 * t], rows[2 * t + 1]) << 16) | i;
        ls = med3_u32(lb, ls, key);
        lb = min(lb, key);
    }
    const uint32_t b = wave_min_u32(lb);
    const uint32_t s = wave_min_u32(lb == b ? ls : lb);  // keys are unique: the lane that owns the best offers its second
    if (lane == 0) {
        keys2[0] = b;
        keys2[1] = s;
    }
}

// staged = [header 128 B | angle f32 n | map_point_id i32 n]; out = version-2 record, zero padded to out_bytes
__global__ __launch_bounds__(1024) void kf_pack_record_kernel(const uint8_t* __restrict__ desc, const float2* __restrict__ xy_un,
                                                              const int8_t* __restrict__ octave, const uint8_t* __restrict__ staged,
                                                              int n, uint8_t* __restrict__ out, size_t out_bytes) {
    __shared__ unsigned long long s_part[16];
    __shared__ int s_bound[16];
    const int tid = threadIdx.x;
    const float* angle = reinterpret_cast<const float*>(staged + kHdrBytes);
    const int32_t* mp = reinterpret_cast<const int32_t*>(staged + kHdrBytes + (size_t)n * 4);
    uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
    const size_t words = out_bytes / 4, hdr_w = kHdrBytes / 4;
    const size_t desc_w = (size_t)n * 8, geo_w = (size_t)n * 4;
    unsigned long long acc = 0;
    int bound = 0;
    // one pass over the body as 32-bit words: word k of the record body (behind the header) is byte 4k..4k+3 of the
    // checksummed range
    for (size_t k = tid; k < words - hdr_w; k += 1024) {
        uint32_t v = 0;
        if (k < desc_w) {
            v = reinterpret_cast<const uint32_t*>(desc)[k];
        } else if (k < desc_w + geo_w) {
            const size_t g = k - desc_w, i = g >> 2;
            const int f = (int)(g & 3);
            v = f == 0 ? __float_as_uint(xy_un[i].x) : f == 1 ? __float_as_uint(xy_un[i].y) : f == 2 ? __float_as_uint(angle[i])
                                                                                                     : (uint32_t)(int32_t)octave[i];
        } else if (k < desc_w + geo_w + (size_t)n) {
            const int32_t id = mp[k - desc_w - geo_w];
            v = (uint32_t)id;
            bound += id >= 0 ? 1 : 0;
        }
        o32[hdr_w + k] = v;
        if (k < desc_w + geo_w + (size_t)n) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned long long i = (unsigned long long)k * 4 + b;
                acc += (unsigned long long)((v >> (8 * b)) & 0xFFu) * (i % 65521ull + 1ull);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        acc += __shfl_xor(acc, off);
        bound += __shfl_xor(bound, off);
    }
    if ((tid & 63) == 0) {
        s_part[tid >> 6] = acc;
        s_bound[tid >> 6] = bound;
    }
    __syncthreads();
    if (tid < 32) {  // header: the staged one with the fields pack2 fills
        uint32_t v = reinterpret_cast<const uint32_t*>(staged)[tid];
        unsigned long long tot = 0;
        int nb = 0;
        for (int w = 0; w < 16; w++) {
            tot += s_part[w];
            nb += s_bound[w];
        }
        tot %= ((1ull << 61) - 1ull);
        if (tid == 0) v = 0x464B4F53u;                       // "SOKF"
        if (tid == 1) v = 2u | ((uint32_t)kHdrBytes << 16);  // version 2, header_bytes
        if (tid == kOffN / 4) v = (uint32_t)n;
        if (tid == kOffChecksum / 4) v = (uint32_t)(tot & 0xFFFFFFFFull);
        if (tid == kOffChecksum / 4 + 1) v = (uint32_t)(tot >> 32);
        if (tid == kOffFlags / 4) v = 1u;
        if (tid == kOffBound / 4) v = (uint32_t)nb;
        o32[tid] = v;
    }
}

}  // namespace

void launch_kf_append(const KfStoreDev& S, const uint8_t* d_src, size_t src_stride, const int32_t* d_job_slot, int n_jobs,
                      hipStream_t s) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(kf_append_kernel, dim3(n_jobs), dim3(256), 0, s, S, d_src, src_stride, d_job_slot);
}

void launch_kf_compact_query(const uint8_t* d_rec, int kp, uint32_t* d_qdesc, uint16_t* d_qidx, int32_t* d_nqv, hipStream_t s) {
    hipLaunchKernelGGL(kf_compact_query_kernel, dim3(1), dim3(256), 0, s, d_rec, kp, d_qdesc, d_qidx, d_nqv);
}

void launch_kf_pack_record(const uint8_t* d_desc, const float2* d_xy_un, const int8_t* d_octave, const uint8_t* d_staged,
                           int n, uint8_t* d_out, size_t out_bytes, hipStream_t s) {
    hipLaunchKernelGGL(kf_pack_record_kernel, dim3(1), dim3(1024), 0, s, d_desc, d_xy_un, d_octave, d_staged, n, d_out, out_bytes);
}

void launch_kf_scan(const KfStoreDev& S, int n_slots, const uint32_t* d_qdesc, int nqv, int query_agent, int th_low,
                    float nn_ratio, int32_t* d_votes, int qper, hipStream_t s) {
    if (n_slots <= 0 || nqv <= 0) return;
    // one query row per lane measures best or equal at every size tried (tools/kfscan_bench.py): the scalar row loads are
    // not what bounds the kernel, and fewer rows per lane leave fewer idle lanes in the last wave
    if (qper != 1 && qper != 2 && qper != 4) qper = 1;
    const int chunks = (nqv + 256 * qper - 1) / (256 * qper);
    const int grid = (n_slots * chunks + 7) / 8 * 8;
#define SO_KF_SCAN(Q)                                                                                                  \
    hipLaunchKernelGGL(kf_scan_kernel<Q>, dim3(grid), dim3(256), 0, s, S.vdesc, S.nv, S.agent, S.kp, n_slots, chunks, \
                       d_qdesc, nqv, query_agent, th_low, nn_ratio, d_votes)
    if (qper == 4) SO_KF_SCAN(4);
    else if (qper == 2) SO_KF_SCAN(2);
    else SO_KF_SCAN(1);
#undef SO_KF_SCAN
}

void launch_kf_pair_topk(const KfStoreDev& S, const int32_t* d_cand, int n_cand, const uint32_t* d_qdesc, int nqv, int K,
                         uint32_t* d_keys, hipStream_t s) {
    if (n_cand <= 0 || nqv <= 0) return;
    static bool attr_set[64] = {};  // more than 64 KB of dynamic LDS has to be asked for once per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf_pair_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(uint32_t) * 4 * kKfMaxKeypoints));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kf_pair_topk_kernel, dim3((nqv + 3) / 4, n_cand), dim3(256), sizeof(uint32_t) * 4 * (size_t)S.kp, s, S,
                       d_cand, d_qdesc, nqv, K, d_keys);
}

void launch_kf_pair_rerun(const KfStoreDev& S, int slot, const uint32_t* d_qdesc, int q, const uint32_t* d_taken,
                          uint32_t* d_keys2, hipStream_t s) {
    hipLaunchKernelGGL(kf_pair_rerun_kernel, dim3(1), dim3(64), 0, s, S, slot, d_qdesc, q, d_taken, d_keys2);
}

}  // namespace so
