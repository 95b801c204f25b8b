// quadtree_kernel.hip — ORBextractor::DistributeOctTree (code/src/ORBextractor.cc:407-689) on the device.
//
// One 1024-thread workgroup per pyramid level, all levels in one launch; the whole tree lives in LDS.  The
// reference walks a std::list and splits one node at a time; the same result is produced here in bulk-synchronous
// "split steps", each of which splits a whole set of nodes at once:
//   * sweep  (ORBextractor.cc:524-593): every non-leaf node of the list is split, in list order;
//   * careful round (:599-664): the non-leaf nodes are ranked by (population, creation sequence) descending (the
//     pointer tie-break of :610 is defined as creation sequence), the number of non-empty children
//     of each is known before splitting, so a prefix sum tells how many of them the reference would have split
//     before its `size >= N` break; exactly those are split, in that order.
// In both cases the reference pushes the non-empty children n1..n4 of each processed node to the FRONT of the
// list, so after a step the list is [children of the last processed node (n4..n1), ..., children of the first
// processed node, then the untouched nodes in their old order]; that position is computed with prefix sums and is
// also the node's slot in LDS (slot == list position).  Keys (candidate indices) stay in one array in which every
// node owns a contiguous, order-preserving slice; a split is a stable 4-way partition of the parent's slice,
// ranked with a block-wide scan of packed 4x16-bit quadrant counters.
// Output per level: the best-response key of every node in list order (first maximum, :667-686).
#include "orb_device.h"

namespace so {

constexpr int kQtThreads = 1024;
constexpr int kQtMaxKeys = kFastCap;  // 10000 candidates per level at most
constexpr int kQtMaxNodes = 1024;     // list never exceeds N + 3 (N <= 1020 on this path)
static_assert((kQtMaxKeys + kQtThreads - 1) / kQtThreads <= 16, "a thread's keys must fit the 4-bit codes of one 64-bit register");

struct QtNode {  // 16 bytes
    int16_t x0, y0, x1, y1;
    uint16_t off, n;  // slice of the key array
    uint32_t seq;     // creation sequence (tie-break of the careful phase)
};

typedef unsigned long long u64;

__device__ __forceinline__ int qt_wave_excl_scan(int v, int lane) {
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
    }
    return incl;
}

// block-wide exclusive scan of one int per thread; returns the exclusive prefix, *total = sum over the block
__device__ __forceinline__ int qt_block_scan(int v, int* s_w, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int incl = qt_wave_excl_scan(v, lane);
    __syncthreads();
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int base = 0, t = 0;
#pragma unroll
    for (int i = 0; i < kQtThreads / 64; i++) {
        const int x = s_w[i];
        if (i < w) base += x;
        t += x;
    }
    *total = t;
    return base + incl - v;
}

__device__ __forceinline__ u64 qt_block_scan64(u64 v, u64* s_w) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u64 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 u = (u64)__shfl_up((long long)incl, off);
        if (lane >= off) incl += u;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u64 base = 0;
#pragma unroll
    for (int i = 0; i < kQtThreads / 64; i++)
        if (i < w) base += s_w[i];
    return base + incl - v;
}

// the same, also returning the block total
__device__ __forceinline__ u64 qt_block_scan64_total(u64 v, u64* s_w, u64* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u64 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 u = (u64)__shfl_up((long long)incl, off);
        if (lane >= off) incl += u;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u64 base = 0, t = 0;
#pragma unroll
    for (int i = 0; i < kQtThreads / 64; i++) {
        const u64 x = s_w[i];
        if (i < w) base += x;
        t += x;
    }
    *total = t;
    return base + incl - v;
}

__device__ __forceinline__ int qt_unpack(u64 v, int q) { return (int)((v >> (16 * q)) & 0xFFFFull); }

struct QtLevelArgs {
    int n_target[kMaxLevels];  // mnFeaturesPerLevel
    int sel_stride;            // slots per level in the output
};

__global__ __launch_bounds__(kQtThreads) void quadtree_kernel(PyramidParams P, QtLevelArgs A,
                                                               const Candidate* __restrict__ cands,
                                                               const CandidateHeader* __restrict__ hdr,
                                                               SelectedKp* __restrict__ sel_out,
                                                               int32_t* __restrict__ count_out) {
    __shared__ uint16_t s_keys[2][kQtMaxKeys];
    __shared__ uint16_t s_knode[2][kQtMaxKeys];
    __shared__ QtNode s_nodes[2][kQtMaxNodes];
    __shared__ u64 s_pre[kQtMaxNodes], s_post[kQtMaxNodes];
    __shared__ uint16_t s_child[kQtMaxNodes][4];
    __shared__ uint16_t s_newslot[kQtMaxNodes];
    __shared__ int s_proc[kQtMaxNodes];     // processing rank of a node in this step, -1 = not split
    __shared__ int s_kv[kQtMaxNodes];       // by processing rank: number of non-empty children, then its prefix
    __shared__ u64 s_w64[kQtThreads / 64];
    __shared__ int s_wi[kQtThreads / 64];
    __shared__ int s_misc[8];

    const int tid = threadIdx.x;
    const int lvl = blockIdx.x;
    const LevelDesc& L = P.lv[lvl];
    const int n = hdr->count[lvl];
    const Candidate* C = cands + hdr->offset[lvl];
    const int N = A.n_target[lvl];
    SelectedKp* out = sel_out + (size_t)lvl * A.sel_stride;
    if (n <= 0) {
        if (tid == 0) count_out[lvl] = 0;
        return;
    }
    const int W = L.w - 2 * kFastBorder, H = L.h - 2 * kFastBorder;
    const int chunk = (n + kQtThreads - 1) / kQtThreads;
    const int p0 = min(tid * chunk, n), p1 = min(p0 + chunk, n);

    // ---------------- roots (:468-511) ----------------
    int n_ini = (int)roundf((float)W / (float)H);
    if (n_ini < 1) n_ini = 1;
    const float hX = (float)W / (float)n_ini;
    int b = 0;  // current buffer
    {
        u64 mine = 0;
        for (int p = p0; p < p1; p++) {
            int r = (int)((float)C[p].x / hX);
            r = min(r, n_ini - 1);
            mine += 1ull << (16 * r);
        }
        const u64 excl = qt_block_scan64(mine, s_w64);
        if (tid == kQtThreads - 1) s_pre[0] = excl + mine;  // totals per root
        __syncthreads();
        const u64 tot = s_pre[0];
        int off_r[4], cnt_r[4], slot_r[4], m0 = 0, o = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            cnt_r[r] = r < n_ini ? qt_unpack(tot, r) : 0;
            off_r[r] = o;
            o += cnt_r[r];
            slot_r[r] = m0;
            if (cnt_r[r] > 0) m0++;
        }
        u64 run = excl;
        for (int p = p0; p < p1; p++) {
            int r = (int)((float)C[p].x / hX);
            r = min(r, n_ini - 1);
            const int pos = off_r[r] + qt_unpack(run, r);
            s_keys[0][pos] = (uint16_t)p;
            s_knode[0][pos] = (uint16_t)slot_r[r];
            run += 1ull << (16 * r);
        }
        if (tid < 4 && tid < n_ini && cnt_r[tid] > 0) {
            QtNode nd;
            nd.x0 = (int16_t)(int)(hX * (float)tid);
            nd.x1 = (int16_t)(int)(hX * (float)(tid + 1));
            nd.y0 = 0;
            nd.y1 = (int16_t)H;
            nd.off = (uint16_t)off_r[tid];
            nd.n = (uint16_t)cnt_r[tid];
            nd.seq = (uint32_t)tid;
            s_nodes[0][slot_r[tid]] = nd;
        }
        if (tid == 0) {
            s_misc[0] = m0;  // list size
            s_misc[1] = 4;   // next creation sequence
        }
        __syncthreads();
    }

    int m = s_misc[0];
    uint32_t seq_base = 4;
    bool careful = false;
    // ---------------- split steps ----------------
    for (int guard = 0; guard < 64; guard++) {
        const QtNode* nodes = s_nodes[b];
        QtNode* nnodes = s_nodes[b ^ 1];
        const uint16_t* keys = s_keys[b];
        const uint16_t* knode = s_knode[b];
        // (B) quadrant of every key that sits in a non-leaf node; packed counts; prefix at the slice borders.
        // The quadrant of a key costs a chain of dependent reads (key -> node slot -> node box; key -> candidate in
        // global memory): it is worked out ONCE per step and kept as a 4-bit code (0 = leaf node, q + 1 otherwise) in a
        // 64-bit register - a thread owns at most ten keys (10000 candidates / 1024 threads) - for the count, the
        // prefix and the partition below (three such chains per step before: 42 -> see DESIGN 3)
        u64 mine = 0, qcode = 0;
        for (int p = p0; p < p1; p++) {
            const QtNode nd = nodes[knode[p]];
            if (nd.n > 1) {
                const Candidate c = C[keys[p]];
                const int xm = nd.x0 + (int)ceilf((float)(nd.x1 - nd.x0) / 2), ym = nd.y0 + (int)ceilf((float)(nd.y1 - nd.y0) / 2);
                const int q = (c.x < xm ? 0 : 1) + (c.y < ym ? 0 : 2);
                mine += 1ull << (16 * q);
                qcode |= (u64)(q + 1) << (4 * (p - p0));
            }
        }
        const u64 excl = qt_block_scan64(mine, s_w64);
        {
            u64 run = excl;
            for (int p = p0; p < p1; p++) {
                const int qc = (int)((qcode >> (4 * (p - p0))) & 15ull);
                if (qc) {
                    const int ni = knode[p];
                    const QtNode nd = nodes[ni];
                    if (p == nd.off) s_pre[ni] = run;
                    run += 1ull << (16 * (qc - 1));
                    if (p == nd.off + nd.n - 1) s_post[ni] = run;
                }
            }
        }
        __syncthreads();
        // per node: children populations
        int cnt[4] = {0, 0, 0, 0}, k_children = 0;
        bool nonleaf = false;
        QtNode me;
        if (tid < m) {
            me = nodes[tid];
            nonleaf = me.n > 1;
            if (nonleaf) {
                const u64 d = s_post[tid] - s_pre[tid];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    cnt[q] = qt_unpack(d, q);
                    k_children += cnt[q] > 0;
                }
            }
        }
        // (A/C/D) which nodes are split and in which order; children prefix in processing order; rank of the untouched
        // nodes in list order; how many of the new nodes can be split again
        const int ne = nonleaf ? (cnt[0] > 1) + (cnt[1] > 1) + (cnt[2] > 1) + (cnt[3] > 1) : 0;
        int nsplit = 0, cp = 0, ur = 0, total_children = 0, n_untouched = 0, n_to_expand = 0;
        bool split = false;
        if (!careful) {
            // sweep: every non-leaf node is split, in list order - processing rank, children prefix, untouched rank and the
            // count of splittable children are four 16-bit fields of ONE block scan (four scans and six barriers before)
            const u64 v = tid < m ? ((u64)(nonleaf ? 1 : 0) | ((u64)(nonleaf ? k_children : 0) << 16) | ((u64)(nonleaf ? 0 : 1) << 32) |
                                     ((u64)ne << 48))
                                  : 0ull;
            u64 tot = 0;
            const u64 ex = qt_block_scan64_total(v, s_w64, &tot);
            nsplit = qt_unpack(tot, 0);
            total_children = qt_unpack(tot, 1);
            n_untouched = qt_unpack(tot, 2);
            n_to_expand = qt_unpack(tot, 3);
            cp = qt_unpack(ex, 1);
            ur = qt_unpack(ex, 2);
            split = tid < m && nonleaf;
            if (tid < m) s_proc[tid] = nonleaf ? qt_unpack(ex, 0) : -1;
            __syncthreads();
        } else {
            // rank by (population, seq) descending among the non-leaf nodes (all of them were created last step)
            int rank = -1;
            if (nonleaf) {
                rank = 0;
                const uint32_t myn = me.n, mys = me.seq;
                for (int j = 0; j < m; j++) {
                    const QtNode o = nodes[j];
                    if (o.n > 1 && (o.n > myn || (o.n == myn && o.seq > mys))) rank++;
                }
            }
            int n_nonleaf = 0;
            (void)qt_block_scan(nonleaf ? 1 : 0, s_wi, &n_nonleaf);
            if (tid < kQtMaxNodes) s_kv[tid] = 0;
            __syncthreads();
            if (nonleaf) s_kv[rank] = k_children - 1;  // growth of the list when this node is split
            __syncthreads();
            int tot_unused = 0;
            const int g = tid < n_nonleaf ? s_kv[tid] : 0;
            const int gpre = qt_block_scan(g, s_wi, &tot_unused);
            // the reference stops right after the first split that makes size >= N (:655-661)
            const bool reaches = tid < n_nonleaf && (m + gpre + g >= N);
            if (tid == 0) s_misc[2] = n_nonleaf;  // default: all of them
            __syncthreads();
            if (reaches) atomicMin(&s_misc[2], tid + 1);
            __syncthreads();
            nsplit = s_misc[2];
            if (tid < m) s_proc[tid] = (nonleaf && rank < nsplit) ? rank : -1;
            __syncthreads();
            split = tid < m && nonleaf && s_proc[tid] >= 0;
            // children prefix in processing order
            if (tid < kQtMaxNodes) s_kv[tid] = 0;
            __syncthreads();
            if (split) s_kv[s_proc[tid]] = k_children;
            __syncthreads();
            const int kv = tid < nsplit ? s_kv[tid] : 0;
            const int cpre_sorted = qt_block_scan(kv, s_wi, &total_children);
            __syncthreads();
            if (tid < nsplit) s_kv[tid] = cpre_sorted;
            ur = qt_block_scan((tid < m && !split) ? 1 : 0, s_wi, &n_untouched);
            __syncthreads();
            if (split) cp = s_kv[s_proc[tid]];
            (void)qt_block_scan(split ? ne : 0, s_wi, &n_to_expand);
        }
        const int m_new = total_children + n_untouched;
        // (E) new nodes; slot == position in the new list
        if (tid < m) {
            if (split) {
                const int hx = (int)ceilf((float)(me.x1 - me.x0) / 2), hy = (int)ceilf((float)(me.y1 - me.y0) / 2);
                const int xm = me.x0 + hx, ym = me.y0 + hy;
                int r = 0, o = me.off;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (cnt[q] > 0) {
                        const int pos = total_children - 1 - (cp + r);
                        QtNode nd;
                        nd.x0 = (int16_t)((q & 1) ? xm : me.x0);
                        nd.x1 = (int16_t)((q & 1) ? me.x1 : xm);
                        nd.y0 = (int16_t)((q & 2) ? ym : me.y0);
                        nd.y1 = (int16_t)((q & 2) ? me.y1 : ym);
                        nd.off = (uint16_t)o;
                        nd.n = (uint16_t)cnt[q];
                        nd.seq = seq_base + 4u * (uint32_t)s_proc[tid] + (uint32_t)q;
                        nnodes[pos] = nd;
                        s_child[tid][q] = (uint16_t)pos;
                        r++;
                    }
                    o += cnt[q];
                }
            } else {
                const int pos = total_children + ur;
                nnodes[pos] = me;
                s_newslot[tid] = (uint16_t)pos;
            }
        }
        __syncthreads();
        // (F) stable 4-way partition of the keys of the split nodes
        {
            uint16_t* keys2 = s_keys[b ^ 1];
            uint16_t* knode2 = s_knode[b ^ 1];
            u64 run = excl;
            for (int p = p0; p < p1; p++) {
                const int ni = knode[p];
                const int q = (int)((qcode >> (4 * (p - p0))) & 15ull) - 1;
                if (q >= 0 && s_proc[ni] >= 0) {
                    const int slot = s_child[ni][q];
                    const int rank = qt_unpack(run, q) - qt_unpack(s_pre[ni], q);
                    const int pos = nnodes[slot].off + rank;
                    keys2[pos] = keys[p];
                    knode2[pos] = (uint16_t)slot;
                } else {
                    keys2[p] = keys[p];
                    knode2[p] = s_newslot[ni];
                }
                if (q >= 0) run += 1ull << (16 * q);
            }
        }
        __syncthreads();
        b ^= 1;
        seq_base += 4u * (uint32_t)nsplit;
        const int m_prev = m;
        m = m_new;
        // termination (:595-598, :650-664)
        if (m >= N || m == m_prev) break;
        if (!careful && (m + n_to_expand * 3 > N)) careful = true;
    }

    // ---------------- best response per node, list order (:667-686) ----------------
    // The reference keeps the FIRST maximum of `response` in the node's insertion order.  A thread per node walking its
    // keys would chain one global read per key (16 on average at level 0); instead every key fetches its own score - all
    // in flight together - and the node's winner is an LDS atomicMax over (score << 16 | 0xFFFF - rank in the node):
    // highest score, lowest rank on ties.
    int* s_best = s_proc;  // (dead after the last split step)
    if (tid < kQtMaxNodes) s_best[tid] = -1;
    __syncthreads();
    {
        const uint16_t* keys = s_keys[b];
        const uint16_t* knode = s_knode[b];
        const QtNode* nodes = s_nodes[b];
        for (int p = p0; p < p1; p++) {
            const int ni = knode[p];
            const int sc = C[keys[p]].score;
            atomicMax(&s_best[ni], (sc << 16) | (0xFFFF - (p - (int)nodes[ni].off)));
        }
    }
    __syncthreads();
    if (tid < m) {
        const QtNode nd = s_nodes[b][tid];
        const int v = s_best[tid];
        const int best = s_keys[b][nd.off + (0xFFFF - (v & 0xFFFF))];
        const int best_score = v >> 16;
        SelectedKp o;
        o.x = (int16_t)(C[best].x + kFastBorder);  // addBorder_kernel, Fast_gpu.cu:461-470
        o.y = (int16_t)(C[best].y + kFastBorder);
        o.level = (uint16_t)lvl;
        o.score = (uint16_t)best_score;
        if (tid < A.sel_stride) out[tid] = o;
    }
    if (tid == 0) count_out[lvl] = min(m, A.sel_stride);
}

void launch_quadtree(const PyramidParams& p, const int* n_target, int sel_stride, const Candidate* d_cands,
                     const CandidateHeader* d_hdr, SelectedKp* d_sel, int32_t* d_count, hipStream_t s) {
    QtLevelArgs a;
    for (int l = 0; l < kMaxLevels; l++) a.n_target[l] = l < p.nlevels ? n_target[l] : 0;
    a.sel_stride = sel_stride;
    hipLaunchKernelGGL(quadtree_kernel, dim3(p.nlevels), dim3(kQtThreads), 0, s, p, a, d_cands, d_hdr, d_sel, d_count);
}

}  // namespace so
