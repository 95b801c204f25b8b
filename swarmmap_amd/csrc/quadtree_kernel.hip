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
// also the node's slot in LDS (slot == list position).  Keys stay in one array in which every node owns a contiguous,
// order-preserving slice; a split is a stable 4-way partition of the parent's slice, ranked with a block-wide scan of
// packed 4x16-bit quadrant counters.
// Output per level: the best-response key of every node in list order (first maximum, :667-686).
//
// The candidate list the reference builds in front of this (GpuFast's kpLoc array, raster order, capped at 10000 per
// level) is never materialised: a key IS the candidate's position (x | y << 16, ROI coordinates), read straight off the
// keep bitmap in raster order by the level's workgroup, and a key's FAST response is read from the score map once, at
// the very end.  Rounds 1-2 had a per-band emit kernel write 8-byte records that every split step then gathered from
// global memory through the key (12-15 us for the launch, a memory round trip per step).  A thread holds the keys of
// its chunk of positions in registers across a step, so keys, key -> node and the node table are single-buffered.
#include "orb_device.h"

namespace so {

constexpr int kQtThreads = 1024;
constexpr int kQtMaxKeys = kFastCap;  // 10000 candidates per level at most
constexpr int kQtMaxNodes = 1024;     // list never exceeds N + 3 (N <= 1020 on this path)
constexpr int kQtChunk = (kQtMaxKeys + kQtThreads - 1) / kQtThreads;  // positions per thread at most (10)
static_assert(kQtChunk <= 16, "a thread's keys must fit the 4-bit codes of one 64-bit register");

struct QtNode {  // 16 bytes
    int16_t x0, y0, x1, y1;
    uint16_t off, n;  // slice of the key array
    uint32_t seq;     // creation sequence (tie-break of the careful phase)
};

typedef unsigned long long u64;

// tools/probe/qt_probe.hip builds this file with -DQT_TIMING: shader-clock stamps of the level-0 workgroup
#ifdef QT_TIMING
__device__ long long qt_stamps[96];
#define QT_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) qt_stamps[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define QT_T(i) do { } while (0)
#endif

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

// LDS map (bytes): keys 40000 | key -> node 2 x 20000 | nodes 16384 | pre, post 2 x 8192 | child 8192 | newslot 2048 |
// proc 4096 | kv 4096 | scan scratch.  While the key array is built, the level's keep words sit (in raster order) in
// the space behind the keys.
constexpr int kQtOffKnode = kQtMaxKeys * 4;
constexpr int kQtOffNodes = kQtOffKnode + 2 * kQtMaxKeys * 2;
constexpr int kQtOffPre = kQtOffNodes + kQtMaxNodes * 16;
constexpr int kQtOffPost = kQtOffPre + kQtMaxNodes * 8;
constexpr int kQtOffChild = kQtOffPost + kQtMaxNodes * 8;
constexpr int kQtOffNewslot = kQtOffChild + kQtMaxNodes * 8;
constexpr int kQtOffProc = kQtOffNewslot + kQtMaxNodes * 2;
constexpr int kQtOffKv = kQtOffProc + kQtMaxNodes * 4;
constexpr int kQtSmemBytes = kQtOffKv + kQtMaxNodes * 4;
constexpr int kQtWordCache = (kQtSmemBytes - kQtOffKnode) / 4;  // keep words the LDS can hold next to the keys
static_assert(sizeof(QtNode) == 16 && kQtOffKnode % 16 == 0 && kQtOffNodes % 16 == 0 && kQtOffPre % 8 == 0, "LDS map");

__global__ __launch_bounds__(kQtThreads) void quadtree_kernel(PyramidParams P, QtLevelArgs A,
                                                               SelectedKp* __restrict__ sel_out,
                                                               int32_t* __restrict__ count_out) {
    __shared__ __align__(16) unsigned char smem[kQtSmemBytes];
    __shared__ u64 s_w64[kQtThreads / 64];
    __shared__ int s_wi[kQtThreads / 64];
    __shared__ int s_misc[8];
    uint32_t* s_key = reinterpret_cast<uint32_t*>(smem);                       // x | y << 16 (ROI coordinates, as GpuFast's kpLoc)
    uint16_t* s_knode2 = reinterpret_cast<uint16_t*>(smem + kQtOffKnode);      // [2][kQtMaxKeys]: key position -> node slot
    QtNode* nodes = reinterpret_cast<QtNode*>(smem + kQtOffNodes);
    u64* s_pre = reinterpret_cast<u64*>(smem + kQtOffPre);
    u64* s_post = reinterpret_cast<u64*>(smem + kQtOffPost);
    uint16_t(*s_child)[4] = reinterpret_cast<uint16_t(*)[4]>(smem + kQtOffChild);
    uint16_t* s_newslot = reinterpret_cast<uint16_t*>(smem + kQtOffNewslot);
    int* s_proc = reinterpret_cast<int*>(smem + kQtOffProc);  // processing rank of a node in this step, -1 = not split
    int* s_kv = reinterpret_cast<int*>(smem + kQtOffKv);      // by processing rank: number of non-empty children, then its prefix
    uint32_t* s_words = reinterpret_cast<uint32_t*>(smem + kQtOffKnode);       // (key build only)

    const int tid = threadIdx.x;
    const int lvl = blockIdx.x;
    const LevelDesc& L = P.lv[lvl];
    const int N = A.n_target[lvl];
    SelectedKp* out = sel_out + (size_t)lvl * A.sel_stride;
    const int W = L.w - 2 * kFastBorder, H = L.h - 2 * kFastBorder;

    // ---------------- keys in raster order off the keep bitmap ----------------
    // Word i of the level in raster order = pixel row i / ntx, tile column i % ntx.  The bitmap is tile-major in memory
    // ([tile][row]): it is read in memory order (coalesced, every load of a thread in flight at once) and laid down in
    // LDS in raster order; thread t then owns a contiguous run of raster words.
    QT_T(0);
    const int ntx = L.ntx;
    const int nwords = L.nty * kTile * ntx;
    const bool cached = nwords <= kQtWordCache;
    if (cached) {
        for (int i = tid; i < nwords; i += kQtThreads) {
            const int tile = i >> 5, r = i & 31;
            const int ty = tile / ntx, tx = tile - ty * ntx;
            s_words[(ty * kTile + r) * ntx + tx] = L.bitmap[i];
        }
        __syncthreads();
    }
    QT_T(1);
    const int wpt = ((nwords + kQtThreads - 1) / kQtThreads) | 1;  // odd: neighbouring threads start in different banks
    const int w0 = min(tid * wpt, nwords), w1 = min(w0 + wpt, nwords);
    const int row_first = ntx > 0 ? w0 / ntx : 0, tx_first = w0 - row_first * ntx;
    auto word_at = [&](int i, int row, int tx) -> uint32_t {
        return cached ? s_words[i] : L.bitmap[(size_t)((row >> 5) * ntx + tx) * kTile + (row & 31)];
    };
    int total = 0, base = 0;
    {
        int cnt = 0, row = row_first, tx = tx_first;
        for (int i = w0; i < w1; i++) {
            cnt += __popc(word_at(i, row, tx));
            if (++tx == ntx) { tx = 0; row++; }
        }
        base = qt_block_scan(cnt, s_wi, &total);
    }
    QT_T(2);
    const int n = min(total, kQtMaxKeys);  // the first 10000 in raster order (Fast.hpp:32)
    if (n <= 0) {
        if (tid == 0) count_out[lvl] = 0;
        return;
    }
    {
        int rank = base, row = row_first, tx = tx_first;
        for (int i = w0; i < w1 && rank < n; i++) {
            uint32_t word = word_at(i, row, tx);
            while (word && rank < n) {
                const int bit = __ffs(word) - 1;
                word &= word - 1;
                s_key[rank++] = (uint32_t)(3 + 32 * tx + bit) | ((uint32_t)(3 + row) << 16);
            }
            if (++tx == ntx) { tx = 0; row++; }
        }
    }
    __syncthreads();  // keys are down; the word cache is dead
    QT_T(3);

    const int chunk = (n + kQtThreads - 1) / kQtThreads;
    const int p0 = min(tid * chunk, n), p1 = min(p0 + chunk, n);
    int b = 0;  // current key -> node buffer

    // ---------------- roots (:468-511): a stable partition of the raster-ordered keys by root ----------------
    int n_ini = (int)roundf((float)W / (float)H);
    if (n_ini < 1) n_ini = 1;
    const float hX = (float)W / (float)n_ini;
    {
        uint32_t kreg[kQtChunk];
        u64 mine = 0;
#pragma unroll
        for (int j = 0; j < kQtChunk; j++) {
            kreg[j] = 0;
            if (p0 + j < p1) {
                kreg[j] = s_key[p0 + j];
                const int r = min((int)((float)(int)(kreg[j] & 0xFFFFu) / hX), n_ini - 1);
                mine += 1ull << (16 * r);
            }
        }
        const u64 excl = qt_block_scan64(mine, s_w64);  // (barriers: every thread holds its keys before any is moved)
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
#pragma unroll
        for (int j = 0; j < kQtChunk; j++)
            if (p0 + j < p1) {
                const int r = min((int)((float)(int)(kreg[j] & 0xFFFFu) / hX), n_ini - 1);
                const int pos = (r == 0 ? off_r[0] : r == 1 ? off_r[1] : r == 2 ? off_r[2] : off_r[3]) + qt_unpack(run, r);
                s_key[pos] = kreg[j];
                s_knode2[pos] = (uint16_t)(r == 0 ? slot_r[0] : r == 1 ? slot_r[1] : r == 2 ? slot_r[2] : slot_r[3]);
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
            nodes[slot_r[tid]] = nd;
        }
        if (tid == 0) s_misc[0] = m0;  // list size
        __syncthreads();
    }

    int m = s_misc[0];
    uint32_t seq_base = 4;
    bool careful = false;
    QT_T(4);
    // ---------------- split steps ----------------
    for (int guard = 0; guard < 64; guard++) {
        const uint16_t* knode = s_knode2 + b * kQtMaxKeys;
        uint16_t* knode_new = s_knode2 + (b ^ 1) * kQtMaxKeys;
        // (B) quadrant of every key that sits in a non-leaf node; packed counts; prefix at the slice borders.  The
        // thread's keys are read ONCE per step into registers (the partition below moves keys in place), the quadrant
        // is kept as a 4-bit code (0 = leaf node, q + 1 otherwise) in a 64-bit register.
        uint32_t kreg[kQtChunk];
        u64 mine = 0, qcode = 0;
#pragma unroll
        for (int j = 0; j < kQtChunk; j++) {
            const int p = p0 + j;
            kreg[j] = 0;
            if (p < p1) {
                kreg[j] = s_key[p];
                const QtNode nd = nodes[knode[p]];
                if (nd.n > 1) {
                    const int cx = (int)(kreg[j] & 0xFFFFu), cy = (int)(kreg[j] >> 16);
                    const int xm = nd.x0 + (int)ceilf((float)(nd.x1 - nd.x0) / 2), ym = nd.y0 + (int)ceilf((float)(nd.y1 - nd.y0) / 2);
                    const int q = (cx < xm ? 0 : 1) + (cy < ym ? 0 : 2);
                    mine += 1ull << (16 * q);
                    qcode |= (u64)(q + 1) << (4 * j);
                }
            }
        }
        const u64 excl = qt_block_scan64(mine, s_w64);
        {
            u64 run = excl;
#pragma unroll
            for (int j = 0; j < kQtChunk; j++) {
                const int p = p0 + j;
                const int qc = (int)((qcode >> (4 * j)) & 15ull);
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
        QT_T(8 + 4 * guard + 0);
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
        QT_T(8 + 4 * guard + 1);
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
                        nodes[pos] = nd;  // (every thread read its own node before the barriers above)
                        s_child[tid][q] = (uint16_t)pos;
                        r++;
                    }
                    o += cnt[q];
                }
            } else {
                const int pos = total_children + ur;
                nodes[pos] = me;
                s_newslot[tid] = (uint16_t)pos;
            }
        }
        __syncthreads();
        QT_T(8 + 4 * guard + 2);
        // (F) stable 4-way partition of the keys of the split nodes, in place (the step's keys are in registers)
        {
            u64 run = excl;
#pragma unroll
            for (int j = 0; j < kQtChunk; j++) {
                const int p = p0 + j;
                if (p < p1) {
                    const int ni = knode[p];
                    const int q = (int)((qcode >> (4 * j)) & 15ull) - 1;
                    if (q >= 0 && s_proc[ni] >= 0) {
                        const int slot = s_child[ni][q];
                        const int rank = qt_unpack(run, q) - qt_unpack(s_pre[ni], q);
                        const int pos = nodes[slot].off + rank;
                        s_key[pos] = kreg[j];
                        knode_new[pos] = (uint16_t)slot;
                    } else {
                        knode_new[p] = s_newslot[ni];  // the key stays where it is
                    }
                    if (q >= 0) run += 1ull << (16 * q);
                }
            }
        }
        __syncthreads();
        QT_T(8 + 4 * guard + 3);
        b ^= 1;
        seq_base += 4u * (uint32_t)nsplit;
        const int m_prev = m;
        m = m_new;
        // termination (:595-598, :650-664)
        if (m >= N || m == m_prev) break;
        if (!careful && (m + n_to_expand * 3 > N)) careful = true;
    }

    // ---------------- best response per node, list order (:667-686) ----------------
    // The reference keeps the FIRST maximum of `response` in the node's insertion order.  Every key fetches its own
    // score from the score map - a thread's loads all in flight together - and the node's winner is an LDS atomicMax
    // over (score << 16 | 0xFFFF - rank in the node): highest score, lowest rank on ties.
    QT_T(5);
    int* s_best = s_proc;  // (dead after the last split step)
    if (tid < kQtMaxNodes) s_best[tid] = -1;
    __syncthreads();
    {
        int sc[kQtChunk];
#pragma unroll
        for (int j = 0; j < kQtChunk; j++) {
            sc[j] = 0;
            if (p0 + j < p1) {
                const uint32_t k = s_key[p0 + j];
                const int px = (int)(k & 0xFFFFu) - 3, py = (int)(k >> 16) - 3;  // pixel of the FAST ROI
                sc[j] = L.score[(size_t)((py >> 5) * ntx + (px >> 5)) * kScoreBlock + kScoreRing + (py & 31) * kTile + (px & 31)];
            }
        }
#pragma unroll
        for (int j = 0; j < kQtChunk; j++)
            if (p0 + j < p1) {
                const int ni = s_knode2[b * kQtMaxKeys + p0 + j];
                atomicMax(&s_best[ni], (sc[j] << 16) | (0xFFFF - (p0 + j - (int)nodes[ni].off)));
            }
    }
    __syncthreads();
    if (tid < m) {
        const QtNode nd = nodes[tid];
        const int v = s_best[tid];
        const uint32_t best = s_key[nd.off + (0xFFFF - (v & 0xFFFF))];
        SelectedKp o;
        o.x = (int16_t)((int)(best & 0xFFFFu) + kFastBorder);  // addBorder_kernel, Fast_gpu.cu:461-470
        o.y = (int16_t)((int)(best >> 16) + kFastBorder);
        o.level = (uint16_t)lvl;
        o.score = (uint16_t)(v >> 16);
        if (tid < A.sel_stride) out[tid] = o;
    }
    if (tid == 0) count_out[lvl] = min(m, A.sel_stride);
    QT_T(6);
#ifdef QT_TIMING
    if (blockIdx.x == 0 && tid == 0) { qt_stamps[7] = m; qt_stamps[90] = n; qt_stamps[91] = careful; }
#endif
}

void launch_quadtree(const PyramidParams& p, const int* n_target, int sel_stride, SelectedKp* d_sel, int32_t* d_count,
                     hipStream_t s) {
    QtLevelArgs a;
    for (int l = 0; l < kMaxLevels; l++) a.n_target[l] = l < p.nlevels ? n_target[l] : 0;
    a.sel_stride = sel_stride;
    hipLaunchKernelGGL(quadtree_kernel, dim3(p.nlevels), dim3(kQtThreads), 0, s, p, a, d_sel, d_count);
}

}  // namespace so
