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
// the very end.  (Rounds 1-2 had a per-band emit kernel write 8-byte records that every split step then gathered from
// global memory through the key: 12-15 us for the launch, a memory round trip per step.)
//
// The level-0 workgroup is the critical path of the whole front end and it runs on ONE compute unit, four waves per
// SIMD: what it costs is instructions issued, not memory.  So: a thread holds the keys of its chunk of positions in
// registers across a step (keys move in place, one LDS read per key per step); the loops over a thread's keys are
// compiled for 1, 2, 4, 6 and 10 keys per thread and the workgroup picks the one its key count needs; scans run on
// DPP row shifts with one barrier each; what a key needs of its node is one 8-byte record.
#include "orb_device.h"

namespace so {

constexpr int kQtThreads = 1024;
constexpr int kQtWaves = kQtThreads / 64;
constexpr int kQtMaxKeys = kFastCap;  // 10000 candidates per level at most
constexpr int kQtMaxNodes = 1024;     // list never exceeds N + 3 (N <= 1020 on this path)
constexpr int kQtChunk = (kQtMaxKeys + kQtThreads - 1) / kQtThreads;  // positions per thread at most (10)
static_assert(kQtChunk <= 10, "a thread's quadrant codes (3 bits per key) must fit one 32-bit register");

typedef unsigned long long u64;

// tools/probe/qt_probe.hip builds this file with -DQT_TIMING: shader-clock stamps of the level-0 workgroup
#ifdef QT_TIMING
__device__ long long qt_stamps[96];
#define QT_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) qt_stamps[(i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define QT_T(i) do { } while (0)
#endif

// ---- scans ----------------------------------------------------------------------------------------------------
// inclusive scan over the 64 lanes of a wave: row shifts inside the rows of 16, then the row totals are broadcast
// to the rows behind them (the DPP sequence LLVM's atomic optimizer emits for gfx9)
__device__ __forceinline__ uint32_t qt_wave_incl(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return (uint32_t)x;
}
// inclusive scan inside every row of 16 lanes
__device__ __forceinline__ uint32_t qt_row_incl(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    return (uint32_t)x;
}

// The wave totals of consecutive scans go to alternating halves of s_w: a wave can only be two scans ahead of the
// slowest one after passing the barrier of the scan in between, so ONE barrier per scan is enough.
struct QtScan {
    u64 (*s_w)[kQtWaves];  // [2][16]
    int parity;
};

// block-wide exclusive scan of two independent 32-bit lanes (the halves of a u64; four 16-bit fields when the sums
// stay below 65536); *total = sums over the block
__device__ __forceinline__ u64 qt_block_scan64(u64 v, QtScan& S, u64* total) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint32_t ilo = qt_wave_incl(lo), ihi = qt_wave_incl(hi);
    u64* buf = S.s_w[S.parity];
    S.parity ^= 1;
    if (lane == 63) buf[w] = (u64)ilo | ((u64)ihi << 32);
    __syncthreads();
    const u64 t = buf[lane & 15];  // every row of 16 lanes holds the 16 wave totals
    const uint32_t tlo = qt_row_incl((uint32_t)t), thi = qt_row_incl((uint32_t)(t >> 32));
    const int prev = (w + 15) & 15;
    uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)tlo, prev), bhi = (uint32_t)__builtin_amdgcn_readlane((int)thi, prev);
    if (w == 0) blo = bhi = 0;
    *total = (u64)(uint32_t)__builtin_amdgcn_readlane((int)tlo, 15) | ((u64)(uint32_t)__builtin_amdgcn_readlane((int)thi, 15) << 32);
    return (u64)(blo + ilo - lo) | ((u64)(bhi + ihi - hi) << 32);
}

__device__ __forceinline__ uint32_t qt_block_scan32(uint32_t v, QtScan& S, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t incl = qt_wave_incl(v);
    u64* buf = S.s_w[S.parity];
    S.parity ^= 1;
    if (lane == 63) buf[w] = incl;
    __syncthreads();
    const uint32_t t = qt_row_incl((uint32_t)buf[lane & 15]);
    uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)t, (w + 15) & 15);
    if (w == 0) base = 0;
    *total = (uint32_t)__builtin_amdgcn_readlane((int)t, 15);
    return base + incl - v;
}

__device__ __forceinline__ int qt_unpack(u64 v, int q) { return (int)((v >> (16 * q)) & 0xFFFFull); }
__device__ __forceinline__ u64 qt_pack4(int a, int b, int c, int d) {
    return (u64)(uint32_t)(a & 0xFFFF) | ((u64)(uint32_t)(b & 0xFFFF) << 16) | ((u64)(uint32_t)(c & 0xFFFF) << 32) |
           ((u64)(uint32_t)(d & 0xFFFF) << 48);
}

// ---- LDS map (bytes) --------------------------------------------------------------------------------------------
// keys 40000 | key -> node 20000 | per node: box 8 B, info 8 B, seq 4 B, pre / post 8 B each, child slots 8 B, child
// bases 8 B, two 4-byte work words | per thread: exclusive prefix 8 B.  While the key array is built, the level's keep
// words sit (in raster order) in the space behind the keys.
constexpr int kQtOffKnode = kQtMaxKeys * 4;
constexpr int kQtOffBox = (kQtOffKnode + kQtMaxKeys * 2 + 15) & ~15;
constexpr int kQtOffInfo = kQtOffBox + kQtMaxNodes * 8;
constexpr int kQtOffSeq = kQtOffInfo + kQtMaxNodes * 8;
constexpr int kQtOffPre = kQtOffSeq + kQtMaxNodes * 4;
constexpr int kQtOffPost = kQtOffPre + kQtMaxNodes * 8;
constexpr int kQtOffChild = kQtOffPost + kQtMaxNodes * 8;
constexpr int kQtOffCbase = kQtOffChild + kQtMaxNodes * 8;
constexpr int kQtOffKv = kQtOffCbase + kQtMaxNodes * 8;
constexpr int kQtOffG = kQtOffKv + kQtMaxNodes * 4;
constexpr int kQtOffTexcl = kQtOffG + kQtMaxNodes * 4;
constexpr int kQtSmemBytes = kQtOffTexcl + kQtThreads * 8;
constexpr int kQtWordCache = (kQtSmemBytes - kQtOffKnode) / 4;  // keep words the LDS can hold next to the keys
static_assert(kQtOffKnode % 16 == 0, "LDS map");

struct QtCtx {
    uint32_t* key;     // x | y << 16 (ROI coordinates, as GpuFast's kpLoc)
    uint16_t* knode;   // key position -> node slot
    uint2* box;        // x0 | y0 << 16, x1 | y1 << 16
    uint2* info;       // off | n << 16 (slice of the key array), xm | ym << 16 (where the node splits)
    uint32_t* seq;     // creation sequence (tie-break of the careful phase)
    u64* pre;          // per node: quadrant counts (4 x 16 bit) in front of its slice / behind it, relative to the
    u64* post;         //           thread that owns that key position
    u64* child;        // per node: slots of its four children, or (new slot | 0x8000) x 4 when it is not split
    u64* cbase;        // per node and quadrant: child's key offset - quadrant count in front of the slice (mod 2^16)
    uint32_t* kv;      // work words per node
    uint32_t* g;
    u64* texcl;        // per thread: exclusive prefix of its quadrant counts
    int* misc;
    int n, N, lvl, W, H, ntx, sel_stride;
    const uint8_t* score;
    SelectedKp* out;
    int32_t* count_out;
};

__device__ __forceinline__ int qt_mid(int a, int b) { return a + (int)ceilf((float)(b - a) / 2); }  // :418-419

// Everything behind the raster-ordered key array, for at most CH keys per thread.
template <int CH>
__device__ __forceinline__ void qt_run(const QtCtx& c, QtScan& S) {
    const int tid = threadIdx.x;
    const int n = c.n, N = c.N;
    const int chunk = (n + kQtThreads - 1) / kQtThreads;  // <= CH
    const float inv_chunk = 1.0f / (float)chunk;
    const int p0 = min(tid * chunk, n), p1 = min(p0 + chunk, n);

    // ---------------- roots (:468-511): a stable partition of the raster-ordered keys by root ----------------
    int n_ini = (int)roundf((float)c.W / (float)c.H);
    if (n_ini < 1) n_ini = 1;
    const float hX = (float)c.W / (float)n_ini;
    int m;
    {
        uint32_t kreg[CH], rcode = 0;
        u64 mine = 0;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            kreg[j] = 0;
            if (p0 + j < p1) {
                kreg[j] = c.key[p0 + j];
                const int r = min((int)((float)(int)(kreg[j] & 0xFFFFu) / hX), n_ini - 1);
                mine += 1ull << (16 * r);
                rcode |= (uint32_t)r << (2 * j);
            }
        }
        u64 tot = 0;
        const u64 excl = qt_block_scan64(mine, S, &tot);  // (barrier: every thread holds its keys before any is moved)
        int off_r[4], cnt_r[4], slot_r[4], m0 = 0, o = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            cnt_r[r] = r < n_ini ? qt_unpack(tot, r) : 0;
            off_r[r] = o;
            o += cnt_r[r];
            slot_r[r] = m0;
            if (cnt_r[r] > 0) m0++;
        }
        const u64 offs = qt_pack4(off_r[0], off_r[1], off_r[2], off_r[3]);
        const u64 slots = qt_pack4(slot_r[0], slot_r[1], slot_r[2], slot_r[3]);
        u64 run = excl + offs;
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (p0 + j < p1) {
                const int r = (int)((rcode >> (2 * j)) & 3u);
                const int pos = qt_unpack(run, r);
                c.key[pos] = kreg[j];
                c.knode[pos] = (uint16_t)qt_unpack(slots, r);
                run += 1ull << (16 * r);
            }
        if (tid < 4 && tid < n_ini && cnt_r[tid] > 0) {
            const int x0 = (int)(hX * (float)tid), x1 = (int)(hX * (float)(tid + 1));
            const int s = slot_r[tid];
            c.box[s] = make_uint2((uint32_t)x0, (uint32_t)x1 | ((uint32_t)c.H << 16));
            c.info[s] = make_uint2((uint32_t)off_r[tid] | ((uint32_t)cnt_r[tid] << 16),
                                   (uint32_t)qt_mid(x0, x1) | ((uint32_t)qt_mid(0, c.H) << 16));
            c.seq[s] = (uint32_t)tid;
        }
        m = m0;
        __syncthreads();
    }
    QT_T(4);

    uint32_t seq_base = 4, seq_prev = 0;
    bool careful = false;
    // ---------------- split steps ----------------
    for (int guard = 0; guard < 64; guard++) {
        // (B) quadrant of every key that sits in a non-leaf node, kept as a 3-bit code (0 = leaf node, q + 1 otherwise);
        // packed counts; the counts in front of / behind a node's slice are noted relative to this thread
        uint32_t kreg[CH], nireg[CH], qcode = 0;
        u64 mine = 0;
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int p = p0 + j;
            kreg[j] = 0;
            nireg[j] = 0;
            if (p < p1) {
                kreg[j] = c.key[p];
                nireg[j] = c.knode[p];
                const uint2 f = c.info[nireg[j]];
                const int off = (int)(f.x & 0xFFFFu), nn = (int)(f.x >> 16);
                if (nn > 1) {
                    const int q = ((kreg[j] & 0xFFFFu) < (f.y & 0xFFFFu) ? 0 : 1) + ((kreg[j] >> 16) < (f.y >> 16) ? 0 : 2);
                    if (p == off) c.pre[nireg[j]] = mine;
                    mine += 1ull << (16 * q);
                    if (p == off + nn - 1) c.post[nireg[j]] = mine;
                    qcode |= (uint32_t)(q + 1) << (3 * j);
                }
            }
        }
        u64 tot_unused = 0;
        const u64 excl = qt_block_scan64(mine, S, &tot_unused);
        c.texcl[tid] = excl;
        __syncthreads();
        QT_T(8 + 4 * guard + 0);
        // per node: children populations
        int cnt[4] = {0, 0, 0, 0}, k_children = 0;
        bool nonleaf = false;
        uint2 my_box = make_uint2(0, 0), my_info = make_uint2(0, 0);
        uint32_t my_seq = 0;
        u64 pre_g = 0;
        if (tid < m) {
            my_box = c.box[tid];
            my_info = c.info[tid];
            my_seq = c.seq[tid];
            const int off = (int)(my_info.x & 0xFFFFu), nn = (int)(my_info.x >> 16);
            nonleaf = nn > 1;
            if (nonleaf) {
                // the thread that owns key position p is p / chunk (exact in float: p < 10000, chunk <= 10)
                const int o0 = (int)(((float)off + 0.5f) * inv_chunk), o1 = (int)(((float)(off + nn - 1) + 0.5f) * inv_chunk);
                pre_g = c.pre[tid] + c.texcl[o0];
                const u64 d = c.post[tid] + c.texcl[o1] - pre_g;
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
        int nsplit = 0, cp = 0, ur = 0, total_children = 0, n_untouched = 0, n_to_expand = 0, proc = -1;
        bool split = false;
        if (!careful) {
            // sweep: every non-leaf node is split, in list order - processing rank, children prefix, untouched rank and the
            // count of splittable children are four 16-bit fields of ONE block scan
            const u64 v = tid < m ? ((u64)(nonleaf ? 1 : 0) | ((u64)(nonleaf ? k_children : 0) << 16) | ((u64)(nonleaf ? 0 : 1) << 32) |
                                     ((u64)ne << 48))
                                  : 0ull;
            u64 tot = 0;
            const u64 ex = qt_block_scan64(v, S, &tot);
            nsplit = qt_unpack(tot, 0);
            total_children = qt_unpack(tot, 1);
            n_untouched = qt_unpack(tot, 2);
            n_to_expand = qt_unpack(tot, 3);
            cp = qt_unpack(ex, 1);
            ur = qt_unpack(ex, 2);
            split = tid < m && nonleaf;
            proc = split ? qt_unpack(ex, 0) : -1;
        } else {
            // rank by (population, seq) descending among the non-leaf nodes.  All of them were created by the previous
            // step, so seq - (that step's first seq) < 4096 and (population << 16 | that) orders them in one compare.
            const uint32_t rk = nonleaf ? (((my_info.x >> 16) << 16) | (my_seq - seq_prev)) : 0u;
            c.kv[tid] = rk;  // (zero for leaves and for tid >= m)
            c.g[tid] = 0;
            if (tid == 0) c.misc[2] = 0x7FFFFFFF;
            __syncthreads();
            int rank = 0;
            if (nonleaf) {
                const uint4* kv4 = reinterpret_cast<const uint4*>(c.kv);
                const int m4 = (m + 3) >> 2;
#pragma unroll 4
                for (int j = 0; j < m4; j++) {
                    const uint4 o = kv4[j];
                    rank += (o.x > rk) + (o.y > rk) + (o.z > rk) + (o.w > rk);
                }
                c.g[rank] = (uint32_t)(k_children - 1) | ((uint32_t)k_children << 16);  // growth of the list | children
            }
            __syncthreads();
            // in processing order: growth prefix (the reference stops right after the first split that makes size >= N,
            // :655-661), children prefix, number of ranked nodes
            const uint32_t gv = c.g[tid];
            u64 tot = 0;
            const u64 ex = qt_block_scan64((u64)gv | ((u64)(gv != 0u ? 1u : 0u) << 32), S, &tot);
            const int n_nonleaf = (int)(tot >> 32);
            const int gpre = (int)(ex & 0xFFFFull), growth = (int)(gv & 0xFFFFu);
            if (gv != 0u && m + gpre + growth >= N) atomicMin(&c.misc[2], tid + 1);
            c.kv[tid] = (uint32_t)((ex >> 16) & 0xFFFFull) + (gv >> 16);  // inclusive children prefix by processing rank
            __syncthreads();
            nsplit = min(c.misc[2], n_nonleaf);
            total_children = nsplit > 0 ? (int)c.kv[nsplit - 1] : 0;
            split = nonleaf && rank < nsplit;
            if (split) {
                proc = rank;
                cp = (int)c.kv[rank] - k_children;
            }
            uint32_t tot2 = 0;
            const uint32_t ex2 = qt_block_scan32(((tid < m && !split) ? 1u : 0u) | ((uint32_t)(split ? ne : 0) << 16), S, &tot2);
            ur = (int)(ex2 & 0xFFFFu);
            n_untouched = (int)(tot2 & 0xFFFFu);
            n_to_expand = (int)(tot2 >> 16);
        }
        QT_T(8 + 4 * guard + 1);
        const int m_new = total_children + n_untouched;
        // (E) new nodes; slot == position in the new list.  (Every thread read its own node before the barriers above.)
        if (tid < m) {
            if (split) {
                const int x0 = (int)(my_box.x & 0xFFFFu), y0 = (int)(my_box.x >> 16), x1 = (int)(my_box.y & 0xFFFFu), y1 = (int)(my_box.y >> 16);
                const int xm = (int)(my_info.y & 0xFFFFu), ym = (int)(my_info.y >> 16);
                int r = 0, o = (int)(my_info.x & 0xFFFFu);
                int slot[4] = {0, 0, 0, 0}, cb[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (cnt[q] > 0) {
                        const int pos = total_children - 1 - (cp + r);
                        const int cx0 = (q & 1) ? xm : x0, cx1 = (q & 1) ? x1 : xm, cy0 = (q & 2) ? ym : y0, cy1 = (q & 2) ? y1 : ym;
                        c.box[pos] = make_uint2((uint32_t)cx0 | ((uint32_t)cy0 << 16), (uint32_t)cx1 | ((uint32_t)cy1 << 16));
                        c.info[pos] = make_uint2((uint32_t)o | ((uint32_t)cnt[q] << 16),
                                                 (uint32_t)qt_mid(cx0, cx1) | ((uint32_t)qt_mid(cy0, cy1) << 16));
                        c.seq[pos] = seq_base + 4u * (uint32_t)proc + (uint32_t)q;
                        slot[q] = pos;
                        cb[q] = o - qt_unpack(pre_g, q);
                        r++;
                    }
                    o += cnt[q];
                }
                c.child[tid] = qt_pack4(slot[0], slot[1], slot[2], slot[3]);
                c.cbase[tid] = qt_pack4(cb[0], cb[1], cb[2], cb[3]);
            } else {
                const int pos = total_children + ur;
                c.box[pos] = my_box;
                c.info[pos] = my_info;
                c.seq[pos] = my_seq;
                const int stay = pos | 0x8000;
                c.child[tid] = qt_pack4(stay, stay, stay, stay);
            }
        }
        __syncthreads();
        QT_T(8 + 4 * guard + 2);
        // (F) stable 4-way partition of the keys of the split nodes, in place (the step's keys are in registers)
        {
            u64 run = excl;
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const int p = p0 + j;
                if (p < p1) {
                    const int ni = (int)nireg[j];
                    const int qc = (int)((qcode >> (3 * j)) & 7u);
                    const int q = qc ? qc - 1 : 0;
                    const int ch = qt_unpack(c.child[ni], q);
                    if (ch & 0x8000) {
                        c.knode[p] = (uint16_t)(ch & 0x7FFF);  // the key stays where it is
                    } else {
                        const int pos = (qt_unpack(c.cbase[ni], q) + qt_unpack(run, q)) & 0xFFFF;
                        c.key[pos] = kreg[j];
                        c.knode[pos] = (uint16_t)ch;
                    }
                    if (qc) run += 1ull << (16 * q);
                }
            }
        }
        __syncthreads();
        QT_T(8 + 4 * guard + 3);
        seq_prev = seq_base;
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
    int* s_best = reinterpret_cast<int*>(c.kv);
    s_best[tid] = -1;
    __syncthreads();
    {
        int sc[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            sc[j] = 0;
            if (p0 + j < p1) {
                const uint32_t k = c.key[p0 + j];
                const int px = (int)(k & 0xFFFFu) - 3, py = (int)(k >> 16) - 3;  // pixel of the FAST ROI
                sc[j] = c.score[(size_t)((py >> 5) * c.ntx + (px >> 5)) * kScoreBlock + kScoreRing + (py & 31) * kTile + (px & 31)];
            }
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
            if (p0 + j < p1) {
                const int ni = c.knode[p0 + j];
                atomicMax(&s_best[ni], (sc[j] << 16) | (0xFFFF - (p0 + j - (int)(c.info[ni].x & 0xFFFFu))));
            }
    }
    __syncthreads();
    if (tid < m) {
        const int off = (int)(c.info[tid].x & 0xFFFFu);
        const int v = s_best[tid];
        const uint32_t best = c.key[off + (0xFFFF - (v & 0xFFFF))];
        SelectedKp o;
        o.x = (int16_t)((int)(best & 0xFFFFu) + kFastBorder);  // addBorder_kernel, Fast_gpu.cu:461-470
        o.y = (int16_t)((int)(best >> 16) + kFastBorder);
        o.level = (uint16_t)c.lvl;
        o.score = (uint16_t)(v >> 16);
        if (tid < c.sel_stride) c.out[tid] = o;
    }
    if (tid == 0) c.count_out[c.lvl] = min(m, c.sel_stride);
    QT_T(6);
#ifdef QT_TIMING
    if (blockIdx.x == 0 && tid == 0) { qt_stamps[7] = m; qt_stamps[90] = n; qt_stamps[91] = careful; }
#endif
}

__device__ __forceinline__ void quadtree_body(const PyramidParams& P, const QtLevelArgs& A, SelectedKp* __restrict__ sel_out,
                                              int32_t* __restrict__ count_out) {
    __shared__ __align__(16) unsigned char smem[kQtSmemBytes];
    __shared__ u64 s_w[2][kQtWaves];
    __shared__ int s_misc[8];
    QtCtx c;
    c.key = reinterpret_cast<uint32_t*>(smem);
    c.knode = reinterpret_cast<uint16_t*>(smem + kQtOffKnode);
    c.box = reinterpret_cast<uint2*>(smem + kQtOffBox);
    c.info = reinterpret_cast<uint2*>(smem + kQtOffInfo);
    c.seq = reinterpret_cast<uint32_t*>(smem + kQtOffSeq);
    c.pre = reinterpret_cast<u64*>(smem + kQtOffPre);
    c.post = reinterpret_cast<u64*>(smem + kQtOffPost);
    c.child = reinterpret_cast<u64*>(smem + kQtOffChild);
    c.cbase = reinterpret_cast<u64*>(smem + kQtOffCbase);
    c.kv = reinterpret_cast<uint32_t*>(smem + kQtOffKv);
    c.g = reinterpret_cast<uint32_t*>(smem + kQtOffG);
    c.texcl = reinterpret_cast<u64*>(smem + kQtOffTexcl);
    c.misc = s_misc;
    uint32_t* s_words = reinterpret_cast<uint32_t*>(smem + kQtOffKnode);  // (key build only)
    QtScan S{s_w, 0};

    const int tid = threadIdx.x;
    const int lvl = blockIdx.x;
    const LevelDesc& L = P.lv[lvl];
    c.N = A.n_target[lvl];
    c.lvl = lvl;
    c.W = L.w - 2 * kFastBorder;
    c.H = L.h - 2 * kFastBorder;
    c.ntx = L.ntx;
    c.sel_stride = A.sel_stride;
    c.score = L.score;
    c.out = sel_out + (size_t)lvl * A.sel_stride;
    c.count_out = count_out;

    // ---------------- keys in raster order off the keep bitmap ----------------
    // Word i of the level in raster order = pixel row i / ntx, tile column i % ntx.  The bitmap is tile-major in memory
    // ([tile][row]): it is read in memory order (coalesced, every load of a thread in flight at once) and laid down in
    // LDS in raster order; thread t then owns a contiguous run of raster words.
    QT_T(0);
    const int ntx = L.ntx;
    const int nwords = L.nty * kTile * ntx;
    const float inv_ntx = 1.0f / (float)max(ntx, 1);
    const bool cached = nwords <= kQtWordCache;
    if (cached) {
        for (int i = tid; i < nwords; i += kQtThreads) {
            const int tile = i >> 5, r = i & 31;
            const int ty = (int)(((float)tile + 0.5f) * inv_ntx), tx = tile - ty * ntx;  // exact: tile < 2^16, ntx <= 128
            s_words[(ty * kTile + r) * ntx + tx] = L.bitmap[i];
        }
        __syncthreads();
    }
    QT_T(1);
    const int wpt = ((nwords + kQtThreads - 1) / kQtThreads) | 1;  // odd: neighbouring threads start in different banks
    const int w0 = min(tid * wpt, nwords), w1 = min(w0 + wpt, nwords);
    const int row_first = (int)(((float)w0 + 0.5f) * inv_ntx), tx_first = w0 - row_first * ntx;  // (w0 < 2^18: exact)
    auto word_at = [&](int i, int row, int tx) -> uint32_t {
        return cached ? s_words[i] : L.bitmap[(size_t)((row >> 5) * ntx + tx) * kTile + (row & 31)];
    };
    uint32_t total = 0, base = 0;
    {
        uint32_t cnt = 0;
        if (cached) {
            for (int i = w0; i < w1; i++) cnt += (uint32_t)__popc(s_words[i]);
        } else {
            int row = row_first, tx = tx_first;
            for (int i = w0; i < w1; i++) {
                cnt += (uint32_t)__popc(word_at(i, row, tx));
                if (++tx == ntx) { tx = 0; row++; }
            }
        }
        base = qt_block_scan32(cnt, S, &total);
    }
    QT_T(2);
    const int n = (int)min(total, (uint32_t)kQtMaxKeys);  // the first 10000 in raster order (Fast.hpp:32)
    c.n = n;
    if (n <= 0) {
        if (tid == 0) count_out[lvl] = 0;
        return;
    }
    {
        int rank = (int)min(base, (uint32_t)n), row = row_first, tx = tx_first;
        for (int i = w0; i < w1 && rank < n; i++) {
            uint32_t word = word_at(i, row, tx);
            while (word && rank < n) {
                const int bit = __ffs(word) - 1;
                word &= word - 1;
                c.key[rank++] = (uint32_t)(3 + 32 * tx + bit) | ((uint32_t)(3 + row) << 16);
            }
            if (++tx == ntx) { tx = 0; row++; }
        }
    }
    __syncthreads();  // keys are down; the word cache is dead
    QT_T(3);

    const int chunk = (n + kQtThreads - 1) / kQtThreads;
    if (chunk <= 1) qt_run<1>(c, S);
    else if (chunk <= 2) qt_run<2>(c, S);
    else if (chunk <= 4) qt_run<4>(c, S);
    else if (chunk <= 6) qt_run<6>(c, S);
    else qt_run<kQtChunk>(c, S);
}

__global__ __launch_bounds__(kQtThreads) void quadtree_kernel(PyramidParams P, QtLevelArgs A,
                                                               SelectedKp* __restrict__ sel_out,
                                                               int32_t* __restrict__ count_out) {
    quadtree_body(P, A, sel_out, count_out);
}
// several members' pyramids in one launch (so_extractor_group): blockIdx.y = member
__global__ __launch_bounds__(kQtThreads) void quadtree_batch_kernel(const ExtractBatchMember* __restrict__ M) {
    const ExtractBatchMember& m = M[blockIdx.y];
    if (m.skip) return;
    quadtree_body(m.P, m.qt, m.qt_sel, m.qt_count);
}

void launch_quadtree_batch(const ExtractBatchMember* d_members, int n_members, int nlevels, hipStream_t s) {
    hipLaunchKernelGGL(quadtree_batch_kernel, dim3(nlevels, n_members), dim3(kQtThreads), 0, s, d_members);
}

void launch_quadtree(const PyramidParams& p, const int* n_target, int sel_stride, SelectedKp* d_sel, int32_t* d_count,
                     hipStream_t s) {
    QtLevelArgs a;
    for (int l = 0; l < kMaxLevels; l++) a.n_target[l] = l < p.nlevels ? n_target[l] : 0;
    a.sel_stride = sel_stride;
    hipLaunchKernelGGL(quadtree_kernel, dim3(p.nlevels), dim3(kQtThreads), 0, s, p, a, d_sel, d_count);
}

}  // namespace so
