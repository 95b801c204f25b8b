/*
 * orb_oracle.c — CPU restatement of SwarmMap's ORB front-end (see orb_oracle.h for status).
 * TEST INFRASTRUCTURE ONLY; never linked into the product.
 *
 * Every function cites the reference file:line it follows.  Compile with
 *   gcc -O2 -march=x86-64-v3 -ffp-contract=off -fPIC -shared
 * (-ffp-contract=off: float stages are defined WITHOUT implicit FMA contraction so that the HIP
 * kernels, built with the same flag, reproduce them bit for bit; explicit fmaf() is used where a
 * fused operation is part of the definition).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31      /* code/src/ORBextractor.cc:76 */
#define HALF_PATCH_SIZE 15 /* :77 */
#define EDGE_THRESHOLD 19  /* :78 */
#define FAST_BORDER 16     /* minBorderX = EDGE_THRESHOLD-3, :695 */

static const int8_t k_pattern[1024] = {
#include "brief_pattern.inc"
};

/* cvRound(float/double): round-half-to-even in the default rounding mode. */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------------------------------------
 * E0  ORBextractor::ORBextractor  (code/src/ORBextractor.cc:340-405)
 * ---------------------------------------------------------------------------------------------- */
void orc_make_tables(const orc_config* cfg, orc_tables* t) {
    memset(t, 0, sizeof(*t));
    const int nl = cfg->nlevels;
    const double sf = (double)cfg->scale_factor; /* member is double, include/ORBextractor.h:108 */
    t->scale[0] = 1.0f;
    t->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        t->scale[i] = (float)((double)t->scale[i - 1] * sf); /* :349 */
        t->sigma2[i] = t->scale[i] * t->scale[i];             /* :350 */
    }
    for (int i = 0; i < nl; i++) {
        t->inv_scale[i] = 1.0f / t->scale[i];
        t->inv_sigma2[i] = 1.0f / t->sigma2[i];
    }
    /* :365-377 geometric split of nfeatures over the levels */
    float factor = (float)(1.0 / sf);
    float desired = (float)((double)((float)cfg->nfeatures * (1.0f - factor)) /
                            (1.0 - pow((double)factor, (double)nl)));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        t->features_per_level[l] = cv_round_f(desired);
        sum += t->features_per_level[l];
        desired *= factor;
    }
    t->features_per_level[nl - 1] = cfg->nfeatures - sum > 0 ? cfg->nfeatures - sum : 0;

    /* :385-402 end of row of the circular patch */
    int v, v0;
    int vmax = (int)floorf(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceilf(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) t->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (t->umax[v0] == t->umax[v0 + 1]) ++v0;
        t->umax[v] = v0;
        ++v0;
    }
}

/* ------------------------------------------------------------------------------------------------
 * CONVENTION SWITCH (test infrastructure of test infrastructure): the arithmetic this oracle DEFINES where the reference's
 * lives in un-vendored OpenCV-CUDA / Eigen / nvcc fast-math (parity unpinned, header of this file) can be swapped, one piece at
 * a time, for its plausible alternative - tools/convention_sensitivity.py runs the whole closed loop under each swap and
 * records how far the trajectory moves.  0 = the conventions every test and the product are held to.
 * ---------------------------------------------------------------------------------------------- */
static int g_convention = 0;
void orc_set_convention(int flags) { g_convention = flags; }
int orc_get_convention(void) { return g_convention; }

void orc_level_size(int w, int h, float inv_scale, int* lw, int* lh) {
    *lw = cv_round_f((float)w * inv_scale); /* :825 */
    *lh = cv_round_f((float)h * inv_scale);
}

/* ------------------------------------------------------------------------------------------------
 * E1  cv::cuda::resize(..., INTER_LINEAR)  (call site code/src/ORBextractor.cc:845)
 * The arithmetic is un-vendored OpenCV-CUDA; CONVENTION DEFINED HERE (parity unpinned): the
 * OpenCV-CUDA "resize_linear" form  src = dst * (1/f), f = dsize/ssize, floor, 4 float-weighted
 * taps accumulated in the order (y1,x1),(y1,x2),(y2,x1),(y2,x2), round-half-even, saturate.
 * ---------------------------------------------------------------------------------------------- */
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh,
                       int dstride) {
    const float fx = (float)(1.0 / ((double)dw / (double)sw));
    const float fy = (float)(1.0 / ((double)dh / (double)sh));
    if (g_convention & ORC_CONV_RESIZE_HALF_PIXEL) {
        /* alternative: pixel CENTRES map onto each other, src = (dst + 0.5) / f - 0.5, clamped at the border (cv::resize on
         * the CPU; the CUDA kernel restated below samples at dst / f) */
        for (int y = 0; y < dh; y++) {
            float src_y = ((float)y + 0.5f) * fy - 0.5f;
            if (src_y < 0.f) src_y = 0.f;
            int y1 = (int)floorf(src_y);
            if (y1 > sh - 1) y1 = sh - 1;
            const int y2r = y1 + 1 < sh ? y1 + 1 : sh - 1;
            const float wy1 = src_y - (float)y1, wy2 = 1.0f - wy1;
            for (int x = 0; x < dw; x++) {
                float src_x = ((float)x + 0.5f) * fx - 0.5f;
                if (src_x < 0.f) src_x = 0.f;
                int x1 = (int)floorf(src_x);
                if (x1 > sw - 1) x1 = sw - 1;
                const int x2r = x1 + 1 < sw ? x1 + 1 : sw - 1;
                const float wx1 = src_x - (float)x1, wx2 = 1.0f - wx1;
                float out = (float)src[y1 * sstride + x1] * (wx2 * wy2);
                out = out + (float)src[y1 * sstride + x2r] * (wx1 * wy2);
                out = out + (float)src[y2r * sstride + x1] * (wx2 * wy1);
                out = out + (float)src[y2r * sstride + x2r] * (wx1 * wy1);
                int v = (int)lrintf(out);
                dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
        return;
    }
    for (int y = 0; y < dh; y++) {
        const float src_y = (float)y * fy;
        const int y1 = (int)floorf(src_y);
        const int y2 = y1 + 1;
        const int y2r = y2 < sh - 1 ? y2 : sh - 1;
        const float wy2 = (float)y2 - src_y; /* weight of row y1 */
        const float wy1 = src_y - (float)y1; /* weight of row y2 */
        for (int x = 0; x < dw; x++) {
            const float src_x = (float)x * fx;
            const int x1 = (int)floorf(src_x);
            const int x2 = x1 + 1;
            const int x2r = x2 < sw - 1 ? x2 : sw - 1;
            const float wx2 = (float)x2 - src_x;
            const float wx1 = src_x - (float)x1;
            float out = (float)src[y1 * sstride + x1] * (wx2 * wy2);
            out = out + (float)src[y1 * sstride + x2r] * (wx1 * wy2);
            out = out + (float)src[y2r * sstride + x1] * (wx2 * wy1);
            out = out + (float)src[y2r * sstride + x2r] * (wx1 * wy1);
            int v = (int)lrintf(out);
            dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* cv::cuda::copyMakeBorder(..., BORDER_REFLECT_101)  (code/src/ORBextractor.cc:846-851) */
void orc_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int border,
                           int dstride) {
    for (int y = -border; y < h + border; y++) {
        const uint8_t* srow = src + reflect101(y, h) * sstride;
        uint8_t* drow = dst + (y + border) * dstride;
        for (int x = -border; x < w + border; x++) drow[x + border] = srow[reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------------------------------------
 * E6  cv::cuda::createGaussianFilter(CV_8UC1, CV_8UC1, Size(7,7), 2, 2, BORDER_REFLECT_101)
 *     (code/src/ORBextractor.cc:835; applied in place at :719,:742).  Un-vendored; CONVENTION DEFINED
 * HERE (parity unpinned): separable float kernel k[i] = float(exp(-(i-3)^2/8)/sum), row pass into a
 * float buffer accumulating taps 0..6 in order (mul then add, no FMA), column pass likewise, then
 * round-half-even + saturate.  The ROI is treated as isolated (reflect-101 of the interior), which
 * equals reading the 19-px border the reference materialises.
 * ---------------------------------------------------------------------------------------------- */
static const float k_gauss7[7] = {0x1.1f5f62p-4f, 0x1.0c70fcp-3f, 0x1.869472p-3f, 0x1.ba95c0p-3f,
                                  0x1.869472p-3f, 0x1.0c70fcp-3f, 0x1.1f5f62p-4f};

void orc_gaussian7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    if (g_convention & ORC_CONV_GAUSS_FIXED8) {
        /* alternative: 8-bit fixed-point weights (sum 256), integer accumulation, one rounding at the end - the form OpenCV's
         * 8-bit filters take on the CPU */
        static const int kq[7] = {18, 34, 49, 54, 49, 34, 18};
        int32_t* ib = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
        for (int y = 0; y < h; y++) {
            const uint8_t* row = src + y * sstride;
            for (int x = 0; x < w; x++) {
                int32_t sum = 0;
                for (int k = 0; k < 7; k++) sum += (int32_t)row[reflect101(x + k - 3, w)] * kq[k];
                ib[y * w + x] = sum;
            }
        }
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int32_t sum = 0;
                for (int k = 0; k < 7; k++) sum += ib[reflect101(y + k - 3, h) * w + x] * kq[k];
                const int v = (sum + 32768) >> 16;
                dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        free(ib);
        return;
    }
    float* buf = (float*)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* row = src + y * sstride;
        for (int x = 0; x < w; x++) {
            float sum = 0.f;
            for (int k = 0; k < 7; k++) sum = sum + (float)row[reflect101(x + k - 3, w)] * k_gauss7[k];
            buf[y * w + x] = sum;
        }
    }
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float sum = 0.f;
            for (int k = 0; k < 7; k++) sum = sum + buf[reflect101(y + k - 3, h) * w + x] * k_gauss7[k];
            int v = (int)lrintf(sum);
            dst[y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(buf);
}

/* ------------------------------------------------------------------------------------------------
 * E2  FAST-9/16  (code/src/cuda/Fast_gpu.cu:63-282)
 * ---------------------------------------------------------------------------------------------- */
/* ring bit k -> (dy,dx), SURVEY.md A.2 (from isKeyPoint2 :224-252 + calcMask :78-183) */
static const int8_t k_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
static const int8_t k_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};

static int has_run9(int m) {
    /* circular run of >= 9 set bits in a 16-bit mask */
    unsigned x = (unsigned)m & 0xffffu;
    x |= x << 16;
    unsigned r = x & (x >> 1);
    r &= r >> 2;
    r &= r >> 4;      /* runs of 8 */
    r &= x >> 8;      /* runs of 9 */
    return (r & 0xffffu) != 0;
}

int orc_fast_is_corner_masks(int mask_dark, int mask_bright) {
    /* isKeyPoint, :189-193, with the table replaced by the predicate it encodes (KAT-pinned) */
    return has_run9(mask_dark) || has_run9(mask_bright);
}

int orc_fast_table_lookup(const uint8_t* table, int mask) {
    /* literal form of :191: popc(mask) > 8 && table[(mask>>3)-63] & (1 << (mask&7)) */
    if (__builtin_popcount((unsigned)mask) <= 8) return 0;
    return (table[(mask >> 3) - 63] & (1 << (mask & 7))) != 0;
}

static void ring_masks(const int* ring, int v, int th, int* mdark, int* mbright) {
    /* diffType :62-67: bit0 "darker" x-v < -th ; bit1 "brighter" x-v > th.  calcMask's early
     * returns (:81,:95,:109,:123) only skip work when no 9-run can exist, so the full masks give
     * the same predicate. */
    int m1 = 0, m2 = 0;
    for (int k = 0; k < 16; k++) {
        int diff = ring[k] - v;
        m1 |= (diff < -th) << k;
        m2 |= (diff > th) << k;
    }
    *mdark = m1;
    *mbright = m2;
}

int orc_fast_score(const uint8_t* img, int stride, int x, int y, int th) {
    /* isKeyPoint2 :220-265 + cornerScore :195-218.  Returns 0 when (x,y) is not a corner at th. */
    const int v = img[y * stride + x];
    /* quick reject on ring bits 4 and 12 (:238-243) */
    {
        int d4 = img[y * stride + x + 3] - v, d12 = img[y * stride + x - 3] - v;
        if (!(d4 < -th || d4 > th || d12 < -th || d12 > th)) return 0;
    }
    int ring[16];
    for (int k = 0; k < 16; k++) ring[k] = img[(y + k_ring_dy[k]) * stride + x + k_ring_dx[k]];
    int m1, m2;
    ring_masks(ring, v, th, &m1, &m2);
    if (!orc_fast_is_corner_masks(m1, m2)) return 0;
    int lo = th + 1, hi = 255; /* binary search in [th+1, 255] */
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        ring_masks(ring, v, mid, &m1, &m2);
        if (orc_fast_is_corner_masks(m1, m2)) lo = mid + 1;
        else hi = mid - 1;
    }
    return lo - 1;
}

/* ------------------------------------------------------------------------------------------------
 * E2+E3  tileCalcKeypoints_kernel + joinDetectAsync  (code/src/cuda/Fast_gpu.cu:283-341,379-388)
 *
 * The reference kernel is racy.  DETERMINISTIC RULE DEFINED HERE (its intent):
 *   tiles are 32x32 ROI pixels anchored at ROI (3,3); tested pixels are ROI [3,dim-3).
 *   pass 1: S1(p) = score(p) if p is a corner at th_high else 0.  p is kept if S1(p) is strictly
 *           greater than S1 of its 8 neighbours.  A tile is "non-empty" if it keeps any pixel.
 *   pass 2: only for empty tiles: S2(p) = score(p) if corner at th_low else 0; p (in an empty tile)
 *           is kept if S2(p) is strictly greater than Sf(q) for its 8 neighbours q, where
 *           Sf(q) = S2(q) if q's tile is empty, S1(q) otherwise.
 *   score(p) (largest threshold at which p is still a corner) does not depend on the threshold the
 *   binary search starts from, so one score map at th_low serves both passes.
 *   Output order: raster (y, x); truncated to `cap` (the reference keeps an arbitrary subset of
 *   10000 when the atomic counter overflows, :308; here: the first `cap` in raster order).
 * ---------------------------------------------------------------------------------------------- */
int orc_fast_detect(const uint8_t* level, int w, int h, int stride, int th_high, int th_low,
                    int16_t* xs, int16_t* ys, uint8_t* scores, int cap) {
    const int rw = w - 2 * FAST_BORDER, rh = h - 2 * FAST_BORDER;
    if (rw < 7 || rh < 7) return 0;
    const uint8_t* roi = level + FAST_BORDER * stride + FAST_BORDER;
    uint8_t* slow = (uint8_t*)calloc((size_t)rw * rh, 1); /* score at th_low, 0 if none */
    for (int i = 3; i < rh - 3; i++)
        for (int j = 3; j < rw - 3; j++) slow[i * rw + j] = (uint8_t)orc_fast_score(roi, stride, j, i, th_low);

    const int ntx = (rw + 31) / 32, nty = (rh + 31) / 32;
    uint8_t* has1 = (uint8_t*)calloc((size_t)ntx * nty, 1);
    uint8_t* keep1 = (uint8_t*)calloc((size_t)rw * rh, 1);
#define S1(ii, jj) (slow[(ii)*rw + (jj)] >= th_high ? slow[(ii)*rw + (jj)] : 0)
    for (int i = 3; i < rh - 3; i++)
        for (int j = 3; j < rw - 3; j++) {
            int s = S1(i, j);
            if (!s) continue;
            int ismax = 1;
            for (int dy = -1; dy <= 1 && ismax; dy++)
                for (int dx = -1; dx <= 1; dx++)
                    if ((dy | dx) && !(s > S1(i + dy, j + dx))) {
                        ismax = 0;
                        break;
                    }
            if (ismax) {
                keep1[i * rw + j] = 1;
                has1[((i - 3) / 32) * ntx + (j - 3) / 32] = 1;
            }
        }
    int n = 0;
    for (int i = 3; i < rh - 3; i++)
        for (int j = 3; j < rw - 3; j++) {
            int keep = 0;
            if (has1[((i - 3) / 32) * ntx + (j - 3) / 32]) {
                keep = keep1[i * rw + j];
            } else {
                int s = slow[i * rw + j];
                if (s) {
                    keep = 1;
                    for (int dy = -1; dy <= 1 && keep; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            if (!(dy | dx)) continue;
                            int qi = i + dy, qj = j + dx, sf;
                            if (qi < 3 || qj < 3 || qi >= rh - 3 || qj >= rw - 3) sf = 0;
                            else if (has1[((qi - 3) / 32) * ntx + (qj - 3) / 32]) sf = S1(qi, qj);
                            else sf = slow[qi * rw + qj];
                            if (!(s > sf)) {
                                keep = 0;
                                break;
                            }
                        }
                }
            }
            if (keep) {
                if (n < cap) {
                    xs[n] = (int16_t)j;
                    ys[n] = (int16_t)i;
                    scores[n] = slow[i * rw + j];
                }
                n++;
            }
        }
#undef S1
    free(slow);
    free(has1);
    free(keep1);
    return n < cap ? n : cap;
}

/* ------------------------------------------------------------------------------------------------
 * E4  ORBextractor::DistributeOctTree + ExtractorNode::DivideNode (code/src/ORBextractor.cc:407-689)
 *
 * Nodes are axis-aligned boxes (UL=(x0,y0), UR=(x1,y0), BL=(x0,y1), BR=(x1,y1)); the std::list is an
 * index-linked list.  The one non-deterministic point of the reference — std::sort over
 * (population, node POINTER) pairs, :610 — is DEFINED HERE as (population, creation sequence):
 * among equal populations the most recently created node is split first.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int x0, y0, x1, y1;
    int key_off, nkeys; /* slice of the key pool (candidate indices, insertion order) */
    int prev, next;
    int no_more;
} onode;

typedef struct {
    onode* nodes;
    int n_nodes, cap_nodes;
    int* pool;
    int n_pool, cap_pool;
    int head, tail, size;
} otree;

static int ot_new_node(otree* t) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes *= 2;
        t->nodes = (onode*)realloc(t->nodes, sizeof(onode) * t->cap_nodes);
    }
    return t->n_nodes++;
}
static int ot_pool_alloc(otree* t, int n) {
    while (t->n_pool + n > t->cap_pool) {
        t->cap_pool *= 2;
        t->pool = (int*)realloc(t->pool, sizeof(int) * t->cap_pool);
    }
    int off = t->n_pool;
    t->n_pool += n;
    return off;
}
static void ot_push_front(otree* t, int id) {
    t->nodes[id].prev = -1;
    t->nodes[id].next = t->head;
    if (t->head >= 0) t->nodes[t->head].prev = id;
    t->head = id;
    if (t->tail < 0) t->tail = id;
    t->size++;
}
static void ot_push_back(otree* t, int id) {
    t->nodes[id].next = -1;
    t->nodes[id].prev = t->tail;
    if (t->tail >= 0) t->nodes[t->tail].next = id;
    t->tail = id;
    if (t->head < 0) t->head = id;
    t->size++;
}
static int ot_erase(otree* t, int id) { /* returns next */
    int p = t->nodes[id].prev, n = t->nodes[id].next;
    if (p >= 0) t->nodes[p].next = n; else t->head = n;
    if (n >= 0) t->nodes[n].prev = p; else t->tail = p;
    t->size--;
    return n;
}

/* DivideNode :407-463; children ids in c[0..3] = n1..n4 (created, not yet linked) */
static void ot_divide(otree* t, int id, const int16_t* xs, const int16_t* ys, int c[4]) {
    const int x0 = t->nodes[id].x0, y0 = t->nodes[id].y0, x1 = t->nodes[id].x1, y1 = t->nodes[id].y1;
    const int halfX = (int)ceilf((float)(x1 - x0) / 2);
    const int halfY = (int)ceilf((float)(y1 - y0) / 2);
    const int xm = x0 + halfX, ym = y0 + halfY;
    const int nk = t->nodes[id].nkeys;
    for (int k = 0; k < 4; k++) {
        c[k] = ot_new_node(t);
        int off = ot_pool_alloc(t, nk);
        onode* n = &t->nodes[c[k]];
        n->key_off = off;
        n->nkeys = 0;
        n->no_more = 0;
        n->prev = n->next = -1;
    }
    onode* n1 = &t->nodes[c[0]];
    onode* n2 = &t->nodes[c[1]];
    onode* n3 = &t->nodes[c[2]];
    onode* n4 = &t->nodes[c[3]];
    n1->x0 = x0; n1->y0 = y0; n1->x1 = xm; n1->y1 = ym;
    n2->x0 = xm; n2->y0 = y0; n2->x1 = x1; n2->y1 = ym;
    n3->x0 = x0; n3->y0 = ym; n3->x1 = xm; n3->y1 = y1;
    n4->x0 = xm; n4->y0 = ym; n4->x1 = x1; n4->y1 = y1;
    const int koff = t->nodes[id].key_off;
    for (int i = 0; i < nk; i++) {
        int ci = t->pool[koff + i];
        onode* dst;
        if ((float)xs[ci] < (float)xm) dst = ((float)ys[ci] < (float)ym) ? n1 : n3;
        else dst = ((float)ys[ci] < (float)ym) ? n2 : n4;
        t->pool[dst->key_off + dst->nkeys++] = ci;
    }
    for (int k = 0; k < 4; k++)
        if (t->nodes[c[k]].nkeys == 1) t->nodes[c[k]].no_more = 1;
}

typedef struct { int size; int seq; } sizeseq; /* seq == node id == creation order */
static int cmp_sizeseq(const void* a, const void* b) {
    const sizeseq* p = (const sizeseq*)a;
    const sizeseq* q = (const sizeseq*)b;
    if (p->size != q->size) return p->size < q->size ? -1 : 1;
    return p->seq < q->seq ? -1 : (p->seq > q->seq ? 1 : 0);
}

int orc_distribute_octree(const int16_t* xs, const int16_t* ys, const uint8_t* scores, int n, int roi_w,
                          int roi_h, int N, int32_t* out_idx, int out_cap) {
    if (n <= 0) return 0;
    otree t;
    t.cap_nodes = 4096; t.n_nodes = 0;
    t.nodes = (onode*)malloc(sizeof(onode) * t.cap_nodes);
    t.cap_pool = n * 8 + 64; t.n_pool = 0;
    t.pool = (int*)malloc(sizeof(int) * t.cap_pool);
    t.head = t.tail = -1; t.size = 0;

    /* :468-496 */
    const int nIni = (int)roundf((float)roi_w / (float)roi_h);
    const float hX = (float)roi_w / (float)nIni;
    int* ini = (int*)malloc(sizeof(int) * (nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; i++) {
        int id = ot_new_node(&t);
        onode* nd = &t.nodes[id];
        nd->x0 = (int)(hX * (float)i);
        nd->x1 = (int)(hX * (float)(i + 1));
        nd->y0 = 0;
        nd->y1 = roi_h;
        nd->key_off = ot_pool_alloc(&t, n);
        nd->nkeys = 0;
        nd->no_more = 0;
        ot_push_back(&t, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; i++) {
        int r = (int)((float)xs[i] / hX);
        onode* nd = &t.nodes[ini[r]];
        t.pool[nd->key_off + nd->nkeys++] = i;
    }
    /* :498-511 */
    for (int lit = t.head; lit >= 0;) {
        if (t.nodes[lit].nkeys == 1) {
            t.nodes[lit].no_more = 1;
            lit = t.nodes[lit].next;
        } else if (t.nodes[lit].nkeys == 0) lit = ot_erase(&t, lit);
        else lit = t.nodes[lit].next;
    }

    int finish = 0;
    sizeseq* vs = (sizeseq*)malloc(sizeof(sizeseq) * (size_t)(n * 4 + 16));
    sizeseq* vprev = (sizeseq*)malloc(sizeof(sizeseq) * (size_t)(n * 4 + 16));
    int nvs = 0;
    while (!finish) {
        int prev_size = t.size;
        int n_to_expand = 0;
        nvs = 0;
        for (int lit = t.head; lit >= 0;) {
            if (t.nodes[lit].no_more) {
                lit = t.nodes[lit].next;
                continue;
            }
            int c[4];
            ot_divide(&t, lit, xs, ys, c);
            for (int k = 0; k < 4; k++) {
                if (t.nodes[c[k]].nkeys > 0) {
                    ot_push_front(&t, c[k]);
                    if (t.nodes[c[k]].nkeys > 1) {
                        n_to_expand++;
                        vs[nvs].size = t.nodes[c[k]].nkeys;
                        vs[nvs].seq = c[k];
                        nvs++;
                    }
                }
            }
            lit = ot_erase(&t, lit);
        }
        if (t.size >= N || t.size == prev_size) {
            finish = 1;
        } else if (t.size + n_to_expand * 3 > N) {
            while (!finish) {
                prev_size = t.size;
                int nprev = nvs;
                memcpy(vprev, vs, sizeof(sizeseq) * (size_t)nprev);
                nvs = 0;
                qsort(vprev, (size_t)nprev, sizeof(sizeseq), cmp_sizeseq);
                for (int j = nprev - 1; j >= 0; j--) {
                    int c[4];
                    int id = vprev[j].seq;
                    ot_divide(&t, id, xs, ys, c);
                    for (int k = 0; k < 4; k++) {
                        if (t.nodes[c[k]].nkeys > 0) {
                            ot_push_front(&t, c[k]);
                            if (t.nodes[c[k]].nkeys > 1) {
                                vs[nvs].size = t.nodes[c[k]].nkeys;
                                vs[nvs].seq = c[k];
                                nvs++;
                            }
                        }
                    }
                    ot_erase(&t, id);
                    if (t.size >= N) break;
                }
                if (t.size >= N || t.size == prev_size) finish = 1;
            }
        }
    }
    /* :667-686 best response per node, first maximum wins */
    int nout = 0;
    for (int lit = t.head; lit >= 0; lit = t.nodes[lit].next) {
        const onode* nd = &t.nodes[lit];
        int best = t.pool[nd->key_off];
        float maxr = (float)scores[best];
        for (int k = 1; k < nd->nkeys; k++) {
            int ci = t.pool[nd->key_off + k];
            if ((float)scores[ci] > maxr) {
                best = ci;
                maxr = (float)scores[ci];
            }
        }
        if (nout < out_cap) out_idx[nout] = best;
        nout++;
    }
    free(vs); free(vprev); free(ini); free(t.nodes); free(t.pool);
    return nout < out_cap ? nout : out_cap;
}

/* ------------------------------------------------------------------------------------------------
 * deterministic float helpers.  The reference calls atan2f / cosf / sinf under nvcc -use_fast_math
 * (CMakeLists.txt:34), i.e. device approximations whose bits are not reproducible off that GPU.
 * DEFINED HERE: polynomial evaluations made only of correctly-rounded +,*,/,fma so that CPU and
 * gfx950 agree bit for bit; |error| < 2e-7 rad (tests/test_oracle_kat.py checks against libm).
 * ---------------------------------------------------------------------------------------------- */
float orc_atan2f(float y, float x) {
    if (g_convention & ORC_CONV_TRIG_LIBM) return atan2f(y, x); /* alternative: libm (correctly rounded to ~1 ulp) */
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (mx == 0.f) return 0.f;
    const float a = mn / mx;
    const float s = a * a;
    float p = 0.0028340641874819994f;
    p = fmaf(p, s, -0.016005029901862144f);
    p = fmaf(p, s, 0.042587608098983765f);
    p = fmaf(p, s, -0.07495445758104324f);
    p = fmaf(p, s, 0.10636754333972931f);
    p = fmaf(p, s, -0.14202570915222168f);
    p = fmaf(p, s, 0.19992484152317047f);
    p = fmaf(p, s, -0.3333306610584259f);
    p = fmaf(p, s, 1.0f);
    float r = a * p;
    if (ay > ax) r = 0x1.921fb6p+0f - r;
    if (x < 0.f) r = 0x1.921fb6p+1f - r;
    if (y < 0.f) r = -r;
    return r;
}

static void orc_sincosf(float a, float* sn, float* cs) {
    if (g_convention & ORC_CONV_TRIG_LIBM) {
        *sn = sinf(a);
        *cs = cosf(a);
        return;
    }
    const float k = rintf(a * 0x1.45f306p-1f); /* 2/pi */
    float r = fmaf(-k, 0x1.921fb6p+0f, a);
    r = fmaf(-k, -0x1.777a5cp-25f, r);
    const float s = r * r;
    float ps = 2.716587005124893e-06f;
    ps = fmaf(ps, s, -0.0001983911934075877f);
    ps = fmaf(ps, s, 0.008333328180015087f);
    ps = fmaf(ps, s, -0.1666666716337204f);
    ps = fmaf(ps, s, 1.0f);
    const float sinr = r * ps;
    float pc = 2.4371513063670136e-05f;
    pc = fmaf(pc, s, -0.001388652715831995f);
    pc = fmaf(pc, s, 0.04166661202907562f);
    pc = fmaf(pc, s, -0.5f);
    pc = fmaf(pc, s, 1.0f);
    const float cosr = pc;
    switch (((int)k) & 3) {
        case 0: *sn = sinr; *cs = cosr; break;
        case 1: *sn = cosr; *cs = -sinr; break;
        case 2: *sn = -sinr; *cs = -cosr; break;
        default: *sn = -cosr; *cs = sinr; break;
    }
}

void orc_sincosf_deg(float deg, float* s, float* c) {
    /* calcOrb_kernel: angle = kpt.angle * (float)(CV_PI/180.f), code/src/cuda/Orb_gpu.cu:77-79 */
    orc_sincosf(deg * 0x1.1df46ap-6f, s, c);
}

/* ------------------------------------------------------------------------------------------------
 * E5  IC_Angle_kernel  (code/src/cuda/Fast_gpu.cu:402-459)
 * ---------------------------------------------------------------------------------------------- */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int32_t* umax) {
    int m_01 = 0, m_10 = 0;
    const uint8_t* c = img + y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * c[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, m_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = c[u + v * stride], val_minus = c[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_sum += u * (val_plus + val_minus);
        }
        m_10 += m_sum;
        m_01 += v * v_sum;
    }
    float kp_dir = orc_atan2f((float)m_01, (float)m_10);
    kp_dir += (float)(kp_dir < 0) * 0x1.921fb6p+2f; /* 2.0f * CV_PI_F */
    kp_dir *= 0x1.ca5dcp+5f;                          /* 180.0f / CV_PI_F */
    return kp_dir;
}

/* ------------------------------------------------------------------------------------------------
 * E7  calcOrb_kernel  (code/src/cuda/Orb_gpu.cu:63-100)
 * sample (row = y + rn(px*b + py*a), col = x + rn(px*a - py*b)); bit = t0 < t1; rn = half-even.
 * ---------------------------------------------------------------------------------------------- */
void orc_brief(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t* desc) {
    float a, b;
    orc_sincosf_deg(angle_deg, &b, &a); /* a = cos, b = sin */
    for (int byte = 0; byte < 32; byte++) {
        int val = 0;
        for (int bit = 0; bit < 8; bit++) {
            const int8_t* p = k_pattern + (byte * 16 + bit * 2) * 2;
            const float p0x = (float)p[0], p0y = (float)p[1], p1x = (float)p[2], p1y = (float)p[3];
            int r0 = (int)lrintf(p0x * b + p0y * a), c0 = (int)lrintf(p0x * a - p0y * b);
            int r1 = (int)lrintf(p1x * b + p1y * a), c1 = (int)lrintf(p1x * a - p1y * b);
            int t0 = img[(y + r0) * stride + x + c0];
            int t1 = img[(y + r1) * stride + x + c1];
            val |= (t0 < t1) << bit;
        }
        desc[byte] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------------
 * E8  ORBextractor::operator() + ComputePyramid + ComputeKeyPointsOctTree
 *     (code/src/ORBextractor.cc:691-855)
 * ---------------------------------------------------------------------------------------------- */
int orc_extract_debug(const orc_config* cfg, const uint8_t* img, int w, int h, int stride,
                      orc_keypoint* kps, uint8_t* desc, int cap, int16_t* cand_x, int16_t* cand_y,
                      uint8_t* cand_score, int32_t* cand_count, int cand_cap_per_level,
                      uint8_t* pyr_out) {
    if (!img || w <= 0 || h <= 0) return 0; /* :750-751 empty image: outputs untouched */
    orc_tables t;
    orc_make_tables(cfg, &t);
    const int nl = cfg->nlevels;
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    uint8_t* lev[ORC_MAX_LEVELS];
    for (int l = 0; l < nl; l++) {
        orc_level_size(w, h, t.inv_scale[l], &lw[l], &lh[l]);
        lev[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
    }
    /* ComputePyramid :837-853 — level l from level l-1 (un-blurred) */
    for (int y = 0; y < h; y++) memcpy(lev[0] + (size_t)y * lw[0], img + (size_t)y * stride, (size_t)w);
    for (int l = 1; l < nl; l++)
        orc_resize_linear(lev[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lev[l], lw[l], lh[l], lw[l]);
    if (pyr_out) {
        size_t off = 0;
        for (int l = 0; l < nl; l++) {
            memcpy(pyr_out + off, lev[l], (size_t)lw[l] * lh[l]);
            off += (size_t)lw[l] * lh[l];
        }
    }

    int16_t* cx = (int16_t*)malloc(sizeof(int16_t) * ORC_FAST_CAP);
    int16_t* cy = (int16_t*)malloc(sizeof(int16_t) * ORC_FAST_CAP);
    uint8_t* cs = (uint8_t*)malloc(ORC_FAST_CAP);
    int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cfg->nfeatures + 64));
    int nout = 0;
    for (int l = 0; l < nl; l++) {
        /* ComputeKeyPointsOctTree :691-744 */
        int nc = orc_fast_detect(lev[l], lw[l], lh[l], lw[l], cfg->ini_th_fast, cfg->min_th_fast, cx, cy, cs,
                                 ORC_FAST_CAP);
        if (cand_count) {
            cand_count[l] = nc;
            int m = nc < cand_cap_per_level ? nc : cand_cap_per_level;
            memcpy(cand_x + (size_t)l * cand_cap_per_level, cx, sizeof(int16_t) * (size_t)m);
            memcpy(cand_y + (size_t)l * cand_cap_per_level, cy, sizeof(int16_t) * (size_t)m);
            memcpy(cand_score + (size_t)l * cand_cap_per_level, cs, (size_t)m);
        }
        int ns = orc_distribute_octree(cx, cy, cs, nc, lw[l] - 2 * FAST_BORDER, lh[l] - 2 * FAST_BORDER,
                                       t.features_per_level[l], sel, cfg->nfeatures + 64);
        /* blur AFTER angles are taken from the un-blurred level (:716-719) */
        uint8_t* blurred = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        orc_gaussian7(lev[l], lw[l], lh[l], lw[l], blurred, lw[l]);
        const int ksize = (int)((float)PATCH_SIZE * t.scale[l]); /* float -> int param, :717 */
        for (int k = 0; k < ns; k++) {
            if (nout >= cap) break;
            const int ci = sel[k];
            const int px = cx[ci] + FAST_BORDER, py = cy[ci] + FAST_BORDER; /* addBorder_kernel */
            orc_keypoint* kp = &kps[nout];
            kp->x = (float)px;
            kp->y = (float)py;
            kp->size = (float)ksize;
            kp->response = (float)cs[ci];
            kp->octave = l;
            kp->class_id = -1;
            kp->angle = orc_ic_angle(lev[l], lw[l], px, py, t.umax);
            orc_brief(blurred, lw[l], px, py, kp->angle, desc + (size_t)nout * 32);
            if (l != 0) { /* :808-814 */
                kp->x *= t.scale[l];
                kp->y *= t.scale[l];
            }
            nout++;
        }
        free(blurred);
    }
    free(cx); free(cy); free(cs); free(sel);
    for (int l = 0; l < nl; l++) free(lev[l]);
    return nout;
}

int orc_extract(const orc_config* cfg, const uint8_t* img, int w, int h, int stride, orc_keypoint* kps,
                uint8_t* desc, int cap) {
    return orc_extract_debug(cfg, img, w, h, stride, kps, desc, cap, 0, 0, 0, 0, 0, 0);
}
