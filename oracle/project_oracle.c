/*
 * project_oracle.c — CPU restatement of the projection + gating half of Fuse / SearchBySim3 / the keyframe-side
 * SearchByProjection overloads, and of the five routines end to end (see project_oracle.h for the conventions).
 * TEST INFRASTRUCTURE ONLY; never linked into the product.
 */
#include "project_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100 /* code/src/ORBmatcher.cc:37 */
#define TH_LOW 50   /* :38 */

/* d = A(3x3, row stride lda) * x + b: one GEMM, double accumulation, one rounding */
static void gemv3(const float* A, int lda, const float* x, const float* b, float* d) {
    for (int r = 0; r < 3; r++) {
        const double s = (double)A[lda * r] * (double)x[0] + (double)A[lda * r + 1] * (double)x[1] +
                         (double)A[lda * r + 2] * (double)x[2];
        d[r] = (float)(s + (double)b[r]);
    }
}

/* KeyFrame::IsInImage, code/src/KeyFrame.cc:816-818 */
static int is_in_image(const orc_frame_view* KF, float x, float y) {
    return (x >= KF->min_x && x < KF->max_x && y >= KF->min_y && y < KF->max_y);
}

/* MapPoint::PredictScale, code/src/MapPoint.cc:476-485 */
static int predict_scale(float max_distance, float current_dist, float log_scale_factor, int n_scale_levels) {
    const float ratio = max_distance / current_dist;
    const float lr = (float)orc_log((double)ratio); /* std::log(float) */
    int nScale = (int)ceilf(lr / log_scale_factor);
    if (nScale > n_scale_levels - 1) nScale = n_scale_levels - 1;
    if (nScale < 0) nScale = 0;
    return nScale;
}

static float norm3(const float* p) { /* cv::norm of a 3x1 CV_32F */
    const double n2 = (double)p[0] * (double)p[0] + (double)p[1] * (double)p[1] + (double)p[2] * (double)p[2];
    return (float)sqrt(n2);
}

static double dot3(const float* a, const float* b) { /* Mat::dot */
    return (double)a[0] * (double)b[0] + (double)a[1] * (double)b[1] + (double)a[2] * (double)b[2];
}

static void clear_query(const orc_window_queries* q, int i) {
    q->active[i] = 0;
    q->u[i] = q->v[i] = q->radius[i] = 0.f;
    q->level[i] = 0;
}

void orc_sim3_decompose(const float* S, float* Rcw, float* tcw, float* Ow) {
    /* cv::Mat sRcw = Scw.rowRange(0,3).colRange(0,3); const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0))); */
    const float scw = (float)sqrt(dot3(S, S));
    const float inv = (float)(1.0 / (double)scw); /* Rcw = sRcw / scw; tcw = Scw.col(3) / scw */
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Rcw[3 * r + c] = S[4 * r + c] * inv;
        tcw[r] = S[4 * r + 3] * inv;
    }
    for (int j = 0; j < 3; j++) { /* Ow = -Rcw.t() * tcw */
        const double s = (double)Rcw[0 + j] * (double)tcw[0] + (double)Rcw[3 + j] * (double)tcw[1] +
                         (double)Rcw[6 + j] * (double)tcw[2];
        Ow[j] = (float)(-s);
    }
}

void orc_sim3_relative(float s12, const float* R12, const float* t12, float* sR12, float* sR21, float* t21) {
    const float inv = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            sR12[3 * r + c] = R12[3 * r + c] * s12; /* s12 * R12 */
            sR21[3 * r + c] = R12[3 * c + r] * inv; /* (1.0 / s12) * R12.t() */
        }
    for (int r = 0; r < 3; r++) { /* t21 = -sR21 * t12 */
        const double s = (double)sR21[3 * r] * (double)t12[0] + (double)sR21[3 * r + 1] * (double)t12[1] +
                         (double)sR21[3 * r + 2] * (double)t12[2];
        t21[r] = (float)(-s);
    }
}

/* The shared tail of Fuse(:779-815), Fuse-Scw(:929-964) and SearchByProjection(KF,Scw)(:299-333) once p3Dc and Ow
 * are known. */
static void project_world_point(const orc_frame_view* KF, const orc_camera* cam, const float* p3Dw, const float* p3Dc,
                                const float* Ow, const float* Pn, float mfMax, float mfMin, float lsf, float th,
                                const orc_window_queries* q, int i) {
    if (p3Dc[2] < 0.0f) return; /* Depth must be positive */
    const float invz = 1.0f / p3Dc[2];
    const float x = p3Dc[0] * invz;
    const float y = p3Dc[1] * invz;
    const float u = cam->fx * x + cam->cx;
    const float v = cam->fy * y + cam->cy;
    if (!is_in_image(KF, u, v)) return; /* Point must be inside the image */
    const float maxDistance = 1.2f * mfMax;
    const float minDistance = 0.8f * mfMin;
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist3D = norm3(PO);
    if (dist3D < minDistance || dist3D > maxDistance) return;
    if (dot3(PO, Pn) < 0.5 * (double)dist3D) return; /* Viewing angle must be less than 60 deg */
    const int nPredictedLevel = predict_scale(mfMax, dist3D, lsf, KF->nlevels);
    q->active[i] = 1;
    q->u[i] = u;
    q->v[i] = v;
    q->radius[i] = th * KF->scale_factors[nPredictedLevel];
    q->level[i] = nPredictedLevel;
}

void orc_fuse_queries(const orc_frame_view* KF, const orc_camera* cam, const float* T, float lsf,
                      const orc_mappoint_view* mp, float th, const orc_window_queries* q) {
    float Ow[3];
    orc_camera_center(T, Ow); /* pKF->GetCameraCenter(): Ow = -Rcw.t() * tcw (KeyFrame::SetPose) */
    for (int i = 0; i < mp->n; i++) {
        clear_query(q, i);
        if (!mp->valid[i]) continue; /* :770-774 */
        const float* p3Dw = mp->Xw + 3 * (size_t)i;
        float p3Dc[3];
        gemv3(T, 4, p3Dw, (const float[3]){T[3], T[7], T[11]}, p3Dc); /* Rcw * p3Dw + tcw */
        project_world_point(KF, cam, p3Dw, p3Dc, Ow, mp->normal + 3 * (size_t)i, mp->max_dist[i], mp->min_dist[i], lsf,
                            th, q, i);
    }
}

void orc_sim3_world_queries(const orc_frame_view* KF, const orc_camera* cam, const float* S, float lsf,
                            const orc_mappoint_view* mp, float th, const orc_window_queries* q) {
    float Rcw[9], tcw[3], Ow[3];
    orc_sim3_decompose(S, Rcw, tcw, Ow);
    for (int i = 0; i < mp->n; i++) {
        clear_query(q, i);
        if (!mp->valid[i]) continue; /* :920 / :290 */
        const float* p3Dw = mp->Xw + 3 * (size_t)i;
        float p3Dc[3];
        gemv3(Rcw, 3, p3Dw, tcw, p3Dc);
        project_world_point(KF, cam, p3Dw, p3Dc, Ow, mp->normal + 3 * (size_t)i, mp->max_dist[i], mp->min_dist[i], lsf,
                            th, q, i);
    }
}

void orc_sim3_pair_queries(const orc_frame_view* KF, const orc_camera* cam, const float* Tsw, const float* sR,
                           const float* t, float lsf, const orc_mappoint_view* mp, float th,
                           const orc_window_queries* q) {
    for (int i = 0; i < mp->n; i++) {
        clear_query(q, i);
        if (!mp->valid[i]) continue; /* :1057-1061 */
        const float* p3Dw = mp->Xw + 3 * (size_t)i;
        float a[3], p[3];
        gemv3(Tsw, 4, p3Dw, (const float[3]){Tsw[3], Tsw[7], Tsw[11]}, a); /* p3Dc1 = R1w * p3Dw + t1w */
        gemv3(sR, 3, a, t, p);                                              /* p3Dc2 = sR21 * p3Dc1 + t21 */
        if (p[2] < 0.0f) continue;
        const float invz = 1.0f / p[2]; /* const float invz = 1.0 / z: a correctly rounded quotient either way */
        const float x = p[0] * invz;
        const float y = p[1] * invz;
        const float u = cam->fx * x + cam->cx;
        const float v = cam->fy * y + cam->cy;
        if (!is_in_image(KF, u, v)) continue;
        const float maxDistance = 1.2f * mp->max_dist[i];
        const float minDistance = 0.8f * mp->min_dist[i];
        const float dist3D = norm3(p); /* cv::norm(p3Dc2) */
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = predict_scale(mp->max_dist[i], dist3D, lsf, KF->nlevels);
        q->active[i] = 1;
        q->u[i] = u;
        q->v[i] = v;
        q->radius[i] = th * KF->scale_factors[nPredictedLevel];
        q->level[i] = nPredictedLevel;
    }
}

void orc_frame_kf_queries(const orc_frame_view* F, const orc_camera* cam, const float* T, float lsf,
                          const orc_mappoint_view* mp, float th, const orc_window_queries* q) {
    float Ow[3];
    orc_camera_center(T, Ow); /* const cv::Mat Ow = -Rcw.t() * tcw, :1362 */
    for (int i = 0; i < mp->n; i++) {
        clear_query(q, i);
        if (!mp->valid[i]) continue; /* !pMP || isBad() || sAlreadyFound.count(pMP), :1377 */
        const float* x3Dw = mp->Xw + 3 * (size_t)i;
        float x3Dc[3];
        gemv3(T, 4, x3Dw, (const float[3]){T[3], T[7], T[11]}, x3Dc);
        const float xc = x3Dc[0];
        const float yc = x3Dc[1];
        const float invzc = (float)(1.0 / (double)x3Dc[2]); /* no depth test in this overload */
        const float u = cam->fx * xc * invzc + cam->cx;
        const float v = cam->fy * yc * invzc + cam->cy;
        if (u < F->min_x || u > F->max_x) continue;
        if (v < F->min_y || v > F->max_y) continue;
        if (!(u >= F->min_x && v >= F->min_y)) continue; /* NaN coordinates never reach GetFeaturesInArea */
        const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
        const float dist3D = norm3(PO);
        const float maxDistance = 1.2f * mp->max_dist[i];
        const float minDistance = 0.8f * mp->min_dist[i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = predict_scale(mp->max_dist[i], dist3D, lsf, F->nlevels);
        q->active[i] = 1;
        q->u[i] = u;
        q->v[i] = v;
        q->radius[i] = th * F->scale_factors[nPredictedLevel];
        q->level[i] = nPredictedLevel;
    }
}

/* ---- the routines end to end ---- */

typedef struct {
    orc_window_queries q;
    void* block;
} query_store;

static query_store alloc_queries(int n) {
    query_store s;
    const size_t m = (size_t)(n > 0 ? n : 1);
    s.block = malloc(m * (1 + 4 * 4));
    uint8_t* b = (uint8_t*)s.block;
    s.q.u = (float*)b;
    s.q.v = s.q.u + m;
    s.q.radius = s.q.v + m;
    s.q.level = (int32_t*)(s.q.radius + m);
    s.q.active = (uint8_t*)(s.q.level + m);
    return s;
}

static int apply_th_low(int n, int32_t* best_idx, const int32_t* best_dist) {
    int nFused = 0;
    for (int i = 0; i < n; i++) {
        if (best_idx[i] >= 0 && best_dist[i] <= TH_LOW) nFused++; /* :873 / :995 */
        else best_idx[i] = -1;
    }
    return nFused;
}

int orc_fuse(const orc_frame_view* KF, const orc_camera* cam, const float* Tcw12, float lsf, const float* inv_sigma2,
             const orc_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist) {
    query_store s = alloc_queries(mp->n);
    orc_fuse_queries(KF, cam, Tcw12, lsf, mp, th, &s.q);
    orc_search_window_best(KF, mp->n, s.q.active, s.q.u, s.q.v, s.q.radius, s.q.level, mp->desc, 1, inv_sigma2,
                           best_idx, best_dist);
    free(s.block);
    return apply_th_low(mp->n, best_idx, best_dist);
}

int orc_fuse_sim3(const orc_frame_view* KF, const orc_camera* cam, const float* Scw12, float lsf,
                  const orc_mappoint_view* mp, float th, int32_t* best_idx, int32_t* best_dist) {
    query_store s = alloc_queries(mp->n);
    orc_sim3_world_queries(KF, cam, Scw12, lsf, mp, th, &s.q);
    orc_search_window_best(KF, mp->n, s.q.active, s.q.u, s.q.v, s.q.radius, s.q.level, mp->desc, 0, NULL, best_idx,
                           best_dist);
    free(s.block);
    return apply_th_low(mp->n, best_idx, best_dist);
}

int orc_search_by_sim3(const orc_frame_view* KF1, const orc_frame_view* KF2, const orc_camera* cam, const float* T1w,
                       const float* T2w, float s12, const float* R12, const float* t12, float lsf1, float lsf2,
                       const orc_mappoint_view* mp1, const orc_mappoint_view* mp2, float th, int32_t* match12) {
    float sR12[9], sR21[9], t21[3];
    orc_sim3_relative(s12, R12, t12, sR12, sR21, t21);
    const int N1 = mp1->n, N2 = mp2->n;
    int32_t* vnMatch1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N1 > 0 ? N1 : 1));
    int32_t* vnMatch2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N2 > 0 ? N2 : 1));
    int32_t* dist = (int32_t*)malloc(sizeof(int32_t) * (size_t)((N1 > N2 ? N1 : N2) + 1));
    { /* Transform from KF1 to KF2 and search, :1054-1127 */
        query_store s = alloc_queries(N1);
        orc_sim3_pair_queries(KF2, cam, T1w, sR21, t21, lsf2, mp1, th, &s.q);
        orc_search_window_best(KF2, N1, s.q.active, s.q.u, s.q.v, s.q.radius, s.q.level, mp1->desc, 0, NULL, vnMatch1,
                               dist);
        for (int i = 0; i < N1; i++)
            if (!(vnMatch1[i] >= 0 && dist[i] <= TH_HIGH)) vnMatch1[i] = -1; /* :1124 */
        free(s.block);
    }
    { /* Transform from KF2 to KF1 and search, :1130-1203 */
        query_store s = alloc_queries(N2);
        orc_sim3_pair_queries(KF1, cam, T2w, sR12, t12, lsf1, mp2, th, &s.q);
        orc_search_window_best(KF1, N2, s.q.active, s.q.u, s.q.v, s.q.radius, s.q.level, mp2->desc, 0, NULL, vnMatch2,
                               dist);
        for (int i = 0; i < N2; i++)
            if (!(vnMatch2[i] >= 0 && dist[i] <= TH_HIGH)) vnMatch2[i] = -1; /* :1200 */
        free(s.block);
    }
    int nFound = 0; /* Check agreement, :1205-1218 */
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                match12[i1] = idx2;
                nFound++;
            }
        }
    }
    free(vnMatch1); free(vnMatch2); free(dist);
    return nFound;
}

int orc_search_by_projection_sim3(const orc_frame_view* KF, const orc_camera* cam, const float* Scw12, float lsf,
                                  const orc_mappoint_view* mp, int th, int32_t* kp_to_point) {
    query_store s = alloc_queries(mp->n);
    orc_sim3_world_queries(KF, cam, Scw12, lsf, mp, (float)th, &s.q); /* radius = th * mvScaleFactors[...], int th */
    int32_t* lo = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mp->n > 0 ? mp->n : 1));
    for (int i = 0; i < mp->n; i++) lo[i] = s.q.level[i] - 1; /* :352 */
    /* GetFeaturesInArea(u, v, radius) without levels, then the explicit octave test: the same set as a level-checked
     * query because max_level = nPredictedLevel >= 0 always enables the check (matcher_oracle.c, features_in_area) */
    const int nm = orc_search_window_greedy(KF, mp->n, s.q.active, s.q.u, s.q.v, s.q.radius, lo, s.q.level, mp->desc,
                                            NULL, TH_LOW, 0, kp_to_point);
    free(lo);
    free(s.block);
    return nm;
}

int orc_search_by_projection_frame_kf(const orc_frame_view* F, const orc_camera* cam, const float* Tcw12, float lsf,
                                      const orc_mappoint_view* mp, const float* mp_angle, float th, int orb_dist,
                                      int check_orientation, int32_t* kp_to_point) {
    query_store s = alloc_queries(mp->n);
    orc_frame_kf_queries(F, cam, Tcw12, lsf, mp, th, &s.q);
    int32_t* lo = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mp->n > 0 ? mp->n : 1));
    int32_t* hi = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mp->n > 0 ? mp->n : 1));
    for (int i = 0; i < mp->n; i++) { /* GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1), :1412 */
        lo[i] = s.q.level[i] - 1;
        hi[i] = s.q.level[i] + 1;
    }
    const int nm = orc_search_window_greedy(F, mp->n, s.q.active, s.q.u, s.q.v, s.q.radius, lo, hi, mp->desc, mp_angle,
                                            orb_dist, check_orientation, kp_to_point);
    free(lo); free(hi);
    free(s.block);
    return nm;
}
