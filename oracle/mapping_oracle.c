/*
 * mapping_oracle.c — CPU restatement of CreateNewMapPoints' per-match body and MapPoint::UpdateNormalAndDepth
 * (see mapping_oracle.h for the conventions).  TEST INFRASTRUCTURE ONLY; never linked into the product.
 */
#include "mapping_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* OpenCV JacobiSVDImpl_<float> on At (n rows of length m: the columns of A), Vt accumulates the rotations */
static void jacobi_svd_f32(float* At, float* Vt, double* W, int m, int n) {
    const float eps = FLT_EPSILON * 2;
    const int max_iter = m > 30 ? m : 30;
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const float t = At[i * m + k];
            sd += (double)t * t;
        }
        W[i] = sd;
        for (int k = 0; k < n; k++) Vt[i * n + k] = 0;
        Vt[i * n + i] = 1;
    }
    for (int iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                float *Ai = At + i * m, *Aj = At + j * m;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt((double)a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = sqrt(p * p + beta * beta); /* hypot(p, beta) */
                float c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = (float)sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < m; k++) {
                    const float t0 = c * Ai[k] + s * Aj[k];
                    const float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0;
                    Aj[k] = t1;
                    a += (double)t0 * t0;
                    b += (double)t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = 1;
                float *Vi = Vt + i * n, *Vj = Vt + j * n;
                for (int k = 0; k < n; k++) {
                    const float t0 = c * Vi[k] + s * Vj[k];
                    const float t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0;
                    Vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) {
            const float t = At[i * m + k];
            sd += (double)t * t;
        }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            const double tw = W[i]; W[i] = W[j]; W[j] = tw;
            for (int k = 0; k < m; k++) { const float t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
            for (int k = 0; k < n; k++) { const float t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
        }
    }
}

void orc_svd4_last_row(const float* A, float* v4) {
    float At[16], Vt[16];
    double W[4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) At[c * 4 + r] = A[r * 4 + c]; /* transpose(src, temp_a) */
    jacobi_svd_f32(At, Vt, W, 4, 4);
    memcpy(v4, Vt + 12, 4 * sizeof(float));
}

static float norm3f(const float* p) {
    return (float)sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
}

static void camera_center(const float* T, float* Ow) { /* KeyFrame::SetPose: Ow = -Rwc * tcw */
    for (int j = 0; j < 3; j++) {
        const double s = (double)T[0 + j] * (double)T[3] + (double)T[4 + j] * (double)T[7] + (double)T[8 + j] * (double)T[11];
        Ow[j] = (float)(-s);
    }
}

void orc_triangulate_matches(const orc_tri_keyframe* k1, const orc_tri_keyframe* k2, float ratioFactor, int32_t n,
                             const float* xy1, const int32_t* octave1, const float* xy2, const int32_t* octave2,
                             uint8_t* ok, float* x3D_out) {
    const float* T1 = k1->Tcw;
    const float* T2 = k2->Tcw;
    float Ow1[3], Ow2[3];
    camera_center(T1, Ow1);
    camera_center(T2, Ow2);
    for (int32_t m = 0; m < n; m++) {
        ok[m] = 0;
        const float kp1x = xy1[2 * m], kp1y = xy1[2 * m + 1], kp2x = xy2[2 * m], kp2y = xy2[2 * m + 1];
        /* Check parallax between rays (:274-281) */
        const float xn1[3] = {(kp1x - k1->cx) * k1->invfx, (kp1y - k1->cy) * k1->invfy, 1.0f};
        const float xn2[3] = {(kp2x - k2->cx) * k2->invfx, (kp2y - k2->cy) * k2->invfy, 1.0f};
        float ray1[3], ray2[3];
        for (int j = 0; j < 3; j++) { /* Rwc * xn: Rwc = Rcw.t() */
            ray1[j] = (float)((double)T1[0 + j] * xn1[0] + (double)T1[4 + j] * xn1[1] + (double)T1[8 + j] * xn1[2]);
            ray2[j] = (float)((double)T2[0 + j] * xn2[0] + (double)T2[4 + j] * xn2[1] + (double)T2[8 + j] * xn2[2]);
        }
        const double dot = (double)ray1[0] * ray2[0] + (double)ray1[1] * ray2[1] + (double)ray1[2] * ray2[2];
        const double n1 = sqrt((double)ray1[0] * ray1[0] + (double)ray1[1] * ray1[1] + (double)ray1[2] * ray1[2]);
        const double n2 = sqrt((double)ray2[0] * ray2[0] + (double)ray2[1] * ray2[1] + (double)ray2[2] * ray2[2]);
        const float cosParallaxRays = (float)(dot / (n1 * n2));
        const float cosParallaxStereo = cosParallaxRays + 1; /* monocular: no stereo bound */
        if (!(cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (double)cosParallaxRays < 0.9998))
            continue; /* No stereo and very low parallax (:315-316) */
        /* Linear Triangulation Method (:289-296) */
        float A[16];
        for (int c = 0; c < 4; c++) {
            A[0 + c] = xn1[0] * T1[8 + c] - T1[0 + c];
            A[4 + c] = xn1[1] * T1[8 + c] - T1[4 + c];
            A[8 + c] = xn2[0] * T2[8 + c] - T2[0 + c];
            A[12 + c] = xn2[1] * T2[8 + c] - T2[4 + c];
        }
        float v[4];
        orc_svd4_last_row(A, v);
        if (v[3] == 0) continue;
        const float iw = (float)(1.0 / (double)v[3]); /* x3D.rowRange(0,3) / x3D.at<float>(3) */
        const float X[3] = {v[0] * iw, v[1] * iw, v[2] * iw};
        /* Check triangulation in front of cameras (:320-326) */
        const float z1 = (float)(((double)T1[8] * X[0] + (double)T1[9] * X[1] + (double)T1[10] * X[2]) + (double)T1[11]);
        if (z1 <= 0) continue;
        const float z2 = (float)(((double)T2[8] * X[0] + (double)T2[9] * X[1] + (double)T2[10] * X[2]) + (double)T2[11]);
        if (z2 <= 0) continue;
        /* Check reprojection error in first keyframe (:328-340) */
        const float sigmaSquare1 = k1->level_sigma2[octave1[m]];
        const float x1 = (float)(((double)T1[0] * X[0] + (double)T1[1] * X[1] + (double)T1[2] * X[2]) + (double)T1[3]);
        const float y1 = (float)(((double)T1[4] * X[0] + (double)T1[5] * X[1] + (double)T1[6] * X[2]) + (double)T1[7]);
        const float invz1 = (float)(1.0 / (double)z1);
        {
            const float u1 = k1->fx * x1 * invz1 + k1->cx;
            const float v1 = k1->fy * y1 * invz1 + k1->cy;
            const float errX1 = u1 - kp1x, errY1 = v1 - kp1y;
            if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaSquare1) continue;
        }
        /* Check reprojection error in second keyframe (:352-364) */
        const float sigmaSquare2 = k2->level_sigma2[octave2[m]];
        const float x2 = (float)(((double)T2[0] * X[0] + (double)T2[1] * X[1] + (double)T2[2] * X[2]) + (double)T2[3]);
        const float y2 = (float)(((double)T2[4] * X[0] + (double)T2[5] * X[1] + (double)T2[6] * X[2]) + (double)T2[7]);
        const float invz2 = (float)(1.0 / (double)z2);
        {
            const float u2 = k2->fx * x2 * invz2 + k2->cx;
            const float v2 = k2->fy * y2 * invz2 + k2->cy;
            const float errX2 = u2 - kp2x, errY2 = v2 - kp2y;
            if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)sigmaSquare2) continue;
        }
        /* Check scale consistency (:378-396) */
        const float nrm1[3] = {X[0] - Ow1[0], X[1] - Ow1[1], X[2] - Ow1[2]};
        const float dist1 = norm3f(nrm1);
        const float nrm2[3] = {X[0] - Ow2[0], X[1] - Ow2[1], X[2] - Ow2[2]};
        const float dist2 = norm3f(nrm2);
        if (dist1 == 0 || dist2 == 0) continue;
        const float ratioDist = dist2 / dist1;
        const float ratioOctave = k1->scale_factors[octave1[m]] / k2->scale_factors[octave2[m]];
        if (ratioDist * ratioFactor < ratioOctave || ratioDist > ratioOctave * ratioFactor) continue;
        ok[m] = 1; /* Triangulation is successful */
        x3D_out[3 * m] = X[0];
        x3D_out[3 * m + 1] = X[1];
        x3D_out[3 * m + 2] = X[2];
    }
}

void orc_update_normal_and_depth(int32_t n_points, const int32_t* off, const float* obs_Ow, const float* Xw,
                                 const float* ref_Ow, const float* ref_level_scale, const float* ref_last_scale,
                                 float* normal, float* max_dist, float* min_dist) {
    for (int32_t p = 0; p < n_points; p++) {
        const int a = off[p], b = off[p + 1];
        if (b <= a) continue; /* if (observations.empty()) return; */
        const float* Pos = Xw + 3 * (size_t)p;
        float nsum[3] = {0.f, 0.f, 0.f};
        int n = 0;
        for (int k = a; k < b; k++) {
            const float* Owi = obs_Ow + 3 * (size_t)k;
            const float normali[3] = {Pos[0] - Owi[0], Pos[1] - Owi[1], Pos[2] - Owi[2]};
            const double nr = sqrt((double)normali[0] * normali[0] + (double)normali[1] * normali[1] + (double)normali[2] * normali[2]);
            const float inv = (float)(1.0 / nr); /* normali / cv::norm(normali) */
            for (int j = 0; j < 3; j++) nsum[j] = nsum[j] + normali[j] * inv;
            n++;
        }
        const float PC[3] = {Pos[0] - ref_Ow[3 * (size_t)p], Pos[1] - ref_Ow[3 * (size_t)p + 1], Pos[2] - ref_Ow[3 * (size_t)p + 2]};
        const float dist = norm3f(PC);
        const float mx = dist * ref_level_scale[p];
        max_dist[p] = mx;
        min_dist[p] = mx / ref_last_scale[p];
        const float invn = (float)(1.0 / (double)n); /* normal / n */
        for (int j = 0; j < 3; j++) normal[3 * (size_t)p + j] = nsum[j] * invn;
    }
}
