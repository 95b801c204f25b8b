/* frame_oracle.c — see frame_oracle.h.  Plain C, scalar, no libm in the arithmetic that must match the GPU. */
#include "frame_oracle.h"
#include "orb_oracle.h" /* orc_get_convention: tools/convention_sensitivity.py */

#include <math.h>
#include <string.h>

#define GRID_COLS 64 /* FRAME_GRID_COLS, code/include/Frame.h:38 */
#define GRID_ROWS 48 /* FRAME_GRID_ROWS, code/include/Frame.h:37 */

/* ln x = e ln2 + 2 atanh(s), s = (m-1)/(m+1), x = m 2^e with m in [sqrt(1/2), sqrt(2)); 12 odd terms, Horner in s^2 */
double orc_log(double x) {
    union { double d; uint64_t u; } b;
    b.d = x;
    int e = (int)((b.u >> 52) & 0x7ff) - 1023;
    b.u = (b.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL; /* m in [1, 2) */
    double m = b.d;
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0;
    p = p * z + 1.0 / 19.0;
    p = p * z + 1.0 / 17.0;
    p = p * z + 1.0 / 15.0;
    p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0;
    p = p * z + 1.0 / 9.0;
    p = p * z + 1.0 / 7.0;
    p = p * z + 1.0 / 5.0;
    p = p * z + 1.0 / 3.0;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

/* cvUndistortPointsInternal with R = I, P = K, criteria COUNT 5 (published algorithm, OpenCV 3.4 / 4.x) */
static void undistort_point(const orc_camera* cam, float u, float v, float* uo, float* vo) {
    const double fx = (double)cam->fx, fy = (double)cam->fy, cx = (double)cam->cx, cy = (double)cam->cy;
    const double k1 = (double)cam->k1, k2 = (double)cam->k2, p1 = (double)cam->p1, p2 = (double)cam->p2,
                 k3 = (double)cam->k3;
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    const int iters = (orc_get_convention() & ORC_CONV_UNDISTORT_20) ? 20 : 5;
    for (int j = 0; j < iters; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
        const double deltaX = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
        const double deltaY = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *uo = (float)(fx * x + cx);
    *vo = (float)(fy * y + cy);
}

void orc_undistort_keypoints(const orc_camera* cam, int32_t n, const float* xy, float* xy_un) {
    if (cam->k1 == 0.0f) { /* mDistCoef.at<float>(0)==0.0 -> mvKeysUn = mvKeys */
        memcpy(xy_un, xy, sizeof(float) * 2 * (size_t)n);
        return;
    }
    for (int32_t i = 0; i < n; i++) undistort_point(cam, xy[2 * i], xy[2 * i + 1], &xy_un[2 * i], &xy_un[2 * i + 1]);
}

void orc_image_bounds(const orc_camera* cam, int32_t width, int32_t height, float* b) {
    if (cam->k1 != 0.0f) {
        float m[4][2];
        undistort_point(cam, 0.0f, 0.0f, &m[0][0], &m[0][1]);
        undistort_point(cam, (float)width, 0.0f, &m[1][0], &m[1][1]);
        undistort_point(cam, 0.0f, (float)height, &m[2][0], &m[2][1]);
        undistort_point(cam, (float)width, (float)height, &m[3][0], &m[3][1]);
        b[0] = m[0][0] < m[2][0] ? m[0][0] : m[2][0]; /* min(mat(0,0), mat(2,0)) */
        b[1] = m[1][0] > m[3][0] ? m[1][0] : m[3][0];
        b[2] = m[0][1] < m[1][1] ? m[0][1] : m[1][1];
        b[3] = m[2][1] > m[3][1] ? m[2][1] : m[3][1];
    } else {
        b[0] = 0.0f;
        b[1] = (float)width;
        b[2] = 0.0f;
        b[3] = (float)height;
    }
}

int32_t orc_assign_features_to_grid(int32_t n, const float* xy, const float* b, int32_t* cell_of, int32_t* cell_start,
                                    int32_t* cell_items) {
    const float inv_w = (float)GRID_COLS / (b[1] - b[0]); /* Frame.cc:259-260 */
    const float inv_h = (float)GRID_ROWS / (b[3] - b[2]);
    const int ncell = GRID_COLS * GRID_ROWS;
    for (int c = 0; c <= ncell; c++) cell_start[c] = 0;
    int32_t inside = 0;
    for (int32_t i = 0; i < n; i++) {
        const int px = (int)roundf((xy[2 * i] - b[0]) * inv_w);
        const int py = (int)roundf((xy[2 * i + 1] - b[2]) * inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) {
            cell_of[i] = -1;
        } else {
            cell_of[i] = px * GRID_ROWS + py;
            cell_start[cell_of[i] + 1]++;
            inside++;
        }
    }
    for (int c = 0; c < ncell; c++) cell_start[c + 1] += cell_start[c];
    {
        static int32_t fill[GRID_COLS * GRID_ROWS];
        for (int c = 0; c < ncell; c++) fill[c] = cell_start[c];
        for (int32_t i = 0; i < n; i++)
            if (cell_of[i] >= 0) cell_items[fill[cell_of[i]]++] = i; /* push_back in index order */
    }
    return inside;
}

void orc_camera_center(const float* T, float* Ow) {
    for (int j = 0; j < 3; j++) { /* -(Rcw^T tcw)_j, double accumulation, one rounding */
        const double s = (double)T[0 + j] * (double)T[3] + (double)T[4 + j] * (double)T[7] + (double)T[8 + j] * (double)T[11];
        Ow[j] = (float)(-s);
    }
}

void orc_is_in_frustum(const orc_camera* cam, const float* b, const float* T, int32_t n, const float* Xw,
                       const float* normal, const float* max_dist, const float* min_dist, float viewing_cos_limit,
                       float log_scale_factor, int32_t n_scale_levels, uint8_t* in_view, float* proj_x,
                       float* proj_y, float* view_cos, int32_t* pred_level) {
    float Ow[3];
    orc_camera_center(T, Ow);
    for (int32_t i = 0; i < n; i++) {
        in_view[i] = 0; /* pMP->mbTrackInView = false */
        const float* P = Xw + 3 * (size_t)i;
        float Pc[3];
        for (int r = 0; r < 3; r++) { /* mRcw*P+mtcw */
            const double s = (double)T[4 * r] * (double)P[0] + (double)T[4 * r + 1] * (double)P[1] +
                             (double)T[4 * r + 2] * (double)P[2];
            Pc[r] = (float)(s + (double)T[4 * r + 3]);
        }
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = cam->fx * Pc[0] * invz + cam->cx;
        const float v = cam->fy * Pc[1] * invz + cam->cy;
        if (u < b[0] || u > b[1]) continue;
        if (v < b[2] || v > b[3]) continue;
        const float maxD = 1.2f * max_dist[i], minD = 0.8f * min_dist[i];
        const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
        const double n2 = (double)PO[0] * (double)PO[0] + (double)PO[1] * (double)PO[1] + (double)PO[2] * (double)PO[2];
        const float dist = (float)sqrt(n2); /* cv::norm */
        if (dist < minD || dist > maxD) continue;
        const float* Pn = normal + 3 * (size_t)i;
        const double dot = (double)PO[0] * (double)Pn[0] + (double)PO[1] * (double)Pn[1] + (double)PO[2] * (double)Pn[2];
        const float vc = (float)(dot / (double)dist);
        if (vc < viewing_cos_limit) continue;
        const float ratio = max_dist[i] / dist; /* PredictScale */
        const float lr = (float)orc_log((double)ratio);
        int nScale = (int)ceilf(lr / log_scale_factor);
        if (nScale > n_scale_levels - 1) nScale = n_scale_levels - 1;
        if (nScale < 0) nScale = 0;
        in_view[i] = 1;
        proj_x[i] = u;
        proj_y[i] = v;
        view_cos[i] = vc;
        pred_level[i] = nScale;
    }
}

void orc_project_last_frame(const orc_camera* cam, const float* b, const float* T, int32_t n, const float* Xw,
                            const uint8_t* has_point, uint8_t* valid, float* u_out, float* v_out) {
    for (int32_t i = 0; i < n; i++) {
        valid[i] = 0;
        if (!has_point[i]) continue; /* pMP && !LastFrame.mvbOutlier[i], ORBmatcher.cc:1248-1250 */
        const float* P = Xw + 3 * (size_t)i;
        float Pc[3];
        for (int r = 0; r < 3; r++) { /* Rcw*x3Dw+tcw, :1253 */
            const double s = (double)T[4 * r] * (double)P[0] + (double)T[4 * r + 1] * (double)P[1] +
                             (double)T[4 * r + 2] * (double)P[2];
            Pc[r] = (float)(s + (double)T[4 * r + 3]);
        }
        const float invzc = (float)(1.0 / (double)Pc[2]); /* const float invzc = 1.0/x3Dc.at<float>(2), :1257 */
        if (!(invzc >= 0.0f)) continue;                    /* if(invzc<0) continue, :1259 (a NaN depth is dropped too) */
        const float u = cam->fx * Pc[0] * invzc + cam->cx;
        const float v = cam->fy * Pc[1] * invzc + cam->cy;
        if (u < b[0] || u > b[1]) continue; /* :1265-1268 */
        if (v < b[2] || v > b[3]) continue;
        if (!(u >= b[0] && v >= b[2])) continue; /* NaN coordinates never reach GetFeaturesInArea */
        valid[i] = 1;
        u_out[i] = u;
        v_out[i] = v;
    }
}
