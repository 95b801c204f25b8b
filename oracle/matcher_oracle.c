/*
 * matcher_oracle.c — CPU restatement of ORBmatcher's tracking routines (see matcher_oracle.h).
 * TEST INFRASTRUCTURE ONLY; never linked into the product.
 */
#include "matcher_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100     /* code/src/ORBmatcher.cc:37 */
#define TH_LOW 50       /* :38 */
#define HISTO_LENGTH 30 /* :39 */

/* ORBmatcher::DescriptorDistance, :1511-1525 */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ---- Frame grid: AssignFeaturesToGrid + PosInGrid, code/src/Frame.cc:277-292,433-443 ---- */
typedef struct {
    int32_t start[ORC_GRID_COLS][ORC_GRID_ROWS + 1]; /* CSR per column-major cell [ix][iy] */
    int32_t* items;
} orc_grid;

static int pos_in_grid(const orc_frame_view* F, int i, int* px, int* py) {
    *px = (int)roundf((F->x[i] - F->min_x) * F->grid_inv_w);
    *py = (int)roundf((F->y[i] - F->min_y) * F->grid_inv_h);
    if (*px < 0 || *px >= ORC_GRID_COLS || *py < 0 || *py >= ORC_GRID_ROWS) return 0;
    return 1;
}

static void grid_build(const orc_frame_view* F, orc_grid* g) {
    int32_t counts[ORC_GRID_COLS][ORC_GRID_ROWS];
    memset(counts, 0, sizeof(counts));
    for (int i = 0; i < F->n; i++) {
        int px, py;
        if (pos_in_grid(F, i, &px, &py)) counts[px][py]++;
    }
    int run = 0;
    for (int ix = 0; ix < ORC_GRID_COLS; ix++) {
        for (int iy = 0; iy < ORC_GRID_ROWS; iy++) {
            g->start[ix][iy] = run;
            run += counts[ix][iy];
        }
        g->start[ix][ORC_GRID_ROWS] = run;
    }
    g->items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(run > 0 ? run : 1));
    memset(counts, 0, sizeof(counts));
    for (int i = 0; i < F->n; i++) { /* push_back in keypoint order */
        int px, py;
        if (pos_in_grid(F, i, &px, &py)) g->items[g->start[px][py] + counts[px][py]++] = i;
    }
}

static int cell_end(const orc_grid* g, int ix, int iy) {
    return iy + 1 < ORC_GRID_ROWS ? g->start[ix][iy + 1] : g->start[ix][ORC_GRID_ROWS];
}

/* Frame::GetFeaturesInArea, code/src/Frame.cc:377-431 */
static int features_in_area(const orc_frame_view* F, const orc_grid* g, float x, float y, float r, int min_level,
                            int max_level, int32_t* out, int cap) {
    int n = 0;
    int nMinCellX = (int)floorf((x - F->min_x - r) * F->grid_inv_w);
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= ORC_GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - F->min_x + r) * F->grid_inv_w);
    if (nMaxCellX > ORC_GRID_COLS - 1) nMaxCellX = ORC_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - F->min_y - r) * F->grid_inv_h);
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= ORC_GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - F->min_y + r) * F->grid_inv_h);
    if (nMaxCellY > ORC_GRID_ROWS - 1) nMaxCellY = ORC_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (min_level > 0) || (max_level >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
            for (int j = g->start[ix][iy]; j < cell_end(g, ix, iy); j++) {
                const int k = g->items[j];
                if (bCheckLevels) {
                    if (F->octave[k] < min_level) continue;
                    if (max_level >= 0)
                        if (F->octave[k] > max_level) continue;
                }
                const float distx = F->x[k] - x;
                const float disty = F->y[k] - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n < cap) out[n] = k;
                    n++;
                }
            }
    return n < cap ? n : cap;
}

int orc_features_in_area(const orc_frame_view* F, float x, float y, float r, int min_level, int max_level,
                         int32_t* out_idx, int cap) {
    orc_grid g;
    grid_build(F, &g);
    int n = features_in_area(F, &g, x, y, r, min_level, max_level, out_idx, cap);
    free(g.items);
    return n;
}

/* ORBmatcher::ComputeThreeMaxima, :1475-1506 */
void orc_three_maxima(const int32_t* sizes, int L, int* ind1, int* ind2, int* ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1; /* callers initialise them to -1 (:1339-1341) */
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        *ind2 = -1;
        *ind3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        *ind3 = -1;
    }
}

/* M1 — :44-121 */
int orc_search_by_projection_mappoints(const orc_frame_view* F, int32_t n_mp, const uint8_t* in_view,
                                       const float* proj_x, const float* proj_y, const float* view_cos,
                                       const int32_t* pred_level, const uint8_t* mp_desc,
                                       const uint8_t* mp_has_obs, float th, float nn_ratio, int32_t* kp_to_mp) {
    orc_grid g;
    grid_build(F, &g);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(F->n > 0 ? F->n : 1));
    for (int k = 0; k < F->n; k++) kp_to_mp[k] = -1;
    int nmatches = 0;
    const int bFactor = th != 1.0f;
    for (int i = 0; i < n_mp; i++) {
        if (!in_view[i]) continue;
        const int lvl = pred_level[i];
        float r = view_cos[i] > 0.998f ? 2.5f : 4.0f; /* RadiusByViewingCos :123-128 */
        if (bFactor) r *= th;
        const int nv = features_in_area(F, &g, proj_x[i], proj_y[i], r * F->scale_factors[lvl], lvl - 1, lvl,
                                        vIndices, F->n);
        if (nv == 0) continue;
        const uint8_t* d_mp = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int j = 0; j < nv; j++) {
            const int idx = vIndices[j];
            /* F.mvpMapPoints[idx] && Observations() > 0: bound on entry, or bound earlier in this call */
            if (F->excluded && F->excluded[idx]) continue;
            if (kp_to_mp[idx] >= 0 && mp_has_obs[kp_to_mp[idx]]) continue;
            const int dist = orc_descriptor_distance(d_mp, F->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = F->octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F->octave[idx];
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            kp_to_mp[bestIdx] = i;
            nmatches++;
        }
    }
    free(vIndices);
    free(g.items);
    return nmatches;
}

/* M2 — :1223-1354 (monocular branch) */
int orc_search_by_projection_lastframe(const orc_frame_view* cur, int32_t n_last, const uint8_t* valid,
                                       const float* u, const float* v, const int32_t* last_octave,
                                       const float* last_angle, const uint8_t* mp_desc,
                                       const uint8_t* mp_has_obs, float th, int check_orientation,
                                       int32_t* kp_to_last) {
    orc_grid g;
    grid_build(cur, &g);
    int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cur->n > 0 ? cur->n : 1));
    int32_t* rot_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_last > 0 ? n_last : 1));
    int32_t* rot_bin = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_last > 0 ? n_last : 1));
    int n_rot = 0;
    int32_t hist[HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    for (int k = 0; k < cur->n; k++) kp_to_last[k] = -1;
    int nmatches = 0;
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n_last; i++) {
        if (!valid[i]) continue;
        const int nLastOctave = last_octave[i];
        const float radius = th * cur->scale_factors[nLastOctave];
        const int nv = features_in_area(cur, &g, u[i], v[i], radius, nLastOctave - 1, nLastOctave + 1, vIndices2,
                                        cur->n);
        if (nv == 0) continue;
        const uint8_t* dMP = mp_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (int j = 0; j < nv; j++) {
            const int i2 = vIndices2[j];
            if (cur->excluded && cur->excluded[i2]) continue;
            if (kp_to_last[i2] >= 0 && mp_has_obs[kp_to_last[i2]]) continue;
            const int dist = orc_descriptor_distance(dMP, cur->desc + (size_t)i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            kp_to_last[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = last_angle[i] - cur->angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rot_items[n_rot] = bestIdx2;
                rot_bin[n_rot] = bin;
                n_rot++;
                hist[bin]++;
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int j = 0; j < n_rot; j++) {
            const int b = rot_bin[j];
            if (b != ind1 && b != ind2 && b != ind3) {
                kp_to_last[rot_items[j]] = -1;
                nmatches--;
            }
        }
    }
    free(vIndices2);
    free(rot_items);
    free(rot_bin);
    free(g.items);
    return nmatches;
}

/* M4 — :375-479 */
int orc_search_for_initialization(const orc_frame_view* F1, const orc_frame_view* F2, float* prev_matched,
                                  int window, float nn_ratio, int check_orientation, int32_t* matches12) {
    orc_grid g;
    grid_build(F2, &g);
    int nmatches = 0;
    for (int i = 0; i < F1->n; i++) matches12[i] = -1;
    int32_t* vIndices2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(F2->n > 0 ? F2->n : 1));
    int* vMatchedDistance = (int*)malloc(sizeof(int) * (size_t)(F2->n > 0 ? F2->n : 1));
    int32_t* vnMatches21 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(F2->n > 0 ? F2->n : 1));
    int32_t* rot_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(F1->n > 0 ? F1->n : 1));
    int32_t* rot_bin = (int32_t*)malloc(sizeof(int32_t) * (size_t)(F1->n > 0 ? F1->n : 1));
    int n_rot = 0;
    int32_t hist[HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    for (int k = 0; k < F2->n; k++) {
        vMatchedDistance[k] = INT_MAX;
        vnMatches21[k] = -1;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i1 = 0; i1 < F1->n; i1++) {
        const int level1 = F1->octave[i1];
        if (level1 > 0) continue;
        const int nv = features_in_area(F2, &g, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, level1,
                                        level1, vIndices2, F2->n);
        if (nv == 0) continue;
        const uint8_t* d1 = F1->desc + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int j = 0; j < nv; j++) {
            const int i2 = vIndices2[j];
            const int dist = orc_descriptor_distance(d1, F2->desc + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if ((float)bestDist < (float)bestDist2 * nn_ratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    matches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = F1->angle[i1] - F2->angle[bestIdx2];
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rot_items[n_rot] = i1;
                    rot_bin[n_rot] = bin;
                    n_rot++;
                    hist[bin]++;
                }
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int j = 0; j < n_rot; j++) {
            const int b = rot_bin[j];
            if (b == ind1 || b == ind2 || b == ind3) continue;
            const int idx1 = rot_items[j];
            if (matches12[idx1] >= 0) {
                matches12[idx1] = -1;
                nmatches--;
            }
        }
    }
    for (int i1 = 0; i1 < F1->n; i1++) /* update prev matched, :472-475 */
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = F2->x[matches12[i1]];
            prev_matched[2 * i1 + 1] = F2->y[matches12[i1]];
        }
    free(vIndices2); free(vMatchedDistance); free(vnMatches21); free(rot_items); free(rot_bin); free(g.items);
    return nmatches;
}

void orc_hamming_top2(const uint8_t* A, int na, const uint8_t* B, int nb, int32_t* best_idx, int32_t* best_dist,
                      int32_t* second_dist) {
    for (int i = 0; i < na; i++) {
        int bd = 256, sd = 256, bi = -1;
        for (int j = 0; j < nb; j++) {
            const int d = orc_descriptor_distance(A + (size_t)i * 32, B + (size_t)j * 32);
            if (d < bd) {
                sd = bd;
                bd = d;
                bi = j;
            } else if (d < sd) {
                sd = d;
            }
        }
        best_idx[i] = bi;
        best_dist[i] = bd;
        second_dist[i] = sd;
    }
}
