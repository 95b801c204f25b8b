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
    const float ox = F->has_grid_origin ? F->grid_min_x : F->min_x; /* the origin the cells were assigned with */
    const float oy = F->has_grid_origin ? F->grid_min_y : F->min_y;
    *px = (int)roundf((F->x[i] - ox) * F->grid_inv_w);
    *py = (int)roundf((F->y[i] - oy) * F->grid_inv_h);
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

/* ================= M3, M5, M6, M7 ================= */
static void apply_rot_hist(const int32_t* hist, const int32_t* items, const int32_t* bins, int n_rot, int32_t* target,
                           int* nmatches) {
    int ind1, ind2, ind3;
    orc_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int j = 0; j < n_rot; j++) {
        const int b = bins[j];
        if (b == ind1 || b == ind2 || b == ind3) continue;
        target[items[j]] = -1;
        (*nmatches)--;
    }
}

static int rot_bin(float a1, float a2) {
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (1.0f / HISTO_LENGTH));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* lower_bound over the flattened node ids (std::map::lower_bound) */
static int fv_lower_bound(const orc_featvec* fv, int from, int id) {
    int k = from;
    while (k < fv->n_nodes && fv->node_id[k] < id) k++;
    return k;
}

int orc_search_by_bow(int variant, int32_t n1, const uint8_t* desc1, const float* angle1, const uint8_t* valid1,
                      const orc_featvec* fv1, int32_t n2, const uint8_t* desc2, const float* angle2,
                      const uint8_t* valid2, const orc_featvec* fv2, float nn_ratio, int check_orientation,
                      int32_t* match_of_2, int32_t* match_of_1) {
    int32_t* m2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 > 0 ? n2 : 1)); /* target -> source */
    int32_t* m1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1)); /* source -> target */
    int32_t* rot_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    int32_t* rot_bins = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    int32_t hist[HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    for (int i = 0; i < n2; i++) m2[i] = -1;
    for (int i = 0; i < n1; i++) m1[i] = -1;
    int nmatches = 0, n_rot = 0;
    int k1 = 0, k2 = 0;
    while (k1 < fv1->n_nodes && k2 < fv2->n_nodes) {
        if (fv1->node_id[k1] == fv2->node_id[k2]) {
            for (int a = fv1->off[k1]; a < fv1->off[k1 + 1]; a++) {
                const int idx1 = fv1->idx[a];
                if (!valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int b = fv2->off[k2]; b < fv2->off[k2 + 1]; b++) {
                    const int idx2 = fv2->idx[b];
                    if (m2[idx2] >= 0) continue; /* vpMapPointMatches[realIdxF] / vbMatched2[idx2] */
                    if (variant == 1 && !valid2[idx2]) continue;
                    const int dist = orc_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                const int pass = variant == 0 ? (bestDist1 <= TH_LOW) : (bestDist1 < TH_LOW);
                if (pass && (float)bestDist1 < nn_ratio * (float)bestDist2) {
                    m2[bestIdx2] = idx1;
                    m1[idx1] = bestIdx2;
                    if (check_orientation) {
                        const int bin = rot_bin(angle1[idx1], angle2[bestIdx2]);
                        rot_items[n_rot] = variant == 0 ? bestIdx2 : idx1; /* :226 vs :563 */
                        rot_bins[n_rot] = bin;
                        n_rot++;
                        hist[bin]++;
                    }
                    nmatches++;
                }
            }
            k1++;
            k2++;
        } else if (fv1->node_id[k1] < fv2->node_id[k2]) {
            k1 = fv_lower_bound(fv1, k1, fv2->node_id[k2]);
        } else {
            k2 = fv_lower_bound(fv2, k2, fv1->node_id[k1]);
        }
    }
    if (check_orientation) {
        /* variant 0 clears vpMapPointMatches[target]; variant 1 clears vpMatches12[source]; the taken flags of
         * variant 1 (vbMatched2) are not reset, and the loop is over anyway */
        if (variant == 0) {
            int ind1, ind2, ind3;
            orc_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
            for (int j = 0; j < n_rot; j++) {
                const int b = rot_bins[j];
                if (b == ind1 || b == ind2 || b == ind3) continue;
                const int t = rot_items[j];
                if (m2[t] >= 0) m1[m2[t]] = -1;
                m2[t] = -1;
                nmatches--;
            }
        } else {
            int ind1, ind2, ind3;
            orc_three_maxima(hist, HISTO_LENGTH, &ind1, &ind2, &ind3);
            for (int j = 0; j < n_rot; j++) {
                const int b = rot_bins[j];
                if (b == ind1 || b == ind2 || b == ind3) continue;
                /* the reference only clears vpMatches12 (vbMatched2 is a local that dies with the call, :590-594);
                 * match_of_2 is an output here, so the dropped pair leaves it too */
                if (m1[rot_items[j]] >= 0) m2[m1[rot_items[j]]] = -1;
                m1[rot_items[j]] = -1;
                nmatches--;
            }
        }
    }
    if (match_of_2) memcpy(match_of_2, m2, sizeof(int32_t) * (size_t)n2);
    if (match_of_1) memcpy(match_of_1, m1, sizeof(int32_t) * (size_t)n1);
    free(m2); free(m1); free(rot_items); free(rot_bins);
    return nmatches;
}

int orc_search_for_triangulation(int32_t n1, const float* x1, const float* y1, const float* angle1,
                                 const uint8_t* desc1, const uint8_t* free1, const orc_featvec* fv1, int32_t n2,
                                 const float* x2, const float* y2, const int32_t* octave2, const float* angle2,
                                 const uint8_t* desc2, const uint8_t* free2, const orc_featvec* fv2,
                                 const float* F12, float ex, float ey, const float* scale_factors2,
                                 const float* level_sigma2_2, int check_orientation, int32_t* matches12) {
    int nmatches = 0, n_rot = 0;
    int32_t* rot_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    int32_t* rot_bins = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 > 0 ? n1 : 1));
    int32_t hist[HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int k1 = 0, k2 = 0;
    while (k1 < fv1->n_nodes && k2 < fv2->n_nodes) {
        if (fv1->node_id[k1] == fv2->node_id[k2]) {
            for (int a = fv1->off[k1]; a < fv1->off[k1 + 1]; a++) {
                const int idx1 = fv1->idx[a];
                if (!free1[idx1]) continue; /* already a MapPoint */
                const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int b = fv2->off[k2]; b < fv2->off[k2 + 1]; b++) {
                    const int idx2 = fv2->idx[b];
                    if (!free2[idx2]) continue; /* vbMatched2 is never set in the reference (:634-720) */
                    const int dist = orc_descriptor_distance(d1, desc2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const float distex = ex - x2[idx2];
                    const float distey = ey - y2[idx2];
                    if (distex * distex + distey * distey < 100 * scale_factors2[octave2[idx2]]) continue;
                    { /* CheckDistEpipolarLine :131-148 */
                        const float a_ = x1[idx1] * F12[0] + y1[idx1] * F12[3] + F12[6];
                        const float b_ = x1[idx1] * F12[1] + y1[idx1] * F12[4] + F12[7];
                        const float c_ = x1[idx1] * F12[2] + y1[idx1] * F12[5] + F12[8];
                        const float num = a_ * x2[idx2] + b_ * y2[idx2] + c_;
                        const float den = a_ * a_ + b_ * b_;
                        if (den == 0) continue;
                        const float dsqr = num * num / den;
                        if (!(dsqr < 3.84 * level_sigma2_2[octave2[idx2]])) continue;
                    }
                    bestIdx2 = idx2;
                    bestDist = dist;
                }
                if (bestIdx2 >= 0) {
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        const int bin = rot_bin(angle1[idx1], angle2[bestIdx2]);
                        rot_items[n_rot] = idx1;
                        rot_bins[n_rot] = bin;
                        n_rot++;
                        hist[bin]++;
                    }
                }
            }
            k1++;
            k2++;
        } else if (fv1->node_id[k1] < fv2->node_id[k2]) {
            k1 = fv_lower_bound(fv1, k1, fv2->node_id[k2]);
        } else {
            k2 = fv_lower_bound(fv2, k2, fv1->node_id[k1]);
        }
    }
    if (check_orientation) apply_rot_hist(hist, rot_items, rot_bins, n_rot, matches12, &nmatches);
    free(rot_items); free(rot_bins);
    return nmatches;
}

void orc_search_window_best(const orc_frame_view* KF, int32_t nq, const uint8_t* valid, const float* u,
                            const float* v, const float* radius, const int32_t* pred_level, const uint8_t* qdesc,
                            int chi2_gate, const float* inv_sigma2, int32_t* best_idx, int32_t* best_dist) {
    orc_grid g;
    grid_build(KF, &g);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(KF->n > 0 ? KF->n : 1));
    for (int i = 0; i < nq; i++) {
        best_idx[i] = -1;
        best_dist[i] = 256;
        if (!valid[i]) continue;
        const int lvl = pred_level[i];
        const int nv = features_in_area(KF, &g, u[i], v[i], radius[i], -1, -1, vIndices, KF->n);
        int bestDist = 256, bestIdx = -1;
        for (int j = 0; j < nv; j++) {
            const int idx = vIndices[j];
            const int kpLevel = KF->octave[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            if (chi2_gate) {
                const float exx = u[i] - KF->x[idx];
                const float eyy = v[i] - KF->y[idx];
                const float e2 = exx * exx + eyy * eyy;
                if (e2 * inv_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, KF->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        best_idx[i] = bestIdx;
        best_dist[i] = bestDist;
    }
    free(vIndices);
    free(g.items);
}

int orc_search_window_greedy(const orc_frame_view* F, int32_t nq, const uint8_t* valid, const float* u,
                             const float* v, const float* radius, const int32_t* min_level,
                             const int32_t* max_level, const uint8_t* qdesc, const float* q_angle, int max_dist,
                             int check_orientation, int32_t* kp_to_query) {
    orc_grid g;
    grid_build(F, &g);
    int32_t* vIndices = (int32_t*)malloc(sizeof(int32_t) * (size_t)(F->n > 0 ? F->n : 1));
    int32_t* rot_items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq > 0 ? nq : 1));
    int32_t* rot_bins = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nq > 0 ? nq : 1));
    int32_t hist[HISTO_LENGTH];
    memset(hist, 0, sizeof(hist));
    int nmatches = 0, n_rot = 0;
    for (int k = 0; k < F->n; k++) kp_to_query[k] = -1;
    for (int i = 0; i < nq; i++) {
        if (!valid[i]) continue;
        const int nv = features_in_area(F, &g, u[i], v[i], radius[i], min_level[i], max_level[i], vIndices, F->n);
        int bestDist = 256, bestIdx = -1;
        for (int j = 0; j < nv; j++) {
            const int idx = vIndices[j];
            if (F->excluded && F->excluded[idx]) continue;
            if (kp_to_query[idx] >= 0) continue;
            const int dist = orc_descriptor_distance(qdesc + (size_t)i * 32, F->desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        if (bestDist <= max_dist) {
            kp_to_query[bestIdx] = i;
            nmatches++;
            if (check_orientation) {
                const int bin = rot_bin(q_angle[i], F->angle[bestIdx]);
                rot_items[n_rot] = bestIdx;
                rot_bins[n_rot] = bin;
                n_rot++;
                hist[bin]++;
            }
        }
    }
    if (check_orientation) apply_rot_hist(hist, rot_items, rot_bins, n_rot, kp_to_query, &nmatches);
    free(vIndices); free(rot_items); free(rot_bins); free(g.items);
    return nmatches;
}

/* MapPoint::ComputeDistinctiveDescriptors, code/src/MapPoint.cc:361-391 */
static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

int orc_distinctive_descriptor(const uint8_t* descs, int32_t n, int32_t* median_out) {
    int* dist = (int*)malloc(sizeof(int) * (size_t)n * (size_t)n);
    int* row = (int*)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) {
        dist[(size_t)i * n + i] = 0;
        for (int j = i + 1; j < n; j++) {
            const int d = orc_descriptor_distance(descs + 32 * (size_t)i, descs + 32 * (size_t)j);
            dist[(size_t)i * n + j] = d;
            dist[(size_t)j * n + i] = d;
        }
    }
    int best_median = 2147483647, best_idx = 0;
    for (int i = 0; i < n; i++) {
        memcpy(row, dist + (size_t)i * n, sizeof(int) * (size_t)n);
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        const int median = row[(int)(0.5 * (n - 1))];
        if (median < best_median) {
            best_median = median;
            best_idx = i;
        }
    }
    free(dist);
    free(row);
    if (median_out) *median_out = best_median;
    return best_idx;
}
