// kfstore.cpp — keyframe store + cross-agent candidate search behind the C ABI (include/swarmorb.h: so_kfstore_*).
//
// The device-side counterpart of what the reference's server does for every new keyframe
// (AgentMediator::CheckOverlapCandidates, code/src/AgentMediator.cc:140-202: look it up in EVERY other agent's whole
// keyframe database; AgentMediator::GetSim3, :204-262: SearchByBoW(pCurrentKF, pKF) per candidate, >= 20 pairs).
// Phase 1 (detection) is one scan of the store on the GPU; phase 2 runs the exact SearchByBoW(KF, KF) semantics of
// code/src/ORBmatcher.cc:481-597 - all features in one vocabulary node - on the few keyframes phase 1 lets through:
// K-lists from the GPU, the order-dependent resolve on the host in the reference's order, an exhausted list re-run on
// the GPU under the current "taken" gate (distances never come from the CPU).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kfstore_internal.h"
#include "so_common.h"

using namespace so;

static_assert(sizeof(so_keyframe_header) == 128, "keyframe record header is 128 bytes");
static_assert(offsetof(so_keyframe_header, agent_id) == 8 && offsetof(so_keyframe_header, n_keypoints) == 12 &&
                  offsetof(so_keyframe_header, checksum) == 32 && offsetof(so_keyframe_header, flags) == 104 &&
                  offsetof(so_keyframe_header, n_map_points) == 108,
              "kfstore_kernels.hip reads the header by offset");

namespace {

constexpr uint32_t kMagic = 0x464B4F53u;  // "SOKF"
constexpr int HISTO_LENGTH = 30;          // code/src/ORBmatcher.cc:39
constexpr int kTopK = 8;

bool header_ok(const so_keyframe_header& h, int kp) {
    if (h.magic != kMagic || h.header_bytes != sizeof(so_keyframe_header)) return false;
    // the device takes "has map-point ids" from flags & 1, the host from the version: a peer's record in which the two
    // disagree is refused instead of being interpreted two ways
    if (h.version == 1) return h.flags == 0 && h.n_keypoints >= 0 && h.n_keypoints <= kp;
    return h.version == 2 && (h.flags & SO_KF_FLAG_MAP_POINTS) && h.n_keypoints >= 0 && h.n_keypoints <= kp &&
           h.n_map_points >= 0 && h.n_map_points <= h.n_keypoints;
}

size_t record_bytes(const so_keyframe_header& h) {
    return h.version == 2 ? so_keyframe_record_size2(h.n_keypoints) : so_keyframe_record_size(h.n_keypoints);
}

// ORBmatcher::ComputeThreeMaxima (code/src/ORBmatcher.cc:1475-1506)
void three_maxima(const int* sizes, int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

int rot_bin(float a1, float a2) {  // ORBmatcher.cc:557-563
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (1.0f / HISTO_LENGTH));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

template <typename T>
hipError_t dev_alloc(T** p, size_t count) {
    return hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
}

}  // namespace

namespace so {

// Validates the headers of `n` gathered / uploaded records lying in DEVICE memory at d_src (host copies of the headers
// in `hdrs`), assigns ring slots (skip[j] != 0 or an invalid header: not stored) and launches the append on `stream`.
int kfstore_append_device(so_kfstore* s, const uint8_t* d_src, size_t src_stride, const so_keyframe_header* hdrs,
                          const uint8_t* skip, int n, int32_t* slots_out, hipStream_t stream) {
    if (n <= 0) return SO_OK;
    if (n > s->job_cap) {
        last_error_ref() = "too many records in one append";
        return SO_ERR_CAPACITY;
    }
    // the job list of the previous append may still be read by its kernel
    SO_HIP(hipEventSynchronize(s->jobs_free));
    int stored = 0;
    for (int j = 0; j < n; j++) {
        int slot = -1;
        if (!(skip && skip[j]) && header_ok(hdrs[j], s->dev.kp)) {
            slot = s->head;
            s->head = (s->head + 1) % s->dev.capacity;
            if (s->count < s->dev.capacity) s->count++;
            KfMeta& m = s->meta[(size_t)slot];
            s->n_desc -= m.nv;
            m.agent = hdrs[j].agent_id;
            m.keyframe_id = hdrs[j].keyframe_id;
            m.n = hdrs[j].n_keypoints;
            m.nv = hdrs[j].version == 2 ? hdrs[j].n_map_points : hdrs[j].n_keypoints;
            m.bytes = record_bytes(hdrs[j]);
            m.used = true;
            s->n_desc += m.nv;
            stored++;
        }
        s->h_jobs[j] = slot;
        if (slots_out) slots_out[j] = slot;
    }
    if (stored == 0) return SO_OK;
    SO_HIP(hipMemcpyAsync(s->d_jobs, s->h_jobs, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream));
    launch_kf_append(s->dev, d_src, src_stride, s->d_jobs, n, stream);
    SO_HIP(hipGetLastError());
    SO_HIP(hipEventRecord(s->jobs_free, stream));
    return SO_OK;
}

// The search proper.  The query record lies in device memory (d_query); its header and per-keypoint angle / binding are
// host copies.  Runs on s->stream; the caller has ordered d_query's producer before it.
int kfstore_search_device(so_kfstore* s, const uint8_t* d_query, const so_keyframe_header& qh, const float* q_angle,
                          const int32_t* q_mp, const so_kf_search_params* p, bool votes_only, int32_t* votes_out,
                          so_kf_candidate* out, int32_t* pairs, int32_t* n_out, int32_t* n_evaluated) {
    hipStream_t st = s->stream;
    const int kp = s->dev.kp, cap = s->dev.capacity;
    const int n1 = qh.n_keypoints;
    for (int k = 0; k < 6; k++) s->stats[k] = 0.0;
    if (n_out) *n_out = 0;
    if (n_evaluated) *n_evaluated = 0;
    // bound keypoints of the query, in keypoint order (the kernel's compaction yields the same list)
    std::vector<int>& qidx = s->scratch_qidx;
    qidx.clear();
    for (int i = 0; i < n1; i++)
        if (!q_mp || q_mp[i] >= 0) qidx.push_back(i);
    const int nqv = (int)qidx.size();
    const int n_slots = s->count;  // the ring fills from slot 0: used slots are [0, count)
    if (votes_out)
        for (int k = 0; k < cap; k++) votes_out[k] = -1;
    launch_kf_compact_query(d_query, kp, s->d_qdesc, s->d_qidx, s->d_nqv, st);
    SO_HIP(hipMemsetAsync(s->d_votes, 0, sizeof(int32_t) * (size_t)cap, st));
    SO_HIP(hipEventRecord(s->ev0, st));
    launch_kf_scan(s->dev, n_slots, s->d_qdesc, nqv, qh.agent_id, p->th_low, p->nn_ratio, s->d_votes, s->scan_qper, st);
    SO_HIP(hipEventRecord(s->ev1, st));
    SO_HIP(hipGetLastError());
    SO_HIP(hipMemcpyAsync(s->h_votes, s->d_votes, sizeof(int32_t) * (size_t)(n_slots > 0 ? n_slots : 1), hipMemcpyDeviceToHost, st));
    SO_HIP(hipMemcpyAsync(s->h_nqv, s->d_nqv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    SO_HIP(hipStreamSynchronize(st));
    if (*s->h_nqv != nqv) {
        last_error_ref() = "keyframe search: the query record's map-point block does not match the bindings passed with it";
        return SO_ERR_INVALID_ARG;
    }
    float ms = 0.f;
    if (n_slots > 0 && nqv > 0 && hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->stats[0] = ms;
    double pairs_cmp = 0.0;
    int scanned = 0;
    std::vector<KfCand> cands;
    for (int k = 0; k < n_slots; k++) {
        const KfMeta& m = s->meta[(size_t)k];
        if (!m.used || m.agent == qh.agent_id) continue;
        scanned++;
        pairs_cmp += (double)nqv * (double)m.nv;
        const int v = s->h_votes[k];
        if (votes_out) votes_out[k] = v;
        if (v >= p->min_votes && v > 0) cands.push_back({k, v});
    }
    s->stats[1] = pairs_cmp;
    s->stats[2] = scanned;
    if (votes_only) return SO_OK;
    // candidates by (votes descending, slot ascending)
    std::stable_sort(cands.begin(), cands.end(), [](const KfCand& a, const KfCand& b) { return a.votes > b.votes; });
    const int max_c = std::min(std::min(p->max_candidates, (int)SO_KF_MAX_CANDIDATES), (int)cands.size());
    if (max_c <= 0 || nqv == 0) return SO_OK;
    if (n_evaluated) *n_evaluated = max_c;
    cands.resize((size_t)max_c);
    return kfstore_match_device(s, qh, q_angle, qidx, cands, p, out, pairs, n_out);
}

// Phase 2 on a given list of store slots: the exact SearchByBoW(KF1, KF2) of code/src/ORBmatcher.cc:481-597 with all
// features in one vocabulary node, per candidate.  The query's compacted rows are in s->d_qdesc (kf_compact_query_kernel
// has run on s->stream), qidx lists their keypoint indices.
int kfstore_match_device(so_kfstore* s, const so_keyframe_header& qh, const float* q_angle, const std::vector<int>& qidx,
                         const std::vector<KfCand>& cands, const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs,
                         int32_t* n_out) {
    hipStream_t st = s->stream;
    const int kp = s->dev.kp;
    const int n1 = qh.n_keypoints, nqv = (int)qidx.size(), max_c = (int)cands.size();
    float ms = 0.f;
    if (n_out) *n_out = 0;
    if (max_c <= 0 || nqv == 0) return SO_OK;
    s->stats[4] = max_c;
    // K-lists of every bound query keypoint against every candidate, the candidates' angles
    for (int c = 0; c < max_c; c++) s->h_cand[c] = cands[(size_t)c].slot;
    SO_HIP(hipMemcpyAsync(s->d_cand, s->h_cand, sizeof(int32_t) * (size_t)max_c, hipMemcpyHostToDevice, st));
    SO_HIP(hipEventRecord(s->ev0, st));
    launch_kf_pair_topk(s->dev, s->d_cand, max_c, s->d_qdesc, nqv, kTopK, s->d_keys, st);
    SO_HIP(hipEventRecord(s->ev1, st));
    SO_HIP(hipGetLastError());
    SO_HIP(hipMemcpyAsync(s->h_keys, s->d_keys, sizeof(uint32_t) * (size_t)max_c * (size_t)nqv * kTopK, hipMemcpyDeviceToHost, st));
    for (int c = 0; c < max_c; c++)
        SO_HIP(hipMemcpyAsync(s->h_angle + (size_t)c * kp, s->dev.angle + (size_t)cands[(size_t)c].slot * kp,
                              sizeof(float) * (size_t)s->meta[(size_t)cands[(size_t)c].slot].n, hipMemcpyDeviceToHost, st));
    SO_HIP(hipStreamSynchronize(st));
    if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->stats[3] = ms;
    int n_pass = 0;
    std::vector<int32_t>& m1 = s->scratch_m1;
    std::vector<uint8_t>& taken = s->scratch_taken;
    std::vector<int>&rot_item = s->scratch_rot_item, &rot_b = s->scratch_rot_b;
    for (int c = 0; c < max_c; c++) {
        const int slot = cands[(size_t)c].slot;
        const KfMeta& M = s->meta[(size_t)slot];
        const int n2 = M.n;
        const float* angle2 = s->h_angle + (size_t)c * kp;
        m1.assign((size_t)n1, -1);           // vpMatches12 as indices, ORBmatcher.cc:492
        taken.assign((size_t)n2, 0);         // vbMatched2, :493
        rot_item.clear();
        rot_b.clear();
        int hist[HISTO_LENGTH] = {0};
        int nm = 0;
        const uint32_t* keys = s->h_keys + (size_t)c * (size_t)nqv * kTopK;
        for (int qi = 0; qi < nqv; qi++) {  // the node's features of KF1 in stored order = ascending keypoint index, :514
            const int idx1 = qidx[(size_t)qi];
            int found = 0, walked = 0, e_idx[2] = {-1, -1}, e_dist[2] = {256, 256};
            for (; walked < kTopK && found < 2; walked++) {
                const uint32_t key = keys[(size_t)qi * kTopK + walked];
                if (key == 0xFFFFFFFFu) break;
                const int idx2 = (int)(key & 0xFFFFu);
                if (taken[(size_t)idx2]) continue;  // :537
                e_idx[found] = idx2;
                e_dist[found] = (int)(key >> 16);
                found++;
            }
            if (found < 2 && walked == kTopK && M.nv > kTopK) {
                // the K best were all taken by earlier keypoints: this one query again, on the GPU, under the gate
                uint32_t* bits = s->h_taken;
                memset(bits, 0, sizeof(uint32_t) * (size_t)((kp + 31) / 32));
                for (int k = 0; k < n2; k++)
                    if (taken[(size_t)k]) bits[k >> 5] |= 1u << (k & 31);
                SO_HIP(hipMemcpyAsync(s->d_taken, bits, sizeof(uint32_t) * (size_t)((kp + 31) / 32), hipMemcpyHostToDevice, st));
                launch_kf_pair_rerun(s->dev, slot, s->d_qdesc, qi, s->d_taken, s->d_keys2, st);
                SO_HIP(hipGetLastError());
                SO_HIP(hipMemcpyAsync(s->h_keys2, s->d_keys2, sizeof(uint32_t) * 2, hipMemcpyDeviceToHost, st));
                SO_HIP(hipStreamSynchronize(st));
                s->stats[5] += 1.0;
                found = 0;
                for (int k = 0; k < 2; k++) {
                    if (s->h_keys2[k] == 0xFFFFFFFFu) break;
                    e_idx[found] = (int)(s->h_keys2[k] & 0xFFFFu);
                    e_dist[found] = (int)(s->h_keys2[k] >> 16);
                    found++;
                }
            }
            if (found == 0) continue;
            const int bestDist1 = e_dist[0], bestIdx2 = e_idx[0], bestDist2 = found > 1 ? e_dist[1] : 256;
            if (bestDist1 < p->th_low && (float)bestDist1 < p->nn_ratio * (float)bestDist2) {  // :550-551
                m1[(size_t)idx1] = bestIdx2;
                taken[(size_t)bestIdx2] = 1;
                if (p->check_orientation) {
                    const int b = rot_bin(q_angle[idx1], angle2[bestIdx2]);
                    rot_item.push_back(idx1);
                    rot_b.push_back(b);
                    hist[b]++;
                }
                nm++;
            }
        }
        if (p->check_orientation) {  // :578-594
            int a, b, cc;
            three_maxima(hist, HISTO_LENGTH, a, b, cc);
            for (size_t j = 0; j < rot_item.size(); j++) {
                if (rot_b[j] == a || rot_b[j] == b || rot_b[j] == cc) continue;
                m1[(size_t)rot_item[j]] = -1;
                nm--;
            }
        }
        if (nm < p->min_matches) continue;  // AgentMediator.cc:259
        so_kf_candidate& o = out[n_pass];
        o.slot = slot;
        o.agent_id = M.agent;
        o.keyframe_id = M.keyframe_id;
        o.n_keypoints = M.n;
        o.votes = cands[(size_t)c].votes;
        o.n_matches = nm;
        o.reserved = 0;
        if (pairs) memcpy(pairs + (size_t)n_pass * (size_t)n1, m1.data(), sizeof(int32_t) * (size_t)n1);
        n_pass++;
    }
    if (n_out) *n_out = n_pass;
    return SO_OK;
}

}  // namespace so

extern "C" {

int so_kfstore_create(int device, int capacity_keyframes, int slot_keypoints, so_kfstore** out) {
    if (!out || capacity_keyframes < 1 || slot_keypoints < 1) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    if (slot_keypoints > kKfMaxKeypoints) {  // phase 2 keeps one key per keypoint and wave in LDS (4 waves x 4 B x keypoints)
        last_error_ref() = "keyframe store: at most 8192 keypoints per keyframe";
        return SO_ERR_CAPACITY;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_kfstore* s = new so_kfstore();
    s->device = device;
    const size_t C = (size_t)capacity_keyframes, KP = (size_t)slot_keypoints;
    s->dev.capacity = capacity_keyframes;
    s->dev.kp = slot_keypoints;
    s->dev.rec_stride = (so_keyframe_record_size2(slot_keypoints) + 255) / 256 * 256;
    s->meta.assign(C, KfMeta{});
    s->job_cap = 4096;
    if (const char* e = getenv("SWARMORB_KF_SCAN_QPER")) s->scan_qper = atoi(e);
    const size_t keys_n = (size_t)SO_KF_MAX_CANDIDATES * KP * kTopK;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->jobs_free, hipEventDisableTiming);
    if (e == hipSuccess) e = dev_alloc(&s->dev.rec, C * s->dev.rec_stride);
    if (e == hipSuccess) e = dev_alloc(&s->dev.vdesc, C * KP * 8);
    if (e == hipSuccess) e = dev_alloc(&s->dev.vidx, C * KP);
    if (e == hipSuccess) e = dev_alloc(&s->dev.angle, C * KP);
    if (e == hipSuccess) e = dev_alloc(&s->dev.nv, C);
    if (e == hipSuccess) e = dev_alloc(&s->dev.agent, C);
    if (e == hipSuccess) e = so::memset_sync(s->dev.nv, 0, sizeof(int32_t) * C);
    if (e == hipSuccess) e = so::memset_sync(s->dev.agent, 0xFF, sizeof(int32_t) * C);
    if (e == hipSuccess) e = dev_alloc(&s->d_stage, (size_t)s->stage_records * s->dev.rec_stride);
    if (e == hipSuccess) e = dev_alloc(&s->d_query, s->dev.rec_stride);
    if (e == hipSuccess) e = dev_alloc(&s->d_qdesc, KP * 8);
    if (e == hipSuccess) e = dev_alloc(&s->d_qidx, KP);
    if (e == hipSuccess) e = dev_alloc(&s->d_nqv, 1);
    if (e == hipSuccess) e = dev_alloc(&s->d_votes, C);
    if (e == hipSuccess) e = dev_alloc(&s->d_jobs, (size_t)s->job_cap);
    if (e == hipSuccess) e = dev_alloc(&s->d_cand, (size_t)SO_KF_MAX_CANDIDATES);
    if (e == hipSuccess) e = dev_alloc(&s->d_keys, keys_n);
    if (e == hipSuccess) e = dev_alloc(&s->d_keys2, 2);
    if (e == hipSuccess) e = dev_alloc(&s->d_taken, (KP + 31) / 32);
    const size_t pin = sizeof(int32_t) * (C + 1 + (size_t)s->job_cap + SO_KF_MAX_CANDIDATES + 2 + (KP + 31) / 32) +
                       sizeof(uint32_t) * keys_n + sizeof(float) * (size_t)SO_KF_MAX_CANDIDATES * KP + 256;
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_pin, pin, hipHostMallocDefault);
    if (e != hipSuccess) {
        so_kfstore_destroy(s);
        return hip_fail(e, "keyframe store init", __FILE__, __LINE__);
    }
    uint8_t* h = s->h_pin;
    auto take = [&](size_t bytes) {
        uint8_t* r = h;
        h += (bytes + 15) / 16 * 16;
        return r;
    };
    s->h_votes = (int32_t*)take(sizeof(int32_t) * C);
    s->h_nqv = (int32_t*)take(sizeof(int32_t));
    s->h_jobs = (int32_t*)take(sizeof(int32_t) * (size_t)s->job_cap);
    s->h_cand = (int32_t*)take(sizeof(int32_t) * SO_KF_MAX_CANDIDATES);
    s->h_keys2 = (uint32_t*)take(sizeof(uint32_t) * 2);
    s->h_taken = (uint32_t*)take(sizeof(uint32_t) * ((KP + 31) / 32));
    s->h_keys = (uint32_t*)take(sizeof(uint32_t) * keys_n);
    s->h_angle = (float*)take(sizeof(float) * (size_t)SO_KF_MAX_CANDIDATES * KP);
    *out = s;
    return SO_OK;
}

void so_kfstore_destroy(so_kfstore* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void* p : {(void*)s->dev.rec, (void*)s->dev.vdesc, (void*)s->dev.vidx, (void*)s->dev.angle, (void*)s->dev.nv,
                    (void*)s->dev.agent, (void*)s->d_stage, (void*)s->d_query, (void*)s->d_qdesc, (void*)s->d_qidx,
                    (void*)s->d_nqv, (void*)s->d_votes, (void*)s->d_jobs, (void*)s->d_cand, (void*)s->d_keys,
                    (void*)s->d_keys2, (void*)s->d_taken})
        if (p) (void)hipFree(p);
    if (s->h_pin) (void)hipHostFree(s->h_pin);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->jobs_free) (void)hipEventDestroy(s->jobs_free);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int so_kfstore_append(so_kfstore* s, const uint8_t* records, size_t stride, int32_t n_records, int32_t* slots_out) {
    if (!s || n_records < 0 || (n_records > 0 && (!records || stride < sizeof(so_keyframe_header)))) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(s->device));
    std::vector<so_keyframe_header> hdrs;
    for (int base = 0; base < n_records; base += s->stage_records) {
        const int n = std::min(s->stage_records, n_records - base);
        hdrs.resize((size_t)n);
        for (int j = 0; j < n; j++) {
            const uint8_t* r = records + (size_t)(base + j) * stride;
            memcpy(&hdrs[(size_t)j], r, sizeof(so_keyframe_header));
            if (!header_ok(hdrs[(size_t)j], s->dev.kp) || record_bytes(hdrs[(size_t)j]) > stride) {
                last_error_ref() = "keyframe store: not a keyframe record, or more keypoints than a slot holds";
                return SO_ERR_INVALID_ARG;
            }
            SO_HIP(hipMemcpyAsync(s->d_stage + (size_t)j * s->dev.rec_stride, r, record_bytes(hdrs[(size_t)j]),
                                  hipMemcpyHostToDevice, s->stream));
        }
        const int rc = kfstore_append_device(s, s->d_stage, s->dev.rec_stride, hdrs.data(), nullptr, n,
                                             slots_out ? slots_out + base : nullptr, s->stream);
        if (rc != SO_OK) return rc;
        SO_HIP(hipStreamSynchronize(s->stream));  // the staging block and the caller's records are free again
    }
    return SO_OK;
}

int so_kfstore_size(const so_kfstore* s, int32_t* n_keyframes, int64_t* n_descriptors) {
    if (!s) return SO_ERR_INVALID_ARG;
    if (n_keyframes) *n_keyframes = s->count;
    if (n_descriptors) *n_descriptors = s->n_desc;
    return SO_OK;
}

}  // extern "C"

namespace {

// host record -> device + host-side views of what the resolve needs
int stage_query(so_kfstore* s, const uint8_t* rec, size_t length, so_keyframe_header* h, std::vector<float>* angle,
                std::vector<int32_t>* mp) {
    if (!rec || length < sizeof(so_keyframe_header)) return SO_ERR_INVALID_ARG;
    memcpy(h, rec, sizeof(*h));
    if (!header_ok(*h, s->dev.kp) || record_bytes(*h) > length) {
        last_error_ref() = "keyframe search: not a keyframe record, or more keypoints than a slot holds";
        return SO_ERR_INVALID_ARG;
    }
    const size_t n = (size_t)h->n_keypoints;
    angle->resize(n);
    mp->resize(n);
    const uint8_t* geo = rec + sizeof(so_keyframe_header) + n * 32;
    for (size_t i = 0; i < n; i++) memcpy(&(*angle)[i], geo + 16 * i + 8, 4);
    if (h->version == 2) {
        if (n) memcpy(mp->data(), geo + 16 * n, 4 * n);
    } else {
        std::fill(mp->begin(), mp->end(), 0);
    }
    SO_HIP(hipMemcpyAsync(s->d_query, rec, record_bytes(*h), hipMemcpyHostToDevice, s->stream));
    return SO_OK;
}

}  // namespace

extern "C" {

int so_kfstore_votes(so_kfstore* s, const uint8_t* query_record, size_t length, int32_t th_low, float nn_ratio,
                     int32_t* votes) {
    if (!s || !votes) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(s->device));
    so_keyframe_header h;
    int rc = stage_query(s, query_record, length, &h, &s->scratch_angle, &s->scratch_mp);
    if (rc != SO_OK) return rc;
    so_kf_search_params p{th_low, nn_ratio, 1, INT_MAX, INT_MAX, 0};
    return kfstore_search_device(s, s->d_query, h, s->scratch_angle.data(), s->scratch_mp.data(), &p, true, votes, nullptr,
                                 nullptr, nullptr, nullptr);
}

int so_kfstore_search(so_kfstore* s, const uint8_t* query_record, size_t length, const so_kf_search_params* p,
                      so_kf_candidate* out, int32_t* pairs, int32_t* n_out, int32_t* n_evaluated) {
    if (!s || !p || !out || !n_out || p->max_candidates < 0 || p->max_candidates > SO_KF_MAX_CANDIDATES) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(s->device));
    so_keyframe_header h;
    int rc = stage_query(s, query_record, length, &h, &s->scratch_angle, &s->scratch_mp);
    if (rc != SO_OK) return rc;
    return kfstore_search_device(s, s->d_query, h, s->scratch_angle.data(), s->scratch_mp.data(), p, false, nullptr, out,
                                 pairs, n_out, n_evaluated);
}

// Phase 2 alone, on the slots the caller names: what the reference runs AFTER its own filtering of the detection result
// (AgentMediator::DetectLoop's covisibility-consistency groups, code/src/AgentMediator.cc:384-456, between
// DetectLoopCandidates and GetSim3).  so_kfstore_votes gives the detection scores, the host picks, this matches.
int so_kfstore_match(so_kfstore* s, const uint8_t* query_record, size_t length, const int32_t* slots, int32_t n_slots,
                     const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs, int32_t* n_out) {
    if (!s || !p || !out || !n_out || n_slots < 0 || n_slots > SO_KF_MAX_CANDIDATES || (n_slots > 0 && !slots)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(s->device));
    so_keyframe_header h;
    int rc = stage_query(s, query_record, length, &h, &s->scratch_angle, &s->scratch_mp);
    if (rc != SO_OK) return rc;
    for (int k = 0; k < 6; k++) s->stats[k] = 0.0;
    std::vector<int>& qidx = s->scratch_qidx;
    qidx.clear();
    for (int i = 0; i < h.n_keypoints; i++)
        if (s->scratch_mp[(size_t)i] >= 0) qidx.push_back(i);
    std::vector<KfCand> cands;
    for (int c = 0; c < n_slots; c++) {
        if (slots[c] < 0 || slots[c] >= s->dev.capacity || !s->meta[(size_t)slots[c]].used) {
            last_error_ref() = "keyframe store: match against an empty or out-of-range slot";
            return SO_ERR_INVALID_ARG;
        }
        cands.push_back({slots[c], 0});
    }
    launch_kf_compact_query(s->d_query, s->dev.kp, s->d_qdesc, s->d_qidx, s->d_nqv, s->stream);
    SO_HIP(hipGetLastError());
    return kfstore_match_device(s, h, s->scratch_angle.data(), qidx, cands, p, out, pairs, n_out);
}

int so_kfstore_read(so_kfstore* s, int32_t slot, uint8_t* record, size_t capacity, size_t* length) {
    if (!s || slot < 0 || slot >= s->dev.capacity || !length) return SO_ERR_INVALID_ARG;
    const KfMeta& m = s->meta[(size_t)slot];
    if (!m.used) {
        last_error_ref() = "keyframe store: empty slot";
        return SO_ERR_INVALID_ARG;
    }
    *length = m.bytes;
    if (!record) return SO_OK;
    if (capacity < m.bytes) return SO_ERR_CAPACITY;
    SO_HIP(hipSetDevice(s->device));
    SO_HIP(hipMemcpyAsync(record, s->dev.rec + (size_t)slot * s->dev.rec_stride, m.bytes, hipMemcpyDeviceToHost, s->stream));
    SO_HIP(hipStreamSynchronize(s->stream));
    return SO_OK;
}

int so_kfstore_last_stats(so_kfstore* s, double* stats6) {
    if (!s || !stats6) return SO_ERR_INVALID_ARG;
    for (int k = 0; k < 6; k++) stats6[k] = s->stats[k];
    return SO_OK;
}

}  // extern "C"
