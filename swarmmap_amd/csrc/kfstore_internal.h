// kfstore_internal.h — the keyframe store handle, shared by kfstore.cpp (which owns it) and exchange.cpp (which appends
// the records an all-gather delivered and searches this rank's new keyframes, both without a host hop for the payload).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/swarmorb.h"
#include "kfstore_device.h"

namespace so {

struct KfMeta {  // host mirror of a slot's header
    int32_t agent = -1;
    uint64_t keyframe_id = 0;
    int32_t n = 0, nv = 0;
    size_t bytes = 0;
    bool used = false;
};

}  // namespace so

struct so_kfstore {
    int device = 0;
    so::KfStoreDev dev{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, jobs_free = nullptr;
    int head = 0, count = 0;  // ring position of the next record, keyframes held
    int64_t n_desc = 0;       // bound-keypoint rows held
    std::vector<so::KfMeta> meta;
    int scan_qper = 0;        // query rows per lane of the scan (0 = by size; SWARMORB_KF_SCAN_QPER)
    // device scratch
    int stage_records = 8;
    uint8_t* d_stage = nullptr;  // host records land here before the append kernel
    uint8_t* d_query = nullptr;  // the record being looked up
    uint32_t* d_qdesc = nullptr;
    uint16_t* d_qidx = nullptr;
    int32_t* d_nqv = nullptr;
    int32_t* d_votes = nullptr;
    int32_t* d_jobs = nullptr;
    int job_cap = 0;
    int32_t* d_cand = nullptr;
    uint32_t* d_keys = nullptr;
    uint32_t* d_keys2 = nullptr;
    uint32_t* d_taken = nullptr;
    // pinned host block and the views into it
    uint8_t* h_pin = nullptr;
    int32_t* h_votes = nullptr;
    int32_t* h_nqv = nullptr;
    int32_t* h_jobs = nullptr;
    int32_t* h_cand = nullptr;
    uint32_t* h_keys2 = nullptr;
    uint32_t* h_taken = nullptr;
    uint32_t* h_keys = nullptr;
    float* h_angle = nullptr;
    double stats[6] = {0, 0, 0, 0, 0, 0};
    // scratch kept from call to call
    std::vector<int> scratch_qidx, scratch_rot_item, scratch_rot_b;
    std::vector<int32_t> scratch_m1, scratch_mp;
    std::vector<uint8_t> scratch_taken;
    std::vector<float> scratch_angle;
};

namespace so {

int kfstore_append_device(so_kfstore* s, const uint8_t* d_src, size_t src_stride, const so_keyframe_header* hdrs,
                          const uint8_t* skip, int n, int32_t* slots_out, hipStream_t stream);
struct KfCand {
    int slot, votes;
};
int kfstore_match_device(so_kfstore* s, const so_keyframe_header& qh, const float* q_angle, const std::vector<int>& qidx,
                         const std::vector<KfCand>& cands, const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs,
                         int32_t* n_out);
int kfstore_search_device(so_kfstore* s, const uint8_t* d_query, const so_keyframe_header& qh, const float* q_angle,
                          const int32_t* q_mp, const so_kf_search_params* p, bool votes_only, int32_t* votes_out,
                          so_kf_candidate* out, int32_t* pairs, int32_t* n_out, int32_t* n_evaluated);

}  // namespace so
