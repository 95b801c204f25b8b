// kfstore_device.h — device-side layout of the keyframe store and the cross-agent candidate search.
//
// HBM layout of one so_kfstore (capacity C keyframes of up to KP keypoints, allocated once):
//   rec     u8   C x rec_stride     the keyframe records as received (version 1 / 2, include/swarmorb.h)
//   vdesc   u32  C x KP x 8         descriptors of the keypoints that carry a map point, compacted in keypoint order:
//                                   the rows SearchByBoW(KF, KF) compares (code/src/ORBmatcher.cc:517-521,535-541)
//   vidx    u16  C x KP             keypoint index of every compacted row
//   angle   f32  C x KP             mvKeysUn[i].angle by keypoint index (rotation histogram of the resolve)
//   nv      i32  C                  compacted rows per keyframe
//   agent   i32  C                  header.agent_id (-1: empty slot)
// At 4096 keyframes x 1024 keypoints: 218 + 134 + 8 + 17 MB = 0.13 % of the 288 GB.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so {

constexpr int kKfMaxKeypoints = 8192;  // per keyframe: 128 KB of dynamic LDS in kf_pair_topk_kernel

struct KfStoreDev {
    uint8_t* rec;
    uint32_t* vdesc;
    uint16_t* vidx;
    float* angle;
    int32_t* nv;
    int32_t* agent;
    int capacity;      // keyframes
    int kp;            // keypoints per slot (KP)
    size_t rec_stride; // bytes
};

// Append: record `src[j]` (device memory, `src_stride` apart, only those with job_slot[j] >= 0) goes to store slot
// job_slot[j]; one workgroup per record copies it and builds the compacted rows.
void launch_kf_append(const KfStoreDev& S, const uint8_t* d_src, size_t src_stride, const int32_t* d_job_slot, int n_jobs,
                      hipStream_t s);
// Query side: the same compaction for a record that is not stored (the keyframe being looked up):
// d_qdesc[nqv x 8] / d_qidx[nqv] / *d_nqv from the record at d_rec.
void launch_kf_compact_query(const uint8_t* d_rec, int kp, uint32_t* d_qdesc, uint16_t* d_qidx, int32_t* d_nqv, hipStream_t s);
// Assemble a version-2 record in device memory from a device-resident frame (descriptors, undistorted keypoints and
// octaves stay in HBM) plus a small staged block [header 128 B | angle f32 n | map_point_id i32 n].
void launch_kf_pack_record(const uint8_t* d_desc, const float2* d_xy_un, const int8_t* d_octave, const uint8_t* d_staged,
                           int n, uint8_t* d_out, size_t out_bytes, hipStream_t s);
// Phase 1: votes[slot] += number of query rows whose best / second-best row of that keyframe pass the ratio test.
// d_votes must be zero on entry; slots with agent == query_agent or agent < 0 are skipped.
// qper = query rows per lane (1, 2 or 4); 0 = chosen from nqv.
void launch_kf_scan(const KfStoreDev& S, int n_slots, const uint32_t* d_qdesc, int nqv, int query_agent, int th_low,
                    float nn_ratio, int32_t* d_votes, int qper, hipStream_t s);
// Phase 2: for candidate c (store slot d_cand[c]) and query row q the K smallest keys dist << 16 | keypoint index of
// the candidate's compacted rows: d_keys[(c * nqv + q) * K + k] (0xFFFFFFFF past the end).
void launch_kf_pair_topk(const KfStoreDev& S, const int32_t* d_cand, int n_cand, const uint32_t* d_qdesc, int nqv, int K,
                         uint32_t* d_keys, hipStream_t s);
// Phase 2, exhausted list: best two keys of query row q against store slot `slot` among the rows whose keypoint index
// is not set in the `taken` bitmap (kp bits).
void launch_kf_pair_rerun(const KfStoreDev& S, int slot, const uint32_t* d_qdesc, int q, const uint32_t* d_taken,
                          uint32_t* d_keys2, hipStream_t s);

}  // namespace so
