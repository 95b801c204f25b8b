// exchange.cpp — cross-agent keyframe descriptor exchange behind the C ABI (include/swarmorb.h: so_exchange_*).
//
// Replaces the server-side candidate query of AgentMediator::CheckOverlapCandidates (code/src/AgentMediator.cc:140-202:
// every new keyframe is looked up in every other agent's BoW inverted index) for agents sharded one per GPU: each tick
// every rank contributes ONE fixed-capacity slot with its newest keyframe's descriptors, filled by a copy kernel
// straight from the device-resident frame (no host hop), one ncclAllGather over xGMI delivers all slots to all ranks,
// and each rank brute-force matches its own slot against every peer's with hamming_top2_kernel on the gathered buffer.
// The payload is 32-64 KB per rank: latency-bound, far below the per-link xGMI limit, hence one padded collective per
// tick and no per-frame collective.
// RCCL is bound at run time (dlopen of librccl.so.1): a process that already runs a RCCL (PyTorch's) gets that very
// copy, and libswarmorb.so carries no link-time dependency on it for single-GPU users.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "dframe_internal.h"
#include "ba_device.h"
#include "kfstore_internal.h"
#include "match_device.h"
#include "so_common.h"

using namespace so;

namespace {

// the slice of rccl.h this file needs (ABI of RCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclUint8 = 1;  // ncclUint8 / ncclChar+1 in ncclDataType_t

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.lib) break;
        }
        if (!R.lib) return;
        R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.lib, "ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.lib, "ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.lib, "ncclCommDestroy");
        R.CommAbort = (decltype(R.CommAbort))dlsym(R.lib, "ncclCommAbort");
        R.AllGather = (decltype(R.AllGather))dlsym(R.lib, "ncclAllGather");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.lib, "ncclGetErrorString");
        R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllGather;
    });
    return R;
}

int rccl_fail(ncclResult_t r, const char* what) {
    const Rccl& R = rccl();
    last_error_ref() = std::string(what) + " failed: " + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error");
    return SO_ERR_HIP;
}

}  // namespace

struct so_exchange {
    int device = 0, rank = 0, world = 1, slot_keypoints = 0;
    size_t slot_bytes = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint8_t* d_slot = nullptr;      // (1 + slot_keypoints) x 32 B: header row + descriptors
    uint8_t* d_gathered = nullptr;  // world slots in rank order
    uint8_t* d_stage = nullptr;     // host descriptors land here before the slot is filled
    int32_t* d_res = nullptr;       // world x 3 x slot_keypoints: best index / best distance / second distance per peer
    uint8_t* h_pin = nullptr;       // pinned: [world headers (32 B each) | results]
    size_t h_pin_bytes = 0;
    std::vector<int32_t> counts;
    std::vector<uint64_t> sums;
    // candidate-search mode (so_exchange_create_store): the slot holds records_per_tick keyframe records, rec_stride apart
    so_kfstore* store = nullptr;
    int records_per_tick = 0;
    size_t rec_stride = 0;
    uint8_t* d_rslot = nullptr;      // this rank's records
    uint8_t* d_rgathered = nullptr;  // world x records_per_tick records in rank order
    uint8_t* d_staged = nullptr;     // header + angles + bindings of a device-resident keyframe
    uint8_t* h_rpin = nullptr;       // pinned: [world x records_per_tick headers | staged block]
    // transport of the candidate-search mode: RCCL (comm != null) or the host's own all-gather (SwarmMap's WebSocket
    // layer, MPI, ...: agents that do not share a node), which moves the slot through pinned host memory
    so_exchange_allgather_fn host_gather = nullptr;
    void* host_user = nullptr;
    uint8_t* h_slot = nullptr;      // pinned: this rank's slot
    uint8_t* h_gathered = nullptr;  // pinned: world slots
    // A tick waits for its collective with a budget (SWARMORB_COLLECTIVE_TIMEOUT_MS / so_exchange_set_timeout, default
    // 30000; 0 = wait for ever): when a peer never enters the tick the survivors get SO_ERR_TIMEOUT instead of hanging,
    // and the handle is dead - every later tick returns SO_ERR_TIMEOUT at once, destroy neither waits for the stuck
    // stream nor frees what the collective may still write.  The FIRST collective of a handle also pays RCCL's lazy
    // connection set-up and whatever start-up skew the ranks have (one still renders its stream while another is done):
    // it gets at least kFirstTickBudgetMs.  The budget covers RCCL collectives only: on the host-transport path
    // (so_exchange_create_store_host) the wait for this rank's own staging copy and the caller's all-gather callback are
    // not bounded here - the callback owns its time-outs.
    int timeout_ms = 30000;
    int collectives_done = 0;
    bool dead = false;
    unsigned* d_stall_sink = nullptr;  // so_exchange_debug_stall
    std::vector<so_keyframe_header> hdrs;
    std::vector<uint8_t> skip;
    std::vector<float> q_angle;
    std::vector<int32_t> q_mp;
};

namespace {

int dead_handle() {
    last_error_ref() = "exchange: a previous tick's collective timed out; the handle is dead (destroy it, re-create the group)";
    return SO_ERR_TIMEOUT;
}

// Waits for everything enqueued on `s` up to the collective, within the handle's budget.
int wait_collective(so_exchange* x, hipStream_t s) {
    if (x->timeout_ms <= 0) {
        SO_HIP(hipStreamSynchronize(s));
        return SO_OK;
    }
    constexpr int kFirstTickBudgetMs = 120000;
    const int budget_ms = x->collectives_done == 0 ? std::max(x->timeout_ms, kFirstTickBudgetMs) : x->timeout_ms;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) {
            x->collectives_done++;
            return SO_OK;
        }
        if (e != hipErrorNotReady) return hip_fail(e, "hipStreamQuery (exchange tick)", __FILE__, __LINE__);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > (double)budget_ms) {
            x->dead = true;
            last_error_ref() = "exchange tick: the collective did not complete within " + std::to_string(budget_ms) +
                               " ms (a peer is not taking part); the handle is dead";
            return SO_ERR_TIMEOUT;
        }
        if (ms > 0.2) std::this_thread::sleep_for(std::chrono::microseconds(50));  // spin for the common case, then nap
    }
}

int tick_common(so_exchange* x, const uint8_t* d_desc, int n, int max_dist, float ratio, int32_t* peer_counts,
                int32_t* peer_candidates) {
    const Rccl& R = rccl();
    hipStream_t s = x->stream;
    const int nk = n < x->slot_keypoints ? n : x->slot_keypoints;
    launch_exchange_fill_slot(reinterpret_cast<const uint4*>(d_desc), nk, x->slot_keypoints, x->rank,
                              reinterpret_cast<uint4*>(x->d_slot), s);
    SO_HIP(hipGetLastError());
    const ncclResult_t r = R.AllGather(x->d_slot, x->d_gathered, x->slot_bytes, kNcclUint8, x->comm, s);
    if (r != 0) return rccl_fail(r, "ncclAllGather");
    // headers of all slots -> host (keypoint counts drive the match launches)
    SO_HIP(hipMemcpy2DAsync(x->h_pin, 32, x->d_gathered, x->slot_bytes, 32, (size_t)x->world, hipMemcpyDeviceToHost, s));
    {
        const int wrc = wait_collective(x, s);
        if (wrc != SO_OK) return wrc;
    }
    for (int p = 0; p < x->world; p++) {
        int32_t hdr[2];
        memcpy(hdr, x->h_pin + 32 * (size_t)p, 8);
        memcpy(&x->sums[(size_t)p], x->h_pin + 32 * (size_t)p + 8, 8);
        if (hdr[0] < 0 || hdr[0] > x->slot_keypoints) {  // a peer built with another slot size, or a corrupted header
            last_error_ref() = "all-gather slot header: keypoint count outside [0, slot_keypoints]";
            return SO_ERR_INVALID_ARG;
        }
        x->counts[(size_t)p] = hdr[0];
        if (hdr[1] != p) {
            last_error_ref() = "all-gather slot order does not follow the rank order";
            return SO_ERR_HIP;
        }
    }
    const int mine = x->counts[(size_t)x->rank];
    const uint4* my_desc = reinterpret_cast<const uint4*>(x->d_gathered + (size_t)x->rank * x->slot_bytes + 32);
    const size_t res_stride = 3 * (size_t)x->slot_keypoints;
    for (int p = 0; p < x->world; p++) {
        if (p == x->rank || mine == 0 || x->counts[(size_t)p] == 0) continue;
        int32_t* dr = x->d_res + res_stride * (size_t)p;
        launch_hamming_top2(my_desc, mine, reinterpret_cast<const uint4*>(x->d_gathered + (size_t)p * x->slot_bytes + 32),
                            x->counts[(size_t)p], dr, dr + x->slot_keypoints, dr + 2 * (size_t)x->slot_keypoints, s);
    }
    SO_HIP(hipGetLastError());
    uint8_t* h_res = x->h_pin + 32 * (size_t)x->world;
    if (mine > 0 && x->world > 1)
        SO_HIP(hipMemcpyAsync(h_res, x->d_res, sizeof(int32_t) * res_stride * (size_t)x->world, hipMemcpyDeviceToHost, s));
    SO_HIP(hipStreamSynchronize(s));
    for (int p = 0; p < x->world; p++) {
        if (peer_counts) peer_counts[p] = x->counts[(size_t)p];
        if (!peer_candidates) continue;
        peer_candidates[p] = 0;
        if (p == x->rank || mine == 0 || x->counts[(size_t)p] == 0) continue;
        const int32_t* hr = reinterpret_cast<const int32_t*>(h_res) + res_stride * (size_t)p;
        const int32_t* bd = hr + x->slot_keypoints;
        const int32_t* sd = hr + 2 * (size_t)x->slot_keypoints;
        int c = 0;
        for (int i = 0; i < mine; i++) c += (bd[i] <= max_dist && (float)bd[i] < ratio * (float)sd[i]) ? 1 : 0;
        peer_candidates[p] = c;
    }
    return SO_OK;
}

// Candidate-search tick, common part: this rank's n_mine records are in d_rslot (unused positions zeroed); all-gather,
// append the peers' records to the store, search every own record against the whole store.
int tick_store_common(so_exchange* x, int n_mine, const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs,
                      int32_t* n_out, const std::vector<const float*>& angles, const std::vector<const int32_t*>& mps) {
    static const Rccl no_rccl;
    const Rccl& R = x->host_gather ? no_rccl : rccl();
    so_kfstore* S = x->store;
    hipStream_t s = S->stream;
    const int K = x->records_per_tick, total = x->world * K;
    const size_t slot_bytes = x->rec_stride * (size_t)K;
    if (x->host_gather) {  // the host's transport: slot down, all-gather on the host, gathered slots up
        SO_HIP(hipMemcpyAsync(x->h_slot, x->d_rslot, slot_bytes, hipMemcpyDeviceToHost, s));
        SO_HIP(hipStreamSynchronize(s));
        const int grc = x->host_gather(x->host_user, x->h_slot, x->h_gathered, slot_bytes);
        if (grc != 0) {
            last_error_ref() = "exchange tick: the host all-gather callback failed (" + std::to_string(grc) + ")";
            return SO_ERR_HIP;
        }
        SO_HIP(hipMemcpyAsync(x->d_rgathered, x->h_gathered, slot_bytes * (size_t)x->world, hipMemcpyHostToDevice, s));
    } else {
        const ncclResult_t r = R.AllGather(x->d_rslot, x->d_rgathered, slot_bytes, kNcclUint8, x->comm, s);
        if (r != 0) return rccl_fail(r, "ncclAllGather");
    }
    so_keyframe_header* hh = reinterpret_cast<so_keyframe_header*>(x->h_rpin);
    SO_HIP(hipMemcpy2DAsync(hh, sizeof(so_keyframe_header), x->d_rgathered, x->rec_stride, sizeof(so_keyframe_header),
                            (size_t)total, hipMemcpyDeviceToHost, s));
    {
        const int wrc = wait_collective(x, s);
        if (wrc != SO_OK) return wrc;
    }
    x->hdrs.assign(hh, hh + total);
    x->skip.assign((size_t)total, 0);
    for (int j = 0; j < total; j++)
        if (j / K == x->rank || x->hdrs[(size_t)j].magic == 0) x->skip[(size_t)j] = 1;  // own records; empty positions
    int rc = kfstore_append_device(S, x->d_rgathered, x->rec_stride, x->hdrs.data(), x->skip.data(), total, nullptr, s);
    if (rc != SO_OK) return rc;
    for (int j = 0; j < n_mine; j++) {
        const so_keyframe_header& h = x->hdrs[(size_t)(x->rank * K + j)];
        rc = kfstore_search_device(S, x->d_rgathered + (size_t)(x->rank * K + j) * x->rec_stride, h, angles[(size_t)j],
                                   mps[(size_t)j], p, false, nullptr, out + (size_t)j * (size_t)p->max_candidates,
                                   pairs ? pairs + (size_t)j * (size_t)p->max_candidates * (size_t)x->slot_keypoints : nullptr,
                                   n_out + j, nullptr);
        if (rc != SO_OK) return rc;
    }
    return SO_OK;
}

// A tick is COLLECTIVE: a rank that returned before the all-gather because of something only IT can see (its frame is
// not ready, one of its records is malformed, ...) would leave every other rank waiting in theirs.  Such a rank takes
// part with zero records and reports its error afterwards; only a null / dead handle returns before the collective.
int tick_store_with_nothing(so_exchange* x, const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs, int32_t* n_out,
                            int n_clear, const char* why) {
    SO_HIP(hipSetDevice(x->device));
    SO_HIP(hipMemsetAsync(x->d_rslot, 0, x->rec_stride * (size_t)x->records_per_tick, x->store->stream));
    if (n_out)
        for (int j = 0; j < n_clear; j++) n_out[j] = 0;
    const std::vector<const float*> none_a;
    const std::vector<const int32_t*> none_m;
    so_kf_search_params dflt;
    memset(&dflt, 0, sizeof(dflt));
    const int rc = tick_store_common(x, 0, p ? p : &dflt, out, pairs, n_out, none_a, none_m);
    if (rc != SO_OK) return rc;
    last_error_ref() = std::string("exchange tick: ") + why + " (the rank took part in the collective with zero records)";
    return SO_ERR_INVALID_ARG;
}

}  // namespace

extern "C" {

int so_exchange_unique_id(uint8_t* id128) {
    if (!id128) return SO_ERR_INVALID_ARG;
    const Rccl& R = rccl();
    if (!R.ok) {
        last_error_ref() = "librccl.so.1 could not be loaded";
        return SO_ERR_NO_DEVICE;
    }
    ncclUniqueId id;
    const ncclResult_t r = R.GetUniqueId(&id);
    if (r != 0) return rccl_fail(r, "ncclGetUniqueId");
    memcpy(id128, id.internal, 128);
    return SO_OK;
}

int so_exchange_create(int device, int rank, int world, const uint8_t* id128, int slot_keypoints, so_exchange** out) {
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world || slot_keypoints < 1) return SO_ERR_INVALID_ARG;
    *out = nullptr;
    const Rccl& R = rccl();
    if (!R.ok) {
        last_error_ref() = "librccl.so.1 could not be loaded";
        return SO_ERR_NO_DEVICE;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_exchange* x = new so_exchange();
    if (const char* e = getenv("SWARMORB_COLLECTIVE_TIMEOUT_MS")) x->timeout_ms = atoi(e) > 0 ? atoi(e) : 0;
    x->device = device;
    x->rank = rank;
    x->world = world;
    x->slot_keypoints = slot_keypoints;
    x->slot_bytes = 32 * (size_t)(1 + slot_keypoints);
    x->counts.assign((size_t)world, 0);
    x->sums.assign((size_t)world, 0);
    x->h_pin_bytes = 32 * (size_t)world + sizeof(int32_t) * 3 * (size_t)slot_keypoints * (size_t)world;
    hipError_t e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&x->d_slot, x->slot_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&x->d_gathered, x->slot_bytes * (size_t)world);
    if (e == hipSuccess) e = hipMalloc((void**)&x->d_stage, 32 * (size_t)slot_keypoints);
    if (e == hipSuccess) e = hipMalloc((void**)&x->d_res, sizeof(int32_t) * 3 * (size_t)slot_keypoints * (size_t)world);
    if (e == hipSuccess) e = hipHostMalloc((void**)&x->h_pin, x->h_pin_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = so::memset_sync(x->d_gathered, 0, x->slot_bytes * (size_t)world);
    if (e != hipSuccess) {
        so_exchange_destroy(x);
        return hip_fail(e, "exchange init", __FILE__, __LINE__);
    }
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    const ncclResult_t r = R.CommInitRank(&x->comm, world, id, rank);  // collective: every rank calls it
    fflush(stdout);  // RCCL prints a version banner through stdio: do not let it trail the host's own output
    if (r != 0) {
        x->comm = nullptr;
        so_exchange_destroy(x);
        return rccl_fail(r, "ncclCommInitRank");
    }
    *out = x;
    return SO_OK;
}

void so_exchange_destroy(so_exchange* x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    if (x->dead) {
        // the stuck collective may never finish: no stream wait, the communicator is aborted (where RCCL offers it) rather
        // than destroyed, and the buffers it may still write stay allocated (hipFree would wait for the device)
        if (x->comm && rccl().CommAbort) (void)rccl().CommAbort(x->comm);
        if (x->h_slot) (void)hipHostFree(x->h_slot);
        if (x->h_gathered) (void)hipHostFree(x->h_gathered);
        delete x;
        return;
    }
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->store && x->store->stream) (void)hipStreamSynchronize(x->store->stream);
    if (x->comm) (void)rccl().CommDestroy(x->comm);
    for (void* p : {(void*)x->d_slot, (void*)x->d_gathered, (void*)x->d_stage, (void*)x->d_res, (void*)x->d_rslot,
                    (void*)x->d_rgathered, (void*)x->d_staged, (void*)x->d_stall_sink})
        if (p) (void)hipFree(p);
    if (x->h_pin) (void)hipHostFree(x->h_pin);
    if (x->h_rpin) (void)hipHostFree(x->h_rpin);
    if (x->h_slot) (void)hipHostFree(x->h_slot);
    if (x->h_gathered) (void)hipHostFree(x->h_gathered);
    if (x->store) so_kfstore_destroy(x->store);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

int so_exchange_tick_dframe(so_exchange* x, const so_dframe* f, int max_dist, float ratio, int32_t* peer_counts,
                            int32_t* peer_candidates) {
    if (!x || !x->comm) return SO_ERR_INVALID_ARG;
    if (x->dead) return dead_handle();
    SO_HIP(hipSetDevice(x->device));
    if (!f || !f->ready || f->device != x->device) {  // rank-local: take part with an empty slot, then report
        const int rc = tick_common(x, nullptr, 0, max_dist, ratio, peer_counts, peer_candidates);
        if (rc != SO_OK) return rc;
        last_error_ref() = "exchange tick: the frame is not ready or lives on another device (the rank took part with an empty slot)";
        return SO_ERR_INVALID_ARG;
    }
    return tick_common(x, f->d_desc, f->n, max_dist, ratio, peer_counts, peer_candidates);
}

int so_exchange_tick(so_exchange* x, const uint8_t* descriptors, int n, int max_dist, float ratio, int32_t* peer_counts,
                     int32_t* peer_candidates) {
    if (!x || !x->comm) return SO_ERR_INVALID_ARG;
    if (x->dead) return dead_handle();
    SO_HIP(hipSetDevice(x->device));
    if (n < 0 || (n > 0 && !descriptors)) {  // rank-local: take part with an empty slot, then report
        const int rc = tick_common(x, nullptr, 0, max_dist, ratio, peer_counts, peer_candidates);
        if (rc != SO_OK) return rc;
        last_error_ref() = "exchange tick: no descriptors (the rank took part with an empty slot)";
        return SO_ERR_INVALID_ARG;
    }
    const int nk = n < x->slot_keypoints ? n : x->slot_keypoints;
    if (nk > 0) SO_HIP(hipMemcpyAsync(x->d_stage, descriptors, 32 * (size_t)nk, hipMemcpyHostToDevice, x->stream));
    return tick_common(x, x->d_stage, nk, max_dist, ratio, peer_counts, peer_candidates);
}

static int attach_store(so_exchange* x, int device, int world, int slot_keypoints, int records_per_tick, int store_keyframes,
                        so_exchange** out);

int so_exchange_create_store(int device, int rank, int world, const uint8_t* id128, int slot_keypoints,
                             int records_per_tick, int store_keyframes, so_exchange** out) {
    if (!out || records_per_tick < 1 || records_per_tick > 64 || store_keyframes < 1) return SO_ERR_INVALID_ARG;
    int rc = so_exchange_create(device, rank, world, id128, slot_keypoints, out);
    if (rc != SO_OK) return rc;
    so_exchange* x = *out;
    *out = nullptr;
    return attach_store(x, device, world, slot_keypoints, records_per_tick, store_keyframes, out);
}

// The same exchange over the HOST's transport instead of RCCL: agents that do not share a node (the reference's robots
// talk to their server over WebSockets, code/src/WebSocket.cc) or a build without RCCL.  `allgather(user, send, recv,
// bytes)` must deliver every rank's `bytes` to every rank in rank order (recv = world x bytes) and return 0; it is
// called once per tick from the ticking thread with pinned host buffers.
int so_exchange_create_store_host(int device, int rank, int world, so_exchange_allgather_fn allgather, void* user,
                                  int slot_keypoints, int records_per_tick, int store_keyframes, so_exchange** out) {
    if (!out || !allgather || world < 1 || rank < 0 || rank >= world || slot_keypoints < 1 || records_per_tick < 1 ||
        records_per_tick > 64 || store_keyframes < 1)
        return SO_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        last_error_ref() = "no usable HIP device";
        return SO_ERR_NO_DEVICE;
    }
    SO_HIP(hipSetDevice(device));
    so_exchange* x = new so_exchange();
    if (const char* e = getenv("SWARMORB_COLLECTIVE_TIMEOUT_MS")) x->timeout_ms = atoi(e) > 0 ? atoi(e) : 0;
    x->device = device;
    x->rank = rank;
    x->world = world;
    x->slot_keypoints = slot_keypoints;
    x->host_gather = allgather;
    x->host_user = user;
    return attach_store(x, device, world, slot_keypoints, records_per_tick, store_keyframes, out);
}

static int attach_store(so_exchange* x, int device, int world, int slot_keypoints, int records_per_tick, int store_keyframes,
                        so_exchange** out) {
    int rc = so_kfstore_create(device, store_keyframes, slot_keypoints, &x->store);
    if (rc != SO_OK) {
        so_exchange_destroy(x);
        return rc;
    }
    x->records_per_tick = records_per_tick;
    x->rec_stride = x->store->dev.rec_stride;
    const size_t slot = x->rec_stride * (size_t)records_per_tick;
    const size_t staged = sizeof(so_keyframe_header) + 8 * (size_t)slot_keypoints;
    hipError_t e = hipMalloc((void**)&x->d_rslot, slot);
    if (e == hipSuccess) e = hipMalloc((void**)&x->d_rgathered, slot * (size_t)world);
    if (e == hipSuccess) e = hipMalloc((void**)&x->d_staged, staged);
    if (e == hipSuccess) e = hipHostMalloc((void**)&x->h_rpin, sizeof(so_keyframe_header) * (size_t)world * (size_t)records_per_tick + staged,
                                           hipHostMallocDefault);
    if (e == hipSuccess) e = so::memset_sync(x->d_rgathered, 0, slot * (size_t)world);
    if (e == hipSuccess && x->host_gather) e = hipHostMalloc((void**)&x->h_slot, slot, hipHostMallocDefault);
    if (e == hipSuccess && x->host_gather) e = hipHostMalloc((void**)&x->h_gathered, slot * (size_t)world, hipHostMallocDefault);
    if (e != hipSuccess) {
        so_exchange_destroy(x);
        return hip_fail(e, "exchange store init", __FILE__, __LINE__);
    }
    *out = x;
    return SO_OK;
}

so_kfstore* so_exchange_store(so_exchange* x) { return x ? x->store : nullptr; }

int so_exchange_read_record(so_exchange* x, int peer, int index, uint8_t* record, size_t capacity, size_t* length) {
    if (!x || !x->store || peer < 0 || peer >= x->world || index < 0 || index >= x->records_per_tick || !length)
        return SO_ERR_INVALID_ARG;
    *length = 0;
    const size_t j = (size_t)peer * (size_t)x->records_per_tick + (size_t)index;
    if (x->hdrs.size() <= j) return SO_OK;  // no tick yet
    const so_keyframe_header& h = x->hdrs[j];
    if (h.magic == 0) return SO_OK;
    if (h.n_keypoints < 0 || h.n_keypoints > x->slot_keypoints) return SO_ERR_INVALID_ARG;
    const size_t bytes = h.version == 2 ? so_keyframe_record_size2(h.n_keypoints) : so_keyframe_record_size(h.n_keypoints);
    *length = bytes;
    if (!record) return SO_OK;
    if (capacity < bytes) return SO_ERR_CAPACITY;
    SO_HIP(hipSetDevice(x->device));
    SO_HIP(so::memcpy_sync(record, x->d_rgathered + j * x->rec_stride, bytes, hipMemcpyDeviceToHost));
    return SO_OK;
}

int so_exchange_tick_records(so_exchange* x, const uint8_t* records, size_t stride, int32_t n_records,
                             const so_kf_search_params* p, so_kf_candidate* out, int32_t* pairs, int32_t* n_out) {
    if (!x || !x->store) return SO_ERR_INVALID_ARG;
    if (x->dead) return dead_handle();
    if (!p || n_records < 0 || n_records > x->records_per_tick || p->max_candidates < 0 || p->max_candidates > SO_KF_MAX_CANDIDATES ||
        (n_records > 0 && (!records || !out || !n_out || stride < sizeof(so_keyframe_header))))
        return tick_store_with_nothing(x, p, out, pairs, n_out, 0, "bad arguments (record count, stride, outputs or search parameters)");
    SO_HIP(hipSetDevice(x->device));
    hipStream_t s = x->store->stream;
    SO_HIP(hipMemsetAsync(x->d_rslot, 0, x->rec_stride * (size_t)x->records_per_tick, s));
    // (a malformed record is a rank-local failure too: zero records, error afterwards - tick_store_with_nothing)
    bool bad_records = false;
    size_t need = 0;
    for (int j = 0; j < n_records; j++) {
        so_keyframe_header h;
        memcpy(&h, records + (size_t)j * stride, sizeof(h));
        const bool v_ok = h.version == 1 || (h.version == 2 && (h.flags & SO_KF_FLAG_MAP_POINTS));
        if (h.magic != 0x464B4F53u || h.n_keypoints < 0 || h.n_keypoints > x->slot_keypoints || !v_ok ||
            (h.version == 2 ? so_keyframe_record_size2(h.n_keypoints) : so_keyframe_record_size(h.n_keypoints)) > stride)
            bad_records = true;
        else
            need += (size_t)h.n_keypoints;
    }
    if (bad_records)
        return tick_store_with_nothing(x, p, out, pairs, n_out, n_records, "not a keyframe record, or more keypoints than the slot holds");
    std::vector<const float*> angles((size_t)n_records);
    std::vector<const int32_t*> mps((size_t)n_records);
    x->q_angle.resize(need ? need : 1);
    x->q_mp.resize(need ? need : 1);
    size_t at = 0;
    for (int j = 0; j < n_records; j++) {
        const uint8_t* rec = records + (size_t)j * stride;
        so_keyframe_header h;
        memcpy(&h, rec, sizeof(h));
        const size_t n = (size_t)h.n_keypoints;
        const size_t bytes = h.version == 2 ? so_keyframe_record_size2(h.n_keypoints) : so_keyframe_record_size(h.n_keypoints);
        if (bytes > stride) return SO_ERR_INVALID_ARG;
        const uint8_t* geo = rec + sizeof(so_keyframe_header) + n * 32;
        for (size_t i = 0; i < n; i++) memcpy(&x->q_angle[at + i], geo + 16 * i + 8, 4);
        if (h.version == 2) {
            if (n) memcpy(&x->q_mp[at], geo + 16 * n, 4 * n);
        } else {
            for (size_t i = 0; i < n; i++) x->q_mp[at + i] = 0;
        }
        angles[(size_t)j] = x->q_angle.data() + at;
        mps[(size_t)j] = x->q_mp.data() + at;
        at += n;
        SO_HIP(hipMemcpyAsync(x->d_rslot + (size_t)j * x->rec_stride, rec, bytes, hipMemcpyHostToDevice, s));
    }
    return tick_store_common(x, n_records, p, out, pairs, n_out, angles, mps);
}

int so_exchange_tick_keyframe(so_exchange* x, const so_dframe* f, const so_keyframe_header* hdr,
                              const int32_t* map_point_id, const so_kf_search_params* p, so_kf_candidate* out,
                              int32_t* pairs, int32_t* n_out) {
    if (!x || !x->store) return SO_ERR_INVALID_ARG;
    if (x->dead) return dead_handle();
    if (!f || !f->ready || !f->mirrors || !hdr || !p || !out || !n_out || p->max_candidates < 0 || p->max_candidates > SO_KF_MAX_CANDIDATES)
        return tick_store_with_nothing(x, p, out, pairs, n_out, n_out ? 1 : 0,
                                       "the frame is not ready / not collected, or header, parameters or outputs are missing");
    if (f->device != x->device)
        return tick_store_with_nothing(x, p, out, pairs, n_out, 1, "frame and exchange live on different devices");
    const int n = f->n < x->slot_keypoints ? f->n : x->slot_keypoints;
    if (n > 0 && !map_point_id) return tick_store_with_nothing(x, p, out, pairs, n_out, 1, "map_point_id is missing");
    SO_HIP(hipSetDevice(x->device));
    hipStream_t s = x->store->stream;
    // staged block: header | angle f32 n | map_point_id i32 n  (descriptors, undistorted keypoints, octaves stay in HBM)
    uint8_t* hs = x->h_rpin + sizeof(so_keyframe_header) * (size_t)x->world * (size_t)x->records_per_tick;
    so_keyframe_header h = *hdr;
    h.n_keypoints = n;
    memcpy(hs, &h, sizeof(h));
    if (n > 0) {
        memcpy(hs + sizeof(h), f->angle.data(), 4 * (size_t)n);
        memcpy(hs + sizeof(h) + 4 * (size_t)n, map_point_id, 4 * (size_t)n);
    }
    const size_t staged = sizeof(h) + 8 * (size_t)n;
    SO_HIP(hipMemsetAsync(x->d_rslot, 0, x->rec_stride * (size_t)x->records_per_tick, s));
    SO_HIP(hipMemcpyAsync(x->d_staged, hs, staged, hipMemcpyHostToDevice, s));
    launch_kf_pack_record(f->d_desc, f->d_xy_un, f->d_octave, x->d_staged, n, x->d_rslot, so_keyframe_record_size2(n), s);
    SO_HIP(hipGetLastError());
    x->q_angle.assign(f->angle.begin(), f->angle.begin() + n);
    x->q_mp.assign(map_point_id, map_point_id + n);
    if (n == 0) {
        x->q_angle.resize(1);
        x->q_mp.resize(1);
    }
    std::vector<const float*> angles{x->q_angle.data()};
    std::vector<const int32_t*> mps{x->q_mp.data()};
    return tick_store_common(x, 1, p, out, pairs, n_out, angles, mps);
}

// Budget of the wait for a tick's collective in ms (0 = wait for ever); the default comes from
// SWARMORB_COLLECTIVE_TIMEOUT_MS (5000).
int so_exchange_set_timeout(so_exchange* x, int milliseconds) {
    if (!x || milliseconds < 0) return SO_ERR_INVALID_ARG;
    x->timeout_ms = milliseconds;
    return SO_OK;
}
int so_exchange_is_dead(const so_exchange* x) { return x && x->dead ? 1 : 0; }
// Test hook: one workgroup spins for `milliseconds` on the stream the next tick's collective is enqueued on - what a
// peer that enters the tick late looks like from this rank (the spin ends by itself; nothing hangs).
int so_exchange_debug_stall(so_exchange* x, int milliseconds) {
    if (!x || milliseconds < 0 || milliseconds > 10000) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(x->device));
    if (!x->d_stall_sink) SO_HIP(hipMalloc((void**)&x->d_stall_sink, 256));
    hipStream_t s = x->store ? x->store->stream : x->stream;
    if (milliseconds == 0) return SO_OK;
    return launch_occupy(1, 256, milliseconds, x->d_stall_sink, s) ? SO_OK : SO_ERR_INVALID_ARG;
}

// Rank `peer`'s slot as gathered by the last tick (tests, host merger): descriptors and the header's checksum.
int so_exchange_read_slot(so_exchange* x, int peer, uint8_t* descriptors, int capacity, int* n_out, uint64_t* checksum) {
    if (!x || !x->comm || peer < 0 || peer >= x->world || !n_out) return SO_ERR_INVALID_ARG;
    const int n = x->counts[(size_t)peer];
    *n_out = n;
    if (checksum) *checksum = x->sums[(size_t)peer];
    if (!descriptors) return SO_OK;
    if (capacity < n) return SO_ERR_CAPACITY;
    SO_HIP(hipSetDevice(x->device));
    if (n > 0)
        SO_HIP(so::memcpy_sync(descriptors, x->d_gathered + (size_t)peer * x->slot_bytes + 32, 32 * (size_t)n, hipMemcpyDeviceToHost));
    return SO_OK;
}

}  // extern "C"
