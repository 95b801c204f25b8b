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

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "dframe_internal.h"
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
};

namespace {

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
    SO_HIP(hipStreamSynchronize(s));
    for (int p = 0; p < x->world; p++) {
        int32_t hdr[2];
        memcpy(hdr, x->h_pin + 32 * (size_t)p, 8);
        memcpy(&x->sums[(size_t)p], x->h_pin + 32 * (size_t)p + 8, 8);
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
    if (e == hipSuccess) e = hipMemset(x->d_gathered, 0, x->slot_bytes * (size_t)world);
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
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->comm) (void)rccl().CommDestroy(x->comm);
    for (void* p : {(void*)x->d_slot, (void*)x->d_gathered, (void*)x->d_stage, (void*)x->d_res})
        if (p) (void)hipFree(p);
    if (x->h_pin) (void)hipHostFree(x->h_pin);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
}

int so_exchange_tick_dframe(so_exchange* x, const so_dframe* f, int max_dist, float ratio, int32_t* peer_counts,
                            int32_t* peer_candidates) {
    if (!x || !f || !f->ready) return SO_ERR_INVALID_ARG;
    if (f->device != x->device) {
        last_error_ref() = "frame and exchange live on different devices";
        return SO_ERR_INVALID_ARG;
    }
    SO_HIP(hipSetDevice(x->device));
    return tick_common(x, f->d_desc, f->n, max_dist, ratio, peer_counts, peer_candidates);
}

int so_exchange_tick(so_exchange* x, const uint8_t* descriptors, int n, int max_dist, float ratio, int32_t* peer_counts,
                     int32_t* peer_candidates) {
    if (!x || n < 0 || (n > 0 && !descriptors)) return SO_ERR_INVALID_ARG;
    SO_HIP(hipSetDevice(x->device));
    const int nk = n < x->slot_keypoints ? n : x->slot_keypoints;
    if (nk > 0) SO_HIP(hipMemcpyAsync(x->d_stage, descriptors, 32 * (size_t)nk, hipMemcpyHostToDevice, x->stream));
    return tick_common(x, x->d_stage, nk, max_dist, ratio, peer_counts, peer_candidates);
}

// Rank `peer`'s slot as gathered by the last tick (tests, host merger): descriptors and the header's checksum.
int so_exchange_read_slot(so_exchange* x, int peer, uint8_t* descriptors, int capacity, int* n_out, uint64_t* checksum) {
    if (!x || peer < 0 || peer >= x->world || !n_out) return SO_ERR_INVALID_ARG;
    const int n = x->counts[(size_t)peer];
    *n_out = n;
    if (checksum) *checksum = x->sums[(size_t)peer];
    if (!descriptors) return SO_OK;
    if (capacity < n) return SO_ERR_CAPACITY;
    SO_HIP(hipSetDevice(x->device));
    if (n > 0)
        SO_HIP(hipMemcpy(descriptors, x->d_gathered + (size_t)peer * x->slot_bytes + 32, 32 * (size_t)n, hipMemcpyDeviceToHost));
    return SO_OK;
}

}  // extern "C"
