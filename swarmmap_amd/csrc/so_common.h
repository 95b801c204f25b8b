// so_common.h — host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "../../include/swarmorb.h"

namespace so {

std::string& last_error_ref();  // thread-local (capi.cpp)

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    last_error_ref() = buf;
    return SO_ERR_HIP;
}

// The runtime multiplexes streams onto a few hardware queues, and a frame's match kernel that shares a queue
// with the local-mapping solver waits behind ~100 queued bundle-adjustment launches (half a millisecond).  The
// contexts of the per-frame tracking path are called one after the other by one thread, so per (creating thread,
// device) all extractors share one stream (role 0) and all matchers another (role 1) instead of holding one
// each; the solver keeps its own.  The shared streams live as long as the process.
hipError_t tracking_stream(int device, int role, hipStream_t* s);  // capi.cpp
// The stream a new extractor (role 0) / matcher (role 1) runs on: the calling thread's shared one, or - after
// so_runtime_private_streams(1), for a thread that drives several agents - a stream of its own (*owned: the handle
// destroys it).
hipError_t context_stream(int device, int role, hipStream_t* s, bool* owned);  // capi.cpp
// Blocking fills / copies WITHOUT the legacy default stream: hipMemset / hipMemcpy run on the null stream, which
// synchronises with every other stream - and fails outright while another thread of the process (another agent) is
// capturing its frame into a hipGraph ("operation would make the legacy stream depend on a capturing blocking stream").
// These run on a non-blocking utility stream of the calling thread and wait for it.
hipError_t memset_sync(void* dst, int value, size_t bytes);                          // capi.cpp
hipError_t memcpy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);  // capi.cpp

}  // namespace so

#define SO_HIP(call)                                                          \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return so::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)
