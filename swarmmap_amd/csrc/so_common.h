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

}  // namespace so

#define SO_HIP(call)                                                          \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return so::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)
