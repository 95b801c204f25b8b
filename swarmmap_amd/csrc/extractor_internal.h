// extractor_internal.h — what the device-resident frame (dframe.cpp) needs from an extractor context besides the
// public C ABI: the stream its frame runs on and the HBM-resident copies of the frame's outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"

struct so_extractor;

namespace so {

struct ExtractorDeviceView {
    hipStream_t stream;
    int device;
    int capacity;            // keypoint capacity of the output arrays
    int nlevels;
    float scale[kMaxLevels];
    // valid once the first frame has sized the context; on the host-quadtree path these are the device views of
    // the host-mapped result buffers (same layout), so consumers need no second code path
    const SelectedKp* meta;  // (x, y, level, score) in level coordinates, output order
    const float* angle;
    const uint8_t* desc;
    const int32_t* total;
};

int extractor_device_view(so_extractor* ex, ExtractorDeviceView* out);

}  // namespace so
