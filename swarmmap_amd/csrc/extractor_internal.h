// extractor_internal.h — what the device-resident frame (dframe.cpp) needs from an extractor context besides the
// public C ABI: the stream its frame runs on and the HBM-resident copies of the frame's outputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_device.h"

struct so_extractor;
struct so_extractor_group;

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

// A consumer's launch captured behind describe in the frame's hipGraph (the device-resident Frame's prepare kernel: a
// separate launch behind a graph starts ~8 us after the graph's last kernel ends).  `fn(owner, stream)` must enqueue
// the same work with the same arguments for a given (owner, revision); the extractor keeps one captured graph per
// owner, re-captured when the revision moves.  The tail set here applies to the NEXT submit only.
typedef void (*ExtractorTailFn)(void* ctx, hipStream_t s);
void extractor_set_graph_tail(so_extractor* ex, void* owner, uint64_t revision, ExtractorTailFn fn);
void extractor_release_graph_tail(so_extractor* ex, void* owner);  // the owner goes away: its graph is dropped
// true when the frame submitted last ran the tail inside its graph (false: chained launches, profiling, no graph)
bool extractor_tail_launched(const so_extractor* ex);

// so_extractor_group internals used by the device-resident frames' group submit (dframe.cpp)
struct FramePrepareArgs;
int extractor_group_size(const so_extractor_group* g);
so_extractor* extractor_group_member(const so_extractor_group* g, int i);
int extractor_group_prepare(so_extractor_group* g, int w, int h);
int extractor_group_submit(so_extractor_group* g, const uint8_t* const* images, int w, int h, int stride, const FramePrepareArgs* preps);

}  // namespace so
