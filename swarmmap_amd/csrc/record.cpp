// record.cpp — compact binary keyframe record.  The reference ships keyframes between agents and
// the server as Boost *text* archives of whole map slices (code/src/MapUpdater.cc:190-230, KeyFrame::serialize,
// code/include/KeyFrame.h:310-406); the cross-agent descriptor all-gather needs only what the loop / merge
// candidate search reads, in a fixed layout a GPU kernel can index: a 128-byte header, then the descriptors
// (n x 32 B, the block the Hamming kernels consume in place), then the keypoint geometry (n x 16 B).
// Everything is little-endian; the record length is a multiple of 32 bytes (one descriptor row).
#include <cstring>

#include "so_common.h"

namespace {

constexpr uint32_t kMagic = 0x464B4F53u;  // "SOKF"
constexpr uint16_t kVersion = 1;

// order-sensitive 61-bit checksum of a byte range (same function as swarmmap_amd.parallel.slot_checksum)
uint64_t checksum(const uint8_t* p, size_t n) {
    const uint64_t mod = (1ull << 61) - 1;
    uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) acc = (acc + (uint64_t)p[i] * ((i % 65521u) + 1u)) % mod;
    return acc;
}

}  // namespace

extern "C" {

size_t so_keyframe_record_size(int32_t n_keypoints) {
    return n_keypoints < 0 ? 0 : sizeof(so_keyframe_header) + (size_t)n_keypoints * 48;
}

int so_keyframe_record_pack(const so_keyframe_header* hdr, const float* xy, const float* angle, const int32_t* octave,
                            const uint8_t* descriptors, uint8_t* out, size_t capacity) {
    if (!hdr || !out || hdr->n_keypoints < 0) return SO_ERR_INVALID_ARG;
    const size_t n = (size_t)hdr->n_keypoints;
    if (n > 0 && (!xy || !angle || !octave || !descriptors)) return SO_ERR_INVALID_ARG;
    if (capacity < so_keyframe_record_size(hdr->n_keypoints)) return SO_ERR_CAPACITY;
    uint8_t* desc = out + sizeof(so_keyframe_header);
    uint8_t* geo = desc + n * 32;
    if (n > 0) memcpy(desc, descriptors, n * 32);
    for (size_t i = 0; i < n; i++) {
        memcpy(geo + 16 * i, xy + 2 * i, 8);
        memcpy(geo + 16 * i + 8, angle + i, 4);
        memcpy(geo + 16 * i + 12, octave + i, 4);
    }
    so_keyframe_header h = *hdr;
    h.magic = kMagic;
    h.version = kVersion;
    h.header_bytes = (uint16_t)sizeof(so_keyframe_header);
    h.checksum = checksum(desc, n * 48);
    memcpy(out, &h, sizeof(h));
    return SO_OK;
}

int so_keyframe_record_unpack(const uint8_t* rec, size_t length, so_keyframe_header* hdr, float* xy, float* angle,
                              int32_t* octave, uint8_t* descriptors, int32_t capacity) {
    if (!rec || !hdr || length < sizeof(so_keyframe_header)) return SO_ERR_INVALID_ARG;
    so_keyframe_header h;
    memcpy(&h, rec, sizeof(h));
    if (h.magic != kMagic || h.version != kVersion || h.header_bytes != sizeof(so_keyframe_header) || h.n_keypoints < 0) {
        so::last_error_ref() = "not a keyframe record (magic / version)";
        return SO_ERR_INVALID_ARG;
    }
    const size_t n = (size_t)h.n_keypoints;
    if (length < so_keyframe_record_size(h.n_keypoints)) return SO_ERR_INVALID_ARG;
    const uint8_t* desc = rec + sizeof(so_keyframe_header);
    const uint8_t* geo = desc + n * 32;
    if (checksum(desc, n * 48) != h.checksum) {
        so::last_error_ref() = "keyframe record checksum mismatch";
        return SO_ERR_INVALID_ARG;
    }
    *hdr = h;
    if (capacity < h.n_keypoints) return SO_ERR_CAPACITY;
    if (descriptors && n > 0) memcpy(descriptors, desc, n * 32);
    for (size_t i = 0; i < n; i++) {
        if (xy) memcpy(xy + 2 * i, geo + 16 * i, 8);
        if (angle) memcpy(angle + i, geo + 16 * i + 8, 4);
        if (octave) memcpy(octave + i, geo + 16 * i + 12, 4);
    }
    return SO_OK;
}

}  // extern "C"
