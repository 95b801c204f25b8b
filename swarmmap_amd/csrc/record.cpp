// record.cpp — compact binary keyframe record.  The reference ships keyframes between agents and
// the server as Boost *text* archives of whole map slices (code/src/MapUpdater.cc:190-230, KeyFrame::serialize,
// code/include/KeyFrame.h:310-406); the cross-agent descriptor all-gather needs only what the loop / merge
// candidate search reads, in a fixed layout a GPU kernel can index: a 128-byte header, then the descriptors
// (n x 32 B, the block the Hamming kernels consume in place), then the keypoint geometry (n x 16 B).
// Everything is little-endian; the record length is a multiple of 32 bytes (one descriptor row).
// Version 2 appends one i32 per keypoint (the id of the map point bound to it, -1 = none) and pads to 32 bytes: the
// candidate search only matches keypoints that carry a map point (code/src/ORBmatcher.cc:517-521,535-541).
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
    h.flags = 0;
    h.n_map_points = 0;
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

size_t so_keyframe_record_size2(int32_t n_keypoints) {
    return n_keypoints < 0 ? 0 : (sizeof(so_keyframe_header) + (size_t)n_keypoints * 52 + 31) / 32 * 32;
}

int so_keyframe_record_pack2(const so_keyframe_header* hdr, const float* xy, const float* angle, const int32_t* octave,
                             const uint8_t* descriptors, const int32_t* map_point_id, uint8_t* out, size_t capacity) {
    if (!hdr || !out || hdr->n_keypoints < 0) return SO_ERR_INVALID_ARG;
    const size_t n = (size_t)hdr->n_keypoints;
    if (n > 0 && !map_point_id) return SO_ERR_INVALID_ARG;
    const size_t total = so_keyframe_record_size2(hdr->n_keypoints);
    if (capacity < total) return SO_ERR_CAPACITY;
    const int rc = so_keyframe_record_pack(hdr, xy, angle, octave, descriptors, out, capacity);
    if (rc != SO_OK) return rc;
    uint8_t* mp = out + sizeof(so_keyframe_header) + n * 48;
    if (n > 0) memcpy(mp, map_point_id, n * 4);
    memset(mp + n * 4, 0, total - (sizeof(so_keyframe_header) + n * 52));
    int32_t bound = 0;
    for (size_t i = 0; i < n; i++) bound += map_point_id[i] >= 0 ? 1 : 0;
    so_keyframe_header h;
    memcpy(&h, out, sizeof(h));
    h.version = 2;
    h.flags = SO_KF_FLAG_MAP_POINTS;
    h.n_map_points = bound;
    h.checksum = checksum(out + sizeof(so_keyframe_header), n * 52);
    memcpy(out, &h, sizeof(h));
    return SO_OK;
}

int so_keyframe_record_unpack2(const uint8_t* rec, size_t length, so_keyframe_header* hdr, float* xy, float* angle,
                               int32_t* octave, uint8_t* descriptors, int32_t* map_point_id, int32_t capacity) {
    if (!rec || !hdr || length < sizeof(so_keyframe_header)) return SO_ERR_INVALID_ARG;
    so_keyframe_header h;
    memcpy(&h, rec, sizeof(h));
    if (h.magic == kMagic && h.version == kVersion) {  // version 1: every keypoint counts as bound
        const int rc = so_keyframe_record_unpack(rec, length, hdr, xy, angle, octave, descriptors, capacity);
        if (rc == SO_OK && map_point_id)
            for (int32_t i = 0; i < hdr->n_keypoints; i++) map_point_id[i] = 0;
        return rc;
    }
    if (h.magic != kMagic || h.version != 2 || h.header_bytes != sizeof(so_keyframe_header) || h.n_keypoints < 0 ||
        !(h.flags & SO_KF_FLAG_MAP_POINTS)) {
        so::last_error_ref() = "not a keyframe record (magic / version)";
        return SO_ERR_INVALID_ARG;
    }
    const size_t n = (size_t)h.n_keypoints;
    if (length < so_keyframe_record_size2(h.n_keypoints)) return SO_ERR_INVALID_ARG;
    const uint8_t* desc = rec + sizeof(so_keyframe_header);
    const uint8_t* geo = desc + n * 32;
    const uint8_t* mp = geo + n * 16;
    if (checksum(desc, n * 52) != h.checksum) {
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
    if (map_point_id && n > 0) memcpy(map_point_id, mp, n * 4);
    return SO_OK;
}

}  // extern "C"
