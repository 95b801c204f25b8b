// Drives swarmmap_amd/host/AgentMediator.{h,cc} (plain g++ over the C ABI) for tests/test_cpp_adapters_gpu.py: reads the
// keyframes the test wrote, stores all but the last `n_query`, looks those up and prints what came back.
//   mediator_smoke <dir> <n_keyframes> <n_query> <max_keypoints> <min_votes> <min_matches>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../../swarmmap_amd/host/AgentMediator.h"

template <class T>
static std::vector<T> load(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    const size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)bytes);
    return v;
}

struct Kf {
    std::vector<int32_t> meta, octave, mp;  // meta: agent, keyframe id
    std::vector<float> xy, angle;
    std::vector<uint8_t> desc;
    ORB_SLAM2::KeyFrameView view() const {
        ORB_SLAM2::KeyFrameView v;
        v.mnClientId = meta[0];
        v.mnId = (unsigned long)meta[1];
        v.N = (int)angle.size();
        v.xy = xy.data(); v.angle = angle.data(); v.octave = octave.data(); v.descriptors = desc.data(); v.mapPointId = mp.data();
        return v;
    }
};

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const std::string dir = argv[1];
    const int n = atoi(argv[2]), nq = atoi(argv[3]), kp = atoi(argv[4]), min_votes = atoi(argv[5]), min_matches = atoi(argv[6]);
    std::vector<Kf> kfs((size_t)n);
    for (int k = 0; k < n; k++) {
        const std::string p = dir + "/kf" + std::to_string(k) + "_";
        kfs[(size_t)k].meta = load<int32_t>(p + "meta.bin");
        kfs[(size_t)k].xy = load<float>(p + "xy.bin");
        kfs[(size_t)k].angle = load<float>(p + "angle.bin");
        kfs[(size_t)k].octave = load<int32_t>(p + "octave.bin");
        kfs[(size_t)k].desc = load<uint8_t>(p + "desc.bin");
        kfs[(size_t)k].mp = load<int32_t>(p + "mp.bin");
    }
    try {
        ORB_SLAM2::AgentMediator med(64, kp);
        for (int k = 0; k < n - nq; k++) med.AddKeyFrame(kfs[(size_t)k].view());
        printf("store %d\n", med.KeyFramesInStore());
        for (int k = n - nq; k < n; k++) {
            const std::vector<ORB_SLAM2::OverlapCandidate> c = med.CheckOverlapCandidates(kfs[(size_t)k].view(), 0.75f, true, min_votes, min_matches, 8);
            printf("query %d candidates %zu\n", k, c.size());
            for (const auto& o : c) {
                unsigned long long h = 1469598103934665603ull;  // FNV-1a of vpMatches12
                for (int v : o.vpMatches12)
                    for (int b = 0; b < 4; b++) h = (h ^ (unsigned char)((unsigned)v >> (8 * b))) * 1099511628211ull;
                printf("cand %d %lu %d %d %d %016llx\n", o.mnClientId, o.mnId, o.slot, o.votes, o.nmatches, h);
            }
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
