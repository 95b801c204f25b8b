// replay_sanitize.cc — TEST INFRASTRUCTURE: drives the replay harness (swarmmap_amd/host/replay.cc + closedloop.cc: the tracking
// thread, the local-mapping thread, the hand-over between them, the map model, the window gather, the packets, so_fleet_run) over
// tests/cpp/mock_swarmorb.cc, for the sanitizer builds of `make -C swarmmap_amd/csrc host-asan host-tsan` (CPU only).
//   replay_sanitize <mode> [frames]     mode: solo | policy | threads | fleet
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "swarmorb.h"

struct so_replay;
extern "C" {
int so_replay_create(int device, int width, int height, int nfeatures, int lba_every, const float* K4, const float* dist5, int keyframe_every,
                     float keyframe_ratio, float plane_z, int local_keyframes, int third_pose, so_replay** out);
void so_replay_destroy(so_replay* r);
const char* so_replay_error(so_replay* r);
int so_replay_set_frames(so_replay* r, const uint64_t* pointers, int n, int on_device);
int so_replay_set_vocabulary(so_replay* r, const uint8_t* centroids, int n, int neighbours);
int so_replay_set_closed_loop(so_replay* r, int kf_every, int delay, int n_free, int n_fixed, int policy);
int so_replay_set_track_chain(so_replay* r, int on);
int so_replay_prime(so_replay* r, int t);
int so_replay_run(so_replay* r, int first_t, int n_steps, int timed);
int so_fleet_run(so_replay** agents, int n_agents, int first_t, int n_steps, int timed);
int so_replay_drain(so_replay* r);
int so_replay_fleet_ticks(so_replay* r, long long* ticks, long long* slots);
int so_replay_set_fleet_offset(so_replay* r, int offset);
int so_replay_finish(so_replay* r);
int so_replay_cl_counts(so_replay* r, int64_t* counts8, double* wait_ms);
int so_replay_log_size(so_replay* r);
}

namespace {
constexpr int W = 752, H = 480;
const float K4[4] = {458.654f, 457.296f, 367.215f, 248.375f};
const float D5[5] = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f, 0.f};

struct Agent {
    so_replay* r = nullptr;
    std::vector<uint8_t> image;
    std::vector<uint64_t> ptrs;
    std::vector<uint8_t> vocab;
};

int make_agent(Agent& a, int frames, int policy, int chain) {
    a.image.assign((size_t)W * H, 127);
    a.ptrs.assign((size_t)frames + 2, (uint64_t)(uintptr_t)a.image.data());
    a.vocab.resize(100 * 32);
    for (size_t i = 0; i < a.vocab.size(); i++) a.vocab[i] = (uint8_t)(i * 131u + 7u);
    if (so_replay_create(0, W, H, 1000, 5, K4, D5, 8, 0.7f, 2.0f, 12, 1, &a.r) != SO_OK) return 1;
    if (so_replay_set_frames(a.r, a.ptrs.data(), (int)a.ptrs.size(), 0) != SO_OK) return 2;
    if (so_replay_set_track_chain(a.r, chain) != SO_OK) return 3;
    if (so_replay_set_vocabulary(a.r, a.vocab.data(), 100, 20) != SO_OK) return 4;
    if (so_replay_set_closed_loop(a.r, 5, 5, 25, 40, policy) != SO_OK) return 5;
    if (so_replay_prime(a.r, 0) != SO_OK) return 6;
    return 0;
}

int finish_agent(Agent& a, const char* what, int frames, int min_jobs = -1) {
    int rc = so_replay_drain(a.r);
    if (rc == SO_OK) rc = so_replay_finish(a.r);
    int64_t c[8] = {0};
    double waited = 0.0;
    so_replay_cl_counts(a.r, c, &waited);
    const int tracked = so_replay_log_size(a.r);
    printf("%s: rc %d, %d frames tracked, jobs %lld windows %lld map slots %lld bad %lld keyframes %lld\n", what, rc, tracked, (long long)c[0], (long long)c[1],
           (long long)c[4], (long long)c[5], (long long)c[6]);
    if (rc != SO_OK) printf("  error: %s\n", so_replay_error(a.r));
    if (min_jobs < 0) min_jobs = frames / 5 - 1;
    const bool ok = rc == SO_OK && tracked == frames && c[0] >= min_jobs && c[1] >= 1 && c[4] > 900;
    so_replay_destroy(a.r);
    a.r = nullptr;
    return ok ? 0 : 1;
}
}  // namespace

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "solo";
    const int frames = argc > 2 ? atoi(argv[2]) : 120;
    int fails = 0;
    if (!strcmp(mode, "solo") || !strcmp(mode, "policy")) {
        // one agent: the caller is the tracking thread, the handle owns the local-mapping thread; both stage paths
        for (int chain = 0; chain < 2; chain++) {
            Agent a;
            if (int e = make_agent(a, frames, !strcmp(mode, "policy") ? 1 : 0, chain)) return 10 + e;
            const int rc = so_replay_run(a.r, 0, frames, 1);
            if (rc != SO_OK) printf("so_replay_run: %d %s\n", rc, so_replay_error(a.r));
            // (the reference's policy makes a keyframe only while local mapping is idle: fewer jobs, timing-dependent)
            fails += finish_agent(a, chain ? "one agent, chained stages" : "one agent, separate calls", frames, !strcmp(mode, "policy") ? 4 : -1);
        }
    } else if (!strcmp(mode, "threads")) {
        // three agents as thread pairs of one process (bench.py --agents-per-gpu)
        std::vector<Agent> ag(3);
        std::vector<std::thread> th;
        std::vector<int> res(ag.size(), 0);
        for (size_t i = 0; i < ag.size(); i++)
            th.emplace_back([&, i] {
                if (make_agent(ag[i], frames, 0, 1)) { res[i] = 1; return; }
                if (so_replay_run(ag[i].r, 0, frames, 1) != SO_OK) res[i] = 1;
                res[i] += finish_agent(ag[i], "agent in a thread", frames);
            });
        for (auto& t : th) t.join();
        for (int v : res) fails += v;
    } else if (!strcmp(mode, "fleet")) {
        // four agents in lockstep on this thread, a local-mapping thread each (bench.py --lockstep)
        std::vector<Agent> ag(4);
        std::vector<so_replay*> hs;
        for (auto& a : ag) {
            if (int e = make_agent(a, frames, 0, 1)) return 20 + e;
            hs.push_back(a.r);
        }
        // (MOCK_BA_SLEEP_US makes the local-mapping jobs longer than five ticks: the elastic ticks leave agents out and take them
        //  back in; two of the agents are a frame / two frames ahead of the fleet's clock, as bench.py's staggered fleets are)
        for (size_t i = 1; i < ag.size(); i += 2) {
            const int off = (int)i / 2 + 1;
            if (so_replay_run(ag[i].r, 0, off, 1) != SO_OK || so_replay_set_fleet_offset(ag[i].r, off) != SO_OK) return 30;
        }
        const int m = frames - 4;  // frames per agent inside the fleet (the run-ahead agents track theirs from their offset on)
        int rc = so_fleet_run(hs.data(), (int)hs.size(), 0, m / 2, 1);
        if (rc == SO_OK) rc = so_fleet_run(hs.data(), (int)hs.size(), m / 2, m - m / 2, 1);
        if (rc != SO_OK) printf("so_fleet_run: %d %s\n", rc, so_replay_error(hs[0]));
        long long ticks = 0, places = 0;
        so_replay_fleet_ticks(hs[0], &ticks, &places);
        printf("fleet: %lld ticks, %lld agent places (%d agents x %d frames)\n", ticks, places, (int)ag.size(), m);
        if (places != (long long)ag.size() * m) fails++;
        if (getenv("MOCK_BA_SLEEP_US") && ticks <= m) {
            printf("the elastic path was not exercised: every tick took every agent\n");
            fails++;
        }
        for (size_t i = 0; i < ag.size(); i++) {  // the agents' last frames, alone again: `frames` each in all
            const int off = (i & 1) ? (int)i / 2 + 1 : 0;
            if (off + m < frames && so_replay_run(ag[i].r, off + m, frames - off - m, 1) != SO_OK) fails++;
        }
        for (size_t i = ag.size(); i-- > 0;) fails += finish_agent(ag[i], "agent of the fleet", frames);
    } else {
        fprintf(stderr, "unknown mode %s\n", mode);
        return 2;
    }
    printf("%s: %s\n", mode, fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
