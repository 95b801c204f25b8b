"""Several agents in one process (one thread pair each, as SwarmMap runs its clients): contexts are per thread,
results must not depend on what the other threads are doing."""
import threading

import numpy as np
import pytest

from swarmmap_amd import synth

pytestmark = pytest.mark.gpu


def _agent_work(S, seed, rounds, out):
    from swarmmap_amd.matcher import FrameView
    ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
    m = S.ORBmatcher(0.9, True)
    opt = S.Optimizer()
    stream = synth.FrameStream(seed=seed)
    fr, last = synth.make_m2_case(seed, 800, 800)
    F = FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"], fr["excluded"])
    c = synth.make_pose_case(seed, 300)
    win = synth.make_ba_problem(seed, 5, 4, 250, max_obs="auto")
    res = []
    ex.submit(stream.frame(0))
    for t in range(rounds):
        kps, desc = ex.collect()
        kps, desc = kps.copy(), desc.copy()
        ex.submit(stream.frame(t + 1))
        nm, k2l = m.SearchByProjectionLastFrame(F, last, 15.0)
        ni, T, outl, _ = opt.PoseOptimization(c["Tcw"], c["intr"], c["Xw"], c["obs"], c["inv_sigma2"])
        item = [kps.tobytes(), desc.tobytes(), nm, k2l.tobytes(), ni, T.tobytes(), outl.tobytes()]
        if t % 4 == 0:
            r = opt.LocalBundleAdjustment(win)
            item += [r["Tcw"].tobytes(), r["Xw"].tobytes()]
        res.append(item)
    ex.collect()
    for o in (ex, m, opt):
        o.close()
    out[seed] = res


def test_four_agents_in_threads_match_sequential_runs():
    import swarmmap_amd as S
    assert S.device_count() > 0, "these tests need a GPU"
    seeds, rounds = [11, 12, 13, 14], 12
    ref = {}
    for s in seeds:
        _agent_work(S, s, rounds, ref)
    par, errs = {}, []

    def run(s):
        try:
            _agent_work(S, s, rounds, par)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=run, args=(s,)) for s in seeds]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    for s in seeds:
        assert par[s] == ref[s], "agent %d: results depend on concurrency" % s


def test_agents_started_at_the_same_moment_on_streams_of_their_own():
    """Eight agents created and run by eight threads at the same time, every handle on a stream of its own
    (so_runtime_private_streams): each allocates its device-resident frame and captures its frame graph while the others
    do the same.  (A fill on the legacy default stream used to fail here: "operation would make the legacy stream depend
    on a capturing blocking stream".)  Every agent's keypoints and descriptors equal a solo run's."""
    import swarmmap_amd as S
    from swarmmap_amd.replay import private_streams
    img = [synth.make_canvas(20 + k, 752, 480) for k in range(3)]

    def work(out, k):
        ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
        f = S.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST)
        res = []
        for t in range(6):
            kps, un, d = f(img[(k + t) % 3])
            res.append((kps.tobytes(), un.tobytes(), d.tobytes()))
        f.close(); ex.close()
        out[k] = res

    solo = {}
    work(solo, 0)
    private_streams(True)
    try:
        par, errs = {}, []
        gate = threading.Barrier(8)

        def run(k):
            try:
                gate.wait(timeout=60)
                work(par, k)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=run, args=(k,)) for k in range(8)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    finally:
        private_streams(False)
    assert not errs, errs
    # every (agent, frame) equals the solo result for the same image
    by_img = {}
    for t in range(6):
        by_img[t % 3] = solo[0][t]
    for k in range(8):
        for t in range(6):
            assert par[k][t] == by_img[(k + t) % 3], (k, t)


def test_tracking_searches_while_another_thread_appends_and_the_map_table_moves():
    """The closed loop's thread pattern: the tracking thread searches the device-resident map (so_track_search_local_map over a
    slot list) while the local-mapping thread appends new map points to the same so_map - far enough for the table to be
    reallocated twice (65536 -> 131072 -> 262144 rows).  The searches hold the table in place while their kernels run
    (so_map's shared lock, taken at submit and released at the end of the wait): every result equals the quiet run's."""
    import swarmmap_amd as S
    from swarmmap_amd import dframe as dfm
    ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST)
    img = synth.make_canvas(5, 752, 480)
    kps, un, d = [a.copy() for a in f(img)]
    rng = np.random.default_rng(3)
    z = rng.uniform(2.0, 8.0, len(kps))
    fx, fy, cx, cy = [float(v) for v in synth.EUROC_K]
    Xw = np.stack([(un[:, 0] - cx) / fx * z, (un[:, 1] - cy) / fy * z, z], 1).astype(np.float32)
    normal = (Xw / np.linalg.norm(Xw, axis=1, keepdims=True)).astype(np.float32)
    dist = np.linalg.norm(Xw, axis=1).astype(np.float32)
    sfac = ex.GetScaleFactors()
    mx, mn = (1.2 * dist * sfac[kps["octave"]]).astype(np.float32), (0.5 * dist / sfac[7]).astype(np.float32)
    dmap = S.DeviceMap()
    dmap.append(Xw, normal, mx, mn, d)
    n0 = len(Xw)
    Tcw = np.array([1, 0, 0, 0.002, 0, 1, 0, -0.001, 0, 0, 1, 0], np.float32)
    log_sf = float(np.log(np.float32(1.2)))
    m = S.ORBmatcher(0.8, True)
    slots = np.arange(n0, dtype=np.int32)[::-1].copy()  # an explicit slot list, as the closed loop passes it
    want = dfm.search_local_map(m, f, dmap, Tcw, n0, 1.0, 0.5, log_sf, local_slot=slots)
    assert want[0] > 300
    stop, errs, n_search = threading.Event(), [], [0]

    def tracker():
        try:
            while not stop.is_set():
                got = dfm.search_local_map(m, f, dmap, Tcw, n0, 1.0, 0.5, log_sf, local_slot=slots)
                assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
                n_search[0] += 1
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=tracker)
    th.start()
    chunk = 4096
    junk = [rng.normal(0, 1, (chunk, 3)).astype(np.float32), rng.normal(0, 1, (chunk, 3)).astype(np.float32),
            np.ones(chunk, np.float32), np.ones(chunk, np.float32), rng.integers(0, 256, (chunk, 32)).astype(np.uint8)]
    try:
        while len(dmap) < 140000:  # two reallocations of the table
            dmap.append(*junk)
    finally:
        stop.set()
        th.join()
    assert not errs, errs
    assert n_search[0] >= 5 and len(dmap) >= 140000
    got = dfm.search_local_map(m, f, dmap, Tcw, n0, 1.0, 0.5, log_sf, local_slot=slots)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    m.close(); dmap.close(); f.close(); ex.close()
