"""The RCCL leg of the cross-agent exchange through the C ABI (so_exchange_*), on one GPU: a world of one rank runs
the real ncclCommInitRank / ncclAllGather / slot fill / header read-back / store append / search code; payload
semantics across ranks are covered by the world-size-2 gloo test (tests/test_exchange_gloo.py).  More than one rank
has not run on hardware (the builder's lease has one GPU)."""
import numpy as np
import pytest

from swarmmap_amd import synth
from swarmmap_amd.parallel import slot_checksum

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


def test_single_rank_rccl_exchange_host_and_device_slots(S):
    from swarmmap_amd.exchange import DeviceExchange, unique_id
    x = DeviceExchange(0, 0, 1, unique_id(), slot_keypoints=1024)
    rng = np.random.default_rng(5)
    for n in (1000, 37, 0, 1024, 2000):  # ragged, empty, full and over-full slots
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        counts, cands = x.tick(desc=desc)
        got, cs = x.read_slot(0)
        want = desc[:1024]
        assert counts[0] == len(want) and cands[0] == 0
        assert np.array_equal(got, want) and cs == slot_checksum(want)
    # the slot filled on the device from a device-resident frame: same bytes as the frame's host copy
    ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, synth.EUROC_K)
    kps, _, d = f(synth.make_canvas(3, 752, 480))
    counts, _ = x.tick(frame_handle=f._h)
    got, cs = x.read_slot(0)
    assert counts[0] == len(kps) and np.array_equal(got, d) and cs == slot_checksum(d)
    f.close(); ex.close(); x.close()


def test_unique_ids_differ_and_bad_arguments_fail(S):
    from swarmmap_amd.exchange import DeviceExchange, unique_id
    a, b = unique_id(), unique_id()
    assert a.nbytes == 128 and not np.array_equal(a, b)
    with pytest.raises(S.SwarmOrbError):
        DeviceExchange(0, 2, 1, a, 64)  # rank outside the world


K_EUROC = (458.654, 457.296, 367.215, 248.375)


def _rec(kf):
    from swarmmap_amd.kfstore import pack_keyframe_record2
    return pack_keyframe_record2(kf["agent"], kf["keyframe_id"], 0.0, kf["Tcw"], K_EUROC, kf["xy"], kf["angle"], kf["octave"],
                                 kf["desc"], kf["map_point_id"])


def test_store_exchange_single_rank_searches_the_whole_store(S):
    """so_exchange_tick_records with one rank: the rank's own records travel through ncclAllGather and are NOT stored
    (the store holds the peers' keyframes); what the peers sent earlier - put there through the store handle, as an
    earlier tick's append would have - is searched in full: candidates and pairs identical to the oracle, including
    keyframes that arrived many ticks before the query."""
    from oracle import oracle_py
    from swarmmap_amd.exchange import StoreExchange, unique_id
    from swarmmap_amd.kfstore import search_params
    kfs = synth.make_kf_store_case(55, n_agents=3, kfs_per_agent=12, n_kp=300, n_places=5)
    mine = [k for k in kfs if k["agent"] == 0]
    peers = [k for k in kfs if k["agent"] != 0]
    x = StoreExchange(0, 0, 1, unique_id(), slot_keypoints=300, records_per_tick=4, store_keyframes=64)
    x.store.append([_rec(k) for k in peers])
    p = search_params(min_votes=15, min_matches=15, max_candidates=8)
    total = 0
    for base in range(0, len(mine), 3):  # three new keyframes per tick, then an empty tick
        chunk = mine[base:base + 3]
        res = x.tick_records([_rec(k) for k in chunk], p)
        assert len(res) == len(chunk)
        for q, got in zip(chunk, res):
            _, ocands, _ = oracle_py.kf_search(q, peers, min_votes=15, min_matches=15, max_candidates=8)
            assert [(c["slot"], c["votes"], c["n_matches"]) for c in got] == [(s, v, nm) for s, v, nm, _ in ocands]
            for c, oc in zip(got, ocands):
                assert np.array_equal(c["match_of_1"], oc[3])
            total += len(got)
    assert x.tick_records([], p) == []
    assert total > 0 and x.store.size()[0] == len(peers)  # own keyframes were not appended
    # a malformed record must not keep the rank out of the collective (the peers would wait for it forever): the tick
    # runs with zero records of this rank and THEN reports the error; the next tick works
    with pytest.raises(S.SwarmOrbError, match="took part in the collective"):
        x.tick_records([np.zeros(4096, np.uint8)], p)
    assert x.read_record(0, 0) is None
    assert len(x.tick_records([_rec(mine[0])], p)) == 1
    x.close()


def test_store_exchange_keyframe_assembled_on_the_device(S):
    """so_exchange_tick_keyframe: the record of a device-resident frame (descriptors, undistorted keypoints, octaves
    from HBM; angles and bindings staged) equals the host-packed record, and its search equals the store's own."""
    from swarmmap_amd.exchange import StoreExchange, unique_id
    from swarmmap_amd.kfstore import pack_keyframe_record2, search_params, unpack_keyframe_record2
    ex = S.ORBextractor(1000, 1.2, 8, 20, 7)
    f = S.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST)
    img = synth.make_canvas(3, 752, 480)
    kps, un, d = [a.copy() for a in f(img)]
    n = len(kps)
    rng = np.random.default_rng(4)
    mp = np.where(rng.random(n) < 0.5, rng.integers(0, 1 << 20, n), -1).astype(np.int32)
    x = StoreExchange(0, 0, 1, unique_id(), slot_keypoints=1024, records_per_tick=2, store_keyframes=16)
    # a peer saw the same image (its keyframe is in the store), another one saw something else
    twin = pack_keyframe_record2(1, 500, 0.0, np.zeros(12), K_EUROC, un, kps["angle"], kps["octave"], d, mp)
    kps2, un2, d2 = [a.copy() for a in f(synth.make_canvas(8, 752, 480))]
    other = pack_keyframe_record2(2, 501, 0.0, np.zeros(12), K_EUROC, un2, kps2["angle"], kps2["octave"], d2,
                                  np.zeros(len(kps2), np.int32))
    x.store.append([other, twin])
    kps_b, _, _ = f(img)  # the frame the tick reads is the one tracked last
    assert len(kps_b) == n
    Tcw = np.arange(12, dtype=np.float32)
    got = x.tick_keyframe(f._h, n, agent_id=0, keyframe_id=77, map_point_id=mp, timestamp=1.5, Tcw=Tcw, K=K_EUROC,
                          params=search_params(), want_pairs=True)
    # the record the kernel assembled = the record the host would have packed from the frame's host mirrors
    want = pack_keyframe_record2(0, 77, 1.5, Tcw, K_EUROC, un, kps["angle"], kps["octave"], d, mp)
    assert np.array_equal(x.read_record(0, 0), want) and x.read_record(0, 1) is None
    assert unpack_keyframe_record2(want)["n_map_points"] == int((mp >= 0).sum())
    nb = int((mp >= 0).sum())
    assert len(got) == 1 and got[0]["slot"] == 1 and got[0]["keyframe_id"] == 500 and got[0]["n_matches"] == nb
    assert np.array_equal(got[0]["match_of_1"][:n][mp >= 0], np.nonzero(mp >= 0)[0])
    f.close(); ex.close(); x.close()


def test_two_ranks_over_the_host_transport_find_each_other(S):
    """World size 2 on ONE GPU.  RCCL refuses two ranks on one device (tools/two_ranks_one_gpu.py: ncclCommInitRank ->
    invalid usage), so the two ranks of this test exchange their slots through so_exchange_create_store_host - the
    transport a deployment without a shared node uses - with a thread rendezvous standing in for the network.  Everything
    around the all-gather is the code the RCCL path runs: headers of both ranks parsed, own records skipped, the peer's
    appended to the rank's own store on the device, own keyframes searched against all of it.  Three agents' keyframes
    revisit a few places; every rank must report exactly what the oracle finds in ITS store, tick by tick - including
    keyframes the peer sent several ticks earlier."""
    import threading
    from oracle import oracle_py
    from swarmmap_amd.exchange import StoreExchange
    from swarmmap_amd.kfstore import search_params
    world, K, kp = 2, 3, 260
    kfs = synth.make_kf_store_case(73, n_agents=world, kfs_per_agent=9, n_kp=kp, n_places=3)
    per_tick = [[[k for k in kfs if k["agent"] == r][3 * t:3 * t + 3][:1 + (t + r) % 3] for t in range(3)] for r in range(world)]
    gate = threading.Barrier(world)
    bufs = [None] * world
    errs, got = [], [[] for _ in range(world)]

    def make_gather(rank):
        def allgather(send, recv):
            bufs[rank] = send.copy()
            gate.wait(timeout=60)
            n = len(send)
            for r in range(world):
                recv[r * n:(r + 1) * n] = bufs[r]
            gate.wait(timeout=60)
        return allgather

    def run(rank):
        try:
            x = StoreExchange.over_host_transport(0, rank, world, make_gather(rank), kp, records_per_tick=K, store_keyframes=32)
            p = search_params(min_votes=10, min_matches=10, max_candidates=8)
            for t in range(3):
                res = x.tick_records([_rec(k) for k in per_tick[rank][t]], p)
                got[rank].append([[(c["slot"], c["agent_id"], c["keyframe_id"], c["votes"], c["n_matches"], c["match_of_1"].tobytes())
                                   for c in r] for r in res])
            got[rank].append(x.store.size()[0])
            x.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            gate.abort()

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    found = 0
    for rank in range(world):
        store = []  # what this rank's store holds, in append order: tick by tick, the peer's records in position order
        for t in range(3):
            for r in range(world):
                if r != rank:
                    store += per_tick[r][t]
            for j, q in enumerate(per_tick[rank][t]):
                _, ocands, _ = oracle_py.kf_search(q, store, min_votes=10, min_matches=10, max_candidates=8)
                want = [(s, store[s]["agent"], store[s]["keyframe_id"], v, nm, m1.tobytes()) for s, v, nm, m1 in ocands]
                assert got[rank][t][j] == want, (rank, t, j)
                found += len(want)
        assert got[rank][3] == len(store)
    assert found > 0


def test_a_tick_whose_collective_does_not_complete_times_out_and_kills_the_handle(S):
    """A peer that never enters a tick must not hang the survivors: the wait for the collective is bounded
    (so_exchange_set_timeout / SWARMORB_COLLECTIVE_TIMEOUT_MS).  One rank over RCCL; the peer that is late is a
    workgroup spinning for 400 ms on the tick's stream in front of the collective (so_exchange_debug_stall - nothing
    hangs for good).  With a 50 ms budget the tick fails with 'collective timed out', the handle is dead, later ticks
    fail at once, close() returns; with a budget above the stall the same tick succeeds."""
    import time
    from swarmmap_amd import SwarmOrbError
    from swarmmap_amd.exchange import StoreExchange, unique_id
    from swarmmap_amd.kfstore import search_params
    kp = 200
    kfs = synth.make_kf_store_case(81, n_agents=1, kfs_per_agent=3, n_kp=kp, n_places=1)
    p = search_params(min_votes=10, min_matches=10, max_candidates=4)
    x = StoreExchange(0, 0, 1, unique_id(), kp, records_per_tick=2, store_keyframes=16)
    x.set_timeout(2000)
    x.debug_stall(300)
    t0 = time.perf_counter()
    x.tick_records([_rec(kfs[0])], p)            # late, but inside the budget
    assert 0.25 < time.perf_counter() - t0 < 1.5 and not x.is_dead()
    x.set_timeout(50)
    x.debug_stall(400)
    t0 = time.perf_counter()
    with pytest.raises(SwarmOrbError, match="timed out"):
        x.tick_records([_rec(kfs[1])], p)
    assert 0.04 < time.perf_counter() - t0 < 0.3 and x.is_dead()
    t0 = time.perf_counter()
    with pytest.raises(SwarmOrbError, match="timed out"):
        x.tick_records([_rec(kfs[2])], p)
    assert time.perf_counter() - t0 < 0.02
    x.close()                                    # does not wait for the stalled stream
    time.sleep(0.5)                              # (let the spinning workgroup end before the next test uses the GPU)


def test_a_rank_with_a_local_problem_still_takes_part_in_the_collective(S):
    """a tick is collective, so a rank whose OWN input is unusable (frame never collected, missing bindings,
    too many records) must not return before the all-gather - its peer would wait for ever.  Two ranks over the host
    transport: rank 1 hands in a bad keyframe at every tick and gets 'invalid argument' back; rank 0's ticks complete
    (its gather callback sees both ranks arrive) and its store simply receives nothing from rank 1."""
    import threading
    from swarmmap_amd import SwarmOrbError
    from swarmmap_amd.exchange import StoreExchange
    from swarmmap_amd.kfstore import search_params
    world, kp = 2, 200
    kfs = synth.make_kf_store_case(83, n_agents=1, kfs_per_agent=4, n_kp=kp, n_places=2)
    gate = threading.Barrier(world)
    bufs, errs, bad_errors, sizes = [None] * world, [], [], []

    def make_gather(rank):
        def allgather(send, recv):
            bufs[rank] = send.copy()
            gate.wait(timeout=20)
            n = len(send)
            for r in range(world):
                recv[r * n:(r + 1) * n] = bufs[r]
            gate.wait(timeout=20)
        return allgather

    def run(rank):
        try:
            x = StoreExchange.over_host_transport(0, rank, world, make_gather(rank), kp, records_per_tick=2, store_keyframes=16)
            p = search_params(min_votes=10, min_matches=10, max_candidates=4)
            for t in range(3):
                if rank == 0:
                    x.tick_records([_rec(kfs[t])], p)
                else:
                    try:
                        if t == 0:
                            x.tick_records([_rec(kfs[0])] * 3, p)        # more records than the slot holds
                        elif t == 1:
                            bad = _rec(kfs[1]).copy(); bad[:4] = 0       # not a keyframe record
                            x.tick_records([bad], p)
                        else:
                            x.tick_keyframe(None, kp, agent_id=1, keyframe_id=9, map_point_id=np.zeros(kp, np.int32),
                                            timestamp=0.0, Tcw=np.zeros(12, np.float32), K=K_EUROC)  # no frame at all
                    except SwarmOrbError as e:
                        bad_errors.append(str(e))
            sizes.append((rank, x.store.size()[0]))
            x.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            gate.abort()

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errs, errs
    assert len(bad_errors) == 3 and all("took part" in e for e in bad_errors), bad_errors
    assert dict(sizes) == {0: 0, 1: 3}   # rank 1 received rank 0's three keyframes, rank 0 nothing
