"""Cross-agent candidate search through the C ABI (so_kfstore_*) against the oracle: detection votes of EVERY stored
keyframe bit-exact, the candidates phase 2 evaluates, their match counts and their (i1, i2) pairs identical, ring
overwrite, version-1 records, exhausted K-lists re-run on the GPU, and size-independent properties at full size.
Reference: code/src/AgentMediator.cc:177-191,204-262; code/src/ORBmatcher.cc:481-597."""
import numpy as np
import pytest

from oracle import oracle_py
from swarmmap_amd import synth

pytestmark = pytest.mark.gpu

K_EUROC = (458.654, 457.296, 367.215, 248.375)


@pytest.fixture(scope="module")
def S():
    import swarmmap_amd
    assert swarmmap_amd.device_count() > 0, "these tests need a GPU"
    return swarmmap_amd


def _rec(kf, v1=False):
    from swarmmap_amd.kfstore import pack_keyframe_record2
    from swarmmap_amd.parallel import pack_keyframe_record
    if v1:
        return pack_keyframe_record(kf["agent"], kf["keyframe_id"], 0.0, kf["Tcw"], K_EUROC, kf["xy"], kf["angle"], kf["octave"],
                                    kf["desc"])
    return pack_keyframe_record2(kf["agent"], kf["keyframe_id"], 0.0, kf["Tcw"], K_EUROC, kf["xy"], kf["angle"], kf["octave"],
                                 kf["desc"], kf["map_point_id"])


def _check_search(store, q, oracle_store, **kw):
    from swarmmap_amd.kfstore import search_params
    votes = store.votes(_rec(q))
    ovotes, ocands, on_eval = oracle_py.kf_search(q, oracle_store, **kw)
    assert np.array_equal(votes[:len(oracle_store)], ovotes), "detection votes differ from the oracle"
    assert (votes[len(oracle_store):] == -1).all()
    cands, n_eval = store.search(_rec(q), search_params(**kw))
    assert n_eval == on_eval
    assert [(c["slot"], c["votes"], c["n_matches"]) for c in cands] == [(s, v, nm) for s, v, nm, _ in ocands]
    for c, (slot, _, _, m1) in zip(cands, ocands):
        assert np.array_equal(c["match_of_1"], m1), "pairs of candidate slot %d differ" % slot
        assert c["agent_id"] == oracle_store[slot]["agent"] and c["keyframe_id"] == oracle_store[slot]["keyframe_id"]
        assert c["n_keypoints"] == len(oracle_store[slot]["desc"])
    return cands


def test_votes_candidates_and_pairs_match_the_oracle(S):
    from swarmmap_amd.kfstore import KeyframeStore, unpack_keyframe_record2
    kfs = synth.make_kf_store_case(21, n_agents=4, kfs_per_agent=10, n_kp=400, n_places=6)
    store = KeyframeStore(64, 400)
    slots = store.append([_rec(k) for k in kfs[:-4]])
    assert slots.tolist() == list(range(len(kfs) - 4))
    n_kf, n_desc = store.size()
    assert n_kf == len(kfs) - 4 and n_desc == sum(int(k["valid"].sum()) for k in kfs[:-4])
    found = 0
    for q in kfs[-4:]:  # the newest keyframe of every agent against everything the others sent before
        found += len(_check_search(store, q, kfs[:-4], min_votes=15, min_matches=15, max_candidates=8))
    assert found > 0
    st = store.last_stats()
    assert st["keyframes_scanned"] > 0 and st["pairs"] > 0 and st["scan_ms"] > 0
    # a stored record comes back byte for byte (the merger reads the candidate's geometry and pose from it)
    back = unpack_keyframe_record2(store.read(5))
    for k in ("desc", "angle", "xy", "octave", "map_point_id"):
        assert np.array_equal(back[k], kfs[5][k]), k
    assert back["agent_id"] == kfs[5]["agent"] and back["keyframe_id"] == kfs[5]["keyframe_id"]
    store.close()


@pytest.mark.parametrize("qper", [1, 2, 4])
def test_scan_variants_agree(S, qper, monkeypatch):
    """The scan kernel with 1, 2 and 4 query rows per lane (ragged query counts around the 64 / 256 boundaries)."""
    from swarmmap_amd.kfstore import KeyframeStore
    monkeypatch.setenv("SWARMORB_KF_SCAN_QPER", str(qper))
    for n_kp, seed in ((63, 1), (257, 2), (700, 3)):
        kfs = synth.make_kf_store_case(seed, n_agents=3, kfs_per_agent=5, n_kp=n_kp, bound_frac=0.9)
        store = KeyframeStore(32, n_kp)
        store.append([_rec(k) for k in kfs[:-1]])
        q = kfs[-1]
        ovotes, _, _ = oracle_py.kf_search(q, kfs[:-1])
        assert np.array_equal(store.votes(_rec(q))[:len(kfs) - 1], ovotes)
        store.close()


def test_ring_overwrites_the_oldest_keyframes(S):
    from swarmmap_amd.kfstore import KeyframeStore
    kfs = synth.make_kf_store_case(33, n_agents=3, kfs_per_agent=9, n_kp=150, n_places=3)
    store = KeyframeStore(10, 150)
    model = [None] * 10
    head = 0
    for base in range(0, 24, 5):  # appended five at a time: the ring wraps twice
        chunk = kfs[base:base + 5]
        slots = store.append([_rec(k) for k in chunk])
        for k in chunk:
            model[head] = k
            head = (head + 1) % 10
        assert slots.tolist() == [(base + j) % 10 for j in range(len(chunk))]
    assert store.size()[0] == 10
    _check_search(store, kfs[-1], model, min_votes=10, min_matches=10)
    store.close()


def test_version1_records_count_every_keypoint_as_bound(S):
    from swarmmap_amd.kfstore import KeyframeStore
    kfs = synth.make_kf_store_case(44, n_agents=2, kfs_per_agent=4, n_kp=200, n_places=2)
    allb = [dict(k, valid=np.ones(len(k["desc"]), np.uint8)) for k in kfs]
    store = KeyframeStore(16, 200)
    store.append([_rec(k, v1=True) for k in allb[:-1]])
    from swarmmap_amd.kfstore import search_params
    q = allb[-1]
    ovotes, ocands, _ = oracle_py.kf_search(q, allb[:-1], min_votes=10, min_matches=10)
    assert np.array_equal(store.votes(_rec(q, v1=True))[:len(kfs) - 1], ovotes)
    cands, _ = store.search(_rec(q, v1=True), search_params(min_votes=10, min_matches=10))
    assert [(c["slot"], c["n_matches"]) for c in cands] == [(s, nm) for s, _, nm, _ in ocands] and len(cands) > 0
    for c, oc in zip(cands, ocands):
        assert np.array_equal(c["match_of_1"], oc[3])
    store.close()


def test_exhausted_lists_are_rerun_on_the_gpu(S):
    """Twenty query keypoints each take their own near-copy in the candidate; a last one sits in the middle of all
    twenty, so the eight best entries of its K-list are taken when its turn comes."""
    from swarmmap_amd.kfstore import KeyframeStore, search_params
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (1, 32)).astype(np.uint8)
    T = np.concatenate([synth.flip_bits(rng, np.repeat(base, 20, 0), 0.03), rng.integers(0, 256, (40, 32)).astype(np.uint8)])
    Q = np.concatenate([synth.flip_bits(rng, T[:20], 0.004), base, rng.integers(0, 256, (9, 32)).astype(np.uint8)])

    def kf(agent, kid, desc):
        n = len(desc)
        return dict(agent=agent, keyframe_id=kid, desc=np.ascontiguousarray(desc), angle=np.full(n, 10.0, np.float32),
                    xy=np.zeros((n, 2), np.float32), octave=np.zeros(n, np.int32), map_point_id=np.arange(n, dtype=np.int32),
                    valid=np.ones(n, np.uint8), Tcw=np.zeros(12, np.float32))
    target, query = kf(1, 7, T), kf(0, 3, Q)
    store = KeyframeStore(4, 64)
    store.append([_rec(target)])
    kw = dict(min_votes=5, min_matches=5)
    ovotes, ocands, _ = oracle_py.kf_search(query, [target], **kw)
    cands, _ = store.search(_rec(query), search_params(**kw))
    assert store.last_stats()["reruns"] >= 1, "the construction no longer exhausts a K-list"
    assert len(cands) == 1 and cands[0]["n_matches"] == ocands[0][2] >= 15
    assert np.array_equal(cands[0]["match_of_1"], ocands[0][3])
    store.close()


def test_empty_store_unbound_query_and_bad_records(S):
    from swarmmap_amd.kfstore import KeyframeStore, search_params
    kfs = synth.make_kf_store_case(2, n_agents=2, kfs_per_agent=2, n_kp=100)
    store = KeyframeStore(8, 100)
    assert store.search(_rec(kfs[0]))[0] == [] and (store.votes(_rec(kfs[0])) == -1).all()
    store.append([_rec(k) for k in kfs[:3]])
    unbound = dict(kfs[3], map_point_id=np.full(len(kfs[3]["desc"]), -1, np.int32), valid=np.zeros(len(kfs[3]["desc"]), np.uint8))
    cands, n_eval = store.search(_rec(unbound), search_params(min_votes=1, min_matches=1))
    assert cands == [] and n_eval == 0
    empty = dict(agent=1, keyframe_id=99, desc=np.zeros((0, 32), np.uint8), angle=np.zeros(0, np.float32),
                 xy=np.zeros((0, 2), np.float32), octave=np.zeros(0, np.int32), map_point_id=np.zeros(0, np.int32),
                 Tcw=np.zeros(12, np.float32))
    assert store.append([_rec(empty)]).tolist() == [3]
    assert store.search(_rec(empty))[0] == []
    big = synth.make_kf_store_case(3, n_agents=1, kfs_per_agent=1, n_kp=300, ragged=False)[0]
    with pytest.raises(S.SwarmOrbError):
        store.append([_rec(big)])  # more keypoints than a slot holds
    garbage = np.zeros(4096, np.uint8)
    with pytest.raises(S.SwarmOrbError):
        store.append([garbage])
    store.close()


def test_full_size_store_properties(S):
    """BASELINE-size: 8 agents x 64 keyframes of 1000 keypoints; properties that need no oracle at this size: an exact
    copy of the query under another agent's name collects one vote per bound keypoint and is matched keypoint by
    keypoint; the query's own agent is never scanned; unrelated keyframes stay below the gate; a second run is
    identical."""
    from swarmmap_amd.kfstore import KeyframeStore, search_params
    rng = np.random.default_rng(77)
    n_kp, n_store = 1000, 512

    def rand_kf(agent, kid):
        mp = np.where(rng.random(n_kp) < 0.4, rng.integers(0, 1 << 30, n_kp), -1).astype(np.int32)
        return dict(agent=agent, keyframe_id=kid, desc=rng.integers(0, 256, (n_kp, 32)).astype(np.uint8),
                    angle=rng.uniform(0, 360, n_kp).astype(np.float32), xy=rng.uniform(0, 752, (n_kp, 2)).astype(np.float32),
                    octave=rng.integers(0, 8, n_kp).astype(np.int32), map_point_id=mp, Tcw=np.zeros(12, np.float32))
    kfs = [rand_kf(k % 8, k) for k in range(n_store)]
    q = rand_kf(3, 9999)
    twin = dict(q, agent=5, keyframe_id=4242)
    kfs[100] = twin
    store = KeyframeStore(n_store, n_kp + 24)
    store.append([_rec(k) for k in kfs])
    v = store.votes(_rec(q))
    nb = int((q["map_point_id"] >= 0).sum())
    assert v[100] == nb
    own = np.array([k["agent"] == 3 for k in kfs])
    assert (v[own] == -1).all() and (v[~own] >= 0).all()
    assert np.delete(v[~own], np.nonzero(np.nonzero(~own)[0] == 100)[0]).max() <= 3
    cands, n_eval = store.search(_rec(q), search_params())
    assert n_eval == 1 and len(cands) == 1 and cands[0]["slot"] == 100 and cands[0]["keyframe_id"] == 4242
    m1 = cands[0]["match_of_1"]
    bound = q["map_point_id"] >= 0
    assert cands[0]["n_matches"] == nb and np.array_equal(m1[bound], np.nonzero(bound)[0]) and (m1[~bound] == -1).all()
    st = store.last_stats()
    assert st["pairs"] == float(nb) * sum(int((k["map_point_id"] >= 0).sum()) for k, o in zip(kfs, own) if not o)
    assert np.array_equal(store.votes(_rec(q)), v)
    store.close()


def test_keyframes_of_kitti_init_size_and_the_keypoint_cap(S):
    """4024 keypoints per keyframe (the KITTI initialisation extractor's 2 x 2000 + 24): phase 2 needs more than 64 KB of
    dynamic LDS; beyond 8192 keypoints the store refuses with SO_ERR_CAPACITY."""
    from swarmmap_amd.kfstore import KeyframeStore, search_params
    kfs = synth.make_kf_store_case(8, n_agents=2, kfs_per_agent=2, n_kp=4024, n_places=1, ragged=False, bound_frac=0.7)
    store = KeyframeStore(4, 4024)
    store.append([_rec(k) for k in kfs[:-1]])
    _check_search(store, kfs[-1], kfs[:-1], min_votes=20, min_matches=20)
    store.close()
    with pytest.raises(S.SwarmOrbError):
        KeyframeStore(4, 8193)


def test_match_alone_on_slots_the_host_picked(S):
    """so_kfstore_votes -> the host filters (the reference's DetectLoop consistency groups sit here,
    code/src/AgentMediator.cc:384-456) -> so_kfstore_match: phase 2 on exactly the slots named, whatever their votes,
    equal to SearchByBoW(KF, KF) of the oracle on each of them."""
    from swarmmap_amd.kfstore import KeyframeStore, search_params
    kfs = synth.make_kf_store_case(31, n_agents=3, kfs_per_agent=6, n_kp=220, n_places=3)
    store = KeyframeStore(32, 220)
    store.append([_rec(k) for k in kfs[:-1]])
    q = kfs[-1]
    votes = store.votes(_rec(q))
    others = [k for k in range(len(kfs) - 1) if kfs[k]["agent"] != q["agent"]]
    picked = sorted(others, key=lambda k: -votes[k])[:2] + sorted(others, key=lambda k: votes[k])[:2]  # two good, two hopeless
    got = store.match(_rec(q), picked, search_params(min_matches=0))
    assert [c["slot"] for c in got] == picked
    for c in got:
        kf = kfs[c["slot"]]
        nm, _, m1 = oracle_py.search_by_bow(1, q, oracle_py._OneNode(len(q["desc"])), kf, oracle_py._OneNode(len(kf["desc"])), 0.75, True)
        assert c["n_matches"] == nm and np.array_equal(c["match_of_1"], m1)
    assert got[0]["n_matches"] >= 15 > got[-1]["n_matches"]
    assert [c["slot"] for c in store.match(_rec(q), picked, search_params(min_matches=15))] == [c["slot"] for c in got if c["n_matches"] >= 15]
    with pytest.raises(S.SwarmOrbError):
        store.match(_rec(q), [31], search_params())  # an empty slot
    store.close()
