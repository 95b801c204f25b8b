"""CPU: the candidate-search oracle (oracle/kfsearch_oracle.c + oracle_py.kf_search) against a literal numpy
restatement of the detection score, and the properties the GPU tests rely on."""
import numpy as np

from oracle import oracle_py
from swarmmap_amd import synth


def _votes_numpy(q, kf, th_low=50, ratio=0.75):
    d1 = np.unpackbits(q["desc"][q["valid"] > 0], axis=1).astype(np.int16)
    d2 = np.unpackbits(kf["desc"][kf["valid"] > 0], axis=1).astype(np.int16)
    if len(d1) == 0:
        return 0
    if len(d2) == 0:
        return 0
    D = (d1[:, None, :] != d2[None, :, :]).sum(2)
    D = np.concatenate([D, np.full((len(d1), 2), 256)], axis=1)  # the scan starts from best = second = 256
    s = np.sort(D, axis=1)
    best, second = s[:, 0], s[:, 1]
    return int(((best < th_low) & (best.astype(np.float32) < np.float32(ratio) * second.astype(np.float32))).sum())


def test_votes_match_a_literal_numpy_scan():
    kfs = synth.make_kf_store_case(3, n_agents=3, kfs_per_agent=4, n_kp=120)
    q = kfs[-1]
    for kf in kfs[:-1]:
        assert oracle_py.kf_votes(q, kf) == _votes_numpy(q, kf)
    # the same place seen again scores high, another place scores ~0
    same = [oracle_py.kf_votes(q, kf) for kf in kfs[:-1] if kf["place"] == q["place"] and kf["agent"] != q["agent"]]
    other = [oracle_py.kf_votes(q, kf) for kf in kfs[:-1] if kf["place"] != q["place"]]
    assert all(v <= 3 for v in other)
    assert not same or min(same) >= 10


def test_search_reports_revisits_of_other_agents_only():
    kfs = synth.make_kf_store_case(11, n_agents=3, kfs_per_agent=8, n_kp=200, n_places=4)
    q, store = kfs[-1], kfs[:-1]
    votes, cands, n_eval = oracle_py.kf_search(q, store, min_votes=15, min_matches=15)
    for k, kf in enumerate(store):
        assert (votes[k] == -1) == (kf["agent"] == q["agent"])
    assert n_eval >= len(cands) > 0
    for slot, v, nm, m1 in cands:
        kf = store[slot]
        assert kf["agent"] != q["agent"] and kf["place"] == q["place"] and nm >= 15 and v == votes[slot]
        # pairs bind bound keypoints only, every target at most once (vbMatched2)
        i1 = np.nonzero(m1 >= 0)[0]
        assert len(i1) == nm and q["valid"][i1].all() and kf["valid"][m1[i1]].all()
        assert len(set(m1[i1].tolist())) == nm
    # candidates come ordered by votes (descending), ties by slot
    vs = [c[1] for c in cands]
    assert vs == sorted(vs, reverse=True)


def test_empty_and_unbound_queries_vote_for_nothing():
    kfs = synth.make_kf_store_case(5, n_agents=2, kfs_per_agent=3, n_kp=64)
    q = dict(kfs[-1])
    q["valid"] = np.zeros_like(q["valid"])
    votes, cands, n_eval = oracle_py.kf_search(q, kfs[:-1])
    assert cands == [] and n_eval == 0 and votes.max() <= 0
