"""The RCCL leg of the cross-agent exchange with MORE THAN ONE RANK: two processes, one GPU each, so_exchange_create_store
over ncclCommInitRank / ncclAllGather (what replaces AgentMediator::CheckOverlapCandidates' server-side query,
code/src/AgentMediator.cc:177-191).  The builder's and the round driver's leases have had one GPU so far (RCCL refuses two
ranks on one device, tools/two_ranks_one_gpu.py): the test SKIPS unless the machine has two, and runs by itself the day
one does.  Same scenario and the same checker as the two-rank host-transport test (tests/test_exchange_gpu.py): every
rank must report exactly what the oracle finds in ITS store, tick by tick."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import pickle, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
rank, world, uid_hex, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
import numpy as np
from swarmmap_amd import synth
from swarmmap_amd.exchange import StoreExchange
from swarmmap_amd.kfstore import pack_keyframe_record2, search_params
K, kp = 3, 260
kfs = synth.make_kf_store_case(73, n_agents=world, kfs_per_agent=9, n_kp=kp, n_places=3)
per_tick = [[k for k in kfs if k["agent"] == rank][3 * t:3 * t + 3][:1 + (t + rank) % 3] for t in range(3)]
rec = lambda k: pack_keyframe_record2(k["agent"], k["keyframe_id"], 0.0, k["Tcw"], synth.EUROC_K, k["xy"], k["angle"], k["octave"],
                                      k["desc"], k["map_point_id"])
x = StoreExchange(rank, rank, world, np.frombuffer(bytes.fromhex(uid_hex), np.uint8).copy(), kp, records_per_tick=K, store_keyframes=32)
x.set_timeout(60000)
p = search_params(min_votes=10, min_matches=10, max_candidates=8)
got = []
for t in range(3):
    res = x.tick_records([rec(k) for k in per_tick[t]], p)
    got.append([[(c["slot"], c["agent_id"], c["keyframe_id"], c["votes"], c["n_matches"], c["match_of_1"].tobytes()) for c in r] for r in res])
got.append(x.store.size()[0])
x.close()
pickle.dump(got, open(out, "wb"))
'''


def test_two_ranks_on_two_gpus_over_rccl_find_each_other(tmp_path):
    import swarmmap_amd
    if swarmmap_amd.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device): %d visible" % swarmmap_amd.device_count())
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle_py
    from swarmmap_amd import synth
    from swarmmap_amd.exchange import unique_id
    world = 2
    uid = bytes(bytearray(unique_id())).hex()
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / ("rank%d.pkl" % r)) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, str(worker), ROOT, str(r), str(world), uid, outs[r]], env=env) for r in range(world)]
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    got = [pickle.load(open(o, "rb")) for o in outs]
    kfs = synth.make_kf_store_case(73, n_agents=world, kfs_per_agent=9, n_kp=260, n_places=3)
    per_tick = [[[k for k in kfs if k["agent"] == r][3 * t:3 * t + 3][:1 + (t + r) % 3] for t in range(3)] for r in range(world)]
    found = 0
    for rank in range(world):
        store = []  # what this rank's store holds, in append order: tick by tick, the peer's records in rank order
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
