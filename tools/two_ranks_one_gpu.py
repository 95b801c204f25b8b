"""Can two RCCL ranks share ONE GPU?  (The builder's lease has one.)  Two processes, both on device 0, a communicator of
world size 2 over so_exchange_create_store, three ticks with real keyframe records.  If RCCL refuses the duplicate device
this prints its error and exits 3.
    python tools/two_ranks_one_gpu.py"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, uid_path, q):
    try:
        from swarmmap_amd import synth
        from swarmmap_amd.exchange import StoreExchange, unique_id
        from swarmmap_amd.kfstore import pack_keyframe_record2, search_params
        if rank == 0:
            uid = unique_id()
            np.save(uid_path + ".tmp.npy", uid)
            os.replace(uid_path + ".tmp.npy", uid_path)
        else:
            for _ in range(600):
                if os.path.exists(uid_path):
                    break
                time.sleep(0.05)
            uid = np.load(uid_path)
        x = StoreExchange(0, rank, 2, uid, slot_keypoints=300, records_per_tick=2, store_keyframes=32)
        kfs = synth.make_kf_store_case(91, n_agents=2, kfs_per_agent=6, n_kp=300, n_places=2)
        mine = [k for k in kfs if k["agent"] == rank]
        found = []
        for t in range(3):
            recs = [pack_keyframe_record2(k["agent"], k["keyframe_id"], 0.0, k["Tcw"], synth.EUROC_K, k["xy"], k["angle"], k["octave"],
                                          k["desc"], k["map_point_id"]) for k in mine[2 * t:2 * t + 2]]
            res = x.tick_records(recs, search_params(min_votes=10, min_matches=10))
            found.append([[(c["agent_id"], c["keyframe_id"], c["n_matches"]) for c in r] for r in res])
        q.put((rank, "ok", x.store.size(), found))
        x.close()
    except BaseException as e:  # noqa: BLE001
        q.put((rank, "error", repr(e), None))


def main():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid_path = "/tmp/so_uid_%d.npy" % os.getpid()
    ps = [ctx.Process(target=worker, args=(r, uid_path, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = []
    try:
        for _ in range(2):
            out.append(q.get(timeout=120))
    except Exception as e:  # noqa: BLE001
        print("timeout / failure:", repr(e))
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    for o in sorted(out):
        print(o)
    sys.exit(0 if len(out) == 2 and all(o[1] == "ok" for o in out) else 3)


if __name__ == "__main__":
    main()
