"""Multi-process CPU test (gloo, world_size 2) of the cross-agent keyframe descriptor all-gather: every rank must
end up with every rank's slot, byte for byte (checksum of checksums), in rank order, for ragged keyframe sizes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swarmmap_amd.parallel import KeyframeExchange, slot_checksum
    x = KeyframeExchange(slot_keypoints=1024)
    results = []
    for tick, n in enumerate([1000, 37 + 500 * rank, 0, 1024, 2000]):  # ragged, empty, full and over-full slots
        rng = np.random.default_rng(1000 * tick + rank)
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        g, counts, sums = x.exchange(desc)
        for r in range(world):
            rr = np.random.default_rng(1000 * tick + r)
            nr = [1000, 37 + 500 * r, 0, 1024, 2000][tick]
            want = rr.integers(0, 256, (nr, 32)).astype(np.uint8)[:1024]
            got = g[r, 1:1 + counts[r]].numpy()
            assert counts[r] == len(want)
            assert np.array_equal(got, want)
            assert sums[r] == slot_checksum(want)
        results.append((counts.tolist(), [int(s) for s in sums]))
    # full keyframe records (descriptors + geometry + pose) through the same collective
    from swarmmap_amd.parallel import pack_keyframe_record
    xr = KeyframeExchange(record_keypoints=1000)  # sized for whole records of up to 1000 keypoints (128 + 48 n bytes)
    assert xr.slot.numel() >= 128 + 48 * 1000
    for tick, n in enumerate([700, 1 + 300 * rank, 0]):
        def make(r, nn):
            g2 = np.random.default_rng(77 * tick + r)
            return dict(xy=g2.uniform(0, 752, (nn, 2)).astype(np.float32), angle=g2.uniform(0, 360, nn).astype(np.float32),
                        octave=g2.integers(0, 8, nn).astype(np.int32), desc=g2.integers(0, 256, (nn, 32)).astype(np.uint8),
                        Tcw=g2.normal(size=12).astype(np.float32))
        mine = make(rank, n)
        rec = pack_keyframe_record(rank, 100 * tick + rank, 0.05 * tick, mine["Tcw"], (458.654, 457.296, 367.215, 248.375),
                                   mine["xy"], mine["angle"], mine["octave"], mine["desc"])
        got = xr.exchange_records(rec)
        for r in range(world):
            want = make(r, [700, 1 + 300 * r, 0][tick])
            assert got[r] is not None and got[r]["agent_id"] == r and got[r]["keyframe_id"] == 100 * tick + r
            for k in ("xy", "angle", "octave", "desc", "Tcw"):
                assert np.array_equal(got[r][k], want[k]), k
    # the payload of so_exchange_tick_records: K version-2 record positions per rank, every rank keeps a store of what
    # the OTHERS sent and looks its own new keyframes up in all of it (the oracle search stands in for the GPU)
    from oracle import oracle_py
    from swarmmap_amd import synth
    from swarmmap_amd.kfstore import pack_keyframe_record2, unpack_keyframe_record2
    K, kp = 3, 160
    stride = KeyframeExchange.record_stride(kp)
    xs = KeyframeExchange(slot_bytes=K * stride)
    kfs = synth.make_kf_store_case(91, n_agents=world, kfs_per_agent=8, n_kp=kp, n_places=3)  # the same on every rank
    store, found = [], []
    for tick in range(4):
        mine = [k for k in kfs if k["agent"] == rank][2 * tick:2 * tick + 2][:1 + (tick + rank) % 2]  # ragged: 1 or 2 records
        recs = [pack_keyframe_record2(k["agent"], k["keyframe_id"], 0.1 * tick, k["Tcw"], (458.654, 457.296, 367.215, 248.375),
                                      k["xy"], k["angle"], k["octave"], k["desc"], k["map_point_id"]) for k in mine]
        got = xs.exchange_record_slots(recs, K, stride)
        for r in range(world):
            theirs = [k for k in kfs if k["agent"] == r][2 * tick:2 * tick + 2][:1 + (tick + r) % 2]
            assert [g is not None for g in got[r]] == [j < len(theirs) for j in range(K)]
            for g, want in zip(got[r], theirs):
                u = unpack_keyframe_record2(g)  # checksum verified inside
                assert u["agent_id"] == r and u["keyframe_id"] == want["keyframe_id"] and u["version"] == 2
                for key in ("desc", "angle", "xy", "octave", "map_point_id"):
                    assert np.array_equal(u[key], want[key]), key
                if r != rank:  # append order = rank order, position order
                    store.append(dict(agent=r, keyframe_id=int(u["keyframe_id"]), desc=u["desc"], angle=u["angle"],
                                      valid=(u["map_point_id"] >= 0).astype(np.uint8)))
        for k in mine:
            _, cands, _ = oracle_py.kf_search(k, store, min_votes=10, min_matches=10)
            found.append((int(k["keyframe_id"]), [(store[s]["keyframe_id"], nm) for s, _, nm, _ in cands]))
    results.append(found)
    q.put((rank, results))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_descriptor_allgather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert out[0][:-1] == out[1][:-1]  # every rank sees the same gathered payload (counts + checksums)
    # candidate search over the per-rank stores: a pair of keyframes of two agents that saw the same place is found from
    # BOTH sides (by the side whose keyframe arrived later), including pairs several ticks apart
    pairs = {r: {(a, b) for a, cl in out[r][-1] for b, _ in cl} for r in (0, 1)}
    assert pairs[0] and pairs[1]
    later_ticks = [(a, b) for r in (0, 1) for a, b in pairs[r] if (a % 1000) // 2 != (b % 1000) // 2]
    assert later_ticks, "no candidate pair spans more than one tick"
