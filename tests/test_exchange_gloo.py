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
    assert out[0] == out[1]  # every rank sees the same gathered payload (counts + checksums)
