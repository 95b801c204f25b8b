"""The RCCL leg of the cross-agent exchange through the C ABI (so_exchange_*), on one GPU: a world of one rank runs
the real ncclCommInitRank / ncclAllGather / slot fill / header read-back code; payload semantics across ranks are
covered by the world-size-2 gloo test (tests/test_exchange_gloo.py), and two communicators in ONE process exercise the
all-gather with more than one slot where two devices are visible."""
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
