"""so_extractor_group / so_dframe_group_submit: several agents' frames through one chain of launches give every member
exactly what a lone so_extractor_submit / so_dframe_submit gives it."""
import numpy as np
import pytest

from swarmmap_amd import synth

pytestmark = pytest.mark.gpu


def _pinned(images):
    import torch
    h, w = images[0].shape
    block = torch.empty((len(images), h, w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for i, im in enumerate(images):
        view[i] = im
    return block, [view[i] for i in range(len(images))]


@pytest.mark.parametrize("n_members,size", [(1, (752, 480)), (3, (752, 480)), (4, (1241, 376)), (8, (640, 360)), (24, (400, 300))])
def test_group_extraction_equals_lone_extraction(n_members, size):
    import swarmmap_amd as S
    assert S.device_count() > 0, "these tests need a GPU"
    w, h = size
    nf = 2000 if w > 1000 else 1000
    frames = 3
    imgs = [[synth.make_canvas(100 + 7 * a + t, w, h) for a in range(n_members)] for t in range(frames)]
    solo = S.ORBextractor(nf, 1.2, 8, 20, 7)
    ref = [[tuple(x.copy() for x in solo(imgs[t][a])) for a in range(n_members)] for t in range(frames)]
    solo.close()
    exs = [S.ORBextractor(nf, 1.2, 8, 20, 7) for _ in range(n_members)]
    grp = S.ExtractorGroup(exs)
    for t in range(frames):
        keep, pinned = _pinned(imgs[t])
        grp.submit(pinned)
        for a, ex in enumerate(exs):
            kps, desc = ex.collect()
            assert kps.tobytes() == ref[t][a][0].tobytes(), (t, a)
            assert desc.tobytes() == ref[t][a][1].tobytes(), (t, a)
        del keep
    grp.close()
    for ex in exs:
        ex.close()


def test_group_members_can_still_run_alone_and_pageable_images_are_refused():
    import swarmmap_amd as S
    w, h = 752, 480
    exs = [S.ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(2)]
    grp = S.ExtractorGroup(exs)
    imgs = [synth.make_canvas(5 + a, w, h) for a in range(2)]
    with pytest.raises(S.SwarmOrbError):
        grp.submit(imgs)  # pageable numpy arrays: not device-visible
    keep, pinned = _pinned(imgs)
    grp.submit(pinned)
    a0 = [tuple(x.copy() for x in ex.collect()) for ex in exs]
    lone = [tuple(x.copy() for x in ex(im)) for ex, im in zip(exs, imgs)]  # the members' own path, same handles
    for g, l in zip(a0, lone):
        assert g[0].tobytes() == l[0].tobytes() and g[1].tobytes() == l[1].tobytes()
    grp.submit(pinned)  # and the group again after that
    for ex, l in zip(exs, lone):
        k, d = ex.collect()
        assert k.tobytes() == l[0].tobytes() and d.tobytes() == l[1].tobytes()
    grp.close()
    for ex in exs:
        ex.close()


def test_group_of_device_resident_frames_equals_lone_frames():
    """so_dframe_group_submit: keypoints, undistorted positions, descriptors, bounds and the grid of every member equal
    the lone so_dframe_submit of the same image (EuRoC lens model)."""
    import swarmmap_amd as S
    w, h, n_members, frames = 752, 480, 4, 3
    imgs = [[synth.make_canvas(300 + 5 * a + t, w, h) for a in range(n_members)] for t in range(frames)]
    ex0 = S.ORBextractor(1000, 1.2, 8, 20, 7)
    f0 = S.DeviceFrame(ex0, synth.EUROC_K, synth.EUROC_DIST)
    ref = []
    for t in range(frames):
        row = []
        for a in range(n_members):
            kps, un, d = f0(imgs[t][a])
            row.append((kps.tobytes(), un.tobytes(), d.tobytes(), f0.bounds.tobytes(), tuple(x.tobytes() for x in f0.grid())))
        ref.append(row)
    f0.close(); ex0.close()
    exs = [S.ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(n_members)]
    frs = [S.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST) for ex in exs]
    grp = S.ExtractorGroup(exs)
    for t in range(frames):
        keep, pinned = _pinned(imgs[t])
        grp.submit(pinned, frames=frs)
        for a, f in enumerate(frs):
            kps, un, d = f.collect()
            got = (kps.tobytes(), un.tobytes(), d.tobytes(), f.bounds.tobytes(), tuple(x.tobytes() for x in f.grid()))
            assert got == ref[t][a], (t, a)
        del keep
    grp.close()
    for f in frs:
        f.close()
    for ex in exs:
        ex.close()


def test_members_that_sit_a_chain_out_are_left_alone():
    """images[i] == NULL: member i takes no part in the chain (so_fleet_run's elastic ticks: an agent whose next frame is extracted
    already, and not collected yet).  Four members rotate three device-resident frames each and take part in different ticks -
    every frame that is collected equals the lone so_dframe_submit of its image, including one that was submitted two chains
    before it is collected while its neighbours went on."""
    import swarmmap_amd as S
    w, h, n_members, ticks = 752, 480, 4, 7
    # which members take part in which tick (member 3 sits out ticks 2 and 3 with its tick-1 frame uncollected)
    takes = [[1, 1, 1, 1], [1, 0, 1, 1], [0, 1, 1, 0], [1, 1, 0, 0], [1, 1, 1, 1], [0, 0, 0, 1], [1, 1, 1, 1]]
    imgs = {(t, a): synth.make_canvas(700 + 11 * a + t, w, h) for t in range(ticks) for a in range(n_members) if takes[t][a]}
    ex0 = S.ORBextractor(1000, 1.2, 8, 20, 7)
    f0 = S.DeviceFrame(ex0, synth.EUROC_K, synth.EUROC_DIST)
    ref = {}
    for key, im in imgs.items():
        kps, un, d = f0(im)
        ref[key] = (kps.tobytes(), un.tobytes(), d.tobytes(), f0.bounds.tobytes(), tuple(x.tobytes() for x in f0.grid()))
    f0.close(); ex0.close()
    exs = [S.ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(n_members)]
    frs = [[S.DeviceFrame(ex, synth.EUROC_K, synth.EUROC_DIST) for _ in range(3)] for ex in exs]
    grp = S.ExtractorGroup(exs)
    rot = [0] * n_members
    pending = [None] * n_members   # (tick, frame) submitted and not collected
    keeps = []

    def collect(a):
        t0, f = pending[a]
        kps, un, d = f.collect()
        got = (kps.tobytes(), un.tobytes(), d.tobytes(), f.bounds.tobytes(), tuple(x.tobytes() for x in f.grid()))
        assert got == ref[(t0, a)], (t0, a)
        pending[a] = None

    for t in range(ticks):
        for a in range(n_members):  # a member collects its earlier frame right before it submits the next one
            if takes[t][a] and pending[a]:
                collect(a)
        present = [a for a in range(n_members) if takes[t][a]]
        keep, pinned = _pinned([imgs[(t, a)] for a in present])
        keeps.append(keep)
        images, frames = [None] * n_members, [None] * n_members
        for a, im in zip(present, pinned):
            images[a] = im
            frames[a] = frs[a][rot[a]]
            pending[a] = (t, frames[a])
            rot[a] = (rot[a] + 1) % 3
        grp.submit(images, frames=frames)
    for a in range(n_members):
        if pending[a]:
            collect(a)
    with pytest.raises(Exception):
        grp.submit([None] * n_members)  # nobody takes part
    grp.close()
    for row in frs:
        for f in row:
            f.close()
    for ex in exs:
        ex.close()


def test_group_with_members_on_streams_of_their_own():
    """so_runtime_private_streams: every member has its own stream, the chain runs on the first member's; the others'
    collects wait for it through an event."""
    import swarmmap_amd as S
    from swarmmap_amd.replay import private_streams
    w, h, n_members = 752, 480, 3
    imgs = [synth.make_canvas(40 + a, w, h) for a in range(n_members)]
    solo = S.ORBextractor(1000, 1.2, 8, 20, 7)
    ref = [tuple(x.copy() for x in solo(im)) for im in imgs]
    solo.close()
    private_streams(True)
    try:
        exs = [S.ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(n_members)]
        grp = S.ExtractorGroup(exs)
        keep, pinned = _pinned(imgs)
        for _ in range(3):
            grp.submit(pinned)
            for a in reversed(range(n_members)):  # the last member first: its stream has nothing of its own to wait for
                k, d = exs[a].collect()
                assert k.tobytes() == ref[a][0].tobytes() and d.tobytes() == ref[a][1].tobytes(), a
        grp.close()
        for ex in exs:
            ex.close()
    finally:
        private_streams(False)


def test_group_refuses_what_it_cannot_batch():
    import swarmmap_amd as S
    a = S.ORBextractor(1000, 1.2, 8, 20, 7)
    b = S.ORBextractor(500, 1.2, 8, 20, 7)  # another configuration
    with pytest.raises(S.SwarmOrbError):
        S.ExtractorGroup([a, b])
    with pytest.raises(S.SwarmOrbError):
        S.ExtractorGroup([a, a])  # the same member twice
    b.close()
    c = S.ORBextractor(1000, 1.2, 8, 20, 7)
    grp = S.ExtractorGroup([a, c])
    keep, pinned = _pinned([synth.make_canvas(1, 752, 480), synth.make_canvas(2, 752, 480)])
    grp.submit(pinned)
    with pytest.raises(S.SwarmOrbError):
        grp.submit(pinned)  # the members' frames have not been collected
    ka, da = a.collect()
    kc, dc = c.collect()
    assert len(ka) > 100 and len(kc) > 100
    keep2, other = _pinned([synth.make_canvas(1, 640, 360), synth.make_canvas(2, 640, 360)])
    with pytest.raises(S.SwarmOrbError):
        grp.submit(other)  # image size changed between frames
    grp.submit(pinned)  # and the group still works after the refusals
    assert a.collect()[0].tobytes() == ka.tobytes() and c.collect()[0].tobytes() == kc.tobytes()
    # the candidate list of the last frame is produced on demand (the device quadtree never builds it)
    xs, ys, sc = a.candidates(0)
    assert len(xs) > len(ka) // 8 and len(xs) == len(ys) == len(sc)
    grp.close()
    a.close(); c.close()
