"""The ten ORBmatcher routines one by one at the sizes their callers use (SURVEY 8a rows M1-M7 + distinctive
descriptors): host ms per call (flatten + stage + launch + wait + resolve, numpy marshalling excluded where it matters:
the C ABI call is what is timed), HIP-event kernel ms per call, queries and candidate pairs per call.
    python tools/matcher_bench.py            # JSON lines -> profiles/r4_matcher.json
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/matcher_bench.py
The batch lines time a new keyframe's whole matcher load (20 SearchForTriangulation + 20 Fuse + 1 Fuse back) as
LocalMapping issues it: one by one over host views, as one so_matcher batch over host views, as one batch over
HBM-resident keyframes."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd  # noqa: E402
from swarmmap_amd import synth  # noqa: E402
from swarmmap_amd.matcher import FeatureVector, FrameView, KFrame  # noqa: E402


def view(fr, excluded=True):
    return FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"],
                     fr.get("excluded") if excluded else None, grid_bounds=fr.get("grid_bounds"))


def timed(m, fn, reps=30):
    fn()
    host, kern = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - t0) * 1e3)
        kern.append(m.last_kernel_ms())
    return float(np.median(host)), float(np.median(kern))


def main():
    sf = synth.SCALE_FACTORS
    out = []

    def rec(name, ref, m, fn, **extra):
        h, k = timed(m, fn)
        r = dict(routine=name, reference=ref, host_ms_per_call=h, kernel_ms_per_call=k, **extra)
        out.append(r)
        print(json.dumps(r), flush=True)

    m = swarmmap_amd.ORBmatcher(0.8, True)
    fr, mps = synth.make_m1_case(1, 1000, 2000)
    rec("M1 SearchByProjection(Frame, vpMapPoints)", "ORBmatcher.cc:44-121", m, lambda: m.SearchByProjectionMapPoints(view(fr), mps, 1.0),
        keypoints=1000, map_points=2000)
    fr2, last = synth.make_m2_case(11)
    rec("M2 SearchByProjection(cur, last)", "ORBmatcher.cc:1223-1354", m, lambda: m.SearchByProjectionLastFrame(view(fr2), last, 15.0),
        keypoints=1000, map_points=1000)
    i1, i2, prev = synth.make_m4_case(21, 2000)
    rec("M4 SearchForInitialization", "ORBmatcher.cc:375-479", m, lambda: m.SearchForInitialization(view(i1, False), view(i2, False), prev.copy(), 100),
        keypoints=2000, window=100)
    kf1, node1, kf2, node2, src = synth.make_bow_case(41, 1000, 1000, p_flip=0.06)
    rng = np.random.default_rng(12)
    kf1["y"] = (kf2["y"][src] + rng.normal(0, 0.8, len(src))).astype(np.float32)
    kf1["x"] = (kf2["x"][src] + rng.uniform(-30, 30, len(src))).astype(np.float32)
    fv1, fv2 = FeatureVector(node1), FeatureVector(node2)
    for variant in (0, 1):
        rec("M3 SearchByBoW variant %d" % variant, "ORBmatcher.cc:150-262" if variant == 0 else "ORBmatcher.cc:481-597", m,
            lambda v=variant: m.SearchByBoW(v, kf1, fv1, kf2, fv2), features=1000, nodes=100)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) + rng.normal(0, 1e-6, (3, 3)).astype(np.float32)
    rec("M5 SearchForTriangulation", "ORBmatcher.cc:599-749", m,
        lambda: m.SearchForTriangulation(kf1, fv1, kf2, fv2, F12, (900.0, 240.0), sf, sf * sf), features=1000, nodes=100)
    c = synth.make_projection_case(101, 1000, 1200, keyframe_bounds=True, prebound_frac=0.1)
    KF, KFx = view(c["frame"], False), view(c["frame"])
    lsf, inv = c["log_scale_factor"], c["inv_level_sigma2"]
    rec("M6 Fuse(pKF, vpMapPoints)", "ORBmatcher.cc:751-891", m, lambda: m.Fuse(KF, c["cam"], c["Tcw"], lsf, inv, c["mp"], 3.0),
        keypoints=1000, map_points=1200)
    rec("M6 Fuse(pKF, Scw, vpPoints)", "ORBmatcher.cc:893-1009", m, lambda: m.FuseSim3(KF, c["cam"], c["Scw"], lsf, c["mp"], 4.0),
        keypoints=1000, map_points=1200)
    p = synth.make_sim3_pair_case(121, 1000)
    K1, K2 = view(p["frame1"], False), view(p["frame2"], False)
    rec("M7 SearchBySim3", "ORBmatcher.cc:1011-1221", m,
        lambda: m.SearchBySim3(K1, K2, p["cam"], p["T1w"], p["T2w"], p["s12"], p["R12"], p["t12"], lsf, lsf, p["mp1"], p["mp2"], 7.5),
        keypoints=1000, map_points="2 x 1000")
    rec("M7 SearchByProjection(pKF, Scw, vpPoints, vpMatched)", "ORBmatcher.cc:264-373", m,
        lambda: m.SearchByProjectionSim3(KFx, c["cam"], c["Scw"], lsf, c["mp"], 10), keypoints=1000, map_points=1200)
    rec("M7 SearchByProjection(Frame, pKF, sAlreadyFound)", "ORBmatcher.cc:1356-1473", m,
        lambda: m.SearchByProjectionKeyFrame(KFx, c["cam"], c["Tcw"], lsf, c["mp"], c["mp"]["angle"], 10.0, 100),
        keypoints=1000, map_points=1200)
    counts = np.concatenate([[1, 2, 3, 64, 130], rng.integers(2, 12, 3000)])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    dd = rng.integers(0, 256, (off[-1], 32)).astype(np.uint8)
    rec("MapPoint::ComputeDistinctiveDescriptors (batch)", "MapPoint.cc:323-392", m, lambda: m.ComputeDistinctiveDescriptors(off, dd),
        map_points=len(counts), observations=int(off[-1]))

    tc = synth.make_triangulation_case(5, 4000)
    rec("LocalMapping::CreateNewMapPoints per-match body (4000 matches, one neighbour)", "LocalMapping.cc:263-420", m,
        lambda: m.TriangulateMatches(tc["kf1"], [tc["kf2"]], tc["ratio_factor"], np.zeros(4000, np.int32), tc["xy1"], tc["octave1"],
                                     tc["xy2"], tc["octave2"]), matches=4000)
    nd = synth.make_normal_depth_case(7, 20000, 12)
    rec("MapPoint::UpdateNormalAndDepth (batch)", "MapPoint.cc:413-465", m,
        lambda: m.UpdateNormalAndDepth(nd["offsets"], nd["obs_Ow"], nd["Xw"], nd["ref_Ow"], nd["ref_level_scale"], nd["ref_last_scale"],
                                       nd["normal"], nd["max_dist"], nd["min_dist"]), map_points=20000, observations=int(nd["offsets"][-1]))

    # ---- a new keyframe's whole matcher load: 20 neighbours ----------------------------------------------------
    nb = 20
    neigh = [synth.make_projection_case(500 + i, 1000, 900, keyframe_bounds=True) for i in range(nb)]
    bows = []
    for i in range(nb):
        a1, n1, a2, n2, s_ = synth.make_bow_case(600 + i, 1000, 1000, p_flip=0.06)
        a1["y"] = (a2["y"][s_] + rng.normal(0, 0.8, len(s_))).astype(np.float32)
        a1["x"] = (a2["x"][s_] + rng.uniform(-30, 30, len(s_))).astype(np.float32)
        bows.append((a1, FeatureVector(n1), a2, FeatureVector(n2)))
    back = synth.make_projection_case(700, 1000, 4000, keyframe_bounds=True)
    bounds = (0.0, float(synth.EUROC[0]), 0.0, float(synth.EUROC[1]))

    def load(batched, resident=None, dmap=None):
        if batched:
            m.batch_begin()
        for i in range(nb):
            a1, f1, a2, f2 = bows[i]
            if resident:
                m.SearchForTriangulationKFrame(a1, f1, resident["tri"][i], a2["free"], F12, (900.0, 240.0))
            else:
                m.SearchForTriangulation(a1, f1, a2, f2, F12, (900.0, 240.0), sf, sf * sf)
        for i in range(nb):
            q = neigh[i]
            if dmap is not None:
                m.FuseKFrameMap(resident["fuse"][i], q["cam"], q["Tcw"], lsf, inv, dmap, map_slots[i], q["mp"].get("valid"), 3.0)
            elif resident:
                m.FuseKFrame(resident["fuse"][i], q["cam"], q["Tcw"], lsf, inv, q["mp"], 3.0)
            else:
                m.Fuse(view(q["frame"], False), q["cam"], q["Tcw"], lsf, inv, q["mp"], 3.0)
        if dmap is not None:
            m.FuseKFrameMap(resident["back"], back["cam"], back["Tcw"], lsf, inv, dmap, map_slots[nb], back["mp"].get("valid"), 3.0)
        elif resident:
            m.FuseKFrame(resident["back"], back["cam"], back["Tcw"], lsf, inv, back["mp"], 3.0)
        else:
            m.Fuse(view(back["frame"], False), back["cam"], back["Tcw"], lsf, inv, back["mp"], 3.0)
        if batched:
            m.batch_end()

    t0 = time.perf_counter(); load(False); load(False); one = (time.perf_counter() - t0) * 500
    rec("keyframe load, 41 calls one by one (host views)", "LocalMapping.cc:197-246,451-481", m, lambda: load(False), calls=41)
    rec("keyframe load, ONE batch (host views)", "LocalMapping.cc:197-246,451-481", m, lambda: load(True), calls=41)
    res = dict(tri=[KFrame(m, FrameView(b[2]["x"], b[2]["y"], b[2]["octave"], b[2]["angle"], b[2]["desc"], bounds, sf), b[3], sf * sf)
                    for b in bows],
               fuse=[KFrame(m, view(q["frame"], False)) for q in neigh], back=KFrame(m, view(back["frame"], False)))
    rec("keyframe load, ONE batch (HBM-resident keyframes)", "LocalMapping.cc:197-246,451-481", m, lambda: load(True, res), calls=41)
    # ... and with the map points where the tracking searches keep them: rows of a device-resident map table
    from swarmmap_amd.dframe import DeviceMap
    dmap, map_slots = DeviceMap(0), []
    for q in neigh + [back]:
        mp = q["mp"]
        n_mp = len(mp["max_dist"])
        first = dmap.append(np.asarray(mp["Xw"], np.float32).reshape(n_mp, 3), np.asarray(mp["normal"], np.float32).reshape(n_mp, 3),
                            mp["max_dist"], mp["min_dist"], np.asarray(mp["desc"], np.uint8).reshape(n_mp, 32))
        map_slots.append(first + np.arange(n_mp, dtype=np.int32))
    rec("keyframe load, ONE batch (HBM-resident keyframes, map points by slot from the resident map)",
        "LocalMapping.cc:197-246,451-481", m, lambda: load(True, res, dmap), calls=41)
    del one
    dmap.close()
    m.close()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r4_matcher.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
