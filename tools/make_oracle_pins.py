"""Regression pins of the CPU oracle (tests/golden/oracle_pins.json): digests of its outputs on seeded inputs, so that an
accidental change of the oracle (the thing every parity claim hangs on) shows up as a failing CPU test.
    python tools/make_oracle_pins.py        # rewrites the fixture; commit it together with the oracle change that caused it
The reference has no golden vectors of its own; these pin OUR restatement, they do not pin the reference."""
import hashlib
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py  # noqa: E402
from swarmmap_amd import synth  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def compute():
    pins = {}
    img = synth.make_canvas(7, 376, 240)
    kps, desc = oracle_py.extract(oracle_py.config(500), img)
    pins["extract_376x240_500"] = {"n": int(len(kps)), "sha256": digest(kps, desc)}
    fr, last = synth.make_m2_case(3, 400, 400)
    from swarmmap_amd.matcher import FrameView
    F = FrameView(fr["x"], fr["y"], fr["octave"], fr["angle"], fr["desc"], fr["bounds"], fr["scale_factors"], fr["excluded"])
    nm, k2l = oracle_py.search_by_projection_lastframe(F, last, 15.0, True)
    pins["m2_400x400"] = {"nmatches": int(nm), "sha256": digest(k2l)}
    cam = oracle_py.camera(synth.EUROC_K, synth.EUROC_DIST)
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    un = oracle_py.undistort_keypoints(cam, xy)
    b = oracle_py.image_bounds(cam, 752, 480)
    g = oracle_py.assign_features_to_grid(un, b)
    pins["frame_prepare"] = {"bounds": [float(v) for v in b], "sha256": digest(un, g["cell_of"], g["cell_items"])}
    c = synth.make_frustum_case(5, 2000)
    r = oracle_py.is_in_frustum(cam, b, c["Tcw"], c["Xw"], c["normal"], c["max_dist"], c["min_dist"], 0.5,
                                np.float32(math.log(1.2)), 8)
    pins["is_in_frustum_2000"] = {"in_view": int(r["in_view"].sum()),
                                  "sha256": digest(r["in_view"], r["proj_x"], r["proj_y"], r["view_cos"], r["pred_level"])}
    w = synth.make_ba_problem(4, 5, 4, 250, max_obs="auto")
    o = oracle_py.bundle_adjust(w)
    # floating point through libm (sin / cos / sqrt of the host's glibc): pinned by value with a tolerance, not by digest
    pins["lba_5x4x250"] = {"chi2_final": float(o["info"]["chi2_final"]), "n_outliers": int(o["info"]["n_outliers"]),
                           "Tcw_of_first_free_keyframe": [float(v) for v in o["Tcw"][int(np.nonzero(w["fixed"] == 0)[0][0])]]}
    pc = synth.make_pose_case(6, 300)
    ni, T, outl, info = oracle_py.pose_optimization(pc["Tcw"], pc["intr"], pc["Xw"], pc["obs"], pc["inv_sigma2"])
    pins["pose_300"] = {"inliers": int(ni), "Tcw": [float(v) for v in T], "outlier_sha256": digest(outl)}
    return pins


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "oracle_pins.json")
    json.dump(compute(), open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)
