"""Where do the C++ closed loop (HIP operators) and the oracle closed loop part ways?  Runs bench.py's stream through both
for N frames, the C++ chain twice (a difference between ITS two runs would be a race, not a rounding flip), and prints the
first frame / keyframe at which counts, poses, local-mapping log rows or keyframe bindings differ.

    python tools/loop_diff.py [N=581] [seed=20221001]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from swarmmap_amd import closedloop, synth  # noqa: E402
from swarmmap_amd.replay import Replay, make_vocabulary  # noqa: E402
from trajectory_common import OracleBackend  # noqa: E402

PLANE_Z = 2.0


def cpp_chain(n, ptrs, st, K, dist, nfeat, vocab):
    rp = Replay(0, st.w, st.h, nfeat, 5, K, dist, plane_z=PLANE_Z, local_keyframes=12, third_pose=True)
    rp.set_frames(ptrs, on_device=False)
    rp.set_vocabulary(vocab)
    rp.set_closed_loop()
    rp.prime(0)
    rp.run(0, n, True)
    rp.drain()
    rp.finish()
    a, cl = rp.log(), rp.closed_loop_log()
    rp.close()
    a.update(cl)
    return a


def first_diff(a, b, what):
    m = min(len(a["poses"]), len(b["poses"]))
    cnt = [t for t in range(m) if any(a[k][t] != b[k][t] for k in ("matches_last", "matches_map", "inliers"))]
    dp = np.abs(a["poses"][:m] - b["poses"][:m]).reshape(m, -1).max(1)
    big = np.nonzero(dp > 2e-6)[0]
    print("== %s: %d frames; frames with different counts: %d (first %s); max pose entry difference %.3g (first > 2e-6 at %s)"
          % (what, m, len(cnt), cnt[:1], dp.max(), big[:1]))
    la, lb = a["lm_log"], b["lm_log"]
    k = min(len(la), len(lb))
    rows = np.nonzero(np.any(la[:k] != lb[:k], axis=1))[0]
    print("   local-mapping log rows that differ: %d of %d (first %s)" % (len(rows), k, rows[:3]))
    for r in rows[:3]:
        print("   ", closedloop.LM_LOG_COLUMNS)
        print("    a", la[r].tolist())
        print("    b", lb[r].tolist())
    for t in cnt[:3]:
        print("    frame %d: a %s  b %s  pose diff %.3g" % (t, [int(a[k][t]) for k in ("matches_last", "matches_map", "inliers")],
                                                         [int(b[k][t]) for k in ("matches_last", "matches_map", "inliers")], dp[t]))
    return cnt, rows


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 581
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20221001
    K, dist, nfeat = synth.EUROC_K, synth.EUROC_DIST, 1000
    st = synth.FrameStream(seed=seed, size=synth.EUROC, K=K, dist=dist)
    block = torch.empty((n + 2, st.h, st.w), dtype=torch.uint8).pin_memory()
    view = block.numpy()
    for t in range(n + 2):
        view[t] = st.frame(t)
    ptrs = [block.data_ptr() + i * st.w * st.h for i in range(n + 2)]
    vocab = make_vocabulary()
    a1 = cpp_chain(n, ptrs, st, K, dist, nfeat, vocab)
    a2 = cpp_chain(n, ptrs, st, K, dist, nfeat, vocab)
    first_diff(a1, a2, "C++ chain, run 1 vs run 2")
    same = all(np.array_equal(x, y) for x, y in zip(a1["kf_bindings"], a2["kf_bindings"]))
    print("   bindings of the two runs equal:", same, " poses bit-equal:", np.array_equal(a1["poses"], a2["poses"]))
    b = closedloop.track(OracleBackend(K, nfeat, dist), None, n, K, vocab, plane_z=PLANE_Z, third_pose=True,
                         frames=[view[t] for t in range(n + 2)])
    first_diff(a1, b, "C++ chain vs oracle chain")
    M = b["map"]
    for i, (ka, kb) in enumerate(zip(a1["kf_bindings"], M.kfs)):
        d = np.nonzero(ka != kb["mp"])[0]
        if len(d):
            print("   first keyframe whose bindings differ: #%d (frame %d): %d keypoints, e.g. kp %d: hip %d oracle %d"
                  % (i, int(a1["kf_t"][i]), len(d), d[0], ka[d[0]], kb["mp"][d[0]]))
            break
    mm = min(len(a1["point_bad"]), len(M.bad))
    d = np.nonzero(a1["point_bad"][:mm] != np.asarray(M.bad[:mm]))[0]
    print("   points: hip %d oracle %d; bad flags differ at %d slots (first %s)" % (len(a1["point_bad"]), len(M.bad), len(d), d[:5]))


if __name__ == "__main__":
    main()
