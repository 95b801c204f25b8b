"""The detection scan of the cross-agent candidate search on one GPU: kernel time and integer-VALU rate over the number
of query rows per lane (SWARMORB_KF_SCAN_QPER = 1 / 2 / 4) and store shapes.
    python tools/kfscan_bench.py            # JSON lines"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from swarmmap_amd.kfstore import KeyframeStore, search_params  # noqa: E402

PEAK = bench.INT_PEAK_TOPS


def main():
    cases = [("8x512 kf, 40% bound", 512, 0.4, 1000), ("8x128 kf, all bound", 128, 1.0, 1000), ("8x64 kf, 2000 kp all bound", 64, 1.0, 2000)]
    for name, per_agent, frac, n_kp in cases:
        rng = np.random.default_rng(11)
        recs = [bench.random_keyframe_records(rng, 1, per_agent, n_kp, frac, first_agent=1 + a) for a in range(8)]
        q = bench.random_keyframe_records(rng, 1, 1, n_kp, frac, first_agent=0)[0]
        for qper in ((0,) if os.environ.get("SWARMORB_KFSCAN_ONLY_DEFAULT") else (1, 2, 4)):  # 0: the library's choice
            os.environ["SWARMORB_KF_SCAN_QPER"] = str(qper)
            store = KeyframeStore(8 * per_agent, n_kp + 24)
            for r in recs:
                store.append(r)
            p = search_params()
            for _ in range(3):
                store.search(q, p, want_pairs=False)
            ms = []
            for _ in range(15):
                store.search(q, p, want_pairs=False)
                ms.append(store.last_stats()["scan_ms"])
            st = store.last_stats()
            m = float(np.median(ms))
            print(json.dumps({"case": name, "qper": qper, "scan_ms": m, "min_ms": float(np.min(ms)), "pairs": st["pairs"],
                              "tera_lane_ops": 16 * st["pairs"] / (m * 1e-3) / 1e12, "frac_of_int_peak": 16 * st["pairs"] / (m * 1e-3) / 1e12 / PEAK,
                              "issued_frac": 18 * st["pairs"] / (m * 1e-3) / 1e12 / PEAK}), flush=True)
            store.close()


if __name__ == "__main__":
    main()
