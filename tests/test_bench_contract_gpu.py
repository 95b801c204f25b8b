"""bench.py honours the driver's contract: one JSON line (the last line of stdout) with the agreed keys, the chained
workload named in config.workload, roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "8", "--no-configs"],
                                  text=True, cwd=ROOT, timeout=900)
    line = out.strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 8 and d["unit"] == "frames/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    w = d["config"]["workload"]
    assert "upload" in w and "UndistortKeyPoints" in w and "isInFrustum" in w and "PoseOptimization over its matches" in w
    assert d["config"]["m2_matches_per_frame"] > 300 and d["config"]["inliers_per_frame"] > 400   # real matches feed the pose
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 2 and c["value"] > 0 and "sample" in c
    assert d["value"] > 10 * c["value"]
