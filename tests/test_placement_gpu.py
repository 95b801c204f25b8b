"""Thread placement helpers (so_device_host_cpus, _lib.pin_process_near_device): what bench.py and the replay loop use to put
an agent's threads behind one L3 next to its GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_device_host_cpus_names_cpus_of_this_host():
    from swarmmap_amd import _lib
    node = _lib.device_host_cpus(0, -1)
    if node is None:
        pytest.skip("the host does not report a NUMA node for the device")
    online = os.sched_getaffinity(0) | node  # (the test process may itself be restricted)
    assert node and node <= online and max(node) < 4096
    groups = [_lib.device_host_cpus(0, s) for s in range(24)]
    assert all(g and g <= node for g in groups)  # every last-level-cache group lies inside the node
    distinct = {frozenset(g) for g in groups}
    n = len(distinct)
    assert n >= 1
    if n < len(groups):  # (a host with more groups per node than probed here: nothing to say about the wrap)
        assert groups[0] == groups[n] and (n == 1 or groups[0] != groups[1])  # slots wrap around the node's groups
    # groups partition (part of) the node: pairwise disjoint
    d = list(distinct)
    assert all(a.isdisjoint(b) for i, a in enumerate(d) for b in d[i + 1:])
    assert _lib.device_host_cpus(10 ** 6, 0) is None  # no such device


def test_pin_process_near_device_moves_every_thread_and_can_be_switched_off():
    code = ("import os, threading, torch\n"
            "torch.cuda.set_device(0); torch.cuda.synchronize()\n"
            "from swarmmap_amd import _lib\n"
            "cpus = _lib.pin_process_near_device(0, 1)\n"
            "masks = [os.sched_getaffinity(int(t)) for t in os.listdir('/proc/self/task')]\n"
            "print('NONE' if cpus is None else int(all(m <= cpus for m in masks) and len(masks) > 1))\n")
    for env_off in (False, True):
        env = dict(os.environ)
        env.pop("SWARMORB_NO_PIN", None)
        if env_off:
            env["SWARMORB_NO_PIN"] = "1"
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        last = out.stdout.strip().splitlines()[-1]
        assert last == "NONE" if env_off else last in ("1", "NONE")  # (NONE without the switch: a host without NUMA information)
