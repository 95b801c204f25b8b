"""The host side of the replay harness (swarmmap_amd/host/replay.cc + closedloop.cc: tracking thread, local-mapping thread, the
hand-over between them, the map model, the window gather, so_fleet_run) under AddressSanitizer + UBSan and under
ThreadSanitizer, on the CPU: built against tests/cpp/mock_swarmorb.cc, a stand-in for the C ABI that returns plausible indices
(no operators - those are the HIP library's, checked on the GPU), driven by tests/cpp/replay_sanitize.cc.  Sanitizers never run
on the GPU box.  A clean log: profiles/r6_host_sanitizers.txt."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "swarmmap_amd", "csrc")


def _make(target):
    r = subprocess.run(["make", "-C", CSRC, target], capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-6000:]
    for mode in ("solo", "policy", "threads", "fleet"):
        assert "%s: ok" % mode in out, out[-3000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "WARNING: ThreadSanitizer" not in out, out[-6000:]


def test_replay_harness_is_clean_under_address_and_undefined_behaviour_sanitizers():
    _make("host-asan")


def test_replay_harness_is_clean_under_thread_sanitizer():
    _make("host-tsan")
