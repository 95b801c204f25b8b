"""Static checks on the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inline_dpp_reads_keep_their_wait_states():
    # ba_dense.hip issues FP64 DPP FMAs from inline assembly; the compiler's hazard recogniser cannot see them
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazard.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 hazard violations" in r.stdout
