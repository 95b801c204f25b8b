#!/usr/bin/env python3
"""The diagonal-block kernel of the blocked solver issues FP64 DPP instructions from inline assembly, which the
compiler's hazard recogniser does not look into: a VGPR read through DPP must have been written at least two wait
states earlier.  This script compiles ba_dense.hip to gfx950 assembly and checks that distance for every DPP
instruction.  Usage: python tools/check_dpp_hazard.py   (exit code 1 on a violation)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "swarmmap_amd", "csrc", "ba_dense.hip")


def regs(tok):
    """VGPR numbers named by an operand like v5, -v[8:9], v[8:9]."""
    m = re.fullmatch(r"-?\|?v\[(\d+):(\d+)\]\|?", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    return {int(m.group(1))} if m else set()


def check(asm_text):
    violations, n_dpp = [], 0
    window = []  # (written vgprs, wait states this instruction provides) of the most recent instructions
    for ln, line in enumerate(asm_text.splitlines(), 1):
        code = line.split(";")[0].strip()
        if not code or code.endswith(":") or code.startswith("."):
            continue
        parts = code.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.startswith("s_nop"):
            window.append((set(), int(args[0]) + 1))
            continue
        if "_dpp" in op:
            n_dpp += 1
            src = regs(args[1])
            dist = 0
            for written, ws in reversed(window):
                if dist >= 2:
                    break
                if written & src:
                    violations.append((ln, code))
                    break
                dist += ws
        written = regs(args[0]) if args and op.startswith("v_") and not op.startswith("v_cmp") else set()
        if op.startswith(("s_", "ds_write", "global_store", "buffer_store", "scratch_store")):
            written = set()
        if op.startswith(("ds_read", "global_load", "buffer_load", "scratch_load")):
            written = regs(args[0])
        window.append((written, 1))
        window = window[-8:]
    return n_dpp, violations


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "ba_dense.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
               "--cuda-device-only", "-I", os.path.dirname(SRC), SRC, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        n, bad = check(open(out).read())
    print("%d DPP instructions, %d hazard violations" % (n, len(bad)))
    for ln, code in bad[:10]:
        print("  line %d: %s" % (ln, code))
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
