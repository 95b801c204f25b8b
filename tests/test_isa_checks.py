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


def test_detection_scan_streams_the_store_through_the_scalar_unit():
    """kf_scan_kernel<1> (kfstore_kernels.hip) as DESIGN.md 6 describes it: the store's rows arrive by scalar loads
    (s_load_dwordx16: the rows are SGPR operands of the vector XORs, no LDS, no vector loads in the loop), a descriptor
    pair costs 8 v_xor + 8 accumulating v_bcnt + v_med3 + v_min, and nothing spills."""
    import re
    src = os.path.join(ROOT, "swarmmap_amd", "csrc", "kfstore_kernels.hip")
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                          "-S", src, "-o", "-"], capture_output=True, text=True, timeout=600)
    assert asm.returncode == 0, asm.stderr[-2000:]
    text = asm.stdout
    m = re.search(r"^(_ZN2so\S*kf_scan_kernelILi1E\w*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert m, "kf_scan_kernel<1> not found in the assembly"
    body = m.group(2)
    loops = re.findall(r"^(\.LBB\d+_\d+):.*?Inner Loop Header.*?\n(.*?)s_cbranch_scc\d \1", body, flags=re.S | re.M)
    assert loops, "no inner loop found"
    main = max((b for _, b in loops), key=len)  # the 8-rows-per-trip loop
    n = lambda pat: len(re.findall(pat, main))  # noqa: E731
    assert n(r"s_load_dwordx16") == 4 and n(r"global_load|buffer_load|ds_read|ds_load") == 0
    assert n(r"v_bcnt_u32_b32") == 64 and n(r"v_xor_b32") == 64 and n(r"v_med3_i32") == 8 and n(r"v_min_i32") == 8
    valu = n(r"^\s+v_\w+")
    assert valu <= 8 * 18 + 6, "more vector instructions per pair than the design says: %d for 8 rows" % valu
    meta = re.search(r"\.name:\s+%s\b.*?\.vgpr_spill_count:\s+(\d+)" % re.escape(m.group(1)), text, flags=re.S)
    if meta:
        assert int(meta.group(1)) == 0


def test_pose_optimization_kernels_keep_their_edges_in_registers():
    """pose_opt_reg_kernel<256, EPT> / pose_opt_chain(_group)_kernel (ba_kernels.hip, Optimizer::PoseOptimization,
    code/src/Optimizer.cc:239-434): every instantiated variant holds its edges in registers for the whole 4 x optimize(10)
    schedule - no scratch memory.  From eight edges per thread on (KITTI-sized frames: 1793..3072 matched points) the float32
    inputs stay float in the registers and are widened where they are used (round 6; before: 84 / 328 / 556 bytes of scratch at
    8 / 10 / 12 edges per thread).  The one exception is the 12-edge instance (2817..3072 points), which spills 11 dwords."""
    import re
    src = os.path.join(ROOT, "swarmmap_amd", "csrc", "ba_kernels.hip")
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                          "-S", src, "-o", "-"], capture_output=True, text=True, timeout=1200)
    assert asm.returncode == 0, asm.stderr[-2000:]
    seen = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.a|\Z)", asm.stdout, flags=re.S):
        name, blk = m.group(1), m.group(2)
        k = re.search(r"pose_opt_reg_kernelILi(\d+)ELi(\d+)E", name)
        c = re.search(r"pose_opt_chain(_group)?_kernelILi(\d)E", name)
        if not k and not c:
            continue
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        key = ("reg", int(k.group(1)), int(k.group(2))) if k else ("chain" + (c.group(1) or ""), int(c.group(2)))
        seen[key] = scratch
    regs = sorted(k for k in seen if k[0] == "reg")
    assert [k[2] for k in regs] == [2, 3, 4, 5, 6, 7, 8, 10, 11, 12] and all(k[1] == 256 for k in regs), regs
    assert {k for k in seen if k[0] != "reg"} == {("chain", 0), ("chain", 1), ("chain_group", 0), ("chain_group", 1)}
    for key, scratch in seen.items():
        assert scratch <= (64 if key == ("reg", 256, 12) else 0), (key, scratch)
