#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE): streams a 1 GiB buffer (larger than the 256 MiB
Infinity Cache) three times with the FAST staging access shape, so the per-dispatch counter value of
`calib_read_dwords_kernel` can be divided by the known 2^30 bytes to calibrate the counter
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section: FETCH_SIZE is not calibrated for 4-byte-per-lane reads)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import swarmmap_amd  # noqa: E402

lib = swarmmap_amd.load_library()
lib.so_debug_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_void_p]
n = 1 << 30
buf = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
sink = torch.zeros(1024, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    assert lib.so_debug_stream_read(buf.data_ptr(), n, sink.data_ptr()) == 0
print("streamed", n, "bytes x3")
