"""Host mirror of the cross-agent keyframe exchange behind the C ABI (so_exchange_* in include/swarmorb.h): RCCL
all-gather of descriptor slots + Hamming top-2 on the gathered buffer, all on the device.  The unique id travels over
torch.distributed when a process group exists (bench.py), else the caller passes it."""
import ctypes as C

import numpy as np

from . import _lib


def _bind(lib):
    if getattr(lib, "_exchange_bound", False):
        return
    vp, i32 = C.c_void_p, C.c_int
    lib.so_exchange_unique_id.argtypes = [vp]
    lib.so_exchange_create.argtypes = [i32, i32, i32, vp, i32, C.POINTER(vp)]
    lib.so_exchange_destroy.argtypes = [vp]
    lib.so_exchange_destroy.restype = None
    lib.so_exchange_tick_dframe.argtypes = [vp, vp, i32, C.c_float, vp, vp]
    lib.so_exchange_tick.argtypes = [vp, vp, i32, i32, C.c_float, vp, vp]
    lib.so_exchange_read_slot.argtypes = [vp, i32, vp, i32, C.POINTER(i32), C.POINTER(C.c_uint64)]
    lib._exchange_bound = True


def unique_id():
    lib = _lib.load_library()
    _bind(lib)
    buf = np.zeros(128, np.uint8)
    _lib.check(lib.so_exchange_unique_id(buf.ctypes.data))
    return buf


class DeviceExchange:
    def __init__(self, device, rank, world, uid, slot_keypoints):
        self._lib = _lib.load_library()
        _bind(self._lib)
        self.rank, self.world, self.slot_keypoints = int(rank), int(world), int(slot_keypoints)
        uid = np.ascontiguousarray(uid, np.uint8)
        assert uid.nbytes == 128
        self._h = C.c_void_p()
        _lib.check(self._lib.so_exchange_create(int(device), self.rank, self.world, uid.ctypes.data, self.slot_keypoints,
                                                C.byref(self._h)))

    @classmethod
    def from_process_group(cls, device, slot_keypoints):
        """Rank 0 draws the id, torch.distributed carries it to the others."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [unique_id().tobytes() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(device, rank, world, np.frombuffer(box[0], np.uint8), slot_keypoints)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_exchange_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def tick(self, desc=None, frame_handle=None, max_dist=50, ratio=0.75):
        """desc: (n, 32) uint8 host array, or frame_handle: a so_dframe handle (device-resident descriptors).
        Returns (peer_counts, peer_candidates)."""
        counts, cands = np.zeros(self.world, np.int32), np.zeros(self.world, np.int32)
        if frame_handle is not None:
            _lib.check(self._lib.so_exchange_tick_dframe(self._h, frame_handle, int(max_dist), float(ratio),
                                                         counts.ctypes.data, cands.ctypes.data))
        else:
            d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
            _lib.check(self._lib.so_exchange_tick(self._h, d.ctypes.data if len(d) else None, len(d), int(max_dist),
                                                  float(ratio), counts.ctypes.data, cands.ctypes.data))
        return counts, cands

    def read_slot(self, peer):
        out = np.zeros((self.slot_keypoints, 32), np.uint8)
        n, cs = C.c_int(0), C.c_uint64(0)
        _lib.check(self._lib.so_exchange_read_slot(self._h, int(peer), out.ctypes.data, self.slot_keypoints, C.byref(n),
                                                   C.byref(cs)))
        return out[:n.value].copy(), int(cs.value)
