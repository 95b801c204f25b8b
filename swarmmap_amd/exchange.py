"""Host mirror of the cross-agent keyframe exchange behind the C ABI (so_exchange_* in include/swarmorb.h).
DeviceExchange: RCCL all-gather of descriptor slots + Hamming top-2 on the gathered buffer (the newest keyframes of the
ranks against each other).  StoreExchange: the reference's candidate search (code/src/AgentMediator.cc:177-191,204-262) -
keyframe RECORDS travel, every rank appends what it receives to an HBM keyframe store and looks its new keyframes up
in the whole store.  The unique id travels over torch.distributed when a process group exists (bench.py), else the
caller passes it."""
import ctypes as C

import numpy as np

from . import _lib


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


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
    lib.so_exchange_create_store.argtypes = [i32, i32, i32, vp, i32, i32, i32, C.POINTER(vp)]
    lib.so_exchange_tick_records.argtypes = [vp, vp, C.c_size_t, C.c_int32, vp, vp, vp, vp]
    lib.so_exchange_tick_keyframe.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.so_exchange_create_store_host.argtypes = [i32, i32, i32, ALLGATHER_FN, vp, i32, i32, i32, C.POINTER(vp)]
    lib.so_exchange_store.argtypes = [vp]
    lib.so_exchange_read_record.argtypes = [vp, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.so_exchange_store.restype = vp
    lib.so_exchange_set_timeout.argtypes = [vp, i32]
    lib.so_exchange_is_dead.argtypes = [vp]
    lib.so_exchange_debug_stall.argtypes = [vp, i32]
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

    def set_timeout(self, milliseconds):
        """Budget of the wait for a tick's collective (0: unbounded).  On expiry the tick raises 'collective timed out'
        and the handle is dead."""
        _lib.check(self._lib.so_exchange_set_timeout(self._h, int(milliseconds)))

    def is_dead(self):
        return bool(self._lib.so_exchange_is_dead(self._h))

    def debug_stall(self, milliseconds):
        _lib.check(self._lib.so_exchange_debug_stall(self._h, int(milliseconds)))

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


class StoreExchange:
    """so_exchange_create_store: up to `records_per_tick` keyframe records per rank and tick, a keyframe store of
    `store_keyframes` records behind it."""

    def __init__(self, device, rank, world, uid, slot_keypoints, records_per_tick=4, store_keyframes=4096):
        from . import kfstore
        self._lib = _lib.load_library()
        _bind(self._lib)
        kfstore._bind(self._lib)
        self.rank, self.world, self.slot_keypoints = int(rank), int(world), int(slot_keypoints)
        self.records_per_tick = int(records_per_tick)
        uid = np.ascontiguousarray(uid, np.uint8)
        assert uid.nbytes == 128
        self._h = C.c_void_p()
        _lib.check(self._lib.so_exchange_create_store(int(device), self.rank, self.world, uid.ctypes.data, self.slot_keypoints,
                                                      self.records_per_tick, int(store_keyframes), C.byref(self._h)))
        # the store behind the communicator (owned by it): a borrowed handle in the KeyframeStore mirror
        self.store = kfstore.KeyframeStore.__new__(kfstore.KeyframeStore)
        self.store._lib = self._lib
        self.store.capacity, self.store.slot_keypoints = int(store_keyframes), self.slot_keypoints
        self.store._h = C.c_void_p(self._lib.so_exchange_store(self._h))
        self.store.close = lambda: None

    @classmethod
    def over_host_transport(cls, device, rank, world, allgather, slot_keypoints, records_per_tick=4, store_keyframes=4096):
        """so_exchange_create_store_host: the same exchange over the caller's own all-gather instead of RCCL.
        allgather(send: uint8 array, recv: writable uint8 array of world x len(send)) -> None."""
        from . import kfstore
        self = cls.__new__(cls)
        self._lib = _lib.load_library()
        _bind(self._lib)
        kfstore._bind(self._lib)
        self.rank, self.world, self.slot_keypoints = int(rank), int(world), int(slot_keypoints)
        self.records_per_tick = int(records_per_tick)

        def _cb(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(send))
                dst = np.ctypeslib.as_array((C.c_uint8 * (nbytes * self.world)).from_address(recv))
                allgather(src, dst)
                return 0
            except Exception:  # noqa: BLE001 - reported through the status of the tick
                return 1

        self._cb = ALLGATHER_FN(_cb)  # (kept alive as long as the exchange)
        self._h = C.c_void_p()
        _lib.check(self._lib.so_exchange_create_store_host(int(device), self.rank, self.world, self._cb, None, self.slot_keypoints,
                                                           self.records_per_tick, int(store_keyframes), C.byref(self._h)))
        self.store = kfstore.KeyframeStore.__new__(kfstore.KeyframeStore)
        self.store._lib = self._lib
        self.store.capacity, self.store.slot_keypoints = int(store_keyframes), self.slot_keypoints
        self.store._h = C.c_void_p(self._lib.so_exchange_store(self._h))
        self.store.close = lambda: None
        return self

    @classmethod
    def from_process_group(cls, device, slot_keypoints, records_per_tick=4, store_keyframes=4096):
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [unique_id().tobytes() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(device, rank, world, np.frombuffer(box[0], np.uint8), slot_keypoints, records_per_tick, store_keyframes)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.store._h = C.c_void_p()
            self._lib.so_exchange_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    set_timeout = DeviceExchange.set_timeout
    is_dead = DeviceExchange.is_dead
    debug_stall = DeviceExchange.debug_stall

    def tick_records(self, records, params=None, want_pairs=True):
        """records: this rank's new keyframes (list of record arrays, at most records_per_tick; may be empty).
        Returns one list of candidate dicts per record."""
        from . import kfstore
        p = params if params is not None else kfstore.search_params()
        n = len(records)
        stride = (max([r.nbytes for r in records] + [128]) + 31) // 32 * 32
        block = np.zeros((max(n, 1), stride), np.uint8)
        for j, r in enumerate(records):
            block[j, :r.nbytes] = r.reshape(-1)
        mc = max(p.max_candidates, 1)
        out = (kfstore.SoKfCandidate * (mc * max(n, 1)))()
        pairs = np.full(max(n, 1) * mc * self.slot_keypoints, -1, np.int32) if want_pairs else None
        n_out = np.zeros(max(n, 1), np.int32)
        _lib.check(self._lib.so_exchange_tick_records(self._h, block.ctypes.data, stride, n, C.byref(p), out,
                                                      pairs.ctypes.data if want_pairs else None, n_out.ctypes.data))
        res = []
        for j in range(n):
            nq = int(block[j, 12:16].view(np.int32)[0])
            pj = pairs[j * mc * self.slot_keypoints:(j + 1) * mc * self.slot_keypoints] if want_pairs else None
            res.append(kfstore.candidates_to_list(out[j * mc:(j + 1) * mc], int(n_out[j]), pj, nq))
        return res

    def read_record(self, peer, index):
        """Record `index` of rank `peer` as the last tick delivered it (None for an unused position)."""
        ln = C.c_size_t(0)
        _lib.check(self._lib.so_exchange_read_record(self._h, int(peer), int(index), None, 0, C.byref(ln)))
        if ln.value == 0:
            return None
        rec = np.zeros(ln.value, np.uint8)
        _lib.check(self._lib.so_exchange_read_record(self._h, int(peer), int(index), rec.ctypes.data, rec.nbytes, C.byref(ln)))
        return rec

    def tick_keyframe(self, frame_handle, n_keypoints, agent_id, keyframe_id, map_point_id, timestamp=0.0, Tcw=None, K=None,
                      params=None, want_pairs=False):
        """One keyframe whose descriptors / undistorted keypoints are device-resident (a collected so_dframe)."""
        from . import kfstore
        from .parallel import SoKeyframeHeader
        p = params if params is not None else kfstore.search_params()
        h = SoKeyframeHeader()
        h.agent_id, h.n_keypoints, h.keyframe_id, h.timestamp = int(agent_id), int(n_keypoints), int(keyframe_id), float(timestamp)
        if Tcw is not None:
            h.Tcw[:] = [float(v) for v in np.asarray(Tcw, np.float32).reshape(12)]
        if K is not None:
            h.K[:] = [float(v) for v in np.asarray(K, np.float32).reshape(4)]
        mp = np.ascontiguousarray(map_point_id, np.int32)
        mc = max(p.max_candidates, 1)
        out = (kfstore.SoKfCandidate * mc)()
        pairs = np.full(mc * self.slot_keypoints, -1, np.int32) if want_pairs else None
        n_out = np.zeros(1, np.int32)
        _lib.check(self._lib.so_exchange_tick_keyframe(self._h, frame_handle, C.byref(h), mp.ctypes.data, C.byref(p), out,
                                                       pairs.ctypes.data if want_pairs else None, n_out.ctypes.data))
        return kfstore.candidates_to_list(out, int(n_out[0]), pairs, min(int(n_keypoints), self.slot_keypoints))
