"""Cross-agent keyframe descriptor exchange: one agent per GPU, RCCL all-gather over xGMI.

In the reference the server asks every other agent's BoW inverted index for loop/merge candidates of each new
keyframe (AgentMediator::CheckOverlapCandidates, code/src/AgentMediator.cc:140-202).  With one agent per GPU the
same question is answered without a server: every `tick`, each rank contributes ONE fixed-capacity slot holding
its newest keyframe's descriptors (header + slot_keypoints x 32 B; 32-64 KB), a single `all_gather_into_tensor`
(backend "nccl" == RCCL on ROCm) delivers all slots to all ranks, and each rank brute-force matches its own slot
against every peer slot with the Hamming top-2 kernel (so_hamming_top2_device).  The payload is latency-bound
(tens of microseconds), far below the per-link xGMI limit, so one collective per tick on padded slots is the
right shape; there is no per-frame collective.

The exchange itself only needs torch.distributed, so it is covered by world_size-2 gloo tests on CPU; the
matching step needs the GPU library.
"""
import numpy as np
import torch
import torch.distributed as dist

HEADER_ROWS = 1  # row 0 of a slot: int32 keypoint count, int32 rank, int64 checksum of the descriptor bytes


def slot_checksum(desc):
    """Order-sensitive 63-bit checksum of a descriptor block (used to verify gathered payloads)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1)
    if d.size == 0:
        return 0
    w = (np.arange(d.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return int((d.astype(np.uint64) * w).sum() % np.uint64((1 << 61) - 1))


# ---- compact binary keyframe record (include/swarmorb.h: so_keyframe_record_*) -------------------------------
import ctypes as _C


class SoKeyframeHeader(_C.Structure):
    _fields_ = [("magic", _C.c_uint32), ("version", _C.c_uint16), ("header_bytes", _C.c_uint16),
                ("agent_id", _C.c_int32), ("n_keypoints", _C.c_int32), ("keyframe_id", _C.c_uint64),
                ("timestamp", _C.c_double), ("checksum", _C.c_uint64), ("Tcw", _C.c_float * 12), ("K", _C.c_float * 4),
                ("flags", _C.c_uint32), ("n_map_points", _C.c_int32), ("reserved", _C.c_uint8 * 16)]


RECORD_HEADER_BYTES = 128
RECORD_HEADER_ROWS = RECORD_HEADER_BYTES // 32


def _rec_lib():
    from . import _lib
    lib = _lib.load_library()
    vp = _C.c_void_p
    lib.so_keyframe_record_size.restype = _C.c_size_t
    lib.so_keyframe_record_size.argtypes = [_C.c_int32]
    lib.so_keyframe_record_pack.argtypes = [_C.POINTER(SoKeyframeHeader), vp, vp, vp, vp, vp, _C.c_size_t]
    lib.so_keyframe_record_unpack.argtypes = [vp, _C.c_size_t, _C.POINTER(SoKeyframeHeader), vp, vp, vp, vp, _C.c_int32]
    return lib, _lib


def pack_keyframe_record(agent_id, keyframe_id, timestamp, Tcw, K, xy, angle, octave, desc, out=None):
    """Returns the record as a uint8 array (128 + 48 n bytes), written into `out` when given."""
    lib, _l = _rec_lib()
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    angle = np.ascontiguousarray(angle, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    h = SoKeyframeHeader()
    h.agent_id, h.n_keypoints, h.keyframe_id, h.timestamp = int(agent_id), n, int(keyframe_id), float(timestamp)
    h.Tcw[:] = [float(v) for v in np.asarray(Tcw, np.float32).reshape(12)]
    h.K[:] = [float(v) for v in np.asarray(K, np.float32).reshape(4)]
    size = lib.so_keyframe_record_size(n)
    if out is None:
        out = np.zeros(size, np.uint8)
    _l.check(lib.so_keyframe_record_pack(_C.byref(h), xy.ctypes.data, angle.ctypes.data, octave.ctypes.data,
                                         desc.ctypes.data, out.ctypes.data, out.nbytes))
    return out[:size]


def unpack_keyframe_record(rec):
    lib, _l = _rec_lib()
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1)
    h = SoKeyframeHeader()
    rc = lib.so_keyframe_record_unpack(rec.ctypes.data, rec.nbytes, _C.byref(h), None, None, None, None, 0)
    if rc not in (0, 4):  # SO_OK or SO_ERR_CAPACITY (header filled, nothing copied)
        _l.check(rc)
    n = h.n_keypoints
    xy, angle, octave, desc = np.zeros((n, 2), np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros((n, 32), np.uint8)
    _l.check(lib.so_keyframe_record_unpack(rec.ctypes.data, rec.nbytes, _C.byref(h), xy.ctypes.data, angle.ctypes.data,
                                           octave.ctypes.data, desc.ctypes.data, n))
    return dict(agent_id=h.agent_id, keyframe_id=h.keyframe_id, timestamp=h.timestamp, checksum=h.checksum,
                Tcw=np.array(h.Tcw[:], np.float32), K=np.array(h.K[:], np.float32), xy=xy, angle=angle, octave=octave, desc=desc)


class KeyframeExchange:
    """torch.distributed mirror of the exchange (gloo on CPU for the world-size-2 tests, nccl = RCCL on GPUs).  The
    product path is so_exchange_* behind the C ABI (swarmmap_amd/exchange.py); this class keeps the payload layout
    testable without GPUs.  A slot holds 1 header row + slot_keypoints descriptor rows of 32 B; pass
    record_keypoints = n to size it for whole keyframe RECORDS of up to n keypoints instead (128 + 48 n bytes)."""

    def __init__(self, slot_keypoints=2024, device=None, group=None, record_keypoints=None, slot_bytes=None):
        if not dist.is_initialized():
            raise RuntimeError("KeyframeExchange needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.slot_keypoints = int(slot_keypoints)
        if record_keypoints is not None:  # rows needed by a record of that many keypoints (header 128 B + 48 B each)
            need = RECORD_HEADER_BYTES + 48 * int(record_keypoints)
            self.slot_keypoints = max(self.slot_keypoints if slot_keypoints != 2024 else 0, (need + 31) // 32 - HEADER_ROWS)
        if slot_bytes is not None:  # an explicit slot size (records_per_tick x record stride of so_exchange_create_store)
            self.slot_keypoints = (int(slot_bytes) + 31) // 32 - HEADER_ROWS
        on_gpu = dist.get_backend(group) == "nccl"
        self.device = torch.device("cuda", device if device is not None else torch.cuda.current_device()) \
            if on_gpu else torch.device("cpu")
        rows = HEADER_ROWS + self.slot_keypoints
        self.slot = torch.zeros((rows, 32), dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros((self.world, rows, 32), dtype=torch.uint8, device=self.device)
        self._host_slot = torch.zeros((rows, 32), dtype=torch.uint8).pin_memory() if on_gpu \
            else torch.zeros((rows, 32), dtype=torch.uint8)

    def exchange(self, desc):
        """All-gather one keyframe's descriptors.  Returns (gathered [world, rows, 32] tensor, counts, checksums)."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = min(len(desc), self.slot_keypoints)
        hs = self._host_slot.numpy()
        hs[0, :] = 0
        hs[0, 0:4] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
        hs[0, 4:8] = np.frombuffer(np.int32(self.rank).tobytes(), np.uint8)
        hs[0, 8:16] = np.frombuffer(np.int64(slot_checksum(desc[:n])).tobytes(), np.uint8)
        hs[HEADER_ROWS:HEADER_ROWS + n] = desc[:n]
        self.slot.copy_(self._host_slot, non_blocking=True)
        if self.device.type == "cuda":
            dist.all_gather_into_tensor(self.gathered, self.slot, group=self.group)  # one RCCL all-gather
        else:  # gloo (CPU tests): list form over views of the same buffer
            dist.all_gather(list(self.gathered.unbind(0)), self.slot, group=self.group)
        hdr = self.gathered[:, 0, :16].cpu().numpy()  # synchronises with the collective
        counts = hdr[:, 0:4].copy().view(np.int32).reshape(-1)
        ranks = hdr[:, 4:8].copy().view(np.int32).reshape(-1)
        sums = hdr[:, 8:16].copy().view(np.int64).reshape(-1)
        assert np.array_equal(ranks, np.arange(self.world)), "all-gather slot order"
        return self.gathered, counts, sums

    def exchange_and_match(self, desc, matcher, max_dist=50, ratio=0.75):
        """Exchange, then match this rank's keyframe against every peer's (GPU).  Returns
        {peer: number of mutual-ratio-test candidates} — what the host merger would be told."""
        gathered, counts, _ = self.exchange(desc)
        torch.cuda.synchronize(self.device)
        row_bytes = gathered.shape[1] * 32
        base = gathered.data_ptr()
        mine = base + self.rank * row_bytes + HEADER_ROWS * 32
        out = {}
        for peer in range(self.world):
            if peer == self.rank or counts[peer] == 0 or counts[self.rank] == 0:
                continue
            theirs = base + peer * row_bytes + HEADER_ROWS * 32
            bi, bd, sd = matcher.hamming_top2_device(mine, int(counts[self.rank]), theirs, int(counts[peer]))
            out[peer] = int(((bd <= max_dist) & (bd < ratio * sd)).sum())
        return out

    # ---- full keyframe records (geometry + pose travel with the descriptors) ----
    def exchange_records(self, record):
        """All-gather one keyframe RECORD per rank (pack_keyframe_record).  A record of n keypoints takes 128 + 48 n
        bytes: create the exchange with record_keypoints = n (a 1000-keypoint record does NOT fit the descriptor slot
        of the same keypoint count).  Returns the list of unpacked records in rank order (None for an empty slot)."""
        rec = np.ascontiguousarray(record, np.uint8).reshape(-1)
        cap = self.slot.numel()
        if rec.nbytes > cap:
            raise ValueError("keyframe record of %d bytes does not fit the %d-byte slot" % (rec.nbytes, cap))
        hs = self._host_slot.numpy().reshape(-1)
        hs[:] = 0
        hs[:rec.nbytes] = rec
        self.slot.copy_(self._host_slot, non_blocking=True)
        if self.device.type == "cuda":
            dist.all_gather_into_tensor(self.gathered, self.slot, group=self.group)
        else:
            dist.all_gather(list(self.gathered.unbind(0)), self.slot, group=self.group)
        flat = self.gathered.reshape(self.world, -1).cpu().numpy()
        out = []
        for r in range(self.world):
            out.append(unpack_keyframe_record(flat[r]) if flat[r, :4].tobytes() == b"SOKF" else None)
        return out

    # ---- the payload of so_exchange_tick_records: records_per_tick fixed-stride record positions per rank ----
    @staticmethod
    def record_stride(slot_keypoints):
        """Stride of a record position = so_keyframe_record_size2(slot_keypoints) rounded up to 256 (kfstore.cpp)."""
        return (((RECORD_HEADER_BYTES + 52 * int(slot_keypoints) + 31) // 32 * 32) + 255) // 256 * 256

    def exchange_record_slots(self, records, records_per_tick, stride):
        """All-gather up to records_per_tick version-1/2 records of this rank.  Returns [rank][position] -> raw record
        bytes (uint8 array) or None for an unused position - what every rank's store append sees."""
        K, stride = int(records_per_tick), int(stride)
        assert len(records) <= K and self.slot.numel() >= K * stride
        hs = self._host_slot.numpy().reshape(-1)
        hs[:] = 0
        for j, r in enumerate(records):
            r = np.ascontiguousarray(r, np.uint8).reshape(-1)
            assert r.nbytes <= stride
            hs[j * stride:j * stride + r.nbytes] = r
        self.slot.copy_(self._host_slot, non_blocking=True)
        if self.device.type == "cuda":
            dist.all_gather_into_tensor(self.gathered, self.slot, group=self.group)
        else:
            dist.all_gather(list(self.gathered.unbind(0)), self.slot, group=self.group)
        flat = self.gathered.reshape(self.world, -1).cpu().numpy()
        out = []
        for r in range(self.world):
            row = []
            for j in range(K):
                rec = flat[r, j * stride:(j + 1) * stride]
                row.append(rec.copy() if rec[:4].tobytes() == b"SOKF" else None)
            out.append(row)
        return out
