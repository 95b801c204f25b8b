"""Host mirror of the keyframe store and the cross-agent candidate search behind the C ABI (so_kfstore_* in
include/swarmorb.h): keyframe records in an HBM ring, every new keyframe looked up in the WHOLE store (detection scan +
exact SearchByBoW(KF, KF) matching of the candidates), the counterpart of AgentMediator::CheckOverlapCandidates /
GetSim3 (code/src/AgentMediator.cc:140-262)."""
import ctypes as C

import numpy as np

from . import _lib
from .parallel import SoKeyframeHeader

MAX_CANDIDATES = 64


class SoKfSearchParams(C.Structure):
    _fields_ = [("th_low", C.c_int32), ("nn_ratio", C.c_float), ("check_orientation", C.c_int32),
                ("min_votes", C.c_int32), ("min_matches", C.c_int32), ("max_candidates", C.c_int32)]


class SoKfCandidate(C.Structure):
    _fields_ = [("slot", C.c_int32), ("agent_id", C.c_int32), ("keyframe_id", C.c_uint64), ("n_keypoints", C.c_int32),
                ("votes", C.c_int32), ("n_matches", C.c_int32), ("reserved", C.c_int32)]


def search_params(th_low=50, nn_ratio=0.75, check_ori=True, min_votes=20, min_matches=20, max_candidates=16):
    return SoKfSearchParams(int(th_low), float(nn_ratio), int(bool(check_ori)), int(min_votes), int(min_matches),
                            int(max_candidates))


def _bind(lib):
    if getattr(lib, "_kfstore_bound", False):
        return
    vp, i32 = C.c_void_p, C.c_int32
    lib.so_keyframe_record_size2.restype = C.c_size_t
    lib.so_keyframe_record_size2.argtypes = [i32]
    lib.so_keyframe_record_pack2.argtypes = [C.POINTER(SoKeyframeHeader), vp, vp, vp, vp, vp, vp, C.c_size_t]
    lib.so_keyframe_record_unpack2.argtypes = [vp, C.c_size_t, C.POINTER(SoKeyframeHeader), vp, vp, vp, vp, vp, i32]
    lib.so_kfstore_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.so_kfstore_destroy.argtypes = [vp]
    lib.so_kfstore_destroy.restype = None
    lib.so_kfstore_append.argtypes = [vp, vp, C.c_size_t, i32, vp]
    lib.so_kfstore_size.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_int64)]
    lib.so_kfstore_votes.argtypes = [vp, vp, C.c_size_t, i32, C.c_float, vp]
    lib.so_kfstore_search.argtypes = [vp, vp, C.c_size_t, C.POINTER(SoKfSearchParams), vp, vp, C.POINTER(i32), C.POINTER(i32)]
    lib.so_kfstore_match.argtypes = [vp, vp, C.c_size_t, vp, i32, C.POINTER(SoKfSearchParams), vp, vp, C.POINTER(i32)]
    lib.so_kfstore_read.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.so_kfstore_last_stats.argtypes = [vp, vp]
    lib._kfstore_bound = True


def _lib_bound():
    lib = _lib.load_library()
    _bind(lib)
    return lib


def record_size2(n):
    return int(_lib_bound().so_keyframe_record_size2(int(n)))


def pack_keyframe_record2(agent_id, keyframe_id, timestamp, Tcw, K, xy, angle, octave, desc, map_point_id, out=None):
    """Version-2 record (128 + 52 n bytes rounded up to 32): version 1 + the id of the map point bound to every
    keypoint (-1 = none)."""
    lib = _lib_bound()
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = len(xy)
    angle = np.ascontiguousarray(angle, np.float32)
    octave = np.ascontiguousarray(octave, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    mp = np.ascontiguousarray(map_point_id, np.int32)
    assert len(angle) == n and len(octave) == n and len(desc) == n and len(mp) == n
    h = SoKeyframeHeader()
    h.agent_id, h.n_keypoints, h.keyframe_id, h.timestamp = int(agent_id), n, int(keyframe_id), float(timestamp)
    h.Tcw[:] = [float(v) for v in np.asarray(Tcw, np.float32).reshape(12)]
    h.K[:] = [float(v) for v in np.asarray(K, np.float32).reshape(4)]
    size = record_size2(n)
    if out is None:
        out = np.zeros(size, np.uint8)
    _lib.check(lib.so_keyframe_record_pack2(C.byref(h), xy.ctypes.data, angle.ctypes.data, octave.ctypes.data,
                                            desc.ctypes.data, mp.ctypes.data, out.ctypes.data, out.nbytes))
    return out[:size]


def unpack_keyframe_record2(rec):
    lib = _lib_bound()
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1)
    h = SoKeyframeHeader()
    rc = lib.so_keyframe_record_unpack2(rec.ctypes.data, rec.nbytes, C.byref(h), None, None, None, None, None, 0)
    if rc not in (0, 4):
        _lib.check(rc)
    n = h.n_keypoints
    xy, angle, octave = np.zeros((n, 2), np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    desc, mp = np.zeros((n, 32), np.uint8), np.zeros(n, np.int32)
    _lib.check(lib.so_keyframe_record_unpack2(rec.ctypes.data, rec.nbytes, C.byref(h), xy.ctypes.data, angle.ctypes.data,
                                              octave.ctypes.data, desc.ctypes.data, mp.ctypes.data, n))
    return dict(agent_id=h.agent_id, keyframe_id=h.keyframe_id, timestamp=h.timestamp, checksum=h.checksum, version=h.version,
                n_map_points=h.n_map_points, Tcw=np.array(h.Tcw[:], np.float32), K=np.array(h.K[:], np.float32), xy=xy,
                angle=angle, octave=octave, desc=desc, map_point_id=mp)


def candidates_to_list(out, n_out, pairs, n_query):
    res = []
    for c in range(n_out):
        o = out[c]
        res.append(dict(slot=o.slot, agent_id=o.agent_id, keyframe_id=int(o.keyframe_id), n_keypoints=o.n_keypoints,
                        votes=o.votes, n_matches=o.n_matches,
                        match_of_1=pairs[c * n_query:(c + 1) * n_query].copy() if pairs is not None else None))
    return res


class KeyframeStore:
    def __init__(self, capacity_keyframes, slot_keypoints, device=0):
        self._lib = _lib_bound()
        self.capacity, self.slot_keypoints = int(capacity_keyframes), int(slot_keypoints)
        self._h = C.c_void_p()
        _lib.check(self._lib.so_kfstore_create(int(device), self.capacity, self.slot_keypoints, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_kfstore_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def append(self, records):
        """records: list of uint8 record arrays (pack_keyframe_record / pack_keyframe_record2).  Returns their slots."""
        if not records:
            return np.zeros(0, np.int32)
        stride = (max(r.nbytes for r in records) + 31) // 32 * 32
        block = np.zeros((len(records), stride), np.uint8)
        for j, r in enumerate(records):
            block[j, :r.nbytes] = r.reshape(-1)
        slots = np.zeros(len(records), np.int32)
        _lib.check(self._lib.so_kfstore_append(self._h, block.ctypes.data, stride, len(records), slots.ctypes.data))
        return slots

    def size(self):
        n, d = C.c_int32(0), C.c_int64(0)
        _lib.check(self._lib.so_kfstore_size(self._h, C.byref(n), C.byref(d)))
        return n.value, d.value

    def votes(self, query_record, th_low=50, nn_ratio=0.75):
        q = np.ascontiguousarray(query_record, np.uint8).reshape(-1)
        v = np.zeros(self.capacity, np.int32)
        _lib.check(self._lib.so_kfstore_votes(self._h, q.ctypes.data, q.nbytes, int(th_low), float(nn_ratio), v.ctypes.data))
        return v

    def search(self, query_record, params=None, want_pairs=True):
        """Returns (list of candidate dicts that reached min_matches, keyframes phase 2 looked at)."""
        p = params if params is not None else search_params()
        q = np.ascontiguousarray(query_record, np.uint8).reshape(-1)
        nq = int(q[12:16].view(np.int32)[0])
        out = (SoKfCandidate * max(p.max_candidates, 1))()
        pairs = np.full(max(p.max_candidates, 1) * max(nq, 1), -1, np.int32) if want_pairs else None
        n_out, n_eval = C.c_int32(0), C.c_int32(0)
        _lib.check(self._lib.so_kfstore_search(self._h, q.ctypes.data, q.nbytes, C.byref(p), out,
                                               pairs.ctypes.data if want_pairs else None, C.byref(n_out), C.byref(n_eval)))
        return candidates_to_list(out, n_out.value, pairs, nq), n_eval.value

    def match(self, query_record, slots, params=None):
        """Phase 2 alone on the given store slots (the host has filtered the detection result itself)."""
        p = params if params is not None else search_params()
        q = np.ascontiguousarray(query_record, np.uint8).reshape(-1)
        nq = int(q[12:16].view(np.int32)[0])
        sl = np.ascontiguousarray(slots, np.int32)
        out = (SoKfCandidate * max(len(sl), 1))()
        pairs = np.full(max(len(sl), 1) * max(nq, 1), -1, np.int32)
        n_out = C.c_int32(0)
        _lib.check(self._lib.so_kfstore_match(self._h, q.ctypes.data, q.nbytes, sl.ctypes.data, len(sl), C.byref(p), out,
                                              pairs.ctypes.data, C.byref(n_out)))
        return candidates_to_list(out, n_out.value, pairs, nq)

    def read(self, slot):
        ln = C.c_size_t(0)
        _lib.check(self._lib.so_kfstore_read(self._h, int(slot), None, 0, C.byref(ln)))
        rec = np.zeros(ln.value, np.uint8)
        _lib.check(self._lib.so_kfstore_read(self._h, int(slot), rec.ctypes.data, rec.nbytes, C.byref(ln)))
        return rec

    def last_stats(self):
        s = np.zeros(6, np.float64)
        _lib.check(self._lib.so_kfstore_last_stats(self._h, s.ctypes.data))
        return dict(scan_ms=s[0], pairs=s[1], keyframes_scanned=int(s[2]), phase2_ms=s[3], evaluated=int(s[4]), reruns=int(s[5]))
