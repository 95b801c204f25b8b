"""Python mirror of ORB_SLAM2::ORBextractor (code/include/ORBextractor.h:49-127) over the C ABI.

Same constructor arguments, call operator and getters as the reference class; keypoints come back as a
numpy structured array with cv::KeyPoint's 28-byte layout, descriptors as an (N, 32) uint8 array.
"""
import ctypes as C

import numpy as np

from . import _lib

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

STAGES = ("pyramid", "fast_score", "fast_low", "compact", "describe", "host_wall", "host_enqueue1", "host_wait1",
          "quadtree", "host_phase2", "host_assemble")


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBextractor:
    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0):
        self._lib = _lib.load_library()
        self._h = C.c_void_p()
        cfg = _lib.SoExtractorConfig(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST),
                                     int(minThFAST), int(device))
        _lib.check(self._lib.so_extractor_create(C.byref(cfg), C.byref(self._h)))
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self.scaleFactor = float(scaleFactor)
        self._cap = int(self._lib.so_extractor_capacity(self._h))
        self._kps = np.zeros(self._cap, KP_DTYPE)
        self._desc = np.zeros((self._cap, 32), np.uint8)

    @property
    def quadtree_on_device(self):
        """True once a frame has sized the context and DistributeOctTree runs as a HIP kernel."""
        return bool(self._lib.so_extractor_quadtree_on_device(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_extractor_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # --- ORBextractor::operator() -------------------------------------------------------------
    def __call__(self, image, mask=None):
        """image: (H, W) uint8 numpy array (host).  mask is ignored, as in the reference."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise ValueError("image must be CV_8UC1 (2-D uint8)")  # assert at ORBextractor.cc:754
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        n = C.c_int(0)
        _lib.check(self._lib.so_extractor_run(self._h, _vp(image), image.shape[1], image.shape[0],
                                               image.strides[0], _vp(self._kps), _vp(self._desc), self._cap,
                                               C.byref(n)))
        return self._kps[:n.value].copy(), self._desc[:n.value].copy()

    def run_device(self, d_ptr, width, height, stride):
        """Image already resident in HBM (device pointer as int)."""
        n = C.c_int(0)
        _lib.check(self._lib.so_extractor_run_device(self._h, C.c_void_p(d_ptr), width, height, stride,
                                                      _vp(self._kps), _vp(self._desc), self._cap, C.byref(n)))
        return self._kps[:n.value], self._desc[:n.value]

    # --- asynchronous form: submit frame t+1, match / optimise frame t, collect ---------------------
    def submit(self, image):
        if image.dtype != np.uint8 or image.ndim != 2 or image.strides[1] != 1:
            raise ValueError("image must be a row-contiguous CV_8UC1 array")
        self._inflight = image  # keep the pixels alive until collect
        _lib.check(self._lib.so_extractor_submit(self._h, _vp(image), image.shape[1], image.shape[0], image.strides[0]))

    def submit_device(self, d_ptr, width, height, stride):
        _lib.check(self._lib.so_extractor_submit_device(self._h, C.c_void_p(d_ptr), width, height, stride))

    def collect(self):
        n = C.c_int(0)
        _lib.check(self._lib.so_extractor_collect(self._h, _vp(self._kps), _vp(self._desc), self._cap, C.byref(n)))
        self._inflight = None
        return self._kps[:n.value], self._desc[:n.value]

    # --- getters (ORBextractor.h:60-87) -------------------------------------------------------
    def _tables(self):
        nl = self.nlevels
        arrs = [np.zeros(nl, np.float32) for _ in range(4)] + [np.zeros(nl, np.int32)]
        _lib.check(self._lib.so_extractor_tables(self._h, *[_vp(a) for a in arrs]))
        return arrs

    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def GetFeaturesPerLevel(self):
        return self._tables()[4]

    # --- stage outputs of the last run (parity tests) ------------------------------------------
    def level(self, l):
        w, h = C.c_int(), C.c_int()
        _lib.check(self._lib.so_extractor_level_size(self._h, l, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _lib.check(self._lib.so_extractor_get_level(self._h, l, _vp(out), out.size))
        return out

    def candidates(self, l, cap=10000):
        xs, ys, sc = np.zeros(cap, np.int16), np.zeros(cap, np.int16), np.zeros(cap, np.uint8)
        n = C.c_int(0)
        _lib.check(self._lib.so_extractor_get_candidates(self._h, l, _vp(xs), _vp(ys), _vp(sc), cap, C.byref(n)))
        return xs[:n.value].copy(), ys[:n.value].copy(), sc[:n.value].copy()

    def set_profiling(self, on=True):
        _lib.check(self._lib.so_extractor_set_profiling(self._h, 1 if on else 0))

    def profile(self):
        ms = np.zeros(len(STAGES), np.float32)
        _lib.check(self._lib.so_extractor_get_profile(self._h, _vp(ms)))
        return dict(zip(STAGES, ms.tolist()))


class ExtractorGroup:
    """Several ORBextractors of one GPU whose frames go through ONE chain of launches (so_extractor_group): submit() takes
    one image per member (pinned host memory, tightly packed, all of one size) and every member is collected as usual.
    `frames` (optional DeviceFrame per member) adds the Frame constructors' kernel to the chain (so_dframe_group_submit)."""

    def __init__(self, extractors):
        self._lib = lib = _lib.load_library()
        self.members = list(extractors)
        n = len(self.members)
        lib.so_extractor_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        lib.so_extractor_group_destroy.argtypes = [C.c_void_p]
        lib.so_extractor_group_destroy.restype = None
        lib.so_extractor_group_submit.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        lib.so_dframe_group_submit.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]
        arr = (C.c_void_p * n)(*[e._h for e in self.members])
        self._h = C.c_void_p()
        _lib.check(lib.so_extractor_group_create(arr, n, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.so_extractor_group_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _images(self, images):
        n = len(self.members)
        if len(images) != n:
            raise ValueError("one image per member")
        present = [im for im in images if im is not None]
        if not present:
            raise ValueError("at least one member must take part")
        h, w = present[0].shape
        for im in present:
            if im.dtype != np.uint8 or im.ndim != 2 or im.shape != (h, w) or im.strides != (w, 1):
                raise ValueError("images must be tightly packed CV_8UC1 arrays of one size")
        self._inflight = list(images)
        return (C.c_void_p * n)(*[None if im is None else im.ctypes.data for im in images]), w, h

    def submit(self, images, frames=None):
        """images[i] None: member i sits this chain out (its frames[i], if given, is ignored)."""
        ptrs, w, h = self._images(images)
        if frames is None:
            _lib.check(self._lib.so_extractor_group_submit(self._h, ptrs, w, h, w))
        else:
            fr = (C.c_void_p * len(frames))(*[None if f is None else f._h for f in frames])
            for f, im in zip(frames, images):
                if im is not None:
                    f._inflight = im
            _lib.check(self._lib.so_dframe_group_submit(self._h, fr, ptrs, w, h, w))
