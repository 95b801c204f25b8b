"""Seeded synthetic inputs: there are no datasets on the build or GPU boxes.

Front-end streams: a large textured canvas (value-noise background, contrast rectangles, small
blobs) viewed through a window that moves along a Lissajous path — cheap to render, rich in FAST
corners at every pyramid level, deterministic from the seed.
"""
import numpy as np

EUROC = (752, 480)   # code/Examples/Monocular/EuRoC.yaml
KITTI = (1241, 376)  # code/Examples/Monocular/KITTI00-02.yaml


def _value_noise(rng, h, w, octaves=4, amplitude=40.0):
    out = np.zeros((h, w), np.float32)
    for o in range(octaves):
        cells = 4 << o
        g = rng.standard_normal((cells + 2, cells + 2)).astype(np.float32)
        ys = np.linspace(0, cells, h, endpoint=False)
        xs = np.linspace(0, cells, w, endpoint=False)
        y0 = ys.astype(int); x0 = xs.astype(int)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        out += (amplitude / (1 << o)) * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    return out


def make_canvas(seed, w, h, n_rect=None, n_blob=None):
    """A (h, w) uint8 textured canvas."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = 110.0 + _value_noise(rng, h, w)
    n_rect = n_rect if n_rect is not None else (w * h) // 2500
    n_blob = n_blob if n_blob is not None else (w * h) // 600
    for _ in range(n_rect):
        rw, rh = rng.integers(8, 90, 2)
        x, y = rng.integers(0, w - 8), rng.integers(0, h - 8)
        img[y:y + rh, x:x + rw] += rng.uniform(-70, 70)
    for _ in range(n_blob):
        s = rng.integers(2, 7)
        x, y = rng.integers(0, w - 8), rng.integers(0, h - 8)
        img[y:y + s, x:x + s] += rng.choice([-1.0, 1.0]) * rng.uniform(25, 90)
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_image(seed, size=EUROC):
    return make_canvas(seed, size[0], size[1])


class FrameStream:
    """Deterministic stream of (h, w) uint8 frames: a window sliding over a canvas."""

    def __init__(self, seed=20221001, size=EUROC, margin=160):
        self.w, self.h = size
        self.margin = margin
        self.canvas = make_canvas(seed, self.w + 2 * margin, self.h + 2 * margin)

    def frame(self, t):
        m = self.margin
        ox = int(round(m + (m - 1) * np.sin(0.013 * t)))
        oy = int(round(m + (m - 1) * np.sin(0.021 * t + 0.5)))
        return np.ascontiguousarray(self.canvas[oy:oy + self.h, ox:ox + self.w])


# ------------------------------------------------------------------------------------------------
# matcher inputs: descriptors = base descriptors with Binomial(256, p) flipped bits,
# geometry = jittered keypoint positions.  Everything is returned as plain numpy arrays.
# ------------------------------------------------------------------------------------------------
SCALE_FACTORS = np.array([1.2 ** i for i in range(8)], np.float32)


def flip_bits(rng, desc, p):
    """Flip each of the 256 bits of every 32-byte descriptor with probability p."""
    mask = np.packbits(rng.random((desc.shape[0], 256)) < p, axis=1)
    return np.bitwise_xor(desc, mask)


def make_frame_arrays(rng, n, size=EUROC, levels=8, distort_margin=6.0, dup_frac=0.02):
    """Random undistorted keypoints (a few fall slightly outside the image like real undistorted points)."""
    w, h = size
    x = rng.uniform(-distort_margin, w + distort_margin, n).astype(np.float32)
    y = rng.uniform(-distort_margin, h + distort_margin, n).astype(np.float32)
    probs = (1 / 1.2) ** np.arange(levels)
    octave = rng.choice(levels, n, p=probs / probs.sum()).astype(np.int32)
    angle = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    ndup = int(n * dup_frac)  # duplicated descriptors -> equal distances -> exercises the tie-break order
    if ndup:
        src = rng.integers(0, n, ndup)
        dst = rng.integers(0, n, ndup)
        desc[dst] = desc[src]
    bounds = (0.0, float(w), 0.0, float(h))  # mnMinX, mnMaxX, mnMinY, mnMaxY (no distortion)
    return dict(x=x, y=y, octave=octave, angle=angle, desc=desc, bounds=bounds, scale_factors=SCALE_FACTORS)


def make_m1_case(seed, n_kp=1000, n_mp=2000, p_flip=0.15, size=EUROC, jitter=2.0, prebound_frac=0.1):
    """Frame + local map points for SearchByProjection(Frame&, vector<MapPoint*>&, th)."""
    rng = np.random.default_rng(seed)
    fr = make_frame_arrays(rng, n_kp, size)
    k = rng.integers(0, n_kp, n_mp)
    outlier = rng.random(n_mp) < 0.25
    proj_x = (fr["x"][k] + rng.normal(0, jitter, n_mp)).astype(np.float32)
    proj_y = (fr["y"][k] + rng.normal(0, jitter, n_mp)).astype(np.float32)
    proj_x[outlier] = rng.uniform(0, size[0], outlier.sum()).astype(np.float32)
    proj_y[outlier] = rng.uniform(0, size[1], outlier.sum()).astype(np.float32)
    desc = flip_bits(rng, fr["desc"][k], p_flip)
    desc[outlier] = rng.integers(0, 256, (int(outlier.sum()), 32)).astype(np.uint8)
    pred_level = np.clip(fr["octave"][k] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    view_cos = np.where(rng.random(n_mp) < 0.5, 0.9995, 0.9).astype(np.float32)
    mps = dict(in_view=(rng.random(n_mp) < 0.9).astype(np.uint8), proj_x=proj_x, proj_y=proj_y, view_cos=view_cos,
               pred_level=pred_level, desc=desc, has_obs=(rng.random(n_mp) < 0.97).astype(np.uint8))
    fr["excluded"] = (rng.random(n_kp) < prebound_frac).astype(np.uint8)
    return fr, mps


def make_m2_case(seed, n_kp=1000, n_last=1000, p_flip=0.1, size=EUROC, jitter=3.0):
    """Current frame + last frame's projected map points for SearchByProjection(cur, last, th, bMono)."""
    rng = np.random.default_rng(seed)
    fr = make_frame_arrays(rng, n_kp, size)
    k = rng.integers(0, n_kp, n_last)
    u = (fr["x"][k] + rng.normal(0, jitter, n_last)).astype(np.float32)
    v = (fr["y"][k] + rng.normal(0, jitter, n_last)).astype(np.float32)
    octave = np.clip(fr["octave"][k] + rng.integers(-1, 2, n_last), 0, 7).astype(np.int32)
    rot = rng.choice([3.0, 40.0, 200.0], n_last, p=[0.8, 0.15, 0.05])  # a dominant rotation + outliers
    angle = ((fr["angle"][k] + rot + rng.normal(0, 2, n_last)) % 360).astype(np.float32)
    last = dict(valid=(rng.random(n_last) < 0.6).astype(np.uint8), u=u, v=v, octave=octave, angle=angle,
                desc=flip_bits(rng, fr["desc"][k], p_flip), has_obs=(rng.random(n_last) < 0.97).astype(np.uint8))
    fr["excluded"] = (rng.random(n_kp) < 0.05).astype(np.uint8)
    return fr, last


def make_m4_case(seed, n_kp=2000, p_flip=0.05, size=EUROC, shift=(12.0, -7.0)):
    """Two frames for SearchForInitialization (F2 = F1 moved by `shift` + noise, partly re-ordered)."""
    rng = np.random.default_rng(seed)
    f1 = make_frame_arrays(rng, n_kp, size, dup_frac=0.01)
    perm = rng.permutation(n_kp)
    f2 = dict(x=(f1["x"][perm] + shift[0] + rng.normal(0, 0.7, n_kp)).astype(np.float32),
              y=(f1["y"][perm] + shift[1] + rng.normal(0, 0.7, n_kp)).astype(np.float32),
              octave=f1["octave"][perm].copy(),
              angle=((f1["angle"][perm] + 5 + rng.normal(0, 3, n_kp)) % 360).astype(np.float32),
              desc=flip_bits(rng, f1["desc"][perm], p_flip), bounds=f1["bounds"],
              scale_factors=f1["scale_factors"])
    prev = np.stack([f1["x"], f1["y"]], 1).astype(np.float32)
    return f1, f2, prev
