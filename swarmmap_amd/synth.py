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
