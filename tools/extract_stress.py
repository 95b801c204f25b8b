"""Stress of the front end's bit-exactness: many different images (textured, noise, low contrast, half textured) through a
lone extractor twice and through an extractor group; every run of an image must give the same bytes.  Catches rare
intra-workgroup races (the quadtree moves keys in place, the scans double-buffer their wave totals).  GPU box.
    python tools/extract_stress.py [images, default 300]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import swarmmap_amd as S  # noqa: E402
from swarmmap_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w, h, A = 752, 480, 4
rng = np.random.default_rng(11)
solo = S.ORBextractor(1000, 1.2, 8, 20, 7)
exs = [S.ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(A)]
grp = S.ExtractorGroup(exs)
block = torch.empty((A, h, w), dtype=torch.uint8).pin_memory()
view = block.numpy()
bad = 0
for t in range(0, n, A):
    imgs = []
    for a in range(A):
        kind = (t + a) % 4
        if kind == 0:
            im = synth.make_canvas(1000 + t + a, w, h)
        elif kind == 1:
            im = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 2:
            im = (synth.make_canvas(2000 + t + a, w, h).astype(np.float32) * 0.15 + 100).astype(np.uint8)
        else:
            im = synth.make_canvas(3000 + t + a, w, h)
            im[:, w // 2:] = 128
        imgs.append(im)
        view[a] = im
    ref = [tuple(x.copy() for x in solo(im)) for im in imgs]
    again = [tuple(x.copy() for x in solo(im)) for im in imgs]
    grp.submit([view[a] for a in range(A)])
    got = [tuple(x.copy() for x in ex.collect()) for ex in exs]
    for a in range(A):
        for name, other in (("second solo run", again[a]), ("group", got[a])):
            if other[0].tobytes() != ref[a][0].tobytes() or other[1].tobytes() != ref[a][1].tobytes():
                bad += 1
                print("MISMATCH image", t + a, "kind", (t + a) % 4, name, len(ref[a][0]), len(other[0]), flush=True)
print("images", n, "mismatches", bad)
sys.exit(1 if bad else 0)
