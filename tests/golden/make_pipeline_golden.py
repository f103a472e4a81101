"""Golden vectors of the input pipeline: Pillow (the library torchvision's transforms call) applied to seeded uint8
images through oracle.pipeline.transform_pil.  Run here (Pillow 12.2.0): python tests/golden/make_pipeline_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pipeline as P  # noqa: E402

CASES = [  # name, (h, w, channels), resize, (top, left) fractions, S, flip, angle
    ('train384_like', (150, 121, 3), 112, (0.5, 0.25), 96, False, 3.7),
    ('eval384_like', (130, 170, 3), 112, None, 96, False, None),
    ('train224_like', (90, 140, 3), 64, (0.3, 0.9), 56, True, None),
    ('eval224_like', (77, 101, 3), (56, 56), None, 56, False, None),
    ('grey_rot_neg', (128, 96, 1), 112, (0.0, 1.0), 96, False, -4.99),
    ('upscale', (40, 50, 3), 64, (1.0, 0.0), 48, False, 0.8),
]


def case_input(name, shape):
    rng = np.random.default_rng(abs(hash(name)) % 2 ** 31 if False else sum(ord(c) for c in name))
    h, w, c = shape
    base = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((np.sin(yy / 7.0) + np.cos(xx / 5.0)) * 60 + 128).clip(0, 255).astype(np.uint8)
    img = ((base.astype(np.int32) + smooth[:, :, None].astype(np.int32) * 3) // 4).astype(np.uint8)
    return img[:, :, 0] if c == 1 else img


def case_params(shape, resize, frac, S):
    h, w = shape[:2]
    rw, rh = P.resized_size(w, h, resize)
    if frac is None:
        top, left = P.center_crop_origin(rw, rh, S)
    else:
        top, left = int(round(frac[0] * (rh - S))), int(round(frac[1] * (rw - S)))
    return rw, rh, top, left


if __name__ == '__main__':
    out = {}
    for name, shape, resize, frac, S, flip, angle in CASES:
        img = case_input(name, shape)
        rw, rh, top, left = case_params(shape, resize, frac, S)
        y = P.transform_pil(img, resize, top, left, S, flip, angle)
        out[name] = y
        print(name, img.shape, (rw, rh), (top, left), y.shape, float(y.mean()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pipeline.npz'), **out)
