"""Input pipeline on the GPU: evk_preprocess_image against the Pillow-generated golden vectors (bit-exact) and, at the real
384 / 224 model sizes, against the oracle's numpy restatement of Pillow on a seeded chest-film-sized image."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_pipeline_golden import CASES, case_input, case_params

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pipeline.npz'))


def _run(img, rw, rh, top, left, S, flip, angle):
    from evoke_amd import pipeline as M
    out = torch.empty(3, S, S, dtype=torch.float32, device='cuda')
    prm = dict(resize_w=rw, resize_h=rh, crop_top=top, crop_left=left, out_size=S, flip=flip, angle=angle)
    M.preprocess_into(out, torch.from_numpy(np.ascontiguousarray(img)).cuda(), prm)
    return out.cpu().numpy()


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_preprocess_matches_pillow_golden(case):
    name, shape, resize, frac, S, flip, angle = case
    img = case_input(name, shape)
    rw, rh, top, left = case_params(shape, resize, frac, S)
    got = _run(img, rw, rh, top, left, S, flip, angle)
    assert np.array_equal(got, GOLD[name]), 'max |d| = %g' % np.abs(got - GOLD[name]).max()


@pytest.mark.parametrize('res,split', [(384, 'train'), (384, 'test'), (224, 'train'), (224, 'val')])
def test_preprocess_full_size_matches_oracle(res, split):
    from evoke_amd import pipeline as M
    from oracle import pipeline as P
    rng = np.random.default_rng(res + len(split))
    h, w = 1210, 1004                                  # a down-scaled portrait chest film keeps the oracle fast
    yy, xx = np.mgrid[0:h, 0:w]
    img = ((np.sin(yy / 37.0) * np.cos(xx / 23.0) * 90 + 128) + rng.integers(-20, 20, (h, w))).clip(0, 255).astype(np.uint8)
    t = M.Transform.for_model(res, split, torch.Generator().manual_seed(11))
    batch, prms = M.preprocess_batch([img, np.repeat(img[:, :, None], 3, axis=2)], t)
    assert batch.shape == (2, 3, res, res)
    for i, prm in enumerate(prms):
        want = P.transform_numpy(img, t.resize, prm['crop_top'], prm['crop_left'], res, prm['flip'], prm['angle'])
        assert np.array_equal(batch[i].cpu().numpy(), want)


def test_preprocess_rejects_bad_window():
    from evoke_amd import pipeline as M
    out = torch.empty(3, 64, 64, dtype=torch.float32, device='cuda')
    img = torch.zeros(80, 80, 3, dtype=torch.uint8, device='cuda')
    with pytest.raises(RuntimeError):
        M.preprocess_into(out, img, dict(resize_w=70, resize_h=70, crop_top=10, crop_left=0, out_size=64, flip=False, angle=None))
