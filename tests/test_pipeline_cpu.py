"""Input pipeline, CPU side: the oracle's numpy restatement of Pillow's integer resampling / rotation is pinned against
Pillow and the committed golden vectors; the host logic of evoke_amd/pipeline.py (size rule, crop / rotation parameters,
collate contracts) is checked against the oracle restatement of the reference's collate functions."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline as P
from tests.golden.make_pipeline_golden import CASES, case_input, case_params

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pipeline.npz'))


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_numpy_restatement_matches_golden_and_pillow(case):
    name, shape, resize, frac, S, flip, angle = case
    img = case_input(name, shape)
    rw, rh, top, left = case_params(shape, resize, frac, S)
    mine = P.transform_numpy(img, resize, top, left, S, flip, angle)
    assert np.array_equal(mine, GOLD[name])
    assert np.array_equal(P.transform_pil(img, resize, top, left, S, flip, angle), GOLD[name])


def test_resized_size_rule():
    assert P.resized_size(2544, 3056, 448) == (448, 538)       # portrait: width is the short side
    assert P.resized_size(3056, 2544, 448) == (538, 448)
    assert P.resized_size(500, 500, 256) == (256, 256)
    assert P.resized_size(300, 200, (224, 224)) == (224, 224)
    assert P.center_crop_origin(538, 448, 384) == (32, 77)


def test_host_mirror_matches_oracle_host_logic():
    from evoke_amd import pipeline as M
    for w, h, size in [(2544, 3056, 448), (3056, 2544, 448), (1000, 999, 256), (300, 200, (224, 224))]:
        assert M.resized_size(w, h, size) == P.resized_size(w, h, size)
    for ang in (3.7, -4.99, 0.0, 360.0, 1e-3):
        assert M.rotate_affine_fixed(ang, 384, 384) == P.rotate_affine_fixed(ang, 384, 384)
    batch = ['files/p10/s1/a.jpg', 'files/p11/s7/b.jpg', 'files/p10/s2/c.jpg']
    mv = [['files/p10/s1/a.jpg', 'files/p10/s1/a2.jpg'], ['files/p11/s7/b.jpg'], ['files/p10/s2/c.jpg', 'files/p10/s2/c2.jpg', 'files/p10/s1/a2.jpg']]
    paths, pids = M.collate_order(batch, mv, True)
    assert (paths, list(pids)) == P.collate_order(batch, mv, True)
    assert paths == batch + ['files/p10/s1/a2.jpg', 'files/p10/s2/c2.jpg']
    assert list(pids) == ['p10_s1', 'p11_s7', 'p10_s2', 'p10_s1', 'p10_s2']
    assert M.collate_order(batch, mv, False)[0] == batch
    with pytest.raises(AssertionError):
        M.collate_order(['a/b/c.jpg'], [[]])

    ids = ['s1', 's2', 's3', 's4']
    vps = [['LATERAL', 'PA', 'AP'], ['unk', 'unk'], ['LATERAL', 'LL'], ['LL', 'XTABLE', 'LATERAL']]
    imgs = [['1a', '1b', '1c'], ['2a', '2b'], ['3a', '3b'], ['4a', '4b', '4c']]
    for seed in range(4):
        r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
        got = M.multiview_collate_order(ids, vps, imgs, r1.randint)
        want = P.multiview_collate_order(ids, vps, imgs, r2.randint)
        assert (got[0], list(got[1])) == want
        assert got[0][1] == '2a' and got[0][3] == '4b' and got[0][0] in ('1b', '1c') and len(got[0]) == 10
        assert list(got[1][:4]) == ids

    a, m = M.pad_tokens([[5, 6, 7], [8]], [[1, 1, 1], [1]])
    oa, om = P.pad_tokens([[5, 6, 7], [8]], [[1, 1, 1], [1]])
    assert a.dtype == torch.int64 and np.array_equal(a.numpy(), oa) and np.array_equal(m.numpy(), om)


def test_transform_parameter_draws():
    from evoke_amd import pipeline as M
    g = torch.Generator().manual_seed(3)
    t = M.Transform.for_model(384, 'train', g)
    for _ in range(20):
        p = t.params(2544, 3056)
        assert (p['resize_w'], p['resize_h']) == (448, 538) and 0 <= p['crop_top'] <= 154 and 0 <= p['crop_left'] <= 64
        assert -5.0 <= p['angle'] <= 5.0 and not p['flip']
    e = M.Transform.for_model(384, 'test').params(3056, 2544)
    assert (e['crop_top'], e['crop_left'], e['angle']) == (32, 77, None)
    e = M.Transform.for_model(224, 'val').params(3056, 2544)
    assert (e['resize_w'], e['resize_h'], e['crop_top'], e['crop_left']) == (224, 224, 0, 0)
    tr = M.Transform.for_model(224, 'train', g)
    flips = [tr.params(1000, 800)['flip'] for _ in range(40)]
    assert any(flips) and not all(flips)
    with pytest.raises(ValueError):
        M.Transform(64, 96, True).params(100, 100)
