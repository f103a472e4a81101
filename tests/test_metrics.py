"""CPU tests: BLEU restatement (oracle/bleu.py) and the product's report-comparison helper against the metric rows of
the reference's released predictions (tests/golden/bleu_*.json.gz = data rows + header rows of test_prediction.csv)."""
import gzip
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('res', ['224x224', '384x384'])
def test_bleu_reproduces_released_metric_rows(res):
    from evoke_amd import metrics
    from oracle import bleu as OB
    d = json.load(gzip.open(os.path.join(GOLDEN, 'bleu_%s.json.gz' % res), 'rt'))
    assert len(d['ground_truth']) == 3852
    want = [d['metrics']['BLEU_%d' % i] for i in range(1, 5)]
    for got in (OB.corpus_bleu(d['ground_truth'], d['generated']), metrics.bleu(d['ground_truth'], d['generated'])):
        for g, w in zip(got, want):
            assert abs(g - w) < 1e-12, (got, want)


def test_token_agreement():
    from evoke_amd import metrics
    s, t = metrics.token_agreement([[1, 2, 3, 0], [4, 5, 0, 0]], [[1, 2, 3, 0], [4, 6, 0, 0]])
    assert s == 0.5 and abs(t - 4 / 5) < 1e-12
