"""CPU oracle (test infrastructure): corpus BLEU-1..4 as computed by pycocoevalcap (the metric behind the header rows
of the reference's released predictions, modules/metrics/metrics.py:17-94 -> pycocoevalcap.bleu, not installed here).

Restated from the published algorithm (pycocoevalcap bleu_scorer.py, option 'closest', single reference): whitespace
tokens, clipped n-gram counts, corpus-level precision with tiny = 1e-15 / small = 1e-9 guards, brevity penalty
exp(1 - ref_len / test_len) when test_len < ref_len.  Pinned by the 12 metric rows of
generated_reports/MIMIC-CXR/resolution-{224x224,384x384}/test_prediction.csv (tests/golden/bleu_*.json.gz)."""
import math
from collections import Counter


def _ngrams(words, n=4):
    c = Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            c[tuple(words[i:i + k])] += 1
    return c


def corpus_bleu(references, hypotheses, n=4):
    """references / hypotheses: lists of strings (one reference per hypothesis) -> [BLEU_1 .. BLEU_n]."""
    tiny, small = 1e-15, 1e-9
    guess = [0] * n
    correct = [0] * n
    testlen = reflen = 0
    for ref, hyp in zip(references, hypotheses):
        r, h = ref.split(), hyp.split()
        testlen += len(h)
        reflen += len(r)
        rc, hc = _ngrams(r, n), _ngrams(h, n)
        for k in range(n):
            guess[k] += max(0, len(h) - k)
        for ng, cnt in hc.items():
            correct[len(ng) - 1] += min(cnt, rc.get(ng, 0))
    bleus = []
    b = 1.0
    for k in range(n):
        b *= (float(correct[k]) + tiny) / (float(guess[k]) + small)
        bleus.append(b ** (1.0 / (k + 1)))
    ratio = (testlen + tiny) / (reflen + small)
    if ratio < 1:
        bleus = [x * math.exp(1 - 1 / ratio) for x in bleus]
    return bleus
