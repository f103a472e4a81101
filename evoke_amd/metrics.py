"""Report comparison helpers: corpus BLEU-1..4 in pycocoevalcap's convention (the metric the
reference reports for its generated reports, modules/metrics/metrics.py:17-94) and token-exact sequence agreement.
CPU text utilities -- not part of the GPU hot path."""
import math
from collections import Counter


def bleu(references, hypotheses, max_n=4):
    """Whitespace tokens, one reference per hypothesis, clipped counts, corpus-level brevity penalty."""
    total = [0] * max_n
    match = [0] * max_n
    len_h = len_r = 0
    for ref, hyp in zip(references, hypotheses):
        r, h = ref.split(), hyp.split()
        len_h += len(h)
        len_r += len(r)
        for n in range(1, max_n + 1):
            hc = Counter(tuple(h[i:i + n]) for i in range(len(h) - n + 1))
            rc = Counter(tuple(r[i:i + n]) for i in range(len(r) - n + 1))
            total[n - 1] += max(0, len(h) - n + 1)
            match[n - 1] += sum(min(c, rc[g]) for g, c in hc.items() if g in rc)
    out, prod = [], 1.0
    for n in range(max_n):
        prod *= (match[n] + 1e-15) / (total[n] + 1e-9)
        out.append(prod ** (1.0 / (n + 1)))
    ratio = (len_h + 1e-15) / (len_r + 1e-9)
    if ratio < 1:
        out = [x * math.exp(1 - 1 / ratio) for x in out]
    return out


def token_agreement(seq_a, seq_b, pad=0):
    """Fraction of identical sequences and of identical non-pad token positions between two (B, L) id tensors/lists."""
    same_seq = same_tok = n_tok = 0
    for a, b in zip(seq_a, seq_b):
        a, b = list(a), list(b)
        same_seq += int(a == b)
        for x, y in zip(a, b):
            if x != pad or y != pad:
                n_tok += 1
                same_tok += int(x == y)
    return same_seq / max(1, len(seq_a)), same_tok / max(1, n_tok)
