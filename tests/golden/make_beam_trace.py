#!/usr/bin/env python3
"""Decision trace of the L = 100 beam case (BASELINE config 5 shape), written by the CPU ORACLE after it has reproduced the
imported reference's token ids of tests/golden/beam384_b4_L100.npz bit for bit (asserted below): per decoded position and study,
the 12 best candidates of the descending sort of modules/caption_model.py:70-74 (flat index = parent beam * (V+1) + word, running
score, log-probability).  The first `beam` columns are the reference's selection at that position, the rest is what a perturbed
scorer could pick instead -- this is what lets the GPU test follow the reference's search step by step (teacher-forced) and price
every disagreement by the reference's own score margin.

Usage:  python tests/golden/make_beam_trace.py [case ...]      (default beam384_b4_L100: about a minute of CPU; needs no /root/reference)
        python tests/golden/make_beam_trace.py beam224 beam224_b4     (the short goldens: seconds)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import beam as OB, functional as O, spec as S  # noqa: E402
from tests.golden.cases import CASES, make_inputs  # noqa: E402

V = 1444


def drift16(P, x, m, seq, cfg):
    """How far does the REFERENCE's own arithmetic move when its decoder runs on 16-bit operands?  Teacher-forced along the returned
    reports: log-probabilities with every 2-D decoder weight rounded to fp16 and the relational memory rounded to fp16 after every
    token, against the fp32 run -> (T, B) max |difference| over the vocabulary.  The relational memory is a 100-step recurrence
    (modules/encoder_decoder.py:274-300) and, on these untrained weights, an expanding one: a relative perturbation of 1e-6 of its
    weights alone moves the log-probabilities of position 99 by 3e-3, fp16 operands by 0.3-0.5.  This profile is the yardstick of
    the GPU test: an engine with 16-bit operands cannot be closer to the reference than the reference is to itself in 16 bits."""
    h = lambda t: t.half().float()          # noqa: E731
    ids = torch.cat([torch.full((seq.shape[0], 1), V - 2), seq[:, :-1]], 1)
    ones = torch.ones_like(ids)
    base = O.r2_forward_logprobs(P, ids, x, ones, m, cfg, O.Ctx())
    P16 = {k: (h(v) if k.startswith('text_decoder.') and v.dim() == 2 and 'lut' not in k else v) for k, v in P.items()}
    orig = O.rm_step
    O.rm_step = lambda *a, **k: h(orig(*a, **k))
    try:
        low = O.r2_forward_logprobs(P16, ids, x, ones, m, cfg, O.Ctx())
    finally:
        O.rm_step = orig
    return (low - base).abs().max(-1).values.t().contiguous()


def main(name='beam384_b4_L100'):
    case = CASES[name]
    gold = np.load(os.path.join(HERE, name + '.npz'))
    inp = make_inputs(case, V)
    cfg = dict(O.DEFAULT_CFG, max_seq_len=case['max_seq_len'], beam_size=case['beam_size'])
    P = S.procedural_state(S.finetune_spec(V))
    tr = []
    with torch.no_grad():
        x, m = O.finetune_encoder_states(P, inp['images'], inp['patient_ids'], case['B'], inp['inc_ids'], inp['inc_masks'], cfg, O.Ctx())
        seq, p = OB.beam_search(P, x, m, cfg, bos=V - 2, eos=V - 1, pad=0, trace=tr, return_scores=True)
        assert seq.tolist() == gold['eval/seq'].tolist(), 'the oracle no longer reproduces the reference ids: trace not written'
        out = {k: torch.stack([t[k] for t in tr]).numpy() for k in ('flat', 'score', 'logp')}      # (T, B, beam + 8)
        out['best_p'] = np.asarray(p, dtype=np.float64)
        out['drift16'] = drift16(P, x, m, seq, cfg).numpy()
    np.savez_compressed(os.path.join(HERE, name + '_trace.npz'), **out)
    beam = case['beam_size']
    margin = out['score'][1:, :, beam - 1] - out['score'][1:, :, beam]
    print(name, 'trace written:', {k: v.shape for k, v in out.items()})
    pos = [t for t in (10, 30, 50, 70, 90, 99) if t < case['max_seq_len']]
    print('16-bit drift of the reference itself at positions %s:' % pos, out['drift16'][pos].T)
    print('selection margin (beam-th minus next candidate) per study: min', margin.min(0), 'median', np.median(margin, 0),
          'positions below 1e-2:', (margin < 1e-2).sum(0))


if __name__ == '__main__':
    for n in (sys.argv[1:] or ['beam384_b4_L100']):
        main(n)
