"""CPU oracle (test infrastructure): R2Gen beam search, restated from
modules/att_model.py:98-137 (_sample_beam), modules/caption_model.py:26-202 (beam_search/beam_step,
group_size=1 path), modules/encoder_decoder.py:344-348, 396-404 (_prepare_feature / core: the whole
prefix is re-decoded every step) and modules/utils.py:160-211 (repeat_tensors, penalty '' = identity).
"""
import torch
import torch.nn.functional as F

from . import functional as O


def _step_logprobs(P, ys, memory, src_mask, cfg, ctx):
    out = O.r2_decode(P, memory, src_mask, ys, O.subsequent_mask(ys.shape[1]), cfg, ctx)
    return F.log_softmax(O._lin(P, 'text_decoder.logit', out[:, -1]), dim=1)


def beam_search(P, enc_states, enc_mask, cfg, bos, eos, pad=0, ctx=None, return_scores=False, trace=None, n_near=8):
    """-> seq (B, max_seq_len) int64, padded with `pad` (AttModel._sample_beam with sample_n=1).
    trace: optional list; one dict per step is appended with the search's own decision record -- the `beam + n_near` best
    candidates of every sample in the order of the descending sort (flat index = parent beam * (V+1) + word, score = running sum
    + log-prob, the log-prob itself) and the running sums after the step, before the -1000 penalty of finished beams.  The first
    `beam` entries are the step's selection; the following ones are what a perturbed scorer could have picked instead."""
    ctx = ctx or O.Ctx()
    beam, max_len = cfg['beam_size'], cfg['max_seq_len']
    att, am = enc_states[:, 1:, :], enc_mask[:, 1:]
    bsz = att.shape[0]
    src_mask = am.unsqueeze(-2)
    memory = O.r2_encode(P, O.r2_att_embed(P, att, am, ctx), src_mask, ctx, cfg)

    ys = torch.full((bsz, 1), bos, dtype=torch.long)
    logprobs = _step_logprobs(P, ys, memory, src_mask, cfg, ctx)           # (B, V+1)
    memory = memory.unsqueeze(1).expand(-1, beam, -1, -1).reshape(bsz * beam, *memory.shape[1:])
    src_mask = src_mask.unsqueeze(1).expand(-1, beam, -1, -1).reshape(bsz * beam, *src_mask.shape[1:])

    beam_seq = torch.zeros(bsz, beam, 0, dtype=torch.long)
    beam_sum = torch.zeros(bsz, beam)
    done = [[] for _ in range(bsz)]
    for t in range(max_len):
        v = logprobs.shape[-1]
        lp = logprobs.reshape(bsz, -1, v)
        nb = lp.shape[1]
        cand = (beam_sum[:, :nb].unsqueeze(-1) + lp).reshape(bsz, -1)
        srt, ix = torch.sort(cand, -1, True)
        if trace is not None:
            k = min(beam + n_near, cand.shape[1])
            trace.append({'flat': ix[:, :k].clone(), 'score': srt[:, :k].clone(), 'logp': lp.reshape(bsz, -1).gather(1, ix[:, :k]),
                          'n_beams': nb})
        ix = ix[:, :beam]
        beam_ix, word_ix = ix // v, ix % v
        state_ix = (beam_ix + torch.arange(bsz).unsqueeze(-1) * nb).reshape(-1)
        if t > 0:
            beam_seq = beam_seq.gather(1, beam_ix.unsqueeze(-1).expand_as(beam_seq))
        beam_seq = torch.cat([beam_seq, word_ix.unsqueeze(-1)], -1)
        beam_sum = beam_sum[:, :nb].gather(1, beam_ix) + lp.reshape(bsz, -1).gather(1, ix)
        ys = ys[state_ix]
        for b in range(bsz):
            is_end = beam_seq[b, :, t] == eos
            if t == max_len - 1:
                is_end = torch.ones_like(is_end)
            for vix in range(beam):
                if is_end[vix]:
                    done[b].append({'seq': beam_seq[b, vix].clone(), 'p': beam_sum[b, vix].item()})
            beam_sum[b, is_end] -= 1000
        if t == max_len - 1:
            break
        ys = torch.cat([ys, beam_seq[:, :, t].reshape(-1, 1)], dim=1)
        logprobs = _step_logprobs(P, ys, memory, src_mask, cfg, ctx)

    seq = torch.full((bsz, max_len), pad, dtype=torch.long)
    scores = []
    for b in range(bsz):
        best = sorted(done[b], key=lambda x: -x['p'])[0]
        seq[b, :best['seq'].shape[0]] = best['seq']
        scores.append(best['p'])
    if return_scores:
        return seq, scores
    return seq


def decode_texts(tokenizer, seq, fallback=True):
    """FineTune.text_decoder_forward_r2gen post-processing -- ...v0623_large_res.py:115-123."""
    texts = tokenizer.decode_batch(seq.cpu().tolist())
    if fallback:
        texts = [t if len(t) > 0 else "there is no evidence of pulmonary." for t in texts]
    return texts
