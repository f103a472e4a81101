"""CPU oracle (test infrastructure): the distilgpt2 cross-attention decoder backend.

Reference: models/language_encoder/language_model.py:161-282 (DistilGPT2TextDecoderModel) -- a thin wrapper whose
arithmetic lives in un-vendored HF `transformers` (pinned 4.23.1; container has 5.15.0):
GPT2LMHeadModel(GPT2Config(add_cross_attention=True, is_decoder=True, ...)) inside an EncoderDecoderModel with a dummy
encoder.  The wrapper itself cannot be constructed under transformers 5.15, so this restatement of the
published GPT-2 block is pinned against the in-container HF GPT2LMHeadModel called directly with `encoder_hidden_states`
(tests/golden/make_golden.py, case 'gpt2') -- "parity unpinned" w.r.t. the reference's own pinned version.

Semantics restated: Conv1D weights are (in, out); pre-LN blocks ln_1 -> causal self-attention -> ln_cross_attn ->
cross-attention (q_attn on the decoder stream, c_attn -> K,V on the encoder states, NO encoder mask: the wrapper passes
the mask in a field HF ignores) -> ln_2 -> MLP(gelu_new); tied LM head; training loss = cross_entropy(logits, input_ids,
ignore_index=pad) with UN-shifted labels (language_model.py:252-254)."""
import math

import torch
import torch.nn.functional as F

PRE = 'text_decoder.decoder.encoder_decoder.decoder.'


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _c1d(P, name, x):
    return x @ P[name + '.weight'] + P[name + '.bias']


def _ln(P, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[name + '.weight'], P[name + '.bias'], eps)


def _heads(x, h):
    b, t, d = x.shape
    return x.view(b, t, h, d // h).permute(0, 2, 1, 3)


def _attend(q, k, v, mask_add):
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if mask_add is not None:
        s = s + mask_add
    o = torch.matmul(torch.softmax(s, dim=-1), v)
    b, h, t, dh = o.shape
    return o.permute(0, 2, 1, 3).reshape(b, t, h * dh)


def gpt2_hidden(P, ids, attention_mask, enc, heads, layers, pre=PRE, past=None, pos0=0):
    """-> (final hidden states (B, L, d), per-layer (k, v) self-attention caches).  attention_mask (B, L_total) or None."""
    t = pre + 'transformer.'
    b, l = ids.shape
    x = P[t + 'wte.weight'][ids] + P[t + 'wpe.weight'][pos0:pos0 + l][None]
    d = x.shape[-1]
    total = pos0 + l
    causal = torch.tril(torch.ones(total, total, dtype=torch.bool))[pos0:total, :total]
    madd = torch.zeros(b, 1, l, total).masked_fill(~causal[None, None], torch.finfo(torch.float32).min)
    if attention_mask is not None:
        madd = madd + (1.0 - attention_mask[:, None, None, :total].float()) * torch.finfo(torch.float32).min
    caches = []
    for i in range(layers):
        p = '%sh.%d.' % (t, i)
        hn = _ln(P, p + 'ln_1', x)
        q, k, v = _c1d(P, p + 'attn.c_attn', hn).split(d, dim=2)
        q, k, v = _heads(q, heads), _heads(k, heads), _heads(v, heads)
        if past is not None:
            k, v = torch.cat([past[i][0], k], 2), torch.cat([past[i][1], v], 2)
        caches.append((k, v))
        x = x + _c1d(P, p + 'attn.c_proj', _attend(q, k, v, madd))
        hn = _ln(P, p + 'ln_cross_attn', x)
        q = _heads(_c1d(P, p + 'crossattention.q_attn', hn), heads)
        ck, cv = _c1d(P, p + 'crossattention.c_attn', enc).split(d, dim=2)
        x = x + _c1d(P, p + 'crossattention.c_proj', _attend(q, _heads(ck, heads), _heads(cv, heads), None))
        hn = _ln(P, p + 'ln_2', x)
        x = x + _c1d(P, p + 'mlp.c_proj', gelu_new(_c1d(P, p + 'mlp.c_fc', hn)))
    return _ln(P, t + 'ln_f', x), caches


def gpt2_logits(P, ids, attention_mask, enc, heads, layers, pre=PRE):
    h, _ = gpt2_hidden(P, ids, attention_mask, enc, heads, layers, pre)
    return h @ P[pre + 'transformer.wte.weight'].t()


def gpt2_train_loss(P, ids, attention_mask, enc, heads, layers, pad=0, pre=PRE):
    """language_model.py:244-254: cross_entropy(logits.permute(0,2,1), input_ids, ignore_index=pad) -- labels not shifted."""
    lg = gpt2_logits(P, ids, attention_mask, enc, heads, layers, pre)
    return F.cross_entropy(lg.permute(0, 2, 1), ids, ignore_index=pad)


def gpt2_spec(vocab, d, layers, n_pos=1024, pre=PRE):
    """name -> (shape, kind) in HF GPT2LMHeadModel.state_dict() order (lm_head.weight is tied to wte and not listed)."""
    from collections import OrderedDict
    s = OrderedDict()
    t = pre + 'transformer.'
    s[t + 'wte.weight'] = ((vocab, d), 'emb')
    s[t + 'wpe.weight'] = ((n_pos, d), 'emb')

    def ln(n):
        s[n + '.weight'] = ((d,), 'ln_w')
        s[n + '.bias'] = ((d,), 'ln_b')

    def c1d(n, i, o):
        s[n + '.weight'] = ((i, o), 'c1d_w')
        s[n + '.bias'] = ((o,), 'bias')
    for i in range(layers):
        p = '%sh.%d.' % (t, i)
        ln(p + 'ln_1')
        c1d(p + 'attn.c_attn', d, 3 * d)
        c1d(p + 'attn.c_proj', d, d)
        ln(p + 'ln_2')
        c1d(p + 'crossattention.c_attn', d, 2 * d)
        c1d(p + 'crossattention.q_attn', d, d)
        c1d(p + 'crossattention.c_proj', d, d)
        ln(p + 'ln_cross_attn')
        c1d(p + 'mlp.c_fc', d, 4 * d)
        c1d(p + 'mlp.c_proj', 4 * d, d)
    ln(t + 'ln_f')
    return s


def beam_search(P, enc, heads, layers, num_beams, max_length, bos, eos, pad, pre=PRE):
    """HF `generate(num_beams=..., max_length=..., use_cache=True, length_penalty=1.0, early_stopping=False)` for a
    decoder with encoder_hidden_states (language_model.py:260-281), restated from the classic BeamSearchScorer algorithm:
    2*num_beams candidates per step, EOS candidates among the top num_beams become finished hypotheses scored
    sum_logprobs / generated_length, a batch is done when its worst kept hypothesis beats the best running score /
    generated_length.  Returns (B, <= max_length) ids starting with BOS, padded with `pad`."""
    B = enc.shape[0]
    nb = num_beams
    encx = enc.repeat_interleave(nb, dim=0)
    seqs = torch.full((B * nb, 1), bos, dtype=torch.long)
    scores = torch.zeros(B, nb)
    scores[:, 1:] = -1e9
    scores = scores.view(-1)
    hyps = [[] for _ in range(B)]          # (score, tokens)
    done = [False] * B
    past = None
    cur_len = 1
    V = P[pre + 'transformer.wte.weight'].shape[0]
    while cur_len < max_length:
        inp = seqs if past is None else seqs[:, -1:]
        h, past = gpt2_hidden(P, inp, None, encx, heads, layers, pre, past, 0 if past is None else cur_len - 1)
        lp = F.log_softmax(h[:, -1] @ P[pre + 'transformer.wte.weight'].t(), dim=-1) + scores[:, None]
        top_s, top_i = torch.topk(lp.view(B, nb * V), 2 * nb, dim=1, largest=True, sorted=True)
        nxt_scores = torch.zeros(B, nb)
        nxt_tok = torch.zeros(B, nb, dtype=torch.long)
        nxt_idx = torch.zeros(B, nb, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nxt_tok[b] = pad
                nxt_idx[b] = b * nb
                continue
            k = 0
            for rank in range(2 * nb):
                tok, bi, sc = int(top_i[b, rank]) % V, int(top_i[b, rank]) // V, float(top_s[b, rank])
                if tok == eos:
                    if rank >= nb:
                        continue
                    gen_len = cur_len + 1 - 1                      # generated tokens incl. EOS, excl. the BOS prompt
                    hyps[b].append((sc / gen_len, torch.cat([seqs[b * nb + bi], torch.tensor([eos])])))
                    hyps[b] = sorted(hyps[b], key=lambda x: -x[0])[:nb]
                else:
                    nxt_scores[b, k], nxt_tok[b, k], nxt_idx[b, k] = sc, tok, b * nb + bi
                    k += 1
                if k == nb:
                    break
            if len(hyps[b]) >= nb:
                best_running = float(top_s[b].max()) / (cur_len + 1 - 1)
                if hyps[b][-1][0] >= best_running:
                    done[b] = True
        idx = nxt_idx.view(-1)
        seqs = torch.cat([seqs[idx], nxt_tok.view(-1, 1)], dim=1)
        scores = nxt_scores.view(-1)
        past = [(k_[idx], v_[idx]) for k_, v_ in past]
        cur_len += 1
        if all(done):
            break
    out = []
    for b in range(B):
        if not done[b]:
            for k in range(nb):
                hyps[b].append((float(scores[b * nb + k]) / (cur_len - 1), seqs[b * nb + k]))
        out.append(sorted(hyps[b], key=lambda x: -x[0])[0][1])
    L = min(max(len(o) for o in out), max_length)
    res = torch.full((B, L), pad, dtype=torch.long)
    for b, o in enumerate(out):
        res[b, :min(len(o), L)] = o[:L]
    return res
