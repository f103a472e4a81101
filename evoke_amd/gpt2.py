"""distilgpt2 cross-attention decoder backend on the HIP engine -- mirror of DistilGPT2TextDecoderModel
(models/language_encoder/language_model.py:161-282).

The reference class wraps HF GPT2LMHeadModel(add_cross_attention=True) in an EncoderDecoderModel with a dummy encoder; its
state_dict keys are `decoder.encoder_decoder.decoder.{transformer.{wte,wpe,h.<i>.*,ln_f},lm_head}.*` (HF Conv1D weights are
(in, out)).  Same constructor `(config, tokenizer)`, same `forward(encoder_hidden_states, encoder_attention_mask,
input_ids, attention_mask, stage)`; `stage='train'` returns the un-shifted cross-entropy of lines 252-254, `stage='test'`
HF-style beam search (2*num_beams candidates, length-normalised finished hypotheses) with K/V caches.
Selected in FineTune by args['text_decoder'] = 'distilgpt2' (the key exists only as a comment in modules/utils.py:78)."""
import math

import torch
import torch.nn as nn

from . import hip as H
from . import ops
from .layers import EmbP, LayerNormP
from .ops import BF16, F32


class Conv1DP(nn.Module):
    """transformers.pytorch_utils.Conv1D parameter holder: weight (in, out), bias (out)."""

    def __init__(self, nf, nx):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(nx, nf) * 0.02)
        self.bias = nn.Parameter(torch.zeros(nf))

    def forward(self, x, resid=None):
        return ops.linear_t(x, self.weight, self.bias, resid=resid)


class _Attn(nn.Module):
    def __init__(self, d, cross):
        super().__init__()
        if cross:
            self.c_attn = Conv1DP(2 * d, d)
            self.q_attn = Conv1DP(d, d)
        else:
            self.c_attn = Conv1DP(3 * d, d)
        self.c_proj = Conv1DP(d, d)


class _MLP(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = Conv1DP(4 * d, d)
        self.c_proj = Conv1DP(d, 4 * d)


class GPT2Block(nn.Module):
    def __init__(self, d, heads, eps=1e-5, p_attn=0.1, p_resid=0.1):
        super().__init__()
        self.ln_1 = LayerNormP(d, eps)
        self.attn = _Attn(d, False)
        self.ln_2 = LayerNormP(d, eps)
        self.crossattention = _Attn(d, True)
        self.ln_cross_attn = LayerNormP(d, eps)
        self.mlp = _MLP(d)
        self.d, self.heads, self.p_attn, self.p_resid = d, heads, p_attn, p_resid

    def _proj(self, conv, a, x):
        if self.training and self.p_resid > 0 and ops.DROPOUT_ENABLED[0]:
            return ops.dropout(conv(a), self.p_resid, True, resid=x)
        return conv(a, resid=x)

    def forward(self, x, enc, key_mask, cache=None, pos=None):
        d, tr = self.d, self.training
        q, k, v = [t.contiguous() for t in self.attn.c_attn(self.ln_1(x)).split(d, dim=2)]
        if cache is not None:                                    # incremental decode: append to the K/V cache
            cache['k'][:, pos:pos + 1], cache['v'][:, pos:pos + 1] = k, v
            k, v = cache['k'][:, :pos + 1].contiguous(), cache['v'][:, :pos + 1].contiguous()
            a = ops.attention(q, k, v, self.heads)
        else:
            a = ops.attention(q, k, v, self.heads, mask=key_mask, causal=True, p_drop=self.p_attn, training=tr)
        x = self._proj(self.attn.c_proj, a, x)
        q = self.crossattention.q_attn(self.ln_cross_attn(x))
        if cache is not None and 'ck' in cache:
            ck, cv = cache['ck'], cache['cv']
        else:
            ck, cv = [t.contiguous() for t in self.crossattention.c_attn(enc).split(d, dim=2)]
            if cache is not None:
                cache['ck'], cache['cv'] = ck, cv
        a = ops.attention(q, ck, cv, self.heads, p_drop=self.p_attn, training=tr)
        x = self._proj(self.crossattention.c_proj, a, x)
        f = ops.activation(self.mlp.c_fc(self.ln_2(x)), H.ACT_GELU_NEW)
        return self._proj(self.mlp.c_proj, f, x)


class _Transformer(nn.Module):
    def __init__(self, vocab, d, layers, heads, n_pos):
        super().__init__()
        self.wte = EmbP(vocab, d)
        self.wpe = EmbP(n_pos, d)
        self.h = nn.ModuleList([GPT2Block(d, heads) for _ in range(layers)])
        self.ln_f = LayerNormP(d, 1e-5)


class _LMHead(nn.Module):
    def __init__(self, weight):
        super().__init__()
        self.weight = weight                      # tied to transformer.wte.weight (same Parameter object)


class _GPT2LMHead(nn.Module):
    def __init__(self, vocab, d, layers, heads, n_pos):
        super().__init__()
        self.transformer = _Transformer(vocab, d, layers, heads, n_pos)
        self.lm_head = _LMHead(self.transformer.wte.weight)


class _EncDec(nn.Module):
    def __init__(self, dec):
        super().__init__()
        self.decoder = dec


class _Holder(nn.Module):
    def __init__(self, dec):
        super().__init__()
        self.encoder_decoder = _EncDec(dec)


class DistilGPT2TextDecoderModel(nn.Module):
    def __init__(self, config, tokenizer):
        super().__init__()
        d = config['decoder_hidden_size']
        self.vocab = config['vocab_size']
        self.heads = config.get('decoder_num_attention_heads', 12)
        self.layers = config['decoder_num_hidden_layers']
        self.decoder = _Holder(_GPT2LMHead(self.vocab, d, self.layers, self.heads, config.get('n_positions', 1024)))
        self.pad_token_id = tokenizer.token_to_id('[PAD]')
        self.eos_token_id = tokenizer.token_to_id('[EOS]')
        self.bos_token_id = tokenizer.token_to_id('[BOS]')
        self.beam_size = config['beam_size']
        self.max_seq_len = config['max_seq_len']
        self.p_embd = 0.1

    @property
    def gpt2(self):
        return self.decoder.encoder_decoder.decoder

    def _hidden(self, ids, attention_mask, enc, caches=None, pos=None):
        t = self.gpt2.transformer
        pe = t.wpe.weight if pos is None else t.wpe.weight[pos:pos + 1]
        x = ops.dropout(ops.embedding(ids.contiguous(), t.wte.weight, pos=pe), self.p_embd, self.training)
        km = attention_mask.to(torch.uint8).contiguous() if attention_mask is not None else None
        for i, blk in enumerate(t.h):
            x = blk(x, enc, km, None if caches is None else caches[i], pos)
        return t.ln_f(x)

    def logits(self, ids, attention_mask, enc, caches=None, pos=None):
        """f32 logits (B, L, pad8(V)) through the tied head."""
        return ops.linear(self._hidden(ids, attention_mask, enc, caches, pos), self.gpt2.lm_head.weight, None, out_f32=True)

    def forward(self, encoder_hidden_states, encoder_attention_mask, input_ids=None, attention_mask=None, stage='train'):
        assert stage in ['train', 'test']
        if stage == 'train':
            lg = self.logits(input_ids, attention_mask, encoder_hidden_states)
            w = (input_ids != self.pad_token_id).to(F32)
            return ops.nll_loss(lg, input_ids.reshape(-1), w.reshape(-1), self.vocab)
        return self.generate(encoder_hidden_states)

    @torch.no_grad()
    def generate(self, enc):
        """HF beam search (num_beams = beam_size, max_length = max_seq_len, length_penalty 1, early_stopping False)."""
        was_training = self.training
        self.eval()
        try:
            nb, max_len, V = self.beam_size, self.max_seq_len, self.vocab
            if 2 * nb > 8:
                raise ValueError('beam_size must be <= 4 for the distilgpt2 backend')
            B, dev, d = enc.shape[0], enc.device, enc.shape[-1]
            encx = enc.repeat_interleave(nb, dim=0).contiguous()
            R = B * nb
            caches = [dict(k=torch.zeros(R, max_len, d, dtype=BF16, device=dev), v=torch.zeros(R, max_len, d, dtype=BF16, device=dev))
                      for _ in range(self.layers)]
            seqs = torch.full((R, 1), self.bos_token_id, dtype=torch.long, device=dev)
            scores = torch.zeros(B, nb, dtype=F32, device=dev)
            scores[:, 1:] = -1e9
            scores = scores.view(-1)
            hyps = [[] for _ in range(B)]
            done = [False] * B
            cur_len = 1
            while cur_len < max_len:
                lg = self.logits(seqs[:, -1:].contiguous(), None, encx, caches, cur_len - 1)
                lp = ops.log_softmax(lg.view(R, -1), V) + scores[:, None]
                top_s, top_i = torch.empty(B, 2 * nb, dtype=F32, device=dev), torch.empty(B, 2 * nb, dtype=torch.long, device=dev)
                H.check(H.lib.evk_topk_rows(H.ptr(lp.view(B, nb * V).contiguous()), H.ptr(top_s), H.ptr(top_i), B, nb * V, 2 * nb, H.stream()))
                ts, ti = top_s.cpu(), top_i.cpu()
                nxt_scores = torch.zeros(B, nb)
                nxt_tok = torch.zeros(B, nb, dtype=torch.long)
                nxt_idx = torch.zeros(B, nb, dtype=torch.long)
                seq_cpu = None
                for b in range(B):
                    if done[b]:
                        nxt_tok[b] = self.pad_token_id
                        nxt_idx[b] = b * nb
                        continue
                    k = 0
                    for rank in range(2 * nb):
                        tok, bi, sc = int(ti[b, rank]) % V, int(ti[b, rank]) // V, float(ts[b, rank])
                        if tok == self.eos_token_id:
                            if rank >= nb:
                                continue
                            if seq_cpu is None:
                                seq_cpu = seqs.cpu()
                            hyps[b].append((sc / cur_len, torch.cat([seq_cpu[b * nb + bi], torch.tensor([tok])])))
                            hyps[b] = sorted(hyps[b], key=lambda x: -x[0])[:nb]
                        else:
                            nxt_scores[b, k], nxt_tok[b, k], nxt_idx[b, k] = sc, tok, b * nb + bi
                            k += 1
                        if k == nb:
                            break
                    if len(hyps[b]) >= nb and hyps[b][-1][0] >= float(ts[b].max()) / cur_len:
                        done[b] = True
                idx = nxt_idx.view(-1).to(dev)
                seqs = torch.cat([seqs.index_select(0, idx), nxt_tok.view(-1, 1).to(dev)], dim=1)
                scores = nxt_scores.view(-1).to(dev)
                for c in caches:
                    c['k'][:, :cur_len] = c['k'][:, :cur_len].index_select(0, idx)
                    c['v'][:, :cur_len] = c['v'][:, :cur_len].index_select(0, idx)
                cur_len += 1
                if all(done):
                    break
            seq_cpu, sc_cpu = seqs.cpu(), scores.cpu()
            out = []
            for b in range(B):
                if not done[b]:
                    for k in range(nb):
                        hyps[b].append((float(sc_cpu[b * nb + k]) / (cur_len - 1), seq_cpu[b * nb + k]))
                out.append(sorted(hyps[b], key=lambda x: -x[0])[0][1])
            L = min(max(len(o) for o in out), max_len)
            res = torch.full((B, L), self.pad_token_id, dtype=torch.long)
            for b, o in enumerate(out):
                res[b, :min(len(o), L)] = o[:L]
            return res.to(dev)
        finally:
            self.train(was_training)
