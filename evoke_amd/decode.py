"""Beam-search report generation on the HIP engine.

Semantics = AttModel._sample_beam (modules/att_model.py:98-137) + CaptionModel.beam_search / beam_step
(modules/caption_model.py:26-202, group_size = 1) + EncoderDecoder.core (modules/encoder_decoder.py:396-404).
The reference re-decodes the WHOLE prefix at every step (O(T^2) decoder work, O(T^2) Python-level relational-memory
steps) and walks a per-sample Python loop with .item() syncs; this engine is mathematically identical but
incremental: per-layer self-attention K/V caches, cross-attention K/V projected once, the relational memory carried
as state, the per-position conditional-LayerNorm inputs taken from the current memory, and all beam bookkeeping
(segmented top-k, EOS handling, the -1000 penalty, best-finished-beam tracking) done on the device.
"""
import math

import torch

from . import hip as H
from . import ops
from .ops import BF16, F32


def _topk(x, k):
    rows, n = x.shape
    vals = torch.empty(rows, k, dtype=F32, device=x.device)
    idx = torch.empty(rows, k, dtype=torch.long, device=x.device)
    H.check(H.lib.evk_topk_rows(H.ptr(x), H.ptr(vals), H.ptr(idx), rows, n, k, H.stream()), 'topk_rows')
    return vals, idx


class _DecoderState:
    """Incremental state of R = batch*beam hypotheses."""

    def __init__(self, dec, enc, src_mask, max_len):
        model = dec.model
        self.dec, self.model = dec, model
        R, d = enc.shape[0], model.d_model
        self.enc, self.src_mask = enc, src_mask
        self.mem = model.rm.init_memory(R, enc.device)
        self.t = 0
        self.kc, self.vc, self.ks, self.vs = [], [], [], []
        for layer in model.decoder.layers:
            self.kc.append(layer.src_attn.linears[1](enc))
            self.vc.append(layer.src_attn.linears[2](enc))
            self.ks.append(torch.zeros(R, max_len, d, dtype=BF16, device=enc.device))
            self.vs.append(torch.zeros(R, max_len, d, dtype=BF16, device=enc.device))

    def reorder(self, ix):
        self.mem = self.mem.index_select(0, ix)
        self.enc = self.enc.index_select(0, ix)
        if self.src_mask is not None:
            self.src_mask = self.src_mask.index_select(0, ix)
        t = self.t
        for i in range(len(self.ks)):
            self.kc[i] = self.kc[i].index_select(0, ix)
            self.vc[i] = self.vc[i].index_select(0, ix)
            if ix.numel() != self.ks[i].shape[0]:
                nk = torch.zeros(ix.numel(), *self.ks[i].shape[1:], dtype=BF16, device=ix.device)
                nv = torch.zeros_like(nk)
                nk[:, :t] = self.ks[i][:, :t].index_select(0, ix)
                nv[:, :t] = self.vs[i][:, :t].index_select(0, ix)
                self.ks[i], self.vs[i] = nk, nv
            else:
                self.ks[i][:, :t] = self.ks[i][:, :t].index_select(0, ix)
                self.vs[i][:, :t] = self.vs[i][:, :t].index_select(0, ix)

    def step(self, it):
        """it (R,) token ids at position self.t -> f32 log-probs (R, V+1) of the next token."""
        model, t = self.model, self.t
        h = model.decoder.layers[0].self_attn.h
        pe = model.tgt_embed[1].pe[0][t:t + 1]
        emb = ops.embedding(it.view(-1, 1).contiguous(), model.tgt_embed[0].lut.weight, pos=pe, scale=math.sqrt(model.d_model))
        memory, self.mem = model.rm.run(emb, self.mem)          # (R, 1, slots*d), carried memory
        x = emb
        for i, layer in enumerate(model.decoder.layers):
            n = layer.sublayer[0].norm(x, memory)
            sa = layer.self_attn
            q = sa.linears[0](n)
            self.ks[i][:, t:t + 1] = sa.linears[1](n)
            self.vs[i][:, t:t + 1] = sa.linears[2](n)
            c = ops.attention(q, self.ks[i][:, :t + 1].contiguous(), self.vs[i][:, :t + 1].contiguous(), h)
            x = sa.linears[3](c, resid=x)
            n = layer.sublayer[1].norm(x, memory)
            ca = layer.src_attn
            c = ops.attention(ca.linears[0](n), self.kc[i], self.vc[i], h, mask=self.src_mask)
            x = ca.linears[3](c, resid=x)
            n = layer.sublayer[2].norm(x, memory)
            x = layer.feed_forward(n, resid=x)
        out = model.decoder.norm(x)
        logits = self.dec.logit(out, out_f32=True)
        self.t = t + 1
        return ops.log_softmax(logits.view(logits.shape[0], -1), self.dec.vocab_size + 1)


@torch.no_grad()
def beam_search(dec, enc_states, enc_mask, args, return_scores=False):
    """-> (B, max_seq_len) int64 token ids padded with [PAD] (= AttModel._sample_beam with sample_n = 1)."""
    was_training = dec.training
    dec.eval()
    try:
        beam, max_len = int(args.get('beam_size', 3)), int(args['max_seq_len'])
        if args.get('group_size', 1) != 1 or args.get('sample_n', 1) != 1:
            raise NotImplementedError('diverse beam search (group_size > 1) is not on the path')
        if beam < 1 or beam > 8 or beam > dec.vocab_size + 1:
            raise ValueError('beam_size must be in [1, 8]')
        enc, src_mask = dec.encode(enc_states, enc_mask)
        B, dev = enc.shape[0], enc.device
        V1 = dec.vocab_size + 1
        st = _DecoderState(dec, enc, src_mask, max_len)
        logp = st.step(torch.full((B,), dec.bos_idx, dtype=torch.long, device=dev))            # (B, V+1)
        base = torch.arange(B, device=dev).unsqueeze(1)
        beam_seq = torch.zeros(B, beam, max_len, dtype=torch.long, device=dev)
        beam_sum = torch.zeros(B, beam, dtype=F32, device=dev)
        best_p = torch.full((B,), -float('inf'), dtype=F32, device=dev)
        best_seq = torch.full((B, max_len), dec.pad_idx, dtype=torch.long, device=dev)
        nb = 1
        for t in range(max_len):
            cand = (beam_sum[:, :nb].unsqueeze(-1) + logp.view(B, nb, V1)).reshape(B, nb * V1).contiguous()
            ys, ix = _topk(cand, beam)
            beam_ix, word_ix = ix // V1, ix % V1
            state_ix = (beam_ix + base * nb).reshape(-1)
            if t > 0:
                beam_seq = beam_seq.gather(1, beam_ix.unsqueeze(-1).expand(-1, -1, max_len))
            beam_seq[:, :, t] = word_ix
            beam_sum = ys.clone()
            st.reorder(state_ix)
            is_end = word_ix == dec.eos_idx
            if t == max_len - 1:
                is_end = torch.ones_like(is_end)
            # finished beams: keep, per sample, the best p seen so far (earlier / lower beam index wins ties)
            p_end = torch.where(is_end, beam_sum, torch.full_like(beam_sum, -float('inf')))
            pv, pi = p_end.max(dim=1)
            better = pv > best_p
            cand_seq = beam_seq.gather(1, pi.view(B, 1, 1).expand(-1, 1, max_len))[:, 0]
            best_seq = torch.where(better.unsqueeze(1), cand_seq, best_seq)
            best_p = torch.where(better, pv, best_p)
            beam_sum = beam_sum - 1000.0 * is_end.to(F32)
            if t == max_len - 1:
                break
            logp = st.step(word_ix.reshape(-1))
            nb = beam
        if return_scores:
            return best_seq, best_p
        return best_seq
    finally:
        dec.train(was_training)
