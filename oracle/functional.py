"""CPU oracle (test infrastructure) -- functional restatement of the EVOKE hot path.

Pure functions over ``P``: a dict {reference state_dict key -> fp32 tensor}.
Every function cites the reference file:line (under /root/reference) it follows.
No module of the product (evoke_amd/) imports this file.

Conventions: N images (anchors first), B anchors/text rows, P patches, T=P+1,
L report tokens, Li indication tokens, D=2048.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# run-time switches (parity is only defined with dropout off: SURVEY section 7 hard parts)
# ----------------------------------------------------------------------------


class Ctx:
    """train: BN uses batch statistics (+ running-stat update); dropout: apply the
    reference's dropout layers (only meaningful for timing, never for parity)."""

    def __init__(self, train=False, dropout=False):
        self.train = train
        self.dropout = dropout and train

    def drop(self, x, p):
        if self.dropout and p > 0:
            return F.dropout(x, p, True)
        return x


def _lin(P, name, x):
    return F.linear(x, P[name + '.weight'], P[name + '.bias'])


def _bn(P, name, x, ctx, affine=True, eps=1e-5, momentum=0.1):
    """torch.nn.BatchNorm{1,2}d semantics (stats over every dim but channel dim 1)."""
    w = P[name + '.weight'] if affine else None
    b = P[name + '.bias'] if affine else None
    if ctx.train and (name + '.num_batches_tracked') in P:
        P[name + '.num_batches_tracked'] += 1
    return F.batch_norm(x, P[name + '.running_mean'], P[name + '.running_var'], w, b, ctx.train, momentum, eps)


# ----------------------------------------------------------------------------
# a1  ResNet-101 trunk -- modules/visual_extractor.py:27-43 (ResNetTemp); arithmetic lives in
# un-vendored torchvision==0.16.2 resnet101 (children()[:-2]); restated from the public
# architecture: Bottleneck x [3,4,23,3], stride on the 3x3 conv ("v1.5"), bias-free convs.
# ----------------------------------------------------------------------------
RESNET_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))


def resnet101_trunk(P, images, ctx, prefix='visual_extractor.model.'):
    x = F.conv2d(images, P[prefix + '0.weight'], None, 2, 3)
    x = F.relu(_bn(P, prefix + '1', x, ctx))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS):
        for b in range(blocks):
            p = '%s%d.%d.' % (prefix, 4 + li, b)
            s = stride if b == 0 else 1
            y = F.relu(_bn(P, p + 'bn1', F.conv2d(x, P[p + 'conv1.weight']), ctx))
            y = F.relu(_bn(P, p + 'bn2', F.conv2d(y, P[p + 'conv2.weight'], None, s, 1), ctx))
            y = _bn(P, p + 'bn3', F.conv2d(y, P[p + 'conv3.weight']), ctx)
            if b == 0:
                x = _bn(P, p + 'downsample.1', F.conv2d(x, P[p + 'downsample.0.weight'], None, s), ctx)
            x = F.relu(y + x)
    return x


def visual_extractor(P, images, ctx):
    """visual_extractor.py:37-43 -> (patch_feats (N,P,2048), avg_feats (N,2048))."""
    f = resnet101_trunk(P, images, ctx)
    n, c = f.shape[:2]
    patch = f.reshape(n, c, -1).permute(0, 2, 1)
    return patch, patch.mean(dim=1)


# ----------------------------------------------------------------------------
# a4  projection heads -- modules/utils_v0511.py:131-208  (Conv1d k=1 -> BN1d -> ReLU -> Conv1d k=1 [-> BN1d no affine])
# ----------------------------------------------------------------------------
def projection_head(P, name, x, ctx, final_bn):
    b, t, c = x.shape
    h = F.conv1d(x.permute(0, 2, 1), P[name + '.head.0.weight'], P[name + '.head.0.bias'])
    h = F.relu(_bn(P, name + '.head.1', h, ctx))
    h = F.conv1d(h, P[name + '.head.3.weight'], P[name + '.head.3.bias'])
    if final_bn:
        h = _bn(P, name + '.head.4', h, ctx, affine=False)
    return h.permute(0, 2, 1)


# ----------------------------------------------------------------------------
# a3  ScaledDotProductAttention -- modules/utils_v0511.py:251-279  (h heads of width d_k = d_model)
# ----------------------------------------------------------------------------
def sdpa_multiview(P, name, q_in, kv_in, ctx, h=8):
    nq, d = q_in.shape
    nk = kv_in.shape[0]
    q = _lin(P, name + '.fc_q', q_in).view(nq, h, d).permute(1, 0, 2)
    k = _lin(P, name + '.fc_k', kv_in).view(nk, h, d).permute(1, 2, 0)
    v = _lin(P, name + '.fc_v', kv_in).view(nk, h, d).permute(1, 0, 2)
    att = torch.softmax(torch.matmul(q, k) / np.sqrt(d), -1)
    att = ctx.drop(att, 0.1)
    out = torch.matmul(att, v).permute(1, 0, 2).reshape(nq, h * d)
    return _lin(P, name + '.fc_o', out)


def _ln(P, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), P[name + '.weight'], P[name + '.bias'], eps)


# a2  multiview_fusion -- models/model_pretrain_finetune_v0623_large_res.py:126-150 / 284-309
def multiview_fusion(P, fc, att, patient_ids, batch_size, ctx, final_bn):
    pid = np.asarray(patient_ids)
    labels = (pid.reshape(-1, 1) == pid.reshape(1, -1))
    np.fill_diagonal(labels, False)
    x = _ln(P, 'layer_norm_1', torch.cat([fc.unsqueeze(1), att], dim=1))
    rows = []
    for i in range(batch_size):
        sib = np.nonzero(labels[i])[0]
        if len(sib) == 0:
            rows.append(x[i])
            continue
        kv = torch.cat([x[j] for j in sib], dim=0).detach()
        o = sdpa_multiview(P, 'multiview_cross_attention', x[i], kv, ctx)
        rows.append(_ln(P, 'layer_norm_2', o + x[i]))
    y = projection_head(P, 'visual_head', torch.stack(rows, 0), ctx, final_bn)
    return y[:, 0, :], y[:, 1:, :]


def no_fusion(P, fc, att, ctx, final_bn):
    """is_multiview_learning=False branch -- ...v0623...:161-165 / 364-368."""
    x = _ln(P, 'layer_norm_1', torch.cat([fc.unsqueeze(1), att], dim=1))
    y = projection_head(P, 'visual_head', x, ctx, final_bn)
    return y[:, 0, :], y[:, 1:, :]


# ----------------------------------------------------------------------------
# a5/a6/a7/a8  BERT pieces -- models/language_encoder/bert_model.py:210-502,548-628 (vendored HF BERT),
# language_model.py:120-154 (TextEncoderModel -> un-vendored HF BertModel, pinned vs in-container HF)
# ----------------------------------------------------------------------------
def extended_mask(mask):
    """modules/utils_v0511.py:697-753 -> additive (B,1,1,S) mask."""
    m = mask[:, None, None, :].to(torch.float32)
    return (1.0 - m) * torch.finfo(torch.float32).min


def bert_attention(P, pre, x, kv, add_mask, heads, ctx, eps=1e-12):
    """BertAttention = BertSelfAttention + BertSelfOutput (bert_model.py:210-412); post-LN."""
    b, t, hd = x.shape
    s = kv.shape[1]
    dh = hd // heads
    q = _lin(P, pre + '.self.query', x).view(b, t, heads, dh).permute(0, 2, 1, 3)
    k = _lin(P, pre + '.self.key', kv).view(b, s, heads, dh).permute(0, 2, 1, 3)
    v = _lin(P, pre + '.self.value', kv).view(b, s, heads, dh).permute(0, 2, 1, 3)
    sc = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if add_mask is not None:
        sc = sc + add_mask
    p = ctx.drop(torch.softmax(sc, dim=-1), 0.1)
    c = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(b, t, hd)
    o = ctx.drop(_lin(P, pre + '.output.dense', c), 0.1)
    return _ln(P, pre + '.output.LayerNorm', o + x, eps)


def bert_ffn(P, pre, x, ctx, eps=1e-12):
    h = F.gelu(_lin(P, pre + '.intermediate.dense', x))
    o = ctx.drop(_lin(P, pre + '.output.dense', h), 0.1)
    return _ln(P, pre + '.output.LayerNorm', o + x, eps)


def bert_layer(P, pre, x, add_mask, heads, ctx):
    """BertLayer.forward (bert_model.py:562-628), encoder configuration."""
    a = bert_attention(P, pre + '.attention', x, x, add_mask, heads, ctx)
    return bert_ffn(P, pre, a, ctx)


def bert_cross_layer(P, pre, x, y, x_mask, y_mask, heads, ctx):
    """BertCrossLayer.forward (bert_model.py:456-502): self-attn(x) -> cross-attn(x, y) -> FFN."""
    a = bert_attention(P, pre + '.attention', x, x, x_mask, heads, ctx)
    c = bert_attention(P, pre + '.crossattention', a, y, y_mask, heads, ctx)
    return bert_ffn(P, pre, c, ctx)


def text_encoder(P, ids, mask, ctx, layers=6, heads=12, pre='text_encoder.encoder'):
    """HF BertModel(...)[0]: embeddings (word+pos+type0 -> LN eps 1e-12) + `layers` post-LN blocks."""
    b, l = ids.shape
    # nn.Embedding(padding_idx=pad_token_id=0): the [PAD] row receives no gradient
    e = (F.embedding(ids, P[pre + '.embeddings.word_embeddings.weight'], padding_idx=0)
         + P[pre + '.embeddings.token_type_embeddings.weight'][0]
         + P[pre + '.embeddings.position_embeddings.weight'][:l][None])
    x = ctx.drop(_ln(P, pre + '.embeddings.LayerNorm', e, 1e-12), 0.1)
    am = extended_mask(mask)
    for i in range(layers):
        x = bert_layer(P, '%s.encoder.layer.%d' % (pre, i), x, am, heads, ctx)
    return x


# ----------------------------------------------------------------------------
# a9-a13  R2Gen memory-driven Transformer -- modules/encoder_decoder.py, modules/att_model.py
# ----------------------------------------------------------------------------
def r2_layernorm(x, gamma, beta, eps=1e-6):
    """encoder_decoder.py:93-103 / 166-179: unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gamma * (x - mean) / (std + eps) + beta


def r2_mha(P, pre, q_in, k_in, v_in, mask, h, ctx):
    """MultiHeadedAttention.forward + attention() -- encoder_decoder.py:20-28, 192-203."""
    b, d = q_in.shape[0], q_in.shape[-1]
    dk = d // h
    q = _lin(P, pre + '.linears.0', q_in).view(b, -1, h, dk).transpose(1, 2)
    k = _lin(P, pre + '.linears.1', k_in).view(b, -1, h, dk).transpose(1, 2)
    v = _lin(P, pre + '.linears.2', v_in).view(b, -1, h, dk).transpose(1, 2)
    sc = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        sc = sc.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p = ctx.drop(F.softmax(sc, dim=-1), 0.1)
    x = torch.matmul(p, v).transpose(1, 2).reshape(b, -1, d)
    return _lin(P, pre + '.linears.3', x)


def r2_ff(P, pre, x):
    return _lin(P, pre + '.w_2', F.relu(_lin(P, pre + '.w_1', x)))


def r2_encode(P, feats, src_mask, ctx, cfg, pre='text_decoder.model.encoder'):
    """Encoder/EncoderLayer/SublayerConnection -- encoder_decoder.py:58-90 (pre-LN residual blocks)."""
    x = feats
    for i in range(cfg['num_layers']):
        lp = '%s.layers.%d' % (pre, i)
        n = r2_layernorm(x, P[lp + '.sublayer.0.norm.gamma'], P[lp + '.sublayer.0.norm.beta'])
        x = x + r2_mha(P, lp + '.self_attn', n, n, n, src_mask, cfg['num_heads'], ctx)
        n = r2_layernorm(x, P[lp + '.sublayer.1.norm.gamma'], P[lp + '.sublayer.1.norm.beta'])
        x = x + r2_ff(P, lp + '.feed_forward', n)
    return r2_layernorm(x, P[pre + '.norm.gamma'], P[pre + '.norm.beta'])


def r2_cln(P, pre, x, memory):
    """ConditionalLayerNorm.forward -- encoder_decoder.py:166-179."""
    dg = _lin(P, pre + '.mlp_gamma.2', F.relu(_lin(P, pre + '.mlp_gamma.0', memory)))
    db = _lin(P, pre + '.mlp_beta.2', F.relu(_lin(P, pre + '.mlp_beta.0', memory)))
    return r2_layernorm(x, P[pre + '.gamma'] + dg, P[pre + '.beta'] + db)


def rm_init_memory(batch, slots, d):
    """RelationalMemory.init_memory -- encoder_decoder.py:263-272."""
    m = torch.zeros(batch, slots, d)
    m[:, :, :slots] = torch.eye(slots)
    return m


def rm_step(P, x_t, memory, cfg, ctx, pre='text_decoder.model.rm'):
    """RelationalMemory.forward_step -- encoder_decoder.py:274-291. memory (B, slots*d) flat in/out."""
    s, d = cfg['rm_num_slots'], cfg['rm_d_model']
    m = memory.reshape(-1, s, d)
    kv = torch.cat([m, x_t.unsqueeze(1)], 1)
    nm = m + r2_mha(P, pre + '.attn', m, kv, kv, None, cfg['rm_num_heads'], ctx)
    nm = nm + F.relu(_lin(P, pre + '.mlp.2', F.relu(_lin(P, pre + '.mlp.0', nm))))
    gates = _lin(P, pre + '.W', x_t.unsqueeze(1)) + _lin(P, pre + '.U', torch.tanh(m))
    ig, fg = torch.split(gates, d, dim=2)
    nm = torch.sigmoid(ig) * torch.tanh(nm) + torch.sigmoid(fg) * m
    return nm.reshape(-1, s * d)


def rm_forward(P, emb, cfg, ctx):
    """RelationalMemory.forward -- encoder_decoder.py:293-300: serial over tokens -> (B, L, slots*d)."""
    b = emb.shape[0]
    mem = rm_init_memory(b, cfg['rm_num_slots'], cfg['rm_d_model']).to(emb)
    outs = []
    for t in range(emb.shape[1]):
        mem = rm_step(P, emb[:, t], mem, cfg, ctx)
        outs.append(mem)
    return torch.stack(outs, dim=1)


def positional_encoding(max_len, d):
    """PositionalEncoding buffer -- encoder_decoder.py:232-238."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def r2_tgt_embed(P, ids, cfg, pre='text_decoder.model.tgt_embed'):
    """Embeddings (x sqrt(d)) + PositionalEncoding -- encoder_decoder.py:217-243."""
    return P[pre + '.0.lut.weight'][ids] * math.sqrt(cfg['d_model']) + P[pre + '.1.pe'][:, :ids.shape[1]]


def subsequent_mask(size):
    """encoder_decoder.py:31-34: (1,size,size) bool, lower-triangular incl. diagonal."""
    return torch.tril(torch.ones(1, size, size, dtype=torch.bool))


def r2_decode(P, memory_enc, src_mask, ids, tgt_mask, cfg, ctx, pre='text_decoder.model.decoder'):
    """Transformer.decode + Decoder/DecoderLayer -- encoder_decoder.py:52-55, 106-141."""
    emb = r2_tgt_embed(P, ids, cfg)
    mem = rm_forward(P, emb, cfg, ctx)
    x = emb
    h = cfg['num_heads']
    for i in range(cfg['num_layers']):
        lp = '%s.layers.%d' % (pre, i)
        n = r2_cln(P, lp + '.sublayer.0.norm', x, mem)
        x = x + r2_mha(P, lp + '.self_attn', n, n, n, tgt_mask, h, ctx)
        n = r2_cln(P, lp + '.sublayer.1.norm', x, mem)
        x = x + r2_mha(P, lp + '.src_attn', n, memory_enc, memory_enc, src_mask, h, ctx)
        n = r2_cln(P, lp + '.sublayer.2.norm', x, mem)
        x = x + r2_ff(P, lp + '.feed_forward', n)
    return r2_layernorm(x, P[pre + '.norm.gamma'], P[pre + '.norm.beta'])


def r2_att_embed(P, att_feats, att_masks, ctx):
    """AttModel.att_embed via pack_wrapper -- att_model.py:28-35, 59-64 (Linear, ReLU, Dropout .5)."""
    x = att_feats * att_masks.unsqueeze(-1)
    return ctx.drop(F.relu(_lin(P, 'text_decoder.att_embed.0', x)), 0.5)


def r2_forward_logprobs(P, ids, enc_states, attention_mask, enc_mask, cfg, ctx):
    """EncoderDecoder._forward -- encoder_decoder.py:385-394 (token 0 of enc_states dropped)."""
    att, am = enc_states[:, 1:, :], enc_mask[:, 1:]
    feats = r2_att_embed(P, att, am, ctx)
    src_mask = am.unsqueeze(-2)
    tgt_mask = attention_mask.unsqueeze(-2) & subsequent_mask(ids.shape[-1]).to(ids)
    enc = r2_encode(P, feats, src_mask, ctx, cfg)
    out = r2_decode(P, enc, src_mask, ids, tgt_mask, cfg, ctx)
    return F.log_softmax(_lin(P, 'text_decoder.logit', out), dim=-1)


# a14  modules/loss.py:5-22
def compute_lm_loss(logp, ids, masks):
    target, mask = ids[:, 1:], masks[:, 1:]
    target = target[:, :logp.shape[1]]
    mask = mask[:, :logp.shape[1]]
    lp = logp[:, :target.shape[1]]
    out = -lp.gather(2, target.long().unsqueeze(2)).squeeze(2) * mask
    return out.sum() / mask.sum()


# ----------------------------------------------------------------------------
# a15-a17 contrastive losses -- models/model_pretrain_finetune_v0623_large_res.py:262-282, 311-351
# ----------------------------------------------------------------------------
def multi_pos_contra_images(g, patient_ids, temp):
    pid = np.asarray(patient_ids)
    labels = torch.from_numpy((pid.reshape(-1, 1) == pid.reshape(1, -1)).astype(np.float32)).to(g)
    labels.fill_diagonal_(0.0)
    idx = torch.argwhere(labels.sum(1) != 0).reshape(-1)
    if len(idx) == 0:
        return torch.tensor([0.0], requires_grad=True)
    g, labels = g[idx], labels[idx][:, idx]
    labels = labels / labels.sum(1, keepdim=True)
    g = F.normalize(g, dim=-1, p=2)
    logits = g @ g.T / temp
    logits = logits.masked_fill(torch.eye(len(idx), dtype=torch.bool), -1e9)
    logits = logits - logits.max(dim=-1, keepdim=True)[0].detach()
    return F.cross_entropy(logits, labels)


def global_alignment_loss(v, t, patient_ids, temp):
    pid = np.asarray(patient_ids)[:v.shape[0]]
    labels = torch.from_numpy((pid.reshape(-1, 1) == pid.reshape(1, -1)).astype(np.float32))
    labels = labels / labels.sum(1, keepdim=True)
    v = F.normalize(v, dim=-1, p=2)
    t = F.normalize(t, dim=-1, p=2)
    l1 = F.cross_entropy(v @ t.t() / temp, labels)
    l2 = F.cross_entropy(t @ v.t() / temp, labels)
    return (l1 + l2) / 2.0


def local_text_token_alignment_loss(patches, tokens, temp):
    sim = tokens @ patches.permute(0, 2, 1)
    sco = F.softmax(sim / math.sqrt(patches.shape[2]), dim=-1)
    att = F.normalize(torch.bmm(sco, patches), dim=-1, p=2)
    tok = F.normalize(tokens, dim=-1, p=2)
    ws = torch.bmm(tok, att.permute(0, 2, 1)) / temp
    b, n1, n2 = ws.shape
    tgt = torch.arange(n1).long().repeat(b)
    l1 = F.cross_entropy(ws.reshape(b * n1, n2), tgt)
    l2 = F.cross_entropy(ws.permute(0, 2, 1).reshape(b * n2, n1), tgt)
    return (l1 + l2) / 2.0


# ----------------------------------------------------------------------------
# a18  FineTune.forward / Pretrain.forward -- ...v0623_large_res.py:152-217, 353-395
# ----------------------------------------------------------------------------
DEFAULT_CFG = dict(
    num_layers=3, d_model=512, d_ff=512, num_heads=8, rm_num_slots=3, rm_num_heads=8, rm_d_model=512,
    fusion_num_heads=8, sk_fusion_num_layers=1, encoder_num_hidden_layers=6, encoder_num_heads=12,
    is_multiview_learning=True, instance_temp=0.5, region_temp=0.5, max_seq_len=100, beam_size=3,
)


def finetune_encoder_states(P, images, patient_ids, batch_size, inc_ids, inc_masks, cfg, ctx, taps=None):
    """Everything of FineTune.forward up to the decoder input (lines 152-203)."""
    att, fc = visual_extractor(P, images, ctx)
    if taps is not None:
        taps['att'], taps['fc'] = att, fc
    if cfg['is_multiview_learning']:
        v_fc, v_att = multiview_fusion(P, fc, att, patient_ids, batch_size, ctx, True)
    else:
        v_fc, v_att = no_fusion(P, fc, att, ctx, True)
    x = torch.cat([v_fc.unsqueeze(1), v_att], dim=1)
    if taps is not None:
        taps['fused'] = x
    enc_mask = torch.ones(x.shape[:2], dtype=torch.long)
    xm = extended_mask(enc_mask)
    heads = cfg['fusion_num_heads']
    if inc_ids is not None:
        y = text_encoder(P, inc_ids, inc_masks, ctx, cfg['encoder_num_hidden_layers'], cfg['encoder_num_heads'])
        y = projection_head(P, 'text_head', y, ctx, True)
        ym = extended_mask(inc_masks)
        for i in range(cfg['sk_fusion_num_layers']):
            x = bert_cross_layer(P, 'multimodal_fusion_layers.%d' % i, x, y, xm, ym, heads, ctx)
    else:
        for i in range(cfg['sk_fusion_num_layers']):
            x = bert_layer(P, 'visual_self_atten_layers.%d' % i, x, xm, heads, ctx)
    if taps is not None:
        taps['enc_states'] = x
    return x, enc_mask


def finetune_forward_train(P, images, report_ids, report_masks, patient_ids, inc_ids=None, inc_masks=None,
                           cfg=DEFAULT_CFG, ctx=None, taps=None):
    ctx = ctx or Ctx()
    x, enc_mask = finetune_encoder_states(P, images, patient_ids, report_ids.shape[0], inc_ids, inc_masks, cfg, ctx, taps)
    logp = r2_forward_logprobs(P, report_ids, x, report_masks, enc_mask, cfg, ctx)
    if taps is not None:
        taps['logp'] = logp
    loss = compute_lm_loss(logp, report_ids, report_masks)
    return {'lm': loss, 'all_loss': loss}


def pretrain_forward(P, images, radgraph_ids, radgraph_masks, patient_ids, cfg=DEFAULT_CFG, ctx=None, taps=None):
    ctx = ctx or Ctx()
    att, fc = visual_extractor(P, images, ctx)
    b = radgraph_ids.shape[0]
    mul = torch.tensor([0.0])
    if cfg['is_multiview_learning']:
        mul = multi_pos_contra_images(fc, patient_ids, cfg['region_temp'])
        v_fc, v_att = multiview_fusion(P, fc, att, patient_ids, b, ctx, False)
    else:
        v_fc, v_att = no_fusion(P, fc, att, ctx, False)
    t = text_encoder(P, radgraph_ids, radgraph_masks, ctx, cfg['encoder_num_hidden_layers'], cfg['encoder_num_heads'])
    t = projection_head(P, 'text_head', t, ctx, False)
    t_fc, t_att = t[:, 0, :], t[:, 1:, :]
    if taps is not None:
        taps.update(v_fc=v_fc, v_att=v_att, t_fc=t_fc, t_att=t_att, fc=fc)
    inst = global_alignment_loss(v_fc, t_fc, patient_ids, cfg['instance_temp'])
    sen = local_text_token_alignment_loss(v_att, t_att, cfg['region_temp'])
    allv = inst + sen + mul if cfg['is_multiview_learning'] else inst + sen
    return {'sen_image_loss': torch.tensor([0.0]), 'sen_text_loss': sen, 'instance_loss': inst,
            'multiview_loss': mul, 'all_loss': allv}
