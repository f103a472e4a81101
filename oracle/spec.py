"""CPU oracle (test infrastructure): state_dict layout of the reference models + procedural weights.

``finetune_spec`` / ``pretrain_spec`` enumerate every tensor of the reference's
``FineTune`` / ``Pretrain`` state_dict (name -> (shape, kind)) -- SURVEY.md section 8b;
checked against the imported reference in tests/golden/make_golden.py (1029 tensors,
347,799,781 trainable parameters at V=1444).

``procedural_fill`` is the bit-exact integer weight generator of SURVEY.md section 8c:
splitmix64 over (fnv1a64(name), flat index) -> uniform value in a per-kind range.  It
is applied by key to the imported reference, to this oracle and to the HIP engine, so
only input seeds and outputs have to be stored as golden fixtures.
"""
from collections import OrderedDict

import numpy as np
import torch

from .functional import RESNET_LAYERS, positional_encoding


def _bn(spec, name, c, affine=True):
    if affine:
        spec[name + '.weight'] = ((c,), 'bn_w')
        spec[name + '.bias'] = ((c,), 'bn_b')
    spec[name + '.running_mean'] = ((c,), 'bn_mean')
    spec[name + '.running_var'] = ((c,), 'bn_var')
    spec[name + '.num_batches_tracked'] = ((), 'nbt')


def _lin(spec, name, out_f, in_f, kind='lin_w'):
    spec[name + '.weight'] = ((out_f, in_f), kind)
    spec[name + '.bias'] = ((out_f,), 'bias')


def _ln(spec, name, d, w='weight', b='bias'):
    spec[name + '.' + w] = ((d,), 'ln_w')
    spec[name + '.' + b] = ((d,), 'ln_b')


def resnet_spec(spec, prefix='visual_extractor.model.'):
    spec[prefix + '0.weight'] = ((64, 3, 7, 7), 'conv')
    _bn(spec, prefix + '1', 64)
    inpl = 64
    for li, (planes, blocks, stride) in enumerate(RESNET_LAYERS):
        for b in range(blocks):
            p = '%s%d.%d.' % (prefix, 4 + li, b)
            spec[p + 'conv1.weight'] = ((planes, inpl, 1, 1), 'conv')
            _bn(spec, p + 'bn1', planes)
            spec[p + 'conv2.weight'] = ((planes, planes, 3, 3), 'conv')
            _bn(spec, p + 'bn2', planes)
            spec[p + 'conv3.weight'] = ((planes * 4, planes, 1, 1), 'conv')
            _bn(spec, p + 'bn3', planes * 4)
            if b == 0:
                spec[p + 'downsample.0.weight'] = ((planes * 4, inpl, 1, 1), 'conv')
                _bn(spec, p + 'downsample.1', planes * 4)
            inpl = planes * 4


def bert_attention_spec(spec, pre, h):
    for n in ('query', 'key', 'value'):
        _lin(spec, '%s.self.%s' % (pre, n), h, h)
    _lin(spec, pre + '.output.dense', h, h)
    _ln(spec, pre + '.output.LayerNorm', h)


def bert_layer_spec(spec, pre, h, inter, cross=False):
    bert_attention_spec(spec, pre + '.attention', h)
    if cross:
        bert_attention_spec(spec, pre + '.crossattention', h)
    _lin(spec, pre + '.intermediate.dense', inter, h)
    _lin(spec, pre + '.output.dense', h, inter)
    _ln(spec, pre + '.output.LayerNorm', h)


def text_encoder_spec(spec, vocab, h=768, layers=6, inter=3072, max_pos=512, types=2, pre='text_encoder.encoder'):
    spec[pre + '.embeddings.word_embeddings.weight'] = ((vocab, h), 'emb')
    spec[pre + '.embeddings.position_embeddings.weight'] = ((max_pos, h), 'emb')
    spec[pre + '.embeddings.token_type_embeddings.weight'] = ((types, h), 'emb')
    _ln(spec, pre + '.embeddings.LayerNorm', h)
    for i in range(layers):
        bert_layer_spec(spec, '%s.encoder.layer.%d' % (pre, i), h, inter)
    _lin(spec, pre + '.pooler.dense', h, h)


def head_spec(spec, name, in_dim, hid, out, final_bn):
    spec[name + '.head.0.weight'] = ((hid, in_dim, 1), 'lin_w')
    spec[name + '.head.0.bias'] = ((hid,), 'bias')
    _bn(spec, name + '.head.1', hid)
    spec[name + '.head.3.weight'] = ((out, hid, 1), 'lin_w')
    spec[name + '.head.3.bias'] = ((out,), 'bias')
    if final_bn:
        _bn(spec, name + '.head.4', out, affine=False)


def sdpa_spec(spec, name='multiview_cross_attention', d=2048, h=8):
    for n in ('fc_q', 'fc_k', 'fc_v'):
        _lin(spec, '%s.%s' % (name, n), h * d, d)
    _lin(spec, name + '.fc_o', d, h * d)


def r2gen_spec(spec, vocab, d=512, dff=512, dvf=2048, layers=3, slots=3, pre='text_decoder'):
    _lin(spec, pre + '.att_embed.0', d, dvf)
    m = pre + '.model'
    for i in range(layers):
        lp = '%s.encoder.layers.%d' % (m, i)
        for j in range(4):
            _lin(spec, '%s.self_attn.linears.%d' % (lp, j), d, d)
        _lin(spec, lp + '.feed_forward.w_1', dff, d)
        _lin(spec, lp + '.feed_forward.w_2', d, dff)
        for j in range(2):
            _ln(spec, '%s.sublayer.%d.norm' % (lp, j), d, 'gamma', 'beta')
    _ln(spec, m + '.encoder.norm', d, 'gamma', 'beta')
    for i in range(layers):
        lp = '%s.decoder.layers.%d' % (m, i)
        for att in ('self_attn', 'src_attn'):
            for j in range(4):
                _lin(spec, '%s.%s.linears.%d' % (lp, att, j), d, d)
        _lin(spec, lp + '.feed_forward.w_1', dff, d)
        _lin(spec, lp + '.feed_forward.w_2', d, dff)
        for j in range(3):
            n = '%s.sublayer.%d.norm' % (lp, j)
            _ln(spec, n, d, 'gamma', 'beta')
            _lin(spec, n + '.mlp_gamma.0', d, slots * d)
            _lin(spec, n + '.mlp_gamma.2', d, d)
            _lin(spec, n + '.mlp_beta.0', d, slots * d)
            _lin(spec, n + '.mlp_beta.2', d, d)
    _ln(spec, m + '.decoder.norm', d, 'gamma', 'beta')
    spec[m + '.tgt_embed.0.lut.weight'] = ((vocab + 1, d), 'emb')
    spec[m + '.tgt_embed.1.pe'] = ((1, 5000, d), 'pe')
    for j in range(4):
        _lin(spec, '%s.rm.attn.linears.%d' % (m, j), d, d)
    _lin(spec, m + '.rm.mlp.0', d, d)
    _lin(spec, m + '.rm.mlp.2', d, d)
    _lin(spec, m + '.rm.W', 2 * d, d)
    _lin(spec, m + '.rm.U', 2 * d, d)
    _lin(spec, pre + '.logit', vocab + 1, d)


def finetune_spec(vocab, fusion_layers=1, max_pos=512):
    """Key order follows FineTune.__init__ (...v0623_large_res.py:22-76)."""
    s = OrderedDict()
    resnet_spec(s)
    text_encoder_spec(s, vocab, max_pos=max_pos)
    _ln(s, 'layer_norm_1', 2048)
    _ln(s, 'layer_norm_2', 2048)
    r2gen_spec(s, vocab)
    head_spec(s, 'visual_head', 2048, 2048, 2048, True)
    head_spec(s, 'text_head', 768, 2048, 2048, True)
    sdpa_spec(s)
    for i in range(fusion_layers):
        bert_layer_spec(s, 'visual_self_atten_layers.%d' % i, 2048, 3072)
    for i in range(fusion_layers):
        bert_layer_spec(s, 'multimodal_fusion_layers.%d' % i, 2048, 3072, cross=True)
    return s


def pretrain_spec(vocab, max_pos=512):
    """Key order follows Pretrain.__init__ (...v0623_large_res.py:221-246)."""
    s = OrderedDict()
    resnet_spec(s)
    text_encoder_spec(s, vocab, max_pos=max_pos)
    _ln(s, 'layer_norm_1', 2048)
    _ln(s, 'layer_norm_2', 2048)
    head_spec(s, 'visual_head', 2048, 2048, 2048, False)
    head_spec(s, 'text_head', 768, 2048, 2048, False)
    sdpa_spec(s)
    return s


BUFFER_KINDS = ('bn_mean', 'bn_var', 'nbt', 'pe')


def n_trainable(spec):
    return sum(int(np.prod(sh)) for sh, k in spec.values() if k not in BUFFER_KINDS)


# ----------------------------------------------------------------------------
# procedural weights
# ----------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name):
    h = 0xcbf29ce484222325
    for ch in name.encode('utf-8'):
        h = ((h ^ ch) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64_uniform(seed, n):
    """n doubles in [0,1): splitmix64 finaliser of (seed + (i+1)*golden), top 53 bits."""
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _range(kind, shape):
    if kind == 'conv':
        fan_in = int(np.prod(shape[1:]))
        b = float(np.sqrt(6.0 / fan_in))
        return -b, b
    if kind == 'lin_w':
        fan_in = int(np.prod(shape[1:]))
        b = float(np.sqrt(3.0 / fan_in))
        return -b, b
    if kind == 'c1d_w':                      # HF Conv1D (in, out)
        b = float(np.sqrt(3.0 / shape[0]))
        return -b, b
    if kind == 'emb':
        return -0.08, 0.08
    if kind in ('bn_w', 'bn_var', 'ln_w'):
        return 0.6, 1.4
    if kind in ('bn_b', 'bn_mean', 'ln_b', 'bias'):
        return -0.1, 0.1
    raise KeyError(kind)


def procedural_tensor(name, shape, kind, salt=0):
    """The last BatchNorm of every bottleneck (`bn3.weight`) gets a small gain (0.09..0.21, like zero-init-residual or a
    trained network): with gain ~1 on RANDOM weights the 33-block trunk amplifies any activation perturbation ~linearly
    in depth under train-mode BN (bf16 storage then differs from fp32 by 0.85 relative at layer4 for ANY
    implementation), which would make the fixtures useless for a bf16 engine; with the small gain it stays ~5e-2."""
    if kind == 'nbt':
        return torch.zeros((), dtype=torch.long)
    if kind == 'pe':
        return positional_encoding(shape[1], shape[2])
    n = int(np.prod(shape)) if len(shape) else 1
    lo, hi = _range(kind, shape)
    if name.endswith('bn3.weight'):
        lo, hi = 0.09, 0.21
    u = splitmix64_uniform((fnv1a64(name) + salt) & 0xFFFFFFFFFFFFFFFF, n)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(shape))


def procedural_state(spec, salt=0):
    return OrderedDict((k, procedural_tensor(k, sh, kind, salt)) for k, (sh, kind) in spec.items())


def det(shape, a=0.37, b=0.11, c=0.0):
    """SURVEY.md 8c input generator: x[i] = sin(i*a + c) + cos(i*b), float64 -> float32, row-major."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return torch.from_numpy((np.sin(i * a + c) + np.cos(i * b)).astype(np.float32).reshape(shape))
