#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the IMPORTED REFERENCE (runs only in the build container,
where /root/reference exists; the GPU box only ever sees the committed fixtures).

Recipe = SURVEY.md section 8c: in-memory shims for absent / drifted third-party packages, nothing is
written under /root/reference.  Weights are procedural (oracle.spec.procedural_state applied by
state_dict key), inputs come from fixed seeds, so the fixtures hold only inputs' seeds + outputs.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only kat,ft224,...]
"""
import argparse
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
warnings.filterwarnings('ignore')

from oracle import spec as S  # noqa: E402
from tests.golden.cases import CASES, make_inputs, reduce_tensor  # noqa: E402


# ----------------------------------------------------------------------------------------------
# shims
# ----------------------------------------------------------------------------------------------
class _Bottleneck(nn.Module):
    def __init__(self, inpl, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        return self.relu(self.bn3(self.conv3(y)) + idt)


class _ResNet101(nn.Module):
    """Stand-in for the un-installed torchvision.models.resnet101 (child order conv1,bn1,relu,maxpool,
    layer1-4,avgpool,fc so that children()[:-2] has indices 0-7)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for i, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))):
            layer = [_Bottleneck(inpl, planes, stride, True)]
            inpl = planes * 4
            layer += [_Bottleneck(inpl, planes, 1, False) for _ in range(blocks - 1)]
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(*layer))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)


def install_shims():
    sys.path.insert(0, REF)
    import transformers  # noqa  (must be imported before the fake torchvision is registered)
    import transformers.modeling_utils  # noqa
    sys.modules['cv2'] = types.ModuleType('cv2')
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvt = types.ModuleType('torchvision.transforms')
    tvm.resnet101 = lambda *a, **k: _ResNet101()
    tv.models, tv.transforms = tvm, tvt
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.transforms': tvt})
    import modules  # noqa
    import modules.metrics  # noqa
    st = types.ModuleType('modules.metrics.metrics')
    st.compute_ce_scores = st.compute_all_scores = lambda *a, **k: {}
    sys.modules['modules.metrics.metrics'] = st
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    if not hasattr(mu, 'find_pruneable_heads_and_indices'):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), torch.zeros(0, dtype=torch.long))
    import models  # noqa
    import models.language_encoder  # noqa
    bs = types.ModuleType('models.language_encoder.beam_search')
    bs.prepare_inputs_for_generation = bs._validate_model_kwargs = bs.beam_search = lambda *a, **k: None
    sys.modules['models.language_encoder.beam_search'] = bs
    torch.Tensor.cuda = lambda self, *a, **k: self


def make_args(tmp, tokenizer, max_seq_len=100, beam_size=3, multiview=True):
    import yaml
    from transformers import BertConfig, BertModel
    d = os.path.join(tmp, 'bert6')
    if not os.path.exists(d):
        BertModel(BertConfig(num_hidden_layers=6)).save_pretrained(d)
    args = yaml.safe_load(open(os.path.join(REF, 'config/finetune_config.yaml')))
    args.update(dict(resnet_checkpoint='', text_checkpoint=d, fusion_checkpoint=d, max_seq_len=max_seq_len,
                     num_layers=3, sk_fusion_num_layers=1, is_multiview_learning=multiview, is_add_indication=True,
                     vocab_size=tokenizer.get_vocab_size(), suppress_UNK=0, task='finetune', data_name='iu_xray',
                     beam_size=beam_size))
    return args


def load_procedural(model, spec):
    sd = model.state_dict()
    extra = [k for k in sd if k not in spec and not k.endswith('position_ids') and not k.endswith('token_type_ids')]
    missing = [k for k in spec if k not in sd]
    assert not extra and not missing, (extra[:5], missing[:5])
    for k, (shape, kind) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), (k, sd[k].shape, shape)
    model.load_state_dict(S.procedural_state(spec), strict=False)


def zero_dropout(model):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0


def hook_taps(model, names, store):
    hs = []
    for tap, modname in names.items():
        mod = model.get_submodule(modname)

        def fn(m, inp, out, tap=tap):
            store[tap] = out
        hs.append(mod.register_forward_hook(fn))
    return hs


GRAD_KEYS = ['visual_extractor.model.0.weight', 'visual_extractor.model.7.2.conv3.weight',
             'visual_extractor.model.5.0.bn2.weight', 'multiview_cross_attention.fc_q.weight',
             'multiview_cross_attention.fc_k.weight', 'layer_norm_2.weight', 'visual_head.head.0.weight',
             'text_encoder.encoder.embeddings.word_embeddings.weight',
             'text_encoder.encoder.encoder.layer.5.output.dense.weight', 'text_head.head.3.weight',
             'multimodal_fusion_layers.0.crossattention.self.key.weight',
             'visual_self_atten_layers.0.attention.self.query.weight',
             'text_decoder.logit.weight', 'text_decoder.model.rm.W.weight', 'text_decoder.model.rm.attn.linears.1.weight',
             'text_decoder.model.decoder.layers.0.sublayer.1.norm.mlp_gamma.0.weight',
             'text_decoder.model.decoder.layers.2.src_attn.linears.2.weight',
             'text_decoder.model.encoder.layers.0.self_attn.linears.0.weight',
             'text_decoder.model.tgt_embed.0.lut.weight', 'text_decoder.att_embed.0.weight',
             # round 3: every stage of the trunk, every transformer family, biases and norm parameters (66 tensors in all)
             'visual_extractor.model.1.weight', 'visual_extractor.model.1.bias',
             'visual_extractor.model.4.0.conv1.weight', 'visual_extractor.model.4.0.conv2.weight', 'visual_extractor.model.4.0.downsample.0.weight',
             'visual_extractor.model.4.2.bn3.weight', 'visual_extractor.model.4.2.bn3.bias',
             'visual_extractor.model.5.0.conv2.weight', 'visual_extractor.model.5.0.downsample.1.bias', 'visual_extractor.model.5.3.conv1.weight',
             'visual_extractor.model.6.0.conv1.weight', 'visual_extractor.model.6.0.downsample.0.weight', 'visual_extractor.model.6.5.conv2.weight',
             'visual_extractor.model.6.11.conv3.weight', 'visual_extractor.model.6.17.bn1.weight', 'visual_extractor.model.6.22.conv2.weight',
             'visual_extractor.model.6.22.bn2.bias', 'visual_extractor.model.7.0.conv2.weight', 'visual_extractor.model.7.0.downsample.0.weight',
             'visual_extractor.model.7.1.conv1.weight', 'visual_extractor.model.7.2.bn3.weight',
             'layer_norm_1.weight', 'layer_norm_1.bias', 'multiview_cross_attention.fc_v.weight', 'multiview_cross_attention.fc_o.weight',
             'multiview_cross_attention.fc_o.bias', 'visual_head.head.1.weight', 'visual_head.head.3.weight', 'text_head.head.0.weight',
             'text_encoder.encoder.embeddings.LayerNorm.weight', 'text_encoder.encoder.encoder.layer.0.attention.self.query.weight',
             'text_encoder.encoder.encoder.layer.2.intermediate.dense.weight', 'text_encoder.encoder.encoder.layer.3.attention.output.dense.bias',
             'multimodal_fusion_layers.0.attention.self.value.weight', 'multimodal_fusion_layers.0.crossattention.output.dense.weight',
             'multimodal_fusion_layers.0.intermediate.dense.weight', 'multimodal_fusion_layers.0.output.LayerNorm.weight',
             'visual_self_atten_layers.0.intermediate.dense.weight', 'visual_self_atten_layers.0.output.dense.bias',
             'text_decoder.logit.bias', 'text_decoder.model.rm.U.weight', 'text_decoder.model.rm.mlp.0.weight', 'text_decoder.model.rm.attn.linears.3.weight',
             'text_decoder.model.decoder.layers.0.self_attn.linears.0.weight', 'text_decoder.model.decoder.layers.1.feed_forward.w_1.weight',
             'text_decoder.model.decoder.layers.1.sublayer.0.norm.gamma', 'text_decoder.model.decoder.layers.2.sublayer.2.norm.mlp_beta.2.weight',
             'text_decoder.model.encoder.layers.2.feed_forward.w_2.weight', 'text_decoder.model.encoder.norm.gamma',
             'text_decoder.model.decoder.norm.beta']


def run_case(name, case, tmp, tokenizer):
    out = {}
    inp = make_inputs(case, tokenizer.get_vocab_size())
    kind = case['kind']
    args = make_args(tmp, tokenizer, case.get('max_seq_len', 100), case.get('beam_size', 3), case.get('multiview', True))
    if kind == 'finetune_gpt2':
        return run_finetune_gpt2(case, inp, args, tokenizer)
    if kind in ('finetune', 'beam'):
        from models.model_pretrain_finetune_v0623_large_res import FineTune
        model = FineTune(args, tokenizer, 'iu_xray')
        spec = S.finetune_spec(tokenizer.get_vocab_size())
    else:
        from models.model_pretrain_finetune_v0623_large_res import Pretrain
        model = Pretrain(args, tokenizer, 'iu_xray')
        spec = S.pretrain_spec(tokenizer.get_vocab_size())
    load_procedural(model, spec)
    zero_dropout(model)
    ntr = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert ntr == S.n_trainable(spec), (ntr, S.n_trainable(spec))
    out['n_trainable'] = np.int64(ntr)
    pid = np.array(inp['patient_ids'])
    for mode in case['modes']:
        model.train(mode == 'train')
        load_procedural(model, spec)           # reset BN running stats between modes
        taps = {}
        if kind == 'finetune':
            fus = 'multimodal_fusion_layers.0' if inp.get('inc_ids') is not None else 'visual_self_atten_layers.0'
            hs = hook_taps(model, {'resnet': 'visual_extractor', 'vhead': 'visual_head', 'fusion': fus,
                                   'logp': 'text_decoder'}, taps)
            model.zero_grad()
            ret = model(inp['images'], inp['ids'], inp['masks'], pid, inp.get('inc_ids'), inp.get('inc_masks'), mode='train')
            out['%s/loss' % mode] = np.float64(ret['all_loss'].item())
            out['%s/tap/att' % mode] = reduce_tensor(taps['resnet'][0])
            out['%s/tap/fc' % mode] = reduce_tensor(taps['resnet'][1])
            out['%s/tap/vhead' % mode] = reduce_tensor(taps['vhead'])
            out['%s/tap/enc_states' % mode] = reduce_tensor(taps['fusion'][0])
            out['%s/tap/logp' % mode] = reduce_tensor(taps['logp'])
            if case.get('logp'):
                from tests.golden.cases import PROBE_IDS
                lp = taps['logp'].detach().double()                                   # (B, L, V+1) log-probabilities of the teacher-forced pass
                tgt = torch.cat([inp['ids'][:, 1:], torch.zeros(inp['ids'].shape[0], 1, dtype=torch.long)], 1)
                out['%s/logp_target' % mode] = lp.gather(2, tgt.unsqueeze(-1)).squeeze(-1).numpy()      # position t predicts ids[:, t + 1]
                out['%s/logp_probe' % mode] = lp[:, :, PROBE_IDS].numpy()
            ret['all_loss'].backward()
            g = dict(model.named_parameters())
            for k in GRAD_KEYS:
                if g[k].grad is not None:
                    out['%s/grad/%s' % (mode, k)] = reduce_tensor(g[k].grad)
            if mode == 'train':
                sd = model.state_dict()
                out['train/bn/running_mean'] = reduce_tensor(sd['visual_extractor.model.7.2.bn3.running_mean'])
                out['train/bn/running_var'] = reduce_tensor(sd['visual_extractor.model.7.2.bn3.running_var'])
            for h in hs:
                h.remove()
        elif kind == 'pretrain':
            hs = hook_taps(model, {'resnet': 'visual_extractor', 'vhead': 'visual_head', 'thead': 'text_head'}, taps)
            model.zero_grad()
            ret = model(inp['images'], inp['ids'], inp['masks'], pid)
            for k in ('sen_text_loss', 'instance_loss', 'multiview_loss', 'all_loss'):
                out['%s/%s' % (mode, k)] = np.float64(ret[k].reshape(-1)[0].item())
            out['%s/tap/fc' % mode] = reduce_tensor(taps['resnet'][1])
            out['%s/tap/vhead' % mode] = reduce_tensor(taps['vhead'])
            out['%s/tap/thead' % mode] = reduce_tensor(taps['thead'])
            ret['all_loss'].backward()
            g = dict(model.named_parameters())
            for k in GRAD_KEYS:
                if k in g and g[k].grad is not None:
                    out['%s/grad/%s' % (mode, k)] = reduce_tensor(g[k].grad)
            for h in hs:
                h.remove()
        elif kind == 'beam':
            with torch.no_grad():
                texts, seq = model(inp['images'], inp['ids'], inp['masks'], pid, inp.get('inc_ids'), inp.get('inc_masks'),
                                   mode='inference')
            out['eval/seq'] = seq.numpy().astype(np.int64)
            out['eval/texts'] = np.array(texts)
    return out


GPT2_DIMS = dict(d=2048, layers=3, heads=8)          # config/finetune_config.yaml:30-32 (decoder_hidden_size / layers / heads)


def run_finetune_gpt2(case, inp, args, tokenizer):
    """config 1: reference FineTune up to encoder_hidden_states (visual extractor, LN1, visual head, indication cross-fusion) ->
    HF GPT2LMHeadModel(add_cross_attention) as DistilGPT2TextDecoderModel.forward would call it (language_model.py:237-258):
    un-shifted cross entropy; generate = HF beam search (language_model.py:271-280)."""
    import torch.nn.functional as F
    from transformers import GPT2Config, GPT2LMHeadModel
    from models.model_pretrain_finetune_v0623_large_res import FineTune
    from oracle import gpt2 as G
    V, d, layers, heads = tokenizer.get_vocab_size(), GPT2_DIMS['d'], GPT2_DIMS['layers'], GPT2_DIMS['heads']
    model = FineTune(args, tokenizer, 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    zero_dropout(model)
    model.eval()
    cfg = GPT2Config(vocab_size=V, n_embd=d, n_layer=layers, n_head=heads, add_cross_attention=True, is_decoder=True,
                     bos_token_id=V - 2, eos_token_id=V - 1, pad_token_id=0)
    dec = GPT2LMHeadModel(cfg).eval()
    sd = {k[len(G.PRE):]: v for k, v in S.procedural_state(G.gpt2_spec(V, d, layers)).items()}
    sd['lm_head.weight'] = sd['transformer.wte.weight']
    dec.load_state_dict(sd, strict=False)
    taps = {}
    hs = hook_taps(model, {'fusion': 'multimodal_fusion_layers.0'}, taps)
    # run the reference up to the decoder: its r2gen decoder's output is discarded, only the fusion tap is used
    model.zero_grad()
    model(inp['images'], inp['ids'], inp['masks'], np.array(inp['patient_ids']), inp['inc_ids'], inp['inc_masks'], mode='train')
    for h in hs:
        h.remove()
    enc = taps['fusion'][0]
    ids, am = inp['ids'], inp['masks']
    lg = dec(input_ids=ids, attention_mask=am, encoder_hidden_states=enc).logits
    loss = F.cross_entropy(lg.permute(0, 2, 1), ids, ignore_index=0)
    model.zero_grad()
    loss.backward()
    out = {'eval/loss': np.float64(loss.item()), 'eval/tap/enc_states': reduce_tensor(enc), 'eval/tap/logits': reduce_tensor(lg)}
    g = dict(model.named_parameters())
    for k in ('visual_extractor.model.7.2.conv3.weight', 'visual_head.head.0.weight', 'multimodal_fusion_layers.0.crossattention.self.key.weight',
              'text_encoder.encoder.encoder.layer.5.output.dense.weight'):
        out['eval/grad/' + k] = reduce_tensor(g[k].grad)
    gd = dict(dec.named_parameters())
    for k in ('transformer.wte.weight', 'transformer.h.0.attn.c_attn.weight', 'transformer.h.1.crossattention.c_attn.weight'):
        out['eval/grad/' + G.PRE + k] = reduce_tensor(gd[k].grad)
    with torch.no_grad():
        seq = dec.generate(input_ids=torch.full((case['B'], 1), V - 2), encoder_hidden_states=enc.detach(), num_beams=case['beam_size'],
                           max_length=case['max_seq_len'], use_cache=True, bos_token_id=V - 2, eos_token_id=V - 1, pad_token_id=0, do_sample=False)
    out['eval/seq'] = seq.numpy().astype(np.int64)
    return out


def run_kat(tmp, tokenizer):
    """Known-answer vectors of SURVEY.md section 8c, recomputed on the imported reference."""
    from models.model_pretrain_finetune_v0623_large_res import Pretrain
    from modules.loss import compute_lm_loss
    from modules import encoder_decoder as ED
    k = {}
    pt = Pretrain.__new__(Pretrain)
    pt.args = {'region_temp': 0.5, 'instance_temp': 0.5}
    a = np.array
    k['multi_pos'] = float(pt.multi_pos_contra_images_v0401(S.det((6, 8)), a(['a', 'b', 'c', 'd', 'a', 'c'])).item())
    k['multi_pos_nosib'] = pt.multi_pos_contra_images_v0401(S.det((4, 8)), a(['a', 'b', 'c', 'd'])).tolist()
    k['global_align'] = float(pt.global_alignment_loss(S.det((4, 8)), S.det((4, 8), a=.23, b=.31, c=1.0),
                                                       a(['a', 'b', 'a', 'd', 'a', 'c'])).item())
    k['local_align'] = float(pt.local_text_token_alignment_loss(S.det((2, 5, 8), a=.19),
                                                                S.det((2, 3, 8), a=.29, b=.07, c=.5)).item())
    lp = torch.log_softmax(S.det((2, 4, 5), a=.41), -1)
    k['lm_loss'] = float(compute_lm_loss(lp, torch.tensor([[3, 1, 4, 2], [3, 0, 2, 0]]),
                                         torch.tensor([[1, 1, 1, 1], [1, 1, 1, 0]])).item())
    k['r2_layernorm'] = ED.LayerNorm(8)(S.det((1, 8)))[0].tolist()
    k['subsequent_mask4'] = ED.subsequent_mask(4).int().tolist()
    k['pe_0_1'] = ED.PositionalEncoding(8, 0.0).pe[0, 1].tolist()
    rm = ED.RelationalMemory(3, 16, 4)
    k['rm_init_memory'] = rm.init_memory(2).tolist()
    k['tok_1'] = tokenizer.encode('[BOS] a 1.2-cm calcified granuloma, unchanged; no_pneumothorax zzzqq . [EOS]').ids
    k['tok_2'] = tokenizer.encode('[CLS] cardiomegaly [SEP] pleural effusion').ids
    k['tok_3'] = tokenizer.encode('Heart SIZE').ids
    k['tok_4'] = tokenizer.encode('[BOS] the heart size is normal . [EOS]').ids
    k['decode_1'] = tokenizer.decode([1442, 6, 20, 22, 8, 10, 5, 1443, 0, 0])
    k['vocab_size'] = tokenizer.get_vocab_size()
    k['bos_eos_pad'] = [tokenizer.token_to_id('[BOS]'), tokenizer.token_to_id('[EOS]'), tokenizer.token_to_id('[PAD]')]
    return k


def run_gpt2():
    """distilgpt2 backend (a20): in-container HF GPT2LMHeadModel with the EVOKE YAML sizes (3 x 2048 x 8 heads), procedural
    weights by key, encoder states from det(); stores the un-shifted CE loss, a logits tap, eval grads and beam sequences."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from oracle import gpt2 as G
    import torch.nn.functional as F
    V, d, layers, heads = 1444, 2048, 3, 8
    cfg = GPT2Config(vocab_size=V, n_embd=d, n_layer=layers, n_head=heads, add_cross_attention=True, is_decoder=True,
                     bos_token_id=V - 2, eos_token_id=V - 1, pad_token_id=0)
    m = GPT2LMHeadModel(cfg).eval()
    spec = G.gpt2_spec(V, d, layers)
    sd = {k[len(G.PRE):]: v for k, v in S.procedural_state(spec).items()}
    sd['lm_head.weight'] = sd['transformer.wte.weight']
    missing = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if 'bias' not in k.split('.')[-1] or 'attn.bias' not in k], missing
    inp = make_inputs(dict(kind='finetune', res=224, pids=[0, 1, 2], B=3, L=12, Li=0), V)
    enc = S.det((3, 50, d), a=.013, b=.007, c=.3) * 0.5
    out = {}
    ids, am = inp['ids'], inp['masks']
    for p in m.parameters():
        p.requires_grad_(True)
    lg = m(input_ids=ids, attention_mask=am, encoder_hidden_states=enc).logits
    loss = F.cross_entropy(lg.permute(0, 2, 1), ids, ignore_index=0)
    loss.backward()
    out['eval/loss'] = np.float64(loss.item())
    out['eval/tap/logits'] = reduce_tensor(lg)
    g = dict(m.named_parameters())
    for k in ('transformer.wte.weight', 'transformer.h.0.attn.c_attn.weight', 'transformer.h.1.crossattention.c_attn.weight',
              'transformer.h.2.mlp.c_proj.weight', 'transformer.h.0.crossattention.q_attn.weight', 'transformer.wpe.weight'):
        out['eval/grad/' + k] = reduce_tensor(g[k].grad)
    with torch.no_grad():
        for nb, ml in ((3, 16), (1, 10)):
            seq = m.generate(input_ids=torch.full((3, 1), V - 2), encoder_hidden_states=enc, num_beams=nb, max_length=ml, use_cache=True,
                             bos_token_id=V - 2, eos_token_id=V - 1, pad_token_id=0, do_sample=False)
            out['eval/seq_b%d' % nb] = seq.numpy().astype(np.int64)
    return out


def run_manifest(tmp, tokenizer):
    """Key / shape / dtype manifests of the imported reference's FineTune.state_dict(), of the optimizer modules/optimizers.py builds
    for it after one step (torch.optim.RAdam, two groups), and of the HF GPT2LMHeadModel(add_cross_attention) state_dict as a
    cvt2distilgpt2 checkpoint stores it (prefix decoder.encoder_decoder.decoder., language_model.py:215-220)."""
    import gzip
    from transformers import GPT2Config, GPT2LMHeadModel
    from models.model_pretrain_finetune_v0623_large_res import FineTune
    from modules.optimizers import build_two_stage_optimizer
    from evoke_amd import checkpoint as CK
    args = make_args(tmp, tokenizer)
    args.update(task='finetune', optim='RAdam', pt_lr=5e-6, ft_lr=5e-5, weight_decay=1e-4, amsgrad=True)
    model = FineTune(args, tokenizer, 'iu_xray')
    opt = build_two_stage_optimizer(args, model)
    inp = make_inputs(CASES['ft224_nomv'], tokenizer.get_vocab_size())
    inp2 = make_inputs(CASES['ft224_noinc'], tokenizer.get_vocab_size())
    for b in (inp, inp2):              # an indication and a no-indication step: every branch receives optimizer state
        ret = model(b['images'], b['ids'], b['masks'], np.array(b['patient_ids']), b.get('inc_ids'), b.get('inc_masks'), mode='train')
        ret['all_loss'].backward()
        opt.step()
        opt.zero_grad()
    V = tokenizer.get_vocab_size()
    cfg = GPT2Config(vocab_size=V, n_embd=GPT2_DIMS['d'], n_layer=GPT2_DIMS['layers'], n_head=GPT2_DIMS['heads'], add_cross_attention=True,
                     is_decoder=True)
    gsd = GPT2LMHeadModel(cfg).state_dict()
    out = {'finetune_state_dict': CK.manifest(model.state_dict()), 'finetune_optimizer': CK.optimizer_manifest(opt.state_dict()),
           'n_named_parameters': len(list(model.named_parameters())),
           'cvt2distilgpt2_decoder': [('decoder.encoder_decoder.decoder.' + k, s, d) for k, s, d in CK.manifest(gsd)]}
    with gzip.open(os.path.join(HERE, 'manifest_finetune.json.gz'), 'wt') as f:
        json.dump(out, f)
    print('manifest_finetune.json.gz written: %d state_dict entries, optimizer groups %s, %d optimizer states, %d decoder keys'
          % (len(out['finetune_state_dict']), [g['n_params'] for g in out['finetune_optimizer']['param_groups']],
             len(out['finetune_optimizer']['state']), len(out['cvt2distilgpt2_decoder'])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    only = set(a.only.split(',')) if a.only else None
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from modules.tokenizers_new import build_my_tokenizer
    tokenizer = build_my_tokenizer(os.path.join(REF, 'config/tokenizer'), data_name='iu_xray')
    tmp = os.path.join(tempfile.gettempdir(), 'evoke_golden')
    os.makedirs(tmp, exist_ok=True)
    if only is None or 'kat' in only:
        json.dump(run_kat(tmp, tokenizer), open(os.path.join(HERE, 'kat.json'), 'w'), indent=1)
        print('kat.json written')
    if only is None or 'manifest' in only:
        run_manifest(tmp, tokenizer)
    if only is None or 'gpt2' in only:
        np.savez_compressed(os.path.join(HERE, 'gpt2.npz'), **run_gpt2())
        print('gpt2.npz written')
    for name, case in CASES.items():
        if only is not None and name not in only:
            continue
        out = run_case(name, case, tmp, tokenizer)
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
        print(name, {k: (v.tolist() if getattr(v, 'ndim', 1) == 0 else v.shape) for k, v in out.items()
                     if 'loss' in k or 'seq' in k})


if __name__ == '__main__':
    main()
