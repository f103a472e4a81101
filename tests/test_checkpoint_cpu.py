"""Checkpoint interoperability against manifests written from the IMPORTED REFERENCE
(tests/golden/make_golden.py --only manifest -> manifest_finetune.json.gz): state_dict keys / shapes / dtypes in order, the
optimizer's group partition and per-index state shapes (modules/optimizers.py:27-46), the reference's checkpoint file format
(modules/trainer_v0401.py:160-202) and the cvt2distilgpt2 key map (models/language_encoder/language_model.py:215-220)."""
import gzip
import json
import os

import pytest
import torch

from evoke_amd import checkpoint as CK

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def man():
    with gzip.open(os.path.join(GOLDEN, 'manifest_finetune.json.gz'), 'rt') as f:
        return json.load(f)


@pytest.fixture(scope='module')
def ft(tokenizer):
    return _finetune(tokenizer)


def _finetune(tokenizer, **kw):
    from evoke_amd.model_pretrain_finetune import FineTune
    from evoke_amd.config import ARGS
    return FineTune(dict(ARGS, task='finetune', optim='RAdam', pt_lr=5e-6, ft_lr=5e-5, weight_decay=1e-4, **kw), tokenizer, 'iu_xray')


def test_state_dict_manifest_equals_the_reference(man, ft):
    model = ft
    mine = [e for e in CK.manifest(model.state_dict()) if not e[0].endswith('position_ids')]
    ref = [tuple(e) for e in man['finetune_state_dict']]
    assert len(ref) == 1029 and len(mine) == len(ref)
    for (k0, s0, d0), (k1, s1, d1) in zip(mine, ref):
        assert (k0, s0, d0) == (k1, s1, d1), ((k0, s0, d0), (k1, s1, d1))           # same keys, same ORDER, same shapes and dtypes
    assert len(list(model.named_parameters())) == man['n_named_parameters']


def test_optimizer_partition_and_state_format_equal_the_reference(man, ft):
    from evoke_amd import ops, optim
    model = ft
    groups = optim.split_param_groups(model.args, model)
    ref = man['finetune_optimizer']
    assert [len(named) for _, named in groups] == [g['n_params'] for g in ref['param_groups']] == [427, 277]
    flat = [p for _, named in groups for _, p in named]
    for i, ent in ref['state'].items():                     # index i of the reference's optimizer state = the same parameter here
        assert list(flat[int(i)].shape) == ent['exp_avg'] == ent['exp_avg_sq'], (i, flat[int(i)].shape, ent)
    # a reference-format optimizer state_dict loads into the engine's optimizer and comes back in the same format
    opt = optim.FusedOptimizer(groups, kind='RAdam', weight_decay=1e-4)
    g = torch.Generator().manual_seed(3)
    sd = {'state': {}, 'param_groups': []}
    idx = 0
    for (lr, named), rg in zip(groups, ref['param_groups']):
        sd['param_groups'].append({'lr': lr * 0.5, 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 1e-4, 'params': list(range(idx, idx + len(named)))})
        idx += len(named)
    for i, ent in ref['state'].items():
        i = int(i)
        sd['state'][i] = {'step': torch.tensor(float(3 + i % 2)), 'exp_avg': torch.randn(ent['exp_avg'], generator=g) * 1e-3,
                          'exp_avg_sq': torch.rand(ent['exp_avg_sq'], generator=g) * 1e-6}
    opt.load_state_dict(sd)
    back = opt.state_dict()
    assert set(back['state']) == set(sd['state'])
    assert [len(gp['params']) for gp in back['param_groups']] == [427, 277] and back['param_groups'][0]['lr'] == groups[0][0] * 0.5
    for i in (0, 5, 300, 426, 427, 500, 701):
        if i in sd['state']:
            for k in ('exp_avg', 'exp_avg_sq'):
                assert torch.equal(back['state'][i][k], sd['state'][i][k]), (i, k)
            assert float(back['state'][i]['step']) == float(sd['state'][i]['step'])
    assert set(CK.optimizer_manifest(back)['state'][0]) == set(ref['state']['0'])
    ops.clear_grad_callbacks()


def test_reference_checkpoint_file_round_trip(tmp_path, ft):
    """{'epoch', 'state_dict', 'optimizer', 'monitor_best'} (trainer_v0401.py:161-166): written by the engine's Trainer, read back by key
    (resume) and through the key-and-shape filter (stage 1 -> stage 2 warm start)."""
    from evoke_amd import ops, optim
    from evoke_amd.trainer import Trainer
    model = ft
    opt = optim.build_two_stage_optimizer(model.args, model)
    tr = Trainer(model, opt, dict(model.args, result_dir=str(tmp_path)), task='finetune', log=lambda s: None)
    path = tr.save_checkpoint(4, save_best=True)
    ck = torch.load(path, map_location='cpu')
    assert sorted(ck) == ['epoch', 'monitor_best', 'optimizer', 'state_dict'] and ck['epoch'] == 4
    assert os.path.exists(os.path.join(os.path.dirname(path), 'model_best.pth'))
    assert sorted(ck['optimizer']) == ['param_groups', 'state']
    saved = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    tr.resume_checkpoint(path)
    assert tr.start_epoch == 5
    for k, v in model.state_dict().items():
        assert torch.equal(v, saved[k]), k
    # stage 1 -> stage 2 warm start: a file holding a foreign key and a tensor of another shape is filtered, the rest loads
    ck['state_dict']['visual_global.head.0.weight'] = torch.zeros(3)
    ck['state_dict']['layer_norm_1.weight'] = torch.zeros(7)
    torch.save(ck, path)
    invalid = tr.load_checkpoint(path)
    assert set(invalid) == {'visual_global.head.0.weight', 'layer_norm_1.weight'}
    ops.clear_grad_callbacks()


def test_cvt2distilgpt2_key_map_and_warm_start(man, tokenizer):
    """A synthetic checkpoint in cvt2distilgpt2's layout (generator under decoder.encoder_decoder.decoder.*, a CvT encoder under
    encoder.*, HF 4.x's persistent mask buffers, the tied lm_head) warm-starts the distilgpt2 backend: every generator tensor lands in
    the parameter of the same name under text_decoder.*, nothing else is touched, what cannot be used is reported."""
    from evoke_amd.model_pretrain_finetune import FineTune
    from evoke_amd.config import ARGS
    args = dict(ARGS, task='finetune', text_decoder='distilgpt2', decoder_hidden_size=2048, decoder_num_hidden_layers=3,
                decoder_num_attention_heads=8)
    model = FineTune(args, tokenizer, 'iu_xray')
    before = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    sd = {}
    for k, shape, dtype in man['cvt2distilgpt2_decoder']:
        sd[k] = torch.randn(shape, generator=g) * 0.02
    sd['decoder.encoder_decoder.decoder.transformer.h.0.attn.bias'] = torch.ones(1, 1, 1024, 1024, dtype=torch.uint8)
    sd['decoder.encoder_decoder.decoder.transformer.h.0.attn.masked_bias'] = torch.tensor(-1e4)
    sd['encoder.cvt.encoder.stages.0.embedding.convolution_embeddings.projection.weight'] = torch.zeros(64, 3, 7, 7)
    sd['decoder.encoder_decoder.decoder.transformer.h.1.mlp.c_fc.bias'] = torch.zeros(17)          # wrong shape: reported invalid
    rep = CK.load_cvt2distilgpt2(model, {'state_dict': sd, 'epoch': 9})
    after = model.state_dict()
    n_tensors = sum(1 for k, _, _ in man['cvt2distilgpt2_decoder'] if not k.endswith(('lm_head.weight', '.attn.bias', '.attn.masked_bias', '.crossattention.bias')))
    assert len(rep['loaded']) == n_tensors - 1 and rep['invalid'] == ['text_decoder.decoder.encoder_decoder.decoder.transformer.h.1.mlp.c_fc.bias']
    assert any(k.startswith('encoder.cvt') for k in rep['skipped']) and any(k.endswith('attn.masked_bias') for k in rep['skipped'])
    for k in rep['loaded']:
        src = sd['decoder.encoder_decoder.decoder.' + k[len(CK.ENGINE_GPT2_PREFIX):]]
        assert torch.equal(after[k], src), k
    for k in after:
        if k not in rep['loaded'] and not k.endswith('lm_head.weight'):          # (the head is the token embedding: it moves with it)
            assert torch.equal(after[k], before[k]), k                           # nothing outside the generator moved
    # the head is tied to the token embedding, as in GPT2LMHeadModel
    assert model.text_decoder.gpt2.lm_head.weight is model.text_decoder.gpt2.transformer.wte.weight or \
        torch.equal(model.text_decoder.gpt2.lm_head.weight, model.text_decoder.gpt2.transformer.wte.weight)
