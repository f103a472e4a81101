"""GPU parity of the HIP engine's FineTune / Pretrain forward+backward against the golden fixtures that were
produced by the IMPORTED REFERENCE (tests/golden/make_golden.py): same procedural weights (by state_dict key),
same hashed inputs.  bf16 operands / f32 accumulation vs the reference's fp32 => tolerances are stated here:
   loss        |d| <= 1e-3 * max(1, |loss|)   (north_star: "loss parity to reference within 1e-3")
   activations relative error of the reduced taps <= 3e-2 (bf16 storage, ~100 layers)
   gradients   reduced taps <= 8e-2
"""
import os

import numpy as np
import pytest
import torch

from tests.golden.cases import CASES, compare_reduced, make_inputs, reduce_tensor
from tests.helpers import ARGS, GOLDEN, V, load_procedural, load_tokenizer

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-3
ACT_TOL = 3e-2
GRAD_TOL = 8e-2


def _gold(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def _hook(model, names, store):
    hs = []
    for tap, modname in names.items():
        def fn(m, inp, out, tap=tap):
            store[tap] = out
        hs.append(model.get_submodule(modname).register_forward_hook(fn))
    return hs


def _report(what, got, want, tol):
    ok, msg = compare_reduced(reduce_tensor(got.float()), want, tol)
    print('   %-28s %s %s' % (what, 'ok ' if ok else 'BAD', msg))
    return ok


@pytest.mark.parametrize('name', [n for n, c in CASES.items() if c['kind'] == 'finetune'])
def test_finetune_matches_reference(name):
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    case, gold = CASES[name], _gold(name)
    inp = make_inputs(case, V)
    args = dict(ARGS, is_multiview_learning=case.get('multiview', True))
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    spec = S.finetune_spec(V)
    ops.set_dropout_enabled(False)
    bad = []
    for mode in case['modes']:
        load_procedural(model, spec)
        model.train(mode == 'train')
        model.zero_grad(set_to_none=True)
        taps = {}
        fus = 'multimodal_fusion_layers.0' if inp['inc_ids'] is not None else 'visual_self_atten_layers.0'
        hs = _hook(model, {'resnet': 'visual_extractor', 'vhead': 'visual_head', 'fusion': fus}, taps)
        ret = model(inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'],
                    inp['inc_masks'], mode='train')
        for h in hs:
            h.remove()
        loss, want = ret['all_loss'].item(), float(gold[mode + '/loss'])
        print('\n[%s/%s] loss hip %.6f ref %.6f  diff %.2e' % (name, mode, loss, want, abs(loss - want)))
        if abs(loss - want) > LOSS_TOL * max(1.0, abs(want)):
            bad.append('%s loss %.6f vs %.6f' % (mode, loss, want))
        for tap, t in (('att', taps['resnet'][0]), ('fc', taps['resnet'][1]), ('vhead', taps['vhead']), ('enc_states', taps['fusion'])):
            if not _report(tap, t, gold['%s/tap/%s' % (mode, tap)], ACT_TOL):
                bad.append('%s %s' % (mode, tap))
        if mode == 'train':
            ret['all_loss'].backward()
            prm = dict(model.named_parameters())
            for k in gold.files:
                if k.startswith('train/grad/'):
                    g = prm[k[len('train/grad/'):]].grad
                    if g is None or not _report(k[11:], g, gold[k], GRAD_TOL):
                        bad.append(k)
            sd = model.state_dict()
            if not _report('bn running_mean', sd['visual_extractor.model.7.2.bn3.running_mean'], gold['train/bn/running_mean'], ACT_TOL):
                bad.append('running_mean')
            if not _report('bn running_var', sd['visual_extractor.model.7.2.bn3.running_var'], gold['train/bn/running_var'], ACT_TOL):
                bad.append('running_var')
    ops.set_dropout_enabled(True)
    assert not bad, bad


@pytest.mark.parametrize('name', [n for n, c in CASES.items() if c['kind'] == 'pretrain'])
def test_pretrain_matches_reference(name):
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import Pretrain
    from oracle import spec as S
    case, gold = CASES[name], _gold(name)
    inp = make_inputs(case, V)
    model = Pretrain(dict(ARGS), load_tokenizer(), 'iu_xray')
    spec = S.pretrain_spec(V)
    ops.set_dropout_enabled(False)
    bad = []
    for mode in case['modes']:
        load_procedural(model, spec)
        model.train(mode == 'train')
        model.zero_grad(set_to_none=True)
        taps = {}
        hs = _hook(model, {'resnet': 'visual_extractor', 'vhead': 'visual_head', 'thead': 'text_head'}, taps)
        ret = model(inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']))
        for h in hs:
            h.remove()
        print('\n[%s/%s]' % (name, mode))
        for k in ('sen_text_loss', 'instance_loss', 'multiview_loss', 'all_loss'):
            got, want = ret[k].reshape(-1)[0].item(), float(gold['%s/%s' % (mode, k)])
            print('   %-16s hip %.6f ref %.6f diff %.2e' % (k, got, want, abs(got - want)))
            if abs(got - want) > LOSS_TOL * max(1.0, abs(want)):
                bad.append('%s %s %.6f vs %.6f' % (mode, k, got, want))
        assert tuple(ret['sen_image_loss'].shape) == (1,)
        for tap, t in (('fc', taps['resnet'][1]), ('vhead', taps['vhead']), ('thead', taps['thead'])):
            if not _report(tap, t, gold['%s/tap/%s' % (mode, tap)], ACT_TOL):
                bad.append('%s %s' % (mode, tap))
        if mode == 'train':
            ret['all_loss'].backward()
            prm = dict(model.named_parameters())
            for k in gold.files:
                if k.startswith('train/grad/'):
                    g = prm[k[len('train/grad/'):]].grad
                    if g is None or not _report(k[11:], g, gold[k], GRAD_TOL):
                        bad.append(k)
    ops.set_dropout_enabled(True)
    assert not bad, bad
