"""GPU parity of the HIP engine's FineTune / Pretrain forward+backward against the golden fixtures that were
produced by the IMPORTED REFERENCE (tests/golden/make_golden.py): same procedural weights (by state_dict key),
same hashed inputs.  The engine computes with 16-bit operands / 16-bit activation storage / f32 accumulation, the
reference in fp32.  Tolerances of the DEFAULT build (fp16 storage, dynamic loss scale; measured values in DESIGN.md "Parity"):

  eval mode (BN running statistics):
     loss        |d| <= 1e-3        the north star's figure (measured 1.2e-4 .. 4.7e-4 on the golden cases)
     activations reduced taps (sum, sum of squares, 61 samples / rms) <= 2e-2   (measured <= 7e-3)
     gradients   66 tensors per case (every trunk stage, every transformer family, biases, norms): energy (sum of squares) within 10 %
                 and cosine of the 61-sample vector >= 0.975 (measured <= 6.6e-2; 300 of 301 comparisons >= 0.99, the lowest 0.979):
                 a 16-bit forward flips a few ReLU gates per layer, so deep-network gradients are not point-wise reproducible
  train mode (BN batch statistics):
     loss        |d| <= 1e-3        (measured 8e-5 .. 4.4e-4)
     BN running statistics vs the reference <= 2e-2
The bf16-storage build of the same kernels (EVK_STORE=bf16) is run by the last test in a child interpreter with ITS tolerances:
loss 5e-3 eval / 6e-2 train (bf16's 8-bit mantissa: a bias of the storage chain, amplified ~linearly in depth by train-mode BN on
random weights, see oracle/bf16_emulation.py), activations 8e-2, gradients 40 % / 0.5.
"""
import os

import numpy as np
import pytest
import torch

from tests.golden.cases import CASES, compare_grad, compare_reduced, make_inputs, reduce_tensor
from tests.helpers import ARGS, GOLDEN, V, load_procedural, load_tokenizer

pytestmark = pytest.mark.gpu

F16 = os.environ.get('EVK_STORE', 'f16').lower() == 'f16'      # the default build
LOSS_TOL = 1e-3 if F16 else 5e-3
LOSS_TOL_TRAIN = 1e-3 if F16 else 6e-2
ACT_TOL = 2e-2 if F16 else 8e-2
GRAD_TOL = 0.10 if F16 else 0.4
# Pretrain: the contrastive losses send gradients ~1e-4 of FineTune's into the trunk; at the initial loss scale (1024) the first bottlenecks'
# 16-bit activation gradients touch fp16's subnormals: energy error 12 % on layer1.0.conv2 at 224^2 (cosine 0.998), <= 6.6 % everywhere else
PT_GRAD_TOL = 0.15 if F16 else 0.4
# train-mode gradients (round 5: batch statistics in all 104 batch norms, i.e. the backward through the statistics as well): measured on the 67
# tensors of ft224_inc -- cosine >= 0.963 (the stem and the first bottlenecks: 0.963-0.98; everything outside the trunk >= 0.998), energy
# <= 4.1 % except the stem batch norm's bias (18 %: a 64-element vector at the end of 104 normalised layers)
TRAIN_GRAD_TOL, TRAIN_GRAD_COS = (0.25, 0.95) if F16 else (0.6, 0.4)
GRAD_COS = 0.975 if F16 else 0.5          # 66 tensors x 5 cases since round 3: measured 300 of 301 comparisons >= 0.99, the lowest 0.979

@pytest.fixture(autouse=True)
def _fresh_loss_scale():
    """the dynamic loss scale is per device and survives a test: start every parity test from the initial scale (an earlier test's
    overflow back-offs would push Pretrain's small trunk gradients towards fp16's subnormals: 12 % of their energy at scale 256)"""
    from evoke_amd import ops
    sc = ops.loss_scaler() if torch.cuda.is_available() else None
    if sc is not None:
        sc.state.copy_(torch.tensor([ops.LOSS_SCALE_INIT, 0.0, 0.0, 0.0]))
    yield


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


def _report_grad(what, got, want, tol=None, cos=None):
    ok, msg = compare_grad(reduce_tensor(got.float()), want, GRAD_TOL if tol is None else tol, GRAD_COS if cos is None else cos)
    print('   grad %-60s %s %s' % (what, 'ok ' if ok else 'BAD', msg))
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
        if abs(loss - want) > (LOSS_TOL if mode == 'eval' else LOSS_TOL_TRAIN):
            bad.append('%s loss %.6f vs %.6f' % (mode, loss, want))
        ret['all_loss'].backward()
        if mode == 'eval':
            for tap, t in (('att', taps['resnet'][0]), ('fc', taps['resnet'][1]), ('vhead', taps['vhead']), ('enc_states', taps['fusion'])):
                if not _report(tap, t, gold['%s/tap/%s' % (mode, tap)], ACT_TOL):
                    bad.append('%s %s' % (mode, tap))
            prm = dict(model.named_parameters())
            for k in gold.files:
                if k.startswith('eval/grad/'):
                    g = prm[k[len('eval/grad/'):]].grad
                    # (bf16 build, 100-token case: back-propagation through 100 recurrent steps on an 8-bit mantissa -- the relational
                    # memory's weight gradients measure 42 - 47 % energy error at cosine 0.97 - 0.99, the 16-token cases <= 36 %)
                    tol = 0.6 if (not F16 and case.get('logp')) else None
                    if g is None or not _report_grad(k[10:], g, gold[k], tol=tol):
                        bad.append(k)
        else:
            assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
            # train mode (batch statistics in all 104 batch norms, dropout off): the same 66 gradient tensors as in eval mode against the
            # reference's -- the backward through the batch statistics (the sum terms of bn_bwd_apply, the statistics epilogues of the
            # data-gradient kernels) is only exercised here
            prm = dict(model.named_parameters())
            for k in gold.files:
                if k.startswith('train/grad/'):
                    g = prm[k[len('train/grad/'):]].grad
                    if g is None or not _report_grad('(train) ' + k[11:], g, gold[k], tol=TRAIN_GRAD_TOL, cos=TRAIN_GRAD_COS):
                        bad.append(k)
            sd = model.state_dict()
            if not _report('bn running_mean', sd['visual_extractor.model.7.2.bn3.running_mean'], gold['train/bn/running_mean'], ACT_TOL):
                bad.append('running_mean')
            if not _report('bn running_var', sd['visual_extractor.model.7.2.bn3.running_var'], gold['train/bn/running_var'], ACT_TOL):
                bad.append('running_var')
    ops.set_dropout_enabled(True)
    assert not bad, bad


LOGP_TOL_TRAIN = 1e-2 if F16 else 8e-2


@pytest.mark.parametrize('name', [n for n, c in CASES.items() if c.get('logp')])
def test_training_forward_log_probabilities_per_position(name):
    """The teacher-forced TRAINING forward at the real report length (L = 100, 384^2, two views): the engine's log-probabilities -- of the
    target token and of 12 probe tokens -- against the imported reference's at EVERY position (tests/golden/ft384_L100.npz), not only
    their mean (the loss).  The relational memory is the same 100-step recurrence as in generation (modules/encoder_decoder.py:293-300)
    and expands perturbations on these weights, so a 16-bit recurrent state shows up here as an error that GROWS with the position;
    asserted <= 1e-2 at every position of every row."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from tests.golden.cases import PROBE_IDS
    from oracle import spec as S
    case, gold = CASES[name], _gold(name)
    inp = make_inputs(case, V)
    model = FineTune(dict(ARGS), load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    ops.set_dropout_enabled(False)
    try:
        model.eval()
        ids, masks = inp['ids'].cuda(), inp['masks'].cuda()
        with torch.no_grad():
            x, enc_mask = model.encoder_states(inp['images'].cuda(), np.array(inp['patient_ids']), case['B'], inp['inc_ids'], inp['inc_masks'])
            logits = model.text_decoder.forward_logits(ids, x, masks, enc_mask)
            lp = torch.log_softmax(logits[..., :V + 1].float(), -1).cpu().double()
    finally:
        ops.set_dropout_enabled(True)
    tgt = torch.cat([inp['ids'][:, 1:], torch.zeros(case['B'], 1, dtype=torch.long)], 1)
    got_t = lp.gather(2, tgt.unsqueeze(-1)).squeeze(-1).numpy()
    got_p = lp[:, :, PROBE_IDS].numpy()
    valid = inp['masks'].numpy().astype(bool)                          # query positions inside the report
    err_t = np.abs(got_t - gold['eval/logp_target']) * valid
    err_p = np.abs(got_p - gold['eval/logp_probe']).max(-1) * valid
    L = case['L']
    pos = [t for t in (0, 10, 30, 50, 70, 90, L - 4) if t < L]
    print('\n[%s] teacher-forced training forward, |log-probability - reference| per position %s:\n   target token %s\n   12 probe tokens (max) %s\n'
          '   largest over all positions: target %.2e, probes %.2e (tolerance %.0e)'
          % (name, pos, np.array2string(err_t[:, pos], precision=4), np.array2string(err_p[:, pos], precision=4), err_t.max(), err_p.max(), LOGP_TOL_TRAIN))
    if os.environ.get('EVK_TEST_DUMP'):
        np.savez(os.path.join(os.environ['EVK_TEST_DUMP'], 'train_logp_%s.npz' % name), err_t=err_t, err_p=err_p)
    assert err_t.max() <= LOGP_TOL_TRAIN and err_p.max() <= LOGP_TOL_TRAIN, (np.argwhere(err_p > LOGP_TOL_TRAIN)[:8].tolist(), err_t.max(), err_p.max())


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
            if abs(got - want) > (LOSS_TOL if mode == 'eval' else LOSS_TOL_TRAIN):
                bad.append('%s %s %.6f vs %.6f' % (mode, k, got, want))
        assert tuple(ret['sen_image_loss'].shape) == (1,)
        ret['all_loss'].backward()
        if mode == 'eval':
            for tap, t in (('fc', taps['resnet'][1]), ('vhead', taps['vhead']), ('thead', taps['thead'])):
                if not _report(tap, t, gold['%s/tap/%s' % (mode, tap)], ACT_TOL):
                    bad.append('%s %s' % (mode, tap))
            prm = dict(model.named_parameters())
            for k in gold.files:
                if k.startswith('eval/grad/'):
                    g = prm[k[len('eval/grad/'):]].grad
                    if g is None or not _report_grad(k[10:], g, gold[k], tol=PT_GRAD_TOL):
                        bad.append(k)
        else:
            assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
            prm = dict(model.named_parameters())          # train-mode gradients against the reference's (round 5), where the fixture holds them
            for k in gold.files:
                if k.startswith('train/grad/'):
                    g = prm[k[len('train/grad/'):]].grad
                    if g is None or not _report_grad('(train) ' + k[11:], g, gold[k], tol=max(PT_GRAD_TOL, TRAIN_GRAD_TOL), cos=TRAIN_GRAD_COS):
                        bad.append(k)
    ops.set_dropout_enabled(True)
    assert not bad, bad


@pytest.mark.parametrize('train', [True, False])
def test_trunk_follows_bf16_emulation(train):
    """HIP trunk vs the CPU emulation of its bf16 storage points (oracle/bf16_emulation.py) and vs the fp32 oracle."""
    from evoke_amd.trunk import ResNet
    from oracle import bf16_emulation as E, functional as O, spec as S
    from tests.helpers import rel_err
    inp = make_inputs(CASES['ft224_inc'], V)
    spec = {}
    S.resnet_spec(spec)
    m = ResNet({})
    m.load_state_dict({k[len('visual_extractor.'):]: v for k, v in S.procedural_state(spec).items()})
    m = m.cuda().train(train)
    with torch.no_grad():
        att, fc = m(inp['images'].cuda())
    emu = E.resnet101_trunk_bf16(S.procedural_state(spec), inp['images'], O.Ctx(train=train))
    ref = O.resnet101_trunk(S.procedural_state(spec), inp['images'], O.Ctx(train=train))
    n, c = emu.shape[:2]
    emu_p, ref_p = emu.reshape(n, c, -1).permute(0, 2, 1), ref.reshape(n, c, -1).permute(0, 2, 1)
    e_emu, e_ref, emu_vs_ref = rel_err(att.float(), emu_p), rel_err(att.float(), ref_p), rel_err(emu_p, ref_p)
    print('\n[trunk train=%s] HIP vs bf16-emulation %.3e | HIP vs fp32 %.3e | emulation vs fp32 %.3e' % (train, e_emu, e_ref, emu_vs_ref))
    assert e_emu <= 6e-2
    assert e_ref <= 1.3 * emu_vs_ref + 1e-2


LOGP_TOL = 8e-3 if F16 else 6e-2          # flat, at every position: |engine log-probability - reference| in the default decode mode (f32 relational memory)


@pytest.mark.parametrize('res', [224, 384])
def test_inference_trunk_with_batchnorm_epilogues_follows_the_eval_forward(res):
    """The inference forward of the visual extractor (eval mode under no_grad: the eval-mode batch norms as scale / shift in the convolution
    epilogues together with identity and ReLU, evk_trunk_forward_inference) against the unfused eval forward of the same module
    (EVK_FOLD_BN=-1: the training runner's conv, bn_finalize, bn_apply per layer), whose arithmetic every mode reproduces bit for bit, and against the fp32
    oracle; and the scale / shift vectors must follow the parameters: after a change of a running statistic the next call recomputes them."""
    from evoke_amd import trunk as T
    from evoke_amd.trunk import ResNet
    from oracle import functional as O, spec as S
    from tests.helpers import rel_err
    inp = make_inputs(CASES['ft224_inc' if res == 224 else 'ft384_inc'], V)
    spec = {}
    S.resnet_spec(spec)
    m = ResNet({})
    m.load_state_dict({k[len('visual_extractor.'):]: v for k, v in S.procedural_state(spec).items()})
    m = m.cuda().eval()
    img = inp['images'].cuda()
    saved = T.FOLD_BN[0]

    def run(fold):
        T.FOLD_BN[0] = fold
        try:
            with torch.no_grad():
                return m(img)[0]
        finally:
            T.FOLD_BN[0] = saved

    att_f = run(1)
    assert getattr(m.model, '_evk_fold', None) is not None, 'the inference path did not run'
    att_u = run(-1)                                            # the training runner (evk_trunk_forward, training = 0)
    for mode in (2, 0):                                        # identity convolutions unfused / nothing fused: the same bits again
        assert torch.equal(run(mode), att_u), 'inference trunk mode %d differs from the training runner' % mode
    ref = O.resnet101_trunk(S.procedural_state(spec), inp['images'], O.Ctx(train=False))
    n, c = ref.shape[:2]
    ref_p = ref.reshape(n, c, -1).permute(0, 2, 1)
    e_fu, e_f, e_u = rel_err(att_f.float(), att_u.float().cpu()), rel_err(att_f.float(), ref_p), rel_err(att_u.float(), ref_p)
    print('\n[inference trunk %d^2] fused vs unfused %.3e | fused vs fp32 oracle %.3e | unfused vs fp32 oracle %.3e' % (res, e_fu, e_f, e_u))
    assert torch.equal(att_f, att_u) and e_f <= (2e-3 if F16 else 2e-2), (e_fu, e_f, e_u)
    # the vectors follow the parameters
    with torch.no_grad():
        m.model.pairs()[5][1].running_var.mul_(4.0)
    att_2, att_2u = run(1), run(-1)
    assert torch.equal(run(2), att_2u)
    assert torch.equal(att_2, att_2u)
    assert rel_err(att_2.float(), att_f.float().cpu()) > 1e-3, 'the running statistic changed but the scale / shift vectors did not'
    # ... also when the native TRAINING forward moved the running statistics through raw pointers (no torch version counter moves, no
    # optimizer step in between: BN recalibration under no_grad) -- ADVICE round 4
    if res == 224:
        m.train()
        with torch.no_grad():
            m(img)
        m.eval()
        att_3, att_3u = run(2), run(-1)
        assert torch.equal(att_3, att_3u), 'stale batch-norm coefficients after a train-mode forward without an optimizer step'
        assert not torch.equal(att_3, att_2), 'the train-mode forward did not move the running statistics'


@pytest.mark.parametrize('name', [n for n, c in CASES.items() if c['kind'] == 'beam' and c['max_seq_len'] <= 40])
def test_beam_search_matches_reference(name):
    """FineTune.forward(mode='inference') end to end (incremental device-side beam search vs the reference's full re-decode) on the
    short goldens.  The engine's log-probabilities carry <= LOGP_TOL of 16-bit noise (asserted position by position in
    test_beam_search_follows_the_reference_decisions), and the reference's OWN margin between the last selected and the first rejected
    candidate is below that at several of these goldens' decisions (tests/golden/<name>_trace.npz: down to 5e-4 / 1.6e-3), so a study's
    token ids must be IDENTICAL to the reference's whenever every decision of that study was taken with a margin >= 2 x LOGP_TOL; for the
    others a different report is accepted only if the fp32 oracle scores it within 2 % of the reference's (teacher forced)."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import beam as OB, functional as O, spec as S
    case, gold, tr = CASES[name], _gold(name), _gold(name + '_trace')
    inp = make_inputs(case, V)
    args = dict(ARGS, max_seq_len=case['max_seq_len'], beam_size=case['beam_size'])
    tok = load_tokenizer()
    model = FineTune(args, tok, 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    with torch.no_grad():
        texts, seq = model(inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'],
                           inp['inc_masks'], mode='inference')
    seq = seq.cpu()
    want = torch.from_numpy(gold['eval/seq'])
    same = (seq == want).all(dim=1)
    beam = case['beam_size']
    min_margin = (tr['score'][:, :, beam - 1] - tr['score'][:, :, beam]).min(0)            # per study, over all decisions
    final_margin = tr['score'][-1, :, 0] - tr['score'][-1, :, 1]                            # ... and the choice of the returned beam
    decisive = np.minimum(min_margin, final_margin) >= 2 * LOGP_TOL
    print('\n[%s] identical sequences: %d / %d; smallest reference margin per study %s (final ranking %s)' % (
        name, int(same.sum()), len(same), np.round(min_margin, 5), np.round(final_margin, 5)))
    print('   hip', seq.tolist())
    print('   ref', want.tolist())
    assert seq.shape == want.shape and seq.dtype == torch.long
    assert texts == tok.decode_batch(seq.tolist()) or all(len(t) > 0 for t in texts)
    for b in range(len(same)):
        if decisive[b]:
            assert bool(same[b]), 'study %d: every reference decision had a margin >= %g, yet the ids differ' % (b, 2 * LOGP_TOL)
    if not bool(same.all()):
        # score both sequences with the fp32 oracle (teacher forced): the engine's choice must be as good as the reference's
        P = S.procedural_state(S.finetune_spec(V))
        cfg = dict(O.DEFAULT_CFG, max_seq_len=case['max_seq_len'], beam_size=case['beam_size'])
        with torch.no_grad():
            x, m = O.finetune_encoder_states(P, inp['images'], inp['patient_ids'], case['B'], inp['inc_ids'], inp['inc_masks'], cfg, O.Ctx())

            def score(sq):
                L = sq.shape[1]
                ids = torch.cat([torch.full((sq.shape[0], 1), V - 2), sq[:, :-1]], 1)
                lp = O.r2_forward_logprobs(P, ids, x, torch.ones_like(ids), m, cfg, O.Ctx())
                tokp = lp.gather(2, sq.unsqueeze(-1)).squeeze(-1)
                # sum until (and including) the first EOS or the end
                out = []
                for b in range(sq.shape[0]):
                    n = L
                    eos = (sq[b] == V - 1).nonzero()
                    if len(eos):
                        n = int(eos[0]) + 1
                    out.append(float(tokp[b, :n].sum()))
                return out
            s_h, s_r = score(seq), score(want)
        print('   oracle score of hip seq', s_h, ' of ref seq', s_r)
        for a, b in zip(s_h, s_r):
            assert a >= b - (0.02 if F16 else 0.05) * max(1.0, abs(b)), (a, b)


BEAM_TRACED = [n for n, c in CASES.items() if c['kind'] == 'beam' and os.path.exists(os.path.join(GOLDEN, n + '_trace.npz'))]


@pytest.mark.parametrize('name', BEAM_TRACED)
def test_beam_search_follows_the_reference_decisions(name):
    """Every beam golden -- BASELINE config 5 at its real decode length (384^2, two views, beam 4, max_seq_len 100) and the short 224^2
    cases -- against the imported reference's token ids (tests/golden/<name>.npz) and the decision trace the oracle wrote after
    reproducing those ids bit for bit (tests/golden/<name>_trace.npz: the best candidates per position and study, with the reference's
    running scores), in the engine's DEFAULT decode mode (relational memory in f32, csrc/rm_f32.hip).

    An untrained network never emits [EOS] and its candidates lie close together: the reference's OWN margin between the last selected
    and the first rejected candidate falls below 1e-2 at 9 / 15 of the 100 positions of the two config-5 studies (minimum 3.4e-3 /
    2.6e-4), which no engine with 16-bit operands can resolve.  So the test pins the search the way it can be pinned exactly:
      1. TEACHER-FORCED: the engine's beam search runs its product path (graph-replayed step, evk_beam_step bookkeeping) while a hook
         replaces every step's log-probabilities by the reference's selection, so that at all positions the engine holds exactly
         the reference's hypotheses.  At EVERY position the engine's log-probabilities of the traced candidates must agree with the
         reference's to the FLAT tolerance LOGP_TOL = 8e-3 (measured <= 5.3e-3), the engine's own top-beam set must equal the reference's
         wherever the reference's margin exceeds 2 x LOGP_TOL and at all but two positions of each study in any case, and the forced
         search must return the reference's ids exactly (bookkeeping, cache row tables, memory re-order).
      2. FREE-RUNNING: the ids must be identical to the reference's up to the first position whose margin is below 2 x LOGP_TOL, and
         the first divergent decision must be between candidates the reference scored within 2 x LOGP_TOL of each other."""
    from evoke_amd import decode
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    assert decode._RM_F32[0], 'the default decode mode carries the relational memory in f32'
    try:
        _beam100(decode, FineTune, S, name, True)
    finally:
        decode._SESSIONS.clear()


def test_beam_search_16bit_recurrence_stays_within_the_reference_16bit_drift():
    """The opt-in 16-bit relational-memory recurrence (EVK_DECODE_RM_F32=0; the default until round 3) on the 100-position golden: its
    log-probabilities drift with depth (0.5 nats at position 90) and are only held to four times what the REFERENCE's own arithmetic
    moves by on 16-bit operands (tests/golden/make_beam_trace.py::drift16).  Kept so that the mode stays correct; it is not the product
    default and no parity claim rests on it."""
    from evoke_amd import decode
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    saved_rm = decode._RM_F32[0]
    decode._RM_F32[0] = False
    decode._SESSIONS.clear()
    try:
        _beam100(decode, FineTune, S, 'beam384_b4_L100', False)
    finally:
        decode._RM_F32[0] = saved_rm
        decode._SESSIONS.clear()


def _beam100(decode, FineTune, S, name, rm_f32):
    case, gold, tr = CASES[name], _gold(name), _gold(name + '_trace')
    beam, T, B, V1 = case['beam_size'], case['max_seq_len'], case['B'], V + 1
    inp = make_inputs(case, V)
    args = dict(ARGS, max_seq_len=T, beam_size=beam)
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    flat = torch.from_numpy(tr['flat']).cuda()            # (T, B, K)
    score = torch.from_numpy(tr['score']).cuda()
    ref_lp = torch.from_numpy(tr['logp']).cuda()
    want = torch.from_numpy(gold['eval/seq'])
    margin = tr['score'][:, :, beam - 1] - tr['score'][:, :, beam]            # (T, B)
    # tolerance per position and study: four times what the REFERENCE's own arithmetic moves by when its decoder runs on 16-bit
    # operands (tests/golden/make_beam_trace.py::drift16: 4e-3 at position 10, 0.2 - 0.45 at position 99 -- the relational memory is
    # an expanding recurrence on these weights), never below 5e-3 (4e-2 in the bf16 build)
    tol = 4.0 * np.maximum.accumulate(tr['drift16'], axis=0) * (1.0 if F16 else 8.0) + (5e-3 if F16 else 4e-2)
    if rm_f32:
        tol = np.full_like(tol, LOGP_TOL)
    rec = {'err': [], 'same': []}

    def forced(t, logp, beam_sum):
        nb = 1 if t == 0 else beam
        ld = logp.shape[-1]
        parent, word = flat[t] // V1, flat[t] % V1                                              # (B, K)
        rows = parent + torch.arange(B, device='cuda').unsqueeze(1) * nb
        got = logp[rows, word]
        rec['err'].append((got - ref_lp[t]).abs().max(dim=1).values.cpu())
        cand = (beam_sum[:, :nb].unsqueeze(-1) + logp.view(B, nb, ld)[:, :, :V1]).reshape(B, -1)
        mine = cand.topk(beam, dim=1).indices.sort(dim=1).values
        rec['same'].append((mine == flat[t][:, :beam].sort(dim=1).values).all(dim=1).cpu())
        new = torch.full_like(logp, -1e30)
        new[rows[:, :beam], word[:, :beam]] = score[t][:, :beam] - beam_sum[:, :nb].gather(1, parent[:, :beam])
        logp.copy_(new)

    state = {'perm': torch.zeros(B, 1, dtype=torch.long, device='cuda'), 'first': [T] * B}

    def watch(t, logp, beam_sum):
        """free-running search: the engine's own selection, translated into the reference's beam numbering for as long as both hold the
        same hypotheses; records the first position at which the two selections differ"""
        nb = 1 if t == 0 else beam
        ld = logp.shape[-1]
        cand = (beam_sum[:, :nb].unsqueeze(-1) + logp.view(B, nb, ld)[:, :, :V1]).reshape(B, -1)
        mine = cand.topk(beam, dim=1).indices                                                   # engine order
        as_ref = state['perm'].gather(1, mine // V1) * V1 + mine % V1
        sel = flat[t][:, :beam]
        hit = as_ref.unsqueeze(2) == sel.unsqueeze(1)                                           # (B, mine, ref)
        ok = hit.any(dim=2).all(dim=1).cpu()
        for b in range(B):
            if state['first'][b] == T and not bool(ok[b]):
                state['first'][b] = t
        state['perm'] = hit.float().argmax(dim=2)

    with torch.no_grad():
        x, enc_mask = model.encoder_states(inp['images'].cuda(), np.array(inp['patient_ids']), B, inp['inc_ids'], inp['inc_masks'])
        seq_forced, p_forced = decode.beam_search(model.text_decoder, x, enc_mask, args, return_scores=True, step_hook=forced)
        assert decode.stats.get('graph') and decode.stats.get('fused_bookkeeping'), decode.stats
        seq_free = decode.beam_search(model.text_decoder, x, enc_mask, args, step_hook=watch)
        seq_plain = decode.beam_search(model.text_decoder, x, enc_mask, args)
    err = torch.stack(rec['err']).numpy()                 # (T, B)
    same = torch.stack(rec['same']).numpy()
    assert err.shape == (T, B)
    if os.environ.get('EVK_TEST_DUMP'):
        np.savez(os.path.join(os.environ['EVK_TEST_DUMP'], 'beam_forced.npz'), err=err, same=same, margin=margin, seq_forced=seq_forced.cpu().numpy(),
                 seq_free=seq_free.cpu().numpy(), first=np.array(state['first']))
    pos = [t for t in (10, 30, 50, 70, 90, 99) if t < T] or [T - 1]
    print('\n[%s] teacher-forced |logp - reference| (max over the 12 traced candidates) at positions %s:\n   engine    %s\n   tolerance %s\n'
          '   largest error / tolerance ratio %.2f; own top-%d set equals the reference at %s of %d positions'
          % (name, pos, np.array2string(err[pos].T, precision=4), np.array2string(tol[pos].T, precision=4), float((err / tol).max()), beam,
             same.sum(0), T))
    assert (err <= tol).all(), 'positions beyond the tolerance: %s' % np.argwhere(err > tol)[:10].tolist()
    bad = ~same & (margin > 2 * tol)
    assert not bad.any(), 'the engine selects another beam set where the reference margin is %s' % margin[bad]
    if rm_f32 and F16:
        assert (same.sum(0) >= T - 2).all(), same.sum(0)
    assert torch.equal(seq_forced.cpu(), want), 'forced search does not return the reference ids: bookkeeping / state re-order differ'
    np.testing.assert_allclose(p_forced.cpu().numpy(), tr['best_p'], atol=2e-2, rtol=0)
    assert torch.equal(seq_free, seq_plain), 'a read-only hook changed the search'
    free = seq_free.cpu()
    for b in range(B):
        first = state['first'][b]
        agree = float((free[b] == want[b]).float().mean())
        print('   study %d free-running: selections identical to the reference for %d positions%s; token agreement of the returned '
              'report %.2f' % (b, first, '' if first == T else ' (reference margin there: %.2e)' % margin[first, b], agree))
        if first == T:
            assert torch.equal(free[b], want[b])
        else:
            assert margin[first, b] < 2 * tol[first, b], 'the search leaves the reference at a decision the reference made with margin %g' % margin[first, b]


def test_distilgpt2_backend_matches_hf_fixture():
    """a20: the distilgpt2 cross-attention decoder backend vs the fixture produced by in-container HF GPT2LMHeadModel."""
    from evoke_amd import ops
    from evoke_amd.gpt2 import DistilGPT2TextDecoderModel
    from oracle import gpt2 as G, spec as S
    gold = _gold('gpt2')
    d, layers, heads = 2048, 3, 8
    args = dict(ARGS, decoder_hidden_size=d, decoder_num_hidden_layers=layers, decoder_num_attention_heads=heads, beam_size=3, max_seq_len=16)
    dec = DistilGPT2TextDecoderModel(args, load_tokenizer())
    sd = {k[len('text_decoder.'):]: v for k, v in S.procedural_state(G.gpt2_spec(V, d, layers)).items()}
    sd['decoder.encoder_decoder.decoder.lm_head.weight'] = sd['decoder.encoder_decoder.decoder.transformer.wte.weight']
    dec.load_state_dict(sd)
    dec = dec.cuda().eval()
    ops.set_dropout_enabled(False)
    inp = make_inputs(dict(kind='finetune', res=224, pids=[0, 1, 2], B=3, L=12, Li=0), V)
    enc = (S.det((3, 50, d), a=.013, b=.007, c=.3) * 0.5).to(ops.BF16).cuda()
    bad = []
    lg = dec.logits(inp['ids'].cuda(), inp['masks'].cuda(), enc)
    loss = dec(enc, None, inp['ids'].cuda(), inp['masks'].cuda(), stage='train')
    print('\n[gpt2] loss hip %.6f ref %.6f diff %.2e' % (loss.item(), float(gold['eval/loss']), abs(loss.item() - float(gold['eval/loss']))))
    if abs(loss.item() - float(gold['eval/loss'])) > LOSS_TOL:
        bad.append('loss')
    if not _report('logits', lg[..., :V], gold['eval/tap/logits'], ACT_TOL):
        bad.append('logits')
    ops.scale_loss(loss).backward()          # the decoder is called directly here: apply the model-level loss scale by hand
    prm = dict(dec.named_parameters())
    for k in gold.files:
        if k.startswith('eval/grad/'):
            g = prm['decoder.encoder_decoder.decoder.' + k[len('eval/grad/'):]].grad
            if g is None or not _report_grad(k[10:], g / ops.loss_scale_value(), gold[k]):
                bad.append(k)
    with torch.no_grad():
        seq = dec(enc, None, stage='test').cpu()
    want = torch.from_numpy(gold['eval/seq_b3'])
    print('   seq hip', seq.tolist())
    print('   seq ref', want.tolist())
    agree = sum(int(a.tolist() == b.tolist()) for a, b in zip(seq, want)) if seq.shape == want.shape else 0
    print('   identical sequences %d / %d' % (agree, len(want)))
    assert seq.dtype == torch.long and seq.shape[0] == 3 and bool((seq[:, 0] == V - 2).all())
    if F16:
        assert seq.shape == want.shape and torch.equal(seq, want), 'distilgpt2 beam generate differs from the HF fixture'
    ops.set_dropout_enabled(True)
    assert not bad, bad


def test_finetune_with_distilgpt2_decoder_matches_reference_composition():
    """BASELINE config 1 end to end: FineTune(args['text_decoder'] = 'distilgpt2') -- the switch the reference keeps in a comment
    (modules/utils.py:78) -- single view, 224^2, batch 2, against tests/golden/ft224_gpt2.npz (imported reference encoder +
    in-container HF GPT2LMHeadModel, see tests/golden/cases.py): eval loss, encoder-state tap, upstream and decoder gradients, and
    the beam-3 generated token ids (mode='inference') which must be identical in the default build."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import gpt2 as G, spec as S
    case, gold = CASES['ft224_gpt2'], _gold('ft224_gpt2')
    inp = make_inputs(case, V)
    d, layers, heads = 2048, 3, 8
    args = dict(ARGS, text_decoder='distilgpt2', decoder_hidden_size=d, decoder_num_hidden_layers=layers, decoder_num_attention_heads=heads,
                beam_size=case['beam_size'], max_seq_len=case['max_seq_len'])
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    sd = {k: v for k, v in S.procedural_state(S.finetune_spec(V)).items() if not k.startswith('text_decoder.')}
    sd.update(S.procedural_state(G.gpt2_spec(V, d, layers)))
    sd[G.PRE + 'lm_head.weight'] = sd[G.PRE + 'transformer.wte.weight']
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.endswith(('position_ids', '.attn.bias', '.attn.masked_bias')) for k in res.missing_keys), (res.unexpected_keys[:4], res.missing_keys[:4])
    model = model.cuda().eval()
    ops.set_dropout_enabled(False)
    bad = []
    taps = {}
    hs = _hook(model, {'fusion': 'multimodal_fusion_layers.0'}, taps)
    model.zero_grad(set_to_none=True)
    ret = model(inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'], inp['inc_masks'], mode='train')
    for h in hs:
        h.remove()
    loss, want = ret['all_loss'].item(), float(gold['eval/loss'])
    print('\n[ft224_gpt2/eval] loss hip %.6f ref %.6f  diff %.2e' % (loss, want, abs(loss - want)))
    if abs(loss - want) > LOSS_TOL:
        bad.append('loss %.6f vs %.6f' % (loss, want))
    if not _report('enc_states', taps['fusion'], gold['eval/tap/enc_states'], ACT_TOL):
        bad.append('enc_states')
    ret['all_loss'].backward()
    prm = dict(model.named_parameters())
    for k in gold.files:
        if k.startswith('eval/grad/'):
            g = prm[k[len('eval/grad/'):]].grad
            if g is None or not _report_grad(k[10:], g, gold[k]):
                bad.append(k)
    with torch.no_grad():
        texts, seq = model(inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'],
                           inp['inc_masks'], mode='inference')
    seq, wseq = seq.cpu(), torch.from_numpy(gold['eval/seq'])
    print('   seq hip', seq.tolist())
    print('   seq ref', wseq.tolist())
    assert len(texts) == case['B'] and seq.dtype == torch.long
    if F16:
        assert seq.shape == wseq.shape and torch.equal(seq, wseq), 'generated token ids differ from the HF fixture'
    # the serving loop over this backend (encoders of the next batch queued ahead of the host-driven generate): forward()'s results
    with torch.no_grad():
        b = (inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'], inp['inc_masks'])
        b2 = (b[0] * 1.05,) + b[1:]
        want2 = [model(*bb, mode='inference') for bb in (b, b2, b)]
        got2 = list(model.generate_pipelined([b, b2, b], mode='inference'))
    assert len(got2) == 3
    for (wt, ws), (gt, gs) in zip(want2, got2):
        assert wt == gt and torch.equal(ws.cpu(), gs.cpu()), 'generate_pipelined (distilgpt2) differs from forward(mode=inference)'
    ops.set_dropout_enabled(True)
    assert not bad, bad


def test_loss_parity_at_realistic_token_count():
    """Eval-mode loss at a realistic token count (8 studies x 2 views, <= 60 report tokens, 30 indication tokens) against the
    fp32 CPU oracle on the same procedural weights.  The north star's goal is 1e-3.  bf16 storage measures |d| = 3.5e-3 on a
    loss of 7.80 (4.5e-4 relative) -- the same as on the 40-token golden cases, i.e. a bias of the bf16 storage chain, not
    token noise (tools/precision_experiment.py: recomputing the final LayerNorm + logits + NLL in fp32 from the engine's
    decoder output leaves it unchanged) -- tolerance 5e-3; the fp16-storage build (EVK_STORE=f16) measures 2.2e-4,
    tolerance 1e-3."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import functional as O
    from oracle import spec as S
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    args = dict(ARGS)
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    ops.set_dropout_enabled(False)
    g = torch.Generator().manual_seed(21)
    B, L, Li = 8, 60, 30
    images = torch.randn(2 * B, 3, 224, 224, generator=g)
    ids = torch.randint(5, V - 2, (B, L), generator=g)
    ids[:, 0] = V - 2
    masks = torch.ones(B, L, dtype=torch.long)
    for i in range(B):
        ln = L - 4 * i
        ids[i, ln - 1] = V - 1
        ids[i, ln:] = 0
        masks[i, ln:] = 0
    inc = torch.randint(5, V - 2, (B, Li), generator=g)
    inc[:, 0] = 1
    incm = torch.ones(B, Li, dtype=torch.long)
    pids = np.array(['p%d_s%d' % (i % B, i % B) for i in range(2 * B)])
    with torch.no_grad():
        hip = model(images.cuda(), ids.cuda(), masks.cuda(), pids, inc, incm, mode='train')['all_loss'].item()
        P = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if not k.endswith('position_ids')}
        ref = O.finetune_forward_train(P, images, ids, masks, pids, inc, incm)['all_loss'].item()
    ops.set_dropout_enabled(True)
    print('\n[realistic] loss hip %.6f oracle %.6f diff %.2e' % (hip, ref, abs(hip - ref)))
    assert abs(hip - ref) <= LOSS_TOL, (hip, ref)



def test_full_size_step_properties():
    """BASELINE config 3 at its full size (384^2, 32 studies x 2 views = 64 images, L = 100, Li = 30): the oracle needs minutes
    there, so the step is pinned through properties that do not depend on the size.
      * eval mode is per-study: the loss of the 32-study batch equals the token-weighted mean of the losses of its two halves
        (LanguageModelCriterion = sum of masked NLL / sum of mask, modules/loss.py:5-22); only GEMM row partitioning differs,
        tolerance 1e-4 on a loss of ~7.4
      * train mode is repeatable: the same step from the same state twice gives bit-identical gradients for every parameter of the
        trunk (fixed reduction orders: split-K slabs, column-sum partial rows, side streams joined by events) and for every GEMM-produced
        weight gradient; the loss and the parameters reduced with f32 atomics agree to 1e-5.  (This test found a one-in-10^6 stale read in
        the gate-statistic accumulators, see csrc/gemm.hip and tests/test_abi.py.)
      * every gradient is finite and the trunk, the text encoder and the decoder all receive one."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    model = FineTune(dict(ARGS), load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    g = torch.Generator().manual_seed(33)
    B, L, Li = 32, 100, 30
    images = torch.randn(2 * B, 3, 384, 384, generator=g).cuda()
    ids = torch.randint(5, V - 2, (B, L), generator=g)
    ids[:, 0] = V - 2
    masks = torch.ones(B, L, dtype=torch.long)
    for i in range(B):
        ln = L - (3 * i) % 41
        ids[i, ln - 1] = V - 1
        ids[i, ln:] = 0
        masks[i, ln:] = 0
    inc = torch.randint(5, V - 2, (B, Li), generator=g)
    inc[:, 0] = 1
    incm = torch.ones(B, Li, dtype=torch.long)
    ids, masks = ids.cuda(), masks.cuda()

    def run(sl, train):          # anchors first, then the second views (the collate order of the reference's loader)
        n = sl.stop - sl.start
        img = torch.cat([images[sl], images[B + sl.start:B + sl.stop]])
        pids = np.array(['p%d_s%d' % (sl.start + i % n, sl.start + i % n) for i in range(2 * n)])
        return model(img, ids[sl], masks[sl], pids, inc[sl], incm[sl], mode='train')['all_loss']

    model.eval()
    ops.set_dropout_enabled(False)
    with torch.no_grad():
        full = run(slice(0, B), False).item()
        h0, h1 = run(slice(0, B // 2), False).item(), run(slice(B // 2, B), False).item()
    n0, n1 = float(masks[:B // 2, 1:].sum()), float(masks[B // 2:, 1:].sum())
    halves = (h0 * n0 + h1 * n1) / (n0 + n1)
    print('\n[full size] eval loss %.6f, token-weighted halves %.6f (|d| %.2e)' % (full, halves, abs(full - halves)))
    assert np.isfinite(full) and abs(full - halves) <= 1e-4, (full, h0, h1)

    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    params = [p for p in model.parameters() if p.requires_grad]
    outs = []
    sc = ops.loss_scaler()                      # a fixed loss scale for both runs: earlier tests leave the dynamic one wherever they drove it
    saved_scale = None if sc is None else sc.state.clone()
    if sc is not None:
        sc.state.copy_(torch.tensor([256.0, 0.0, 0.0, 0.0]))
    for _ in range(2):
        model.load_state_dict(state)
        for p in params:
            p.grad = None
        torch.manual_seed(5)
        loss = run(slice(0, B), True)
        loss.backward()                 # the model's forward already attached the loss scale to the graph (ops.scale_loss)
        ops.join_side_streams()
        torch.cuda.synchronize()
        outs.append((loss.item(), [None if p.grad is None else p.grad.detach().clone() for p in params]))
    ops.set_dropout_enabled(True)
    if sc is not None:
        sc.state.copy_(saved_scale)
    (l0, g0), (l1, g1) = outs
    # the bias / LayerNorm-parameter column sums and the embedding scatter accumulate with f32 atomics (as the reference's CUDA kernels
    # do): their last bits depend on arrival order -- 1e-5 relative; the loss and everything the trunk runner produces are bit-stable
    assert l0 == l1, (l0, l1)            # the LM criterion sums per-row values with a tree reduction: bit-stable
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    got = {'visual_extractor': 0, 'text_encoder': 0, 'text_decoder': 0}
    exact = loose = 0
    gmax = max(float(a.norm()) for a in g0 if a is not None)
    for n, a, b in zip(names, g0, g1):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        assert bool(torch.isfinite(a).all()), n
        if torch.equal(a, b):
            exact += 1
        else:
            assert not n.startswith('visual_extractor'), 'trunk gradient of %s differs between two runs of the same step' % n
            # (a key bias has a zero gradient in exact arithmetic -- softmax is shift invariant -- so its buffer holds rounding noise only:
            # the bound is relative to the parameter's own norm plus 1e-7 of the largest gradient norm of the step)
            d, lim = float((a - b).norm()), 1e-5 * float(a.norm()) + 1e-7 * gmax
            assert d <= lim, 'gradient of %s differs by %.3e (limit %.3e) between two runs of the same step' % (n, d, lim)
            loose += 1
        for k in got:
            if k in n and float(a.abs().sum()) > 0:
                got[k] += 1
    print('[full size] train loss %.6f / %.6f; %d gradients bit-identical, %d within 1e-5 (atomic column sums); with gradient: %s' % (l0, l1, exact, loose, got))
    assert exact >= 300
    assert all(v > 0 for v in got.values()), got



def test_full_size_beam_decode_properties():
    """BASELINE config 5 at its full size (384^2, 64 studies x 2 views, beam 4, max_seq_len 100) through FineTune.forward(mode='inference'):
      * integer output, so two runs of the same batch must give IDENTICAL token ids (graph replays, device-side beam bookkeeping)
      * beam search is per study (modules/caption_model.py:26-202 keeps every study's beams apart): decoding the first 32 studies alone
        must reproduce their rows of the 64-study result; the GEMMs see a different row count, so a near-tie may flip -- at least 99 % of
        the token positions must agree and at least 90 % of the studies must match exactly
      * every sequence stays inside the vocabulary and is zero after its first 0 (the reference's end-of-sequence convention)."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    args = dict(ARGS)
    args['beam_size'], args['max_seq_len'] = 4, 100
    model = FineTune(args, load_tokenizer(), 'mimic_cxr')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    g = torch.Generator().manual_seed(44)
    B, Li = 64, 30
    images = torch.randn(2 * B, 3, 384, 384, generator=g).cuda()
    ids = torch.zeros(B, 100, dtype=torch.long).cuda()
    masks = torch.ones(B, 100, dtype=torch.long).cuda()
    inc = torch.randint(5, V - 2, (B, Li), generator=g)
    inc[:, 0] = 1
    incm = torch.ones(B, Li, dtype=torch.long)

    def run(n):
        img = torch.cat([images[:n], images[B:B + n]])
        pids = np.array(['p%d_s%d' % (i % n, i % n) for i in range(2 * n)])
        with torch.no_grad():
            out = model(img, ids[:n], masks[:n], pids, inc[:n], incm[:n], mode='inference')
        seq = out[1] if isinstance(out, (tuple, list)) else out
        return seq.detach().cpu()

    a, b = run(B), run(B)
    assert a.shape[0] == B and a.dtype == torch.long
    assert torch.equal(a, b), 'beam search produced different token ids on the same batch'
    assert int(a.min()) >= 0 and int(a.max()) < V + 1
    for row in a:
        z = (row == 0).nonzero()
        if len(z):
            assert int(row[int(z[0]):].abs().sum()) == 0
    h = run(B // 2)
    agree = float((h == a[:B // 2]).float().mean())
    exact = float((h == a[:B // 2]).all(dim=1).float().mean())
    print('\n[full size] decode: repeatable; half batch vs full batch: %.4f of the tokens, %.3f of the studies identical; mean length %.1f'
          % (agree, exact, float((a != 0).sum(1).float().mean())))
    assert agree >= 0.99 and exact >= 0.90, (agree, exact)


EDGE = {
    # name: (views per study, report length L, true lengths, indication length Li, indication true lengths)
    'one_study_one_view': ([1], 12, [12], 6, [6]),
    'four_views_and_one': ([4, 1], 20, [20, 9], 8, [8, 1]),
    'max_seq_len': ([2, 2], 100, [100, 57], 30, [30, 30]),
    'ragged_everything': ([1, 3, 2], 33, [33, 5, 17], 11, [2, 11, 6]),
}


@pytest.mark.parametrize('name', list(EDGE))
def test_edge_geometries_match_oracle(name):
    """Eval-mode FineTune loss against the fp32 oracle on the geometries the reference's collate can produce but the golden
    cases do not hold: a single study with a single view (no siblings: the multi-view fusion is skipped for it,
    models/model_pretrain_finetune_v0623_large_res.py:126-150), studies with 4 / 3 views next to single-view ones, a report
    at max_seq_len = 100, ragged report and indication lengths down to 1 real indication token."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import functional as O
    from oracle import spec as S
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    views, L, lens, Li, ilens = EDGE[name]
    B = len(views)
    model = FineTune(dict(ARGS), load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    ops.set_dropout_enabled(False)
    g = torch.Generator().manual_seed(77)
    images = torch.randn(sum(views), 3, 224, 224, generator=g)
    pids = np.array(['p%d_s%d' % (i, i) for i, v in enumerate(views) for _ in range(v)])
    ids = torch.randint(5, V - 2, (B, L), generator=g)
    ids[:, 0] = V - 2
    masks = torch.ones(B, L, dtype=torch.long)
    for i, ln in enumerate(lens):
        ids[i, ln - 1] = V - 1
        ids[i, ln:] = 0
        masks[i, ln:] = 0
    inc = torch.randint(5, V - 2, (B, Li), generator=g)
    inc[:, 0] = 1
    incm = torch.ones(B, Li, dtype=torch.long)
    for i, ln in enumerate(ilens):
        inc[i, ln:] = 0
        incm[i, ln:] = 0
    with torch.no_grad():
        hip = model(images.cuda(), ids.cuda(), masks.cuda(), pids, inc, incm, mode='train')['all_loss'].item()
        P = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if not k.endswith('position_ids')}
        ref = O.finetune_forward_train(P, images, ids, masks, pids, inc, incm)['all_loss'].item()
    ops.set_dropout_enabled(True)
    print('\n[%s] loss hip %.6f oracle %.6f diff %.2e' % (name, hip, ref, abs(hip - ref)))
    assert np.isfinite(hip) and abs(hip - ref) <= LOSS_TOL, (hip, ref)


@pytest.mark.skipif(not F16, reason='already running in the bf16-storage build')
def test_bf16_storage_build_passes_the_gpu_suite():
    """The bf16-storage build of the same kernels (EVK_STORE=bf16 -> libevoke_hip_bf16.so) in a child interpreter (the storage format is
    fixed per process) with that build's tolerances (module docstring): every model-level parity test against its reference and one
    representative per kernel family (tests/conftest.py: BF16_CHILD_SELECT), unscaled gradients (the unit-level contrastive-gradient check
    runs only there).  When the whole GPU suite is collected the child was started at the end of collection and has been running beside this
    process's tests (tests/conftest.py); here it is only joined.  A hand-picked run of this test starts it now."""
    from tests import conftest as CT
    if not CT.BF16_CHILD:
        import gc
        from evoke_amd import trunk
        for v in trunk._WsLease._pool.values():      # hand this process's idle HBM back before the child allocates its own
            del v[:]
        gc.collect()
        torch.cuda.empty_cache()
        CT.start_bf16_child()
    proc, log = CT.BF16_CHILD['proc'], CT.BF16_CHILD['log']
    try:
        rc = proc.wait(timeout=1500)
    finally:
        if proc.poll() is None:
            proc.kill()
    log.flush()
    log.seek(0)
    tail = '\n'.join(log.read().splitlines()[-30:])
    print(tail)
    log.close()
    os.unlink(log.name)
    CT.BF16_CHILD.clear()
    assert rc == 0, tail


def test_pipelined_generation_equals_per_batch_inference():
    """FineTune.generate_pipelined (encoders of batch k+1 on a second stream while batch k is decoded) returns, batch for batch, exactly
    what forward(mode='inference') returns: three different batches (different images, view structure and indications)."""
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    args = dict(ARGS, max_seq_len=24, beam_size=3)
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    batches = []
    for k, name in enumerate(('beam224', 'ft224_inc', 'beam224_b4')):
        inp = make_inputs(CASES[name], V)
        batches.append((inp['images'].cuda() * (1.0 + 0.1 * k), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'],
                        inp['inc_masks']))
    with torch.no_grad():
        want = [model(*b, mode='inference') for b in batches]
        got = list(model.generate_pipelined(batches, mode='inference'))
        again = list(model.generate_pipelined(batches, mode='sample'))
    assert len(got) == 3
    for (wt, ws), (gt, gs), (st, sg) in zip(want, got, again):
        assert torch.equal(ws.cpu(), gs.cpu()), 'pipelined generation changed the token ids'
        assert wt == gt == st
        assert isinstance(sg, list) and len(sg) == len(gt)


@pytest.mark.parametrize('env', [
    {'EVK_DECODE_THREADS': '0'},                                                   # the default: four searches in flight, the calling thread issues every token step round-robin
    {'EVK_DECODE_THREADS': '1', 'EVK_DECODE_DEPTH': '2', 'EVK_DECODE_BURST': '1', 'EVK_DECODE_AHEAD': '1'},     # host threads, two searches, one step per native call
    {'EVK_DECODE_THREADS': '1', 'EVK_DECODE_DEPTH': '3', 'EVK_DECODE_BURST': '5'},     # three searches in flight, bursts that do not divide the loop
    {'EVK_DECODE_THREADS': '1', 'EVK_EXPERIMENTAL': '1', 'EVK_ENC_RESERVE_CUS': '8'},                       # encoders on a CU-masked stream (hip.masked_stream)
    {'EVK_DECODE_THREADS': '1', 'EVK_FOLD_BN': '1'},                               # batch norms in every convolution epilogue that has one
    {'EVK_DECODE_THREADS': '1', 'EVK_FOLD_BN': '0'},                               # inference runner, nothing fused
    {'EVK_DECODE_THREADS': '1', 'EVK_FOLD_BN': '-1'},                              # the training runner in eval mode (before round 4)
], ids=['one_thread', 'threads_burst1', 'depth3_burst5', 'cu_mask', 'fused_bn', 'unfused_runner', 'training_runner'])
def test_pipelined_generation_modes_return_the_same_ids(env, monkeypatch):
    """Every way generate_pipelined can drive the searches (host threads issuing native bursts of token steps / one thread round-robin,
    2, 3 or 4 searches in flight, encoders on a CU-masked stream, every setting of the inference trunk) returns the ids of
    forward(mode='inference') over the training runner, five batches of three different structures, in batch order."""
    from evoke_amd import trunk as T
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args = dict(ARGS, max_seq_len=20, beam_size=3)
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    batches = []
    for k, name in enumerate(('beam224', 'ft224_inc', 'beam224_b4', 'beam224', 'ft224_inc')):
        inp = make_inputs(CASES[name], V)
        batches.append((inp['images'].cuda() * (1.0 + 0.05 * k), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'],
                        inp['inc_masks']))
    with torch.no_grad():
        monkeypatch.setattr(T, 'FOLD_BN', [-1])
        want = [model(*b, mode='inference')[1].cpu() for b in batches]
        monkeypatch.setattr(T, 'FOLD_BN', [int(env.get('EVK_FOLD_BN', '2'))])
        got = [seq.cpu() for _, seq in model.generate_pipelined(batches, mode='inference')]
    assert len(got) == len(want)
    for k, (w, g) in enumerate(zip(want, got)):
        assert torch.equal(w, g), 'batch %d: %s changed the token ids' % (k, env)


def test_pipelined_generation_with_asynchronously_uploaded_batches():
    """The serving loop as a loader drives it: every batch is uploaded with `.to(device, non_blocking=True)` from pinned memory on the
    CALLER's stream when the generator asks for it -- behind a long-running kernel, so that the copy lands late -- and dropped by the
    caller at once.  generate_pipelined must order its encoder stream after that upload for EVERY batch (not once at the start) and keep
    the blocks alive for the stream that reads them: results equal forward(mode='inference') batch for batch."""
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    args = dict(ARGS, max_seq_len=16, beam_size=3)
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    host = []
    for k, name in enumerate(('beam224', 'ft224_inc', 'beam224_b4', 'beam224')):
        inp = make_inputs(CASES[name], V)
        host.append(((inp['images'] * (1.0 + 0.07 * k)).pin_memory(), inp['ids'].pin_memory(), inp['masks'].pin_memory(), np.array(inp['patient_ids']),
                     inp['inc_ids'], inp['inc_masks']))
    with torch.no_grad():
        want = [model(b[0].cuda(), b[1].cuda(), b[2].cuda(), *b[3:], mode='inference')[1].cpu() for b in host]

        def loader():
            for b in host:
                torch.cuda._sleep(40_000_000)                    # ~20 ms of GPU time in front of the copies on the caller's stream
                yield (b[0].to('cuda', non_blocking=True), b[1].to('cuda', non_blocking=True), b[2].to('cuda', non_blocking=True), *b[3:])
                # (the caller keeps no reference: the blocks return to the allocator as soon as generate_pipelined lets go of them)
        got = [seq.cpu() for _, seq in model.generate_pipelined(loader(), mode='inference')]
    assert len(got) == len(want)
    for k, (w, g) in enumerate(zip(want, got)):
        assert torch.equal(w, g), 'batch %d: pipelined generation over asynchronously uploaded inputs changed the token ids' % k


def test_beam_session_follows_the_encoder_mask_of_every_batch():
    """A persistent beam session replays a captured per-token step that holds POINTERS: the encoder key mask must live in a buffer the
    session owns and be refreshed per batch.  Two consecutive searches over the same encoder states with different non-trivial masks:
    the second must equal what a fresh session returns for its mask (and differ from the first)."""
    from evoke_amd import decode
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    case = CASES['beam224']
    inp = make_inputs(case, V)
    args = dict(ARGS, max_seq_len=12, beam_size=3)
    model = FineTune(args, load_tokenizer(), 'iu_xray')
    load_procedural(model, S.finetune_spec(V))
    model.eval()
    with torch.no_grad():
        x, m = model.encoder_states(inp['images'].cuda(), np.array(inp['patient_ids']), case['B'], inp['inc_ids'], inp['inc_masks'])
        T = x.shape[1]
        m_a, m_b = torch.ones(x.shape[0], T, dtype=torch.long, device='cuda'), torch.ones(x.shape[0], T, dtype=torch.long, device='cuda')
        m_a[:, T // 2:] = 0                                      # first batch: the second half of the patches is masked out
        m_b[:, 1:T // 3] = 0                                     # second batch: the first third
        decode._SESSIONS.clear()
        seq_a = decode.beam_search(model.text_decoder, x, m_a, args).cpu()
        del m_a                                                  # the first mask's memory may be reused
        filler = torch.zeros(x.shape[0], T, dtype=torch.uint8, device='cuda')      # noqa: F841 -- takes the freed block
        seq_b = decode.beam_search(model.text_decoder, x, m_b, args).cpu()          # same session (same geometry), new mask
        decode._SESSIONS.clear()
        fresh_b = decode.beam_search(model.text_decoder, x, m_b, args).cpu()
        decode._SESSIONS.clear()
    assert torch.equal(seq_b, fresh_b), 'the replayed step read a stale encoder mask'
    assert not torch.equal(seq_a, seq_b), 'the two masks should lead to different reports (test is vacuous otherwise)'


@pytest.mark.skipif(not F16, reason='the bf16-storage build has fp32\'s exponent range')
def test_forward_overflow_of_the_fp16_build_is_reported_not_propagated():
    """fp16 storage has a range cliff (+-65504) that the loss scale cannot repair in the FORWARD.  A freshly initialised network in eval
    mode (batch-norm running statistics 0 / 1 through 33 un-normalised bottlenecks) reaches it: the engine must say so -- a RuntimeError
    that names EVK_STORE=bf16 -- instead of returning reports decoded from NaN features; the same network in train mode (batch statistics)
    stays in range and must NOT trip the guard."""
    from evoke_amd import ops
    from evoke_amd.model_pretrain_finetune import FineTune
    torch.manual_seed(3)
    case = CASES['beam224']
    inp = make_inputs(case, V)
    args = dict(ARGS, max_seq_len=6, beam_size=2)
    model = FineTune(args, load_tokenizer(), 'iu_xray').cuda()
    big = inp['images'].cuda() * 8.0
    model.eval()
    with pytest.raises(RuntimeError, match='EVK_STORE=bf16'):
        with torch.no_grad():
            model(big, inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'], inp['inc_masks'], mode='inference')
    ops.check_forward_guard(block=True)              # the verdict was consumed: nothing pending
    model.train()
    ret = model(inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids']), inp['inc_ids'], inp['inc_masks'], mode='train')
    torch.cuda.synchronize()
    ops.check_forward_guard(block=True)              # train-mode batch statistics keep every activation O(1): no report
    assert torch.isfinite(ret['all_loss']).item()
