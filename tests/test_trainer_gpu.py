"""Step harness on the GPU: FusedOptimizer state in torch.optim format (checkpoint interoperability with the reference's
`optimizer.state_dict()`, trainer_v0401.py:160-189) and a resumed FineTune run through evoke_amd.trainer.Trainer."""
import os

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(16, 24), (24,), (8, 4, 3, 3), (5, 7)]
    return [torch.randn(*s, generator=g) for s in shapes]


@pytest.mark.parametrize('kind', ['RAdam', 'AdamW'])
def test_fused_optimizer_state_dict_round_trip(kind):
    from evoke_amd import ops, optim
    init = _params()
    grads = [[torch.randn(*p.shape, generator=torch.Generator().manual_seed(10 * s + i)) * 0.05 for i, p in enumerate(init)] for s in range(4)]

    def torch_opt(ps):
        if kind == 'RAdam':
            return torch.optim.RAdam(ps, lr=5e-3)
        return torch.optim.Adam(ps, lr=5e-3, amsgrad=True)

    # reference trajectory: 4 torch steps (gradients clipped to 0.1 like the trainer does)
    ref = [nn.Parameter(p.clone().cuda()) for p in init]
    ropt = torch_opt(ref)
    ref_state_after2 = None
    for s in range(4):
        for p, g in zip(ref, grads[s]):
            p.grad = g.cuda().clamp(-0.1, 0.1)
        ropt.step()
        if s == 1:
            import copy
            ref_state_after2 = copy.deepcopy(ropt.state_dict())

    def fused(ps):
        return optim.FusedOptimizer([(5e-3, [('p%d' % i, p) for i, p in enumerate(ps)])], kind=kind, amsgrad=kind == 'AdamW', clip_value=0.1)

    def fstep(opt, ps, gs):
        opt.zero_grad()
        for p, g in zip(ps, gs):
            ops.grad_buffer(p).add_(g.cuda() * ops.LOSS_SCALE)      # gradient buffers carry the loss scale (1 in the bf16 build)
            ops.grad_done(p)
        opt.step()

    mine = [nn.Parameter(p.clone().cuda()) for p in init]
    fo = fused(mine)
    fstep(fo, mine, grads[0])
    fstep(fo, mine, grads[1])
    sd = fo.state_dict()
    assert set(sd) == {'state', 'param_groups'} and sd['param_groups'][0]['params'] == [0, 1, 2, 3]
    for i in range(4):
        for k in ref_state_after2['state'][i]:
            a, b = sd['state'][i][k], ref_state_after2['state'][i][k]
            assert torch.allclose(torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu(), rtol=1e-5, atol=1e-7), (i, k)

    # a fresh engine optimizer resumed from the TORCH state continues on the torch trajectory
    ops.clear_grad_callbacks()
    again = [nn.Parameter(p.detach().clone()) for p in mine]
    fo2 = fused(again)
    fo2.load_state_dict(ref_state_after2)
    fstep(fo2, again, grads[2])
    fstep(fo2, again, grads[3])
    for a, b in zip(again, ref):
        assert torch.allclose(a.detach().cpu(), b.detach().cpu(), rtol=2e-5, atol=2e-6)
    ops.clear_grad_callbacks()


def test_trainer_finetune_steps_and_resume(tmp_path):
    from evoke_amd import distributed as D, ops, optim
    from evoke_amd.model_pretrain_finetune import FineTune
    from evoke_amd.trainer import Trainer
    from tests.helpers import ARGS, V, load_tokenizer
    args = dict(ARGS, task='finetune', pt_lr=5e-5, ft_lr=5e-4, optim='RAdam', weight_decay=5e-5, amsgrad=True,
                result_dir=str(tmp_path), monitor_mode='max', monitor_metric='BLEU_4', save_period=1)

    def batches(inc, n):
        out = []
        for i in range(n):
            g = torch.Generator().manual_seed(7 * i + (3 if inc else 0))
            ids = torch.randint(5, V, (2, 12), generator=g)
            ids[:, 0] = V - 2
            b = [['s%d' % i] * 2, torch.randn(2, 3, 224, 224, generator=g), ids, torch.ones(2, 12, dtype=torch.long), ['p%d_s%d' % (i, j) for j in range(2)]]
            if inc:
                b += [torch.randint(5, V, (2, 6), generator=g), torch.ones(2, 6, dtype=torch.long)]
            out.append(tuple(b))
        return out

    def make():
        ops.clear_grad_callbacks()
        torch.manual_seed(5)
        m = FineTune(args, load_tokenizer(), 'mimic_cxr').cuda()
        o = optim.build_two_stage_optimizer(args, m, clip_value=0.1)
        return m, o, D.GradReducer.for_optimizer(o)

    m, o, red = make()
    tr = Trainer(m, o, args, reducer=red, log=lambda s: None)
    log1 = tr.train_epoch_finetune(batches(True, 2), batches(False, 2), epoch=1)
    assert log1['train_loss'] > 0 and log1['train_loss'] == log1['train_loss']
    tr.end_of_epoch(1, {'val_BLEU_4': 0.1})
    ck = os.path.join(str(tmp_path), 'checkpoint', 'current_checkpoint.pth')
    log2 = tr.train_epoch_finetune(None, batches(False, 2), epoch=2)

    m2, o2, red2 = make()
    tr2 = Trainer(m2, o2, dict(args, resume=ck), reducer=red2, log=lambda s: None)
    assert tr2.start_epoch == 2
    log2b = tr2.train_epoch_finetune(None, batches(False, 2), epoch=2)
    # same weights + optimizer state + inputs (dropout 0): the resumed epoch reproduces the original one
    assert abs(log2b['train_loss'] - log2['train_loss']) <= 2e-2 * abs(log2['train_loss']), (log2b, log2)
    ops.clear_grad_callbacks()
