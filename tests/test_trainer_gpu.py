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
            ops.grad_buffer(p).add_(g.cuda() * ops.loss_scale_value())      # gradient buffers carry the loss scale (1 in the bf16 build)
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


def test_dynamic_loss_scale_skips_the_whole_step_on_overflow():
    """Device-side dynamic loss scaling (csrc/eltwise.hip: evk_grad_nonfinite / evk_optim_step_dyn / evk_optim_bump /
    evk_loss_scale_update) driven through the C ABI: a non-finite element anywhere in the flat gradient skips EVERY parameter of
    the step, leaves the per-parameter step counts alone and halves the scale; clean steps follow torch.optim.RAdam fed the
    unscaled, world-averaged gradients, and the scale doubles after `interval` clean steps."""
    from evoke_amd import hip as H
    n, world, scale0, interval = 4096, 4, 512.0, 3
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * 0.05 for _ in range(6)]
    ref = nn.Parameter(p0.clone().cuda())
    ropt = torch.optim.RAdam([ref], lr=5e-3, weight_decay=1e-4)
    p, m, v = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    steps = torch.zeros(2, dtype=torch.int32).cuda()          # two "parameters" sharing the launch
    state = torch.tensor([scale0, 0.0, 0.0, 0.0]).cuda()
    scales, skipped_at = [], (1, 4)
    for i, gr in enumerate(grads):
        cur = float(state[0])
        summed = (gr * world * cur).cuda()                     # what the all-reduce SUM of `world` scaled shard gradients holds
        if i in skipped_at:
            summed[n // 2] = float('inf') if i == 1 else float('nan')
        before = p.clone()
        H.check(H.lib.evk_grad_nonfinite(H.ptr(summed), n, H.ptr(state), H.stream()))
        H.check(H.lib.evk_optim_step_dyn(H.ptr(p), H.ptr(summed), H.ptr(m), H.ptr(v), None, None, n, 0, 5e-3, 0.9, 0.999, 1e-8, 1e-4, 0.1,
                                         H.ptr(steps), H.ptr(state), 1.0 / world, 0, H.stream()))
        H.check(H.lib.evk_optim_bump(H.ptr(steps), 2, H.ptr(state), H.stream()))
        H.check(H.lib.evk_loss_scale_update(H.ptr(state), 2.0, 0.5, interval, 1.0, 65536.0, H.stream()))
        if i in skipped_at:
            assert torch.equal(p, before), 'a skipped step moved parameters'
        else:
            ref.grad = gr.cuda().clamp(-0.1, 0.1)
            ropt.step()
        scales.append(float(state[0]))
    assert steps.tolist() == [4, 4] and float(state[3]) == 2.0 and float(state[2]) == 0.0
    # 512 -> ok 512 -> inf 256 -> ok, ok (2 good) -> nan 128 -> ok (1 good)
    assert scales == [512.0, 256.0, 256.0, 256.0, 128.0, 128.0], scales
    assert torch.allclose(p, ref.detach(), rtol=2e-5, atol=2e-6)
    # growth: `interval` clean steps double the scale
    for _ in range(interval):
        H.check(H.lib.evk_loss_scale_update(H.ptr(state), 2.0, 0.5, interval, 1.0, 65536.0, H.stream()))
    assert float(state[0]) == 256.0, float(state[0])


def test_fused_optimizer_overflow_skip_keeps_the_torch_trajectory():
    """FusedOptimizer.step() with an overflowed gradient in ONE parameter: no parameter of any group moves, state_dict step counts
    do not advance, and the following clean steps continue on torch.optim.RAdam's trajectory (fp16-storage build only: the
    bf16 build has no scaler)."""
    from evoke_amd import ops, optim
    if ops.loss_scaler() is None:
        pytest.skip('bf16-storage build: no loss scaler')
    init = _params(3)
    grads = [[torch.randn(*q.shape, generator=torch.Generator().manual_seed(31 * s + i)) * 0.05 for i, q in enumerate(init)] for s in range(3)]
    ref = [nn.Parameter(q.clone().cuda()) for q in init]
    ropt = torch.optim.RAdam(ref, lr=5e-3)
    mine = [nn.Parameter(q.clone().cuda()) for q in init]
    fo = optim.FusedOptimizer([(5e-3, [('p%d' % i, q) for i, q in enumerate(mine)])], kind='RAdam', clip_value=0.1)
    sc0 = ops.loss_scale_value()
    for s in range(3):
        for attempt in range(2 if s == 1 else 1):                # step 1 is first tried with an overflow, then repeated clean
            fo.zero_grad()
            for i, (q, gq) in enumerate(zip(mine, grads[s])):
                gsc = gq.cuda() * ops.loss_scale_value()
                if s == 1 and attempt == 0 and i == 2:
                    gsc.view(-1)[5] = float('inf')
                ops.grad_buffer(q).add_(gsc)
                ops.grad_done(q)
            before = [q.detach().clone() for q in mine]
            fo.step()
            if s == 1 and attempt == 0:
                assert all(torch.equal(a, b.detach()) for a, b in zip(before, mine))
                assert ops.loss_scale_value() == sc0 * 0.5
        for q, gq in zip(ref, grads[s]):
            q.grad = gq.cuda().clamp(-0.1, 0.1)
        ropt.step()
    assert [int(e['step']) for e in fo.state_dict()['state'].values()] == [3, 3, 3, 3]
    for a, b in zip(mine, ref):
        assert torch.allclose(a.detach().cpu(), b.detach().cpu(), rtol=2e-5, atol=2e-6)
    ops.loss_scaler().state.copy_(torch.tensor([sc0, 0.0, 0.0, 0.0]))          # leave the process-wide scaler as found
    ops.clear_grad_callbacks()


@pytest.mark.parametrize('task', ['finetune', 'pretrain'])
def test_step_graph_replays_the_eager_trajectory(task):
    """Whole-step HIP-graph capture (evoke_amd/graph.py) through Trainer.train_step: six optimizer steps on batches of ONE
    structure (different data each step) -- two eager warm-up calls, the capture, three replays -- follow the same loss sequence
    and end at the same parameters as six eager steps of an identically initialised model (dropout off; the only
    non-determinism left is the summation order of f32 atomics)."""
    from evoke_amd import distributed as D, ops, optim
    from evoke_amd.model_pretrain_finetune import FineTune, Pretrain
    from evoke_amd.trainer import Trainer
    from tests.helpers import ARGS, V, load_tokenizer
    args = dict(ARGS, task=task, pt_lr=5e-5, ft_lr=5e-4, optim='RAdam', weight_decay=5e-5, amsgrad=True)
    ops.set_dropout_enabled(False)

    def batch(i):
        g = torch.Generator().manual_seed(11 * i + 1)
        ids = torch.randint(5, V - 2, (2, 12), generator=g)
        ids[:, 0] = V - 2 if task == 'finetune' else 1
        b = [torch.randn(3, 3, 224, 224, generator=g), ids, torch.ones(2, 12, dtype=torch.long), ['q%d_s0' % i, 'q%d_s1' % i, 'q%d_s0' % i]]
        if task == 'finetune':
            inc = torch.randint(5, V - 2, (2, 6), generator=g)
            inc[:, 0] = 1
            b += [inc, torch.ones(2, 6, dtype=torch.long)]
        return tuple(b)

    def run(graphs):
        ops.clear_grad_callbacks()
        torch.manual_seed(5)
        m = (FineTune if task == 'finetune' else Pretrain)(args, load_tokenizer(), 'mimic_cxr').cuda().train()
        o = optim.build_two_stage_optimizer(args, m, clip_value=0.1)
        tr = Trainer(m, o, dict(args, evk_step_graphs=graphs), reducer=D.GradReducer.for_optimizer(o), task=task, log=lambda s: None)
        losses = [float(tr.train_step(batch(i))['all_loss'].reshape(-1)[0]) for i in range(6)]
        torch.cuda.synchronize()
        captured = [sg.graph is not None or (sg.failed is not None and 'multi-dimensional memcpy' in str(sg.failed)) for _, sg in tr._graphs.values()]
        for _, sg in tr._graphs.values():
            if sg.failed is not None:      # torch issued a 2-D device copy inside the step: the plan is refused, the step stays eager
                print('   step capture refused: %s' % sg.failed)
        steps = [int(e['step']) for e in o.state_dict()['state'].values()]
        return losses, [st['p'].detach().clone() for st in o.flat], captured, steps

    l_e, p_e, cap_e, steps_e = run(False)
    l_g, p_g, cap_g, steps_g = run(True)
    ops.clear_grad_callbacks()
    ops.set_dropout_enabled(True)
    print('\n[step graph %s] eager %s\n                   graph %s' % (task, ['%.5f' % v for v in l_e], ['%.5f' % v for v in l_g]))
    assert cap_e == [] and cap_g == [True], (cap_e, cap_g)
    assert steps_e == steps_g and max(steps_g) == 6
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(a)), (l_e, l_g)
    for a, b in zip(p_e, p_g):
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-5), float((a - b).abs().max())
