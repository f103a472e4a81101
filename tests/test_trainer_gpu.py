"""Step harness on the GPU: FusedOptimizer state in torch.optim format (checkpoint interoperability with the reference's
`optimizer.state_dict()`, trainer_v0401.py:160-189) and a resumed FineTune run through evoke_amd.trainer.Trainer."""
import os

import pytest
import torch
import numpy as np
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
        captured = [sg.graph is not None and (sg.plan is not None or os.environ.get('EVK_STEP_GRAPH_MODE') == 'hipgraph') for _, sg in tr._graphs.values()]
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


def test_reference_order_step_with_a_torch_optimizer_follows_the_oracle():
    """The drop-in boundary under the reference's OWN step, nothing from evoke_amd.optim / evoke_amd.trainer:
        optimizer.zero_grad(); loss = model(...)['all_loss']; loss.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 0.1); optimizer.step()         (modules/trainer_v0401.py:428-435)
    with the optimizer modules/optimizers.py:27-46 builds (torch.optim.RAdam over the two name-split groups).  The fp16-storage
    build back-propagates under a loss scale; because no FusedOptimizer owns these parameters the scale must be gone from `p.grad`
    when backward() returns -- otherwise clip_grad_value_ clamps every element to +-0.1 and the run trains wrongly, silently.
    Four steps on the golden inputs (train-mode BN, dropout 0) against the CPU oracle doing the same four steps in fp32: the loss
    sequence must agree to 2e-3 and the per-step decrease to 10 %; the unclipped gradient maximum must be the oracle's (0.34, not
    0.34 x scale)."""
    from evoke_amd import ops, optim
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import functional as O, spec as S
    from tests.golden.cases import CASES, make_inputs
    from tests.helpers import ARGS, V, load_procedural, load_tokenizer
    lrs = (2e-5, 2e-4)
    case = CASES['ft224_inc']
    inp = make_inputs(case, V)
    spec = S.finetune_spec(V)
    ftk = ('text_decoder', 'visual_self_atten_layers', 'multimodal_fusion_layers', 'visual_head', 'text_head')      # optimizers.py:29-33

    # oracle trajectory (CPU, fp32)
    P = S.procedural_state(spec)
    keys = [k for k, (sh, kind) in spec.items() if kind not in S.BUFFER_KINDS]
    for k in keys:
        P[k].requires_grad_(True)
    oopt = torch.optim.RAdam([{'params': [P[k] for k in keys if not any(s in k for s in ftk)], 'lr': lrs[0]},
                              {'params': [P[k] for k in keys if any(s in k for s in ftk)], 'lr': lrs[1]}], weight_decay=1e-4)
    want, want_gmax = [], []
    for _ in range(4):
        oopt.zero_grad()
        ret = O.finetune_forward_train(P, inp['images'], inp['ids'], inp['masks'], inp['patient_ids'], inp['inc_ids'], inp['inc_masks'],
                                       O.DEFAULT_CFG, O.Ctx(train=True))
        want.append(ret['all_loss'].item())
        ret['all_loss'].backward()
        want_gmax.append(max(P[k].grad.abs().max().item() for k in keys if P[k].grad is not None))
        torch.nn.utils.clip_grad_value_([P[k] for k in keys], 0.1)
        oopt.step()

    # the engine under the reference's step
    ops.clear_grad_callbacks()
    ops.set_dropout_enabled(False)
    model = FineTune(dict(ARGS), load_tokenizer(), 'iu_xray')
    load_procedural(model, spec)
    model.train()
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    assert not any(ops.grads_owned(p) for _, p in named)
    opt = torch.optim.RAdam([{'params': [p for n, p in named if not any(s in n for s in ftk)], 'lr': lrs[0]},
                             {'params': [p for n, p in named if any(s in n for s in ftk)], 'lr': lrs[1]}], weight_decay=1e-4)
    images, ids, masks, pids = inp['images'].cuda(), inp['ids'].cuda(), inp['masks'].cuda(), np.array(inp['patient_ids'])
    got, got_gmax = [], []
    for _ in range(4):
        opt.zero_grad()
        ret = model(images, ids, masks, pids, inp['inc_ids'], inp['inc_masks'], mode='train')
        got.append(ret['all_loss'].item())
        ret['all_loss'].backward()
        got_gmax.append(max(float(p.grad.abs().max()) for _, p in named if p.grad is not None))
        torch.nn.utils.clip_grad_value_(model.parameters(), 0.1)
        opt.step()
    ops.set_dropout_enabled(True)
    print('\n[reference-order step] oracle %s\n                       engine %s\n   max |grad| before the clip: oracle %s engine %s (loss scale %g)'
          % (['%.5f' % v for v in want], ['%.5f' % v for v in got], ['%.3f' % v for v in want_gmax], ['%.3f' % v for v in got_gmax],
             ops.loss_scale_value()))
    from tests.helpers import F16_BUILD
    loss_tol, step_tol, g_tol = (2e-3, 0.10, 0.15) if F16_BUILD else (6e-2, 0.5, 0.4)      # bf16 storage: its train-mode tolerances (test_model_gpu.py)
    for a, b in zip(got, want):
        assert abs(a - b) <= loss_tol, (got, want)
    for i in range(3):
        assert abs((got[i] - got[i + 1]) - (want[i] - want[i + 1])) <= step_tol * abs(want[i] - want[i + 1]), (got, want)
    for a, b in zip(got_gmax, want_gmax):
        assert abs(a - b) <= g_tol * b, (got_gmax, want_gmax)


def test_unowned_gradients_overflow_surfaces_as_zero_gradients():
    """evk_grads_multi through ops._grads_multi: scan (mode 0) + unscale (mode 1) over several tensors of odd sizes and alignments;
    an inf anywhere zeroes EVERY gradient and the following evk_loss_scale_update halves the scale; mode 2 re-applies the scale."""
    from evoke_amd import ops
    sc = ops.loss_scaler()
    if sc is None:
        pytest.skip('bf16-storage build: no loss scaler')
    saved = sc.state.clone()
    try:
        sc.state.copy_(torch.tensor([512.0, 0.0, 0.0, 0.0]))
        g = torch.Generator().manual_seed(3)
        base = [torch.randn(n, generator=g) for n in (5, 70001, 131072, 3, 1 << 18)]
        big = torch.zeros(70001 + 8).cuda()
        grads = [base[0].cuda(), big[1:1 + 70001].copy_(base[1]), base[2].cuda(), base[3].cuda(), base[4].cuda().view(512, 512)]
        ref = [b.clone() for b in base]
        ops._grads_multi(grads, 0, sc)
        ops._grads_multi(grads, 1, sc)
        sc.update()
        for a, b in zip(grads, ref):
            assert torch.equal(a.cpu().reshape(-1), b / 512.0)
        assert sc.value() == 512.0 and float(big[0]) == 0.0 and float(big[70002:].abs().sum()) == 0.0
        ops._grads_multi(grads, 2, sc)
        for a, b in zip(grads, ref):
            assert torch.equal(a.cpu().reshape(-1), b)
        grads[1][12345] = float('inf')
        grads[3][2] = float('nan')
        ops._grads_multi(grads, 0, sc)
        ops._grads_multi(grads, 1, sc)
        sc.update()
        assert all(float(a.abs().sum()) == 0.0 for a in grads)
        assert sc.value() == 256.0 and sc.skipped() == int(saved[3].item() * 0) + 1
    finally:
        sc.state.copy_(saved)


def test_step_graphs_follow_eager_through_lr_changes_mixed_step_kinds_and_decodes():
    """Replayed steps against eager steps through everything a captured step must not bake in:
      * the learning rate changes between replays (an lr scheduler: the update kernel reads hyper-parameters from a device buffer);
      * indication / no-indication steps alternate: they touch different parameter subsets (multimodal_fusion_layers + text encoder
        vs visual_self_atten_layers), so per-parameter step counts diverge inside one flat group and every parameter must be updated
        with ITS OWN count (torch.optim semantics), whichever graph or eager step runs next;
      * a beam-search decode between replays fills the relational memory's cached q | k | v weights; after more replays the next decode
        must rebuild them (the optimizer kernel rewrites parameters through raw pointers: ops.WEIGHT_EPOCH is the only signal).
    Two models run in lock step, one eager, one through the step graphs, and after every step the graph model is re-seated on the eager
    model's state: whole trajectories of this random-init network are not reproducible even eager against eager (the last bits of the
    f32-atomic bias sums get amplified to 1e-2 in the loss within a few steps -- tools/graph_divergence.py,
    profiles/r03_a_graph_divergence.txt), single steps from a common state are."""
    from evoke_amd import distributed as D, ops, optim
    from evoke_amd.model_pretrain_finetune import FineTune
    from evoke_amd.trainer import Trainer
    from tests.helpers import ARGS, V, load_tokenizer
    args = dict(ARGS, task='finetune', pt_lr=5e-5, ft_lr=5e-4, optim='RAdam', weight_decay=5e-5, max_seq_len=12, beam_size=2)
    ops.set_dropout_enabled(False)
    ops.clear_grad_callbacks()

    def batch(i, inc):
        g = torch.Generator().manual_seed(11 * i + 1)
        ids = torch.randint(5, V - 2, (2, 12), generator=g)
        ids[:, 0] = V - 2
        b = [torch.randn(3, 3, 224, 224, generator=g), ids, torch.ones(2, 12, dtype=torch.long), ['q%d_s0' % i, 'q%d_s1' % i, 'q%d_s0' % i]]
        if inc:
            t = torch.randint(5, V - 2, (2, 6), generator=g)
            t[:, 0] = 1
            b += [t, torch.ones(2, 6, dtype=torch.long)]
        return tuple(b)

    def make(graphs):
        torch.manual_seed(5)
        m = FineTune(args, load_tokenizer(), 'mimic_cxr').cuda().train()
        o = optim.build_two_stage_optimizer(args, m, clip_value=0.1)
        return m, o, Trainer(m, o, dict(args, evk_step_graphs=graphs), reducer=D.GradReducer.for_optimizer(o), task='finetune', log=lambda s: None)

    m_e, o_e, t_e = make(False)
    m_g, o_g, t_g = make(True)

    def reseat():
        for sd, ss in zip(o_g.flat, o_e.flat):
            for k in ('p', 'm', 'v', 'steps', 'shadow'):
                sd[k].copy_(ss[k])
        for bd, bs in zip(m_g.buffers(), m_e.buffers()):
            bd.copy_(bs)
        ops.WEIGHT_EPOCH[0] += 1

    reseat()
    plan = [True] * 4 + [False] * 4 + ['lr'] + [True, False, True, False] + ['decode'] + [True, True] + ['decode']
    qkv, worst = [], 0.0
    for i, what in enumerate(plan):
        if what == 'lr':
            for o in (o_e, o_g):
                for g in o.param_groups:
                    g['lr'] *= 0.25
        elif what == 'decode':
            m_g.eval()
            b = batch(100, True)
            with torch.no_grad():
                m_g(b[0].cuda(), b[1].cuda(), b[2].cuda(), np.array(b[3]), b[4], b[5], mode='inference')
            rm = m_g.text_decoder.model.rm
            # the derived weights the decode step ran on: the f32 stack of the (default) f32 recurrence, or the 16-bit q | k | v cache of the
            # 16-bit recurrence (EVK_DECODE_RM_F32=0)
            from evoke_amd import decode as DEC
            if DEC._RM_F32[0]:
                fw = m_g.text_decoder.model._evk_fused_decode[1]
                want = torch.cat([rm.attn.linears[k].weight.detach().float() for k in range(3)], 0)
                assert torch.equal(fw.rm32_wqkv, want), 'decode ran on stale relational-memory weights'
            else:
                want = torch.cat([ops.shadow(rm.attn.linears[k].weight).view(512, 512) for k in range(3)], 0)
                assert rm._qkv_cache is not None and torch.equal(rm._qkv_cache[1], want), 'decode ran on stale relational-memory weights'
            qkv.append(want.float().clone())
            m_g.train()
        else:
            before = [st['p'].clone() for st in o_e.flat]
            le = float(t_e.train_step(batch(i, what))['all_loss'].reshape(-1)[0])
            lg = float(t_g.train_step(batch(i, what))['all_loss'].reshape(-1)[0])
            assert abs(le - lg) <= 2e-5 * max(1.0, abs(le)), (i, le, lg)
            for st_e, st_g, p0 in zip(o_e.flat, o_g.flat, before):
                de, dg = st_e['p'] - p0, st_g['p'] - p0
                rel = float((dg - de).norm() / (de.norm() + 1e-30))
                worst = max(worst, rel)
                assert rel <= 1e-3, 'step %d (%s): the replayed update differs from the eager one by %.3e of its norm' % (i, what, rel)
                assert torch.equal(st_e['steps'], st_g['steps']), 'per-parameter step counts differ after step %d' % i
            reseat()
    torch.cuda.synchronize()
    replays = [sg.graph is not None and sg.plan is not None for _, sg in t_g._graphs.values()]
    steps = sorted(set(int(e['step']) for e in o_g.state_dict()['state'].values()))
    ops.clear_grad_callbacks()
    ops.set_dropout_enabled(True)
    print('\n[graphs vs eager, lock step] worst relative difference of a replayed update: %.2e; per-parameter step counts %s' % (worst, steps))
    assert t_e._graphs == {} and replays == [True, True], replays         # both structures captured AND replayed through the C++ plan
    assert steps == [6, 8, 14], steps              # 8 indication + 6 plain steps; shared parameters took part in all 14
    assert not torch.equal(qkv[0], qkv[1]), 'the two decodes saw the same weights: the steps in between did not train'


