"""Which parameters leave the eager trajectory, and at which step, when training steps are replayed from captured graphs with mixed
step kinds (indication / no indication)?  Prints, per step, the parameters whose values differ between the two runs.
usage: python tools/graph_divergence.py [iiiinnnninin]   (EVK_STEP_GRAPH_MODE=hipgraph: hipGraphLaunch instead of the C++ replayer)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from evoke_amd import distributed as D, ops, optim
from evoke_amd.model_pretrain_finetune import FineTune
from evoke_amd.trainer import Trainer
from tests.helpers import ARGS, V, load_tokenizer

args = dict(ARGS, task='finetune', pt_lr=5e-5, ft_lr=5e-4, optim='RAdam', weight_decay=5e-5, max_seq_len=12, beam_size=2)
ops.set_dropout_enabled(False)
plan = [c if c == 'L' else c == 'i' for c in (sys.argv[1] if len(sys.argv) > 1 else 'iiiinnnninin')]   # i = indication step, n = no indication, L = lr x 0.25


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


EPS = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8


def run(graphs):
    ops.clear_grad_callbacks()
    torch.manual_seed(5)
    m = FineTune(args, load_tokenizer(), 'mimic_cxr').cuda().train()
    o = optim.FusedOptimizer(optim.split_param_groups(args, m), kind='RAdam', weight_decay=5e-5, clip_value=0.1, eps=EPS)
    tr = Trainer(m, o, dict(args, evk_step_graphs=graphs), reducer=D.GradReducer.for_optimizer(o), task='finetune', log=lambda s: None)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    rec = []
    for i, what in enumerate(plan):
        if what == 'L':
            for g_ in o.param_groups:
                g_['lr'] *= 0.25
            rec.append(rec[-1])
            continue
        loss = float(tr.train_step(batch(i, what))['all_loss'].reshape(-1)[0])
        torch.cuda.synchronize()
        sums = {n: (float(p.detach().double().sum()), float(p.detach().double().abs().sum())) for n, p in m.named_parameters() if p.requires_grad}
        steps = [st['steps'].cpu().tolist() for st in o.flat]
        gsum = [float(st['g'].abs().sum()) for st in o.flat]
        rec.append((loss, sums, steps, gsum))
    return names, rec


names, e = run(False)
_, e2 = run(False)
_, g = run(True)
print('eps', EPS, '| eager run 1 vs eager run 2 losses:', [('%.6f' % a[0], '%.6f' % b[0]) for a, b in zip(e, e2) if a[0] != b[0]] or 'identical')
for i, (a, b) in enumerate(zip(e, g)):
    bad = [n for n in names if abs(a[1][n][0] - b[1][n][0]) > 1e-7 * (a[1][n][1] + 1e-12)]
    print('step %2d (%s) loss eager %.6f graph %.6f | differing parameters %d / %d | steps equal %s | leftover |grad| eager %s graph %s'
          % (i, 'lr' if plan[i] == 'L' else 'inc' if plan[i] else 'no_inc', a[0], b[0], len(bad), len(names), a[2] == b[2], a[3], b[3]))
    print('      kinds of differing parameters:', sorted(set(n.split('.')[-2] + '.' + n.split('.')[-1] for n in bad)))
    for n in bad[:4]:
        print('      %-70s eager %.9g graph %.9g' % (n, a[1][n][0], b[1][n][0]))
