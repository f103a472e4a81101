"""torch.profiler over one eager training step: every GPU memcpy activity with the CPU op and Python line that issued it.
usage: python tools/find_memcpy_ops.py [finetune|pretrain] [B views res L Li]"""
import sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, '.')
import bench
from evoke_amd import distributed as D, ops, optim
from evoke_amd.model_pretrain_finetune import FineTune, Pretrain
from tests.helpers import load_tokenizer

TASK = sys.argv[1] if len(sys.argv) > 1 else 'finetune'
SHAPE = [int(v) for v in sys.argv[2:7]] if len(sys.argv) >= 7 else [32, 2, 384, 100, 30]
dev = torch.device('cuda', 0)
args = bench.make_args(TASK)
model = (FineTune if TASK == 'finetune' else Pretrain)(args, load_tokenizer(), 'mimic_cxr').to(dev)
model.train()
opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
red = D.GradReducer.for_optimizer(opt)
batch = bench.synth_batch(TASK, SHAPE[0], SHAPE[1], SHAPE[2], SHAPE[3], SHAPE[4], dev, 1000)


def step():
    ops.advance_seed_epoch()
    opt.zero_grad()
    red.begin(TASK)
    if TASK == 'finetune':
        loss = model(batch['images'], batch['ids'], batch['masks'], batch['pids'], batch['inc'], batch['inc_masks'], mode='train')['all_loss']
    else:
        loss = model(batch['images'], batch['ids'], batch['masks'], batch['pids'])['all_loss']
    loss.backward()
    red.finish()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()


def chain(e):
    out, cur = [], e.cpu_parent
    while cur is not None and len(out) < 7:
        out.append('%s%s' % (cur.name, list(cur.input_shapes)[:3] if getattr(cur, 'input_shapes', None) else ''))
        cur = cur.cpu_parent
    return ' <- '.join(out)


for e in evs:
    if e.name in ('hipMemcpyAsync', 'hipMemcpy2DAsync', 'hipMemsetAsync', 'hipMemcpyWithStream'):
        print('RT call %-18s issued by: %s' % (e.name, chain(e)))

# aggregated: which host ops issue the step's memcpy / memset runtime calls, and the torch-native GPU kernels (glue) by the op that launched them
import collections
agg = collections.Counter()
for e in evs:
    if e.name in ('hipMemcpyAsync', 'hipMemcpy2DAsync', 'hipMemsetAsync', 'hipMemcpyWithStream'):
        agg[(e.name, chain(e)[:160])] += 1
print('---- aggregated runtime copies / memsets per step')
for (n, c), k in agg.most_common(40):
    print('%4d  %-18s %s' % (k, n, c))
glue = collections.Counter()
for e in evs:
    if e.name == 'hipLaunchKernel' or e.name == 'hipExtModuleLaunchKernel':
        par = e.cpu_parent
        if par is not None and par.name.startswith('aten::'):
            glue[(par.name, chain(par)[:120])] += 1
print('---- aggregated aten:: kernel launches per step (torch glue)')
for (n, c), k in glue.most_common(40):
    print('%4d  %-24s %s' % (k, n, c))
