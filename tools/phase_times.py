"""GPU time of the phases of one FineTune training step on the main stream (events at phase boundaries; side streams on)."""
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench  # noqa: E402
from evoke_amd import distributed as D, ops, optim  # noqa: E402
from evoke_amd.model_pretrain_finetune import FineTune  # noqa: E402
from evoke_amd.config import load_default_tokenizer as load_tokenizer  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(9233)
args = bench.make_args('finetune')
model = FineTune(args, load_tokenizer(), 'mimic_cxr').to(dev).train()
opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
red = D.GradReducer.for_optimizer(opt)
b = bench.synth_batch('finetune', 32, 2, 384, 100, 30, dev, 1000)
ev = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev.setdefault(name, []).append(e)


model.visual_extractor.register_forward_hook(lambda m, i, o: mark('trunk_fwd_end'))


def _trunk_out_hook(m, i, o):
    if o.requires_grad:
        o.register_hook(lambda g: mark('trunk_bwd_start'))


model.visual_extractor.model.register_forward_hook(_trunk_out_hook)


def step():
    mark('start')
    opt.zero_grad()
    red.begin('finetune')
    ret = model(b['images'], b['ids'], b['masks'], b['pids'], b['inc'], b['inc_masks'], mode='train')
    mark('fwd_end')
    ret['all_loss'].backward()
    mark('bwd_end')
    for (name, d), st in list(ops._side_streams.items()):            # when each side stream runs dry (everything the backward queued on it)
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        ev.setdefault('side:' + name, []).append(e)
    red.finish()
    mark('joined')
    opt.step()
    mark('opt_end')


for _ in range(3):
    step()
torch.cuda.synchronize()
ev.clear()
t0 = time.perf_counter()
N = 6
for _ in range(N):
    step()
torch.cuda.synchronize()
print('wall %.2f ms/step' % (1e3 * (time.perf_counter() - t0) / N))
order = ['start', 'trunk_fwd_end', 'fwd_end', 'trunk_bwd_start', 'bwd_end', 'joined', 'opt_end']
for a, c in zip(order[:-1], order[1:]):
    ms = sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[c])) / N
    print('%-16s -> %-16s %7.2f ms' % (a, c, ms))
print('opt_end -> next start %.2f ms' % (sum(x.elapsed_time(y) for x, y in zip(ev['opt_end'][:-1], ev['start'][1:])) / (N - 1)))
for k in sorted(ev):
    if k.startswith('side:'):
        print('start -> %-12s dry %7.2f ms   (main stream: bwd_end at %.2f, joined at %.2f)' % (
            k[5:], sum(x.elapsed_time(y) for x, y in zip(ev['start'], ev[k])) / N, sum(x.elapsed_time(y) for x, y in zip(ev['start'], ev['bwd_end'])) / N,
            sum(x.elapsed_time(y) for x, y in zip(ev['start'], ev['joined'])) / N))
