"""Host issue time against GPU time per phase of one FineTune training step: is any phase of the step waiting for the host?
Steps issued from an idle GPU (full synchronisation before each) give the pure host cost of enqueuing the forward, the backward and
the optimizer; free-running steps give the GPU durations between phase events.  usage: python tools/host_phases.py"""
import sys
import time
import torch
sys.path.insert(0, '.')
import bench
from evoke_amd import distributed as D, ops, optim
from evoke_amd.model_pretrain_finetune import FineTune
from evoke_amd.config import load_default_tokenizer

dev = torch.device('cuda', 0)
torch.manual_seed(9233)
args = bench.make_args('finetune')
model = FineTune(args, load_default_tokenizer(), 'mimic_cxr').to(dev).train()
opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
red = D.GradReducer.for_optimizer(opt)
b = bench.synth_batch('finetune', 32, 2, 384, 100, 30, dev, 1000)


def step(host):
    t = [time.perf_counter()]
    opt.zero_grad()
    red.begin('finetune')
    ret = model(b['images'], b['ids'], b['masks'], b['pids'], b['inc'], b['inc_masks'], mode='train')
    t.append(time.perf_counter())
    ret['all_loss'].backward()
    t.append(time.perf_counter())
    red.finish()
    opt.step()
    t.append(time.perf_counter())
    if host is not None:
        host.append([1e3 * (y - x) for x, y in zip(t[:-1], t[1:])])


for _ in range(3):
    step(None)
host = []
for _ in range(5):
    torch.cuda.synchronize()
    step(host)
torch.cuda.synchronize()
avg = [sum(h[i] for h in host) / len(host) for i in range(3)]
print('host issue time from an idle GPU: forward %.1f ms, backward %.1f ms, reducer + optimizer %.1f ms (total %.1f)' % (*avg, sum(avg)))
t0 = time.perf_counter()
free = []
for _ in range(6):
    step(free)
torch.cuda.synchronize()
print('free-running: %.1f ms per step; host loop: forward %.1f, backward %.1f, optimizer %.1f ms (includes waiting for launch-queue slots)' % (
    1e3 * (time.perf_counter() - t0) / 6, *[sum(h[i] for h in free) / len(free) for i in range(3)]))
