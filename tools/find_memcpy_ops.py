"""torch.profiler over one eager training step: every GPU memcpy activity with the CPU op and Python line that issued it.
usage: python tools/find_memcpy_ops.py"""
import sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, '.')
import bench
from evoke_amd import distributed as D, ops, optim
from evoke_amd.model_pretrain_finetune import FineTune
from tests.helpers import load_tokenizer

dev = torch.device('cuda', 0)
args = bench.make_args('finetune')
model = FineTune(args, load_tokenizer(), 'mimic_cxr').to(dev)
model.train()
opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
red = D.GradReducer.for_optimizer(opt)
batch = bench.synth_batch('finetune', 32, 2, 384, 100, 30, dev, 1000)


def step():
    ops.advance_seed_epoch()
    opt.zero_grad()
    red.begin('finetune')
    loss = model(batch['images'], batch['ids'], batch['masks'], batch['pids'], batch['inc'], batch['inc_masks'], mode='train')['all_loss']
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
by_corr = {}
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU:
        continue
for e in evs:
    nm = e.name
    if 'emcpy' in nm or 'emset' in nm:
        print('GPU/RT activity:', nm, 'dur %.1f us' % e.cuda_time_total if hasattr(e, 'cuda_time_total') else '', '| parent:', getattr(e.cpu_parent, 'name', None),
              '| stack:', [s for s in (e.stack or (e.cpu_parent.stack if e.cpu_parent else []) or []) if 'evoke_amd' in s or 'bench' in s][:3])
