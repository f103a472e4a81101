"""Where the HOST time of one eager training step goes: cProfile over steps issued from an idle GPU (each step followed by a full
synchronisation, so queue back-pressure does not show up as host time).  usage: python tools/host_profile.py [steps]"""
import cProfile
import pstats
import sys
import time
import torch
sys.path.insert(0, '.')
import bench
from evoke_amd import distributed as D, ops, optim
from evoke_amd.model_pretrain_finetune import FineTune
from tests.helpers import load_tokenizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda', 0)
torch.manual_seed(9233)
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


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
tot = 0.0
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    step()
    pr.disable()
    tot += time.perf_counter() - t0
torch.cuda.synchronize()
print('host issue time per step (under cProfile): %.1f ms' % (1e3 * tot / steps))
st = pstats.Stats(pr)
st.sort_stats('tottime')
st.print_stats(28)
