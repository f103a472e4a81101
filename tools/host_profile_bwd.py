"""Host time of the BACKWARD of one training step by Python function (autograd's device thread switched off, so that cProfile -- which
only sees the thread it was enabled in -- covers the backward nodes): steps issued from an idle GPU.  usage: python tools/host_profile_bwd.py [steps]"""
import cProfile
import os
import pstats
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from evoke_amd import distributed as D, ops, optim                 # noqa: E402
from evoke_amd.model_pretrain_finetune import FineTune             # noqa: E402
from evoke_amd.config import load_default_tokenizer                # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda', 0)
torch.manual_seed(9233)
args = bench.make_args('finetune')
model = FineTune(args, load_default_tokenizer(), 'mimic_cxr').to(dev)
model.train()
opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
red = D.GradReducer.for_optimizer(opt)
batch = bench.synth_batch('finetune', 32, 2, 384, 100, 30, dev, 1000)
torch.autograd.set_multithreading_enabled(False)
pr_f, pr_b = cProfile.Profile(), cProfile.Profile()
tf = tb = 0.0
for it in range(3 + steps):
    torch.cuda.synchronize()
    ops.advance_seed_epoch()
    opt.zero_grad()
    red.begin('finetune')
    t0 = time.perf_counter()
    if it >= 3:
        pr_f.enable()
    loss = model(batch['images'], batch['ids'], batch['masks'], batch['pids'], batch['inc'], batch['inc_masks'], mode='train')['all_loss']
    pr_f.disable()
    t1 = time.perf_counter()
    if it >= 3:
        pr_b.enable()
    loss.backward()
    pr_b.disable()
    t2 = time.perf_counter()
    if it >= 3:
        tf += t1 - t0
        tb += t2 - t1
    red.finish()
    opt.step()
torch.cuda.synchronize()
print('host issue time per step under cProfile: forward %.1f ms, backward %.1f ms' % (1e3 * tf / steps, 1e3 * tb / steps))
for name, pr in (('FORWARD', pr_f), ('BACKWARD', pr_b)):
    print('=' * 30, name, '(per %d steps)' % steps)
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(22)
    st.sort_stats('cumulative').print_stats(30)
