"""Loss trajectory of repeated FineTune training steps on one fixed synthetic batch (memorisation): a sanity check that the
step harness optimises, and a side-by-side of the two storage builds (run once per build: EVK_STORE=bf16 / f16).
usage: [EVK_STORE=f16] python tools/train_sanity.py [steps]"""
import os
import sys
import torch
sys.path.insert(0, '.')
import bench as BN
from evoke_amd import hip as H, ops, optim
from evoke_amd.model_pretrain_finetune import FineTune
from tests.helpers import load_tokenizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
torch.manual_seed(9233)
ops.manual_seed(9233)
args = BN.make_args('finetune')
args.update(pt_lr=5e-5, ft_lr=5e-4)
dev = torch.device('cuda', 0)
model = FineTune(args, load_tokenizer(), 'mimic_cxr').to(dev)
model.train()
opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
batch = BN.synth_batch('finetune', 16, 2, 224, 60, 30, dev, 5)
out = []
for it in range(steps):
    opt.zero_grad()
    ret = model(batch['images'], batch['ids'], batch['masks'], batch['pids'], batch['inc'], batch['inc_masks'], mode='train')
    ret['all_loss'].backward()
    opt.step()
    if it % 10 == 0 or it == steps - 1:
        out.append((it, float(ret['all_loss'])))
print('store=%s loss_scale=%g  ' % (H.STORE, ops.loss_scale_value()) + '  '.join('%d:%.4f' % t for t in out))
assert all(x == x for _, x in out) and out[-1][1] < out[0][1], 'loss did not decrease'
