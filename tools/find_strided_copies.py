"""Which Python lines issue device-to-device copies of NON-contiguous tensors in one training step?  (torch turns row-pitched ones into
hipMemcpy2DAsync, which a stream capture records as a memcpy node the step replayer cannot re-issue -- csrc/replay.hip.)
usage: python tools/find_strided_copies.py [finetune|pretrain] [B views res L Li]"""
import collections
import sys
import traceback
import torch
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
seen = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if '/evoke_amd/' in fr.filename or fr.filename.endswith('bench.py'):
            return '%s:%d %s' % (fr.filename.split('/')[-1], fr.lineno, fr.line)
    return '?'


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        if torch.is_tensor(self):
            o = a[0] if name == 'copy_' and a and torch.is_tensor(a[0]) else None
            if name in ('to', 'cuda') and not self.is_cuda and ('cuda' in str(a) + str(k) or name == 'cuda'):
                seen[('H2D ' + name, tuple(self.shape), (), site())] += 1
            if name == 'copy_' and o is not None and self.is_cuda and not o.is_cuda:
                seen[('H2D copy_', tuple(self.shape), (), site())] += 1
            if name in ('clone',) and self.is_cuda and self.is_contiguous():
                seen[('D2D clone', tuple(self.shape), (), site())] += 1
            if name == 'copy_' and o is not None and self.is_cuda and o.is_cuda and self.is_contiguous() and o.is_contiguous() and o.dtype == self.dtype and o.shape == self.shape:
                seen[('D2D copy_', tuple(self.shape), (), site())] += 1
        src = a[0] if name == 'copy_' and a and torch.is_tensor(a[0]) else self
        if torch.is_tensor(src) and src.is_cuda and not src.is_contiguous() and src.numel() > 0:
            seen[(name, tuple(src.shape), tuple(src.stride()), site())] += 1
        elif name == 'copy_' and self.is_cuda and not self.is_contiguous() and self.numel() > 0:
            seen[(name + ' (strided dst)', tuple(self.shape), tuple(self.stride()), site())] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)


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


step()
torch.cuda.synchronize()
for n in ('contiguous', 'clone', 'copy_', 'reshape', 'to', 'cuda', 'flatten', 'view_as', 'type_as', 'float', 'half'):
    wrap(n)
_si = torch.Tensor.__setitem__


def _set(self, k, v):
    if self.is_cuda:
        seen[('__setitem__', tuple(self.shape), tuple(self.stride()), site())] += 1
    return _si(self, k, v)


torch.Tensor.__setitem__ = _set
step()
torch.cuda.synchronize()
for (name, shape, stride, where), c in sorted(seen.items(), key=lambda kv: kv[0][3]):
    print('%3d x %-22s shape %-26s stride %-26s %s' % (c, name, shape, stride, where))
