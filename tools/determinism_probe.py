"""Where do two runs of the same training step first differ?  Full-size FineTune step (384^2, 32 studies x 2 views) run several times from
the same state with forward hooks on every module: prints, in call order, the first modules whose output bits are not the same in all runs,
then the parameters whose gradients differ.  usage: python tools/determinism_probe.py [runs] [B] [res]"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import os
from evoke_amd import ops
if os.environ.get('PROBE_NO_SIDE') == '1':
    ops.SIDE_STREAMS_ENABLED[0] = False
from evoke_amd.model_pretrain_finetune import FineTune
from oracle import spec as S
from tests.helpers import ARGS, V, load_procedural, load_tokenizer

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
res = int(sys.argv[3]) if len(sys.argv) > 3 else 384
model = FineTune(dict(ARGS), load_tokenizer(), 'iu_xray')
load_procedural(model, S.finetune_spec(V))
g = torch.Generator().manual_seed(33)
L, Li = 100, 30
images = torch.randn(2 * B, 3, res, res, generator=g).cuda()
ids = torch.randint(5, V - 2, (B, L), generator=g)
ids[:, 0] = V - 2
masks = torch.ones(B, L, dtype=torch.long)
for i in range(B):
    ln = L - (3 * i) % 41
    ids[i, ln - 1] = V - 1
    ids[i, ln:] = 0
    masks[i, ln:] = 0
inc = torch.randint(5, V - 2, (B, Li), generator=g)
inc[:, 0] = 1
incm = torch.ones(B, Li, dtype=torch.long)
pids = np.array(['p%d_s%d' % (i % B, i % B) for i in range(2 * B)])
ids, masks = ids.cuda(), masks.cuda()


def bits(t):
    t = t.detach().contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        v = t.view(torch.int16).to(torch.int64)
    elif t.dtype == torch.float32:
        v = t.view(torch.int32).to(torch.int64)
    else:
        v = t.to(torch.int64)
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003 + 1
    return int((v.reshape(-1) * w).sum().item())


order, rec = [], {}


def hook(name):
    def fn(m, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for j, o in enumerate(outs):
            if torch.is_tensor(o):
                key = '%s[%d]' % (name, j)
                if key not in rec:
                    order.append(key)
                rec.setdefault(key, []).append(bits(o))
    return fn


border, brec = [], {}


def bhook(name):
    def fn(m, gin, gout):
        for j, o in enumerate(gout):
            if torch.is_tensor(o):
                key = '%s.grad_out[%d]' % (name, j)
                if key not in brec:
                    border.append(key)
                brec.setdefault(key, []).append(bits(o))
    return fn


for n, m in model.named_modules():
    if n:
        m.register_forward_hook(hook(n))
        if len(list(m.children())) == 0 or n.count('.') <= 1:
            m.register_full_backward_hook(bhook(n))
model.train()
ops.set_dropout_enabled(False)
state = {k: v.clone() for k, v in model.state_dict().items()}
params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
losses, grads = [], []
for r in range(runs):
    model.load_state_dict(state)
    for _, p in params:
        p.grad = None
    loss = model(images, ids, masks, pids, inc, incm, mode='train')['all_loss']
    loss.backward()                 # the model's forward already attached the loss scale to the graph (ops.scale_loss)
    ops.join_side_streams()
    torch.cuda.synchronize()
    losses.append(loss.item())
    grads.append([None if p.grad is None else bits(p.grad) for _, p in params])
print('losses', losses)
bad = [k for k in order if len(set(rec[k])) > 1]
print('%d of %d module outputs differ between runs; first in call order:' % (len(bad), len(order)))
for k in bad[:25]:
    print('   ', k, rec[k])
bbad = [k for k in border if len(set(brec[k])) > 1]
print('%d of %d module output-gradients differ; first in backward order:' % (len(bbad), len(border)))
for k in bbad[:12]:
    print('   ', k, brec[k])
print('   identical before the first difference:', border[:border.index(bbad[0])][-6:] if bbad else border[-6:])
gb = [params[i][0] for i in range(len(params)) if len(set(gr[i] for gr in grads)) > 1]
print('%d of %d parameter gradients differ; first:' % (len(gb), len(params)), gb[:20])
import collections
same = collections.Counter(params[i][0].split('.')[0] + '.' + params[i][0].split('.')[1] for i in range(len(params)) if len(set(gr[i] for gr in grads)) == 1 and grads[0][i] is not None)
print('identical gradients by module:', dict(same))
idn = [params[i][0] for i in range(len(params)) if len(set(gr[i] for gr in grads)) == 1 and grads[0][i] is not None]
print('identical trunk gradients:', [n for n in idn if n.startswith('visual_extractor')])
diff = collections.Counter('.'.join(n.split('.')[:3]) for n in gb if not n.startswith('visual_extractor'))
print('differing outside the trunk:', dict(diff))
print('none-grad params:', sum(1 for i in range(len(params)) if grads[0][i] is None))
