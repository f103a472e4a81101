"""Debug helper: per-block relative error of the HIP trunk vs the CPU oracle (train-mode BN)."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import functional as O, spec as S
from tests.golden.cases import CASES, make_inputs
from tests.helpers import V, rel_err
from evoke_amd.trunk import ResNet, _Stem, _MaxPool

train = len(sys.argv) < 2 or sys.argv[1] == 'train'
inp = make_inputs(CASES['ft224_inc'], V)
spec = {}
S.resnet_spec(spec)
P = S.procedural_state(spec)
ctx = O.Ctx(train=train)
images = inp['images']
# oracle, capturing block outputs
ref = []
x = F.conv2d(images, P['visual_extractor.model.0.weight'], None, 2, 3); ref.append(('stem conv', x))
x = F.relu(O._bn(P, 'visual_extractor.model.1', x, ctx)); ref.append(('stem bn', x))
x = F.max_pool2d(x, 3, 2, 1); ref.append(('maxpool', x))
for li, (planes, blocks, stride) in enumerate(O.RESNET_LAYERS):
    for b in range(blocks):
        p = 'visual_extractor.model.%d.%d.' % (4 + li, b)
        s = stride if b == 0 else 1
        y = F.relu(O._bn(P, p + 'bn1', F.conv2d(x, P[p + 'conv1.weight']), ctx))
        y = F.relu(O._bn(P, p + 'bn2', F.conv2d(y, P[p + 'conv2.weight'], None, s, 1), ctx))
        y = O._bn(P, p + 'bn3', F.conv2d(y, P[p + 'conv3.weight']), ctx)
        if b == 0:
            x = O._bn(P, p + 'downsample.1', F.conv2d(x, P[p + 'downsample.0.weight'], None, s), ctx)
        x = F.relu(y + x)
        ref.append(('layer%d.%d' % (li + 1, b), x))
m = ResNet({})
sd = {k[len('visual_extractor.'):]: v for k, v in S.procedural_state(spec).items()}
m.load_state_dict(sd)
m = m.cuda().train(train)
t = m.model
got = []
with torch.no_grad():
    x = _Stem.apply(images.cuda(), t[0].weight); got.append(x)
    x = t[1](x, relu=True); got.append(x)
    x = _MaxPool.apply(x); got.append(x)
    for li in range(4, 8):
        for blk in t[li]:
            x = blk(x); got.append(x)
for (name, r), g in zip(ref, got):
    r = r.permute(0, 2, 3, 1)
    print('%-12s rel err %.4e   rms %.3e  frac zero ref %.3f' % (name, rel_err(g.float(), r), r.pow(2).mean().sqrt().item(), (r == 0).float().mean().item()))
