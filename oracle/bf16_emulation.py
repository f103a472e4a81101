"""CPU oracle (test infrastructure): the ResNet-101 trunk of oracle.functional with the ENGINE'S bf16 storage points
emulated (weights, conv outputs and BN outputs rounded to bf16, statistics / accumulation in f32).

Why: with train-mode BatchNorm the trunk amplifies any perturbation of its activations roughly linearly in depth on
random weights; the fp32 reference and a bf16-storage engine therefore legitimately differ by tens of percent at
layer4 on the synthetic golden cases (measured: 0.85 relative), while eval mode stays at ~1.5e-2.  Comparing the HIP
engine with this emulation separates "wrong kernel" from "bf16 storage": the two follow the same trajectory."""
import torch
import torch.nn.functional as F

from . import functional as O


def _r(t):
    return t.to(torch.bfloat16).float()


def resnet101_trunk_bf16(P, images, ctx, prefix='visual_extractor.model.', taps=None):
    w = lambda k: _r(P[k])
    x = _r(F.conv2d(_r(images), w(prefix + '0.weight'), None, 2, 3))
    x = _r(F.relu(O._bn(P, prefix + '1', x, ctx)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(O.RESNET_LAYERS):
        for b in range(blocks):
            p = '%s%d.%d.' % (prefix, 4 + li, b)
            s = stride if b == 0 else 1
            y = _r(F.relu(O._bn(P, p + 'bn1', _r(F.conv2d(x, w(p + 'conv1.weight'))), ctx)))
            y = _r(F.relu(O._bn(P, p + 'bn2', _r(F.conv2d(y, w(p + 'conv2.weight'), None, s, 1)), ctx)))
            y3 = _r(F.conv2d(y, w(p + 'conv3.weight')))
            if b == 0:
                x = _r(O._bn(P, p + 'downsample.1', _r(F.conv2d(x, w(p + 'downsample.0.weight'), None, s)), ctx))
            x = _r(F.relu(O._bn(P, p + 'bn3', y3, ctx) + x))
            if taps is not None:
                taps.append(('layer%d.%d' % (li + 1, b), x))
    return x
