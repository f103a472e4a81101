"""Eval-mode forward of the visual extractor under no_grad, folded batch norms (evk_trunk_forward_inference) against the unfused eval forward:
usage: python tools/trunk_infer_bench.py [images=128] [res=384] [fold=both|0|1]"""
import sys
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from evoke_amd import trunk as T
from evoke_amd.trunk import ResNet

n, res = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), (int(sys.argv[2]) if len(sys.argv) > 2 else 384)
which = sys.argv[3] if len(sys.argv) > 3 else 'both'
torch.manual_seed(5)
m = ResNet({}).cuda().train()
img = torch.randn(n, 3, res, res, device='cuda')
with torch.no_grad():
    for _ in range(30):                           # running statistics of a network that has seen data (fresh ones overflow fp16 in eval mode)
        m(img[:16])
m.eval()
for fold in {'both': (-1, 0, 1, 2, -1, 0, 1, 2), '-1': (-1,), '0': (0,), '1': (1,), '2': (2,)}[which]:
    T.FOLD_BN[0] = fold
    with torch.no_grad():
        for _ in range(3):
            m(img)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m(img)
        e1.record()
        torch.cuda.synchronize()
    print('fold=%s  %.2f ms per forward of %d images at %d^2' % (fold, e0.elapsed_time(e1) / 10, n, res))
