"""Fused optimizer launch (evk_optim_step_dyn) on a flat buffer of the FineTune model's size: time and achieved HBM rate for an
aligned run and for one that starts 1 element off a 16-byte boundary.  usage: python tools/optim_probe.py"""
import sys
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H

n = 76_000_000
dev = 'cuda'
p, g, m, v = (torch.randn(n + 8, device=dev) * 0.01 for _ in range(4))
v.abs_()
sh = torch.empty(n + 8, device=dev, dtype=H.STORE_DTYPE)
step = torch.zeros(1, dtype=torch.int32, device=dev)
state = torch.tensor([1024.0, 0, 0, 0], device=dev)
for off in (0, 1):
    def fn():
        H.check(H.lib.evk_optim_step_dyn(p.data_ptr() + 4 * off, g.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, None,
                                         sh.data_ptr() + 2 * off, n, 0, 1e-4, 0.9, 0.999, 1e-8, 0.0, 0.1, step.data_ptr(), state.data_ptr(), 1.0, 1,
                                         H.stream()))
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print('offset %d: %.3f ms for %d elements, %.2f TB/s (16 B read + 18 B written per element)' % (off, ms, n, n * 34 / ms / 1e9))
