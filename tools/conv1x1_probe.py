"""Cold-cache timing of the trunk's 1x1 convolutions (forward with batch-norm statistics, gated data gradient) against
their algorithmic HBM bytes: the kernels DESIGN.md section 3 names as the step's bound.
usage: python tools/conv1x1_probe.py [--iters N] [filter]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H

BF = torch.bfloat16
# (images, H, Ci, Co) of the 1x1 convolutions of ResNet-101 at 384^2, 64 images
SHAPES = [(64, 96, 64, 256), (64, 96, 256, 64), (64, 48, 128, 512), (64, 48, 512, 128),
          (64, 24, 256, 1024), (64, 24, 1024, 256), (64, 12, 512, 2048), (64, 12, 2048, 512)]


def cases():
    for n, h, ci, co in SHAPES:
        g = H.conv_geom(n, h, h, ci, co, 1, 1, 1, 0)
        M = n * h * h
        x = torch.randn(M, ci, device='cuda').to(BF)
        w = (torch.randn(co, ci, device='cuda') / ci ** 0.5).to(BF)
        y = torch.empty(M, co, device='cuda', dtype=BF)
        nb = H.lib.evk_conv_stats_bytes(M, co)
        part = torch.empty(nb // 4, device='cuda')
        nblk = C.c_int(0)
        gate = torch.randn(M, ci, device='cuda').to(BF)
        resid = torch.randn(M, ci, device='cuda').to(BF)
        dx = torch.empty(M, ci, device='cuda', dtype=BF)
        st = H.stream

        def fwd(x=x, w=w, y=y, g=g, part=part, nb=nb, nblk=nblk):
            H.check(H.lib.evk_conv2d_fwd_stats(H.ptr(x), H.ptr(w), H.ptr(y), C.byref(g), H.ptr(part), nb, C.byref(nblk), st()))

        def dgrad(y=y, w=w, resid=resid, gate=gate, dx=dx, g=g):
            H.check(H.lib.evk_conv2d_dgrad_gated(H.ptr(y), H.ptr(w), H.ptr(resid), H.ptr(gate), H.ptr(dx), C.byref(g), st()))

        def dgrad_plain(y=y, w=w, gate=gate, dx=dx, g=g):
            H.check(H.lib.evk_conv2d_dgrad_gated(H.ptr(y), H.ptr(w), None, H.ptr(gate), H.ptr(dx), C.byref(g), st()))

        yield 'fwd+stats   %6d x %4d -> %4d' % (M, ci, co), fwd, 2 * M * (ci + co), 2.0 * M * ci * co
        yield 'dgrad gate  %6d x %4d <- %4d' % (M, ci, co), dgrad_plain, 2 * M * (co + 2 * ci), 2.0 * M * ci * co
        yield 'dgrad g+res %6d x %4d <- %4d' % (M, ci, co), dgrad, 2 * M * (co + 3 * ci), 2.0 * M * ci * co


def time_cold(fn, iters):
    junk = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')
    fn()
    tot = 0.0
    for i in range(iters):
        junk.fill_(i & 255)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / iters


if __name__ == '__main__':
    iters = 10
    if '--iters' in sys.argv:
        i = sys.argv.index('--iters')
        iters = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    sel = sys.argv[1] if len(sys.argv) > 1 else ''
    for name, fn, nbytes, flops in cases():
        if sel and sel not in name:
            continue
        ms = time_cold(fn, iters)
        print('%-36s %8.1f us  %7.0f GB/s  %7.1f TF/s' % (name, ms * 1e3, nbytes / ms / 1e6, flops / ms / 1e9), flush=True)
