"""Isolated, cold-cache timing of the stem convolution (forward + batch-norm partials, weight gradient incl. the slab reduction) on
64 images of 384^2: the halo kernels of csrc/stem.hip by default, the implicit-GEMM tile path with EVK_STEM_HALO=0."""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H

N, Hh, W = 64, 384, 384
img = torch.randn(N, 3, Hh, W, device='cuda')
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.1
xpad = torch.empty(N, Hh + 6, W + 8, 4, dtype=H.STORE_DTYPE, device='cuda')
wp = torch.empty(64, 7, 8, 4, dtype=H.STORE_DTYPE, device='cuda')
H.check(H.lib.evk_stem_pack_image(H.ptr(img), H.ptr(xpad), N, Hh, W, H.stream()))
H.check(H.lib.evk_stem_pack_weight(H.ptr(w), H.ptr(wp), H.stream()))
y = torch.empty(N, Hh // 2, W // 2, 64, dtype=H.STORE_DTYPE, device='cuda')
dy = torch.randn(N, Hh // 2, W // 2, 64, device='cuda').to(H.STORE_DTYPE)
nb = max(H.lib.evk_conv_stats_bytes(N * (Hh // 2) * (W // 2), 64), H.lib.evk_stem_halo_part_bytes(N, Hh, W))
part = torch.empty(nb // 4, device='cuda')
nblk = C.c_int32(0)
wsb = H.lib.evk_stem_wgrad_ws_bytes(N, Hh, W)
ws = torch.empty(wsb // 4, device='cuda')
dwp = torch.zeros(64, 7, 8, 4, device='cuda')
junk = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')


def run(fn, iters=10):
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


fl = 2.0 * N * (Hh // 2) * (W // 2) * 64 * 147
t = run(lambda: H.check(H.lib.evk_stem_fwd_stats(H.ptr(xpad), H.ptr(wp), H.ptr(y), N, Hh, W, H.ptr(part), nb, C.byref(nblk), H.stream())))
print('stem fwd+stats 64x384x384   %.3f ms  %6.1f TF/s (7x7x3 taps)' % (t, fl / t / 1e9))
t = run(lambda: H.check(H.lib.evk_stem_wgrad(H.ptr(dy), H.ptr(xpad), H.ptr(dwp), N, Hh, W, H.ptr(ws), wsb, H.stream())))
print('stem wgrad     64x384x384   %.3f ms  %6.1f TF/s (incl. the slab reduction)' % (t, fl / t / 1e9))
