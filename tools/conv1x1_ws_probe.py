"""Weight-stationary 1x1 convolution kernel (csrc/conv1x1.hip) against the tile GEMM path on the trunk's short-K shapes: correctness
(output + statistics partials) and cold-cache timing.  usage: python tools/conv1x1_ws_probe.py"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H

BF = H.STORE_DTYPE
SHAPES = [(64, 24, 256, 1024), (64, 96, 64, 256), (64, 48, 128, 512), (64, 12, 512, 2048), (5, 14, 256, 1024),
          (64, 24, 1024, 256), (64, 48, 512, 128)]   # images, H, Ci, Co


def cold(fn, iters=8):
    junk = torch.empty(768 << 20, dtype=torch.uint8, device='cuda')
    fn()
    tot = 0.0
    for i in range(iters):
        junk.fill_(i & 255)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / iters


for (N, Hh, Ci, Co) in SHAPES:
    torch.manual_seed(1)
    g = H.conv_geom(N, Hh, Hh, Ci, Co, 1, 1, 1, 0)
    M = N * Hh * Hh
    x = (torch.randn(M, Ci, device='cuda') * 0.7).to(BF)
    w = (torch.randn(Co, Ci, device='cuda') * 0.05).to(BF)
    y0, y1 = torch.empty(M, Co, device='cuda', dtype=BF), torch.empty(M, Co, device='cuda', dtype=BF)
    nb0 = H.lib.evk_conv_stats_bytes(M, Co)
    part0 = torch.empty(nb0 // 4, device='cuda')
    n0 = C.c_int32(0)
    f0 = lambda: H.check(H.lib.evk_gemm_conv1x1_ref(H.ptr(x), H.ptr(w), H.ptr(y0), C.byref(g), H.ptr(part0), nb0, C.byref(n0), H.stream())) \
        if hasattr(H.lib, 'evk_gemm_conv1x1_ref') else H.check(H.lib.evk_conv2d_fwd_stats_tile(H.ptr(x), H.ptr(w), H.ptr(y0), C.byref(g), H.ptr(part0), nb0, C.byref(n0), H.stream()))
    nb1 = H.lib.evk_conv1x1_ws_part_bytes(M, Ci, Co)
    part1 = torch.empty(nb1 // 4, device='cuda')
    n1 = C.c_int32(0)
    f1 = lambda: H.check(H.lib.evk_conv1x1_ws_fwd(H.ptr(x), H.ptr(w), H.ptr(y1), M, Ci, Co, H.ptr(part1), nb1, C.byref(n1), H.stream()))
    f0(); f1()
    torch.cuda.synchronize()
    ref = x.float() @ w.float().t()
    e0 = float((y0.float() - ref).abs().max()); e1 = float((y1.float() - ref).abs().max())
    s0 = part0[:n0.value * 2 * Co].view(n0.value, 2, Co).sum(0)
    s1 = part1[:n1.value * 2 * Co].view(n1.value, 2, Co).sum(0)
    es = float((s1[0] - ref.sum(0)).abs().max() / (ref.abs().sum(0).max())), float((s1[1] - (ref ** 2).sum(0)).abs().max() / (ref ** 2).sum(0).max())
    t0, t1 = cold(f0), cold(f1)
    byts = (M * Ci + M * Co) * 2
    print('fwd  M=%7d K=%4d N=%5d  tile %.3f ms (%.2f TB/s)  ws %.3f ms (%.2f TB/s)  x%.2f  | max err tile %.3g ws %.3g  stats rel err %.2e / %.2e'
          % (M, Ci, Co, t0, byts / t0 / 1e9, t1, byts / t1 / 1e9, t0 / t1, e0, e1, es[0], es[1]), flush=True)
    # data gradient with skip + gate + gate statistics:  dx[M][Co'] = gate(dy[M][Ci'] . Wt^T + skip) with K = Ci, N = Co here
    dy = x
    wt = w.t().contiguous()                         # the data-gradient entry takes the weights as stored: [K][N]
    skip = (torch.randn(M, Co, device='cuda') * 0.3).to(BF)
    gate = torch.relu(torch.randn(M, Co, device='cuda')).to(BF)
    d1 = torch.empty(M, Co, device='cuda', dtype=BF)
    fd = lambda: H.check(H.lib.evk_conv1x1_ws_dgrad(H.ptr(dy), H.ptr(wt), H.ptr(skip), H.ptr(gate), H.ptr(d1), M, Ci, Co, H.ptr(part1), nb1, C.byref(n1), H.stream()))
    fd()
    torch.cuda.synchronize()
    gref = (ref + skip.float()) * (gate.float() > 0)
    ed = float((d1.float() - gref).abs().max())
    sd = part1[:n1.value * 2 * Co].view(n1.value, 2, Co).sum(0)
    esd = float((sd[0] - gref.sum(0)).abs().max() / gref.abs().sum(0).max()), float((sd[1] - (gref * gate.float()).sum(0)).abs().max() / (gref * gate.float()).abs().sum(0).max())
    td = cold(fd)
    byd = (M * Ci + 3 * M * Co) * 2
    print('dgrad                          ws %.3f ms (%.2f TB/s)                      | max err %.3g  gate stats rel err %.2e / %.2e' % (td, byd / td / 1e9, ed, esd[0], esd[1]), flush=True)
