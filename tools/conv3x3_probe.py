"""In-kernel timing of csrc/conv3x3.hip (evk_conv3x3_halo_debug_stamps): per workgroup the shader-clock cycles of the prologue,
the K loop and the epilogue, and the clock the chip held (s_memtime / s_memrealtime x 100 MHz)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H

BF = H.STORE_DTYPE


def run(N, Hh, W, Ci, Co, iters=30):
    x = torch.randn(N, Hh, W, Ci, device='cuda').to(BF)
    w = (torch.randn(Co, 3, 3, Ci, device='cuda') * 0.02).to(BF)
    y = torch.empty(N, Hh, W, Co, device='cuda', dtype=BF)
    nb = H.lib.evk_conv3x3_halo_part_bytes(N, Hh, W, Co)
    part = torch.empty(nb // 4, device='cuda')
    nblk = C.c_int32(0)
    call = lambda: H.check(H.lib.evk_conv3x3_halo(H.ptr(x), H.ptr(w), H.ptr(y), N, Hh, W, Ci, Co, None, 0, None, 0, H.ptr(part), None, nb,
                                                  C.byref(nblk), H.stream()))
    for _ in range(iters):          # warm the clocks
        call()
    torch.cuda.synchronize()
    nwg = nblk.value // 4 * (Co // 128)
    st = torch.zeros(nwg * 8, dtype=torch.int64, device='cuda')
    H.check(H.lib.evk_conv3x3_halo_debug_stamps(H.ptr(st)))
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    H.check(H.lib.evk_conv3x3_halo_debug_stamps(None))
    s = st.view(nwg, 8).cpu().double()
    pro, loop, epi, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
    real = (s[:, 5] - s[:, 4]) / 100.0           # us
    steps = Ci // 64 * 9
    print('%dx%dx%d %d->%d: %d workgroups, %d steps' % (N, Hh, W, Ci, Co, nwg, steps))
    print('  cycles  prologue %.0f  loop %.0f (%.0f per step; 1280 = MFMA-bound)  epilogue %.0f  total %.0f' % (
        pro.median(), loop.median(), loop.median() / steps, epi.median(), tot.median()))
    print('  workgroup lifetime %.1f us median, %.1f max; clock %.2f GHz; launch span %.1f us' % (
        real.median(), real.max(), (tot / real).median() / 1000.0, (s[:, 5].max() - s[:, 4].min()) / 100.0))


def run_wgrad(N, Hh, W, Ci, Co, iters=30):
    x = torch.randn(N, Hh, W, Ci, device='cuda').to(BF)
    dy = torch.randn(N, Hh, W, Co, device='cuda').to(BF)
    dw = torch.zeros(Co, 3, 3, Ci, device='cuda')
    nb = H.lib.evk_conv3x3_wgrad_halo_ws_bytes(N, Hh, W, Ci, Co)
    ws = torch.empty(nb // 4, device='cuda')
    call = lambda: H.check(H.lib.evk_conv3x3_wgrad_halo(H.ptr(dy), H.ptr(x), H.ptr(dw), N, Hh, W, Ci, Co, H.ptr(ws), nb, H.stream()))
    for _ in range(iters):
        call()
    torch.cuda.synchronize()
    nsplit = nb // (9 * Co * Ci * 4)
    nwg = nsplit * (Ci // 64) * (Co // 64)
    st = torch.zeros(nwg * 8, dtype=torch.int64, device='cuda')
    H.check(H.lib.evk_conv3x3_halo_debug_stamps(H.ptr(st)))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record()
    torch.cuda.synchronize()
    H.check(H.lib.evk_conv3x3_halo_debug_stamps(None))
    s = st.view(nwg, 8).cpu().double()
    pro, loop, epi, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
    real = (s[:, 5] - s[:, 4]) / 100.0
    px = N * Hh * W
    mf = 2.0 * px * 9 * 64 * 64 / nsplit / 16384 / 8 * 16 * 2          # MFMA cycles per SIMD per workgroup at 16 cycles per MFMA, 2 waves per SIMD
    print('wgrad %dx%dx%d %d->%d: %d workgroups (%d K-slices), %.1f us per call incl. the split-K reduction' % (N, Hh, W, Ci, Co, nwg, nsplit, a.elapsed_time(b) * 1e3 / iters))
    print('  cycles  prologue %.0f  loop %.0f (MFMA-bound %.0f)  epilogue %.0f  total %.0f' % (pro.median(), loop.median(), mf, epi.median(), tot.median()))
    print('  workgroup lifetime %.1f us median, %.1f max; clock %.2f GHz; launch span %.1f us' % (
        real.median(), real.max(), (tot / real).median() / 1000.0, (s[:, 5].max() - s[:, 4].min()) / 100.0))


if __name__ == '__main__':
    run_wgrad(64, 24, 24, 256, 256)
    run_wgrad(64, 48, 48, 128, 128)
    run_wgrad(64, 12, 12, 512, 512)
    run_wgrad(64, 96, 96, 64, 64)
    run(64, 24, 24, 256, 256)
    run(128, 24, 24, 256, 256)
    run(64, 48, 48, 128, 128)
    run(64, 12, 12, 512, 512)
