"""Isolated timing of the GEMM / implicit-GEMM kernel on representative shapes of the FineTune 384^2 bs32 step."""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H, ops

BF = H.STORE_DTYPE


def time_it(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def gemm_case(M, N, K, a_mode, b_mode):
    A = torch.randn((M, K) if a_mode == H.A_PLAIN else (K, M), device='cuda').to(BF)
    B = torch.randn((N, K) if b_mode == H.B_PLAIN else (K, N), device='cuda').to(BF)
    acc = a_mode == H.A_KSTR
    Cm = torch.zeros(M, N, device='cuda', dtype=torch.float32 if acc else BF)
    lda = K if a_mode == H.A_PLAIN else M
    ldb = K if b_mode == H.B_PLAIN else N
    return lambda: ops.gemm(A, B, Cm, M, N, K, a_mode=a_mode, b_mode=b_mode, lda=lda, ldb=ldb, ldc=N, accumulate=acc)


def conv_case(N, Hh, Ci, Co, k, stride, kind):
    g = H.conv_geom(N, Hh, Hh, Ci, Co, k, k, stride, k // 2)
    x = torch.randn(N, Hh, Hh, Ci, device='cuda').to(BF)
    w = torch.randn(Co, k, k, Ci, device='cuda').to(BF)
    y = torch.randn(N, g.Ho, g.Wo, Co, device='cuda').to(BF)
    dx = torch.empty_like(x)
    dw = torch.zeros(Co, k, k, Ci, device='cuda')
    nb = H.lib.evk_conv2d_wgrad_ws_bytes(C.byref(g))
    ws = torch.empty(max(nb // 4, 1), device='cuda')
    st = H.stream
    if kind == 'fwd':
        return lambda: H.check(H.lib.evk_conv2d_fwd(H.ptr(x), H.ptr(w), H.ptr(y), C.byref(g), st()))
    if kind == 'dgrad_flip':          # the data gradient as the trunk runs it: forward convolution over flipped weights + gate + gate statistics
        wt = torch.randn(Ci, k, k, Co, device='cuda').to(BF)
        gate = torch.relu(torch.randn_like(x)).to(BF)
        pb = max(H.lib.evk_conv_stats_bytes(N * Hh * Hh, Ci), H.lib.evk_conv3x3_halo_part_bytes(N, Hh, Hh, Ci))
        part = torch.empty(pb // 4, device='cuda')
        nblk = C.c_int32(0)
        return lambda: H.check(H.lib.evk_conv2d_dgrad_flipped_gated_stats(H.ptr(y), H.ptr(wt), None, H.ptr(gate), H.ptr(dx), C.byref(g),
                                                                          H.ptr(part), pb, C.byref(nblk), st()))
    if kind == 'dgrad_s2':            # 3x3 / stride 2 data gradient + gate + gate statistics: by output parity (default) or the gathering GEMM
        import os
        gate = torch.relu(torch.randn_like(x)).to(BF)
        pb = H.lib.evk_conv_stats_bytes(N * Hh * Hh, Ci)
        part = torch.empty(pb // 4, device='cuda')
        nblk = C.c_int32(0)
        if os.environ.get('EVK_S2_PARITY', '1') != '0':
            wc = torch.empty(9 * Ci * Co, device='cuda', dtype=BF)
            wsb = H.lib.evk_conv3x3s2_dgrad_parity_ws_bytes(C.byref(g))
            ws2 = torch.empty(wsb // 2, device='cuda', dtype=BF)
            H.check(H.lib.evk_conv3x3s2_class_weights(H.ptr(w), H.ptr(wc), Co, Ci, st()))
            return lambda: H.check(H.lib.evk_conv3x3s2_dgrad_parity(H.ptr(y), H.ptr(wc), H.ptr(gate), H.ptr(dx), C.byref(g), H.ptr(ws2), wsb, H.ptr(part),
                                                                    pb, C.byref(nblk), st()))
        return lambda: H.check(H.lib.evk_conv2d_dgrad_gated_stats(H.ptr(y), H.ptr(w), None, H.ptr(gate), H.ptr(dx), C.byref(g), H.ptr(part), pb,
                                                                  C.byref(nblk), st()))
    if kind == 'fwd_stats':
        pb = max(H.lib.evk_conv_stats_bytes(N * g.Ho * g.Wo, Co), H.lib.evk_conv3x3_halo_part_bytes(N, Hh, Hh, Co))
        part = torch.empty(pb // 4, device='cuda')
        nblk = C.c_int32(0)
        return lambda: H.check(H.lib.evk_conv2d_fwd_stats(H.ptr(x), H.ptr(w), H.ptr(y), C.byref(g), H.ptr(part), pb, C.byref(nblk), st()))
    if kind == 'dgrad':
        return lambda: H.check(H.lib.evk_conv2d_dgrad(H.ptr(y), H.ptr(w), H.ptr(dx), C.byref(g), st()))
    return lambda: H.check(H.lib.evk_conv2d_wgrad(H.ptr(y), H.ptr(x), H.ptr(dw), C.byref(g), H.ptr(ws), nb, st()))


CASES = [
    ('gemm 4640x16384x2048 NT', lambda: gemm_case(4640, 16384, 2048, 0, 0), 2 * 4640 * 16384 * 2048),
    ('gemm 4640x2048x16384 NT', lambda: gemm_case(4640, 2048, 16384, 0, 0), 2 * 4640 * 16384 * 2048),
    ('gemm 8192x8192x8192 NT', lambda: gemm_case(8192, 8192, 8192, 0, 0), 2 * 8192 ** 3),
    ('gemm 4640x16384x2048 NN(dX)', lambda: gemm_case(4640, 16384, 2048, 0, 1), 2 * 4640 * 16384 * 2048),
    ('gemm 16384x2048x4640 TN(dW)', lambda: gemm_case(16384, 2048, 4640, 3, 1), 2 * 4640 * 16384 * 2048),
    ('conv3x3 l3 fwd 64x24x24 256->256', lambda: conv_case(64, 24, 256, 256, 3, 1, 'fwd'), 2 * 36864 * 256 * 2304),
    ('conv3x3 l3 fwd+stats', lambda: conv_case(64, 24, 256, 256, 3, 1, 'fwd_stats'), 2 * 36864 * 256 * 2304),
    ('conv3x3 l3 dgrad flipped+gate+stats', lambda: conv_case(64, 24, 256, 256, 3, 1, 'dgrad_flip'), 2 * 36864 * 256 * 2304),
    ('conv3x3 l2 fwd+stats 64x48x48 128->128', lambda: conv_case(64, 48, 128, 128, 3, 1, 'fwd_stats'), 2 * 147456 * 128 * 1152),
    ('conv3x3 l4 fwd+stats 64x12x12 512->512', lambda: conv_case(64, 12, 512, 512, 3, 1, 'fwd_stats'), 2 * 9216 * 512 * 4608),
    ('conv3x3 l3 fwd+stats 128 images (decode encoder)', lambda: conv_case(128, 24, 256, 256, 3, 1, 'fwd_stats'), 2 * 73728 * 256 * 2304),
    ('conv3x3 l1 fwd+stats 64x96x96 64->64', lambda: conv_case(64, 96, 64, 64, 3, 1, 'fwd_stats'), 2 * 589824 * 64 * 576),
    ('conv3x3 l1 dgrad flipped+gate+stats 64->64', lambda: conv_case(64, 96, 64, 64, 3, 1, 'dgrad_flip'), 2 * 589824 * 64 * 576),
    ('conv3x3 s2 dgrad+gate+stats l2 64x96x96 128->128', lambda: conv_case(64, 96, 128, 128, 3, 2, 'dgrad_s2'), 2 * 147456 * 128 * 1152),
    ('conv3x3 s2 dgrad+gate+stats l3 64x48x48 256->256', lambda: conv_case(64, 48, 256, 256, 3, 2, 'dgrad_s2'), 2 * 36864 * 256 * 2304),
    ('conv3x3 s2 dgrad+gate+stats l4 64x24x24 512->512', lambda: conv_case(64, 24, 512, 512, 3, 2, 'dgrad_s2'), 2 * 9216 * 512 * 4608),
    ('conv3x3 l3 dgrad', lambda: conv_case(64, 24, 256, 256, 3, 1, 'dgrad'), 2 * 36864 * 256 * 2304),
    ('conv3x3 l3 wgrad', lambda: conv_case(64, 24, 256, 256, 3, 1, 'wgrad'), 2 * 36864 * 256 * 2304),
    ('conv3x3 l1 wgrad 64->64', lambda: conv_case(64, 96, 64, 64, 3, 1, 'wgrad'), 2 * 589824 * 64 * 576),
    ('conv1x1 l1 wgrad 256->64', lambda: conv_case(64, 96, 256, 64, 1, 1, 'wgrad'), 2 * 589824 * 64 * 256),
    ('conv1x1 l3 fwd 256->1024', lambda: conv_case(64, 24, 256, 1024, 1, 1, 'fwd'), 2 * 36864 * 256 * 1024),
    ('conv1x1 l3 fwd 1024->256', lambda: conv_case(64, 24, 1024, 256, 1, 1, 'fwd'), 2 * 36864 * 256 * 1024),
    ('conv1x1 l3 dgrad 1024->256', lambda: conv_case(64, 24, 1024, 256, 1, 1, 'dgrad'), 2 * 36864 * 256 * 1024),
    ('conv1x1 l3 fwd+stats 1024->256', lambda: conv_case(64, 24, 1024, 256, 1, 1, 'fwd_stats'), 2 * 36864 * 256 * 1024),
    ('conv1x1 l3 dgrad flipped+gate+stats 256->1024', lambda: conv_case(64, 24, 256, 1024, 1, 1, 'dgrad_flip'), 2 * 36864 * 256 * 1024),
    ('conv1x1 l2 fwd+stats 512->128', lambda: conv_case(64, 48, 512, 128, 1, 1, 'fwd_stats'), 2 * 147456 * 128 * 512),
    ('conv1x1 l3 wgrad 1024->256', lambda: conv_case(64, 24, 1024, 256, 1, 1, 'wgrad'), 2 * 36864 * 256 * 1024),
    ('conv3x3 l2 fwd 64x48x48 128->128', lambda: conv_case(64, 48, 128, 128, 3, 1, 'fwd'), 2 * 147456 * 128 * 1152),
    ('conv1x1 l1 fwd 64x96x96 64->256', lambda: conv_case(64, 96, 64, 256, 1, 1, 'fwd'), 2 * 589824 * 64 * 256),
    ('conv1x1 l1 fwd 256->64', lambda: conv_case(64, 96, 256, 64, 1, 1, 'fwd'), 2 * 589824 * 64 * 256),
    ('conv3x3 l4 fwd 64x12x12 512->512', lambda: conv_case(64, 12, 512, 512, 3, 1, 'fwd'), 2 * 9216 * 512 * 4608),
    ('gemm 3200x512x512 NT', lambda: gemm_case(3200, 512, 512, 0, 0), 2 * 3200 * 512 * 512),
    ('gemm 3200x512x1536 NT', lambda: gemm_case(3200, 512, 1536, 0, 0), 2 * 3200 * 512 * 1536),
    ('gemm 4608x512x512 NT', lambda: gemm_case(4608, 512, 512, 0, 0), 2 * 4608 * 512 * 512),
    ('gemm 960x768x768 NT', lambda: gemm_case(960, 768, 768, 0, 0), 2 * 960 * 768 * 768),
    ('gemm 3200x1536x512 NT', lambda: gemm_case(3200, 1536, 512, 0, 0), 2 * 3200 * 512 * 1536),
    ('gemm 768x512x512 NT (rm decode)', lambda: gemm_case(768, 512, 512, 0, 0), 2 * 768 * 512 * 512),
    ('gemm 768x1024x512 NT (rm decode)', lambda: gemm_case(768, 1024, 512, 0, 0), 2 * 768 * 512 * 1024),
    ('gemm 256x4608x1536 NT (cln mlp 1)', lambda: gemm_case(256, 4608, 1536, 0, 0), 2 * 256 * 4608 * 1536),
    ('gemm 256x512x512 NT (decode proj)', lambda: gemm_case(256, 512, 512, 0, 0), 2 * 256 * 512 * 512),
    ('gemm 256x1448x512 NT (logits)', lambda: gemm_case(256, 1448, 512, 0, 0), 2 * 256 * 1448 * 512),
    ('gemm 96x512x512 NT (skinny)', lambda: gemm_case(96, 512, 512, 0, 0), 2 * 96 * 512 * 512),
]

def time_cold(fn, iters=10):
    """each launch timed on its own, after a 768 MB fill that evicts L2 and the 256 MB MALL (what a training step sees)."""
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
    cold = '--cold' in sys.argv
    if cold:
        sys.argv.remove('--cold')
    sel = sys.argv[1] if len(sys.argv) > 1 else ''
    for name, mk, flops in CASES:
        if sel and sel not in name:
            continue
        fn = mk()
        ms = time_cold(fn) if cold else time_it(fn)
        print('%-40s %8.3f ms  %8.1f TF/s' % (name, ms, flops / ms / 1e9), flush=True)
