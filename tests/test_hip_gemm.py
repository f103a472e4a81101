"""GPU parity tests of the MFMA GEMM / implicit-GEMM family (evoke_amd/csrc/gemm.hip, conv.hip) through the C ABI.

Reference = torch CPU fp32 math on the same bf16-rounded operands, so the only differences are the f32
accumulation order and the final bf16 rounding of the output (tolerances stated per test)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import STORE_DTYPE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def H():
    from evoke_amd import hip
    return hip


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(STORE_DTYPE)


def close(got, want, rtol, atol):
    got, want = got.float().cpu(), want.float()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), 'max err %.4g (tol %.4g) at %s' % (
        err.max().item(), tol.flatten()[err.argmax()].item(), err.argmax().item())


def base_desc(H, A, B, Cm, M, N, K, a_mode=0, b_mode=0, lda=None, ldb=None, ldc=None):
    d = H.Gemm()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.M, d.N, d.K, d.a_mode, d.b_mode = M, N, K, a_mode, b_mode
    d.lda, d.ldb, d.ldc = lda or 0, ldb or 0, ldc or 0
    d.batch_outer = d.batch_inner = 1
    d.alpha, d.c_dtype = 1.0, H.dt(Cm)
    return d


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 136, 72), (4640, 2048, 512), (37, 64, 256), (300, 1445, 512), (1, 8, 8),
                                   (96, 512, 512), (200, 1536, 1024), (256, 1445, 512), (96, 512, 1536),
                                   (8200, 8264, 72)])       # last: 33 x 33 tiles of 256 x 256 (16-wave kernel), ragged in M, N and K
def test_gemm_nt(H, M, N, K):
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias = torch.randn(N)
    res = rnd(M, N, seed=3)
    ldc = (N + 7) // 8 * 8
    for c_dtype in (STORE_DTYPE, torch.float32):
        for act in (H.ACT_NONE, H.ACT_GELU):
            c = torch.zeros(M, ldc, dtype=c_dtype, device='cuda')
            ad, bd, biasd = a.cuda(), b.cuda(), bias.cuda()
            resd = torch.zeros(M, ldc, dtype=STORE_DTYPE, device='cuda')
            resd[:, :N] = res.cuda()
            d = base_desc(H, ad, bd, c, M, N, K, lda=K, ldb=K, ldc=ldc)
            d.bias, d.resid, d.ldr, d.r_dtype, d.act, d.alpha = biasd.data_ptr(), resd.data_ptr(), ldc, H.BF16, act, 0.5
            H.gemm_launch(d)
            torch.cuda.synchronize()
            want = 0.5 * (a.float() @ b.float().t()) + bias
            if act == H.ACT_GELU:
                want = F.gelu(want)
            want = want + res.float()
            close(c[:, :N], want, 1e-2 if c_dtype == STORE_DTYPE else 2e-4, 2e-3 * K ** 0.5 if c_dtype == torch.float32 else 2e-2 * (K / 64) ** 0.5)
            assert float(c[:, N:].abs().sum()) == 0.0


def test_gemm_batched_heads(H):
    # attention-style: Q (B,T,Hh*dh) x K (B,S,Hh*dh) -> scores (B,Hh,T,S) f32, then P.V with V K-strided
    Bz, T, S, Hh, dh = 3, 37, 50, 4, 64
    q, k, v = rnd(Bz, T, Hh * dh, seed=1), rnd(Bz, S, Hh * dh, seed=2), rnd(Bz, S, Hh * dh, seed=3)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    Sp = 56
    sc = torch.zeros(Bz, Hh, T, Sp, dtype=torch.float32, device='cuda')
    d = base_desc(H, qd, kd, sc, T, S, dh, lda=Hh * dh, ldb=Hh * dh, ldc=Sp)
    d.batch_outer, d.batch_inner = Bz, Hh
    d.sAo, d.sAi, d.sBo, d.sBi, d.sCo, d.sCi = T * Hh * dh, dh, S * Hh * dh, dh, Hh * T * Sp, T * Sp
    d.alpha = 0.125
    H.gemm_launch(d)
    want = torch.einsum('bthd,bshd->bhts', q.float().view(Bz, T, Hh, dh), k.float().view(Bz, S, Hh, dh)) * 0.125
    close(sc[..., :S], want, 2e-4, 2e-3)
    p = torch.softmax(want, -1).to(STORE_DTYPE)
    pd = torch.zeros(Bz, Hh, T, Sp, dtype=STORE_DTYPE, device='cuda')
    pd[..., :S] = p.cuda()
    ctx = torch.zeros(Bz, T, Hh * dh, dtype=STORE_DTYPE, device='cuda')
    d = base_desc(H, pd, vd, ctx, T, dh, Sp, b_mode=H.B_KSTR, lda=Sp, ldb=Hh * dh, ldc=Hh * dh)
    d.K = Sp  # padded K: the pad columns of P are zero, V rows beyond S must not be read -> use K = 56 only if V padded
    d.K = S if S % 8 == 0 else Sp
    vpad = torch.zeros(Bz, Sp, Hh * dh, dtype=STORE_DTYPE, device='cuda')
    vpad[:, :S] = vd
    d.B = vpad.data_ptr()
    d.batch_outer, d.batch_inner = Bz, Hh
    d.sAo, d.sAi, d.sBo, d.sBi, d.sCo, d.sCi = Hh * T * Sp, T * Sp, Sp * Hh * dh, dh, T * Hh * dh, dh
    H.gemm_launch(d)
    want = torch.einsum('bhts,bshd->bthd', p.float(), v.float().view(Bz, S, Hh, dh)).reshape(Bz, T, Hh * dh)
    close(ctx, want, 1e-2, 1e-2)


@pytest.mark.parametrize('M,N,K', [(4640, 512, 2048), (333, 136, 1448), (64, 64, 8)])
def test_gemm_nn_dx(H, M, N, K):
    # dX[M,N] = dY[M,K] . W[K,N]   (B K-strided)
    dy, w = rnd(M, K, seed=4), rnd(K, N, seed=5, scale=0.1)
    c = torch.zeros(M, N, dtype=STORE_DTYPE, device='cuda')
    dyd, wd = dy.cuda(), w.cuda()
    d = base_desc(H, dyd, wd, c, M, N, K, b_mode=H.B_KSTR, lda=K, ldb=N, ldc=N)
    H.gemm_launch(d)
    close(c, dy.float() @ w.float(), 1e-2, 2e-2 * (K / 64) ** 0.5 * 0.1)


@pytest.mark.parametrize('M,N,K', [(512, 2048, 4640), (1445, 512, 300), (64, 72, 128)])
def test_gemm_tn_dw_accumulate(H, M, N, K):
    # dW[M,N] += dY[K,M]^T . X[K,N]   (both K-strided, split-K with f32 atomics)
    ldy = (M + 7) // 8 * 8
    dy = torch.zeros(K, ldy, dtype=STORE_DTYPE)
    dy[:, :M] = rnd(K, M, seed=6)
    x = rnd(K, N, seed=7)
    c0 = torch.randn(M, N)
    c = c0.clone().cuda()
    dyd, xd = dy.cuda(), x.cuda()
    d = base_desc(H, dyd, xd, c, M, N, K, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=ldy, ldb=N, ldc=N)
    d.accumulate = 1
    nb = H.lib.evk_gemm_workspace_bytes(C.byref(d))
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device='cuda')
    d.workspace, d.workspace_bytes = ws.data_ptr(), nb
    H.gemm_launch(d)
    want = c0 + dy[:, :M].float().t() @ x.float()
    close(c, want, 2e-4, 3e-3 * K ** 0.5)


@pytest.mark.parametrize('M,N,K,batch', [
    (256, 1024, 9216, 1),        # layer3 conv1 / conv3 weight gradient (K = a quarter of bs 64 x 24 x 24): 16 tiles x 16 K-slices
    (1024, 256, 4672, 1),        # ... the expanding one, K-slices with a ragged last step (4672 = 73 x 64)
    (512, 2048, 1200, 1),        # layer4 shape, few steps per slice
    (128, 128, 70, 1),           # one tile, two steps, the second with 6 rows of data
    (128, 256, 40, 1),           # a single, partly filled step
    (2048, 2048, 4640, 1),       # a head / fusion linear: 256 tiles, ONE K-slice -> the tile is added to C directly (no slabs)
    (512, 512, 3200, 3),         # q | k | v of a transformer block as one batched product (batch strides on A, B and C)
])
def test_weight_gradient_kernel_tn(H, M, N, K, batch):
    """csrc/gemm_tn.hip through evk_gemm_launch (A_KSTR x B_KSTR, accumulate): dW[z][M][N] += dY[z][K][M]^T . X[z][K][N] on the shapes of
    the trunk's pointwise convolutions and of the linear layers, against fp32 CPU math on the same 16-bit
    operands; C starts non-zero (accumulation), K tails are ragged, and the routed kernel must be the new one."""
    import os
    g = torch.Generator().manual_seed(31)
    dy = (torch.randn(batch, K, M, generator=g)).to(STORE_DTYPE)
    x = (torch.randn(batch, K, N, generator=g)).to(STORE_DTYPE)
    c0 = torch.randn(batch, M, N, generator=g)
    for forced_off in (False, True):
        # second pass: the tile path of gemm.hip on the same problem (EVK_GEMM_TN is read once per process, so it is compared through the
        # public switch of the router: a K-strided tap stride of 0 with b_klog = 0 is what selects the new kernel)
        c = c0.clone().cuda()
        dyd, xd = dy.cuda(), x.cuda()
        d = base_desc(H, dyd, xd, c, M, N, K, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=M, ldb=N, ldc=N)
        d.accumulate = 1
        d.batch_outer, d.batch_inner = batch, 1
        d.sAo, d.sBo, d.sCo = K * M, K * N, M * N
        if forced_off:
            d.alpha = 1.0000001192092896        # any alpha != 1 keeps the launch on the tile path (alpha is applied there)
        nb = H.lib.evk_gemm_workspace_bytes(C.byref(d))
        ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device='cuda')
        d.workspace, d.workspace_bytes = ws.data_ptr(), nb
        H.gemm_launch(d)
        want = c0 + torch.einsum('zkm,zkn->zmn', dy.float(), x.float())
        close(c, want, 2e-4, 3e-3 * K ** 0.5)


CONVS = [  # N, Hi, Wi, Ci, Co, KH, stride, pad
    (2, 12, 12, 64, 64, 3, 1, 1), (3, 14, 10, 128, 64, 3, 2, 1), (2, 8, 8, 256, 512, 1, 2, 0),
    (2, 9, 9, 64, 256, 1, 1, 0), (1, 24, 24, 256, 256, 3, 1, 1), (2, 7, 7, 512, 512, 3, 1, 1), (2, 11, 13, 256, 64, 1, 1, 0),
    (5, 31, 29, 64, 64, 3, 1, 1),
]


def _conv_ref(x, w, stride, pad):
    return F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad)


@pytest.mark.parametrize('cfg', CONVS)
def test_conv_fwd_dgrad_wgrad(H, cfg):
    N, Hi, Wi, Ci, Co, KH, stride, pad = cfg
    x = rnd(N, Hi, Wi, Ci, seed=8)
    w = rnd(Co, KH, KH, Ci, seed=9, scale=(2.0 / (KH * KH * Ci)) ** 0.5)
    g = H.conv_geom(N, Hi, Wi, Ci, Co, KH, KH, stride, pad)
    xd, wd = x.cuda(), w.cuda()
    y = torch.zeros(N, g.Ho, g.Wo, Co, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_conv2d_fwd(H.ptr(xd), H.ptr(wd), H.ptr(y), C.byref(g), H.stream()))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    close(y, yr.detach().permute(0, 2, 3, 1), 1e-2, 2e-2)
    dy = rnd(N, g.Ho, g.Wo, Co, seed=10)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    dyd = dy.cuda()
    dx = torch.full((N, Hi, Wi, Ci), 7.0, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_conv2d_dgrad(H.ptr(dyd), H.ptr(wd), H.ptr(dx), C.byref(g), H.stream()))
    close(dx, xr.grad.permute(0, 2, 3, 1), 1e-2, 2e-2 * (KH * KH * Co / 64) ** 0.5 * w.float().std().item())
    dw = torch.zeros(Co, KH, KH, Ci, dtype=torch.float32, device='cuda')
    nb = H.lib.evk_conv2d_wgrad_ws_bytes(C.byref(g))
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device='cuda')
    H.check(H.lib.evk_conv2d_wgrad(H.ptr(dyd), H.ptr(xd), H.ptr(dw), C.byref(g), H.ptr(ws), nb, H.stream()))
    dw2 = torch.zeros_like(dw)          # fallback path without workspace (f32 atomics)
    H.check(H.lib.evk_conv2d_wgrad(H.ptr(dyd), H.ptr(xd), H.ptr(dw2), C.byref(g), None, 0, H.stream()))
    close(dw2, wr.grad.permute(0, 2, 3, 1), 2e-4, 3e-3 * (N * g.Ho * g.Wo) ** 0.5)
    close(dw, wr.grad.permute(0, 2, 3, 1), 2e-4, 3e-3 * (N * g.Ho * g.Wo) ** 0.5)


AFFINE = [  # N, Hi, Wi, Ci, Co, KH, stride, pad, resid, relu -- the routes of evk_conv2d_fwd_affine (and two geometries it refuses)
    (4, 24, 24, 256, 1024, 1, 1, 0, True, True),      # conv3 of a layer3 bottleneck + identity + ReLU: weight-stationary kernel
    (3, 48, 48, 128, 512, 1, 1, 0, True, True),       # conv3 of layer2 (K = 128), a pixel count that is no multiple of the tile
    (2, 96, 96, 64, 256, 1, 1, 0, True, False),       # conv3 of layer1 (K = 64), identity without ReLU
    (64, 24, 24, 1024, 256, 1, 1, 0, False, True),    # conv1 of layer3 at the bench's batch (256 strips): strip GEMM
    (64, 24, 24, 1024, 256, 1, 1, 0, True, True),     # strip GEMM with an identity
    (2, 24, 24, 256, 256, 3, 1, 1, False, True),      # conv2: halo kernel
    (2, 24, 24, 256, 256, 3, 1, 1, True, False),      # halo kernel with a residual, no ReLU
    (2, 48, 48, 128, 128, 3, 2, 1, False, True),      # the stride-2 conv2 of a layer's first block: tile GEMM's geometry, refused
    (2, 24, 24, 512, 1024, 1, 2, 0, False, False),    # shortcut convolution (1x1 stride 2): refused
]


@pytest.mark.parametrize('cfg', AFFINE)
def test_conv_with_batchnorm_epilogue(H, cfg):
    """evk_bn_eval_coeffs + evk_conv2d_fwd_affine (the inference form of conv -> eval-mode BatchNorm2d (-> + identity) -> ReLU,
    modules/visual_extractor.py:30-43 under model.eval()) on every route that has the epilogue -- weight-stationary, strip, halo -- against
    fp32 torch (F.conv2d -> F.batch_norm(training=False) -> + residual -> relu) on the same 16-bit operands, and BIT FOR BIT against the
    library's unfused eval forward of the same layer (evk_conv2d_fwd_stats -> evk_bn_apply): the epilogue reproduces its arithmetic, so a
    batch whose size sends a layer down another route decodes to the same tokens.  Geometries of the tile GEMM are refused (the trunk runs
    conv + bn there)."""
    N, Hi, Wi, Ci, Co, KH, stride, pad, with_resid, relu = cfg
    g = H.conv_geom(N, Hi, Wi, Ci, Co, KH, KH, stride, pad)
    gen = torch.Generator().manual_seed(77)
    x = rnd(N, Hi, Wi, Ci, seed=51)
    w = rnd(Co, KH, KH, Ci, seed=52, scale=(2.0 / (KH * KH * Ci)) ** 0.5)
    gamma = 0.5 + torch.rand(Co, generator=gen)
    beta = torch.randn(Co, generator=gen) * 0.2
    mean = torch.randn(Co, generator=gen) * 0.3
    var = 0.3 + torch.rand(Co, generator=gen)
    resid = rnd(N, g.Ho, g.Wo, Co, seed=53) if with_resid else None
    xd, wd = x.cuda(), w.cuda()
    gd, bd, md, vd = gamma.cuda(), beta.cuda(), mean.cuda(), var.cuda()          # (kept alive: a temporary's block would be handed to the next one)
    scale = torch.empty(Co, dtype=torch.float32, device='cuda')
    shift = torch.empty(Co, dtype=torch.float32, device='cuda')
    H.check(H.lib.evk_bn_eval_coeffs(H.ptr(gd), H.ptr(bd), H.ptr(md), H.ptr(vd), C.c_float(1e-5), H.ptr(scale), H.ptr(shift), Co, H.stream()), 'bn_eval_coeffs')
    sc = gamma / torch.sqrt(var + 1e-5)
    close(scale, sc, 1e-5, 1e-6)
    close(shift, beta - mean * sc, 1e-5, 2e-6)
    y = torch.full((N, g.Ho, g.Wo, Co), 7.0, dtype=STORE_DTYPE, device='cuda')
    rd = resid.cuda() if with_resid else None
    rc = H.lib.evk_conv2d_fwd_affine(H.ptr(xd), H.ptr(wd), H.ptr(y), C.byref(g), H.ptr(scale), H.ptr(shift), H.ptr(rd) if rd is not None else None,
                                     int(relu), H.stream())
    if not H.lib.evk_conv2d_fwd_affine_routes(C.byref(g)):
        assert rc == -3 and float(y.float().min()) == 7.0          # EVK_EUNSUPPORTED, nothing written
        return
    H.check(rc, 'conv2d_fwd_affine')
    yr = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, stride, pad)
    yr = F.batch_norm(yr, mean, var, gamma, beta, training=False, eps=1e-5)
    if with_resid:
        yr = yr + resid.float().permute(0, 3, 1, 2)
    if relu:
        yr = torch.relu(yr)
    rt = 2.0 ** -7 if STORE_DTYPE == torch.bfloat16 else 2.0 ** -9        # (two roundings, as in the unfused forward)
    close(y, yr.permute(0, 2, 3, 1), rt, 4e-3 if STORE_DTYPE == torch.float16 else 3e-2)       # (the conv output's rounding, scaled by gamma / sigma, where the shift cancels it)
    y0 = torch.empty_like(y)
    y1 = torch.empty_like(y)
    nblk = C.c_int32(0)
    H.check(H.lib.evk_conv2d_fwd_stats(H.ptr(xd), H.ptr(wd), H.ptr(y0), C.byref(g), None, 0, C.byref(nblk), H.stream()), 'conv2d_fwd_stats')
    H.check(H.lib.evk_bn_apply(H.ptr(y0), H.ptr(scale), H.ptr(shift), H.ptr(rd) if rd is not None else None, H.ptr(y1), N * g.Ho * g.Wo, Co, int(relu),
                               H.stream()), 'bn_apply')
    assert torch.equal(y, y1), 'the fused epilogue differs from conv -> bn_apply (max %.3e)' % float((y.float() - y1.float()).abs().max())


@pytest.mark.parametrize('N,Hh,W', [(2, 32, 32), (1, 64, 48), (2, 64, 128), (3, 40, 256)])       # the last two: the halo kernel of stem.hip
def test_stem(H, N, Hh, W):
    g = torch.Generator().manual_seed(11)
    img = torch.randn(N, 3, Hh, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    imgd, wdev = img.cuda(), w.cuda()
    xpad = torch.empty(N, Hh + 6, W + 8, 4, dtype=STORE_DTYPE, device='cuda')
    wp = torch.empty(64, 7, 8, 4, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_stem_pack_image(H.ptr(imgd), H.ptr(xpad), N, Hh, W, H.stream()))
    H.check(H.lib.evk_stem_pack_weight(H.ptr(wdev), H.ptr(wp), H.stream()))
    y = torch.zeros(N, Hh // 2, W // 2, 64, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_stem_fwd(H.ptr(xpad), H.ptr(wp), H.ptr(y), N, Hh, W, H.stream()))
    if H.lib.evk_stem_halo_supported(N, Hh, W) == 1:
        # the same call with the batch-norm partials: sums of the unrounded result
        nb = max(H.lib.evk_conv_stats_bytes(N * (Hh // 2) * (W // 2), 64), H.lib.evk_stem_halo_part_bytes(N, Hh, W))
        part = torch.full((nb // 4,), 3.0, device='cuda')
        nblk = C.c_int32(0)
        y2 = torch.zeros_like(y)
        H.check(H.lib.evk_stem_fwd_stats(H.ptr(xpad), H.ptr(wp), H.ptr(y2), N, Hh, W, H.ptr(part), nb, C.byref(nblk), H.stream()))
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
        yf = F.conv2d(img.to(STORE_DTYPE).float(), w.to(STORE_DTYPE).float(), None, 2, 3).permute(0, 2, 3, 1).reshape(-1, 64)
        sums = part[:nblk.value * 128].view(nblk.value, 2, 64).sum(0).cpu()
        assert float((sums[0] - yf.sum(0)).abs().max()) <= 1e-5 * float(yf.abs().sum(0).max()) + 1e-3
        assert float((sums[1] - (yf ** 2).sum(0)).abs().max()) <= 1e-5 * float((yf ** 2).sum(0).max()) + 1e-3
    ir = img.to(STORE_DTYPE).float().requires_grad_(False)
    wr = w.to(STORE_DTYPE).float().requires_grad_(True)
    yr = F.conv2d(ir, wr, None, 2, 3)
    close(y, yr.detach().permute(0, 2, 3, 1), 1e-2, 3e-2)
    dy = rnd(N, Hh // 2, W // 2, 64, seed=12)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    dwp = torch.zeros(64, 7, 8, 4, dtype=torch.float32, device='cuda')
    dyd = dy.cuda()
    nb = H.lib.evk_stem_wgrad_ws_bytes(N, Hh, W)
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device='cuda')
    H.check(H.lib.evk_stem_wgrad(H.ptr(dyd), H.ptr(xpad), H.ptr(dwp), N, Hh, W, H.ptr(ws), nb, H.stream()))
    dw = torch.ones(64, 3, 7, 7, dtype=torch.float32, device='cuda')
    H.check(H.lib.evk_stem_unpack_wgrad(H.ptr(dwp), H.ptr(dw), H.stream()))
    close(dw - 1.0, wr.grad, 2e-4, 3e-3 * (N * Hh * W / 4) ** 0.5)


def test_gemm_rejects_bad_args(H):
    a = torch.zeros(8, 12, dtype=STORE_DTYPE, device='cuda')
    c = torch.zeros(8, 8, dtype=STORE_DTYPE, device='cuda')
    d = base_desc(H, a, a, c, 8, 8, 12, lda=12, ldb=12, ldc=8)
    with pytest.raises(RuntimeError):
        H.gemm_launch(d)


# Every kernel ROUTE the library can be switched to (csrc: evk_tunable, honoured only under EVK_EXPERIMENTAL=1) besides the measured default: the
# routes that were the default in earlier rounds and stay as fallbacks for geometries the specialised kernels refuse.  One child interpreter
# (the switches are read once per process) runs the kernel suites and one model-level parity case with ALL of them forced at once.
FALLBACK_ROUTES = dict(
    EVK_EXPERIMENTAL='1',
    EVK_TILE256='1',                 # 256 x 256 / 16-wave tile for every plain NT product with M, N >= 256 (default: >= 1024 big tiles only)
    EVK_GEMM_TN='0',                 # weight gradients on the tile kernel instead of csrc/gemm_tn.hip
    EVK_GEMM_STRIP='0',              # contracting 1x1 convolutions on the tile kernel instead of the strip GEMM
    EVK_CONV1X1_WS='0',              # expanding 1x1 convolutions on the tile kernel instead of the weight-stationary kernel
    EVK_CONV3X3_HALO='0', EVK_CONV3X3_WGRAD_HALO='0', EVK_STEM_HALO='0',          # implicit-GEMM 3x3 / stem instead of the halo-tile kernels
    EVK_S2_PARITY='0', EVK_DGRAD_FLIP='0', EVK_DOWN_COMPACT='0',                  # gathering data gradients instead of parity classes / flipped weights / compact shortcut
    EVK_BN_GATE_STATS='0', EVK_BN_XSTATS='0', EVK_BN_FOLD='0',                    # batch-norm backward sums by their own reductions
    EVK_BEAM_STEP_FAST='0', EVK_DECODE_RB_SPLIT='0')                              # batched-walk beam step, one workgroup per row block


@pytest.mark.skipif(os.environ.get('EVK_TILE256') is not None, reason='already running on the fallback routes')
def test_kernel_suite_on_the_fallback_routes():
    """gemm.hip picks the 256 x 256 / 16-wave tile only for plain NT products of >= 1024 big tiles, the convolutions go to halo / strip /
    weight-stationary kernels, batch-norm backward sums ride on data-gradient epilogues ...: FALLBACK_ROUTES forces the other side of every such
    choice (batched attention scores, linears with bias / activation / residual epilogues and ragged edges on the big tile; every convolution
    on the implicit-GEMM tile path; the trunk runner without its fused statistics), and the kernel tests plus one FineTune parity case run once
    more that way in a child interpreter.  Left out: the tests that assert a specialised kernel TOOK a launch (the strip-GEMM routing test, the
    halo weight-gradient and stride-2 parity tests, which end with `routed entry point == specialised kernel, bit for bit`)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(here, 'test_hip_gemm.py'), os.path.join(here, 'test_hip_ops.py'),
                        os.path.join(here, 'test_model_gpu.py'), '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', '-k',
                        # the tests whose launches pass through a forced switch: products and linears (big tile, tile-kernel weight gradients),
                        # convolutions / stem / trunk runner / batch norm (implicit-GEMM and unfused-statistics routes), beam step, row blocks
                        '(gemm_ or linear or attention or conv or stem or trunk or batchnorm or dgrad or gate_statistics or beam_step or rowblock or '
                        '(finetune_matches_reference and ft224_inc)) and not strip_gemm_is_what and not fallback_routes '
                        'and not halo_conv3x3_weight_gradient and not stride2_conv3x3_data_gradient_by_output_parity'],
                       env=dict(os.environ, **FALLBACK_ROUTES), capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    tail = '\n'.join((r.stdout + r.stderr).splitlines()[-20:])
    print(tail)
    assert r.returncode == 0, tail


WS_SHAPES = [(980, 256, 1024), (4608, 64, 256), (2304, 128, 512), (1154, 512, 2048), (2309, 1024, 256), (1000, 512, 128), (1, 64, 256)]


@pytest.mark.parametrize('M,K,N', WS_SHAPES)
def test_weight_stationary_pointwise_conv_forward_and_data_gradient(H, M, K, N):
    """csrc/conv1x1.hip through its own C-ABI entry points (evk_conv1x1_ws_fwd / _dgrad), ragged M included: the output against the
    fp32 product of the same rounded operands (one output rounding: 2^-8 relative in bf16, 2^-11 in fp16, plus the f32 summation order),
    the per-channel statistics partials against the sums of the UNROUNDED product, and the tile GEMM path as a second witness."""
    assert H.lib.evk_conv1x1_ws_supported(M, K, N) == 1
    x, w = rnd(M, K, seed=11, scale=0.7).cuda(), rnd(N, K, seed=12, scale=0.05).cuda()
    ref = x.float() @ w.float().t()
    y = torch.empty(M, N, dtype=STORE_DTYPE, device='cuda')
    nb = H.lib.evk_conv1x1_ws_part_bytes(M, K, N)
    part = torch.zeros(nb // 4, device='cuda')
    nblk = C.c_int32(0)
    H.check(H.lib.evk_conv1x1_ws_fwd(H.ptr(x), H.ptr(w), H.ptr(y), M, K, N, H.ptr(part), nb, C.byref(nblk), H.stream()))
    torch.cuda.synchronize()
    rt = 2.0 ** -8 if STORE_DTYPE == torch.bfloat16 else 2.0 ** -10
    close(y, ref.cpu(), rt, 1e-3)
    s = part[:nblk.value * 2 * N].view(nblk.value, 2, N).sum(0)
    assert float((s[0] - ref.sum(0)).abs().max()) <= 1e-5 * float(ref.abs().sum(0).max()) + 1e-4
    assert float((s[1] - (ref ** 2).sum(0)).abs().max()) <= 1e-5 * float((ref ** 2).sum(0).max()) + 1e-4
    # second witness: the tile path on the same operands rounds the same f32 sums (summation order differs: allow one ulp)
    g = H.conv_geom(1, 1, M, K, N, 1, 1, 1, 0)
    y2 = torch.empty(M, N, dtype=STORE_DTYPE, device='cuda')
    nb2 = H.lib.evk_conv_stats_bytes(M, N)
    part2 = torch.zeros(nb2 // 4, device='cuda')
    n2 = C.c_int32(0)
    H.check(H.lib.evk_conv2d_fwd_stats_tile(H.ptr(x), H.ptr(w), H.ptr(y2), C.byref(g), H.ptr(part2), nb2, C.byref(n2), H.stream()))
    torch.cuda.synchronize()
    close(y, y2.float().cpu(), 2 * rt, 1e-3)
    # data gradient: dx = gate(dy . W + skip), W stored [K][N]; gate statistics = (sum g, sum g * z) per channel
    wt = w.t().contiguous()
    skip = rnd(M, N, seed=13, scale=0.3).cuda()
    gate = torch.relu(rnd(M, N, seed=14).float()).to(STORE_DTYPE).cuda()
    dx = torch.empty(M, N, dtype=STORE_DTYPE, device='cuda')
    part.zero_()
    H.check(H.lib.evk_conv1x1_ws_dgrad(H.ptr(x), H.ptr(wt), H.ptr(skip), H.ptr(gate), H.ptr(dx), M, K, N, H.ptr(part), nb, C.byref(nblk), H.stream()))
    torch.cuda.synchronize()
    gref = (ref + skip.float()) * (gate.float() > 0)
    close(dx, gref.cpu(), rt, 1e-3)
    s = part[:nblk.value * 2 * N].view(nblk.value, 2, N).sum(0)
    gz = gref * gate.float()
    assert float((s[0] - gref.sum(0)).abs().max()) <= 1e-5 * float(gref.abs().sum(0).max()) + 1e-4
    assert float((s[1] - gz.sum(0)).abs().max()) <= 1e-5 * float(gz.abs().sum(0).max()) + 1e-4
    # third partial row: sum over pixels of dx * (stat_x - mean) with the ROUNDED dx (what a later pass over dx would have read)
    sx = rnd(M, N, seed=15, scale=1.5).cuda()
    mean = torch.randn(N, device='cuda') * 0.5
    part3 = torch.zeros(nb // 4 * 3 // 2, device='cuda')
    dx2 = torch.empty_like(dx)
    H.check(H.lib.evk_conv1x1_ws_dgrad_xstat(H.ptr(x), H.ptr(wt), H.ptr(skip), H.ptr(gate), H.ptr(dx2), M, K, N, H.ptr(sx), H.ptr(mean),
                                             H.ptr(part3), part3.numel() * 4, C.byref(nblk), H.stream()))
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx)
    s3 = part3[:nblk.value * 3 * N].view(nblk.value, 3, N).sum(0)
    assert torch.allclose(s3[:2], s, rtol=0, atol=0)
    want = (dx.float() * (sx.float() - mean)).sum(0)
    scale = float((dx.float() * (sx.float() - mean)).abs().sum(0).max())
    assert float((s3[2] - want).abs().max()) <= 1e-5 * scale + 1e-4


STRIP_SHAPES = [(2304, 128, 256), (1000, 256, 64), (289, 128, 128), (5000, 384, 192), (36864, 256, 1024), (1, 128, 64), (577, 128, 640)]


@pytest.mark.parametrize('M,N,K', STRIP_SHAPES)
def test_strip_gemm_forward_statistics_and_gated_data_gradient(H, M, N, K):
    """csrc/gemm_strip.hip through its own entry point: one to sixteen K steps (the pipeline's prologue / clamped tail loads), ragged
    last strip, both epilogues.  Against the fp32 product of the same rounded operands (one output rounding + the f32 summation order);
    statistics partials against the sums of the UNROUNDED / gated result."""
    assert H.lib.evk_gemm_strip_supported(M, N, K) == 1
    a, b = rnd(M, K, seed=41, scale=0.7).cuda(), rnd(N, K, seed=42, scale=(1.0 / K) ** 0.5).cuda()
    ref = (a.float() @ b.float().t()).cpu()
    c = torch.full((M, N), 9.0, dtype=STORE_DTYPE, device='cuda')
    nb = H.lib.evk_gemm_strip_part_bytes(M, N)
    part = torch.full((nb // 4,), 3.0, device='cuda')
    nblk = C.c_int32(0)
    H.check(H.lib.evk_gemm_strip(H.ptr(a), K, H.ptr(b), K, H.ptr(c), N, M, N, K, None, 0, None, 0, H.ptr(part), None, nb, C.byref(nblk), H.stream()))
    torch.cuda.synchronize()
    rt = 2.0 ** -8 if STORE_DTYPE == torch.bfloat16 else 2.0 ** -10
    close(c, ref, rt, 2e-3)
    assert nblk.value * 2 * N * 4 == nb
    s = part.view(nblk.value, 2, N).sum(0).cpu()
    assert float((s[0] - ref.sum(0)).abs().max()) <= 1e-5 * float(ref.abs().sum(0).max()) + 1e-4
    assert float((s[1] - (ref ** 2).sum(0)).abs().max()) <= 1e-5 * float((ref ** 2).sum(0).max()) + 1e-4
    skip = rnd(M, N, seed=43, scale=0.3).cuda()
    gate = torch.relu(rnd(M, N, seed=44).float()).to(STORE_DTYPE).cuda()
    c2 = torch.full((M, N), 5.0, dtype=STORE_DTYPE, device='cuda')
    part.fill_(3.0)
    H.check(H.lib.evk_gemm_strip(H.ptr(a), K, H.ptr(b), K, H.ptr(c2), N, M, N, K, H.ptr(skip), N, H.ptr(gate), N, None, H.ptr(part), nb, C.byref(nblk),
                                 H.stream()))
    torch.cuda.synchronize()
    gref = (ref + skip.float().cpu()) * (gate.float().cpu() > 0)
    close(c2, gref, rt, 2e-3)
    s = part.view(nblk.value, 2, N).sum(0).cpu()
    gz = gref * gate.float().cpu()
    assert float((s[0] - gref.sum(0)).abs().max()) <= 2e-5 * float(gref.abs().sum(0).max()) + 1e-3
    assert float((s[1] - gz.sum(0)).abs().max()) <= 2e-5 * float(gz.abs().sum(0).max()) + 1e-3


def test_strip_gemm_is_what_the_contracting_pointwise_convolutions_take(H):
    """Routing: Bottleneck.conv1 forward and conv3's data gradient over transposed weights at layer3's size go to the strip kernel and
    agree with the tile path (EVK_A_PLAIN / EVK_B_KSTR loaders) to one output ulp."""
    N, Hh, Ci, Co = 64, 24, 1024, 256
    g = H.conv_geom(N, Hh, Hh, Ci, Co, 1, 1, 1, 0)
    M = N * Hh * Hh
    assert H.lib.evk_gemm_strip_routes(M, Co, Ci, 0, 0) == 1
    x, w = rnd(M, Ci, seed=45, scale=0.7).cuda(), rnd(Co, Ci, seed=46, scale=Ci ** -0.5).cuda()
    y = torch.empty(M, Co, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_conv2d_fwd(H.ptr(x), H.ptr(w), H.ptr(y), C.byref(g), H.stream()))
    d = base_desc(H, x, w, torch.empty_like(y), M, Co, Ci, lda=Ci, ldb=Ci, ldc=Co)
    y2 = torch.empty_like(y)
    d.C = y2.data_ptr()
    H.gemm_launch(d)
    torch.cuda.synchronize()
    rt = 2.0 ** -8 if STORE_DTYPE == torch.bfloat16 else 2.0 ** -10
    close(y, y2.float().cpu(), 2 * rt, 2e-3)
    # data gradient of a Co = 4 Ci pointwise convolution over weights transposed by evk_conv_flip_weights
    g3 = H.conv_geom(N, Hh, Hh, Co, Ci, 1, 1, 1, 0)          # conv3: 256 -> 1024
    w3 = rnd(Ci, 1, 1, Co, seed=47, scale=Co ** -0.5).cuda()
    wt = torch.empty(Co, 1, 1, Ci, dtype=STORE_DTYPE, device='cuda')
    one = lambda v: (C.c_int32 * 1)(v)
    H.check(H.lib.evk_conv_flip_weights((C.c_void_p * 1)(w3.data_ptr()), (C.c_void_p * 1)(wt.data_ptr()), one(Ci), one(Co), one(1), one(1), 1, H.stream()))
    dy = rnd(M, Ci, seed=48).cuda()
    gate = torch.relu(rnd(M, Co, seed=49).float()).to(STORE_DTYPE).cuda()
    dx, dx2 = torch.empty(M, Co, dtype=STORE_DTYPE, device='cuda'), torch.empty(M, Co, dtype=STORE_DTYPE, device='cuda')
    nb = max(H.lib.evk_conv_stats_bytes(M, Co), H.lib.evk_gemm_strip_part_bytes(M, Co))
    p1, p2 = torch.zeros(nb // 4, device='cuda'), torch.zeros(nb // 4, device='cuda')
    n1, n2 = C.c_int32(0), C.c_int32(0)
    H.check(H.lib.evk_conv2d_dgrad_flipped_gated_stats(H.ptr(dy), H.ptr(wt), None, H.ptr(gate), H.ptr(dx), C.byref(g3), H.ptr(p1), nb, C.byref(n1), H.stream()))
    H.check(H.lib.evk_conv2d_dgrad_gated_stats(H.ptr(dy), H.ptr(w3), None, H.ptr(gate), H.ptr(dx2), C.byref(g3), H.ptr(p2), nb, C.byref(n2), H.stream()))
    torch.cuda.synchronize()
    close(dx, dx2.float().cpu(), 2 * rt, 2e-3)
    s1 = p1[:n1.value * 2 * Co].view(n1.value, 2, Co).sum(0)
    s2 = p2[:n2.value * 2 * Co].view(n2.value, 2, Co).sum(0)
    assert float((s1 - s2).abs().max()) <= 2e-5 * float(s2.abs().max()) + 1e-3


HALO_SHAPES = [(2, 24, 24, 256, 256), (5, 7, 7, 64, 128), (3, 14, 14, 128, 256), (2, 48, 48, 128, 128), (3, 12, 12, 512, 512),
               (3, 10, 23, 64, 128), (1, 16, 20, 64, 384), (7, 5, 40, 128, 128), (2, 96, 96, 64, 64), (3, 20, 20, 64, 192), (5, 9, 31, 64, 64)]


@pytest.mark.parametrize('N,Hh,W,Ci,Co', HALO_SHAPES)
def test_halo_conv3x3_forward_and_flipped_data_gradient(H, N, Hh, W, Ci, Co):
    """csrc/conv3x3.hip through its own entry point (evk_conv3x3_halo): tiles of whole rows that cross image boundaries (7 x 7 and
    5-row images: several images per tile), ragged last tiles, one to eight 64-channel chunks.  Forward against the fp32 convolution
    of the same rounded operands (one output rounding + the f32 summation order), the batch-norm partials against the sums of the
    UNROUNDED result, the tile path as a second witness; then the data gradient as a forward convolution over the flipped weights with
    the residual / ReLU gate / gate statistics epilogue."""
    assert H.lib.evk_conv3x3_halo_supported(N, Hh, W, Ci, Co) == 1
    x = rnd(N, Hh, W, Ci, seed=21, scale=0.7).cuda()
    w = rnd(Co, 3, 3, Ci, seed=22, scale=(2.0 / (9 * Ci)) ** 0.5).cuda()
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1).contiguous()
    y = torch.full((N, Hh, W, Co), 9.0, dtype=STORE_DTYPE, device='cuda')
    nb = H.lib.evk_conv3x3_halo_part_bytes(N, Hh, W, Co)
    part = torch.full((nb // 4,), 3.0, device='cuda')
    nblk = C.c_int32(0)
    H.check(H.lib.evk_conv3x3_halo(H.ptr(x), H.ptr(w), H.ptr(y), N, Hh, W, Ci, Co, None, 0, None, 0, H.ptr(part), None, nb, C.byref(nblk), H.stream()))
    torch.cuda.synchronize()
    rt = 2.0 ** -8 if STORE_DTYPE == torch.bfloat16 else 2.0 ** -10
    close(y, ref, rt, 2e-3)
    assert nblk.value * 2 * Co * 4 == nb
    s = part.view(nblk.value, 2, Co).sum(0).cpu()
    flat = ref.reshape(-1, Co)
    assert float((s[0] - flat.sum(0)).abs().max()) <= 1e-5 * float(flat.abs().sum(0).max()) + 1e-4
    assert float((s[1] - (flat ** 2).sum(0)).abs().max()) <= 1e-5 * float((flat ** 2).sum(0).max()) + 1e-4
    # second witness: the implicit-GEMM tile path (EVK_A_CONV loader) on the same operands
    g = H.conv_geom(N, Hh, W, Ci, Co, 3, 3, 1, 1)
    d = H.Gemm()
    y2 = torch.empty_like(y)
    d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), y2.data_ptr()
    d.M, d.N, d.K, d.a_mode, d.b_mode = N * Hh * W, Co, 9 * Ci, H.A_CONV, H.B_PLAIN
    d.lda, d.ldb, d.ldc = Ci, 9 * Ci, Co
    d.batch_outer = d.batch_inner = 1
    d.alpha, d.c_dtype = 1.0, H.dt(y2)
    d.g = g
    d.g.sN, d.g.sH, d.g.sW = Hh * W * Ci, W * Ci, Ci
    H.gemm_launch(d)
    torch.cuda.synchronize()
    close(y, y2.float().cpu(), 2 * rt, 2e-3)
    # data gradient of a Co -> Ci ... here: the same kernel as "dy [.., Ci] -> dx [.., Co]" over weights flipped by evk_conv_flip_weights
    wf = torch.empty(Ci, 3, 3, Co, dtype=STORE_DTYPE, device='cuda')          # wt[ci][2-kh][2-kw][co] = w[co][kh][kw][ci]
    ptrs_w, ptrs_wt = (C.c_void_p * 1)(w.data_ptr()), (C.c_void_p * 1)(wf.data_ptr())
    one = lambda v: (C.c_int32 * 1)(v)
    H.check(H.lib.evk_conv_flip_weights(ptrs_w, ptrs_wt, one(Co), one(Ci), one(3), one(3), 1, H.stream()))
    torch.cuda.synchronize()
    assert torch.equal(wf.cpu(), w.cpu().flip(1, 2).permute(3, 1, 2, 0).contiguous())
    if H.lib.evk_conv3x3_halo_supported(N, Hh, W, Co, Ci) == 1:
        # dgrad of the forward conv (Ci -> Co): input dy has Co channels, output dx has Ci channels, weights wf [Ci][3][3][Co]
        dy = rnd(N, Hh, W, Co, seed=23).cuda()
        skip = rnd(N, Hh, W, Ci, seed=24, scale=0.3).cuda()
        gate = torch.relu(rnd(N, Hh, W, Ci, seed=25).float()).to(STORE_DTYPE).cuda()
        dx = torch.full((N, Hh, W, Ci), 5.0, dtype=STORE_DTYPE, device='cuda')
        nb2 = H.lib.evk_conv3x3_halo_part_bytes(N, Hh, W, Ci)
        part2 = torch.full((nb2 // 4,), 3.0, device='cuda')
        H.check(H.lib.evk_conv3x3_halo(H.ptr(dy), H.ptr(wf), H.ptr(dx), N, Hh, W, Co, Ci, H.ptr(skip), Ci, H.ptr(gate), Ci, None, H.ptr(part2),
                                       nb2, C.byref(nblk), H.stream()))
        torch.cuda.synchronize()
        dref = F.conv_transpose2d(dy.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        gref = (dref + skip.float().cpu()) * (gate.float().cpu() > 0)
        close(dx, gref, rt, 2e-2 * (9 * Co / 64) ** 0.5 * float(w.float().std()))
        s = part2.view(nblk.value, 2, Ci).sum(0).cpu()
        gf, zf = gref.reshape(-1, Ci), gate.float().cpu().reshape(-1, Ci)
        assert float((s[0] - gf.sum(0)).abs().max()) <= 2e-5 * float(gf.abs().sum(0).max()) + 1e-3
        assert float((s[1] - (gf * zf).sum(0)).abs().max()) <= 2e-5 * float((gf * zf).abs().sum(0).max()) + 1e-3


WG_HALO_SHAPES = [(2, 24, 24, 256, 256), (5, 7, 7, 64, 128), (3, 14, 14, 128, 64), (2, 48, 48, 128, 128), (3, 12, 12, 192, 320),
                  (3, 10, 23, 64, 64), (2, 96, 96, 64, 64), (7, 6, 40, 128, 64), (64, 8, 8, 64, 64)]


@pytest.mark.parametrize('N,Hh,W,Ci,Co', WG_HALO_SHAPES)
def test_halo_conv3x3_weight_gradient(H, N, Hh, W, Ci, Co):
    """csrc/conv3x3.hip, evk_conv3x3_wgrad_halo: dw += dy^T x over all nine taps from one dy tile + one x halo tile in LDS, K-slices through
    f32 slabs.  Against the fp32 weight gradient of the same rounded operands (f32 accumulation order only: tolerance ~ sqrt(pixels)
    ulps of f32 on sums of 16-bit products), accumulation into a non-zero dw, tiles at the top / bottom edge of an image, tiles whose
    pixel count is not a multiple of the 32-pixel MFMA step, uneven K-slices."""
    assert H.lib.evk_conv3x3_wgrad_halo_supported(N, Hh, W, Ci, Co) == 1
    x = rnd(N, Hh, W, Ci, seed=31, scale=0.7)
    dy = rnd(N, Hh, W, Co, seed=32, scale=0.5)
    xr = x.float().permute(0, 3, 1, 2)
    wr = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(xr, wr, None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    want = wr.grad.permute(0, 2, 3, 1).contiguous()                   # [Co][3][3][Ci]
    nb = H.lib.evk_conv3x3_wgrad_halo_ws_bytes(N, Hh, W, Ci, Co)
    ws = torch.full((nb // 4,), float('nan'), device='cuda')
    dw = torch.full((Co, 3, 3, Ci), 0.5, dtype=torch.float32, device='cuda')
    xd, dyd = x.cuda(), dy.cuda()
    H.check(H.lib.evk_conv3x3_wgrad_halo(H.ptr(dyd), H.ptr(xd), H.ptr(dw), N, Hh, W, Ci, Co, H.ptr(ws), nb, H.stream()))
    torch.cuda.synchronize()
    close(dw - 0.5, want, 2e-4, 3e-3 * (N * Hh * W) ** 0.5)
    # the routed entry point takes the same kernel: identical result
    g = H.conv_geom(N, Hh, W, Ci, Co, 3, 3, 1, 1)
    nb2 = H.lib.evk_conv2d_wgrad_ws_bytes(C.byref(g))
    assert nb2 >= nb
    ws2 = torch.empty(nb2 // 4, device='cuda')
    dw2 = torch.full((Co, 3, 3, Ci), 0.5, dtype=torch.float32, device='cuda')
    H.check(H.lib.evk_conv2d_wgrad(H.ptr(dyd), H.ptr(xd), H.ptr(dw2), C.byref(g), H.ptr(ws2), nb2, H.stream()))
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)


@pytest.mark.parametrize('N,Ho,Ci,Co', [(2, 8, 128, 128), (1, 12, 256, 64), (3, 5, 64, 256), (2, 6, 512, 512)])
def test_stride2_conv3x3_data_gradient_by_output_parity(H, N, Ho, Ci, Co):
    """conv.hip, evk_conv3x3s2_dgrad_parity: the data gradient of a 3x3 / stride 2 / pad 1 convolution as four small stride-1 convolutions
    over dy (one per parity class of the input pixel) + the interleaving pass with the ReLU gate and the gate statistics, against the fp32
    transposed convolution of the same rounded operands and against the gathering GEMM (evk_conv2d_dgrad_gated_stats)."""
    Hi = 2 * Ho
    g = H.conv_geom(N, Hi, Hi, Ci, Co, 3, 3, 2, 1)
    assert g.Ho == Ho and H.lib.evk_conv3x3s2_dgrad_parity_supported(C.byref(g)) == 1
    w = rnd(Co, 3, 3, Ci, seed=51, scale=(2.0 / (9 * Co)) ** 0.5).cuda()
    dy = rnd(N, Ho, Ho, Co, seed=52).cuda()
    gate = torch.relu(rnd(N, Hi, Hi, Ci, seed=53).float()).to(STORE_DTYPE).cuda()
    wc = torch.empty(9 * Ci * Co, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_conv3x3s2_class_weights(H.ptr(w), H.ptr(wc), Co, Ci, H.stream()))
    wsb = H.lib.evk_conv3x3s2_dgrad_parity_ws_bytes(C.byref(g))
    ws = torch.empty(wsb // 2, dtype=STORE_DTYPE, device='cuda')
    nb = H.lib.evk_conv_stats_bytes(N * Hi * Hi, Ci)
    part, part2 = torch.full((nb // 4,), 3.0, device='cuda'), torch.zeros(nb // 4, device='cuda')
    n1, n2 = C.c_int32(0), C.c_int32(0)
    dx = torch.full((N, Hi, Hi, Ci), 5.0, dtype=STORE_DTYPE, device='cuda')
    H.check(H.lib.evk_conv3x3s2_dgrad_parity(H.ptr(dy), H.ptr(wc), H.ptr(gate), H.ptr(dx), C.byref(g), H.ptr(ws), wsb, H.ptr(part), nb, C.byref(n1), H.stream()))
    dx2 = torch.empty_like(dx)
    H.check(H.lib.evk_conv2d_dgrad_gated_stats(H.ptr(dy), H.ptr(w), None, H.ptr(gate), H.ptr(dx2), C.byref(g), H.ptr(part2), nb, C.byref(n2), H.stream()))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dy.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), None, 2, 1, output_padding=1).permute(0, 2, 3, 1)
    gref = ref * (gate.float().cpu() > 0)
    rt = 2.0 ** -8 if STORE_DTYPE == torch.bfloat16 else 2.0 ** -10
    close(dx, gref, rt, 2e-3)
    close(dx, dx2.float().cpu(), 2 * rt, 2e-3)
    s1 = part[:n1.value * 2 * Ci].view(n1.value, 2, Ci).sum(0).cpu()
    gf, zf = dx.float().cpu().reshape(-1, Ci), gate.float().cpu().reshape(-1, Ci)           # statistics of the stored (rounded) gradient
    assert float((s1[0] - gf.sum(0)).abs().max()) <= 2e-5 * float(gf.abs().sum(0).max()) + 1e-3
    assert float((s1[1] - (gf * zf).sum(0)).abs().max()) <= 2e-5 * float((gf * zf).abs().sum(0).max()) + 1e-3
    s2 = part2[:n2.value * 2 * Ci].view(n2.value, 2, Ci).sum(0).cpu()
    assert float((s1 - s2).abs().max()) <= 4 * rt * float(gf.abs().sum(0).max()) + 1e-2


def test_halo_conv3x3_refuses_what_it_cannot_tile(H):
    assert H.lib.evk_conv3x3_halo_supported(1, 8, 200, 64, 128) == 0       # the halo of even one 200-pixel row exceeds the LDS buffer
    assert H.lib.evk_conv3x3_halo_supported(4, 4, 5, 64, 128) == 1         # tiny images: many per tile
    assert H.lib.evk_conv3x3_halo_supported(2, 24, 24, 96, 128) == 0       # C % 64
    assert H.lib.evk_conv3x3_halo_supported(2, 24, 24, 64, 32) == 0        # Co % 64
    assert H.lib.evk_conv3x3_halo_supported(2, 24, 24, 128, 64) == 0       # 64-channel tiles: single-chunk inputs only
    assert H.lib.evk_conv3x3_wgrad_halo_supported(7, 5, 40, 128, 64) == 0  # no divisor of H = 5 gives a tile of >= 48 pixels that fits
    assert H.lib.evk_conv3x3_wgrad_halo_supported(2, 24, 24, 96, 64) == 0  # Ci % 64
    x = torch.zeros(2, 24, 24, 96, dtype=STORE_DTYPE, device='cuda')
    w = torch.zeros(128, 3, 3, 96, dtype=STORE_DTYPE, device='cuda')
    y = torch.zeros(2, 24, 24, 128, dtype=STORE_DTYPE, device='cuda')
    assert H.lib.evk_conv3x3_halo(H.ptr(x), H.ptr(w), H.ptr(y), 2, 24, 24, 96, 128, None, 0, None, 0, None, None, 0, None, H.stream()) != 0


def test_weight_stationary_kernel_refuses_what_it_cannot_tile(H):
    """Shapes outside the instantiated set are reported as unsupported and the entry point returns an error (no fallback inside it)."""
    assert H.lib.evk_conv1x1_ws_supported(100, 96, 256) == 0
    assert H.lib.evk_conv1x1_ws_supported(100, 64, 200) == 0
    x = torch.zeros(100, 96, dtype=STORE_DTYPE, device='cuda')
    w = torch.zeros(256, 96, dtype=STORE_DTYPE, device='cuda')
    y = torch.zeros(100, 256, dtype=STORE_DTYPE, device='cuda')
    part = torch.zeros(1 << 16, device='cuda')
    n = C.c_int32(0)
    assert H.lib.evk_conv1x1_ws_fwd(H.ptr(x), H.ptr(w), H.ptr(y), 100, 96, 256, H.ptr(part), part.numel() * 4, C.byref(n), H.stream()) != 0
