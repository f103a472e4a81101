import os, sys, ctypes as C
import torch
sys.path.insert(0, '.')
from evoke_amd import hip as H
BF = H.STORE_DTYPE
torch.manual_seed(0)
for (M, K, N) in [(36864, 256, 1024), (784, 64, 256), (3136, 128, 512), (4 * 49 * 16, 256, 1024), (196, 512, 2048)]:
    x = (torch.randn(M, K, device='cuda') * 0.7).to(BF)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(BF)
    wt = w.t().contiguous()
    skip = (torch.randn(M, N, device='cuda') * 0.3).to(BF)
    gate = torch.relu(torch.randn(M, N, device='cuda')).to(BF)
    nb = H.lib.evk_conv1x1_ws_part_bytes(M, K, N)
    outs = []
    for rep in range(3):
        y = torch.empty(M, N, device='cuda', dtype=BF); d = torch.empty(M, N, device='cuda', dtype=BF)
        p1 = torch.zeros(nb // 4, device='cuda'); p2 = torch.zeros(nb // 4, device='cuda')
        n1, n2 = C.c_int32(0), C.c_int32(0)
        H.check(H.lib.evk_conv1x1_ws_fwd(H.ptr(x), H.ptr(w), H.ptr(y), M, K, N, H.ptr(p1), nb, C.byref(n1), H.stream()))
        H.check(H.lib.evk_conv1x1_ws_dgrad(H.ptr(x), H.ptr(wt), H.ptr(skip), H.ptr(gate), H.ptr(d), M, K, N, H.ptr(p2), nb, C.byref(n2), H.stream()))
        torch.cuda.synchronize()
        outs.append((y.clone(), d.clone(), p1.clone(), p2.clone()))
    same = [all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:]]
    ref = x.float() @ w.float().t()
    gref = (ref + skip.float()) * (gate.float() > 0)
    bad_y = int((outs[0][0].float() - ref).abs().gt(0.02 + 0.01 * ref.abs()).sum())
    bad_d = int((outs[0][1].float() - gref).abs().gt(0.02 + 0.01 * gref.abs()).sum())
    print('M=%d K=%d N=%d deterministic=%s bad_y=%d bad_d=%d' % (M, K, N, same, bad_y, bad_d), flush=True)
