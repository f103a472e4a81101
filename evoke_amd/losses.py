"""Losses of the path on HIP kernels.

  lm_loss                       modules/loss.py:5-22 (LanguageModelCriterion / compute_lm_loss), log_softmax fused
  multi_pos_contra_images       models/model_pretrain_finetune_v0623_large_res.py:262-282
  global_alignment              ...:311-329
  local_text_token_alignment    ...:331-351

The contrastive similarity products are tiny (<= 128 x 2048) but loss parity wants f32-level accuracy, while the
MFMA GEMM takes bf16 operands: each f32 operand is split into hi + lo bf16 parts and the three significant
cross terms hi.hi + hi.lo + lo.hi are accumulated in ONE bf16 GEMM launch by concatenating them along K
(relative error ~2^-16 instead of 2^-8).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hip as H
from . import ops
from .ops import BF16, F32, _pad8


# ----------------------------------------------------------------------------------------------------
# LM loss
# ----------------------------------------------------------------------------------------------------
def lm_loss(logits, ids, masks, V):
    """logits f32 (B, L, ld); position t predicts ids[:, t+1] weighted by masks[:, t+1]; the last position has weight 0."""
    B, L = ids.shape
    # built with cat (kernels), not with slice assignment: a copy between two row-pitched views is issued as hipMemcpy2DAsync, which a
    # stream capture records as a memcpy node the step replayer cannot re-issue (csrc/replay.hip)
    target = torch.cat([ids[:, 1:], torch.zeros(B, 1, dtype=torch.long, device=logits.device)], 1)
    w = torch.cat([masks[:, 1:].to(F32), torch.zeros(B, 1, dtype=F32, device=logits.device)], 1)
    return ops.nll_loss(logits, target.view(-1), w.view(-1), V)


# ----------------------------------------------------------------------------------------------------
# split-precision small matmuls (f32 in / f32 out, 3-D batched)
# ----------------------------------------------------------------------------------------------------
def _split(x):
    hi = x.to(BF16)
    lo = (x - hi.to(F32)).to(BF16)
    return hi, lo


def _pad_last(x, mult):
    """zero-pad the last dimension to a multiple of `mult`.  Built with cat (a kernel): a slice assignment into the padded buffer is
    a copy between row-pitched views, which torch issues as a memcpy -- a node the step replayer cannot re-issue (ops.pitched_copy)."""
    n = x.shape[-1]
    p = (n + mult - 1) // mult * mult
    if p == n:
        return x.contiguous()
    return torch.cat([x, x.new_zeros(*x.shape[:-1], p - n)], dim=-1)


def _mm_nt(a, b):
    """(G,M,K) x (G,N,K)^T -> (G,M,N)."""
    G, M, K = a.shape
    N = b.shape[1]
    ah, al = _split(_pad_last(a, 8))
    bh, bl = _split(_pad_last(b, 8))
    A = torch.cat([ah, ah, al], dim=2).contiguous()
    Bm = torch.cat([bh, bl, bh], dim=2).contiguous()
    K3 = A.shape[2]
    out = torch.empty(G, M, N, dtype=F32, device=a.device)
    ops.gemm(A, Bm, out, M, N, K3, lda=K3, ldb=K3, ldc=N, batch=(G, 1), sA=(M * K3, 0), sB=(N * K3, 0), sC=(M * N, 0))
    return out


def _mm_nn(a, b):
    """(G,M,S) x (G,S,N) -> (G,M,N)."""
    G, M, S = a.shape
    N = b.shape[2]
    ah, al = _split(_pad_last(a, 8))
    Sp = ah.shape[2]
    bp = _pad_last(b, 8)
    if Sp != S:
        bp = torch.cat([bp, bp.new_zeros(G, Sp - S, bp.shape[2])], dim=1)          # (cat, not slice assignment: see _pad_last)
    bh, bl = _split(bp)
    A = torch.cat([ah, ah, al], dim=2).contiguous()
    Bm = torch.cat([bh, bl, bh], dim=1).contiguous()
    K3, ldb = 3 * Sp, Bm.shape[2]
    out = torch.empty(G, M, N, dtype=F32, device=a.device)
    ops.gemm(A, Bm, out, M, N, K3, b_mode=H.B_KSTR, lda=K3, ldb=ldb, ldc=N, batch=(G, 1), sA=(M * K3, 0), sB=(K3 * ldb, 0),
             sC=(M * N, 0))
    return out


def _mm_tn(a, b):
    """(G,S,M)^T x (G,S,N) -> (G,M,N)."""
    G, S, M = a.shape
    N = b.shape[2]
    ah, al = _split(_pad_last(a, 8))
    bh, bl = _split(_pad_last(b, 8))
    A = torch.cat([ah, ah, al], dim=1).contiguous()
    Bm = torch.cat([bh, bl, bh], dim=1).contiguous()
    lda, ldb = A.shape[2], Bm.shape[2]
    out = torch.empty(G, M, N, dtype=F32, device=a.device)
    ops.gemm(A, Bm, out, M, N, 3 * S, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=lda, ldb=ldb, ldc=N, batch=(G, 1),
             sA=(3 * S * lda, 0), sB=(3 * S * ldb, 0), sC=(M * N, 0))
    return out


class _MatmulNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _mm_nt(a, b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous()
        da = _mm_nn(dc, b) if ctx.needs_input_grad[0] else None
        db = _mm_tn(dc, a) if ctx.needs_input_grad[1] else None
        return da, db


class _MatmulNN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return _mm_nn(a, b)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        dc = dc.contiguous()
        da = _mm_nt(dc, b) if ctx.needs_input_grad[0] else None
        db = _mm_tn(a, dc) if ctx.needs_input_grad[1] else None
        return da, db


def matmul_nt(a, b):
    """a (.., M, K) @ b (.., N, K)^T in ~f32 accuracy on the bf16 MFMA kernel."""
    two = a.dim() == 2
    out = _MatmulNT.apply(a.unsqueeze(0) if two else a, b.unsqueeze(0) if two else b)
    return out.squeeze(0) if two else out          # (squeeze is a view both ways; out[0] back-propagates through a zero-fill + memcpy)


def matmul_nn(a, b):
    two = a.dim() == 2
    out = _MatmulNN.apply(a.unsqueeze(0) if two else a, b.unsqueeze(0) if two else b)
    return out.squeeze(0) if two else out


# ----------------------------------------------------------------------------------------------------
# row kernels with autograd
# ----------------------------------------------------------------------------------------------------
class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        nrm = torch.empty(rows, dtype=F32, device=x.device)
        H.check(H.lib.evk_l2norm_fwd(H.ptr(x), H.ptr(y), H.ptr(nrm), rows, D, H.stream()), 'l2norm_fwd')
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        D = y.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        H.check(H.lib.evk_l2norm_bwd(H.ptr(dy), H.ptr(y), H.ptr(nrm), H.ptr(dx), y.numel() // D, D, H.stream()), 'l2norm_bwd')
        return dx


def l2_normalize(x):
    """F.normalize(x, dim=-1, p=2) on f32 rows."""
    assert x.dtype == F32
    return _L2Norm.apply(x.contiguous())


class _SoftCE(torch.autograd.Function):
    """mean over rows of -sum_c t[c] log_softmax(z)[c]  (F.cross_entropy with probability targets)."""

    @staticmethod
    def forward(ctx, z, t, diag_mask):
        rows, Cn = z.shape
        acc = torch.zeros(1, dtype=F32, device=z.device)
        H.check(H.lib.evk_softce(H.ptr(z), H.ptr(t), H.ptr(acc), None, None, rows, Cn, C.c_float(1.0 / rows), int(diag_mask),
                                 H.stream()), 'softce')
        ctx.save_for_backward(z, t)
        ctx.diag = diag_mask
        return acc[0]

    @staticmethod
    def backward(ctx, dl):
        z, t = ctx.saved_tensors
        rows, Cn = z.shape
        dz = torch.empty_like(z)
        gs = dl.to(F32).reshape(1).contiguous()
        H.check(H.lib.evk_softce(H.ptr(z), H.ptr(t), None, H.ptr(dz), H.ptr(gs), rows, Cn, C.c_float(1.0 / rows), int(ctx.diag),
                                 H.stream()), 'softce')
        return dz, None, None


def soft_cross_entropy(z, t, diag_mask=False):
    assert z.dtype == F32 and t.dtype == F32 and z.shape == t.shape and z.dim() == 2
    return _SoftCE.apply(z.contiguous(), t.contiguous(), diag_mask)


class _Softmax(torch.autograd.Function):
    """f32 softmax over the last dim of (G, T, S) scores (token -> patch attention of the local loss)."""

    @staticmethod
    def forward(ctx, s):
        G, T, S = s.shape
        p = torch.empty_like(s)
        H.check(H.lib.evk_softmax_fwd(H.ptr(s), H.ptr(p), None, H.F32, None, 0, 0, 0, G, 1, T, S, S, S, C.c_float(0.0), 0,
                                      H.stream()), 'softmax_fwd')
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        p, = ctx.saved_tensors
        G, T, S = p.shape
        dp = dp.contiguous()
        ds = torch.empty_like(p)
        H.check(H.lib.evk_softmax_bwd(H.ptr(dp), H.F32, S, H.ptr(p), H.ptr(ds), H.F32, G * T, S, S, C.c_float(1.0), C.c_float(0.0), 0,
                                      H.stream()), 'softmax_bwd')
        return ds


# ----------------------------------------------------------------------------------------------------
# the three contrastive losses
# ----------------------------------------------------------------------------------------------------
def _same_study(pid):
    pid = np.asarray(pid)
    return (pid.reshape(-1, 1) == pid.reshape(1, -1)).astype(np.float32)


def multi_pos_contra_images(global_image_embed, patient_ids, temp, gather=None):
    """Image-image multi-positive InfoNCE over rows that have a same-study sibling; with `gather` (cross-rank
    all-gather with autograd) the negatives / positives span all ranks."""
    g = global_image_embed.to(F32)
    pid = np.asarray(patient_ids)
    if gather is not None:
        g, pid = gather(g, pid)
    labels = _same_study(pid)
    np.fill_diagonal(labels, 0.0)
    idx = np.nonzero(labels.sum(1) != 0)[0]
    if len(idx) == 0:
        return torch.zeros(1, device=global_image_embed.device, requires_grad=True)      # no host->device copy (graph-capturable)
    labels = labels[idx][:, idx]
    labels = labels / labels.sum(1, keepdims=True)
    t = ops.upload(labels, g.device)
    gi = l2_normalize(g.index_select(0, ops.index_tensor(idx, g.device)))
    logits = matmul_nt(gi, gi) / temp
    # the reference subtracts the detached row max after the -1e9 diagonal fill: a no-op for log-softmax
    return soft_cross_entropy(logits, t, diag_mask=True)


def global_alignment(global_image_embed, global_text_embed, patient_ids, temp, gather=None):
    v, t = global_image_embed.to(F32), global_text_embed.to(F32)
    pid = np.asarray(patient_ids)[:v.shape[0]]
    if gather is not None:
        # image and text globals of the SAME studies: one exchange of [v | t] rows (one id exchange + one row all-gather instead of two each)
        d = v.shape[1]
        vt, pid = gather(torch.cat([v, t], 1), pid)
        v, t = vt[:, :d].contiguous(), vt[:, d:].contiguous()
    labels = _same_study(pid)
    labels = labels / labels.sum(1, keepdims=True)
    tg = ops.upload(labels, v.device)
    v, t = l2_normalize(v), l2_normalize(t)
    sim = matmul_nt(v, t) / temp
    sim_t = matmul_nt(t, v) / temp
    return (soft_cross_entropy(sim, tg) + soft_cross_entropy(sim_t, tg)) / 2.0


def local_text_token_alignment(local_image_embed, local_text_embed, temp):
    p, t = local_image_embed.to(F32).contiguous(), local_text_embed.to(F32).contiguous()
    b, n1, d = t.shape
    sco = _Softmax.apply((matmul_nt(t, p) / math.sqrt(d)).contiguous())
    att = l2_normalize(matmul_nn(sco, p))
    tn = l2_normalize(t)
    ws = matmul_nt(tn, att) / temp                        # (b, n1, n1)
    eye = torch.eye(n1, dtype=F32, device=t.device).repeat(b, 1)
    l1 = soft_cross_entropy(ws.reshape(b * n1, n1), eye)
    l2 = soft_cross_entropy(ws.transpose(1, 2).reshape(b * n1, n1), eye)
    return (l1 + l2) / 2.0
