"""Data parallelism of the path: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm),
gloo on CPU for the tests.  The reference has no distributed path at all (nn.DataParallel only, disabled:
modules/trainer_v0401.py:28-29); what is built here follows SURVEY.md section 8e:

  * studies shard across ranks (all views of a study stay on one rank); no collective on the forward data path
    except ONE exchange step for the contrastive negatives of Pretrain (autograd all-gather of the global image /
    text embeddings + 64-bit study-id hashes, variable rows per rank -> padded);
  * gradient all-reduce (SUM; the 1/world of the mean is applied by the optimizer kernel together with the loss-scale
    division, so the loss is NOT pre-divided and 16-bit activation gradients keep the full loss scale) on the FLAT f32 gradient buffers of
    evoke_amd.optim.FusedOptimizer: contiguous buckets, launched asynchronously from the backward as soon as every
    parameter of a bucket has its gradient (parameters are laid out in forward order, so the decoder / fusion /
    multi-view buckets reduce while the ResNet backward is still running); parameters that get no gradient in a step
    (BERT pooler, the unused fusion branch) simply contribute zeros;
  * BatchNorm statistics are per-rank (reference semantics = per-process batch statistics).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import ops


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('EVK_FORCE_DIST', '0') == '1'      # rehearsal: a 1-rank RCCL group that still runs every collective
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def fnv1a64(s):
    h = 0xcbf29ce484222325
    for ch in str(s).encode('utf-8'):
        h = ((h ^ ch) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h - (1 << 64) if h >= (1 << 63) else h


class _AllGatherRows(torch.autograd.Function):
    """cat over ranks of x (rows_r, D) padded to the max row count.  Every rank then evaluates the SAME global loss on
    the gathered tensor, so d(loss)/d(x_local) is complete from the local slice alone; the backward multiplies it by
    world so that the mean over ranks (all-reduce SUM, then 1/world in the optimizer) leaves the global term un-averaged."""

    @staticmethod
    def forward(ctx, x, counts):
        world, rank = dist.get_world_size(), dist.get_rank()
        mx = max(counts)
        pad = x.new_zeros(mx, *x.shape[1:])
        pad[:x.shape[0]] = x
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad.contiguous())
        ctx.rank, ctx.counts, ctx.world = rank, counts, world
        return torch.cat([o[:c] for o, c in zip(outs, counts)], 0)

    @staticmethod
    def backward(ctx, dy):
        start = sum(ctx.counts[:ctx.rank])
        return dy[start:start + ctx.counts[ctx.rank]] * float(ctx.world), None


def gather_rows(x, patient_ids):
    """(x (rows, D), ids (rows,) str) -> (all ranks' rows, all ranks' ids) ; identity when not distributed."""
    if world_size() == 1:
        return x, np.asarray(patient_ids)
    world = dist.get_world_size()
    n = torch.tensor([x.shape[0]], dtype=torch.long, device=x.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    counts = [int(t.item()) for t in ns]
    mx = max(counts)
    h = torch.zeros(mx, dtype=torch.long, device=x.device)
    h[:x.shape[0]] = torch.tensor([fnv1a64(p) for p in patient_ids], dtype=torch.long, device=x.device)
    hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    ids = np.concatenate([t[:c].cpu().numpy() for t, c in zip(hs, counts)])
    return _AllGatherRows.apply(x, counts), ids


class GradReducer:
    """Bucketed asynchronous all-reduce over flat gradient buffers.

    flat_grads: list of 1-D f32 tensors; params: list (same length) of [(param, offset, numel), ...] describing which
    slice of the flat buffer each parameter's gradient lives in (forward order).

    A parameter may receive several partial gradients in one backward (the relational-memory weights are used once per
    token) and some receive none (statically unused branches), so "bucket complete" is decided from the per-parameter
    callback counts LEARNED during the first step of each step kind (`begin(key)`, e.g. 'inc' / 'no_inc'); the first
    step of a kind reduces everything at `finish()` without overlap."""

    def __init__(self, flat_grads, params, bucket_bytes=128 << 20, overlap=True):
        self.flat, self.overlap = flat_grads, overlap
        self.buckets = []          # (flat_index, start, end)
        self.bucket_of = {}        # id(param) -> bucket index
        elems = max(1, bucket_bytes // 4)
        for fi, (g, plist) in enumerate(zip(flat_grads, params)):
            total = g.numel()
            cur_start, cur_ps = 0, []
            for i, (p, off, n) in enumerate(plist):
                cur_ps.append(p)
                end = plist[i + 1][1] if i + 1 < len(plist) else total
                if end - cur_start >= elems or i + 1 == len(plist):
                    b = len(self.buckets)
                    self.buckets.append((fi, cur_start, end))
                    for q in cur_ps:
                        self.bucket_of[id(q)] = b
                    cur_start, cur_ps = end, []
        self.learned = {}
        self.handles = []
        self.begin('default')
        ops.register_grad_callback(self.on_grad)

    @classmethod
    def for_optimizer(cls, opt, **kw):
        params = []
        for g, st in zip(opt.param_groups, opt.flat):
            params.append([(p, o, p.numel()) for p, o in zip(g['params'], st['offsets'])])
        opt.world = world_size()          # the optimizer kernel divides the summed gradients by the rank count
        return cls(opt.flat_grads(), params, **kw)

    def begin(self, key='default'):
        self.key = key
        self.counts = {}
        self.launched = [False] * len(self.buckets)
        exp = self.learned.get(key)
        self.learning = exp is None
        self.pending = [0] * len(self.buckets)
        if exp is not None:
            for pid, c in exp.items():
                self.pending[self.bucket_of[pid]] += c

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if world_size() == 1 and not (dist.is_initialized() and os.environ.get('EVK_FORCE_DIST', '0') == '1'):
            return
        fi, s, e = self.buckets[b]
        if self.flat[fi].is_cuda:
            # Gradients are accumulated in place from several streams (main, wgrad, rm, text).  The collective is issued from
            # a dedicated stream that waits for all of them, so the main stream is NOT held back at every bucket boundary
            # (RCCL orders itself after the stream that is current at the call); finish() joins the handles.
            cur = torch.cuda.current_stream()
            comm = ops.side_stream('comm')
            comm.wait_stream(cur)
            ops.join_side_streams(into=comm, skip='comm')
            with torch.cuda.stream(comm):
                self.handles.append(dist.all_reduce(self.flat[fi][s:e], op=dist.ReduceOp.SUM, async_op=True))
            return
        self.handles.append(dist.all_reduce(self.flat[fi][s:e], op=dist.ReduceOp.SUM, async_op=True))

    def on_grad(self, p):
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        self.counts[id(p)] = self.counts.get(id(p), 0) + 1
        if not self.learning:
            self.pending[b] -= 1
            if self.overlap and self.pending[b] == 0:
                self._launch(b)

    def finish(self):
        """Call after backward(): reduces what was not launched from the backward, waits for everything."""
        if self.flat and self.flat[0].is_cuda:
            ops.join_side_streams()
        for b in range(len(self.buckets)):
            self._launch(b)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.learning:
            self.learned[self.key] = dict(self.counts)
        self.begin(self.key)
