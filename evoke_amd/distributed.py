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


def forced():
    """EVK_FORCE_DIST=1 with a live process group: the rehearsal mode -- every collective of the data-parallel path (bucketed gradient
    sum, update-mask union, the contrastive exchange) is ISSUED even on one rank, where each of them is the identity."""
    return dist.is_available() and dist.is_initialized() and os.environ.get('EVK_FORCE_DIST', '0') == '1'


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


GATHER_CAP = int(os.environ.get('EVK_GATHER_CAP', '1024'))       # most rows (images / studies) ONE rank may contribute to an exchange


def gather_rows(x, patient_ids):
    """(x (rows, D), ids (rows,) str) -> (all ranks' rows, all ranks' ids) ; identity when not distributed.
    Two collectives per call: ONE all-gather of a fixed-capacity int64 record per rank -- [row count, 64-bit study-id hashes ...] -- read
    back with one host copy (the row counts are shapes: the host must know them), then the autograd all-gather of the rows padded to
    the largest count."""
    if world_size() == 1 and not forced():
        return x, np.asarray(patient_ids)
    world = dist.get_world_size()
    n = x.shape[0]
    if n > GATHER_CAP:
        raise ValueError('gather_rows: %d rows on this rank exceed EVK_GATHER_CAP = %d (every rank must use the same capacity)' % (n, GATHER_CAP))
    rec = np.zeros(1 + GATHER_CAP, dtype=np.int64)
    rec[0] = n
    rec[1:1 + n] = [fnv1a64(p) for p in patient_ids]
    if x.is_cuda:
        # The record depends on host data only (the batch's study ids), the read-back must not wait for the GPU work already queued on the
        # compute stream (the whole forward up to the loss: a host that stalls there twice per step turns a host-bound step into a
        # serialised one -- 23.6 -> 28.0 ms on the 224^2 Pretrain step with a 1-rank group).  So the id exchange runs on the 'comm' stream,
        # which is idle during the forward; every rank issues it at the same point of its program, so the collective order still agrees.
        comm = ops.side_stream('comm')
        with torch.cuda.stream(comm):
            mine = ops.upload(rec, x.device)
            every = torch.empty(world * (1 + GATHER_CAP), dtype=torch.long, device=x.device)
            dist.all_gather_into_tensor(every, mine.contiguous())
            every = every.cpu()          # (blocking copy on 'comm': waits for the all-gather only)
    else:
        mine = torch.from_numpy(rec)
        every = torch.empty(world * (1 + GATHER_CAP), dtype=torch.long)
        dist.all_gather_into_tensor(every, mine.contiguous())
    every = every.numpy().reshape(world, 1 + GATHER_CAP)
    counts = [int(c) for c in every[:, 0]]
    ids = np.concatenate([every[r, 1:1 + c] for r, c in enumerate(counts)])
    return _AllGatherRows.apply(x, counts), ids


def batch_structure(kind, patient_ids):
    """Key of everything that decides WHICH parameters receive gradients in a step (the reducer learns per-parameter gradient
    counts per key): the step kind (indication / no indication / pretrain) and whether any anchor has a sibling view -- a batch
    without siblings never runs the multi-view attention and layer_norm_2 (models/...v0623_large_res.py:137-139)."""
    pid = np.asarray(patient_ids)
    return (kind, len(np.unique(pid)) != len(pid))


class GradReducer:
    """Bucketed asynchronous gradient sum over flat gradient buffers, launched from the backward pass.

    flat_grads: list of 1-D f32 tensors; params: list (same length) of [(param, offset, numel), ...] describing which
    slice of the flat buffer each parameter's gradient lives in (forward order).

    ORDER: every rank issues exactly one collective sequence per bucket per step, in DESCENDING bucket index (the order the
    backward completes them in, as torch's DDP does): bucket b is launched only after bucket b+1, however early its own gradients
    arrive.  Ranks therefore never disagree on the order of collectives, whatever their batches look like (a rank whose batch has
    no sibling views skips the multi-view attention; that bucket then simply goes out at finish()).
    COMPLETENESS: a parameter may receive several partial gradients in one backward and some receive none (statically unused
    branches), so "bucket complete" is decided from per-parameter callback counts LEARNED during the first step of each batch
    structure (`begin(key)`; batch_structure()); the first step of a structure reduces everything at finish() without overlap.  A
    gradient that arrives for a bucket that has already gone out is an error, not a silent corruption.
    MODE (EVK_GRAD_SYNC): 'allreduce' = one all_reduce (SUM) per bucket on f32; 'direct' = reduce-scatter + all-gather written as
    all_to_all_single (every rank sends chunk j straight to rank j: on MI355X's xGMI mesh that is one transfer per point-to-point
    link, all seven links at once, instead of a ring that is bound by one link), a local sum of the received chunks, and
    all_gather_into_tensor; '16bit' = the bucket is cast to the library's 16-bit storage format, all-reduced in that format (half
    the bytes on the links) and cast back -- an overflow of the scaled fp16 gradients becomes inf, which the optimizer's overflow
    scan turns into a skipped step on every rank."""

    active = False          # set by begin(): more than one rank takes part in this step's gradient sum

    def __init__(self, flat_grads, params, bucket_bytes=128 << 20, overlap=True, mode=None):
        self.flat, self.overlap = flat_grads, overlap
        # default 'allreduce': RCCL's own all-reduce (it picks its algorithm for the xGMI mesh).  'direct' -- reduce-scatter + all-gather written
        # as one all_to_all_single (every rank sends chunk j straight to rank j: one transfer per point-to-point link, all seven at once) + a
        # local sum + all_gather_into_tensor, 2.3 ms against ~16 ms for a ring on the 1.39 GB of FineTune gradients by SURVEY.md section 5's
        # arithmetic -- is verified with gloo (two ranks, three modes) but has never met RCCL with more than one rank (no multi-GPU node was
        # available to the builder), so it stays opt-in: EVK_GRAD_SYNC=direct on a node is the first thing to measure there.
        self.mode = mode or os.environ.get('EVK_GRAD_SYNC') or 'allreduce'
        if self.mode not in ('allreduce', 'direct', '16bit'):
            raise ValueError('EVK_GRAD_SYNC must be allreduce, direct or 16bit')
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
        self.opt = None            # weakref to the FusedOptimizer whose flat buffers these are (for_optimizer): receives the union below
        self.touched_union = None  # uint8 per parameter (group order): 1 = SOME rank produced a gradient for it this step
        self.issued = []           # bucket indices in the order their collectives were issued this step (tests read it)
        self.last_issued, self.last_early = [], 0          # the same of the step finish() completed last, and how many went out before finish()
        # timing=True (bench.py): HIP events on the 'comm' stream around every bucket's collective (from "its producers are done" to "its
        # result is there") and on the main stream around finish()'s wait for them; read with comm_stats() after a device synchronisation
        self.timing = False
        self._ev_comm, self._ev_exposed = [], []
        self.begin('default')
        ops.register_grad_callback(self.on_grad)

    @classmethod
    def for_optimizer(cls, opt, **kw):
        params = []
        for g, st in zip(opt.param_groups, opt.flat):
            params.append([(p, o, p.numel()) for p, o in zip(g['params'], st['offsets'])])
        opt.world = world_size()          # the optimizer kernel divides the summed gradients by the rank count
        red = cls(opt.flat_grads(), params, **kw)
        import weakref
        red.opt = weakref.ref(opt)
        return red

    def begin(self, key='default'):
        self.key = key
        self.counts = {}
        n = len(self.buckets)
        self.launched = [False] * n
        self.active = self._active()                  # (constant over a step: asked once, not per gradient)
        self.streams = [set() for _ in range(n)]      # HIP streams that produced gradients of the bucket
        self.next = n - 1                             # the only bucket that may go out now
        self.issued = []
        exp = self.learned.get(key)
        self.learning = exp is None
        self.pending = [0] * n
        if exp is not None:
            for pid, c in exp.items():
                self.pending[self.bucket_of[pid]] += c

    def _active(self):
        return world_size() > 1 or forced()

    def _collective(self, buf):
        """the gradient sum of one bucket in the configured mode; returns async handles (possibly none: the blocking steps of a
        mode run on the communication stream and are ordered by it)"""
        world = dist.get_world_size()
        if self.mode == 'direct' and (world > 1 or forced()) and buf.numel() % world == 0 and buf.numel() >= 1024 * world:
            recv = torch.empty_like(buf)
            dist.all_to_all_single(recv, buf)                                   # chunk j of every rank -> rank j
            shard = recv.view(world, -1).sum(0)                                 # this rank's slice of the sum
            return [dist.all_gather_into_tensor(buf, shard, async_op=True)]
        if self.mode == '16bit' and buf.numel() > 0:
            if buf.is_cuda:
                from . import hip as H
                low = torch.empty(buf.numel(), dtype=ops.BF16, device=buf.device)
                H.check(H.lib.evk_cast(H.ptr(buf), H.F32, H.ptr(low), H.BF16, buf.numel(), H.stream()), 'cast')
                dist.all_reduce(low, op=dist.ReduceOp.SUM)
                H.check(H.lib.evk_cast(H.ptr(low), H.BF16, H.ptr(buf), H.F32, buf.numel(), H.stream()), 'cast')
            else:
                low = buf.to(ops.BF16)
                dist.all_reduce(low, op=dist.ReduceOp.SUM)
                buf.copy_(low)
            return []
        return [dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)]

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        self.issued.append(b)
        if not self._active():
            return
        fi, s, e = self.buckets[b]
        if self.flat[fi].is_cuda:
            # Gradients are accumulated in place from several streams (main, wgrad, rm, text).  The collective is issued from a
            # dedicated stream that waits for the streams THIS bucket's gradients came from (not for every side stream: the
            # weight-gradient stream of an early bucket has nothing to do with the relational memory's), so the main stream is
            # not held back at a bucket boundary (RCCL orders itself after the stream that is current at the call).
            cur = ops.H.current_stream()
            comm = ops.side_stream('comm')
            comm.wait_stream(cur)
            for st in self.streams[b]:
                if st != comm and st != cur:
                    comm.wait_stream(st)
            with torch.cuda.stream(comm):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(comm)
                    for h in self._collective(self.flat[fi][s:e]):
                        h.wait()          # (stream-level: 'comm' waits for the collective's own stream; the host does not block)
                    e1.record(comm)
                    self._ev_comm.append((e0, e1, 4 * (e - s)))
                else:
                    self.handles += self._collective(self.flat[fi][s:e])
            return
        self.handles += self._collective(self.flat[fi][s:e])

    def _drain(self):
        """launch, in descending order, every bucket that is complete and whose successors have all gone out"""
        while self.next >= 0 and (self.launched[self.next] or self.pending[self.next] == 0):
            self._launch(self.next)
            self.next -= 1

    def on_grad(self, p):
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        if self.launched[b] and self.active:
            raise RuntimeError('GradReducer: a gradient arrived for bucket %d after its collective was issued (structure key %r: '
                               'the learned gradient counts do not describe this batch)' % (b, self.key))
        self.counts[id(p)] = self.counts.get(id(p), 0) + 1
        if self.active and self.flat[self.buckets[b][0]].is_cuda:          # (one rank: no collective will wait for anything)
            self.streams[b].add(ops.H.current_stream())
            # Weight gradients are launched on the 'wgrad' side stream (ops.wgrad_stream, the trunk runner's second stream) while the
            # callback runs on the stream that issued the backward op: the collective of this bucket must wait for that stream too.
            # (Without it the all-reduce could start before the last weight-gradient kernel had added its part, which then landed on
            # top of the reduced values of THIS rank only: parameters drifting apart across ranks, seen once in the two-rank test.)
            wg = ops.existing_side_stream('wgrad')
            if wg is not None:
                self.streams[b].add(wg)
        if not self.learning:
            self.pending[b] -= 1
            if self.overlap and self.pending[b] == 0 and b == self.next:
                self._drain()

    def _share_touched(self):
        """Which parameters received a gradient this step is a property of the rank's BATCH (a shard without sibling views never runs the
        multi-view attention and layer_norm_2; indication / no-indication shards use different fusion branches).  After the sum every rank
        holds the other ranks' gradients for those parameters too, so the update mask must be the UNION over ranks: a rank that masked
        them out locally would skip an update (and a step count) the others apply, and the replicas would part for good.  One MAX
        all-reduce of a uint8 per parameter, issued after the last bucket on every rank; the optimizer reads the result on the device."""
        opt = self.opt() if self.opt is not None else None
        if opt is None or not self._active():
            self.touched_union = None
            return
        bits = np.frombuffer(bytes(1 if id(p) in opt._touched else 0 for g in opt.param_groups for p in g['params']), dtype=np.uint8).copy()
        dev = self.flat[0].device
        t = ops.upload(bits, dev).clone() if dev.type == 'cuda' else torch.from_numpy(bits)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.touched_union = t
        opt.set_global_touched(t)

    def comm_stats(self, steps):
        """(timing=True) per-step figures of the last `steps` steps; synchronise the device first.  comm_ms = sum over buckets of the time
        between "the bucket's producers are done" and "its collective has finished" on the communication stream (collectives of different
        buckets serialise on RCCL's stream, so the sum is the stream's busy time); exposed_ms = how long the main stream stood still in
        finish() waiting for the last collectives and the update-mask exchange -- the part of comm_ms the backward did not hide."""
        n = max(1, steps)
        comm = sum(a.elapsed_time(b) for a, b, _ in self._ev_comm) / n
        exposed = sum(a.elapsed_time(b) for a, b in self._ev_exposed) / n
        out = dict(comm_ms_per_step=comm, exposed_comm_ms_per_step=exposed, collectives_per_step=len(self._ev_comm) / n,
                   buckets=len(self.buckets), bucket_bytes=[4 * (e - s) for _, s, e in self.buckets],
                   reduced_bytes_per_step=sum(nb for _, _, nb in self._ev_comm) / n)
        self._ev_comm, self._ev_exposed = [], []
        return out

    def finish(self):
        """Call after backward(): reduces what was not launched from the backward (descending order), waits for everything."""
        early = len(self.issued)        # collectives that went out from inside the backward
        if self.flat and self.flat[0].is_cuda:
            ops.join_side_streams()
            for st in self.streams:
                st.clear()              # everything is joined into the current stream now
        for b in range(len(self.buckets) - 1, -1, -1):
            self._launch(b)
        self.next = -1
        for h in self.handles:
            h.wait()
        self.handles = []
        on_gpu = bool(self.flat) and self.flat[0].is_cuda and self._active()
        if on_gpu and self.timing:
            m0 = torch.cuda.Event(enable_timing=True)
            m0.record()
        if on_gpu:
            ops.H.current_stream().wait_stream(ops.side_stream('comm'))
        self._share_touched()
        if on_gpu and self.timing:
            m1 = torch.cuda.Event(enable_timing=True)
            m1.record()
            self._ev_exposed.append((m0, m1))
        if self.learning:
            self.learned[self.key] = dict(self.counts)
        self.last_issued, self.last_early = list(self.issued), early          # (begin() resets the step's bookkeeping)
        self.begin(self.key)
