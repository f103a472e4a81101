"""Host-side operators of the engine: torch.autograd.Function wrappers whose forward AND backward enqueue
hand-written HIP kernels through the C ABI (evoke_amd/hip.py -> libevoke_hip.so).  PyTorch is plumbing here:
device memory, streams, the autograd tape.  No op in this file computes on the CPU or through torch math
kernels for the heavy work; tiny O(C) vector updates (e.g. adding a [C] partial into a .grad) use torch.

Conventions: activations are contiguous bf16 CUDA tensors [rows.., features]; parameters are f32 masters with a
cached bf16 "shadow" that the GEMMs read; parameter gradients are accumulated in f32 directly into ``p.grad``
(the backward returns None for them and fires ``grad_ready`` callbacks, which the DDP reducer listens to).
"""
import contextlib
import os
from .config import tunable
import ctypes as C
import math

import torch

from . import hip as H

BF16, F32 = H.STORE_DTYPE, torch.float32      # BF16 = the 16-bit storage dtype of the loaded library (bf16, or fp16 under EVK_STORE=f16)

# Dynamic loss scale of the fp16-storage build (the default; EVK_STORE=bf16 has fp32's exponent range and runs with scale 1):
# FineTune / Pretrain return `all_loss` through scale_loss(), whose backward multiplies the incoming gradient by the CURRENT
# scale -- a device scalar, never read back by the host -- so every 16-bit activation gradient of the step stays in fp16's
# normal range.  The f32 parameter gradients carry the same factor (`p.grad` holds SCALED gradients); FusedOptimizer.step()
# looks for non-finite gradients after the all-reduce, skips the WHOLE step when it finds one, divides the scale out before
# clipping otherwise, and backs the scale off / grows it (torch.cuda.amp.GradScaler's rule: x0.5 on overflow, x2 after
# `interval` clean steps), all in device code (csrc/eltwise.hip).  The value of the loss is unchanged.
LOSS_SCALE_INIT = float(os.environ.get('EVK_LOSS_SCALE', '1024' if H.STORE == 'f16' else '1'))
LOSS_SCALE_DYNAMIC = H.STORE == 'f16' and os.environ.get('EVK_LOSS_SCALE_STATIC', '0') != '1'


class LossScaler:
    """state (device f32[4]): [0] scale, [1] consecutive good steps, [2] overflow seen this step, [3] skipped steps."""
    GROWTH, BACKOFF, INTERVAL, MIN, MAX = 2.0, 0.5, int(os.environ.get('EVK_LOSS_SCALE_INTERVAL', '2000')), 1.0, 2.0 ** 24

    def __init__(self, device):
        self.state = torch.tensor([LOSS_SCALE_INIT, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)

    def check(self, flat_grad):
        H.check(H.lib.evk_grad_nonfinite(H.ptr(flat_grad), flat_grad.numel(), H.ptr(self.state), H.stream()), 'grad_nonfinite')

    def update(self):
        if LOSS_SCALE_DYNAMIC:
            H.check(H.lib.evk_loss_scale_update(H.ptr(self.state), self.GROWTH, self.BACKOFF, self.INTERVAL, self.MIN, self.MAX, H.stream()),
                    'loss_scale_update')
        else:
            self.state[2:3].zero_()

    def value(self):
        """host read-back (synchronises): tests / checkpoints only"""
        return float(self.state[0].item())

    def skipped(self):
        return int(self.state[3].item())


_scalers = {}


def loss_scaler(device=None):
    """The per-device loss scaler, or None when the build needs none (bf16 storage with scale 1)."""
    if LOSS_SCALE_INIT == 1.0 and not LOSS_SCALE_DYNAMIC:
        return None
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != 'cuda':
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    sc = _scalers.get(key)
    if sc is None:
        sc = _scalers[key] = LossScaler(torch.device('cuda', key))
    return sc


def loss_scale_value(device=None):
    """Current loss scale as a float (host sync; 1.0 when the build has no scaler)."""
    sc = loss_scaler(device)
    return sc.value() if sc is not None else 1.0


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, owner):
        ctx.save_for_backward(scale)
        ctx.owner = owner
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        scale, = ctx.saved_tensors
        owner = ctx.owner() if ctx.owner is not None else None
        if owner is not None:
            _unowned_backward_begin(owner, g.device)
        return g * scale, None, None


def scale_loss(loss, owner=None):
    """`loss` with a backward that multiplies the gradient seed by the current loss scale.  owner = the model (nn.Module) the loss
    belongs to: parameters of it that no FusedOptimizer has re-homed get their gradients UNSCALED when the backward pass ends, so the
    reference's own step (trainer_v0401.py:432-435: backward -> clip_grad_value_ -> torch.optim step) sees true gradients."""
    if not (torch.is_tensor(loss) and loss.requires_grad and loss.is_cuda):
        return loss
    sc = loss_scaler(loss.device)
    if sc is None:
        return loss
    import weakref
    return _ScaleGrad.apply(loss, sc.state[0], weakref.ref(owner) if owner is not None else None)


def grads_owned(p):
    """True when a live FusedOptimizer holds p's gradient in its flat buffers (it divides the loss scale out itself)."""
    ref = getattr(p, '_evk_optimizer', None)
    return ref is not None and ref() is not None


_chunk_tables = {}
GRAD_CHUNK = 1 << 16


def _chunk_table(grads):
    """device table of {pointer, element count} pieces for evk_grads_multi, cached on the buffers' addresses"""
    import numpy as np
    key = tuple((g.data_ptr(), g.numel()) for g in grads)
    ent = _chunk_tables.get(key)
    if ent is None:
        rows = []
        for ptr_, n in key:
            for o in range(0, n, GRAD_CHUNK):
                rows.append((ptr_ + 4 * o, min(GRAD_CHUNK, n - o)))
        tab = np.zeros(len(rows), dtype=np.dtype([('p', np.uint64), ('n', np.int32), ('pad', np.int32)]))
        tab['p'] = [r[0] for r in rows]
        tab['n'] = [r[1] for r in rows]
        dev = torch.from_numpy(tab.view(np.uint8).copy()).to(grads[0].device)
        if len(_chunk_tables) >= 8:
            _chunk_tables.pop(next(iter(_chunk_tables)))
        ent = _chunk_tables[key] = (dev, len(rows))
    return ent


def _grads_multi(grads, mode, scaler):
    for g in grads:
        if g.dtype != torch.float32 or not g.is_cuda or not (g.is_contiguous() or g.is_contiguous(memory_format=torch.channels_last)):
            raise RuntimeError('evoke_amd: parameter gradients are dense f32 device tensors; got %s %s' % (g.dtype, tuple(g.stride())))
    tab, n = _chunk_table(grads)
    H.check(H.lib.evk_grads_multi(H.ptr(tab), n, mode, H.ptr(scaler.state), H.stream()), 'grads_multi')


def _unowned_backward_begin(owner, device):
    """Called when the backward pass of a scaled loss starts (from the root node).  Parameters whose gradients no FusedOptimizer owns:
    gradients already present (accumulation over several backward calls) go back under the scale, and an engine callback unscales
    everything -- after an inf / NaN scan whose verdict zeroes all of them and backs the scale off -- when the pass has finished."""
    params = [p for p in owner.parameters() if p.requires_grad and not grads_owned(p)]
    if not params:
        return
    scaler = loss_scaler(device)
    old = [p.grad for p in params if p.grad is not None and p.grad.is_cuda]
    if old:
        _grads_multi(old, 2, scaler)

    # Mixed ownership: when a live FusedOptimizer holds OTHER parameters of this model, the step's overflow verdict and the scale
    # update belong to ITS step() -- it scans its flat buffers, divides by state[0] and then updates.  Updating here as well would clear
    # the overflow flag (or move the scale) before that division: owned gradients unscaled by the wrong factor, an overflow unseen,
    # update() twice per step.  So this callback only unscales / zeroes and leaves state[1:] alone in that case.
    mixed = any(grads_owned(p) for p in owner.parameters() if p.requires_grad)

    def finish():
        join_side_streams()
        grads = [p.grad for p in params if p.grad is not None and p.grad.is_cuda]
        if grads:
            _grads_multi(grads, 0, scaler)
            _grads_multi(grads, 1, scaler)
        if not mixed:
            scaler.update()
    torch.autograd.Variable._execution_engine.queue_callback(finish)


def overflow_steps(device=None):
    """How many backward passes / optimizer steps so far hit a non-finite gradient (host sync; 0 without a scaler).  Semantics for
    gradients NO FusedOptimizer owns (the reference's own step: backward -> clip_grad_value_ -> torch.optim step): on overflow they
    are ZEROED and the scale is halved -- the torch optimizer that follows still runs, i.e. it applies weight decay and momentum with
    a zero gradient and advances its step count (torch.cuda.amp.GradScaler would skip optimizer.step() instead).  A caller that wants the
    GradScaler behaviour compares this counter before and after backward() and skips its optimizer.step() when it moved."""
    sc = loss_scaler(device)
    return sc.skipped() if sc is not None else 0


# ----------------------------------------------------------------------------------------------------
# forward-overflow guard (fp16 storage)
# ----------------------------------------------------------------------------------------------------
FWD_GUARD = [H.STORE == 'f16' and os.environ.get('EVK_FWD_GUARD', '1') != '0']
_guards = {}          # device index -> {'slots': int32[K] device counters, 'host': pinned int32[K], 'next': i, 'pending': [(slot, event, what)]}
_GUARD_SLOTS = 16


class ForwardOverflow(RuntimeError):
    pass


def _guard_state(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    g = _guards.get(key)
    if g is None:
        g = _guards[key] = dict(slots=torch.zeros(_GUARD_SLOTS, dtype=torch.int32, device=dev),
                                host=torch.zeros(_GUARD_SLOTS, dtype=torch.int32).pin_memory(), next=0, pending=[])
    return g


def guard_finite(t, what):
    """fp16 storage only: scan the 16-bit activation tensor `t` (the trunk output) for inf / NaN WITHOUT stalling the pipeline -- one scan
    launch, one 4-byte copy into pinned memory and an event; the verdict is read by check_forward_guard() once the event has passed (at
    the next guarded forward, at every optimizer step, and -- blocking -- wherever the host synchronises anyway: generation results,
    trainer read-backs).  A forward activation beyond fp16's +-65504 cannot be repaired by the loss scale (that guards the backward)."""
    if not FWD_GUARD[0] or not t.is_cuda or t.dtype != BF16 or torch.cuda.is_current_stream_capturing():
        return
    check_forward_guard(t.device)
    g = _guard_state(t.device)
    if len(g['pending']) >= _GUARD_SLOTS - 1:
        check_forward_guard(t.device, block=True)
    i = g['next']
    g['next'] = (i + 1) % _GUARD_SLOTS
    cnt = g['slots'][i:i + 1]
    cnt.zero_()
    x = t if t.is_contiguous() else t.contiguous()
    H.check(H.lib.evk_act_nonfinite(H.ptr(x), x.numel(), H.ptr(cnt), H.stream()), 'act_nonfinite')
    g['host'][i:i + 1].copy_(cnt, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    g['pending'].append((i, ev, what))


def check_forward_guard(device=None, block=False):
    """Raise ForwardOverflow if a guarded forward tensor held inf / NaN.  block=False only looks at scans the GPU has already finished."""
    if not FWD_GUARD[0] or not _guards:
        return
    keys = list(_guards) if device is None else [torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()]
    for key in keys:
        g = _guards.get(key)
        if g is None:
            continue
        keep, bad = [], None
        for i, ev, what in g['pending']:
            if block:
                ev.synchronize()
            if block or ev.query():
                if int(g['host'][i]) != 0 and bad is None:
                    bad = what
            else:
                keep.append((i, ev, what))
        g['pending'] = keep
        if bad is not None:
            raise ForwardOverflow(
                'evoke_amd: %s holds inf / NaN in the FORWARD pass of the fp16-storage build: an activation left fp16\'s range (+-65504).  The '
                'dynamic loss scale protects the backward only.  Typical cause: batch-norm running statistics that were never trained (a freshly '
                'initialised network in eval mode) or weights far from a trained checkpoint.  Run with EVK_STORE=bf16 (libevoke_hip_bf16.so: fp32\'s '
                'exponent range, no loss scaling) or load trained weights.' % bad)


# ----------------------------------------------------------------------------------------------------
# parameter shadows + direct gradient accumulation
# ----------------------------------------------------------------------------------------------------
_shadows = {}
_grad_callbacks = []


def register_grad_callback(fn):
    _grad_callbacks.append(fn)


def clear_grad_callbacks():
    del _grad_callbacks[:]


def _pad8(n):
    return (n + 7) // 8 * 8


def shadow(p, pad_rows=False):
    """bf16 copy of an f32 master weight (same physical layout), refreshed when the master changed."""
    key = id(p)
    ent = _shadows.get(key)
    ver = p._version
    if ent is not None and ent[1] == ver and ent[2] is p and ent[0].device == p.device:
        return ent[0]
    n = p.numel()
    if ent is not None and ent[2] is p and ent[0].device == p.device:
        sh = ent[0]
    else:
        rows = p.shape[0]
        tot = n if not pad_rows else _pad8(rows) * (n // rows)
        sh = torch.zeros(tot, dtype=BF16, device=p.device)
    H.check(H.lib.evk_cast(H.ptr(p), H.F32, H.ptr(sh), H.BF16, n, H.stream()), 'cast')
    _shadows[key] = (sh, ver, p)
    return sh


WEIGHT_EPOCH = [0]          # bumped by FusedOptimizer.step(): the optimizer kernel rewrites parameters through raw pointers, which does not advance
                            # torch's per-tensor version counters -- inference-time caches of derived weights key on (this, the versions)


def pitched_copy(x):
    """Contiguous copy of a row-pitched view (a slice along an inner dimension).  torch issues `.contiguous()` of such a view as
    hipMemcpy2DAsync; inside a stream capture that becomes a memcpy node whose parameters ROCm 7.2 does not report back, so the step
    replayer (csrc/replay.hip) refuses the plan.  An elementwise kernel does the same copy and keeps the step replayable; autograd sees
    a multiplication by one."""
    return x if x.is_contiguous() else torch.mul(x, 1)


class _SplitFirst(torch.autograd.Function):
    """(x[:, 0, :], x[:, 1:, :]) as two contiguous tensors.  The plain slices are views whose backward writes each gradient into a
    zero tensor through a slice copy (memcpy nodes in a captured step, see pitched_copy); here the forward is two elementwise copies
    and the backward ONE concatenation."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return torch.mul(x[:, 0, :], 1), torch.mul(x[:, 1:, :], 1)

    @staticmethod
    def backward(ctx, ga, gb):
        B, T, D = ctx.shape
        if ga is None:
            ga = gb.new_zeros(B, D)
        if gb is None:
            gb = ga.new_zeros(B, T - 1, D)
        return torch.cat([ga.unsqueeze(1), gb], 1)


def split_first_token(x):
    """x (B, T, D) -> global token (B, D), local tokens (B, T-1, D) (...v0623_large_res.py:260, 387-388)"""
    return _SplitFirst.apply(x)


def copy_kernel(dst, src):
    """dst[...] = src for two contiguous device tensors of one dtype and size, as a KERNEL (a word copy through evk_cast).  torch's
    copy_ of such a pair is hipMemcpyAsync, which a stream capture records as a memcpy node -- and ROCm 7.2 does not report those nodes'
    parameters back, so a captured launch sequence that is to be re-issued by the library's own replayer must not contain one."""
    nb = dst.numel() * dst.element_size()
    if dst.dtype != src.dtype or nb != src.numel() * src.element_size() or not (dst.is_contiguous() and src.is_contiguous()) or nb % 4:
        return dst.copy_(src)
    H.check(H.lib.evk_cast(H.ptr(src), H.F32, H.ptr(dst), H.F32, nb // 4, H.stream()), 'copy')
    return dst


def set_shadow_fresh(p, sh):
    """Used by the fused optimizer, which writes the bf16 shadow itself."""
    _shadows[id(p)] = (sh, p._version, p)


def grad_buffer(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
    return p.grad


def grad_done(p):
    for fn in _grad_callbacks:
        fn(p)


# ----------------------------------------------------------------------------------------------------
# side streams: latency-bound branches (the relational-memory recurrence, the indication text encoder) run on their
# own HIP stream concurrently with the fat ResNet kernels; autograd replays each op's backward on its forward stream.
# ----------------------------------------------------------------------------------------------------
# HIP queue priority of each named side stream in the eager step (-1 = above the default).  All default: the HIP runtime gives every priority
# class hardware queues of its own, and a relational-memory stream in the high-priority class makes the eager step 60-110 % slower (48.3 ->
# 78.2 ms at 384^2, 30.8 -> 65.9 ms at 224^2: profiles/r05_stream_priorities.txt, r05_hw_queues.txt); EVK_RM_STREAM_PRIO reproduces that measurement.
SIDE_STREAM_PRIORITY = {'rm': int(tunable('EVK_RM_STREAM_PRIO', '0'))}
_side_streams = {}
_side_raw = set()            # (device, raw handle) of every side stream
_main_raw = {}              # device -> raw handle of the stream remembered in _main_stream
SIDE_STREAMS_ENABLED = [True]


def side_stream(name, device=None):
    """The named side stream of `device`; the stream current at the call is remembered as that device's main stream."""
    dev = H._cur_device() if device is None else device
    key = (name, dev)
    st = _side_streams.get(key)
    if st is None:
        # (picking side streams that share no hardware queue -- H.concurrent_streams -- changes nothing for the eager step: 48.20 -> 48.16 ms)
        st = _side_streams[key] = torch.cuda.Stream(device=dev, priority=SIDE_STREAM_PRIORITY.get(name, 0))
        _side_raw.add((dev, st.cuda_stream))
    raw = H._raw_stream(dev)
    if (dev, raw) not in _side_raw and _main_raw.get(dev) != raw:        # (raw handles: no Stream objects built or compared per call)
        _main_raw[dev] = raw
        _main_stream[dev] = torch.cuda.current_stream(dev)
    return st


_main_stream = {}


def existing_side_stream(name, device=None):
    """the named side stream of `device` if it has been created, else None (never creates one)"""
    dev = H._cur_device() if device is None else device
    return _side_streams.get((name, dev))


def _capturing(st):
    with torch.cuda.stream(st):
        return torch.cuda.is_current_stream_capturing()


def join_side_streams(into=None, skip=None):
    """Make `into` (default: the current stream) wait for everything queued on the side streams and on the main stream
    (called before the optimizer / gradient all-reduce, because parameter gradients are accumulated in place from those
    streams).  `skip` names a side stream to leave out.  While a step is being captured in a HIP graph only the streams that
    have joined the capture are waited for (an idle side stream has nothing queued and is not part of the graph)."""
    cur = H.current_stream() if into is None else into
    cap = _capturing(cur)
    for (name, dev), st in _side_streams.items():
        if dev == cur.device.index and st != cur and name != skip and (not cap or _capturing(st)):
            cur.wait_stream(st)
    main = _main_stream.get(cur.device.index)
    if main is not None and main != cur and (not cap or _capturing(main)):
        cur.wait_stream(main)


class _Uploader:
    """Host -> device upload of small arrays without stalling the host behind queued compute: a blocking
    `torch.tensor(list, device=...)` / `.to(device)` of pageable memory is a synchronous copy ON THE CURRENT STREAM and therefore
    waits for every kernel queued before it (the whole ResNet forward at the multi-view grouping point).  Here the data goes
    through a ring of pinned staging buffers on a dedicated copy stream; the compute stream only waits for the copy's event."""

    def __init__(self, device, slots=16, cap=1 << 18):
        self.device = device
        self.bufs = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.i = 0
        self.stream = torch.cuda.Stream(device=device)

    def put(self, arr):
        import numpy as np
        arr = np.ascontiguousarray(arr)
        nb = arr.nbytes
        tdt = torch.from_numpy(arr[:0].reshape(-1)).dtype
        if nb and torch.cuda.is_current_stream_capturing():
            # HIP-graph capture of a whole step (evoke_amd/graph.py): the copy becomes a memcpy node that re-reads ITS OWN pinned
            # staging buffer at every replay, so the buffer is dedicated to the graph (kept alive by the capture's owner) and no
            # event of the ring is synchronised while the stream is capturing
            # ... and the copy is a KERNEL that reads the pinned (device-mapped) buffer, not a memcpy: ROCm 7.2 does not report the
            # parameters of captured host-to-device memcpy nodes back faithfully and the step replayer (csrc/replay.hip) cannot
            # re-issue what it cannot read
            pad = (nb + 3) // 4 * 4
            host = np.zeros(pad, dtype=np.uint8)
            host[:nb] = arr.reshape(-1).view(np.uint8)
            stage = torch.from_numpy(host).pin_memory()
            CAPTURE_KEEPALIVE.append(stage)
            dst = torch.empty(pad, dtype=torch.uint8, device=self.device)
            H.check(H.lib.evk_cast(stage.data_ptr(), H.F32, H.ptr(dst), H.F32, pad // 4, H.stream()), 'upload (word copy from pinned memory)')
            return dst[:nb].view(tdt).view(arr.shape)
        if nb > self.bufs[0].numel() or nb == 0:
            return torch.from_numpy(arr).to(self.device)          # oversize / empty: plain (blocking) path
        k = self.i % len(self.bufs)
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()                          # the copy issued `slots` uploads ago; long done
        stage = self.bufs[k][:nb]
        stage.numpy()[:] = arr.reshape(-1).view(np.uint8)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            t = stage.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.events[k] = ev
        main.wait_event(ev)
        t.record_stream(main)
        return t.view(tdt).view(arr.shape)


_uploaders = {}
CAPTURE_KEEPALIVE = []          # pinned staging buffers referenced by memcpy nodes of captured step graphs


def upload(arr, device):
    """device tensor (same dtype / shape) from a host numpy array, uploaded asynchronously (see _Uploader)."""
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.from_numpy(arr).to(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    up = _uploaders.get(key)
    if up is None:
        up = _uploaders[key] = _Uploader(torch.device('cuda', key))
    return up.put(arr)


def index_tensor(values, device):
    """int64 device tensor from a host list / array, uploaded asynchronously."""
    import numpy as np
    device = torch.device(device)
    if device.type != 'cuda':
        return torch.as_tensor(values, dtype=torch.long, device=device)
    return upload(np.asarray(values, dtype=np.int64).reshape(-1), device)


WGRAD_SIDE_STREAM = [tunable('EVK_LINEAR_WGRAD_SIDE', '1') == '1']      # linear-layer dW on the 'wgrad' stream


_wgrad_join_queued = [False]


def _wgrad_join_callback():
    _wgrad_join_queued[0] = False
    cur = H.current_stream()
    cap = _capturing(cur)
    for (name, dev), st in _side_streams.items():
        if name == 'wgrad' and dev == cur.device.index and (not cap or _capturing(st)):
            cur.wait_stream(st)


def join_wgrad_at_backward_end():
    """Parameter gradients computed on the 'wgrad' stream must be visible to whoever reads `.grad` after `backward()`
    returns: queue (once per backward pass) an autograd-engine callback that makes the calling stream wait for it -- the
    engine itself only joins the streams of the graph's own nodes."""
    if not _wgrad_join_queued[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join_callback)
            _wgrad_join_queued[0] = True
        except RuntimeError:          # not inside a backward pass (direct call of a backward function in a test)
            pass


class wgrad_stream:
    """context: side stream 'wgrad' ordered after the current stream; `tensors` are kept alive for it."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.prev = None

    def __enter__(self):
        if SIDE_STREAMS_ENABLED[0] and WGRAD_SIDE_STREAM[0]:
            side = side_stream('wgrad')
            join_wgrad_at_backward_end()
            cur = H.current_stream()
            side.wait_stream(cur)
            for t in self.tensors:
                t.record_stream(side)
            # (set_stream directly: torch.cuda.stream()'s context manager costs ~10 us per entry, 60 entries per backward)
            self.prev = cur
            torch.cuda.set_stream(side)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_stream(self.prev)
            self.prev = None
        return False


def _z(*shape, dtype=BF16, device='cuda'):
    return torch.zeros(*shape, dtype=dtype, device=device)


def _e(*shape, dtype=BF16, device='cuda'):
    return torch.empty(*shape, dtype=dtype, device=device)


# ----------------------------------------------------------------------------------------------------
# raw GEMM helper
# ----------------------------------------------------------------------------------------------------
def gemm(A, B, Cm, M, N, K, a_mode=H.A_PLAIN, b_mode=H.B_PLAIN, lda=0, ldb=0, ldc=0, bias=None, resid=None, ldr=0,
         act=H.ACT_NONE, alpha=1.0, accumulate=False, batch=(1, 1), sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0),
         a_off=0, b_off=0, c_off=0, b_klog=0, b_tapstride=0, splitk=0, bias_stride=0):
    d = H.Gemm()
    d.A = A.data_ptr() + a_off * A.element_size()
    d.B = B.data_ptr() + b_off * B.element_size()
    d.C = Cm.data_ptr() + c_off * Cm.element_size()
    d.bias = bias.data_ptr() if bias is not None else None
    d.resid = resid.data_ptr() if resid is not None else None
    d.M, d.N, d.K, d.a_mode, d.b_mode = M, N, K, a_mode, b_mode
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.batch_outer, d.batch_inner = batch
    d.sAo, d.sAi = sA
    d.sBo, d.sBi = sB
    d.sCo, d.sCi = sC
    d.sRo, d.sRi = sR
    d.alpha, d.act = alpha, act
    d.c_dtype = H.dt(Cm)
    d.r_dtype = H.dt(resid) if resid is not None else H.BF16
    d.accumulate = 1 if accumulate else 0
    d.splitk = splitk
    d.b_klog, d.b_tapstride = b_klog, b_tapstride
    d.bias_stride_inner = bias_stride
    if accumulate:
        nb = H.lib.evk_gemm_workspace_bytes(C.byref(d))
        if nb > 0:
            ws = torch.empty(nb // 4, dtype=F32, device=Cm.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), nb
    H.check(H.lib.evk_gemm_launch(C.byref(d), H.stream()), 'gemm')


def _rows(x):
    return x.numel() // x.shape[-1]


# ----------------------------------------------------------------------------------------------------
# linear:  y = act(x W^T + b) (+ resid)
# ----------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, resid, act, out_f32):
        K = x.shape[-1]
        N = W.shape[0]
        M = _rows(x)
        Np = _pad8(N)
        w = shadow(W, pad_rows=(N != Np))
        y = _e(M, Np, dtype=F32 if out_f32 else BF16, device=x.device) if Np == N else _z(M, Np, dtype=F32 if out_f32 else BF16, device=x.device)
        gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=Np, bias=b, resid=resid, ldr=Np, act=act)
        ctx.save_for_backward(x, y if act == H.ACT_RELU else None)
        ctx.W, ctx.b, ctx.act, ctx.dims = W, b, act, (M, N, K, Np)
        ctx.has_resid = resid is not None
        return y.view(*x.shape[:-1], Np)

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        M, N, K, Np = ctx.dims
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        dres = dy if ctx.has_resid else None
        if ctx.act == H.ACT_RELU:
            g = _e(M, Np, device=dy.device)
            H.check(H.lib.evk_act_bwd(H.ptr(dy), H.ptr(y), H.ptr(g), M * Np, H.ACT_RELU, H.stream()), 'act_bwd')
            dy = g
        elif ctx.act != H.ACT_NONE:
            raise RuntimeError('fused activation %d has no fused backward; use ops.activation' % ctx.act)
        dx = None
        if ctx.needs_input_grad[0]:
            w = shadow(W, pad_rows=(N != Np))
            dx = _e(M, K, device=dy.device)
            gemm(dy, w, dx, M, K, Np, b_mode=H.B_KSTR, lda=Np, ldb=K, ldc=K)
            dx = dx.view(x.shape)
        wg, bg = W.requires_grad, b is not None and b.requires_grad
        if wg or bg:
            # parameter gradients feed nothing before the optimizer: second stream, overlapping the rest of the backward --
            # unless dy itself is handed on as the residual branch's gradient (autograd may then accumulate into it in place)
            with (wgrad_stream(dy, x) if dres is None else contextlib.nullcontext()):
                if wg:
                    gemm(dy, x, grad_buffer(W), N, K, M, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=Np, ldb=K, ldc=K, accumulate=True)
                if bg:
                    H.check(H.lib.evk_colsum(H.ptr(dy), H.ptr(grad_buffer(b)), M, N, Np, H.stream()), 'colsum')
            if wg:
                grad_done(W)
            if bg:
                grad_done(b)
        return dx, None, None, dres, None, None


def linear(x, W, b=None, resid=None, act=H.ACT_NONE, out_f32=False):
    """nn.Linear / Conv1d(k=1) on the last dim.  Returns [.., pad8(N)] (pad columns are zero)."""
    if W.dim() == 3:                      # Conv1d weight (out, in, 1): same memory as (out, in)
        assert W.shape[2] == 1
    assert x.dtype == BF16 and x.is_contiguous()
    if resid is not None:
        assert resid.is_contiguous() and resid.shape[-1] == _pad8(W.shape[0]) and act != H.ACT_RELU
    return _Linear.apply(x, W, b, resid, act, out_f32)


class _LinearT(torch.autograd.Function):
    """HF Conv1D: y = x W + b with W stored (in, out)  (GPT-2: transformers.pytorch_utils.Conv1D)."""

    @staticmethod
    def forward(ctx, x, W, b, resid):
        K, N = W.shape
        M = _rows(x)
        y = _e(M, N, device=x.device)
        gemm(x, shadow(W), y, M, N, K, b_mode=H.B_KSTR, lda=K, ldb=N, ldc=N, bias=b, resid=resid, ldr=N)
        ctx.save_for_backward(x)
        ctx.W, ctx.b, ctx.dims, ctx.has_resid = W, b, (M, N, K), resid is not None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        M, N, K = ctx.dims
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _e(M, K, device=dy.device)
            gemm(dy, shadow(W), dx, M, K, N, lda=N, ldb=N, ldc=K)            # dX = dY W^T : B[n'=k][k'=n] = W[k][n]
            dx = dx.view(x.shape)
        if W.requires_grad:
            gemm(x, dy, grad_buffer(W), K, N, M, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=K, ldb=N, ldc=N, accumulate=True)
            grad_done(W)
        if b is not None and b.requires_grad:
            H.check(H.lib.evk_colsum(H.ptr(dy), H.ptr(grad_buffer(b)), M, N, N, H.stream()), 'colsum')
            grad_done(b)
        return dx, None, None, (dy if ctx.has_resid else None)


class _SiblingLinears(torch.autograd.Function):
    """n nn.Linear layers with EQUAL shapes applied to the SAME input (q / k / v projections; k / v of a cross-attention):
    forward = one batched GEMM over the concatenated bf16 weight shadows (shared A, per-batch bias, one contiguous output
    plane per layer); backward = one GEMM for dX over the concatenated output gradients; the n weight / bias gradients go to
    the weight-gradient stream.  Replaces n GEMMs + (n - 1) gradient-accumulation adds per direction."""

    @staticmethod
    def forward(ctx, x, holder):
        lins = holder.lins
        n, N, K = len(lins), lins[0].weight.shape[0], x.shape[-1]
        R = x.numel() // K
        W = torch.cat([shadow(m.weight).view(-1) for m in lins]).view(n * N, K)
        b = torch.cat([m.bias.detach() for m in lins])
        out = torch.empty(n, R, N, dtype=BF16, device=x.device)
        gemm(x, W, out, R, N, K, lda=K, ldb=K, ldc=N, batch=(1, n), sB=(0, N * K), sC=(0, R * N), bias=b, bias_stride=N)
        ctx.save_for_backward(x, W)
        ctx.lins, ctx.dims = lins, (n, N, K, R)
        shape = x.shape[:-1] + (N,)
        return tuple(out[i].view(shape) for i in range(n))

    @staticmethod
    def backward(ctx, *dys):
        x, W = ctx.saved_tensors
        lins = ctx.lins
        n, N, K, R = ctx.dims
        dev = x.device
        dy = torch.cat([(g if g is not None else torch.zeros(R, N, dtype=BF16, device=dev)).reshape(R, N).to(BF16) for g in dys], dim=1).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(R, K, dtype=BF16, device=dev)
            gemm(dy, W, dx, R, K, n * N, b_mode=H.B_KSTR, lda=n * N, ldb=K, ldc=K)
            dx = dx.view(x.shape)
        with wgrad_stream(dy, x):
            for i, m in enumerate(lins):
                if m.weight.requires_grad:
                    gemm(dy, x, grad_buffer(m.weight), N, K, R, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=n * N, ldb=K, ldc=K, accumulate=True, a_off=i * N)
                if m.bias is not None and m.bias.requires_grad:
                    H.check(H.lib.evk_colsum(dy.data_ptr() + 2 * i * N, H.ptr(grad_buffer(m.bias)), R, N, n * N, H.stream()), 'colsum')
        for m in lins:
            if m.weight.requires_grad:
                grad_done(m.weight)
            if m.bias is not None and m.bias.requires_grad:
                grad_done(m.bias)
        return dx, None


class _LinHolder:
    def __init__(self, lins):
        self.lins = lins


FUSED_SIBLINGS = [True]


def sibling_linears(x, lins):
    """[lin(x) for lin in lins] for parameter holders with .weight (N, K) / .bias (N) of equal shape (N, K multiples of 8)."""
    N, K = lins[0].weight.shape
    if (not FUSED_SIBLINGS[0] or not x.is_cuda or len(lins) < 2 or N % 8 or K % 8 or any(m.weight.shape != (N, K) or m.bias is None for m in lins)):
        return tuple(linear(x, m.weight, m.bias) for m in lins)
    return _SiblingLinears.apply(x.contiguous(), _LinHolder(list(lins)))


def linear_t(x, W, b=None, resid=None):
    """x (.., in) @ W (in, out) + b (+ resid): the Conv1D layout of HF GPT-2 (in and out multiples of 8)."""
    assert x.dtype == BF16 and x.is_contiguous() and W.shape[0] % 8 == 0 and W.shape[1] % 8 == 0
    return _LinearT.apply(x, W, b, resid)


# ----------------------------------------------------------------------------------------------------
# activations / dropout
# ----------------------------------------------------------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        y = torch.empty_like(x)
        H.check(H.lib.evk_act_fwd(H.ptr(x), H.ptr(y), x.numel(), act, H.stream()), 'act_fwd')
        ctx.act = act
        ctx.save_for_backward(x if act in (H.ACT_GELU, H.ACT_GELU_NEW) else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        ref, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        H.check(H.lib.evk_act_bwd(H.ptr(dy), H.ptr(ref), H.ptr(dx), dy.numel(), ctx.act, H.stream()), 'act_bwd')
        return dx, None


def activation(x, act):
    assert x.dtype == BF16 and x.is_contiguous()
    return _Act.apply(x, act)


_seed_state = [0x5EED5EED]
DROPOUT_ENABLED = [True]      # tests switch the reference's dropout layers off (parity is defined without dropout)


def set_dropout_enabled(on):
    DROPOUT_ENABLED[0] = bool(on)


def next_seed():
    _seed_state[0] = (_seed_state[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return _seed_state[0]


_seed_epoch = {}


def advance_seed_epoch(device=None):
    """Device-side dropout epoch (evk_set_seed_epoch): call once at the start of every training step.  The increment is a
    device op, so a captured step graph advances it at every replay and its frozen seed arguments still draw fresh masks."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    ep = _seed_epoch.get(dev)
    if ep is None:
        ep = _seed_epoch[dev] = torch.zeros(1, dtype=torch.int64, device=torch.device('cuda', dev))
        H.check(H.lib.evk_set_seed_epoch(H.ptr(ep)), 'set_seed_epoch')
    ep.add_(1)
    return ep


def manual_seed(s):
    _seed_state[0] = (int(s) * 2654435761 + 0x5EED5EED) & 0xFFFFFFFFFFFFFFFF


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, resid, p, seed):
        y = torch.empty_like(x)
        H.check(H.lib.evk_dropout(H.ptr(x), H.ptr(resid), H.ptr(y), x.numel(), C.c_float(p), C.c_uint64(seed), H.stream()), 'dropout')
        ctx.p, ctx.seed, ctx.has_res = p, seed, resid is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        H.check(H.lib.evk_dropout(H.ptr(dy), None, H.ptr(dx), dy.numel(), C.c_float(ctx.p), C.c_uint64(ctx.seed), H.stream()), 'dropout')
        return dx, (dy if ctx.has_res else None), None, None


def dropout(x, p, training, resid=None):
    """nn.Dropout (+ optional fused residual add: y = dropout(x) + resid)."""
    if not training or p <= 0.0 or not DROPOUT_ENABLED[0]:
        assert resid is None, 'fuse the residual into the producing GEMM when dropout is inactive'
        return x
    assert x.dtype == BF16 and x.is_contiguous() and (resid is None or resid.is_contiguous())
    return _Dropout.apply(x, resid, float(p), next_seed())


def linear_dropout_resid(x, W, b, resid, p, training):
    """dense -> dropout -> + residual (HF BertSelfOutput/BertOutput): GEMM-epilogue residual when dropout is off."""
    if training and p > 0.0 and DROPOUT_ENABLED[0]:
        return dropout(linear(x, W, b), p, True, resid=resid)
    return linear(x, W, b, resid=resid)


# ----------------------------------------------------------------------------------------------------
# LayerNorm family
# ----------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, dgam, dbet, mode, eps):
        D = x.shape[-1]
        rows = _rows(x)
        y = _e(*x.shape, device=x.device)
        mean = _e(rows, dtype=F32, device=x.device)
        rstd = _e(rows, dtype=F32, device=x.device)
        H.check(H.lib.evk_layernorm_fwd(H.ptr(x), H.dt(x), H.ptr(y), H.BF16, H.ptr(gamma), H.ptr(beta), H.ptr(dgam), H.ptr(dbet),
                                        H.BF16, H.ptr(mean), H.ptr(rstd), rows, D, mode, C.c_float(eps), H.stream()), 'ln_fwd')
        ctx.save_for_backward(x, mean, rstd, dgam)
        ctx.gamma, ctx.beta, ctx.mode, ctx.eps = gamma, beta, mode, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, dgam = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        D = x.shape[-1]
        rows = _rows(x)
        dy = dy.contiguous()
        dx = _e(*x.shape, dtype=x.dtype, device=x.device)
        want_vec = gamma.requires_grad
        gg = grad_buffer(gamma) if want_vec else None
        gb = grad_buffer(beta) if want_vec else None
        ddg = ddb = None
        if dgam is not None:
            ddg = _e(*dgam.shape, device=x.device)
            ddb = _e(*dgam.shape, device=x.device)
        H.check(H.lib.evk_layernorm_bwd(H.ptr(dy), H.dt(dy), H.ptr(x), H.dt(x), H.ptr(gamma), H.ptr(dgam), H.BF16, H.ptr(mean),
                                        H.ptr(rstd), H.ptr(dx), H.dt(dx), H.ptr(gg), H.ptr(gb), H.ptr(ddg), H.ptr(ddb), rows, D,
                                        ctx.mode, C.c_float(ctx.eps), H.stream()), 'ln_bwd')
        if want_vec:
            grad_done(gamma)
            grad_done(beta)
        return dx, None, None, ddg, ddb, None, None


def layernorm(x, gamma, beta, eps=1e-5, mode=0, dgam=None, dbet=None):
    """mode 0: torch.nn.LayerNorm; mode 1: R2Gen LayerNorm (encoder_decoder.py:93-103); dgam/dbet: CLN deltas."""
    assert x.is_contiguous() and x.dtype in (BF16, F32)
    return _LayerNorm.apply(x, gamma, beta, dgam, dbet, mode, eps)


# ----------------------------------------------------------------------------------------------------
# multi-head attention core on projected q/k/v ([B,T,H*dh] / [B,S,H*dh] bf16, heads interleaved)
# ----------------------------------------------------------------------------------------------------
FUSED_ATTENTION = [tunable('EVK_FUSED_ATTENTION', '1') == '1']


class _FusedAttention(torch.autograd.Function):
    """ONE kernel each way (csrc/attn.hip): the 32 x S score tile of a workgroup stays in LDS, softmax by wavefront shuffles,
    K / V tiles staged through LDS; P (16-bit) is the only intermediate in HBM.  dK / dV are two K-strided batched GEMMs."""

    @staticmethod
    def forward(ctx, q, k, v, mask, heads, scale, causal, p_drop, seed):
        B, T, HD = q.shape
        S = k.shape[1]
        dh = HD // heads
        Sp = _pad8(S)
        dev = q.device
        P = _e(B, heads, T, Sp, device=dev)
        Pd = _e(B, heads, T, Sp, device=dev) if p_drop > 0 else None
        out = _e(B, T, HD, device=dev)
        mb = mq = 0
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous()
            if mask.dim() == 2:
                mb, mq = mask.shape[1], 0
            else:
                mb, mq = mask.shape[1] * mask.shape[2], mask.shape[2]
        H.check(H.lib.evk_attention_fwd(H.ptr(q), H.ptr(k), H.ptr(v), H.ptr(out), H.ptr(P), H.ptr(Pd), H.ptr(mask), mb, mq, int(causal), B, heads,
                                        T, S, dh, C.c_float(scale), C.c_float(p_drop), C.c_uint64(seed), H.stream()), 'attention_fwd')
        ctx.save_for_backward(q, k, v, P, Pd if Pd is not None else P)
        ctx.cfg = (B, T, S, Sp, HD, heads, dh, scale, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, P, Pv = ctx.saved_tensors
        B, T, S, Sp, HD, heads, dh, scale, p_drop, seed = ctx.cfg
        dev = q.device
        dout = dout.contiguous()
        dS = _e(B, heads, T, Sp, device=dev)
        dQ = _e(B, T, HD, device=dev)
        H.check(H.lib.evk_attention_bwd(H.ptr(dout), H.ptr(k), H.ptr(v), H.ptr(P), H.ptr(dS), H.ptr(dQ), B, heads, T, S, dh, C.c_float(scale),
                                        C.c_float(p_drop), C.c_uint64(seed), H.stream()), 'attention_bwd')
        dV = dK = None
        if ctx.needs_input_grad[2]:          # dV = P'^T . dO
            dV = _e(B, S, HD, device=dev)
            gemm(Pv, dout, dV, S, dh, T, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=Sp, ldb=HD, ldc=HD, batch=(B, heads),
                 sA=(heads * T * Sp, T * Sp), sB=(T * HD, dh), sC=(S * HD, dh))
        if ctx.needs_input_grad[1]:          # dK = dS^T . Q   (dS carries the 1/sqrt(d) factor)
            dK = _e(B, S, HD, device=dev)
            gemm(dS, q, dK, S, dh, T, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=Sp, ldb=HD, ldc=HD, batch=(B, heads),
                 sA=(heads * T * Sp, T * Sp), sB=(T * HD, dh), sC=(S * HD, dh))
        return dQ, dK, dV, None, None, None, None, None, None


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, heads, scale, causal, p_drop, seed):
        B, T, HD = q.shape
        S = k.shape[1]
        dh = HD // heads
        Sp = _pad8(S)
        dev = q.device
        sc = _e(B, heads, T, Sp, dtype=F32, device=dev)
        gemm(q, k, sc, T, S, dh, lda=HD, ldb=HD, ldc=Sp, alpha=scale, batch=(B, heads),
             sA=(T * HD, dh), sB=(S * HD, dh), sC=(heads * T * Sp, T * Sp))
        P = _e(B, heads, T, Sp, device=dev)
        Pd = _e(B, heads, T, Sp, device=dev) if p_drop > 0 else None
        mb = mq = 0
        if mask is not None:
            assert mask.dtype == torch.uint8 and mask.is_contiguous()
            if mask.dim() == 2:
                mb, mq = mask.shape[1], 0
            else:
                mb, mq = mask.shape[1] * mask.shape[2], mask.shape[2]
        H.check(H.lib.evk_softmax_fwd(H.ptr(sc), H.ptr(P), H.ptr(Pd), H.BF16, H.ptr(mask), mb, mq, int(causal), B, heads, T, S, Sp, Sp,
                                      C.c_float(p_drop), C.c_uint64(seed), H.stream()), 'softmax_fwd')
        Pv = Pd if Pd is not None else P
        # V rows beyond S are never multiplied by a non-zero probability, but must be readable: K = S (KSTR has no K%8 rule
        # on B; A_PLAIN needs K%8 -> use Sp and rely on zero pad columns of P with k < S guard on V via kend = Sp?)
        out = _e(B, T, HD, device=dev)
        vv = v
        if Sp != S:
            vv = _z(B, Sp, HD, device=dev)
            vv[:, :S].copy_(v)
        gemm(Pv, vv, out, T, dh, Sp, b_mode=H.B_KSTR, lda=Sp, ldb=HD, ldc=HD, batch=(B, heads),
             sA=(heads * T * Sp, T * Sp), sB=(Sp * HD, dh), sC=(T * HD, dh))
        ctx.save_for_backward(q, k, vv, P, Pv)
        ctx.cfg = (B, T, S, Sp, HD, heads, dh, scale, p_drop, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, vv, P, Pv = ctx.saved_tensors
        B, T, S, Sp, HD, heads, dh, scale, p_drop, seed = ctx.cfg
        dev = q.device
        dout = dout.contiguous()
        # dP' = dO . V^T   (f32 [B,h,T,Sp])
        dP = _e(B, heads, T, Sp, dtype=F32, device=dev)
        gemm(dout, vv, dP, T, Sp, dh, lda=HD, ldb=HD, ldc=Sp, batch=(B, heads),
             sA=(T * HD, dh), sB=(Sp * HD, dh), sC=(heads * T * Sp, T * Sp))
        # dV = P'^T . dO
        dV = _e(B, S, HD, device=dev)
        gemm(Pv, dout, dV, S, dh, T, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=Sp, ldb=HD, ldc=HD, batch=(B, heads),
             sA=(heads * T * Sp, T * Sp), sB=(T * HD, dh), sC=(S * HD, dh))
        dS = _e(B, heads, T, Sp, device=dev)
        H.check(H.lib.evk_softmax_bwd(H.ptr(dP), H.F32, Sp, H.ptr(P), H.ptr(dS), H.BF16, B * heads * T, S, Sp, C.c_float(scale),
                                      C.c_float(p_drop), C.c_uint64(seed), H.stream()), 'softmax_bwd')
        # dQ = dS . K   ;  dK = dS^T . Q      (dS already carries the 1/sqrt(d) factor)
        dQ = _e(B, T, HD, device=dev)
        kk = k
        if Sp != S:
            kk = _z(B, Sp, HD, device=dev)
            kk[:, :S].copy_(k)
        gemm(dS, kk, dQ, T, dh, Sp, b_mode=H.B_KSTR, lda=Sp, ldb=HD, ldc=HD, batch=(B, heads),
             sA=(heads * T * Sp, T * Sp), sB=(Sp * HD, dh), sC=(T * HD, dh))
        dK = _e(B, S, HD, device=dev)
        gemm(dS, q, dK, S, dh, T, a_mode=H.A_KSTR, b_mode=H.B_KSTR, lda=Sp, ldb=HD, ldc=HD, batch=(B, heads),
             sA=(heads * T * Sp, T * Sp), sB=(T * HD, dh), sC=(S * HD, dh))
        return dQ, dK, dV, None, None, None, None, None, None


FUSED_MAX_DH = [int(tunable('EVK_ATTN_FUSED_MAX_DH', '1000000'))]


def attention(q, k, v, heads, mask=None, causal=False, p_drop=0.0, training=False, scale=None):
    """softmax(q k^T * scale [masked]) v per head.  mask: uint8 [B,S] (keys) or [B,T,S], 1 = attend."""
    assert q.dtype == BF16 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    dh = q.shape[-1] // heads
    if scale is None:
        scale = 1.0 / math.sqrt(dh)
    p = float(p_drop) if (training and DROPOUT_ENABLED[0]) else 0.0
    # heads wider than FUSED_MAX_DH (EVK_ATTN_FUSED_MAX_DH) take the batched-GEMM path: the fused kernel re-streams K and V for every
    # 32-row query tile, which costs more than the f32 score round trip once a head is thousands of dims wide
    if FUSED_ATTENTION[0] and dh <= FUSED_MAX_DH[0] and H.lib.evk_attention_supported(k.shape[1], dh):
        return _FusedAttention.apply(q, k, v, mask, heads, float(scale), causal, p, next_seed() if p > 0 else 0)
    return _Attention.apply(q, k, v, mask, heads, float(scale), causal, p, next_seed() if p > 0 else 0)


# ----------------------------------------------------------------------------------------------------
# embedding (f32 table -> bf16 rows), optional scale / positional table / constant row
# ----------------------------------------------------------------------------------------------------
class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, pos, extra, scale, padding_idx, pos0=None, out32=None):
        rows, L = ids.numel(), ids.shape[-1]
        D = table.shape[1]
        out = _e(*ids.shape, D, device=table.device)
        if out32 is not None:
            assert out32.dtype == F32 and out32.is_contiguous() and out32.numel() == rows * D
        H.check(H.lib.evk_embedding_fwd(H.ptr(table), H.ptr(ids), H.ptr(pos), H.ptr(extra), H.ptr(out), H.BF16, rows, D, L,
                                        C.c_float(scale), table.shape[0], H.ptr(pos0), H.ptr(out32), H.stream()), 'embedding_fwd')
        ctx.save_for_backward(ids)
        ctx.table, ctx.pos, ctx.extra, ctx.scale, ctx.padding_idx = table, pos, extra, scale, padding_idx
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, = ctx.saved_tensors
        dout = dout.contiguous()
        rows, L = ids.numel(), ids.shape[-1]
        table, pos, extra = ctx.table, ctx.pos, ctx.extra
        D = table.shape[1]
        st = H.stream()
        if table.requires_grad:
            H.check(H.lib.evk_embedding_bwd(H.ptr(dout), H.dt(dout), H.ptr(ids), H.ptr(grad_buffer(table)), rows, D,
                                            C.c_float(ctx.scale), ctx.padding_idx, table.shape[0], st), 'embedding_bwd')
            grad_done(table)
        if pos is not None and pos.requires_grad:
            pid = torch.arange(L, device=ids.device).repeat(rows // L)
            H.check(H.lib.evk_embedding_bwd(H.ptr(dout), H.dt(dout), H.ptr(pid), H.ptr(grad_buffer(pos)), rows, D,
                                            C.c_float(1.0), -1, pos.shape[0], st), 'embedding_bwd')
            grad_done(pos)
        if extra is not None and extra.requires_grad:
            zid = torch.zeros(rows, dtype=torch.long, device=ids.device)
            # `extra` is row 0 of a (types, D) table: accumulate into that row
            H.check(H.lib.evk_embedding_bwd(H.ptr(dout), H.dt(dout), H.ptr(zid), H.ptr(grad_buffer(extra)), rows, D,
                                            C.c_float(1.0), -1, extra.shape[0], st), 'embedding_bwd')
            grad_done(extra)
        return None, None, None, None, None, None, None, None


def embedding(ids, table, pos=None, extra=None, scale=1.0, padding_idx=-1, pos0=None, out32=None):
    """out[r] = table[ids[r]]*scale + pos[r % L] + extra[0]; `pos` is a (>=L, D) table, `extra` a (types, D) table (row 0 used).
    pos0: device scalar added to the position index; out32: optional f32 buffer that receives the same rows unrounded."""
    assert ids.dtype == torch.long and ids.is_contiguous()
    return _Embedding.apply(ids, table, pos, extra, float(scale), int(padding_idx), pos0, out32)


# ----------------------------------------------------------------------------------------------------
# masked NLL over padded f32 logits  (encoder_decoder.py:393 + loss.py:9-22)
# ----------------------------------------------------------------------------------------------------
class _NllLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, wmask, V):
        rows = _rows(logits)
        ld = logits.shape[-1]
        lse = _e(rows, dtype=F32, device=logits.device)
        row = _e(rows, dtype=F32, device=logits.device)
        H.check(H.lib.evk_log_softmax_nll_rows(H.ptr(logits), H.ptr(lse), H.ptr(target), H.ptr(wmask), H.ptr(row), rows, V, ld, H.stream()),
                'nll_rows')
        # per-row values summed by torch's tree reduction: the loss has the same bits on every run (f32 atomics did not)
        acc = torch.stack((row.sum(), wmask.sum()))
        ctx.save_for_backward(logits, lse, target, wmask, acc)
        ctx.V = V
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, dloss):
        logits, lse, target, wmask, acc = ctx.saved_tensors
        rows, ld = _rows(logits), logits.shape[-1]
        gs = (dloss.to(F32) / acc[1]).reshape(1).contiguous()
        dl = _e(*logits.shape, device=logits.device)
        H.check(H.lib.evk_nll_bwd(H.ptr(logits), H.ptr(lse), H.ptr(target), H.ptr(wmask), H.ptr(gs), H.ptr(dl), rows, ctx.V, ld, ld,
                                  H.stream()), 'nll_bwd')
        return dl, None, None, None


def nll_loss(logits, target, wmask, V):
    """logits f32 [rows.., ld>=V]; target int64 [rows]; wmask f32 [rows] -> sum(-logp[target]*w)/sum(w)."""
    assert logits.dtype == F32 and logits.is_contiguous() and target.dtype == torch.long and wmask.dtype == F32
    return _NllLoss.apply(logits, target.contiguous(), wmask.contiguous(), V)


def log_softmax(logits, V, out=None):
    """f32 [rows, ld] -> f32 log-probs [rows, V] (decode path, no autograd)."""
    rows, ld = _rows(logits), logits.shape[-1]
    if out is None:
        out = _e(rows, V, dtype=F32, device=logits.device)
    assert out.dtype == F32 and out.is_contiguous() and tuple(out.shape) == (rows, V)
    H.check(H.lib.evk_log_softmax_nll_fwd(H.ptr(logits), H.ptr(out), None, None, None, None, rows, V, ld, V, H.stream()), 'log_softmax')
    return out
