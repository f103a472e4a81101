"""Optimizers of the training step on the HIP engine -- mirror of modules/optimizers.py:17-68.

`build_two_stage_optimizer(args, model)` keeps the reference's rules: non-finetune tasks use one group at pt_lr;
finetune splits parameters by name into {ResNet, text encoder, LN1/LN2, multi-view attention} at pt_lr and
{text_decoder, visual_self_atten_layers, multimodal_fusion_layers, visual_head, text_head} at ft_lr
(optimizers.py:27-33); 'AdamW' means torch.optim.Adam(weight_decay, amsgrad) (L2-coupled), 'RAdam' torch.optim.RAdam.

The engine flattens every group into contiguous f32 buffers (params / grads / exp_avg / exp_avg_sq [/ max]) plus the
bf16 GEMM-operand shadow, so that one fused kernel per group does clip_grad_value_ (trainer_v0401.py:262,434,455) +
the update + the shadow refresh, and the DDP all-reduce runs on the flat gradient buffer without copies.
"""
import torch

from . import hip as H
from . import ops


def split_param_groups(args, model):
    """optimizers.py:17-46 -> [(lr, [(name, param), ...]), ...]."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    if args.get('task', 'finetune') != 'finetune':
        return [(args['pt_lr'], named)]
    keys = ('text_decoder', 'visual_self_atten_layers', 'multimodal_fusion_layers', 'visual_head', 'text_head')
    ft = [(n, p) for n, p in named if any(k in n for k in keys)]
    pt = [(n, p) for n, p in named if not any(k in n for k in keys)]
    return [(args['pt_lr'], pt), (args['ft_lr'], ft)]


def _pad8(n):
    return (n + 7) // 8 * 8


def _view_like(flat_chunk, p):
    """A view of `flat_chunk` (numel elements) with p's logical shape AND physical layout."""
    if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
        co, ci, kh, kw = p.shape
        return flat_chunk.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return flat_chunk.view(p.shape)


class FusedOptimizer(torch.optim.Optimizer):
    """kind: 'RAdam' | 'AdamW' (= Adam + amsgrad flag, as the reference names it) | 'Adam'."""

    def __init__(self, groups, kind='RAdam', weight_decay=0.0, amsgrad=False, betas=(0.9, 0.999), eps=1e-8, clip_value=0.1):
        param_groups = [{'params': [p for _, p in named], 'lr': lr} for lr, named in groups]
        super().__init__(param_groups, dict(lr=groups[0][0], weight_decay=weight_decay, betas=betas, eps=eps))
        self.kind = 0 if kind == 'RAdam' else 1
        self.amsgrad = bool(amsgrad) and kind == 'AdamW'
        self.clip_value = clip_value
        self.steps = 0
        self.flat = []
        self._touched = set()
        self._global_touched = None     # uint8 per parameter over all groups: the union over ranks (GradReducer._share_touched), device tensor
        # Per-parameter step counts live ON THE DEVICE (st['steps'], int32 per parameter): a step that the dynamic loss scaler
        # skips (non-finite gradient anywhere) must not advance them and the host never reads the skip decision back.
        self.world = 1              # ranks whose gradients the all-reduce SUMS into the flat buffers (set by GradReducer)
        self._step_zeroes = False       # True once a step() with in-kernel gradient zeroing has run (CUDA, world == 1)
        ops.register_grad_callback(self._on_grad)
        import weakref
        me = weakref.ref(self)
        for g in self.param_groups:
            for p in g['params']:
                p._evk_optimizer = me          # ops.grads_owned: this optimizer divides the loss scale out of p.grad itself
        for g in self.param_groups:
            ps = g['params']
            dev = ps[0].device
            offs, tot = [], 0
            for p in ps:
                offs.append(tot)
                rows = p.shape[0] if p.dim() >= 2 else 1
                n = p.numel()
                if p.dim() == 2 and rows % 8:
                    n = _pad8(rows) * (p.numel() // rows)      # room for the zero-padded rows of the bf16 shadow
                tot += _pad8(n)
            fp = torch.zeros(tot, dtype=torch.float32, device=dev)
            fg = torch.zeros(tot, dtype=torch.float32, device=dev)
            sh = torch.zeros(tot, dtype=ops.BF16, device=dev) if dev.type == 'cuda' else None
            for p, o in zip(ps, offs):
                n = p.numel()
                v = _view_like(fp[o:o + n], p)
                v.copy_(p.data)
                p.data = v
                p.grad = _view_like(fg[o:o + n], p)
            st = dict(p=fp, g=fg, m=torch.zeros_like(fp), v=torch.zeros_like(fp), vmax=torch.zeros_like(fp) if self.amsgrad else None,
                      shadow=sh, offsets=offs, steps=torch.zeros(len(ps), dtype=torch.int32, device=dev),
                      offs_dev=torch.tensor(offs, dtype=torch.int64, device=dev), coef=torch.zeros(4 * len(ps), dtype=torch.float32, device=dev),
                      hp=torch.zeros(8, dtype=torch.float32, device=dev), hp_host=None, masks={})
            self.flat.append(st)
            if sh is not None:
                H.check(H.lib.evk_cast(H.ptr(fp), H.F32, H.ptr(sh), H.BF16, tot, H.stream()), 'cast')
                for p, o in zip(ps, offs):
                    rows = p.shape[0] if p.dim() >= 2 else 1
                    n = p.numel() if not (p.dim() == 2 and rows % 8) else _pad8(rows) * (p.numel() // rows)
                    ops.set_shadow_fresh(p, sh[o:o + n])

    def _on_grad(self, p):
        self._touched.add(id(p))

    def set_global_touched(self, bits):
        """bits: uint8 tensor, one entry per parameter in param_groups order, 1 where ANY rank produced a gradient this step (the
        data-parallel reducer's MAX all-reduce).  The next step() updates exactly those parameters on every rank."""
        n = sum(len(g['params']) for g in self.param_groups)
        if bits.numel() != n:
            raise ValueError('set_global_touched: %d entries for %d parameters' % (bits.numel(), n))
        self._global_touched = bits

    # ---- torch.optim-format state (what the reference checkpoints with `optimizer.state_dict()`, trainer_v0401.py:160-189)
    def state_dict(self):
        state, groups, idx = {}, [], 0
        for g, st in zip(self.param_groups, self.flat):
            ids = []
            counts = st['steps'].cpu().tolist()          # checkpoint time: the one host read of the device step counts
            for p, o, k in zip(g['params'], st['offsets'], counts):
                if k > 0:
                    n = p.numel()
                    ent = {'step': torch.tensor(float(k)),
                           'exp_avg': _view_like(st['m'][o:o + n], p).detach().clone(memory_format=torch.contiguous_format),
                           'exp_avg_sq': _view_like(st['v'][o:o + n], p).detach().clone(memory_format=torch.contiguous_format)}
                    if st['vmax'] is not None:
                        ent['max_exp_avg_sq'] = _view_like(st['vmax'][o:o + n], p).detach().clone(memory_format=torch.contiguous_format)
                    state[idx] = ent
                ids.append(idx)
                idx += 1
            hp = {k: v for k, v in g.items() if k != 'params'}
            hp['params'] = ids
            groups.append(hp)
        return {'state': state, 'param_groups': groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        if len(sd['param_groups']) != len(self.param_groups):
            raise ValueError('loaded state dict has a different number of parameter groups')
        for g, st, lg in zip(self.param_groups, self.flat, sd['param_groups']):
            counts = [0] * len(g['params'])
            if len(lg['params']) != len(g['params']):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for k, v in lg.items():
                if k != 'params':
                    g[k] = v
            for pi, (p, o, idx) in enumerate(zip(g['params'], st['offsets'], lg['params'])):
                n = p.numel()
                ent = sd['state'].get(idx)
                for name, buf in (('exp_avg', st['m']), ('exp_avg_sq', st['v']), ('max_exp_avg_sq', st['vmax'])):
                    if buf is None:
                        continue
                    view = _view_like(buf[o:o + n], p)
                    if ent is not None and name in ent:
                        view.copy_(ent[name].to(view.device))
                    else:
                        view.zero_()
                if ent is not None:
                    counts[pi] = int(float(ent['step']))
            st['steps'].copy_(torch.tensor(counts, dtype=torch.int32))

    def flat_grads(self):
        return [st['g'] for st in self.flat]

    def zero_grad(self, set_to_none=False):
        """Gradients live in the flat buffers the kernels accumulate into: they are zeroed, never freed.  On one rank step() has
        already consumed (zeroed) every gradient it read, so the pass is skipped unless gradients were produced since; with
        several ranks the all-reduce may have written into parameters this rank did not touch, so the buffers are always cleared."""
        if self._touched or self.world > 1 or not self._step_zeroes:
            for st in self.flat:
                st['g'].zero_()
            self._touched.clear()

    def _hp_tuple(self, g):
        b1, b2 = g['betas']
        return (float(g['lr']), float(b1), float(b2), float(g['eps']), float(g['weight_decay']), float(self.clip_value or 0.0), 0.0, 0.0)

    def sync_hparams(self):
        """Per-group hyper-parameters live in a small device buffer the update kernel reads (lr schedulers, load_state_dict and plain
        `group['lr'] = x` change the HOST dict): upload what changed.  Called by step() and, for a captured step, before every replay
        (a capture cannot upload, it only records the pointer)."""
        for g, st in zip(self.param_groups, self.flat):
            hp = self._hp_tuple(g)
            if st.get('hp_host') != hp:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError('FusedOptimizer: hyper-parameters changed inside a stream capture; call sync_hparams() before capturing')
                import numpy as np
                st['hp'].copy_(ops.upload(np.asarray(hp, dtype=np.float32), st['hp'].device))
                st['hp_host'] = hp

    def _touched_mask(self, g, st):
        """device uint8 mask of the group's parameters that received a gradient this step (cached per distinct set: a step kind
        always touches the same parameters)"""
        bits = bytes(1 if id(p) in self._touched else 0 for p in g['params'])
        if not any(bits):
            return None
        m = st['masks'].get(bits)
        if m is None:
            import numpy as np
            m = ops.upload(np.frombuffer(bits, dtype=np.uint8).copy(), st['p'].device).clone()
            if not torch.cuda.is_current_stream_capturing():      # a tensor born inside a capture belongs to the graph's pool
                st['masks'][bits] = m
        return m

    @torch.no_grad()
    def step(self, closure=None):
        """torch.optim semantics: parameters that received no gradient this step are skipped (no weight decay, no
        step-count increment) -- e.g. the BERT pooler always, visual_self_atten_layers on indication batches -- and every parameter
        is updated with ITS OWN step count (bias corrections / RAdam rectification per parameter, evaluated on the device).  One
        launch pair per group.  fp16 storage: the gradients carry the loss scale; a non-finite gradient anywhere (after the
        all-reduce, so every rank sees it) skips the whole step."""
        self.steps += 1
        ops.WEIGHT_EPOCH[0] += 1
        on_gpu = bool(self.flat) and self.flat[0]['p'].is_cuda
        if not on_gpu:
            raise RuntimeError('FusedOptimizer.step: the update runs in libevoke_hip.so; parameters must live on the GPU')
        ops.join_side_streams()
        if not torch.cuda.is_current_stream_capturing():
            ops.check_forward_guard(self.flat[0]['p'].device)          # (non-blocking: forward scans the GPU has already finished)
        scaler = ops.loss_scaler(self.flat[0]['p'].device)
        sstate = H.ptr(scaler.state) if scaler is not None else None
        if scaler is not None:
            for st in self.flat:
                scaler.check(st['g'])
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hparams()
        inv_world = 1.0 / float(self.world)
        if self.world > 1 and self._global_touched is None:
            raise RuntimeError('FusedOptimizer.step with world %d: the update mask must be the union over ranks of the parameters that '
                               'received gradients (GradReducer.finish() provides it); a rank-local mask lets the replicas drift apart' % self.world)
        first = 0
        for g, st in zip(self.param_groups, self.flat):
            n_g = len(g['params'])
            if self._global_touched is not None:
                mask = self._global_touched[first:first + n_g]          # on the device: no rank-local decision, no host read-back
            else:
                mask = self._touched_mask(g, st)
            first += n_g
            if mask is None:
                continue
            H.check(H.lib.evk_optim_group_step(H.ptr(st['p']), H.ptr(st['g']), H.ptr(st['m']), H.ptr(st['v']), H.ptr(st['vmax']), H.ptr(st['shadow']),
                                               st['p'].numel(), self.kind, H.ptr(st['hp']), H.ptr(st['offs_dev']), len(g['params']),
                                               H.ptr(st['steps']), H.ptr(mask), H.ptr(st['coef']), sstate, inv_world, int(self.world == 1),
                                               H.stream()), 'optim_group_step')
        if scaler is not None:
            scaler.update()
        self._step_zeroes = self.world == 1
        self._touched.clear()
        self._global_touched = None

    def replay_hook(self):
        """Host bookkeeping of a step that was just captured in a HIP graph (evoke_amd/graph.py): the returned function is called
        after every replay.  The optimizer kernel rewrote the parameters through raw pointers, so everything keyed on the weights'
        epoch (inference-time caches of derived weights, ops.WEIGHT_EPOCH) must see a new epoch, exactly as after an eager step()."""
        def hook():
            self.steps += 1
            ops.WEIGHT_EPOCH[0] += 1
        return hook


def build_two_stage_optimizer(args, model, clip_value=0.1):
    kind = args.get('optim', 'RAdam')
    return FusedOptimizer(split_param_groups(args, model), kind=kind, weight_decay=args.get('weight_decay', 0.0),
                          amsgrad=args.get('amsgrad', False), clip_value=clip_value)


def build_lr_scheduler(args, optimizer):
    if args.get('lr_scheduler') == 'StepLR':
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=args['step_size'], gamma=args['gamma'])
    return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode=args.get('monitor_mode', 'min'))
