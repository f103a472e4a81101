"""Step harness of the training path -- the minimal counterpart of modules/trainer_v0401.py.

What is mirrored (reference file:line):
  * one optimizer step = zero_grad -> forward -> backward -> clip_grad_value_(0.1) -> step
    (PTrainer._train_epoch :256-263, FTrainer._train_epoch :426-436, :450-456); here the clip lives inside the fused
    optimizer kernel and the gradient all-reduce (GradReducer) is started from the backward;
  * FTrainer's epoch order: all indication batches, then all no-indication batches (:423-463); PTrainer's five running
    loss sums averaged over the batch count (:244-251, :285-288);
  * checkpoint dict {'epoch', 'state_dict', 'optimizer', 'monitor_best'} written to
    <result_dir>/checkpoint/{current_checkpoint,model_best}.pth (:160-176), `resume` (:178-189) and the shape-filtered
    partial `load` that seeds stage 2 from stage-1 weights (:191-202);
  * best-metric tracking / early stop of BaseTrainer.train (:89-109) and the lr-scheduler step (:567-570).
What is NOT: dataset/dataloader construction, the external text metrics (CheXbert, RadGraph, METEOR...), CSV dumps.

Difference by design: the reference reads every loss back with .item() each step (5 device syncs per pretrain step);
the harness accumulates the loss scalars on the device and reads them once per log point / epoch end.
"""
import os
from math import inf

import torch

from . import distributed as D
from . import ops


def filter_state_for(model_state, loaded_state):
    """trainer_v0401.py:195-198: keep entries whose key exists in the model with the same shape; returns (valid, invalid keys)."""
    valid = {k: v for k, v in loaded_state.items() if k in model_state and v.shape == model_state[k].shape}
    invalid = {k for k in loaded_state if k not in valid}
    return valid, invalid


class Trainer:
    LOSS_KEYS = ('all_loss', 'sen_image_loss', 'sen_text_loss', 'instance_loss', 'multiview_loss')

    def __init__(self, model, optimizer, args, lr_scheduler=None, reducer=None, task='finetune', is_save_checkpoint=True, log=print):
        self.model, self.optimizer, self.args = model, optimizer, args
        self.lr_scheduler, self.reducer = lr_scheduler, reducer
        self.task, self.is_save_checkpoint, self.log = task, is_save_checkpoint, log
        self.epochs = args.get('epochs', 1)
        self.save_period = args.get('save_period', 1)
        self.mnt_mode = args.get('monitor_mode', 'max')
        assert self.mnt_mode in ('min', 'max', 'off')
        self.mnt_metric = 'val_' + args.get('monitor_metric', 'BLEU_4')
        self.mnt_best = inf if self.mnt_mode == 'min' else -inf
        self.early_stop = args.get('early_stop', inf)
        self.start_epoch = 1
        self.not_improved = 0
        self.checkpoint_dir = os.path.join(args.get('result_dir', '.'), 'checkpoint')
        self.world = D.world_size()
        # engine-namespaced key (the reference has no counterpart): capture each batch structure's whole step once and replay its
        # launch sequence from C++ (evoke_amd/graph.py); opt-in, see bench.py
        self.step_graphs = bool(args.get('evk_step_graphs', False))
        self._graphs = {}
        if args.get('resume'):
            self.resume_checkpoint(args['resume'])
        if args.get('load'):
            self.load_checkpoint(args['load'])

    # ------------------------------------------------------------------ checkpoints (trainer_v0401.py:160-202)
    def save_checkpoint(self, epoch, save_best=False):
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        state = {'epoch': epoch, 'state_dict': self.model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                 'monitor_best': self.mnt_best}
        path = os.path.join(self.checkpoint_dir, 'current_checkpoint.pth')
        torch.save(state, path)
        if save_best:
            torch.save(state, os.path.join(self.checkpoint_dir, 'model_best.pth'))
        return path

    def resume_checkpoint(self, path):
        ck = torch.load(str(path), map_location='cpu')
        self.model.load_state_dict(ck['state_dict'])
        self.start_epoch = ck['epoch'] + 1
        self.mnt_best = ck['monitor_best']
        self.optimizer.load_state_dict(ck['optimizer'])

    def load_checkpoint(self, path):
        loaded = torch.load(str(path), map_location='cpu')['state_dict']
        current = self.model.state_dict()
        valid, invalid = filter_state_for(current, loaded)
        self.log('invalid states for pretrained_model %s' % sorted(invalid))
        current.update(valid)
        self.model.load_state_dict(current, strict=False)
        return invalid

    # ------------------------------------------------------------------ one optimizer step
    MAX_STEP_GRAPHS = 8

    def _graphed_step(self, batch, kind):
        """train_step through evoke_amd.graph.StepGraph: one captured HIP graph per batch STRUCTURE (tensor shapes + multi-view
        grouping of patient_ids + step kind), inputs copied into that structure's static device buffers before every call.  A
        structure runs eagerly until it has been seen `warmup` times; at most MAX_STEP_GRAPHS structures are kept (LRU)."""
        from . import ops
        from .graph import StepGraph, structure_key
        dev = next(self.model.parameters()).device
        tens = [batch[0], batch[1], batch[2]] + [batch[i] for i in (4, 5) if len(batch) > i]
        key = structure_key(tens, batch[3], extra=(kind, self.model.training))
        ent = self._graphs.pop(key, None)
        if ent is None:
            static = [torch.empty(t.shape, dtype=t.dtype, device=dev) for t in tens]
            pids = list(batch[3])
            model, opt, red, task = self.model, self.optimizer, self.reducer, self.task

            def fn():
                ops.advance_seed_epoch()
                opt.zero_grad()
                if red is not None:
                    red.begin(D.batch_structure(kind, pids))
                if task == 'finetune':
                    ret = model(static[0], static[1], static[2], pids, *static[3:5], mode='train')
                else:
                    ret = model(static[0], static[1], static[2], pids)
                ret['all_loss'].backward()
                if red is not None:
                    red.finish()
                opt.step()
                return {k: v.detach() for k, v in ret.items() if torch.is_tensor(v)}
            ent = (static, StepGraph(fn, warmup=2, make_on_replay=getattr(opt, 'replay_hook', None)))
            while len(self._graphs) >= self.MAX_STEP_GRAPHS:
                self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = ent                     # most recently used last
        static, sg = ent
        for dst, src in zip(static, tens):
            dst.copy_(src, non_blocking=True)
        if hasattr(self.optimizer, 'sync_hparams'):
            self.optimizer.sync_hparams()           # lr scheduler / load_state_dict since the capture: the kernels read a device buffer
        return sg()

    def train_step(self, batch, kind=None):
        """batch: the loader tuple without the leading image ids -- pretrain (images, radgraph_ids, radgraph_masks,
        patient_ids); finetune (images, input_ids, attention_masks, patient_ids[, inc_ids, inc_masks]).
        Returns the model's loss dict (device tensors, detached)."""
        dev = next(self.model.parameters()).device
        if self.step_graphs and dev.type == 'cuda' and hasattr(self.optimizer, 'replay_hook') and D.world_size() == 1:
            k = kind or (('inc' if len(batch) >= 6 else 'no_inc') if self.task == 'finetune' else 'pretrain')
            return self._graphed_step(batch, k)
        images = batch[0].to(dev, non_blocking=True)
        from . import ops
        if dev.type == 'cuda':
            ops.advance_seed_epoch()
        self.optimizer.zero_grad()
        if self.task == 'finetune':
            has_inc = len(batch) >= 6
            if self.reducer is not None:
                self.reducer.begin(D.batch_structure(kind or ('inc' if has_inc else 'no_inc'), batch[3]))
            ids, masks = batch[1].to(dev), batch[2].to(dev)
            if has_inc:
                ret = self.model(images, ids, masks, batch[3], batch[4], batch[5], mode='train')
            else:
                ret = self.model(images, ids, masks, batch[3], mode='train')
        else:
            if self.reducer is not None:
                self.reducer.begin(D.batch_structure(kind or 'pretrain', batch[3]))
            ret = self.model(images, batch[1].to(dev), batch[2].to(dev), batch[3])
        loss = ret['all_loss']
        loss.backward()              # never pre-divided by world: the reducer SUMS, the optimizer kernel applies 1/world
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()                    # clip_grad_value_(0.1) is fused into the update kernel
        return {k: v.detach() for k, v in ret.items() if torch.is_tensor(v)}

    # ------------------------------------------------------------------ epochs
    def _run(self, loader, sums, log_every, tag, epoch):
        n = 0
        for batch_idx, batch in enumerate(loader):
            ret = self.train_step(batch[1:])
            for k in sums:
                if k in ret:            # device-side accumulation: no .item() per step
                    v = ret[k].float().reshape(-1)[0].to(ret['all_loss'].device)
                    sums[k] = v if sums[k] is None else sums[k] + v
            n += 1
            if log_every and batch_idx % log_every == 0:
                self.log('Epoch %d, %s step %d, all_loss: %.4f, lr: %s' % (epoch, tag, batch_idx, float(ret['all_loss']),
                                                                         self.optimizer.param_groups[0]['lr']))
        return n

    def train_epoch_pretrain(self, loader, epoch=1, log_every=2000):
        """PTrainer._train_epoch training part (:242-300): returns {'epoch', 'train_<loss>': mean over batches}."""
        self.model.train()
        sums = {k: None for k in self.LOSS_KEYS}
        n = self._run(loader, sums, log_every, 'pretrain', epoch)
        ops.check_forward_guard(block=True)          # fp16 storage: a forward activation that left the range is an error, not a NaN loss in a log
        return {'epoch': epoch, **{'train_' + k: (float(v) if v is not None else 0.0) / max(n, 1) for k, v in sums.items()}}

    def train_epoch_finetune(self, loader_inc, loader_not_inc, epoch=1, log_every=2000):
        """FTrainer._train_epoch training part (:418-465): indication batches first, then the others."""
        self.model.train()
        sums = {'all_loss': None}
        n = 0
        if loader_inc is not None:
            n += self._run(loader_inc, sums, log_every, 'indication', epoch)
        n += self._run(loader_not_inc, sums, log_every, 'no-indication', epoch)
        ops.check_forward_guard(block=True)
        return {'train_loss': (float(sums['all_loss']) if sums['all_loss'] is not None else 0.0) / max(n, 1), 'epoch': epoch}

    @torch.no_grad()
    def generate_epoch(self, loader_inc, loader_not_inc):
        """FTrainer validation/test loops (:470-494): mode='sample' over both loaders -> (image ids, generated, ground truth)."""
        self.model.eval()
        dev = next(self.model.parameters()).device
        ids_all, res, gts = [], [], []
        for loader in (loader_inc, loader_not_inc):
            if loader is None:
                continue
            for batch in loader:
                images = batch[1].to(dev)
                ids, masks = batch[2].to(dev), batch[3].to(dev)
                extra = tuple(batch[5:7]) if len(batch) >= 7 else ()
                gen, gt = self.model(images, ids, masks, batch[4], *extra, mode='sample')
                res.extend(gen)
                gts.extend(gt)
                ids_all.extend(batch[0])
        return ids_all, res, gts

    # ------------------------------------------------------------------ best tracking (BaseTrainer.train :89-121)
    def end_of_epoch(self, epoch, log):
        """Returns (best, stop).  Saves the checkpoint like BaseTrainer.train; steps the lr scheduler like :567-570."""
        best = False
        if self.mnt_mode != 'off' and self.mnt_metric in log:
            v = log[self.mnt_metric]
            improved = (self.mnt_mode == 'min' and v <= self.mnt_best) or (self.mnt_mode == 'max' and v >= self.mnt_best)
            if improved:
                self.mnt_best, self.not_improved, best = v, 0, True
            else:
                self.not_improved += 1
        if self.lr_scheduler is not None:
            if self.args.get('lr_scheduler') == 'StepLR':
                self.lr_scheduler.step()
            elif ('val_' + self.args.get('lr_monitor_metric', '')) in log:
                self.lr_scheduler.step(log['val_' + self.args['lr_monitor_metric']])
        if epoch % self.save_period == 0 and self.is_save_checkpoint and D.rank() == 0:
            self.save_checkpoint(epoch, save_best=best)
        return best, self.not_improved > self.early_stop
