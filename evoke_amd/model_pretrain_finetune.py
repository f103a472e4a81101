"""Drop-in mirror of models/model_pretrain_finetune_v0623_large_res.py (FineTune 21-217, Pretrain 220-395) on the
MI355X engine: same constructor (args dict, tokenizer, data_name), same forward signatures / return values, same
state_dict keys -- main_224.py / main_384.py can import these classes instead of the reference's.

The compute runs on hand-written HIP kernels (evoke_amd.ops / trunk / layers -> libevoke_hip.so).  There is no CPU
path: calling forward without a GPU and the built library raises.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import hip as H
from .config import tunable
from . import losses, ops
from .layers import (BertLayer, EncoderDecoder, LayerNormP, ProjectionHead, ScaledDotProductAttention, TextEncoderModel,
                     key_mask, multiview_fusion)
from .ops import BF16, F32
from .trunk import ResNet

NO_FINDING = "there is no evidence of pulmonary."


def _to_device_async(t, device):
    if t.is_cuda:
        return t
    return ops.upload(t.contiguous().numpy(), device)


class _Base(nn.Module):
    def __str__(self):
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + '\nTrainable parameters: {}'.format(params / 1e6)

    def visual_forward_mimic_cxr(self, images):
        att_feats, fc_feats = self.visual_extractor(images)
        ops.guard_finite(att_feats, 'the visual extractor output (ResNet-101 patch features)')      # fp16 storage: forward range guard
        return fc_feats, att_feats

    @staticmethod
    def _side_branch(name, fn, uses=()):
        """Run fn() on a side HIP stream (after everything queued so far); returns (result, stream or None).  `uses`: the input tensors
        of the branch -- the caching allocator must know that the side stream reads them (forward AND backward, which autograd replays on
        this stream), or a block freed during the backward can be handed out again while the branch's last kernels still read it."""
        if not ops.SIDE_STREAMS_ENABLED[0]:
            return fn(), None
        main = torch.cuda.current_stream()
        side = ops.side_stream(name)
        side.wait_stream(main)
        for t in uses:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(side)
        with torch.cuda.stream(side):
            out = fn()
        return out, side

    def _image_tokens(self, images, patient_ids, batch_size):
        """visual_forward + (multiview_fusion | plain LN1) + visual_head -> (fc (N,D) bf16, tokens (B,T,D) bf16)."""
        fc, att = self.visual_forward(images)
        x = torch.cat([fc.unsqueeze(1), att], dim=1)
        x = self.layer_norm_1(x)
        if self.args['is_multiview_learning']:
            x = multiview_fusion(x, patient_ids, batch_size, self.multiview_cross_attention, self.layer_norm_2)
        elif x.shape[0] != batch_size:
            x = x[:batch_size].contiguous()
        return fc, self.visual_head(x)


class FineTune(_Base):
    def __init__(self, args: dict, tokenizer: object, data_name: str):
        super().__init__()
        self.args = args
        self.tokenizer = tokenizer
        visual_dim = 2048
        self.visual_extractor = ResNet(args)
        self.text_encoder = TextEncoderModel(args, tokenizer)
        self.layer_norm_1 = LayerNormP(visual_dim)
        self.layer_norm_2 = LayerNormP(visual_dim)
        self.decoder_kind = args.get('text_decoder', 'r2gen')          # modules/utils.py:78 (commented-out switch)
        if self.decoder_kind == 'distilgpt2':
            from .gpt2 import DistilGPT2TextDecoderModel
            self.text_decoder = DistilGPT2TextDecoderModel(args, tokenizer)
        elif self.decoder_kind == 'r2gen':
            self.text_decoder = EncoderDecoder(args, tokenizer)
        else:
            raise ValueError('text_decoder must be r2gen or distilgpt2')
        text_dim = self.text_encoder.encoder.hidden_size
        self.visual_head = ProjectionHead(visual_dim, args['output_dim'], args['output_dim'], final_bn=True)
        self.text_head = ProjectionHead(text_dim, args['output_dim'], args['output_dim'], final_bn=True)
        self.multiview_cross_attention = ScaledDotProductAttention(visual_dim, visual_dim, visual_dim, h=8)
        nl, heads, hid = args['sk_fusion_num_layers'], args['fusion_num_heads'], args['output_dim']
        self.visual_self_atten_layers = nn.ModuleList([BertLayer(hid, heads) for _ in range(nl)])
        self.multimodal_fusion_layers = nn.ModuleList([BertLayer(hid, heads, cross=True) for _ in range(nl)])
        for layers in (self.visual_self_atten_layers, self.multimodal_fusion_layers):
            for m in layers.modules():
                if hasattr(m, 'weight') and isinstance(m.weight, nn.Parameter) and m.weight.dim() == 2:
                    nn.init.normal_(m.weight, std=0.02)
                    nn.init.zeros_(m.bias)
        self.visual_forward = self.visual_forward_mimic_cxr
        self.text_decoder_forward = self.text_decoder_forward_r2gen if self.decoder_kind == 'r2gen' else self.text_decoder_forward_gpt2

    def text_decoder_forward_gpt2(self, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask, mode='train'):
        if mode == 'train':
            return self.text_decoder(encoder_hidden_states, encoder_attention_mask, input_ids, attention_mask, stage='train')
        output = self.text_decoder(encoder_hidden_states, encoder_attention_mask, stage='test')
        gen_texts = self.tokenizer.decode_batch(output.cpu().tolist())
        ops.check_forward_guard(output.device, block=True)
        gt_texts = self.tokenizer.decode_batch(input_ids.cpu().tolist())
        if mode == 'sample':
            return [[t if len(t) > 0 else NO_FINDING for t in gen_texts], gt_texts]
        if mode == 'test':
            return [gen_texts, gt_texts]
        return [[t if len(t) > 0 else NO_FINDING for t in gen_texts], output]

    def freeze_encoder_models(self, freeze_image_encoder: bool, freeze_text_encoder: bool):
        if freeze_image_encoder:
            for m in (self.visual_extractor, self.visual_head):
                for p in m.parameters():
                    p.requires_grad = False
        if freeze_text_encoder:
            for m in (self.text_encoder, self.text_head):
                for p in m.parameters():
                    p.requires_grad = False

    def encoder_states(self, images, patient_ids, batch_size, inc_ids=None, inc_masks=None):
        """lines 152-203: image tokens -> (indication cross-fusion | visual self-attention) -> (B,T,D) bf16."""
        device = images.device
        y = side = None
        if inc_ids is not None:
            # the indication branch depends on the text only: run it on a side stream under the ResNet
            # the loaders hand the indication tokens over on the CPU (trainer_v0401.py:430): pinned ring + copy stream, never a
            # pageable copy on the compute stream (that one blocks the host until the stream has drained)
            inc_ids, inc_masks = _to_device_async(inc_ids, device), _to_device_async(inc_masks, device)
            y, side = self._side_branch('text', lambda: self.text_head(self.text_encoder(input_ids=inc_ids, attention_mask=inc_masks)), uses=(inc_ids, inc_masks))
        _, x = self._image_tokens(images, patient_ids, batch_size)
        enc_mask = torch.ones(x.shape[:2], dtype=torch.long, device=device)
        enc_mask.evk_all_ones = True          # host-side knowledge of the mask content (no .all() readback downstream)
        if inc_ids is not None:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                y.record_stream(torch.cuda.current_stream())
            ym = key_mask(inc_masks)
            for layer in self.multimodal_fusion_layers:
                x = layer(x, y, None, ym)
        else:
            for layer in self.visual_self_atten_layers:
                x = layer(x, None, None, None)
        return x, enc_mask

    def text_decoder_forward_r2gen(self, input_ids, attention_mask, encoder_hidden_states, encoder_attention_mask, mode='train',
                                   pending=None):
        if mode == 'train':
            logits = self.text_decoder.forward_logits(input_ids, encoder_hidden_states, attention_mask, encoder_attention_mask, pending)
            return losses.lm_loss(logits, input_ids, attention_mask, self.text_decoder.vocab_size + 1)
        from .decode import beam_search
        output = beam_search(self.text_decoder, encoder_hidden_states, encoder_attention_mask, self.args)
        gen_texts = self.tokenizer.decode_batch(output.cpu().tolist())
        ops.check_forward_guard(output.device, block=True)          # (the .cpu() above has synchronised: the verdict is there)
        gt_texts = self.tokenizer.decode_batch(input_ids.cpu().tolist())
        if mode == 'sample':
            return [[t if len(t) > 0 else NO_FINDING for t in gen_texts], gt_texts]
        if mode == 'test':
            return [gen_texts, gt_texts]
        return [[t if len(t) > 0 else NO_FINDING for t in gen_texts], output]

    @torch.no_grad()
    def generate_pipelined(self, batches, mode='inference', depth=None):
        """forward(..., mode=mode) over a sequence of batches, as a generator of its return values, with the device kept busy across
        batches.  The decode of a batch is a chain of ~100 x ~56 small dependent kernels: latency bound, most of the GPU idles while it
        runs, and the host can only issue it at the GPU's pace (a launch queue is finite).  So `depth` batches are decoded at the same time,
        each on its own HIP stream with its own persistent beam session and its own host thread, which hands the whole token
        loop to the launch queue from native code (evk_replay_run_n: the interpreter is not on the per-token path, the GIL is free);
        this thread meanwhile queues the visual extractor / fusion / text encoders of the following batch on a further stream.  Per batch the
        kernels, their order and therefore the results are exactly those of forward() (tests/test_model_gpu.py).  batches: iterable of
        (images, report_ids, report_masks, patient_ids, inc_ids, inc_masks).  depth: EVK_DECODE_DEPTH or 4.  (The distilgpt2 backend has no
        stepwise sessions: for it only the encoders run ahead, _generate_encoders_ahead.)"""
        import os
        from collections import deque
        if mode not in ('sample', 'inference'):
            raise ValueError
        if self.decoder_kind != 'r2gen':
            yield from self._generate_encoders_ahead(batches, mode)
            return
        from .decode import beam_search
        depth = max(1, int(depth if depth is not None else os.environ.get('EVK_DECODE_DEPTH', '4')))
        # EVK_DECODE_THREADS=0 (default since round 5): this thread issues the token steps of all searches round-robin; 1: one host thread per
        # search in flight issues its token loop (evk_replay_run_n, GIL released).  With every stream at the default priority and four searches
        # in flight the single issuing thread measures 137-138 k tokens/s, the threads 132-135 k (profiles/r05_decode_stream_priorities.txt)
        threaded = os.environ.get('EVK_DECODE_THREADS', '0') != '0'
        burst = max(1, int(os.environ.get('EVK_DECODE_BURST', '8')))          # token steps per native call ...
        ahead = max(1, int(os.environ.get('EVK_DECODE_AHEAD', '2')))          # ... and bursts a search may be ahead of the GPU
        pool = None
        if threaded:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=depth, thread_name_prefix='evk-decode')
        dev = torch.cuda.current_device()
        was_training = self.training
        self.eval()
        cur = torch.cuda.current_stream()
        # EVK_ENC_RESERVE_CUS = 8 / 16 / 24: the encoder stream keeps off that many of every 32 compute units (hip.masked_stream).  A
        # convolution grid fills every CU and the searches' ~10 us kernels then queue behind its tiles for a workgroup slot (4-5 x their lone
        # duration); measured on the decode workload, though, the encoders lose more than the searches gain -- 117 k tokens/s without a
        # mask, 103 / 91 / 67 k with 8 / 16 / 24 reserved: the pipeline is bound by the sum of the GPU work, not by the chains' latency -- so
        # the default is 0 (no mask)
        reserve = int(tunable('EVK_ENC_RESERVE_CUS', '0'))
        enc_s = H.masked_stream(cur.device, reserve) if reserve > 0 else torch.cuda.Stream()
        # Every stream at the DEFAULT priority (EVK_DECODE_PRIO: comma list to override).  Rounds 3-4 ran the searches on high-priority streams
        # ("their small kernels go first"); the HIP runtime gives each priority class hardware queues of its own, and the extra active queues
        # cost more than the ordering buys: 115.6 k tokens/s with two high-priority searches, 120.1 k with two default ones, 127.5 k with three
        # (105 k with three high-priority ones in round 3) -- profiles/r05_decode_stream_priorities.txt.
        # (streams picked so that no two share a hardware queue -- H.concurrent_streams -- measured no difference here: 115.1 vs 116.3 k tokens/s)
        prios = [int(v) for v in tunable('EVK_DECODE_PRIO', '').split(',') if v.strip() != '']
        dec_s = [torch.cuda.Stream(priority=(prios[i] if i < len(prios) else 0)) for i in range(depth)]
        for st in [enc_s] + dec_s:
            st.wait_stream(cur)

        def encode(batch):
            images, report_ids, report_masks, patient_ids, inc_ids, inc_masks = batch
            # the batch was produced on the CALLER's stream (a loader's `.to(device, non_blocking=True)`, a pre-processing kernel) after
            # this generator started: the encoder stream must wait for that work, batch by batch, and the caching allocator must know that
            # enc_s reads these blocks -- or a block the caller drops is handed to the next upload while the ResNet still reads it
            enc_s.wait_stream(torch.cuda.current_stream())
            for t in (images, report_ids, report_masks, inc_ids, inc_masks):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(enc_s)
            with torch.cuda.stream(enc_s):
                x, enc_mask = self.encoder_states(images, patient_ids, report_ids.shape[0], inc_ids, inc_masks)
                ev = torch.cuda.Event()
                ev.record(enc_s)
            return x, enc_mask, ev, report_ids

        def start(enc, slot):
            """-> [slot, generator of the stepwise search, report ids, result or None, host copy of the result, future]"""
            x, enc_mask, ev, report_ids = enc
            with torch.cuda.stream(dec_s[slot]):
                dec_s[slot].wait_event(ev)
                x.record_stream(dec_s[slot])
                enc_mask.record_stream(dec_s[slot])
                try:
                    gen = beam_search(self.text_decoder, x, enc_mask, self.args, slot=slot, as_iterator=True, burst=burst if threaded else 0)
                    job = [slot, gen, report_ids, None, None, None]
                except NotImplementedError:              # no session path for this geometry: the whole search in one go
                    return [slot, None, report_ids, beam_search(self.text_decoder, x, enc_mask, self.args), None, None]
                if threaded:
                    # everything in front of the token loop (cross-attention K / V, position 0, on first use the graph capture, which
                    # must not meet another thread's allocations) from THIS thread; the loop itself from a worker
                    if not advance(job):
                        job[5] = pool.submit(drive, job)
                return job

        def advance(job):
            """issue the next token step of a job; True when the search has been issued completely"""
            if job[1] is None:
                return True
            with torch.cuda.stream(dec_s[job[0]]):
                try:
                    next(job[1])
                    return False
                except StopIteration as e:
                    job[3] = e.value
                    job[1] = None
                    return True

        def drive(job):
            """(worker thread) the rest of one search: the token loop as native bursts (the GIL is released inside), then the download"""
            torch.cuda.set_device(dev)               # the current device, stream and grad mode are per thread
            marks = deque()
            with torch.no_grad():
                while not advance(job):
                    # at most `ahead` bursts queued: a thread that fills the launch queue to the brim blocks inside the runtime, and
                    # with it every other thread's launches (measured: the next batch's encoders then wait for the search to drain)
                    marks.append(torch.cuda.Event())
                    marks[-1].record(dec_s[job[0]])
                    if len(marks) > ahead:
                        marks.popleft().synchronize()
                with torch.cuda.stream(dec_s[job[0]]):
                    job[4] = job[3].cpu()

        def finish(job):
            # a worker clears job[1] (in advance()) BEFORE its download: whoever sees job[1] is None must still join the worker -- otherwise this
            # thread downloads the result a second time, frees the slot while the worker is inside .cpu() on that stream, and an exception raised
            # in the worker's tail is lost
            if job[5] is not None:
                job[5].result()
            seq, report_ids = job[3], job[2]
            if job[4] is None:
                with torch.cuda.stream(dec_s[job[0]]):
                    job[4] = seq.cpu()               # waits for that decode stream only
            ops.check_forward_guard(seq.device)      # verdicts of the encoder passes that have finished
            gen_texts = self.tokenizer.decode_batch(job[4].tolist())
            gen_texts = [t if len(t) > 0 else NO_FINDING for t in gen_texts]
            if mode == 'sample':
                return [gen_texts, self.tokenizer.decode_batch(report_ids.cpu().tolist())]
            return [gen_texts, seq]

        try:
            it = iter(batches)
            jobs = deque()                           # in flight, oldest first
            free = list(range(depth))
            staged = None                            # (threaded) the encoders of the batch after the ones in flight, already queued
            exhausted = False
            while True:
                while free and (staged is not None or not exhausted):     # fill the free slots: encoders, then the search joins the rotation
                    enc, staged = staged, None
                    if enc is None:
                        batch = next(it, None)
                        if batch is None:
                            exhausted = True
                            break
                        enc = encode(batch)
                    jobs.append(start(enc, free.pop(0)))
                if not jobs:
                    break
                if threaded:
                    if staged is None and not exhausted:                   # the workers issue the searches; this thread gets one batch ahead
                        batch = next(it, None)
                        if batch is None:
                            exhausted = True
                        else:
                            staged = encode(batch)
                    if jobs[0][5] is not None:
                        jobs[0][5].result()                                 # (re-raises what the worker raised)
                        jobs[0][1] = None
                else:
                    for job in list(jobs):           # one token step of every search in flight
                        advance(job)
                while jobs and jobs[0][1] is None:   # results leave in batch order
                    job = jobs.popleft()
                    yield finish(job)
                    free.append(job[0])
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            for st in [enc_s] + dec_s:
                cur.wait_stream(st)
            self.train(was_training)

    def _generate_encoders_ahead(self, batches, mode):
        """generate_pipelined for a decoder backend without stepwise sessions (distilgpt2: its generate() drives the search from the host,
        language_model.py:271-280): the encoders of batch k + 1 are queued on a second stream BEFORE batch k is generated, so the GPU
        runs them in the gaps of that host-driven search.  Results are those of forward(mode=mode), batch for batch."""
        was_training = self.training
        self.eval()
        cur = torch.cuda.current_stream()
        enc_s = torch.cuda.Stream()
        enc_s.wait_stream(cur)

        def encode(batch):
            images, report_ids, report_masks, patient_ids, inc_ids, inc_masks = batch
            enc_s.wait_stream(torch.cuda.current_stream())           # (the batch may have been uploaded on the caller's stream just now)
            for t in (images, report_ids, report_masks, inc_ids, inc_masks):
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(enc_s)
            with torch.cuda.stream(enc_s):
                x, enc_mask = self.encoder_states(images, patient_ids, report_ids.shape[0], inc_ids, inc_masks)
                ev = torch.cuda.Event()
                ev.record(enc_s)
            return x, enc_mask, ev, report_ids, report_masks

        def generate(enc):
            x, enc_mask, ev, report_ids, report_masks = enc
            here = torch.cuda.current_stream()
            here.wait_event(ev)
            x.record_stream(here)
            enc_mask.record_stream(here)
            ret = self.text_decoder_forward(report_ids.to(x.device), report_masks.to(x.device), x, enc_mask, mode=mode)
            return [ret[0], ret[1]]

        try:
            prev = None
            for batch in batches:
                enc = encode(batch)
                if prev is not None:
                    yield generate(prev)
                prev = enc
            if prev is not None:
                yield generate(prev)
        finally:
            cur.wait_stream(enc_s)
            self.train(was_training)

    def forward(self, images, report_ids, report_masks, patient_ids, inc_ids=None, inc_masks=None, mode='train'):
        if mode not in ('train', 'sample', 'inference'):
            raise ValueError
        report_ids, report_masks = report_ids.to(images.device), report_masks.to(images.device)
        kw = {}
        if mode == 'train' and self.text_decoder_forward == self.text_decoder_forward_r2gen:
            kw['pending'] = self.text_decoder.start_memory(report_ids)       # overlaps with the image path
        x, enc_mask = self.encoder_states(images, patient_ids, report_ids.shape[0], inc_ids, inc_masks)
        ret = self.text_decoder_forward(report_ids, report_masks, x, enc_mask, mode=mode, **kw)
        if mode == 'train':
            ret = ops.scale_loss(ret, self)
            return {'lm': ret, 'all_loss': ret}
        return [ret[0], ret[1]]


class Pretrain(_Base):
    def __init__(self, args: dict, tokenizer: object, data_name: str):
        super().__init__()
        self.args = args
        self.tokenizer = tokenizer
        visual_dim = 2048
        self.visual_extractor = ResNet(args)
        self.text_encoder = TextEncoderModel(args, tokenizer)
        self.layer_norm_1 = LayerNormP(visual_dim)
        self.layer_norm_2 = LayerNormP(visual_dim)
        text_dim = self.text_encoder.encoder.hidden_size
        self.visual_head = ProjectionHead(visual_dim, args['output_dim'], args['output_dim'], final_bn=False)
        self.text_head = ProjectionHead(text_dim, args['output_dim'], args['output_dim'], final_bn=False)
        self.multiview_cross_attention = ScaledDotProductAttention(visual_dim, visual_dim, visual_dim, h=8)
        self.visual_forward = self.visual_forward_mimic_cxr
        self.gather = None          # set by evoke_amd.distributed for the cross-rank contrastive negatives

    def obtain_text_embeds(self, input_ids, attention_mask):
        t = self.text_head(self.text_encoder(input_ids=input_ids, attention_mask=attention_mask))
        return ops.split_first_token(t)

    def multi_pos_contra_images_v0401(self, global_image_embed, patient_ids):
        return losses.multi_pos_contra_images(global_image_embed, patient_ids, self.args['region_temp'], self.gather)

    def global_alignment_loss(self, global_image_embed, global_text_embed, patient_ids):
        return losses.global_alignment(global_image_embed, global_text_embed, patient_ids, self.args['instance_temp'], self.gather)

    def local_text_token_alignment_loss(self, local_image_embed, local_text_embed):
        return losses.local_text_token_alignment(local_image_embed, local_text_embed, self.args['region_temp'])

    def forward(self, images, radgraph_ids, radgraph_masks, patient_ids):
        device = images.device
        b = radgraph_ids.shape[0]
        rid, rmask = radgraph_ids.to(device), radgraph_masks.to(device)
        (t_fc, t_att), side = self._side_branch('text', lambda: self.obtain_text_embeds(rid, rmask), uses=(rid, rmask))
        fc, tok = self._image_tokens(images, patient_ids, b)
        mul_pos_loss = torch.tensor([0.0])
        if self.args['is_multiview_learning']:
            mul_pos_loss = self.multi_pos_contra_images_v0401(fc, patient_ids)
        v_fc, v_att = ops.split_first_token(tok)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            t_fc.record_stream(torch.cuda.current_stream())
            t_att.record_stream(torch.cuda.current_stream())
        instance_loss = self.global_alignment_loss(v_fc, t_fc, patient_ids)
        sen_text_loss = self.local_text_token_alignment_loss(v_att, t_att)
        all_loss = instance_loss + sen_text_loss
        if self.args['is_multiview_learning']:
            all_loss = all_loss + mul_pos_loss
        return {'sen_image_loss': torch.tensor([0.0]), 'sen_text_loss': sen_text_loss, 'instance_loss': instance_loss,
                'multiview_loss': mul_pos_loss, 'all_loss': ops.scale_loss(all_loss, self)}
