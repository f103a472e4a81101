#!/usr/bin/env python3
"""bench.py -- throughput of the EVOKE hot path on MI355X (contract: see the task statement / DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one full training step of FineTune on one synthetic batch per rank: forward + backward + gradient
all-reduce + clip_grad_value_(0.1) + two-group RAdam (the reference's FTrainer._train_epoch body,
modules/trainer_v0401.py:426-435).  Workload at N=1 = BASELINE.json configs[2] per GPU: 2-view 384x384, 32 studies
(64 images) per GPU, report length 100, indication length 30, V = 1444 (the shipped IU-Xray tokenizer; the MIMIC vocab is
not shipped), random-init weights, inputs resident in HBM before the timed region.  Weak scaling: every rank processes
its own 32 studies; value = all ranks' studies / max-over-ranks time.

Prints ONE JSON line on rank 0 with the metric plus `roofline` (GEMM / implicit-GEMM kernel family, HIP-event timed
inside the library over the timed region) and `cpu_baseline` (the CPU oracle = restatement of the reference, timed on
this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

V = 1444
# algorithmic forward+backward FLOPs per study counted on the reference (SURVEY.md section 8d / BASELINE.md section 2)
ALG_GFLOP_PER_STUDY = {('finetune', 384): 453.2, ('finetune', 224): 60.3 * 2.868, ('pretrain', 384): 394.2, ('pretrain', 224): 141.4}
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md, chip-level parameters (dense; 5 PF is 2:1 sparse)


def make_args(task):
    from evoke_amd.config import ARGS
    a = dict(ARGS)
    a.update(task=task, optim='RAdam', pt_lr=5e-6, ft_lr=5e-5, weight_decay=1e-4, amsgrad=True)
    return a


def study_layout(B, views, seed):
    """study index of every image, anchors (one per study) first, then the other views of the same studies
    (dataloaders_v0623.py:60-116).  views: int, or 'u4' = 1..4 views per study, uniform, seeded (BASELINE config 4:
    'Multi-view CXR (<=4 views/study, variable)', SURVEY.md section 8d: seed 7) -- ragged image counts per rank."""
    if str(views) == 'u4':
        rng = np.random.RandomState(seed)
        per = rng.randint(1, 5, size=B)
    else:
        per = np.full(B, int(views))
    studies = list(range(B))
    for s in range(B):
        studies += [s] * (int(per[s]) - 1)
    return studies


def synth_batch(kind, B, views, res, L, Li, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    studies = study_layout(B, views, 7 + (seed % 1000))
    N = len(studies)
    images = torch.randn(N, 3, res, res, generator=g)
    ids = torch.randint(5, V - 2, (B, L), generator=g)
    if kind == 'finetune':
        ids[:, 0], ids[:, -1] = V - 2, V - 1          # [BOS] ... [EOS]
    else:
        ids[:, 0] = 1                                  # [CLS]
    masks = torch.ones(B, L, dtype=torch.long)
    inc = torch.randint(5, V - 2, (B, max(Li, 1)), generator=g)
    inc[:, 0] = 1
    inc_masks = torch.ones(B, max(Li, 1), dtype=torch.long)
    pids = np.array(['p%08d_s%08d' % (seed * 1000 + s, s) for s in studies])
    return dict(images=images.to(device), ids=ids.to(device), masks=masks.to(device), inc=inc.to(device),
                inc_masks=inc_masks.to(device), pids=pids)


def cpu_baseline(model, kind, res, L, Li, seconds_budget=25.0):
    """The CPU oracle (oracle/, restatement of the reference pinned by tests/golden) timed on the host cores."""
    from oracle import functional as O
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, int(os.environ.get('EVK_CPU_THREADS', '16'))))    # a 1-GPU box has a 16-CPU share
    torch.set_num_threads(threads)
    P = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().clone().contiguous()
         for k, v in model.state_dict().items() if not k.endswith('position_ids')}
    train_keys = [k for k, v in P.items() if v.is_floating_point() and not any(s in k for s in ('running_', '.pe', 'num_batches'))]
    for k in train_keys:
        P[k].requires_grad_(True)
    opt = torch.optim.RAdam([P[k] for k in train_keys], lr=5e-6, weight_decay=1e-4)
    B = 2
    b = synth_batch(kind, B, 2, res, L, Li, 'cpu', 7)
    ctx = O.Ctx(train=True, dropout=True)
    times = []
    t_all = time.time()
    for it in range(7):            # 1 warm-up + up to 6 timed steps: ~10-15 s of CPU work, bounded by seconds_budget
        t0 = time.time()
        opt.zero_grad()
        if kind == 'finetune':
            ret = O.finetune_forward_train(P, b['images'], b['ids'], b['masks'], b['pids'], b['inc'], b['inc_masks'], O.DEFAULT_CFG, ctx)
        else:
            ret = O.pretrain_forward(P, b['images'], b['ids'], b['masks'], b['pids'], O.DEFAULT_CFG, ctx)
        ret['all_loss'].backward()
        torch.nn.utils.clip_grad_value_([P[k] for k in train_keys], 0.1)
        opt.step()
        times.append(time.time() - t0)
        print('[cpu_baseline] step %d: %.2f s (%d threads)' % (it, times[-1], threads), file=sys.stderr, flush=True)
        if time.time() - t_all > seconds_budget and it >= 1:
            break
    t = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return dict(value=B / t, unit='studies/s', cores=threads, kind='port',
                sample='oracle (CPU restatement of the reference, fp32 eager) %s train step, batch %d x 2 views %dx%d, '
                       'median of %d steps after 1 warm-up' % (kind, B, res, res, max(1, len(times) - 1)))


def decode_record(model, a, rank, world, dev, with_cpu):
    """BASELINE.json configs[4]: EVOKE-384 inference, beam search (default beam 4), bs 64 studies x 2 views, max_seq_len 100,
    replicas only.  One step = the whole FineTune.forward(mode='inference') of one batch (visual extractor + fusion + beam search);
    every hypothesis is extended for all max_seq_len steps as in the reference (finished beams keep running with -1000), so
    tokens = batch x max_seq_len per step.  Returns the record dict (rank 0) or None.
    roofline: the per-token decode step (one replay of the captured step: 52 kernels) against HBM -- achieved = ALGORITHMIC bytes of one
    step (the decoder-step weights once + per hypothesis the self-attention cache up to t, the cross-attention
    K/V of the 144 patches and the 1536-wide memory row, in 16-bit) / measured step time (HIP events around the replays)."""
    from evoke_amd import decode as DEC, hip as H, metrics
    from evoke_amd.config import load_default_tokenizer as load_tokenizer
    beam, B, L = a.beam, a.decode_batch, 100
    model.eval()
    model.args['beam_size'], model.args['max_seq_len'] = beam, L
    b = synth_batch('finetune', B, a.views, a.res, L, 30, dev, 2000 + rank)

    def step(bb=b):
        with torch.no_grad():
            return model(bb['images'], bb['ids'], bb['masks'], bb['pids'], bb['inc'], bb['inc_masks'], mode='inference')[1]

    # Replicas: no rank waits for another inside the measurement.  A rank that fails reports an infinite time, and the ONE collective
    # (max over ranks) sits outside the guarded region so that every rank reaches it whatever happened -- a barrier inside would hang
    # the whole job on a single rank's exception.
    n = max(2, min(a.steps, 16))          # batches in the timed region (the pipeline fills during the first and drains during the last)
    step_ms, dt, err = [], float('inf'), None
    pipelined = os.environ.get('EVK_DECODE_PIPELINE', '1') != '0'
    depth = max(1, int(os.environ.get('EVK_DECODE_DEPTH', '4'))) if pipelined else 1
    try:
        for _ in range(max(1, min(a.warmup, 2))):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pipelined:
            # the serving loop: encoders of batch k+1 overlap the decode of batch k (FineTune.generate_pipelined); same per-batch results
            tup = (b['images'], b['ids'], b['masks'], b['pids'], b['inc'], b['inc_masks'])
            for _, seq in model.generate_pipelined([tup] * n, mode='inference', depth=depth):
                ev0, ev1, cnt = DEC.stats['step_events']
                step_ms.append((ev0, ev1, cnt))
        else:
            for _ in range(n):
                seq = step()
                ev0, ev1, cnt = DEC.stats['step_events']
                step_ms.append((ev0, ev1, cnt))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    except Exception as e:          # noqa: BLE001
        err = e
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    if err is not None:
        raise err
    if rank != 0:
        return None
    dt = float(tt.item())
    if not (dt < float('inf')):
        raise RuntimeError('decode failed on another rank')
    per_step_ms = sum(e0.elapsed_time(e1) for e0, e1, c in step_ms) / max(1, sum(c for _, _, c in step_ms))
    td = model.text_decoder
    w_params = sum(p.numel() for m in (td.model.decoder, td.model.rm, td.logit) for p in m.parameters())
    R = B * beam
    t_mean = (L - 1) / 2.0
    state_bytes = (t_mean * 3 * 2 * 512 + 144 * 3 * 2 * 512 + 1536) * 2.0
    alg_bytes = 2.0 * w_params + R * state_bytes
    # achieved = the algorithmic bytes of EVERY token step issued in the timed region / the region's wall time (encoders of the batches
    # included: they share the GPU with the searches in flight) -- reproducible from ms_per_batch alone: L * alg_bytes / ms_per_batch.
    # per_search_step_ms (HIP events around one search's replays) is reported beside it; with `depth` searches in flight it is LONGER than
    # ms_per_batch / L, which is the effective time per token step of the pipeline.
    ms_per_batch = 1e3 * dt / n
    eff_step_ms = ms_per_batch / L
    ach = L * alg_bytes / (ms_per_batch * 1e-3) / 1e9
    n_launch = DEC.stats.get('step_launches')
    mean_len = float((seq != 0).sum(1).float().mean().item())
    rec = {
        'metric': 'decode tokens/sec (beam search, %d^2)' % a.res, 'value': B * L * n * world / dt, 'unit': 'tokens/s', 'steps': n,
        # `value` counts token POSITIONS run (every hypothesis is extended for all max_seq_len positions, as the reference does);
        # emitted_tokens_per_s counts only the tokens of the returned reports (up to and excluding the padding after [EOS])
        'positions_per_s': B * L * n * world / dt, 'emitted_tokens_per_s': B * mean_len * n * world / dt,
        'ms_per_batch': ms_per_batch, 'higher_is_better': True, 'dtype': H.STORE,
        'config': {'workload': 'EVOKE-%d inference: beam=%d, batch %d studies x %d views, max_seq_len %d (all steps run, as the reference), '
                               'incremental decoder state, visual extractor + fusion included, V=%d, random-init weights'
                               % (a.res, beam, B, a.views, L, V), 'parallelism': 'replicas x%d' % world,
                   'mean_generated_len': mean_len, 'hip_graph_step': DEC.stats.get('graph'),
                   'pipelined_encoders': pipelined, 'searches_in_flight': depth,
                   'host_thread_per_search': pipelined and os.environ.get('EVK_DECODE_THREADS', '0') != '0',
                   'inference_trunk_mode': int(__import__('evoke_amd.trunk', fromlist=['FOLD_BN']).FOLD_BN[0]),
                   'relational_memory': 'f32 (default)' if DEC._RM_F32[0] else '16-bit (EVK_DECODE_RM_F32=0)',
                   'fused_beam_bookkeeping': DEC.stats.get('fused_bookkeeping'), 'launches_per_token_step': n_launch},
        'roofline': {'bound': 'hbm', 'kernel': 'per-token decode step (captured once, re-issued by csrc/replay.hip: relational-memory step, 3 decoder layers, '
                                               'logits, log-softmax, beam step = %s launches)' % (n_launch if n_launch else '~50'),
                     'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0, 'traffic': None,
                     'algorithmic_bytes_per_step': alg_bytes, 'step_ms': eff_step_ms, 'steps_per_batch': L, 'hypotheses': R,
                     'per_search_step_ms': per_step_ms, 'searches_in_flight': depth,
                     'overlap_factor': (per_step_ms / eff_step_ms) if eff_step_ms > 0 else None,
                     'note': 'achieved = steps_per_batch x algorithmic_bytes_per_step / ms_per_batch (whole pipeline: encoders of the next batch and `searches_in_flight` '
                             'independent searches share the GPU); step_ms = ms_per_batch / steps_per_batch, so step_ms x steps_per_batch = ms_per_batch by construction; '
                             'per_search_step_ms = HIP events around ONE search\'s replays under that overlap (a chain of small dependent kernels)'},
    }
    if with_cpu:
        try:
            rec['cpu_baseline'], rec['parity'] = decode_cpu_baseline(model, a, dev, beam, L)
        except Exception as e:          # noqa: BLE001
            rec['cpu_baseline'] = {'value': None, 'unit': 'tokens/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: %r' % (e,)}
    return rec


def decode_cpu_baseline(model, a, dev, beam, L):
    """The CPU oracle's generation (oracle/beam.py: full-prefix re-decode per step, as the reference) on a bounded sample -- 2
    studies x 2 views -- and, on the SAME inputs and weights, the engine's sequences: token agreement + BLEU-4 of the engine's
    reports against the oracle's (evoke_amd/metrics.py)."""
    from evoke_amd import metrics
    from oracle import beam as OB, functional as O
    from evoke_amd.config import load_default_tokenizer as load_tokenizer
    threads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get('EVK_CPU_THREADS', '16'))))
    torch.set_num_threads(threads)
    Bc = 2
    b = synth_batch('finetune', Bc, a.views, a.res, L, 30, 'cpu', 11)
    # The comparison runs on the PROCEDURAL weights of the golden fixtures (oracle/spec.py: a well-conditioned network, the one the
    # parity tests pin), in a second model object of the same architecture.  On the benchmark's random-init weights -- eval-mode BN
    # with untrained running statistics -- near-ties decide almost every token and the agreement figure only measured that
    # (0.02 ... 0.7 from run to run); the CPU timing does not depend on the weights.
    from evoke_amd.model_pretrain_finetune import FineTune
    from oracle import spec as S
    from evoke_amd.config import load_state_by_key as load_procedural
    pm = FineTune(dict(model.args), load_tokenizer(), 'mimic_cxr')
    load_procedural(pm, S.procedural_state(S.finetune_spec(V)), device=dev)
    pm.eval()
    from evoke_amd import decode as DEC
    picks = []          # the engine's own selection per position: (Bc, beam) flat candidate indices (parent beam * (V+1) + word), engine beam order

    def watch(t, logp, beam_sum):
        nb = 1 if t == 0 else beam
        cand = (beam_sum[:, :nb].unsqueeze(-1) + logp.view(Bc, nb, logp.shape[-1])[:, :, :V + 1]).reshape(Bc, -1)
        picks.append(cand.topk(beam, dim=1).indices.cpu())

    with torch.no_grad():
        xs, ms = pm.encoder_states(b['images'].to(dev), b['pids'], Bc, b['inc'], b['inc_masks'])
        # the shipped decode mode (relational memory in f32 unless EVK_DECODE_RM_F32=0) ...
        default_f32 = bool(DEC._RM_F32[0])
        hip_seq = DEC.beam_search(pm.text_decoder, xs, ms, dict(pm.args, beam_size=beam, max_seq_len=L), step_hook=watch).cpu()
        # ... and the same search in the OTHER mode of the recurrence, for comparison
        picks_def, saved = list(picks), DEC._RM_F32[0]
        del picks[:]
        DEC._RM_F32[0] = not default_f32
        DEC._SESSIONS.clear()
        try:
            hip_seq_alt = DEC.beam_search(pm.text_decoder, xs, ms, dict(pm.args, beam_size=beam, max_seq_len=L), step_hook=watch).cpu()
        finally:
            DEC._RM_F32[0] = saved
            DEC._SESSIONS.clear()
        picks_alt = list(picks)
    P = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu() for k, v in pm.state_dict().items() if not k.endswith('position_ids')}
    del pm
    cfg = dict(O.DEFAULT_CFG, max_seq_len=L, beam_size=beam)
    t0 = time.time()
    with torch.no_grad():
        x, m = O.finetune_encoder_states(P, b['images'], b['pids'], Bc, b['inc'], b['inc_masks'], cfg, O.Ctx())
        trace = []
        ref_seq = OB.beam_search(P, x, m, cfg, V - 2, V - 1, trace=trace)
    dt = time.time() - t0
    tok = load_tokenizer()
    same_seq, same_tok = metrics.token_agreement(hip_seq.tolist(), ref_seq.tolist())
    bl = metrics.bleu(OB.decode_texts(tok, ref_seq), OB.decode_texts(tok, hip_seq))
    base = dict(value=Bc * L / dt, unit='tokens/s', cores=threads, kind='port',
                sample='oracle beam search (full-prefix re-decode per step, as the reference) incl. encoder, beam %d, %d studies x %d views %dx%d, '
                       '%d steps, one run of %.1f s' % (beam, Bc, a.views, a.res, a.res, L, dt))
    def oracle_score(sq):          # teacher-forced fp32 log-probability the oracle gives a sequence (no [EOS] on these weights: all L tokens)
        ids = torch.cat([torch.full((sq.shape[0], 1), V - 2, dtype=sq.dtype), sq[:, :-1]], 1)
        with torch.no_grad():
            lp = O.r2_forward_logprobs(P, ids, x, torch.ones_like(ids), m, cfg, O.Ctx())
        return [round(float(v), 4) for v in lp.gather(2, sq.unsqueeze(-1)).squeeze(-1).sum(1)]
    sc_h, sc_r = oracle_score(hip_seq), oracle_score(ref_seq)
    pref = []
    for hs, rs in zip(hip_seq.tolist(), ref_seq.tolist()):
        k = 0
        while k < min(len(hs), len(rs)) and hs[k] == rs[k]:
            k += 1
        pref.append(k)
    # where do the two searches part, and how close was that decision for the REFERENCE arithmetic?  Both searches are replayed as sets of
    # hypotheses (token tuples); at the first position where the sets differ, the oracle's own scores of the candidates say how far
    # apart the last candidate it selected and the first one it rejected were (its decision margin) and where the engine's pick ranked
    V1 = V + 1

    def first_divergence(picks):
        first_div = []
        for s_ in range(Bc):
            eng, ref, rec_ = [()], [()], None
            for t in range(L):
                e_new = [eng[int(f) // V1] + (int(f) % V1,) for f in picks[t][s_]]
                flat = trace[t]['flat'][s_].tolist()
                r_all = [ref[f // V1] + (f % V1,) for f in flat]
                r_new = r_all[:beam]
                if set(e_new) != set(r_new):
                    sc = trace[t]['score'][s_].tolist()
                    rank_of = {h: i for i, h in enumerate(r_all)}
                    odd = [h for h in e_new if h not in set(r_new)]
                    ranks = [rank_of.get(h, -1) for h in odd]
                    gap = max([sc[beam - 1] - sc[r] for r in ranks if r >= 0] or [sc[beam - 1] - sc[-1]])
                    rec_ = dict(position=t, oracle_margin_selected_vs_rejected=round(sc[beam - 1] - sc[beam], 5),
                                oracle_score_gap_to_engine_pick=round(gap, 5), engine_pick_oracle_rank=ranks)
                    break
                eng, ref = e_new, r_new
            first_div.append(rec_)
        return first_div

    same_seq_alt, same_tok_alt = metrics.token_agreement(hip_seq_alt.tolist(), ref_seq.tolist())
    mode_name = lambda f32: 'relational memory in f32 (csrc/rm_f32.hip)' if f32 else '16-bit relational memory (EVK_DECODE_RM_F32=0)'      # noqa: E731
    par = dict(mode=mode_name(default_f32) + ' -- the shipped default, the mode `value` above was measured in',
               first_divergent_decision=first_divergence(picks_def), identical_sequences=same_seq, token_agreement=same_tok,
               bleu4_vs_oracle=bl[3], studies=Bc, common_prefix_tokens=pref,
               oracle_logprob_of_engine_sequences=sc_h, oracle_logprob_of_oracle_sequences=sc_r,
               other_mode=dict(mode=mode_name(not default_f32), first_divergent_decision=first_divergence(picks_alt),
                               identical_sequences=same_seq_alt, token_agreement=same_tok_alt),
               note='engine (16-bit operands; f32 relational-memory recurrence) vs CPU oracle (fp32) on the same inputs and the procedural weights of the golden '
                    'fixtures, beam %d, %d positions; an untrained network never emits [EOS] and its logit gaps are tiny (the oracle\'s own selection margin is '
                    'below 1e-2 at a tenth of the positions), so once one near-tie resolves differently the rest of the sequence differs -- '
                    'first_divergent_decision gives, per study, the position at which the two searches first select different hypothesis sets, '
                    'the oracle (fp32) margin between its last selected and first rejected candidate there, and the oracle score gap to what the engine '
                    'picked instead (None = the searches never part); tests/test_model_gpu.py holds the engine to |log-probability - reference| <= 8e-3 at '
                    'every position of every beam golden (teacher forced) and to the reference ids wherever its margins exceed twice that' % (beam, L))
    return base, par


def bench_decode(a, rank, world, dev):
    from evoke_amd.model_pretrain_finetune import FineTune
    from evoke_amd.config import load_default_tokenizer as load_tokenizer
    torch.manual_seed(9233)
    model = FineTune(make_args('test'), load_tokenizer(), 'mimic_cxr').to(dev)
    # a freshly initialised network in eval mode runs its 33 bottlenecks on batch-norm running statistics 0 / 1 and the un-normalised
    # activations leave fp16's range (the engine's forward guard raises, evoke_amd/ops.py: guard_finite): give the statistics the values a
    # trained model has -- 40 train-mode forward passes over the synthetic batch (the default bench line decodes with the model it has just
    # trained for warmup + steps iterations, which has the same effect)
    model.train()
    wb = synth_batch('finetune', min(a.decode_batch, 16), a.views, a.res, 100, 30, dev, 3000 + rank)
    with torch.no_grad():
        for _ in range(40):
            model.visual_extractor(wb['images'])
    model.eval()
    rec = decode_record(model, a, rank, world, dev, with_cpu=(world == 1 and not a.no_cpu_baseline))
    if rank == 0:
        rec.update(n_gpus=world, warmup=a.warmup, scaling='weak', vs_baseline=None, data='synthetic', ms_per_step=rec['ms_per_batch'])
        print(json.dumps(rec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='finetune', choices=['finetune', 'pretrain', 'decode'])
    ap.add_argument('--beam', type=int, default=4)
    ap.add_argument('--decode-batch', type=int, default=64, help='studies per GPU of the decode workload (BASELINE config 5: 64)')
    ap.add_argument('--no-decode', action='store_true', help='skip the decode sub-record of the default line')
    ap.add_argument('--res', type=int, default=384)
    ap.add_argument('--batch', type=int, default=32, help='studies per GPU')
    ap.add_argument('--views', default='2', help="views per study: an integer, or 'u4' = 1..4 per study (uniform, seed 7 + rank): config 4")
    ap.add_argument('--config', type=int, default=0, help='4: BASELINE config 4 = --workload pretrain --res 384 --batch 16 --views u4')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-launches', default='', help='write one CSV row per profiled launch (GEMM shapes + ms) to this path')
    ap.add_argument('--no-prof', action='store_true', help='disable the in-library HIP-event timing of kernel families')
    ap.add_argument('--no-side-streams', action='store_true', help='issue the whole step on one stream (experiment)')
    ap.add_argument('--graph', type=int, default=-1, help='1: capture the whole training step in a HIP graph, 0: eager launches, -1: default')
    a = ap.parse_args()
    if a.config == 4:
        a.workload, a.res, a.batch, a.views = 'pretrain', 384, 16, 'u4'
    a.views = a.views if a.views == 'u4' else int(a.views)

    from evoke_amd import distributed as D
    # EVK_DIST_BACKEND=gloo: a REHEARSAL of the N > 1 code path on a box with fewer GPUs than ranks (RCCL refuses two ranks per device): the
    # ranks share the devices there are (rank r on device r mod count) and gloo carries the collectives.  Its numbers mean nothing; the
    # point is that bench.py's multi-rank branch (barriers, max-over-ranks timing, reducer, comm statistics, replicated decode) has run.
    backend = os.environ.get('EVK_DIST_BACKEND') or None
    if backend == 'gloo':
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    rank, world, local = D.init_distributed(backend)
    if world != a.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run --nproc-per-node %d' % (world, a.gpus, a.gpus))
    if backend == 'gloo':
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from evoke_amd import hip as H, ops, optim
    from evoke_amd.model_pretrain_finetune import FineTune, Pretrain
    from evoke_amd.config import load_default_tokenizer as load_tokenizer
    torch.manual_seed(9233)
    ops.manual_seed(9233 + rank)
    if a.workload == 'decode':
        return bench_decode(a, rank, world, dev)
    args = make_args(a.workload)
    kind = a.workload
    L, Li = (100, 30) if kind == 'finetune' else (40, 0)
    model = (FineTune if kind == 'finetune' else Pretrain)(args, load_tokenizer(), 'mimic_cxr').to(dev)
    model.train()
    if kind == 'pretrain' and (world > 1 or D.forced()):
        model.gather = D.gather_rows
    opt = optim.build_two_stage_optimizer(args, model, clip_value=0.1)
    red = D.GradReducer.for_optimizer(opt)
    batch = synth_batch(kind, a.batch, a.views, a.res, L, Li, dev, 1000 + rank)

    def step_eager():
        ops.advance_seed_epoch()
        opt.zero_grad()
        red.begin(D.batch_structure(kind, batch['pids']))       # the structure key the trainer uses: ragged per-rank batches (config 4) differ in it
        if kind == 'finetune':
            ret = model(batch['images'], batch['ids'], batch['masks'], batch['pids'], batch['inc'], batch['inc_masks'], mode='train')
        else:
            ret = model(batch['images'], batch['ids'], batch['masks'], batch['pids'])
        loss = ret['all_loss']
        loss.backward()              # not pre-divided by world: all-reduce SUM, 1/world inside the optimizer kernel
        red.finish()
        opt.step()
        return loss.detach()

    # Step replay (evoke_amd/graph.py + csrc/replay.hip): the step is stream-captured once and its launch sequence re-issued from C++ on
    # lanes that ARE the capture's streams (round 5: a capture probe notes which stream created which node; the minimum path cover of
    # rounds 2-4 scrambled trunk forward + weight gradients into one lane and ran 71 ms).  Measured on one box (profiles/r05_replay_sweep.txt):
    # eager 48.6 ms (host issue 29 ms), replayed 49.2 ms at equal stream priorities (host issue 9 ms), 52.7 ms with the main lane at the
    # higher priority -- and 48.1 ms once the relational-memory lane (a chain of ~1400 dependent launches of a few microseconds) runs at the
    # higher priority too (evoke_amd/graph.py: LANE_PRIORITY; 224^2: 30.8 eager, 30.0 replayed).  The same priority on the EAGER step's
    # relational-memory stream makes that step 60 % slower (78 ms), so it is a property of the replay lanes only.
    # default: replayed on one rank (measured <= the eager step at every workload, profiles/r05_replay_sweep.txt + r05_stream_priorities.txt);
    # eager whenever a process group exists -- the bucketed RCCL calls are issued from inside the backward and are not part of a capture
    # (the capture needs three untimed calls: with fewer warm-up steps than that the default stays eager, so that exactly W warm-up steps run)
    use_graph = a.graph if a.graph >= 0 else (1 if (world == 1 and not D.forced() and a.warmup >= 3) else 0)
    step = step_eager
    if use_graph:
        from evoke_amd.graph import StepGraph
        step = StepGraph(step_eager, warmup=2, make_on_replay=opt.replay_hook)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Every stream of the step at the DEFAULT HIP priority.  (Rounds 2-4 gave the step's own stream the higher priority: +-0.2 ms for the eager
    # step on this round's boxes -- 48.0 vs 47.8 ms -- and 3-8 ms SLOWER for the replayed one.  The runtime gives each priority class hardware
    # queues of its own, and a fifth active queue is where the step's slow mode begins (profiles/r05_hw_queues.txt); with N > 1 the RCCL
    # stream is one more active queue, so the safe configuration keeps everything in one class.  EVK_MAIN_PRIO=1 restores the old behaviour.)
    main_stream = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get('EVK_MAIN_PRIO', '0') == '1' else torch.cuda.current_stream()
    main_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(main_stream)
    n_warm = max(a.warmup, 3) if use_graph else a.warmup          # capture happens on the 3rd call: keep it out of the timed steps
    # Guard of the replayed default (untimed, inside the warm-up): the replayed step has a slow mode when the runtime's hardware queues are
    # laid out badly (72-111 ms instead of 47, profiles/r05_hw_queues.txt).  The shipped configuration has not shown it in any run, but a
    # benchmark must not depend on that: each warm-up step is timed on its own (a device synchronisation on both sides), and if the
    # replayed steps (call 4 onwards) take more than 1.25 x the eager ones (calls 1-2) the timed region runs the eager step instead.
    graph_guard = None
    warm_ms = []
    for _ in range(n_warm):
        if use_graph:
            torch.cuda.synchronize()
            tw = time.perf_counter()
        step()
        if use_graph:
            torch.cuda.synchronize()
            warm_ms.append(1e3 * (time.perf_counter() - tw))
    if use_graph and len(warm_ms) >= 4 and getattr(step, 'graph', None) is not None:
        eager_ms, replay_ms = min(warm_ms[:2]), min(warm_ms[3:])
        graph_guard = {'eager_warmup_step_ms': eager_ms, 'replayed_warmup_step_ms': replay_ms, 'kept_replay': bool(replay_ms <= 1.25 * eager_ms)}
        if not graph_guard['kept_replay']:
            step, use_graph = step_eager, 0
    barrier()
    skipped0 = ops.overflow_steps(dev)
    red.timing = bool(red.active)          # events around the collectives (a process group exists: N > 1, or the EVK_FORCE_DIST=1 rehearsal)
    t0 = time.perf_counter()
    losses = []
    for i in range(a.steps):
        losses.append(step())
    host_dt = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    comm = red.comm_stats(a.steps) if red.timing else None
    if comm is not None:
        comm['backend'] = torch.distributed.get_backend()
    red.timing = False
    skipped = ops.overflow_steps(dev) - skipped0
    # host cost of ISSUING one step, measured from an idle GPU (inside the timed loop the launch queue is full whenever the GPU is
    # the bottleneck, so the loop's host time mostly measures the GPU): two more steps, each issued after a full synchronisation
    issue = []
    for _ in range(2):
        barrier()
        t1 = time.perf_counter()
        step()
        issue.append(time.perf_counter() - t1)
    barrier()
    host_issue_ms = 1e3 * min(issue)
    fam, launched_flops = ({}, 0.0)
    if not a.no_prof:
        # Per-kernel durations for the roofline: ONE more step of the same workload right after the timed region, with a
        # HIP-event pair around every launch (on the launch's own stream) and the side streams folded into the main one --
        # concurrent kernels share the CUs, which would inflate every event-timed duration, and the ~6k event records
        # cost ~15 ms of host time that must not sit inside the timed steps.
        # The step is enqueued behind a bounded GPU-side spin (torch.cuda._sleep, ~1.5 steps long) so that every launch and
        # event is already queued when the GPU reaches it: the event pairs then time kernels, not host launch latency.
        ops.SIDE_STREAMS_ENABLED[0] = False
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(20_000_000)
        e1.record()
        torch.cuda.synchronize()
        cyc_per_ms = 20_000_000 / max(e0.elapsed_time(e1), 1e-3)
        torch.cuda._sleep(int(cyc_per_ms * min(400.0, 1.5e3 * dt / a.steps + 30.0)))
        H.prof_enable(True)
        step_eager()
        barrier()
        ops.SIDE_STREAMS_ENABLED[0] = True
        if a.dump_launches:
            H._dump_path = a.dump_launches.encode()
            H.check(H.lib.evk_prof_dump_to(H._dump_path))
        fam, launched_flops = H.prof_collect()
        H.prof_enable(False)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt = float(tt.item())
    # BASELINE.json's metric is two numbers: the train-step studies/s above and decode tokens/s (config 5) -- measured on the same
    # model object right after the timed training region (every rank: replicas)
    dec_rec = None
    if kind == 'finetune' and not a.no_decode:
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        try:
            dec_rec = decode_record(model, a, rank, world, dev, with_cpu=(world == 1 and not a.no_cpu_baseline))
        except Exception as e:          # noqa: BLE001 -- the training measurement must still be reported
            dec_rec = {'metric': 'decode tokens/sec', 'value': None, 'error': repr(e)}
        model.train()
    if rank != 0:
        return
    studies = a.batch * world * a.steps
    out = {
        'metric': 'studies/sec (train step, %s-view %d^2)' % (a.views, a.res), 'value': studies / dt, 'unit': 'studies/s', 'n_gpus': world,
        'steps': a.steps, 'warmup': n_warm, 'ms_per_step': 1e3 * dt / a.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': H.STORE, 'data': 'synthetic',
        'config': {'workload': 'EVOKE-%d %s-view %s train step (fwd+bwd+allreduce+clip+RAdam), %d studies (%d images) per GPU, '
                               'L=%d, Li=%d, V=%d, random-init weights' % (a.res, a.views, kind, a.batch, int(batch['images'].shape[0]), L, Li, V),
                   'parallelism': 'dp%d' % world, 'rccl_ranks': D.world_size(), 'backend': (torch.distributed.get_backend() if D.world_size() > 1 or D.forced() else None),
                   'grad_sync': (red.mode + (' (EVK_FORCE_DIST=1: 1-rank process group, every collective issued)' if D.forced() and world == 1 else ''))
                   if red.active else 'none (1 rank, no process group)',
                   'comm': comm, 'loss_last': float(losses[-1].item()),
                   # a dynamic-loss-scale overflow skips the optimizer update of that step (fp16 storage): steps of the timed region it skipped
                   'overflow_skipped_steps': skipped, 'loss_scale': ops.loss_scale_value(dev),
                   'batches': 'one synthetic batch per rank, fed every step',
                   'host_launch_ms_per_step': host_issue_ms, 'host_loop_ms_per_step': 1e3 * host_dt / a.steps,
                   'step_graph': bool(use_graph and getattr(step, 'graph', None) is not None), 'step_graph_guard': graph_guard,
                   'step_replay_plan': getattr(step, 'info', None)},
    }
    if fam:
        ms, n = fam['gemm']
        alg = ALG_GFLOP_PER_STUDY.get((kind, a.res))
        psteps = 1                                   # the extra, event-timed step
        alg_flops = (alg * 1e9 * a.batch * psteps) if alg else launched_flops
        ach = alg_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM bytes of the family per step: rocprofv3 PMC passes of THIS command (tools/traffic_json.py), valid only while the
        # committed file carries the fingerprint of the sources that are running; null otherwise (never a stale constant)
        traffic, tsrc = None, None
        from evoke_amd.build import source_fingerprint
        for tname in sorted((f for f in os.listdir(os.path.join(REPO, 'profiles')) if f.endswith('traffic.json')), reverse=True):
            tj = json.load(open(os.path.join(REPO, 'profiles', tname)))
            if (tj.get('fingerprint') == source_fingerprint() and tj.get('workload') == [kind, a.res, a.batch, a.views]
                    and tj.get('store') == H.STORE):
                traffic, tsrc = tj.get('gemm_family_hbm_bytes_per_step'), 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per step)' % tname
                break
        out['roofline'] = {'bound': 'mfma', 'kernel': 'MFMA GEMM / convolution family (gemm_kernel tiles, halo-tile 3x3 kernels, strip GEMM, weight-stationary 1x1, attention, relational memory)', 'achieved': ach,
                           'peak': MFMA_BF16_DENSE_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / MFMA_BF16_DENSE_PEAK_TFLOPS,
                           'traffic': traffic, 'traffic_source': tsrc,
                           'launches_per_step': n / psteps, 'event_timed': '1 extra step after the timed region, single stream', 'avg_launch_us': 1e3 * ms / max(n, 1),
                           'gemm_ms_per_step': ms / psteps, 'launched_tflops': launched_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                           'family_ms_per_step': {k: v[0] / psteps for k, v in fam.items()}}
    if kind == 'finetune' and not a.no_decode and dec_rec is not None:
        out['decode'] = dec_rec
    if world == 1 and not a.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline(model, kind, a.res, L, Li)
        except Exception as e:          # the GPU measurement must still be reported
            out['cpu_baseline'] = {'value': None, 'unit': 'studies/s', 'cores': os.cpu_count(), 'kind': 'port', 'sample': 'failed: %r' % (e,)}
    print(json.dumps(out))


if __name__ == '__main__':
    # ONE JSON line on stdout: libraries that write banners to fd 1 (RCCL prints its version block there at communicator creation)
    # are sent to stderr for the duration of the run; the real stdout is restored for the result line
    sys.stdout.flush()
    _real_stdout = os.dup(1)
    os.dup2(2, 1)
    _print = print

    def print(*a, **k):          # noqa: A001 -- module-level override used by main() / bench_decode() for the result line
        if k.get('file') is None:
            sys.stdout.flush()
            os.dup2(_real_stdout, 1)
            _print(*a, **k)
            sys.stdout.flush()
            os.dup2(2, 1)
        else:
            _print(*a, **k)
    try:
        main()
    finally:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
