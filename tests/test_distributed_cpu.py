"""CPU (gloo, world_size 2) tests of the data-parallel layer: the autograd all-gather used for the cross-rank
contrastive negatives and the bucketed flat-gradient reducer (evoke_amd/distributed.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from evoke_amd import distributed as D, ops
    from oracle import functional as O
    D.init_distributed('gloo')
    try:
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(3, 16, generator=g), torch.randn(2, 16, generator=g)]          # ragged rows per rank
        pids = [np.array(['a', 'b', 'c']), np.array(['a', 'c'])]
        W0 = torch.randn(16, 16, generator=g) * 0.3
        b0 = torch.randn(16, generator=g) * 0.1

        # ---- distributed: local projection, gathered global loss (+ a local term), mean all-reduce of flat grads
        flat = torch.zeros(16 * 16 + 16)
        W = W0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        red = D.GradReducer([flat], [[(W, 0, 256), (b, 256, 16)]], bucket_bytes=512)
        y = xs[rank] @ W + b
        Y, ids = D.gather_rows(y, pids[rank])
        loss = O.multi_pos_contra_images(Y, ids, 0.5) + y.pow(2).mean()
        (loss / world).backward()
        flat[:256] = W.grad.reshape(-1)
        flat[256:] = b.grad
        red.on_grad(W)
        red.on_grad(b)
        red.finish()

        # ---- single process on the concatenated batch
        Wr = W0.clone().requires_grad_(True)
        br = b0.clone().requires_grad_(True)
        yr = torch.cat(xs) @ Wr + br
        lr = O.multi_pos_contra_images(yr, np.concatenate(pids), 0.5) + sum((xs[r] @ Wr + br).pow(2).mean() for r in range(world)) / world
        lr.backward()
        ok_ids = len(set(ids.tolist())) == 3 and ids[0] == ids[3] and ids[2] == ids[4]
        err_w = (flat[:256].reshape(16, 16) - Wr.grad).abs().max().item()
        err_b = (flat[256:] - br.grad).abs().max().item()
        # second step exercises the learned-count overlap path
        flat.zero_()
        red.begin('default')
        flat[:256] = 1.0
        red.on_grad(W)
        flat[256:] = 2.0
        red.on_grad(b)
        red.finish()
        ok2 = bool((flat[:256] == world).all() and (flat[256:] == 2.0 * world).all())
        q.put((rank, ok_ids, err_w, err_b, ok2, len(red.buckets)))
    finally:
        dist.destroy_process_group()
        ops.clear_grad_callbacks()


def test_allgather_autograd_and_reducer_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_ids, err_w, err_b, ok2, nb in res:
        assert ok_ids, 'study-id hashes do not line up across ranks'
        assert err_w < 1e-5 and err_b < 1e-5, (rank, err_w, err_b)
        assert ok2
        assert nb == 2


def test_reducer_single_process_is_identity():
    from evoke_amd import distributed as D, ops
    flat = torch.arange(10, dtype=torch.float32)
    p = torch.nn.Parameter(torch.zeros(10))
    red = D.GradReducer([flat], [[(p, 0, 10)]])
    red.on_grad(p)
    red.finish()
    assert torch.equal(flat, torch.arange(10, dtype=torch.float32))
    assert D.gather_rows(torch.ones(2, 3), ['x', 'y'])[0].shape == (2, 3)
    ops.clear_grad_callbacks()


def _order_worker(rank, world, port, q, mode):
    """Two ranks whose batches differ in STRUCTURE: rank 0's studies have sibling views (the multi-view attention parameters receive
    gradients), rank 1's have none (they receive nothing); gradients also arrive in a different order on the two ranks.  The reducer
    must issue the same collective sequence on both (descending bucket index), must not hang, and every bucket must hold the sum of
    the shards."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from evoke_amd import distributed as D, ops
    D.init_distributed('gloo')
    try:
        n = 4096
        names = ['trunk', 'multiview_attention', 'fusion', 'decoder']
        ps = [torch.nn.Parameter(torch.zeros(n)) for _ in names]
        flat = torch.zeros(4 * n)
        red = D.GradReducer([flat], [[(p, i * n, n) for i, p in enumerate(ps)]], bucket_bytes=4 * n, mode=mode)
        assert len(red.buckets) == 4
        out = []
        for step in range(3):
            pids = ['a', 'b', 'a'] if rank == 0 else ['c', 'd', 'e']          # rank 1: no study has two views
            key = D.batch_structure('inc', pids)
            red.begin(key)
            flat.zero_()
            # backward order: decoder, fusion, (multi-view attention only with siblings), trunk -- rank 1 sees fusion before decoder
            order = [3, 2, 1, 0] if rank == 0 else [2, 3, 0]
            for i in order:
                flat[i * n:(i + 1) * n] = float((rank + 1) * (i + 1) * (step + 1))
                red.on_grad(ps[i])
            issued_before_finish = list(red.issued)
            red.finish()
            want = [sum((r + 1) * (i + 1) * (step + 1) for r in range(world) if not (r == 1 and i == 1)) for i in range(4)]
            got = [float(flat[i * n]) for i in range(4)]
            same = all(bool((flat[i * n:(i + 1) * n] == flat[i * n]).all()) for i in range(4))
            out.append((key, issued_before_finish, got, want, same))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()
        ops.clear_grad_callbacks()


def _run_order(mode):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_reducer_fixed_order_with_different_batch_structures_gloo():
    res = _run_order('allreduce')
    assert res[0][0][0] == ('inc', True) and res[1][0][0] == ('inc', False)         # the structure keys differ across ranks
    for rank in (0, 1):
        for step, (key, early, got, want, same) in enumerate(res[rank]):
            assert same and got == want, (rank, step, got, want)
            assert early == sorted(early, reverse=True), early                      # never bucket b before bucket b + 1
            if step == 0:
                assert early == []                                                  # first step of a structure: learned, reduced at finish()
    # once the counts are learned both ranks overlap all four buckets with their backward, in the same order: rank 1 has learned, for
    # ITS structure, that bucket 1 (multi-view attention) receives nothing and sends its zeros as soon as buckets 3 and 2 have gone out;
    # its early 'fusion' gradient (bucket 2 before bucket 3) waits for bucket 3
    assert res[0][2][1] == [3, 2, 1, 0]
    assert res[1][2][1] == [3, 2, 1, 0]


def test_reducer_direct_and_16bit_sync_modes_gloo():
    for mode in ('direct', '16bit'):
        res = _run_order(mode)
        for rank in (0, 1):
            for step, (key, early, got, want, same) in enumerate(res[rank]):
                assert same and got == want, (mode, rank, step, got, want)


def _union_worker(rank, world, port, q):
    """the optimizer's update mask must be the UNION over ranks of the parameters that received gradients (a rank whose
    shard has no sibling views holds, after the sum, the other rank's multi-view gradients and must apply them too)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from evoke_amd import distributed as D, ops, optim
    D.init_distributed('gloo')
    try:
        names = ['trunk', 'multiview_cross_attention', 'pooler', 'text_decoder.a', 'visual_self_atten_layers.b']
        ps = [torch.nn.Parameter(torch.zeros(64)) for _ in names]
        opt = optim.FusedOptimizer([(1e-3, list(zip(names[:3], ps[:3]))), (1e-2, list(zip(names[3:], ps[3:])))])
        red = D.GradReducer.for_optimizer(opt, bucket_bytes=256)
        assert opt.world == world
        # rank 0: siblings + indication batch -> multi-view attention and the decoder; rank 1: no siblings, no indication -> self-attention
        # branch and the decoder.  Nobody touches the pooler.
        mine = [0, 1, 3] if rank == 0 else [0, 3, 4]
        red.begin(D.batch_structure('inc' if rank == 0 else 'no_inc', ['a', 'a'] if rank == 0 else ['b', 'c']))
        for i in mine:
            ps[i].grad.fill_(float(rank + 1))
            opt._on_grad(ps[i])
            red.on_grad(ps[i])
        red.finish()
        union = red.touched_union.tolist()
        sums = [float(p.grad[0]) for p in ps]
        q.put((rank, union, opt._global_touched is red.touched_union, sums))
    finally:
        dist.destroy_process_group()
        ops.clear_grad_callbacks()


def test_update_mask_is_the_union_over_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_union_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, union, handed_over, sums in res:
        assert union == [1, 1, 0, 1, 1], (rank, union)             # identical on both ranks; the pooler stays untouched everywhere
        assert handed_over
        assert sums == [3.0, 1.0, 0.0, 3.0, 2.0], (rank, sums)     # the sum of the shards, wherever a shard produced a gradient
