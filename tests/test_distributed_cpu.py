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
