"""Data-parallel orchestration (bayes_sim_ig_amd/dp.py) on CPU with the gloo
backend, world_size 2: the gradient exchange (through the C ABI's communicator,
bsig_comm_allreduce / _broadcast over a gloo-backed exchange, host buffers), loss
reduction and shard bounds are exercised with the oracle as the local gradient
engine (the HIP engine needs a GPU).  Property: R ranks with B/R rows each == one rank with the B
rows (same weights after every update, up to fp32 summation order)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, ROOT)
    from bayes_sim_ig_amd import dp
    from oracle import estimators as oest
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(0)
    n, i, d, b, updates = 64, 12, 3, 16, 6
    x, y = torch.randn(n, i, generator=gen), torch.rand(n, d, generator=gen)
    ids = np.random.RandomState(1).randint(0, n, (updates, b))
    kw = dict(input_dim=i, output_dim=d, output_lows=None, output_highs=None, n_gaussians=3,
              full_covariance=False, hidden_layers=(8,), activation=torch.nn.Tanh, lr=1e-2,
              eps_noise=0.0)
    torch.manual_seed(100 + rank)               # different init: broadcast must fix it
    model = oest.OracleMDNN(**kw)
    params = [p for p in model.parameters()]
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    group = dp.DataParallel().init_comm()     # gloo group: the group's collectives behind bsig_comm_*
    assert group.transport == 'torch' and group.comm is not None
    group.broadcast(flat)
    off = 0
    with torch.no_grad():
        for p in params:
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
    opt = torch.optim.Adam(params, lr=1e-2)
    flat_grad = torch.zeros_like(flat)
    lo, hi = dp.shard_bounds(b, world, rank)
    state = {'it': 0, 'loss': []}

    def grad():
        rows = ids[state['it']][lo:hi]
        opt.zero_grad()
        loss = model.mdn_loss_fn(*model(x[rows]), y[rows]) * (hi - lo) / b   # 1/B_global
        loss.backward()
        flat_grad.copy_(torch.cat([p.grad.reshape(-1) for p in params]))
        state['loss'].append(float(loss) * b / (hi - lo))

    def apply():
        o = 0
        for p in params:
            p.grad.copy_(flat_grad[o:o + p.numel()].view_as(p))
            o += p.numel()
        opt.step()
        state['it'] += 1

    evals = []
    dp.run_updates(updates, {0, updates - 1}, grad,
                   lambda: group.allreduce_sum(flat_grad), apply,
                   lambda: evals.append(float(model.mdn_loss_fn(*model(x[:8 + 4 * rank]),
                                                                y[:8 + 4 * rank]))))
    tl, te = group.mean_losses(torch.tensor(state['loss']), torch.tensor(evals), 8 + 4 * rank)
    if rank == 0:
        torch.save({'flat': torch.cat([p.detach().reshape(-1) for p in params]),
                    'train': tl, 'test': te, 'evals0': evals}, out)
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_equals_single_rank(tmp_path):
    from oracle import estimators as oest
    out = str(tmp_path / 'dp.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    # single-process reference with the same global batch
    gen = torch.Generator().manual_seed(0)
    n, i, d, b, updates = 64, 12, 3, 16, 6
    x, y = torch.randn(n, i, generator=gen), torch.rand(n, d, generator=gen)
    ids = np.random.RandomState(1).randint(0, n, (updates, b))
    torch.manual_seed(100)                      # rank 0's init
    model = oest.OracleMDNN(input_dim=i, output_dim=d, output_lows=None, output_highs=None,
                            n_gaussians=3, full_covariance=False, hidden_layers=(8,),
                            activation=torch.nn.Tanh, lr=1e-2, eps_noise=0.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for it in range(updates):
        opt.zero_grad()
        model.mdn_loss_fn(*model(x[ids[it]]), y[ids[it]]).backward()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(res['flat'], flat, rtol=1e-4, atol=1e-6)
    # count-weighted held-out mean over the two shards (8 and 12 rows)
    assert res['test'].shape == (2,) and res['train'].shape == (updates,)
    assert torch.isfinite(res['test']).all()


def test_shard_bounds_cover_everything():
    from bayes_sim_ig_amd import dp
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
