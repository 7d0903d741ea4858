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


def _worker(rank, world, port, out, timeout_rank=-1):
    """timeout_rank >= 0: that rank's FIRST attempt at the call reports the persistent kernels' time-out bit
    (bit 1 of the flag word) in its logs -- the way a resident launch whose bounded polls gave up does.
    The logs of a call are summed over the ranks (bsig_fit_run_dp), so EVERY rank reads a non-zero flag
    at the same call: all restore the snapshot they took at its start and repeat it together."""
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
    import copy
    snap = (copy.deepcopy(model.state_dict()), copy.deepcopy(opt.state_dict()))
    attempts = 0

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

    while True:
        attempts += 1
        evals = []
        dp.run_updates(updates, {0, updates - 1}, grad,
                       lambda: group.allreduce_sum(flat_grad), apply,
                       lambda: evals.append(float(model.mdn_loss_fn(*model(x[:8 + 4 * rank]),
                                                                    y[:8 + 4 * rank]))))
        # the call's logs travel as one packed sum over the ranks, the flag word with them
        flag = torch.tensor([2.0 if (rank == timeout_rank and attempts == 1) else 0.0])
        group.allreduce_sum(flag)
        if float(flag) == 0.0:
            break
        # group-wide restore: every rank saw the flag, every rank goes back to the call's start
        assert attempts == 1
        model.load_state_dict(snap[0])
        opt.load_state_dict(snap[1])
        state['it'], state['loss'] = 0, []
    tl, te = group.mean_losses(torch.tensor(state['loss']), torch.tensor(evals), 8 + 4 * rank)
    counts = group.gather_counts(8 + 4 * rank)
    if rank == 0:
        torch.save({'flat': torch.cat([p.detach().reshape(-1) for p in params]),
                    'train': tl, 'test': te, 'evals0': evals, 'attempts': attempts, 'counts': counts}, out)
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


def test_four_ranks_unequal_eval_shards_and_a_group_wide_restore(tmp_path):
    """world_size 4 (minibatch 16 = 4 rows per rank), held-out shards of 8 / 12 / 16 / 20 rows, and rank 2
    reporting a persistent-kernel time-out on the first attempt: every rank must repeat the call from
    its snapshot, and the result must be the undisturbed single-rank run."""
    from oracle import estimators as oest
    out = str(tmp_path / 'dp4.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(4, port, out, 2), nprocs=4, join=True)
    res = torch.load(out)
    assert res['attempts'] == 2 and res['counts'] == [8, 12, 16, 20]
    gen = torch.Generator().manual_seed(0)
    n, i, d, b, updates = 64, 12, 3, 16, 6
    x, y = torch.randn(n, i, generator=gen), torch.rand(n, d, generator=gen)
    ids = np.random.RandomState(1).randint(0, n, (updates, b))
    torch.manual_seed(100)                      # rank 0's init
    model = oest.OracleMDNN(input_dim=i, output_dim=d, output_lows=None, output_highs=None,
                            n_gaussians=3, full_covariance=False, hidden_layers=(8,),
                            activation=torch.nn.Tanh, lr=1e-2, eps_noise=0.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    evals = []
    for it in range(updates):
        opt.zero_grad()
        model.mdn_loss_fn(*model(x[ids[it]]), y[ids[it]]).backward()
        opt.step()
        if it in (0, updates - 1):
            # the group's held-out NLL: count-weighted mean over the four shards x[:8], x[:12], x[:16], x[:20]
            tot = sum(float(model.mdn_loss_fn(*model(x[:m]), y[:m])) * m for m in (8, 12, 16, 20))
            evals.append(tot / 56.0)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(res['flat'], flat, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(res['test'], torch.tensor(evals), rtol=1e-4, atol=1e-6)
    assert res['train'].shape == (updates,)


def test_shard_bounds_cover_everything():
    from bayes_sim_ig_amd import dp
    for n in (0, 1, 7, 100, 1001):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
