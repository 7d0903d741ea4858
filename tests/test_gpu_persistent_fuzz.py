"""Randomised shapes through the persistent MDNN kernel and its variants (factor rows, wide
heads, full covariance, narrow zero-padded trunks) against the per-phase kernels on the same
chunk: same jitter draws, the arithmetic differs only in summation order."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
_ENV = ('BSIG_NO_PERSISTENT', 'BSIG_MDNN_WIDE_HEADS', 'BSIG_NO_TRUNK_PAD')


@pytest.fixture(scope='module')
def B():
    import bayes_sim_ig_amd as pkg
    pkg._lib.require_gpu()
    pkg.MDNN.VERBOSE = False
    return pkg


@pytest.fixture(autouse=True)
def _guards():
    import bayes_sim_ig_amd as pkg
    old = pkg.MDNN.EPS_NOISE
    yield
    pkg.MDNN.EPS_NOISE = old
    for k in _ENV:
        os.environ.pop(k, None)


def _case(seed):
    r = np.random.RandomState(seed)
    full = bool(r.randint(2))
    c = dict(d=int(r.randint(1, 9 if full else 13)), k=int(r.randint(1, 13)), full=full,
             summarizer=['summary_start', 'summary_corrdiff', 'summary_corr'][r.randint(3)],
             sd=int(r.randint(2, 13)), ad=int(r.randint(1, 5)), t=int(r.randint(3, 16)),
             batch=int(r.randint(1, 105)), n_updates=int(r.randint(1, 21)),
             hidden=[(128, 128), (128, 128), (24, 24), (7, 128), (128, 33)][r.randint(5)],
             lazy=bool(r.randint(2)), eps=[0.0, 1e-5][r.randint(2)], wide=bool(r.randint(2)))
    c['n'] = int(r.randint(max(c['batch'] // 2, 5), 700))
    return c


def _run(B, c, persistent):
    import bench
    for k in _ENV:
        os.environ.pop(k, None)
    if not persistent:
        os.environ['BSIG_NO_PERSISTENT'] = '1'
    elif c['wide']:
        os.environ['BSIG_MDNN_WIDE_HEADS'] = '1'
    B.MDNN.EPS_NOISE = c['eps']
    cfg = dict(task='fuzz', model='MDNN', summarizer=c['summarizer'], t=c['t'], sd=c['sd'], ad=c['ad'],
               d=c['d'], k=c['k'], hidden=list(c['hidden']), n_feat=0, pairs=c['n'], full=c['full'])
    theta, states, actions = bench.synth_pairs(cfg, c['n'], 3, DEV)
    torch.manual_seed(5)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    n_train = max(int(c['n'] * 0.8), 1)
    ids = np.random.RandomState(5).randint(0, n_train, (c['n_updates'], c['batch']))
    lazy = c['lazy'] and c['summarizer'] != 'summary_start'
    summ = bs._summarize(states, actions, lazy=lazy)
    torch.manual_seed(6)
    logs = bs.model.run_training(summ, theta, c['n_updates'], c['batch'], ids_table=ids)
    return logs, bs.model.state_dict(), int(B._lib.load().bsig_fit_is_persistent(bs.model._plan))


@pytest.mark.parametrize('seed', list(range(24)))
def test_random_shapes_persistent_equals_phase_kernels(B, seed):
    c = _case(seed)
    la, wa, pa = _run(B, c, True)
    lb, wb, pb = _run(B, c, False)
    assert pa == 2 and pb == 0, c
    for key in ('train_loss', 'test_loss'):
        assert len(la[key]) == len(lb[key]), c
        assert np.allclose(la[key], lb[key], rtol=1e-4, atol=1e-4, equal_nan=True), (c, key, la[key], lb[key])
    for (k1, v1), (k2, v2) in zip(wa.items(), wb.items()):
        assert k1 == k2 and v1.shape == v2.shape
        assert torch.allclose(v1, v2, atol=2e-4, rtol=2e-3), (c, k1, float((v1 - v2).abs().max()))
