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


# ---- the persistent kernel of the linear heads (MDRFF; csrc/fit_persistent.hip): its tiling picks
#      16- or 32-row tiles, k-slices of 96 / 192 / 288 columns, row owners on workgroups of their own
#      (one or two rows each) or on tile workgroups, evaluation owners with 1..8 rows -- random shapes
#      walk through those corners
def _rff_case(seed):
    r = np.random.RandomState(1000 + seed)
    c = dict(d=int(r.randint(1, 41)), k=int(r.randint(1, 13)),
             n_feat=int([200, 500, 512, 1024, 2048, 4096, 96, 300][r.randint(8)]),
             sd=int(r.randint(2, 9)), ad=int(r.randint(1, 4)), t=int(r.randint(3, 13)),
             batch=int(r.randint(1, 113)), n_updates=int(r.randint(1, 25)),
             eps=[0.0, 1e-5][r.randint(2)])
    # (the row kernel keeps at most 8 elements per lane: D <= 8 * (64 // K))
    c['d'] = min(c['d'], 8 * (64 // c['k']))
    c['n'] = int(r.randint(max(c['batch'] // 2, 5), 900))
    return c


def _rff_run(B, c, persistent):
    import bench
    for k in _ENV:
        os.environ.pop(k, None)
    if not persistent:
        os.environ['BSIG_NO_PERSISTENT'] = '1'
    B.MDNN.EPS_NOISE = c['eps']
    cfg = dict(task='fuzz', model='MDRFF', summarizer='summary_start', t=c['t'], sd=c['sd'], ad=c['ad'],
               d=c['d'], k=c['k'], hidden=[], n_feat=c['n_feat'], pairs=c['n'])
    theta, states, actions = bench.synth_pairs(cfg, c['n'], 3, DEV)
    torch.manual_seed(5)
    np.random.seed(5)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    n_train = max(int(c['n'] * 0.8), 1)
    ids = np.random.RandomState(5).randint(0, n_train, (c['n_updates'], c['batch']))
    summ = bs._summarize(states, actions)
    torch.manual_seed(6)
    logs = bs.model.run_training(summ, theta, c['n_updates'], c['batch'], ids_table=ids)
    return logs, bs.model.state_dict(), int(B._lib.load().bsig_fit_is_persistent(bs.model._plan))


@pytest.mark.parametrize('seed', list(range(20)))
def test_random_rff_shapes_persistent_equals_phase_kernels(B, seed):
    import ctypes as C
    c = _rff_case(seed)
    geo = (C.c_int32 * 16)()
    covered = B._lib.load().bsig_debug_persist_geometry(c['batch'], c['n_feat'], c['d'], c['k'],
                                                        c['n'] - max(int(c['n'] * 0.8), 1), geo)
    la, wa, pa = _rff_run(B, c, True)
    lb, wb, pb = _rff_run(B, c, False)
    assert pb == 0, c
    if covered:
        assert pa == 1, (c, list(geo))
    for key in ('train_loss', 'test_loss'):
        assert len(la[key]) == len(lb[key]), c
        assert np.allclose(la[key], lb[key], rtol=1e-4, atol=1e-4, equal_nan=True), (c, list(geo), key, la[key], lb[key])
    for (k1, v1), (k2, v2) in zip(wa.items(), wb.items()):
        assert k1 == k2 and v1.shape == v2.shape
        assert torch.allclose(v1, v2, atol=2e-4, rtol=2e-3), (c, list(geo), k1, float((v1 - v2).abs().max()))


def test_rff_shapes_with_owners_on_tile_workgroups():
    """The layout for heads that leave no CUs to spare -- the row owners are ALSO tile workgroups --
    is forced (BSIG_PERSIST_MIXED=1, read once per process: a fresh process) on three of the random
    shapes and on the ShadowHand head, against the per-phase kernels."""
    import subprocess
    import sys
    code = '''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch, ctypes as C
import bayes_sim_ig_amd as B
import test_gpu_persistent_fuzz as F
B._lib.require_gpu(); B.MDNN.VERBOSE = False
cases = [F._rff_case(s) for s in (5, 11, 15)]
cases.append(dict(d=32, k=4, n_feat=4096, sd=5, ad=2, t=11, batch=100, n_updates=12, eps=1e-5, n=600))
for c in cases:
    geo = (C.c_int32 * 16)()
    assert B._lib.load().bsig_debug_persist_geometry(c['batch'], c['n_feat'], c['d'], c['k'], 200, geo)
    assert geo[12] > 0, ('owners on tile workgroups expected', list(geo))
    la, wa, pa = F._rff_run(B, c, True)
    lb, wb, pb = F._rff_run(B, c, False)
    assert pa == 1 and pb == 0, c
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(la[key], lb[key], rtol=1e-4, atol=1e-4, equal_nan=True), (c, key, la[key], lb[key])
    for (k1, v1), (k2, v2) in zip(wa.items(), wb.items()):
        assert torch.allclose(v1, v2, atol=2e-4, rtol=2e-3), (c, k1, float((v1 - v2).abs().max()))
print('mixed layout ok')
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BSIG_PERSIST_MIXED='1')
    res = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'mixed layout ok' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
