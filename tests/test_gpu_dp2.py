"""Two-rank data-parallel fit on the HIP engine (two processes sharing the one
GPU of the test box, gloo as the exchange backend since RCCL refuses two ranks
on one device): sharded minibatches + summed flat gradients must reproduce the
single-process fit on the union minibatch."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

N, I, D, K, B, NU = 120, 130, 3, 4, 24, 12      # per rank: 120 pairs, minibatch 24
N_TRAIN = 96


def _data(rank):
    gen = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(N, I, generator=gen)
    y = torch.rand(N, D, generator=gen)
    ids = np.random.RandomState(200 + rank).randint(0, N_TRAIN, (NU, B))
    return x, y, ids


def _model(pkg, n_feat, eps=0.0):
    torch.manual_seed(3)
    np.random.seed(3)
    pkg.MDNN.VERBOSE = False
    pkg.MDNN.EPS_NOISE = eps
    if n_feat == 'mdnn':
        return pkg.MDNN(input_dim=I, output_dim=D, output_lows=np.zeros(D),
                        output_highs=np.ones(D), n_gaussians=K, lr=2e-3,
                        activation=torch.nn.Tanh, full_covariance=False,
                        hidden_layers=(128, 128), device='cuda:0')
    return pkg.MDRFF(input_dim=I, output_dim=D, output_lows=np.zeros(D), output_highs=np.ones(D),
                     n_gaussians=K, lr=2e-3, activation=torch.nn.Tanh, full_covariance=False,
                     n_feat=n_feat, sigma=3.0, device='cuda:0')


def _worker(rank, world, port, out, n_feat, eps=0.0):
    import sys
    sys.path.insert(0, ROOT)
    import bayes_sim_ig_amd as pkg
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    m = _model(pkg, n_feat, eps).enable_data_parallel()
    x, y, ids = _data(rank)
    logs = m.run_training(x.cuda(), y.cuda(), NU, B, test_frac=0.2, ids_table=ids)
    if rank == 0:
        torch.save({'logs': logs, 'flat': m._flat.cpu(),
                    'persistent': int(pkg._lib.load().bsig_fit_is_persistent(m._plan))}, out)
    dist.barrier()
    dist.destroy_process_group()


# 50 features (not a multiple of 4): per-phase kernels; 512: every update one launch of the persistent
# kernel per rank (gradients out, Adam step of the reduced gradients in); 'mdnn': the same
# with the persistent kernel of the two-layer MDNN
@pytest.mark.parametrize('n_feat', [50, 512, 'mdnn'])
def test_two_rank_fit_equals_union_minibatch(tmp_path, n_feat):
    import bayes_sim_ig_amd as pkg
    out = str(tmp_path / 'dp2.pt')
    mp.spawn(_worker, args=(2, 29600 + os.getpid() % 1000, out, n_feat), nprocs=2, join=True)
    res = torch.load(out)
    assert res['persistent'] == {50: 0, 512: 1, 'mdnn': 2}[n_feat]
    (x0, y0, i0), (x1, y1, i1) = _data(0), _data(1)
    # single process: [train0; train1; test0; test1], minibatch = both ranks' rows
    x = torch.cat([x0[:N_TRAIN], x1[:N_TRAIN], x0[N_TRAIN:], x1[N_TRAIN:]])
    y = torch.cat([y0[:N_TRAIN], y1[:N_TRAIN], y0[N_TRAIN:], y1[N_TRAIN:]])
    ids = np.concatenate([i0, i1 + N_TRAIN], axis=1)
    m = _model(pkg, n_feat)
    logs = m.run_training(x.cuda(), y.cuda(), NU, 2 * B, test_frac=0.2, ids_table=ids)
    pkg.MDNN.EPS_NOISE = 1e-5
    np.testing.assert_allclose(res['logs']['train_loss'], logs['train_loss'], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(res['logs']['test_loss'], logs['test_loss'], rtol=2e-5, atol=1e-6)
    # every weight within 1e-4 relative, but for a handful whose gradients sit at rounding level: Adam's
    # m / sqrt(v) turns an ulp of difference in such a gradient (the two paths add the ranks' shares in
    # different orders) into a difference of the size of a step -- those stay within 2e-5 absolute
    diff = (res['flat'] - m._flat.cpu()).abs()
    loose = diff > 1e-6 + 1e-4 * m._flat.cpu().abs()
    assert int(loose.sum()) <= 3 and float(diff.max()) < 2e-5, (int(loose.sum()), float(diff.max()))


def test_two_rank_fit_with_rank_local_jitter_scale(tmp_path):
    """EPS_NOISE = 1e-5 (the reference default): the jitter scale EPS_NOISE * mean(L_d) of mdnn.py:115
    is taken over the rank's own 24 rows instead of the 48 of the union minibatch, and the ranks
    draw their own noise.  Measured effect at R = 2 against the single-process fit on the union
    minibatch (itself with jitter): every logged loss within the north-star 1e-4 (observed
    ~1e-6: a 1e-5-relative perturbation of sigma either way)."""
    import bayes_sim_ig_amd as pkg
    out = str(tmp_path / 'dp2j.pt')
    try:
        mp.spawn(_worker, args=(2, 29700 + os.getpid() % 1000, out, 512, 1e-5), nprocs=2, join=True)
        res = torch.load(out)
        (x0, y0, i0), (x1, y1, i1) = _data(0), _data(1)
        x = torch.cat([x0[:N_TRAIN], x1[:N_TRAIN], x0[N_TRAIN:], x1[N_TRAIN:]])
        y = torch.cat([y0[:N_TRAIN], y1[:N_TRAIN], y0[N_TRAIN:], y1[N_TRAIN:]])
        ids = np.concatenate([i0, i1 + N_TRAIN], axis=1)
        m = _model(pkg, 512, 1e-5)
        logs = m.run_training(x.cuda(), y.cuda(), NU, 2 * B, test_frac=0.2, ids_table=ids)
    finally:
        pkg.MDNN.EPS_NOISE = 1e-5
    for key in ('train_loss', 'test_loss'):
        got, exp = np.asarray(res['logs'][key]), np.asarray(logs[key])
        print(key, 'max rel effect of rank-local jitter:', float(np.max(np.abs(got - exp) / np.abs(exp))))
        np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-6)


RESIDENT_RETRIES = []      # (test name, attempt) of every resident fit that had to be repeated in this session


def _fit_clean(build, fits=1, retries=1):
    """build() -> (BayesSim, theta, states, actions); its fit(s) with any time-out fallback turned into a
    retry on a fresh model (one resident call in some 38 000 has timed out on this pool without a known
    cause: DESIGN.md 5 -- a test of what the resident path computes should not fail on that).  The retry
    is NOT silent: it is recorded in RESIDENT_RETRIES, printed, and test_resident_retries_stay_rare bounds
    the session's count.  Returns (model wrapper, logs, resident calls of the fits)."""
    import warnings
    for attempt in range(retries + 1):
        bs, theta, states, actions, seed = build()
        np.random.seed(seed)
        before = bs.model._dp.resident_calls()
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('error', RuntimeWarning)      # a time-out fallback is a failure here
                logs = []
                for _ in range(fits):
                    logs += bs.fit(theta, states, actions)
            torch.cuda.synchronize()
            return bs, logs, bs.model._dp.resident_calls() - before
        except RuntimeWarning as w:
            RESIDENT_RETRIES.append((os.environ.get('PYTEST_CURRENT_TEST', '?'), attempt, str(w)[:120]))
            print('RESIDENT RETRY', RESIDENT_RETRIES[-1])
            if attempt == retries:
                raise


@pytest.mark.parametrize('config', ['cfg5', 'cfg3', 'cfg2'])
def test_rank_resident_across_the_exchange_is_bitwise_the_per_update_launches(monkeypatch, config):
    """BSIG_DP_RESIDENT: ONE launch per run_training call, the gradients handed to the exchange
    stream's all-reduce and back per update (fit_persistent.hip XR, fit_persistent_mdnn.hip), against a launch + all-reduce
    per update -- on a 1-rank RCCL group (all this pool can run): the same kernels' arithmetic in the
    same order, so parameters and logs must be bit-identical; and the resident launches must really
    have run (not timed out into the per-phase fallback, which would agree as well)."""
    import warnings
    import bench
    import bayes_sim_ig_amd as pkg
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29581')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        cfg = dict(bench.CONFIGS[config])      # (cfg3: the MDNN kernel, fit_persistent_mdnn.hip, on factor rows)
        theta, states, actions = bench.synth_pairs(cfg, 5000, 21, 'cuda:0')
        out = {}
        def build():
            bs = bench.build_gpu_model(pkg, cfg, 'cuda:0', 31)
            bs.model.enable_data_parallel()
            return bs, theta, states, actions, 32
        for mode in ('0', '1'):
            monkeypatch.setenv('BSIG_DP_RESIDENT', mode)
            bs, logs, calls = _fit_clean(build, fits=2)      # (the second fit right behind the first)
            assert not getattr(bs.model, '_no_persistent', False)
            out[mode] = (logs, bs.model._flat.clone(), calls)
    finally:
        if created:
            dist.destroy_process_group()
    assert out['0'][2] == 0 and out['1'][2] == 10, (out['0'][2], out['1'][2])
    assert torch.equal(out['0'][1], out['1'][1])
    for a, b in zip(out['0'][0], out['1'][0]):
        assert a == b


def test_resident_rank_without_a_usable_exchange_stream_runs_one_launch_per_update(monkeypatch):
    """No candidate stream answers the probe in time (forced: BSIG_DP_XR_PROBE_OK_US=0): the call must run
    one launch + all-reduce per update -- no resident call, no time-out, the same bits."""
    import warnings
    import bench
    import bayes_sim_ig_amd as pkg
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29582')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        cfg = dict(bench.CONFIGS['cfg5'])
        theta, states, actions = bench.synth_pairs(cfg, 2000, 22, 'cuda:0')
        out = {}
        for mode, ok_us in (('0', None), ('1', '0')):
            monkeypatch.setenv('BSIG_DP_RESIDENT', mode)
            if ok_us is None:
                monkeypatch.delenv('BSIG_DP_XR_PROBE_OK_US', raising=False)
            else:
                monkeypatch.setenv('BSIG_DP_XR_PROBE_OK_US', ok_us)
            def build():
                bs = bench.build_gpu_model(pkg, cfg, 'cuda:0', 33)
                bs.model.enable_data_parallel()
                return bs, theta, states, actions, 34
            bs, logs, calls = _fit_clean(build)
            out[mode] = (logs, bs.model._flat.clone(), calls)
    finally:
        if created:
            dist.destroy_process_group()
    assert out['1'][2] == 0
    assert torch.equal(out['0'][1], out['1'][1]) and out['0'][0] == out['1'][0]


def test_resident_launch_that_times_out_falls_back_to_one_launch_per_update(monkeypatch):
    """The exchange of one resident call never answers (forced: BSIG_DP_XR_DROP_CALL): the launch's bounded
    polls give up, the fit restores its snapshot, this model's rank stops staying resident (NOT: stops
    using the persistent kernel) and the repeated fit is bit for bit the launch-per-update fit."""
    import bench
    import bayes_sim_ig_amd as pkg
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29584')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        cfg = dict(bench.CONFIGS['cfg5'])
        theta, states, actions = bench.synth_pairs(cfg, 3000, 23, 'cuda:0')
        monkeypatch.setenv('BSIG_DP_RESIDENT', '0')
        ref = bench.build_gpu_model(pkg, cfg, 'cuda:0', 35)
        ref.model.enable_data_parallel()
        np.random.seed(36)
        ref_logs = ref.fit(theta, states, actions)
        monkeypatch.setenv('BSIG_DP_RESIDENT', '1')
        lib = pkg._lib.load()
        bs = bench.build_gpu_model(pkg, cfg, 'cuda:0', 35)
        bs.model.enable_data_parallel()
        np.random.seed(36)
        with pytest.warns(RuntimeWarning, match='resident across the gradient exchange timed out'):
            monkeypatch.setenv('BSIG_DP_XR_DROP_CALL', '1')      # the communicator's second call: chunk 1 of 3
            logs = bs.fit(theta, states, actions)
        torch.cuda.synchronize()
    finally:
        if created:
            dist.destroy_process_group()
    assert not getattr(bs.model, '_no_persistent', False)
    assert lib.bsig_comm_resident_mode(bs.model._dp.comm) == 0
    assert logs == ref_logs and torch.equal(bs.model._flat, ref.model._flat)


@pytest.mark.parametrize('config', ['cfg5', 'cfg3'])
def test_resident_rank_takes_its_adam_step_from_the_exchanged_gradients(monkeypatch, config):
    """A 1-rank group's all-reduce is the identity: a resident rank that kept its OWN gradients instead of
    reloading the buffer the exchange stream reduced would pass every other test.  With the exchanged
    gradients tripled (BSIG_DEBUG_GRAD_EXCHANGE_SCALE=3: three identical peers) the resident rank and the
    launch-per-update rank must still agree bit for bit -- and both must differ from the unscaled fit."""
    import bench
    import bayes_sim_ig_amd as pkg
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29585')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        cfg = dict(bench.CONFIGS[config])
        theta, states, actions = bench.synth_pairs(cfg, 2000, 25, 'cuda:0')
        out = {}
        for mode, scale in (('0', '3'), ('1', '3'), ('1', None)):
            monkeypatch.setenv('BSIG_DP_RESIDENT', mode)
            if scale is None:
                monkeypatch.delenv('BSIG_DEBUG_GRAD_EXCHANGE_SCALE', raising=False)
            else:
                monkeypatch.setenv('BSIG_DEBUG_GRAD_EXCHANGE_SCALE', scale)
            def build():
                bs = bench.build_gpu_model(pkg, cfg, 'cuda:0', 41)
                bs.model.enable_data_parallel()
                return bs, theta, states, actions, 42
            bs, logs, calls = _fit_clean(build)
            assert not getattr(bs.model, '_no_persistent', False)
            out[(mode, scale)] = (logs, bs.model._flat.clone(), calls)
    finally:
        if created:
            dist.destroy_process_group()
    assert out[('0', '3')][2] == 0 and out[('1', '3')][2] == 2 and out[('1', None)][2] == 2
    assert torch.equal(out[('0', '3')][1], out[('1', '3')][1])
    assert out[('0', '3')][0] == out[('1', '3')][0]
    assert not torch.equal(out[('1', '3')][1], out[('1', None)][1])


def test_resident_retries_stay_rare():
    """(runs last in this file) The resident fits above may repeat a fit whose launch timed out -- the
    unexplained 1-in-38 000 of DESIGN.md 5.  This session's fits are a few hundred resident calls: more
    than ONE retry would be a regression of the hand-off, not that residue."""
    print('resident fits repeated in this session:', RESIDENT_RETRIES)
    assert len(RESIDENT_RETRIES) <= 1, RESIDENT_RETRIES
