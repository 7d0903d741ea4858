"""The persistent update kernel (csrc/fit_persistent.hip) and the feature cache:
parity with the oracle (teacher-forced chunks, full weight vectors), with the
one-kernel-per-phase path it replaces, and across the launch schedule."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


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
    pkg.MDNN.USE_GRAPH = True
    for k in ('BSIG_NO_PERSISTENT', 'BSIG_NO_FEAT_CACHE', 'BSIG_PERSIST_FAST_ROWS'):
        os.environ.pop(k, None)


def _cfg(d, k, n_feat, sd=5, ad=2):
    return dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=sd, ad=ad,
                d=d, k=k, hidden=[], n_feat=n_feat, pairs=1000)


def _chunk(B, cfg, n=1000, batch=100, n_updates=100, seed=3, eps=None, env=None):
    """One teacher-forced chunk on a freshly built model -> (logs, flat weights, model)."""
    import bench
    for k in ('BSIG_NO_PERSISTENT', 'BSIG_NO_FEAT_CACHE'):
        os.environ.pop(k, None)
    os.environ.update(env or {})
    if eps is not None:
        B.MDNN.EPS_NOISE = eps
    theta, states, actions = bench.synth_pairs(cfg, n, seed, DEV)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    n_train = n - int(n * 0.2)
    ids = np.random.RandomState(5).randint(0, n_train, (n_updates, batch))
    summ = bs._summarize(states, actions)
    logs = bs.model.run_training(summ, theta, n_updates, batch, ids_table=ids)
    return logs, bs.model._flat.clone(), bs, (theta, states, actions, ids)


# (D, K, n_feat): two head blocks x two k-slices; the cfg2 head on 1024 features;
# 11 head blocks x 16 k-slices (176 tile workgroups: two rows per owner workgroup)
# ; the reference's own feature counts (bayes_sim.py:81 n_feat=200, MDRFF default 500:
# one / two k-slices with a masked tail)
SHAPES = [(3, 5, 512), (13, 10, 1024), (40, 4, 4096), (2, 10, 200), (5, 3, 500)]


@pytest.mark.parametrize('d,k,n_feat', SHAPES)
def test_persistent_chunk_matches_oracle(B, d, k, n_feat):
    """Teacher-forced chunk (EPS_NOISE=0, same start weights, same ids) against
    the fp32 oracle: all 6+6 logged losses within 1e-4 relative (the north-star
    bound), end weights within 2e-4 absolute of the oracle's."""
    import bench
    from oracle import summarize as osum
    cfg = _cfg(d, k, n_feat)
    torch.set_num_threads(8)
    logs, flat, bs, (theta, states, actions, ids) = _chunk(B, cfg, eps=0.0)
    assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 1
    ora = bench.build_oracle(cfg, bs.model.rff.d, 77, 0.0, freqs=bs.model.rff.freqs.cpu().numpy())
    bs0 = bench.build_gpu_model(B, cfg, DEV, 77)
    ora.load_state_dict({kk: v.cpu() for kk, v in bs0.model.state_dict().items()})
    ora.rff.freqs = bs0.model.rff.freqs.cpu()
    ref = ora.run_training(osum.SUMMARIZERS['summary_start'](states.cpu(), actions.cpu()),
                           theta.cpu(), 100, 100, ids_table=ids)
    for key in ('train_loss', 'test_loss'):
        got, exp = np.array(logs[key]), np.array(ref[key])
        assert got.shape == exp.shape == (6,)
        assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0)), (key, got, exp)
    sd = bs.model.state_dict()
    for name, v in ora.state_dict().items():
        assert torch.allclose(sd[name].cpu(), v, atol=2e-4, rtol=1e-3), name


@pytest.mark.parametrize('n_feat', [512, 200])
@pytest.mark.parametrize('eps', [0.0, 1e-5])
def test_persistent_equals_phase_kernels(B, eps, n_feat):
    """Same chunk through the persistent kernel and through the per-phase
    kernels it replaces (BSIG_NO_PERSISTENT=1): the jitter RNG streams are the
    same, the arithmetic differs only in summation order."""
    cfg = _cfg(4, 6, n_feat)
    logs_p, flat_p, _, _ = _chunk(B, cfg, eps=eps)
    logs_k, flat_k, _, _ = _chunk(B, cfg, eps=eps, env={'BSIG_NO_PERSISTENT': '1'})
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs_p[key], logs_k[key], rtol=2e-5, atol=2e-5), (key, logs_p, logs_k)
    assert torch.allclose(flat_p, flat_k, atol=5e-5, rtol=1e-3)
    assert not torch.equal(flat_p, torch.zeros_like(flat_p))


def test_feature_cache_is_bitwise_neutral(B):
    """Projecting each distinct training row once (feature cache) or every
    gathered minibatch row (BSIG_NO_FEAT_CACHE=1) feeds identical features to
    identical kernels: logs and weights are bitwise equal."""
    cfg = _cfg(3, 5, 512)
    for env in ({}, {'BSIG_NO_PERSISTENT': '1'}):
        a = _chunk(B, cfg, eps=1e-5, env=dict(env))
        b = _chunk(B, cfg, eps=1e-5, env=dict(env, BSIG_NO_FEAT_CACHE='1'))
        assert a[0] == b[0]
        assert torch.equal(a[1], b[1])


def test_persistent_graph_flag_and_rerun_are_bitwise(B):
    """Evaluations replayed from a graph or launched directly, and a second
    identical run: bitwise identical (fixed summation orders, no atomics)."""
    cfg = _cfg(3, 5, 512)
    out = []
    for use_graph in (True, False, True):
        B.MDNN.USE_GRAPH = use_graph
        out.append(_chunk(B, cfg, eps=1e-5)[:2])
    for logs, flat in out[1:]:
        assert logs == out[0][0]
        assert torch.equal(flat, out[0][1])


@pytest.mark.parametrize('n,batch,n_updates', [(1000, 64, 37), (1000, 7, 11), (60, 100, 5), (1000, 104, 20)])
def test_persistent_ragged_shapes_match_phase_kernels(B, n, batch, n_updates):
    """Minibatches that are not a multiple of the MFMA tile, a chunk smaller
    than one minibatch (ids repeat), odd update counts (runs of 1..n between
    evaluations)."""
    cfg = _cfg(3, 5, 512)
    a = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=1e-5)
    b = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=1e-5,
               env={'BSIG_NO_PERSISTENT': '1'})
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(a[2].model._plan) == 1
    assert lib.bsig_fit_is_persistent(b[2].model._plan) == 0
    for key in ('train_loss', 'test_loss'):
        assert len(a[0][key]) == len(b[0][key])
        assert np.allclose(a[0][key], b[0][key], rtol=2e-5, atol=2e-5), (key, a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=5e-5, rtol=1e-3)


def test_fit_updates_entry_point_splits_runs(B):
    """bsig_fit_updates(5) + bsig_fit_updates(7) == bsig_fit_updates(12):
    the launch schedule does not change the arithmetic."""
    L = B._lib
    lib = L.load()
    cfg = _cfg(3, 5, 512)
    _, _, bs, _ = _chunk(B, cfg, eps=1e-5)
    m = bs.model
    st = L.stream()
    res = []
    for runs in ((12,), (5, 7), (1, 1, 10)):
        with torch.no_grad():
            m._flat.copy_(torch.linspace(-0.05, 0.05, m._flat.numel(), device=DEV))
        L.check(lib.bsig_fit_begin(m._plan, 99, 100, st))
        for r in runs:
            L.check(lib.bsig_fit_updates(m._plan, r, st))
        torch.cuda.synchronize()
        res.append(m._flat.clone())
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])
    assert torch.isfinite(res[0]).all()


def test_nonfinite_features_raise_through_persistent_path(B):
    """NaN in a training summary reaches the persistent kernel's finiteness
    flag -> AssertionError (the reference asserts isfinite, mdnn.py:120-124)."""
    import bench
    cfg = _cfg(3, 5, 512)
    theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
    states[5, 0, 0] = float('nan')
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    ids = np.random.RandomState(5).randint(0, 800, (100, 100))
    ids[0, 0] = 5
    with pytest.raises(AssertionError):
        bs.model.run_training(bs._summarize(states, actions), theta, 100, 100, ids_table=ids)


@pytest.mark.parametrize('eps', [0.0, 1e-5])
def test_data_parallel_rank_in_persistent_kernel_is_bitwise(B, eps):
    """A data-parallel rank runs each update as ONE launch of the persistent
    kernel (gradients out -> all-reduce -> the Adam step taken by the next launch
    while it loads its tiles).  On a 1-rank group that is the same arithmetic as
    the resident single-rank run: weights and losses bitwise equal."""
    import torch.distributed as dist
    cfg = _cfg(4, 6, 512)
    logs_p, flat_p, bs_p, _ = _chunk(B, cfg, eps=eps)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29578')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        import bench
        B.MDNN.EPS_NOISE = eps
        theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
        bs = bench.build_gpu_model(B, cfg, DEV, 77)
        bs.model.enable_data_parallel()
        ids = np.random.RandomState(5).randint(0, 800, (100, 100))
        logs_d = bs.model.run_training(bs._summarize(states, actions), theta, 100, 100,
                                       ids_table=ids)
        flat_d = bs.model._flat.clone()
        lib = B._lib.load()
        assert lib.bsig_fit_is_persistent(bs.model._plan) == 1
    finally:
        if created:
            dist.destroy_process_group()
    # (the resident run evaluates the held-out rows inside its launch, the data-parallel
    # rank with the per-phase kernels: same weights bit for bit, another summation order)
    assert logs_d['train_loss'] == logs_p['train_loss']
    assert np.allclose(logs_d['test_loss'], logs_p['test_loss'], rtol=1e-6, atol=1e-6)
    assert torch.equal(flat_d, flat_p)


@pytest.mark.parametrize('n,batch,n_updates', [(1000, 100, 100), (1000, 100, 3), (60, 100, 5), (300, 64, 37)])
@pytest.mark.parametrize('eps', [0.0, 1e-5])
def test_in_launch_evaluations_equal_the_evaluation_graphs(B, eps, n, batch, n_updates):
    """The held-out evaluations run inside the launch of the persistent kernel
    (tile workgroups in their wait window, owners after their rows) or as
    separate graphs between launches (BSIG_NO_INKERNEL_EVAL=1): same weights bit
    for bit (same updates, same jitter streams), same held-out NLLs up to the
    summation order; two passes of held-out rows, a ragged pass, an evaluation
    after every update."""
    cfg = _cfg(4, 6, 512)
    a = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
    os.environ['BSIG_NO_INKERNEL_EVAL'] = '1'
    try:
        b = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
    finally:
        os.environ.pop('BSIG_NO_INKERNEL_EVAL', None)
    assert a[0]['train_loss'] == b[0]['train_loss']
    assert len(a[0]['test_loss']) == len(b[0]['test_loss'])
    assert np.allclose(a[0]['test_loss'], b[0]['test_loss'], rtol=2e-6, atol=2e-6), (a[0], b[0])
    assert torch.equal(a[1], b[1])


def test_time_out_is_recovered_on_the_phase_kernels(B):
    """A persistent launch needs every workgroup resident at once.  With 200 of the 256 CUs held by another
    kernel (bsig_debug_spin on a side stream, 1.5 s; workgroups that take a CU's whole LDS, so that the small
    tiles of this head cannot move in beside them) its bounded polls give up and raise bit 1 of
    the flag word; BayesSim.fit then restores the parameters / Adam moments / RNG states it saved
    at its start and repeats the loop on the per-phase kernels: the result is bit for bit the
    per-phase fit (BSIG_NO_PERSISTENT=1), not a partially updated model."""
    import os
    import bench
    cfg = dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=5, ad=2,
               d=3, k=4, hidden=[], n_feat=512, pairs=2000)
    theta, states, actions = bench.synth_pairs(cfg, 2000, 3, DEV)
    lib = B._lib.load()
    # reference: the per-phase kernels
    os.environ['BSIG_NO_PERSISTENT'] = '1'
    try:
        ref = bench.build_gpu_model(B, cfg, DEV, 77)
        np.random.seed(11); torch.manual_seed(11)
        ref_logs = ref.fit(theta, states, actions)
    finally:
        os.environ.pop('BSIG_NO_PERSISTENT', None)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    np.random.seed(11); torch.manual_seed(11)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    import ctypes as C
    B._lib.check(lib.bsig_debug_spin(200, 160 * 1024, 1500, C.c_void_p(side.cuda_stream)))
    logs = bs.fit(theta, states, actions)
    torch.cuda.synchronize()
    assert getattr(bs.model, '_no_persistent', False), 'the launch was expected to time out'
    assert lib.bsig_fit_is_persistent(bs.model._plan) == 0
    assert [lg['test_loss'] for lg in logs] == [lg['test_loss'] for lg in ref_logs]
    assert torch.equal(bs.model._flat, ref.model._flat)
    # and a model that is not disturbed keeps the persistent kernel
    ok = bench.build_gpu_model(B, cfg, DEV, 77)
    ok.fit(theta, states, actions)
    assert lib.bsig_fit_is_persistent(ok.model._plan) == 1 and not getattr(ok.model, '_no_persistent', False)


def test_time_out_of_a_data_parallel_rank_is_recovered_by_the_group(B):
    """The same with a data-parallel rank (one launch per update, RCCL exchange, here a 1-rank group): the
    time-out bit travels in the logs bsig_fit_run_dp sums over the ranks, so EVERY rank of a group reads it
    at the same chunk -- the rank that timed out has kept enqueueing its all-reduces meanwhile, nobody
    hangs --, restores its snapshot and repeats the loop on the per-phase kernels with its peers: bit for
    bit the per-phase data-parallel fit."""
    import ctypes as C
    import bench
    import torch.distributed as dist
    cfg = dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=5, ad=2,
               d=3, k=4, hidden=[], n_feat=512, pairs=2000)
    theta, states, actions = bench.synth_pairs(cfg, 2000, 3, DEV)
    lib = B._lib.load()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29579')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        os.environ['BSIG_NO_PERSISTENT'] = '1'
        try:
            ref = bench.build_gpu_model(B, cfg, DEV, 77)
            ref.model.enable_data_parallel()
            np.random.seed(11); torch.manual_seed(11)
            ref_logs = ref.fit(theta, states, actions)
        finally:
            os.environ.pop('BSIG_NO_PERSISTENT', None)
        bs = bench.build_gpu_model(B, cfg, DEV, 77)
        bs.model.enable_data_parallel()
        np.random.seed(11); torch.manual_seed(11)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        B._lib.check(lib.bsig_debug_spin(200, 160 * 1024, 1500, C.c_void_p(side.cuda_stream)))
        logs = bs.fit(theta, states, actions)
        torch.cuda.synchronize()
    finally:
        if created:
            dist.destroy_process_group()
    assert getattr(bs.model, '_no_persistent', False), 'the launches were expected to time out'
    assert lib.bsig_fit_is_persistent(bs.model._plan) == 0
    assert [lg['test_loss'] for lg in logs] == [lg['test_loss'] for lg in ref_logs]
    assert torch.equal(bs.model._flat, ref.model._flat)


# (D, K, n_feat): K = 4 exact on 4 lanes with one / two sweeps per wavefront; K = 10 and 16 on 16 lanes (the reference
# YAMLs' 10 components); K = 5 and 8 on 8 lanes; K = 3 on 4 lanes; D = 40, K = 4: three sweeps (two per wavefront)
FAST_ROW_SHAPES = [(32, 4, 1024), (13, 10, 512), (17, 5, 512), (3, 16, 256), (40, 4, 512), (20, 8, 512), (6, 3, 256),
                   (32, 10, 512)]


@pytest.mark.parametrize('eps', [0.0, 1e-5])
@pytest.mark.parametrize('d,k,n_feat', FAST_ROW_SHAPES)
def test_fast_row_owners_match_the_generic_row(B, d, k, n_feat, eps):
    """Round 6: the row owners' lean two-wavefront row (u_own_update_fast / diag_row_fast_core: component count padded
    to 4 / 8 / 16 lanes, DPP reductions, per-wavefront granules) against the shape-generic row it replaces
    (BSIG_PERSIST_FAST_ROWS=0, read per launch): same chunk, same start weights, same minibatch ids, same jitter
    draws per (row, dimension, component) -- only summation orders differ.  Every logged loss within 2e-5, the
    weights within 5e-5 (the bound the persistent kernel is held to against the per-phase kernels), and
    the two runs must not be the SAME bits (that would mean the switch does nothing)."""
    cfg = _cfg(d, k, n_feat)
    a = _chunk(B, cfg, eps=eps)
    b = _chunk(B, cfg, eps=eps, env={'BSIG_PERSIST_FAST_ROWS': '0'})
    os.environ.pop('BSIG_PERSIST_FAST_ROWS', None)
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(a[2].model._plan) == 1 and lib.bsig_fit_is_persistent(b[2].model._plan) == 1
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(a[0][key], b[0][key], rtol=2e-5, atol=2e-5), (key, a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=5e-5, rtol=1e-3)
    import ctypes as C
    geo = (C.c_int32 * 16)()
    if lib.bsig_debug_persist_geometry(100, n_feat, d, k, 200, geo) and geo[12] == 0 and k <= 16:
        # (owners on CUs of their own and K <= 16: the lean row ran in `a`)
        assert not torch.equal(a[1], b[1])
