"""The persistent update kernel of the two-layer MDNN (csrc/fit_persistent_mdnn.hip):
parity with the oracle (teacher-forced chunks, all logged losses and the full
weight vector), with the per-phase kernels it replaces, across the launch
schedule, and the finiteness flag."""
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
    os.environ.pop('BSIG_NO_PERSISTENT', None)


def _cfg(d, k, summarizer='summary_start', t=11, sd=5, ad=2, hidden=(128, 128)):
    return dict(task='synthetic', model='MDNN', summarizer=summarizer, t=t, sd=sd, ad=ad,
                d=d, k=k, hidden=list(hidden), n_feat=0, pairs=1000)


def _chunk(B, cfg, n=1000, batch=100, n_updates=100, seed=3, eps=None, env=None):
    """One teacher-forced chunk on a freshly built model -> (logs, flat weights, BayesSim, data)."""
    import bench
    os.environ.pop('BSIG_NO_PERSISTENT', None)
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


# (D, K, summarizer, sd, ad, T): I = 70 (one k-slice, ragged tail); the Pendulum
# head (K=10, D=2) on I = 40; cross-correlation summaries I = 722 (3 k-slices, I % 4 = 2);
# a ShadowHand-sized head (Nh = 260) on I = 1170
SHAPES = [(3, 5, 'summary_start', 5, 2, 11), (2, 10, 'summary_start', 3, 1, 21),
          (17, 5, 'summary_corrdiff', 10, 8, 12), (32, 4, 'summary_start', 97, 20, 11)]


@pytest.mark.parametrize('d,k,summarizer,sd,ad,t', SHAPES)
def test_persistent_mdnn_chunk_matches_oracle(B, d, k, summarizer, sd, ad, t):
    """Teacher-forced chunk (EPS_NOISE=0, same start weights, same ids) against the
    fp32 oracle: all 6+6 logged losses within 1e-4 relative (the north-star
    bound), end weights within 2e-4 absolute of the oracle's."""
    import bench
    from oracle import summarize as osum
    cfg = _cfg(d, k, summarizer, t, sd, ad)
    torch.set_num_threads(8)
    logs, flat, bs, (theta, states, actions, ids) = _chunk(B, cfg, eps=0.0)
    assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 2
    ora = bench.build_oracle(cfg, bs.model.input_dim, 77, 0.0)
    bs0 = bench.build_gpu_model(B, cfg, DEV, 77)
    ora.load_state_dict({kk: v.cpu() for kk, v in bs0.model.state_dict().items()})
    ref = ora.run_training(osum.SUMMARIZERS[summarizer](states.cpu(), actions.cpu()),
                           theta.cpu(), 100, 100, ids_table=ids)
    for key in ('train_loss', 'test_loss'):
        got, exp = np.array(logs[key]), np.array(ref[key])
        assert got.shape == exp.shape == (6,)
        assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0)), (key, got, exp)
    # (an element whose gradient hovers around zero takes Adam steps of +-lr whose sign
    # is decided by rounding: the per-phase kernels show the same isolated outliers)
    sd_ = bs.model.state_dict()
    for name, v in ora.state_dict().items():
        diff = (sd_[name].cpu() - v).abs()
        assert int((diff > 2e-4 + 1e-3 * v.abs()).sum()) <= 3 and float(diff.max()) < 2e-3, name


@pytest.mark.parametrize('eps', [0.0, 1e-5])
def test_persistent_mdnn_equals_phase_kernels(B, eps):
    """Same chunk through the persistent kernel and through the per-phase kernels
    it replaces (BSIG_NO_PERSISTENT=1): same jitter RNG streams, the arithmetic
    differs only in summation order."""
    cfg = _cfg(4, 6, 'summary_corrdiff', 12, 7, 3)
    logs_p, flat_p, bs_p, _ = _chunk(B, cfg, eps=eps)
    logs_k, flat_k, bs_k, _ = _chunk(B, cfg, eps=eps, env={'BSIG_NO_PERSISTENT': '1'})
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(bs_p.model._plan) == 2
    assert lib.bsig_fit_is_persistent(bs_k.model._plan) == 0
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs_p[key], logs_k[key], rtol=3e-5, atol=3e-5), (key, logs_p, logs_k)
    assert torch.allclose(flat_p, flat_k, atol=1e-4, rtol=1e-3)
    assert not torch.equal(flat_p, torch.zeros_like(flat_p))


def test_persistent_mdnn_reruns_are_bitwise(B):
    """Every cross-workgroup sum is taken in a fixed order: two runs (jitter on)
    give identical bits."""
    cfg = _cfg(3, 5, 'summary_start', 11, 40, 6)
    a = _chunk(B, cfg, eps=1e-5)
    b = _chunk(B, cfg, eps=1e-5)
    assert a[0] == b[0]
    assert torch.equal(a[1], b[1])


@pytest.mark.parametrize('n,batch,n_updates', [(1000, 64, 37), (1000, 7, 11), (60, 100, 5), (1000, 104, 20)])
def test_persistent_mdnn_ragged_shapes_match_phase_kernels(B, n, batch, n_updates):
    """Minibatches that are not a multiple of the owners' 4 rows or of the MFMA
    tile, a chunk smaller than one minibatch, odd update counts."""
    cfg = _cfg(3, 5, 'summary_start', 11, 9, 2)
    a = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=1e-5)
    b = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=1e-5,
               env={'BSIG_NO_PERSISTENT': '1'})
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(a[2].model._plan) == 2
    assert lib.bsig_fit_is_persistent(b[2].model._plan) == 0
    for key in ('train_loss', 'test_loss'):
        assert len(a[0][key]) == len(b[0][key])
        assert np.allclose(a[0][key], b[0][key], rtol=3e-5, atol=3e-5), (key, a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=1e-4, rtol=1e-3)


def test_narrower_two_layer_trunks_run_zero_padded_in_the_persistent_kernel(B):
    """A (24, 24) tanh trunk (the reference's tests/regression_tests.py:59) is stored zero-padded
    to [128, 128]: the persistent kernel covers it, the padding stays exactly zero, and the fit
    matches the per-phase kernels on the unpadded network and the oracle."""
    import bench
    from oracle import summarize as osum
    cfg = _cfg(2, 10, 'summary_start', 21, 3, 1, hidden=(24, 24))
    torch.set_num_threads(8)
    logs, flat, bs, (theta, states, actions, ids) = _chunk(B, cfg, eps=0.0)
    m = bs.model
    assert B._lib.load().bsig_fit_is_persistent(m._plan) == 2
    assert m.net[0].weight.shape == (24, 40) and m.net[2].weight.shape == (24, 24)
    assert m.pi.weight.shape == (10, 24) and m.state_dict()['mu.weight'].shape == (20, 24)
    # every stored element outside the parameters' views is still exactly zero
    mask = torch.ones_like(m._flat, dtype=torch.bool)
    for o, full, real in m._param_slices:
        blk = mask[o:o + int(np.prod(full))].view(full)
        blk[tuple(slice(0, d) for d in real)] = False
    assert float(m._flat[mask].abs().max()) == 0.0
    assert float(m._exp_avg[mask].abs().max()) == 0.0
    ref, ora = _oracle_chunk(B, cfg, bs, theta, states, actions, ids)
    for key in ('train_loss', 'test_loss'):
        got, exp = np.array(logs[key]), np.array(ref[key])
        assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0)), (key, got, exp)
    os.environ['BSIG_NO_TRUNK_PAD'] = '1'
    try:
        logs_u, flat_u, bs_u, _ = _chunk(B, cfg, eps=0.0)
    finally:
        os.environ.pop('BSIG_NO_TRUNK_PAD', None)
    assert B._lib.load().bsig_fit_is_persistent(bs_u.model._plan) == 0
    assert bs_u.model._flat.numel() < m._flat.numel()
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(logs[key], logs_u[key], rtol=3e-5, atol=3e-5)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), bs_u.model.state_dict().items()):
        assert k1 == k2 and torch.allclose(v1, v2, atol=1e-4, rtol=1e-3), k1


def test_other_trunks_keep_the_phase_kernels(B):
    """Three hidden layers, a trunk wider than 128: per-phase kernels."""
    for hidden in ((64, 64, 64), (256, 128)):
        cfg = _cfg(3, 5, hidden=hidden)
        _, _, bs, _ = _chunk(B, cfg, n_updates=5, eps=0.0)
        assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 0


def test_nonfinite_summary_raises_through_persistent_mdnn(B):
    """NaN in a training summary reaches the kernel's finiteness flag ->
    AssertionError (the reference asserts isfinite, mdnn.py:120-124)."""
    import bench
    cfg = _cfg(3, 5)
    theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
    states[5, 0, 0] = float('nan')
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    ids = np.random.RandomState(5).randint(0, 800, (100, 100))
    ids[0, 0] = 5
    with pytest.raises(AssertionError):
        bs.model.run_training(bs._summarize(states, actions), theta, 100, 100, ids_table=ids)


@pytest.mark.parametrize('eps', [0.0, 1e-5])
def test_data_parallel_rank_in_persistent_mdnn_kernel_is_bitwise(B, eps):
    """A data-parallel rank runs each update as ONE launch of the MDNN persistent
    kernel (gradients out -> all-reduce -> the Adam step taken by the weights'
    owners at the start of the next launch).  On a 1-rank group that is the same
    arithmetic as the resident single-rank run: weights and losses bitwise equal."""
    import bench
    import torch.distributed as dist
    cfg = _cfg(4, 6, 'summary_corrdiff', 12, 7, 3)
    logs_p, flat_p, bs_p, _ = _chunk(B, cfg, eps=eps)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29579')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        B.MDNN.EPS_NOISE = eps
        theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
        bs = bench.build_gpu_model(B, cfg, DEV, 77)
        bs.model.enable_data_parallel()
        ids = np.random.RandomState(5).randint(0, 800, (100, 100))
        logs_d = bs.model.run_training(bs._summarize(states, actions), theta, 100, 100,
                                       ids_table=ids)
        flat_d = bs.model._flat.clone()
        assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 2
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
    """The held-out evaluations run inside the launch of the MDNN persistent kernel
    or as separate graphs between launches (BSIG_NO_INKERNEL_EVAL=1): same weights
    bit for bit, same held-out NLLs up to the summation order."""
    cfg = _cfg(4, 6, 'summary_corrdiff', 12, 7, 3)
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


# ---- wide heads (Nh > 272): the head-block workgroups form the head outputs -----------------
def _oracle_chunk(B, cfg, bs, theta, states, actions, ids, n_updates=100, batch=100):
    import bench
    from oracle import summarize as osum
    ora = bench.build_oracle(cfg, bs.model.input_dim, 77, 0.0)
    bs0 = bench.build_gpu_model(B, cfg, DEV, 77)
    ora.load_state_dict({kk: v.cpu() for kk, v in bs0.model.state_dict().items()})
    return ora.run_training(osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu()),
                            theta.cpu(), n_updates, batch, ids_table=ids), ora


# the reference YAMLs' 10 components (cfg/ant.yaml:69-70: D = 17 -> Nh = 350; a ShadowHand-sized
# theta, D = 32 -> Nh = 650, 21 head blocks) and a head that is not a multiple of 32 or 16
@pytest.mark.parametrize('d,k,summarizer,sd,ad,t', [(17, 10, 'summary_corrdiff', 10, 8, 12),
                                                    (32, 10, 'summary_start', 97, 20, 11),
                                                    (25, 7, 'summary_start', 5, 2, 11)])
def test_wide_heads_chunk_matches_oracle(B, d, k, summarizer, sd, ad, t):
    cfg = _cfg(d, k, summarizer, t, sd, ad)
    torch.set_num_threads(8)
    logs, flat, bs, (theta, states, actions, ids) = _chunk(B, cfg, eps=0.0)
    assert k * (1 + 2 * d) > 272
    assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 2
    ref, ora = _oracle_chunk(B, cfg, bs, theta, states, actions, ids)
    for key in ('train_loss', 'test_loss'):
        got, exp = np.array(logs[key]), np.array(ref[key])
        assert got.shape == exp.shape == (6,)
        assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0)), (key, got, exp)
    sd_ = bs.model.state_dict()
    for name, v in ora.state_dict().items():
        diff = (sd_[name].cpu() - v).abs()
        assert int((diff > 2e-4 + 1e-3 * v.abs()).sum()) <= 3 and float(diff.max()) < 2e-3, name


@pytest.mark.parametrize('eps', [0.0, 1e-5])
@pytest.mark.parametrize('n,batch,n_updates', [(1000, 100, 100), (1000, 37, 11), (60, 100, 5)])
def test_wide_heads_equal_the_owner_resident_heads(B, eps, n, batch, n_updates):
    """The wide path forced onto a head that also fits the owners (BSIG_MDNN_WIDE_HEADS=1):
    the same update, head products summed in another order; and against the per-phase
    kernels.  Reruns are bitwise."""
    cfg = _cfg(4, 6, 'summary_corrdiff', 12, 7, 3)
    a = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
    os.environ['BSIG_MDNN_WIDE_HEADS'] = '1'
    try:
        b = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
        b2 = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
        assert B._lib.load().bsig_fit_is_persistent(b[2].model._plan) == 2
    finally:
        os.environ.pop('BSIG_MDNN_WIDE_HEADS', None)
    # (two summation orders of the head products; the jitter noise amplifies the 1e-7 differences
    # of the early updates to a few 1e-5 by update 100 -- the weights stay within 1e-5)
    for key in ('train_loss', 'test_loss'):
        assert len(a[0][key]) == len(b[0][key])
        assert np.allclose(a[0][key], b[0][key], rtol=1e-4, atol=1e-4), (key, a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=1e-4, rtol=1e-3)
    assert b[0] == b2[0] and torch.equal(b[1], b2[1])


def test_wide_heads_data_parallel_rank_is_bitwise(B):
    import bench
    import torch.distributed as dist
    cfg = _cfg(17, 10, 'summary_corrdiff', 12, 10, 8)
    logs_p, flat_p, bs_p, _ = _chunk(B, cfg, eps=1e-5)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29579')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        B.MDNN.EPS_NOISE = 1e-5
        theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
        bs = bench.build_gpu_model(B, cfg, DEV, 77)
        bs.model.enable_data_parallel()
        ids = np.random.RandomState(5).randint(0, 800, (100, 100))
        logs_d = bs.model.run_training(bs._summarize(states, actions), theta, 100, 100, ids_table=ids)
        flat_d = bs.model._flat.clone()
        assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 2
    finally:
        if created:
            dist.destroy_process_group()
    assert logs_d['train_loss'] == logs_p['train_loss']
    assert np.allclose(logs_d['test_loss'], logs_p['test_loss'], rtol=1e-6, atol=1e-6)
    assert torch.equal(flat_d, flat_p)


# ---- full covariance (the refit of BayesSim.predict, bayes_sim.py:148-179; fullCovariance: True) ----
# (D, K, summarizer, sd, ad, T): the pendulum refit's head (D = 2, K = 10 -> Nh = 60), a
# mid-sized theta (Nh = 140), and a head past the owners' LDS (D = 8, K = 10 -> Nh = 450: wide)
@pytest.mark.parametrize('d,k,summarizer,sd,ad,t', [(2, 10, 'summary_start', 3, 1, 21),
                                                    (6, 5, 'summary_corrdiff', 7, 3, 12),
                                                    (8, 10, 'summary_start', 5, 2, 11)])
def test_full_covariance_chunk_matches_oracle(B, d, k, summarizer, sd, ad, t):
    """40 updates: at the benches' lr = 1e-3 these full-covariance fits leave the teacher-forced
    horizon early -- by update 60 the per-phase kernels, the persistent kernel and the oracle
    (identical to 1e-7 through update 40) are 1e-5 apart, by update 100 1e-2..1e-1
    (tools/micro/full_probe.py); the reference does not reproduce itself there either."""
    cfg = dict(_cfg(d, k, summarizer, t, sd, ad), full=True)
    torch.set_num_threads(8)
    logs, flat, bs, (theta, states, actions, ids) = _chunk(B, cfg, eps=0.0, n_updates=40)
    assert bs.model.Lower is not None
    assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == 2
    ref, ora = _oracle_chunk(B, cfg, bs, theta, states, actions, ids, n_updates=40)
    for key in ('train_loss', 'test_loss'):
        got, exp = np.array(logs[key]), np.array(ref[key])
        assert got.shape == exp.shape == (6,)
        assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.abs(exp), 1.0)), (key, got, exp)
    sd_ = bs.model.state_dict()
    for name, v in ora.state_dict().items():
        diff = (sd_[name].cpu() - v).abs()
        assert int((diff > 2e-4 + 1e-3 * v.abs()).sum()) <= 3 and float(diff.max()) < 2e-3, name


@pytest.mark.parametrize('eps', [0.0, 1e-5])
@pytest.mark.parametrize('n,batch,n_updates', [(1000, 100, 40), (300, 37, 11)])
def test_full_covariance_persistent_equals_phase_kernels(B, eps, n, batch, n_updates):
    """Same jitter draws (one Philox value per (row, d, k) in both), same operations per
    component: only the GEMM summation orders differ."""
    cfg = dict(_cfg(4, 6, 'summary_corrdiff', 12, 7, 3), full=True)
    a = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
    b = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps, env={'BSIG_NO_PERSISTENT': '1'})
    a2 = _chunk(B, cfg, n=n, batch=batch, n_updates=n_updates, eps=eps)
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(a[2].model._plan) == 2
    assert lib.bsig_fit_is_persistent(b[2].model._plan) == 0
    for key in ('train_loss', 'test_loss'):
        assert len(a[0][key]) == len(b[0][key])
        assert np.allclose(a[0][key], b[0][key], rtol=1e-4, atol=1e-4), (key, a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=1e-4, rtol=1e-3)
    assert a[0] == a2[0] and torch.equal(a[1], a2[1])            # reruns bitwise


@pytest.mark.parametrize('config', ['cfg3', 'ant_yaml', 'cfg4'])
def test_fewer_rows_per_owner_is_bitwise_the_four_row_layout(config):
    """Where the chip has the CUs, an owner workgroup takes two minibatch rows, or one, instead of four (less
    slab bytes through its CU's memory pipe, shorter per-row loops: fit_persistent_mdnn.hip, mdnn_geom).  The
    k-slice sums keep their four-way split and their order (slab_quads_sum), every other per-row product is
    unchanged: the trained weights of a teacher-forced chunk (EPS_NOISE = 0) are bit for bit those of the
    four-row layout (BSIG_MDNN_MR=4, read once per process: two fresh processes) -- two rows on summary rows
    from factor rows (cfg3) and with wide heads (cfg/ant.yaml), one row on a narrow first layer (cfg4)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for mr in ('4', '0'):
        env = dict(os.environ, BSIG_MDNN_MR=mr)
        res = subprocess.run([sys.executable, os.path.join(root, 'tools', 'ab_bitwise.py'), config], env=env,
                             capture_output=True, text=True, timeout=600)
        m = re.search(r'weights ([0-9a-f]{16}) logs [0-9a-f]{16} test_loss\[-1\] (\S+)', res.stdout)
        assert res.returncode == 0 and m, res.stdout[-2000:] + res.stderr[-3000:]
        out.append((m.group(1), float(m.group(2))))
    assert out[0][0] == out[1][0], out
    assert abs(out[0][1] - out[1][1]) <= 2e-6 * max(1.0, abs(out[0][1])), out


# (D, K, summarizer, sd, ad, T): the reference YAMLs' 10 components on 16 lanes (Pendulum head; a 13-dimensional one);
# 5 components on 8 lanes; 4 exact (ShadowHand head, two sweeps over the pair); 8 exact; 3 on 4
MDNN_FAST_ROW_SHAPES = [(2, 10, 'summary_start', 3, 1, 21), (13, 10, 'summary_start', 4, 1, 21),
                        (17, 5, 'summary_start', 10, 8, 12), (32, 4, 'summary_start', 30, 5, 11),
                        (6, 8, 'summary_start', 5, 2, 11), (3, 3, 'summary_start', 5, 2, 11)]


@pytest.mark.parametrize('eps', [0.0, 1e-5])
@pytest.mark.parametrize('d,k,summarizer,sd,ad,t', MDNN_FAST_ROW_SHAPES)
def test_fast_rows_of_the_mdnn_owners_match_the_generic_row(B, d, k, summarizer, sd, ad, t, eps):
    """Round 6: the lean two-wavefront row in the MDNN owners (mdnn_fast_row / diag_row_fast_core; first layers of at
    most 4096 inputs) against the shape-generic row (BSIG_MDNN_FAST_ROWS=0, read per launch): same chunk, same
    start weights, ids and jitter draws per (row, dimension, component); only summation orders differ.  The
    bounds are the ones the persistent kernel is held to against the per-phase kernels."""
    cfg = _cfg(d, k, summarizer, t, sd, ad)
    try:
        a = _chunk(B, cfg, eps=eps)
        b = _chunk(B, cfg, eps=eps, env={'BSIG_MDNN_FAST_ROWS': '0'})
    finally:
        os.environ.pop('BSIG_MDNN_FAST_ROWS', None)
    lib = B._lib.load()
    assert lib.bsig_fit_is_persistent(a[2].model._plan) == 2 and lib.bsig_fit_is_persistent(b[2].model._plan) == 2
    for key in ('train_loss', 'test_loss'):
        assert np.allclose(a[0][key], b[0][key], rtol=3e-5, atol=3e-5), (key, a[0], b[0])
    assert torch.allclose(a[1], b[1], atol=1e-4, rtol=1e-3)
    assert not torch.equal(a[1], b[1])       # (the switch does something)
