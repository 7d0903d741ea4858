"""GPU parity of the fit path: teacher-forced 100-update chunks against the
reference-generated golden vectors (EPS_NOISE=0, same start weights, same
minibatch id table), BayesSim end to end on the reference's own pendulum
fixture, graph replay == direct launches, and size-independent properties at
BASELINE sizes."""
import os

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def B():
    import bayes_sim_ig_amd as pkg
    pkg._lib.require_gpu()
    pkg.MDNN.VERBOSE = False
    return pkg


@pytest.fixture(autouse=True)
def _eps_guard():
    import bayes_sim_ig_amd as pkg
    old = pkg.MDNN.EPS_NOISE
    yield
    pkg.MDNN.EPS_NOISE = old
    pkg.MDNN.USE_GRAPH = True


CHUNKS = {
    'mdnn_start': dict(cls='MDNN', summarizer='summary_start', d=2, k=10,
                       hidden=(24, 24), full=False),
    'mdnn_corrdiff_full': dict(cls='MDNN', summarizer='summary_corrdiff', d=3, k=3,
                               hidden=(16, 16), full=True),
    'mdrff_corrdiff': dict(cls='MDRFF', summarizer='summary_corrdiff', d=4, k=4,
                           hidden=[], full=False),
    # f4: the reference's BayesSim built with modelClass 'MDRFF_Matern32_2.0' (bayes_sim.py:72-81)
    'mdrff_matern32': dict(cls='MDRFF', summarizer='summary_corrdiff', d=4, k=4,
                           hidden=[], full=False, kernel='Matern32', sigma=2.0),
}


def _chunk_model(B, tag, g, input_dim):
    kw = CHUNKS[tag]
    d = kw['d']
    common = dict(input_dim=input_dim, output_dim=d, output_lows=np.zeros(d),
                  output_highs=np.ones(d), n_gaussians=kw['k'],
                  full_covariance=kw['full'], lr=float(g['lr']),
                  activation=torch.nn.Tanh, device=DEV)
    if kw['cls'] == 'MDRFF':
        m = B.MDRFF(n_feat=200, sigma=kw.get('sigma', 4.0), kernel=kw.get('kernel', 'RBF'),
                    freqs=g['rff.freqs'], **common)
        np.testing.assert_array_equal(m.rff.sigma.cpu().numpy(), g['rff.sigma'])
    else:
        m = B.MDNN(hidden_layers=kw['hidden'], **common)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items()
                       if k.startswith('w0.')})
    return m


@pytest.mark.parametrize('tag', list(CHUNKS))
@pytest.mark.parametrize('use_graph', [True, False])
def test_teacher_forced_chunk_matches_reference(B, tag, use_graph):
    g = golden('chunk_%s.npz' % tag)
    B.MDNN.EPS_NOISE = 0.0
    B.MDNN.USE_GRAPH = use_graph
    states = torch.from_numpy(g['states']).to(DEV)
    actions = torch.from_numpy(g['actions']).to(DEV)
    theta = torch.from_numpy(g['theta']).to(DEV)
    summ = getattr(B.summarizers, CHUNKS[tag]['summarizer'])(states, actions)
    np.testing.assert_allclose(summ.cpu().numpy(), g['summaries'], rtol=1e-6, atol=1e-7)
    m = _chunk_model(B, tag, g, summ.shape[1])
    logs = m.run_training(summ, theta, int(g['n_updates']), int(g['batch']),
                          ids_table=g['ids'])
    # north-star tolerance: held-out NLL within 1e-4 relative
    np.testing.assert_allclose(logs['test_loss'], g['test_loss'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logs['train_loss'], g['train_loss'], rtol=1e-4, atol=1e-5)
    n_train = int(states.shape[0] * 0.8)
    mog = m.predict_MoGs(summ[n_train:n_train + 1])[0]
    # fitted MoG weights / means / covariances within 1e-4 relative (+ tiny abs)
    np.testing.assert_allclose(mog.a, g['mog.a'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(np.stack([c.m for c in mog.xs]), g['mog.ms'],
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(np.stack([c.S for c in mog.xs]), g['mog.Ss'],
                               rtol=1e-4, atol=1e-7)
    nll = -mog.eval(g['theta'][n_train:n_train + 1].astype(np.float64), log=True)
    np.testing.assert_allclose(nll, g['mog.nll_true'], rtol=1e-4, atol=1e-5)


def test_graph_replay_equals_direct_launches(B):
    g = golden('chunk_mdnn_start.npz')
    B.MDNN.EPS_NOISE = 0.0
    states = torch.from_numpy(g['states']).to(DEV)
    actions = torch.from_numpy(g['actions']).to(DEV)
    theta = torch.from_numpy(g['theta']).to(DEV)
    summ = B.summary_start(states, actions)
    out = []
    for use_graph in (True, False, True):
        B.MDNN.USE_GRAPH = use_graph
        m = _chunk_model(B, 'mdnn_start', g, summ.shape[1])
        logs = m.run_training(summ, theta, 100, 100, ids_table=g['ids'])
        out.append((logs, m._flat.clone()))
    for logs, flat in out[1:]:
        assert logs == out[0][0]                  # bitwise: same kernels, same order
        assert torch.equal(flat, out[0][1])


def test_second_call_uses_fresh_optimizer_and_same_plan(B):
    """run_training twice on one model: Adam state restarts (mdnn.py:203) and
    the result equals the oracle doing the same."""
    from oracle import estimators as oest
    g = golden('chunk_mdnn_start.npz')
    B.MDNN.EPS_NOISE = 0.0
    states, actions = torch.from_numpy(g['states']), torch.from_numpy(g['actions'])
    theta = torch.from_numpy(g['theta'])
    summ = B.summary_start(states.to(DEV), actions.to(DEV))
    m = _chunk_model(B, 'mdnn_start', g, 40)
    o = oest.OracleMDNN(input_dim=40, output_dim=2, output_lows=np.zeros(2),
                        output_highs=np.ones(2), n_gaussians=10, full_covariance=False,
                        hidden_layers=(24, 24), activation=torch.nn.Tanh,
                        lr=float(g['lr']), eps_noise=0.0)
    o.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w0.')})
    ids2 = np.random.RandomState(3).randint(0, 200, (40, 50))
    for ids, nu, bs in ((g['ids'], 100, 100), (ids2, 40, 50)):
        lg = m.run_training(summ, theta.to(DEV), nu, bs, ids_table=ids)
        lo = o.run_training(summ.cpu(), theta, nu, bs, ids_table=ids)
        np.testing.assert_allclose(lg['test_loss'], lo['test_loss'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(lg['train_loss'], lo['train_loss'], rtol=1e-4, atol=1e-5)


def test_numpy_rng_minibatch_stream(B):
    """Without ids_table the global numpy RNG is consumed exactly like the
    reference's n_updates sequential randint calls (mdnn.py:221)."""
    g = golden('chunk_mdnn_start.npz')
    B.MDNN.EPS_NOISE = 0.0
    summ = B.summary_start(torch.from_numpy(g['states']).to(DEV),
                           torch.from_numpy(g['actions']).to(DEV))
    theta = torch.from_numpy(g['theta']).to(DEV)
    m1 = _chunk_model(B, 'mdnn_start', g, 40)
    np.random.seed(21 + 7)           # the seed make_golden.py used for the ids
    l1 = m1.run_training(summ, theta, 100, 100)
    after = np.random.randint(0, 1 << 30)
    np.random.seed(21 + 7)
    for _ in range(100):
        np.random.randint(0, 200, 100)
    assert after == np.random.randint(0, 1 << 30)
    np.testing.assert_allclose(l1['test_loss'], g['test_loss'], rtol=1e-4, atol=1e-5)


def test_bayessim_on_reference_pendulum_fixture(B):
    """The reference's own regression data through BayesSim.run_training /
    predict (regression_tests.py:46-89, one run_training call)."""
    g = golden('pendulum_ref.npz')
    B.MDNN.EPS_NOISE = 0.0
    n = g['params'].shape[0]
    sa = torch.from_numpy(g['data']).reshape(n, -1, 4).to(DEV)
    cfg = {'modelClass': 'MDNN', 'summarizerFxn': 'summary_start', 'trainTrajLen': 10,
           'components': 10, 'hiddenLayers': (128, 128), 'lr': 5e-4}
    bsim = B.BayesSim(model_cfg=cfg, obs_dim=3, act_dim=1, params_dim=2,
                      params_lows=np.array([0.01] * 2), params_highs=np.array([2.0] * 2),
                      prior=None, proposal=None, device=DEV)
    bsim.model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items()
                                if k.startswith('w0.')})
    np.random.seed(9)
    logs = bsim.run_training(torch.from_numpy(g['params']).to(DEV), sa[:, :, :3].contiguous(),
                             sa[:, :, 3:].contiguous())
    np.testing.assert_allclose(logs['test_loss'], g['test_loss'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logs['train_loss'], g['train_loss'], rtol=1e-4, atol=1e-5)
    tsa = torch.from_numpy(g['true_data']).reshape(1, -1, 4).to(DEV)
    mog = bsim.predict(tsa[:, :, :3].contiguous(), tsa[:, :, 3:].contiguous())
    nll = -mog.eval(g['true_params'].reshape(1, -1).astype(np.float64), log=True)
    np.testing.assert_allclose(nll, g['mog.nll_true'], rtol=1e-4)
    np.testing.assert_allclose(mog.a, g['mog.a'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(np.stack([c.m for c in mog.xs]), g['mog.ms'], rtol=1e-4, atol=1e-6)


def test_bayessim_multi_trajectory_refit_runs(B):
    g = golden('pendulum_ref.npz')
    sa = torch.from_numpy(g['data'][:300]).reshape(300, -1, 4).to(DEV)
    cfg = {'modelClass': 'MDRFF_Matern32_2.0', 'summarizerFxn': 'summary_corrdiff',
           'trainTrajLen': 10, 'components': 3, 'hiddenLayers': (16,), 'lr': 1e-3,
           'nFeat': 64}
    np.random.seed(1)
    torch.manual_seed(1)
    bsim = B.BayesSim(model_cfg=cfg, obs_dim=3, act_dim=1, params_dim=2,
                      params_lows=np.array([0.01] * 2), params_highs=np.array([2.0] * 2),
                      prior=None, device=DEV, proposal=None)
    logs = bsim.fit(torch.from_numpy(g['params'][:300]).to(DEV), sa[:, :, :3].contiguous(),
                    sa[:, :, 3:].contiguous())
    assert len(logs) == 1 and len(logs[0]['test_loss']) == 6
    assert np.isfinite(logs[0]['test_loss']).all()
    tsa = torch.from_numpy(g['true_data']).reshape(1, -1, 4).to(DEV).repeat(2, 1, 1)
    mog = bsim.predict(tsa[:, :, :3].contiguous(), tsa[:, :, 3:].contiguous())
    assert mog.ndim == 2 and np.isfinite(mog.eval(np.array([[1.0, 0.5]]))).all()


@pytest.mark.parametrize('n_samples,tol', [(2000, 1e-4), (10000, 1e-3)])
def test_bayessim_multi_trajectory_refit_matches_oracle(B, n_samples, tol):
    """BayesSim.predict on several trajectories (bayes_sim.py:148-179): REFIT_SAMPLES points
    drawn from the per-trajectory MoGs are refitted by a fresh unconditional full-covariance
    MDNN (input_dim 1, trunk (128, 128)).  The same flow with the oracle's MDNN on the same
    samples, start weights (same torch-RNG order) and minibatch ids (same numpy-RNG order)
    must give the same posterior.  2000 samples = 100 updates: the teacher-forced horizon,
    1e-4; the reference's 10^4 samples = 500 updates: past the horizon where the reference
    stops reproducing itself, 1e-3."""
    from oracle import estimators as oest
    g = golden('pendulum_ref.npz')
    B.MDNN.EPS_NOISE = 0.0
    n = g['params'].shape[0]
    sa = torch.from_numpy(g['data']).reshape(n, -1, 4).to(DEV)
    cfg = {'modelClass': 'MDNN', 'summarizerFxn': 'summary_start', 'trainTrajLen': 10,
           'components': 3, 'hiddenLayers': (128, 128), 'lr': 5e-4, 'fullCovariance': True}
    torch.manual_seed(3)
    bsim = B.BayesSim(model_cfg=cfg, obs_dim=3, act_dim=1, params_dim=2,
                      params_lows=np.array([0.01] * 2), params_highs=np.array([2.0] * 2),
                      prior=None, proposal=None, device=DEV)
    np.random.seed(9)
    bsim.run_training(torch.from_numpy(g['params']).to(DEV), sa[:, :, :3].contiguous(),
                      sa[:, :, 3:].contiguous())
    tsa = sa[[5, 17, 40]]                               # three "real" trajectories
    st, ac = tsa[:, :, :3].contiguous(), tsa[:, :, 3:].contiguous()
    old = B.BayesSim.REFIT_SAMPLES
    B.BayesSim.REFIT_SAMPLES = n_samples
    try:
        np.random.seed(31)
        torch.manual_seed(32)
        mog = bsim.predict(st, ac)
        # the same flow, the refit by the oracle
        np.random.seed(31)
        torch.manual_seed(32)
        mogs = bsim.model.predict_MoGs(bsim._summarize(st, ac))
        o = oest.OracleMDNN(input_dim=1, output_dim=2, output_lows=np.array([0.01] * 2),
                            output_highs=np.array([2.0] * 2), n_gaussians=3, full_covariance=True,
                            hidden_layers=(128, 128), activation=torch.nn.Tanh, lr=5e-4,
                            eps_noise=0.0)
        smpls = np.concatenate([m.gen(n_samples=n_samples // 3) for m in mogs], axis=0)
        smpls = torch.from_numpy(smpls).float()
        inp = torch.zeros(smpls.shape[0], 1)
        o.run_training(inp, smpls, B.BayesSim.REFIT_EPOCHS * n_samples // 100, 100)
        w, ms, ls = o.predict_mog_params(inp[0:1])[0]
    finally:
        B.BayesSim.REFIT_SAMPLES = old
    ref = B.pdf.MoG(a=w, ms=ms, Ls=ls)
    np.testing.assert_allclose(mog.a, ref.a, rtol=tol, atol=tol * 1e-2)
    got_m, ref_m = np.stack([c.m for c in mog.xs]), np.stack([c.m for c in ref.xs])
    got_s, ref_s = np.stack([c.S for c in mog.xs]), np.stack([c.S for c in ref.xs])
    np.testing.assert_allclose(got_m, ref_m, rtol=tol, atol=tol * 1e-1)
    # covariances: relative to each component's largest entry (off-diagonals pass through 0)
    scale = np.abs(ref_s).max(axis=(1, 2), keepdims=True)
    assert (np.abs(got_s - ref_s) <= tol * scale + 1e-9).all(), (got_s - ref_s) / scale
    th = np.array([[1.0, 0.5]])
    np.testing.assert_allclose(mog.eval(th, log=True), ref.eval(th, log=True), rtol=tol, atol=tol)


def test_jitter_noise_path_is_finite_and_seeded(B):
    """EPS_NOISE = 1e-5 (the reference default) with in-kernel Philox noise:
    same torch seed -> identical run; different seed -> ~1e-5-level change."""
    g = golden('chunk_mdnn_start.npz')
    B.MDNN.EPS_NOISE = 1e-5
    summ = B.summary_start(torch.from_numpy(g['states']).to(DEV),
                           torch.from_numpy(g['actions']).to(DEV))
    theta = torch.from_numpy(g['theta']).to(DEV)
    res = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        m = _chunk_model(B, 'mdnn_start', g, 40)
        res.append(m.run_training(summ, theta, 100, 100, ids_table=g['ids'])['test_loss'])
    assert res[0] == res[1]
    assert res[0] != res[2]
    np.testing.assert_allclose(res[0], res[2], rtol=5e-3)
    np.testing.assert_allclose(res[0], g['test_loss'], rtol=5e-3)   # vs EPS=0 golden


# ---- size-independent properties at BASELINE sizes ---------------------------
def test_summary_start_full_size_rows(B):
    n, t, sd, ad = 100_000, 11, 211, 20
    gen = torch.Generator(device=DEV).manual_seed(0)
    s = torch.randn(n, t, sd, device=DEV, generator=gen)
    a = torch.rand(n, t, ad, device=DEV, generator=gen)
    out = B.summary_start(s, a)
    assert out.shape == (n, 2310)
    ref = torch.cat([s[:, :10], a[:, :10]], -1).reshape(n, -1)
    assert torch.equal(out, ref)             # pure copy: bit-exact, every row


def test_crosscorr_full_size_properties(B):
    n, t, sd, ad = 50_000, 51, 60, 8          # cfg3: Ant, 50k trajectories, 2.36 GB out
    gen = torch.Generator(device=DEV).manual_seed(1)
    s = torch.randn(n, t, sd, device=DEV, generator=gen)
    a = torch.rand(n, t, ad, device=DEV, generator=gen)
    out = B.summary_corrdiff(s, a)
    assert out.shape == (n, 11802)
    sf = (s[:, :5, 1:] - s[:, :5, :-1]).reshape(n, -1)
    af = a[:, :5].reshape(n, -1)
    idx = torch.randint(0, n, (64,), device=DEV)
    ref = (sf[idx].unsqueeze(2) * af[idx].unsqueeze(1)).reshape(64, -1)
    assert torch.equal(out[idx, :-2], ref)
    # linearity in the actions: corr(s, 2a) == 2 corr(s, a) exactly (power of 2)
    out2 = B.summary_corrdiff(s[:1000], 2 * a[:1000])
    assert torch.equal(out2[:, :-2], 2 * out[:1000, :-2])
    assert torch.equal(out2[:, -2:], out[:1000, -2:])
    torch.testing.assert_close(out[:, -2], sf.mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out[:, -1], sf.std(1), rtol=1e-5, atol=1e-6)


def test_signature_full_size_properties(B):
    n, t, sd, ad = 100_000, 11, 17, 4         # cfg4(B): 22 channels, depth 3 -> 11154 wide, 4.46 GB
    gen = torch.Generator(device=DEV).manual_seed(2)
    s = torch.randn(n, t, sd, device=DEV, generator=gen)
    a = torch.rand(n, t, ad, device=DEV, generator=gen)
    out = B.summary_signatory(s, a)
    d = 22
    assert out.shape == (n, d + d * d + d ** 3)
    l1 = out[:, :d]
    l2 = out[:, d:d + d * d].reshape(n, d, d)
    # level 1 = total increment; shuffle identity S1_i S1_j = S2_ij + S2_ji
    torch.testing.assert_close(l1[:, 1:18], s[:, -1] - s[:, 0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(l1[:, :, None] * l1[:, None, :], l2 + l2.transpose(1, 2),
                               rtol=1e-4, atol=2e-4)
    # invariance to re-parametrisation is broken by the time channel, but
    # reversing time negates odd levels of the non-time channels' level 1
    l3 = out[:, d + d * d:].reshape(n, d, d, d)
    sym = l3 + l3.permute(0, 1, 3, 2) + l3.permute(0, 2, 3, 1)   # ijk + ikj + kij
    torch.testing.assert_close(l2[:, :, :, None] * l1[:, None, None, :], sym,
                               rtol=1e-4, atol=1e-3)


def test_rff_gemm_full_size_against_torch(B):
    """cfg5-shaped projection at a scaled batch vs torch fp64 on sampled rows."""
    b, i, mf = 8192, 2310, 2048
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(b, i, device=DEV, generator=gen) * 0.3
    np.random.seed(4)
    rff = B.RFF(2 * mf, i, 4.0, quasi_random=False, device=DEV)
    feats = rff.to_features(x)
    assert feats.shape == (b, 2 * mf)
    idx = torch.randint(0, b, (32,), device=DEV)
    inner = x[idx].double() @ (rff.freqs.double() / rff.sigma.double()).T
    ref = rff.a * torch.cat([torch.cos(inner), torch.sin(inner)], 1)
    torch.testing.assert_close(feats[idx].double(), ref, rtol=0, atol=5e-6)
    # cos^2 + sin^2 = a^2 for every feature pair, every row
    torch.testing.assert_close(feats[:, :mf] ** 2 + feats[:, mf:] ** 2,
                               torch.full((b, mf), float(rff.a) ** 2, device=DEV),
                               rtol=1e-5, atol=1e-8)


def test_data_parallel_path_single_rank_equals_fused(B):
    """The data-parallel engine path (gradient graph -> RCCL all-reduce of the
    flat gradient buffer -> flat Adam graph) on a 1-rank nccl group must
    reproduce the fused single-rank path (same kernels for the gradient, Adam
    as a separate kernel instead of a GEMM epilogue)."""
    import os
    import torch.distributed as dist
    g = golden('chunk_mdrff_corrdiff.npz')
    B.MDNN.EPS_NOISE = 0.0
    states = torch.from_numpy(g['states']).to(DEV)
    actions = torch.from_numpy(g['actions']).to(DEV)
    theta = torch.from_numpy(g['theta']).to(DEV)
    summ = B.summary_corrdiff(states, actions)
    m1 = _chunk_model(B, 'mdrff_corrdiff', g, summ.shape[1])
    l1 = m1.run_training(summ, theta, 100, 100, ids_table=g['ids'])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        m2 = _chunk_model(B, 'mdrff_corrdiff', g, summ.shape[1]).enable_data_parallel()
        l2 = m2.run_training(summ, theta, 100, 100, ids_table=g['ids'])
    finally:
        if created:
            dist.destroy_process_group()
    np.testing.assert_allclose(l2['test_loss'], l1['test_loss'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(l2['train_loss'], l1['train_loss'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(l2['test_loss'], g['test_loss'], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(m2._flat, m1._flat, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('case', [
    dict(n=50, i=7, d=1, k=1, hidden=(8,), full=False, nu=3, bs=16, tf=0.2),
    dict(n=5, i=4, d=2, k=2, hidden=(8, 8), full=True, nu=1, bs=3, tf=0.2),
    dict(n=40, i=9, d=3, k=4, hidden=(16,), full=False, nu=7, bs=100, tf=0.5),   # batch > n_train
    dict(n=30, i=6, d=2, k=3, hidden=(), full=False, nu=5, bs=8, tf=0.0),        # no held-out rows
    dict(n=64, i=130, d=4, k=5, hidden=None, full=False, nu=6, bs=32, tf=0.25),  # MDRFF
    dict(n=33, i=5, d=6, k=2, hidden=(12,), full=True, nu=4, bs=10, tf=0.2),
    # the reference's default trunk: the persistent MDNN kernel (fit_persistent_mdnn.hip)
    dict(n=50, i=7, d=1, k=1, hidden=(128, 128), full=False, nu=3, bs=16, tf=0.2),
    dict(n=5, i=4, d=2, k=2, hidden=(128, 128), full=False, nu=1, bs=3, tf=0.2),
    dict(n=40, i=9, d=3, k=4, hidden=(128, 128), full=False, nu=7, bs=100, tf=0.5),
    dict(n=30, i=6, d=2, k=3, hidden=(128, 128), full=False, nu=5, bs=8, tf=0.0),
    dict(n=33, i=301, d=6, k=2, hidden=(128, 128), full=False, nu=4, bs=10, tf=0.2),
    dict(n=33, i=5, d=6, k=2, hidden=(128, 128), full=True, nu=4, bs=10, tf=0.2),
    dict(n=5, i=4, d=2, k=2, hidden=(128, 128), full=True, nu=1, bs=3, tf=0.2),
    # the unconditional refit of BayesSim.predict: one constant input (bayes_sim.py:160-176)
    dict(n=200, i=1, d=2, k=3, hidden=(128, 128), full=False, nu=6, bs=20, tf=0.2),
])
def test_run_training_edge_cases_match_oracle(B, case):
    """Shapes off the beaten path (K=1, D=1, one update, batch larger than the
    training split, empty held-out split, no trunk, full covariance) against
    the oracle from the same weights and ids, EPS_NOISE=0."""
    from oracle import estimators as oest
    B.MDNN.EPS_NOISE = 0.0
    c = case
    gen = torch.Generator().manual_seed(c['n'] * 13 + c['i'])
    x = torch.randn(c['n'], c['i'], generator=gen)
    y = torch.rand(c['n'], c['d'], generator=gen) * 2.0 + 1.0
    lows, highs = np.full(c['d'], 0.5), np.full(c['d'], 3.5)
    common = dict(input_dim=c['i'], output_dim=c['d'], output_lows=lows, output_highs=highs,
                  n_gaussians=c['k'], full_covariance=c['full'], lr=2e-3,
                  activation=torch.nn.Tanh)
    torch.manual_seed(5)
    np.random.seed(5)
    if c['hidden'] is None:
        m = B.MDRFF(n_feat=32, sigma=3.0, device=DEV, **common)
        o = oest.OracleMDRFF(n_feat=32, sigma=3.0, freqs=m.rff.freqs.cpu().numpy(),
                             eps_noise=0.0, **common)
    else:
        m = B.MDNN(hidden_layers=c['hidden'], device=DEV, **common)
        o = oest.OracleMDNN(hidden_layers=c['hidden'], eps_noise=0.0, **common)
    o.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    n_train = max(int(c['n'] * (1.0 - c['tf'])), 1)
    ids = np.random.RandomState(1).randint(0, n_train, (c['nu'], c['bs']))
    got = m.run_training(x.to(DEV), y.to(DEV), c['nu'], c['bs'], test_frac=c['tf'], ids_table=ids)
    if c['hidden'] == (128, 128):
        assert B._lib.load().bsig_fit_is_persistent(m._plan) == 2
    if c['tf'] == 0.0:
        # empty held-out split: torch 2.10's MultivariateNormal refuses an empty
        # batch (the reference would crash there); the HIP path logs NaN like
        # mean-of-empty.  Compare the updates themselves.
        opt = torch.optim.Adam(o.parameters(), lr=o.lr)
        yn = o.normalize_samples(y)
        every, ref = max(c['nu'] // 5, 1), {'train_loss': []}
        for it in range(c['nu']):
            opt.zero_grad()
            loss = o.mdn_loss_fn(*o(x[ids[it]]), yn[ids[it]])
            loss.backward()
            opt.step()
            if it % every == 0 or it + 1 == c['nu']:
                ref['train_loss'].append(loss.item())
    else:
        ref = o.run_training(x, y, c['nu'], c['bs'], test_frac=c['tf'], ids_table=ids)
    assert len(got['train_loss']) == len(ref['train_loss'])
    np.testing.assert_allclose(got['train_loss'], ref['train_loss'], rtol=1e-4, atol=1e-5)
    if c['tf'] == 0.0:
        assert all(np.isnan(v) for v in got['test_loss'])
    else:
        np.testing.assert_allclose(got['test_loss'], ref['test_loss'], rtol=1e-4, atol=1e-5)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), o.state_dict().items()):
        np.testing.assert_allclose(v1.cpu().numpy(), v2.numpy(), rtol=2e-3, atol=2e-5, err_msg=k1)


def test_nonfinite_input_raises_assertion(B):
    """The reference asserts isfinite on forward outputs / loss (mdnn.py:120-124,
    162-174); the device-side flag surfaces as the same AssertionError."""
    B.MDNN.EPS_NOISE = 0.0
    torch.manual_seed(0)
    m = B.MDNN(input_dim=4, output_dim=2, output_lows=np.zeros(2), output_highs=np.ones(2),
               n_gaussians=2, full_covariance=False, hidden_layers=(8,),
               activation=torch.nn.Tanh, lr=1e-3, device=DEV)
    x = torch.randn(20, 4, device=DEV)
    y = torch.rand(20, 2, device=DEV)
    y[3, 1] = float('nan')
    with pytest.raises(AssertionError):
        m.run_training(x, y, 5, 20, test_frac=0.2, ids_table=np.tile(np.arange(16), (5, 2))[:, :20] % 16)
    xb = x.clone()
    xb[0, 0] = float('inf')
    with pytest.raises(AssertionError):
        m.forward(xb)


def test_fit_defers_the_summarizer_finiteness_assert(B):
    """BayesSim.fit raises the cross-correlation summarizer's isfinite assert
    (summarizers.py:120) with the chunks' logs, after the last chunk is enqueued,
    instead of synchronising the host once per chunk; a direct summarizer call
    still asserts at once."""
    import bench
    cfg = dict(task='synthetic', model='MDNN', summarizer='summary_corrdiff', t=12, sd=6, ad=2,
               d=3, k=4, hidden=[128, 128], n_feat=0, pairs=2000)
    theta, states, actions = bench.synth_pairs(cfg, 2000, 3, DEV)
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    logs = bs.fit(theta, states, actions)
    assert len(logs) == 2 and all(np.isfinite(lg['test_loss']).all() for lg in logs)
    bad = states.clone()
    bad[1500, 3, 2] = float('inf')                    # second chunk, a held-out row
    with pytest.raises(AssertionError):
        B.summary_corrdiff(bad[1000:], actions[1000:])
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    with pytest.raises(AssertionError):
        bs.fit(theta, bad, actions)


def test_fit_projects_blocks_of_chunks_at_once(B):
    """BayesSim.fit with an MDRFF summarises and RFF-projects blocks of chunks in one
    launch each (the features are a pure function of the row) and hands every chunk
    its rows' features; BSIG_NO_FIT_PREPROJECT=1 projects per chunk inside
    bsig_fit_begin.  Same minibatches, same jitter streams: the logs agree to fp32
    GEMM summation order (another tile shape for the large product)."""
    import os
    import bench
    cfg = dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=5, ad=2,
               d=4, k=6, hidden=[], n_feat=512, pairs=3500)
    theta, states, actions = bench.synth_pairs(cfg, 3500, 3, DEV)
    out = []
    for env in ('0', '1'):
        os.environ['BSIG_NO_FIT_PREPROJECT'] = env
        try:
            bs = bench.build_gpu_model(B, cfg, DEV, 77)
            np.random.seed(11)
            out.append((bs.fit(theta, states, actions), bs.model._flat.clone()))
        finally:
            os.environ.pop('BSIG_NO_FIT_PREPROJECT', None)
    (la, fa), (lb, fb) = out
    assert len(la) == len(lb) == 4                       # 1000 + 1000 + 1000 + 500 pairs
    # (measured: bitwise equal at these shapes -- the projection sums a row's products in the same
    # order whatever the number of rows in the launch, tools/micro/preproject_tol_probe.py; the
    # tolerance leaves room for a split-K plan at another row count, nothing more)
    for x, y in zip(la, lb):
        np.testing.assert_allclose(x['train_loss'], y['train_loss'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x['test_loss'], y['test_loss'], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(fa, fb, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('variant', ['full_cov', 'no_persistent', 'no_inkernel_eval', 'no_graph'])
def test_fit_hands_each_chunk_its_own_features_on_graph_paths(B, variant):
    """The same comparison for MDRFF plans whose updates or evaluations replay HIP graphs (full
    covariance, BSIG_NO_PERSISTENT=1, evaluation graphs between the launches): the graphs carry
    the feature block's address as a kernel argument, and BayesSim.fit hands every chunk another
    slice of the block's features -- chunks 2..N must train on THEIR features (a graph captured
    on chunk 1's slice would not), bsig_fit_set_features / bsig_fit_begin."""
    import os
    import bench
    cfg = dict(task='synthetic', model='MDRFF', summarizer='summary_start', t=11, sd=5, ad=2,
               d=3, k=4, hidden=[], n_feat=256, pairs=3500, full=variant == 'full_cov')
    env = {'no_persistent': {'BSIG_NO_PERSISTENT': '1'},
           'no_inkernel_eval': {'BSIG_NO_INKERNEL_EVAL': '1'}}.get(variant, {})
    B.MDNN.USE_GRAPH = variant != 'no_graph'
    B.MDNN.EPS_NOISE = 0.0
    theta, states, actions = bench.synth_pairs(cfg, 3500, 3, DEV)
    out = []
    os.environ.update(env)
    try:
        for pre in ('0', '1'):
            os.environ['BSIG_NO_FIT_PREPROJECT'] = pre
            bs = bench.build_gpu_model(B, cfg, DEV, 77)
            np.random.seed(11)
            out.append((bs.fit(theta, states, actions), bs.model._flat.clone()))
            expect = 0 if variant in ('full_cov', 'no_persistent') else 1
            assert B._lib.load().bsig_fit_is_persistent(bs.model._plan) == expect
    finally:
        for k in list(env) + ['BSIG_NO_FIT_PREPROJECT']:
            os.environ.pop(k, None)
    (la, fa), (lb, fb) = out
    assert len(la) == len(lb) == 4
    # (measured: bitwise equal at these shapes -- the projection sums a row's products in the same
    # order whatever the number of rows in the launch, tools/micro/preproject_tol_probe.py; the
    # tolerance leaves room for a split-K plan at another row count, nothing more)
    for x, y in zip(la, lb):
        np.testing.assert_allclose(x['train_loss'], y['train_loss'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(x['test_loss'], y['test_loss'], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(fa, fb, rtol=1e-5, atol=1e-6)


def test_full_size_chunk_protocol_fit(B):
    """cfg5-shaped fit (ShadowHand MDRFF-4096, summary_start) over 20 chunks with
    the reference defaults (EPS_NOISE=1e-5, numpy-RNG ids): finite, 6+6 logs per
    chunk, held-out NLL improves, seeded runs bitwise reproducible."""
    import bench
    cfg = dict(bench.CONFIGS['cfg5'])
    theta, states, actions = bench.synth_pairs(cfg, 20_000, 7, DEV)
    finals = []
    for rep in range(2):
        torch.manual_seed(11)
        bs = bench.build_gpu_model(B, cfg, DEV, 11)
        np.random.seed(12)
        logs = bs.fit(theta, states, actions)
        assert len(logs) == 20 and all(len(lg['test_loss']) == 6 for lg in logs)
        flat = np.array([lg['test_loss'] for lg in logs])
        assert np.isfinite(flat).all()
        assert flat[-1, -1] < flat[0, 0] - 1.0          # the posterior sharpens
        finals.append((flat, bs.model._flat.clone()))
    np.testing.assert_array_equal(finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1])


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('name', ['cfg5', 'cfg2', 'cfg4'])
def test_baseline_shaped_chunk_matches_oracle(B, name, seed):
    """One teacher-forced 1000-pair chunk at the BASELINE shapes (ShadowHand
    MDRFF-4096 / Cartpole MDRFF-1024 / ShadowHand MDNN on the reference-rule depth-1
    signature, I=232) against the fp32 oracle: held-out NLL within the north-star 1e-4 relative,
    DIRECTLY, on six data seeds (the bench line's single figure per configuration is one draw --
    measured over these seeds 2e-6 on cfg5, 1e-6 on cfg2 / cfg4).  cfg4 feeds summary_signatory
    output (summarizers.py:144-168) into run_training.  (cfg4b: the test below.)"""
    import bench
    cfg = dict(bench.CONFIGS[name])
    theta, states, actions = bench.synth_pairs(cfg, 1000, seed, DEV)
    torch.set_num_threads(8)
    res = bench.nll_check(B, cfg, theta, states, actions, DEV)
    print('%s seed %d: |hip - oracle| / |oracle| = %.2e' % (name, seed, res['rel_diff']))
    assert res['rel_diff'] < 1e-4, res


def test_scaled_batch_fit_matches_oracle(B):
    """SURVEY.md 8(d) scaled-batch mode (one chunk, minibatch 8192, one optimizer) teacher-forced
    against the oracle: every logged loss within 1e-4 relative (measured 1e-7).  Exercises what
    only this regime reaches: a 65536-entry id table drawn after `bsig_fit_begin` is enqueued,
    the float4 staging copy, split-K head products with their 2048 exp partials, the finish
    kernel's 128-row slabs."""
    import bench
    cfg = dict(bench.CONFIGS['cfg5'])
    theta, states, actions = bench.synth_pairs(cfg, 20000, 3, DEV)
    torch.set_num_threads(8)
    res = bench.scaled_nll_check(B, cfg, theta, states, actions, DEV, 8192)
    assert res['max_rel_diff_all_logs'] < 1e-4, res


def test_scaled_batch_in_launch_combine_is_bitwise_the_reduce_kernel(B):
    """The head forward product of a large minibatch combines its K slices INSIDE the launch
    (csrc/gemm_wide.h: every workgroup writes its slab through, the last arriver of a row tile adds
    the slabs in slice order and the bias).  Same sums in the same order as the separate reduce kernel
    (BSIG_GEMM_NO_COMBINE=1): with EPS_NOISE = 0 (no exp partial sums, whose grouping differs) the two
    fits must agree BIT FOR BIT -- every loss, every weight -- over 40 updates of minibatch 8192 and
    their 20000-row evaluations: ~10^4 cross-workgroup hand-offs, any stale slab word would show."""
    import bench
    cfg = dict(bench.CONFIGS['cfg5'])
    theta, states, actions = bench.synth_pairs(cfg, 40000, 9, DEV)
    old = B.MDNN.EPS_NOISE
    B.MDNN.EPS_NOISE = 0.0
    try:
        ids = np.random.RandomState(8).randint(0, 32000, (40, 8192))
        runs = {}
        for env in ('0', '1'):
            os.environ['BSIG_GEMM_NO_COMBINE'] = env
            try:
                bs = bench.build_gpu_model(B, cfg, DEV, 78)
                logs = bs.model.run_training(bs._summarize(states, actions), theta, 40, 8192, ids_table=ids)
                torch.cuda.synchronize()
                runs[env] = (logs, bs.model._flat.clone())
            finally:
                os.environ.pop('BSIG_GEMM_NO_COMBINE', None)
        for key in ('train_loss', 'test_loss'):
            np.testing.assert_array_equal(np.asarray(runs['0'][0][key]), np.asarray(runs['1'][0][key]))
        assert torch.equal(runs['0'][1], runs['1'][1])
    finally:
        B.MDNN.EPS_NOISE = old


@pytest.mark.parametrize('batch', [2112, 2100])
def test_scaled_batch_fit_ragged_minibatches(B, batch):
    """The large-minibatch path at minibatch sizes that are no multiple of the whole-width kernels'
    tiles: 2112 = 33 x 64 rows (both products on gemm_wide_kernel, K slices of 1056 rows) and 2100
    (the forward product with a ragged last row tile, the gradient -- its contraction is no multiple
    of 32 -- on the generic kernels at the padded dO pitch): every logged loss within 1e-4 of the oracle."""
    import bench
    cfg = dict(bench.CONFIGS['cfg5'])
    theta, states, actions = bench.synth_pairs(cfg, 6000, 5, DEV)
    torch.set_num_threads(8)
    res = bench.scaled_nll_check(B, cfg, theta, states, actions, DEV, batch, n=6000, n_updates=6)
    assert res['max_rel_diff_all_logs'] < 1e-4, res


def _fp64_oracle(bench, cfg, in_dim, w0, freqs):
    o = bench.build_oracle(cfg, in_dim, 77, 0.0, freqs=freqs).double()
    o.load_state_dict({k: v.double() for k, v in w0.items()})
    o.output_lows, o.output_highs = o.output_lows.double(), o.output_highs.double()
    if freqs is not None:
        o.rff.freqs, o.rff.sigma = o.rff.freqs.double(), o.rff.sigma.double()
    return o


def _bracket_chunk(B, name, seed, lazy, n_updates=100, must_factor=None, f32_threads=(8,), hip_env=None):
    """One teacher-forced chunk three ways: HIP, the fp32 oracle, the fp64 oracle (same start
    weights, same ids, EPS_NOISE = 0).  Returns the three log dicts and the GPU model (with
    several f32_threads: a list of fp32 oracle logs, one per thread count -- each thread count is
    another summation order of the same fp32 arithmetic)."""
    import bench
    from oracle import summarize as osum
    B.MDNN.EPS_NOISE = 0.0
    cfg = dict(bench.CONFIGS[name])
    torch.set_num_threads(8)
    theta, states, actions = bench.synth_pairs(cfg, 1000, seed, DEV)
    ids = np.random.RandomState(5).randint(0, 800, (n_updates, 100))
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    w0 = {k: v.cpu().clone() for k, v in bs.model.state_dict().items()}
    summ = bs._summarize(states, actions, lazy=lazy)
    if lazy:      # the factor rows must reach the kernel as factor rows (f2), not expanded
        assert isinstance(summ, B.summarizers.CrossCorrFactors)
    os.environ.update(hip_env or {})
    try:
        hip = bs.model.run_training(summ, theta, n_updates, 100, ids_table=ids)
        torch.cuda.synchronize()
    finally:
        for k in (hip_env or {}):
            os.environ.pop(k, None)
    if lazy if must_factor is None else must_factor:
        lib = B._lib.load()
        assert lib.bsig_fit_is_persistent(bs.model._plan) == 2
        assert lib.bsig_fit_accepts_factor_rows(bs.model._plan, summ.s_dim, summ.a_dim) == 1
    s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
    f32s = []
    for nt in f32_threads:
        torch.set_num_threads(nt)
        o32 = bench.build_oracle(cfg, s_cpu.shape[1], 77, 0.0)
        o32.load_state_dict(w0)
        f32s.append(o32.run_training(s_cpu, theta.cpu(), n_updates, 100, ids_table=ids))
        del o32
    torch.set_num_threads(8)
    o64 = _fp64_oracle(bench, cfg, s_cpu.shape[1], w0, None)
    f64 = o64.run_training(s_cpu.double(), theta.cpu().double(), n_updates, 100, ids_table=ids)
    return hip, (f32s[0] if len(f32s) == 1 else f32s), f64, bs


def _assert_bracket(hip, f32, f64):
    """|hip - f64| <= |cpu_f32 - f64| + 1e-4 |f64| at every logging point."""
    for key in ('test_loss', 'train_loss'):
        h, a, r = (np.asarray(v[key], dtype=np.float64) for v in (hip, f32, f64))
        bound = np.abs(a - r) + 1e-4 * np.abs(r) + 1e-6
        assert (np.abs(h - r) <= bound).all(), (key, h - r, a - r)


@pytest.mark.parametrize('lazy', [False, True])
@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_cfg3_chunk_within_reference_fp32_noise_of_fp64(B, seed, lazy):
    """cfg3 (Ant MDNN on 11802-wide cross-correlations): the first layer sums 11802 fp32
    products per output, so two fp32 evaluation orders of the SAME chunk differ by more than
    1e-4 in the held-out NLL -- the reference's own fp32 CPU path included.  The yardstick is
    therefore the same teacher-forced chunk run by the oracle in fp64: the HIP path must be
    as close to it as the reference's fp32 path is, plus the north-star 1e-4:
        |hip - f64| <= |cpu_f32 - f64| + 1e-4 |f64|     at every logging point.
    lazy: the summaries reach the update kernel as cross-correlation FACTOR rows (f2; what
    BayesSim.fit runs) -- the mdnn_updates_kernel<., FAC> path against the oracle itself.
    Beside the bracket, the direct north-star statement on the final held-out NLL,
    |hip - cpu_f32| <= 1e-4 |cpu_f32| (summarizers.py:106-119 into mdnn.py:228-242), wherever
    the reference's own fp32 path reproduces the fp64 chunk to half that tolerance (seed 3, the
    bench line's seed: 7e-6; seed 4: the fp32 oracle itself is 2e-4 from fp64 -- the bracket
    is the statement there)."""
    hip, f32, f64, _ = _bracket_chunk(B, 'cfg3', seed, lazy)
    _assert_bracket(hip, f32, f64)
    g, r, r64 = hip['test_loss'][-1], f32['test_loss'][-1], f64['test_loss'][-1]
    print('cfg3 seed %d lazy %d: |hip-cpu32|/|cpu32| = %.2e, |cpu32-f64|/|f64| = %.2e'
          % (seed, lazy, abs(g - r) / abs(r), abs(r - r64) / abs(r64)))
    if abs(r - r64) <= 0.5e-4 * abs(r64):
        assert abs(g - r) <= 1e-4 * abs(r), (g, r, r64)
    if seed == 3:
        assert abs(g - r) <= 1e-4 * abs(r), (g, r, r64)


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_cfg4b_chunk_within_reference_fp32_noise_of_fp64(B, seed):
    """cfg4b (ShadowHand MDNN on the depth-3 signature of 22 channels, I = 11154: as wide a first layer
    as cfg3's).  On five of the six seeds every fp32 path reproduces the fp64 chunk to 1e-5 and the
    direct north-star statement holds; on seed 1 the chunk is ill-conditioned for ANY fp32 arithmetic:
    the reference's own fp32 path ends 3.0e-4 (relative) off its fp64 self, round 5's row arithmetic
    2.7e-4, round 6's (another summation order in the row-wise NLL, same formulas and IEEE operations)
    5.8e-4 (`tools/micro/diag_mdnn_rows_parity.py`).  Round 5 asserted 1e-4 DIRECTLY on six seeds and
    passed seed 1 by the luck of landing next to the fp32 oracle; the statement that survives a change of
    summation order is the one cfg3 already uses -- HIP must be one more member of the family of the
    reference's fp32 evaluation orders:
        |hip - f64| <= 2 max_i |cpu_f32_i - f64| + 1e-4 |f64|     at every logging point
    over two evaluation orders of the fp32 oracle (8 threads / 1 thread), and DIRECTLY within 1e-4 of
    the fp32 oracle wherever that oracle is itself within 0.5e-4 of fp64 (a relaxation of the north
    star on the ill-conditioned seed, named as such here and in bench.py's nll_criterion)."""
    hip, f32s, f64, _ = _bracket_chunk(B, 'cfg4b', seed, False, must_factor=False, f32_threads=(8, 1))
    for key in ('test_loss', 'train_loss'):
        h, r = (np.asarray(v[key], dtype=np.float64) for v in (hip, f64))
        dev = np.max([np.abs(np.asarray(a[key], dtype=np.float64) - r) for a in f32s], axis=0)
        assert (np.abs(h - r) <= 2.0 * dev + 1e-4 * np.abs(r) + 1e-6).all(), (key, h - r, dev)
    g, r64 = hip['test_loss'][-1], f64['test_loss'][-1]
    r = f32s[0]['test_loss'][-1]
    d32 = max(abs(a['test_loss'][-1] - r64) for a in f32s)
    print('cfg4b seed %d: |hip-cpu32|/|cpu32| = %.2e, max |cpu32-f64|/|f64| = %.2e' % (seed, abs(g - r) / abs(r), d32 / abs(r64)))
    if d32 <= 0.5e-4 * abs(r64):
        assert abs(g - r) <= 1e-4 * abs(r), (g, r, r64)


@pytest.mark.parametrize('lazy', [False, True])
def test_cfg3_chunk_from_factor_rows_matches_oracle_directly(B, lazy):
    """bench.nll_check on cfg3 (what the bench line's per_config reports) with the summaries as
    factor rows and as materialised rows: the FAC kernel path vs the fp32 oracle at 1e-4."""
    import bench
    cfg = dict(bench.CONFIGS['cfg3'])
    theta, states, actions = bench.synth_pairs(cfg, 1000, 3, DEV)
    torch.set_num_threads(8)
    res = bench.nll_check(B, cfg, theta, states, actions, DEV, lazy=lazy)
    assert res['rel_diff'] < 1e-4, res


def _horizon(x, r, theta):
    """Index of the first logging point where x has left r by more than theta (relative)."""
    x, r = np.asarray(x, dtype=np.float64), np.asarray(r, dtype=np.float64)
    off = np.nonzero(np.abs(x - r) > theta * np.abs(r))[0]
    return int(off[0]) if off.size else len(r)


def _reference_orders(bench, cfg, s_cpu, theta_cpu, w0, ids, n_updates, n_perms):
    """The reference's fp32 arithmetic on one teacher-forced chunk in 1 + n_perms EVALUATION ORDERS: as
    it is, and with the input columns (and the first layer's weight columns with them) permuted -- the
    same network, the same fp32 operations, another grouping of the 56-105 k-term sums.  (8 threads
    against 1 is NOT another order: same blocking, the logs agree to three digits.)  Returns the log
    dicts and the end weights (first-layer columns back in the original order)."""
    k1 = [k for k, v in w0.items() if v.dim() == 2 and v.shape[1] == s_cpu.shape[1]][0]
    logs, weights = [], []
    for i in range(1 + n_perms):
        o = bench.build_oracle(cfg, s_cpu.shape[1], 77, 0.0)
        w, x = dict(w0), s_cpu
        perm = None
        if i > 0:
            perm = torch.from_numpy(np.random.RandomState(100 + i).permutation(s_cpu.shape[1]))
            w[k1] = w0[k1][:, perm].contiguous()
            x = s_cpu[:, perm].contiguous()
        o.load_state_dict(w)
        logs.append(o.run_training(x, theta_cpu, n_updates, 100, ids_table=ids))
        sd = {k: v.double() for k, v in o.state_dict().items()}
        if perm is not None:
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())
            sd[k1] = sd[k1][:, inv]
        weights.append(sd)
        del o
    return logs, weights


def _wide_chunk(B, name, seed, n_updates, n_perms, hip_env=None, lazy=True):
    """HIP, the reference's fp32 arithmetic in 1 + n_perms evaluation orders, and the fp64 oracle on one
    teacher-forced chunk (same start weights, same ids, EPS_NOISE = 0)."""
    import bench
    from oracle import summarize as osum
    B.MDNN.EPS_NOISE = 0.0
    cfg = dict(bench.CONFIGS[name])
    torch.set_num_threads(8)
    theta, states, actions = bench.synth_pairs(cfg, 1000, seed, DEV)
    ids = np.random.RandomState(5).randint(0, 800, (100, 100))[:n_updates]
    bs = bench.build_gpu_model(B, cfg, DEV, 77)
    w0 = {k: v.cpu().clone() for k, v in bs.model.state_dict().items()}
    summ = bs._summarize(states, actions, lazy=lazy)
    os.environ.update(hip_env or {})
    try:
        hip = bs.model.run_training(summ, theta, n_updates, 100, ids_table=ids)
        torch.cuda.synchronize()
    finally:
        for k in (hip_env or {}):
            os.environ.pop(k, None)
    persistent = int(B._lib.load().bsig_fit_is_persistent(bs.model._plan))
    w_hip = {k: v.cpu().double() for k, v in bs.model.state_dict().items()}
    s_cpu = osum.SUMMARIZERS[cfg['summarizer']](states.cpu(), actions.cpu())
    f32_logs, f32_w = _reference_orders(bench, cfg, s_cpu, theta.cpu(), w0, ids, n_updates, n_perms)
    o64 = _fp64_oracle(bench, cfg, s_cpu.shape[1], w0, None)
    f64 = o64.run_training(s_cpu.double(), theta.cpu().double(), n_updates, 100, ids_table=ids)
    w64 = {k: v.clone() for k, v in o64.state_dict().items()}
    del bs, o64
    return hip, f32_logs, f64, w_hip, f32_w, w64, persistent


def _assert_envelope(hip, f32_logs, f64, factor=2.0, sigmas=3.0):
    """HIP as one more member of the family of the reference's fp32 evaluation orders (EIGHT of them:
    round 4 had factor 3 over four).  With d_i = |cpu_f32(order i) - f64| and h = |hip - f64| at a
    logging point, h must be within
        max( factor * max_i d_i ,  10 ** (mean_i log10 d_i + sigmas * std_i log10 d_i) )  +  1e-4 |f64|.
    The first term is the plain envelope (factor 2).  The second is the same statement in the only
    scale these chunks have: their deviation from the fp64 chunk grows 10- to 20-fold per 20 updates,
    so the orders' deviations at one logging point are spread over a decade or more (log-normal to
    the eye) and "twice the furthest of eight" is a lead of a few updates, not a different trajectory
    -- (cfg/anymal.yaml, seed 4) sits at 3.0 x the furthest order after 60 updates and INSIDE the
    orders after 80 and 100.  A path that leaves the family (a wrong gradient, a lost update) is out
    by orders of magnitude at every later point and fails both."""
    for key in ('test_loss', 'train_loss'):
        r = np.asarray(f64[key], dtype=np.float64)
        h = np.abs(np.asarray(hip[key], dtype=np.float64) - r)
        d = np.array([np.abs(np.asarray(f[key], dtype=np.float64) - r) for f in f32_logs])   # [orders, points]
        env = d.max(axis=0)
        ok = d.min(axis=0) > 0.0                              # (an order that is exact at a point: no log statistics)
        lg = np.log10(np.where(ok[None, :], d, 1.0))
        stat = np.where(ok, 10.0 ** (lg.mean(axis=0) + sigmas * lg.std(axis=0, ddof=1)), 0.0)
        bound = np.maximum(factor * env, stat) + 1e-4 * np.abs(r) + 1e-6
        print('  %-10s |hip-f64|/|f64|            %s\n  %-10s furthest of %d orders       %s\n  %-10s mean + %g sigma (log10)    %s' %
              (key, h / np.abs(r), '', len(f32_logs), env / np.abs(r), '', sigmas, stat / np.abs(r)))
        assert (h <= bound).all(), (key, h / np.abs(r), env / np.abs(r), stat / np.abs(r))


@pytest.mark.parametrize('name,seed', [('anymal_yaml', 3), ('anymal_yaml', 4), ('shadow_more', 4)])
def test_wide_crosscorr_chunk_stays_inside_the_reference_fp32_envelope(B, name, seed):
    """cfg/anymal.yaml (I = 56402) and cfg/shadow_hand_more.yaml (I = 105002) as shipped, 100
    teacher-forced updates through the path BayesSim.fit takes (factor rows into the streamed first
    layer).  These chunks are ill-conditioned in fp32: most of the 56-105 k inputs are near-zero
    products whose first-layer gradients are sums with heavy cancellation, Adam turns a noise-level
    gradient into a full-size step, and the deviation from the fp64 chunk grows 20- to 100-fold per 20
    updates -- for the reference's OWN fp32 arithmetic: with the input columns permuted (another
    grouping of the same fp32 sums) its held-out NLL is 5e-5 .. 1e-3 from the fp64 chunk after 40 and
    60 updates where the unpermuted run happens to sit at 2e-6 and 3e-4 (tools/parity_wide_diag.py;
    8 threads against 1 is no second order, the logs agree to three digits).  What is asserted:
      * the ENVELOPE -- at every logging point HIP is no further from the fp64 chunk than TWICE
        the furthest of EIGHT reference evaluation orders, or than their log-normal spread allows
        (mean + 3 sigma of log10 deviation: _assert_envelope) (+ the north-star 1e-4);
      * the HORIZON -- HIP stays within 1e-3 of the fp64 chunk at least as long as the earliest of
        those orders (no slack), and never leaves before update 40;
    and, with 20 updates, every loss within 1e-4 of the fp32 oracle
    (test_wide_crosscorr_chunk_20_updates_matches_oracle)."""
    hip, f32_logs, f64, _, _, _, persistent = _wide_chunk(B, name, seed, 100, n_perms=7)
    assert persistent == 2
    _assert_envelope(hip, f32_logs, f64)
    for key in ('test_loss', 'train_loss'):
        t_hip = _horizon(hip[key], f64[key], 1e-3)
        t_f32 = [_horizon(f[key], f64[key], 1e-3) for f in f32_logs]
        print('%s seed %d %s: horizon hip %d, reference orders %s' % (name, seed, key, t_hip, t_f32))
        assert t_hip >= min(t_f32), (key, t_hip, t_f32)
        assert t_hip >= 2, (key, hip[key], f64[key])       # never before update 40


@pytest.mark.parametrize('name,seed', [('anymal_yaml', 4), ('shadow_more', 4)])
def test_wide_chunk_weights_as_close_to_fp64_as_the_reference_orders(B, name, seed):
    """The other half of the argument, as an assertion (it used to live in
    tools/parity_weights_probe.py): BEFORE the fp32 paths part, the HIP weights are as close to the
    fp64 chunk's as the reference's fp32 weights are -- per parameter tensor,
        mean|W_hip - W_f64| <= c * max over the reference evaluation orders of mean|W_cpu32 - W_f64|
    with c = 2 after 1, 5 AND 20 updates -- unless the orders THEMSELVES are further apart than that
    for the tensor: then c is their own spread, max / min over the orders (after 20 updates the chunk's
    amplification has set in and the orders differ up to four-fold: the test prints the spread it
    used); + 1e-9 for tensors every path gets right to the last bit.  Measured in round 4: after 1
    update every ratio is 0.7-1.2; after 5, 0.6-1.3 (shadow_hand_more) and 1.1-1.9 (anymal); after 20,
    1.1 and 3.2 at worst."""
    for n_up in (1, 5, 20):
        _, _, _, w_hip, f32_w, w64, persistent = _wide_chunk(B, name, seed, n_up, n_perms=3)
        assert persistent == 2
        for k in w_hip:
            dh = float((w_hip[k] - w64[k]).abs().mean())
            dcs = [float((w[k] - w64[k]).abs().mean()) for w in f32_w]
            spread = max(dcs) / max(min(dcs), 1e-300)
            c = max(2.0, spread)
            print('%s seed %d, %2d updates, %-16s mean|hip-f64| %.3e = %.2f x the furthest order; orders %s '
                  '(their spread %.2f -> c = %.2f)' % (name, seed, n_up, k, dh, dh / max(max(dcs), 1e-300),
                                                      ' '.join('%.3e' % d for d in dcs), spread, c))
            assert dh <= c * max(dcs) + 1e-9, (name, seed, n_up, k, dh, dcs)


@pytest.mark.parametrize('name,seed', [('anymal_yaml', 3)])
def test_wide_chunk_with_fp64_sums_is_no_closer_to_fp64(B, name, seed):
    """SURVEY.md section 7 ("hard parts") asks for an instantiation of the wide first-layer products
    that accumulates in fp64, to extend the horizon.  Built (BSIG_DEBUG_F64_ACC_MIN_K=64: every product
    of the per-phase path with a contraction of 64 terms or more -- the 56 402-term forward sums, the
    100-term dW sums -- has its products and its sum in fp64, rounded to fp32 once:
    csrc/gemm_f32.hip gemm_f64acc_kernel) and MEASURED: the chunk leaves the fp64 trajectory exactly as
    the fp32 paths do (held-out NLL 7e-5 / 9e-4 off after 40 / 60 updates against 4e-5 / 8e-4 with fp32
    sums).  The summation ORDER is therefore not what the chunk amplifies: the cancelling sums amplify
    the rounding of their fp32 INPUTS (the back-propagated dz1, the activations), which only a run with
    everything in fp64 -- the oracle's -- removes.  Asserted: that run is one more member of the
    reference's fp32 envelope (and the debug path computes the same chunk: first logging points at
    1e-4)."""
    env = {'BSIG_NO_PERSISTENT': '1', 'BSIG_DEBUG_F64_ACC_MIN_K': '64'}
    hip, f32_logs, f64, _, _, _, persistent = _wide_chunk(B, name, seed, 100, n_perms=7, hip_env=env, lazy=False)
    assert persistent == 0
    _assert_envelope(hip, f32_logs, f64)
    for key in ('test_loss', 'train_loss'):
        np.testing.assert_allclose(hip[key][:2], f64[key][:2], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('name', ['anymal_yaml', 'shadow_more'])
def test_wide_crosscorr_chunk_20_updates_matches_oracle(B, name):
    """... and before the fp32 paths part: 20 teacher-forced updates, every loss within the
    north-star 1e-4 of the fp32 oracle."""
    hip, f32, _, _ = _bracket_chunk(B, name, 3, lazy=True, n_updates=20, must_factor=True)
    for key in ('test_loss', 'train_loss'):
        np.testing.assert_allclose(hip[key], f32[key], rtol=1e-4, atol=1e-6)
