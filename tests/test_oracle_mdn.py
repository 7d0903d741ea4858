"""Pin the oracle's MDNN / MDRFF / density restatement against outputs of the
reference itself (tests/golden/mdn_step_*.npz, chunk_*.npz, pdf_cases.npz,
pendulum_ref.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import density as oden
from oracle import estimators as oest
from oracle import summarize as osum

STEP_CASES = {
    'diag': dict(cls='MDNN', input_dim=40, output_dim=2, n_gaussians=10,
                 full_covariance=False, hidden_layers=(24, 24), lr=5e-4),
    'full': dict(cls='MDNN', input_dim=12, output_dim=5, n_gaussians=3,
                 full_covariance=True, hidden_layers=(16,), lr=1e-3),
    'mdrff': dict(cls='MDRFF', input_dim=302, output_dim=13, n_gaussians=4,
                  full_covariance=False, lr=1e-3, n_feat=64, sigma=4.0),
    'clamp': dict(cls='MDNN', input_dim=6, output_dim=3, n_gaussians=5,
                  full_covariance=False, hidden_layers=(8,), lr=1e-3),
}


def build(case, g, eps_noise, prefix='w0.'):
    kw = dict(STEP_CASES[case])
    cls = kw.pop('cls')
    d = kw['output_dim']
    kw.update(output_lows=np.zeros(d), output_highs=np.ones(d),
              activation=torch.nn.Tanh, eps_noise=eps_noise)
    if cls == 'MDRFF':
        m = oest.OracleMDRFF(freqs=g['rff.freqs'], **kw)
    else:
        m = oest.OracleMDNN(**kw)
    m.load_state_dict({k[len(prefix):]: torch.from_numpy(v)
                       for k, v in g.items() if k.startswith(prefix)})
    return m


@pytest.mark.parametrize('tag', ['diag_eps0', 'diag_eps1e5', 'full_eps0',
                                 'full_eps1e5', 'mdrff_eps0', 'mdrff_eps1e5',
                                 'clamp_eps1e5'])
def test_one_step_matches_reference(tag):
    g = golden('mdn_step_%s.npz' % tag)
    case = tag.split('_')[0]
    m = build(case, g, float(g['eps_noise']))
    x, y = torch.from_numpy(g['x']), torch.from_numpy(g['y'])
    noise = torch.from_numpy(g['noise'])
    if case == 'mdrff':
        np.testing.assert_array_equal(m.rff.to_features(x).numpy(),
                                      g['rff.features'])
    opt = torch.optim.Adam(m.parameters(), lr=float(g['lr']))
    opt.zero_grad()
    w, mu, l_d, low = m(x, noise=noise)
    loss = m.mdn_loss_fn(w, mu, l_d, low, y)
    loss.backward()
    np.testing.assert_array_equal(w.detach().numpy(), g['weights'])
    np.testing.assert_array_equal(mu.detach().numpy(), g['mu'])
    np.testing.assert_array_equal(l_d.detach().numpy(), g['L_d'])
    if 'L' in g:
        np.testing.assert_array_equal(low.detach().numpy(), g['L'])
    assert loss.item() == pytest.approx(float(g['loss']), rel=1e-7)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g['grad.' + k],
                                   rtol=1e-5, atol=1e-9)
    opt.step()
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g['w1.' + k], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('tag', ['diag_eps1e5', 'full_eps1e5', 'clamp_eps1e5',
                                 'mdrff_eps0'])
def test_closed_form_head_matches_reference(tag):
    """fp64 closed forms vs the reference's autograd."""
    g = golden('mdn_step_%s.npz' % tag)
    case = tag.split('_')[0]
    m = build(case, g, float(g['eps_noise'])).double()
    if m.output_lows is not None:
        m.output_lows = m.output_lows.double()
    x = torch.from_numpy(g['x']).double()
    if case == 'mdrff':
        m.rff.freqs, m.rff.sigma = m.rff.freqs.double(), m.rff.sigma.double()
        h = m.rff.to_features(x)
    else:
        h = m.net(x)
    heads = [m.pi(h), m.mu(h), m.Diag(h)] + ([m.Lower(h)] if m.Lower is not None else [])
    o = torch.cat(heads, dim=1).detach().requires_grad_(True)
    kk, d = m.n_gaussians, m.output_dim
    loss, d_o, aux = oest.mdn_head_closed_form(
        o.detach().numpy(), g['y'], d, kk, m.Lower is not None,
        eps_noise=float(g['eps_noise']), noise=g['noise'])
    np.testing.assert_allclose(aux['weights'], g['weights'], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(aux['l_d'], g['L_d'], rtol=2e-6)
    assert loss == pytest.approx(float(g['loss']), rel=2e-6)
    # gradient w.r.t. the head pre-activations, via fp64 autograd of the oracle
    w = torch.softmax(o[:, :kk], -1).clamp(oest.MIN_WEIGHT, 1.0)
    w = w / w.sum(1, keepdim=True)
    mu = o[:, kk:kk + d * kk].reshape(-1, d, kk)
    l_d = torch.exp(o[:, kk + d * kk:kk + 2 * d * kk]).reshape(-1, d, kk)
    l_d = l_d + torch.from_numpy(g['noise']).double() * (float(g['eps_noise']) * l_d.mean())
    low = None
    if m.Lower is not None:
        low = o[:, kk + 2 * d * kk:].reshape(-1, m.L_size, kk)
    # fp64 loss without the fp32 result buffer
    rows, cols = np.tril_indices(d, -1)
    res = []
    for k in range(kk):
        tri = torch.diag_embed(l_d[:, :, k])
        if low is not None:
            tri[:, rows, cols] = low[:, :, k]
        lp = torch.distributions.MultivariateNormal(
            mu[:, :, k], scale_tril=tri).log_prob(torch.from_numpy(g['y']).double())
        res.append(lp.clamp(-oest.LL_LIMIT, oest.LL_LIMIT)
                   + w[:, k].clamp(oest.MIN_WEIGHT, 1.0).log())
    l64 = -torch.logsumexp(torch.stack(res, 1), 1).mean()
    l64.backward()
    assert loss == pytest.approx(l64.item(), rel=1e-12)
    np.testing.assert_allclose(d_o, o.grad.numpy(), rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize('tag,kw', [
    ('mdnn_start', dict(cls='MDNN', summarizer='summary_start', d=2, k=10,
                        hidden=(24, 24), full=False)),
    ('mdnn_corrdiff_full', dict(cls='MDNN', summarizer='summary_corrdiff', d=3,
                                k=3, hidden=(16, 16), full=True)),
    ('mdrff_corrdiff', dict(cls='MDRFF', summarizer='summary_corrdiff', d=4,
                            k=4, hidden=[], full=False)),
    # bayes_sim.py:72-81: 'MDRFF_Matern32_2.0' -> Student-t frequencies, sigma 2.0
    ('mdrff_matern32', dict(cls='MDRFF', summarizer='summary_corrdiff', d=4,
                            k=4, hidden=[], full=False, kernel='Matern32', sigma=2.0)),
])
def test_teacher_forced_chunk_matches_reference(tag, kw):
    g = golden('chunk_%s.npz' % tag)
    states, actions = torch.from_numpy(g['states']), torch.from_numpy(g['actions'])
    theta = torch.from_numpy(g['theta'])
    summ = osum.SUMMARIZERS[kw['summarizer']](states, actions)
    np.testing.assert_array_equal(summ.numpy(), g['summaries'])
    d = kw['d']
    common = dict(input_dim=summ.shape[1], output_dim=d, output_lows=np.zeros(d),
                  output_highs=np.ones(d), n_gaussians=kw['k'],
                  full_covariance=kw['full'], lr=float(g['lr']),
                  activation=torch.nn.Tanh, eps_noise=0.0)
    if kw['cls'] == 'MDRFF':
        m = oest.OracleMDRFF(n_feat=200, sigma=kw.get('sigma', 4.0),
                             kernel=kw.get('kernel', 'RBF'), freqs=g['rff.freqs'], **common)
        np.testing.assert_array_equal(m.rff.sigma.numpy(), g['rff.sigma'])
    else:
        m = oest.OracleMDNN(hidden_layers=kw['hidden'], **common)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items()
                       if k.startswith('w0.')})
    logs = m.run_training(summ, theta, int(g['n_updates']), int(g['batch']),
                          ids_table=g['ids'])
    np.testing.assert_allclose(logs['train_loss'], g['train_loss'], rtol=2e-5)
    np.testing.assert_allclose(logs['test_loss'], g['test_loss'], rtol=2e-5)
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g['w1.' + k], rtol=2e-3, atol=2e-5)
    n_train = int(states.shape[0] * 0.8)
    a, ms, ls = m.predict_mog_params(summ[n_train:n_train + 1])[0]
    np.testing.assert_allclose(a, g['mog.a'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(np.stack(ms), g['mog.ms'], rtol=1e-4, atol=1e-6)
    mean, covs = oden.mog_moments(ms, ls)
    np.testing.assert_allclose(covs, g['mog.Ss'], rtol=1e-3, atol=1e-8)
    nll = -oden.mog_logpdf(a, ms, ls, g['theta'][n_train:n_train + 1])
    np.testing.assert_allclose(nll, g['mog.nll_true'], rtol=1e-4)


RFF_VARIANTS = ['cos_rbf'] + ['%s.%s' % (k, m) for k in ('Matern12', 'Matern32', 'Matern52',
                                                           'Laplace')
                              for m in ('cossin', 'cos')]


def rff_variant_args(tag):
    """Constructor arguments make_golden.gen_rff_variants used for a case."""
    if tag == 'cos_rbf':
        return dict(n_feat=64, d=302, sigma=4.0, cos_only=True, kernel='RBF'), 'x'
    kern, mode = tag.split('.')
    return dict(n_feat=48, d=150, sigma=[0.5 + 0.01 * j for j in range(150)],
                cos_only=(mode == 'cos'), kernel=kern), 'x150'


@pytest.mark.parametrize('tag', RFF_VARIANTS)
def test_rff_variants_match_reference(tag):
    """f4: cos-only feature map + offsets (rff.py:98-102,122-126) and the
    Matern / Laplace Student-t draws (rff.py:151-184), against the reference's
    own RFF objects: frequencies and offsets bit-equal (same numpy-RNG call
    order: normal, chisquare, then rand), global RNG left in the same state."""
    g = golden('rff_variants.npz')
    args, xk = rff_variant_args(tag)
    np.random.seed(int(g[tag + '.seed']))
    r = oest.OracleRFF(**args)
    assert int(np.random.randint(0, 1 << 30)) == int(g[tag + '.rng_after'])
    np.testing.assert_array_equal(r.freqs.numpy(), g[tag + '.freqs'])
    assert float(r.a) == float(g[tag + '.a'])
    if args['cos_only']:
        np.testing.assert_array_equal(r.offset.numpy(), g[tag + '.offset'])
    feats = r.to_features(torch.from_numpy(g[xk])).numpy()
    assert feats.shape == g[tag + '.features'].shape == (12, args['n_feat'])
    np.testing.assert_allclose(feats, g[tag + '.features'], rtol=0, atol=1e-6)


def test_numpy_ids_stream_is_batchable():
    """n_updates sequential randint(0,n,B) calls == one randint(0,n,(U,B))
    call on the legacy global RNG (mdnn.py:221) — lets the product path draw
    the whole id table up front without changing the stream."""
    np.random.seed(5)
    seq = np.stack([np.random.randint(0, 800, 100) for _ in range(100)])
    np.random.seed(5)
    one = np.random.randint(0, 800, (100, 100))
    np.testing.assert_array_equal(seq, one)


def test_density_matches_reference_pdf():
    g = golden('pdf_cases.npz')
    for tag in ('full', 'diag'):
        ls = g['Ls_' + tag]
        lp = oden.mog_logpdf(g['a'], g['ms'], ls, g['x'])
        np.testing.assert_allclose(lp, g['logpdf_' + tag], rtol=1e-12)
        _, covs = oden.mog_moments(g['ms'], ls)
        np.testing.assert_allclose(covs, g['S_' + tag], rtol=1e-12)
    keep, a = oden.prune(np.array([0.6, 0.001, 0.397, 0.002]), 0.005)
    np.testing.assert_allclose(a, g['pruned_a'], rtol=1e-14)
    np.testing.assert_allclose(g['ms'][keep], g['pruned_ms'])


def test_pendulum_reference_fixture():
    """The reference's own regression data (slice) through the oracle."""
    g = golden('pendulum_ref.npz')
    n = g['params'].shape[0]
    sa = torch.from_numpy(g['data']).reshape(n, -1, 4)
    summ = osum.summary_start(sa[:, :, :3], sa[:, :, 3:])
    m = oest.OracleMDNN(input_dim=40, output_dim=2,
                        output_lows=np.array([0.01] * 2),
                        output_highs=np.array([2.0] * 2), n_gaussians=10,
                        full_covariance=False, hidden_layers=(128, 128),
                        activation=torch.nn.Tanh, lr=5e-4, eps_noise=0.0)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items()
                       if k.startswith('w0.')})
    logs = m.run_training(summ, torch.from_numpy(g['params']), 100, 100,
                          ids_table=g['ids'])
    np.testing.assert_allclose(logs['test_loss'], g['test_loss'], rtol=1e-4)
    np.testing.assert_allclose(logs['train_loss'], g['train_loss'], rtol=1e-4)
    tsa = torch.from_numpy(g['true_data']).reshape(1, -1, 4)
    a, ms, ls = m.predict_mog_params(
        osum.summary_start(tsa[:, :, :3], tsa[:, :, 3:]))[0]
    nll = -oden.mog_logpdf(a, ms, ls, g['true_params'].reshape(1, -1))
    np.testing.assert_allclose(nll, g['mog.nll_true'], rtol=1e-3)
