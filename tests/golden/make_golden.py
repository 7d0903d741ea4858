#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself.

Runs only in the build container (needs /root/reference).  The reference is
imported from where it lies; nothing of its source is copied.  Two absent,
unpinned third-party modules are stubbed in memory so the import succeeds
(SURVEY.md §8c): ``ghalton`` (only reached for RFF input_dim <= 100 and pdf
Halton sampling — never exercised below) and ``signatory`` (its use is not
exercised; signature parity is pinned by mathematical KATs instead).

Oracle of record: torch 2.10.0 CPU (the reference pins torch==1.8.0,
setup.py:18), numpy 2.2, fp32, default thread count.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True

_gh = types.ModuleType('ghalton')
_gh.EA_PERMS = list(range(4096))


class _NoHalton:
    def __init__(self, perms):
        raise RuntimeError('ghalton is stubbed; quasi-random path not pinned')


_gh.GeneralizedHalton = _NoHalton
sys.modules['ghalton'] = _gh
sys.modules['signatory'] = types.ModuleType('signatory')
sys.path.insert(0, '/root/reference')

import io                                                    # noqa: E402
import contextlib                                            # noqa: E402
import numpy as np                                           # noqa: E402
import torch                                                 # noqa: E402
from bayes_sim_ig.bayes_sim import BayesSim                  # noqa: E402
from bayes_sim_ig.models.mdnn import MDNN                    # noqa: E402
from bayes_sim_ig.models.mdrff import MDRFF                  # noqa: E402
from bayes_sim_ig.utils import summarizers as ref_sum        # noqa: E402
from bayes_sim_ig.utils import pdf as ref_pdf                # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print('wrote', name, os.path.getsize(path), 'bytes')


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy()
            for k, v in model.state_dict().items()}


# ------------------------------------------------------------ summarizers
def gen_summaries():
    g = torch.Generator().manual_seed(101)
    cases = {
        # name: (N, T, sd, ad)
        'cartpole': (7, 21, 4, 1),      # T > 10
        'ant': (3, 12, 60, 8),          # sd > 50 -> 5 waypoints
        'short': (4, 6, 5, 2),          # T < 10: all steps used by corr
        'single_pad': (1, 6, 3, 1),     # N == 1, T < 10: summary_start pads
        'pendulum': (5, 10, 3, 1),      # T == 10
    }
    out = {}
    for name, (n, t, sd, ad) in cases.items():
        s = torch.randn(n, t, sd, generator=g)
        a = torch.rand(n, t, ad, generator=g)
        out[name + '.states'] = s.numpy()
        out[name + '.actions'] = a.numpy()
        for fn in ('summary_start', 'summary_waypts', 'summary_corr',
                   'summary_corrdiff'):
            if name == 'short' and fn in ('summary_start', 'summary_waypts'):
                continue   # reference padding raises for N > 1
            res = quiet(getattr(ref_sum, fn), s, a)
            out[name + '.' + fn] = res.numpy()
    out['signature_depth.d'] = np.array([2, 5, 6, 22, 23, 69, 110, 111, 232,
                                         12100, 12101])
    out['signature_depth.depth'] = np.array(
        [ref_sum.signature_depth(int(d)) for d in out['signature_depth.d']])
    save('summaries.npz', **out)


# ---------------------------------------------------------- MDN one step
def one_step_case(tag, model_ctor, input_dim, out_dim, batch, eps_noise, seed):
    """forward tuple, loss, grads and weights after ONE Adam step."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    MDNN.EPS_NOISE = eps_noise
    model = quiet(model_ctor)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, input_dim, generator=g)
    y = torch.rand(batch, out_dim, generator=g)
    out = {'x': x.numpy(), 'y': y.numpy(), 'eps_noise': np.float64(eps_noise)}
    for k, v in sd_np(model).items():
        out['w0.' + k] = v
    if hasattr(model, 'rff'):
        out['rff.freqs'] = model.rff.freqs.numpy()
        out['rff.sigma'] = model.rff.sigma.numpy()
        out['rff.a'] = np.float64(model.rff.a)
        out['rff.features'] = model.rff.to_features(x).numpy()
    # the jitter noise the reference will draw: rand_like(L_d) right after
    # manual_seed == torch.rand(shape) after the same seed
    noise_seed = seed + 2
    torch.manual_seed(noise_seed)
    noise = torch.rand(batch, out_dim, model.n_gaussians)
    out['noise'] = noise.numpy()
    opt = torch.optim.Adam(model.parameters(), lr=model.lr)
    opt.zero_grad()
    torch.manual_seed(noise_seed)
    w, mu, l_d, low = model(x)
    loss = model.mdn_loss_fn(w, mu, l_d, low, y)
    loss.backward()
    out['weights'], out['mu'], out['L_d'] = (w.detach().numpy(),
                                             mu.detach().numpy(),
                                             l_d.detach().numpy())
    if low is not None:
        out['L'] = low.detach().numpy()
    out['loss'] = np.float64(loss.item())
    for k, p in model.named_parameters():
        out['grad.' + k] = p.grad.detach().numpy().copy()
    opt.step()
    for k, v in sd_np(model).items():
        out['w1.' + k] = v
    out['lr'] = np.float64(model.lr)
    save('mdn_step_%s.npz' % tag, **out)
    MDNN.EPS_NOISE = 1.0e-5


def gen_one_step():
    lows2, highs2 = np.zeros(2), np.ones(2)
    for eps, sfx in ((0.0, 'eps0'), (1.0e-5, 'eps1e5')):
        one_step_case(
            'diag_' + sfx,
            lambda: MDNN(input_dim=40, output_dim=2, output_lows=lows2,
                         output_highs=highs2, n_gaussians=10,
                         full_covariance=False, hidden_layers=(24, 24),
                         activation=torch.nn.Tanh, lr=5e-4),
            40, 2, 16, eps, 11)
        one_step_case(
            'full_' + sfx,
            lambda: MDNN(input_dim=12, output_dim=5,
                         output_lows=np.zeros(5), output_highs=np.ones(5),
                         n_gaussians=3, full_covariance=True,
                         hidden_layers=(16,), activation=torch.nn.Tanh,
                         lr=1e-3),
            12, 5, 9, eps, 12)
        one_step_case(
            'mdrff_' + sfx,
            lambda: MDRFF(input_dim=302, output_dim=13,
                          output_lows=np.zeros(13), output_highs=np.ones(13),
                          n_gaussians=4, lr=1e-3, activation=torch.nn.Tanh,
                          full_covariance=False, n_feat=64, kernel='RBF',
                          sigma=4.0),
            302, 13, 10, eps, 13)
    # clamp-active case: huge logits push softmax weights under MIN_WEIGHT
    def clamp_model():
        m = MDNN(input_dim=6, output_dim=3, output_lows=np.zeros(3),
                 output_highs=np.ones(3), n_gaussians=5,
                 full_covariance=False, hidden_layers=(8,),
                 activation=torch.nn.Tanh, lr=1e-3)
        with torch.no_grad():
            m.pi.weight.mul_(40.0)
            m.pi.bias.copy_(torch.tensor([9.0, -9.0, 4.0, -14.0, 0.0]))
        return m
    one_step_case('clamp_eps1e5', clamp_model, 6, 3, 12, 1.0e-5, 14)


# ------------------------------------------- teacher-forced 100-update chunk
def synth_pairs(n, t, sd, ad, d, seed):
    """Small theta-dependent synthetic (theta, states, actions) pairs."""
    g = torch.Generator().manual_seed(seed)
    theta = torch.rand(n, d, generator=g)
    w1 = torch.randn(d, sd, generator=g) / d ** 0.5
    act = torch.rand(n, t, ad, generator=g)
    bm = torch.randn(ad, sd, generator=g) / ad ** 0.5
    s = 0.1 * torch.randn(n, sd, generator=g)
    states = []
    for i in range(t):
        states.append(s)
        s = 0.9 * s + 0.1 * torch.tanh(theta @ w1) + 0.1 * (act[:, i] @ bm) \
            + 0.01 * torch.randn(n, sd, generator=g)
    return theta, torch.stack(states, dim=1), act


def chunk_case(tag, model_class, summarizer, n, t, sd, ad, d, k, hidden,
               full_cov, seed, n_updates=100, batch=100):
    MDNN.EPS_NOISE = 0.0
    theta, states, actions = synth_pairs(n, t, sd, ad, d, seed)
    lows, highs = np.zeros(d), np.ones(d)
    cfg = {'modelClass': model_class, 'summarizerFxn': summarizer,
           'trainTrajLen': t, 'components': k, 'hiddenLayers': hidden,
           'lr': 1e-3, 'fullCovariance': full_cov}
    torch.manual_seed(seed)
    np.random.seed(seed)
    bsim = quiet(BayesSim, model_cfg=cfg, obs_dim=sd, act_dim=ad,
                 params_dim=d, params_lows=lows, params_highs=highs,
                 prior=None, proposal=None, device='cpu')
    out = {'theta': theta.numpy(), 'states': states.numpy(),
           'actions': actions.numpy(), 'lr': np.float64(1e-3),
           'n_updates': np.int64(n_updates), 'batch': np.int64(batch)}
    for kk, v in sd_np(bsim.model).items():
        out['w0.' + kk] = v
    if hasattr(bsim.model, 'rff'):
        out['rff.freqs'] = bsim.model.rff.freqs.numpy()
        out['rff.sigma'] = bsim.model.rff.sigma.numpy()
    # record the id table the reference will draw (same seed, same calls)
    n_train = max(int(n * 0.8), 1)
    np.random.seed(seed + 7)
    ids = np.stack([np.random.randint(0, n_train, batch)
                    for _ in range(n_updates)])
    out['ids'] = ids
    np.random.seed(seed + 7)
    summ = quiet(bsim.summarizer_fxn, states, actions)
    out['summaries'] = summ.numpy()
    logs = quiet(bsim.model.run_training, x_data=summ, y_data=theta,
                 n_updates=n_updates, batch_size=batch, test_frac=0.2)
    out['train_loss'] = np.array(logs['train_loss'])
    out['test_loss'] = np.array(logs['test_loss'])
    for kk, v in sd_np(bsim.model).items():
        out['w1.' + kk] = v
    # posterior at the first held-out trajectory (single point -> no refit)
    xs = summ[n_train:n_train + 1]
    mog = bsim.model.predict_MoGs(xs)[0]
    out['mog.a'] = mog.a
    out['mog.ms'] = np.stack([g.m for g in mog.xs])
    out['mog.Ss'] = np.stack([g.S for g in mog.xs])
    th_true = theta[n_train:n_train + 1].numpy().astype(np.float64)
    out['mog.nll_true'] = -mog.eval(th_true, log=True)
    save('chunk_%s.npz' % tag, **out)
    MDNN.EPS_NOISE = 1.0e-5


def gen_chunks():
    chunk_case('mdnn_start', 'MDNN', 'summary_start', 250, 12, 3, 1, 2, 10,
               (24, 24), False, 21)
    chunk_case('mdnn_corrdiff_full', 'MDNN', 'summary_corrdiff', 200, 12, 4,
               2, 3, 3, (16, 16), True, 22)


def gen_chunk_mdrff():
    """MDRFF chunk with input_dim > 100 so the numpy frequency path is taken
    (BayesSim hard-codes n_feat=200, bayes_sim.py:81)."""
    chunk_case('mdrff_corrdiff', 'MDRFF', 'summary_corrdiff', 200, 21, 4, 1,
               4, 4, [], False, 23)


# ------------------------------------- f4: cos-only features, Matern draws
def gen_rff_variants():
    """rff.py:98-102,122-126 (cos-only feature map + its offsets) and
    rff.py:151-184 (Matern12/32/52 Student-t frequency draws) from the
    reference's own RFF class, input_dim > 100 so that no ghalton is reached.
    Call order pinned: draw_freqs (normal, then chisquare) BEFORE the offsets'
    np.random.rand."""
    from bayes_sim_ig.models.rff import RFF
    g = torch.Generator().manual_seed(301)
    x = torch.randn(12, 302, generator=g) * 0.5
    out = {'x': x.numpy()}
    np.random.seed(31)
    r = RFF(64, 302, 4.0, cos_only=True, quasi_random=False, kernel='RBF')
    out['cos_rbf.seed'] = np.int64(31)
    out['cos_rbf.freqs'] = r.freqs.numpy()
    out['cos_rbf.offset'] = r.offset.numpy()
    out['cos_rbf.a'] = np.float64(r.a)
    out['cos_rbf.features'] = r.to_features(x).numpy()
    out['cos_rbf.rng_after'] = np.int64(np.random.randint(0, 1 << 30))
    for i, kern in enumerate(('Matern12', 'Matern32', 'Matern52', 'Laplace')):
        seed = 40 + i
        for cos_only in (False, True):
            tag = '%s.%s' % (kern, 'cos' if cos_only else 'cossin')
            np.random.seed(seed)
            r = RFF(48, 150, [0.5 + 0.01 * j for j in range(150)],
                    cos_only=cos_only, quasi_random=False, kernel=kern)
            out[tag + '.seed'] = np.int64(seed)
            out[tag + '.freqs'] = r.freqs.numpy()
            out[tag + '.a'] = np.float64(r.a)
            if cos_only:
                out[tag + '.offset'] = r.offset.numpy()
            xs = x[:, :150] * 0.05      # heavy-tailed freqs: keep |inner| moderate
            out[tag + '.features'] = r.to_features(xs).numpy()
            out[tag + '.rng_after'] = np.int64(np.random.randint(0, 1 << 30))
    out['x150'] = (x[:, :150] * 0.05).numpy()
    save('rff_variants.npz', **out)


def gen_chunk_mdrff_matern():
    """bayes_sim.py:72-81: 'MDRFF_Matern32_2.0' -> kernel Matern32, sigma 2.0,
    n_feat 200; teacher-forced chunk like chunk_mdrff_corrdiff."""
    chunk_case('mdrff_matern32', 'MDRFF_Matern32_2.0', 'summary_corrdiff', 200,
               21, 4, 1, 4, 4, [], False, 24)


# ------------------------------------------------------------------- pdf
def gen_pdf():
    rs = np.random.RandomState(5)
    d, k = 3, 4
    a = rs.dirichlet(np.ones(k))
    ms = rs.randn(k, d)
    rows, cols = np.tril_indices(d, -1)
    ls_full = np.concatenate([np.exp(0.3 * rs.randn(k, d)),
                              0.4 * rs.randn(k, rows.size)], axis=1)
    ls_diag = ls_full[:, :d]
    x = rs.randn(6, d)
    out = {'a': a, 'ms': ms, 'Ls_full': ls_full, 'Ls_diag': ls_diag, 'x': x}
    for tag, ls in (('full', ls_full), ('diag', ls_diag)):
        mog = ref_pdf.MoG(a=a, ms=list(ms), Ls=list(ls))
        out['logpdf_' + tag] = mog.eval(x, log=True)
        out['pdf_' + tag] = mog.eval(x, log=False)
        out['S_' + tag] = np.stack([g.S for g in mog.xs])
        out['P_' + tag] = np.stack([g.P for g in mog.xs])
        out['logdetP_' + tag] = np.array([g.logdetP for g in mog.xs])
        np.random.seed(77)
        out['gen_' + tag] = mog.gen(n_samples=50)
    mog = ref_pdf.MoG(a=np.array([0.6, 0.001, 0.397, 0.002]), ms=list(ms),
                      Ls=list(ls_full))
    mog.prune_negligible_components(threshold=0.005)
    out['pruned_a'] = mog.a
    out['pruned_ms'] = np.stack([g.m for g in mog.xs])
    uni = ref_pdf.Uniform(np.zeros(d), np.ones(d) * 2.0)
    out['uniform_logpdf'] = uni.eval(np.abs(x[:, :d]) % 2.0 * 0.99 + 0.005)
    out['uniform_x'] = np.abs(x[:, :d]) % 2.0 * 0.99 + 0.005
    np.random.seed(78)
    out['uniform_gen'] = uni.gen(n_samples=5)
    save('pdf_cases.npz', **out)


# ------------------------------------------- reference's own test fixture
def gen_pendulum():
    """A slice of the data file the reference's regression test holds
    (bayes_sim_ig/tests/data/pendulum_train_data_ones_policy_rnd.npz) plus
    the reference's outputs on it: BayesSim(MDNN, summary_start) as in
    regression_tests.py:46-89 shortened to one run_training call."""
    src = '/root/reference/bayes_sim_ig/tests/data/'
    loaded = np.load(src + 'pendulum_train_data_ones_policy_rnd.npz')
    n = 1000
    params = loaded['params'][:n].astype(np.float32)
    data = loaded['data'][:n].astype(np.float32)
    true = np.load(src + 'pendulum_true_data_ones_policy_rnd.npz')
    out = {'params': params, 'data': data,
           'true_params': true['params'].astype(np.float32),
           'true_data': true['data'].astype(np.float32)}
    MDNN.EPS_NOISE = 0.0
    th = torch.from_numpy(params)
    sa = torch.from_numpy(data).reshape(n, -1, 4)
    states, actions = sa[:, :, :3], sa[:, :, 3:]
    cfg = {'modelClass': 'MDNN', 'summarizerFxn': 'summary_start',
           'trainTrajLen': 10, 'components': 10, 'hiddenLayers': (128, 128),
           'lr': 5e-4}
    torch.manual_seed(2)
    np.random.seed(2)
    lows, highs = np.array([0.01] * 2), np.array([2.0] * 2)
    bsim = quiet(BayesSim, model_cfg=cfg, obs_dim=3, act_dim=1, params_dim=2,
                 params_lows=lows, params_highs=highs, prior=None,
                 proposal=None, device='cpu')
    for kk, v in sd_np(bsim.model).items():
        out['w0.' + kk] = v
    np.random.seed(9)
    out['ids'] = np.stack([np.random.randint(0, 800, 100) for _ in range(100)])
    np.random.seed(9)
    logs = quiet(bsim.run_training, th, states, actions)
    out['train_loss'] = np.array(logs['train_loss'])
    out['test_loss'] = np.array(logs['test_loss'])
    for kk, v in sd_np(bsim.model).items():
        out['w1.' + kk] = v
    tsa = torch.from_numpy(out['true_data']).reshape(1, -1, 4)
    mog = quiet(bsim.predict, tsa[:, :, :3], tsa[:, :, 3:])
    out['mog.a'] = mog.a
    out['mog.ms'] = np.stack([g.m for g in mog.xs])
    out['mog.Ss'] = np.stack([g.S for g in mog.xs])
    out['mog.nll_true'] = -mog.eval(
        out['true_params'].reshape(1, -1).astype(np.float64), log=True)
    save('pendulum_ref.npz', **out)
    MDNN.EPS_NOISE = 1.0e-5


if __name__ == '__main__':
    torch.set_num_threads(8)
    # no argument: every fixture; otherwise only the named generators
    # (round 6 added gen_rff_variants / gen_chunk_mdrff_matern without
    # rewriting the older files: `make_golden.py gen_rff_variants gen_chunk_mdrff_matern`)
    todo = sys.argv[1:] or ['gen_summaries', 'gen_one_step', 'gen_chunks',
                            'gen_chunk_mdrff', 'gen_rff_variants',
                            'gen_chunk_mdrff_matern', 'gen_pdf', 'gen_pendulum']
    for name in todo:
        globals()[name]()
