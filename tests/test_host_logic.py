"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol
include/bsig.h declares, the Python mirror keeps the reference's interface
(state_dict keys, torch-RNG init order, numpy-RNG frequency draw, error
behaviour) and the product path refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden

import bayes_sim_ig_amd as B
from bayes_sim_ig_amd import _lib, pdf
from oracle import estimators as oest

NO_GPU = not torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'bsig.h')).read()
    declared = set(re.findall(r'\b(bsig_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.exported_symbols())
    assert _lib.load().bsig_version() >= 100


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, 'bayes_sim_ig_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'oracle' not in src.replace('Oracle of record', ''), fn


def test_summary_dim_matches_reference_shapes():
    g = golden('summaries.npz')
    for case in ('cartpole', 'ant', 'short', 'pendulum'):
        s, a = g[case + '.states'], g[case + '.actions']
        for fn in ('summary_start', 'summary_corr', 'summary_corrdiff'):
            if case + '.' + fn in g:
                assert B.summarizers.summary_dim(fn, s.shape[1], s.shape[2], a.shape[2]) \
                    == g[case + '.' + fn].shape[1]
    for d, depth in zip(g['signature_depth.d'], g['signature_depth.depth']):
        assert B.signature_depth(int(d)) == int(depth)
    assert B.summarizers.summary_dim('summary_signatory', 11, 211, 20) == 232
    assert B.summarizers.summary_dim('summary_signatory', 11, 17, 4) == 22 + 22 ** 2 + 22 ** 3


@pytest.mark.parametrize('full', [False, True])
def test_mdnn_matches_reference_interface(full):
    kw = dict(input_dim=40, output_dim=3, output_lows=np.zeros(3), output_highs=np.ones(3),
              n_gaussians=10, full_covariance=full, hidden_layers=(24, 24),
              activation=torch.nn.Tanh, lr=5e-4)
    torch.manual_seed(3)
    m = B.MDNN(device='cpu', **kw)
    torch.manual_seed(3)
    o = oest.OracleMDNN(**kw)
    assert list(m.state_dict()) == list(o.state_dict())
    for a, b in zip(m.state_dict().values(), o.state_dict().values()):
        assert torch.equal(a, b)                   # same torch-RNG init order
    assert m.L_size == 3 and m.n_gaussians == 10 and m.lr == 5e-4
    assert isinstance(m, torch.nn.Module) and m.activation is torch.nn.Tanh
    # parameters are views of one flat buffer; load_state_dict writes through
    sd = {k: torch.full_like(v, 0.25) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    used = sum(p.numel() for p in m.parameters())
    assert float(m._flat.sum()) == pytest.approx(0.25 * used)
    assert m._flat.numel() % 4 == 0
    assert all(p.grad is not None and p.grad.data_ptr() >= m._flat_grad.data_ptr()
               for p in m.parameters())


def test_mdrff_frequency_draw_consumes_numpy_rng_like_reference():
    np.random.seed(11)
    m = B.MDRFF(input_dim=302, output_dim=13, output_lows=np.zeros(13),
                output_highs=np.ones(13), n_gaussians=4, lr=1e-3, activation=torch.nn.Tanh,
                full_covariance=False, n_feat=64, sigma=4.0)
    np.random.seed(11)
    ref = np.random.normal(0.0, 1.0, (32, 302))
    np.testing.assert_array_equal(m.rff.freqs.numpy(), ref.astype(np.float32))
    assert m.rff.a == pytest.approx(np.sqrt(2.0 / 64))
    assert list(m.state_dict()) == ['pi.weight', 'pi.bias', 'mu.weight', 'mu.bias',
                                    'Diag.0.weight', 'Diag.0.bias']
    assert m.pi.weight.shape == (4, 64)
    with pytest.raises(ValueError):
        B.RFF(64, 302, 4.0, kernel='Bogus', quasi_random=False)


@pytest.mark.parametrize('tag', ['cos_rbf'] + ['%s.%s' % (k, m)
                                               for k in ('Matern12', 'Matern32', 'Matern52',
                                                         'Laplace')
                                               for m in ('cossin', 'cos')])
def test_rff_variant_draws_are_the_references(tag):
    """f4 (host half): the product RFF consumes the numpy RNG like the reference's RFF for the
    cos-only map (freqs, then 2*pi*rand offsets, rff.py:98-102) and for the Matern / Laplace
    Student-t draws (normal, then chisquare, rff.py:151-184): frequencies, offsets and the RNG
    state afterwards equal the reference-generated golden bit for bit."""
    g = golden('rff_variants.npz')
    if tag == 'cos_rbf':
        args = dict(n_feat=64, d=302, sigma=4.0, cos_only=True, kernel='RBF')
    else:
        kern, mode = tag.split('.')
        args = dict(n_feat=48, d=150, sigma=[0.5 + 0.01 * j for j in range(150)],
                    cos_only=(mode == 'cos'), kernel=kern)
    np.random.seed(int(g[tag + '.seed']))
    r = B.RFF(quasi_random=False, device='cpu', **args)
    assert int(np.random.randint(0, 1 << 30)) == int(g[tag + '.rng_after'])
    np.testing.assert_array_equal(r.freqs.numpy(), g[tag + '.freqs'])
    assert float(r.a) == float(g[tag + '.a'])
    if args['cos_only']:
        np.testing.assert_array_equal(r.offset.numpy(), g[tag + '.offset'])
    else:
        assert r.offset is None


@pytest.mark.skipif(not NO_GPU, reason='checks the no-GPU failure mode')
def test_product_path_fails_loudly_without_gpu():
    m = B.MDNN(input_dim=4, output_dim=2, output_lows=np.zeros(2), output_highs=np.ones(2),
               n_gaussians=2, full_covariance=False, hidden_layers=(8,),
               activation=torch.nn.Tanh, lr=1e-3)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.forward(torch.zeros(3, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.run_training(torch.zeros(10, 4), torch.zeros(10, 2), 2, 2)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        B.summary_start(torch.zeros(1, 12, 3), torch.zeros(1, 12, 1))


def test_bayessim_constructs_by_name_like_reference():
    cfg = {'modelClass': 'MDRFF_Matern52_2.5', 'summarizerFxn': 'summary_corrdiff',
           'trainTrajLen': 21, 'components': 4, 'hiddenLayers': (128, 128), 'lr': 1e-3}
    B.MDNN.VERBOSE = False
    np.random.seed(0)
    bs = B.BayesSim(model_cfg=cfg, obs_dim=4, act_dim=1, params_dim=13,
                    params_lows=np.zeros(13), params_highs=np.ones(13), prior=None)
    assert isinstance(bs.model, B.MDRFF) and bs.model.rff.n_feat == 200
    assert bs.model.input_dim == 302 and float(bs.model.rff.sigma[0, 0]) == 2.5
    assert B.BayesSim.NUM_GRAD_UPDATES == 100 and B.BayesSim.MINIBATCH_SIZE == 100
    assert B.BayesSim.get_n_trajs_per_batch(2500, 2000) == 500
    with pytest.raises(NameError):
        B.BayesSim(model_cfg=dict(cfg, summarizerFxn='nope'), obs_dim=4, act_dim=1,
                   params_dim=13, params_lows=np.zeros(13), params_highs=np.ones(13),
                   prior=None)


def test_bayessim_matern_model_by_name_draws_the_references_model():
    """f4: BayesSim(modelClass='MDRFF_Matern32_2.0') under the seeds make_golden.chunk_case used
    builds the reference's model: same torch-RNG head init, same numpy-RNG Student-t frequencies
    (bayes_sim.py:72-81, rff.py:167-170), sigma 2.0."""
    g = golden('chunk_mdrff_matern32.npz')
    cfg = {'modelClass': 'MDRFF_Matern32_2.0', 'summarizerFxn': 'summary_corrdiff',
           'trainTrajLen': 21, 'components': 4, 'hiddenLayers': [], 'lr': 1e-3,
           'fullCovariance': False}
    B.MDNN.VERBOSE = False
    torch.manual_seed(24)
    np.random.seed(24)
    bs = B.BayesSim(model_cfg=cfg, obs_dim=4, act_dim=1, params_dim=4,
                    params_lows=np.zeros(4), params_highs=np.ones(4), prior=None)
    np.testing.assert_array_equal(bs.model.rff.freqs.numpy(), g['rff.freqs'])
    np.testing.assert_array_equal(bs.model.rff.sigma.numpy(), g['rff.sigma'])
    for k, v in bs.model.state_dict().items():
        np.testing.assert_array_equal(v.numpy(), g['w0.' + k])


def test_compat_aliases():
    from bayes_sim_ig_amd import compat
    compat.install()
    from bayes_sim_ig.bayes_sim import BayesSim
    from bayes_sim_ig.models.mdnn import MDNN
    from bayes_sim_ig.models.mdrff import MDRFF
    from bayes_sim_ig.utils.summarizers import summary_corrdiff
    from bayes_sim_ig.utils import pdf as p2
    assert BayesSim is B.BayesSim and MDNN is B.MDNN and MDRFF is B.MDRFF
    assert summary_corrdiff is B.summary_corrdiff and p2 is pdf


def test_pdf_matches_reference():
    g = golden('pdf_cases.npz')
    for tag in ('full', 'diag'):
        mog = pdf.MoG(a=g['a'], ms=list(g['ms']), Ls=list(g['Ls_' + tag]))
        np.testing.assert_allclose(mog.eval(g['x'], log=True), g['logpdf_' + tag], rtol=1e-12)
        np.testing.assert_allclose(mog.eval(g['x'], log=False), g['pdf_' + tag], rtol=1e-12)
        np.testing.assert_allclose(np.stack([c.S for c in mog.xs]), g['S_' + tag], rtol=1e-12)
        np.testing.assert_allclose(np.stack([c.P for c in mog.xs]), g['P_' + tag], rtol=1e-9)
        np.testing.assert_allclose([c.logdetP for c in mog.xs], g['logdetP_' + tag], rtol=1e-12)
        np.random.seed(77)
        np.testing.assert_allclose(mog.gen(n_samples=50), g['gen_' + tag], rtol=1e-12)
    mog = pdf.MoG(a=np.array([0.6, 0.001, 0.397, 0.002]), ms=list(g['ms']),
                  Ls=list(g['Ls_full']))
    mog.prune_negligible_components(threshold=0.005)
    np.testing.assert_allclose(mog.a, g['pruned_a'], rtol=1e-14)
    np.testing.assert_allclose(np.stack([c.m for c in mog.xs]), g['pruned_ms'])
    uni = pdf.Uniform(np.zeros(3), np.ones(3) * 2.0)
    np.testing.assert_allclose(uni.eval(g['uniform_x']), g['uniform_logpdf'])
    np.random.seed(78)
    np.testing.assert_allclose(uni.gen(n_samples=5), g['uniform_gen'])
    # product / quotient round trip (py3 __truediv__)
    prior = pdf.Gaussian(m=np.zeros(3), S=4.0 * np.eye(3))
    back = (mog * prior) / prior
    np.testing.assert_allclose(back.a, mog.a, rtol=1e-9)
    np.testing.assert_allclose(back.xs[0].m, mog.xs[0].m, rtol=1e-8, atol=1e-10)


def test_quasi_random_frequencies_are_not_collinear():
    """RFF with input_dim <= 100 draws its frequencies from a low-discrepancy sequence
    (reference rff.py:113-117 via mdrff.py:23).  A PLAIN Halton sequence would make every
    coordinate with a prime base above m the same ramp (for m = 100, I = 40: 14 collinear,
    one-sided columns); the stand-in for ghalton's generalized sequence must not."""
    import numpy as np
    from bayes_sim_ig_amd import rff
    m, d = 100, 40                          # BayesSim's default nFeat = 200 on pendulum summary_start
    pts = rff.halton_points(m, d)
    assert pts.shape == (m, d) and pts.min() > 0.0 and pts.max() < 1.0
    assert np.array_equal(pts, rff.halton_points(m, d))          # deterministic
    f = rff.draw_freqs('RBF', m, d, quasi_random=True)
    c = np.corrcoef(f.T)
    np.fill_diagonal(c, 0.0)
    assert np.abs(c).max() < 0.5                                 # iid normal columns: ~0.3
    assert np.abs(f.mean(0)).max() < 0.3 and 0.8 < f.std(0).min() and f.std(0).max() < 1.2
    # low discrepancy survives the scrambling: every coordinate fills its 10 deciles evenly
    counts = np.stack([np.histogram(pts[:, j], bins=10, range=(0, 1))[0] for j in range(d)])
    assert counts.min() >= 5 and counts.max() <= 15           # iid uniform: 2..20


def test_narrow_two_layer_trunk_is_stored_zero_padded():
    """A two-layer tanh trunk narrower than 128 lives zero-padded to [128, 128] in the flat
    buffer (so that the persistent update kernel covers it); the module surface -- parameter
    shapes, state_dict keys, load_state_dict -- is the reference's."""
    import numpy as np
    import torch
    from bayes_sim_ig_amd import MDNN
    kw = dict(input_dim=40, output_dim=2, output_lows=np.zeros(2), output_highs=np.ones(2),
              n_gaussians=10, full_covariance=True, activation=torch.nn.Tanh, lr=1e-3)
    torch.manual_seed(3)
    m = MDNN(hidden_layers=(24, 24), **kw)
    torch.manual_seed(3)
    ref = MDNN(hidden_layers=(128, 128), **kw)
    assert m._hidden == [24, 24] and m._hidden_stored == [128, 128]
    assert m._flat.numel() == ref._flat.numel()
    sd = m.state_dict()
    assert list(sd) == list(ref.state_dict())
    assert sd['net.fcon0.weight'].shape == (24, 40) and sd['net.fcon1.weight'].shape == (24, 24)
    assert sd['pi.weight'].shape == (10, 24) and sd['Lower.weight'].shape == (10, 24)
    total_real = sum(v.numel() for v in sd.values())
    assert int((m._flat != 0).sum()) <= total_real
    # the same torch-RNG stream as an unpadded (24, 24) reference-shaped model
    torch.manual_seed(3)
    lin = torch.nn.Linear(40, 24)
    assert torch.equal(sd['net.fcon0.weight'], lin.weight.detach())
    # load_state_dict writes through the views into the flat buffer
    new = {k: torch.full_like(v, 0.5) for k, v in sd.items()}
    m.load_state_dict(new)
    assert float(m._flat.sum()) == 0.5 * total_real
    m.net[0].weight.grad.fill_(1.0)
    assert float(m._flat_grad.sum()) == 24 * 40
    # three layers, relu, wide trunks: stored as they are
    assert MDNN(hidden_layers=(24, 24, 24), **kw)._hidden_stored == [24, 24, 24]
    assert MDNN(hidden_layers=(24, 24), **dict(kw, activation=torch.nn.ReLU))._hidden_stored == [24, 24]
    assert MDNN(hidden_layers=(256, 64), **kw)._hidden_stored == [256, 64]


def test_linear_head_tiling_planner():
    """The tiling of the persistent kernel of the linear heads (csrc/fit_persistent.hip, host
    arithmetic: bsig_debug_persist_geometry): the ShadowHand head gets 32 x 192 tiles on 198 CUs and
    two rows per owner on 50 of the CUs to spare; every covered shape respects the chip (<= 256
    workgroups, <= 160 KB of LDS each), owns every minibatch row and every held-out row."""
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    names = ['NT', 'KS', 'n_blocks', 'k_slices', 'G', 'T', 'n_owner', 'R', 'NE', 'RE', 'eval_passes', 'lds', 'mixed']

    def geom(batch, feat, d, k, max_test):
        out = (C.c_int32 * 16)()
        ok = lib.bsig_debug_persist_geometry(batch, feat, d, k, max_test, out)
        return ok, dict(zip(names, list(out)[:13]))

    ok, g = geom(100, 4096, 32, 4, 200)          # cfg5: 260 x 4096 heads
    assert ok and (g['NT'], g['KS'], g['n_blocks'], g['k_slices']) == (2, 192, 9, 22)
    assert (g['G'], g['T'], g['n_owner'], g['R'], g['mixed']) == (198, 248, 50, 2, 0)
    ok, g = geom(100, 1024, 13, 10, 200)         # cfg2: 270 x 1024
    assert ok and g['NT'] == 1 and g['KS'] == 96 and g['mixed'] == 0
    rng = np.random.RandomState(0)
    n_ok = 0
    for _ in range(300):
        batch, k = int(rng.randint(1, 113)), int(rng.randint(1, 17))
        d = int(rng.randint(1, min(40, 8 * (64 // k)) + 1))
        feat = int(rng.choice([96, 200, 300, 500, 512, 1024, 2048, 4096, 8192]))
        max_test = int(rng.randint(0, 400))
        ok, g = geom(batch, feat, d, k, max_test)
        if not ok:
            continue
        n_ok += 1
        nh = k * (1 + 2 * d)
        assert g['KS'] in (96, 192, 288) and g['NT'] in (1, 2)
        assert g['n_blocks'] * 16 * g['NT'] >= nh and g['k_slices'] * g['KS'] >= feat
        assert g['G'] == g['n_blocks'] * g['k_slices'] <= 256 and g['G'] <= g['T'] <= 256
        assert g['n_owner'] * g['R'] >= batch and 1 <= g['R'] <= 8 and g['n_owner'] <= g['T']
        assert g['lds'] <= 160 * 1024
        if g['eval_passes'] > 0:
            assert g['NE'] * g['RE'] >= max_test and g['RE'] <= 8 and g['NE'] <= g['T']
            assert g['eval_passes'] * batch >= max_test
        # owners on workgroups of their own unless the chip has none to spare
        assert g['mixed'] == max(0, g['n_owner'] - (g['T'] - g['G']))
    assert n_ok > 200
    assert geom(200, 4096, 32, 4, 0)[0] == 0        # minibatches beyond 112 rows: the v1 kernel / per-phase kernels


def test_mdnn_kernel_workgroup_planner():
    """The layout of the persistent kernel of the two-layer MDNN (csrc/fit_persistent_mdnn.hip, host
    arithmetic: bsig_debug_persist_mdnn_geometry): an owner workgroup takes ONE minibatch row where the
    chip has the CUs for a workgroup per row (narrow first layers), else two, else four; first layers with
    more tiles than CUs are streamed (8 rows per owner); every covered shape respects the chip."""
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    names = ['k_slices', 'G1', 'n_owner', 'mr', 'n_small', 'wide', 'stream', 'eval_passes', 'lds', 'Nh']

    def geom(batch, inp, d, k, full=0, max_test=200):
        out = (C.c_int32 * 16)()
        ok = lib.bsig_debug_persist_mdnn_geometry(batch, inp, d, k, full, max_test, out)
        return ok, dict(zip(names, list(out)[:10]))

    ok, g = geom(100, 232, 32, 4)                 # cfg4: one k-slice, a workgroup per row
    assert ok and (g['k_slices'], g['G1'], g['mr'], g['n_owner'], g['wide'], g['stream']) == (1, 4, 1, 100, 0, 0)
    ok, g = geom(100, 11802, 10, 5)               # cfg3-like: 47 k-slices, two rows per owner
    assert ok and (g['k_slices'], g['G1'], g['mr'], g['n_owner'], g['stream']) == (47, 188, 2, 50, 0)
    ok, g = geom(100, 11802, 17, 10)              # cfg/ant.yaml: wide heads
    assert ok and g['wide'] == 1 and g['mr'] == 2 and g['G1'] + g['n_owner'] + g['n_small'] <= 256
    ok, g = geom(100, 56402, 13, 10)              # cfg/anymal.yaml: streamed first layer
    assert ok and g['stream'] == 1 and g['mr'] == 8 and g['n_owner'] == 13
    ok, g = geom(100, 232, 4, 4, full=1)          # full covariance keeps four rows per owner
    assert ok and g['mr'] == 4 and g['n_owner'] == 25
    rng = np.random.RandomState(1)
    n_ok = 0
    for _ in range(300):
        batch, k = int(rng.randint(1, 105)), int(rng.randint(1, 17))
        d = int(rng.randint(1, min(40, 8 * (64 // k)) + 1))
        inp = int(rng.choice([3, 40, 190, 256, 257, 1290, 2310, 11154, 11802, 12300, 13000, 56402]))
        ok, g = geom(batch, inp, d, k, 0, int(rng.randint(0, 400)))
        if not ok:
            continue
        n_ok += 1
        assert g['Nh'] == k * (1 + 2 * d) and g['k_slices'] * 256 >= inp
        assert g['G1'] + g['n_owner'] + g['n_small'] <= 256 and g['lds'] <= 160 * 1024
        assert g['n_owner'] * g['mr'] >= batch and g['mr'] in (1, 2, 4, 8)
        if not g['stream']:
            assert g['G1'] == 4 * g['k_slices']
            # the fewest rows per owner the chip has CUs for
            for mr in (1, 2):
                if mr < g['mr']:
                    assert 4 * g['k_slices'] + -(-batch // mr) + g['n_small'] > 256
        else:
            assert g['mr'] == 8 and inp % 2 == 0
    assert n_ok > 150
