"""GPU parity tests: every HIP entry point (through the C ABI, via the ctypes
mirror) against the oracle / the reference-generated golden vectors.
Tolerances are written next to each comparison; integer/index and pure-copy
work is bit-exact."""
import ctypes as C

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


# ----------------------------------------------------------------- summarizers
CASES = ['cartpole', 'ant', 'short', 'single_pad', 'pendulum']


@pytest.mark.parametrize('case', CASES)
def test_summary_start_matches_reference(B, case):
    g = golden('summaries.npz')
    for fn in ('summary_start', 'summary_waypts'):
        key = case + '.' + fn
        s = torch.from_numpy(g[case + '.states']).to(DEV)
        a = torch.from_numpy(g[case + '.actions']).to(DEV)
        out = getattr(B.summarizers, fn)(s, a)
        if key in g:      # pure copy: bit-exact against the reference's output
            np.testing.assert_array_equal(out.cpu().numpy(), g[key])
        else:             # N>1 padding (reference raises): own-last-row repeat
            from oracle import summarize as osum
            np.testing.assert_array_equal(
                out.cpu().numpy(), getattr(osum, fn)(s.cpu(), a.cpu()).numpy())


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('fn', ['summary_corr', 'summary_corrdiff'])
def test_crosscorr_matches_reference(B, case, fn):
    g = golden('summaries.npz')
    s = torch.from_numpy(g[case + '.states']).to(DEV)
    a = torch.from_numpy(g[case + '.actions']).to(DEV)
    out = getattr(B.summarizers, fn)(s, a).cpu().numpy()
    ref = g[case + '.' + fn]
    assert out.shape == ref.shape
    # outer product: one fp32 subtract + one fp32 multiply per entry -> exact
    np.testing.assert_array_equal(out[:, :-2], ref[:, :-2])
    # mean / unbiased std: different summation order, 1e-6 relative
    np.testing.assert_allclose(out[:, -2:], ref[:, -2:], rtol=1e-6, atol=1e-7)


def test_summarizers_cpu_inputs_roundtrip(B):
    """CPU tensors (the zero probe of bayes_sim.py:57-60) hop to the GPU and
    come back on the CPU."""
    out = B.summary_start(torch.zeros(1, 12, 3), torch.zeros(1, 12, 1))
    assert out.device.type == 'cpu' and out.shape == (1, 40)
    assert float(out.abs().sum()) == 0.0
    with pytest.raises(AssertionError):
        B.summary_start(torch.zeros(4, 3), torch.zeros(4, 1))
    with pytest.raises(AssertionError):
        B.summary_corr(torch.zeros(2, 1, 3, device=DEV), torch.zeros(2, 1, 1, device=DEV))


def test_summarizers_empty_batch(B):
    s = torch.zeros(0, 12, 4, device=DEV)
    a = torch.zeros(0, 12, 2, device=DEV)
    assert B.summary_start(s, a).shape == (0, 60)
    assert B.summary_corrdiff(s, a).shape == (0, 10 * 3 * 10 * 2 + 2)
    assert B.summary_signatory(s, a).shape == (0, 7 + 49 + 343)


@pytest.mark.parametrize('n,t,sd,ad,depth', [
    (5, 6, 4, 1, 3), (3, 11, 17, 4, 3), (4, 9, 30, 2, 2), (2, 11, 211, 20, 1),
    (3, 5, 2, 1, 3), (2, 21, 60, 8, 2), (3, 4, 25, 6, 3)])
def test_signature_matches_oracle(B, n, t, sd, ad, depth):
    from oracle import summarize as osum
    gen = torch.Generator().manual_seed(n * 100 + sd)
    s = torch.randn(n, t, sd, generator=gen)
    a = torch.rand(n, t, ad, generator=gen)
    out = B.summary_signatory(s.to(DEV), a.to(DEV), depth=depth).cpu().numpy()
    ref = osum.summary_signatory(s, a, depth=depth, dtype=torch.float64).numpy()
    assert out.shape == ref.shape
    # fp32 Horner recursion vs fp64 oracle: 2e-5 of the row scale
    scale = np.abs(ref).max(axis=1, keepdims=True)
    np.testing.assert_allclose(out / scale, ref / scale, atol=2e-5)
    if depth == osum.signature_depth(1 + sd + ad):      # default-depth call
        out2 = B.summary_signatory(s.to(DEV), a.to(DEV)).cpu().numpy()
        np.testing.assert_array_equal(out, out2)


def test_signature_known_answers(B):
    """KATs from the definition, through the kernel: the kernel prepends the
    time channel 1..L, so feed paths whose first channel is that ramp."""
    s = torch.tensor([[[0.5], [-1.0], [0.25]]], device=DEV)       # x channel
    a = torch.zeros(1, 3, 1, device=DEV)                          # dummy channel
    out = B.summary_signatory(s, a, depth=2).cpu().numpy()[0]
    # path [(1,.5,0),(2,-1,0),(3,.25,0)]: KAT of SURVEY §8c on channels (t,x)
    d = 3
    l1, l2 = out[:d], out[d:].reshape(d, d)
    np.testing.assert_allclose(l1, [2, -0.25, 0], atol=1e-6)
    np.testing.assert_allclose(l2[:2, :2], [[2, 1.125], [-1.625, 0.03125]], atol=1e-6)
    assert np.abs(l2[2]).max() == 0 and np.abs(l2[:, 2]).max() == 0


# ------------------------------------------------------------------------ GEMM
def _gemm(B, a, b, m, n, k, a_km, b_km, epi=0, act=0, bias=None, aux=None,
          alpha=1.0, a_rows=None, b_rows=None, ldc=None):
    lib = B._lib.load()
    cols = 2 * n if epi == B._lib.EPI_COS_SIN else n
    ldc = ldc or cols
    c = torch.full((m, ldc), float('nan'), device=DEV)
    ws = torch.empty(int(lib.bsig_gemm_workspace_bytes(m, n, k)) // 4 + 1, device=DEV)
    B._lib.check(lib.bsig_gemm_f32(
        B._lib.ptr(a), a.stride(0), a_km, B._lib.ptr(a_rows), B._lib.ptr(b), b.stride(0),
        b_km, B._lib.ptr(b_rows), B._lib.ptr(c), ldc, m, n, k, epi, act,
        B._lib.ptr(bias), B._lib.ptr(aux), aux.stride(0) if aux is not None else 0,
        alpha, B._lib.ptr(ws), ws.numel() * 4, B._lib.stream()))
    return c[:, :cols]


@pytest.mark.parametrize('m,n,k', [(100, 128, 40), (100, 2048, 2310), (100, 260, 4096),
                                   (7, 5, 3), (33, 65, 130), (1, 175, 128),
                                   (512, 384, 200), (200, 128, 11802)])
@pytest.mark.parametrize('a_km,b_km', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(B, m, n, k, a_km, b_km):
    if k > 5000 and (a_km or b_km):
        pytest.skip('k-major long-K case not on the path')
    gen = torch.Generator().manual_seed(m + n + k)
    a_t = torch.randn(m, k, generator=gen)
    b_t = torch.randn(n, k, generator=gen)
    ref = (a_t.double() @ b_t.double().T).numpy()
    a = (a_t.T if a_km else a_t).contiguous().to(DEV)
    b = (b_t.T if b_km else b_t).contiguous().to(DEV)
    out = _gemm(B, a, b, m, n, k, a_km, b_km).cpu().numpy()
    # fp32 fma chain vs fp64: ~1e-7 * sum|a.b| -> 3e-6 of sqrt(k) scale
    np.testing.assert_allclose(out, ref, atol=3e-6 * np.sqrt(k) * 4, rtol=1e-5)


@pytest.mark.parametrize('tile', [3, 4, 5])           # 128x64, 128x96, 96x128
@pytest.mark.parametrize('a_km,b_km', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('splits', [1, 3])
def test_gemm_every_tile_shape(B, tile, a_km, b_km, splits):
    """The tile shapes the planner only picks for large problems, forced on a small ragged
    problem (edges in M, N and K) in every operand layout, unsplit and split-K."""
    import os
    m, n, k = 300, 260, 530
    gen = torch.Generator().manual_seed(tile * 7 + a_km * 2 + b_km)
    a_t = torch.randn(m, k, generator=gen)
    b_t = torch.randn(n, k, generator=gen)
    ref = (a_t.double() @ b_t.double().T).numpy()
    a = (a_t.T if a_km else a_t).contiguous().to(DEV)
    b = (b_t.T if b_km else b_t).contiguous().to(DEV)
    os.environ['BSIG_GEMM_TILE'], os.environ['BSIG_GEMM_SPLITS'] = str(tile), str(splits)
    try:
        out = _gemm(B, a, b, m, n, k, a_km, b_km).cpu().numpy()
    finally:
        os.environ.pop('BSIG_GEMM_TILE'), os.environ.pop('BSIG_GEMM_SPLITS')
    np.testing.assert_allclose(out, ref, atol=3e-6 * np.sqrt(k) * 4, rtol=1e-5)


@pytest.mark.parametrize('shape', ['fwd', 'dw'])
def test_gemm_large_minibatch_shapes(B, shape):
    """The scaled-batch update's products as the planner runs them (least-padding tile,
    split K): head forward 8192 x 260 x 4096 on gathered feature rows with bias, and
    dW = dO^T F[ids] with the gather on the contraction index, vs fp64 on sampled outputs."""
    L = B._lib
    gen = torch.Generator(device=DEV).manual_seed(11)
    pool, bsz, nh, f = 20000, 8192, 260, 4096
    feats = torch.randn(pool, f, device=DEV, generator=gen) * 0.02
    ids = torch.randint(0, pool, (bsz,), device=DEV, generator=gen).to(torch.int32)
    if shape == 'fwd':
        w = torch.randn(nh, f, device=DEV, generator=gen)
        bias = torch.randn(nh, device=DEV, generator=gen)
        out = _gemm(B, feats, w, bsz, nh, f, 0, 0, epi=L.EPI_BIAS, bias=bias, a_rows=ids)
        rows = torch.arange(0, bsz, 97, device=DEV)
        ref = feats[ids.long()[rows]].double() @ w.double().T + bias.double()
        torch.testing.assert_close(out[rows].double(), ref, rtol=1e-5, atol=2e-5)
    else:
        d_o = torch.randn(bsz, nh, device=DEV, generator=gen) * 0.01
        out = _gemm(B, d_o, feats, nh, f, bsz, 1, 1, b_rows=ids)
        cols = torch.arange(0, f, 61, device=DEV)
        ref = d_o.double().T @ feats[ids.long()][:, cols].double()
        torch.testing.assert_close(out[:, cols].double(), ref, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('a_km,b_km', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_lean_gemm_is_bitwise_the_old_kernel(B, a_km, b_km):
    """gemm_lean_kernel (csrc/gemm_lean.h: the K loop without vector-ALU work, two LDS images, one
    barrier per step) keeps every output's fma chain: bit for bit gemm_mfma_kernel, on ragged
    problems (edges in M, N and K -- the guarded last step), every tile shape, unsplit and split K,
    gathered rows / gathered contraction rows.  (tools/gemm_lean_ab.py runs 240 such cases.)"""
    import os
    gen = torch.Generator().manual_seed(17 + 2 * a_km + b_km)
    pool = 700
    for (m, n, k) in ((300, 260, 530), (100, 512, 2310), (129, 97, 64)):
        def pitched(rows, cols):
            buf = torch.zeros(rows, (cols + 3) // 4 * 4)
            buf[:, :cols] = torch.randn(rows, cols, generator=gen)
            return buf.to(DEV)[:, :cols]
        a = pitched(k, m) if a_km else pitched(pool, k)
        b = pitched(pool, n) if b_km else pitched(n, k)
        a_rows = None if a_km else torch.randint(0, pool, (m,), generator=gen).to(torch.int32).to(DEV)
        b_rows = torch.randint(0, pool, (k,), generator=gen).to(torch.int32).to(DEV) if b_km else None
        for tile in (0, 1, 2, 3, 4, 5):
            for splits in (1, 3):
                os.environ['BSIG_GEMM_TILE'], os.environ['BSIG_GEMM_SPLITS'] = str(tile), str(splits)
                try:
                    out = {}
                    for lean in ('0', '1'):
                        os.environ['BSIG_GEMM_LEAN'] = lean
                        out[lean] = _gemm(B, a, b, m, n, k, a_km, b_km, a_rows=a_rows, b_rows=b_rows).clone()
                finally:
                    for key in ('BSIG_GEMM_TILE', 'BSIG_GEMM_SPLITS', 'BSIG_GEMM_LEAN'):
                        os.environ.pop(key, None)
                assert torch.equal(out['0'], out['1']), (m, n, k, tile, splits)


def test_wide_gradient_gemm_at_the_padded_pitch(B):
    """dW = dO^T F[ids] with dO at the pitch the fit engine keeps it at for large minibatches
    (ceil16(Nh) = 272 floats): the whole-width gradient kernel (csrc/gemm_wide.h) against fp64 on
    sampled columns and against the generic kernels everywhere (another summation order)."""
    import os
    gen = torch.Generator(device=DEV).manual_seed(12)
    pool, bsz, nh, f = 20000, 8192, 260, 4096
    feats = torch.randn(pool, f, device=DEV, generator=gen) * 0.02
    ids = torch.randint(0, pool, (bsz,), device=DEV, generator=gen).to(torch.int32)
    d_o = torch.full((bsz, 272), float('nan'), device=DEV)       # (the padding columns may hold anything)
    d_o[:, :nh] = torch.randn(bsz, nh, device=DEV, generator=gen) * 0.01
    out = _gemm(B, d_o[:, :nh], feats, nh, f, bsz, 1, 1, b_rows=ids).clone()
    os.environ['BSIG_GEMM_WIDE'] = '0'
    try:
        ref32 = _gemm(B, d_o[:, :nh], feats, nh, f, bsz, 1, 1, b_rows=ids).clone()
    finally:
        os.environ.pop('BSIG_GEMM_WIDE')
    cols = torch.arange(0, f, 61, device=DEV)
    ref = d_o[:, :nh].double().T @ feats[ids.long()][:, cols].double()
    torch.testing.assert_close(out[:, cols].double(), ref, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(out, ref32, rtol=1e-5, atol=2e-6)
    assert not torch.equal(out, ref32)      # (it IS the other kernel: 16x16x4 tiles, 4 K slices)


def test_gemm_gather_and_epilogues(B):
    L = B._lib
    gen = torch.Generator().manual_seed(5)
    m, n, k, pool = 100, 96, 302, 800
    x = torch.randn(pool, k + 2, generator=gen).to(DEV)[:, :k]      # ld = 304 (vec)
    xs = torch.randn(pool, k + 1, generator=gen).to(DEV)[:, :k]     # ld = 303 (scalar)
    w = (torch.randn(n, k, generator=gen) * 0.1).to(DEV)
    bias = torch.randn(n, generator=gen).to(DEV)
    ids = torch.randint(0, pool, (m,), generator=gen).to(torch.int32).to(DEV)
    for src in (x, xs):
        ref = src[ids.long()].double() @ w.double().T
        out = _gemm(B, src, w, m, n, k, 0, 0, a_rows=ids)
        np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=2e-5)
        out = _gemm(B, src, w, m, n, k, 0, 0, epi=L.EPI_BIAS_ACT, act=L.ACT_TANH,
                    bias=bias, a_rows=ids)
        np.testing.assert_allclose(out.cpu().numpy(),
                                   torch.tanh(ref + bias.double()).cpu().numpy(), atol=5e-6)
        out = _gemm(B, src, w, m, n, k, 0, 0, epi=L.EPI_COS_SIN, alpha=0.25, a_rows=ids)
        exp = 0.25 * torch.cat([torch.cos(ref), torch.sin(ref)], 1)
        np.testing.assert_allclose(out.cpu().numpy(), exp.cpu().numpy(), atol=2e-6)
        out = _gemm(B, src, w, m, n, k, 0, 0, epi=L.EPI_COS_OFF, alpha=0.5, bias=bias,
                    a_rows=ids)
        np.testing.assert_allclose(out.cpu().numpy(),
                                   (0.5 * torch.cos(ref + bias.double())).cpu().numpy(),
                                   atol=2e-6)
    # dW = dY^T X[ids] : both operands k-major, gather on the contraction index
    dy = torch.randn(m, n, generator=gen).to(DEV)
    ref = dy.double().T @ x[ids.long()].double()
    out = _gemm(B, dy, x, n, k, m, 1, 1, b_rows=ids)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=3e-5)
    # dX = (dY W) * tanh'(h)
    h = torch.tanh(torch.randn(m, k, generator=gen)).to(DEV)
    ref = (dy.double() @ w.double()) * (1 - h.double() ** 2)
    out = _gemm(B, dy, w, m, k, n, 0, 1, epi=L.EPI_MUL_DACT, act=L.ACT_TANH, aux=h)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=2e-5)


def test_gemm_is_bitwise_reproducible(B):
    gen = torch.Generator().manual_seed(9)
    a = torch.randn(100, 2310, generator=gen).to(DEV)
    b = torch.randn(512, 2310, generator=gen).to(DEV)
    o1 = _gemm(B, a, b, 100, 512, 2310, 0, 0).clone()
    o2 = _gemm(B, a, b, 100, 512, 2310, 0, 0)
    assert torch.equal(o1, o2)


def test_gemm_identity_asymmetric(B):
    """A = I against an asymmetric B catches transposed fragment layouts."""
    n = 96
    eye = torch.eye(n, device=DEV)
    b = (torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) % 17) - 3.0
    out = _gemm(B, eye, b, n, n, n, 0, 0)          # I @ b^T
    assert torch.equal(out, b.T.contiguous())
    out = _gemm(B, eye, b, n, n, n, 1, 1)          # I^T @ b (k-major both)
    assert torch.equal(out, b)


def test_rff_features_match_reference(B):
    g = golden('mdn_step_mdrff_eps0.npz')
    np.random.seed(0)
    rff = B.RFF(64, 302, 4.0, cos_only=False, quasi_random=False, device=DEV,
                freqs=g['rff.freqs'])
    assert float(rff.a) == pytest.approx(float(g['rff.a']))
    out = rff.to_features(torch.from_numpy(g['x']).to(DEV)).cpu().numpy()
    # |inner| ~ 15: fp32 GEMM error 1e-6 passes straight through cos/sin
    np.testing.assert_allclose(out, g['rff.features'], atol=3e-6)
    # frequency draw consumes the numpy RNG exactly like the reference
    np.random.seed(13)
    r2 = B.RFF(64, 302, 4.0, quasi_random=False, device='cpu')
    np.random.seed(13)
    np.testing.assert_array_equal(r2.freqs.numpy(),
                                  np.random.normal(0, 1, (32, 302)).astype(np.float32))
    with pytest.raises(ValueError):
        B.RFF(64, 302, 4.0, kernel='Nope', quasi_random=False)


RFF_VARIANTS = ['cos_rbf'] + ['%s.%s' % (k, m) for k in ('Matern12', 'Matern32', 'Matern52',
                                                           'Laplace')
                              for m in ('cossin', 'cos')]


@pytest.mark.parametrize('tag', RFF_VARIANTS)
def test_rff_variants_match_reference(B, tag):
    """f4: bsig_rff_project(cos_only=1) (rff.py:122-126, offsets :98-102) and the cos/sin map on
    Matern / Laplace frequencies (rff.py:151-184) against features the reference's own RFF objects
    produced (tests/golden/rff_variants.npz); the draw itself is bit-equal."""
    g = golden('rff_variants.npz')
    if tag == 'cos_rbf':
        args, xk = dict(n_feat=64, d=302, sigma=4.0, cos_only=True, kernel='RBF'), 'x'
    else:
        kern, mode = tag.split('.')
        args, xk = dict(n_feat=48, d=150, sigma=[0.5 + 0.01 * j for j in range(150)],
                        cos_only=(mode == 'cos'), kernel=kern), 'x150'
    np.random.seed(int(g[tag + '.seed']))
    rff = B.RFF(quasi_random=False, device=DEV, **args)
    assert int(np.random.randint(0, 1 << 30)) == int(g[tag + '.rng_after'])
    np.testing.assert_array_equal(rff.freqs.cpu().numpy(), g[tag + '.freqs'])
    if args['cos_only']:
        np.testing.assert_array_equal(rff.offset.cpu().numpy(), g[tag + '.offset'])
    out = rff.to_features(torch.from_numpy(g[xk]).to(DEV)).cpu().numpy()
    assert out.shape == g[tag + '.features'].shape
    # the fp32 error of the inner product (~1e-7 |inner| per term order) passes straight through
    # cos / sin: 3e-6 where |inner| <= 8 (every RBF / Matern32 / Matern52 entry), scaled with
    # |inner| for the Cauchy-tailed Matern12 / Laplace frequencies (|inner| up to 1.8e3)
    sig = rff.sigma.cpu().numpy().astype(np.float64)
    inner = g[xk].astype(np.float64) @ (g[tag + '.freqs'].astype(np.float64) / sig).T
    if not args['cos_only']:
        inner = np.concatenate([inner, inner], axis=1)
    tol = 3e-6 * np.maximum(1.0, np.abs(inner) / 8.0)
    err = np.abs(out - g[tag + '.features'])
    assert (err <= tol).all(), (tag, float((err / tol).max()))
    # and against the fp64 value of the same map, same bound
    if args['cos_only']:
        f64 = float(g[tag + '.a']) * np.cos(inner + g[tag + '.offset'].astype(np.float64))
    else:
        half = inner[:, :inner.shape[1] // 2]
        f64 = float(g[tag + '.a']) * np.concatenate([np.cos(half), np.sin(half)], axis=1)
    assert (np.abs(out - f64) <= tol).all()


def test_rff_cos_only_large_shape_vs_fp64(B):
    """cos-only epilogue on a GEMM-sized problem (several tiles, ragged edges)."""
    np.random.seed(5)
    rff = B.RFF(1000, 302, 4.0, cos_only=True, quasi_random=False, device=DEV)
    x = torch.randn(777, 302, generator=torch.Generator().manual_seed(6))
    out = rff.to_features(x.to(DEV)).cpu().numpy()
    inner = x.numpy().astype(np.float64) @ (rff.freqs.cpu().numpy().astype(np.float64) / 4.0).T
    f64 = float(rff.a) * np.cos(inner + rff.offset.cpu().numpy().astype(np.float64))
    np.testing.assert_allclose(out, f64, rtol=0, atol=3e-6)


# ------------------------------------------------------------------- MDN head
def _head_dims(B, d, k, full, eps):
    hd = B._lib.HeadDims()
    hd.out_dim, hd.n_comp, hd.full_cov = d, k, 1 if full else 0
    hd.eps_noise, hd.min_weight, hd.ll_limit = eps, 1e-5, 1e5
    return hd


@pytest.mark.parametrize('b,d,k,full,eps', [
    (16, 2, 10, False, 0.0), (16, 2, 10, False, 1e-5), (9, 5, 3, True, 1e-5),
    (100, 32, 4, False, 1e-5), (100, 17, 5, False, 0.0), (37, 13, 10, True, 1e-5),
    (1, 1, 1, False, 1e-5), (300, 3, 7, True, 0.0), (64, 32, 4, True, 1e-5)])
def test_head_nll_and_grad_match_closed_form(B, b, d, k, full, eps):
    from oracle import estimators as oest
    lib = B._lib.load()
    hd = _head_dims(B, d, k, full, eps)
    nh = int(lib.bsig_head_width(C.byref(hd)))
    gen = torch.Generator().manual_seed(b * 7 + d)
    o = torch.randn(b, nh, generator=gen) * 0.5
    o[:, :k] *= 6.0                       # some weights hit the MIN_WEIGHT clamp
    y = torch.rand(b, d, generator=gen)
    noise = torch.rand(b, d, k, generator=gen)
    loss_ref, d_ref, _ = oest.mdn_head_closed_form(o.numpy(), y.numpy(), d, k, full,
                                                   eps_noise=eps, noise=noise.numpy())
    od, yd, nd = o.to(DEV), y.to(DEV), noise.to(DEV)
    loss = torch.zeros(1, device=DEV)
    d_o = torch.full((b, nh), float('nan'), device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(int(lib.bsig_head_workspace_bytes(C.byref(hd), b)) // 4 + 64, device=DEV)
    B._lib.check(lib.bsig_mdn_head_nll(
        C.byref(hd), B._lib.ptr(od), nh, B._lib.ptr(yd), d, None, b, b, B._lib.ptr(nd), 0, 0,
        B._lib.ptr(loss), B._lib.ptr(d_o), B._lib.ptr(flag), B._lib.ptr(ws), ws.numel() * 4,
        B._lib.stream()))
    assert int(flag.item()) == 0
    # fp32 kernel vs fp64 closed form
    assert float(loss.item()) == pytest.approx(loss_ref, rel=2e-6, abs=2e-6)
    gscale = np.abs(d_ref).max()
    np.testing.assert_allclose(d_o.cpu().numpy(), d_ref, rtol=2e-4, atol=2e-6 * gscale)


@pytest.mark.parametrize('tag', ['diag_eps0', 'diag_eps1e5', 'full_eps1e5', 'clamp_eps1e5',
                                 'mdrff_eps1e5'])
def test_forward_tuple_and_loss_match_reference(B, tag):
    g = golden('mdn_step_%s.npz' % tag)
    m = _build(B, tag, g)
    x = torch.from_numpy(g['x']).to(DEV)
    noise = torch.from_numpy(g['noise']).to(DEV)
    w, mu, l_d, low = m.forward(x, noise=noise)
    # GEMM summation order differs from MKL: 2e-6 abs on O(1) outputs
    np.testing.assert_allclose(w.cpu().numpy(), g['weights'], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(mu.cpu().numpy(), g['mu'], atol=3e-6)
    np.testing.assert_allclose(l_d.cpu().numpy(), g['L_d'], rtol=5e-6)
    if 'L' in g:
        np.testing.assert_allclose(low.cpu().numpy(), g['L'], atol=3e-6)
    loss = m.mdn_loss_fn(w, mu, l_d, low, torch.from_numpy(g['y']).to(DEV))
    assert float(loss.item()) == pytest.approx(float(g['loss']), rel=1e-5)
    # mdn_loss_fn on the reference's own tuple
    loss = m.mdn_loss_fn(*[None if k not in g else torch.from_numpy(g[k])
                           for k in ('weights', 'mu', 'L_d', 'L')],
                         torch.from_numpy(g['y']))
    assert float(loss.item()) == pytest.approx(float(g['loss']), rel=2e-6)


STEP_CFG = {
    'diag': dict(cls='MDNN', input_dim=40, output_dim=2, n_gaussians=10,
                 full_covariance=False, hidden_layers=(24, 24), lr=5e-4),
    'full': dict(cls='MDNN', input_dim=12, output_dim=5, n_gaussians=3,
                 full_covariance=True, hidden_layers=(16,), lr=1e-3),
    'mdrff': dict(cls='MDRFF', input_dim=302, output_dim=13, n_gaussians=4,
                  full_covariance=False, lr=1e-3, n_feat=64, sigma=4.0),
    'clamp': dict(cls='MDNN', input_dim=6, output_dim=3, n_gaussians=5,
                  full_covariance=False, hidden_layers=(8,), lr=1e-3),
}


def _build(B, tag, g):
    kw = dict(STEP_CFG[tag.split('_')[0]])
    cls = kw.pop('cls')
    d = kw['output_dim']
    B.MDNN.EPS_NOISE = float(g['eps_noise'])
    kw.update(output_lows=np.zeros(d), output_highs=np.ones(d),
              activation=torch.nn.Tanh, device=DEV)
    if cls == 'MDRFF':
        m = B.MDRFF(freqs=g['rff.freqs'], **kw)
    else:
        m = B.MDNN(**kw)
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items()
                       if k.startswith('w0.')})
    return m


@pytest.fixture(autouse=True)
def _restore_eps():
    yield
    import bayes_sim_ig_amd as pkg
    pkg.MDNN.EPS_NOISE = 1e-5


@pytest.mark.parametrize('tag', ['diag_eps0', 'diag_eps1e5', 'full_eps0', 'full_eps1e5',
                                 'mdrff_eps0', 'mdrff_eps1e5', 'clamp_eps1e5'])
def test_one_step_grads_and_adam_match_reference(B, tag):
    g = golden('mdn_step_%s.npz' % tag)
    m = _build(B, tag, g)
    x, y = torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['y']).to(DEV)
    loss = m.loss_and_grad(x, y, noise=torch.from_numpy(g['noise']).to(DEV))
    assert float(loss.item()) == pytest.approx(float(g['loss']), rel=1e-5)
    for k, p in m.named_parameters():
        ref = g['grad.' + k]
        scale = max(np.abs(ref).max(), 1e-8)
        # fp32 GEMM chains in a different order than MKL + autograd
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3, atol=2e-5 * scale,
                                   err_msg=k)
    # Adam in isolation: feed the reference's own gradients, compare weights
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.grad.copy_(torch.from_numpy(g['grad.' + k]))
    m.adam_step(1)
    lr = float(g['lr'])
    for k, v in m.state_dict().items():
        # update = lr * g / (|g| + 1e-8): agree to 1e-4 of the step size
        np.testing.assert_allclose(v.cpu().numpy(), g['w1.' + k], rtol=0, atol=1e-4 * lr,
                                   err_msg=k)


@pytest.mark.parametrize('tag', ['diag_eps1e5', 'full_eps1e5', 'mdrff_eps0'])
def test_reference_autograd_pattern(B, tag):
    """loss = mdn_loss_fn(*model(x), y); loss.backward(); Adam.step() — the
    reference's own training idiom (mdnn.py:230-234) — works on the HIP model
    and reproduces the reference's gradients and updated weights."""
    g = golden('mdn_step_%s.npz' % tag)
    m = _build(B, tag, g)
    x, y = torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['y']).to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=float(g['lr']))
    opt.zero_grad()
    loss = m.mdn_loss_fn(*m.forward(x, noise=torch.from_numpy(g['noise']).to(DEV)), y)
    assert loss.requires_grad
    loss.backward()
    assert float(loss.item()) == pytest.approx(float(g['loss']), rel=1e-5)
    for k, p in m.named_parameters():
        ref = g['grad.' + k]
        scale = max(np.abs(ref).max(), 1e-8)
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-3, atol=2e-5 * scale,
                                   err_msg=k)
    opt.step()
    assert not torch.equal(m._flat, torch.zeros_like(m._flat))
    with torch.no_grad():                      # inference path: plain tensors
        out = m.mdn_loss_fn(*m.forward(x, noise=torch.from_numpy(g['noise']).to(DEV)), y)
    assert not out.requires_grad


# ------------------------------------------------------------------- row copies
@pytest.mark.parametrize('n,cols,ld_src,ld_dst,gather', [
    (37, 2310, 2312, 2312, False),     # the staging copy of a ShadowHand chunk: 16-byte rows, float4 path
    (37, 2310, 2312, 2312, True),      # ... gathered through a row table
    (5, 2310, 2311, 2312, False),      # unaligned source rows: scalar path
    (64, 259, 260, 264, True),         # a tail of three columns after the float4 part
    (300, 40, 40, 40, False),          # narrow rows (scalar kernel, 64 threads)
    (20000, 302, 304, 304, False),     # more rows than workgroups (grid stride)
])
def test_copy_rows_is_exact(B, n, cols, ld_src, ld_dst, gather):
    """bsig_copy_rows (the staging / gather copy of the fit, mdnn.py:208-218): bit-exact, columns
    past `cols` of the destination untouched, for the float4 and the scalar kernel."""
    lib, L = B._lib.load(), B._lib
    g = torch.Generator(device='cpu').manual_seed(n + cols)
    n_src = n + 11
    src = torch.randn(n_src, ld_src, generator=g).to(DEV)
    dst = torch.full((n, ld_dst), -7.0, device=DEV)
    rows = torch.randint(0, n_src, (n,), generator=g, dtype=torch.int32).to(DEV) if gather else None
    L.check(lib.bsig_copy_rows(L.ptr(src), ld_src, None if rows is None else L.ptr(rows), L.ptr(dst),
                               ld_dst, n, cols, L.stream()))
    torch.cuda.synchronize()
    want = src[rows.long()] if gather else src[:n]
    assert torch.equal(dst[:, :cols], want[:, :cols])
    assert bool((dst[:, cols:] == -7.0).all())


@pytest.mark.parametrize('k,scale', [(4, 1.0), (128, 1.0), (272, 1.0), (128, 1e-3), (128, 300.0)])
def test_fp32_mfma_is_a_chain_of_fmas_in_ascending_k(B, k, scale):
    """The MDNN kernel's owners of one or two rows form their head outputs as fmaf chains on the vector ALU
    and claim the bits of the 16x16x4 MFMA path (persist_mdnn_device.h: heads_fma_chain;
    tools/micro/mfma_order_probe.hip).  That rests on the hardware adding the four products of a
    v_mfma_f32_16x16x4_f32 one fused multiply-add at a time in ascending k: all 256 outputs of a chain of
    k / 4 such instructions equal the fmaf chain bit for bit, for several magnitudes and lengths."""
    import ctypes as C
    lib = B._lib.load()
    g = torch.Generator(device='cpu').manual_seed(k + int(scale * 7))
    a = ((torch.rand(16, k, generator=g) * 2 - 1) * scale).to(DEV)
    b = ((torch.rand(k, 16, generator=g) * 2 - 1) * 3.0).to(DEV)
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    B._lib.check(lib.bsig_debug_mfma_vs_fma(a.data_ptr(), b.data_ptr(), k, bad.data_ptr(), B._lib.stream()))
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


def test_forward_of_more_row_tiles_than_combine_tickets(B):
    """A head whose width is already a multiple of 16 (K = 16, D = 8 diagonal: Nh = 272) keeps the padded
    output pitch at ANY row count, so a forward() of more than 64 * 1024 rows used to pass the in-launch
    K-slice combine 1094 row tiles for 1024 ticket words (round-5 advisor finding): tickets and the
    exp-sum partials behind them aliased.  The combine now needs ceil(m / 64) <= its ticket capacity;
    the rows of a 70 000-row forward() must equal the same rows pushed through in two halves."""
    torch.manual_seed(3)
    np.random.seed(3)
    d, k = 8, 16
    m = B.MDRFF(input_dim=120, output_dim=d, output_lows=np.zeros(d), output_highs=np.ones(d),
                n_gaussians=k, lr=1e-3, activation=torch.nn.Tanh, full_covariance=False,
                n_feat=64, sigma=4.0, device=DEV)
    assert m.pi.weight.shape[0] + m.mu.weight.shape[0] + getattr(m, 'Diag')[0].weight.shape[0] == 272
    n = 70000
    x = torch.randn(n, 120, generator=torch.Generator().manual_seed(4)).to(DEV)
    old = B.MDNN.EPS_NOISE
    B.MDNN.EPS_NOISE = 1e-5          # the jitter scale comes from the exp-sum partials the tickets aliased
    try:
        torch.manual_seed(9)
        whole = [t.clone() for t in m.forward(x)[:3]]
        halves = []
        for lo, hi in ((0, 35000), (35000, n)):
            torch.manual_seed(9)
            halves.append([t.clone() for t in m.forward(x[lo:hi])[:3]])
    finally:
        B.MDNN.EPS_NOISE = old
    for i, name in enumerate(('weights', 'mu')):
        got = whole[i].cpu().numpy()
        exp = np.concatenate([h[i].cpu().numpy() for h in halves])
        np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-7, err_msg=name)
    # L_d carries the jitter u * 1e-5 * mean(exp(pre)) of ITS batch (different draws per call): compare
    # its batch mean, which a wrong jitter scale (aliased partials) moves by orders of magnitude more
    got = whole[2].cpu().numpy().mean()
    exp = np.concatenate([h[2].cpu().numpy() for h in halves]).mean()
    assert abs(got - exp) <= 1e-4 * abs(exp)
    assert np.isfinite(whole[2].cpu().numpy()).all()
