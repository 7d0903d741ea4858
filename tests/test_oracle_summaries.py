"""Pin the oracle's summarizers against outputs of the reference itself
(tests/golden/summaries.npz) and the signature restatement against KATs."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import summarize as osum
from oracle import signature as osig

CASES = ['cartpole', 'ant', 'short', 'single_pad', 'pendulum']
FNS = ['summary_start', 'summary_waypts', 'summary_corr', 'summary_corrdiff']


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('fn', FNS)
def test_summarizer_matches_reference(case, fn):
    g = golden('summaries.npz')
    key = case + '.' + fn
    if key not in g:
        pytest.skip('reference raises for this case (N>1 padding bug)')
    s = torch.from_numpy(g[case + '.states'])
    a = torch.from_numpy(g[case + '.actions'])
    out = getattr(osum, fn)(s, a).numpy()
    assert out.shape == g[key].shape
    # bit-exact: same ops in the same order on the same bytes
    np.testing.assert_array_equal(out, g[key])


def test_summary_dim_formula():
    g = golden('summaries.npz')
    for case in CASES:
        s = g[case + '.states']
        a = g[case + '.actions']
        for fn in FNS:
            key = case + '.' + fn
            if key in g:
                assert osum.summary_dim(fn, s.shape[1], s.shape[2], a.shape[2]) \
                    == g[key].shape[1]


def test_signature_depth_rule():
    g = golden('summaries.npz')
    for d, depth in zip(g['signature_depth.d'], g['signature_depth.depth']):
        assert osum.signature_depth(int(d)) == int(depth)


# --- signature known-answer tests from the mathematical definition --------
def _sig(path, depth):
    p = torch.tensor(path, dtype=torch.float64).unsqueeze(0)
    return osig.signature(p, depth)[0].numpy()


def test_signature_kat_corner():
    out = _sig([[0, 0], [1, 0], [1, 1]], 3)
    exp = [1, 1, 0.5, 1, 0, 0.5, 1 / 6, 0.5, 0, 0.5, 0, 0, 0, 1 / 6]
    np.testing.assert_allclose(out, exp, atol=1e-15)


def test_signature_kat_straight_line():
    # levels of a straight line are delta^{(x)k}/k!
    out = _sig([[0, 0], [2, 3]], 3)
    exp = [2, 3, 2, 3, 3, 4.5, 4 / 3, 2, 2, 3, 2, 3, 3, 4.5]
    np.testing.assert_allclose(out, exp, atol=1e-14)
    # invariant to subdividing the line
    out2 = _sig([[0, 0], [0.5, 0.75], [1.0, 1.5], [2, 3]], 3)
    np.testing.assert_allclose(out2, exp, atol=1e-14)


def test_signature_kat_time_augmented():
    out = _sig([[1, 0.5], [2, -1], [3, 0.25]], 2)
    exp = [2, -0.25, 2, 1.125, -1.625, 0.03125]
    np.testing.assert_allclose(out, exp, atol=1e-15)


@pytest.mark.parametrize('d,length', [(2, 5), (4, 7), (6, 4)])
def test_signature_chen_vs_bruteforce(d, length):
    rs = np.random.RandomState(d * 10 + length)
    path = rs.randn(length, d)
    chen = _sig(path, 3)
    brute = osig.signature_brute(path, 3)
    np.testing.assert_allclose(chen, brute, rtol=1e-12, atol=1e-13)


def test_signature_shuffle_identity():
    # S1_i * S1_j = S2_ij + S2_ji
    rs = np.random.RandomState(3)
    d = 5
    out = _sig(rs.randn(9, d), 2)
    s1, s2 = out[:d], out[d:].reshape(d, d)
    np.testing.assert_allclose(np.outer(s1, s1), s2 + s2.T, atol=1e-12)


def test_summary_signatory_layout():
    g = torch.Generator().manual_seed(0)
    s = torch.randn(3, 6, 4, generator=g)
    a = torch.rand(3, 6, 1, generator=g)
    out = osum.summary_signatory(s, a)        # d = 6 -> depth 3 -> 258
    assert out.shape == (3, 6 + 36 + 216)
    # level 1 = last - first of [t | s | a]; time channel increments to L-1
    np.testing.assert_allclose(out[:, 0].numpy(), 5.0)
    np.testing.assert_allclose(out[:, 1:5].numpy(),
                               (s[:, -1] - s[:, 0]).numpy(), atol=1e-6)
    # depth-1 case (ShadowHand-sized channel count)
    s = torch.randn(2, 5, 211, generator=g)
    a = torch.rand(2, 5, 20, generator=g)
    out = osum.summary_signatory(s, a)
    assert out.shape == (2, 232)
