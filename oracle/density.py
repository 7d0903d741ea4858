"""ORACLE (test infrastructure only) — mixture-of-Gaussians density, numpy fp64.

Restates the parts of bayes_sim_ig/utils/pdf.py that the posterior report
needs: ``Gaussian(m, L=...)`` (pdf.py:241-251), ``Gaussian.eval`` joint
log-pdf (pdf.py:329-334), ``MoG.eval`` (pdf.py:452-469),
``MoG.prune_negligible_components`` (pdf.py:562-570).
Pinned against tests/golden/pdf_cases.npz (outputs of the reference).
"""
import numpy as np
from scipy.special import logsumexp


def unpack_tril(l_flat, ndim):
    """1-D [diag | strict-lower in np.tril_indices(ndim,-1) order] -> T
    (pdf.py:243-247)."""
    l_flat = np.asarray(l_flat, dtype=np.float64)
    t = np.diag(l_flat[:ndim])
    if 1 < ndim < l_flat.shape[0]:
        r, c = np.tril_indices(ndim, -1)
        t[r, c] = l_flat[ndim:]
    return t


def gaussian_logpdf(x, m, l_flat):
    """log N(x | m, T T^T); x [n, D]."""
    m = np.asarray(m, dtype=np.float64)
    t = unpack_tril(l_flat, m.size)
    cov = t @ t.T
    prec = np.linalg.inv(cov)
    logdet_p = -2.0 * np.sum(np.log(np.diagonal(t)))
    xm = np.atleast_2d(np.asarray(x, dtype=np.float64)) - m
    lp = -np.sum((xm @ prec) * xm, axis=1)
    lp += logdet_p - m.size * np.log(2.0 * np.pi)
    return 0.5 * lp


def mog_logpdf(a, ms, ls, x):
    """log sum_k a_k N(x | m_k, T_k T_k^T)."""
    ps = np.array([gaussian_logpdf(x, m, l) for m, l in zip(ms, ls)]).T
    return logsumexp(ps + np.log(np.asarray(a, dtype=np.float64)), axis=1)


def mog_moments(ms, ls):
    """Component means and covariance matrices."""
    covs = []
    for m, l in zip(ms, ls):
        t = unpack_tril(l, np.asarray(m).size)
        covs.append(t @ t.T)
    return np.asarray(ms, dtype=np.float64), np.asarray(covs)


def prune(a, threshold):
    """Indices kept and re-spread weights, pdf.py:562-570."""
    a = np.asarray(a, dtype=np.float64)
    drop = np.nonzero(a < threshold)[0]
    keep = np.array([i for i in range(a.size) if i not in drop], dtype=int)
    new_a = np.delete(a, drop)
    new_a = new_a + np.sum(a[drop]) / keep.size
    return keep, new_a
