"""Uniform / Gaussian / mixture-of-Gaussians densities (numpy, host side).

Counterpart of the reference's bayes_sim_ig/utils/pdf.py for the posterior
objects ``predict_MoGs`` / ``BayesSim.predict`` return (mdnn.py:288,
bayes_sim.py:116-179).  This is post-training, CPU, tiny work in the
reference too ("Speed is not a major concern", pdf.py:10-12) and is NOT part
of the accelerated hot path; it exists so that callers get the same object
API: ``Gaussian(m=, P=|U=|S=|L=)`` / ``Gaussian(Pm=, ...)``, ``MoG(a, ms=,
Ls=|Ss=|Ps=|Us=)`` / ``MoG(a, xs=)``, ``.eval``, ``.gen``, ``*``, ``/``,
``prune_negligible_components``, ``kl``.

Deviations from the reference, on purpose:
  * ``__truediv__`` is defined (the reference only defines the py2
    ``__div__``, pdf.py:363,525, so ``mog / proposal`` raises on py3);
  * ``MoG.calc_mean_and_cov`` returns the moment-matched mean/covariance
    (the reference reads a non-existent ``x.sigma``, pdf.py:549-554);
  * ``MoG.__mul__`` / ``__truediv__`` use the correct product/quotient
    normaliser (the reference's transcription flips one sign, pdf.py:511,533);
  * Halton sampling uses the scrambled Halton sequence of rff.halton_points
    (``ghalton`` is not available).
"""
import numpy as np
import scipy.stats
from scipy.special import erfinv, logsumexp


def discrete_sample(p, n_samples=1):
    """Indices drawn from the discrete distribution ``p`` (pdf.py:60-76):
    one uniform per sample compared against the cumulative weights."""
    edges = np.cumsum(np.asarray(p)[:-1])[np.newaxis, :]
    draws = np.random.rand(n_samples, 1)
    return np.sum((draws > edges).astype(int), axis=1)


def _halton(n, dim):
    from .rff import halton_points
    return halton_points(n, dim)


class Uniform:
    """Axis-aligned uniform density (pdf.py:79-196)."""

    def __init__(self, lb_array=None, ub_array=None):
        assert len(lb_array) == len(ub_array)
        self.lb_array, self.ub_array = lb_array, ub_array
        self.param_dim = len(lb_array)

    def __str__(self):
        return ('Uniform: \nlower bounds:\n' + str(self.lb_array) +
                '\nupper bounds:\n' + str(self.ub_array))

    def gen(self, n_samples=1, method='random'):
        lo, hi = np.asarray(self.lb_array), np.asarray(self.ub_array)
        if method == 'halton':
            return lo + _halton(n_samples, self.param_dim) * (hi - lo)
        if method != 'random':
            raise ValueError('Unknown gen method ' + method)
        # The reference draws dimension by dimension and then reshapes the
        # concatenation row-major (pdf.py:149-158); reproduce that stream.
        cols = [np.random.uniform(lo[i], hi[i], size=n_samples)
                for i in range(self.param_dim)]
        return np.concatenate(cols, axis=0).reshape(-1, self.param_dim)

    def eval(self, x, ii=None, log=True, debug=False):
        if ii is None:
            ii = np.arange(self.param_dim)
        lo, hi = np.asarray(self.lb_array)[ii], np.asarray(self.ub_array)[ii]
        x = np.atleast_2d(x)
        inside = (x > lo) & (x < hi)
        dens = np.full((x.shape[0],), 1.0 / np.prod(hi - lo))
        dens[~np.all(inside, axis=1)] = 0.0
        if not log:
            return dens
        if not inside.any():
            raise ValueError('log prob. not defined outside of truncation')
        with np.errstate(divide='ignore'):
            return np.log(dens)


def _unpack_lower(l_flat, ndim):
    """[diag | strict-lower in np.tril_indices(ndim,-1) order] -> T (pdf.py:241-247)."""
    l_flat = np.asarray(l_flat)
    t = np.diag(l_flat[:ndim])
    if 1 < ndim < l_flat.shape[0]:
        rows, cols = np.tril_indices(ndim, -1)
        t[rows, cols] = l_flat[ndim:]
    return t


class Gaussian:
    """Gaussian with cached natural and moment parameters: m, S, P = S^-1,
    Pm = P m, C (S = C^T C), logdetP (pdf.py:199-412)."""

    def __init__(self, m=None, P=None, U=None, S=None, Pm=None, L=None):
        if m is None and Pm is None:
            raise ValueError('Mean information missing.')
        from_cov = False
        if P is not None:
            self.P = np.asarray(P)
            chol = np.linalg.cholesky(self.P)
            self.C = np.linalg.inv(chol)
            self.logdetP = 2.0 * np.sum(np.log(np.diagonal(chol)))
        elif U is not None:
            U = np.asarray(U)
            self.P = U.T @ U
            self.C = np.linalg.inv(U.T)
            self.logdetP = 2.0 * np.sum(np.log(np.diagonal(U)))
        elif L is not None and m is not None:
            self.C = _unpack_lower(L, np.asarray(m).size).T
            self.P = np.linalg.inv(self.C.T @ self.C)
            self.logdetP = -2.0 * np.sum(np.log(np.diagonal(self.C)))
        elif S is not None:
            from_cov = True
            self.S = np.asarray(S)
            self.P = np.linalg.inv(self.S)
            self.C = np.linalg.cholesky(self.S).T
            self.logdetP = -2.0 * np.sum(np.log(np.diagonal(self.C)))
        else:
            raise ValueError('Precision information missing.')
        if not from_cov:
            self.S = self.C.T @ self.C
        if m is not None:
            self.m = np.asarray(m)
            self.Pm = self.P @ self.m
        else:
            self.Pm = np.asarray(Pm)
            self.m = self.S @ self.Pm if from_cov else np.linalg.solve(self.P, self.Pm)
        self.ndim = self.m.size

    def gen(self, n_samples=1, method='random'):
        if method == 'random':
            z = np.random.randn(n_samples, self.ndim)
        elif method == 'halton':
            z = erfinv(2 * _halton(int(n_samples), self.ndim) - 1) * np.sqrt(2)
        else:
            raise ValueError('Unknown gen method ' + method)
        return z @ self.C + self.m

    def eval(self, x, ii=None, log=True):
        if ii is None:
            xm = x - self.m
            lp = -np.sum((xm @ self.P) * xm, axis=1)
            lp += self.logdetP - self.ndim * np.log(2.0 * np.pi)
            lp *= 0.5
        else:
            cov = self.S[ii][:, ii]
            jit = 1.e-5 * cov.mean() * np.diag(np.random.rand(cov.shape[0]))
            lp = scipy.stats.multivariate_normal.logpdf(x, self.m[ii], cov + jit)
            lp = np.array([lp]) if x.shape[0] == 1 else lp
        return lp if log else np.exp(lp)

    def _assign(self, other):
        for k in ('m', 'P', 'C', 'S', 'Pm', 'logdetP'):
            setattr(self, k, getattr(other, k))
        return other

    def __mul__(self, other):
        assert isinstance(other, Gaussian)
        return Gaussian(P=self.P + other.P, Pm=self.Pm + other.Pm)

    def __imul__(self, other):
        return self._assign(self * other)

    def __truediv__(self, other):
        """Quotient of Gaussians (may be improper)."""
        assert isinstance(other, Gaussian)
        return Gaussian(P=self.P - other.P, Pm=self.Pm - other.Pm)

    __div__ = __truediv__

    def __itruediv__(self, other):
        return self._assign(self / other)

    __idiv__ = __itruediv__

    def __pow__(self, power, modulo=None):
        return Gaussian(P=power * self.P, Pm=power * self.Pm)

    def __ipow__(self, power):
        return self._assign(self ** power)

    def kl(self, other):
        assert isinstance(other, Gaussian) and self.ndim == other.ndim
        dm = other.m - self.m
        return 0.5 * (np.sum(other.P * self.S) + dm @ other.P @ dm +
                      self.logdetP - other.logdetP - self.ndim)


class MoG:
    """Mixture of Gaussians (pdf.py:414-582)."""

    def __init__(self, a, ms=None, Ps=None, Us=None, Ss=None, xs=None, Ls=None):
        if ms is not None:
            if Ps is not None:
                self.xs = [Gaussian(m=m, P=p) for m, p in zip(ms, Ps)]
            elif Us is not None:
                self.xs = [Gaussian(m=m, U=u) for m, u in zip(ms, Us)]
            elif Ss is not None:
                self.xs = [Gaussian(m=m, S=s) for m, s in zip(ms, Ss)]
            elif Ls is not None:
                self.xs = [Gaussian(m=m, L=l) for m, l in zip(ms, Ls)]
            else:
                raise ValueError('Precision information missing.')
        elif xs is not None:
            self.xs = xs
        else:
            raise ValueError('Mean information missing.')
        self.a = np.asarray(a)
        self.ndim = self.xs[0].ndim
        self.n_components = self.ncomp = len(self.xs)

    @property
    def weights(self):
        return self.a

    @property
    def components(self):
        return self.xs

    def gen(self, n_samples=1, method='random'):
        which = discrete_sample(self.a, n_samples)
        counts = [int(np.sum(which == i)) for i in range(self.n_components)]
        return np.concatenate([g.gen(n_samples=c, method=method)
                               for g, c in zip(self.xs, counts)], axis=0)

    def eval(self, x, ii=None, log=True, debug=False):
        ps = np.array([g.eval(x, ii, log) for g in self.xs]).T
        res = logsumexp(ps + np.log(self.a), axis=1) if log else ps @ self.a
        if debug:
            print('weights\n', self.a, '\nps\n', ps, '\nres\n', res)
        return res

    def __str__(self):
        mus = np.array([g.m.tolist() for g in self.xs])
        diag_s = np.array([np.diagonal(g.S).tolist() for g in self.xs])
        return ('MoG:\nweights:\n' + str(self.a) + '\nmeans:\n' + str(mus) +
                '\ndiagS:\n' + str(diag_s))

    def _reweighted(self, ys, other, sign):
        """Mixing coefficients after multiplying (sign=+1) or dividing
        (sign=-1) every component x by ``other`` (y = x*other or x/other):
        log c = 1/2 [logdetP_x + s logdetP_o - logdetP_y
                     - x.m' P_x x.m - s o.m' P_o o.m + y.m' P_y y.m].
        This is the normaliser of the Gaussian product/quotient (as in the
        epsilon_free_inference original); the reference's transcription
        (pdf.py:506-511, 528-533) has the sign of the y term flipped, on a
        path its callers never reach (proposal=None, bayes_sim_main.py:154)."""
        logc = np.empty_like(self.a, dtype=float)
        for i, (x, y) in enumerate(zip(self.xs, ys)):
            v = x.logdetP + sign * other.logdetP - y.logdetP
            v -= x.m @ x.P @ x.m
            v -= sign * (other.m @ other.P @ other.m)
            v += y.m @ y.P @ y.m
            logc[i] = 0.5 * v
        la = np.log(self.a) + logc
        return np.exp(la - logsumexp(la))

    def __mul__(self, other):
        assert isinstance(other, Gaussian)
        ys = [x * other for x in self.xs]
        return MoG(a=self._reweighted(ys, other, +1.0), xs=ys)

    def __imul__(self, other):
        res = self * other
        self.a, self.xs = res.a, res.xs
        return res

    def __truediv__(self, other):
        assert isinstance(other, Gaussian)
        ys = [x / other for x in self.xs]
        return MoG(a=self._reweighted(ys, other, -1.0), xs=ys)

    __div__ = __truediv__

    def __itruediv__(self, other):
        res = self / other
        self.a, self.xs = res.a, res.xs
        return res

    __idiv__ = __itruediv__

    def calc_mean_and_cov(self):
        ms = np.array([g.m for g in self.xs])
        mean = self.a @ ms
        second = sum(w * (g.S + np.outer(g.m, g.m)) for w, g in zip(self.a, self.xs))
        return mean, second - np.outer(mean, mean)

    def project_to_gaussian(self):
        m, s = self.calc_mean_and_cov()
        return Gaussian(m=m, S=s)

    def prune_negligible_components(self, threshold):
        """Drop components lighter than ``threshold`` and spread their mass
        evenly over the survivors (pdf.py:562-570)."""
        drop = np.nonzero(self.a < threshold)[0]
        lost = np.sum(self.a[drop])
        self.n_components -= drop.size
        self.a = np.delete(self.a, drop) + lost / self.n_components
        self.xs = [g for i, g in enumerate(self.xs) if i not in drop]

    def kl(self, other, n_samples=10000):
        x = self.gen(n_samples)
        t = self.eval(x, log=True) - other.eval(x, log=True)
        return np.mean(t), np.std(t, ddof=1) / np.sqrt(n_samples)


def fit_mog(x, n_components, w=None, tol=1.0e-9, maxiter=float('inf'), verbose=False):
    """EM fit of a mixture to (optionally weighted) samples (pdf.py:585-642)."""
    x = x[:, np.newaxis] if x.ndim == 1 else x
    n, dim = x.shape
    a = np.ones(n_components) / n_components
    ms = np.random.randn(n_components, dim)
    covs = [np.eye(dim) for _ in range(n_components)]

    def joint(allow_singular):
        lj = np.array([scipy.stats.multivariate_normal.logpdf(
            x, ms[k], covs[k], allow_singular=allow_singular)
            for k in range(n_components)])
        lj = lj.reshape(n_components, -1) + np.log(a)[:, np.newaxis]
        lx = logsumexp(lj, axis=0)
        return lj, lx, (np.mean(lx) if w is None else np.dot(w, lx))

    lj, lx, prev = joint(False)
    it = 0
    while True:
        resp = np.exp(lj - lx)
        if w is not None:
            resp = resp * w
        mass = np.sum(resp, axis=1)
        a = mass / n if w is None else mass
        ms = (resp @ x) / mass[:, np.newaxis]
        for k in range(n_components):
            xm = x - ms[k]
            covs[k] = (xm.T * resp[k]) @ xm / mass[k]
        lj, lx, cur = joint(True)
        it += 1
        if verbose:
            print('Iteration = {0}, log likelihood = {1}, diff = {2}'.format(it, cur, cur - prev))
        if cur - prev < tol or it > maxiter:
            break
        prev = cur
    return MoG(a=a, ms=ms, Ss=covs)
