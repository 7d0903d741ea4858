"""ORACLE (test infrastructure only) — RFF / MDNN / MDRFF on torch-CPU.

Restates bayes_sim_ig/models/{rff,mdnn,mdrff}.py of the reference with the
same op sequence (per-component ``MultivariateNormal.log_prob`` loop, fp32
``result`` buffer, ``torch.optim.Adam``, numpy-RNG minibatch ids), so that
it is (a) the parity checker for the HIP path and (b) the "port" CPU
baseline timed by bench.py.  Pinned against tests/golden/mdn_*.npz and
tests/golden/chunk_*.npz (outputs of the reference itself in the build
container; torch 2.10 CPU vs the reference's pinned torch 1.8 — noted in
tests/golden/README.md).

``mdn_head_closed_form`` is a second, independent fp64 numpy evaluation of
the head forward / NLL / backward formulas used to check
the kernels tighter than fp32 autograd allows.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
from torch.distributions.multivariate_normal import MultivariateNormal

LL_LIMIT = 1.0e5       # mdnn.py:22
MIN_WEIGHT = 1.0e-5    # mdnn.py:23
EPS_NOISE = 1.0e-5     # mdnn.py:24


# --------------------------------------------------------------------- RFF
def draw_rff_freqs(kernel, m, d):
    """Non-quasi-random frequency draw from the GLOBAL numpy RNG,
    rff.py:111-120 (quasi_random=False branch) + rff.py:135-184.
    The quasi-random branch needs ``ghalton`` (absent, unpinned): PARITY
    UNPINNED for input_dim <= 100; all BASELINE GPU configs have
    input_dim > 100 and take this branch (mdrff.py:23)."""
    shape = (m, d)
    if kernel == 'RBF':
        return np.random.normal(0.0, 1.0, shape)
    nu = {'Laplace': 1, 'Matern12': 1, 'Matern32': 3, 'Matern52': 5}.get(kernel)
    if nu is None:
        raise ValueError("Kernel {} is not recognised.".format(kernel))
    g = np.random.normal(0, 1, shape)
    return g * np.sqrt(nu / np.random.chisquare(nu, shape))


class OracleRFF:
    """rff.py:53-132 (cos/sin and cos-only feature maps)."""

    def __init__(self, n_feat, d, sigma, cos_only=False, kernel='RBF',
                 freqs=None):
        self.n_feat, self.d = n_feat, int(d)
        if isinstance(sigma, (list, tuple, np.ndarray)):
            assert len(sigma) == d
            sig = np.asarray(sigma, dtype=np.float32)
        else:
            sig = np.ones(d, dtype=np.float32) * sigma
        self.sigma = torch.from_numpy(sig).float().reshape(1, -1)
        self.cos_only = cos_only
        self.offset = None
        if kernel not in ('RBF', 'Laplace', 'Matern12', 'Matern32', 'Matern52'):
            raise ValueError("Kernel {} is not recognised.".format(kernel))
        if cos_only:
            f = draw_rff_freqs(kernel, n_feat, d) if freqs is None else freqs
            self.offset = torch.from_numpy(
                2.0 * np.pi * np.random.rand(1, n_feat)).float()
            self.a = np.sqrt(1.0 / float(n_feat))
        else:
            assert n_feat % 2 == 0
            f = draw_rff_freqs(kernel, n_feat // 2, d) if freqs is None else freqs
            self.a = np.sqrt(1.0 / float(n_feat / 2))
        self.freqs = torch.from_numpy(np.asarray(f)).float()

    def to_features(self, x):
        inner = torch.matmul(x, (self.freqs / self.sigma).T)
        if self.cos_only:
            return self.a * torch.cos(inner + self.offset)
        return self.a * torch.cat([torch.cos(inner), torch.sin(inner)], dim=-1)


# -------------------------------------------------------------------- MDNN
class OracleMDNN(nn.Module):
    """mdnn.py:21-289.  Same submodule names -> same ``state_dict`` keys
    (net.fcon{l}, pi, mu, Diag.0, Lower) and same torch-RNG init order."""

    def __init__(self, input_dim, output_dim, output_lows, output_highs,
                 n_gaussians, full_covariance, hidden_layers, activation, lr,
                 device='cpu', eps_noise=EPS_NOISE, **kwargs):
        super().__init__()
        self.output_dim, self.n_gaussians = output_dim, n_gaussians
        self.lr, self.activation, self.device = lr, activation, device
        self.eps_noise = eps_noise
        self.output_lows = self.output_highs = None
        if output_lows is not None:
            self.output_lows = torch.from_numpy(np.asarray(output_lows)).float()
            self.output_highs = torch.from_numpy(np.asarray(output_highs)).float()
        layers, width = OrderedDict(), input_dim
        for i, h in enumerate(hidden_layers):
            layers['fcon%d' % i] = nn.Linear(width, h)
            layers['nl%d' % i] = activation()
            width = h
        self.net = nn.Sequential(layers) if len(hidden_layers) > 0 else None
        self.pi = nn.Linear(width, n_gaussians)
        self.mu = nn.Linear(width, output_dim * n_gaussians)
        self.Diag = nn.Sequential(nn.Linear(width, output_dim * n_gaussians))
        self.L_size = int(0.5 * output_dim * (output_dim - 1))
        self.Lower = None
        if self.L_size > 0 and full_covariance:
            self.Lower = nn.Linear(width, self.L_size * n_gaussians)

    # mdnn.py:89-125
    def forward(self, x, noise=None):
        h = self.net(x) if self.net is not None else x
        w = torch.softmax(self.pi(h), -1)
        w = torch.clamp(w, MIN_WEIGHT, 1.0)
        w = w / torch.sum(w, dim=1, keepdim=True)
        mu = self.mu(h).reshape(-1, self.output_dim, self.n_gaussians)
        l_d = torch.exp(self.Diag(h)).reshape(-1, self.output_dim, self.n_gaussians)
        eps = self.eps_noise * l_d.mean()              # not detached, :115
        u = torch.rand_like(l_d) if noise is None else noise
        l_d = l_d + u.detach() * eps
        low = None
        if self.Lower is not None:
            low = self.Lower(h).reshape(-1, self.L_size, self.n_gaussians)
        assert torch.isfinite(w).all()
        assert torch.isfinite(mu).all()
        assert torch.isfinite(l_d).all()
        if low is not None:
            assert torch.isfinite(low).all()
        return w, mu, l_d, low

    # mdnn.py:127-178
    def mdn_loss_fn(self, weights, mu, l_d, low, y):
        b = y.size()[0]
        result = torch.zeros(b, self.n_gaussians)       # fp32 always, :149
        rows, cols = np.tril_indices(self.output_dim, -1)
        for k in range(self.n_gaussians):
            tri = torch.diag_embed(l_d[:, :, k])
            if low is not None:
                tri[:, rows, cols] = low[:, :, k]
            lp = MultivariateNormal(loc=mu[:, :, k], scale_tril=tri).log_prob(y)
            lp = torch.clamp(lp, -LL_LIMIT, LL_LIMIT)
            wk = torch.clamp(weights[:, k], MIN_WEIGHT, 1.0)
            result[:, k] = lp + wk.log()
            assert torch.isfinite(lp).all()
            assert torch.isfinite(wk).all()
            assert torch.isfinite(result).all()
        return (-1.0 * torch.logsumexp(result, dim=1)).mean()

    def normalize_samples(self, params):                # mdnn.py:245-248
        return (params - self.output_lows) / (self.output_highs - self.output_lows)

    # mdnn.py:180-243
    def run_training(self, x_data, y_data, n_updates, batch_size,
                     test_frac=0.2, ids_table=None, noise_fn=None,
                     verbose=False):
        """``ids_table`` [n_updates, batch] overrides the numpy-RNG draw
        (teacher forcing); ``noise_fn(shape)`` overrides ``rand_like``."""
        assert x_data.shape[0] == y_data.shape[0]
        self.train()
        opt = torch.optim.Adam(self.parameters(), lr=self.lr)
        if self.output_lows is not None:
            y_data = self.normalize_samples(y_data)
        n_tot = x_data.shape[0]
        n_train = max(int(n_tot * (1.0 - test_frac)), 1)
        x_tr, y_tr = x_data[:n_train], y_data[:n_train]
        x_te, y_te = x_data[n_train:], y_data[n_train:]
        train_log, test_log = [], []
        every = max(n_updates // 5, 1)
        for it in range(n_updates):
            if ids_table is None:
                ids = np.random.randint(0, len(x_tr), batch_size)
            else:
                ids = ids_table[it]
            xb, yb = x_tr[ids], y_tr[ids]
            opt.zero_grad()
            nz = None if noise_fn is None else noise_fn(
                (xb.shape[0], self.output_dim, self.n_gaussians))
            loss = self.mdn_loss_fn(*self.forward(xb, noise=nz), yb)
            loss.backward()
            opt.step()
            if it % every == 0 or it + 1 == n_updates:
                nz = None if noise_fn is None else noise_fn(
                    (x_te.shape[0], self.output_dim, self.n_gaussians))
                te = self.mdn_loss_fn(*self.forward(x_te, noise=nz), y_te).item()
                train_log.append(loss.item())
                test_log.append(te)
                if verbose:
                    print(f'loss: train {loss.item():0.4f} test {te:0.4f}')
        return {'train_loss': train_log, 'test_loss': test_log}

    fit = run_training

    # mdnn.py:250-289 (with the full-covariance row index fixed: the
    # reference reads L[:, :, comp] for every point, mdnn.py:281, which only
    # works for one test point; the oracle uses L[pt, :, comp])
    def predict_mog_params(self, xs, noise=None):
        w, mu, l_d, low = self.forward(xs, noise=noise)
        span = self.output_highs - self.output_lows
        rows, cols = np.tril_indices(self.output_dim, -1)
        out = []
        for p in range(xs.shape[0]):
            ms, ls = [], []
            for k in range(self.n_gaussians):
                m = mu[p, :, k]
                tri = torch.diag_embed(l_d[p, :, k])
                if low is not None:
                    tri[rows, cols] = low[p, :, k]
                if self.output_lows is not None:
                    m = m * span + self.output_lows
                    tri = torch.diag(span) @ tri
                packed = torch.diag(tri)
                if low is not None:
                    packed = torch.cat([packed, tri[rows, cols]])
                ms.append(m.detach().numpy())
                ls.append(packed.detach().numpy())
            out.append((w[p].detach().numpy(), ms, ls))
        return out


class OracleMDRFF(OracleMDNN):
    """mdrff.py:14-30: MDNN heads on fixed cos/sin random features."""

    def __init__(self, input_dim, output_dim, output_lows, output_highs,
                 n_gaussians, lr, activation, full_covariance, device='cpu',
                 n_feat=500, kernel='RBF', sigma=1.0, freqs=None,
                 eps_noise=EPS_NOISE, **kwargs):
        super().__init__(n_feat, output_dim, output_lows, output_highs,
                         n_gaussians, hidden_layers=[], lr=lr,
                         activation=activation,
                         full_covariance=full_covariance, device=device,
                         eps_noise=eps_noise)
        if freqs is None and input_dim <= 100:
            raise NotImplementedError(
                'quasi-random (ghalton) frequencies: parity unpinned; pass freqs=')
        self.rff = OracleRFF(n_feat, input_dim, sigma, cos_only=False,
                             kernel=kernel, freqs=freqs)

    def forward(self, x, noise=None):
        return super().forward(self.rff.to_features(x), noise=noise)


# --------------------------------------------- closed-form fp64 head check
def mdn_head_closed_form(head_out, y, out_dim, n_comp, full_cov,
                         eps_noise=0.0, noise=None):
    """fp64 numpy evaluation of the MDN head on raw head outputs.

    head_out [B, Nh] = [logits K | mu D*K | pre_diag D*K | lower Ls*K] with
    index d*K+k inside each block (mdnn.py:109-119).  Returns
    (loss, d_head_out, dict(weights, mu, l_d, lower, lse)).
    Formulas: SURVEY Appendix A.1-A.3, derived from mdnn.py:108-178.
    """
    o = np.asarray(head_out, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    b, d, k = o.shape[0], out_dim, n_comp
    ls = d * (d - 1) // 2 if full_cov else 0
    logits = o[:, :k]
    mu = o[:, k:k + d * k].reshape(b, d, k)
    pre = o[:, k + d * k:k + 2 * d * k].reshape(b, d, k)
    low = o[:, k + 2 * d * k:k + 2 * d * k + ls * k].reshape(b, ls, k) if ls else None
    s = np.exp(logits - logits.max(axis=1, keepdims=True))
    s /= s.sum(axis=1, keepdims=True)
    c = np.clip(s, MIN_WEIGHT, 1.0)
    csum = c.sum(axis=1, keepdims=True)
    w = c / csum
    sig0 = np.exp(pre)
    eps = eps_noise * sig0.mean()
    u = np.zeros_like(sig0) if noise is None else np.asarray(noise, np.float64)
    sig = sig0 + u * eps
    rows, cols = np.tril_indices(d, -1)
    logp = np.zeros((b, k))
    g_mu_unit = np.zeros((b, d, k))      # d logp / d mu
    g_sig_unit = np.zeros((b, d, k))     # d logp / d sigma
    g_low_unit = np.zeros((b, ls, k)) if ls else None
    for bi in range(b):
        for ki in range(k):
            r = y[bi] - mu[bi, :, ki]
            if ls:
                t = np.diag(sig[bi, :, ki])
                t[rows, cols] = low[bi, :, ki]
                v = np.linalg.solve(t, r)
                q = np.linalg.solve(t.T, v)
                logp[bi, ki] = (-0.5 * v @ v - np.log(sig[bi, :, ki]).sum()
                                - 0.5 * d * math.log(2 * math.pi))
                g_mu_unit[bi, :, ki] = q
                g_sig_unit[bi, :, ki] = q * v - 1.0 / sig[bi, :, ki]
                g_low_unit[bi, :, ki] = np.outer(q, v)[rows, cols]
            else:
                z = r / sig[bi, :, ki]
                logp[bi, ki] = (-0.5 * z @ z - np.log(sig[bi, :, ki]).sum()
                                - 0.5 * d * math.log(2 * math.pi))
                g_mu_unit[bi, :, ki] = z / sig[bi, :, ki]
                g_sig_unit[bi, :, ki] = (z * z - 1.0) / sig[bi, :, ki]
    lp = np.clip(logp, -LL_LIMIT, LL_LIMIT)
    wc = np.clip(w, MIN_WEIGHT, 1.0)
    rr = lp + np.log(wc)
    mx = rr.max(axis=1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(rr - mx).sum(axis=1))
    loss = -lse.mean()
    gamma = np.exp(rr - lse[:, None])
    sc = -gamma / b
    g_lp = sc * ((logp >= -LL_LIMIT) & (logp <= LL_LIMIT))
    d_mu = g_lp[:, None, :] * g_mu_unit
    d_sig = g_lp[:, None, :] * g_sig_unit
    d_sig0 = d_sig + (eps_noise / sig0.size) * np.sum(u * d_sig)
    d_pre = d_sig0 * sig0
    g_w = sc / wc * ((w >= MIN_WEIGHT) & (w <= 1.0))
    g_c = (g_w - (g_w * w).sum(axis=1, keepdims=True)) / csum
    g_s = g_c * ((s >= MIN_WEIGHT) & (s <= 1.0))
    d_logits = s * (g_s - (g_s * s).sum(axis=1, keepdims=True))
    parts = [d_logits, d_mu.reshape(b, -1), d_pre.reshape(b, -1)]
    if ls:
        parts.append((g_lp[:, None, :] * g_low_unit).reshape(b, -1))
    aux = dict(weights=w, mu=mu, l_d=sig, lower=low, lse=lse, logp=logp)
    return loss, np.concatenate(parts, axis=1), aux
