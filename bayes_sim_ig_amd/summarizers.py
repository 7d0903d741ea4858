"""Trajectory summarizers on MI355X — host-side mirror of the reference's
bayes_sim_ig/utils/summarizers.py (same names, arguments and error
behaviour); the arithmetic runs in libbsig_hip (csrc/summarizers.hip).

Every function maps states[N,T,sd], actions[N,Ta,ad] (fp32) to a [N,F]
tensor on the inputs' device.  The rows live in HBM with a 16-byte aligned
pitch (ld = F rounded up to 4 floats) and the returned tensor is the [:, :F]
view, so the estimator's GEMM loaders can use 128-bit loads.
"""
import ctypes as C

import torch

from . import _lib

KIND_START, KIND_CORR, KIND_CORRDIFF, KIND_SIGNATURE = 0, 1, 2, 3


def _prep(states, actions):
    assert len(states.shape) == 3, 'Need states: ntraj x n_steps x state_dim'
    assert len(actions.shape) == 3, 'Need actions: ntraj x n_steps x state_dim'
    assert states.shape[0] == actions.shape[0]
    _lib.require_gpu()
    home = states.device
    dev = home if states.is_cuda else torch.device('cuda', torch.cuda.current_device())
    s = states.to(device=dev, dtype=torch.float32).contiguous()
    a = actions.to(device=dev, dtype=torch.float32).contiguous()
    return s, a, home


def _alloc(n, width, device, out):
    ld = _lib.round_up(width, 4)
    if out is not None:
        assert out.is_cuda and out.dtype == torch.float32 and out.dim() == 2
        assert out.shape[0] >= n and out.stride(1) == 1 and out.stride(0) >= width
        return out, out.stride(0)
    return torch.empty((n, ld), dtype=torch.float32, device=device), ld


def _finish(buf, n, width, home, out):
    res = buf[:n, :width]
    if out is None and home != buf.device:
        res = res.to(home)
    return res


def summary_dim(name, traj_len, obs_dim, act_dim, depth=0):
    """Width F of ``name``'s output without touching the GPU (what
    bayes_sim.py:57-60 obtains by pushing a zero trajectory through)."""
    kind = {'summary_start': 0, 'summary_waypts': 0, 'summary_corr': 1,
            'summary_corrdiff': 2, 'summary_signatory': 3}[name]
    return int(_lib.load().bsig_summary_dim(kind, traj_len, obs_dim, act_dim, depth))


def pad_states_actions(states, actions, tgt_actions_len=None):
    """Reference summarizers.py:20-62 (plain slicing / repeat; no arithmetic).
    Unlike the reference, padding works for any batch size: each trajectory
    repeats its own last step."""
    assert len(states.shape) == 3, 'Need states: ntraj x n_steps x state_dim'
    assert len(actions.shape) == 3, 'Need actions: ntraj x n_steps x state_dim'
    if tgt_actions_len is None:
        tgt_actions_len = states.shape[1]

    def fit(seq):
        if seq.shape[1] >= tgt_actions_len:
            return seq[:, :tgt_actions_len, :]
        tail = seq[:, -1:, :].expand(-1, tgt_actions_len - seq.shape[1], -1)
        return torch.cat([seq, tail], dim=1)
    states, actions = fit(states), fit(actions)
    assert states.shape[1] == actions.shape[1]
    return states, actions


def summary_start(states, actions, max_t=10, out=None):
    """Reference summarizers.py:65-70."""
    s, a, home = _prep(states, actions)
    n, sd, ad = s.shape[0], s.shape[2], a.shape[2]
    width = max_t * (sd + ad)
    buf, ld = _alloc(n, width, s.device, out)
    _lib.check(_lib.load().bsig_summary_start(
        _lib.ptr(s), _lib.ptr(a), _lib.ptr(buf), n, s.shape[1], a.shape[1],
        sd, ad, max_t, ld, _lib.stream()))
    return _finish(buf, n, width, home, out)


def summary_waypts(states, actions, n_waypts=10, out=None):
    """Reference summarizers.py:73-87: the crop to ``n_waypts`` precedes the
    stride computation, so the waypoints are the first ``n_waypts`` steps."""
    return summary_start(states, actions, max_t=n_waypts, out=out)


class CrossCorrFactors:
    """A cross-correlation summary that is not materialised.

    summarizers.py:106-119 builds, per trajectory, the outer product of S state features
    and A action features (+ mean and std of the state features): 47 KB per Ant row,
    420 KB per ShadowHand row, 98 % of it products of the same S + A numbers.  This handle
    keeps the factor rows ``[sf | af | mean | std | 1]`` (S + A + 3 floats per trajectory,
    include/bsig.h) and behaves like the ``[N, S*A + 2]`` summary wherever the estimator
    only needs its rows: ``MDNN.run_training`` hands the factor rows to the fit engine,
    whose first-layer tiles form ``sf[i] * af[j]`` on the fly -- the same single fp32
    multiply the summarizer does, so the first-layer inputs are bit-identical -- and no
    ``[N, S*A + 2]`` tensor, staging copy or per-update gather of 47 KB rows exists (only
    the held-out fifth of a chunk, read once per evaluation, is expanded into rows).
    ``materialize()`` (or any tensor use via ``torch.as_tensor`` semantics: indexing a
    column, ``.cpu()``, arithmetic) expands the rows with bsig_crosscorr_expand.
    """

    def __init__(self, factors, s_dim, a_dim):
        self.factors, self.s_dim, self.a_dim = factors, int(s_dim), int(a_dim)

    # --- the tensor surface BayesSim / MDNN.run_training touch
    @property
    def shape(self):
        return torch.Size((self.factors.shape[0], self.s_dim * self.a_dim + 2))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    @property
    def device(self):
        return self.factors.device

    @property
    def is_cuda(self):
        return self.factors.is_cuda

    @property
    def dtype(self):
        return torch.float32

    def __len__(self):
        return self.factors.shape[0]

    def __getitem__(self, rows):
        """Row selection keeps the factored form; anything else materialises."""
        if isinstance(rows, slice) or (torch.is_tensor(rows) and rows.dim() == 1) or \
                isinstance(rows, (list, range)):
            return CrossCorrFactors(self.factors[rows], self.s_dim, self.a_dim)
        return self.materialize()[rows]

    def materialize(self, out=None):
        """The ``[N, S*A + 2]`` summary rows (16-byte aligned pitch, like the summarizers)."""
        lib = _lib.require_gpu()
        n, width = self.shape
        fac = self.factors if self.factors.stride(1) == 1 else self.factors.contiguous()
        with _lib.on_device(fac.device):
            buf, ld = _alloc(n, width, fac.device, out)
            _lib.check(lib.bsig_crosscorr_expand(
                _lib.ptr(fac), fac.stride(0) if n > 1 else fac.shape[1], n, self.s_dim, self.a_dim,
                _lib.ptr(buf), ld, _lib.stream(fac.device)))
        return buf[:n, :width]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # any torch function applied to a handle sees the materialised tensor
        def expand(a):
            if isinstance(a, CrossCorrFactors):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(expand(v) for v in a)
            return a
        return func(*[expand(a) for a in args], **{k: expand(v) for k, v in (kwargs or {}).items()})

    def cpu(self):
        return self.materialize().cpu()

    def to(self, *args, **kwargs):
        return self.materialize().to(*args, **kwargs)


def cross_correlation(states, actions, use_state_diff=False, out=None,
                      check_finite=True, lazy=False):
    """Reference summarizers.py:90-122.  ``lazy=True`` returns a CrossCorrFactors handle
    instead of the materialised ``[N, S*A + 2]`` tensor."""
    if lazy:
        return _cross_correlation_factors(states, actions, use_state_diff, check_finite)
    s, a, home = _prep(states, actions)
    n, t, sd = s.shape
    ad = a.shape[2]
    assert t > 1                      # summarizers.py:94
    width = summary_dim('summary_corrdiff' if use_state_diff else 'summary_corr',
                        t, sd, ad)
    buf, ld = _alloc(n, width, s.device, out)
    # check_finite may be a 1-element int32 device tensor: the kernel raises its flag there
    # and the caller asserts later (BayesSim.fit: one read-back for all chunks)
    deferred = torch.is_tensor(check_finite)
    flag = check_finite if deferred else (
        torch.zeros(1, dtype=torch.int32, device=s.device) if check_finite else None)
    _lib.check(_lib.load().bsig_crosscorr(
        _lib.ptr(s), _lib.ptr(a), _lib.ptr(buf), n, t, a.shape[1], sd, ad,
        1 if use_state_diff else 0, ld, _lib.ptr(flag), _lib.stream()))
    if check_finite is not None and check_finite is not False and not deferred:
        assert int(flag.item()) == 0  # summarizers.py:120
    return _finish(buf, n, width, home, out)


def _cross_correlation_factors(states, actions, use_state_diff, check_finite):
    s, a, home = _prep(states, actions)
    n, t, sd = s.shape
    ad = a.shape[2]
    assert t > 1                      # summarizers.py:94
    lib = _lib.load()
    s_dim, a_dim = C.c_int32(), C.c_int32()
    _lib.check(lib.bsig_crosscorr_factor_dims(t, sd, ad, C.byref(s_dim), C.byref(a_dim)))
    ld = _lib.round_up(s_dim.value + a_dim.value + 3, 4)
    fac = torch.empty((n, ld), dtype=torch.float32, device=s.device)
    deferred = torch.is_tensor(check_finite)
    flag = check_finite if deferred else (
        torch.zeros(1, dtype=torch.int32, device=s.device) if check_finite else None)
    with _lib.on_device(s.device):
        _lib.check(lib.bsig_crosscorr_factors(
            _lib.ptr(s), _lib.ptr(a), _lib.ptr(fac), n, t, a.shape[1], sd, ad,
            1 if use_state_diff else 0, ld, _lib.ptr(flag), _lib.stream(s.device)))
    if check_finite is not None and check_finite is not False and not deferred:
        assert int(flag.item()) == 0  # summarizers.py:120
    return CrossCorrFactors(fac, s_dim.value, a_dim.value)


def summary_corrdiff(states, actions, out=None, check_finite=True, lazy=False):
    return cross_correlation(states, actions, use_state_diff=True, out=out,
                             check_finite=check_finite, lazy=lazy)


def summary_corr(states, actions, out=None, check_finite=True, lazy=False):
    return cross_correlation(states, actions, use_state_diff=False, out=out,
                             check_finite=check_finite, lazy=lazy)


def signature_depth(ndim):
    """Reference summarizers.py:133-141."""
    max_output_dim = 110 ** 2
    for depth in reversed(range(4)):
        if ndim ** depth <= max_output_dim:
            return depth
    return 1


def summary_signatory(states, actions, depth=None, out=None):
    """Reference summarizers.py:144-168 with the signature computed by
    csrc/summarizers.hip instead of ``signatory``.  All N rows are returned
    (the reference drops N % 10 rows when N > 10000)."""
    assert len(states.shape) == 3, 'states should be batch x time x state_dim'
    s, a, home = _prep(states, actions)
    n, length, sd = s.shape
    ad = a.shape[2]
    assert a.shape[1] == length
    if depth is None:
        depth = signature_depth(1 + sd + ad)
    width = summary_dim('summary_signatory', length, sd, ad, depth)
    buf, ld = _alloc(n, width, s.device, out)
    _lib.check(_lib.load().bsig_signature(
        _lib.ptr(s), _lib.ptr(a), _lib.ptr(buf), n, length, sd, ad, depth, ld,
        _lib.stream()))
    return _finish(buf, n, width, home, out)
